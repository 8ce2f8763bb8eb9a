#!/bin/bash
# Run on the GPU box: the eight-rank bench on one device (tests' test_eight_context_readiness[ranks]) N times; prints what a failing run's contexts reported.
N=${1:-10}; shift
for i in $(seq 1 $N); do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 8 --device-map 0,0,0,0,0,0,0,0 --steps 3 --warmup 1 --streams 64 --frames 16 --regions 1 --region-ms 1 --no-side-workloads --no-cpu-baseline "$@" 2>gpurun_out/ranks8_err_$i.txt | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('run $i parity', d['parity_in_run'])
if not d['parity_in_run']:
    pc=d.get('parity_check'); print(json.dumps(pc)[:1500]); print(''.join(l for l in open('gpurun_out/ranks8_err_$i.txt') if 'replay mismatch' in l)[-1500:])
"
done
