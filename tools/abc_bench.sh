#!/bin/bash
# Run on the GPU box: the pipelined step of up to three builds (atracdenc_amd/libat3hip.so, _prev.so, _eq.so), alternating.
# usage: tools/abc_bench.sh [bench.py flags]
for i in 1 2; do
for L in libat3hip.so libat3hip_prev.so libat3hip_eq.so; do
  AT3HIP_LIB=$PWD/atracdenc_amd/$L python bench.py --no-side-workloads --no-cpu-baseline --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
done; done
