#!/bin/bash
# Run on the GPU box: the pipelined step of atracdenc_amd/libat3hip_prev.so (a build of an older commit, made by hand) and
# of atracdenc_amd/libat3hip.so, alternating on the same box. usage: tools/ab_bench.sh [bench.py flags]
for i in 1 2 3; do
for L in prev cur; do
  if [ $L = prev ]; then export AT3HIP_LIB=$PWD/atracdenc_amd/libat3hip_prev.so; else export AT3HIP_LIB=$PWD/atracdenc_amd/libat3hip.so; fi
  python bench.py --no-side-workloads --no-cpu-baseline --steps 30 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['value'], d['ms_per_step'])"
done; done
