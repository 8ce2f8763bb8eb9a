#!/bin/bash
# Run on the GPU box: the pipelined step (bench.py, 4 extra regions) for several (library, bench flags) variants, alternating,
# on the same box. usage: tools/ab_step.sh "lib.so|flags" "lib2.so|flags" ...
for rep in 1 2 3; do
for V in "$@"; do
  L=${V%%|*}; F=${V#*|}; [ "$F" = "$V" ] && F=""
  AT3HIP_LIB=$PWD/$L python bench.py --no-side-workloads --no-cpu-baseline --no-parity --regions 4 $F 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %10.0f frames/s  %.4f ms  (min %.4f)' % ('$V'[-44:], d['value'], d['ms_per_step'], d['timing']['ms_per_step_min']))"
done; done
