#!/bin/bash
# Run on the GPU box: the pipelined step against dynamic-LDS pads on single launches (AT3HIP_PAD_* of an AT3HIP_DEBUG_KNOBS
# build): which kernels gain from keeping other kernels' workgroups off their CUs. usage: tools/pad_sweep.sh lib_dbg.so "VAR=bytes [VAR=bytes]" ...
L=$1; shift
for rep in 1 2; do
for V in "" "$@"; do
  env $V AT3HIP_LIB=$PWD/$L python bench.py --no-side-workloads --no-cpu-baseline --no-parity --regions 4 --gain-wgs 6 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%-44s %10.0f frames/s  %.4f ms  (min %.4f)' % ('$V'[-44:], d['value'], d['ms_per_step'], d['timing']['ms_per_step_min']))"
done; done
