#!/bin/bash
# Run on the GPU box: the ATRAC1 and ATRAC3plus benches (tools/at1_bench.py, tools/at3p_bench.py) for each library given, alternating.
# usage: tools/ab_wide.sh libA.so libB.so
for rep in 1 2 3; do for L in "$@"; do
AT3HIP_LIB=$PWD/$L python tools/at1_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('at1  %-34s' % '$L'[-34:], d.get('value'), d.get('ms_per_step'), d.get('device_ms'))"
AT3HIP_LIB=$PWD/$L python tools/at3p_bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('at3p %-34s' % '$L'[-34:], d.get('value'), d.get('ms_per_step'), d.get('device_ms'))"
done; done
