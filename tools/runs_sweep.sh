#!/bin/bash
# Run on the GPU box: isolated (synchronous) stage timings of the front-end kernels for different run counts.
# usage: tools/runs_sweep.sh "<runs values>" [bench args]
VALS=$1; shift
for R in $VALS; do
  python bench.py --runs $R --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads --sync-steps "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
st=d['stage_ms_per_step']
print('runs=$R', 'qmf_ms=%.4f'%st['qmf_ms'], 'qmf_mdct_ms=%.4f'%st['qmf_mdct_ms'], 'iso_k1=%.4f'%d['roofline']['isolated']['avg_launch_ms'], 'step=%.4f'%d['ms_per_step'])
"
done
