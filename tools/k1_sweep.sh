#!/bin/bash
# usage (on the GPU box): tools/k1_sweep.sh  -> whole-job frames/s, roofline.frac and the isolated K1 launch time for AT3HIP_FRAMES_PER_WG = auto, 3, 4, 5, 6, 8
for f in 0 3 4 5 6 8; do
  echo -n "fpw=$f "
  AT3HIP_FRAMES_PER_WG=$f python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['roofline']['frac'], d['roofline']['isolated']['avg_launch_ms'], d['roofline']['isolated']['frac'])"
done
