# Run on the GPU box: the pipelined step with the stage-timing events on every step, every 8th, never (AT3HIP_OPT_TIMING_EVERY), alternating.
python -m pytest tests -x -q -m gpu -k "timing_every or option_values or fused or pipelined" 2>&1 | tail -3
for rep in 1 2 3; do for TE in 1 8 0; do
python bench.py --no-side-workloads --no-cpu-baseline --no-parity --regions 4 --timing-every $TE 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('every=$TE %10.0f frames/s  %.4f ms  (min %.4f) k1 %.4f ms n=%s' % (d['value'], d['ms_per_step'], d['timing']['ms_per_step_min'], d['roofline']['avg_launch_ms'], d['roofline'].get('launches_timed')))"
done; done
