#!/bin/bash
# Run on the GPU box: parity slice of the current build, then base / new libraries alternating: isolated kernel times and the
# pipelined step for noise, LP4 and burst.  usage: tools/ab_lazy.sh libA.so libB.so
export TMPDIR=/tmp
A=${1:-atracdenc_amd/lib_base.so}; B=${2:-atracdenc_amd/lib_new.so}
O=gpurun_out/ab_lazy; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
echo "== isolated kernels, noise LP2"; bash tools/ab_kernels.sh $A $B 2>&1 | sed 's/k_gain[a-z_]*=[0-9.]* //g; s/k_state[a-z_]*=[0-9.]* //g'
echo "== isolated kernels, noise LP4"; bash tools/ab_kernels.sh $A $B -- --bitrate 66150 2>&1 | sed 's/k_gain[a-z_]*=[0-9.]* //g; s/k_state[a-z_]*=[0-9.]* //g'
echo "== isolated kernels, burst"; bash tools/ab_kernels.sh $A $B -- --input burst 2>&1 | sed 's/k_gain[a-z_]*=[0-9.]* //g; s/k_state[a-z_]*=[0-9.]* //g'
echo "== step, noise LP2"; bash tools/ab_step.sh "$A" "$B"
echo "== step, LP4"; bash tools/ab_step.sh "$A|--bitrate 66150" "$B|--bitrate 66150"
echo "== step, burst"; bash tools/ab_step.sh "$A|--input burst" "$B|--input burst"
echo "== step, tones"; bash tools/ab_step.sh "$A|--input tones" "$B|--input tones"
