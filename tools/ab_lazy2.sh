#!/bin/bash
# Run on the GPU box: parity slice, base / new alternating (isolated k_alloc_pack + step) for noise LP2 / LP4 / burst / tones, phase cycles of the new build
export TMPDIR=/tmp
A=${1:-atracdenc_amd/lib_base.so}; B=${2:-atracdenc_amd/lib_new.so}
O=gpurun_out/ab_lazy; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
F='s/k_gain[a-z_]*=[0-9.]* //g; s/k_state[a-z_]*=[0-9.]* //g; s/k_loud[a-z_]*=[0-9.]* //g; s/k_mdct_sub<[a-z]*>=[0-9.]* //g; s/k_qmf_sub8=[0-9.]* //g'
echo "== isolated kernels, noise LP2"; bash tools/ab_kernels.sh $A $B 2>&1 | sed "$F"
echo "== isolated kernels, noise LP4"; bash tools/ab_kernels.sh $A $B -- --bitrate 66150 2>&1 | sed "$F"
echo "== isolated kernels, burst"; bash tools/ab_kernels.sh $A $B -- --input burst 2>&1 | sed "$F"
echo "== isolated kernels, tones"; bash tools/ab_kernels.sh $A $B -- --input tones 2>&1 | sed "$F"
echo "== step, noise LP2"; bash tools/ab_step.sh "$A" "$B"
echo "== step, LP4"; bash tools/ab_step.sh "$A|--bitrate 66150" "$B|--bitrate 66150"
echo "== step, burst"; bash tools/ab_step.sh "$A|--input burst" "$B|--input burst"
echo "== step, tones"; bash tools/ab_step.sh "$A|--input tones" "$B|--input tones"
bash tools/alloc_phase_cycles.sh 2>&1 | grep -v amdgpu.ids
bash tools/alloc_phase_cycles.sh --input tones 2>&1 | grep -v amdgpu.ids
