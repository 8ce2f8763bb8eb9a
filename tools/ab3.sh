#!/bin/bash
# Run on the GPU box: three libraries alternating: isolated k_alloc_pack and the pipelined step (noise LP2, LP4), phase cycles of the current tree
export TMPDIR=/tmp
F='s/k_gain[a-z_]*=[0-9.]* //g; s/k_state[a-z_]*=[0-9.]* //g; s/k_loud[a-z_]*=[0-9.]* //g; s/k_mdct_sub<[a-z]*>=[0-9.]* //g; s/k_qmf_sub8=[0-9.]* //g'
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
echo "== isolated, noise LP2"; bash tools/ab_kernels.sh "$@" 2>&1 | sed "$F"
echo "== isolated, LP4"; bash tools/ab_kernels.sh "$@" -- --bitrate 66150 2>&1 | sed "$F"
echo "== step, noise LP2"; bash tools/ab_step.sh "$@"
A=(); for L in "$@"; do A+=("$L|--bitrate 66150"); done
echo "== step, LP4"; bash tools/ab_step.sh "${A[@]}"
A=(); for L in "$@"; do A+=("$L|--input tones"); done
echo "== step, tones"; bash tools/ab_step.sh "${A[@]}"
bash tools/alloc_phase_cycles.sh 2>&1 | grep -v amdgpu.ids
