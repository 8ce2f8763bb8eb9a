#!/bin/bash
# Round-5 closing measurement on the GPU box: headline + no-gain (fused kernel) profiles with PMC passes, the per-input, LP4 and
# shard summaries, k_alloc_pack's phase cycles. Summaries are copied to profiles/ by hand afterwards.
export TMPDIR=/tmp
bash tools/profile_gpu.sh r05 > gpurun_out/r05_profile.log 2>&1
BENCH_ARGS=--no-gain bash tools/profile_gpu.sh r05_nogain > gpurun_out/r05_profile_nogain.log 2>&1
bash tools/profile_inputs.sh r05 burst tones > gpurun_out/r05_profile_inputs.log 2>&1
EXTRA="--bitrate 66150" bash tools/profile_inputs.sh r05lp4 noise > gpurun_out/r05_profile_lp4.log 2>&1
bash tools/profile_shard.sh r05 > gpurun_out/r05_profile_shard.log 2>&1
bash tools/profile_shard.sh r05_nogain --no-gain > gpurun_out/r05_profile_shard_nogain.log 2>&1
bash tools/timeline.sh > gpurun_out/r05_timeline.txt 2>&1
bash tools/alloc_phase_cycles.sh > gpurun_out/r05_alloc_phase_cycles.txt 2>&1
bash tools/alloc_phase_cycles.sh --input burst >> gpurun_out/r05_alloc_phase_cycles.txt 2>&1
bash tools/alloc_phase_cycles.sh --bitrate 66150 >> gpurun_out/r05_alloc_phase_cycles.txt 2>&1
ls gpurun_out | head -40
bash tools/profile_at1.sh > /dev/null 2>&1
bash tools/profile_at3p.sh > /dev/null 2>&1
bash tools/pmc_kernel.sh r05_iso --sync-steps --no-side-workloads --regions 0 --no-parity > /dev/null 2>&1
