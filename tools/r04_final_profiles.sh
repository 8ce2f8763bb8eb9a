#!/bin/bash
# Round-4 closing measurement on the GPU box: headline + no-gain (fused kernel) profiles with PMC passes, the per-input, LP4 and
# shard summaries, k_alloc_pack's phase cycles. Summaries are copied to profiles/ by hand afterwards.
export TMPDIR=/tmp
bash tools/profile_gpu.sh r04 > gpurun_out/r04_profile.log 2>&1
BENCH_ARGS=--no-gain bash tools/profile_gpu.sh r04_nogain > gpurun_out/r04_profile_nogain.log 2>&1
bash tools/profile_inputs.sh r04 burst tones > gpurun_out/r04_profile_inputs.log 2>&1
EXTRA="--bitrate 66150" bash tools/profile_inputs.sh r04lp4 noise > gpurun_out/r04_profile_lp4.log 2>&1
bash tools/profile_shard.sh r04 > gpurun_out/r04_profile_shard.log 2>&1
bash tools/profile_shard.sh r04_nogain --no-gain > gpurun_out/r04_profile_shard_nogain.log 2>&1
bash tools/timeline.sh > gpurun_out/r04_timeline.txt 2>&1
bash tools/alloc_phase_cycles.sh > gpurun_out/r04_alloc_phase_cycles.txt 2>&1
bash tools/alloc_phase_cycles.sh --input burst >> gpurun_out/r04_alloc_phase_cycles.txt 2>&1
bash tools/alloc_phase_cycles.sh --bitrate 66150 >> gpurun_out/r04_alloc_phase_cycles.txt 2>&1
ls gpurun_out | head -40
