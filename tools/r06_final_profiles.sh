#!/bin/bash
# Round-5 closing measurement on the GPU box: headline + no-gain (fused kernel) profiles with PMC passes, the per-input, LP4 and
# shard summaries, k_alloc_pack's phase cycles. Summaries are copied to profiles/ by hand afterwards.
export TMPDIR=/tmp
bash tools/profile_gpu.sh r06 > gpurun_out/r06_profile.log 2>&1
BENCH_ARGS=--no-gain bash tools/profile_gpu.sh r06_nogain > gpurun_out/r06_profile_nogain.log 2>&1
bash tools/profile_inputs.sh r06 burst tones > gpurun_out/r06_profile_inputs.log 2>&1
EXTRA="--bitrate 66150" bash tools/profile_inputs.sh r06lp4 noise > gpurun_out/r06_profile_lp4.log 2>&1
bash tools/profile_shard.sh r06 > gpurun_out/r06_profile_shard.log 2>&1
bash tools/profile_shard.sh r06_nogain --no-gain > gpurun_out/r06_profile_shard_nogain.log 2>&1
bash tools/timeline.sh > gpurun_out/r06_timeline.txt 2>&1
bash tools/alloc_phase_cycles.sh > gpurun_out/r06_alloc_phase_cycles.txt 2>&1
bash tools/alloc_phase_cycles.sh --input burst >> gpurun_out/r06_alloc_phase_cycles.txt 2>&1
bash tools/alloc_phase_cycles.sh --bitrate 66150 >> gpurun_out/r06_alloc_phase_cycles.txt 2>&1
ls gpurun_out | head -40
bash tools/profile_at1.sh > /dev/null 2>&1
bash tools/profile_at3p.sh > /dev/null 2>&1
bash tools/pmc_kernel.sh r06_iso --sync-steps --no-side-workloads --regions 0 --no-parity > /dev/null 2>&1
# round 6: the fused kernel alone (lab), its traffic, its power / clock behaviour, the untraced stage timeline
bash tools/k1/rocprof_lab.sh atracdenc_amd/libat3hip.so > gpurun_out/r06_k1_rocprof_lab.txt 2>&1
bash tools/k1/traffic.sh 64x64 atracdenc_amd/libat3hip.so > gpurun_out/r06_k1_traffic.txt 2>&1
bash tools/k1/traffic.sh 1024x128 atracdenc_amd/libat3hip.so >> gpurun_out/r06_k1_traffic.txt 2>&1
python tools/k1/power_probe.py atracdenc_amd/libat3hip.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_k1_power_probe.txt
python tools/k1/lab.py --reps 1 --chain 1,2 build_ab/lib_stamps.so 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_k1_phase_cycles.txt
AT3HIP_LIB=$PWD/build_ab/lib_ev.so python tools/event_timeline.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_event_timeline.txt
AT3HIP_LIB=$PWD/build_ab/lib_ev.so python tools/event_timeline.py --input tones 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_event_timeline.txt
AT3HIP_LIB=$PWD/build_ab/lib_ev.so python tools/event_timeline.py --input tones --prior 2 2>&1 | grep -v amdgpu.ids >> gpurun_out/r06_event_timeline.txt
