#!/bin/bash
# Copies the summaries tools/r04_final_profiles.sh left under gpurun_out/ into profiles/ (tracked).
cd "$(dirname "$0")/../gpurun_out" || exit 1
cp prof_r04/summary_isolated.txt ../profiles/r04_isolated_rocprof_summary.txt
cp prof_r04/summary.txt ../profiles/r04_pipelined_rocprof_summary.txt
cp prof_r04/k1_traffic.json ../profiles/k1_traffic.json
cp prof_r04/pipeline_traffic.json ../profiles/pipeline_traffic.json
cp prof_r04_nogain/summary_isolated.txt ../profiles/r04_nogain_isolated_rocprof_summary.txt
cp prof_r04_nogain/summary.txt ../profiles/r04_nogain_pipelined_rocprof_summary.txt
cp prof_r04_nogain/k1_traffic.json ../profiles/k1_traffic_nogain.json
cp prof_r04_burst/summary_isolated.txt ../profiles/r04_input_burst_isolated_rocprof_summary.txt
cp prof_r04_tones/summary_isolated.txt ../profiles/r04_input_tones_isolated_rocprof_summary.txt
cp prof_r04lp4_noise/summary_isolated.txt ../profiles/r04_lp4_isolated_rocprof_summary.txt
cp prof_r04_shard/summary_isolated.txt ../profiles/r04_shard_isolated_rocprof_summary.txt
cp prof_r04_nogain_shard/summary_isolated.txt ../profiles/r04_nogain_shard_isolated_rocprof_summary.txt
cp r04_timeline.txt ../profiles/r04_pipelined_timeline.txt
grep -v amdgpu r04_alloc_phase_cycles.txt > ../profiles/r04_alloc_phase_cycles.txt
