#!/bin/bash
# Copies the summaries tools/r05_final_profiles.sh left under gpurun_out/ into profiles/ (tracked).
cd "$(dirname "$0")/../gpurun_out" || exit 1
cp prof_r05/summary_isolated.txt ../profiles/r05_isolated_rocprof_summary.txt
cp prof_r05/summary.txt ../profiles/r05_pipelined_rocprof_summary.txt
cp prof_r05/k1_traffic.json ../profiles/k1_traffic.json
cp prof_r05/pipeline_traffic.json ../profiles/pipeline_traffic.json
cp prof_r05_nogain/summary_isolated.txt ../profiles/r05_nogain_isolated_rocprof_summary.txt
cp prof_r05_nogain/summary.txt ../profiles/r05_nogain_pipelined_rocprof_summary.txt
cp prof_r05_nogain/k1_traffic.json ../profiles/k1_traffic_nogain.json
cp prof_r05_burst/summary_isolated.txt ../profiles/r05_input_burst_isolated_rocprof_summary.txt
cp prof_r05_tones/summary_isolated.txt ../profiles/r05_input_tones_isolated_rocprof_summary.txt
cp prof_r05lp4_noise/summary_isolated.txt ../profiles/r05_lp4_isolated_rocprof_summary.txt
cp prof_r05_shard/summary_isolated.txt ../profiles/r05_shard_isolated_rocprof_summary.txt
cp prof_r05_nogain_shard/summary_isolated.txt ../profiles/r05_nogain_shard_isolated_rocprof_summary.txt
cp r05_timeline.txt ../profiles/r05_pipelined_timeline.txt
grep -v amdgpu r05_alloc_phase_cycles.txt > ../profiles/r05_alloc_phase_cycles.txt
cp at1_summary.txt ../profiles/r05_at1_rocprof_summary.txt
cp at3p_summary.txt ../profiles/r05_at3p_rocprof_summary.txt
cp pmc_r05_iso/summary.txt ../profiles/r05_isolated_pmc_summary.txt
