#!/bin/bash
# Copies the summaries tools/r06_final_profiles.sh left under gpurun_out/ into profiles/ (tracked).
cd "$(dirname "$0")/../gpurun_out" || exit 1
cp prof_r06/summary_isolated.txt ../profiles/r06_isolated_rocprof_summary.txt
cp prof_r06/summary.txt ../profiles/r06_pipelined_rocprof_summary.txt
cp prof_r06/k1_traffic.json ../profiles/k1_traffic.json
cp prof_r06/pipeline_traffic.json ../profiles/pipeline_traffic.json
cp prof_r06_nogain/summary_isolated.txt ../profiles/r06_nogain_isolated_rocprof_summary.txt
cp prof_r06_nogain/summary.txt ../profiles/r06_nogain_pipelined_rocprof_summary.txt
cp prof_r06_nogain/k1_traffic.json ../profiles/k1_traffic_nogain.json
cp prof_r06_burst/summary_isolated.txt ../profiles/r06_input_burst_isolated_rocprof_summary.txt
cp prof_r06_tones/summary_isolated.txt ../profiles/r06_input_tones_isolated_rocprof_summary.txt
cp prof_r06lp4_noise/summary_isolated.txt ../profiles/r06_lp4_isolated_rocprof_summary.txt
cp prof_r06_shard/summary_isolated.txt ../profiles/r06_shard_isolated_rocprof_summary.txt
cp prof_r06_nogain_shard/summary_isolated.txt ../profiles/r06_nogain_shard_isolated_rocprof_summary.txt
cp r06_timeline.txt ../profiles/r06_pipelined_timeline.txt
grep -v amdgpu r06_alloc_phase_cycles.txt > ../profiles/r06_alloc_phase_cycles.txt
cp at1_summary.txt ../profiles/r06_at1_rocprof_summary.txt
cp at3p_summary.txt ../profiles/r06_at3p_rocprof_summary.txt
cp pmc_r06_iso/summary.txt ../profiles/r06_isolated_pmc_summary.txt
