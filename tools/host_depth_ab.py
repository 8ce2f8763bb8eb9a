import sys, json
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
for rep in range(2):
    for s16 in (False, True):
        for D in (2, 3, 4):
            r = bench.host_pipeline_workload(64, 64, steps=300, s16=s16, depth=D)
            print(rep, "s16" if s16 else "f32", "depth", D, r.get("value"), r.get("ms_per_step"), r.get("frac_of_pcie_bound"), r.get("h2d_copy_alone_ms"), r.get("error"), flush=True)
