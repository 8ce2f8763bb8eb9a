import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import bench
for s16 in (False, True, True):
    r = bench.host_pipeline_workload(64, 64, s16=s16)
    print(s16, r.get("value"), r.get("ms_per_step"), r.get("serial_calls_ms_per_step"), r.get("h2d_copy_alone_ms"), r.get("error"))
