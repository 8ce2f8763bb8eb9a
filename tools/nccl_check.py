import os, sys, torch
sys.path.insert(0, os.getcwd())
from atracdenc_amd import dist as d
rank, lr, world = d.env_world()
torch.cuda.set_device(lr)
dist = d.init("nccl", lr)
dist.barrier()
print("max", d.max_over_ranks(1.25 + rank, dist, device="cuda"), "objs", d.gather_objects({"r": rank}, dist))
dist.destroy_process_group()
