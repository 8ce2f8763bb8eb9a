"""N>1 path on CPU: two processes over gloo shard the streams, encode their shard (the CPU oracle stands in for
the GPU encoder - the sharding / barrier / MAX-time / gather plumbing is what is under test) and together
reproduce the single-process result exactly."""
import hashlib
import os
import subprocess
import sys

import numpy as np

from at3_testlib import LP2, SIGNALS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import hashlib, os, sys, time
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
from at3_testlib import LP2, SIGNALS, oracle
from atracdenc_amd import dist as D
rank, local_rank, world = D.env_world()
dist = D.init("gloo")
n_total, nb = 5, 6
first, count = D.shard_streams(n_total, world, rank)
names = sorted(SIGNALS)
dist.barrier()
t0 = time.perf_counter()
sums = {}
for i in range(first, first + count):
    frames, _ = oracle().encode(SIGNALS[names[i]](nb), LP2)
    sums[i] = hashlib.md5(frames.tobytes()).hexdigest()
dist.barrier()
elapsed = D.max_over_ranks(time.perf_counter() - t0 + 0.01 * rank, dist)
parts = D.gather_objects((rank, first, count, sums, elapsed), dist)
if rank == 0:
    import json
    print("RESULT " + json.dumps(parts))
dist.destroy_process_group()
'''


def test_shard_streams_partition():
    from atracdenc_amd.dist import shard_streams
    for total in (1, 7, 64, 8192):
        for world in (1, 2, 3, 8):
            parts = [shard_streams(total, world, r) for r in range(world)]
            assert parts[0][0] == 0
            assert sum(c for _, c in parts) == total
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_two_process_gloo_matches_single(oracle, tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    parts = json.loads(line[len("RESULT "):])
    assert [p[0] for p in parts] == [0, 1]
    merged = {}
    for _, first, count, sums, elapsed in parts:
        assert len(sums) == count
        merged.update({int(k): v for k, v in sums.items()})
        assert abs(elapsed - parts[0][4]) < 1e-9          # every rank holds the same MAX
    names = sorted(SIGNALS)
    assert sorted(merged) == list(range(5))
    for i in range(5):
        frames, _ = oracle.encode(SIGNALS[names[i]](6), LP2)
        assert merged[i] == hashlib.md5(frames.tobytes()).hexdigest()
