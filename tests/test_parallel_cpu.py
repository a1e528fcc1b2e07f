"""CPU, world_size 2 over gloo: the N>1 sharding logic (range split, all-gather of Jacobian partials,
local EC sum).  The oracle stands in for the device here (test double); on the GPU box the same
code runs with GpuBackend (tests/test_parallel_gpu.py)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from oracle import bn254 as B, cbind as OC
from sha2_on_cq_halo2_amd.parallel import shard_range, sharded_multiexp, column_owner
from tests.util import random_points, random_scalars

class OracleBackend:
    def best_multiexp(self, c, b): return OC.best_multiexp(c, b) if len(c) else np.zeros(12, dtype=np.uint64)
    def g1_sum(self, pts):
        acc = np.zeros(12, dtype=np.uint64)
        for p in pts:
            out = np.zeros(12, dtype=np.uint64)
            OC.lib().cqo_g1_add(acc.ctypes.data, np.ascontiguousarray(p).ctypes.data, out.ctypes.data)
            acc = out
        return acc

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
for n in (1, 2, 7, 200):
    sc = B.to_mont_limbs(random_scalars(n, 5 + n)); pts = B.points_to_mont_limbs(random_points(n, 6 + n))
    lo, hi = shard_range(n, rank, world)
    got = sharded_multiexp(OracleBackend(), sc[lo:hi], pts[lo:hi])
    exp = OC.best_multiexp(sc, pts)
    assert np.array_equal(OC.g1_to_affine(got), OC.g1_to_affine(exp)), (n, rank)
# weak-scaling timing reduction used by bench.py: max over ranks
t = torch.tensor([1.0 + rank], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == float(world)
assert [column_owner(c, 8, world) for c in range(8)] == [0, 0, 0, 0, 1, 1, 1, 1] and [column_owner(c, 3, world) for c in range(3)] == [0, 0, 1]
dist.barrier()
dist.destroy_process_group()
sys.stdout.write("rank %%d ok\n" %% rank); sys.stdout.flush()
'''


def test_shard_range_covers_everything():
    from sha2_on_cq_halo2_amd.parallel import shard_range

    for n in (0, 1, 5, 8, 1000003):
        for world in (1, 2, 3, 8):
            pieces = [shard_range(n, r, world) for r in range(world)]
            assert pieces[0][0] == 0 and pieces[-1][1] == n
            assert all(pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in pieces]
            assert max(sizes) - min(sizes) <= 1


def test_sharded_multiexp_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("rank ") == 2 and r.stdout.count(" ok") == 2, r.stdout  # both ranks finished (lines may interleave)
