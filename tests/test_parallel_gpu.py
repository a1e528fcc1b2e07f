"""GPU, 2 ranks on one card over gloo: sharded multiexp with the real GpuBackend (HIP MSM per rank,
all-gather of Jacobian partials, local EC sum) equals the single-GPU result and the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch.distributed as dist
from oracle import bn254 as B, cbind as OC
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.parallel import GpuBackend, shard_range, sharded_multiexp
from tests.util import random_points
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ctx = Context(0)
be = GpuBackend(ctx)
n = 5000
pts = np.tile(B.points_to_mont_limbs(random_points(500, 4)), (10, 1))
rs = np.random.RandomState(3)
sc = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)
lo, hi = shard_range(n, rank, world)
got = sharded_multiexp(be, sc[lo:hi], pts[lo:hi])
exp = OC.best_multiexp(sc, pts)
assert np.array_equal(OC.g1_to_affine(got), OC.g1_to_affine(exp))
assert np.array_equal(OC.g1_to_affine(ctx.best_multiexp(sc, pts)), OC.g1_to_affine(exp))
dist.barrier(); dist.destroy_process_group(); ctx.close()
sys.stdout.write("rank %%d ok\n" %% rank); sys.stdout.flush()
'''


def test_sharded_multiexp_two_ranks_one_gpu(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29633", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("rank ") == 2 and r.stdout.count(" ok") == 2, r.stdout  # both ranks finished (lines may interleave)


SHARD_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ctx = Context(0)
wl = ShaCqWorkload(ctx, 12, pairs=2)          # same seed on every rank: same pk, same witness
single = wl.prove(seed=9)
# MSMs by point range only (every rank still transforms every column) ...
wl.pk.set_sharding(rank, world, columns=False)
assert wl.prove(seed=9) == single, "sharded proof differs from the single-GPU proof"
# ... and with the column transforms split by owner, the transformed columns broadcast (gloo through the host hooks)
wl.pk.set_sharding(rank, world, columns=True)
sharded = wl.prove(seed=9)
assert sharded == single, "column-sharded proof differs from the single-GPU proof"
# The random polynomial is committed with round 2 or in a launch of its own, by a timing-dependent choice that the ranks
# must make together (one launch more = one all-gather more): pin it differently per rank, then both ways.
for late in (("1", "0"), ("0", "1"), ("1", "1"), ("0", "0")):
    os.environ["CQ_RANDOM_LATE"] = late[rank]
    assert wl.prove(seed=9) == single, "sharded proof differs with CQ_RANDOM_LATE=%%s" %% (late,)
del os.environ["CQ_RANDOM_LATE"]
wl.pk.set_sharding(0, 1)
assert wl.prove(seed=9) == single
dist.barrier(); dist.destroy_process_group(); ctx.close()
sys.stdout.write("rank %%d ok %%s\n" %% (rank, hashlib.sha256(single).hexdigest()[:12])); sys.stdout.flush()
'''


def test_sharded_create_proof_two_ranks_one_gpu(tmp_path):
    """cq_pk_set_sharding / cq_pk_set_column_sharding: every commitment is split by point range across 2 ranks (window
    tables for the rank's slices only), partials are all-gathered and summed; independent column transforms are split
    by owner and broadcast; the proof bytes equal the unsharded ones on both ranks."""
    script = tmp_path / "shard_worker.py"
    script.write_text(SHARD_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("rank ") == 2 and r.stdout.count(" ok") == 2, r.stdout  # both ranks finished (lines may interleave)


PLONK_SHARD_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaPlonkWorkload
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ctx = Context(0)
wl = ShaPlonkWorkload(ctx, 11, pairs=3)   # gates + permutation: advice / z cosets are column-sharded too; 3 lookups over 2 ranks: uneven
for opener in ("gwc", "shplonk"):
    wl.pk.set_opener(opener)
    wl.pk.set_sharding(0, 1)
    single = wl.prove(seed=4)
    wl.pk.set_sharding(rank, world, columns=True)
    assert wl.prove(seed=4) == single, opener
dist.barrier(); dist.destroy_process_group(); ctx.close()
sys.stdout.write("rank %%d ok\n" %% rank); sys.stdout.flush()
'''


def test_sharded_general_circuit_two_ranks_one_gpu(tmp_path):
    """The general-PLONK path (gates, permutation, both openers) under MSM + column sharding, 2 ranks over gloo."""
    script = tmp_path / "plonk_shard_worker.py"
    script.write_text(PLONK_SHARD_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29649", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("rank ") == 2 and r.stdout.count(" ok") == 2, r.stdout


def test_rccl_path_single_rank(ctx):
    """The RCCL transport itself, on the one GPU this box has: a one-rank communicator, the collective self-check
    (ncclAllGather + grouped ncclBroadcasts on device buffers), and a proof in single-rank sharded mode -- every
    collective of the sharded prover is issued through RCCL -- byte-equal to the plain proof."""
    from sha2_on_cq_halo2_amd.api import rccl_unique_id
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    wl = ShaCqWorkload(ctx, 12, pairs=2)
    plain = wl.prove(seed=9)
    ctx.comm_init_rccl(0, 1, rccl_unique_id())
    try:
        ctx.comm_selftest()
        wl.pk.set_sharding(0, 1, transport="rccl")
        for late in ("0", "1"):
            os.environ["CQ_RANDOM_LATE"] = late
            assert wl.prove(seed=9) == plain
    finally:
        os.environ.pop("CQ_RANDOM_LATE", None)
        ctx.comm_destroy()
        wl.pk.set_sharding(0, 1)
    assert wl.prove(seed=9) == plain
    wl.close()
