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
# resident mode: transformed columns stay on their owner (one lookup and two advice columns per rank here), slices of b_0,
# of the partial quotients and of the GWC batch polynomial travel point to point (gloo isend / irecv through the host hook)
wl.pk.set_sharding(rank, world, columns=True, resident=True)
assert wl.prove(seed=9) == single, "resident-sharded proof differs from the single-GPU proof"
one = ShaCqWorkload(ctx, 11, pairs=1)         # ONE lookup over two ranks: rank 1 owns no lookup (no partial quotient of its own)
one_single = one.prove(seed=3)
one.pk.set_sharding(rank, world, columns=True, resident=True)
assert one.prove(seed=3) == one_single, "resident-sharded proof with a lookup-less rank differs"
one.pk.set_sharding(0, 1); one.close()
wl.pk.set_sharding(rank, world, columns=True)
# The random polynomial is committed with round 2 or in a launch of its own, by a timing-dependent choice that the ranks
# must make together (one launch more = one all-gather more): pin it differently per rank, then both ways.
for late in (("1", "0"), ("0", "1"), ("1", "1"), ("0", "0")):
    os.environ["CQ_RANDOM_LATE"] = late[rank]
    assert wl.prove(seed=9) == single, "sharded proof differs with CQ_RANDOM_LATE=%%s" %% (late,)
del os.environ["CQ_RANDOM_LATE"]
# a rank that cannot set its proof up (injected on rank 1): BOTH ranks return an error, neither waits for the other
from sha2_on_cq_halo2_amd import CqError
os.environ["CQ_TEST_FAIL_SETUP"] = "2"
try:
    wl.prove(seed=9)
    raise SystemExit("the injected set-up failure went unnoticed on rank %%d" %% rank)
except CqError as e:
    assert ("CQ_TEST_FAIL_SETUP" in str(e)) == (rank == 1) and ("another rank" in str(e)) == (rank == 0), str(e)
del os.environ["CQ_TEST_FAIL_SETUP"]
assert wl.prove(seed=9) == single, "proof after the agreed failure differs"
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
        os.environ.pop("CQ_RANDOM_LATE", None)
        wl.pk.set_sharding(0, 1, transport="rccl", resident=True)  # the resident flow over the one rank (its exchanges are local copies)
        assert wl.prove(seed=9) == plain
    finally:
        os.environ.pop("CQ_RANDOM_LATE", None)
        ctx.comm_destroy()
        wl.pk.set_sharding(0, 1)
    assert wl.prove(seed=9) == plain
    wl.close()


BIG_SHARD_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
k = %(k)d
ctx = Context(0)
wl = ShaCqWorkload(ctx, k)                      # BASELINE shape: 8 advice columns, 4 lookups, the SHA trace of that size
n = 1 << k
assert ctx.msm_table_width(wl.params.g_dev, n) == %(bits)d
single = wl.prove(seed=21)
for columns, resident in ((True, False), (False, False), (True, True)):
    wl.pk.set_sharding(rank, world, columns=columns, resident=resident)
    lo = 0 if rank == 0 else n // 2
    # the rank's slice has tables of its own, at the width its length calls for; the whole-array tables are gone
    assert ctx.msm_table_width(wl.params.g_dev + 64 * lo, n // 2) == 15 and ctx.msm_table_width(wl.params.g_dev, n) == 0
    assert wl.prove(seed=21) == single, "sharded proof differs (columns=%%s, resident=%%s)" %% (columns, resident)
wl.pk.set_sharding(0, 1)
assert ctx.msm_table_width(wl.params.g_dev, n) == %(bits)d and wl.prove(seed=21) == single
dist.barrier(); dist.destroy_process_group(); wl.close(); ctx.close()
sys.stdout.write("rank %%d ok %%s\n" %% (rank, hashlib.sha256(single).hexdigest()[:12])); sys.stdout.flush()
'''


@pytest.mark.parametrize("k,bits", [(18, 15), (20, 17)])
def test_sharded_create_proof_at_baseline_sizes(tmp_path, k, bits):
    """BASELINE configs[2] / configs[3] sizes under 2-rank sharding (gloo callbacks, both ranks on this card): the k = 20 key
    has 17-bit window tables unsharded and 15-bit ones for the 2^19-point slices; column transforms on and off; the proof
    bytes equal the unsharded proof's on both ranks."""
    script = tmp_path / "big_shard_worker.py"
    script.write_text(BIG_SHARD_WORKER % {"root": ROOT, "k": k, "bits": bits})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29660 + k), str(script)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("rank ") == 2 and r.stdout.count(" ok") == 2, r.stdout
    digests = {ln.split()[-1] for ln in r.stdout.splitlines() if ln.startswith("rank ")}
    assert len(digests) == 1, r.stdout


def test_rank_local_setup_failure_is_agreed_not_hung(ctx):
    """A rank that cannot set its proof up (allocation failure: injected with CQ_TEST_FAIL_SETUP) must not leave its peers in
    a collective: the outcome of the set-up is agreed in a one-word all-gather before anything else is exchanged, and every
    rank returns an error.  Exercised here over the one-rank RCCL communicator (the handshake itself is an ncclAllGather)."""
    from sha2_on_cq_halo2_amd import CqError
    from sha2_on_cq_halo2_amd.api import rccl_unique_id
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    wl = ShaCqWorkload(ctx, 10, pairs=2)
    plain = wl.prove(seed=2)
    ctx.comm_init_rccl(0, 1, rccl_unique_id())
    try:
        wl.pk.set_sharding(0, 1, transport="rccl")
        os.environ["CQ_TEST_FAIL_SETUP"] = "1"  # rank 0 + 1
        with pytest.raises(CqError) as e:
            wl.prove(seed=2)
        assert "CQ_TEST_FAIL_SETUP" in str(e.value)
        del os.environ["CQ_TEST_FAIL_SETUP"]
        assert wl.prove(seed=2) == plain  # the communicator is still good: nothing was left half-exchanged
    finally:
        os.environ.pop("CQ_TEST_FAIL_SETUP", None)
        ctx.comm_destroy()
        wl.pk.set_sharding(0, 1)
    wl.close()


RESIDENT_WORKER = r'''
import os, sys, hashlib
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ctx = Context(0)
for k, pairs in ((11, 2), (12, 4), (10, 3)):
    wl = ShaCqWorkload(ctx, k, pairs=pairs)
    single = wl.prove(seed=7)
    wl.pk.set_sharding(rank, world, columns=True, resident=True)
    for late in ("0", "1"):
        os.environ["CQ_RANDOM_LATE"] = late
        assert wl.prove(seed=7) == single, "resident-sharded proof differs (world %%d, k %%d, %%d lookups, late %%s)" %% (world, k, pairs, late)
    del os.environ["CQ_RANDOM_LATE"]
    wl.pk.set_sharding(0, 1)
    wl.close()
dist.barrier(); dist.destroy_process_group(); ctx.close()
sys.stdout.write("rank %%d ok\n" %% rank); sys.stdout.flush()
'''


@pytest.mark.parametrize("world", [3, 4])
def test_resident_sharding_three_and_four_ranks_one_gpu(tmp_path, world):
    """Resident column sharding over 3 ranks (point ranges of unequal length: 2^k and 2^k - 1 are not multiples of 3; one
    rank without a lookup when there are two) and 4 ranks (one lookup per rank at pairs = 4; three lookups over four ranks),
    gloo point-to-point through the host hook, every rank on this card: proof bytes equal the unsharded proof's."""
    script = tmp_path / "resident_worker.py"
    script.write_text(RESIDENT_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world,
                        "--master-addr", "127.0.0.1", "--master-port", str(29690 + world), str(script)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count(" ok") == world, r.stdout
