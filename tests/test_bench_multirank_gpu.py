"""GPU: the N > 1 path of bench.py, end to end on ONE card -- what the driver's scaling run executes with
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`, here with CQ_BENCH_DIST_BACKEND=gloo (every rank on
cuda:0, the sharded legs over the library's host-callback transport instead of RCCL, which refuses two ranks on one device):
the replicas headline with its barriers and MAX reduction, the configs[3] leg in child processes (k20_in_children), the JSON
merge on rank 0.  The line is validated, and the sharded proof's digest must equal the unsharded proof's."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(world, extra_env, port, args=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", CQ_BENCH_DIST_BACKEND="gloo", GPU_MAX_HW_QUEUES="8", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1", *args]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=700, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]  # ONE line, from rank 0
    return json.loads(lines[0])


def test_bench_two_ranks_on_one_gpu(ctx):
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    kk, steps = 16, 2
    line = _run_bench(2, {"CQ_BENCH_K20_K": str(kk), "CQ_BENCH_K20_STEPS": str(steps), "CQ_BENCH_K20_TIMEOUT": "300"}, 29731)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["unit"] == "Mscalar/s" and line["value"] > 0 and line["higher_is_better"] is True
    assert line["config"]["dist_backend"] == "gloo" and "replicas x2" in line["config"]["parallelism"]
    # whole-job aggregate: two proofs per step
    assert abs(line["value"] - 2 * line["config"]["msm_scalars_per_proof"] / (line["ms_per_step"] / 1e3) / 1e6) < 1e-6 * line["value"]
    assert "roofline" in line and line["roofline"]["launches"] > 0
    k20 = line["config3_k20"]
    assert "error" not in k20, k20
    assert k20["n_gpus"] == 2 and k20["k"] == kk and k20["scaling"] == "strong"
    wl = ShaCqWorkload(ctx, kk, seed=0x5348413243515F)
    wl.fill_witness()
    unsharded = hashlib.sha256(wl.prove(seed=100 + steps - 1)).hexdigest()[:16]
    wl.close()
    for leg in ("msm_and_columns_sharded", "msm_sharded_only", "resident"):
        assert k20[leg]["proof_sha256"] == unsharded, (leg, k20[leg])
        assert k20[leg]["ms_per_proof"] > 0 and k20[leg]["msm_kernel_mscalar_per_s"] > 0


def test_bench_shard_mode_two_ranks_on_one_gpu():
    """CQ_BENCH_MODE=shard: the headline itself is ONE proof over both ranks (strong scaling)."""
    line = _run_bench(2, {"CQ_BENCH_MODE": "shard"}, 29741, args=("--no-k20", "--k", "16"))
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and "one proof" in line["config"]["parallelism"]
    assert abs(line["value"] - line["config"]["msm_scalars_per_proof"] / (line["ms_per_step"] / 1e3) / 1e6) < 1e-6 * line["value"]
