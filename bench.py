#!/usr/bin/env python3
"""bench.py -- create_proof (CQ lookup + KZG commitment hot path) for the SHA-shaped CQ circuit on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` (N>1: launched by torch.distributed.run,
one rank per GPU).  Prints ONE JSON line on rank 0.

A "step" is one full proof at k=18 (BASELINE.json configs[2]: "k=18 64-block SHA256, 1xMI355X: full
CQ lookup commit + MSM + NTT on GPU"): SHA-256 trace words resident in HBM -> limb/spread witness
fill -> advice commitments -> CQ rounds 1 and 2 -> vanishing / quotient -> evaluations -> GWC
opening; the host only runs the Blake2b transcript, draws the blinding scalars and folds MSM window
sums.  `ms_per_step` is therefore the proof-generation wall-clock and `value` the MSM scalars
committed per second of proving (the metric's two halves); the MSM and NTT kernels' own rates are
reported next to them.  Multi-GPU: independent proofs per rank (weak scaling, no data-path collective;
DESIGN.md section "Multi-GPU").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto 4 hardware queues by default; a context uses three (main, upload, side
# compute), so two contexts in one process (the `two_in_flight` leg) would share queues and serialise.  Must be set
# before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
MSM_BYTES_PER_SCALAR = 96  # 32 B scalar + 64 B affine base
NTT_BYTES_PER_ELEM = 64  # read once + write once
CPU_BASELINE_K = 18  # the metric's own configuration, proven once by the CPU restatement (~20 s on 16 threads)
CPU_BASELINE_SMALL_K = 16  # kept beside it: BASELINE configs[1] (16 SHA blocks)
PMC_TRAFFIC_FILE = "profiles/r03_pmc_traffic_k18_proof.json"
PMC_SQ_FILE = "profiles/r03_pmc_sq_accumulate_k18.json"
PMC_SQ_NTT_FILE = "profiles/r03_pmc_sq_ntt_k18.json"
PMC_FALLBACK = {"profiles/r03_pmc_traffic_k18_proof.json": "profiles/r02_pmc_traffic_k18_proof.json",
                "profiles/r03_pmc_sq_accumulate_k18.json": "profiles/r02_pmc_sq_accumulate_k18.json"}


def _pmc_path(name):
    """The committed PMC summary of this round, or the previous round's when this round's is not there yet."""
    path = os.path.join(ROOT, name)
    return (path, name) if os.path.exists(path) else (os.path.join(ROOT, PMC_FALLBACK[name]), PMC_FALLBACK[name])


class Dist:
    """The job's process group.  Default: "nccl" (= RCCL over xGMI), one rank per GPU, the library's own RCCL transport
    for the sharded legs.  CQ_BENCH_DIST_BACKEND=gloo: the SAME code path on ONE GPU -- every rank on cuda:0, the timing
    reductions on CPU tensors, the sharded legs through the library's host-callback transport -- so that
    `torchrun --nproc-per-node 2 bench.py --gpus 2` can be rehearsed (and is tested, tests/test_bench_multirank_gpu.py) on a
    box with one card; it measures nothing about xGMI."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = os.environ.get("CQ_BENCH_DIST_BACKEND", "nccl")
        assert self.backend in ("nccl", "gloo"), self.backend
        self.device_index = 0 if self.backend == "gloo" else self.local_rank
        self.transport = "rccl" if self.backend == "nccl" else "callback"

    def init(self):
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(self.device_index)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", self.device_index))
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)

    def barrier(self):
        import torch

        if self.world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, values):
        """element-wise MAX of a list of floats over the ranks"""
        if self.world == 1:
            return list(values)
        import torch
        import torch.distributed as dist

        t = torch.tensor(list(values), dtype=torch.float64, device=f"cuda:{self.device_index}" if self.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu()]

    def shard(self, ctx, pk, columns=True, resident=False):
        """one proof over all ranks: MSM point ranges + column transforms (multi-GPU)"""
        import torch

        if self.transport == "rccl":
            if not getattr(ctx, "_bench_comm", False):
                ctx.comm_init_from_torch(device=torch.device("cuda", self.device_index))
                ctx.comm_selftest()
                ctx._bench_comm = True
            pk.set_sharding(self.rank, self.world, transport="rccl", columns=columns, resident=resident)
        else:
            pk.set_sharding(self.rank, self.world, transport="callback", columns=columns, resident=resident)

    def finish(self):
        if self.world > 1:
            import torch.distributed as dist

            dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--k", type=int, default=18)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-plonk-variant", action="store_true")
    ap.add_argument("--no-in-flight", action="store_true")
    ap.add_argument("--no-generic-rng", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip host_advice / dense_witness / ntt_ext / msm_standalone / batched_k22")
    ap.add_argument("--no-k20", action="store_true", help="skip the BASELINE configs[3] leg (k=20, one proof over all ranks)")
    ap.add_argument("--k20-child", action="store_true", help=argparse.SUPPRESS)  # internal: the configs[3] leg in a process of its own
    args = ap.parse_args()
    if args.k20_child:
        return k20_child_main()

    D = Dist()
    rank, world, local_rank = D.rank, D.world, D.device_index
    import torch

    D.init()

    from sha2_on_cq_halo2_amd import Context
    from sha2_on_cq_halo2_amd.api import PROF_MSM_ACCUMULATE, PROF_MSM_ENTRIES, PROF_NTT_PASS
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    stream = torch.cuda.Stream(device=local_rank)  # HIP stream the library enqueues on
    ctx = Context(local_rank, stream.cuda_stream)

    k = args.k
    # ---- setup (untimed): SRS + table SRS from a seeded toxic waste (built on the GPU), proving key,
    #      SHA-256 trace words uploaded to HBM ----
    # CQ_BENCH_MODE=shard: ONE proof, every commitment sharded by point range across the ranks with an
    # all-gather of Jacobian partials over RCCL (strong scaling); default: one independent proof per rank.
    shard = os.environ.get("CQ_BENCH_MODE", "replicas") == "shard" and world > 1
    wl = ShaCqWorkload(ctx, k, seed=0x5348413243515F + (0 if shard else rank))
    if shard:
        D.shard(ctx, wl.pk)

    def step(i):
        wl.fill_witness()
        return wl.prove(seed=1000 + i)

    barrier = D.barrier

    proof = b""
    ctx.profile_enable(True)  # before the warm-up: the profiling events are created on first use, not in the timed region
    for i in range(args.warmup):
        proof = step(i)
    ctx.profile_read(PROF_MSM_ACCUMULATE)
    ctx.profile_read(PROF_NTT_PASS)
    ctx.profile_read(PROF_MSM_ENTRIES)
    import gc

    gc.disable()  # a generation-2 collection of the interpreter (torch's module graph) is milliseconds; it is not proving.
    # (No gc.collect() here: tens of milliseconds of idle GPU, and the first proof after an idle gap runs 0.5-1 ms slower.)
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    for i in range(args.steps):
        ts = time.perf_counter()
        proof = step(i)  # returns with the proof bytes on the host: every step ends synchronised
        step_ms.append((time.perf_counter() - ts) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    acc_ms, acc_calls = ctx.profile_read(PROF_MSM_ACCUMULATE)
    ntt_ms, ntt_calls = ctx.profile_read(PROF_NTT_PASS)
    _, msm_entries = ctx.profile_read(PROF_MSM_ENTRIES)
    ctx.profile_enable(False)

    elapsed = D.max_over_ranks([elapsed])[0]

    scalars_per_step = wl.msm_scalars_per_proof()
    proofs = 1 if shard else world
    value = scalars_per_step * args.steps * proofs / elapsed / 1e6

    out = {
        "metric": "sha256_cq_create_proof_k18: MSM Mscalar/s over proof-generation wall-clock (ms_per_step)",
        "value": value,
        "unit": "Mscalar/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if shard else "weak",
        "vs_baseline": None,
        "dtype": "u256-montgomery",
        "data": "synthetic (SHA-256 trace of bytes i mod 251; KZG/table SRS from a seeded toxic waste, built on the GPU)",
        "config": {
            "workload": f"k={k} {wl.blocks}-block SHA-256-shaped CQ circuit: {2 * wl.pairs} advice columns, {wl.pairs} width-2 static "
            f"lookups into 2^12-entry (dense, spread) tables, degree 3; full create_proof (GWC, Blake2b), proof {len(proof)} B",
            "k": k,
            "msm_scalars_per_proof": scalars_per_step,
            "msm_scalars_in_gpu_launches": wl.msm_scalars_in_launches(),
            "msm_scalars_full_width": wl.msm_scalars_full_width(),
            "ntt_elems_per_proof": wl.ntt_elems_per_proof(),
            "parallelism": (f"one proof, MSM point ranges and column transforms sharded x{world} (RCCL all-gather / broadcast)" if shard else
                            f"replicas x{world} (one independent proof per GPU, no collective)"),
            "dist_backend": D.backend if world > 1 else None,
        },
        "proof_wall_s": elapsed / args.steps,
        "step_ms_all_this_rank": [round(x, 3) for x in step_ms],
        "step_ms_this_rank": {"min": min(step_ms), "median": sorted(step_ms)[len(step_ms) // 2], "max": max(step_ms)},
        "proofs_per_s": args.steps * proofs / elapsed,
        # `value` counts every scalar the reference's create_proof would hand to best_multiexp, including the 8n advice
        # scalars of this witness (<= 24-bit limbs on 9.4 % of the rows) and the 4n of the f commitments (derived by
        # linearity here); this is the same rate over the scalars that are uniformly distributed field elements only
        "dense_equivalent_mscalar_per_s": wl.msm_scalars_full_width() * args.steps * proofs / elapsed / 1e6,
    }
    # ---- BASELINE configs[3]: ONE k = 20 proof over all the ranks of the job (strong scaling).  Collective: every rank
    #      takes part; rank 0 keeps the figures.  A watchdog prints the line without it should a collective hang. ----
    k20 = None
    if not args.no_k20 and k == 18:
        if world == 1 and os.environ.get("CQ_BENCH_K20_CHILD", "0") != "1":  # (the env switch: the child path on one GPU, for testing)
            try:
                k20 = config3_k20(ctx, D)
            except Exception as e:  # noqa: BLE001
                k20 = {"error": f"{type(e).__name__}: {e}"}
        else:
            k20 = k20_in_children(rank)
    if rank == 0:
        if k20 is not None:
            out["config3_k20"] = k20
        _finish_line(out, ctx, wl, args, k, acc_ms, acc_calls, ntt_ms, ntt_calls, msm_entries, world)
    ctx.close()
    D.finish()


def _finish_line(out, ctx, wl, args, k, acc_ms, acc_calls, ntt_ms, ntt_calls, msm_entries, world, light=False):
    """Rank 0: roofline blocks and secondary legs, then the ONE JSON line (light: no further GPU work)."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    peak = None if light else modmul_peak(ctx)
    if True:
        # dominant kernel: msm_accumulate_kernel (bucket accumulation).  Algorithmic bytes = 96 B per
        # (scalar, base) pair; the kernel is VALU-bound (256-bit modular arithmetic), so frac is small.
        acc_s = acc_ms / 1e3
        # the accumulate launches take in the scalars of every MSM the backend runs (the f commitments of the proof are
        # derived from the advice commitments by linearity and are not among them)
        units = wl.msm_scalars_in_launches() * args.steps
        achieved = MSM_BYTES_PER_SCALAR * units / acc_s / 1e9 if acc_s > 0 else 0.0
        out["roofline"] = {
            "kernel": "msm_accumulate_kernel",
            # what limits the kernel is VALU issue (256-bit modular integer arithmetic; no MFMA form exists); achieved /
            # peak / frac stay the HBM figures the metric asks for, valu_frac below is the informative fraction
            "bound": "valu",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            # PMC figures are NOT measured by this run: they are read from the committed rocprofv3 --pmc passes
            "traffic": pmc_traffic("msm_accumulate_kernel") if k == 18 else None,
            "traffic_source": f"{_pmc_path(PMC_TRAFFIC_FILE)[1]} (committed rocprofv3 --pmc pass over the same k=18 proof, not this run)" if k == 18 else None,
            "valu_issue_utilisation_pmc": pmc_valu_issue() if k == 18 else None,
            "valu_issue_utilisation_source": f"{_pmc_path(PMC_SQ_FILE)[1]} (committed rocprofv3 --pmc pass, not this run)" if k == 18 else None,
            "launches": int(acc_calls),
            "avg_launch_ms": acc_ms / max(acc_calls, 1),
            "msm_kernel_mscalar_per_s": units / acc_s / 1e6 if acc_s > 0 else 0.0,
            # the bound that applies: XYZZ mixed additions (8M + 2S = 10 Montgomery products each) against the
            # chip's measured Montgomery-product rate
            "mixed_additions": int(msm_entries),
            "modmul_per_s": 10 * msm_entries / acc_s if acc_s > 0 else 0.0,
            "modmul_peak_measured": peak,
            "valu_frac": (10 * msm_entries / acc_s) / peak if acc_s > 0 and peak else None,
            "note": "VALU-bound integer kernel (no MFMA applies), so the HBM fraction is small by construction; "
                    "valu_frac = Montgomery products executed / measured chip peak (cq_bench_modmul_dev), see DESIGN.md",
        }
        # NTT: measured stand-alone (8 columns of 2^k, lagrange_to_coeff -- the advice transform of the proof), because
        # inside a proof most passes run on the side stream underneath the MSM tail kernels and their event-bracketed
        # times include the kernels they share the GPU with (kept as `in_proof`).
        sa_error = None
        try:
            if light:
                raise RuntimeError("skipped (watchdog path)")
            sa_elems, sa_ms, sa_calls = ntt_standalone(ctx, k)
        except Exception as e:  # noqa: BLE001 -- reported, not papered over with the overlapped in-proof figure
            sa_elems, sa_ms, sa_calls, sa_error = 0, 0.0, 0, f"{type(e).__name__}: {e}"
        ntt_ach = NTT_BYTES_PER_ELEM * sa_elems / (sa_ms / 1e3) / 1e9 if sa_ms > 0 else 0.0
        ntt_elems = wl.ntt_elems_per_proof() * args.steps
        out["roofline_ntt"] = {
            "kernel": "ntt_pass_kernel (all passes of a transform)",
            "bound": "hbm",
            "achieved": ntt_ach,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ntt_ach / HBM_PEAK_GBS,
            "traffic": pmc_traffic("ntt_pass_kernel<6u, 4u>") if k == 18 else None,  # the 2^18 transform's passes
            "traffic_source": f"{_pmc_path(PMC_TRAFFIC_FILE)[1]} (committed rocprofv3 --pmc pass, not this run)" if k == 18 else None,
            # what actually bounds a pass: VALU issue (0.95 between a launch's first and last workgroup completions, ~0.71
            # over a two-generation launch: profiles/r03_ntt_workgroup_timeline.txt), not HBM
            "valu_issue_utilisation_pmc": pmc_valu_issue_ntt() if k == 18 else None,
            "valu_source": f"{PMC_SQ_NTT_FILE} (committed rocprofv3 --pmc pass over the stand-alone transform, not this run)" if k == 18 else None,
            "error": sa_error,
            "launches": int(sa_calls),
            "melem_per_s": sa_elems / (sa_ms / 1e3) / 1e6 if sa_ms > 0 else 0.0,
            "sample": f"10 x lagrange_to_coeff of 8 columns of 2^{k} (3 passes each), nothing else on the GPU",
            "in_proof": {"launches": int(ntt_calls), "melem_per_s": ntt_elems / (ntt_ms / 1e3) / 1e6 if ntt_ms > 0 else 0.0,
                         "note": "overlapped with MSM tail kernels (side stream): pass times include the co-running kernels"},
        }
        # secondary legs: a failure there must not cost the line
        def leg(name, fn):
            try:
                out[name] = fn()
            except Exception as e:  # noqa: BLE001
                out[name] = {"error": f"{type(e).__name__}: {e}"}

        if light:
            print(json.dumps(out), flush=True)
            return
        if world == 1 and not args.no_generic_rng:
            leg("generic_rng", lambda: generic_rng(wl, max(3, min(args.steps, 10))))
        if world == 1 and not args.no_in_flight:
            leg("batched", lambda: batched(ctx, wl, max(12, args.steps)))
            leg("two_in_flight", lambda: two_in_flight(ctx, wl, local_rank, max(10, args.steps)))
        if world == 1 and not args.no_plonk_variant:
            leg("plonk_variant", lambda: plonk_variant(ctx, wl, max(2, min(args.steps, 5))))
        if world == 1 and not args.no_extra_legs:
            leg("host_advice", lambda: host_advice(wl, max(5, min(args.steps, 10))))
            leg("dense_witness", lambda: dense_witness(ctx, wl, max(5, min(args.steps, 10))))
            leg("ntt_ext", lambda: ntt_ext(ctx, k))
            leg("msm_standalone", lambda: msm_standalone(ctx))
            leg("batched_k22", lambda: batched_k22(ctx))
        if world == 1 and not args.no_cpu_baseline:
            leg("cpu_baseline", lambda: cpu_baseline(ctx))
        print(json.dumps(out), flush=True)


_MODMUL_PEAK = []


def modmul_peak(ctx):
    """Chip-wide Montgomery products per second in the field form the MSM kernels use (lazy 9 x 29-bit limbs),
    measured now with the library's microbenchmark (independent chains of products in registers, no memory traffic)."""
    if not _MODMUL_PEAK:
        import torch

        lanes, iters = 256 * 256 * 8, 4096
        buf = ctx.alloc(lanes * 32)
        ctx._chk(ctx.lib.cq_bench_modmul_dev(ctx.h, buf.ptr, lanes, iters, 2))
        torch.cuda.synchronize()
        ctx.sync()
        t0 = time.perf_counter()
        ctx._chk(ctx.lib.cq_bench_modmul_dev(ctx.h, buf.ptr, lanes, iters, 2))
        ctx.sync()
        _MODMUL_PEAK.append(lanes * iters / (time.perf_counter() - t0))
        buf.free()
    return _MODMUL_PEAK[0]


def batched(ctx, wl, count):
    """Secondary figure (not `value`): BASELINE configs[4]'s shape on one GPU -- `count` independent instances of the
    circuit through cq_create_proof_batch (independent witnesses, transcripts and RNG streams; the library keeps up to
    `lanes` of them in flight, each on a stream and host thread of its own)."""
    import torch

    host = [c.download((wl.n, 4)) for c in wl.cols]
    cols = []
    for _ in range(count):  # every instance gets a witness of its own in HBM (the same SHA trace: what is timed is proving)
        cols.append([ctx.to_device(h) for h in host])
    ptrs = [[c.ptr for c in mine] for mine in cols]
    res = {"instances": count}
    single = wl.pk.create_proof_dev(ptrs[0], seed=500)
    for lanes in (2, 3, 4):
        wl.pk.create_proof_batch(ptrs[:lanes], [1 + i for i in range(lanes)], lanes=lanes)  # warm the lanes (twiddles, arenas)
        torch.cuda.synchronize()
        # three batches, the median reported (all listed)
        samples = []
        for _ in range(3):
            t0 = time.perf_counter()
            proofs = wl.pk.create_proof_batch(ptrs, [500 + i for i in range(count)], lanes=lanes)
            samples.append(time.perf_counter() - t0)
        dt = sorted(samples)[1]
        res[f"lanes_{lanes}"] = {"proofs_per_s": count / dt, "ms_per_proof_effective": dt / count * 1e3,
                                 "mscalar_per_s": count * wl.msm_scalars_per_proof() / dt / 1e6,
                                 "batch_wall_s_samples": samples, "reported": "median of 3 batches",
                                 "first_proof_equals_single": proofs[0] == single}
    for mine in cols:
        for b in mine:
            b.free()
    return res


def two_in_flight(ctx, wl, device, reps):
    """Secondary figure (not `value`): throughput with TWO independent proofs in flight on the one GPU -- a second
    context (own stream, proving key, witness) driven by a second host thread.  The latency-bound tail of one
    proof's MSM launches is filled by the other proof's kernels (BASELINE configs[4]: batches of independent proofs)."""
    import threading

    from sha2_on_cq_halo2_amd import Context
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    ctxs = [Context(device), Context(device)]  # library-owned streams
    wls = [ShaCqWorkload(c, wl.k, seed=0x5348413243515F + 1000 + i) for i, c in enumerate(ctxs)]
    for w in wls:
        w.prove(seed=1)
        w.prove(seed=2)
    bar = threading.Barrier(3, timeout=300)  # a failing worker breaks the barrier instead of hanging the bench
    errors = []

    def worker(w):
        try:
            bar.wait()
            for i in range(reps):
                w.fill_witness()
                w.prove(seed=50 + i)
            bar.wait()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            bar.abort()

    ts = [threading.Thread(target=worker, args=(w,), daemon=True) for w in wls]
    for t in ts:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join(timeout=60)
    if errors:
        raise errors[0]
    for c in ctxs:
        c.close()
    return {"proofs": 2 * reps, "proofs_per_s": 2 * reps / dt, "mscalar_per_s": 2 * reps * wl.msm_scalars_per_proof() / dt / 1e6,
            "ms_per_proof_effective": dt / (2 * reps) * 1e3}


def plonk_variant(ctx, wl, steps):
    """Secondary figure (not `value`): the same SHA-shaped circuit with a custom gate (rotation), a selector
    column and copy constraints around the lookups -- the general-PLONK path (gate interpreter, permutation
    grand product by scan, multi-point openings), with both multi-open schemes."""
    import torch

    from sha2_on_cq_halo2_amd.sha_circuit import ShaPlonkWorkload

    pw = ShaPlonkWorkload(ctx, wl.k, seed=0x5348413243515F, share=wl)
    res = {"workload": f"k={wl.k}: {2 * wl.pairs + 2} advice + 1 fixed columns, 2 gates (one with Rotation::next), "
                       f"{wl.pairs} static lookups, permutation over 2 columns ({2 ** wl.k - 7} copy constraints)"}
    for opener in ("gwc", "shplonk"):
        pw.pk.set_opener(opener)
        proof = pw.prove(seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            proof = pw.prove(seed=2 + i)
        torch.cuda.synchronize()
        res[opener] = {"ms_per_proof": (time.perf_counter() - t0) / steps * 1e3, "proof_bytes": len(proof)}
    return res


def _uniform_scalars(count, seed):
    rs = np.random.RandomState(seed)
    a = rs.randint(0, 2**63, size=(count, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)  # below the modulus whatever the rest
    return a


def _median(xs):
    return sorted(xs)[len(xs) // 2]


def msm_standalone(ctx, sizes=(16, 18, 20, 22), reps=5):
    """MSM throughput as the metric defines it -- m / t over the WALL time of one dense `best_multiexp`
    (arithmetic.rs:132-159) with uniform scalars over a resident 2^m-point SRS (`commit`, kzg/commitment.rs:539-543):
    digits, sort, bucket accumulation, bucket reduction, read-back and the host's fold, result on the host.  `single` = one
    MSM per call (latency-bound at small m: the tail of a launch is a chain of dependent group additions), `batch8` = eight
    MSMs over the same bases in one launch (cq_msm_batch_dev; what a round of create_proof issues)."""
    from sha2_on_cq_halo2_amd import ParamsKZG
    from sha2_on_cq_halo2_amd.api import fr_to_mont

    res = {}
    s = fr_to_mont(0x5348413243515F * 0x9E3779B97F4A7C15 + 12345)
    for m in sizes:
        n = 1 << m
        params = ParamsKZG.setup_from_toxic_waste(ctx, m, s)
        host = _uniform_scalars(n, 100 + m)
        cols = [ctx.to_device(np.roll(host, 7 * j, axis=0)) for j in range(8)]
        ctx.best_multiexp_dev(cols[0], _Ptr(params.g_dev), n)
        ctx.msm_batch_dev([c.ptr for c in cols], params.g_dev, n)
        single, batch = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.best_multiexp_dev(cols[0], _Ptr(params.g_dev), n)
            single.append(time.perf_counter() - t0)
        for _ in range(reps):
            t0 = time.perf_counter()
            ctx.msm_batch_dev([c.ptr for c in cols], params.g_dev, n)
            batch.append(time.perf_counter() - t0)
        res[f"2^{m}"] = {"single_ms": _median(single) * 1e3, "single_mscalar_per_s": n / _median(single) / 1e6,
                         "batch8_ms": _median(batch) * 1e3, "batch8_mscalar_per_s": 8 * n / _median(batch) / 1e6,
                         "table_window_bits": ctx.msm_table_width(params.g_dev, n)}
        for c in cols:
            c.free()
        params.close()
    res["note"] = "wall time per call, median of %d, scalars and bases resident in HBM, result (one Jacobian point per MSM) on the host" % reps
    return res


class _Ptr:
    """a bare device pointer where the API takes a DevBuf"""

    def __init__(self, ptr):
        self.ptr = ptr


def ntt_ext(ctx, k, batch=8, reps=10):
    """SURVEY 8(d)(3) at the extended size: `coeff_to_extended` (poly/domain.rs:252-266: coset shift, zero padding to
    2^extended_k, NTT) of `batch` columns of 2^k coefficients, and `extended_to_coeff` (:293-315) of one column; wall time
    per call, nothing else on the GPU.  Elements counted at the transform size (2^extended_k)."""
    from sha2_on_cq_halo2_amd.api import EvaluationDomain

    dom = EvaluationDomain(ctx, 3, k)
    n, ext = 1 << k, dom.extended_len
    src = ctx.to_device(_uniform_scalars(batch * n, 11))
    dst = ctx.alloc(batch * ext * 32)
    out = {"extended_k": dom.extended_len.bit_length() - 1}
    ctx._chk(ctx.lib.cq_coeff_to_extended_dev(dom.h, src.ptr, dst.ptr, batch))
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx._chk(ctx.lib.cq_coeff_to_extended_dev(dom.h, src.ptr, dst.ptr, batch))
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    out["coeff_to_extended"] = {"batch": batch, "ms_per_call": dt * 1e3, "melem_per_s": batch * ext / dt / 1e6,
                                "algorithmic_gb_per_s": batch * (32 * n + 32 * ext) / dt / 1e9}
    ctx._chk(ctx.lib.cq_extended_to_coeff_dev(dom.h, dst.ptr, src.ptr))
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx._chk(ctx.lib.cq_extended_to_coeff_dev(dom.h, dst.ptr, src.ptr))
    ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    out["extended_to_coeff"] = {"batch": 1, "ms_per_call": dt * 1e3, "melem_per_s": ext / dt / 1e6}
    for b in (src, dst):
        b.free()
    dom.close()
    return out


def host_advice(wl, steps):
    """The real caller's path: `WitnessCollection` hands over HOST advice columns (plonk/prover.rs:223-260), so the proof
    starts with their upload (cq_create_proof_host; 8 x 8 MiB at k = 18 over PCIe).  Never `value`: the headline starts
    with the columns resident in HBM."""
    host = [c.download((wl.n, 4)) for c in wl.cols]
    wl.pk.create_proof(host, seed=1)
    ms = []
    for i in range(steps):
        t0 = time.perf_counter()
        proof = wl.pk.create_proof(host, seed=1000 + i)
        ms.append((time.perf_counter() - t0) * 1e3)
    same = proof == wl.pk.create_proof_dev([c.ptr for c in wl.cols], seed=1000 + steps - 1)
    return {"ms_per_step_median": _median(ms), "ms_per_step_mean": sum(ms) / len(ms), "steps": steps, "upload_bytes": sum(h.nbytes for h in host),
            "proof_equals_device_resident": bool(same), "note": "pageable host columns -> pinned staging -> HBM inside the timed call"}


def dense_witness(ctx, wl, steps):
    """The same circuit with EVERY usable row assigned (the SHA trace of the headline fills 9.4 % of them): random table
    rows, so the eight advice commitments are MSMs over 2^k non-zero 12- / 24-bit scalars each."""
    from sha2_on_cq_halo2_amd.sha_circuit import small_to_mont, spread16

    N, u = wl.cfg.size, wl.pk.usable_rows
    idx = np.arange(N)
    dense_m, spread_m = small_to_mont(idx), small_to_mont(spread16(idx))
    rs = np.random.RandomState(5)
    cols = []
    for p in range(wl.pairs):
        rows = rs.randint(0, N, size=u)
        for tab in (dense_m, spread_m):
            col = np.zeros((wl.n, 4), dtype=np.uint64)
            col[:u] = tab[rows]
            cols.append(ctx.to_device(col))
    ptrs = [c.ptr for c in cols]
    wl.pk.create_proof_dev(ptrs, seed=1)
    ms = []
    for i in range(steps):
        t0 = time.perf_counter()
        wl.pk.create_proof_dev(ptrs, seed=2000 + i)
        ms.append((time.perf_counter() - t0) * 1e3)
    for c in cols:
        c.free()
    return {"ms_per_step_median": _median(ms), "ms_per_step_mean": sum(ms) / len(ms), "steps": steps, "assigned_rows_per_column": int(u),
            "mscalar_per_s": wl.msm_scalars_per_proof() / (_median(ms) / 1e3) / 1e6}


def batched_k22(ctx, instances=2, lanes=2, batches=3):
    """BASELINE configs[4] at its own size: the per-GPU unit of "k=22 batched proofs (16 independent SHA256 instances),
    8 x MI355X" is two k = 22 (1024-block) instances; here through cq_create_proof_batch with two lanes."""
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    wl = ShaCqWorkload(ctx, 22, seed=0x5348413243515F)
    try:
        host = [c.download((wl.n, 4)) for c in wl.cols]
        cols = [[ctx.to_device(h) for h in host] for _ in range(instances)]
        ptrs = [[c.ptr for c in mine] for mine in cols]
        single = wl.pk.create_proof_dev(ptrs[0], seed=500)
        t0 = time.perf_counter()
        wl.pk.create_proof_dev(ptrs[0], seed=500)
        one = time.perf_counter() - t0
        wl.pk.create_proof_batch(ptrs, [1 + i for i in range(instances)], lanes=lanes)
        samples = []
        for _ in range(batches):
            t0 = time.perf_counter()
            proofs = wl.pk.create_proof_batch(ptrs, [500 + i for i in range(instances)], lanes=lanes)
            samples.append(time.perf_counter() - t0)
        dt = _median(samples)
        for mine in cols:
            for b in mine:
                b.free()
        return {"k": 22, "blocks": wl.blocks, "instances": instances, "lanes": lanes, "ms_per_proof_effective": dt / instances * 1e3,
                "proofs_per_s": instances / dt, "mscalar_per_s": instances * wl.msm_scalars_per_proof() / dt / 1e6,
                "one_at_a_time_ms": one * 1e3, "batch_wall_s_samples": samples, "reported": "median of %d batches" % batches,
                "first_proof_equals_single": proofs[0] == single}
    finally:
        wl.close()


def pmc_traffic(kernel):
    """HBM bytes per launch from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same
    k=18 proof (profiles/README.md); None if the summary is missing.  FETCH_SIZE is taken at face value
    (the gather of 64-B points is not the wide coalesced stream the gfx950 x2 correction applies to); the
    doubled figure is kept alongside in the profile file."""
    path, _ = _pmc_path(PMC_TRAFFIC_FILE)
    try:
        with open(path) as f:
            d = json.load(f)
        for key in ("cq::" + kernel, "void cq::" + kernel):
            if key in d:
                return d[key]["hbm_bytes_per_launch_raw"]
        return None
    except Exception:
        return None


def ntt_standalone(ctx, k, batch=8, reps=10):
    """(elements, summed pass time in ms, pass launches) of `reps` batched lagrange_to_coeff transforms, timed by the
    library's HIP events around every ntt_pass_kernel launch."""
    import numpy as np

    from sha2_on_cq_halo2_amd.api import PROF_NTT_PASS, EvaluationDomain

    dom = EvaluationDomain(ctx, 3, k)
    n = 1 << k
    rs = np.random.RandomState(7)
    a = rs.randint(0, 2**63, size=(batch * n, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    src, dst = ctx.to_device(a), ctx.alloc(batch * n * 32)
    ctx._chk(ctx.lib.cq_lagrange_to_coeff_dev(dom.h, src.ptr, dst.ptr, batch))
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_read(PROF_NTT_PASS)
    for _ in range(reps):
        ctx._chk(ctx.lib.cq_lagrange_to_coeff_dev(dom.h, src.ptr, dst.ptr, batch))
    ctx.sync()
    ms, calls = ctx.profile_read(PROF_NTT_PASS)
    ctx.profile_enable(False)
    dom.close()
    return batch * n * reps, ms, calls


def pmc_valu_issue():
    """VALU issue utilisation of the accumulate launch that holds most of a proof's additions (SQ_INSTS_VALU x 4
    cycles / (1024 SIMDs x GPU cycles)), from the committed rocprofv3 --pmc pass over the same k=18 proof
    (profiles/README.md); None if the summary is missing."""
    path, _ = _pmc_path(PMC_SQ_FILE)
    try:
        with open(path) as f:
            rows = json.load(f)["kernels"]["msm_accumulate_kernel"]
        top = max(r["valu_wave_instructions"] for r in rows)
        same = [r["valu_issue_utilisation"] for r in rows if r["valu_wave_instructions"] > 0.99 * top]
        # the mean over the traced proofs' launches of that kind: one launch's ratio can read a few 1e-3 above 1 because
        # the cycle count is GRBM_GUI_ACTIVE averaged over the eight XCDs, which do not all run for the whole launch
        return min(1.0, sum(same) / len(same))
    except Exception:
        return None


def pmc_valu_issue_ntt():
    """VALU issue utilisation of a 2^18 x 8 NTT pass launch (mean over the committed pass's launches); None if missing."""
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_SQ_NTT_FILE)) as f:
            rows = json.load(f)["kernels"]["ntt_pass_kernel"]
        vals = [r["valu_issue_utilisation"] for r in rows if r["valu_issue_utilisation"]]
        return sum(vals) / len(vals)
    except Exception:
        return None


def k20_in_children(rank):
    """N > 1: the collective configs[3] leg runs in a CHILD process per rank (same RANK / WORLD_SIZE, rendezvous on another
    port, a communicator and a library context of its own), so that nothing it does -- a hung collective, a fault inside
    RCCL on hardware this code has never met -- can take the headline measurement of the parent with it.  Rank 0's child
    prints the block as JSON; a child that does not finish in time is killed (by pid) and reported as an error."""
    import subprocess

    # The child's rendezvous is its own: another port, and rank 0 of the children hosts the store itself -- under torchrun
    # TORCHELASTIC_USE_AGENT_STORE tells c10d that the launcher's agent already serves MASTER_PORT, so with it inherited
    # every child would wait as a client for a server nobody started (found by the two-ranks-on-one-GPU rehearsal).
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1000 + int(os.environ.get("WORLD_SIZE", "1")))
    env.setdefault("CQ_COMM_TIMEOUT_S", "45")  # a collective that does not complete fails that leg (library-side time-out), not the child
    timeout = float(os.environ.get("CQ_BENCH_K20_TIMEOUT", "300"))
    import tempfile

    errlog = os.path.join(tempfile.gettempdir(), f"cq_bench_k20_child_rank{rank}.err")

    def err_tail():
        try:
            with open(errlog, errors="replace") as f:
                return f.read()[-600:].strip()
        except OSError:
            return ""

    try:
        with open(errlog, "w") as ef:
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--k20-child"], env=env, stdout=subprocess.PIPE,
                                 stderr=ef, cwd=os.path.dirname(os.path.abspath(__file__)) or ".")
    except OSError as e:
        return {"error": f"could not start the child process: {e}"}
    timed_out = False
    try:
        stdout, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        timed_out = True
        p.kill()
        stdout, _ = p.communicate()  # what the child had printed: it reports after every leg
    if rank != 0:
        return None
    for line in reversed(stdout.decode(errors="replace").strip().splitlines()):
        if line.startswith("{"):
            try:
                res = json.loads(line)
            except ValueError:
                continue
            if timed_out:
                res["error"] = f"timed out after {timeout:.0f} s in a later leg (a collective did not complete); the legs listed had finished; headline unaffected"
                res["stderr_tail"] = err_tail()
            return res
    if timed_out:
        return {"error": f"timed out after {timeout:.0f} s (a collective did not complete); headline unaffected", "stderr_tail": err_tail()}
    return {"error": f"child exited with code {p.returncode} and no result; headline unaffected", "stderr_tail": err_tail()}


def k20_child_main():
    D = Dist()
    import torch

    D.init()
    from sha2_on_cq_halo2_amd import Context

    stream = torch.cuda.Stream(device=D.device_index)
    ctx = Context(D.device_index, stream.cuda_stream)
    try:
        res = config3_k20(ctx, D)
    except Exception as e:  # noqa: BLE001
        res = {"error": f"{type(e).__name__}: {e}"}
    if D.rank == 0:
        print(json.dumps(res), flush=True)
    sys.stdout.flush()
    os._exit(0)  # no teardown of communicators whose peers may be gone


def config3_k20(ctx, D, steps=None, warmup=2):
    """BASELINE configs[3]: one k = 20 (256-block) proof, proven by ALL ranks of the job together -- every MSM sharded by
    point range (ncclAllGather of the Jacobian partials), the independent column transforms sharded by owner
    (ncclBroadcast), both issued by the library on device buffers over the context's RCCL communicator.  At N = 1 the
    same proof unsharded: the driver's N = 1, 2, 4, 8 runs give the strong-scaling curve.  `msm_kernel_mscalar_per_s` =
    MSM scalars of the proof / the slowest rank's accumulate-kernel time (the figure the 1 -> 8 MSM-scaling target is
    about); the end-to-end time is bounded by what stays replicated (multi-GPU).  CQ_BENCH_K20_K (default 20)
    lets the tests run the same leg on a smaller circuit."""
    import hashlib

    from sha2_on_cq_halo2_amd.api import PROF_MSM_ACCUMULATE
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    world = D.world
    kk = int(os.environ.get("CQ_BENCH_K20_K", "20"))
    steps = steps or int(os.environ.get("CQ_BENCH_K20_STEPS", "4"))
    wl = ShaCqWorkload(ctx, kk, seed=0x5348413243515F)  # the same instance on every rank
    res = {"k": kk, "blocks": wl.blocks, "n_gpus": world, "scaling": "strong", "steps": steps, "dist_backend": D.backend if world > 1 else None,
           "msm_scalars_in_gpu_launches": wl.msm_scalars_in_launches()}

    def report():
        if D.rank == 0 and os.environ.get("CQ_BENCH_K20_PROGRESS", "1") == "1" and world > 1:
            print(json.dumps(res), flush=True)  # after every leg: a parent that has to kill the child keeps what was measured

    def run(label):
        try:
            run_leg(label)
        except Exception as e:  # noqa: BLE001 -- this leg is lost (e.g. a collective timed out and the communicator was aborted); the next
            res[label] = {"error": f"{type(e).__name__}: {e}"}  # one builds a new communicator
            ctx._bench_comm = False
        report()

    def run_leg(label):
        for i in range(warmup):
            wl.fill_witness()
            wl.prove(seed=10 + i)
        ctx.profile_enable(True)
        ctx.profile_read(PROF_MSM_ACCUMULATE)
        D.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            wl.fill_witness()
            proof = wl.prove(seed=100 + i)
        D.barrier()
        dt = time.perf_counter() - t0
        acc_ms, _ = ctx.profile_read(PROF_MSM_ACCUMULATE)
        ctx.profile_enable(False)
        dt, acc_ms = D.max_over_ranks([dt, acc_ms])
        res[label] = {"ms_per_proof": dt / steps * 1e3,
                      "msm_mscalar_per_s_wall": wl.msm_scalars_per_proof() * steps / dt / 1e6,
                      "msm_kernel_mscalar_per_s": wl.msm_scalars_in_launches() * steps / (acc_ms / 1e3) / 1e6 if acc_ms > 0 else 0.0,
                      "accumulate_ms_per_proof_slowest_rank": acc_ms / steps,
                      "proof_sha256": hashlib.sha256(proof).hexdigest()[:16]}

    if world == 1:
        run("unsharded")
    else:
        # simplest exchange first (all-gathers only), so that a failure in a later mode costs the least
        for label, kw in (("msm_sharded_only", dict(columns=False)), ("msm_and_columns_sharded", dict(columns=True)),
                          ("resident", dict(columns=True, resident=True))):  # resident: columns stay on their owner, slices travel point to point
            try:
                D.shard(ctx, wl.pk, **kw)
            except Exception as e:  # noqa: BLE001
                res[label] = {"error": f"set-up: {type(e).__name__}: {e}"}
                ctx._bench_comm = False
                report()
                continue
            run(label)
        res["parallelism"] = (f"one proof over {world} ranks: MSM point ranges /{world} (window tables for the rank's slices), "
                              "column transforms by owner (broadcast) or resident on their owner (point-to-point slices); " + ("RCCL all-gather of 96-B partials + grouped broadcasts of columns" if D.transport == "rccl"
                                                                else "host-callback transport over gloo (one-GPU rehearsal of the N > 1 path)"))
    wl.close()
    return res


def generic_rng(wl, steps):
    """The headline draws its blinding from the library's own xoshiro256**, which `cq_create_proof` recognises and runs
    inline on eight threads with jump-ahead.  A real caller hands over an opaque `RngCore` callback (prover.rs:65): one
    indirect call per u64 on the library's helper thread.  This leg proves the same instance with such a callback
    (the same generator behind a function pointer the library does not know) -- the time a Rust caller sees -- and
    with the caller's bulk form registered (cq_pk_set_rng_fill)."""
    import torch

    ptrs = [c.ptr for c in wl.cols]
    res = {"note": "same proof bytes as the headline for the same seed (tests/test_prover_gpu.py); "
                   f"{8 * wl.n + 8 * (wl.n - wl.pk.usable_rows + 1) * 2 * wl.pairs} RNG words per proof"}
    for name, fill in (("per_word_callback", None), ("bulk_fill_callback", "opaque")):
        wl.pk.set_rng_fill(fill)
        wl.fill_witness()
        wl.pk.create_proof_dev(ptrs, seed=1, opaque_rng=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            wl.fill_witness()
            wl.pk.create_proof_dev(ptrs, seed=1000 + i, opaque_rng=True)
        torch.cuda.synchronize()
        res[name] = {"ms_per_step": (time.perf_counter() - t0) / steps * 1e3, "steps": steps}
    wl.pk.set_rng_fill(None)
    return res


def cpu_baseline(ctx):
    """CPU leg (rank 0, N=1): the plain-C restatement of the reference's create_proof (oracle/, kind 'port') proving
    the metric's own k = 18 instance once, and the k = 16 instance (BASELINE configs[1]) as `sample_small`; the GPU
    proves the same instances and the proofs are compared byte for byte."""
    from oracle import cbind as OC

    OC.lib(native=True)  # -O3 -march=native, built on this host

    def _cpu_prove(ctx, k, threads):
        """(cpu seconds, gpu seconds, proofs identical, workload) for one create_proof of the bench circuit at 2^k rows."""
        from sha2_on_cq_halo2_amd.api import fr_to_mont
        from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload, small_to_mont, spread16

        n = 1 << k
        wl = ShaCqWorkload(ctx, k, seed=0x5348413243515F)
        wl.prove(seed=77)
        t0 = time.perf_counter()
        gpu_proof = wl.prove(seed=77)
        gpu_s = time.perf_counter() - t0
        g, gl = wl.params.download()
        tl, t0pts = wl.cfg.download()
        N = wl.cfg.size
        idx = np.arange(N)
        tvals = [small_to_mont(idx), small_to_mont(spread16(idx))]
        tqs = [wl.dense.download_qs(), wl.spread.download_qs()]
        advice = [c.download((n, 4)) for c in wl.cols]
        lookups = [[(2 * p, 0), (2 * p + 1, 1)] for p in range(wl.pairs)]
        la = OC.keygen_l_active(k, 5)
        OC.lib().cqo_set_num_threads(threads)
        t0 = time.perf_counter()
        cpu_proof = OC.create_proof(k, 2 * wl.pairs, lookups, tvals, tqs, g, gl, tl, t0pts, g[1:], la, fr_to_mont(0xC0FFEE + k),
                                    advice, 77)
        cpu_s = time.perf_counter() - t0
        res = (cpu_s, gpu_s, cpu_proof == gpu_proof, wl.blocks, wl.msm_scalars_per_proof())
        wl.close()
        return res

    # a 1-GPU job's CPU share on the GPU box is 16 cores (the box exposes more)
    threads = min(OC.lib().cqo_num_threads(), int(os.environ.get("CQ_CPU_BASELINE_THREADS", "16")))

    def entry(k):
        cpu_s, gpu_s, same, blocks, scalars = _cpu_prove(ctx, k, threads)
        return {
            "value": scalars / cpu_s / 1e6,
            "unit": "Mscalar/s",
            "cores": int(threads),
            "kind": "port",
            "k": k,
            "sample": f"one full create_proof of the bench circuit at k={k} ({blocks} SHA blocks, {scalars} MSM scalars): "
            f"C restatement of the reference's algorithms (incl. its serial sparse commits and per-row inversions), "
            f"{cpu_s:.2f} s on {threads} OpenMP threads; the GPU proves the same instance in {gpu_s * 1e3:.1f} ms "
            f"(proof bytes identical: {same})",
            "proof_wall_s": cpu_s,
            "gpu_proof_wall_s_same_sample": gpu_s,
            "speedup_same_sample": cpu_s / gpu_s,
            "proof_bytes_identical": bool(same),
        }

    out = entry(CPU_BASELINE_K)
    try:
        out["sample_small"] = entry(CPU_BASELINE_SMALL_K)
    except Exception as e:  # noqa: BLE001
        out["sample_small"] = {"error": f"{type(e).__name__}: {e}"}
    return out


if __name__ == "__main__":
    main()
