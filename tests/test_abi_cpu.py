"""CPU: the C-ABI library loads and exports every symbol include/cq_halo2.h declares (no compute
calls without a GPU), and contexts fail loudly without a device."""
import ctypes as C

from sha2_on_cq_halo2_amd import header_symbols, load


def test_every_declared_symbol_is_exported():
    lib = load()
    syms = header_symbols()
    assert len(syms) > 50
    missing = [s for s in syms if not hasattr(lib, s)]
    assert missing == []


def test_version_and_host_only_entry_points():
    lib = load()
    assert b"gfx950" in lib.cq_version()
    # host-only helpers work without a GPU
    st = (C.c_uint64 * 4)()
    lib.cq_xoshiro256ss_seed(7, st)
    from oracle.bn254 import Xoshiro256ss

    ref = Xoshiro256ss(7)
    assert list(st) == ref.s
    assert [lib.cq_xoshiro256ss_next_u64(st) for _ in range(5)] == [ref.next_u64() for _ in range(5)]


def test_g1_sum_host_helper_matches_oracle():
    import numpy as np

    from oracle import bn254 as B
    from tests.util import random_points

    lib = load()
    pts = random_points(5, 3)
    jac = np.zeros((5, 12), dtype=np.uint64)
    jac[:, :8] = B.points_to_mont_limbs(pts)
    jac[:, 8:] = B.to_mont_limbs([1], B.Q_MOD)[0]
    out = np.zeros(12, dtype=np.uint64)
    assert lib.cq_g1_sum(jac.ctypes.data, 5, out.ctypes.data) == 0
    acc = None
    for p in pts:
        acc = B.g1_add(acc, p)
    aff = np.zeros(8, dtype=np.uint64)
    assert lib.cq_g1_to_affine(out.ctypes.data, aff.ctypes.data) == 0
    assert B.points_from_mont_limbs(aff.reshape(1, 8))[0] == acc


def test_no_gpu_is_an_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        return
    from sha2_on_cq_halo2_amd import Context, CqError

    try:
        Context(0)
        assert False, "Context() must fail without a GPU"
    except CqError as e:
        assert e.code == -3


def test_product_and_tools_never_import_the_oracle():
    """oracle/ is the checker: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    offenders = []
    for d in ("sha2_on_cq_halo2_amd", "tools"):
        for base, _, files in os.walk(os.path.join(root, d)):
            for f in files:
                if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(base, f)).read(), re.M):
                    offenders.append(os.path.join(base, f))
    assert offenders == []
    # bench.py: the only oracle import sits inside cpu_baseline()
    src = open(os.path.join(root, "bench.py")).read()
    for m in re.finditer(r"^\s*(from|import)\s+oracle\b", src, re.M):
        head = src[:m.start()]
        assert head.rfind("def cpu_baseline") > max(head.rfind("\ndef " + n) for n in ("main", "plonk_variant", "two_in_flight", "ntt_standalone"))


def test_xoshiro_parallel_fill_is_the_same_stream():
    """cq_xoshiro256ss_fill (jump-ahead, several threads) yields exactly the words and the final state of repeated
    cq_xoshiro256ss_next_u64 -- for lengths around the per-thread threshold, odd lengths, and one to eight threads."""
    import numpy as np

    lib = load()
    for seed, count, threads in [(1, 10, 8), (2, 70000, 1), (3, 131072, 2), (4, 200001, 3), (5, 1 << 20, 8), (6, (1 << 20) + 12345, 7)]:
        st_a = (C.c_uint64 * 4)()
        st_b = (C.c_uint64 * 4)()
        lib.cq_xoshiro256ss_seed(seed, st_a)
        lib.cq_xoshiro256ss_seed(seed, st_b)
        got = np.zeros(count, dtype=np.uint64)
        lib.cq_xoshiro256ss_fill(st_a, got.ctypes.data, count, threads)
        exp = np.zeros(count, dtype=np.uint64)
        lib.cq_xoshiro256ss_fill(st_b, exp.ctypes.data, count, 1)  # serial path
        assert np.array_equal(got, exp) and list(st_a) == list(st_b), (seed, count, threads)
        # and the serial path is the per-word generator
        st_c = (C.c_uint64 * 4)()
        lib.cq_xoshiro256ss_seed(seed, st_c)
        head = [lib.cq_xoshiro256ss_next_u64(st_c) for _ in range(min(count, 50))]
        assert head == [int(x) for x in exp[: len(head)]]


def test_hot_kernels_keep_their_register_budget_and_do_not_spill():
    """Reads the gfx950 code objects inside the built library (tools/code_object_audit.py: offload bundles of .hip_fatbin,
    AMDGPU metadata notes): the kernels a proof's time is made of must not use scratch memory, and must stay within the
    register budget their occupancy was tuned at (accumulate and combine: three waves per SIMD; an NTT pass: four, the
    same four its 37 KB of LDS allow).  A `#pragma unroll` that silently gives up (compiler traps) or
    one more live value in a loop shows up here as scratch bytes or a lost wave, not as a timing weeks later."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    load()  # builds the library if it is not there
    spec = importlib.util.spec_from_file_location("code_object_audit", os.path.join(root, "tools", "code_object_audit.py"))
    audit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(audit)
    ks = audit.kernels()
    assert len(ks) > 100
    # interpreters / generic fallbacks that index private arrays at run time by design
    allowed = ("gate_eval_kernel", "_ZN2cq17scan_apply_kernel", "msm_part_scatter_kernelILj0ELj0E")
    spilling = sorted(n for n, k in ks.items() if k["scratch"] and not any(a in n for a in allowed))
    assert not spilling, spilling

    def some(sub):
        hits = [k for n, k in ks.items() if sub in n]
        assert hits, sub
        return hits

    for k in some("msm_accumulate_kernel") + some("msm_combine_level_kernel"):
        assert k["scratch"] == 0 and audit.waves_per_simd(k["vgpr"]) >= 3, k
    for k in some("ntt_pass_kernelILj6ELj4E") + some("ntt_pass_kernelILj5ELj5E") + some("ntt_pass_kernelILj4ELj6E"):
        assert k["scratch"] == 0 and audit.waves_per_simd(k["vgpr"]) >= 4, k
    for k in some("batch_invert_kernel") + some("msm_rowcol") + some("msm_weighted") + some("cq_quotient_kernel") + some("lincomb_kernel") + some("block_eval_kernel"):
        assert k["scratch"] == 0, k


def test_rust_ffi_module_is_generated_from_the_header_and_complete():
    """include/cq_halo2_sys.rs (the reference-side binding) is derived from the header by
    tools/gen_rust_ffi.py: it must be up to date and declare every function, struct, callback and constant."""
    import os
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_rust_ffi.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    rs = open(os.path.join(root, "include", "cq_halo2_sys.rs")).read()
    declared = set(re.findall(r"pub fn (cq_\w+)\(", rs))
    assert declared == set(header_symbols())
    header = open(os.path.join(root, "include", "cq_halo2.h")).read()
    for name in re.findall(r"^#define\s+(CQ_[A-Z0-9_]+)\s+\(?-?\d", header, flags=re.M):
        assert f"pub const {name}:" in rs, name
    for c_name in re.findall(r"\}\s*(cq_\w+)\s*;", header) + re.findall(r"typedef struct (cq_\w+) \1;", header):
        assert "pub struct " + "".join(w.capitalize() for w in c_name.split("_")) + " " in rs, c_name
    assert rs.count("{") == rs.count("}") and rs.count("(") == rs.count(")")
    # the layouts the .rs promises are the C ones: field count and order of every struct follow the header
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(cq_\w+)\s*;", re.sub(r"/\*.*?\*/", "", header, flags=re.S), flags=re.S):
        c_fields = [re.findall(r"(\w+)\s*(?:\[\w*\])?\s*$", f.strip())[0] for f in m.group(1).split(";") if f.strip()]
        rname = "".join(w.capitalize() for w in m.group(2).split("_"))
        body = re.search(r"pub struct " + rname + r" \{(.*?)\n\}", rs, flags=re.S).group(1)
        assert re.findall(r"pub (\w+):", body) == c_fields, rname
