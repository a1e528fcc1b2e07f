"""A compiled C caller of include/cq_halo2.h (tests/abi/consumer.c, gcc -std=c11 -Wall -Wextra -Werror -pedantic, linked
against libcq_halo2.so): the header is usable from plain C as written, the struct layouts the ctypes binding assumes are the
compiler's, and -- on the GPU -- what the C caller gets from best_fft / best_multiexp / commit_lagrange / create_proof
(host advice columns, as `WitnessCollection` hands them over, plonk/prover.rs:223-260) equals the oracle's results."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "abi", "consumer.c")
OUT_DIR = os.path.join(ROOT, "tests", "abi", "build")
EXE = os.path.join(OUT_DIR, "consumer")
LIBDIR = os.path.join(ROOT, "sha2_on_cq_halo2_amd")


def _build():
    os.makedirs(OUT_DIR, exist_ok=True)
    lib = os.path.join(LIBDIR, "libcq_halo2.so")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(SRC), os.path.getmtime(lib),
                                                            os.path.getmtime(os.path.join(ROOT, "include", "cq_halo2.h"))):
        return EXE
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
           "-L", LIBDIR, "-lcq_halo2", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, "the header / consumer does not compile as strict C11:\n" + r.stderr
    return EXE


def _kv(text):
    return dict(line.split(" ", 1) for line in text.strip().splitlines() if " " in line)


def test_header_is_plain_c_and_struct_layouts_match_the_binding():
    from sha2_on_cq_halo2_amd.api import _BufferRng, _CqCircuit, _CqPlonk

    r = subprocess.run([_build(), "layout"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    kv = _kv(r.stdout)
    for cname, cls in (("cq_plonk", _CqPlonk), ("cq_circuit", _CqCircuit), ("cq_buffer_rng", _BufferRng)):
        assert int(kv["sizeof." + cname]) == C.sizeof(cls), cname
        fields = [k.split(".", 1)[1] for k in kv if k.startswith(cname + ".")]
        assert fields == [f[0] for f in cls._fields_], cname  # same members, same order
        for f in fields:
            assert int(kv[cname + "." + f]) == getattr(cls, f).offset, (cname, f)
    assert "gfx950" in kv["version"]


def test_binding_signatures_are_derived_from_the_header():
    """sha2_on_cq_halo2_amd/_lib.py parses the header for every prototype (no hand-written argtypes to drift): each is
    parsed, understood, exported by the library, and nothing the library exports under cq_* is missing from the header."""
    import re

    from sha2_on_cq_halo2_amd import _lib

    protos = _lib.header_prototypes()
    text = _lib._header_text()
    text = re.sub(r"typedef[^;{]*;", "", text)
    declared = set(re.findall(r"\b(cq_[a-z0-9_]+)\s*\(", text))
    assert declared == {p.name for p in protos} and len(protos) == len(declared) > 100
    sig = _lib.signatures()
    lib = _lib.load()
    for p in protos:
        fn = getattr(lib, p.name)
        assert fn.argtypes == sig[p.name][1] and len(fn.argtypes) == len(p.params)
        assert fn.restype == sig[p.name][0]
    assert lib.cq_version.restype is C.c_char_p and lib.cq_ctx_destroy.restype is None and lib.cq_pk_proof_size.restype is C.c_size_t
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(LIBDIR, "libcq_halo2.so")], capture_output=True, text=True)
    exported = {line.split()[-1] for line in nm.stdout.splitlines() if line.split() and line.split()[-1].startswith("cq_")}
    assert exported - declared == set(), "exported but not declared in include/cq_halo2.h"


@pytest.mark.gpu
def test_c_consumer_results_equal_the_oracle(ctx):
    from oracle import cbind as OC
    from sha2_on_cq_halo2_amd import ParamsKZG, StaticTable, TableConfig
    from sha2_on_cq_halo2_amd.api import domain_omega

    k, seed, N = 9, 20260105, 64
    n = 1 << k
    r = subprocess.run([_build(), "run", str(k), str(seed)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("done"), r.stdout[-2000:] + r.stderr[-2000:]
    kv = _kv(r.stdout)

    # the same inputs, drawn the way the consumer draws them (cq_xoshiro256ss_*, top limb masked to 60 bits)
    lib = ctx.lib
    st = (C.c_uint64 * 4)()
    lib.cq_xoshiro256ss_seed(seed, st)

    def draw(count):
        a = np.array([lib.cq_xoshiro256ss_next_u64(st) for _ in range(4 * count)], dtype=np.uint64).reshape(count, 4)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a

    s = draw(1)[0]
    assert kv["s"].split() == ["%016x" % int(v) for v in s]
    scalars = draw(n)
    params = ParamsKZG.setup_from_toxic_waste(ctx, k, s)
    g, gl = params.download()

    def fnv(arr):
        h = 0xCBF29CE484222325
        for b in np.ascontiguousarray(arr).view(np.uint8).reshape(-1).tolist():
            h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
        return h

    fft = OC.best_fft(scalars, domain_omega(k), k)
    assert int(kv["best_fft_fnv"], 16) == fnv(fft)
    assert kv["best_fft_first"].split() == ["%016x" % int(v) for v in fft[0]]
    msm = OC.g1_to_affine(OC.best_multiexp(scalars, gl))
    assert kv["best_multiexp"].split() == kv["commit_lagrange"].split() == ["%016x" % int(v) for v in msm.reshape(-1)]

    # the proof: the same key built through the Python binding feeds the C restatement of the reference prover
    from sha2_on_cq_halo2_amd.sha_circuit import small_to_mont, spread16

    cfg = TableConfig.setup_from_toxic_waste(ctx, N, s)
    idx = np.arange(N)
    tvals = [small_to_mont(idx), small_to_mont(spread16(idx))]
    tabs = [StaticTable.setup_from_toxic_waste(ctx, v, s) for v in tvals]
    tl, t0 = cfg.download()
    vk = np.array([0xC0FFEE, 0, 0, 0], dtype=np.uint64)
    bf = 5
    usable = n - (bf + 1)
    rows = (np.arange(usable) * 7 + 3) % N
    advice = [np.zeros((n, 4), dtype=np.uint64), np.zeros((n, 4), dtype=np.uint64)]
    advice[0][:usable] = tvals[0][rows]
    advice[1][:usable] = tvals[1][rows]
    expect = OC.create_proof(k, 2, [[(0, 0), (1, 1)]], tvals, [t.download_qs() for t in tabs], g, gl, tl, t0, g[1:],
                             OC.keygen_l_active(k, bf), vk, advice, seed + 1)
    assert kv["proof"] == expect.hex()
    assert int(kv["short_buffer_rc"]) == -1 and int(kv["lookup_miss_rc"]) == -4  # CQ_ERR_ARG, CQ_ERR_LOOKUP
    for o in tabs + [cfg, params]:
        o.close()
