"""GPU parity: HIP NTT (through the C ABI) vs the CPU oracle's restatement of `best_fft`."""
import numpy as np
import pytest

from oracle import bn254 as B
from oracle import poly as OP

pytestmark = pytest.mark.gpu


def _omega(log_n):
    w = B.FR_ROOT_OF_UNITY
    for _ in range(log_n, B.FR_S):
        w = w * w % B.R_MOD
    return w


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13])
def test_best_fft_matches_oracle(ctx, log_n):
    rng = B.Xoshiro256ss(100 + log_n)
    n = 1 << log_n
    vals = [B.fr_random(rng) for _ in range(n)]
    w = _omega(log_n)
    exp = OP.best_fft(list(vals), w, log_n)
    got = ctx.best_fft(B.to_mont_limbs(vals), B.to_mont_limbs([w])[0], log_n)
    assert np.array_equal(got, B.to_mont_limbs(exp))


def test_best_fft_inverse_roundtrip_large(ctx):
    """Size-independent property at a BASELINE size (k=18): iNTT(NTT(a)) * n^-1 == a."""
    log_n = 18
    n = 1 << log_n
    rs = np.random.RandomState(5)
    # random canonical-looking residues < 2^253 (valid Montgomery residues since < r)
    a = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    w = _omega(log_n)
    winv = pow(w, B.R_MOD - 2, B.R_MOD)
    f = ctx.best_fft(a, B.to_mont_limbs([w])[0], log_n)
    b = ctx.best_fft(f, B.to_mont_limbs([winv])[0], log_n)
    # compare on a sample (python big-int conversion of 2^18 elements is slow)
    idx = rs.randint(0, n, size=64)
    ninv = pow(n, B.R_MOD - 2, B.R_MOD)
    got = [v * ninv % B.R_MOD for v in B.from_mont_limbs(b[idx])]
    exp = B.from_mont_limbs(a[idx])
    assert got == exp


def test_best_fft_len_mismatch(ctx):
    from sha2_on_cq_halo2_amd import CqError

    with pytest.raises(CqError):
        ctx.best_fft(np.zeros((3, 4), dtype=np.uint64), np.zeros(4, dtype=np.uint64), 2)


def test_best_fft_two_level_twiddles(monkeypatch):
    """Domains above 2^24 have no full twiddle table: the inter-pass twiddle is composed from two small tables (and
    brought back to the canonical range the products' bounds assume).  Forced here at a small size, on a context of
    its own so that its twiddle cache is built under the switch; against the C oracle's best_fft."""
    from oracle import cbind as OC
    from sha2_on_cq_halo2_amd import Context

    monkeypatch.setenv("CQ_NTT_NO_FULL_TABLE", "1")
    c2 = Context(0)
    try:
        for log_n in (9, 13, 14):
            rs = np.random.RandomState(log_n)
            a = rs.randint(0, 2**63, size=(1 << log_n, 4), dtype=np.int64).astype(np.uint64)
            a[:, 3] &= np.uint64((1 << 60) - 1)
            w = B.to_mont_limbs([_omega(log_n)])[0]
            assert np.array_equal(c2.best_fft(a, w, log_n), OC.best_fft(a, w, log_n))
    finally:
        c2.close()


@pytest.mark.parametrize("log_n", [18, 19, 20])
def test_best_fft_full_size_matches_c_oracle(ctx, log_n):
    """The BASELINE sizes themselves, element for element against the C restatement of `best_fft`
    (arithmetic.rs:171-274): 2^18 and 2^19 take three Stockham passes (k = 18 and its extended domain), 2^20 four."""
    from oracle import cbind as OC

    rs = np.random.RandomState(1000 + log_n)
    a = rs.randint(0, 2**63, size=(1 << log_n, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    a[::7919] = 0  # zeros and a maximal element among the inputs
    a[1] = B.to_mont_limbs([B.R_MOD - 1])[0]
    w = B.to_mont_limbs([_omega(log_n)])[0]
    assert np.array_equal(ctx.best_fft(a, w, log_n), OC.best_fft(a, w, log_n))


_VARIANT_WORKER = r'''
import hashlib, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from oracle import bn254 as B
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.api import EvaluationDomain
def omega(log_n):
    w = B.FR_ROOT_OF_UNITY
    for _ in range(log_n, B.FR_S):
        w = w * w %% B.R_MOD
    return w
ctx = Context(0)
h = hashlib.sha256()
for log_n in (13, 14, 16, 17, 18, 19, 20):
    rs = np.random.RandomState(500 + log_n)
    a = rs.randint(0, 2**63, size=(1 << log_n, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    h.update(ctx.best_fft(a, B.to_mont_limbs([omega(log_n)])[0], log_n).tobytes())
dom = EvaluationDomain(ctx, 3, 15)            # coset shift on the way in, zero padding, truncation and output factors
rs = np.random.RandomState(9)
a = rs.randint(0, 2**63, size=(1 << 15, 4), dtype=np.int64).astype(np.uint64); a[:, 3] &= np.uint64((1 << 60) - 1)
co = dom.lagrange_to_coeff(a); ext = dom.coeff_to_extended(co); back = dom.extended_to_coeff(ext)
for x in (co, ext, back): h.update(np.ascontiguousarray(x).tobytes())
print("digest", h.hexdigest())
'''


def test_kernel_variants_give_identical_transforms(tmp_path):
    """The pass-kernel variants kept behind switches -- wide passes (CQ_NTT_WIDE=1: 7..9 bits per pass, swizzled LDS image,
    mid-pass reduction) and the software-pipelined persistent kernel (CQ_NTT_PIPE=1) -- produce bit for bit the transforms of
    the default plan, at 2^13..2^20 and through the EvaluationDomain wrappers (each variant in a process of its own: the
    switches are read once)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "ntt_variant.py"
    script.write_text(_VARIANT_WORKER % {"root": root})
    digests = {}
    for name, env in (("default", {}), ("wide", {"CQ_NTT_WIDE": "1"}), ("pipe", {"CQ_NTT_PIPE": "1"})):
        e = {k: v for k, v in os.environ.items() if k not in ("CQ_NTT_WIDE", "CQ_NTT_PIPE")}
        e.update(env)
        r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        digests[name] = r.stdout.strip().split()[-1]
    assert digests["wide"] == digests["default"] and digests["pipe"] == digests["default"], digests
