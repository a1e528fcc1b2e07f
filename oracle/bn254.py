"""TEST INFRASTRUCTURE ONLY -- CPU oracle for BN254 field/curve arithmetic.

This is a big-integer restatement of the arithmetic the reference
(aleph-zero-foundation/sha2-on-cq-halo2, a fork of PSE halo2 + halo2curves 0.3.1)
uses on its CQ / KZG proving path.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import anything under `oracle/`; the product
(`sha2_on_cq_halo2_amd/`) never does.

Parity status: PINNED by the reference's in-tree known-answer tests
(see `tests/test_oracle_kat.py`): `from_u512` KATs (bn256/fr.rs:347-367,
bn256/fq.rs:331-351), every field constant (bn256/fr.rs:29-118,
bn256/fq.rs:29-91), generator / curve constant (bn256/curve.rs:66-68).
The reference itself is Rust and cannot be built in this environment (no
cargo/rustc), so end-to-end proof bytes are pinned only through these KATs
plus the algebraic identities the reference's own tests use.

All values handled here are *canonical* integers (0 <= a < p); the Montgomery
limb layout the reference keeps in memory (`[u64;4]` of a*2^256 mod p,
derive/field.rs) is produced/consumed by `to_mont_limbs` / `from_mont_limbs`.
"""
from __future__ import annotations

import numpy as np

MASK64 = (1 << 64) - 1


def _limbs(l):
    return l[0] | (l[1] << 64) | (l[2] << 128) | (l[3] << 192)


# ---------------------------------------------------------------------------
# Fr -- BN254 scalar field (bn256/fr.rs:25-118)
# ---------------------------------------------------------------------------
R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FR_INV = 0xC2E1F593EFFFFFFF  # -(r^-1) mod 2^64, fr.rs:39
FR_R = (1 << 256) % R_MOD
FR_R2 = (1 << 512) % R_MOD
FR_R3 = (1 << 768) % R_MOD
FR_S = 28  # fr.rs:72
FR_GENERATOR = 7  # fr.rs:70
# fr.rs:78-83 (from_raw => canonical value)
FR_ROOT_OF_UNITY = _limbs(
    [0xD34F1ED960C37C9C, 0x3215CF6DD39329C8, 0x98865EA93DD31F74, 0x03DDB9F5166D18B7]
)
FR_TWO_INV = _limbs(
    [0xA1F0FAC9F8000001, 0x9419F4243CDCB848, 0xDC2822DB40C0AC2E, 0x183227397098D014]
)
FR_ROOT_OF_UNITY_INV = _limbs(
    [0x0ED3E50A414E6DBA, 0xB22625F59115ABA7, 0x1BBE587180F34361, 0x048127174DAABC26]
)
FR_DELTA = _limbs(
    [0x870E56BBE533E9A2, 0x5B5F898E5E963F25, 0x64EC26AAD4C86E71, 0x09226B6E22C6F0CA]
)
FR_ZETA = _limbs(
    [0xB8CA0B2D36636F23, 0xCC37A73FEC2BC5E9, 0x048B6E193FD84104, 0x30644E72E131A029]
)

# ---------------------------------------------------------------------------
# Fq -- BN254 base field (bn256/fq.rs:25-91)
# ---------------------------------------------------------------------------
Q_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
FQ_INV = 0x87D20782E4866389
FQ_R = (1 << 256) % Q_MOD
FQ_R2 = (1 << 512) % Q_MOD
FQ_R3 = (1 << 768) % Q_MOD
FQ_ZETA = _limbs([0x5763473177FFFFFE, 0xD4F263F1ACDB5C4F, 0x59E26BCEA0D48BAC, 0x0])

G1_GEN = (1, 2)  # bn256/curve.rs:66-67
G1_B = 3  # bn256/curve.rs:68


def inv_mod(a: int, p: int) -> int:
    """Field inversion (reference: Fermat, fr.rs:200-209); zero has no inverse."""
    if a % p == 0:
        raise ZeroDivisionError("inverse of zero")
    return pow(a, p - 2, p)


def from_u512(limbs8, p: int) -> int:
    """derive/field.rs:29-47: d0*R2 + d1*R3 in Montgomery form == (d0 + 2^256 d1) mod p."""
    d0 = _limbs(limbs8[0:4])
    d1 = _limbs(limbs8[4:8])
    return (d0 + (d1 << 256)) % p


def from_bytes_wide(b: bytes, p: int = R_MOD) -> int:
    """`from_bytes_wide` used by Challenge255 (transcript.rs:300-309): LE 512-bit mod p."""
    assert len(b) == 64
    return int.from_bytes(b, "little") % p


def to_repr(a: int) -> bytes:
    """`to_repr`: canonical 32-byte little-endian (fr.rs:245-261)."""
    return int(a).to_bytes(32, "little")


# -- Montgomery limb I/O ----------------------------------------------------
def to_mont_limbs(vals, p: int = R_MOD) -> np.ndarray:
    """canonical ints -> uint64[n,4] of a*2^256 mod p (the reference's in-memory form)."""
    R = (1 << 256) % p
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = (int(v) * R) % p
        out[i, 0] = m & MASK64
        out[i, 1] = (m >> 64) & MASK64
        out[i, 2] = (m >> 128) & MASK64
        out[i, 3] = (m >> 192) & MASK64
    return out


def from_mont_limbs(arr: np.ndarray, p: int = R_MOD):
    """uint64[n,4] Montgomery limbs -> list of canonical ints."""
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    rinv = inv_mod((1 << 256) % p, p)
    out = []
    for row in arr:
        m = int(row[0]) | (int(row[1]) << 64) | (int(row[2]) << 128) | (int(row[3]) << 192)
        assert m < p, "non-canonical Montgomery residue"
        out.append((m * rinv) % p)
    return out


def points_to_mont_limbs(pts) -> np.ndarray:
    """affine points -> uint64[n,8] (x limbs, y limbs), identity = all zero
    (derive/curve.rs:453-463: identity is (0,0))."""
    xs = [0 if P is None else P[0] for P in pts]
    ys = [0 if P is None else P[1] for P in pts]
    return np.concatenate([to_mont_limbs(xs, Q_MOD), to_mont_limbs(ys, Q_MOD)], axis=1)


def points_from_mont_limbs(arr: np.ndarray):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 8)
    xs = from_mont_limbs(arr[:, 0:4], Q_MOD)
    ys = from_mont_limbs(arr[:, 4:8], Q_MOD)
    return [None if (x == 0 and y == 0) else (x, y) for x, y in zip(xs, ys)]


def jac_from_mont_limbs(arr: np.ndarray):
    """uint64[n,12] Jacobian (x,y,z) Montgomery limbs -> affine points."""
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 12)
    xs = from_mont_limbs(arr[:, 0:4], Q_MOD)
    ys = from_mont_limbs(arr[:, 4:8], Q_MOD)
    zs = from_mont_limbs(arr[:, 8:12], Q_MOD)
    return [jac_to_affine((x, y, z)) for x, y, z in zip(xs, ys, zs)]


# ---------------------------------------------------------------------------
# G1: y^2 = x^3 + 3 over Fq.  Affine = (x, y) or None; Jacobian = (X, Y, Z), Z=0 identity.
# Formulas are textbook; the reference's outputs are canonical affine encodings
# (derive/curve.rs:399-412, 635-646) so any correct group law is bit-exact.
# ---------------------------------------------------------------------------
JAC_ID = (0, 0, 0)


def is_on_curve(P) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - x * x * x - G1_B) % Q_MOD == 0


def jac_double(P):
    X, Y, Z = P
    if Z == 0 or Y == 0:
        return JAC_ID
    p = Q_MOD
    A = X * X % p
    B = Y * Y % p
    C = B * B % p
    D = 2 * ((X + B) * (X + B) - A - C) % p
    E = 3 * A % p
    F = E * E % p
    X3 = (F - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y * Z % p
    return (X3, Y3, Z3)


def jac_add(P, Q):
    p = Q_MOD
    X1, Y1, Z1 = P
    X2, Y2, Z2 = Q
    if Z1 == 0:
        return Q
    if Z2 == 0:
        return P
    Z1Z1 = Z1 * Z1 % p
    Z2Z2 = Z2 * Z2 % p
    U1 = X1 * Z2Z2 % p
    U2 = X2 * Z1Z1 % p
    S1 = Y1 * Z2 * Z2Z2 % p
    S2 = Y2 * Z1 * Z1Z1 % p
    if U1 == U2:
        if S1 == S2:
            return jac_double(P)
        return JAC_ID
    H = (U2 - U1) % p
    Rr = (S2 - S1) % p
    HH = H * H % p
    HHH = H * HH % p
    V = U1 * HH % p
    X3 = (Rr * Rr - HHH - 2 * V) % p
    Y3 = (Rr * (V - X3) - S1 * HHH) % p
    Z3 = Z1 * Z2 * H % p
    return (X3, Y3, Z3)


def jac_add_affine(P, A):
    if A is None:
        return P
    return jac_add(P, (A[0], A[1], 1))


def jac_neg(P):
    return (P[0], (-P[1]) % Q_MOD, P[2])


def to_jac(A):
    return JAC_ID if A is None else (A[0], A[1], 1)


def jac_to_affine(P):
    """derive/curve.rs:399-412."""
    X, Y, Z = P
    if Z == 0:
        return None
    zi = inv_mod(Z, Q_MOD)
    zi2 = zi * zi % Q_MOD
    return (X * zi2 % Q_MOD, Y * zi2 * zi % Q_MOD)


def g1_add(A, B):
    return jac_to_affine(jac_add(to_jac(A), to_jac(B)))


def g1_neg(A):
    return None if A is None else (A[0], (-A[1]) % Q_MOD)


def jac_mul(P, k: int):
    """scalar mul (reference: MSB-first double-and-add, derive/curve.rs:914-935)."""
    k %= R_MOD
    acc = JAC_ID
    for bit in bin(k)[2:] if k else "":
        acc = jac_double(acc)
        if bit == "1":
            acc = jac_add(acc, P)
    return acc


def g1_mul(A, k: int):
    return jac_to_affine(jac_mul(to_jac(A), k))


def g1_to_bytes(A) -> bytes:
    """Compressed encoding, derive/curve.rs:635-646: LE x with sign(y) in bit 7 of byte 31;
    identity = 32 zero bytes."""
    if A is None:
        return bytes(32)
    x, y = A
    b = bytearray(x.to_bytes(32, "little"))
    b[31] |= (y & 1) << 7
    return bytes(b)


def batch_to_affine(jacs):
    """Montgomery-trick normalisation (derive/curve.rs:362-397), identities skipped."""
    p = Q_MOD
    pref = []
    acc = 1
    for (_, _, Z) in jacs:
        pref.append(acc)
        if Z != 0:
            acc = acc * Z % p
    acc = inv_mod(acc, p)
    out = [None] * len(jacs)
    for i in range(len(jacs) - 1, -1, -1):
        X, Y, Z = jacs[i]
        if Z == 0:
            continue
        zi = pref[i] * acc % p
        acc = acc * Z % p
        zi2 = zi * zi % p
        out[i] = (X * zi2 % p, Y * zi2 * zi % p)
    return out


# ---------------------------------------------------------------------------
# Deterministic RNG used by the parity harness.  The reference's `create_proof`
# takes any `RngCore` (plonk/prover.rs:65); its tests use OsRng, so no golden
# stream exists.  We fix xoshiro256** (seeded through splitmix64) as *our*
# harness RNG -- the oracle and the product consume the same u64 stream.
# ---------------------------------------------------------------------------
class Xoshiro256ss:
    def __init__(self, seed: int):
        s = []
        z = seed & MASK64
        for _ in range(4):
            z = (z + 0x9E3779B97F4A7C15) & MASK64
            x = z
            x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
            x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK64
            x = x ^ (x >> 31)
            s.append(x)
        self.s = s

    @staticmethod
    def _rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & MASK64

    def next_u64(self) -> int:
        s = self.s
        result = (self._rotl((s[1] * 5) & MASK64, 7) * 9) & MASK64
        t = (s[1] << 17) & MASK64
        s[2] ^= s[0]
        s[3] ^= s[1]
        s[1] ^= s[2]
        s[0] ^= s[3]
        s[2] ^= t
        s[3] = self._rotl(s[3], 45)
        return result

    def words(self, n: int) -> np.ndarray:
        return np.array([self.next_u64() for _ in range(n)], dtype=np.uint64)


class BufferRng:
    """Replays a pre-drawn u64 stream (what the C ABI's rng callback sees)."""

    def __init__(self, words):
        self.words = [int(w) for w in words]
        self.pos = 0

    def next_u64(self) -> int:
        w = self.words[self.pos]
        self.pos += 1
        return w


def fr_random(rng) -> int:
    """`Fr::random`: 8 x next_u64, low limb first, wide-reduced (fr.rs:159-170)."""
    return from_u512([rng.next_u64() for _ in range(8)], R_MOD)
