"""TEST INFRASTRUCTURE ONLY -- byte layout of `ProvingKey::write` in `SerdeFormat::RawBytes`.

Restates `plonk.rs:349-362` (ProvingKey), `plonk.rs:92-113` (VerifyingKey), `permutation.rs:109-148`
(permutation keys), `poly.rs:163-170` (Polynomial), `helpers.rs:98-140` (bit packing, polynomial slices) and the
raw encodings of `derive/field.rs` / `derive/curve.rs` `write_raw` (four little-endian u64 Montgomery limbs per
field element; x || y per affine point).  Parity status: no serialized key ships with the reference, so the
layout is "parity unpinned" beyond what the cited code fixes.
"""
from __future__ import annotations

import struct

from .bn254 import Q_MOD, R_MOD


def _fr_raw(v: int) -> bytes:
    return ((v % R_MOD) * (1 << 256) % R_MOD).to_bytes(32, "little")


def _fq_raw(v: int) -> bytes:
    return ((v % Q_MOD) * (1 << 256) % Q_MOD).to_bytes(32, "little")


def _point_raw(pt) -> bytes:
    if pt is None:
        return bytes(64)
    return _fq_raw(pt[0]) + _fq_raw(pt[1])


def _poly(values) -> bytes:
    return struct.pack(">I", len(values)) + b"".join(_fr_raw(v) for v in values)


def _slice(polys) -> bytes:
    return struct.pack(">I", len(polys)) + b"".join(_poly(p) for p in polys)


def pack_selectors(selectors, n: int) -> bytes:
    """`pack` over chunks of 8 booleans, least significant bit first (helpers.rs:98-105, plonk.rs:103-110)."""
    out = bytearray()
    for sel in selectors:
        assert len(sel) == n
        for i in range(0, n, 8):
            b = 0
            for j, bit in enumerate(sel[i:i + 8]):
                b |= (1 if bit else 0) << j
            out.append(b)
    return bytes(out)


def proving_key_to_bytes(pk, fixed_commitments, perm_commitments, selectors=()) -> bytes:
    """pk: oracle/cq_prover.ProvingKey."""
    n = pk.domain.n
    out = struct.pack(">I", pk.circuit.k) + struct.pack(">I", len(fixed_commitments))
    out += b"".join(_point_raw(c) for c in fixed_commitments)
    out += b"".join(_point_raw(c) for c in perm_commitments)
    out += pack_selectors(selectors, n)
    out += _poly(pk.l0) + _poly(pk.l_last) + _poly(pk.l_active_row)
    out += _slice(pk.fixed_values) + _slice(pk.fixed_polys) + _slice(pk.fixed_cosets)
    out += _slice(pk.permutations) + _slice(pk.perm_polys) + _slice(pk.perm_cosets)
    return out
