"""The synthetic "SHA-shaped CQ circuit" the bench and the large-k tests prove.

The reference has no SHA-256 circuit yet (sha/src/main.rs only holds a toy circuit and the table
generators of sha/src/tables.rs), so the k=14..22 "N-block SHA256" configurations are defined HERE:

  * witness: the 32-bit words of a real SHA-256 compression trace -- per round t the working
    variables a_t, e_t, the schedule word W_t, Sigma0(a), Sigma1(e), Maj(a,b,c), Ch(e,f,g) and T1 --
    for `blocks` chained blocks of the message bytes (i mod 251);
  * each word is split into 12/10/10-bit limbs (LongLimbs, sha/src/tables.rs:70-75,135-154) and every
    limb is placed, with its bit-spread form, in one of `pairs` (dense, spread) advice column pairs;
  * constraint system: 2*pairs advice columns, one width-2 static (CQ) lookup per pair against the
    tables dense = [0, 4096) and spread = bit-spread([0, 4096)) -- both columns unique-valued, as the
    reference's vector lookup requires (plonk/static_lookup.rs:82-85); no gates, so degree 3 and
    extended domain 2n (static_lookup.rs:181-190).
"""
from __future__ import annotations

import struct

import numpy as np

from .api import Context, ParamsKZG, ProvingKey, StaticTable, TableConfig, fr_to_mont

_K = [
    0x428A2F98, 0x71374491, 0xB5C0FBCF, 0xE9B5DBA5, 0x3956C25B, 0x59F111F1, 0x923F82A4, 0xAB1C5ED5,
    0xD807AA98, 0x12835B01, 0x243185BE, 0x550C7DC3, 0x72BE5D74, 0x80DEB1FE, 0x9BDC06A7, 0xC19BF174,
    0xE49B69C1, 0xEFBE4786, 0x0FC19DC6, 0x240CA1CC, 0x2DE92C6F, 0x4A7484AA, 0x5CB0A9DC, 0x76F988DA,
    0x983E5152, 0xA831C66D, 0xB00327C8, 0xBF597FC7, 0xC6E00BF3, 0xD5A79147, 0x06CA6351, 0x14292967,
    0x27B70A85, 0x2E1B2138, 0x4D2C6DFC, 0x53380D13, 0x650A7354, 0x766A0ABB, 0x81C2C92E, 0x92722C85,
    0xA2BFE8A1, 0xA81A664B, 0xC24B8B70, 0xC76C51A3, 0xD192E819, 0xD6990624, 0xF40E3585, 0x106AA070,
    0x19A4C116, 0x1E376C08, 0x2748774C, 0x34B0BCB5, 0x391C0CB3, 0x4ED8AA4A, 0x5B9CCA4F, 0x682E6FF3,
    0x748F82EE, 0x78A5636F, 0x84C87814, 0x8CC70208, 0x90BEFFFA, 0xA4506CEB, 0xBEF9A3F7, 0xC67178F2,
]
_H0 = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
_M = 0xFFFFFFFF


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & _M


def sha256_trace_words(blocks: int) -> np.ndarray:
    """uint32 words of the compression trace of `blocks` chained 64-byte blocks (bytes i mod 251):
    8 words per round, 64 rounds per block."""
    msg = bytes(i % 251 for i in range(64 * blocks))
    h = list(_H0)
    out = []
    for b in range(blocks):
        w = list(struct.unpack(">16I", msg[64 * b: 64 * b + 64]))
        for t in range(16, 64):
            s0 = _rotr(w[t - 15], 7) ^ _rotr(w[t - 15], 18) ^ (w[t - 15] >> 3)
            s1 = _rotr(w[t - 2], 17) ^ _rotr(w[t - 2], 19) ^ (w[t - 2] >> 10)
            w.append((w[t - 16] + s0 + w[t - 7] + s1) & _M)
        a, bb, c, d, e, f, g, hh = h
        for t in range(64):
            S1 = _rotr(e, 6) ^ _rotr(e, 11) ^ _rotr(e, 25)
            ch = (e & f) ^ (~e & _M & g)
            t1 = (hh + S1 + ch + _K[t] + w[t]) & _M
            S0 = _rotr(a, 2) ^ _rotr(a, 13) ^ _rotr(a, 22)
            maj = (a & bb) ^ (a & c) ^ (bb & c)
            out.extend((a, e, w[t], S0, S1, maj, ch, t1))
            t2 = (S0 + maj) & _M
            hh, g, f, e, d, c, bb, a = g, f, e, (d + t1) & _M, c, bb, a, (t1 + t2) & _M
        h = [(x + y) & _M for x, y in zip(h, (a, bb, c, d, e, f, g, hh))]
    return np.array(out, dtype=np.uint32)


def spread16(x: np.ndarray) -> np.ndarray:
    """bit i -> bit 2i for 16-bit inputs."""
    x = x.astype(np.uint64) & np.uint64(0xFFFF)
    x = (x | (x << np.uint64(8))) & np.uint64(0x00FF00FF)
    x = (x | (x << np.uint64(4))) & np.uint64(0x0F0F0F0F)
    x = (x | (x << np.uint64(2))) & np.uint64(0x33333333)
    x = (x | (x << np.uint64(1))) & np.uint64(0x55555555)
    return x


def small_to_mont(vals) -> np.ndarray:
    """small non-negative ints -> uint64[n,4] Montgomery limbs."""
    return np.stack([fr_to_mont(int(v)) for v in vals])


TABLE_BITS = 12
BLOCKS_FOR_K = {14: 1, 16: 16, 18: 64, 20: 256, 22: 1024}


class ShaCqWorkload:
    """Proving key + device-resident witness for the SHA-shaped CQ circuit at 2^k rows."""

    def __init__(self, ctx: Context, k: int, pairs: int = 4, blocks: int | None = None, seed: int = 0x5348413243515F,
                 table_bits: int = TABLE_BITS, share: "ShaCqWorkload | None" = None):
        import ctypes as C

        self.ctx, self.k, self.pairs = ctx, k, pairs
        n = 1 << k
        self.n = n
        N = 1 << table_bits
        if share is not None:  # same SRS / tables as another workload of the same size (setup is not what is measured)
            assert share.k == k and share.cfg.size == N
            self.params, self.cfg, self.dense, self.spread = share.params, share.cfg, share.dense, share.spread
        else:
            s = fr_to_mont(seed * 0x9E3779B97F4A7C15 + 12345)
            self.params = ParamsKZG.setup_from_toxic_waste(ctx, k, s)
            self.cfg = TableConfig.setup_from_toxic_waste(ctx, N, s)
            idx = np.arange(N)
            self.dense = StaticTable.setup_from_toxic_waste(ctx, small_to_mont(idx), s)
            self.spread = StaticTable.setup_from_toxic_waste(ctx, small_to_mont(spread16(idx)), s)
        lookups = [[(2 * p, self.dense), (2 * p + 1, self.spread)] for p in range(pairs)]
        # b0 bound over the circuit SRS itself: srs_g1_len = n  =>  [s^1 .. s^(n-1)]_1 = g[1..]
        self.pk = self._make_pk(lookups)
        u = self.pk.usable_rows
        if blocks is None:
            blocks = BLOCKS_FOR_K.get(k, max(1, (u * pairs) // (3 * 8 * 64)))
        words = sha256_trace_words(blocks)
        max_words = (u * pairs) // 3
        if words.shape[0] > max_words:
            words = words[:max_words]
        self.blocks = blocks
        self.nwords = int(words.shape[0])
        self.words_dev = ctx.to_device(words)
        self.cols = [ctx.alloc(n * 32) for _ in range(2 * pairs)]
        self.fill_witness()
        self._extra_witness(words)

    def _make_pk(self, lookups):
        return ProvingKey(self.ctx, self.params, self.k, 2 * self.pairs, lookups, self.cfg, self.params.g_dev + 64,
                          fr_to_mont(0xC0FFEE + self.k))

    def _extra_witness(self, words):
        pass

    def fill_witness(self):
        """SHA word -> limb witness fill on the GPU (cq_sha_witness_fill_dev)."""
        import ctypes as C

        ctx = self.ctx
        for c in self.cols:
            ctx._chk(ctx.lib.cq_dev_memset(ctx.h, c.ptr, 0, self.n * 32))
        arr = (C.c_void_p * len(self.cols))(*[c.ptr for c in self.cols])
        ctx._chk(ctx.lib.cq_sha_witness_fill_dev(ctx.h, self.words_dev.ptr, self.nwords, self.pairs, self.n, arr))

    def prove(self, seed: int = 1) -> bytes:
        return self.pk.create_proof_dev([c.ptr for c in self.cols], seed=seed)

    def close(self, shared: bool = True):
        """Frees the witness and the proving key, and (shared=True) the SRS / table objects -- 17 window tables per SRS
        array, 4.3 GiB each at k = 22."""
        objs = [self.pk, self.words_dev] + list(self.cols)
        if shared:
            objs += [self.dense, self.spread, self.cfg, self.params]
        for o in objs:
            o.close()

    # work counts of one proof (for throughput figures)
    def msm_scalars_per_proof(self) -> int:
        L, A, n, N = self.pairs, 2 * self.pairs, self.n, self.cfg.size
        dense = A * n + L * n + 2 * L * (n - 1) + n + 2 * n + (n - 1)  # advice, f, (p, b0), random, h pieces, W
        table = L * N + 2 * L * N + L * 2 * N                          # m, (a, a0), q_a
        return dense + table

    def msm_scalars_in_launches(self) -> int:
        """What the MSM kernels of this backend actually take in: the L commitments to f are obtained as
        theta * [a_2p] + [a_2p+1] from the advice commitments (plain advice columns as lookup inputs), not by an MSM."""
        return self.msm_scalars_per_proof() - self.pairs * self.n

    def msm_scalars_full_width(self) -> int:
        """The subset of msm_scalars_in_launches whose scalars are uniformly distributed field elements: p and b0, the
        random polynomial, the h pieces, the opening witness, and a / a0 / q_a over the table (non-zero where m is).
        Not counted: the advice columns (<= 24-bit limbs on 9.4 % of the rows) and m (small multiplicities)."""
        L, n, N = self.pairs, self.n, self.cfg.size
        return 2 * L * (n - 1) + n + 2 * n + (n - 1) + 2 * L * N + L * 2 * N

    def ntt_elems_per_proof(self) -> int:
        L, A, n = self.pairs, 2 * self.pairs, self.n
        return (2 * L + A) * n + 2 * L * 2 * n + 2 * n  # iNTT(b,f), iNTT(advice), coset NTTs, iNTT(ext)


class ShaPlonkWorkload(ShaCqWorkload):
    """The SHA-shaped CQ circuit plus what a real SHA-2 circuit wraps around its lookups: a custom gate with a
    rotation, a fixed selector, and copy constraints.  Two more advice columns: `w` = limb0 + 2^16 * limb1 of
    each row (gate `q * (a0 + 2^16 * a2 - w)`), `w2[r] = w[r+1]` (gate `q * (w2 - w@next)` and, for every row, a
    copy constraint (w2, r) == (w, r+1)); permutation over (w, w2) -> two product sets at degree 3."""

    legacy_lookup = False  # class attribute: set before construction to add a halo2 (permutation-based) lookup as well

    def _make_pk(self, lookups):
        from . import plonk as GP

        n, pairs = self.n, self.pairs
        cs = GP.ConstraintSystem()
        adv = [cs.advice_column() for _ in range(2 * pairs + 2)]
        q = cs.fixed_column()
        tcol = cs.fixed_column() if self.legacy_lookup else None
        w, w2 = adv[2 * pairs], adv[2 * pairs + 1]
        cs.enable_equality(w)
        cs.enable_equality(w2)
        qe = cs.query_fixed(q)
        cs.create_gate("recompose", [qe * (cs.query_advice(adv[0]) + cs.query_advice(adv[2]) * (1 << 16) - cs.query_advice(w))])
        cs.create_gate("shift", [qe * (cs.query_advice(w2) - cs.query_advice(w, 1))])
        for lk in lookups:
            cs.lookup_static("limb", [(adv[c], t) for c, t in lk])
        if tcol is not None:  # the first limb column against a fixed column holding 0 .. 4095 (plonk/lookup.rs, `lookup_any`)
            cs.lookup("limb12", [(cs.query_advice(adv[0]), cs.query_fixed(tcol))])
        self.cs = cs
        u = n - (cs.blinding_factors() + 1)
        self.rows = u - 1  # gate / copy rows: w@next must stay inside the usable rows
        qcol = np.zeros((n, 4), dtype=np.uint64)
        qcol[: self.rows] = fr_to_mont(1)
        # copy constraints (w2, r) == (w, r+1): disjoint 2-cycles, written directly as Assembly.mapping
        mapping = np.zeros((2, n, 2), dtype=np.uint32)
        mapping[0, :, 0], mapping[1, :, 0] = 0, 1
        mapping[:, :, 1] = np.arange(n, dtype=np.uint32)
        r = np.arange(self.rows, dtype=np.uint32)
        mapping[1, r, 0], mapping[1, r, 1] = 0, r + 1
        mapping[0, r + 1, 0], mapping[0, r + 1, 1] = 1, r
        fixed = [qcol]
        if tcol is not None:
            fixed.append(small_to_mont(np.arange(n, dtype=np.uint64) & np.uint64(0xFFF)))
        return ProvingKey(self.ctx, self.params, self.k, 0, [], self.cfg, self.params.g_dev + 64, fr_to_mont(0xC0FFEE + self.k),
                          cs=cs, fixed=fixed, permutation=mapping)

    def _extra_witness(self, words):
        n, pairs, u = self.n, self.pairs, self.pk.usable_rows
        wv = words.astype(np.uint64)
        limbs = np.stack([wv >> 20, (wv >> 10) & 0x3FF, wv & 0x3FF], axis=1).reshape(-1)  # (x, y, z) of tables.rs:135-154
        grid = np.zeros(u * pairs, dtype=np.uint64)
        grid[: limbs.shape[0]] = limbs
        grid = grid.reshape(u, pairs)  # limb t -> row t // pairs, pair t % pairs (cq_sha_witness_fill_dev)
        wcol = grid[:, 0] + (grid[:, 1] << np.uint64(16))
        wm = np.zeros((n, 4), dtype=np.uint64)
        wm[:u] = small_to_mont(wcol)
        w2m = np.zeros((n, 4), dtype=np.uint64)
        w2m[: u - 1] = wm[1:u]
        self.cols += [self.ctx.to_device(wm), self.ctx.to_device(w2m)]
