"""GPU parity: SHA table generators (sha/src/tables.rs), StaticTableValues::new (the reference's
O(N^2) cached-quotient construction) and the ParamsKZG RawBytes formats."""
import numpy as np
import pytest

from oracle import bn254 as B
from oracle import kzg
from oracle import sha_tables as ST

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("limbs", ["test", "short"])
def test_sha_synthesis_tables_match_oracle(ctx, limbs):
    from sha2_on_cq_halo2_amd.api import SHA_CH, SHA_MAJ, SHA_ROT0, SHA_ROT1

    first, second = ST.LIMBS[limbs]
    for kind, gen in ((SHA_ROT0, ST.create_rot0_table), (SHA_ROT1, ST.create_rot1_table), (SHA_MAJ, ST.create_maj_table),
                      (SHA_CH, ST.create_ch_table)):
        got = ctx.sha_synthesis_table(kind, first, second)
        exp = np.array(gen(limbs), dtype=np.uint64)
        assert np.array_equal(got, exp)


def test_sha_table_kats_on_gpu(ctx):
    """The reference's own KAT rows (sha/src/tables.rs:168-211), TestLimbs 4/2/2."""
    from sha2_on_cq_halo2_amd.api import SHA_CH, SHA_MAJ, SHA_ROT0, SHA_ROT1

    def rows(a):
        return {tuple(int(v) for v in r) for r in a}

    t = rows(ctx.sha_synthesis_table(SHA_ROT0, 4, 2))
    assert len(t) == 256 and (0, 0, 0, 0) in t and (0b1100, 1, 0, 0b100) in t and (0b1010, 1, 0b10, 0b110) in t
    t = rows(ctx.sha_synthesis_table(SHA_ROT1, 4, 2))
    assert (0b1100, 1, 0, 0b11101001) in t and (0b1010, 1, 0b10, 0b00011101) in t
    t = rows(ctx.sha_synthesis_table(SHA_MAJ, 4, 2))
    assert (0b1100, 1, 0, 0) in t and (0b1010, 1, 0b10, 0b10) in t
    t = rows(ctx.sha_synthesis_table(SHA_CH, 4, 2))
    assert (0b1100, 1, 0, 0) in t and (0b1010, 1, 0b10, 0) in t
    d = rows(ctx.sha_decomposition_table(4, 2, 10))
    assert len(d) == 1024 and (0, 0, 0, 0) in d
    assert (0b1010101010, 0b1010, 0b10, 0b10) in d and (0b1110110110, 0b1011, 0b01, 0b10) in d


@pytest.mark.parametrize("limbs,k", [("test", 10), ("short", 12), ("long", 14)])
def test_sha_decomposition_table_matches_oracle(ctx, limbs, k):
    first, second = ST.LIMBS[limbs]
    got = ctx.sha_decomposition_table(first, second, k)
    assert np.array_equal(got, np.array(ST.create_decomposition_table(limbs, k), dtype=np.uint64))


@pytest.mark.parametrize("N", [16, 64])
def test_static_table_new_matches_reference_construction(ctx, N):
    """qs from the GPU run of StaticTableValues::new == the oracle's restatement == the closed form."""
    from sha2_on_cq_halo2_amd import StaticTable

    s = B.fr_random(B.Xoshiro256ss(N))
    rng = B.Xoshiro256ss(N + 1)
    values = []
    while len(values) < N:
        v = rng.next_u64() % 100000
        if v not in values:
            values.append(v)
    tsrs = kzg.TableSRS(N - 1, s)
    exp = kzg.StaticTableValues(values, tsrs.g1).qs
    t = StaticTable.new(ctx, B.to_mont_limbs(values), B.points_to_mont_limbs(tsrs.g1))
    assert np.array_equal(t.download_qs(), B.points_to_mont_limbs(exp))
    assert exp == kzg.StaticTableValues.qs_closed_form(values, s)


def test_static_table_rejects_duplicate_values(ctx):
    from sha2_on_cq_halo2_amd import CqError, StaticTable

    vals = B.to_mont_limbs([1, 2, 3, 2])
    with pytest.raises(CqError):
        StaticTable.setup_from_toxic_waste(ctx, vals, B.to_mont_limbs([5])[0])


def test_params_raw_roundtrip_and_validation(ctx):
    from sha2_on_cq_halo2_amd import CqError, ParamsKZG
    from tests.util import jac_limbs_to_affine, random_scalars

    k = 5
    s = B.fr_random(B.Xoshiro256ss(3))
    op = kzg.ParamsKZG(k, s)
    raw = (k).to_bytes(4, "little") + B.points_to_mont_limbs(op.g).tobytes() + B.points_to_mont_limbs(op.g_lagrange).tobytes()
    raw_with_g2 = raw + bytes(256)  # g2 | s_g2 tail is ignored
    for data, checked in ((raw_with_g2, True), (raw, False)):
        p = ParamsKZG.read_raw(ctx, data, checked=checked)
        assert p.write_raw() == raw
        a = random_scalars(1 << k, 8)
        assert jac_limbs_to_affine(p.commit(B.to_mont_limbs(a))) == B.jac_to_affine(op.commit(a))
    bad = bytearray(raw_with_g2)
    bad[4 + 64 * 3 + 5] ^= 0x40  # corrupt g[3].x: no longer on the curve
    with pytest.raises(CqError):
        ParamsKZG.read_raw(ctx, bytes(bad), checked=True)
    ParamsKZG.read_raw(ctx, bytes(bad), checked=False)  # RawBytesUnchecked performs no checks
    with pytest.raises(CqError):
        ParamsKZG.read_raw(ctx, raw[:100], checked=False)


@pytest.mark.parametrize("N", [2, 16, 256, 4096])
def test_static_table_fk_equals_reference_construction(ctx, N):
    """FK-style preprocessing (one size-2N convolution and one DFT over G1) gives the same cached quotients as
    `StaticTableValues::new` (static_lookup.rs:78-126), checked against the closed form with the toxic waste and,
    for small N, against the O(N^2) construction run on the GPU."""
    from sha2_on_cq_halo2_amd import ParamsKZG, StaticTable

    s = B.fr_random(B.Xoshiro256ss(31 + N))
    sm = B.to_mont_limbs([s])[0]
    rng = B.Xoshiro256ss(N)
    values = []
    while len(values) < N:  # unique values
        v = B.fr_random(rng) if N > 16 else rng.next_u64() % 1000
        if v not in values:
            values.append(v)
    vm = B.to_mont_limbs(values)
    k = max(N.bit_length() - 1, 1)
    srs = ParamsKZG.setup_from_toxic_waste(ctx, k, sm).download()[0][:N]
    fk = StaticTable.new_fk(ctx, vm, srs).download_qs()
    closed = StaticTable.setup_from_toxic_waste(ctx, vm, sm).download_qs()
    assert np.array_equal(fk, closed)
    if N <= 256:
        assert np.array_equal(fk, StaticTable.new(ctx, vm, srs).download_qs())
