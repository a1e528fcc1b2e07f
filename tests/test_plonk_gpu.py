"""GPU parity: `create_proof` for general circuits (custom gates with rotations, fixed / instance columns,
the permutation argument over several product sets, optionally next to a static lookup).  Proof BYTES must
equal the oracle's for the same key, witness, public inputs and RNG stream; larger proofs (several scan
tiles, extended domain 4n) are checked with the acceptance verifier."""
import os

import numpy as np
import pytest

from oracle import bn254 as B
from oracle import cq_prover as CP
from oracle import cq_verifier as CV
from tests.plonk_fixtures import TABLE, c_msm, chain_circuit, oracle_env, plonk_api_circuit, to_backend_cs

pytestmark = pytest.mark.gpu
P = B.R_MOD


def _backend_pk(ctx, fx, k, s, b0=None, b0_on_params=False):
    from sha2_on_cq_halo2_amd import ParamsKZG, ProvingKey, StaticTable, TableConfig

    sm = B.to_mont_limbs([s])[0]
    n = 1 << k
    gparams = ParamsKZG.setup_from_toxic_waste(ctx, k, sm)
    gtables, gcfg, b0_arg = {}, None, None
    if fx["tables"]:
        gcfg = TableConfig.setup_from_toxic_waste(ctx, len(TABLE), sm)
        gtables = {name: StaticTable.setup_from_toxic_waste(ctx, B.to_mont_limbs(v), sm) for name, v in fx["tables"].items()}
        b0_arg = (gparams.g_dev + 64) if b0_on_params else B.points_to_mont_limbs(b0)
    cs = to_backend_cs(fx["circuit"], gtables)
    fixed = [B.to_mont_limbs(c) for c in fx["fixed"]]
    mapping = np.array(fx["mapping"], dtype=np.uint32)
    gpk = ProvingKey(ctx, gparams, k, 0, [], gcfg, b0_arg, B.to_mont_limbs([424242])[0], cs=cs, fixed=fixed, permutation=mapping)
    return gpk, gparams


def _advice_cols(fx, n):
    return [B.to_mont_limbs(list(c) + [0] * (n - len(c))) for c in fx["advice"]]


@pytest.mark.parametrize("kw", [dict(), dict(degree5=True), dict(with_lookup=True), dict(lookup_expr=True), dict(plookup=True),
                                dict(plookup=True, with_lookup=True, degree5=True)],
                         ids=["deg3", "deg5", "lookup", "lookup-expr", "plookup", "plookup+cq+deg5"])
def test_plonk_proof_bytes_match_oracle(ctx, kw):
    k = 5
    n = 1 << k
    fx = oracle_env(k, **kw)
    gpk, _ = _backend_pk(ctx, fx, k, fx["s"], b0=fx["pk"].b0_g1_bound)
    # the proving key the backend derived: sigma / fixed commitments equal the oracle's
    fcm, pcm = gpk.vk_commitments()
    assert np.array_equal(fcm, B.points_to_mont_limbs(B.batch_to_affine([fx["params"].commit_lagrange(c) for c in fx["pk"].fixed_values])))
    assert np.array_equal(pcm, B.points_to_mont_limbs(B.batch_to_affine([fx["params"].commit_lagrange(c) for c in fx["pk"].permutations])))
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(77), instances=fx["instances"])
    proof = gpk.create_proof(_advice_cols(fx, n), seed=77, instances=[B.to_mont_limbs(i) for i in fx["instances"]])
    assert len(proof) == gpk.proof_size == len(tr.proof)
    assert proof == tr.proof


@pytest.mark.parametrize("opener", ["gwc", "shplonk"])
def test_plonk_proof_bytes_match_oracle_k11(ctx, opener):
    """Byte parity beyond one scan tile (n = 2048: two tiles per product set, extended domain 8192, 5 permutation
    columns in two sets, a degree-5 gate and a static lookup together).  The Python prover runs with the C
    restatement's multiexp and SRS points so that it finishes in seconds."""
    k = 11
    fx = oracle_env(k, fast=True, degree5=True, with_lookup=True)
    gpk, _ = _backend_pk(ctx, fx, k, fx["s"], b0=fx["pk"].b0_g1_bound)
    gpk.set_opener(opener)
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(31), msm=c_msm, instances=fx["instances"], opener=opener)
    proof = gpk.create_proof(_advice_cols(fx, 1 << k), seed=31, instances=[B.to_mont_limbs(i) for i in fx["instances"]])
    assert proof == tr.proof


def test_plonk_api_shape_proof_bytes(ctx):
    """halo2_proofs/tests/plonk_api.rs `MyCircuit`, legacy lookup included: 6 chained product sets at degree 4."""
    k = 5
    fx = oracle_env(k, builder=plonk_api_circuit)
    gpk, _ = _backend_pk(ctx, fx, k, fx["s"])
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(3), instances=fx["instances"])
    proof = gpk.create_proof(_advice_cols(fx, 1 << k), seed=3, instances=[B.to_mont_limbs(i) for i in fx["instances"]])
    assert proof == tr.proof and len(proof) == gpk.proof_size


@pytest.mark.parametrize("kw", [dict(), dict(degree5=True), dict(with_lookup=True)], ids=["deg3", "deg5", "lookup"])
def test_shplonk_proof_bytes_match_oracle(ctx, kw):
    """ProverSHPLONK (poly/kzg/multiopen/shplonk/prover.rs:120-286) instead of ProverGWC."""
    k = 5
    fx = oracle_env(k, **kw)
    gpk, _ = _backend_pk(ctx, fx, k, fx["s"], b0=fx["pk"].b0_g1_bound)
    gpk.set_opener("shplonk")
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(21), instances=fx["instances"], opener="shplonk")
    proof = gpk.create_proof(_advice_cols(fx, 1 << k), seed=21, instances=[B.to_mont_limbs(i) for i in fx["instances"]])
    assert len(proof) == gpk.proof_size == len(tr.proof)
    assert proof == tr.proof
    gpk.set_opener("gwc")
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(21), instances=fx["instances"])
    assert gpk.create_proof(_advice_cols(fx, 1 << k), seed=21, instances=[B.to_mont_limbs(i) for i in fx["instances"]]) == tr.proof


def test_plonk_instance_too_large_and_missing(ctx):
    from sha2_on_cq_halo2_amd import CqError

    k = 5
    fx = oracle_env(k)
    gpk, _ = _backend_pk(ctx, fx, k, fx["s"])
    cols = _advice_cols(fx, 1 << k)
    with pytest.raises(CqError):  # Error::InstanceTooLarge (prover.rs:108-110)
        gpk.create_proof(cols, seed=1, instances=[B.to_mont_limbs([1] * (gpk.usable_rows + 1))])
    with pytest.raises(CqError):  # Error::InvalidInstances (:73-82)
        gpk.create_proof(cols, seed=1)


@pytest.mark.parametrize("k,kw", [(11, dict(degree5=True)), (12, dict(with_lookup=True)), (10, dict(lookup_expr=True))],
                         ids=["k11-deg5", "k12-lookup", "k10-lookup-expr"])
def test_plonk_large_proof_verifies(ctx, k, kw):
    """Beyond the sizes the Python prover reaches: several scan tiles per product set, batched multi-point
    openings; the proof must satisfy every verifier equation, and a broken copy constraint must not."""
    n = 1 << k
    fx = chain_circuit(k, **kw)
    s = B.fr_random(B.Xoshiro256ss(k))
    gpk, gparams = _backend_pk(ctx, fx, k, s, b0_on_params=True)  # b0_g1_bound = g[1..], srs_g1_len = n
    fcm, pcm = gpk.vk_commitments()
    to_pts = B.points_from_mont_limbs
    inst = [B.to_mont_limbs(i) for i in fx["instances"]]
    cols = _advice_cols(fx, n)
    proof = gpk.create_proof(cols, seed=5, instances=inst)

    def verify(pr, opener="gwc"):
        return CV.verify_proof(pr, fx["circuit"], 424242, s, fx["tables"], len(TABLE), n, instances=fx["instances"],
                               fixed_commitments=to_pts(fcm), perm_commitments=to_pts(pcm), opener=opener)

    assert verify(proof)
    gpk.set_opener("shplonk")
    assert verify(gpk.create_proof(cols, seed=5, instances=inst), "shplonk")
    gpk.set_opener("gwc")
    # break one copy constraint while keeping every gate satisfied: b[0] is tied to a fixed cell only by the permutation
    adv = [list(c) for c in fx["advice"]]
    adv[1][0] = (adv[1][0] + 5) % P
    adv[2][0] = (adv[0][0] + adv[1][0]) % P
    bad_cols = [B.to_mont_limbs(list(c) + [0] * (n - len(c))) for c in adv]
    bad = gpk.create_proof(bad_cols, seed=5, instances=inst)
    assert not verify(bad)


@pytest.mark.parametrize("kw", [dict(degree5=True), dict(with_lookup=True)], ids=["deg5", "lookup"])
def test_proving_key_raw_bytes_roundtrip(ctx, kw):
    """`ProvingKey::write` / `read` in SerdeFormat::RawBytes (plonk.rs:349-403): the backend's stream equals the
    oracle's serialization of the same key byte for byte (cosets included), and a key READ from those bytes
    (polynomials uploaded, nothing recomputed) proves to the same bytes."""
    from oracle import serde as SD
    from sha2_on_cq_halo2_amd import CqError, ProvingKey

    k = 5
    n = 1 << k
    fx = oracle_env(k, **kw)
    gpk, gparams = _backend_pk(ctx, fx, k, fx["s"], b0=fx["pk"].b0_g1_bound)
    fixed_cm = B.batch_to_affine([fx["params"].commit_lagrange(c) for c in fx["pk"].fixed_values])
    perm_cm = B.batch_to_affine([fx["params"].commit_lagrange(c) for c in fx["pk"].permutations])
    selectors = [[(r * 7 + 1) % 3 == 0 for r in range(n)], [r % 2 == 1 for r in range(n)]]
    want = SD.proving_key_to_bytes(fx["pk"], fixed_cm, perm_cm, selectors)
    got = gpk.to_bytes(SD.pack_selectors(selectors, n), 2)
    assert got == want
    # read it back: no fixed columns / mapping handed over, only the circuit's shape
    gtables, gcfg, b0_arg = {}, None, None
    if fx["tables"]:
        gcfg, gtables = gpk._keep[1], {name: t for name, t in zip(fx["tables"], gpk._keep[2])}
        b0_arg = B.points_to_mont_limbs(fx["pk"].b0_g1_bound)
    cs = to_backend_cs(fx["circuit"], gtables)
    rpk = ProvingKey(ctx, gparams, k, 0, [], gcfg, b0_arg, B.to_mont_limbs([424242])[0], cs=cs, raw=want, num_selectors=2)
    cols = _advice_cols(fx, n)
    inst = [B.to_mont_limbs(i) for i in fx["instances"]]
    assert rpk.create_proof(cols, seed=3, instances=inst) == gpk.create_proof(cols, seed=3, instances=inst)
    assert rpk.to_bytes(SD.pack_selectors(selectors, n), 2) == want
    # RawBytes checks: a limb vector >= the modulus is rejected; truncated input too
    bad = bytearray(want)
    off = 8 + 64 * (len(fixed_cm) + len(perm_cm)) + 2 * (n // 8) + 4  # first element of l0
    bad[off:off + 32] = b"\xff" * 32
    with pytest.raises(CqError):
        ProvingKey(ctx, gparams, k, 0, [], gcfg, b0_arg, B.to_mont_limbs([424242])[0], cs=cs, raw=bytes(bad), num_selectors=2)
    ProvingKey(ctx, gparams, k, 0, [], gcfg, b0_arg, B.to_mont_limbs([424242])[0], cs=cs, raw=bytes(bad), num_selectors=2, checked=False)
    with pytest.raises(CqError):
        ProvingKey(ctx, gparams, k, 0, [], gcfg, b0_arg, B.to_mont_limbs([424242])[0], cs=cs, raw=want[:-5], num_selectors=2)


def test_legacy_lookup_failure_is_an_error(ctx):
    """An input that is not a table row: Error::ConstraintSystemFailure (lookup/prover.rs:431-437)."""
    from sha2_on_cq_halo2_amd import CqError

    k = 5
    fx = oracle_env(k, plookup=True)
    gpk, _ = _backend_pk(ctx, fx, k, fx["s"])
    adv = [list(c) for c in fx["advice"]]
    adv[3][0] = 4  # (4, 9) is not in the table
    cols = [B.to_mont_limbs(list(c) + [0] * ((1 << k) - len(c))) for c in adv]
    with pytest.raises(CqError) as e:
        gpk.create_proof(cols, seed=1, instances=[B.to_mont_limbs(i) for i in fx["instances"]])
    assert e.value.code == -4
    with pytest.raises(ValueError):
        CP.create_proof(fx["params"], fx["pk"], adv, B.Xoshiro256ss(1), instances=fx["instances"])


@pytest.mark.parametrize("k,usable,distinct", [(3, 2, 2), (5, 26, 7), (10, 1018, 50), (11, 2042, 2042), (12, 4090, 300),
                                               (14, 16378, 5000), (16, 65530, 65530), (16, 40000, 1), (17, 131066, 9)])
def test_permute_expression_pair_matches_oracle(ctx, k, usable, distinct):
    """`permute_expression_pair` (lookup/prover.rs:400-502) on the device -- a bitonic sort of the canonical values, LDS and
    streaming steps, first-occurrence claims, leftovers into the repeated rows -- against the oracle's restatement
    (oracle/plonk.py), value for value: small domains (one block), one LDS tile, several merge phases; inputs with few
    and with all-distinct values; a table with duplicates; a usable range that is not the whole domain."""
    from oracle import plonk as OP
    from sha2_on_cq_halo2_amd import CqError

    rs = np.random.RandomState(k * 1000 + distinct)
    rng = B.Xoshiro256ss(k + distinct)
    values = [B.fr_random(rng) for _ in range(min(distinct, 64))] + [int(v) for v in rs.randint(0, 1 << 62, size=max(0, distinct - 64))]
    values = list(dict.fromkeys(values))
    # table: every value at least once, padded with repeats of the first ones; input: values drawn from the table
    table = [values[i % len(values)] for i in range(usable)]
    table = [table[i] for i in rs.permutation(usable)]
    inp = [values[int(i)] for i in rs.randint(0, min(len(values), usable), size=usable)]
    n, bf = 1 << k, (1 << k) - usable - 1
    exp_in, exp_tab = OP.permute_expression_pair(n, bf, inp, table, B.Xoshiro256ss(1))
    got_in, got_tab = ctx.permute_expression_pair(k, B.to_mont_limbs(inp), B.to_mont_limbs(table))
    assert np.array_equal(got_in, B.to_mont_limbs(exp_in[:usable]))
    assert np.array_equal(got_tab, B.to_mont_limbs(exp_tab[:usable]))
    if usable > 1 and len(values) > 1:  # an input value the table does not hold
        bad = list(inp)
        bad[usable // 2] = (max(values) + 1) % P
        if bad[usable // 2] not in values:
            with pytest.raises(CqError) as e:
                ctx.permute_expression_pair(k, B.to_mont_limbs(bad), B.to_mont_limbs(table))
            assert e.value.code == -4


def test_permute_expression_pair_edge_cases(ctx):
    """Empty and one-row inputs, a usable range equal to the whole domain, and argument errors."""
    from oracle import plonk as OP
    from sha2_on_cq_halo2_amd import CqError

    a, t = ctx.permute_expression_pair(4, np.zeros((0, 4), dtype=np.uint64), np.zeros((0, 4), dtype=np.uint64))
    assert a.shape == (0, 4) and t.shape == (0, 4)
    one = B.to_mont_limbs([7])
    a, t = ctx.permute_expression_pair(0, one, one)  # domain of one row
    assert np.array_equal(a, one) and np.array_equal(t, one)
    vals = [5, 3, 5, 1, 3, 3, 1, 5]  # usable == 2^k: no padding slots at all
    tab = [1, 3, 5, 9, 9, 9, 2, 2]
    exp_in, exp_tab = OP.permute_expression_pair(8, -1, vals, tab, B.Xoshiro256ss(1))
    a, t = ctx.permute_expression_pair(3, B.to_mont_limbs(vals), B.to_mont_limbs(tab))
    assert np.array_equal(a, B.to_mont_limbs(exp_in[:8])) and np.array_equal(t, B.to_mont_limbs(exp_tab[:8]))
    with pytest.raises(CqError):  # more rows than the domain holds
        ctx.permute_expression_pair(2, B.to_mont_limbs(vals), B.to_mont_limbs(tab))
    with pytest.raises(CqError) as e:  # 4 is not in the table
        ctx.permute_expression_pair(3, B.to_mont_limbs([4] + vals[1:]), B.to_mont_limbs(tab))
    assert e.value.code == -4


@pytest.mark.parametrize("seed", range(int(os.environ.get("CQ_FUZZ_SEEDS", "12"))))  # CQ_FUZZ_SEEDS=100 for a long run
def test_random_circuits_proof_bytes_match_oracle(ctx, seed):
    """Fuzz: random gates (random expression trees, rotations in [-2, 2]), random column mix, random copy
    constraints, optional legacy / static lookups, random witness (constraints NOT satisfied -- the prover never
    checks them): the bytes must still agree, for both multi-open schemes."""
    from tests.plonk_fixtures import random_circuit

    k = 5
    fx = oracle_env(k, builder=random_circuit, seed=seed)
    gpk, _ = _backend_pk(ctx, fx, k, fx["s"], b0=fx["pk"].b0_g1_bound)
    cols = _advice_cols(fx, 1 << k)
    inst = [B.to_mont_limbs(i) for i in fx["instances"]]
    for opener in ("gwc", "shplonk"):
        gpk.set_opener(opener)
        tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(100 + seed), instances=fx["instances"], opener=opener)
        assert gpk.create_proof(cols, seed=100 + seed, instances=inst) == tr.proof, (seed, opener)


def test_multi_phase_proof_bytes_match_oracle(ctx):
    """Three phases (circuit.rs FirstPhase..ThirdPhase): the phase-1 and phase-2 advice columns are computed from the
    challenges squeezed after the earlier phases' commitments (prover.rs:436-463) through the phase callback; gates
    read the challenges (Expression::Challenge).  Bytes equal the oracle's; a single-shot call is refused."""
    from sha2_on_cq_halo2_amd import CqError

    k = 5
    n = 1 << k
    fx = oracle_env(k, phases=True)
    cs = fx["circuit"]
    assert cs.num_phases() == 3
    gpk, _ = _backend_pk(ctx, fx, k, fx["s"])
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(41), instances=fx["instances"])
    bufs = []
    for col in fx["advice"]:
        vals = [0] * n if callable(col) else list(col) + [0] * (n - len(col))
        bufs.append(ctx.to_device(B.to_mont_limbs(vals)))
    calls = []

    def phase_fn(phase, challenges):
        calls.append((phase, list(challenges)))
        out = {}
        for c_, col in enumerate(fx["advice"]):
            if callable(col) and cs.phase_of(c_) == phase:
                vals = col(challenges)
                out[c_] = B.to_mont_limbs(list(vals) + [0] * (n - len(vals)))
        return out

    inst = [B.to_mont_limbs(i) for i in fx["instances"]]
    proof = gpk.create_proof_phases(bufs, phase_fn, seed=41, instances=inst)
    assert [c_[0] for c_ in calls] == [1, 2]
    assert calls[0][1][0] == tr.challenges["user"][0] and calls[1][1] == tr.challenges["user"]
    assert proof == tr.proof
    with pytest.raises(CqError):
        gpk.create_proof_dev([b_.ptr for b_ in bufs], seed=41, instances=inst)
