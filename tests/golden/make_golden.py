"""Generates tests/golden/*.json.  Run from the repo root:  python tests/golden/make_golden.py

Two kinds of vectors:
  * `reference_kats.json`: the known-answer values the REFERENCE's own tests hold for this path (copied as data,
    each with the file:line it comes from) -- the pins of the oracle.
  * `oracle_vectors.json`: inputs -> outputs computed by the pinned oracle in this container (the Rust reference
    cannot be built here and holds no golden proof: its end-to-end tests draw from OsRng).  MSM / NTT / evaluation
    vectors and full proofs (seeded inputs, hex-encoded outputs) that the GPU tests replay through the C ABI
    without importing the oracle's arithmetic.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import bn254 as B  # noqa: E402
from oracle import cq_prover as CP  # noqa: E402
from oracle import poly as OP  # noqa: E402
from tests.plonk_fixtures import chain_circuit, oracle_env, plonk_api_circuit  # noqa: E402
from tests.util import random_points, random_scalars  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hexl(v):
    return "%064x" % v


def reference_kats():
    return {
        "fr_from_u512_aaaa": {"source": "arithmetic/curves/src/bn256/fr.rs:347-367", "input_u64x8": ["aaaaaaaaaaaaaaaa"] * 8,
                              "raw_limbs": ["7e7140b5196b9e6f", "9abac9e4157b6172", "f04bc41062fd7322", "1185fa9c9fef6326"]},
        "fq_from_u512_aaaa": {"source": "arithmetic/curves/src/bn256/fq.rs:331-351", "input_u64x8": ["aaaaaaaaaaaaaaaa"] * 8,
                              "raw_limbs": ["1f8905a172affa8a", "de45ad177dcf3306", "aaa7987907d73ae2", "24d349431d468e30"]},
        "fr_constants": {"source": "arithmetic/curves/src/bn256/fr.rs:29-118",
                         "modulus": hexl(B.R_MOD), "inv": "c2e1f593efffffff", "s": 28, "generator": 7,
                         "r2_limbs": ["1bb8e645ae216da7", "53fe3ab1e35c59e3", "8c49833d53bb8085", "0216d0b17f4e44a5"]},
        "fq_constants": {"source": "arithmetic/curves/src/bn256/fq.rs:29-91", "modulus": hexl(B.Q_MOD), "inv": "87d20782e4866389"},
        "g1": {"source": "arithmetic/curves/src/bn256/curve.rs:66-68", "generator": [1, 2], "b": 3},
        "sha_tables_test_limbs": {
            "source": "sha/src/tables.rs:168-211",
            "rot0": [[12, 1, 0, 4], [10, 1, 2, 6]], "rot1": [[12, 1, 0, 233], [10, 1, 2, 29]],
            "maj": [[12, 1, 0, 0], [10, 1, 2, 2]], "ch": [[12, 1, 0, 0], [10, 1, 2, 0]],
            "decomposition_k10": [[682, 10, 2, 2], [950, 11, 1, 2]], "lengths": {"synthesis": 256, "decomposition": 1024}},
    }


def proof_vector(name, fx, seed, opener="gwc"):
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(seed), instances=fx["instances"], opener=opener)
    return {"circuit": name, "k": fx["circuit"].k, "toxic_waste_seed": 0x6371, "rng_seed": seed, "opener": opener,
            "proof_len": len(tr.proof), "proof_sha256": hashlib.sha256(tr.proof).hexdigest(), "proof_hex": tr.proof.hex()}


def oracle_vectors():
    out = {}
    # best_multiexp: seeded scalars / points (tests/util.py), result as affine x, y
    sc, pts = random_scalars(64, 11), random_points(64, 12)
    x, y = B.jac_to_affine(OP.best_multiexp(sc, pts))
    out["best_multiexp_n64"] = {"scalars_seed": 11, "points_seed": 12, "x": hexl(x), "y": hexl(y)}
    # best_fft, k = 6, omega of the 2^6 domain
    dom = OP.EvaluationDomain(3, 6)
    a = random_scalars(64, 13)
    r = OP.best_fft(list(a), dom.omega, 6)
    out["best_fft_k6"] = {"input_seed": 13, "output_sha256": hashlib.sha256(b"".join(B.to_repr(v) for v in r)).hexdigest(), "first": hexl(r[0]), "last": hexl(r[-1])}
    # eval_polynomial / kate_division
    z = random_scalars(1, 14)[0]
    q = OP.kate_division(a, z)
    out["poly_k6"] = {"poly_seed": 13, "point_seed": 14, "eval": hexl(OP.eval_polynomial(a, z)),
                      "kate_division_sha256": hashlib.sha256(b"".join(B.to_repr(v) for v in q)).hexdigest()}
    # full proofs (k = 5): the circuits of tests/plonk_fixtures.py
    proofs = []
    for name, kw in (("chain", {}), ("chain", {"degree5": True}), ("chain", {"with_lookup": True}), ("chain", {"lookup_expr": True}),
                     ("chain", {"plookup": True})):
        fx = oracle_env(5, **kw)
        v = proof_vector(name, fx, 77)
        v["variant"] = kw
        proofs.append(v)
    fx = oracle_env(5, with_lookup=True)
    v = proof_vector("chain", fx, 21, opener="shplonk")
    v["variant"] = {"with_lookup": True}
    proofs.append(v)
    fx = oracle_env(5, builder=plonk_api_circuit)
    v = proof_vector("plonk_api", fx, 3)
    v["variant"] = {}
    proofs.append(v)
    out["proofs"] = proofs
    return out


if __name__ == "__main__":
    json.dump(reference_kats(), open(os.path.join(HERE, "reference_kats.json"), "w"), indent=1)
    json.dump(oracle_vectors(), open(os.path.join(HERE, "oracle_vectors.json"), "w"), indent=1)
    print("written", os.listdir(HERE))
