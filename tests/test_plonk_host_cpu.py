"""CPU: host-side pieces of the general-PLONK path that need no GPU -- the copy-constraint bookkeeping
exported by libcq_halo2 (`permutation::keygen::Assembly`) against the oracle's restatement, and the
`ConstraintSystem` mirror / gate-program lowering."""
import numpy as np

from oracle import bn254 as B
from oracle import plonk as PL
from tests.plonk_fixtures import chain_circuit, to_backend_cs


def test_permutation_assembly_matches_oracle():
    from sha2_on_cq_halo2_amd import _lib
    from sha2_on_cq_halo2_amd import plonk as GP

    lib = _lib.load()
    n = 16
    gcols = [GP.Column(GP.COL_ADVICE, 0), GP.Column(GP.COL_ADVICE, 1), GP.Column(GP.COL_FIXED, 0)]
    ocols = [(PL.ADVICE, 0), (PL.ADVICE, 1), (PL.FIXED, 0)]
    ga = GP.PermutationAssembly(lib, n, gcols)
    oa = PL.Assembly(n, ocols)
    rng = B.Xoshiro256ss(3)
    for _ in range(60):
        lc, rc = rng.next_u64() % 3, rng.next_u64() % 3
        lr, rr = rng.next_u64() % n, rng.next_u64() % n
        ga.copy(gcols[lc], lr, gcols[rc], rr)
        oa.copy(ocols[lc], lr, ocols[rc], rr)
        assert ga.mapping.tolist() == [[list(c) for c in col] for col in oa.mapping]
    try:
        ga.copy(gcols[0], n, gcols[1], 0)
        assert False
    except IndexError:
        pass
    try:
        ga.copy(GP.Column(GP.COL_INSTANCE, 0), 0, gcols[1], 0)
        assert False
    except ValueError:
        pass


def test_constraint_system_mirror_and_gate_programs():
    from sha2_on_cq_halo2_amd import plonk as GP

    fx = chain_circuit(5, degree5=True)
    cs = to_backend_cs(fx["circuit"], {})
    assert cs.degree() == 5 and cs.blinding_factors() == 5 and cs.minimum_rows() == 8
    constants = []
    progs = [g.compile(constants) for g in cs.gates]
    # interpret the postfix programs on the host and compare with the oracle's expression evaluation
    rng = B.Xoshiro256ss(11)
    vals = {}

    def get(kind, col, rot):
        return vals.setdefault((kind, col, rot), B.fr_random(rng))

    kinds = {GP.GATE_ADVICE: PL.ADVICE, GP.GATE_FIXED: PL.FIXED, GP.GATE_INSTANCE: PL.INSTANCE}
    for prog, g in zip(progs, fx["circuit"].gates):
        st, pc = [], 0
        while pc < len(prog):
            op, arg = prog[pc] & 0xFF, prog[pc] >> 8
            pc += 1
            if op == GP.GATE_CONST:
                st.append(constants[arg])
            elif op in kinds:
                rot = prog[pc] - (1 << 32) if prog[pc] >> 31 else prog[pc]
                pc += 1
                st.append(get(kinds[op], arg, rot))
            elif op == GP.GATE_NEG:
                st.append(-st.pop() % B.R_MOD)
            elif op == GP.GATE_ADD:
                b_, a_ = st.pop(), st.pop()
                st.append((a_ + b_) % B.R_MOD)
            elif op == GP.GATE_MUL:
                b_, a_ = st.pop(), st.pop()
                st.append(a_ * b_ % B.R_MOD)
            else:
                st.append(st.pop() * constants[arg] % B.R_MOD)
        assert len(st) == 1 and st[0] == PL.expr_eval(g, get)


def test_host_glv_scalar_mul_matches_double_and_add(tmp_path):
    """csrc/glv.hpp: the endomorphism-split, 4-bit-window scalar multiplication the prover uses for the f commitments
    (k = k1 + k2 lambda, phi(x, y) = (zeta x, y); constants of bn256/curve.rs:69-83, fq.rs ZETA) against plain
    double-and-add over the host group law, on edge scalars (0, 1, -1, lambda, (r - 1) / 2 and its successor) and 300 random
    ones; and phi(P) = lambda P itself.  HIP-free: g++ on tests/host/glv_check.cpp."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "glv_check")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "sha2_on_cq_halo2_amd", "csrc"),
                        os.path.join(root, "tests", "host", "glv_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_variable_time_inversion_matches_the_constant_time_one(tmp_path):
    """field.hpp: the variable-time safegcd inversion (the lone lane of `batch_invert_kernel`: trailing zeros of g in one shift,
    the next 4-6 bits cleared by one multiple of f, batches until g = 0) returns the words of the constant-time one for both
    BN254 fields on 0, +-1, small values and their negatives, 2^i and 2^i - 1 for every i, 20 000 random elements, equals
    Fermat's x^(p-2) on a chain of values, and x * inv(x) = 1.  HIP-free: g++ on tests/host/inv_check.cpp."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "inv_check")
    r = subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "sha2_on_cq_halo2_amd", "csrc"),
                        os.path.join(root, "tests", "host", "inv_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok") and "0 disagreements" in r.stdout, r.stdout + r.stderr
