"""Sanitizer passes over the CPU side (this container only -- no GPU sanitizers on this pool):
  * oracle/cq_oracle.c built with -fsanitize=address,undefined (`make -C oracle asan`) under the oracle's own tests
    (C vs Python agreement on fields, MSM, FFT, a full create_proof, the pairing);
  * the prover's HIP-free host glue (worker pool, jump-ahead RNG fill, transcript hash; tests/host/host_stress.cpp) under
    -fsanitize=thread and -fsanitize=address,undefined, in the threading patterns csrc/prover.hip uses.
Any sanitizer report fails the test (halt_on_error / non-zero exit)."""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gcc_lib(name):
    return subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True, check=True).stdout.strip()


def test_c_oracle_under_asan_ubsan():
    lib = os.path.join(ROOT, "oracle", "libcq_oracle_asan.so")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.exists(lib), r.stderr
    env = dict(os.environ, LD_PRELOAD=_gcc_lib("libasan.so") + " " + _gcc_lib("libubsan.so"), CQ_ORACLE_LIB=lib, OMP_NUM_THREADS="4",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",  # (the interpreter itself is not leak-clean)
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    code = ("import sys, pytest; from oracle import cbind as OC; "
            "assert OC.lib()._name.endswith('libcq_oracle_asan.so'), OC.lib()._name; "
            "sys.exit(pytest.main(['-x', '-q', '-p', 'no:cacheprovider', 'tests/test_oracle_c.py', 'tests/test_oracle_prover.py', "
            "'tests/test_oracle_pairing.py']))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=ROOT, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_glue_under_sanitizers(tmp_path, sanitizer):
    exe = str(tmp_path / "host_stress")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=" + sanitizer, "-Wall", "-Wextra", "-Werror",
                        "-I", os.path.join(ROOT, "sha2_on_cq_halo2_amd", "csrc"), os.path.join(ROOT, "tests", "host", "host_stress.cpp"),
                        "-o", exe, "-pthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr
    # the transcript hash against hashlib (RFC 7693 with the reference's personalisation, transcript.rs:179-184)
    kv = dict(line.split(" ", 1) for line in r.stdout.strip().splitlines() if " " in line)
    msg = bytes((i * 7 + 1) & 0xFF for i in range(1000))
    for n in (0, 1, 128, 129, 1000):
        tag = "blake2b_empty" if n == 0 else "blake2b_%d" % n
        assert kv[tag] == hashlib.blake2b(msg[:n], digest_size=64, person=b"Halo2-Transcript").hexdigest(), tag
