"""Throughput with TWO proofs in flight on one GPU: two contexts (own streams, workloads, proving keys), one host
thread each (ctypes releases the GIL during library calls).  The latency-bound tail of one proof's MSM launches
is filled by the other proof's kernels.   python3 tools/two_in_flight.py [k] [proofs per thread]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10


def run(nthreads):
    ctxs = [Context(0) for _ in range(nthreads)]
    wls = [ShaCqWorkload(c, k, seed=0x5348413243515F + i) for i, c in enumerate(ctxs)]
    for w in wls:
        w.prove(seed=1)
        w.prove(seed=2)
    bar = threading.Barrier(nthreads + 1)

    def worker(w):
        bar.wait()
        for i in range(reps):
            w.fill_witness()
            w.prove(seed=10 + i)
        bar.wait()

    ts = [threading.Thread(target=worker, args=(w,)) for w in wls]
    for t in ts:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    for c in ctxs:
        c.close()
    return nthreads * reps / dt, dt / reps * 1e3


for nt in (1, 2, 3):
    pps, ms = run(nt)
    print("k=%d, %d in flight: %.1f proofs/s (%.2f ms per round of %d)" % (k, nt, pps, ms, nt), flush=True)
