"""Throughput of cq_create_proof_batch against the number of lanes (bench.py's `batched` leg on its own).
   python3 tools/batch_lanes.py [k] [instances]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = Context(0)
wl = ShaCqWorkload(ctx, k)
host = [c.download((wl.n, 4)) for c in wl.cols]
cols = [[ctx.to_device(h) for h in host] for _ in range(count)]
ptrs = [[c.ptr for c in mine] for mine in cols]
for rep in range(2):
    for lanes in (1, 2, 3, 4):
        wl.pk.create_proof_batch(ptrs[:lanes], [1 + i for i in range(lanes)], lanes=lanes)
        ctx.sync()
        t0 = time.perf_counter()
        wl.pk.create_proof_batch(ptrs, [500 + i for i in range(count)], lanes=lanes)
        dt = time.perf_counter() - t0
        print("k=%d lanes=%d: %.1f proofs/s (%.2f ms per proof)" % (k, lanes, count / dt, dt / count * 1e3), flush=True)
