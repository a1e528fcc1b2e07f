import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ctx = Context(0)
t = time.time(); wl = ShaCqWorkload(ctx, k); ctx.sync(); print("setup %.2f s, blocks=%d words=%d usable=%d" % (time.time()-t, wl.blocks, wl.nwords, wl.pk.usable_rows), flush=True)
for i in range(4):
    t = time.time(); proof = wl.prove(seed=1); dt = time.time() - t
    print("prove k=%d: %.1f ms  proof %d bytes  msm Mscalar/s %.1f" % (k, dt*1e3, len(proof), wl.msm_scalars_per_proof()/dt/1e6), flush=True)
import hashlib; print("proof sha256", hashlib.sha256(proof).hexdigest())
