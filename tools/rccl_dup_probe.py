"""Probe (development aid): can two ranks that share ONE GPU form an RCCL communicator on this box?
(NCCL refuses duplicate devices by default.)  Launched as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29711 tools/rccl_dup_probe.py
Prints one line per rank; a refusal shows up as an exception text, not a hang (the group has a 60 s timeout)."""
import datetime
import os
import sys

import torch
import torch.distributed as dist

rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=int(os.environ["WORLD_SIZE"]), timeout=datetime.timedelta(seconds=60),
                            device_id=torch.device("cuda", 0))
    t = torch.full((4,), float(rank + 1), device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank %d: all_reduce over a shared GPU gave %s" % (rank, t.tolist()), flush=True)
    dist.destroy_process_group()
except Exception as e:  # noqa: BLE001
    print("rank %d: refused: %s: %s" % (rank, type(e).__name__, str(e)[:400]), flush=True)
    sys.exit(0)
