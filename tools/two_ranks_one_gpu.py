"""Can smr_comm_exchange meet a SECOND rank on a one-GPU box?  Two processes, both on device 0, bootstrap over gloo, then the
library's communicator (RCCL) with world = 2.  NCCL refuses two ranks on one device ("Duplicate GPU detected"); whether this
RCCL build does is what this script finds out.  Prints one JSON line per rank."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
out = {"rank": rank, "world": world}
try:
    from summerset_amd import comm
    c = comm.Comm.from_torch_distributed("cpu")
    out["init"] = "ok"
    seg = [[0 if (s + 2 * d) % 5 == 1 else 17 * s + 5 * d + 3 for d in range(world)] for s in range(world)]
    pat = lambda s, d, n: ((torch.arange(n, dtype=torch.int64) * 7 + 31 * s + 101 * d) % 251).to(torch.uint8)
    sbuf = torch.cat([pat(rank, d, n) for d, n in enumerate(seg[rank])]).to(dev)
    rs = [seg[s][rank] for s in range(world)]
    rbuf = torch.full((sum(rs) + 8,), 0xEE, dtype=torch.uint8, device=dev)
    c.exchange(sbuf, seg[rank], rbuf, rs)
    torch.cuda.synchronize()
    want = torch.cat([pat(s, rank, n) for s, n in enumerate(rs)])
    out["exchange"] = "ok" if torch.equal(rbuf[:-8].cpu(), want) and bool((rbuf[-8:] == 0xEE).all()) else "WRONG BYTES"
    t = torch.tensor([rank + 1], dtype=torch.int64, device=dev)
    c.all_reduce(t, comm.SUM)
    out["all_reduce"] = int(t.item())
    out["info"] = c.info()
    c.close()
except Exception as e:                                             # noqa: BLE001
    out["error"] = "%s: %s" % (type(e).__name__, e)
print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
