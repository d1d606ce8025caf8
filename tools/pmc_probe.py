"""Tiny workload for rocprofv3 --pmc passes: a few ticks of the bench shape and three RS encodes
(the encode is the calibration point: its byte counts are known exactly)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from summerset_amd import MultiPaxosCluster, RSCodewordBatch, stream
dev = torch.device("cuda")
G, R, S, W, H = 65536, 5, 32, 512, 4
cap = W + 4
eng = MultiPaxosCluster(G, R, W, win_reserve=W // 8, outbox_cap=cap); eng.preset_leader(0)
st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=8, drop_p=0.1, timeout_frac=0.0, hb_every=H, rand_rows=S + 4, max_drop=2)
for t in range(8):
    x = st.tick(t)
    d = {k: torch.from_numpy(v).to(dev) for k, v in x.items() if isinstance(v, np.ndarray)}
    eng.tick(timeout_rep=None, timeout_src=None, req_target=d["req_target"], req_cnt=d["req_cnt"], req_val=d["req_val"],
             ackctl=d["ackctl"], heartbeat=st.heartbeat(t))
torch.cuda.synchronize()
data = torch.randint(0, 256, (65536, 4099), dtype=torch.uint8, device=dev)     # 268 MB: past the 256 MB L3
cw = RSCodewordBatch.from_data(data, 3, 2)
for _ in range(3):
    cw.compute_parity()
torch.cuda.synchronize()
print("done")
