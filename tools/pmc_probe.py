"""Small workload for rocprofv3 passes (--kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE -- each its own run):
ticks of the bench shape on the DEFAULT workload of bench.py (10 % ack loss, 1 % of the groups changing leader inside the
run) or the steady one (--timeouts 0), then -- the calibration point, its byte counts are known exactly -- three RS(3,2)
encodes of 65 536 codewords (268 MB, past the 256 MB L3), and with --extra the Raft / EPaxos reply kernels of their legs
and the wire ingest kernels of theirs.
usage: python tools/pmc_probe.py [--timeouts 0.01] [--ticks 16] [--extra]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from summerset_amd import MultiPaxosCluster, RSCodewordBatch, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--timeouts", type=float, default=0.01)
ap.add_argument("--ticks", type=int, default=16)
ap.add_argument("--extra", action="store_true")
ap.add_argument("--straggler-ticks", type=int, default=workloads.HEADLINE["straggler_ticks"], help="as bench.py: ticks a group in a leader change stays on the straggler list")
ap.add_argument("--batch", type=int, default=workloads.HEADLINE["batch"], help="as bench.py: ticks per smr_mp_run_ticks call (0: one smr_mp_tick call per tick)")
a = ap.parse_args()
dev = torch.device("cuda")
G, R, S, W, H = 65536, 5, 32, 512, 4
cap = W + 4
# cluster, stream and launch mode from summerset_amd/workloads.py: what bench.py times and the BASELINE-size tests check
eng = workloads.headline_cluster(G, W=W, R=R, straggler_ticks=a.straggler_ticks)
st = workloads.headline_stream(G, a.ticks, a.timeouts, a.ticks, S=S, W=W, R=R, H=H)
pool = [{k: torch.from_numpy(v).to(dev) for k, v in st.tick(t).items() if k in ("req_cnt", "req_val", "ackctl")} for t in range(4)]
def tick_args(t):
    e = {k: torch.from_numpy(v).to(dev) for k, v in st.tick_events(t).items()}
    fired = bool((st.timeout_tick == t).any())
    return dict(timeout_rep=e["timeout_rep"] if fired else None, timeout_src=e["timeout_src"] if fired else None, req_target=e["req_target"],
                heartbeat=st.heartbeat(t), **pool[t % 4])


workloads.drive_headline(eng, tick_args, 0, a.ticks, batch=a.batch)
torch.cuda.synchronize()
# the same workload through the fused tick kernel: two launches of 16 ticks each (smr_mp_run_ticks)
eng2 = MultiPaxosCluster(G, R, W, win_reserve=W // 8, outbox_cap=cap)
eng2.preset_leader(0)
batch = []
for t in range(32):
    e = {k: torch.from_numpy(v).to(dev) for k, v in st.tick_events(t % a.ticks).items()}
    fired = bool((st.timeout_tick == t % a.ticks).any()) and t < a.ticks
    batch.append(dict(timeout_rep=e["timeout_rep"] if fired else None, timeout_src=e["timeout_src"] if fired else None,
                      req_target=e["req_target"], heartbeat=st.heartbeat(t), **pool[t % 4]))
eng2.run_ticks(batch)
torch.cuda.synchronize()
del eng2
data = torch.randint(0, 256, (65536, 4099), dtype=torch.uint8, device=dev)     # 268 MB: past the 256 MB L3
cw = RSCodewordBatch.from_data(data, 3, 2)
for _ in range(3):
    cw.compute_parity()
torch.cuda.synchronize()
if a.extra:
    import bench
    bench.raft_leg(torch, dev)
    bench.epaxos_leg(torch, dev)
    torch.cuda.synchronize()
    bench.wire_ingest_leg(torch, dev)             # the peer-traffic parser's two passes
    torch.cuda.synchronize()
    bench.reply_ingest_leg(torch, dev)            # round 3: Raft replies parsed into the [R][G] arrays
    torch.cuda.synchronize()
    bench.epaxos_cluster_leg(torch, dev, ticks=4)  # round 3: the one-launch cluster tick (ep_cluster_tick_kernel)
    torch.cuda.synchronize()
    bench.rspaxos_leg(torch, dev, ticks=8, warmup=4)   # round 3: the one-pass from_data + encode + fan-out (rs_from_data_xtime) and the rsp_* handlers
    torch.cuda.synchronize()
    bench.rspaxos_payload_leg(torch, dev, ticks=8, warmup=4)   # round 4: the payload store (ps_put_kernel<3>, ps_plan* / ps_bytes* kernels)
    torch.cuda.synchronize()
print("done")
