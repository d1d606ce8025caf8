import sys; sys.path.insert(0, '/root/repo')
import torch, bench
dev = torch.device("cuda")
for cp in (0.005, 0.0, 0.005, 0.0):
    r = bench.raft_leg(torch, dev, conflict_p=cp)
    print("conflict_p", cp, "us/tick batched %.2f" % r["us_per_tick"], "per-call replies kernel %.2f" % r["roofline"]["avg_launch_us"], "per-call tick %.2f" % r["one_call_per_handler"]["us_per_tick"])
