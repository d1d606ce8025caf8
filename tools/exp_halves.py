"""Experiment: the job's groups as P independent sub-populations (one MultiPaxosCluster each, G / P groups, the same global
stream: group_base = the part's offset), each on its own HIP stream, so one part's latency-bound round kernel overlaps
another part's.  Prints slots/s and ms per tick of the whole population for P in --parts.
  python tools/exp_halves.py --parts 1 2 4 [--timeouts 0] [--threads]"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--parts", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--groups", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--timeouts", type=float, default=0.01)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--straggler-ticks", type=int, default=4)
    ap.add_argument("--threads", action="store_true", help="one host thread per part (the C call releases the GIL)")
    args = ap.parse_args()
    import torch
    from summerset_amd import MultiPaxosCluster, stream
    dev = torch.device("cuda", 0)
    G, R, S, W, H = args.groups, 5, 32, 512, 4
    cap, n_ticks, POOL = W + 4, args.warmup + args.steps, 4
    for P in args.parts:
        Gp = G // P
        parts = []
        for p in range(P):
            eng = MultiPaxosCluster(Gp, R, W, win_reserve=W // 8, outbox_cap=cap, straggler_ticks=args.straggler_ticks)
            eng.preset_leader(0)
            st = stream.MultiPaxosStream(Gp, R, S, cap=cap, n_ticks=n_ticks, drop_p=0.1, timeout_frac=args.timeouts, hb_every=H,
                                         rand_rows=S + 4, max_drop=2, timeout_span=max(n_ticks, 72), group_base=p * Gp)
            pool = [{k: torch.from_numpy(v).to(dev) for k, v in st.tick(t).items() if k in ("req_cnt", "req_val", "ackctl")} for t in range(POOL)]
            ev = [{k: torch.from_numpy(v).to(dev) for k, v in st.tick_events(t).items() if isinstance(v, np.ndarray)} for t in range(n_ticks)]
            fired = [bool((st.timeout_tick == t).any()) for t in range(n_ticks)]
            parts.append((eng, st, pool, ev, fired, torch.cuda.Stream(device=dev)))

        def targs(part, t):
            eng, st, pool, ev, fired, _ = part
            p_, e = pool[t % POOL], ev[t]
            return dict(timeout_rep=e["timeout_rep"] if fired[t] else None, timeout_src=e["timeout_src"] if fired[t] else None,
                        req_target=e["req_target"], req_cnt=p_["req_cnt"], req_val=p_["req_val"], ackctl=p_["ackctl"], heartbeat=st.heartbeat(t))

        def drive(part, t0, t1):
            for b0 in range(t0, t1, args.batch):
                part[0].run_ticks([targs(part, t) for t in range(b0, min(b0 + args.batch, t1))], stream=part[5].cuda_stream)

        def run(t0, t1):
            if args.threads and P > 1:
                ths = [threading.Thread(target=drive, args=(part, t0, t1)) for part in parts]
                [t.start() for t in ths]
                [t.join() for t in ths]
            else:                                   # one host thread: batches dealt round robin over the parts
                for b0 in range(t0, t1, args.batch):
                    for part in parts:
                        drive(part, b0, min(b0 + args.batch, t1))

        run(0, args.warmup)
        torch.cuda.synchronize()
        c0 = sum(part[0].counters(r)["commits"] for part in parts for r in range(R))
        t0 = time.perf_counter()
        run(args.warmup, n_ticks)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        c1 = sum(part[0].counters(r)["commits"] for part in parts for r in range(R))
        print("parts %d threads %d timeouts %g: %.3e slots/s, %.4f ms per tick of %d groups" % (P, int(args.threads), args.timeouts, (c1 - c0) / el, el / args.steps * 1e3, G), flush=True)
        del parts
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
