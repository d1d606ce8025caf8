"""Round 6 probe: config 5's EPaxos tick (phase by phase, two launches: the batched kernel + the listed lanes one by one) with
the 65 536 groups as S independent clusters of 65 536 / S groups, each on its own stream.  Groups never talk to each other; the
second launch is ONE lane's chain (~125 us on an idle chip), so another slice's batched kernel can run beside it."""
import sys, time
import numpy as np
import torch
from summerset_amd import EPaxosReplicaGroup, ep_cluster

G, R, W, K, T = 65536, 5, 32, 64, 12
dev = torch.device("cuda:0")
rng = np.random.default_rng(0x5EED5EED)
zipf = 1.0 / np.arange(1, K + 1) ** 0.99
zipf /= zipf.sum()
keys = [[torch.from_numpy(rng.choice(K, G, p=zipf).astype(np.uint8)).to(dev) for _ in range(R)] for _ in range(T + 2)]
ref = None
for rep in range(2):
    for S in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
        Gs = G // S
        cl, outs, streams, call = [], [], [], []
        for i in range(S):
            reps = [EPaxosReplicaGroup(Gs, R, me=r, window=W, n_keys=K, execute=True) for r in range(R)]
            c = ep_cluster.EPaxosCluster(reps, phase_major=True)
            o = c.new_outputs(dev)
            ca = torch.zeros((R, Gs), dtype=torch.uint8, device=dev)
            for s_ in range(R):
                o[s_]["committed"] = ca[s_]
            cl.append(c); outs.append(o); call.append(ca); streams.append(torch.cuda.Stream(device=dev))
        ks = [[[keys[t][r][i * Gs:(i + 1) * Gs] for r in range(R)] for i in range(S)] for t in range(T + 2)]
        torch.cuda.synchronize()
        for t in range(2):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    cl[i].tick(ks[t][i], out=outs[i])
        torch.cuda.synchronize()
        cnt = [torch.zeros((), dtype=torch.int64, device=dev) for _ in range(S)]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        main = torch.cuda.current_stream()
        e0.record(main)
        for i in range(S):
            streams[i].wait_event(e0)
        for t in range(2, T + 2):
            for i in range(S):
                with torch.cuda.stream(streams[i]):
                    cl[i].tick(ks[t][i], out=outs[i])
                    cnt[i] += call[i].sum()
        for i in range(S):
            main.wait_stream(streams[i])
        e1.record(main)
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / T
        n = sum(int(c.item()) for c in cnt)
        ex = sum(int(r.exec_dump()["counters"][0]) for c in cl for r in c.reps)
        if ref is None:
            ref = (n, ex)
        print("slices %d: %.1f us per tick of all %d groups, commits %d executed %d same %s" % (S, us, G, n, ex, (n, ex) == ref), flush=True)
        for c in cl:
            c.close()
        del cl, outs, call
