"""Per-step wall-clock stamps of the one-launch EPaxos cluster tick (a -DEPC_STAMPS build: tools/build_file_variant.sh epc_stamps
ep_engine.hip -DEPC_STAMPS): where a block's time goes, step by step and wavefront by wavefront.  100 MHz counter (10 ns).
usage: SUMMERSET_HIP_LIB=summerset_amd/variants/libsummerset_hip_epc_stamps.so python tools/dbg_epc_stamps.py [execute 0|1] [pm]
(pm: the leaders' steps phase by phase, the order bench.py's best figure runs)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from summerset_amd import EPaxosReplicaGroup, _lib, ep_cluster

EXEC = len(sys.argv) > 1 and sys.argv[1] == "1"
PM = len(sys.argv) > 2 and sys.argv[2] == "pm"
dev = torch.device("cuda")
G, R, W, K = 65536, 5, 32, 64
reps = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=EXEC) for r in range(R)]
cl = ep_cluster.EPaxosCluster(reps, phase_major=PM)
rng = np.random.default_rng(0x5EED5EED)
zipf = 1.0 / np.arange(1, K + 1) ** 0.99
zipf /= zipf.sum()
outs = cl.new_outputs(dev)
for t in range(8):
    keys = [torch.from_numpy(rng.choice(K, G, p=zipf).astype(np.uint8)).to(dev) for _ in range(R)]
    cl.tick(keys, out=outs)
torch.cuda.synchronize()
L = _lib.load()
L.smr_dbg_epc_stamps.restype = C.c_int
buf = np.zeros(8 * 5 * 64, np.uint64)
n = L.smr_dbg_epc_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
s = buf.reshape(8, 5, 64)
names = ["P"] + ["A%d" % i for i in range(R)] + ([x % l for x in ("R%d", "Ac%d", "AR%d", "C%d") for l in range(R)] if PM else
                                               [x for l in range(R) for x in ("R%d" % l, "Ac%d" % l, "AR%d" % l, "C%d" % l)]) + ["end"]
print("execute =", EXEC, " (us since the block's first stamp; one row per wavefront q; block = stamped block k)")
base = int(s[0][:, 0].min())
print("block starts / ends (us): " + "  ".join("b%d: %.0f-%.0f" % (k * 128 + 5, (int(s[k][:, 0].min()) - base) / 100.0, (int(s[k][:, :len(names)].max()) - base) / 100.0) for k in range(8)))
for k in (0, 3, 7):
    t0 = s[k][:, 0].min()
    print("block", k * 128 + 5, "starts (relative to block 5) at %.1f us" % ((int(t0) - int(s[0][:, 0].min())) / 100.0))
    for q in range(R):
        row = [(int(s[k][q][t]) - int(t0)) / 100.0 for t in range(len(names))]
        print("  q%d " % q + " ".join("%s=%.0f" % (nm, v) for nm, v in zip(names, row)))
    d = [(int(s[k][:, t + 1].max()) - int(s[k][:, t].max())) / 100.0 for t in range(len(names) - 1)]
    print("  step durations (max over wavefronts): " + " ".join("%s:%.1f" % (nm, v) for nm, v in zip(names, d)))
# sub-stamps of ONE PreAccept handler (third acceptor step; fourth for wavefront 2), slots 32..38 of each wavefront's row:
# 0 step entry, 1 first round of loads landed, 2 row padded, 3 sequence numbers landed + max, 4 record built, 5 stores issued
# (and landed: the stamp waits), 6 reply in LDS
for k in (0, 3):
    for q in range(R):
        sub = [int(x) for x in s[k][q][32:48]]
        if not sub[0]: continue
        print("  block %d q%d PreAccept handler sub-stamps (us since entry): " % (k * 128 + 5, q) +
              " ".join("%d:%s" % (j, ("%.2f" % ((sub[j] - sub[0]) / 100.0)) if sub[j] else "-") for j in range(7)))
        # one CommitNotice step: 8 entry, 9 handler done (stores landed), 10 the hinted attempt's loads landed, 11 hinted attempt done,
        # 12 / 13 the general walk entered / done (either a miss of the hint or another row's bar), 15 step done
        if sub[8]:
            print("  block %d q%d CommitNotice step sub-stamps (us since entry): " % (k * 128 + 5, q) +
                  " ".join("%d:%s" % (j, ("%.2f" % ((sub[j] - sub[8]) / 100.0)) if sub[j] else "-") for j in range(8, 16)))
