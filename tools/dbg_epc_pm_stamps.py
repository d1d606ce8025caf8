"""Per-phase wall-clock stamps of the batched phase-by-phase EPaxos cluster tick (ep_cluster_tick_pm_kernel; a -DEPC_STAMPS build:
tools/build_file_variant.sh epc_stamps ep_engine.hip -DEPC_STAMPS).  100 MHz counter.
usage: SUMMERSET_HIP_LIB=summerset_amd/variants/libsummerset_hip_epc_stamps.so python tools/dbg_epc_pm_stamps.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from summerset_amd import EPaxosReplicaGroup, _lib, ep_cluster

dev = torch.device("cuda")
G, R, W, K = 65536, 5, 32, 64
reps = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=True) for r in range(R)]
cl = ep_cluster.EPaxosCluster(reps, phase_major=True)
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
L.smr_dbg_epc_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
s = buf.reshape(8, 5, 64)
names = ["P", "P.done", "A", "A.done", "R", "R.done", "Ac", "AR", "AR.done", "C", "C.done", "scalars stored"]
base = int(s[0][:, 0].min())
print("block starts / ends (us): " + "  ".join("b%d: %.0f-%.0f" % (k * 128 + 5, (int(s[k][:, 0].min()) - base) / 100.0, (int(s[k][:, :len(names)].max()) - base) / 100.0) for k in range(4)))
for k in (0, 1, 2, 3):
    t0 = int(s[k][:, 0].min())
    for q in range(R):
        print("  b%d q%d " % (k * 128 + 5, q) + " ".join("%s=%.1f" % (nm, (int(s[k][q][t]) - t0) / 100.0) for t, nm in enumerate(names)))
for q in range(R):
    t0 = int(s[0][q][40])
    print("  cleanup kernel, replica %d, wavefront 0: " % q + " ".join("%d:%.1f" % (k, (int(s[0][q][40 + k]) - t0) / 100.0) for k in range(12) if s[0][q][40 + k]))
print(cl.batch_stats())
