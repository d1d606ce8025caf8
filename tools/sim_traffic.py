"""Bytes the kernels of the late engines touch per call, counted by the emulator (tests/hostsim: every load / store
outside a lane's stack goes through a hook).  An upper bound of a kernel's algorithmic HBM traffic -- no caches, no
merging of accesses, and every lane counted even where lanes re-read what another lane loaded -- until rocprofv3
PMC passes on the device replace it.  Usage: python tools/sim_traffic.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import hostsim
    import torch
    import rsp_cluster as rc
    from summerset_amd import EPaxosReplicaGroup, RSPaxosReplicaGroup
    with hostsim.patched() as lib:
        out = (C.c_ulonglong * 4)()

        def measure(label, G, fn):
            lib.hipsim_traffic(out, 1)
            fn()
            lib.hipsim_traffic(out, 0)
            print("%-58s %7.1f B loaded  %6.1f B stored per group  (%4.1f loads, %4.1f stores)"
                  % (label, out[0] / G, out[1] / G, out[2] / G, out[3] / G))

        G, R, W = 512, 5, 32
        # RSPaxos leader, steady state: one batch + 4 AcceptReplies (f = 1)
        e = rc.NumpyEngine(RSPaxosReplicaGroup(G, R, me=0, window=W, fault_tolerance=1), "cpu")
        e.preset_leader(0)
        b0 = (1 << 8) | 1
        ballot = np.full((R, G), b0, np.uint64)
        fl = np.ones((R, G), np.uint8); fl[0] = 0
        for t in range(3):
            val = (1 + t * G + np.arange(G)).astype(np.uint32)
            if t < 2:
                a = e.req_batch(val); e.accept_replies(a["a_slot"][0], ballot, fl)
                continue
            res = {}
            measure("rsp_req_batch_kernel", G, lambda: res.update(a=e.req_batch(val)))
            measure("rsp_accept_replies_kernel (4 replies, commit + execute)", G, lambda: e.accept_replies(res["a"]["a_slot"][0], ballot, fl))
        f = rc.NumpyEngine(RSPaxosReplicaGroup(G, R, me=1, window=W, fault_tolerance=1), "cpu")
        f.preset_leader(0)
        u8 = lambda v: np.full(G, v, np.uint8)
        measure("rsp_accept_kernel (follower)", G, lambda: f.accept(u8(1), u8(0), np.zeros(G, np.uint32), np.full(G, b0, np.uint64),
                                                                     np.arange(1, G + 1).astype(np.uint32), u8(2)))
        measure("rsp_heartbeat_kernel<0> (learns one commit)", G, lambda: f.heartbeat(u8(1), u8(0), np.full(G, b0, np.uint64),
                                                                                     np.ones(G, np.uint32), np.zeros(G, np.uint32), np.zeros(G, np.uint32)))
        # EPaxos command leader with execution: propose + 4 agreeing replies
        K = 64
        ep = EPaxosReplicaGroup(G, R, me=0, window=W, n_keys=K, execute=True)
        rng = np.random.default_rng(1)
        flags = torch.from_numpy(fl.copy()); bal = torch.ones((R, G), dtype=torch.int64)
        for t in range(4):
            key = torch.from_numpy(rng.integers(0, K, G).astype(np.uint8))
            if t < 3:
                m = ep.handle_req_batch(key)
                ep.handle_msg_pre_accept_reply(m["col"], bal, m["seq"].unsqueeze(0).repeat(R, 1).contiguous(),
                                               m["deps"].unsqueeze(0).repeat(R, 1, 1).contiguous(), flags)
                continue
            res = {}
            measure("ep_propose_kernel + ep_execute_kernel (nothing to run)", G, lambda: res.update(m=ep.handle_req_batch(key)))
            m = res["m"]
            measure("ep_pre_accept_replies_kernel<5> + ep_execute_kernel", G,
                    lambda: ep.handle_msg_pre_accept_reply(m["col"], bal, m["seq"].unsqueeze(0).repeat(R, 1).contiguous(),
                                                           m["deps"].unsqueeze(0).repeat(R, 1, 1).contiguous(), flags))


if __name__ == "__main__":
    main()
