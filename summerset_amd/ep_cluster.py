"""Closed-loop EPaxos cluster on the device: R `EPaxosReplicaGroup` objects (one per replica id, every one holding all G
groups) wired into BASELINE config 5's tick -- EVERY replica proposes one instance per group per tick; the PreAccepts fan
out to all peers; their replies come back; each command leader decides fast / slow path; slow-path Accepts and their
replies; CommitNotices to all peers.  Message order: senders ascending, each receiver handles one sender's message at a
time (the order tests/ep_cluster.py fixes for the numpy backends; `tests/test_zz_ep_cluster_gpu.py` checks this driver
against that one).  Every message stays a device tensor between the handlers: the only host work per tick is the
handler calls themselves (R proposals + R (R - 1) PreAccepts + R reply tallies + R accept-reply tallies + R (R - 1)
CommitNotices, plus Accept rounds when a slow path was taken)."""

NO_KEY = 0xFF
NONE = -1                                       # Option::None in a DepSet (0xFFFFFFFF as int32)


def tick(reps, keys, drop=None, always_accept_round=False, phase_major=False):
    """reps[r]: EPaxosReplicaGroup of replica r; keys[r]: uint8 [G] device tensor, replica r's proposal per group
    (0xFF = none); drop[(s, q)] (optional): bool [G], the PreAccept from s to q is lost (with its reply).
    phase_major: the command leaders' part phase by phase (every leader's PreAcceptReplies, then every Accept round, every
    AcceptReply tally, every CommitNotice) instead of leader by leader -- the order `smr_ep_cluster_set_mode(c, 2)` runs.
    Returns per command leader dict(col, proposed, decision, committed, seq, deps) of device tensors."""
    import torch
    R = len(reps)
    G = keys[0].shape[0]
    dev = keys[0].device
    u8 = lambda v: torch.full((G,), v, dtype=torch.uint8, device=dev)
    i64 = lambda v: torch.full((G,), v, dtype=torch.int64, device=dev)
    pa = [reps[r].handle_req_batch(keys[r], None) for r in range(R)]
    rep = {}
    for q in range(R):
        for s in range(R):
            if s == q:
                continue
            fl = pa[s]["flags"]
            if drop is not None and (s, q) in drop:
                fl = torch.where(drop[(s, q)], torch.zeros_like(fl), fl)
            rep[(q, s)] = reps[q].handle_msg_pre_accept(dict(flags=fl, peer=u8(s), col=pa[s]["col"], ballot=i64(s + 1), seq=pa[s]["seq"],
                                                             deps=pa[s]["deps"], key=keys[s]))
    zero_f, zero_b = torch.zeros(G, dtype=torch.uint8, device=dev), torch.zeros(G, dtype=torch.int64, device=dev)
    none_d = torch.full((R, G), NONE, dtype=torch.int32, device=dev)
    dec, slow, aflags, aballot, committed = [None] * R, [None] * R, [None] * R, [None] * R, [None] * R

    def replies(s):
        flags = torch.stack([zero_f if q == s else rep[(q, s)]["flags"] for q in range(R)])
        ballot = torch.stack([zero_b if q == s else rep[(q, s)]["ballot"] for q in range(R)])
        seq = torch.stack([zero_b if q == s else rep[(q, s)]["seq"] for q in range(R)])
        deps = torch.stack([none_d if q == s else rep[(q, s)]["deps"] for q in range(R)])
        dec[s] = reps[s].handle_msg_pre_accept_reply(pa[s]["col"], ballot, seq, deps, flags)
        slow[s] = (dec[s]["decision"] == 2).to(torch.uint8)

    def accepts(s):
        aflags[s] = torch.zeros((R, G), dtype=torch.uint8, device=dev)
        aballot[s] = torch.zeros((R, G), dtype=torch.int64, device=dev)
        if always_accept_round or bool(slow[s].any()):           # (.any() is the one device -> host read of the tick)
            for q in range(R):
                if q == s:
                    continue
                ar = reps[q].handle_msg_accept(dict(flags=slow[s], peer=u8(s), col=pa[s]["col"], ballot=i64(s + 1), seq=dec[s]["seq"],
                                                    deps=dec[s]["deps"], key=keys[s]))
                aflags[s][q] = ar["flags"]
                aballot[s][q] = ar["ballot"]

    def accept_replies(s):
        acc = reps[s].handle_msg_accept_reply(pa[s]["col"], aballot[s], aflags[s])
        committed[s] = ((dec[s]["decision"] == 3) | (acc["committed"] == 1)).to(torch.uint8)

    def commits(s):
        for q in range(R):
            if q == s:
                continue
            reps[q].handle_msg_commit_notice(dict(flags=committed[s], peer=u8(s), col=pa[s]["col"], ballot=i64(s + 1), seq=dec[s]["seq"],
                                                  deps=dec[s]["deps"], key=keys[s]))

    phases = (replies, accepts, accept_replies, commits)
    if phase_major:
        for ph in phases:
            for s in range(R):
                ph(s)
    else:
        for s in range(R):
            for ph in phases:
                ph(s)
    return [dict(col=pa[s]["col"], proposed=pa[s]["flags"], decision=dec[s]["decision"], committed=committed[s], seq=dec[s]["seq"],
                 deps=dec[s]["deps"]) for s in range(R)]


class EPaxosCluster:
    """The same closed loop as ONE C-ABI call per tick (`smr_ep_cluster_tick`, include/summerset_hip.h) -- and, by default, as
    ONE LAUNCH: a block of the kernel is the R replicas (a wavefront each) of 64 groups, the handlers are steps of that kernel,
    the peers' replies cross wavefronts through the cluster's reply stacks behind block barriers, execution runs behind its
    handler on the same lane.  `per_handler_launches=True` keeps round 2's path: the handler kernels launched back to back
    by the library (115 launches per tick at R = 5 with execution on) -- the decomposition the one-launch tick is checked
    against.  No Python, no torch glue and no host read inside a tick either way (the Accept round runs wherever a leader of
    the tile took the slow path).  `reps` stay usable on their own (dump, exec_dump, the per-handler calls)."""

    def __init__(self, reps, per_handler_launches=False, phase_major=False):
        import ctypes as C
        from . import _lib
        self.reps, self.R, self.G = list(reps), len(reps), reps[0].G
        self._L = _lib.load()
        arr = (C.c_void_p * self.R)(*[r._h for r in self.reps])
        h = C.c_void_p()
        _lib.check(self._L.smr_ep_cluster_create(arr, self.R, C.byref(h)))
        self._h = h
        self._held = None                      # the last tick's drop masks: alive until the next tick's have replaced them
        if per_handler_launches or phase_major:                  # (phase_major: `tick(.., phase_major=True)`'s order, see there)
            _lib.check(self._L.smr_ep_cluster_set_mode(self._h, (1 if per_handler_launches else 0) | (2 if phase_major else 0)))

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_ep_cluster_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def batch_stats(self):
        """(replica, group, tick) lanes whose PreAccept / CommitNotice phase left the batched step of the phase-by-phase
        one-launch tick and ran handler by handler (`smr_ep_cluster_batch_stats`)"""
        import ctypes as C
        from . import _lib
        out = (C.c_uint64 * 2)()
        _lib.check(self._L.smr_ep_cluster_batch_stats(self._h, out))
        return dict(pre_accept_lanes_one_by_one=int(out[0]), commit_lanes_one_by_one=int(out[1]))

    def new_outputs(self, dev):
        """one set of the tick's output arrays (per command leader); pass it back as `tick(..., out=)` to reuse it"""
        import torch
        R, G = self.R, self.G
        return [dict(proposed=torch.empty(G, dtype=torch.uint8, device=dev), col=torch.empty(G, dtype=torch.int32, device=dev),
                     seq0=torch.empty(G, dtype=torch.int64, device=dev), deps0=torch.empty((R, G), dtype=torch.int32, device=dev),
                     decision=torch.empty(G, dtype=torch.uint8, device=dev), committed=torch.empty(G, dtype=torch.uint8, device=dev),
                     seq=torch.empty(G, dtype=torch.int64, device=dev), deps=torch.empty((R, G), dtype=torch.int32, device=dev))
                for _ in range(R)]

    def tick(self, keys, drop=None, stream=None, out=None):
        """keys[r]: uint8 [G] device tensor (0xFF = no proposal); drop[(s, q)] (optional): bool / uint8 [G], the PreAccept
        from s to q is lost.  Returns per command leader dict(col, proposed, decision, committed, seq, deps) like `tick`
        (fresh arrays, or the caller's `out` from `new_outputs`)."""
        import ctypes as C
        import torch
        from . import _lib
        R, dev = self.R, keys[0].device
        outs = (_lib.EpClusterOut * R)()
        res = out if out is not None else self.new_outputs(dev)
        for s in range(R):
            for n, _ in _lib.EpClusterOut._fields_:
                setattr(outs[s], n, res[s][n].data_ptr())
        kp = (C.c_void_p * R)(*[k.data_ptr() for k in keys])
        dp, masks = None, None
        if drop:
            # uint8 copies of bool masks are temporaries of THIS call, and the call only enqueues work: they stay referenced
            # by the object until the next tick (on whatever stream) has replaced them, so the allocator cannot hand their
            # memory out while a kernel of this tick may still read it
            masks = {k: (v if v.dtype == torch.uint8 else v.to(torch.uint8)).contiguous() for k, v in drop.items()}
            dp = (C.c_void_p * (R * R))(*[(masks[(s, q)].data_ptr() if (s, q) in masks else None) for s in range(R) for q in range(R)])
        _lib.check(self._L.smr_ep_cluster_tick(self._h, kp, dp, outs, _lib.stream_ptr(stream)))
        self._held = (masks, list(keys), res)
        return [dict(col=o["col"], proposed=o["proposed"], decision=o["decision"], committed=o["committed"], seq=o["seq"], deps=o["deps"])
                for o in res]
