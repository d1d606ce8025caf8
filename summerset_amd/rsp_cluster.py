"""Lock-step schedule of an RSPaxos cluster out of R per-replica handler objects (RSPaxosReplicaGroup behind the
numpy adapter below, or -- in tests -- the CPU oracle's objects with the same interface; with `spread.SpreadReplica`
around them the replicas of a group live on different ranks and every handler's output is exchanged).  One tick:
  1. HearTimeouts -> become_a_leader (step-up Heartbeat, Prepare, Reconstruct list)
  2. the client batch of the tick at its target replica -> Accepts (one shard per peer)
  3. step-up Heartbeats and the replies to them
  4. Prepares -> PrepareReply batches -> the Accepts the quorum lets the new leader send
  5. Reconstruct reads and their replies, then the Heartbeat the new leader injects behind them
  6. Accepts -> AcceptReplies -> commits (commit bar run gated on shard availability, execution)
  7. every `hb_every` ticks: the leaders' periodic Heartbeats (commit learning) and the replies
Message order: senders ascending, receivers ascending.  `drop[(kind, s, q)]` (optional): bool [G], the message
of that kind from s to q is lost in these groups."""
import numpy as np

NULL, NO_REP = 0xFFFFFFFF, 0xFF


class NumpyEngine:
    """RSPaxosReplicaGroup (device tensors) behind the RspOracle-style numpy interface"""

    def __init__(self, eng, dev):
        import torch
        self.e, self.dev, self.torch = eng, dev, torch
        self.G, self.R, self.W, self.me = eng.G, eng.R, eng.W, eng.me
        self._polls = []                                         # what every call so far executed (the engine keeps the last call's list)

    def _t(self, a):
        if a is None:
            return None
        v = a.view(np.int64) if a.dtype == np.uint64 else (a.view(np.int32) if a.dtype == np.uint32 else a)
        return self.torch.from_numpy(np.ascontiguousarray(v)).to(self.dev)

    @staticmethod
    def _n(d):
        conv = {"int64": np.uint64, "int32": np.uint32, "uint8": np.uint8}
        return {k: v.cpu().numpy().view(conv[str(v.dtype).split(".")[-1]]) for k, v in d.items()}

    def preset_leader(self, leader):
        self.e.preset_leader(leader)

    def __getattr__(self, name):
        if name in ("req_batch", "accept", "accept_replies", "become_leader", "prepare", "prepare_replies", "reconstruct",
                    "reconstruct_reply", "heartbeat", "bcast_heartbeat"):
            fn = getattr(self.e, name)

            def call(*a, **kw):
                out = fn(*[self._t(x) for x in a], **{k: self._t(v) for k, v in kw.items()})
                self._polls.append(self.e.exec_poll())
                return None if out is None else self._n(out)
            return call
        raise AttributeError(name)

    def dump(self):
        return self.e.dump()

    def is_leader(self):
        return (self.e.dump()["leader"] == self.me).astype(np.uint8)

    def take_executed(self):
        """like RspOracle.take_executed: everything since the last call, group-major, execution order per group"""
        cols = [np.concatenate([p[i] for p in self._polls]) if self._polls else np.zeros(0, np.uint32) for i in range(3)]
        self._polls = []
        k = np.argsort(cols[0], kind="stable")
        return tuple(c[k] for c in cols)


def _lost(drop, kind, s, q, G):
    if drop is None or (kind, s, q) not in drop:
        return np.zeros(G, bool)
    return drop[(kind, s, q)]


def _deliver_heartbeat(reps, s, hb_flags, ballot, commit, exec_, snap, drop):
    R, G = len(reps), len(hb_flags)
    for q in range(R):
        if q == s:
            continue
        fl = (hb_flags & ~_lost(drop, "hb", s, q, G)).astype(np.uint8)
        if not fl.any():
            continue
        rp = reps[q].heartbeat(flags=fl, peer=np.full(G, s, np.uint8), ballot=ballot, commit_bar=commit, exec_bar=exec_,
                               snap_bar=snap)
        back = (rp["reply"].astype(bool) & ~_lost(drop, "hb", q, s, G)).astype(np.uint8)
        if back.any():                                           # the follower's Heartbeat back to its leader: no reply to it
            reps[s].heartbeat(flags=back, peer=np.full(G, q, np.uint8), ballot=rp["ballot"], commit_bar=rp["commit_bar"],
                              exec_bar=rp["exec_bar"], snap_bar=rp["snap_bar"])


def _deliver_accepts(reps, s, acc, drop, log):
    """the Accepts replica s has to send (a list of up to W slots per group), one list entry at a time"""
    R, G = len(reps), len(acc["a_n"])
    for k in range(int(acc["a_n"].max()) if len(acc["a_n"]) else 0):
        live = (acc["a_n"] > k)
        slot, val = np.ascontiguousarray(acc["a_slot"][k]), np.ascontiguousarray(acc["a_val"][k])
        ballot = np.zeros((R, G), np.uint64); flags = np.zeros((R, G), np.uint8)
        for q in range(R):
            if q == s:
                continue
            fl = (live & ~_lost(drop, "accept", s, q, G)).astype(np.uint8)
            ar = reps[q].accept(flags=fl, peer=np.full(G, s, np.uint8), slot=slot, ballot=acc["a_ballot"], val=val,
                                mask=np.full(G, 1 << q, np.uint8))
            got = (ar["r_ballot"] != 0) & ~_lost(drop, "accept_reply", q, s, G)
            flags[q] = got.astype(np.uint8); ballot[q] = ar["r_ballot"]
        res = reps[s].accept_replies(slot=slot, ballot=ballot, flags=flags)
        log.append(dict(kind="commit", s=s, slot=slot, val=val, committed=res["committed"] & live.astype(np.uint8)))


def tick(reps, val, target, timeouts=None, drop=None, heartbeat=False):
    """val[G]: the tick's client batch (NULL = none), target[G]: the replica it is sent to; timeouts[r][G]:
    HearTimeout source seen by replica r (NO_REP = none).  Returns the tick's commit log."""
    R, G = len(reps), len(val)
    log = []
    u8 = lambda v: np.full(G, v, np.uint8)
    # 1. HearTimeouts
    bl = [reps[r].become_leader(timeouts[r] if timeouts is not None else u8(NO_REP)) for r in range(R)]
    # 2. client batches
    acc = [reps[r].req_batch(np.where(target == r, val, NULL).astype(np.uint32)) for r in range(R)]
    # 3. step-up heartbeats
    for s in range(R):
        if bl[s]["hb_flags"].any():
            _deliver_heartbeat(reps, s, bl[s]["hb_flags"], bl[s]["hb_ballot"], bl[s]["hb_commit"], bl[s]["hb_exec"], bl[s]["hb_snap"], drop)
    # 4. Prepare phase
    late = []
    for s in range(R):
        if not bl[s]["p_flags"].any():
            continue
        for q in range(R):
            if q == s:
                continue
            fl = (bl[s]["p_flags"] & ~_lost(drop, "prepare", s, q, G)).astype(np.uint8)
            pr = reps[q].prepare(flags=fl, peer=u8(s), trig=bl[s]["p_trig"], ballot=bl[s]["p_ballot"])
            n = np.where(_lost(drop, "prepare_reply", q, s, G), 0, pr["pr_n"]).astype(np.uint32)
            if n.any():
                log.append(dict(kind="prepare_reply", s=s, q=q, rows=int(n.sum()), voted=int((pr["pr_vbal"] > 0).sum())))
                late.append((s, reps[s].prepare_replies(peer=u8(q), pr_n=n, pr_trig=pr["pr_trig"], pr_endp=pr["pr_endp"],
                                                        pr_ballot=pr["pr_ballot"], pr_vbal=pr["pr_vbal"], pr_vval=pr["pr_vval"],
                                                        pr_vmask=pr["pr_vmask"])))
    # 5. reconstruction reads
    for s in range(R):
        if not bl[s]["rc_n"].any():
            continue
        for q in range(R):
            if q == s:
                continue
            fl = ((bl[s]["rc_n"] > 0) & ~_lost(drop, "recon", s, q, G)).astype(np.uint8)
            rr = reps[q].reconstruct(flags=fl, rc_n=bl[s]["rc_n"], rc_slot=bl[s]["rc_slot"])
            fl2 = ((rr["rr_n"] > 0) & ~_lost(drop, "recon_reply", q, s, G)).astype(np.uint8)
            if fl2.any():
                log.append(dict(kind="recon_reply", s=s, q=q, rows=int(rr["rr_n"][fl2.astype(bool)].sum())))
                reps[s].reconstruct_reply(flags=fl2, **rr)
        # "inject a heartbeat after every chunk to keep peers happy" (leadership.rs:173-183): a plain broadcast with the
        # NEW ballot (bal_max_seen = bal_prep_sent by now) and the bars of the step-up moment; the sender does not hear it
        _deliver_heartbeat(reps, s, (bl[s]["rc_n"] > 0).astype(np.uint8), bl[s]["p_ballot"], bl[s]["hb_commit"], bl[s]["hb_exec"],
                           bl[s]["hb_snap"], drop)
    # 6. Accept phase: the client batches, then what the Prepare quorum released
    for s in range(R):
        _deliver_accepts(reps, s, acc[s], drop, log)
    for s, a in late:
        if a["a_n"].any():
            log.append(dict(kind="re_accept", s=s, n=int(a["a_n"].sum()), empty=int(((a["a_val"] == 0) & (np.arange(len(a["a_val"]))[:, None] < a["a_n"][None, :])).sum())))
        _deliver_accepts(reps, s, a, drop, log)
    # 7. periodic heartbeats of the replicas that lead
    if heartbeat:
        leaders = [reps[r].is_leader() if hasattr(reps[r], "is_leader") else (reps[r].dump()["leader"] == r).astype(np.uint8)
                   for r in range(R)]
        for s in range(R):
            fl = leaders[s]
            if not fl.any():
                continue
            hb = reps[s].bcast_heartbeat(fl)
            _deliver_heartbeat(reps, s, fl, hb["ballot"], hb["commit_bar"], hb["exec_bar"], hb["snap_bar"], drop)
    return log


class SteadyLoop:
    """The steady state of a co-located RSPaxos cluster (one prepared leader, no timeouts) with every message a DEVICE tensor
    between the handlers -- the closed loop `tick` above drives through numpy, here without a host round trip, so that a tick
    can be timed and captured into a HIP graph: per tick the leader's `handle_req_batch` on the tick's batches, the followers'
    `handle_msg_accept` with the mask of the ONE shard they were sent, the leader's `handle_msg_accept_reply` tally with the
    shard-availability gate behind it (durability.rs:140-160), and -- on a heartbeat tick -- the leader's Heartbeat, the
    followers' handling of it and their Heartbeats back.  Same handler calls, same order as `tick` for such a tick
    (tests/test_zz_rsp_steady_gpu.py holds it against `tick` on five oracles).  The shard fan-out (follower q gets shard q of
    every new codeword, rspaxos/request.rs:127-142) is `encode`: from_data + RS encode + the R shard stores in ONE pass over the
    serialized batches (smr_rs_from_data_encode_fanout) -- the co-located stand-in for the Accepts' payload.
    Every output and scratch tensor is made once and written again every tick."""

    def __init__(self, reps, leader=0, one_launch=False):
        """one_launch: the tick's handlers as ONE kernel launch (`smr_rsp_cluster_steady_tick`: a block = the R replicas of 64 groups,
        messages through LDS) instead of one call per handler; `reps` must then be RSPaxosReplicaGroup objects"""
        self.reps, self.R, self.G, self.s = list(reps), len(reps), reps[0].G, int(leader)
        self.stores = None                       # [R, G, shard_len]: store q = what replica q holds of the tick's codewords (shard q)
        self._stores = {}                        # (the buffers behind it, by `encode`'s slot)
        self._b = None
        self._cl = None
        if one_launch:
            import ctypes as C
            from . import _lib
            self._L = _lib.load()
            arr = (C.c_void_p * self.R)(*[r._h for r in self.reps])
            h = C.c_void_p()
            _lib.check(self._L.smr_rsp_cluster_create(arr, self.R, C.byref(h)))
            self._cl = h
            self._held = None

    def close(self):
        if getattr(self, "_cl", None):
            self._L.smr_rsp_cluster_destroy(self._cl)
            self._cl = None

    def __del__(self):
        self.close()

    _KINDS = (("accept", True), ("accept_reply", False), ("hb", True), ("hb", False))   # (kind, leader -> q?) in lost_dev's order

    def _tick_one_launch(self, val, lost, heartbeat, stream=None):
        import ctypes as C
        import torch
        from . import _lib
        R, s, dev = self.R, self.s, val.device
        if self._b is None:
            self._b = dict(committed=torch.zeros(self.G, dtype=torch.uint8, device=dev))
        lp, keep = None, []
        if lost:
            ptrs = []
            for kind, out in self._KINDS:
                for q in range(R):
                    m = None if q == s else lost.get((kind, s, q) if out else (kind, q, s))
                    if m is not None:
                        m = (m if m.dtype == torch.uint8 else m.to(torch.uint8)).contiguous()
                        keep.append(m)
                    ptrs.append(None if m is None else m.data_ptr())
            lp = (C.c_void_p * (4 * R))(*ptrs)
        _lib.check(self._L.smr_rsp_cluster_steady_tick(self._cl, s, val.data_ptr(), lp, int(bool(heartbeat)), self._b["committed"].data_ptr(),
                                                       _lib.stream_ptr(stream)))
        self._held = (keep, val)                  # alive until the next tick has replaced them (the call only enqueues work)
        return self._b["committed"]

    def _bufs(self, dev):
        import torch
        if self._b is None:
            R, G, W = self.R, self.G, self.reps[0].W
            z = lambda dt, *sh: torch.zeros(sh or (G,), dtype=dt, device=dev)
            hb = lambda: dict(ballot=z(torch.int64), commit_bar=z(torch.int32), exec_bar=z(torch.int32), snap_bar=z(torch.int32))
            self._b = dict(acc=dict(a_n=z(torch.int32), a_slot=z(torch.int32, W, G), a_val=z(torch.int32, W, G), a_ballot=z(torch.int64)),
                           peer=[torch.full((G,), q, dtype=torch.uint8, device=dev) for q in range(R)],
                           mask=[torch.full((G,), 1 << q, dtype=torch.uint8, device=dev) for q in range(R)],
                           ballot=z(torch.int64, R, G), r_slot=z(torch.int32, R, G), committed=dict(committed=z(torch.uint8)),
                           ones=torch.ones(G, dtype=torch.uint8, device=dev), hb=hb(),
                           rp=[dict(reply=z(torch.uint8), **hb()) for _ in range(R)], back=dict(reply=z(torch.uint8), **hb()))
        return self._b

    def encode(self, data, out=None, stream=None, slot=0):
        """from_data + compute_parity of the tick's batches (`data`: uint8 [G, L]) and every replica's shard store, one pass.
        `slot`: which of the replicas' shard-store buffers receives the fan-out -- a host that encodes batch k + 1 while tick k
        is still being processed (another stream) alternates two, as a leader with a batch in flight does."""
        import torch
        from .rscoding import RSCodewordBatch, rs_shard_len
        sl = rs_shard_len(int(data.shape[1]), self.R // 2 + 1)
        if self._stores.get(slot) is None or self._stores[slot].shape[2] != sl:
            self._stores[slot] = torch.empty((self.R, self.G, sl), dtype=torch.uint8, device=data.device)
        self.stores = self._stores[slot]
        d = self.R // 2 + 1                       # rspaxos/mod.rs:599-609: RS(majority, R - majority)
        return RSCodewordBatch.from_data_and_encode(data, d, self.R - d, stream=stream, out=out, fan_out=self.stores)

    def encode_stores(self, data, stream=None, slot=0):
        """`encode` with every shard written ONCE (round 4): from_data + RS encode straight into the R replicas' shard
        stores (`smr_rs_from_data_encode_stores`) -- store q = shard q of the tick's codewords = what replica q holds; the leader's
        codeword is those stores (the returned batch is a view of them: `RSCodewordBatch.from_data_and_encode_stores`).  `encode`
        wrote a contiguous codeword AND the stores: 10 shard_len per batch instead of 5."""
        import torch
        from .rscoding import RSCodewordBatch, rs_shard_len
        d = self.R // 2 + 1                       # rspaxos/mod.rs:599-609: RS(majority, R - majority)
        sl = rs_shard_len(int(data.shape[1]), d)
        if self._stores.get(slot) is None or self._stores[slot].shape[2] != sl:
            self._stores[slot] = torch.empty((self.R, self.G, sl), dtype=torch.uint8, device=data.device)
        self.stores = self._stores[slot]
        return RSCodewordBatch.from_data_and_encode_stores(data, d, self.R - d, stores=self.stores, stream=stream)

    def tick(self, val, lost=None, heartbeat=False):
        """val: int32 [G] batch tokens (NULL = none); lost: optional dict (kind, from, to) -> bool [G] like `tick`'s drop.
        Returns the leader's `committed` flags [G] (valid until the next tick)."""
        import torch
        if self._cl is not None:
            return self._tick_one_launch(val, lost, heartbeat)
        R, s, dev = self.R, self.s, val.device
        reps, b = self.reps, self._bufs(dev)
        gone = lambda kind, a, c: None if lost is None else lost.get((kind, a, c))
        acc = reps[s].req_batch(val, out=b["acc"])
        live = (acc["a_n"] > 0).to(torch.uint8)
        slot, tok = acc["a_slot"][0], acc["a_val"][0]
        for q in range(R):
            if q == s:
                continue
            fl = live if gone("accept", s, q) is None else (live & ~gone("accept", s, q).to(torch.uint8))
            reps[q].accept(flags=fl, peer=b["peer"][s], slot=slot, ballot=acc["a_ballot"], val=tok, mask=b["mask"][q],
                           out=dict(r_ballot=b["ballot"][q], r_slot=b["r_slot"][q]))
        flags = (b["ballot"] != 0).to(torch.uint8)                  # (row s stays zero: nobody writes it)
        if lost is not None:
            for q in range(R):
                if q != s and gone("accept_reply", q, s) is not None:
                    flags[q] &= ~gone("accept_reply", q, s).to(torch.uint8)
        res = reps[s].accept_replies(slot=slot, ballot=b["ballot"], flags=flags, out=b["committed"])
        committed = res["committed"] & live
        if heartbeat:
            hb = reps[s].bcast_heartbeat(b["ones"], out=b["hb"])
            for q in range(R):
                if q == s:
                    continue
                fl = b["ones"] if gone("hb", s, q) is None else (b["ones"] & ~gone("hb", s, q).to(torch.uint8))
                rp = reps[q].heartbeat(flags=fl, peer=b["peer"][s], ballot=hb["ballot"], commit_bar=hb["commit_bar"], exec_bar=hb["exec_bar"],
                                       snap_bar=hb["snap_bar"], out=b["rp"][q])
                back = rp["reply"]
                if gone("hb", q, s) is not None:
                    back = back & ~gone("hb", q, s).to(torch.uint8)
                reps[s].heartbeat(flags=back, peer=b["peer"][q], ballot=rp["ballot"], commit_bar=rp["commit_bar"], exec_bar=rp["exec_bar"],
                                  snap_bar=rp["snap_bar"], out=b["back"])
        return committed
