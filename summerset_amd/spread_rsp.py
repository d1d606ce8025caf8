"""Layout L2 (BASELINE config 4's "1 -> 8 GPU shard over xGMI") for the RSPaxos replica engine,
device-resident: the steady state of `rsp_cluster.SteadyLoop` with the replicas of every group on DIFFERENT ranks.

The job's groups are block-partitioned over the ranks (shard.group_range); replica r of block b lives on rank
(b + r) mod world (as in spread_mp / spread_ep), so block b is led from rank b and each of its Accepts -- header AND
the follower's shard of the batch's codeword, the one exchange of the path with real bytes (rspaxos/request.rs:127-142:
one shard per peer; messages.rs:468-595) -- crosses to another rank, and each AcceptReply crosses back.  The stand-in for
`server/transport.rs:208-275` (`send_msg`).

Per tick, per rank:
  A  for the block it leads: from_data + RS encode of the tick's batches with every follower's shard written STRAIGHT into
     that follower's slice of the send buffer (smr_rs_from_data_encode_scatter: nothing is read again to fill send buffers),
     `handle_req_batch`, the Accept header (flags, slot, ballot, token) copied in front of every shard
  X1 ONE all_to_all_single (device tensors, static split sizes)
  B  for every (block, follower) it holds: `handle_msg_accept` on views of the receive buffer with the mask of the one shard
     it was sent; the reply ballots are written by the kernel straight into the backward send buffer
  X2 ONE all_to_all_single
  C  the leader's `handle_msg_accept_reply` tally (majority + f, the shard-availability gate behind it)
  on a heartbeat tick two more exchanges: the leader's Heartbeat out, the followers' Heartbeats back.
Results are bit for bit the co-located loop's (tests/test_spread_rsp_gloo.py: world_size 2 over gloo with the emulator
build of the engine, against `rsp_cluster.SteadyLoop` in one process; tests/test_spread_rsp.py: every rank in one process)."""
import numpy as np

from . import shard
from .rscoding import RSCodewordBatch, rs_shard_len
from .rspaxos import RSPaxosReplicaGroup

NULL = 0xFFFFFFFF
_A16 = lambda n: (n + 15) // 16 * 16   # noqa: E731


def home(block, replica, world):
    return (block + replica) % world


class SpreadRSPaxos:
    """one rank's part of the job.  `exchange(kind, send, recv, in_split, out_split)` (optional) replaces the collective --
    tests with every rank in one process."""

    LEADER = 0

    def __init__(self, total_groups, population, window, rank, world, device, data_len, fault_tolerance=1, exchange=None, payload=False):
        """payload: every replica that lives here gets an `RSPaxosPayloadStore` (and every follower a staging store): the leader's
        batches go into its store (`put`), an Accept's shard is `extract`ed from that store into the message, what arrives is
        `ingest`ed into the receiver's staging store and `follow` runs behind every handler -- the bytes of layout L2 in the
        product's store instead of in this module's per-tick buffers (off: the fused encode writes the shards straight into the
        send buffers and nothing keeps them)"""
        import torch
        self.torch = torch
        self.R, self.W, self.rank, self.world, self.device, self.L = int(population), int(window), int(rank), int(world), device, int(data_len)
        self.d = self.R // 2 + 1
        self.sl = rs_shard_len(self.L, self.d)
        self.exchange = exchange
        self.comm = None                                        # set by bind_comm(): the exchanges run inside the library (RCCL)
        self.bytes_sent = 0
        self.n_groups = {b: shard.group_range(total_groups, world, b) for b in range(world)}
        self.reps = {}                                          # (block, replica) -> RSPaxosReplicaGroup, the ones that live here
        for b in range(world):
            lo, hi = self.n_groups[b]
            for r in range(self.R):
                if home(b, r, world) == rank and hi > lo:
                    e = RSPaxosReplicaGroup(hi - lo, self.R, me=r, window=self.W, fault_tolerance=fault_tolerance)
                    e.preset_leader(self.LEADER)
                    self.reps[(b, r)] = e
        self.lead = [b for b in range(world) if (b, self.LEADER) in self.reps]      # (at most one: block `rank`)
        self.stores, self.staging, self._pmsg = {}, {}, {}
        if payload:
            from .rsp_payload import RSPaxosPayloadStore
            for (b, r), e in self.reps.items():
                self.stores[(b, r)] = RSPaxosPayloadStore(e.G, self.R, self.W, max_data_len=self.L)
                if r != self.LEADER:
                    self.staging[(b, r)] = RSPaxosPayloadStore(e.G, self.R, self.W, max_data_len=self.L)
        self._plans = {k: self._plan(k) for k in ("accept", "accept_reply", "hb", "hb_back")}
        self._bufs = {}
        self.cw = {}

    # ---- static plans: who sends what to whom, the same list on every rank --------------------------------------------
    def _msg_bytes(self, kind, b):
        lo, hi = self.n_groups[b]
        G = hi - lo
        if kind == "accept":                                    # ballot u64 | slot u32 | token u32 | flags u8 | pad | the shard
            return _A16(G * 17) + _A16(G * self.sl)
        if kind == "accept_reply":
            return _A16(G * 8)                                  # r_ballot (0 = no reply)
        if kind == "hb":
            return _A16(G * 20)                                 # ballot u64 | commit u32 | exec u32 | snap u32
        return _A16(G * 21)                                     # hb_back: the same + reply u8

    def _plan(self, kind):
        torch = self.torch
        msgs = []                                               # (src, dst, block, follower)
        for b in range(self.world):
            lo, hi = self.n_groups[b]
            if hi <= lo:
                continue
            hl = home(b, self.LEADER, self.world)
            for q in range(self.R):
                if q == self.LEADER:
                    continue
                hq = home(b, q, self.world)
                msgs.append((hl, hq, b, q) if kind in ("accept", "hb") else (hq, hl, b, q))
        send = sorted([m for m in msgs if m[0] == self.rank], key=lambda m: m[1])       # stable: canonical order per destination
        recv = sorted([m for m in msgs if m[1] == self.rank], key=lambda m: m[0])
        in_split, out_split = [0] * self.world, [0] * self.world
        soff, roff, o = {}, {}, 0
        for m in send:
            soff[(m[2], m[3])] = o
            n = self._msg_bytes(kind, m[2])
            o += n
            in_split[m[1]] += n
        n_send, o = o, 0
        for m in recv:
            roff[(m[2], m[3])] = o
            n = self._msg_bytes(kind, m[2])
            o += n
            out_split[m[0]] += n
        return dict(soff=soff, roff=roff, in_split=in_split, out_split=out_split,
                    sbuf=torch.zeros(max(n_send, 16), dtype=torch.uint8, device=self.device),
                    rbuf=torch.zeros(max(o, 16), dtype=torch.uint8, device=self.device))

    def _collective(self, kind):
        import torch.distributed as dist
        p = self._plans[kind]
        self.bytes_sent += sum(p["in_split"])
        if self.exchange is not None:
            self.exchange(kind, self)
        elif self.world > 1 and self.comm is not None:          # the library's exchange (smr_comm_exchange: RCCL send / recv pairs)
            self.comm.exchange(p["sbuf"], p["in_split"], p["rbuf"], p["out_split"])
        elif self.world > 1:
            dist.all_to_all_single(p["rbuf"][:sum(p["out_split"])], p["sbuf"][:sum(p["in_split"])], output_split_sizes=p["out_split"],
                                   input_split_sizes=p["in_split"])
        else:
            p["rbuf"][:sum(p["out_split"])].copy_(p["sbuf"][:sum(p["in_split"])])

    # ---- the tick inside the library (round 6: smr_rsp_spread_*, csrc/rsp_spread.hip) --------------------------------------------
    def use_library_tick(self):
        """from now on the tick's phases are the library's segments (`smr_rsp_spread_segment`: encode + scatter, handlers, header
        copies, loss masks -- no torch op in a tick) and, with `bind_comm`, `tick` is ONE C call (`smr_rsp_spread_tick`, exchanges
        inside).  The plans' send / receive buffers become views of the library's (same layout: what the tests read out of them
        stays where it was).  Not with payload stores (those stay this module's)."""
        import ctypes as C
        from . import _lib
        from .spread_ep import _tensor_over
        if self.stores:
            raise ValueError("the library tick carries the shards in the exchange's buffers: payload=False")
        self._L = _lib.load()
        order = sorted(self.reps)
        n = len(order)
        arr = (C.c_void_p * max(n, 1))(*[self.reps[k]._h for k in order])
        blocks = (C.c_uint32 * max(n, 1))(*[k[0] for k in order])
        ids = (C.c_uint8 * max(n, 1))(*[k[1] for k in order])
        groups = (C.c_uint32 * self.world)(*[self.n_groups[b][1] - self.n_groups[b][0] for b in range(self.world)])
        h = C.c_void_p()
        _lib.check(self._L.smr_rsp_spread_create(arr, blocks, ids, n, groups, self.world, self.rank, self.R, self.W, self.L, C.byref(h)))
        self._lib_h = h
        for k, kind in enumerate(("accept", "accept_reply", "hb", "hb_back")):
            sp, rp = C.c_void_p(), C.c_void_p()
            sb, rb = (C.c_uint64 * self.world)(), (C.c_uint64 * self.world)()
            _lib.check(self._L.smr_rsp_spread_buffers(h, k, C.byref(sp), sb, C.byref(rp), rb))
            p = self._plans[kind]
            assert [int(x) for x in sb] == p["in_split"] and [int(x) for x in rb] == p["out_split"], kind
            p["sbuf"] = _tensor_over(self.torch, sp.value, max(sum(p["in_split"]), 16), self.device)
            p["rbuf"] = _tensor_over(self.torch, rp.value, max(sum(p["out_split"]), 16), self.device)
        if self.comm is not None:
            _lib.check(self._L.smr_rsp_spread_bind_comm(h, self.comm._h))
        self._lib_committed = {b: self.torch.zeros(self.n_groups[b][1] - self.n_groups[b][0], dtype=self.torch.uint8, device=self.device) for b in self.lead}
        return self

    def close_library_tick(self):
        if getattr(self, "_lib_h", None):
            self._L.smr_rsp_spread_destroy(self._lib_h)
            self._lib_h = None

    def _lib_args(self, data, val, lost, heartbeat):
        """(data ptr, val ptr, lost table, heartbeat, committed ptr) of this tick; kept until the next tick has replaced them"""
        import ctypes as C
        torch, R = self.torch, self.R
        b = self.lead[0] if self.lead else None
        d = data[b].contiguous() if b is not None else None
        v = val[b].contiguous() if b is not None else None
        tab, masks = None, None
        if lost:
            kinds = (("accept", lambda q: (0, q)), ("accept_reply", lambda q: (q, 0)), ("hb", lambda q: (0, q)), ("hb", lambda q: (q, 0)))
            masks, ptrs = [], []
            for bb in range(self.world):
                for kind, ft in kinds:
                    for q in range(R):
                        g = lost.get(bb, {}).get((kind,) + ft(q)) if q else None
                        if g is not None:
                            g = (g if g.dtype == torch.uint8 else g.to(torch.uint8)).contiguous()
                            masks.append(g)
                        ptrs.append(None if g is None else g.data_ptr())
            tab = (C.c_void_p * len(ptrs))(*ptrs)
        self._lib_held = (d, v, masks, tab)
        self._lib_call = (None if d is None else d.data_ptr(), None if v is None else v.data_ptr(), tab, 1 if heartbeat else 0,
                          None if b is None else self._lib_committed[b].data_ptr())

    def _lib_segment(self, seg):
        from . import _lib
        d, v, tab, hb, c = self._lib_call
        _lib.check(self._L.smr_rsp_spread_segment(self._lib_h, seg, d, v, tab, hb, c, _lib.stream_ptr(None)))

    def bind_comm(self, comm):
        """every exchange of the tick (Accepts + shards out, AcceptReplies back, the two heartbeat legs) through the library:
        `comm` (summerset_amd.comm.Comm) -> `smr_comm_exchange` on the plans' own buffers with their static split sizes.
        None: back to torch.distributed.all_to_all_single (gloo jobs)."""
        if comm is not None and (comm.world != self.world or comm.rank != self.rank):
            raise ValueError("the communicator is rank %d of %d, the job's rank is %d of %d" % (comm.rank, comm.world, self.rank, self.world))
        self.comm = comm
        if getattr(self, "_lib_h", None):
            from . import _lib
            _lib.check(self._L.smr_rsp_spread_bind_comm(self._lib_h, comm._h if comm is not None else None))

    # ---- typed views of a message inside a buffer -------------------------------------------------------------------------
    def _fields(self, buf, off, G, spec):
        torch = self.torch
        out, o = {}, off
        for name, dt, width in spec:
            n = G * width
            out[name] = buf[o:o + n].view(dt)
            o += n
        return out, o

    def _accept_msg(self, buf, off, G):
        torch = self.torch
        f, o = self._fields(buf, off, G, (("ballot", torch.int64, 8), ("slot", torch.int32, 4), ("val", torch.int32, 4), ("flags", torch.uint8, 1)))
        base = off + _A16(G * 17)
        f["shard"] = buf[base:base + G * self.sl].view(G, self.sl)
        f["header"] = buf[off:off + G * 17]
        return f

    def _hb_msg(self, buf, off, G, back):
        torch = self.torch
        spec = (("ballot", torch.int64, 8), ("commit_bar", torch.int32, 4), ("exec_bar", torch.int32, 4), ("snap_bar", torch.int32, 4))
        if back:
            spec = spec + (("reply", torch.uint8, 1),)
        return self._fields(buf, off, G, spec)[0]

    def _b(self, key, make):
        if key not in self._bufs:
            self._bufs[key] = make()
        return self._bufs[key]

    # ---- the tick's phases ----------------------------------------------------------------------------------------------------
    def phase_a(self, data, val, lost=None, heartbeat=False):
        """leaders: data[b] uint8 [G_b, L] = the tick's serialized batches, val[b] int32 [G_b] their tokens (NULL = none)"""
        if getattr(self, "_lib_h", None):
            self._lib_args(data, val, lost, heartbeat)
            return self._lib_segment(0)
        torch = self.torch
        R, s, p = self.R, self.LEADER, self._plans["accept"]
        for b in self.lead:
            lo, hi = self.n_groups[b]
            G = hi - lo
            eng = self.reps[(b, s)]
            msgs = {q: self._accept_msg(p["sbuf"], p["soff"][(b, q)], G) for q in range(R) if q != s}
            if not self.stores:
                if b not in self.cw:
                    self.cw[b] = RSCodewordBatch(G, self.L, self.d, R - self.d, device=self.device, zero=False)
                RSCodewordBatch.from_data_and_encode(data[b], self.d, R - self.d, out=self.cw[b],
                                                     shard_dst=[None if q == s else msgs[q]["shard"] for q in range(R)])
            acc = eng.req_batch(val[b], out=self._b(("acc", b), lambda: dict(
                a_n=torch.zeros(G, dtype=torch.int32, device=self.device), a_slot=torch.zeros((self.W, G), dtype=torch.int32, device=self.device),
                a_val=torch.zeros((self.W, G), dtype=torch.int32, device=self.device), a_ballot=torch.zeros(G, dtype=torch.int64, device=self.device))))
            live = (acc["a_n"] > 0).to(torch.uint8)
            self._bufs[("live", b)] = live
            if self.stores:                                     # the codeword lives in the leader's store; every Accept's shard is taken out of it
                from .rsp_payload import REQS
                st = self.stores[(b, s)]
                st.put(acc, data[b])
                st.follow(eng)
                for q in range(R):
                    if q != s:
                        self._pmsg[b] = st.extract(acc["a_slot"][0], self._b(("mask", b, q), lambda: torch.full((G,), 1 << q, dtype=torch.uint8, device=self.device)),
                                                   REQS, live, out=self._pmsg.get(b))
                        msgs[q]["shard"].copy_(self._pmsg[b]["buf"][q, :, :self.sl])
            first = None
            for q in range(R):
                if q == s:
                    continue
                m = msgs[q]
                if first is None:                               # the header once, then byte copies of it
                    m["ballot"].copy_(acc["a_ballot"]); m["slot"].copy_(acc["a_slot"][0]); m["val"].copy_(acc["a_val"][0]); m["flags"].copy_(live)
                    first = m
                else:
                    m["header"].copy_(first["header"])
                # every follower's flags are written from `live`, never inherited through the header copy: the first
                # follower's header carries THAT follower's losses, and a partial `lost` dict used to leak them
                g = None if lost is None else lost.get(b, {}).get(("accept", s, q))
                m["flags"].copy_(live if g is None else (live & ~g.to(torch.uint8)))

    def phase_b(self, lost=None):
        """followers: handle_msg_accept on what arrived; the reply ballots land in the backward send buffer"""
        if getattr(self, "_lib_h", None):
            return self._lib_segment(1)
        torch = self.torch
        s, pa, pr = self.LEADER, self._plans["accept"], self._plans["accept_reply"]
        for (b, q), eng in self.reps.items():
            if q == s:
                continue
            lo, hi = self.n_groups[b]
            G = hi - lo
            m = self._accept_msg(pa["rbuf"], pa["roff"][(b, q)], G)
            o = pr["soff"][(b, q)]
            r_ballot = pr["sbuf"][o:o + G * 8].view(torch.int64)
            mask_q = self._b(("mask", b, q), lambda: torch.full((G,), 1 << q, dtype=torch.uint8, device=self.device))
            if self.stores:                                     # what arrived -> the staging store (the message's header + its one shard)
                from .rsp_payload import REQS
                pm = self._b(("pmsg_in", b, q), lambda: dict(buf=torch.zeros((self.R, G, self.stores[(b, q)].group_stride), dtype=torch.uint8, device=self.device),
                                                             dlen=torch.full((G,), self.L, dtype=torch.int32, device=self.device)))
                pm["buf"][q, :, :self.sl].copy_(m["shard"])
                self.staging[(b, q)].ingest(dict(buf=pm["buf"], tok=m["val"], mask=mask_q, dlen=pm["dlen"]), m["slot"], REQS, m["flags"])
            eng.accept(flags=m["flags"], peer=self._b(("peer", b, s), lambda: torch.full((G,), s, dtype=torch.uint8, device=self.device)), slot=m["slot"],
                       ballot=m["ballot"], val=m["val"], mask=mask_q,
                       out=dict(r_ballot=r_ballot, r_slot=self._b(("r_slot", b, q), lambda: torch.zeros(G, dtype=torch.int32, device=self.device))))
            if self.stores:
                self.stores[(b, q)].follow(eng, [(self.staging[(b, q)], REQS)])
            g = None if lost is None else lost.get(b, {}).get(("accept_reply", q, s))
            if g is not None:
                r_ballot.masked_fill_(g, 0)                     # a lost reply

    def phase_c(self):
        """leaders: the AcceptReply tally; returns {block: committed flags [G_b]}"""
        if getattr(self, "_lib_h", None):
            self._lib_segment(2)
            return {b: self._lib_committed[b] for b in self.lead}
        torch = self.torch
        R, s, pr = self.R, self.LEADER, self._plans["accept_reply"]
        out = {}
        for b in self.lead:
            lo, hi = self.n_groups[b]
            G = hi - lo
            ballot = self._b(("ballot", b), lambda: torch.zeros((R, G), dtype=torch.int64, device=self.device))
            for q in range(R):
                if q != s:
                    o = pr["roff"][(b, q)]
                    ballot[q].copy_(pr["rbuf"][o:o + G * 8].view(torch.int64))
            flags = (ballot != 0).to(torch.uint8)
            acc = self._bufs[("acc", b)]
            res = self.reps[(b, s)].accept_replies(slot=acc["a_slot"][0], ballot=ballot, flags=flags,
                                                   out=self._b(("committed", b), lambda: dict(committed=torch.zeros(G, dtype=torch.uint8, device=self.device))))
            out[b] = res["committed"] & self._bufs[("live", b)]
            if self.stores:
                self.stores[(b, s)].follow(self.reps[(b, s)])
        return out

    def phase_hb_out(self, lost=None):
        if getattr(self, "_lib_h", None):
            return self._lib_segment(3)
        torch = self.torch
        R, s, p = self.R, self.LEADER, self._plans["hb"]
        for b in self.lead:
            lo, hi = self.n_groups[b]
            G = hi - lo
            ones = self._b(("ones", b), lambda: torch.ones(G, dtype=torch.uint8, device=self.device))
            first = None
            for q in range(R):
                if q == s:
                    continue
                m = self._hb_msg(p["sbuf"], p["soff"][(b, q)], G, False)
                if first is None:
                    self.reps[(b, s)].bcast_heartbeat(ones, out=m)
                    first = p["soff"][(b, q)]
                else:
                    o = p["soff"][(b, q)]
                    p["sbuf"][o:o + G * 20].copy_(p["sbuf"][first:first + G * 20])

    def phase_hb_in(self, lost=None):
        if getattr(self, "_lib_h", None):
            return self._lib_segment(4)
        torch = self.torch
        s, p, pb = self.LEADER, self._plans["hb"], self._plans["hb_back"]
        for (b, q), eng in self.reps.items():
            if q == s:
                continue
            lo, hi = self.n_groups[b]
            G = hi - lo
            m = self._hb_msg(p["rbuf"], p["roff"][(b, q)], G, False)
            back = self._hb_msg(pb["sbuf"], pb["soff"][(b, q)], G, True)
            fl = self._b(("ones", b), lambda: torch.ones(G, dtype=torch.uint8, device=self.device))
            g = None if lost is None else lost.get(b, {}).get(("hb", s, q))
            if g is not None:
                fl = fl & ~g.to(torch.uint8)
            eng.heartbeat(flags=fl, peer=self._b(("peer", b, s), lambda: torch.full((G,), s, dtype=torch.uint8, device=self.device)), ballot=m["ballot"],
                          commit_bar=m["commit_bar"], exec_bar=m["exec_bar"], snap_bar=m["snap_bar"], out=back)
            if self.stores:
                self.stores[(b, q)].follow(eng)
            g = None if lost is None else lost.get(b, {}).get(("hb", q, s))
            if g is not None:
                back["reply"].masked_fill_(g, 0)

    def phase_hb_back(self):
        if getattr(self, "_lib_h", None):
            return self._lib_segment(5)
        torch = self.torch
        R, s, pb = self.R, self.LEADER, self._plans["hb_back"]
        for b in self.lead:
            lo, hi = self.n_groups[b]
            G = hi - lo
            for q in range(R):
                if q == s:
                    continue
                m = self._hb_msg(pb["rbuf"], pb["roff"][(b, q)], G, True)
                self.reps[(b, s)].heartbeat(flags=m["reply"], peer=self._b(("peer", b, q), lambda: torch.full((G,), q, dtype=torch.uint8, device=self.device)),
                                            ballot=m["ballot"], commit_bar=m["commit_bar"], exec_bar=m["exec_bar"], snap_bar=m["snap_bar"],
                                            out=self._b(("hb_scratch", b), lambda: dict(reply=torch.zeros(G, dtype=torch.uint8, device=self.device),
                                                                                       ballot=torch.zeros(G, dtype=torch.int64, device=self.device),
                                                                                       commit_bar=torch.zeros(G, dtype=torch.int32, device=self.device),
                                                                                       exec_bar=torch.zeros(G, dtype=torch.int32, device=self.device),
                                                                                       snap_bar=torch.zeros(G, dtype=torch.int32, device=self.device))))
            if self.stores:
                self.stores[(b, s)].follow(self.reps[(b, s)])

    def tick(self, data, val, lost=None, heartbeat=False):
        """data / val: per led block (see phase_a); lost[b][(kind, from, to)] = bool [G_b] (optional).  Returns {block: committed}"""
        if getattr(self, "_lib_h", None) and self.exchange is None and (self.comm is not None or self.world == 1):
            from . import _lib                                   # ONE call: segments and exchanges inside the library
            self._lib_args(data, val, lost, heartbeat)
            d, v, tab, hb, c = self._lib_call
            _lib.check(self._L.smr_rsp_spread_tick(self._lib_h, d, v, tab, hb, c, _lib.stream_ptr(None)))
            self.bytes_sent += sum(sum(self._plans[k]["in_split"]) for k in (("accept", "accept_reply", "hb", "hb_back") if heartbeat else ("accept", "accept_reply")))
            return {b: self._lib_committed[b] for b in self.lead}
        self.phase_a(data, val, lost, heartbeat)
        self._collective("accept")
        self.phase_b(lost)
        self._collective("accept_reply")
        committed = self.phase_c()
        if heartbeat:
            self.phase_hb_out(lost)
            self._collective("hb")
            self.phase_hb_in(lost)
            self._collective("hb_back")
            self.phase_hb_back()
        return committed

    def commits(self):
        return sum(int(self.reps[(b, self.LEADER)].dump()["counters"][0]) for b in self.lead)


def _copy_between(ranks, kind):
    """the all-to-all of a job whose ranks all live in this process"""
    for s_, ps in enumerate(ranks):
        p = ps._plans[kind]
        so = 0
        for d, n in enumerate(p["in_split"]):
            q = ranks[d]._plans[kind]
            ro = sum(q["out_split"][:s_])
            assert q["out_split"][s_] == n
            q["rbuf"][ro:ro + n].copy_(p["sbuf"][so:so + n])
            so += n


class in_process:
    """every rank of the job inside one process (one device, or the emulator): same objects, plans and buffers, the collective a
    copy; every rank finishes a phase before any rank starts the next"""

    def __init__(self, total_groups, population, window, world, device, data_len, fault_tolerance=1, payload=False):
        self.ranks = [SpreadRSPaxos(total_groups, population, window, r, world, device, data_len, fault_tolerance, exchange=lambda k, me: None,
                                    payload=payload) for r in range(world)]

    def tick(self, data, val, lost=None, heartbeat=False):
        rs = self.ranks

        def coll(kind):
            for r in rs:
                r.bytes_sent += sum(r._plans[kind]["in_split"])
            _copy_between(rs, kind)
        for r in rs:
            r.phase_a(data, val, lost, heartbeat)
        coll("accept")
        for r in rs:
            r.phase_b(lost)
        coll("accept_reply")
        committed = {}
        for r in rs:
            committed.update(r.phase_c())
        if heartbeat:
            for r in rs:
                r.phase_hb_out(lost)
            coll("hb")
            for r in rs:
                r.phase_hb_in(lost)
            coll("hb_back")
            for r in rs:
                r.phase_hb_back()
        return committed
