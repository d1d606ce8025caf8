"""Layout L2 for the EPaxos cluster -- BASELINE config 5 as it is written: "EPaxos, 65 536 groups x 5
replicas, 8 x MI355X, dependency-graph + fast-quorum kernel with RCCL all-to-all".

The job's groups are block-partitioned over the ranks (shard.group_range); replica r of block b lives on rank
(b + r) mod world, one `EPaxosReplicaGroup` per (block, replica) on its home rank.  EPaxos has no leader: EVERY replica
proposes, so every rank sends and receives in every exchange.  One tick of the closed loop of `ep_cluster.tick`
(PreAccept fan-out, PreAcceptReplies, fast / slow decision, slow-path Accepts, AcceptReplies, CommitNotices) is five
exchanges, each ONE `all_to_all_single` on device tensors with static split sizes -- the stand-in for
`server/transport.rs:208-275` (`send_msg` / `bcast_msg`) under `epaxos/messages.rs`' handlers.  A message is the
handler's own output tensors, byte-viewed and concatenated into the send buffer by one `torch.cat` (fields widest
first, a message padded to 8 bytes, so the receiver's tensors are aligned VIEWS of the receive buffer: no unpack
pass); messages between two replicas of one rank are handed over as they are.  Nothing of the data path touches the
host: no `.cpu()`, no counts to agree on first (the Accept round always runs; where no instance took the slow path
its flags are zero and the handlers ignore it).

Two schedules:
  * `ordered=False` (default): all command leaders tally together -- 5 exchanges per tick.  An acceptor then handles the
    tick's Accepts before the tick's CommitNotices instead of leader by leader; instances of different rows do not read
    each other on these paths, so the protocol state is bit for bit `ep_cluster.tick`'s (tests/test_spread_ep*.py).
    Dependency-graph execution does look across rows when it runs: with `execute=True` this schedule is bit for bit
    `ep_cluster.tick(.., phase_major=True)` -- the co-located loop with the command leaders' steps phase by phase, the order
    `smr_ep_cluster_set_mode(c, 2)` runs and the oracle cluster is checked in (round 3; tests/test_zzy_spread_ep_gpu.py) --
    not the default leader-by-leader loop.  For that one:
  * `ordered=True` (the default where `execute=True`): the Accept / AcceptReply / CommitNotice exchanges once per command
    leader, leaders ascending -- 2 + 3 R exchanges per tick, the exact handler order of `ep_cluster.tick`, execution state
    included.
"""
from . import shard
from .epaxos import EPaxosReplicaGroup

NONE = -1                                       # Option::None in a DepSet (0xFFFFFFFF as int32)
# per-group bytes of a message's fields, widest first (R = population); s -> q kinds carry the instance, q -> s the answer
_KINDS = {
    "pre_accept": (("seq", 8, 1), ("deps", 4, "R"), ("col", 4, 1), ("flags", 1, 1), ("key", 1, 1)),
    "pa_reply": (("ballot", 8, 1), ("seq", 8, 1), ("deps", 4, "R"), ("flags", 1, 1)),
    "accept": (("seq", 8, 1), ("deps", 4, "R"), ("col", 4, 1), ("flags", 1, 1), ("key", 1, 1)),
    "acc_reply": (("ballot", 8, 1), ("flags", 1, 1)),
    "commit": (("seq", 8, 1), ("deps", 4, "R"), ("col", 4, 1), ("flags", 1, 1), ("key", 1, 1)),
}


def home(block, replica, world):
    """the rank replica `replica` of block `block` lives on"""
    return (block + replica) % world


class SpreadEPaxos:
    def __init__(self, total_groups, population, rank, world, device, window=32, n_keys=64, optimized_quorum=True,
                 execute=False, ordered=None):
        import torch
        self.torch, self.R, self.rank, self.world, self.device = torch, int(population), int(rank), int(world), device
        self.ordered = bool(execute) if ordered is None else bool(ordered)
        self.range = {b: shard.group_range(total_groups, world, b) for b in range(world)}
        self.reps = {}                                         # (block, replica) -> EPaxosReplicaGroup, the ones that live here
        for b in range(world):
            lo, hi = self.range[b]
            for r in range(self.R):
                if hi > lo and home(b, r, world) == rank:
                    self.reps[(b, r)] = EPaxosReplicaGroup(hi - lo, self.R, me=r, window=window, n_keys=n_keys,
                                                          optimized_quorum=optimized_quorum, execute=execute)
        self.bytes_sent = 0
        self.peers = None                                      # set by in_process(): every rank's object
        self.comm = None                                       # set by bind_comm(): the exchanges run inside the library (RCCL)
        self._plans = {}
        leaders = [[s] for s in range(self.R)] if self.ordered else [list(range(self.R))]
        self._leader_sets = leaders
        for kind in ("pre_accept", "pa_reply"):
            self._plans[(kind, 0)] = self._plan(kind, list(range(self.R)))
        for i, ls in enumerate(leaders):
            for kind in ("accept", "acc_reply", "commit"):
                self._plans[(kind, i)] = self._plan(kind, ls)
        z = lambda n, dt, v=0: {b: torch.full((n, hi - lo) if n else (hi - lo,), v, dtype=dt, device=device)
                                for b, (lo, hi) in self.range.items() if hi > lo}
        self._zero_f, self._zero_b, self._none_d = z(0, torch.uint8), z(0, torch.int64), z(self.R, torch.int32, NONE)
        self._const = {}

    # ---- static plan: every rank derives the same message list, in the same order ----------------------------------
    def _msg_bytes(self, kind, b):
        lo, hi = self.range[b]
        n = sum(w * (self.R if c == "R" else c) for _, w, c in _KINDS[kind]) * (hi - lo)
        return (n + 7) & ~7

    def _plan(self, kind, leaders):
        torch, W = self.torch, self.world
        to_acceptor = kind in ("pre_accept", "accept", "commit")
        msgs = []                                              # (src rank, dst rank, block, from replica, to replica)
        for b in range(W):
            lo, hi = self.range[b]
            if hi <= lo:
                continue
            for s in leaders:
                for q in range(self.R):
                    if q != s:
                        a, z = (s, q) if to_acceptor else (q, s)
                        msgs.append((home(b, a, W), home(b, z, W), b, a, z))
        send = sorted([m for m in msgs if m[0] == self.rank and m[1] != self.rank], key=lambda m: m[1])   # stable
        recv = sorted([m for m in msgs if m[1] == self.rank and m[0] != self.rank], key=lambda m: m[0])
        in_split, out_split, soff, roff = [0] * W, [0] * W, {}, {}
        off = 0
        for m in send:
            soff[m[2:]] = off
            off += self._msg_bytes(kind, m[2])
            in_split[m[1]] += self._msg_bytes(kind, m[2])
        n_send, off = off, 0
        for m in recv:
            roff[m[2:]] = off
            off += self._msg_bytes(kind, m[2])
            out_split[m[0]] += self._msg_bytes(kind, m[2])
        return dict(kind=kind, send=[m[2:] for m in send], soff=soff, roff=roff, in_split=in_split, out_split=out_split, n_send=n_send, n_recv=off,
                    sbuf=torch.zeros(max(n_send, 8), dtype=torch.uint8, device=self.device),
                    rbuf=torch.zeros(max(off, 8), dtype=torch.uint8, device=self.device),
                    pad={n: torch.zeros(n, dtype=torch.uint8, device=self.device) for n in range(1, 8)})

    # ---- one exchange -----------------------------------------------------------------------------------------------
    def _post(self, plan, out):
        """out[(b, a, z)] = the message's field dict.  Messages for another rank go into the send buffer (ONE cat),
        the ones for a replica of this rank into `local`."""
        torch, parts, local = self.torch, [], {}
        for key in plan["send"]:
            n = 0
            for name, _, _ in _KINDS[plan["kind"]]:
                p = out[key][name].contiguous().view(torch.uint8).reshape(-1)
                parts.append(p)
                n += p.numel()
            if n & 7:
                parts.append(plan["pad"][8 - (n & 7)])
        if parts:
            torch.cat(parts, out=plan["sbuf"][:plan["n_send"]])
        for key, m in out.items():
            if home(key[0], key[2], self.world) == self.rank:
                local[key] = m
        plan["local"] = local
        self.bytes_sent += plan["n_send"]

    def _collective(self, plan):
        import torch.distributed as dist
        if self.world > 1:                                     # (the buffers are never empty tensors; the collective sees exactly the planned bytes)
            if self.comm is not None:                          # the library's exchange (smr_comm_exchange: RCCL send / recv pairs)
                self.comm.exchange(plan["sbuf"], plan["in_split"], plan["rbuf"], plan["out_split"])
            else:
                dist.all_to_all_single(plan["rbuf"][:plan["n_recv"]], plan["sbuf"][:plan["n_send"]], output_split_sizes=plan["out_split"],
                                       input_split_sizes=plan["in_split"])

    # ---- the tick inside the library (round 6: smr_ep_spread_*, csrc/ep_spread.hip) ---------------------------------------------
    def use_library_tick(self):
        """from now on `tick` is the library's: the schedule, the message plan, the packing and -- with `bind_comm` -- the exchanges
        run inside `smr_ep_spread_tick` (one C call per tick); without a communicator (a gloo job, `in_process`) the segments are
        `smr_ep_spread_segment` calls and this class only moves the exchange's buffers.  Same messages, same order, same state."""
        import ctypes as C
        from . import _lib
        self._L = _lib.load()
        order = sorted(self.reps)
        self._lib_order = order
        n = len(order)
        arr = (C.c_void_p * max(n, 1))(*[self.reps[k]._h for k in order])
        blocks = (C.c_uint32 * max(n, 1))(*[k[0] for k in order])
        ids = (C.c_uint8 * max(n, 1))(*[k[1] for k in order])
        groups = (C.c_uint32 * self.world)(*[self.range[b][1] - self.range[b][0] for b in range(self.world)])
        h = C.c_void_p()
        _lib.check(self._L.smr_ep_spread_create(arr, blocks, ids, n, groups, self.world, self.rank, self.R, 1 if self.ordered else 0, C.byref(h)))
        self._lib_h = h
        assert self._L.smr_ep_spread_n_exchanges(h) == self.exchanges_per_tick()
        if self.comm is not None:
            _lib.check(self._L.smr_ep_spread_bind_comm(h, self.comm._h))
        # the exchanges' buffers as tensors (for a host that moves them itself)
        self._lib_bufs = []
        for k in range(self.exchanges_per_tick()):
            sp, rp = C.c_void_p(), C.c_void_p()
            sb, rb = (C.c_uint64 * self.world)(), (C.c_uint64 * self.world)()
            _lib.check(self._L.smr_ep_spread_buffers(h, k, C.byref(sp), sb, C.byref(rp), rb))
            self._lib_bufs.append(dict(send=sp.value, recv=rp.value, in_split=[int(x) for x in sb], out_split=[int(x) for x in rb]))
        self._lib_outs = None
        return self

    def _lib_args(self, keys, drop):
        import ctypes as C
        from . import _lib
        torch, order, R = self.torch, self._lib_order, self.R
        n = len(order)
        if self._lib_outs is None:
            self._lib_outs = {}
            for (b, s) in order:
                G = self.range[b][1] - self.range[b][0]
                e = lambda shape, dt: torch.empty(shape, dtype=dt, device=self.device)   # noqa: E731
                self._lib_outs[(b, s)] = dict(proposed=e(G, torch.uint8), col=e(G, torch.int32), seq0=e(G, torch.int64), deps0=e((R, G), torch.int32),
                                              decision=e(G, torch.uint8), committed=e(G, torch.uint8), seq=e(G, torch.int64), deps=e((R, G), torch.int32))
        outs = (_lib.EpClusterOut * max(n, 1))()
        for i, k in enumerate(order):
            for f, _ in _lib.EpClusterOut._fields_:
                setattr(outs[i], f, self._lib_outs[k][f].data_ptr())
        kp = (C.c_void_p * max(n, 1))(*[keys[k].data_ptr() for k in order])
        dp, masks = None, None
        if drop:
            masks = {k: (v if v.dtype == torch.uint8 else v.to(torch.uint8)).contiguous() for k, v in drop.items()}
            dp = (C.c_void_p * max(n * R, 1))(*[(masks[(b, s, q)].data_ptr() if (b, s, q) in masks else None) for (b, s) in order for q in range(R)])
        self._lib_held = (masks, dict(keys))               # alive until the next tick's have replaced them
        return kp, dp, outs

    def _lib_results(self):
        self.out = {k: dict(col=o["col"], proposed=o["proposed"], decision=o["decision"], committed=o["committed"], seq=o["seq"], deps=o["deps"])
                    for k, o in self._lib_outs.items()}
        import ctypes as C
        from . import _lib
        info = (C.c_uint64 * 2)()
        _lib.check(self._L.smr_ep_spread_info(self._lib_h, info))
        self.bytes_sent = int(info[1])
        return self.out

    def _lib_steps(self, keys, drop=None):
        """the library's segments, yielding before every exchange what `_collective` / `in_process` need to move its buffers"""
        from . import _lib
        kp, dp, outs = self._lib_args(keys, drop)
        nx = self.exchanges_per_tick()
        for seg in range(nx + 1):
            _lib.check(self._L.smr_ep_spread_segment(self._lib_h, seg, kp, dp, outs, _lib.stream_ptr(None)))
            if seg < nx:
                yield self._lib_plan(seg)
        self._lib_results()

    def _lib_plan(self, k):
        """exchange k's buffers as the dict `_collective` takes (tensors over the library's memory)"""
        b = self._lib_bufs[k]
        if "sbuf" not in b:
            n_send, n_recv = sum(b["in_split"]), sum(b["out_split"])
            b.update(kind=k, n_send=n_send, n_recv=n_recv, sbuf=_tensor_over(self.torch, b["send"], max(n_send, 8), self.device),
                     rbuf=_tensor_over(self.torch, b["recv"], max(n_recv, 8), self.device))
        return b

    def close_library_tick(self):
        if getattr(self, "_lib_h", None):
            self._L.smr_ep_spread_destroy(self._lib_h)
            self._lib_h = None

    def bind_comm(self, comm):
        """every exchange of the tick through the library: `comm` (summerset_amd.comm.Comm, this rank's end of the job's
        communicator) -> `smr_comm_exchange` on the plans' own send / receive buffers with their static split sizes, on the
        stream the handlers' kernels run on.  None: back to torch.distributed.all_to_all_single (gloo jobs)."""
        if comm is not None and (comm.world != self.world or comm.rank != self.rank):
            raise ValueError("the communicator is rank %d of %d, the job's rank is %d of %d" % (comm.rank, comm.world, self.rank, self.world))
        self.comm = comm
        if getattr(self, "_lib_h", None):
            from . import _lib
            _lib.check(self._L.smr_ep_spread_bind_comm(self._lib_h, comm._h if comm is not None else None))

    def _get(self, plan, key):
        """the message (block, from, to) as tensors: views of the receive buffer, or the sender's own tensors"""
        if key in plan["local"]:
            return plan["local"][key]
        torch, (b, _, _) = self.torch, key
        lo, hi = self.range[b]
        G, off, m = hi - lo, plan["roff"][key], {}
        dt = {8: torch.int64, 4: torch.int32, 1: torch.uint8}
        for name, w, c in _KINDS[plan["kind"]]:
            c = self.R if c == "R" else c
            n = w * c * G
            t = plan["rbuf"][off:off + n].view(dt[w])
            m[name] = t.reshape(c, G) if c > 1 else t
            off += n
        return m

    def _c(self, b, kind, v):
        """constant per-group tensors (peer id, the default ballot s + 1 of an instance's own leader)"""
        k = (b, kind, v)
        if k not in self._const:
            lo, hi = self.range[b]
            self._const[k] = self.torch.full((hi - lo,), v, dtype=self.torch.uint8 if kind == "u8" else self.torch.int64, device=self.device)
        return self._const[k]

    # ---- the tick, as the compute stages between the exchanges ----------------------------------------------------------
    def _steps(self, keys, drop=None):
        """yields the plan of every exchange once this rank has posted its messages for it; the caller runs the
        collective (or, inside one process, the copy) and resumes.  self.out[(b, s)] = the leader's results."""
        torch, R = self.torch, self.R
        mine = sorted(self.reps)
        pa, dec, acc = {}, {}, {}
        # -- propose; PreAccept to every peer
        out = {}
        for b, s in mine:
            pa[(b, s)] = self.reps[(b, s)].handle_req_batch(keys[(b, s)], None)
            for q in range(R):
                if q != s:
                    fl = pa[(b, s)]["flags"]
                    if drop is not None and (b, s, q) in drop:
                        fl = torch.where(drop[(b, s, q)], torch.zeros_like(fl), fl)
                    out[(b, s, q)] = dict(seq=pa[(b, s)]["seq"], deps=pa[(b, s)]["deps"], col=pa[(b, s)]["col"], flags=fl, key=keys[(b, s)])
        p = self._plans[("pre_accept", 0)]
        self._post(p, out)
        yield p
        # -- acceptors: one sender's PreAccept at a time, senders ascending
        out = {}
        for b, q in mine:
            for s in range(R):
                if s != q:
                    m = self._get(p, (b, s, q))
                    r = self.reps[(b, q)].handle_msg_pre_accept(dict(m, peer=self._c(b, "u8", s), ballot=self._c(b, "i64", s + 1)))
                    out[(b, q, s)] = r
        p = self._plans[("pa_reply", 0)]
        self._post(p, out)
        yield p
        replies = {}
        for b, s in mine:
            replies[(b, s)] = {q: self._get(p, (b, q, s)) for q in range(R) if q != s}
        for i, leaders in enumerate(self._leader_sets):
            # -- command leaders: the fast-quorum decision; Accepts where the slow path was taken
            out = {}
            for b, s in mine:
                if s not in leaders:
                    continue
                rp = replies[(b, s)]
                st = lambda f, own: torch.stack([own if q == s else rp[q][f] for q in range(R)])
                d = self.reps[(b, s)].handle_msg_pre_accept_reply(pa[(b, s)]["col"], st("ballot", self._zero_b[b]), st("seq", self._zero_b[b]),
                                                                  st("deps", self._none_d[b]), st("flags", self._zero_f[b]))
                d["slow"] = (d["decision"] == 2).to(torch.uint8)
                dec[(b, s)] = d
                for q in range(R):
                    if q != s:
                        out[(b, s, q)] = dict(seq=d["seq"], deps=d["deps"], col=pa[(b, s)]["col"], flags=d["slow"], key=keys[(b, s)])
            p = self._plans[("accept", i)]
            self._post(p, out)
            yield p
            out = {}
            for b, q in mine:
                for s in leaders:
                    if s != q:
                        m = self._get(p, (b, s, q))
                        out[(b, q, s)] = self.reps[(b, q)].handle_msg_accept(dict(m, peer=self._c(b, "u8", s), ballot=self._c(b, "i64", s + 1)))
            p = self._plans[("acc_reply", i)]
            self._post(p, out)
            yield p
            # -- command leaders: the slow-path tally; CommitNotice to every peer
            out = {}
            for b, s in mine:
                if s not in leaders:
                    continue
                ar = {q: self._get(p, (b, q, s)) for q in range(R) if q != s}
                st = lambda f, own: torch.stack([own if q == s else ar[q][f] for q in range(R)])
                a = self.reps[(b, s)].handle_msg_accept_reply(pa[(b, s)]["col"], st("ballot", self._zero_b[b]), st("flags", self._zero_f[b]))
                d = dec[(b, s)]
                committed = ((d["decision"] == 3) | (a["committed"] == 1)).to(torch.uint8)
                acc[(b, s)] = committed
                for q in range(R):
                    if q != s:
                        out[(b, s, q)] = dict(seq=d["seq"], deps=d["deps"], col=pa[(b, s)]["col"], flags=committed, key=keys[(b, s)])
            p = self._plans[("commit", i)]
            self._post(p, out)
            yield p
            for b, q in mine:
                for s in leaders:
                    if s != q:
                        m = self._get(p, (b, s, q))
                        self.reps[(b, q)].handle_msg_commit_notice(dict(m, peer=self._c(b, "u8", s), ballot=self._c(b, "i64", s + 1)))
        self.out = {(b, s): dict(col=pa[(b, s)]["col"], proposed=pa[(b, s)]["flags"], decision=dec[(b, s)]["decision"], committed=acc[(b, s)],
                                 seq=dec[(b, s)]["seq"], deps=dec[(b, s)]["deps"]) for b, s in mine}

    def tick(self, keys, drop=None):
        """keys[(b, r)]: uint8 [groups of block b] device tensor for every replica that lives here (0xFF = no proposal);
        drop[(b, s, q)] (optional): bool per group, the PreAccept from s to q is lost (with its reply).  Returns
        {(b, s): dict(col, proposed, decision, committed, seq, deps)} for the command leaders of this rank."""
        if getattr(self, "_lib_h", None):
            if self.comm is not None or self.world == 1:               # ONE call: segments and exchanges inside the library
                from . import _lib
                kp, dp, outs = self._lib_args(keys, drop)
                _lib.check(self._L.smr_ep_spread_tick(self._lib_h, kp, dp, outs, _lib.stream_ptr(None)))
                return self._lib_results()
            for plan in self._lib_steps(keys, drop):                   # (a gloo job: the library's segments, torch moves the buffers)
                self._collective(plan)
            return self.out
        for plan in self._steps(keys, drop):
            self._collective(plan)
        return self.out

    def exchanges_per_tick(self):
        return 2 + 3 * len(self._leader_sets)

    def committed(self, out=None):
        """instances this rank's command leaders committed in the last tick (a device scalar)"""
        out = self.out if out is None else out
        return sum(o["committed"].sum() for o in out.values())


def _tensor_over(torch, ptr, nbytes, device):
    """a uint8 tensor over `nbytes` of device (or, on the emulator, host) memory the library owns"""
    import ctypes as C
    import numpy as np
    if str(device).startswith("cuda"):
        class _Mem:                                          # __cuda_array_interface__: torch.as_tensor wraps device memory without a copy
            pass
        m = _Mem()
        m.__cuda_array_interface__ = dict(shape=(int(nbytes),), typestr="|u1", data=(int(ptr), False), version=2)
        return torch.as_tensor(m, device=device)
    return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * int(nbytes)).from_address(int(ptr))))


class in_process:
    """All `world` ranks of a spread EPaxos job inside one process (one device, or the emulator): the same objects,
    plans and buffers as the multi-process job; only the collective is a copy.  Every rank runs a stage before any rank
    starts the next one -- the order the collectives impose on separate processes."""

    def __init__(self, total_groups, population, world, device, **kw):
        self.ranks = [SpreadEPaxos(total_groups, population, r, world, device, **kw) for r in range(world)]

    def tick(self, keys, drop=None):
        """keys / drop for every (block, replica) of the job; every rank picks its own"""
        gens = [(r._lib_steps(keys, drop) if getattr(r, "_lib_h", None) else r._steps(keys, drop)) for r in self.ranks]
        while True:
            plans = []
            for g in gens:
                try:
                    plans.append(next(g))
                except StopIteration:
                    pass
            if not plans:
                break
            assert len(plans) == len(self.ranks)
            for s_, p in enumerate(plans):
                so = 0
                for d, n in enumerate(p["in_split"]):
                    q = plans[d]
                    ro = sum(q["out_split"][:s_])
                    assert q["out_split"][s_] == n and q["kind"] == p["kind"]
                    if n:
                        q["rbuf"][ro:ro + n].copy_(p["sbuf"][so:so + n])
                    so += n
        out = {}
        for r in self.ranks:
            out.update(r.out)
        return out
