"""Layout L2 (the north star's "RCCL all-to-all over xGMI standing in for the inter-replica message
fan-out") for the MultiPaxos cluster engine, device-resident.

The job's groups are block-partitioned over the ranks (shard.group_range); replica r of block b lives on rank
(b + r) mod world, so a group's five replicas sit on up to five GPUs and EVERY message of the protocol crosses a
rank boundary: the leader's Accepts (and a candidate's Prepares, step-up Heartbeats) after R1, the followers'
AcceptReplies and PrepareReplies after R2, the all-to-all Heartbeats after R3 -- the stand-in for
`server/transport.rs:208-275` (`send_msg` / `bcast_msg`).

Per block it takes part in, a rank holds one `MultiPaxosCluster` with the block's groups in which only its own replicas
are LIVE (`smr_mp_set_live`); the others are images, filled by the exchange.  One exchange = every live replica's piece
packed by a kernel straight into the send buffer (`smr_mp_image_pack`, fixed-size images: include/summerset_hip.h),
ONE `all_to_all_single` on device tensors with static split sizes, and the received pieces unpacked into the images
(`smr_mp_image_unpack`).  Nothing of the data path touches the host: no `.cpu()`, no counts to agree on first.  Three
exchanges per tick (two without a heartbeat round), each a single collective.

Results are bit for bit the co-located engine's (tests/test_spread_mp_gloo.py: world_size 2 over gloo with the
emulator build of the engine on every rank, against the single-process run, every tick).  Group freezes (`overflow`,
a harness guard) are per rank here: a run that overflows a ring is outside what the two layouts agree on."""
import numpy as np

from . import _lib, shard
from ._lib import MpImageOp, check, stream_ptr
from .multipaxos import MultiPaxosCluster

OUTBOX, ACKS, PREPARE_REPLIES, HEARTBEAT = 0, 1, 2, 3


def home(block, replica, world):
    """the rank replica `replica` of block `block` lives on"""
    return (block + replica) % world


class _Msg:
    __slots__ = ("src", "dst", "block", "kind", "rep", "other", "size", "soff", "roff", "dup_of")

    def __init__(self, src, dst, block, kind, rep, other=0):
        self.src, self.dst, self.block, self.kind, self.rep, self.other = src, dst, block, kind, rep, other
        self.size = self.soff = self.roff = 0
        self.dup_of = None


class SpreadMultiPaxos:
    def __init__(self, total_groups, population, window, rank, world, device, slots_per_tick, ovf_cap=8192, **cluster_kw):
        import torch
        self.R, self.W, self.rank, self.world, self.device, self.S = int(population), int(window), int(rank), int(world), device, int(slots_per_tick)
        self.ovf_cap = int(ovf_cap)
        self.torch = torch
        self._L = _lib.load()
        self.blocks = {}                                       # block -> (cluster, live replicas, lo, hi)
        for b in range(world):
            live = [r for r in range(self.R) if home(b, r, world) == rank]
            lo, hi = shard.group_range(total_groups, world, b)
            if live and hi > lo:
                cl = MultiPaxosCluster(hi - lo, self.R, self.W, straggler_ticks=0, **cluster_kw)
                check(self._L.smr_mp_set_live(cl._h, sum(1 << r for r in live)))
                self.blocks[b] = (cl, live, lo, hi)
        self.n_groups = {b: shard.group_range(total_groups, world, b) for b in range(world)}
        self.bytes_sent = 0
        self.peers = None                                      # set by in_process(): every rank's object, for a job inside one process
        self.comm = None                                       # set by bind_comm(): the exchanges run inside the library (RCCL)
        # the three exchanges: every rank derives the same global message lists, in the same order
        self._plans = {ph: self._plan(ph) for ph in ("outbox", "replies", "heartbeat")}
        # the tick's orchestration lives in the library: one C call per segment of the tick (smr_mp_spread_segment), the
        # collectives between them stay here
        import ctypes as C
        self._order = sorted(self.blocks)                      # the order the blocks' inputs are handed over in
        cls = (C.c_void_p * max(len(self._order), 1))(*[self.blocks[b][0]._h for b in self._order])
        names = ("outbox", "replies", "heartbeat")
        pk = (C.c_void_p * 3)(*[self._plans[n]["pack"] for n in names])
        up = (C.c_void_p * 3)(*[self._plans[n]["unpack"] for n in names])
        h = C.c_void_p()
        check(self._L.smr_mp_spread_create(cls, len(self._order), pk, up, C.byref(h)))
        self._spread = h

    # ---- static plan --------------------------------------------------------------------------------------------
    def _ranks_of(self, b):
        return sorted({home(b, r, self.world) for r in range(self.R)})

    def _messages(self, phase):
        out = []
        for b in range(self.world):
            lo, hi = self.n_groups[b]
            if hi <= lo:
                continue
            for r in range(self.R):
                hr = home(b, r, self.world)
                if phase in ("outbox", "heartbeat"):           # a replica's piece goes to every other rank that holds the block
                    out += [_Msg(hr, d, b, OUTBOX if phase == "outbox" else HEARTBEAT, r) for d in self._ranks_of(b) if d != hr]
                else:
                    out += [_Msg(hr, d, b, PREPARE_REPLIES, r) for d in self._ranks_of(b) if d != hr]
                    for q in range(self.R):                    # follower q's answers to sender r travel to r's rank
                        hq = home(b, q, self.world)
                        if q != r and hq != hr:
                            out.append(_Msg(hq, hr, b, ACKS, r, q))
        return out

    def _img_bytes(self, b, kind):
        lo, hi = self.n_groups[b]
        cl = self.blocks[b][0] if b in self.blocks else None
        if cl is not None:
            return int(self._L.smr_mp_image_bytes(cl._h, kind, self.S, self.ovf_cap))
        # a block this rank holds no replica of never appears in its messages
        raise AssertionError("image size of a block this rank takes no part in")

    def _plan(self, phase):
        torch = self.torch
        msgs = self._messages(phase)
        send = sorted([m for m in msgs if m.src == self.rank], key=lambda m: m.dst)       # stable: canonical order per dst
        recv = sorted([m for m in msgs if m.dst == self.rank], key=lambda m: m.src)
        in_split, out_split = [0] * self.world, [0] * self.world
        off, first = 0, {}
        for m in send:
            m.size, m.soff = self._img_bytes(m.block, m.kind), off
            key = (m.block, m.kind, m.rep, m.other)
            m.dup_of = first.get(key)                          # the same piece for another rank: packed once, copied
            first.setdefault(key, m)
            off += m.size
            in_split[m.dst] += m.size
        n_send = off
        off = 0
        for m in recv:
            m.size, m.roff = self._img_bytes(m.block, m.kind), off
            off += m.size
            out_split[m.src] += m.size
        plan = dict(send=send, recv=recv, in_split=in_split, out_split=out_split,
                    sbuf=torch.zeros(max(n_send, 16), dtype=torch.uint8, device=self.device),
                    rbuf=torch.zeros(max(off, 16), dtype=torch.uint8, device=self.device))
        # the exchange's operations never change: handed to the library once (smr_mp_image_plan_create), an exchange is
        # then 3 launches to pack (clear headers, pack, duplicate) + 1 to unpack instead of one or two per image
        plan["pack"] = self._make_plan(send, plan["sbuf"].data_ptr(), True)
        plan["unpack"] = self._make_plan(recv, plan["rbuf"].data_ptr(), False)
        return plan

    def _make_plan(self, msgs, base, sending):
        import ctypes as C
        ops = (MpImageOp * max(len(msgs), 1))()
        for o, m in zip(ops, msgs):
            o.cluster, o.kind, o.rep, o.other = self.blocks[m.block][0]._h, m.kind, m.rep, m.other
            o.img_dev, o.img_bytes = base + (m.soff if sending else m.roff), m.size
            o.copy_of_dev = (base + m.dup_of.soff) if (sending and m.dup_of is not None) else None
        h = C.c_void_p()
        check(self._L.smr_mp_image_plan_create(ops, len(msgs), self.S, self.ovf_cap, C.byref(h)))
        return h

    def close(self):
        if getattr(self, "_spread", None):
            self._L.smr_mp_spread_destroy(self._spread)
            self._spread = None
        for p in getattr(self, "_plans", {}).values():
            for k in ("pack", "unpack"):
                if p.get(k):
                    self._L.smr_mp_image_plan_destroy(p[k])
                    p[k] = None

    def __del__(self):
        self.close()

    # ---- one exchange: pack -> ONE all_to_all_single -> unpack ---------------------------------------------------------
    def _pack(self, phase, stream=None):
        p = self._plans[phase]
        check(self._L.smr_mp_image_plan_run(p["pack"], 0, stream_ptr(stream)))

    def _unpack(self, phase, stream=None):
        check(self._L.smr_mp_image_plan_run(self._plans[phase]["unpack"], 1, stream_ptr(stream)))

    def _collective(self, phase):
        """the exchange itself: ONE all_to_all_single over the plan's send / receive buffers (RCCL over xGMI)"""
        import torch.distributed as dist
        p = self._plans[phase]
        self.bytes_sent += sum(p["in_split"])
        if self.world > 1:
            if self.peers is not None:                         # all ranks of the job in THIS process (tests, one device)
                _copy_between(self.peers, phase)
            else:
                # (the buffers are padded to a minimum size: the collective sees exactly the planned bytes)
                dist.all_to_all_single(p["rbuf"][:sum(p["out_split"])], p["sbuf"][:sum(p["in_split"])], output_split_sizes=p["out_split"],
                                       input_split_sizes=p["in_split"])

    def bind_comm(self, comm):
        """the collectives into the library: `comm` (summerset_amd.comm.Comm, one per rank of the job) and the three exchanges'
        buffers / split sizes are handed to the spread object once (`smr_mp_spread_bind_comm`); `tick` is then ONE C call
        (`smr_mp_spread_tick`: segments and RCCL exchanges enqueued back to back on the stream).  None unbinds."""
        import ctypes as C
        if comm is None:
            check(self._L.smr_mp_spread_bind_comm(self._spread, None, None, None, None, None, self.world))
            self.comm = None
            return
        names = ("outbox", "replies", "heartbeat")
        sd = (C.c_void_p * 3)(*[self._plans[n]["sbuf"].data_ptr() for n in names])
        rd = (C.c_void_p * 3)(*[self._plans[n]["rbuf"].data_ptr() for n in names])
        sb = (C.c_uint64 * (3 * self.world))(*[int(x) for n in names for x in self._plans[n]["in_split"]])
        rb = (C.c_uint64 * (3 * self.world))(*[int(x) for n in names for x in self._plans[n]["out_split"]])
        check(self._L.smr_mp_spread_bind_comm(self._spread, comm._h, C.byref(sd), sb, C.byref(rd), rb, self.world))
        self.comm = comm

    def abort_tick(self):
        """after a segment or a collective failed: the object takes a new tick (and bind_comm) again; the blocks hold a partly
        run tick and are the caller's to restore"""
        check(self._L.smr_mp_spread_abort_tick(self._spread))

    def _exchange(self, phase, stream=None):
        self._pack(phase, stream)
        self._collective(phase)
        self._unpack(phase, stream)

    def _inputs(self, inputs):
        from ._lib import MpTickIn
        arr = (MpTickIn * max(len(self._order), 1))()
        for a, b in zip(arr, self._order):
            x = inputs[b]
            rv = x.get("req_val")
            ptr = lambda t: None if t is None else t.data_ptr()
            a.timeout_rep_dev, a.timeout_src_dev, a.req_target_dev = ptr(x.get("timeout_rep")), ptr(x.get("timeout_src")), ptr(x.get("req_target"))
            a.req_cnt_dev, a.req_val_dev, a.S = ptr(x.get("req_cnt")), ptr(rv), (0 if rv is None else int(rv.shape[0]))
            a.ackctl_dev = ptr(x.get("ackctl"))
        return arr

    def segment(self, k, arr, heartbeat, stream=None):
        check(self._L.smr_mp_spread_segment(self._spread, k, arr, int(bool(heartbeat)), stream_ptr(stream)))

    # ---- the tick ---------------------------------------------------------------------------------------------------
    def tick(self, inputs, heartbeat=False, stream=None):
        """inputs[b] = dict(timeout_rep, timeout_src, req_target, req_cnt, req_val, ackctl) of block b's groups (device
        tensors; every rank that holds block b passes the same arrays -- the streams are keyed by global group id).
        Three or four library calls (the segments between the collectives) and two or three collectives."""
        arr = self._inputs(inputs)
        if self.comm is not None:                              # the exchanges are the library's: the whole tick is one call
            check(self._L.smr_mp_spread_tick(self._spread, arr, int(bool(heartbeat)), stream_ptr(stream)))
            self.bytes_sent += sum(sum(self._plans[p]["in_split"]) for p in (("outbox", "replies", "heartbeat") if heartbeat else ("outbox", "replies")))
            return
        self.segment(0, arr, heartbeat, stream)
        self._collective("outbox")
        self.segment(1, arr, heartbeat, stream)
        self._collective("replies")
        self.segment(2, arr, heartbeat, stream)
        if heartbeat:
            self._collective("heartbeat")
            self.segment(3, arr, heartbeat, stream)

    def tick_call_by_call(self, inputs, heartbeat=False, stream=None):
        """the same tick with every round and every pack / unpack its own host call (rounds 1-2; kept for the comparison)"""
        for b, (cl, _, _, _) in self.blocks.items():
            x = inputs[b]
            cl.round_local(x.get("timeout_rep"), x.get("timeout_src"), x.get("req_target"), x.get("req_cnt"), x.get("req_val"), stream=stream)
        self._exchange("outbox", stream)
        for b, (cl, _, _, _) in self.blocks.items():
            cl.round_deliver(stream=stream)
        self._exchange("replies", stream)
        for b, (cl, _, _, _) in self.blocks.items():
            cl.round_replies(inputs[b].get("ackctl"), publish_heartbeat=heartbeat, stream=stream)
        if heartbeat:
            self._exchange("heartbeat", stream)
            for b, (cl, _, _, _) in self.blocks.items():
                cl.round_heartbeat(stream=stream)
        for b, (cl, _, _, _) in self.blocks.items():
            cl.end_tick()

    def preset_leader(self, rep=0):
        for cl, _, _, _ in self.blocks.values():
            cl.preset_leader(rep)

    def commits(self):
        """leader-side commits of the replicas that live here"""
        return sum(cl.counters(r)["commits"] for cl, live, _, _ in self.blocks.values() for r in live)

    def dropped_overflow_entries(self):
        """overflow-list entries lost because an image's list ran full (must be 0: size ovf_cap for the workload)"""
        n = 0
        for p in self._plans.values():
            for m, buf in [(m, p["sbuf"]) for m in p["send"]] + [(m, p["rbuf"]) for m in p["recv"]]:
                off = m.soff if buf is p["sbuf"] else m.roff
                n += int(buf[off:off + 16].view(self.torch.int32)[2].item())
        return n


def _copy_between(peers, phase):
    """the all-to-all of a job whose ranks all live in this process: rank s's segment for d -> d's segment from s.  Runs
    once per exchange, when the LAST rank arrives (the ranks are stepped one after the other, see in_process).  The segment
    pairs of an exchange never change: their views are made once, and an exchange is ONE multi-tensor copy (round 5: twelve
    `copy_` calls at ~10 us of host time each were a tenth of a virtual-rank tick)."""
    peers[0]._arrived[phase] = peers[0]._arrived.get(phase, 0) + 1
    if peers[0]._arrived[phase] < len(peers):
        return
    peers[0]._arrived[phase] = 0
    pairs = peers[0]._pairs.get(phase) if hasattr(peers[0], "_pairs") else None
    if pairs is None:
        dsts, srcs = [], []
        for s_, ps in enumerate(peers):
            p = ps._plans[phase]
            so = 0
            for d, n in enumerate(p["in_split"]):
                q = peers[d]._plans[phase]
                ro = sum(q["out_split"][:s_])
                assert q["out_split"][s_] == n
                if n:
                    dsts.append(q["rbuf"][ro:ro + n]); srcs.append(p["sbuf"][so:so + n])
                so += n
        pairs = (dsts, srcs)
        if not hasattr(peers[0], "_pairs"):
            peers[0]._pairs = {}
        peers[0]._pairs[phase] = pairs
    if pairs[0]:
        peers[0].torch._foreach_copy_(pairs[0], pairs[1])


class in_process:
    """All `world` ranks of a spread job inside one process (one device, or the emulator): the same objects, plans,
    pack / unpack kernels and buffers as the multi-process job; only the collective is a copy.  tick() steps every rank
    through a round before any rank starts the next one -- the order the collectives impose on separate processes."""

    def __init__(self, total_groups, population, window, world, device, slots_per_tick, **kw):
        self.ranks = [SpreadMultiPaxos(total_groups, population, window, r, world, device, slots_per_tick, **kw) for r in range(world)]
        self.ranks[0]._arrived = {}
        for r in self.ranks:
            r.peers = self.ranks
        # on a device every virtual rank gets a stream of its own: ranks are processes on GPUs of their own in the real job, so
        # nothing orders rank 1's kernels behind rank 0's -- only the collectives do
        self._streams = None
        if getattr(device, "type", str(device)) == "cuda":
            import torch
            self._streams = [torch.cuda.Stream(device=device) for _ in self.ranks]

    def preset_leader(self, rep=0):
        for r in self.ranks:
            r.preset_leader(rep)

    def tick(self, inputs, heartbeat=False):
        """every rank through a segment before any rank starts the next one -- the order the collectives impose on separate
        processes; the collective between two segments is a copy between the ranks' buffers"""
        rs = self.ranks
        arrs = [r._inputs(inputs) for r in rs]
        sts = self._streams
        if sts is not None:
            import torch
            main = torch.cuda.current_stream()
            for st in sts:
                st.wait_stream(main)                           # the tick's inputs were made on the caller's stream

        def segment(k):
            for i, (r, a) in enumerate(zip(rs, arrs)):
                r.segment(k, a, heartbeat, stream=None if sts is None else sts[i].cuda_stream)

        def collective(phase):
            for r in rs:
                r.bytes_sent += sum(r._plans[phase]["in_split"])
            if sts is not None:                                # the collective: every rank has packed; the copies; every rank goes on
                for st in sts:
                    main.wait_stream(st)
            for r in rs:
                _copy_between(rs, phase)
            if sts is not None:
                for st in sts:
                    st.wait_stream(main)
        segment(0)
        collective("outbox")
        segment(1)
        collective("replies")
        segment(2)
        if heartbeat:
            collective("heartbeat")
            segment(3)
        if sts is not None:
            for st in sts:
                main.wait_stream(st)

    def tick_call_by_call(self, inputs, heartbeat=False):
        rs = self.ranks

        def each(fn):
            for r in rs:
                for b, (cl, _, _, _) in r.blocks.items():
                    fn(cl, inputs[b])

        def exchange(phase):
            for r in rs:
                r._pack(phase)
            for r in rs:
                _copy_between(rs, phase)
            for r in rs:
                r._unpack(phase)
        each(lambda cl, x: cl.round_local(x.get("timeout_rep"), x.get("timeout_src"), x.get("req_target"), x.get("req_cnt"), x.get("req_val")))
        exchange("outbox")
        each(lambda cl, x: cl.round_deliver())
        exchange("replies")
        each(lambda cl, x: cl.round_replies(x.get("ackctl"), publish_heartbeat=heartbeat))
        if heartbeat:
            exchange("heartbeat")
            each(lambda cl, x: cl.round_heartbeat())
        each(lambda cl, x: cl.end_tick())
