"""The shard bytes behind an RSPaxos replica's instances (`smr_rsp_pstore_*`, csrc/rsp_payload.hip).

`RSPaxosReplicaGroup` keeps a codeword as (batch token, mask of shards present); `RSPaxosPayloadStore` keeps what the
reference keeps in `inst.reqs_cw` and `inst.voted.1` (src/protocols/rspaxos/mod.rs:168-233): the shards, in HBM, as two
planes (REQS, VOTED) of a ring of `window` rows x `population` shards x G groups.  The engine decides which shards exist
where, the store makes the bytes follow:

    acc = replica.req_batch(tokens)               # handle_req_batch: which slot, which ballot
    store.put(acc, data)                          # from_data + compute_parity of the serialized batches (request.rs:71-101)
    ...                                           # any handler call of `replica`: accept, prepare_replies, reconstruct_reply, ...
    store.follow(replica, sources=[(peer_store, REQS), ...])

`follow` takes, for every ring cell, the shards the engine's mask has and the row lacks from the sources that hold the same
token (the sender's `subset_copy`, the receiver's `inst.reqs_cw = ...` / `absorb_other`, rscoding.rs:255-346) and rebuilds
what is still missing from any `majority` shards present (`reconstruct_data` on commit and at the prepare quorum,
`compute_parity` for the re-Accepts: durability.rs:140-160, messages.rs:227-259).  `get_data` is `RSCodeword::get_data`
for the instances a handler executed.  Every call is one C-ABI call on device tensors; there is no CPU path.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, stream_ptr

REQS, VOTED = 0, 1
NULL = 0xFFFFFFFF


def _ptr(t):
    return None if t is None else t.data_ptr()


class RSPaxosPayloadStore:
    _CREATE = "smr_rsp_pstore_create"

    def __init__(self, n_groups, population=5, window=32, max_data_len=4096, num_data_shards=None):
        self.G, self.R, self.W = int(n_groups), int(population), int(window)
        self.d = int(num_data_shards) if num_data_shards is not None else self.R // 2 + 1     # majority (mod.rs:606-611)
        self.max_data_len = int(max_data_len)
        self._L = _lib.load()
        h = C.c_void_p()
        check(getattr(self._L, self._CREATE)(self.G, self.R, self.d, self.W, self.max_data_len, C.byref(h)))
        self._h = h
        rs, ss, gs = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._L.smr_rsp_pstore_layout(self._h, 0, None, C.byref(rs), C.byref(ss), C.byref(gs)))
        self.row_stride, self.shard_stride, self.group_stride = int(rs.value), int(ss.value), int(gs.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_rsp_pstore_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    # ---- the data path --------------------------------------------------------------------------------------------
    def put(self, accepts, data, lens=None, stream=None):
        """`accepts`: what `RSPaxosReplicaGroup.req_batch` returned (device tensors a_n, a_slot, a_val); `data`: uint8 [G, L]
        serialized batches (rows contiguous), `lens`: int32 [G] per-group lengths <= L or None"""
        if data.dim() != 2 or data.stride(1) != 1 or int(data.shape[0]) != self.G:
            raise _lib.SummersetError(_lib.SMR_ERR_ARG, "data must be uint8 [G, L] with contiguous rows")
        check(self._L.smr_rsp_pstore_put(self._h, _ptr(accepts["a_n"]), _ptr(accepts["a_slot"]), _ptr(accepts["a_val"]), data.data_ptr(),
                                         int(data.stride(0)), _ptr(lens), int(data.shape[1]), stream_ptr(stream)))

    def follow(self, replica, sources=(), sel=None, stream=None):
        """`replica`: the RSPaxosReplicaGroup whose bytes these are; `sources`: (store, plane) pairs a shard may come from (None:
        an empty seat); `sel`: uint8 [G] or None -- in group g only sources[sel[g]] may give shards (a handler's `peer` array)"""
        n = len(sources)
        arr = (C.c_void_p * max(n, 1))(*[None if s is None else s[0]._h for s in sources])
        planes = (C.c_uint8 * max(n, 1))(*[0 if s is None else int(s[1]) for s in sources])
        check(self._L.smr_rsp_pstore_follow(self._h, replica._h, n, arr, C.cast(planes, C.c_void_p), _ptr(sel), stream_ptr(stream)))

    @staticmethod
    def follow_many(stores, replicas, source=None, stream=None):
        """`follow` for several replicas that consumed ONE sender's message: stores[k] follows replicas[k], each with the single
        source `source` = (store, plane), which is none of them -- two launches for all of them"""
        n = len(stores)
        sa = (C.c_void_p * n)(*[s._h for s in stores])
        ra = (C.c_void_p * n)(*[r._h for r in replicas])
        check(stores[0]._L.smr_rsp_pstore_follow_many(n, sa, ra, None if source is None else source[0]._h, 0 if source is None else int(source[1]),
                                                      stream_ptr(stream)))

    def put_follow_all(self, replica, accepts, data, followers=(), follower_replicas=(), lens=None, stream=None):
        """a co-located leader's tick of the byte path in ONE call and four launches (`smr_rsp_pstore_put_follow_all`):
        `put(accepts, data, lens)` + `follow(replica)` + `follow_many(followers, follower_replicas, source=(self, REQS))`"""
        if data.dim() != 2 or data.stride(1) != 1 or int(data.shape[0]) != self.G:
            raise _lib.SummersetError(_lib.SMR_ERR_ARG, "data must be uint8 [G, L] with contiguous rows")
        n = len(followers)
        sa = (C.c_void_p * max(n, 1))(*[s._h for s in followers])
        ra = (C.c_void_p * max(n, 1))(*[r._h for r in follower_replicas])
        check(self._L.smr_rsp_pstore_put_follow_all(self._h, replica._h, _ptr(accepts["a_n"]), _ptr(accepts["a_slot"]), _ptr(accepts["a_val"]),
                                                    data.data_ptr(), int(data.stride(0)), _ptr(lens), int(data.shape[1]), n, sa, ra,
                                                    stream_ptr(stream)))

    def get_data(self, slot, group=None, expect=None, stream=None):
        """serialized batches of the instances (group[i] or i, slot[i]) -> (uint8 [n, max_data_len], int32 [n] lengths, bool [n] ok)"""
        import torch
        n = int(slot.shape[0])
        out = torch.zeros((n, self.max_data_len), dtype=torch.uint8, device=slot.device)
        ln = torch.zeros(n, dtype=torch.int32, device=slot.device)
        ok = torch.zeros(n, dtype=torch.uint8, device=slot.device)
        check(self._L.smr_rsp_pstore_get_data(self._h, n, _ptr(group), _ptr(slot), _ptr(expect), out.data_ptr(), self.max_data_len,
                                              ln.data_ptr(), ok.data_ptr(), stream_ptr(stream)))
        return out, ln, ok.bool()

    # ---- a message's payload between replicas that do not share a device ----------------------------------------------
    def extract(self, slot, mask, plane=REQS, flags=None, out=None, stream=None):
        """sender: `subset_copy` of the rows `slot` [G] (shards `mask` [G]) into a message: dict(buf uint8 [R, G, group_stride],
        tok, mask, dlen [G]) -- `out`: an earlier message to refill"""
        import torch
        dev = slot.device
        m = out if out is not None else dict(buf=torch.zeros((self.R, self.G, self.group_stride), dtype=torch.uint8, device=dev),
                                             tok=torch.zeros(self.G, dtype=torch.int32, device=dev),
                                             mask=torch.zeros(self.G, dtype=torch.uint8, device=dev),
                                             dlen=torch.zeros(self.G, dtype=torch.int32, device=dev))
        check(self._L.smr_rsp_pstore_extract(self._h, int(plane), _ptr(flags), _ptr(slot), _ptr(mask), m["buf"].data_ptr(), m["tok"].data_ptr(),
                                             m["mask"].data_ptr(), m["dlen"].data_ptr(), stream_ptr(stream)))
        return m

    def ingest(self, msg, slot, plane=REQS, flags=None, stream=None):
        """receiver: the message becomes row `slot` of this (staging) store, replacing what the row held"""
        check(self._L.smr_rsp_pstore_ingest(self._h, int(plane), _ptr(flags), _ptr(slot), msg["tok"].data_ptr(), msg["mask"].data_ptr(),
                                            msg["dlen"].data_ptr(), msg["buf"].data_ptr(), stream_ptr(stream)))

    def emit_accepts(self, slot, ballot, mask, stride, plane=REQS, flags=None, stream=None):
        """the Accepts (slot, ballot, shards `mask` of the row) as wire frames with their payload: (uint8 [G, stride], int32 [G] lengths;
        0 = nothing to send, -1 = does not fit `stride`)"""
        import torch
        frames = torch.zeros((self.G, int(stride)), dtype=torch.uint8, device=slot.device)
        ln = torch.zeros(self.G, dtype=torch.int32, device=slot.device)
        check(self._L.smr_rsp_pstore_emit_accepts(self._h, int(plane), _ptr(flags), _ptr(slot), _ptr(ballot), _ptr(mask), frames.data_ptr(), int(stride),
                                                  ln.data_ptr(), stream_ptr(stream)))
        return frames, ln

    # ---- host-side reads ------------------------------------------------------------------------------------------
    def dump(self, plane=REQS):
        tok, av, ln = np.zeros((self.W, self.G), np.uint32), np.zeros((self.W, self.G), np.uint8), np.zeros((self.W, self.G), np.uint32)
        check(self._L.smr_rsp_pstore_dump(self._h, int(plane), tok.ctypes.data_as(C.c_void_p), av.ctypes.data_as(C.c_void_p),
                                          ln.ctypes.data_as(C.c_void_p)))
        return dict(tok=tok, avail=av, dlen=ln)

    def read_row(self, slot, plane=REQS):
        """uint8 [population, G, group_stride]: every shard of the slot's row (bytes beyond a shard's length are padding)"""
        buf = np.zeros((self.R, self.G, self.group_stride), np.uint8)
        check(self._L.smr_rsp_pstore_read_row(self._h, int(plane), int(slot), buf.ctypes.data_as(C.c_void_p)))
        return buf

    def plane_ptr(self, plane=REQS):
        p = C.c_void_p()
        check(self._L.smr_rsp_pstore_layout(self._h, int(plane), C.byref(p), None, None, None))
        return int(p.value)

    def voted_alias_ptr(self):
        """device pointer of the u8 [W][G] array: bit k = the VOTED shard k of that cell is read from the REQS plane"""
        p = C.c_void_p()
        check(self._L.smr_rsp_pstore_voted_alias(self._h, C.byref(p), None))
        return int(p.value)

    def voted_alias(self):
        """uint8 [W, G] on the host: which VOTED shards are aliases of the REQS row's"""
        a = np.zeros((self.W, self.G), np.uint8)
        check(self._L.smr_rsp_pstore_voted_alias(self._h, None, a.ctypes.data_as(C.c_void_p)))
        return a

    def counters(self):
        c = np.zeros(4, np.uint64)
        check(self._L.smr_rsp_pstore_counters(self._h, c.ctypes.data_as(C.c_void_p)))
        return dict(copied=int(c[0]), rebuilt=int(c[1]), unsatisfied=int(c[2]), rekeyed=int(c[3]))

    def delivered(self):
        """of `counters()["copied"]`: shards a sender's `put_follow_all` wrote here from its put launch (no copy out of its row)"""
        c = np.zeros(1, np.uint64)
        check(self._L.smr_rsp_pstore_debug_delivered(self._h, c.ctypes.data_as(C.c_void_p)))
        return int(c[0])


class CRaftPayloadStore(RSPaxosPayloadStore):
    """The shard bytes behind a CRaft replica's log (`smr_craft_pstore_*`): what the reference keeps in `LogEntry::reqs_cw`
    (src/protocols/craft/mod.rs:129-150), one plane keyed by log index.  `CRaftLeaderGroup` (leader or follower) keeps an
    entry's codeword as its avail_shards_map; the store makes the bytes follow:

        first = leader.handle_req_batch_emit(n_new)           # one entry per group: at slot log_len - 1
        store.put(leader, slot, data)                         # from_data + compute_parity, every shard (craft/request.rs:71-76)
        follower.handle_msg_append_entries(..., entry_mask=send[q])   # one's own shard, or the data shards in full-copy mode
        fstore.follow(follower, sources=[store])              # absorb what the message carried (craft/messages.rs:133-146); on
                                                              # commit, reconstruct_data's shards are rebuilt (messages.rs:193-233)
    `get_data`, `extract` / `ingest`, `dump`, `read_row` are the RSPaxos store's, on plane 0 (REQS)."""
    _CREATE = "smr_craft_pstore_create"

    def put(self, replica, slot, data, lens=None, stream=None):
        """`replica`: the CRaftLeaderGroup that appended; `slot`: int32 [G] the entry's log index per group (-1: none);
        `data`: uint8 [G, L] serialized batches (rows contiguous), `lens`: int32 [G] or None"""
        if data.dim() != 2 or data.stride(1) != 1 or int(data.shape[0]) != self.G:
            raise _lib.SummersetError(_lib.SMR_ERR_ARG, "data must be uint8 [G, L] with contiguous rows")
        check(self._L.smr_craft_pstore_put(self._h, replica._h, slot.data_ptr(), data.data_ptr(), int(data.stride(0)), _ptr(lens),
                                           int(data.shape[1]), stream_ptr(stream)))

    def follow(self, replica, sources=(), sel=None, stream=None):
        """`sources`: the stores a shard may come from (None: an empty seat); `sel`: uint8 [G] or None -- in group g only
        sources[sel[g]] may give shards (the sender of the message the handler consumed)"""
        n = len(sources)
        arr = (C.c_void_p * max(n, 1))(*[None if s is None else s._h for s in sources])
        check(self._L.smr_craft_pstore_follow(self._h, replica._h, n, arr, _ptr(sel), stream_ptr(stream)))

    @staticmethod
    def follow_many(stores, replicas, source=None, stream=None):
        """`follow` for several followers that consumed ONE leader's AppendEntries: stores[k] follows replicas[k], each with the
        single source `source` (a store, none of them) -- three launches for all of them"""
        n = len(stores)
        sa = (C.c_void_p * n)(*[s._h for s in stores])
        ra = (C.c_void_p * n)(*[r._h for r in replicas])
        check(stores[0]._L.smr_craft_pstore_follow_many(n, sa, ra, None if source is None else source._h, stream_ptr(stream)))

    def put_follow_all(self, replica, slot, data, followers=(), follower_replicas=(), lens=None, stream=None):
        """`put(replica, slot, data, lens)` + `follow(replica)` + `follow_many(followers, follower_replicas, source=self)` in ONE call
        and four launches (`smr_craft_pstore_put_follow_all`); behind the leader's append AND the followers' handlers"""
        if data.dim() != 2 or data.stride(1) != 1 or int(data.shape[0]) != self.G:
            raise _lib.SummersetError(_lib.SMR_ERR_ARG, "data must be uint8 [G, L] with contiguous rows")
        n = len(followers)
        sa = (C.c_void_p * max(n, 1))(*[s._h for s in followers])
        ra = (C.c_void_p * max(n, 1))(*[r._h for r in follower_replicas])
        check(self._L.smr_craft_pstore_put_follow_all(self._h, replica._h, slot.data_ptr(), data.data_ptr(), int(data.stride(0)), _ptr(lens),
                                                      int(data.shape[1]), n, sa, ra, stream_ptr(stream)))

    def emit_accepts(self, *a, **kw):
        raise _lib.SummersetError(_lib.SMR_ERR_STATE, "Accept frames are RSPaxos'")


class RSPaxosReplicaWithPayload:
    """An `RSPaxosReplicaGroup` and its payload store as one object: the replica's handlers, each followed by `store.follow`,
    and `req_batch` by the `put` of the serialized batches.  A handler that consumes a message carrying shards names where
    they come from -- the peers' stores (`set_peers`, indexed by replica id) stand for the message's payload:
        accept(peer=s, ...)            shard {me} of the sender's REQS plane        (Accept, request.rs:127-142)
        prepare_replies(peer=q, ...)   the voted shards of q's VOTED plane          (PrepareReply, messages.rs:55-83)
        reconstruct_reply(...)         the REQS plane of whoever answered the Reconstruct (ReconstructReply, messages.rs:467-515)
    every other handler moves no bytes between replicas: what its commit-bar run or prepare quorum asks for is rebuilt from
    the shards the row holds.  What a co-located cluster (every replica's store in this GPU's HBM) runs per handler; the
    lock-step schedule consumes every message in the tick that produced it, so a sender's row still holds the token the
    message named when `follow` runs."""
    HANDLERS = ("accept", "accept_replies", "become_leader", "prepare", "prepare_replies", "reconstruct", "reconstruct_reply", "heartbeat",
                "bcast_heartbeat")

    def __init__(self, replica, store, payload=None, staging=None):
        """`staging`: a second store of the same geometry -- the replica then takes NOTHING from its peers' stores: a message's payload
        is extracted at the sender (`subset_copy`), ingested into `staging`, and `follow` names the staging store as its only source
        (what replicas on different devices do, with the wire between `extract` and `ingest`)"""
        self.replica, self.store, self.payload, self.staging = replica, store, payload, staging   # payload(val) -> (data uint8 [G, L], lens int32 [G] or None)
        self.G, self.R, self.W, self.me = replica.G, replica.R, replica.W, replica.me
        self.cluster, self.shared, self._msg = [], {}, None

    def set_peers(self, cluster):
        """`cluster`: the R objects of this kind, by replica id"""
        self.cluster = list(cluster)
        self.shared = cluster[0].shared                                   # (who answered the last Reconstruct: see reconstruct_reply)

    def req_batch(self, val, data=None, lens=None, stream=None, out=None):
        acc = self.replica.req_batch(val, stream=stream, out=out)
        if data is None and self.payload is not None:
            data, lens = self.payload(val)
        if data is not None:
            self.store.put(acc, data, lens, stream=stream)
        self.store.follow(self.replica, stream=stream)
        return acc

    def reconstruct(self, *a, **kw):
        out = self.replica.reconstruct(*a, **kw)
        self.shared["responder"] = self.me                               # the reply names no sender: the host knows who it asked
        self.store.follow(self.replica, stream=kw.get("stream"))
        return out

    def _carry(self, sender, plane, rows, stream):
        """ship the named rows' shards from `sender`'s plane into my staging store: rows = [(flags, slot, mask)] device tensors"""
        for flags, slot, mask in rows:
            self._msg = sender.store.extract(slot, mask, plane, flags, out=self._msg, stream=stream)
            self.staging.ingest(self._msg, slot, REQS, flags, stream=stream)

    def _follow_peers(self, plane, sel, stream):
        """co-located: the peers' stores are the payload -- in group g only the store of replica sel[g] (None: any)"""
        srcs = [None if o is self else (o.store, plane) for o in self.cluster]
        self.store.follow(self.replica, srcs, sel=sel, stream=stream)

    @staticmethod
    def _uniform(peer):
        p = int(peer[0].item())
        if not bool((peer == p).all().item()):
            raise _lib.SummersetError(_lib.SMR_ERR_ARG, "one message = one sender: the `peer` array of a call with a staging store must be uniform")
        return p

    def accept(self, flags, peer, slot, ballot, val, mask, stream=None, out=None):
        sender = self._uniform(peer) if self.staging is not None else None
        if sender is not None:                                           # the Accept's shard, as the sender holds it NOW (before my handler runs)
            self._carry(self.cluster[sender], REQS, [(flags, slot, mask)], stream)
        res = self.replica.accept(flags, peer, slot, ballot, val, mask, stream=stream, out=out)
        if sender is not None:
            self.store.follow(self.replica, [(self.staging, REQS)], stream=stream)
        else:
            self._follow_peers(REQS, peer, stream)
        return res

    def prepare_replies(self, peer, pr_n, pr_trig, pr_endp, pr_ballot, pr_vbal, pr_vval, pr_vmask, stream=None):
        if self.staging is not None:                                     # row k of the reply = the voted shards of slot trig + k
            sender, n = self._uniform(peer), int(pr_n.max().item()) if pr_n.numel() else 0
            rows = [(((pr_n > k) & (pr_vbal[k] != 0)).to(pr_vmask.dtype), pr_trig + k, pr_vmask[k].contiguous()) for k in range(min(n, self.W))]
            self._carry(self.cluster[sender], VOTED, rows, stream)
        res = self.replica.prepare_replies(peer, pr_n, pr_trig, pr_endp, pr_ballot, pr_vbal, pr_vval, pr_vmask, stream=stream)
        if self.staging is not None:
            self.store.follow(self.replica, [(self.staging, REQS)], stream=stream)
        else:
            self._follow_peers(VOTED, peer, stream)
        return res

    def reconstruct_reply(self, flags, rr_n, rr_slot, rr_bal, rr_val, rr_mask, stream=None, responder=None):
        who = self.shared.get("responder") if responder is None else responder
        if self.staging is not None:
            n = int(rr_n.max().item()) if rr_n.numel() else 0
            rows = [(((rr_n > k) & (flags != 0)).to(rr_mask.dtype), rr_slot[k].contiguous(), rr_mask[k].contiguous()) for k in range(min(n, self.W))]
            self._carry(self.cluster[who], REQS, rows, stream)
        self.replica.reconstruct_reply(flags, rr_n, rr_slot, rr_bal, rr_val, rr_mask, stream=stream)
        if self.staging is not None:
            self.store.follow(self.replica, [(self.staging, REQS)], stream=stream)
        else:
            import torch
            sel = None if who is None else torch.full((self.G,), int(who), dtype=torch.uint8, device=flags.device)
            self._follow_peers(REQS, sel, stream)

    def __getattr__(self, name):
        fn = getattr(self.replica, name)
        if name not in self.HANDLERS:
            return fn

        def call(*a, **kw):                                              # handlers that move no bytes between replicas
            out = fn(*a, **kw)
            self.store.follow(self.replica, stream=kw.get("stream"))
            return out
        return call

    def executed_data(self, device, stream=None):
        """the serialized batches of the commands the LAST handler call executed (smr_rsp_exec_poll's list, in its order):
        (groups, slots, tokens, data uint8 [n, max_data_len], lens, ok) -- data / lens / ok are None when nothing ran"""
        import torch
        g, s, v = self.replica.exec_poll()
        if len(g) == 0:
            return g, s, v, None, None, None
        t = lambda a: torch.from_numpy(a.view(np.int32).copy()).to(device)
        data, ln, ok = self.store.get_data(t(s), group=t(g), expect=t(v), stream=stream)
        return g, s, v, data, ln, ok
