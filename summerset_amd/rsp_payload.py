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
    def __init__(self, n_groups, population=5, window=32, max_data_len=4096, num_data_shards=None):
        self.G, self.R, self.W = int(n_groups), int(population), int(window)
        self.d = int(num_data_shards) if num_data_shards is not None else self.R // 2 + 1     # majority (mod.rs:606-611)
        self.max_data_len = int(max_data_len)
        self._L = _lib.load()
        h = C.c_void_p()
        check(self._L.smr_rsp_pstore_create(self.G, self.R, self.d, self.W, self.max_data_len, C.byref(h)))
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

    def counters(self):
        c = np.zeros(4, np.uint64)
        check(self._L.smr_rsp_pstore_counters(self._h, c.ctypes.data_as(C.c_void_p)))
        return dict(copied=int(c[0]), rebuilt=int(c[1]), unsatisfied=int(c[2]), rekeyed=int(c[3]))


class RSPaxosReplicaWithPayload:
    """An `RSPaxosReplicaGroup` and its payload store as one object: the replica's handlers, each followed by `store.follow`,
    and `req_batch` by the `put` of the serialized batches.  A handler that consumes a message carrying shards names where
    they come from -- the peers' stores (`set_peers`, indexed by replica id) stand for the message's payload:
        accept(peer=s, ...)            shard {me} of the sender's REQS plane        (Accept, request.rs:127-142)
        prepare_replies(peer=q, ...)   the voted shards of q's VOTED plane          (PrepareReply, messages.rs:55-83)
        reconstruct_reply(...)         any peer's REQS plane (the call names no sender; ReconstructReply, messages.rs:467-515)
    every other handler moves no bytes between replicas: what its commit-bar run or prepare quorum asks for is rebuilt from
    the shards the row holds.  What a co-located cluster (every replica's store in this GPU's HBM) runs per handler; the
    lock-step schedule consumes every message in the tick that produced it, so a sender's row still holds the token the
    message named when `follow` runs."""
    HANDLERS = ("accept", "accept_replies", "become_leader", "prepare", "prepare_replies", "reconstruct", "reconstruct_reply", "heartbeat",
                "bcast_heartbeat")
    CARRIES = {"accept": REQS, "prepare_replies": VOTED, "reconstruct_reply": REQS}

    def __init__(self, replica, store, payload=None):
        self.replica, self.store, self.payload = replica, store, payload     # payload(val) -> (data uint8 [G, L], lens int32 [G] or None)
        self.G, self.R, self.W, self.me = replica.G, replica.R, replica.W, replica.me
        self.peers = []

    def set_peers(self, cluster):
        """`cluster`: the R objects of this kind, by replica id"""
        self.peers = [None if o is self else o.store for o in cluster]

    def req_batch(self, val, data=None, lens=None, stream=None, out=None):
        acc = self.replica.req_batch(val, stream=stream, out=out)
        if data is None and self.payload is not None:
            data, lens = self.payload(val)
        if data is not None:
            self.store.put(acc, data, lens, stream=stream)
        self.store.follow(self.replica, stream=stream)
        return acc

    def __getattr__(self, name):
        fn = getattr(self.replica, name)
        if name not in self.HANDLERS:
            return fn

        def call(*a, **kw):
            out = fn(*a, **kw)
            plane = self.CARRIES.get(name)
            if plane is None:
                self.store.follow(self.replica, stream=kw.get("stream"))
            else:
                peer = kw.get("peer", a[1] if name == "accept" and len(a) > 1 else (a[0] if name == "prepare_replies" and a else None))
                self.store.follow(self.replica, [None if p is None else (p, plane) for p in self.peers], sel=peer, stream=kw.get("stream"))
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
