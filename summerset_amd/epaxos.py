"""Host-side handle of the batched EPaxos command leader / acceptor (G groups, one replica id).

Mirrors `EPaxosReplica` (src/protocols/epaxos/mod.rs) on the pre-execution path:
`handle_req_batch`, `handle_msg_pre_accept`, `handle_msg_pre_accept_reply` (the
fast-quorum decision), `handle_msg_accept`, `handle_msg_accept_reply`,
`handle_msg_commit_notice`; with `execute=True` every call is followed by the
dependency-graph execution the reference runs from `handle_logged_commit_slot`
(`attempt_execution`, `handle_cmd_result`; state read back by `exec_dump`); with
`recovery=True` the explicit prepare of a suspected peer's row: `heartbeat_timeout`,
`handle_msg_exp_prepare`, `handle_msg_exp_prepare_reply` (heartbeat.rs:17-125,
messages.rs:511-821), and `row=` on the other handlers for the instances a replica then
leads outside its own row.  Thin: every
method is one C-ABI call; messages are device tensors with one entry per group,
DepSets are int32 tensors [R, G] with -1 (0xFFFFFFFF) = None.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EpCfg, EpDumpBufs, EpExpPrepare, EpExpPrepareReply, EpMsg, check, stream_ptr

NONE, NO_KEY = 0xFFFFFFFF, 0xFF


def _ptr(t):
    return None if t is None else t.data_ptr()


class EPaxosReplicaGroup:
    def __init__(self, n_groups, population=5, me=0, window=32, n_keys=64, optimized_quorum=True, execute=False, recovery=False):
        self.G, self.R, self.me, self.W, self.K = int(n_groups), int(population), int(me), int(window), int(n_keys)
        self.execute, self.recovery = bool(execute), bool(recovery)
        cfg = EpCfg(self.G, self.R, self.me, int(optimized_quorum), int(self.execute), self.W, self.K, int(self.recovery))
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_ep_replica_create(C.byref(cfg), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_ep_replica_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _out(self, dev, with_deps=True):
        import torch
        G, R = self.G, self.R
        return dict(flags=torch.zeros(G, dtype=torch.uint8, device=dev), col=torch.zeros(G, dtype=torch.int32, device=dev),
                    ballot=torch.zeros(G, dtype=torch.int64, device=dev), seq=torch.zeros(G, dtype=torch.int64, device=dev),
                    deps=torch.zeros((R, G), dtype=torch.int32, device=dev) if with_deps else None)

    @staticmethod
    def _msg(d):
        return EpMsg(*[_ptr(d.get(k)) for k in ("flags", "peer", "col", "ballot", "seq", "deps", "key", "row")])

    def handle_req_batch(self, key, exploded=None, stream=None):
        """propose key[g] (0xFF = nothing) per group; returns the PreAccept tensors (flags, col, seq, deps)"""
        out = self._out(key.device)
        m = self._msg(out)
        check(self._L.smr_ep_propose(self._h, _ptr(key), _ptr(exploded), C.byref(m), stream_ptr(stream)))
        return out

    def handle_msg_pre_accept(self, msg, stream=None):
        out = self._out(msg["flags"].device)
        check(self._L.smr_ep_handle_pre_accept(self._h, C.byref(self._msg(msg)), C.byref(self._msg(out)),
                                               stream_ptr(stream)))
        return out

    def handle_msg_accept(self, msg, stream=None):
        out = self._out(msg["flags"].device)
        check(self._L.smr_ep_handle_accept(self._h, C.byref(self._msg(msg)), C.byref(self._msg(out)), stream_ptr(stream)))
        return out

    def handle_msg_commit_notice(self, msg, stream=None):
        check(self._L.smr_ep_handle_commit_notice(self._h, C.byref(self._msg(msg)), stream_ptr(stream)))

    def handle_msg_pre_accept_reply(self, col, ballot, seq, deps, flags, order=None, exploded=None, row=None, stream=None):
        """replies [R, G] (deps [R, R, G]) to the instance (row[g], col[g]) I lead (row None: my own row); returns
        decision / seq / deps"""
        import torch
        dev, G, R = flags.device, self.G, self.R
        r = dict(decision=torch.zeros(G, dtype=torch.uint8, device=dev), seq=torch.zeros(G, dtype=torch.int64, device=dev),
                 deps=torch.zeros((R, G), dtype=torch.int32, device=dev))
        check(self._L.smr_ep_handle_pre_accept_replies_at(self._h, _ptr(row), _ptr(col), _ptr(ballot), _ptr(seq), _ptr(deps),
                                                          _ptr(flags), _ptr(order), _ptr(exploded), _ptr(r["decision"]),
                                                          _ptr(r["seq"]), _ptr(r["deps"]), stream_ptr(stream)))
        return r

    def handle_msg_accept_reply(self, col, ballot, flags, order=None, row=None, stream=None):
        import torch
        r = dict(committed=torch.zeros(self.G, dtype=torch.uint8, device=flags.device))
        check(self._L.smr_ep_handle_accept_replies_at(self._h, _ptr(row), _ptr(col), _ptr(ballot), _ptr(flags), _ptr(order),
                                                      _ptr(r["committed"]), stream_ptr(stream)))
        return r

    # ---- explicit prepare (recovery=True) ----
    def heartbeat_timeout(self, src, exploded=None, stream=None):
        """HearTimeout { peer: src[g] } (0xFF: none) -> the ExpPrepare broadcasts: n [G], col / ballot [W, G]"""
        import torch
        dev, G, W = src.device, self.G, self.W
        o = dict(n=torch.zeros(G, dtype=torch.int32, device=dev), col=torch.zeros((W, G), dtype=torch.int32, device=dev),
                 ballot=torch.zeros((W, G), dtype=torch.int64, device=dev))
        check(self._L.smr_ep_heartbeat_timeout(self._h, _ptr(src), _ptr(exploded), _ptr(o["n"]), _ptr(o["col"]), _ptr(o["ballot"]),
                                               stream_ptr(stream)))
        return o

    def _xp_reply_bufs(self, dev, per_peer):
        import torch
        G, R = self.G, self.R
        lead = (R,) if per_peer else ()
        return dict(flags=torch.zeros(lead + (G,), dtype=torch.uint8, device=dev), voted_bal=torch.zeros(lead + (G,), dtype=torch.int64, device=dev),
                    voted_status=torch.zeros(lead + (G,), dtype=torch.uint8, device=dev),
                    voted_seq=torch.zeros(lead + (G,), dtype=torch.int64, device=dev),
                    voted_deps=torch.zeros(lead + (R, G), dtype=torch.int32, device=dev),
                    voted_key=torch.zeros(lead + (G,), dtype=torch.uint8, device=dev))

    @staticmethod
    def _xp_reply(d):
        return EpExpPrepareReply(*[_ptr(d[k]) for k in ("flags", "voted_bal", "voted_status", "voted_seq", "voted_deps", "voted_key")])

    def handle_msg_exp_prepare(self, msg, stream=None):
        """msg: flags / peer / row / col / new_ballot [G] -> the ExpPrepareReply tensors"""
        out = self._xp_reply_bufs(msg["flags"].device, False)
        m = EpExpPrepare(*[_ptr(msg[k]) for k in ("flags", "peer", "row", "col", "new_ballot")])
        check(self._L.smr_ep_handle_exp_prepare(self._h, C.byref(m), C.byref(self._xp_reply(out)), stream_ptr(stream)))
        return out

    def handle_msg_exp_prepare_reply(self, row, col, new_ballot, replies, order=None, stream=None):
        """replies: dict of [R, G] tensors (voted_deps [R, R, G]) to my ExpPrepare of (row[g], col[g]) -> decision / ballot /
        seq / deps / key of what is broadcast"""
        import torch
        dev, G, R = row.device, self.G, self.R
        r = dict(decision=torch.zeros(G, dtype=torch.uint8, device=dev), ballot=torch.zeros(G, dtype=torch.int64, device=dev),
                 seq=torch.zeros(G, dtype=torch.int64, device=dev), deps=torch.zeros((R, G), dtype=torch.int32, device=dev),
                 key=torch.zeros(G, dtype=torch.uint8, device=dev))
        check(self._L.smr_ep_handle_exp_prepare_replies(self._h, _ptr(row), _ptr(col), _ptr(new_ballot), C.byref(self._xp_reply(replies)),
                                                        _ptr(order), _ptr(r["decision"]), _ptr(r["ballot"]), _ptr(r["seq"]),
                                                        _ptr(r["deps"]), _ptr(r["key"]), stream_ptr(stream)))
        return r

    def xp_dump(self):
        G, R, W = self.G, self.R, self.W
        d = dict(acks=np.zeros((R, W, G), np.uint8), max_bal=np.zeros((R, W, G), np.uint64), avoid=np.zeros((R, W, G), np.uint8),
                 has=np.zeros((R, W, G), np.uint8), vstatus=np.zeros((R, W, R, G), np.uint8), vseq=np.zeros((R, W, R, G), np.uint64),
                 vkey=np.zeros((R, W, R, G), np.uint8), vdeps=np.zeros((R, W, R, R, G), np.uint32), counters=np.zeros(4, np.uint64))
        check(self._L.smr_ep_xp_dump(self._h, *[d[k].ctypes.data_as(C.c_void_p) for k in
                                                ("acks", "max_bal", "avoid", "has", "vstatus", "vseq", "vkey", "vdeps", "counters")]))
        return d

    def dump(self):
        G, R, W, K = self.G, self.R, self.W, self.K
        shapes = dict(len=((R, G), np.uint32), commit_bars=((R, G), np.uint32), bal=((R, W, G), np.uint64),
                      seq=((R, W, G), np.uint64), status=((R, W, G), np.uint8), key=((R, W, G), np.uint8),
                      deps=((R, W, G, R), np.uint32), pa_acks=((R, W, G), np.uint8), acc_acks=((R, W, G), np.uint8),
                      bk=((R, W, G), np.uint8), highest_cols=((K, R, G), np.uint32), counters=((3,), np.uint64))
        out, bufs = {}, EpDumpBufs()
        for n in _lib.EP_DUMP_FIELDS:
            out[n] = np.zeros(*shapes[n])
            setattr(bufs, n, out[n].ctypes.data_as(C.c_void_p))
        check(self._L.smr_ep_dump(self._h, C.byref(bufs)))
        return out

    def exec_dump(self):
        """execution state: exec_bars [R, G], kv [n_keys, G] (token of the last Put), digest [G], counters [6]"""
        G, R, K = self.G, self.R, self.K
        d = dict(exec_bars=np.zeros((R, G), np.uint32), kv=np.zeros((K, G), np.uint64), digest=np.zeros(G, np.uint64),
                 counters=np.zeros(6, np.uint64))
        check(self._L.smr_ep_exec_dump(self._h, *[d[k].ctypes.data_as(C.c_void_p)
                                                  for k in ("exec_bars", "kv", "digest", "counters")]))
        return d

    def exec_poll(self):
        """(group, row, col) of the commands the last handler call submitted: group-major, submission order per group"""
        n = C.c_uint64()
        check(self._L.smr_ep_exec_poll(self._h, None, None, None, 0, C.byref(n)))     # count only
        k = max(int(n.value), 1)
        g, r, c = np.zeros(k, np.uint32), np.zeros(k, np.uint8), np.zeros(k, np.uint32)
        check(self._L.smr_ep_exec_poll(self._h, g.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                       c.ctypes.data_as(C.c_void_p), k, C.byref(n)))
        return g[:n.value], r[:n.value], c[:n.value]
