"""Host-side handle of the batched EPaxos command leader / acceptor (G groups, one replica id).

Mirrors `EPaxosReplica` (src/protocols/epaxos/mod.rs) on the pre-execution path:
`handle_req_batch`, `handle_msg_pre_accept`, `handle_msg_pre_accept_reply` (the
fast-quorum decision), `handle_msg_accept`, `handle_msg_accept_reply`,
`handle_msg_commit_notice`; with `execute=True` every call is followed by the
dependency-graph execution the reference runs from `handle_logged_commit_slot`
(`attempt_execution`, `handle_cmd_result`; state read back by `exec_dump`).  Thin: every
method is one C-ABI call; messages are device tensors with one entry per group,
DepSets are int32 tensors [R, G] with -1 (0xFFFFFFFF) = None.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import EpCfg, EpDumpBufs, EpMsg, check, stream_ptr

NONE, NO_KEY = 0xFFFFFFFF, 0xFF


def _ptr(t):
    return None if t is None else t.data_ptr()


class EPaxosReplicaGroup:
    def __init__(self, n_groups, population=5, me=0, window=32, n_keys=64, optimized_quorum=True, execute=False):
        self.G, self.R, self.me, self.W, self.K = int(n_groups), int(population), int(me), int(window), int(n_keys)
        self.execute = bool(execute)
        cfg = EpCfg(self.G, self.R, self.me, int(optimized_quorum), int(self.execute), self.W, self.K)
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
        return EpMsg(*[_ptr(d.get(k)) for k in ("flags", "peer", "col", "ballot", "seq", "deps", "key")])

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

    def handle_msg_pre_accept_reply(self, col, ballot, seq, deps, flags, order=None, exploded=None, stream=None):
        """replies [R, G] (deps [R, R, G]) to my instance (me, col[g]); returns decision / seq / deps"""
        import torch
        dev, G, R = flags.device, self.G, self.R
        r = dict(decision=torch.zeros(G, dtype=torch.uint8, device=dev), seq=torch.zeros(G, dtype=torch.int64, device=dev),
                 deps=torch.zeros((R, G), dtype=torch.int32, device=dev))
        check(self._L.smr_ep_handle_pre_accept_replies(self._h, _ptr(col), _ptr(ballot), _ptr(seq), _ptr(deps), _ptr(flags),
                                                       _ptr(order), _ptr(exploded), _ptr(r["decision"]), _ptr(r["seq"]),
                                                       _ptr(r["deps"]), stream_ptr(stream)))
        return r

    def handle_msg_accept_reply(self, col, ballot, flags, order=None, stream=None):
        import torch
        r = dict(committed=torch.zeros(self.G, dtype=torch.uint8, device=flags.device))
        check(self._L.smr_ep_handle_accept_replies(self._h, _ptr(col), _ptr(ballot), _ptr(flags), _ptr(order),
                                                   _ptr(r["committed"]), stream_ptr(stream)))
        return r

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
