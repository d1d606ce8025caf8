"""Host-side handle of the batched Raft leader (G groups, one replica id).

Mirrors the leader half of `RaftReplica` (src/protocols/raft/mod.rs:237-330):
`handle_req_batch` log append and `handle_msg_append_entries_reply`
(raft/messages.rs:222-388).  Thin: every method is one C-ABI call.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RaftCfg, RaftDumpBufs, check

_T = {"role": np.uint8, "leader": np.uint8, "curr_term": np.uint64, "entry_term": np.uint64}


def _ptr(t):
    return None if t is None else t.data_ptr()


class RaftLeaderGroup:
    def __init__(self, n_groups, population=5, leader_id=0, window=64, term=1, commit_extra=0):
        self.G, self.R, self.W, self.me = int(n_groups), int(population), int(window), int(leader_id)
        cfg = RaftCfg(self.G, self.R, self.me, commit_extra, 0, self.W, term)
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_raft_leader_create(C.byref(cfg), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_raft_leader_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @staticmethod
    def _stream(stream):
        if stream is None:
            import torch
            return torch.cuda.current_stream().cuda_stream
        return int(stream)

    def handle_req_batch(self, n_new, stream=None):
        """append n_new[g] entries of the current term to every group's log"""
        check(self._L.smr_raft_leader_append(self._h, _ptr(n_new), self._stream(stream)))

    def handle_msg_append_entries_reply(self, reply_term, end_slot, flags, conflict_term=None, conflict_slot=None,
                                        order=None, stream=None):
        """one AppendEntriesReply per (peer, group); device tensors shaped [R, G]"""
        check(self._L.smr_raft_leader_handle_replies(self._h, _ptr(reply_term), _ptr(end_slot), _ptr(conflict_term),
                                                     _ptr(conflict_slot), _ptr(flags), _ptr(order),
                                                     self._stream(stream)))

    def dump(self):
        G, W, R = self.G, self.W, self.R
        out, bufs = {}, RaftDumpBufs()
        for name in _lib.RAFT_DUMP_FIELDS:
            shape = (R, G) if name in ("next_slot", "try_next_slot", "match_slot") else \
                ((W, G) if name == "entry_term" else (G,))
            out[name] = np.zeros(shape, _T.get(name, np.uint32))
            setattr(bufs, name, out[name].ctypes.data_as(C.c_void_p))
        check(self._L.smr_raft_leader_dump(self._h, C.byref(bufs)))
        return out

    def total_commits(self):
        n = C.c_uint64()
        check(self._L.smr_raft_leader_total_commits(self._h, C.byref(n)))
        return int(n.value)
