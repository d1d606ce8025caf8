"""Host-side handle of the batched near-quorum-read path of MultiPaxos (G groups, one replica id).

Mirrors `MultiPaxosReplica`'s quorum-read methods (src/protocols/multipaxos/quorumread.rs): `refresh_highest_slot`
(:8-26), `inspect_highest_slot` (:30-73), `handle_msg_read_query` (:75-188), `handle_msg_read_query_reply` (:190-346)
and the `ReadQueryBookkeeping` set-up of `treat_read_only_reqs` (request.rs:55-101).  Thin: every method is one C-ABI
call on device tensors with one entry per group (include/summerset_hip.h, `smr_qread_*`, for the data model)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import QreadCfg, QreadLog, QreadReplies, check, stream_ptr

NONE, SLOT, VALUE = 0, 1, 2                       # Option<(slot, Option<value>)>
PENDING, NOT_FOUND, RETRY, GOT_VALUE = 0, 1, 2, 3


def _ptr(t):
    return None if t is None else (t if isinstance(t, int) else t.data_ptr())


class QuorumReadGroup:
    def __init__(self, n_groups, population=5, replica_id=0, n_keys=16, max_reads=4, n_queries=2):
        self.G, self.R, self.me = int(n_groups), int(population), int(replica_id)
        self.K, self.B, self.Q = int(n_keys), int(max_reads), int(n_queries)
        cfg = QreadCfg(self.G, self.R, self.me, self.K, self.B, self.Q)
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_qread_create(C.byref(cfg), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_qread_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _replies(self, lead, device):
        import torch
        shape = tuple(lead) + (self.B, self.G)
        return dict(state=torch.zeros(shape, dtype=torch.uint8, device=device), slot=torch.zeros(shape, dtype=torch.int32, device=device),
                    val=torch.zeros(shape, dtype=torch.int32, device=device))

    @staticmethod
    def _rs(d):
        return QreadReplies(_ptr(d["state"]), _ptr(d["slot"]), _ptr(d["val"]))

    def refresh_highest_slot(self, slot, put_keys, stream=None):
        """slot[g] (int32, -1 = no batch) gets a batch whose Puts write put_keys[B, g] (uint8, 0xFF = not a Put)"""
        check(self._L.smr_qread_refresh_highest_slot(self._h, _ptr(slot), _ptr(put_keys), stream_ptr(stream)))

    def handle_msg_read_query(self, keys, n, log, stable_leader=None, kv=None, stream=None):
        """log = dict(start_slot, log_end [G] int32; status uint8, token int32 [W, G]), or the `QreadLog` view of a replica
        of the MultiPaxos cluster engine (`MultiPaxosCluster.replica_log_view`).  Returns (replies, from_leader)"""
        import torch
        out = self._replies((), keys.device)
        fl = torch.zeros(self.G, dtype=torch.uint8, device=keys.device)
        lg = log if isinstance(log, QreadLog) else QreadLog(_ptr(log["start_slot"]), _ptr(log["log_end"]), _ptr(log["status"]),
                                                            _ptr(log["token"]), int(log["status"].shape[0]), 0, None, None)
        rs = self._rs(out)
        check(self._L.smr_qread_handle_read_query(self._h, _ptr(keys), _ptr(n), _ptr(stable_leader), _ptr(kv), C.byref(lg),
                                                  C.byref(rs), _ptr(fl), stream_ptr(stream)))
        return out, fl

    def inspect_highest_slot(self, keys, n, log, stream=None):
        return self.handle_msg_read_query(keys, n, log, stream=stream)[0]

    def issue(self, q, n, own, stream=None):
        rs = self._rs(own)
        check(self._L.smr_qread_issue(self._h, int(q), _ptr(n), C.byref(rs), stream_ptr(stream)))

    def handle_msg_read_query_reply(self, q, replies, flags, order=None, stream=None):
        """replies: dict of [R, B, G] tensors; flags [R, G] (bit0 present, bit1 from_leader).  Returns outcome, out_val [B, G], done [G]"""
        import torch
        dev = flags.device
        outcome = torch.zeros((self.B, self.G), dtype=torch.uint8, device=dev)
        out_val = torch.zeros((self.B, self.G), dtype=torch.int32, device=dev)
        done = torch.zeros(self.G, dtype=torch.uint8, device=dev)
        rs = self._rs(replies)
        check(self._L.smr_qread_handle_replies(self._h, int(q), C.byref(rs), _ptr(flags), _ptr(order), _ptr(outcome), _ptr(out_val),
                                               _ptr(done), stream_ptr(stream)))
        return outcome, out_val, done

    def dump(self):
        G, K, B, Q = self.G, self.K, self.B, self.Q
        out = dict(highest_slot=np.zeros((K, G), np.uint32), live=np.zeros((Q, G), np.uint8), n=np.zeros((Q, G), np.uint8),
                   rq_acks=np.zeros((Q, G), np.uint8), mx_state=np.zeros((Q, B, G), np.uint8), mx_slot=np.zeros((Q, B, G), np.uint32),
                   mx_val=np.zeros((Q, B, G), np.uint32), counters=np.zeros(4, np.uint64))
        check(self._L.smr_qread_dump(self._h, *[out[k].ctypes.data_as(C.c_void_p) for k in
                                                ("highest_slot", "live", "n", "rq_acks", "mx_state", "mx_slot", "mx_val", "counters")]))
        return out


GET, PUT = 0, 1


class KvStateMachine:
    """`StateMachineExecutorTask::execute` (src/server/statemach.rs:193-202) for G groups, state resident on the device:
    commands [rows, G] are applied row after row; a Get returns the value, a Put the old value (0 = None)."""

    def __init__(self, n_groups, n_keys=16):
        self.G, self.K = int(n_groups), int(n_keys)
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_kv_create(self.G, self.K, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_kv_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def execute(self, kind, key, val, stream=None):
        import torch
        res = torch.zeros(kind.shape, dtype=torch.int32, device=kind.device)
        check(self._L.smr_kv_execute(self._h, int(kind.shape[0]), _ptr(kind), _ptr(key), _ptr(val), _ptr(res), stream_ptr(stream)))
        return res

    def table_ptr(self):
        p = C.c_void_p()
        check(self._L.smr_kv_table(self._h, C.byref(p)))
        return p.value

    def dump(self):
        out = np.zeros((self.K, self.G), np.uint32)
        check(self._L.smr_kv_dump(self._h, out.ctypes.data_as(C.c_void_p)))
        return out


class StringKvStateMachine:
    """`StateMachineExecutorTask::execute` on `HashMap<String, String>` (src/server/statemach.rs:21-63,193-202) for G groups, keys
    and values as bytes, state resident on the device (`smr_skv_*`): a hash table + an append-only heap per group"""

    def __init__(self, n_groups, slots=256, heap_bytes=1 << 16):
        self.G, self.slots, self.heap_bytes = int(n_groups), int(slots), int(heap_bytes)
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_skv_create(self.G, self.slots, self.heap_bytes, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_skv_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def execute(self, kind, payload, key_off, key_len, val_off, val_len, stream=None):
        """kind uint8 [rows, G] (GET / PUT / else none); payload uint8 [n]; offsets / lengths int32 [rows, G].
        -> (state uint8, off int32, len int32) [rows, G]: None / Some(bytes of the group's heap strip) / refused"""
        import torch
        dev = kind.device
        st = torch.zeros(kind.shape, dtype=torch.uint8, device=dev)
        off = torch.zeros(kind.shape, dtype=torch.int32, device=dev)
        ln = torch.zeros(kind.shape, dtype=torch.int32, device=dev)
        check(self._L.smr_skv_execute(self._h, int(kind.shape[0]), _ptr(kind), _ptr(payload), int(payload.numel()), _ptr(key_off), _ptr(key_len),
                                      _ptr(val_off), _ptr(val_len), _ptr(st), _ptr(off), _ptr(ln), stream_ptr(stream)))
        return st, off, ln

    def heap_ptr(self):
        p, n = C.c_void_p(), C.c_uint64()
        check(self._L.smr_skv_heap(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def heap_bytes_of(self, group, off, length):
        """host copy of bytes [off, off + length) of `group`'s heap strip (tests; a device consumer reads them in place)"""
        buf = (C.c_uint8 * max(int(length), 1))()
        check(self._L.smr_skv_read(self._h, int(group), int(off), int(length), buf))
        return bytes(buf[:int(length)])

    def stats(self):
        nk, hu, fl = np.zeros(self.G, np.uint32), np.zeros(self.G, np.uint32), np.zeros(self.G, np.uint8)
        check(self._L.smr_skv_stats(self._h, nk.ctypes.data_as(C.c_void_p), hu.ctypes.data_as(C.c_void_p), fl.ctypes.data_as(C.c_void_p)))
        return dict(n_keys=nk, heap_used=hu, full=fl)
