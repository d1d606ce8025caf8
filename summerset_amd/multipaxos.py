"""Host-side handle of the batched MultiPaxos / RSPaxos cluster.

Mirrors what `MultiPaxosReplica` (src/protocols/multipaxos/mod.rs:387-514) keeps
and does, for G independent groups x R replicas stepped in lock-step (schedule
LS-1).  Thin: every method is one C-ABI call.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MpCfg, MpDumpBufs, MpGroupState, MpTickIn, SummersetError, check, stream_ptr

_DUMP_T = {"leader": np.uint8, "bal_prep_sent": np.uint64, "bal_prepared": np.uint64, "bal_max_seen": np.uint64,
           "s_bal": np.uint64, "s_status": np.uint8, "s_reqs": np.uint32, "s_vbal": np.uint64,
           "s_vreqs": np.uint32, "s_flags": np.uint8, "s_acks": np.uint8, "s_packs": np.uint8,
           "s_pmax": np.uint64, "s_ltrig": np.uint32, "s_lendp": np.uint32, "s_src": np.uint8,
           "s_rtrig": np.uint32, "s_rendp": np.uint32, "overflow": np.uint8}


# smr_mp_ack (include/summerset_hip.h): one PeerMsg::AcceptReply { slot, ballot } from `peer`
ACK_DTYPE = np.dtype([("group", "<u4"), ("slot", "<u4"), ("ballot", "<u8"), ("peer", "<u4"), ("reserved", "<u4")])


def _ptr(t):
    return None if t is None else t.data_ptr()


class MultiPaxosCluster:
    def __init__(self, n_groups, population=5, window=64, win_reserve=None, outbox_cap=None, commit_extra=0,
                 commit_list_cap=0, straggler_ticks=0):
        """straggler_ticks: ticks a group stays on the engine's side stream after a HearTimeout
        (0 = engine default, 0xFF = off); a scheduling knob only, results do not depend on it."""
        self.G, self.R, self.W = int(n_groups), int(population), int(window)
        self.win_reserve = self.W // 4 if win_reserve is None else int(win_reserve)
        self.cap = self.W + 4 if outbox_cap is None else int(outbox_cap)
        cfg = MpCfg(self.G, self.R, commit_extra, int(straggler_ticks), 0, self.W, self.win_reserve, self.cap, commit_list_cap)
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_mp_cluster_create(C.byref(cfg), C.byref(h)))
        self._h = h
        self.commit_list_cap = commit_list_cap

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_mp_cluster_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def preset_leader(self, rep=0):
        check(self._L.smr_mp_preset_leader(self._h, rep))

    def set_role_rotation(self, on=True):
        """row y of the bulk round launches runs replica (y + leader[g]) mod R of group g instead of replica y: wavefronts of
        one role, whoever leads (smr_mp_set_role_rotation; needs straggler_ticks > 0).  Same results bit for bit."""
        check(self._L.smr_mp_set_role_rotation(self._h, 1 if on else 0))

    def tick(self, timeout_rep=None, timeout_src=None, req_target=None, req_cnt=None, req_val=None, ackctl=None,
             heartbeat=False, stream=None):
        """One lock-step tick; arguments are device tensors (uint8 / uint32), see smr_mp_tick."""
        S = 0 if req_val is None else int(req_val.shape[0])
        check(self._L.smr_mp_tick(self._h, _ptr(timeout_rep), _ptr(timeout_src), _ptr(req_target), _ptr(req_cnt),
                                  _ptr(req_val), S, _ptr(ackctl), int(heartbeat), stream_ptr(stream)))

    def run_ticks(self, ticks, stream=None):
        """a batch of consecutive ticks (each a dict of `tick`'s arguments) through the fused tick kernel: one launch per
        16 ticks instead of five per tick, same results (`smr_mp_run_ticks`).  With straggler_ticks on, the bulk kernels still run
        tick by tick and the straggler list's groups go through the whole batch in one side-stream launch"""
        n = len(ticks)
        arr = (MpTickIn * max(n, 1))()
        for a, t in zip(arr, ticks):
            rv = t.get("req_val")
            a.timeout_rep_dev, a.timeout_src_dev = _ptr(t.get("timeout_rep")), _ptr(t.get("timeout_src"))
            a.req_target_dev, a.req_cnt_dev, a.req_val_dev = _ptr(t.get("req_target")), _ptr(t.get("req_cnt")), _ptr(rv)
            a.S = 0 if rv is None else int(rv.shape[0])
            a.ackctl_dev, a.do_heartbeat = _ptr(t.get("ackctl")), int(bool(t.get("heartbeat", False)))
        check(self._L.smr_mp_run_ticks(self._h, arr, n, stream_ptr(stream)))

    # the four rounds individually (multi-GPU driver / hosts with real I/O)
    def round_local(self, timeout_rep=None, timeout_src=None, req_target=None, req_cnt=None, req_val=None,
                    stream=None):
        S = 0 if req_val is None else int(req_val.shape[0])
        check(self._L.smr_mp_round_local(self._h, _ptr(timeout_rep), _ptr(timeout_src), _ptr(req_target),
                                         _ptr(req_cnt), _ptr(req_val), S, stream_ptr(stream)))

    def round_deliver(self, stream=None):
        check(self._L.smr_mp_round_deliver(self._h, stream_ptr(stream)))

    def round_replies(self, ackctl=None, publish_heartbeat=False, stream=None):
        check(self._L.smr_mp_round_replies(self._h, _ptr(ackctl), int(publish_heartbeat), stream_ptr(stream)))

    def round_heartbeat(self, stream=None):
        check(self._L.smr_mp_round_heartbeat(self._h, stream_ptr(stream)))

    def end_tick(self):
        check(self._L.smr_mp_end_tick(self._h))

    def ack_matrix_ptr(self, rep):
        p, n = C.c_void_p(), C.c_uint64()
        check(self._L.smr_mp_ack_matrix(self._h, rep, C.byref(p), C.byref(n)))
        return p.value, n.value

    # AcceptReply records <-> ack matrix (a host with real sockets; the multi-GPU exchange, spread.mp_exchange_acks)
    def collect_acks(self, rep, out, n_out, stream=None):
        """every AcceptReply that reached replica `rep` this tick (after R2) as ACK_DTYPE records into the device
        uint8 tensor `out` (capacity len(out) // 24 records); their number into the device int64 tensor `n_out`"""
        check(self._L.smr_mp_collect_acks(self._h, rep, _ptr(out), out.numel() // ACK_DTYPE.itemsize, _ptr(n_out),
                                          stream_ptr(stream)))

    def deliver_acks(self, rep, recs, n, dropped=None, stream=None):
        """the first n ACK_DTYPE records of the device uint8 tensor `recs` into replica `rep`'s ack matrix (between R2
        and R3); records answering no Accept of this tick are ignored and counted in the device int64 `dropped`"""
        check(self._L.smr_mp_deliver_acks(self._h, rep, _ptr(recs), int(n), _ptr(dropped), stream_ptr(stream)))

    def deliver_acks_conn(self, rep, ing, dropped=None, stream=None):
        """`deliver_acks` for the per-connection segments of a `wire.MpIngestConn`'s last call (smr_mp_deliver_acks_conn)"""
        off, grp, peer = ing.conn
        check(self._L.smr_mp_deliver_acks_conn(self._h, rep, _ptr(ing.acks), ing.ack_cap, _ptr(off), _ptr(grp), _ptr(peer), _ptr(ing.cnt), ing.n_conn,
                                               _ptr(dropped), stream_ptr(stream)))

    def clear_acks(self, rep, stream=None):
        check(self._L.smr_mp_clear_acks(self._h, rep, stream_ptr(stream)))

    def read_group_state(self, group, rep):
        st = MpGroupState()
        check(self._L.smr_mp_read_group_state(self._h, group, rep, C.byref(st)))
        return st

    def replica_log_view(self, rep):
        """replica `rep`'s log as the quorum-read responder reads it, in place on the device (`QuorumReadGroup.
        handle_msg_read_query(..., log=view)`): start_slot, log end, Status and batch token per slot"""
        from ._lib import QreadLog
        view = QreadLog()
        check(self._L.smr_mp_replica_log_view(self._h, rep, C.byref(view)))
        return view

    def dump(self, rep, g0=0, n=None):
        """Canonical per-replica state as host numpy arrays (same shape as the oracle's dump); with g0 / n only the
        groups [g0, g0 + n) (g0 a multiple of 64): the arrays of an oracle that runs just those groups."""
        W, R = self.W, self.R
        G = self.G - g0 if n is None else int(n)
        out, bufs = {}, MpDumpBufs()
        for name in _lib.MP_DUMP_FIELDS:
            t = _DUMP_T.get(name, np.uint32)
            shape = (R, G) if name == "peer_exec_bar" else ((W, G) if name.startswith("s_") else (G,))
            out[name] = np.zeros(shape, t)
            setattr(bufs, name, out[name].ctypes.data_as(C.c_void_p))
        check(self._L.smr_mp_dump_range(self._h, rep, int(g0), G, C.byref(bufs)))
        return out

    def counters(self, rep):
        arr = (C.c_uint64 * 3)()
        check(self._L.smr_mp_counters(self._h, rep, C.byref(arr)))
        return {"commits": int(arr[0]), "redirects": int(arr[1]), "rejects": int(arr[2])}

    def debug_generic_units(self, rep):
        n = C.c_uint64()
        check(self._L.smr_mp_debug_generic_units(self._h, rep, C.byref(n)))
        return int(n.value)

    def debug_folded_batches(self, rep):
        """client batches of `rep` that the previous tick's quorum-tally launch appended (`smr_mp_run_ticks`, round 6)"""
        n = C.c_uint64()
        check(self._L.smr_mp_debug_folded_batches(self._h, rep, C.byref(n)))
        return int(n.value)

    def straggler_stats(self):
        """(capacity of the straggler list, groups the last mark pass wanted on it): more wanted than capacity = the list
        overflowed and the rest stayed with the bulk kernels (`smr_mp_straggler_stats`)"""
        arr = (C.c_uint64 * 2)()
        check(self._L.smr_mp_straggler_stats(self._h, C.byref(arr)))
        return int(arr[0]), int(arr[1])

    def poll_commits(self, rep, cap=None):
        cap = self.commit_list_cap if cap is None else cap
        g = np.zeros(cap, np.uint32)
        s = np.zeros(cap, np.uint32)
        n = C.c_uint64()
        check(self._L.smr_mp_poll_commits(self._h, rep, g.ctypes.data_as(C.c_void_p),
                                          s.ctypes.data_as(C.c_void_p), cap, C.byref(n)))
        if n.value > cap:
            raise SummersetError(_lib.SMR_ERR_STATE, "commit list overflowed: %d > %d" % (n.value, cap))
        return g[:n.value].copy(), s[:n.value].copy()

    def profile_enable(self, on=True):
        check(self._L.smr_mp_profile_enable(self._h, int(on)))

    def profile_read(self, which):
        ms, n = C.c_double(), C.c_uint64()
        check(self._L.smr_mp_profile_read(self._h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value
