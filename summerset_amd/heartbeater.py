"""Host-side handle of the batched `Heartbeater` (src/server/heartbeat.rs): G groups, one replica id.  Thin: every method
is one C-ABI call on device tensors with one entry per group (include/summerset_hip.h, `smr_hb_*`).  Clocks and random
draws are explicit arguments."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import HbCfg, check, stream_ptr

ALL, NONE = 0xFE, 0xFF


def _ptr(t):
    return None if t is None else t.data_ptr()


class Heartbeater:
    def __init__(self, n_groups, population=5, replica_id=0, hear_timeout_min_ms=1200, hear_timeout_max_ms=2000,
                 send_interval_ms=20, now_ms=0):
        """defaults: multipaxos/mod.rs:123-149 (`hb_hear_timeout_{min,max}` 1200 / 2000, `hb_send_interval_ms` 20)"""
        self.G, self.R, self.me = int(n_groups), int(population), int(replica_id)
        cfg = HbCfg(self.G, self.R, self.me, int(hear_timeout_min_ms), int(hear_timeout_max_ms), int(send_interval_ms))
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_hb_create(C.byref(cfg), int(now_ms), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_hb_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_sending(self, sending, stream=None):
        check(self._L.smr_hb_set_sending(self._h, _ptr(sending), stream_ptr(stream)))

    def kickoff_hear_timer(self, peer, now_ms, draw, stream=None):
        """peer[G] uint8 (a peer id, ALL, or NONE = no call); draw[R, G] int32 / uint32: the random draw per timer"""
        check(self._L.smr_hb_kickoff_hear_timer(self._h, _ptr(peer), int(now_ms), _ptr(draw), stream_ptr(stream)))

    def poll(self, now_ms, stream=None):
        """-> (timeouts [R, G], send_ticked [G]) uint8 device tensors: the HeartbeatEvents delivered at now_ms"""
        import torch
        dev = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
        t = torch.zeros((self.R, self.G), dtype=torch.uint8, device=dev)
        s = torch.zeros(self.G, dtype=torch.uint8, device=dev)
        check(self._L.smr_hb_poll(self._h, int(now_ms), _ptr(t), _ptr(s), stream_ptr(stream)))
        return t, s

    def clear_reply_cnts(self, peer, stream=None):
        check(self._L.smr_hb_clear_reply_cnts(self._h, _ptr(peer), stream_ptr(stream)))

    def update_bcast_cnts(self, flags, stream=None):
        import torch
        d = torch.zeros(self.G, dtype=torch.uint8, device=flags.device)
        check(self._L.smr_hb_update_bcast_cnts(self._h, _ptr(flags), _ptr(d), stream_ptr(stream)))
        return d

    def update_heard_cnt(self, peer, stream=None):
        check(self._L.smr_hb_update_heard_cnt(self._h, _ptr(peer), stream_ptr(stream)))

    def dump(self):
        R, G = self.R, self.G
        d = dict(deadline=np.zeros((R, G), np.uint64), exploded=np.zeros((R, G), np.uint8), is_sending=np.zeros(G, np.uint8),
                 next_tick=np.zeros(G, np.uint64), cnt0=np.zeros((R, G), np.uint64), cnt1=np.zeros((R, G), np.uint64),
                 rep=np.zeros((R, G), np.uint8), alive=np.zeros(G, np.uint8))
        check(self._L.smr_hb_dump(self._h, *[d[k].ctypes.data_as(C.c_void_p) for k in
                                             ("deadline", "exploded", "is_sending", "next_tick", "cnt0", "cnt1", "rep", "alive")]))
        return d


def hear_timeouts_for_engine(timeouts, me):
    """`Heartbeater.poll`'s timeouts[R, G] of replica `me` as the (timeout_rep, timeout_src) arrays the MultiPaxos cluster
    engine takes (`smr_mp_round_local`: HeartbeatEvent::HearTimeout on replica timeout_rep about peer timeout_src), on the
    device: the lowest peer id whose timer fired (one event per group and tick), SMR_NO_REPLICA where none did"""
    import torch
    R = timeouts.shape[0]
    ids = torch.arange(R, dtype=torch.uint8, device=timeouts.device)[:, None].expand_as(timeouts)
    src = torch.where(timeouts != 0, ids, torch.full_like(ids, NONE)).amin(dim=0)
    rep = torch.where(src != NONE, torch.full_like(src, int(me)), torch.full_like(src, NONE))
    return rep, src
