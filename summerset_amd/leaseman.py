"""Host-side handle of the batched `LeaseManager` (src/server/leaseman.rs): G groups, one replica id.  Thin: every method
is one C-ABI call on device tensors with one entry per group (include/summerset_hip.h, `smr_lease_*`).  The clock is an
explicit argument.

Notices and actions travel as three u64 per group (num, meta, bar); `pack_notice` / `unpack_actions` are the bit layouts
of include/summerset_hip.h for hosts that build them from scalars."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import LeaseCfg, check, stream_ptr

ALL = 0xFF
ACT_CAP = 20
N_NONE, N_NEW_GRANTS, N_DO_REVOKE, N_CLEAR_HELD, N_RECV_MSG = range(5)
GUARD, GUARD_REPLY, PROMISE, PROMISE_REPLY, REVOKE, REVOKE_REPLY = range(6)
A_SEND, A_BCAST, A_NEXT_REFRESH, A_GRANT_REMOVED, A_LEASE_CLEARED, A_GRANT_TIMEOUT, A_LEASE_TIMEOUT, A_HIGHER_NUMBER, \
    A_GUARD_ACCEPT_BAR = range(1, 10)
PH_GUARD_SENT, PH_GUARD_HELD, PH_PROMISE_SENT, PH_PROMISE_HELD = 1, 2, 4, 8


def _ptr(t):
    return None if t is None else t.data_ptr()


def pack_notice(kind, peer=0, peers=0, msg=0, held=0, has_bar=0):
    """meta word(s) of a notice; works on ints, numpy arrays and torch tensors (int64) alike"""
    return kind | (peer << 8) | (peers << 16) | (msg << 24) | (held << 32) | (has_bar << 40)


def unpack_actions(meta):
    """-> dict of kind / peer / mask / msg / flag from the action meta words"""
    return dict(kind=meta & 0xFF, peer=(meta >> 8) & 0xFF, mask=(meta >> 16) & 0xFF, msg=(meta >> 24) & 0xFF, flag=(meta >> 32) & 0xFF)


class LeaseManager:
    def __init__(self, n_groups, population=5, replica_id=0, expire_timeout_ms=2000, hb_send_interval_ms=20):
        self.G, self.R, self.me = int(n_groups), int(population), int(replica_id)
        cfg = LeaseCfg(self.G, self.R, self.me, int(expire_timeout_ms), int(hb_send_interval_ms))
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_lease_create(C.byref(cfg), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_lease_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _dev(self):
        import torch
        return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")

    def step(self, now_ms, num=None, meta=None, bar=None, out=None, stream=None):
        """num / meta / bar: int64 device tensors [G] (None: timers only) -> (act_n [G] uint8, act_num, act_meta, act_bar
        [ACT_CAP, G] int64); `out` = a tuple of those four to reuse"""
        import torch
        if out is None:
            dev = self._dev()
            out = (torch.zeros(self.G, dtype=torch.uint8, device=dev),) + tuple(
                torch.zeros((ACT_CAP, self.G), dtype=torch.int64, device=dev) for _ in range(3))
        n, anum, ameta, abar = out
        check(self._L.smr_lease_step(self._h, int(now_ms), _ptr(num), _ptr(meta), _ptr(bar), _ptr(n), _ptr(anum), _ptr(ameta),
                                     _ptr(abar), stream_ptr(stream)))
        return out

    def attempt_refresh(self, now_ms, call, peers, stream=None):
        import torch
        o = torch.zeros(self.G, dtype=torch.uint8, device=call.device)
        check(self._L.smr_lease_attempt_refresh(self._h, int(now_ms), _ptr(call), _ptr(peers), _ptr(o), stream_ptr(stream)))
        return o

    def sets(self, stream=None):
        """-> (grant_set, lease_set, lease_cnt) uint8 device tensors [G]"""
        import torch
        dev = self._dev()
        g, l, c = (torch.zeros(self.G, dtype=torch.uint8, device=dev) for _ in range(3))
        check(self._L.smr_lease_sets(self._h, _ptr(g), _ptr(l), _ptr(c), stream_ptr(stream)))
        return g, l, c

    def dump_raw(self):
        R, G = self.R, self.G
        d = dict(active_num=np.zeros(G, np.uint64), phase=np.zeros((R, G), np.uint8), grant_deadline=np.zeros((R, G), np.uint64),
                 hold_deadline=np.zeros((R, G), np.uint64), refresh_mark=np.zeros(G, np.uint8))
        check(self._L.smr_lease_dump(self._h, *[d[k].ctypes.data_as(C.c_void_p) for k in
                                                ("active_num", "phase", "grant_deadline", "hold_deadline", "refresh_mark")]))
        return d

    def dump(self):
        """the state in the oracle's terms (oracle/lease_oracle.c `orc_lease_dump`)"""
        r = self.dump_raw()
        ph = r["phase"]
        bit = lambda b: ((ph & b) != 0).astype(np.uint8)
        w = (1 << np.arange(self.R, dtype=np.uint32))[:, None]
        fold = lambda m: (m.astype(np.uint32) * w).sum(axis=0).astype(np.uint8)
        ls = fold(bit(PH_PROMISE_HELD))
        return dict(active_num=r["active_num"], grant_set=fold(bit(PH_PROMISE_SENT)), lease_set=ls,
                    lease_cnt=(1 + np.array([bin(int(x)).count("1") for x in ls])).astype(np.uint8),
                    guards_sent=fold(bit(PH_GUARD_SENT)), guards_held=fold(bit(PH_GUARD_HELD)), refresh_mark=r["refresh_mark"],
                    ps_deadline=np.where(bit(PH_PROMISE_SENT) != 0, r["grant_deadline"], 0).astype(np.uint64),
                    gh_deadline=np.where(bit(PH_GUARD_HELD) != 0, r["hold_deadline"], 0).astype(np.uint64),
                    ph_deadline=np.where(bit(PH_PROMISE_HELD) != 0, r["hold_deadline"], 0).astype(np.uint64))
