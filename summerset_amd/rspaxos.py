"""Host-side handle of the batched RSPaxos replica (G groups, one replica id).

Mirrors `RSPaxosReplica` (src/protocols/rspaxos/): every method is one handler of the reference for
all groups at once -- `req_batch` = `handle_req_batch`, `accept` = `handle_msg_accept`,
`accept_replies` = `handle_msg_accept_reply` per peer, `become_leader` = `become_a_leader` on a
HearTimeout, `prepare` / `prepare_replies` = `handle_msg_prepare` / `handle_msg_prepare_reply`,
`reconstruct` / `reconstruct_reply`, `heartbeat` / `bcast_heartbeat` = `heard_heartbeat` /
`bcast_heartbeats` -- one C-ABI call each; messages are device tensors with one entry per group,
lists are [W, G] with a count.  A request batch is a 32-bit token (0 = the empty batch, NULL = none),
a codeword is (token, mask of shards present); the shard bytes belong to `RSCodewordBatch`.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RspAccepts, RspCfg, RspDumpBufs, RspHeartbeat, RspPrepareReply, RspShards, check, stream_ptr

NULL, NO_REP = 0xFFFFFFFF, 0xFF


def _ptr(t):
    return None if t is None else t.data_ptr()


class RSPaxosReplicaGroup:
    def __init__(self, n_groups, population=5, me=0, window=32, fault_tolerance=0):
        self.G, self.R, self.me, self.W = int(n_groups), int(population), int(me), int(window)
        cfg = RspCfg(self.G, self.R, self.me, int(fault_tolerance), 0, self.W)
        h = C.c_void_p()
        self._L = _lib.load()
        check(self._L.smr_rsp_replica_create(C.byref(cfg), C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.smr_rsp_replica_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def preset_leader(self, leader):
        check(self._L.smr_rsp_preset_leader(self._h, int(leader)))

    # ---- output buffers -------------------------------------------------------------------------
    def _z(self, dev, dtype, *shape, fill=0):
        import torch
        return torch.full(shape or (self.G,), fill, dtype=dtype, device=dev)

    def _accepts(self, dev, out=None):
        import torch
        d = out if out is not None else dict(a_n=self._z(dev, torch.int32), a_slot=self._z(dev, torch.int32, self.W, self.G),
                                             a_val=self._z(dev, torch.int32, self.W, self.G), a_ballot=self._z(dev, torch.int64))
        return d, RspAccepts(_ptr(d["a_n"]), _ptr(d["a_slot"]), _ptr(d["a_val"]), _ptr(d["a_ballot"]))

    def _hb(self, dev, prefix, with_flags, out=None):
        import torch
        d = out if out is not None else {prefix + "ballot": self._z(dev, torch.int64), prefix + "commit": self._z(dev, torch.int32),
                                         prefix + "exec": self._z(dev, torch.int32), prefix + "snap": self._z(dev, torch.int32)}
        if with_flags and out is None:
            d[prefix + "flags"] = self._z(dev, torch.uint8)
        return d, RspHeartbeat(_ptr(d.get(prefix + "flags")), _ptr(d[prefix + "ballot"]), _ptr(d[prefix + "commit"]),
                               _ptr(d[prefix + "exec"]), _ptr(d[prefix + "snap"]))

    # ---- handlers -------------------------------------------------------------------------------
    # `out` (where a handler takes it): the dict an earlier call of the same handler returned -- its tensors are written again
    # instead of fresh ones being made and filled (every kernel writes every element of its per-group outputs; of a [W, G]
    # list the first a_n[g] entries): what a loop that runs the same handlers every tick, or inside a HIP graph, passes
    def req_batch(self, val, stream=None, out=None):
        d, s = self._accepts(val.device, out)
        check(self._L.smr_rsp_req_batch(self._h, _ptr(val), C.byref(s), stream_ptr(stream)))
        return d

    def accept(self, flags, peer, slot, ballot, val, mask, stream=None, out=None):
        import torch
        d = out if out is not None else dict(r_ballot=self._z(flags.device, torch.int64), r_slot=self._z(flags.device, torch.int32))
        check(self._L.smr_rsp_handle_accept(self._h, _ptr(flags), _ptr(peer), _ptr(slot), _ptr(ballot), _ptr(val), _ptr(mask),
                                            _ptr(d["r_ballot"]), _ptr(d["r_slot"]), stream_ptr(stream)))
        return d

    def accept_replies(self, slot, ballot, flags, order=None, stream=None, out=None):
        import torch
        d = out if out is not None else dict(committed=self._z(flags.device, torch.uint8))
        check(self._L.smr_rsp_handle_accept_replies(self._h, _ptr(slot), _ptr(ballot), _ptr(flags), _ptr(order),
                                                    _ptr(d["committed"]), stream_ptr(stream)))
        return d

    def become_leader(self, src, stream=None):
        import torch
        dev = src.device
        d, hb = self._hb(dev, "hb_", True)
        d.update(p_flags=self._z(dev, torch.uint8), p_trig=self._z(dev, torch.int32), p_ballot=self._z(dev, torch.int64),
                 rc_n=self._z(dev, torch.int32), rc_slot=self._z(dev, torch.int32, self.W, self.G))
        check(self._L.smr_rsp_become_leader(self._h, _ptr(src), C.byref(hb), _ptr(d["p_flags"]), _ptr(d["p_trig"]),
                                            _ptr(d["p_ballot"]), _ptr(d["rc_n"]), _ptr(d["rc_slot"]), stream_ptr(stream)))
        return d

    def _pr(self, d):
        return RspPrepareReply(*[_ptr(d[k]) for k in ("pr_n", "pr_trig", "pr_endp", "pr_ballot", "pr_vbal", "pr_vval", "pr_vmask")])

    def prepare(self, flags, peer, trig, ballot, stream=None):
        import torch
        dev, W, G = flags.device, self.W, self.G
        d = dict(pr_n=self._z(dev, torch.int32), pr_trig=self._z(dev, torch.int32), pr_endp=self._z(dev, torch.int32),
                 pr_ballot=self._z(dev, torch.int64), pr_vbal=self._z(dev, torch.int64, W, G),
                 pr_vval=self._z(dev, torch.int32, W, G, fill=-1), pr_vmask=self._z(dev, torch.uint8, W, G))
        check(self._L.smr_rsp_handle_prepare(self._h, _ptr(flags), _ptr(peer), _ptr(trig), _ptr(ballot), C.byref(self._pr(d)),
                                             stream_ptr(stream)))
        return d

    def prepare_replies(self, peer, pr_n, pr_trig, pr_endp, pr_ballot, pr_vbal, pr_vval, pr_vmask, stream=None):
        d, s = self._accepts(peer.device)
        src = dict(pr_n=pr_n, pr_trig=pr_trig, pr_endp=pr_endp, pr_ballot=pr_ballot, pr_vbal=pr_vbal, pr_vval=pr_vval, pr_vmask=pr_vmask)
        check(self._L.smr_rsp_handle_prepare_replies(self._h, _ptr(peer), C.byref(self._pr(src)), C.byref(s), stream_ptr(stream)))
        return d

    def reconstruct(self, flags, rc_n, rc_slot, stream=None):
        import torch
        dev, W, G = flags.device, self.W, self.G
        d = dict(rr_n=self._z(dev, torch.int32), rr_slot=self._z(dev, torch.int32, W, G), rr_bal=self._z(dev, torch.int64, W, G),
                 rr_val=self._z(dev, torch.int32, W, G, fill=-1), rr_mask=self._z(dev, torch.uint8, W, G))
        s = RspShards(*[_ptr(d[k]) for k in ("rr_n", "rr_slot", "rr_bal", "rr_val", "rr_mask")])
        check(self._L.smr_rsp_handle_reconstruct(self._h, _ptr(flags), _ptr(rc_n), _ptr(rc_slot), C.byref(s), stream_ptr(stream)))
        return d

    def reconstruct_reply(self, flags, rr_n, rr_slot, rr_bal, rr_val, rr_mask, stream=None):
        s = RspShards(_ptr(rr_n), _ptr(rr_slot), _ptr(rr_bal), _ptr(rr_val), _ptr(rr_mask))
        check(self._L.smr_rsp_handle_reconstruct_reply(self._h, _ptr(flags), C.byref(s), stream_ptr(stream)))

    def heartbeat(self, flags, peer, ballot, commit_bar, exec_bar, snap_bar, stream=None, out=None):
        import torch
        dev = flags.device
        if out is not None:
            o = dict(ballot=out["ballot"], commit=out["commit_bar"], exec=out["exec_bar"], snap=out["snap_bar"])
            reply = out["reply"]
            hb = RspHeartbeat(None, _ptr(o["ballot"]), _ptr(o["commit"]), _ptr(o["exec"]), _ptr(o["snap"]))
        else:
            o, hb = self._hb(dev, "", False)
            reply = self._z(dev, torch.uint8)
        inp = RspHeartbeat(_ptr(flags), _ptr(ballot), _ptr(commit_bar), _ptr(exec_bar), _ptr(snap_bar))
        check(self._L.smr_rsp_handle_heartbeat(self._h, _ptr(peer), C.byref(inp), _ptr(reply), C.byref(hb), stream_ptr(stream)))
        return dict(reply=reply, ballot=o["ballot"], commit_bar=o["commit"], exec_bar=o["exec"], snap_bar=o["snap"])

    def bcast_heartbeat(self, flags, stream=None, out=None):
        if out is not None:
            o = dict(ballot=out["ballot"], commit=out["commit_bar"], exec=out["exec_bar"], snap=out["snap_bar"])
            hb = RspHeartbeat(None, _ptr(o["ballot"]), _ptr(o["commit"]), _ptr(o["exec"]), _ptr(o["snap"]))
        else:
            o, hb = self._hb(flags.device, "", False)
        check(self._L.smr_rsp_bcast_heartbeat(self._h, _ptr(flags), C.byref(hb), stream_ptr(stream)))
        return dict(ballot=o["ballot"], commit_bar=o["commit"], exec_bar=o["exec"], snap_bar=o["snap"])

    def dump(self):
        G, R, W = self.G, self.R, self.W
        u8, u32, u64 = np.uint8, np.uint32, np.uint64
        types = dict(leader=u8, bal_prep_sent=u64, bal_prepared=u64, bal_max_seen=u64, len=u32, commit_bar=u32, exec_bar=u32,
                     snap_bar=u32, digest=u64, s_bal=u64, s_status=u8, s_val=u32, s_mask=u8, s_vbal=u64, s_vval=u32, s_vmask=u8,
                     s_flags=u8, s_ltrig=u32, s_lendp=u32, s_packs=u8, s_aacks=u8, s_pmax=u64, s_rsrc=u8, s_rtrig=u32, s_rendp=u32)
        out, bufs = {}, RspDumpBufs()
        for n in _lib.RSP_DUMP_FIELDS:
            if n == "peer_exec_bar":
                out[n] = np.zeros((R, G), u32)
            elif n == "counters":
                out[n] = np.zeros(4, u64)
            else:
                out[n] = np.zeros((W, G) if n.startswith("s_") else G, types[n])
            setattr(bufs, n, out[n].ctypes.data_as(C.c_void_p))
        check(self._L.smr_rsp_dump(self._h, C.byref(bufs)))
        return out

    def exec_poll(self):
        """(group, slot, token) of the commands the last handler call executed: group-major, execution order per group"""
        n = C.c_uint64()
        check(self._L.smr_rsp_exec_poll(self._h, None, None, None, 0, C.byref(n)))
        k = max(int(n.value), 1)
        g, s, v = np.zeros(k, np.uint32), np.zeros(k, np.uint32), np.zeros(k, np.uint32)
        check(self._L.smr_rsp_exec_poll(self._h, g.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p),
                                        v.ctypes.data_as(C.c_void_p), k, C.byref(n)))
        return g[:n.value], s[:n.value], v[:n.value]
