"""ctypes front-end of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the summerset_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_SRCS = ["rs_oracle.c", "mp_oracle.c", "raft_oracle.c", "ep_oracle.c", "rsp_oracle.c", "qr_oracle.c", "bitmap_oracle.c", "hb_oracle.c", "lease_oracle.c"]

CTL_IDENTITY = 0x00FAC688
NO_LEADER = 0xFF


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    srcs = [os.path.join(_HERE, s) for s in _SRCS if os.path.exists(os.path.join(_HERE, s))]
    if not force and os.path.exists(_LIB_PATH):
        if all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs):
            return _LIB_PATH
    cmd = ["gcc", "-O2", "-fPIC", "-std=c11", "-shared", "-o", _LIB_PATH] + srcs
    subprocess.check_call(cmd)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _declare(L):
    u8, u32, u64, i32, vp = C.c_uint8, C.c_uint32, C.c_uint64, C.c_int, C.c_void_p
    L.orc_gf_mul.restype = u8; L.orc_gf_mul.argtypes = [u8, u8]
    L.orc_gf_div.restype = u8; L.orc_gf_div.argtypes = [u8, u8]
    L.orc_gf_exp.restype = u8; L.orc_gf_exp.argtypes = [u8, i32]
    L.orc_gf_tables.argtypes = [vp, vp]
    L.orc_rs_matrix.restype = i32; L.orc_rs_matrix.argtypes = [i32, i32, vp]
    L.orc_rs_shard_len.restype = u64; L.orc_rs_shard_len.argtypes = [u64, i32]
    L.orc_rs_encode.restype = i32; L.orc_rs_encode.argtypes = [i32, i32, vp, u64, vp]
    L.orc_rs_encode_batch.restype = i32
    L.orc_rs_encode_batch.argtypes = [i32, i32, vp, u64, u64, u64, vp, u64]
    L.orc_rs_reconstruct.restype = i32; L.orc_rs_reconstruct.argtypes = [i32, i32, vp, u64, vp, i32]
    L.orc_rs_verify.restype = i32; L.orc_rs_verify.argtypes = [i32, i32, vp, u64]
    L.orc_bincode_string.restype = u64; L.orc_bincode_string.argtypes = [vp, u64, vp]
    L.orc_bincode_reqbatch_put.restype = u64
    L.orc_bincode_reqbatch_put.argtypes = [u64, u64, vp, u64, vp, u64, vp]
    L.orc_mp_new.restype = vp; L.orc_mp_new.argtypes = [u32, u8, u32, u32, u32, u8, i32]
    L.orc_mp_free.argtypes = [vp]
    L.orc_mp_preset_leader.argtypes = [vp, u8]
    L.orc_mp_tick.argtypes = [vp, vp, vp, vp, vp, vp, u32, vp, i32]
    L.orc_mp_dump.argtypes = [vp, u8] + [vp] * 26
    L.orc_mp_total_commits.restype = u64; L.orc_mp_total_commits.argtypes = [vp, u8]
    L.orc_mp_take_commits.restype = u64; L.orc_mp_take_commits.argtypes = [vp, u8, vp, vp, u64]
    L.orc_raft_new.restype = vp; L.orc_raft_new.argtypes = [u32, u8, u32, u8, u64, u8]
    L.orc_raft_free.argtypes = [vp]
    L.orc_raft_leader_append.argtypes = [vp, vp]
    L.orc_raft_leader_append_emit.argtypes = [vp, vp, vp]
    L.orc_raft_gather_entries.argtypes = [vp, vp, u32] + [vp] * 9
    L.orc_raft_handle_replies.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.orc_raft_dump.argtypes = [vp] + [vp] * 11
    L.orc_raft_total_commits.restype = u64; L.orc_raft_total_commits.argtypes = [vp]
    L.orc_raft_counters.argtypes = [vp, vp]
    L.orc_raft_ring_guard_hits.argtypes = [vp]
    L.orc_raft_ring_guard_hits.restype = C.c_uint64
    L.orc_raft_preset.argtypes = [vp, u8, u8, u64, u8]
    L.orc_raft_handle_append_entries.argtypes = [vp] + [vp] * 7 + [u32] + [vp] * 7
    L.orc_raft_become_candidate.argtypes = [vp] + [vp] * 5
    L.orc_raft_handle_request_vote.argtypes = [vp] + [vp] * 7
    L.orc_raft_handle_vote_replies.argtypes = [vp] + [vp] * 6
    L.orc_raft_dump_votes.argtypes = [vp] + [vp] * 4
    L.orc_qr_new.restype = vp; L.orc_qr_new.argtypes = [u32, u8, u8, u32, u32, u32]
    L.orc_qr_free.argtypes = [vp]
    L.orc_qr_refresh_highest_slot.argtypes = [vp, vp, vp]
    L.orc_qr_handle_read_query.argtypes = [vp] + [vp] * 8 + [u32] + [vp] * 4
    L.orc_qr_issue.argtypes = [vp, u32] + [vp] * 4
    L.orc_qr_handle_replies.argtypes = [vp, u32] + [vp] * 8
    L.orc_qr_dump.argtypes = [vp] + [vp] * 8
    L.orc_craft_enable.argtypes = [vp, u8, u8]
    L.orc_craft_switch_assignment_mode.argtypes = [vp, vp]
    L.orc_craft_bcast_heartbeats.argtypes = [vp] + [vp] * 5
    L.orc_craft_assignment.argtypes = [vp, vp, vp]
    L.orc_craft_dump.argtypes = [vp] + [vp] * 5
    L.orc_craft_handle_append_entries.argtypes = [vp] + [vp] * 8 + [u32] + [vp] * 7
    L.orc_craft_handle_reconstruct.argtypes = [vp, vp, vp, vp, u32, vp, vp, vp]
    L.orc_craft_dump_masks.argtypes = [vp, vp, vp]
    L.orc_craft_take_reconstructs.argtypes = [vp, u32, vp, vp, vp]
    L.orc_craft_handle_reconstruct_reply.argtypes = [vp, vp, vp, vp, vp, u32]
    L.orc_ep_new.restype = vp; L.orc_ep_new.argtypes = [u32, u8, u8, u32, u32, u8]
    L.orc_ep_free.argtypes = [vp]
    L.orc_ep_propose.argtypes = [vp] + [vp] * 6
    L.orc_ep_handle_pre_accept.argtypes = [vp] + [vp] * 12
    L.orc_ep_handle_pre_accept_replies.argtypes = [vp] + [vp] * 11
    L.orc_ep_handle_accept.argtypes = [vp] + [vp] * 10
    L.orc_ep_handle_accept_replies.argtypes = [vp] + [vp] * 6
    L.orc_ep_handle_commit_notice.argtypes = [vp] + [vp] * 8
    L.orc_ep_heartbeat_timeout.argtypes = [vp] + [vp] * 5
    L.orc_ep_handle_exp_prepare.argtypes = [vp] + [vp] * 11
    L.orc_ep_handle_exp_prepare_replies.argtypes = [vp] + [vp] * 15
    L.orc_ep_xp_dump.argtypes = [vp] + [vp] * 9
    L.orc_ep_dump.argtypes = [vp] + [vp] * 12
    L.orc_ep_set_execute.argtypes = [vp, u8]
    L.orc_ep_exec_dump.argtypes = [vp] + [vp] * 4
    L.orc_ep_take_submissions.restype = u64; L.orc_ep_take_submissions.argtypes = [vp, vp, vp, vp, u64]
    L.orc_rsp_new.restype = vp; L.orc_rsp_new.argtypes = [u32, u8, u8, u32, u8]
    L.orc_rsp_free.argtypes = [vp]
    L.orc_rsp_preset_leader.argtypes = [vp, u8]
    L.orc_rsp_req_batch.argtypes = [vp] + [vp] * 5
    L.orc_rsp_accept.argtypes = [vp] + [vp] * 8
    L.orc_rsp_accept_replies.argtypes = [vp] + [vp] * 5
    L.orc_rsp_become_leader.argtypes = [vp] + [vp] * 11
    L.orc_rsp_prepare.argtypes = [vp] + [vp] * 11
    L.orc_rsp_prepare_replies.argtypes = [vp] + [vp] * 12
    L.orc_rsp_reconstruct.argtypes = [vp] + [vp] * 8
    L.orc_rsp_reconstruct_reply.argtypes = [vp] + [vp] * 6
    L.orc_rsp_heartbeat.argtypes = [vp] + [vp] * 11
    L.orc_rsp_bcast_heartbeat.argtypes = [vp] + [vp] * 5
    L.orc_rsp_dump.argtypes = [vp] + [vp] * 27
    L.orc_rsp_take_executed.restype = u64; L.orc_rsp_take_executed.argtypes = [vp, vp, vp, vp, u64]


# ---------------------------------------------------------------- GF / RS ---
def gf_mul(a, b):
    return lib().orc_gf_mul(a, b)


def gf_exp(a, n):
    return lib().orc_gf_exp(a, n)


def rs_matrix(d, p):
    m = np.zeros((d + p, d), np.uint8)
    rc = lib().orc_rs_matrix(d, p, _p(m))
    if rc != 0:
        raise ValueError("bad RS scheme")
    return m


def rs_shard_len(data_len, d):
    return int(lib().orc_rs_shard_len(data_len, d))


def rs_encode(d, p, data):
    """from_data geometry + compute_parity for one codeword -> [p, shard_len]."""
    data = np.ascontiguousarray(data, np.uint8)
    if d <= 0:
        raise ValueError("num_data_shards is zero")
    sl = rs_shard_len(data.size, d)
    par = np.zeros((p, sl), np.uint8)
    rc = lib().orc_rs_encode(d, p, _p(data), data.size, _p(par))
    if rc != 0:
        raise ValueError("codeword is null / bad scheme")
    return par


def rs_encode_batch(d, p, data, data_len, cw_stride, n_cw, par_stride=None):
    sl = rs_shard_len(data_len, d)
    if par_stride is None:
        par_stride = p * sl
    par = np.zeros(n_cw * par_stride, np.uint8)
    rc = lib().orc_rs_encode_batch(d, p, _p(data), data_len, cw_stride, n_cw, _p(par), par_stride)
    if rc != 0:
        raise ValueError("codeword is null / bad scheme")
    return par


def rs_reconstruct(d, p, shards, present, data_only=False):
    """shards [d+p, shard_len] (modified in place), present [d+p] bool."""
    shards = np.ascontiguousarray(shards, np.uint8)
    pres = np.ascontiguousarray(present, np.uint8).copy()
    rc = lib().orc_rs_reconstruct(d, p, _p(shards), shards.shape[1], _p(pres), int(data_only))
    if rc != 0:
        raise ValueError("too few shards present")
    return shards, pres.astype(bool)


def rs_verify(d, p, shards):
    shards = np.ascontiguousarray(shards, np.uint8)
    return bool(lib().orc_rs_verify(d, p, _p(shards), shards.shape[1]))


def bincode_string(s):
    s = np.frombuffer(bytes(s), np.uint8)
    out = np.zeros(s.size + 9, np.uint8)
    n = lib().orc_bincode_string(_p(s), s.size, _p(out))
    return out[:n].copy()


def bincode_reqbatch_put(client, req_id, key, value):
    k = np.frombuffer(bytes(key), np.uint8)
    v = np.frombuffer(bytes(value), np.uint8)
    out = np.zeros(k.size + v.size + 64, np.uint8)
    n = lib().orc_bincode_reqbatch_put(client, req_id, _p(k), k.size, _p(v), v.size, _p(out))
    return out[:n].copy()


# --------------------------------------------------------------- MultiPaxos -
MP_SCALARS = ["leader", "bal_prep_sent", "bal_prepared", "bal_max_seen", "start_slot", "log_len",
              "accept_bar", "commit_bar", "exec_bar", "snap_bar"]
MP_SLOTS = [("s_bal", np.uint64), ("s_status", np.uint8), ("s_reqs", np.uint32), ("s_vbal", np.uint64),
            ("s_vreqs", np.uint32), ("s_flags", np.uint8), ("s_acks", np.uint8), ("s_packs", np.uint8),
            ("s_pmax", np.uint64), ("s_ltrig", np.uint32), ("s_lendp", np.uint32), ("s_src", np.uint8),
            ("s_rtrig", np.uint32), ("s_rendp", np.uint32)]
_MP_SCALAR_T = {"leader": np.uint8, "bal_prep_sent": np.uint64, "bal_prepared": np.uint64,
                "bal_max_seen": np.uint64}


class MpOracle:
    """G groups x R replicas of the literal MultiPaxos restatement (LS-1)."""

    def __init__(self, G, R=5, W=64, win_reserve=None, cap=None, commit_extra=0, record_commits=True):
        self.G, self.R, self.W = G, R, W
        self.win_reserve = W // 4 if win_reserve is None else win_reserve
        self.cap = W + 4 if cap is None else cap
        self.h = lib().orc_mp_new(G, R, W, self.win_reserve, self.cap, commit_extra, int(record_commits))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mp_free(self.h)
            self.h = None

    def preset_leader(self, rep=0):
        lib().orc_mp_preset_leader(self.h, rep)

    def tick(self, timeout_rep=None, timeout_src=None, req_target=None, req_cnt=None, req_val=None,
             ackctl=None, heartbeat=False):
        S = 0 if req_val is None else req_val.shape[0]
        for a, t in ((timeout_rep, np.uint8), (timeout_src, np.uint8), (req_target, np.uint8),
                     (req_cnt, np.uint32), (req_val, np.uint32), (ackctl, np.uint32)):
            assert a is None or (a.dtype == t and a.flags.c_contiguous)
        if ackctl is not None:
            assert ackctl.shape == (self.cap, self.G)
        lib().orc_mp_tick(self.h, _p(timeout_rep), _p(timeout_src), _p(req_target), _p(req_cnt),
                          _p(req_val), S, _p(ackctl), int(heartbeat))

    def dump(self, rep):
        G, W, R = self.G, self.W, self.R
        out = {}
        for n in MP_SCALARS:
            out[n] = np.zeros(G, _MP_SCALAR_T.get(n, np.uint32))
        out["peer_exec_bar"] = np.zeros((R, G), np.uint32)
        for n, t in MP_SLOTS:
            out[n] = np.zeros((W, G), t)
        out["overflow"] = np.zeros(G, np.uint8)
        args = [out[n] for n in MP_SCALARS] + [out["peer_exec_bar"]] + [out[n] for n, _ in MP_SLOTS] \
            + [out["overflow"]]
        lib().orc_mp_dump(self.h, rep, *[_p(a) for a in args])
        return out

    def total_commits(self, rep):
        return int(lib().orc_mp_total_commits(self.h, rep))

    def take_commits(self, rep, max_n=1 << 24):
        g = np.zeros(max_n, np.uint32)
        s = np.zeros(max_n, np.uint32)
        n = int(lib().orc_mp_take_commits(self.h, rep, _p(g), _p(s), max_n))
        assert n <= max_n
        return g[:n].copy(), s[:n].copy()


# --------------------------------------------------------------------- Raft -
RAFT_FIELDS = ["role", "curr_term", "log_len", "last_commit", "last_snap", "next_slot", "try_next_slot",
               "match_slot", "entry_term", "leader", "start_slot"]
_RAFT_T = {"role": np.uint8, "leader": np.uint8, "curr_term": np.uint64, "entry_term": np.uint64}


class RaftOracle:
    """G groups of the literal Raft leader restatement."""

    def __init__(self, G, R=5, W=64, leader_id=0, term=1, commit_extra=0):
        self.G, self.R, self.W = G, R, W
        self.h = lib().orc_raft_new(G, R, W, leader_id, term, commit_extra)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_raft_free(self.h)
            self.h = None

    def append(self, n_new):
        assert n_new.dtype == np.uint32
        lib().orc_raft_leader_append(self.h, _p(n_new))

    def append_emit(self, n_new):
        """append + [R, G] first slot sent to each peer (0xFFFFFFFF = nothing)"""
        first = np.zeros((self.R, self.G), np.uint32)
        lib().orc_raft_leader_append_emit(self.h, _p(n_new), _p(first))
        return first

    def gather_entries(self, first, K):
        G = self.G
        m = dict(flags=np.zeros(G, np.uint8), leader=np.zeros(G, np.uint8), term=np.zeros(G, np.uint64),
                 prev_slot=np.zeros(G, np.uint32), prev_term=np.zeros(G, np.uint64), n_entries=np.zeros(G, np.uint32),
                 entry_term=np.zeros((K, G), np.uint64), leader_commit=np.zeros(G, np.uint32), last_snap=np.zeros(G, np.uint32))
        lib().orc_raft_gather_entries(self.h, _p(np.ascontiguousarray(first)), K, *[_p(m[k]) for k in (
            "flags", "leader", "term", "prev_slot", "prev_term", "n_entries", "entry_term", "leader_commit", "last_snap")])
        return m

    def handle_replies(self, reply_term, end_slot, flags, conflict_term=None, conflict_slot=None, order=None):
        assert reply_term.dtype == np.uint64 and end_slot.dtype == np.uint32 and flags.dtype == np.uint8
        lib().orc_raft_handle_replies(self.h, _p(reply_term), _p(end_slot), _p(conflict_term), _p(conflict_slot),
                                      _p(flags), _p(order))

    def dump(self):
        G, W, R = self.G, self.W, self.R
        out = {}
        for n in RAFT_FIELDS:
            shape = (R, G) if n in ("next_slot", "try_next_slot", "match_slot") else ((W, G) if n == "entry_term" else (G,))
            out[n] = np.zeros(shape, _RAFT_T.get(n, np.uint32))
        lib().orc_raft_dump(self.h, *[_p(out[n]) for n in RAFT_FIELDS])
        return out

    def total_commits(self):
        return int(lib().orc_raft_total_commits(self.h))

    def counters(self):
        c = np.zeros(4, np.uint64)
        lib().orc_raft_counters(self.h, _p(c))
        return c

    def ring_guard_hits(self):
        """entries a follower's AppendEntries handling skipped because they had left the term ring (a harness rule shared
        with the engine, not the reference's): a parity run should keep this at 0"""
        return int(lib().orc_raft_ring_guard_hits(self.h))

    # ---- follower side and elections ----
    def preset(self, role, leader, term, voted_for=0xFF):
        lib().orc_raft_preset(self.h, role, leader, term, voted_for)

    def handle_append_entries(self, flags, leader, term, prev_slot, prev_term, n_entries, entry_term, leader_commit,
                              last_snap):
        G = self.G
        K = entry_term.shape[0]
        assert entry_term.dtype == np.uint64 and entry_term.shape == (K, G) and entry_term.flags.c_contiguous
        r = dict(flags=np.zeros(G, np.uint8), term=np.zeros(G, np.uint64), end_slot=np.zeros(G, np.uint32),
                 conflict_term=np.zeros(G, np.uint64), conflict_slot=np.zeros(G, np.uint32))
        lib().orc_raft_handle_append_entries(self.h, _p(flags), _p(leader), _p(term), _p(prev_slot), _p(prev_term),
                                             _p(n_entries), _p(entry_term), K, _p(leader_commit), _p(last_snap),
                                             _p(r["flags"]), _p(r["term"]), _p(r["end_slot"]), _p(r["conflict_term"]),
                                             _p(r["conflict_slot"]))
        return r

    def become_candidate(self, timeout_src):
        G = self.G
        r = dict(flags=np.zeros(G, np.uint8), term=np.zeros(G, np.uint64), last_slot=np.zeros(G, np.uint32),
                 last_term=np.zeros(G, np.uint64))
        lib().orc_raft_become_candidate(self.h, _p(timeout_src), _p(r["flags"]), _p(r["term"]), _p(r["last_slot"]),
                                        _p(r["last_term"]))
        return r

    def handle_request_vote(self, flags, candidate, term, last_slot, last_term):
        G = self.G
        r = dict(flags=np.zeros(G, np.uint8), term=np.zeros(G, np.uint64))
        lib().orc_raft_handle_request_vote(self.h, _p(flags), _p(candidate), _p(term), _p(last_slot), _p(last_term),
                                           _p(r["flags"]), _p(r["term"]))
        return r

    def handle_vote_replies(self, term, flags, order=None, granted=None):
        G, R = self.G, self.R
        r = dict(hb_prev_slot=np.zeros((R, G), np.uint32), elected=np.zeros(G, np.uint8))
        granted = np.ones((R, G), np.uint8) if granted is None else granted
        lib().orc_raft_handle_vote_replies(self.h, _p(term), _p(granted), _p(flags), _p(order), _p(r["hb_prev_slot"]),
                                           _p(r["elected"]))
        return r

    def dump_votes(self):
        G = self.G
        r = dict(voted_for=np.zeros(G, np.uint8), votes=np.zeros(G, np.uint8), n_exec=np.zeros(G, np.uint64),
                 n_trunc=np.zeros(G, np.uint64))
        lib().orc_raft_dump_votes(self.h, _p(r["voted_for"]), _p(r["votes"]), _p(r["n_exec"]), _p(r["n_trunc"]))
        return r


EP_NONE = 0xFFFFFFFF
EP_NO_KEY = 0xFF


class CRaftOracle(RaftOracle):
    """the CRaft leader variant of the restatement (oracle/raft_oracle.c, orc_craft_*)"""

    def __init__(self, G, R=5, W=64, leader_id=0, term=1, fault_tolerance=1, repeat_threshold=3):
        super().__init__(G, R, W, leader_id, term, 0)
        lib().orc_craft_enable(self.h, fault_tolerance, repeat_threshold)

    def bcast_heartbeats(self):
        R, G = self.R, self.G
        m = dict(hb_flags=np.zeros((R, G), np.uint8), prev_slot=np.zeros((R, G), np.uint32), prev_term=np.zeros((R, G), np.uint64),
                 leader_commit=np.zeros(G, np.uint32), last_snap=np.zeros(G, np.uint32))
        lib().orc_craft_bcast_heartbeats(self.h, *[_p(m[k]) for k in ("hb_flags", "prev_slot", "prev_term", "leader_commit", "last_snap")])
        return m

    def switch_assignment_mode(self, to_full_copy):
        assert to_full_copy.dtype == np.uint8
        lib().orc_craft_switch_assignment_mode(self.h, _p(to_full_copy))

    def assignment(self):
        persist, send = np.zeros(self.G, np.uint32), np.zeros((self.R, self.G), np.uint32)
        lib().orc_craft_assignment(self.h, _p(persist), _p(send))
        return persist, send

    def dump_craft(self):
        R, G = self.R, self.G
        out = dict(full_copy_mode=np.zeros(G, np.uint8), peer_alive=np.zeros(G, np.uint8), hb_replied=np.zeros((R, G), np.uint64),
                   hb_seen=np.zeros((R, G), np.uint64), hb_repeat=np.zeros((R, G), np.uint8))
        lib().orc_craft_dump(self.h, *[_p(out[k]) for k in ("full_copy_mode", "peer_alive", "hb_replied", "hb_seen", "hb_repeat")])
        return out

    # ---- the CRaft follower (craft/messages.rs:14-254, :622-663) ----
    def handle_append_entries(self, flags, leader, term, prev_slot, prev_term, n_entries, entry_term, leader_commit, last_snap,
                              entry_mask=None):
        """entry_mask [K, G] uint8: avail_shards_map of every sent entry's codeword (None: every shard)"""
        G = self.G
        K = entry_term.shape[0]
        if entry_mask is None:
            entry_mask = np.full((K, G), (1 << self.R) - 1, np.uint8)
        assert entry_term.dtype == np.uint64 and entry_term.shape == (K, G) and entry_term.flags.c_contiguous
        assert entry_mask.dtype == np.uint8 and entry_mask.shape == (K, G) and entry_mask.flags.c_contiguous
        r = dict(flags=np.zeros(G, np.uint8), term=np.zeros(G, np.uint64), end_slot=np.zeros(G, np.uint32),
                 conflict_term=np.zeros(G, np.uint64), conflict_slot=np.zeros(G, np.uint32))
        lib().orc_craft_handle_append_entries(self.h, _p(flags), _p(leader), _p(term), _p(prev_slot), _p(prev_term), _p(n_entries),
                                              _p(entry_term), _p(entry_mask), K, _p(leader_commit), _p(last_snap), _p(r["flags"]),
                                              _p(r["term"]), _p(r["end_slot"]), _p(r["conflict_term"]), _p(r["conflict_slot"]))
        return r

    def handle_reconstruct(self, n, slot, term):
        """Reconstruct { slots }: n [G], slot / term [K, G] -> the ReconstructReply: n [G], has / mask [K, G]"""
        G, K = self.G, slot.shape[0]
        r = dict(n=np.zeros(G, np.uint32), has=np.zeros((K, G), np.uint8), mask=np.zeros((K, G), np.uint8))
        lib().orc_craft_handle_reconstruct(self.h, _p(n), _p(slot), _p(term), K, _p(r["n"]), _p(r["has"]), _p(r["mask"]))
        return r

    def dump_masks(self):
        d = dict(mask=np.zeros((self.W, self.G), np.uint8), counters=np.zeros(2, np.uint64))
        lib().orc_craft_dump_masks(self.h, _p(d["mask"]), _p(d["counters"]))
        return d

    def take_reconstructs(self, K=16):
        """the Reconstruct { slots } the reply handler queued since the last call: dict(n [G], slot / term [K, G])"""
        r = dict(n=np.zeros(self.G, np.uint32), slot=np.zeros((K, self.G), np.uint32), term=np.zeros((K, self.G), np.uint64))
        lib().orc_craft_take_reconstructs(self.h, K, _p(r["n"]), _p(r["slot"]), _p(r["term"]))
        return r

    def handle_reconstruct_reply(self, peer, n, slot, mask):
        lib().orc_craft_handle_reconstruct_reply(self.h, _p(peer), _p(n), _p(slot), _p(mask), slot.shape[0])



class EpOracle:
    """G groups of the literal EPaxos command-leader / acceptor restatement (replica `me`)."""

    def __init__(self, G, R=5, me=0, W=32, n_keys=64, optimized_quorum=True, execute=False):
        self.G, self.R, self.me, self.W, self.n_keys = G, R, me, W, n_keys
        self.h = lib().orc_ep_new(G, R, me, W, n_keys, int(optimized_quorum))
        if execute:
            lib().orc_ep_set_execute(self.h, 1)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ep_free(self.h)
            self.h = None

    def propose(self, key, exploded=None):
        G, R = self.G, self.R
        m = dict(flags=np.zeros(G, np.uint8), col=np.zeros(G, np.uint32), seq=np.zeros(G, np.uint64),
                 deps=np.zeros((R, G), np.uint32))
        lib().orc_ep_propose(self.h, _p(key), _p(exploded), _p(m["flags"]), _p(m["col"]), _p(m["seq"]), _p(m["deps"]))
        return m

    def handle_pre_accept(self, flags, peer, col, ballot, seq, deps, key, row=None):
        """row: the slot's row where it is not the sender's (an instance under explicit prepare); None = peer"""
        G, R = self.G, self.R
        r = dict(flags=np.zeros(G, np.uint8), ballot=np.zeros(G, np.uint64), seq=np.zeros(G, np.uint64),
                 deps=np.zeros((R, G), np.uint32))
        lib().orc_ep_handle_pre_accept(self.h, _p(flags), _p(peer), _p(col), _p(ballot), _p(seq), _p(deps), _p(key),
                                       _p(r["flags"]), _p(r["ballot"]), _p(r["seq"]), _p(r["deps"]), _p(row))
        return r

    def handle_pre_accept_replies(self, col, ballot, seq, deps, flags, order=None, exploded=None, row=None):
        G, R = self.G, self.R
        assert deps.shape == (R, R, G) and ballot.shape == (R, G)
        r = dict(decision=np.zeros(G, np.uint8), seq=np.zeros(G, np.uint64), deps=np.zeros((R, G), np.uint32))
        lib().orc_ep_handle_pre_accept_replies(self.h, _p(col), _p(ballot), _p(seq), _p(deps), _p(flags), _p(order),
                                               _p(exploded), _p(r["decision"]), _p(r["seq"]), _p(r["deps"]), _p(row))
        return r

    def handle_accept(self, flags, peer, col, ballot, seq, deps, key, row=None):
        G = self.G
        r = dict(flags=np.zeros(G, np.uint8), ballot=np.zeros(G, np.uint64))
        lib().orc_ep_handle_accept(self.h, _p(flags), _p(peer), _p(col), _p(ballot), _p(seq), _p(deps), _p(key),
                                   _p(r["flags"]), _p(r["ballot"]), _p(row))
        return r

    def handle_commit_notice(self, flags, peer, col, ballot, seq, deps, key, row=None):
        lib().orc_ep_handle_commit_notice(self.h, _p(flags), _p(peer), _p(col), _p(ballot), _p(seq), _p(deps), _p(key), _p(row))

    def handle_accept_replies(self, col, ballot, flags, order=None, row=None):
        r = dict(committed=np.zeros(self.G, np.uint8))
        lib().orc_ep_handle_accept_replies(self.h, _p(col), _p(ballot), _p(flags), _p(order), _p(r["committed"]), _p(row))
        return r

    # ---- explicit prepare (heartbeat.rs:17-125, messages.rs:511-821) ----
    def heartbeat_timeout(self, src, exploded=None):
        """HearTimeout { peer: src[g] } (0xFF: none) -> the ExpPrepare broadcasts: n [G], col / ballot [W, G]"""
        G, W = self.G, self.W
        o = dict(n=np.zeros(G, np.uint32), col=np.zeros((W, G), np.uint32), ballot=np.zeros((W, G), np.uint64))
        lib().orc_ep_heartbeat_timeout(self.h, _p(src), _p(exploded), _p(o["n"]), _p(o["col"]), _p(o["ballot"]))
        return o

    def handle_exp_prepare(self, flags, peer, row, col, new_ballot):
        G, R = self.G, self.R
        r = dict(flags=np.zeros(G, np.uint8), voted_bal=np.zeros(G, np.uint64), status=np.zeros(G, np.uint8),
                 seq=np.zeros(G, np.uint64), deps=np.zeros((R, G), np.uint32), key=np.zeros(G, np.uint8))
        lib().orc_ep_handle_exp_prepare(self.h, _p(flags), _p(peer), _p(row), _p(col), _p(new_ballot), _p(r["flags"]),
                                        _p(r["voted_bal"]), _p(r["status"]), _p(r["seq"]), _p(r["deps"]), _p(r["key"]))
        return r

    def handle_exp_prepare_replies(self, row, col, new_ballot, voted_bal, voted_status, voted_seq, voted_deps, voted_key, flags,
                                   order=None):
        G, R = self.G, self.R
        assert voted_deps.shape == (R, R, G) and new_ballot.shape == (R, G)
        r = dict(decision=np.zeros(G, np.uint8), ballot=np.zeros(G, np.uint64), seq=np.zeros(G, np.uint64),
                 deps=np.zeros((R, G), np.uint32), key=np.zeros(G, np.uint8))
        lib().orc_ep_handle_exp_prepare_replies(self.h, _p(row), _p(col), _p(new_ballot), _p(voted_bal), _p(voted_status),
                                                _p(voted_seq), _p(voted_deps), _p(voted_key), _p(flags), _p(order),
                                                _p(r["decision"]), _p(r["ballot"]), _p(r["seq"]), _p(r["deps"]), _p(r["key"]))
        return r

    def xp_dump(self):
        G, R, W = self.G, self.R, self.W
        d = dict(acks=np.zeros((R, W, G), np.uint8), max_bal=np.zeros((R, W, G), np.uint64), avoid=np.zeros((R, W, G), np.uint8),
                 has=np.zeros((R, W, G), np.uint8), vstatus=np.zeros((R, W, R, G), np.uint8), vseq=np.zeros((R, W, R, G), np.uint64),
                 vkey=np.zeros((R, W, R, G), np.uint8), vdeps=np.zeros((R, W, R, R, G), np.uint32), counters=np.zeros(4, np.uint64))
        lib().orc_ep_xp_dump(self.h, *[_p(d[k]) for k in ("acks", "max_bal", "avoid", "has", "vstatus", "vseq", "vkey", "vdeps", "counters")])
        return d

    def dump(self):
        G, R, W, K = self.G, self.R, self.W, self.n_keys
        d = dict(len=np.zeros((R, G), np.uint32), commit_bars=np.zeros((R, G), np.uint32),
                 bal=np.zeros((R, W, G), np.uint64), seq=np.zeros((R, W, G), np.uint64),
                 status=np.zeros((R, W, G), np.uint8), key=np.zeros((R, W, G), np.uint8),
                 deps=np.zeros((R, W, G, R), np.uint32), pa_acks=np.zeros((R, W, G), np.uint8),
                 acc_acks=np.zeros((R, W, G), np.uint8), bk=np.zeros((R, W, G), np.uint8),
                 highest_cols=np.zeros((K, R, G), np.uint32), counters=np.zeros(3, np.uint64))
        lib().orc_ep_dump(self.h, *[_p(d[k]) for k in ("len", "commit_bars", "bal", "seq", "status", "key", "deps", "pa_acks",
                                                       "acc_acks", "bk", "highest_cols", "counters")])
        return d

    def take_submissions(self):
        """(group, row, col) of the commands submitted since the last call: group-major, submission order per group"""
        cap = 1 << 16
        while True:
            g, r, c = np.zeros(cap, np.uint32), np.zeros(cap, np.uint8), np.zeros(cap, np.uint32)
            n = lib().orc_ep_take_submissions(self.h, _p(g), _p(r), _p(c), cap)
            if n <= cap:
                return g[:n], r[:n], c[:n]
            raise RuntimeError("more than %d submissions between two polls" % cap)

    def exec_dump(self):
        """execution state: exec_bars [R, G], kv [n_keys, G] (token of the last Put), digest [G], counters [6]"""
        G, R, K = self.G, self.R, self.n_keys
        d = dict(exec_bars=np.zeros((R, G), np.uint32), kv=np.zeros((K, G), np.uint64), digest=np.zeros(G, np.uint64),
                 counters=np.zeros(6, np.uint64))
        lib().orc_ep_exec_dump(self.h, _p(d["exec_bars"]), _p(d["kv"]), _p(d["digest"]), _p(d["counters"]))
        return d


# ------------------------------------------------------------- RSPaxos ---
RSP_SCALARS = (("leader", np.uint8), ("bal_prep_sent", np.uint64), ("bal_prepared", np.uint64), ("bal_max_seen", np.uint64),
               ("len", np.uint32), ("commit_bar", np.uint32), ("exec_bar", np.uint32), ("snap_bar", np.uint32))
RSP_SLOTS = (("s_bal", np.uint64), ("s_status", np.uint8), ("s_val", np.uint32), ("s_mask", np.uint8), ("s_vbal", np.uint64),
             ("s_vval", np.uint32), ("s_vmask", np.uint8), ("s_flags", np.uint8), ("s_ltrig", np.uint32), ("s_lendp", np.uint32),
             ("s_packs", np.uint8), ("s_aacks", np.uint8), ("s_pmax", np.uint64), ("s_rsrc", np.uint8), ("s_rtrig", np.uint32),
             ("s_rendp", np.uint32))
RSP_NULL = 0xFFFFFFFF


class RspOracle:
    """G groups of the literal RSPaxos replica restatement (replica `me`); messages are dicts of numpy arrays with one
    entry per group, lists as [W, G] with a count."""

    def __init__(self, G, R=5, me=0, W=32, fault_tolerance=0):
        self.G, self.R, self.me, self.W = G, R, me, W
        self.h = lib().orc_rsp_new(G, R, me, W, fault_tolerance)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_rsp_free(self.h)
            self.h = None

    def preset_leader(self, leader):
        lib().orc_rsp_preset_leader(self.h, leader)

    def _accepts(self):
        G, W = self.G, self.W
        return dict(a_n=np.zeros(G, np.uint32), a_slot=np.zeros((W, G), np.uint32), a_val=np.zeros((W, G), np.uint32),
                    a_ballot=np.zeros(G, np.uint64))

    def req_batch(self, val):
        o = self._accepts()
        lib().orc_rsp_req_batch(self.h, _p(val), *[_p(o[k]) for k in ("a_n", "a_slot", "a_val", "a_ballot")])
        return o

    def accept(self, flags, peer, slot, ballot, val, mask):
        o = dict(r_ballot=np.zeros(self.G, np.uint64), r_slot=np.zeros(self.G, np.uint32))
        lib().orc_rsp_accept(self.h, _p(flags), _p(peer), _p(slot), _p(ballot), _p(val), _p(mask), _p(o["r_ballot"]), _p(o["r_slot"]))
        return o

    def accept_replies(self, slot, ballot, flags, order=None):
        o = dict(committed=np.zeros(self.G, np.uint8))
        lib().orc_rsp_accept_replies(self.h, _p(slot), _p(ballot), _p(flags), _p(order), _p(o["committed"]))
        return o

    def become_leader(self, src):
        G, W = self.G, self.W
        o = dict(hb_flags=np.zeros(G, np.uint8), hb_ballot=np.zeros(G, np.uint64), hb_commit=np.zeros(G, np.uint32),
                 hb_exec=np.zeros(G, np.uint32), hb_snap=np.zeros(G, np.uint32), p_flags=np.zeros(G, np.uint8),
                 p_trig=np.zeros(G, np.uint32), p_ballot=np.zeros(G, np.uint64), rc_n=np.zeros(G, np.uint32),
                 rc_slot=np.zeros((W, G), np.uint32))
        lib().orc_rsp_become_leader(self.h, _p(src), *[_p(o[k]) for k in ("hb_flags", "hb_ballot", "hb_commit", "hb_exec", "hb_snap",
                                                                           "p_flags", "p_trig", "p_ballot", "rc_n", "rc_slot")])
        return o

    def prepare(self, flags, peer, trig, ballot):
        G, W = self.G, self.W
        o = dict(pr_n=np.zeros(G, np.uint32), pr_trig=np.zeros(G, np.uint32), pr_endp=np.zeros(G, np.uint32),
                 pr_ballot=np.zeros(G, np.uint64), pr_vbal=np.zeros((W, G), np.uint64), pr_vval=np.zeros((W, G), np.uint32),
                 pr_vmask=np.zeros((W, G), np.uint8))
        lib().orc_rsp_prepare(self.h, _p(flags), _p(peer), _p(trig), _p(ballot),
                              *[_p(o[k]) for k in ("pr_n", "pr_trig", "pr_endp", "pr_ballot", "pr_vbal", "pr_vval", "pr_vmask")])
        return o

    def prepare_replies(self, peer, pr_n, pr_trig, pr_endp, pr_ballot, pr_vbal, pr_vval, pr_vmask):
        o = self._accepts()
        lib().orc_rsp_prepare_replies(self.h, _p(peer), _p(pr_n), _p(pr_trig), _p(pr_endp), _p(pr_ballot), _p(pr_vbal), _p(pr_vval),
                                      _p(pr_vmask), *[_p(o[k]) for k in ("a_n", "a_slot", "a_val", "a_ballot")])
        return o

    def reconstruct(self, flags, rc_n, rc_slot):
        G, W = self.G, self.W
        o = dict(rr_n=np.zeros(G, np.uint32), rr_slot=np.zeros((W, G), np.uint32), rr_bal=np.zeros((W, G), np.uint64),
                 rr_val=np.zeros((W, G), np.uint32), rr_mask=np.zeros((W, G), np.uint8))
        lib().orc_rsp_reconstruct(self.h, _p(flags), _p(rc_n), _p(rc_slot), *[_p(o[k]) for k in ("rr_n", "rr_slot", "rr_bal", "rr_val", "rr_mask")])
        return o

    def reconstruct_reply(self, flags, rr_n, rr_slot, rr_bal, rr_val, rr_mask):
        lib().orc_rsp_reconstruct_reply(self.h, _p(flags), _p(rr_n), _p(rr_slot), _p(rr_bal), _p(rr_val), _p(rr_mask))

    def heartbeat(self, flags, peer, ballot, commit_bar, exec_bar, snap_bar):
        G = self.G
        o = dict(reply=np.zeros(G, np.uint8), ballot=np.zeros(G, np.uint64), commit_bar=np.zeros(G, np.uint32),
                 exec_bar=np.zeros(G, np.uint32), snap_bar=np.zeros(G, np.uint32))
        lib().orc_rsp_heartbeat(self.h, _p(flags), _p(peer), _p(ballot), _p(commit_bar), _p(exec_bar), _p(snap_bar),
                                *[_p(o[k]) for k in ("reply", "ballot", "commit_bar", "exec_bar", "snap_bar")])
        return o

    def bcast_heartbeat(self, flags):
        G = self.G
        o = dict(ballot=np.zeros(G, np.uint64), commit_bar=np.zeros(G, np.uint32), exec_bar=np.zeros(G, np.uint32),
                 snap_bar=np.zeros(G, np.uint32))
        lib().orc_rsp_bcast_heartbeat(self.h, _p(flags), *[_p(o[k]) for k in ("ballot", "commit_bar", "exec_bar", "snap_bar")])
        return o

    def is_leader(self):
        return (self.dump()["leader"] == self.me).astype(np.uint8)

    def take_executed(self):
        """(group, slot, token) of the commands executed since the last call: group-major, execution order per group"""
        cap = 1 << 18
        g, s, v = np.zeros(cap, np.uint32), np.zeros(cap, np.uint32), np.zeros(cap, np.uint32)
        n = lib().orc_rsp_take_executed(self.h, _p(g), _p(s), _p(v), cap)
        assert n <= cap
        return g[:n], s[:n], v[:n]

    def dump(self):
        G, R, W = self.G, self.R, self.W
        d = {n: np.zeros(G, t) for n, t in RSP_SCALARS}
        d["peer_exec_bar"] = np.zeros((R, G), np.uint32)
        d["digest"] = np.zeros(G, np.uint64)
        for n, t in RSP_SLOTS:
            d[n] = np.zeros((W, G), t)
        d["counters"] = np.zeros(4, np.uint64)
        order = [n for n, _ in RSP_SCALARS] + ["peer_exec_bar", "digest"] + [n for n, _ in RSP_SLOTS] + ["counters"]
        lib().orc_rsp_dump(self.h, *[_p(d[k]) for k in order])
        return d


class QrOracle:
    """G groups of the literal quorum-read restatement (oracle/qr_oracle.c)"""

    def __init__(self, G, R=5, me=0, K=16, B=4, Q=2):
        self.G, self.R, self.me, self.K, self.B, self.Q = G, R, me, K, B, Q
        self.h = lib().orc_qr_new(G, R, me, K, B, Q)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_qr_free(self.h)
            self.h = None

    def refresh_highest_slot(self, slot, put_keys):
        assert slot.dtype == np.uint32 and put_keys.dtype == np.uint8 and put_keys.shape == (self.B, self.G)
        lib().orc_qr_refresh_highest_slot(self.h, _p(slot), _p(put_keys))

    def handle_read_query(self, keys, n, log, stable_leader=None, kv=None):
        B, G = self.B, self.G
        out = dict(state=np.zeros((B, G), np.uint8), slot=np.zeros((B, G), np.uint32), val=np.zeros((B, G), np.uint32))
        fl = np.zeros(G, np.uint8)
        lib().orc_qr_handle_read_query(self.h, _p(keys), _p(n), _p(stable_leader), _p(kv), _p(log["start_slot"]), _p(log["log_end"]),
                                       _p(log["status"]), _p(log["token"]), log["status"].shape[0], _p(out["state"]), _p(out["slot"]),
                                       _p(out["val"]), _p(fl))
        return out, fl

    def issue(self, q, n, own):
        lib().orc_qr_issue(self.h, q, _p(n), _p(own["state"]), _p(own["slot"]), _p(own["val"]))

    def handle_replies(self, q, replies, flags, order=None):
        B, G = self.B, self.G
        outcome, out_val, done = np.zeros((B, G), np.uint8), np.zeros((B, G), np.uint32), np.zeros(G, np.uint8)
        lib().orc_qr_handle_replies(self.h, q, _p(replies["state"]), _p(replies["slot"]), _p(replies["val"]), _p(flags), _p(order),
                                    _p(outcome), _p(out_val), _p(done))
        return outcome, out_val, done

    def dump(self):
        G, K, B, Q = self.G, self.K, self.B, self.Q
        out = dict(highest_slot=np.zeros((K, G), np.uint32), live=np.zeros((Q, G), np.uint8), n=np.zeros((Q, G), np.uint8),
                   rq_acks=np.zeros((Q, G), np.uint8), mx_state=np.zeros((Q, B, G), np.uint8), mx_slot=np.zeros((Q, B, G), np.uint32),
                   mx_val=np.zeros((Q, B, G), np.uint32), counters=np.zeros(4, np.uint64))
        lib().orc_qr_dump(self.h, *[_p(out[k]) for k in ("highest_slot", "live", "n", "rq_acks", "mx_state", "mx_slot", "mx_val", "counters")])
        return out


# ------------------------------------------------------------- Bitmap ---
class _BitmapS(C.Structure):
    _fields_ = [("size", C.c_uint8), ("bits", C.c_uint64)]


class Bitmap:
    """`Bitmap` of src/utils/bitmap.rs restated (oracle/bitmap_oracle.c); errors as ValueError, the panics as AssertionError"""

    def __init__(self, size, ones=False, _from=None):
        self.m = _BitmapS()
        L = lib()
        rc = L.orc_bitmap_new(C.byref(self.m), size, int(ones)) if _from is None else \
            L.orc_bitmap_from(C.byref(self.m), size, (C.c_uint8 * max(len(_from), 1))(*_from), len(_from))
        if rc:
            raise AssertionError("invalid bitmap size %d" % size if _from is None or size == 0 else "index out of bound")

    @classmethod
    def from_ones(cls, size, ones):
        return cls(size, _from=sorted(set(int(x) for x in ones)))

    def set(self, idx, flag):
        if lib().orc_bitmap_set(C.byref(self.m), idx, int(flag)):
            raise ValueError("index %d out of bound" % idx)

    def get(self, idx):
        out = C.c_int()
        if lib().orc_bitmap_get(C.byref(self.m), idx, C.byref(out)):
            raise ValueError("index %d out of bound" % idx)
        return bool(out.value)

    def size(self):
        return int(lib().orc_bitmap_size(C.byref(self.m)))

    def count(self):
        return int(lib().orc_bitmap_count(C.byref(self.m)))

    def flip(self):
        lib().orc_bitmap_flip(C.byref(self.m))

    def union(self, other):
        if lib().orc_bitmap_union(C.byref(self.m), C.byref(other.m)):
            raise ValueError("unioning sizes mismatch: %d != %d" % (self.size(), other.size()))

    def clear(self):
        lib().orc_bitmap_clear(C.byref(self.m))

    def to_vec(self):
        out = (C.c_uint8 * 64)()
        n = lib().orc_bitmap_to_vec(C.byref(self.m), out)
        return [int(out[i]) for i in range(n)]

    def iter(self):
        return [(i, self.get(i)) for i in range(self.size())]

    def mask(self):
        lib().orc_bitmap_mask.restype = C.c_uint64
        return int(lib().orc_bitmap_mask(C.byref(self.m)))

    def bincode(self):
        out = (C.c_uint8 * 32)()
        n = lib().orc_bitmap_bincode(C.byref(self.m), out)
        return bytes(out[:n])

    def __eq__(self, other):
        return self.size() == other.size() and self.mask() == other.mask()


# ------------------------------------------------------------- Heartbeater ---
HB_ALL, HB_NONE = 0xFE, 0xFF


class HbOracle:
    """`Heartbeater` of src/server/heartbeat.rs for G groups (oracle/hb_oracle.c): explicit clock and random draws"""

    def __init__(self, G, R=5, me=0, hear_min_ms=1200, hear_max_ms=2000, send_ms=20, now_ms=0):
        L = lib()
        L.orc_hb_new.restype = C.c_void_p
        L.orc_hb_new.argtypes = [C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]
        for n in ("orc_hb_free", "orc_hb_set_sending", "orc_hb_clear_reply_cnts", "orc_hb_update_heard_cnt"):
            getattr(L, n).argtypes = [C.c_void_p] + ([C.c_void_p] if n != "orc_hb_free" else [])
            getattr(L, n).restype = None
        L.orc_hb_kickoff_hear_timer.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_hb_poll.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        L.orc_hb_update_bcast_cnts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hb_dump.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        self.G, self.R, self.me = G, R, me
        self.h = L.orc_hb_new(G, R, me, hear_min_ms, hear_max_ms, send_ms, now_ms)
        if not self.h:
            raise ValueError("invalid heartbeat configuration")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_hb_free(self.h)
            self.h = None

    def set_sending(self, sending):
        lib().orc_hb_set_sending(self.h, _p(np.ascontiguousarray(sending, np.uint8)))

    def kickoff_hear_timer(self, peer, now_ms, draw):
        lib().orc_hb_kickoff_hear_timer(self.h, _p(np.ascontiguousarray(peer, np.uint8)), int(now_ms), _p(np.ascontiguousarray(draw, np.uint32)))

    def poll(self, now_ms):
        t, s = np.zeros((self.R, self.G), np.uint8), np.zeros(self.G, np.uint8)
        lib().orc_hb_poll(self.h, int(now_ms), _p(t), _p(s))
        return t, s

    def clear_reply_cnts(self, peer):
        lib().orc_hb_clear_reply_cnts(self.h, _p(np.ascontiguousarray(peer, np.uint8)))

    def update_bcast_cnts(self, flags):
        d = np.zeros(self.G, np.uint8)
        lib().orc_hb_update_bcast_cnts(self.h, _p(np.ascontiguousarray(flags, np.uint8)), _p(d))
        return d

    def update_heard_cnt(self, peer):
        lib().orc_hb_update_heard_cnt(self.h, _p(np.ascontiguousarray(peer, np.uint8)))

    def dump(self):
        R, G = self.R, self.G
        d = dict(deadline=np.zeros((R, G), np.uint64), exploded=np.zeros((R, G), np.uint8), is_sending=np.zeros(G, np.uint8),
                 next_tick=np.zeros(G, np.uint64), cnt0=np.zeros((R, G), np.uint64), cnt1=np.zeros((R, G), np.uint64),
                 rep=np.zeros((R, G), np.uint8), alive=np.zeros(G, np.uint8))
        lib().orc_hb_dump(self.h, *[_p(d[k]) for k in ("deadline", "exploded", "is_sending", "next_tick", "cnt0", "cnt1", "rep", "alive")])
        return d


class LeaseOracle:
    """`LeaseManager` of src/server/leaseman.rs for G groups (oracle/lease_oracle.c): explicit clock, one notice per group
    and call.  Notices / messages / actions are small tuples so the restated reference tests read like the originals."""
    N_NONE, N_NEW_GRANTS, N_DO_REVOKE, N_CLEAR_HELD, N_RECV_MSG = range(5)
    GUARD, GUARD_REPLY, PROMISE, PROMISE_REPLY, REVOKE, REVOKE_REPLY = range(6)
    A_SEND, A_BCAST, A_NEXT_REFRESH, A_GRANT_REMOVED, A_LEASE_CLEARED, A_GRANT_TIMEOUT, A_LEASE_TIMEOUT, A_HIGHER_NUMBER, \
        A_GUARD_ACCEPT_BAR = range(1, 10)
    ALL = 0xFF
    ACT_CAP = 20

    def __init__(self, G, R=5, me=0, expire_ms=2000, hb_send_ms=20):
        L = lib()
        L.orc_lease_new.restype = C.c_void_p
        L.orc_lease_new.argtypes = [C.c_uint32, C.c_uint8, C.c_uint8, C.c_uint64, C.c_uint64]
        L.orc_lease_free.argtypes = [C.c_void_p]
        L.orc_lease_step.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 16
        L.orc_lease_attempt_refresh.argtypes = [C.c_void_p, C.c_uint64] + [C.c_void_p] * 3
        L.orc_lease_dump.argtypes = [C.c_void_p] + [C.c_void_p] * 10
        self._L, self.G, self.R, self.me = L, G, R, me
        self._h = L.orc_lease_new(G, R, me, expire_ms, hb_send_ms)
        if not self._h:
            raise ValueError("invalid lease manager configuration")     # new_and_setup's logged_err! paths

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.orc_lease_free(self._h)
            self._h = None

    def step_arrays(self, now, kind, num, peer, peers, msg, held, has_bar, bar):
        """raw form: arrays of G entries in, (act_n[G], dict of [ACT_CAP, G] arrays) out"""
        G, A = self.G, self.ACT_CAP
        ins = [np.ascontiguousarray(kind, np.uint8), np.ascontiguousarray(num, np.uint64), np.ascontiguousarray(peer, np.uint8),
               np.ascontiguousarray(peers, np.uint8), np.ascontiguousarray(msg, np.uint8), np.ascontiguousarray(held, np.uint8),
               np.ascontiguousarray(has_bar, np.uint8), np.ascontiguousarray(bar, np.uint64)]
        n = np.zeros(G, np.uint8)
        out = dict(num=np.zeros((A, G), np.uint64), kind=np.zeros((A, G), np.uint8), peer=np.zeros((A, G), np.uint8),
                   mask=np.zeros((A, G), np.uint8), msg=np.zeros((A, G), np.uint8), flag=np.zeros((A, G), np.uint8),
                   bar=np.zeros((A, G), np.uint64))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._L.orc_lease_step(self._h, int(now), *[p(a) for a in ins], p(n),
                               *[p(out[k]) for k in ("num", "kind", "peer", "mask", "msg", "flag", "bar")])
        return n, out

    def attempt_refresh_arrays(self, now, call, peers):
        call = np.ascontiguousarray(call, np.uint8)
        peers = np.ascontiguousarray(peers, np.uint8)
        out = np.zeros(self.G, np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._L.orc_lease_attempt_refresh(self._h, int(now), p(call), p(peers), p(out))
        return out

    def dump(self):
        R, G = self.R, self.G
        d = dict(active_num=np.zeros(G, np.uint64), grant_set=np.zeros(G, np.uint8), lease_set=np.zeros(G, np.uint8),
                 lease_cnt=np.zeros(G, np.uint8), guards_sent=np.zeros(G, np.uint8), guards_held=np.zeros(G, np.uint8),
                 refresh_mark=np.zeros(G, np.uint8), ps_deadline=np.zeros((R, G), np.uint64),
                 gh_deadline=np.zeros((R, G), np.uint64), ph_deadline=np.zeros((R, G), np.uint64))
        self._L.orc_lease_dump(self._h, *[d[k].ctypes.data_as(C.c_void_p) for k in
                                          ("active_num", "grant_set", "lease_set", "lease_cnt", "guards_sent", "guards_held",
                                           "refresh_mark", "ps_deadline", "gh_deadline", "ph_deadline")])
        return d
