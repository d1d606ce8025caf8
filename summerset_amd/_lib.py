"""ctypes binding of libsummerset_hip.so -- the only way Python reaches the engine.

There is NO CPU fallback: if the HIP library is missing or a call fails, the
caller gets an exception.  (The CPU oracle lives under oracle/ and is test
infrastructure only; this package never imports it.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsummerset_hip.so")

SMR_OK, SMR_ERR_ARG, SMR_ERR_DEVICE, SMR_ERR_STATE = 0, -1, -2, -3
SMR_NO_REPLICA = 0xFF
SMR_MAX_REPLICAS = 8
SMR_CTL_IDENTITY = 0x00FAC688


class SummersetError(RuntimeError):
    """Mirror of the reference's `SummersetError(String)` (src/utils/error.rs:7)."""

    def __init__(self, code, msg):
        super().__init__("[%d] %s" % (code, msg))
        self.code = code
        self.msg = msg


class MpCfg(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("population", C.c_uint8), ("commit_extra", C.c_uint8),
                ("straggler_ticks", C.c_uint8), ("reserved1", C.c_uint8), ("window", C.c_uint32),
                ("win_reserve", C.c_uint32), ("outbox_cap", C.c_uint32), ("commit_list_cap", C.c_uint32)]


class HbCfg(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("population", C.c_uint8), ("replica_id", C.c_uint8), ("hear_timeout_min_ms", C.c_uint64),
                ("hear_timeout_max_ms", C.c_uint64), ("send_interval_ms", C.c_uint64)]


class LeaseCfg(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("population", C.c_uint8), ("replica_id", C.c_uint8), ("expire_timeout_ms", C.c_uint64),
                ("hb_send_interval_ms", C.c_uint64)]


class MpTickIn(C.Structure):
    _fields_ = [("timeout_rep_dev", C.c_void_p), ("timeout_src_dev", C.c_void_p), ("req_target_dev", C.c_void_p),
                ("req_cnt_dev", C.c_void_p), ("req_val_dev", C.c_void_p), ("S", C.c_uint32), ("ackctl_dev", C.c_void_p),
                ("do_heartbeat", C.c_int)]


class MpImageOp(C.Structure):
    _fields_ = [("cluster", C.c_void_p), ("kind", C.c_int), ("rep", C.c_uint8), ("other", C.c_uint8), ("img_dev", C.c_void_p),
                ("img_bytes", C.c_uint64), ("copy_of_dev", C.c_void_p)]


class MpGroupState(C.Structure):
    _fields_ = [("leader", C.c_uint8), ("overflow", C.c_uint8), ("bal_prep_sent", C.c_uint64),
                ("bal_prepared", C.c_uint64), ("bal_max_seen", C.c_uint64), ("start_slot", C.c_uint32),
                ("log_len", C.c_uint32), ("accept_bar", C.c_uint32), ("commit_bar", C.c_uint32),
                ("exec_bar", C.c_uint32), ("snap_bar", C.c_uint32),
                ("peer_exec_bar", C.c_uint32 * SMR_MAX_REPLICAS)]


MP_DUMP_FIELDS = ["leader", "bal_prep_sent", "bal_prepared", "bal_max_seen", "start_slot", "log_len",
                  "accept_bar", "commit_bar", "exec_bar", "snap_bar", "peer_exec_bar", "s_bal", "s_status",
                  "s_reqs", "s_vbal", "s_vreqs", "s_flags", "s_acks", "s_packs", "s_pmax", "s_ltrig",
                  "s_lendp", "s_src", "s_rtrig", "s_rendp", "overflow"]


class MpDumpBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in MP_DUMP_FIELDS]


class QreadCfg(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("population", C.c_uint8), ("replica_id", C.c_uint8), ("n_keys", C.c_uint8),
                ("max_reads", C.c_uint8), ("n_queries", C.c_uint32)]


class QreadReplies(C.Structure):
    _fields_ = [("state", C.c_void_p), ("slot", C.c_void_p), ("val", C.c_void_p)]


class QreadLog(C.Structure):
    _fields_ = [("start_slot", C.c_void_p), ("log_end", C.c_void_p), ("status", C.c_void_p), ("token", C.c_void_p),
                ("window", C.c_uint32), ("mp_layout", C.c_uint32), ("run_lo", C.c_void_p), ("run_hi", C.c_void_p),
                ("run_leader", C.c_void_p), ("run_rep", C.c_uint32)]


class RaftCfg(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("population", C.c_uint8), ("leader_id", C.c_uint8),
                ("commit_extra", C.c_uint8), ("execute", C.c_uint8), ("window", C.c_uint32),
                ("term", C.c_uint64)]


RAFT_DUMP_FIELDS = ["role", "curr_term", "log_len", "last_commit", "last_snap", "next_slot", "try_next_slot",
                    "match_slot", "entry_term", "leader", "start_slot"]


class RaftDumpBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in RAFT_DUMP_FIELDS]


# every symbol include/summerset_hip.h declares: (name, restype, argtypes)
_vp, _u8, _u32, _u64, _i = C.c_void_p, C.c_uint8, C.c_uint32, C.c_uint64, C.c_int
class RaftAppendEntries(C.Structure):
    _fields_ = [("flags", C.c_void_p), ("leader", C.c_void_p), ("term", C.c_void_p), ("prev_slot", C.c_void_p),
                ("prev_term", C.c_void_p), ("n_entries", C.c_void_p), ("entry_term", C.c_void_p),
                ("max_entries", C.c_uint32), ("leader_commit", C.c_void_p), ("last_snap", C.c_void_p), ("entry_mask", C.c_void_p)]


class RaftAppendReply(C.Structure):
    _fields_ = [("flags", C.c_void_p), ("term", C.c_void_p), ("end_slot", C.c_void_p), ("conflict_term", C.c_void_p),
                ("conflict_slot", C.c_void_p)]


class EpCfg(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("population", C.c_uint8), ("me", C.c_uint8), ("optimized_quorum", C.c_uint8),
                ("execute", C.c_uint8), ("window", C.c_uint32), ("n_keys", C.c_uint32), ("recovery", C.c_uint32)]


class EpMsg(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("flags", "peer", "col", "ballot", "seq", "deps", "key", "row")]


class EpExpPrepare(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("flags", "peer", "row", "col", "new_ballot")]


class EpExpPrepareReply(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("flags", "voted_bal", "voted_status", "voted_seq", "voted_deps", "voted_key")]


class RspCfg(C.Structure):
    _fields_ = [("n_groups", C.c_uint32), ("population", C.c_uint8), ("me", C.c_uint8), ("fault_tolerance", C.c_uint8),
                ("reserved0", C.c_uint8), ("window", C.c_uint32)]


def _ptr_struct(name, fields):
    return type(name, (C.Structure,), {"_fields_": [(n, C.c_void_p) for n in fields]})


RspAccepts = _ptr_struct("RspAccepts", ("n", "slot", "val", "ballot"))
RspHeartbeat = _ptr_struct("RspHeartbeat", ("flags", "ballot", "commit_bar", "exec_bar", "snap_bar"))
RspPrepareReply = _ptr_struct("RspPrepareReply", ("n", "trig", "endp", "ballot", "vbal", "vval", "vmask"))
RspShards = _ptr_struct("RspShards", ("n", "slot", "bal", "val", "mask"))
RSP_DUMP_FIELDS = ("leader", "bal_prep_sent", "bal_prepared", "bal_max_seen", "len", "commit_bar", "exec_bar", "snap_bar",
                   "peer_exec_bar", "digest", "s_bal", "s_status", "s_val", "s_mask", "s_vbal", "s_vval", "s_vmask", "s_flags",
                   "s_ltrig", "s_lendp", "s_packs", "s_aacks", "s_pmax", "s_rsrc", "s_rtrig", "s_rendp", "counters")
RspDumpBufs = _ptr_struct("RspDumpBufs", RSP_DUMP_FIELDS)


EP_DUMP_FIELDS = ("len", "commit_bars", "bal", "seq", "status", "key", "deps", "pa_acks", "acc_acks", "bk", "highest_cols",
                  "counters")


class EpDumpBufs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in EP_DUMP_FIELDS]


class WireRaftMsg(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("has_conflict", C.c_uint8), ("granted", C.c_uint8), ("n_entries", C.c_uint32)] + \
               [(n, C.c_uint64) for n in ("term", "prev_slot", "prev_term", "leader_commit", "last_snap", "end_slot",
                                           "conflict_term", "conflict_slot", "last_slot", "last_term")]


class WireMsg(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("has_voted", C.c_uint8), ("slot", C.c_uint64), ("ballot", C.c_uint64),
                ("trigger_slot", C.c_uint64), ("endprep_slot", C.c_uint64), ("accept_bar", C.c_uint64),
                ("voted_ballot", C.c_uint64), ("reqs_off", C.c_uint64), ("reqs_len", C.c_uint64),
                ("commit_bar", C.c_uint64), ("exec_bar", C.c_uint64), ("snap_bar", C.c_uint64),
                ("rq_client", C.c_uint64), ("rq_req_id", C.c_uint64), ("n_replies", C.c_uint64),
                ("replies_off", C.c_uint64), ("replies_len", C.c_uint64), ("from_leader", C.c_uint8)]


class WireCodeword(C.Structure):
    _fields_ = [("num_data_shards", C.c_uint8), ("num_parity_shards", C.c_uint8), ("avail_mask", C.c_uint32),
                ("data_len", C.c_uint64), ("shard_len", C.c_uint64), ("shard_off", C.c_uint64 * 16)]


class RaftTick(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("n_new", "reply_term", "end_slot", "conflict_term", "conflict_slot", "flags", "order")]


class EpClusterOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("proposed", "col", "seq0", "deps0", "decision", "committed", "seq", "deps")]


class WireEpMsg(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("row", C.c_uint8), ("n_deps", C.c_uint32), ("col", C.c_uint64), ("ballot", C.c_uint64),
                ("seq", C.c_uint64), ("reqs_off", C.c_uint64), ("reqs_len", C.c_uint64)]


class WireRspMsg(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("has_voted", C.c_uint8), ("n_items", C.c_uint32), ("slot", C.c_uint64),
                ("ballot", C.c_uint64), ("trigger_slot", C.c_uint64), ("endprep_slot", C.c_uint64), ("voted_ballot", C.c_uint64),
                ("commit_bar", C.c_uint64), ("exec_bar", C.c_uint64), ("snap_bar", C.c_uint64)]


SYMBOLS = [
    ("smr_last_error", C.c_char_p, []),
    ("smr_device_count", _i, []),
    ("smr_abi_version", _u32, []),
    ("smr_rs_matrix", _i, [_i, _i, _vp]),
    ("smr_rs_shard_len", _u64, [_u64, _i]),
    ("smr_rs_encode", _i, [_vp, _u64, _u64, _u64, _i, _i, _vp, _u64, _u64, _vp]),
    ("smr_rs_encode_lut", _i, [_vp, _u64, _u64, _u64, _i, _i, _vp, _u64, _u64, _vp]),
    ("smr_rs_from_data_encode", _i, [_vp, _u64, _u64, _u64, _i, _i, _vp, _u64, _vp]),
    ("smr_rs_from_data_encode_fanout", _i, [_vp, _u64, _u64, _u64, _i, _i, _vp, _u64, _vp, _u64, _u64, _u32, _vp]),
    ("smr_rs_from_data_encode_stores", _i, [_vp, _u64, _u64, _u64, _i, _i, _vp, _u64, _u64, _vp]),
    ("smr_rs_from_data_encode_scatter", _i, [_vp, _u64, _u64, _u64, _i, _i, _vp, _u64, _vp, _u64, _vp]),
    ("smr_rs_reconstruct", _i, [_vp, _u64, _u64, _u64, _u64, _i, _i, _u32, _i, _vp]),
    ("smr_rs_verify", _i, [_vp, _u64, _u64, _u64, _u64, _i, _i, _vp, _vp]),
    ("smr_mp_cluster_create", _i, [C.POINTER(MpCfg), C.POINTER(_vp)]),
    ("smr_mp_cluster_destroy", None, [_vp]),
    ("smr_mp_preset_leader", _i, [_vp, _u8]),
    ("smr_mp_tick", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp, _i, _vp]),
    ("smr_mp_run_ticks", _i, [_vp, C.POINTER(MpTickIn), _u32, _vp]),
    ("smr_mp_set_live", _i, [_vp, _u32]),
    ("smr_mp_set_role_rotation", _i, [_vp, _i]),
    ("smr_mp_image_bytes", C.c_int64, [_vp, _i, _u32, _u32]),
    ("smr_mp_image_pack", _i, [_vp, _i, _u8, _u8, _vp, _u64, _u32, _u32, _vp]),
    ("smr_mp_image_unpack", _i, [_vp, _i, _u8, _u8, _vp, _u64, _u32, _u32, _vp]),
    ("smr_mp_image_plan_create", _i, [C.POINTER(MpImageOp), _u32, _u32, _u32, C.POINTER(_vp)]),
    ("smr_mp_image_plan_destroy", None, [_vp]),
    ("smr_mp_image_plan_run", _i, [_vp, _i, _vp]),
    ("smr_mp_round_local", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("smr_mp_round_deliver", _i, [_vp, _vp]),
    ("smr_mp_round_replies", _i, [_vp, _vp, _i, _vp]),
    ("smr_mp_round_heartbeat", _i, [_vp, _vp]),
    ("smr_mp_end_tick", _i, [_vp]),
    ("smr_mp_spread_create", _i, [_vp, _u32, _vp, _vp, _vp]),
    ("smr_mp_spread_destroy", None, [_vp]),
    ("smr_mp_spread_segment", _i, [_vp, _i, _vp, _i, _vp]),
    ("smr_mp_spread_set_concurrent", _i, [_vp, _i]),
    ("smr_mp_spread_abort_tick", _i, [_vp]),
    ("smr_mp_ack_matrix", _i, [_vp, _u8, C.POINTER(_vp), C.POINTER(_u64)]),
    ("smr_mp_deliver_acks", _i, [_vp, _u8, _vp, _u64, _vp, _vp]),
    ("smr_mp_deliver_acks_conn", _i, [_vp, _u8, _vp, _u64, _vp, _vp, _vp, _vp, _u32, _vp, _vp]),
    ("smr_mp_collect_acks", _i, [_vp, _u8, _vp, _u64, _vp, _vp]),
    ("smr_mp_clear_acks", _i, [_vp, _u8, _vp]),
    ("smr_mp_read_group_state", _i, [_vp, _u32, _u8, C.POINTER(MpGroupState)]),
    ("smr_mp_dump", _i, [_vp, _u8, C.POINTER(MpDumpBufs)]),
    ("smr_mp_dump_range", _i, [_vp, C.c_uint8, _u32, _u32, C.POINTER(MpDumpBufs)]),
    ("smr_mp_counters", _i, [_vp, _u8, C.POINTER(_u64 * 3)]),
    ("smr_mp_debug_generic_units", _i, [_vp, _u8, C.POINTER(_u64)]),
    ("smr_mp_debug_folded_batches", _i, [_vp, _u8, C.POINTER(_u64)]),
    ("smr_mp_debug_stamps", _i, [_vp, _vp]),
    ("smr_comm_unique_id", _i, [_vp, _u64]),
    ("smr_comm_init_rank", _i, [_vp, _u64, _u32, _u32, C.POINTER(_vp)]),
    ("smr_comm_destroy", None, [_vp]),
    ("smr_comm_exchange", _i, [_vp, _vp, C.POINTER(_u64), _vp, C.POINTER(_u64), _u32, _vp]),
    ("smr_comm_all_reduce_u64", _i, [_vp, _vp, _u64, _i, _vp]),
    ("smr_comm_info", _i, [_vp, C.POINTER(_u64 * 5)]),
    ("smr_mp_spread_bind_comm", _i, [_vp, _vp, C.POINTER(_vp * 3), C.POINTER(_u64), C.POINTER(_vp * 3), C.POINTER(_u64), _u32]),
    ("smr_mp_spread_tick", _i, [_vp, _vp, _i, _vp]),
    ("smr_mp_straggler_stats", _i, [_vp, C.POINTER(_u64 * 2)]),
    ("smr_mp_poll_commits", _i, [_vp, _u8, _vp, _vp, _u64, C.POINTER(_u64)]),
    ("smr_mp_profile_enable", _i, [_vp, _i]),
    ("smr_mp_profile_read", _i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_u64)]),
    ("smr_raft_leader_create", _i, [C.POINTER(RaftCfg), C.POINTER(_vp)]),
    ("smr_raft_leader_destroy", None, [_vp]),
    ("smr_raft_leader_append", _i, [_vp, _vp, _vp]),
    ("smr_raft_leader_handle_replies", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_raft_leader_handle_wire_replies", _i, [_vp, _vp, _u64, _vp, _vp, _u32, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    ("smr_raft_leader_run_ticks", _i, [_vp, _vp, _u32, _vp]),
    ("smr_raft_leader_dump", _i, [_vp, C.POINTER(RaftDumpBufs)]),
    ("smr_raft_leader_total_commits", _i, [_vp, C.POINTER(_u64)]),
    ("smr_raft_replica_preset", _i, [_vp, _u8, _u8, _u64, _u8]),
    ("smr_raft_replica_handle_append_entries", _i, [_vp, C.POINTER(RaftAppendEntries), C.POINTER(RaftAppendReply), _vp]),
    ("smr_raft_replica_become_candidate", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_raft_replica_handle_request_vote", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_raft_replica_handle_vote_replies", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_raft_replica_dump_votes", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("smr_raft_leader_append_emit", _i, [_vp, _vp, _vp, _vp]),
    ("smr_raft_leader_gather_entries", _i, [_vp, _vp, C.POINTER(RaftAppendEntries), _vp]),
    ("smr_raft_cluster_replicate", _i, [_vp, _u32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(RaftAppendEntries), C.POINTER(RaftAppendReply), _vp]),
    ("smr_raft_cluster_tick", _i, [_vp, _vp, _vp, _u32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(RaftAppendEntries), C.POINTER(RaftAppendReply), _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_raft_craft_enable", _i, [_vp, _u8, _u8]),
    ("smr_raft_craft_bcast_heartbeats", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_raft_craft_switch_assignment_mode", _i, [_vp, _vp, _vp]),
    ("smr_raft_craft_assignment", _i, [_vp, _vp, _vp, _vp]),
    ("smr_raft_craft_dump", _i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_raft_craft_handle_reconstruct", _i, [_vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp]),
    ("smr_raft_craft_dump_masks", _i, [_vp, _vp, _vp]),
    ("smr_raft_ring_guard_hits", _i, [_vp, C.POINTER(_u64)]),
    ("smr_raft_craft_poll_reconstructs", _i, [_vp, _u32, _vp, _vp, _vp, _vp]),
    ("smr_raft_craft_handle_reconstruct_reply", _i, [_vp, _vp, _vp, _vp, _vp, _u32, _vp]),
    ("smr_ep_replica_create", _i, [C.POINTER(EpCfg), C.POINTER(_vp)]),
    ("smr_ep_replica_destroy", None, [_vp]),
    ("smr_ep_propose", _i, [_vp, _vp, _vp, C.POINTER(EpMsg), _vp]),
    ("smr_ep_handle_pre_accept", _i, [_vp, C.POINTER(EpMsg), C.POINTER(EpMsg), _vp]),
    ("smr_ep_handle_accept", _i, [_vp, C.POINTER(EpMsg), C.POINTER(EpMsg), _vp]),
    ("smr_ep_handle_commit_notice", _i, [_vp, C.POINTER(EpMsg), _vp]),
    ("smr_ep_handle_pre_accept_replies", _i, [_vp] + [_vp] * 11),
    ("smr_ep_handle_accept_replies", _i, [_vp] + [_vp] * 6),
    ("smr_ep_handle_pre_accept_replies_at", _i, [_vp] + [_vp] * 12),
    ("smr_ep_handle_accept_replies_at", _i, [_vp] + [_vp] * 7),
    ("smr_ep_heartbeat_timeout", _i, [_vp] + [_vp] * 6),
    ("smr_ep_handle_exp_prepare", _i, [_vp, C.POINTER(EpExpPrepare), C.POINTER(EpExpPrepareReply), _vp]),
    ("smr_ep_handle_exp_prepare_replies", _i, [_vp, _vp, _vp, _vp, C.POINTER(EpExpPrepareReply)] + [_vp] * 7),
    ("smr_ep_xp_dump", _i, [_vp] + [_vp] * 9),
    ("smr_ep_cluster_create", _i, [_vp, _u32, _vp]),
    ("smr_ep_cluster_destroy", None, [_vp]),
    ("smr_ep_cluster_tick", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("smr_ep_cluster_set_mode", _i, [_vp, _u32]),
    ("smr_ep_cluster_batch_stats", _i, [_vp, _vp]),
    ("smr_ep_spread_create", _i, [_vp, _vp, _vp, _u32, _vp, _u32, _u32, _u8, _i, _vp]),
    ("smr_ep_spread_destroy", None, [_vp]),
    ("smr_ep_spread_n_exchanges", _i, [_vp]),
    ("smr_ep_spread_buffers", _i, [_vp, _u32, _vp, _vp, _vp, _vp]),
    ("smr_ep_spread_bind_comm", _i, [_vp, _vp]),
    ("smr_ep_spread_segment", _i, [_vp, _u32, _vp, _vp, _vp, _vp]),
    ("smr_ep_spread_abort_tick", _i, [_vp]),
    ("smr_ep_spread_tick", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("smr_ep_spread_info", _i, [_vp, _vp]),
    ("smr_rsp_spread_create", _i, [_vp, _vp, _vp, _u32, _vp, _u32, _u32, _u8, _u32, _u64, _vp]),
    ("smr_rsp_spread_destroy", None, [_vp]),
    ("smr_rsp_spread_buffers", _i, [_vp, _u32, _vp, _vp, _vp, _vp]),
    ("smr_rsp_spread_bind_comm", _i, [_vp, _vp]),
    ("smr_rsp_spread_segment", _i, [_vp, _u32, _vp, _vp, _vp, _i, _vp, _vp]),
    ("smr_rsp_spread_abort_tick", _i, [_vp]),
    ("smr_rsp_spread_tick", _i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    ("smr_rsp_spread_info", _i, [_vp, _vp]),
    ("smr_ep_dump", _i, [_vp, C.POINTER(EpDumpBufs)]),
    ("smr_ep_exec_dump", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("smr_ep_exec_poll", _i, [_vp, _vp, _vp, _vp, _u64, C.POINTER(_u64)]),
    ("smr_rsp_replica_create", _i, [C.POINTER(RspCfg), C.POINTER(_vp)]),
    ("smr_rsp_replica_destroy", None, [_vp]),
    ("smr_rsp_cluster_create", _i, [_vp, _u32, _vp]),
    ("smr_rsp_cluster_destroy", None, [_vp]),
    ("smr_rsp_cluster_steady_tick", _i, [_vp, _u8, _vp, _vp, _i, _vp, _vp]),
    ("smr_rsp_preset_leader", _i, [_vp, _u8]),
    ("smr_rsp_req_batch", _i, [_vp, _vp, C.POINTER(RspAccepts), _vp]),
    ("smr_rsp_handle_accept", _i, [_vp] + [_vp] * 9),
    ("smr_rsp_handle_accept_replies", _i, [_vp] + [_vp] * 6),
    ("smr_rsp_become_leader", _i, [_vp, _vp, C.POINTER(RspHeartbeat), _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_rsp_handle_prepare", _i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(RspPrepareReply), _vp]),
    ("smr_rsp_handle_prepare_replies", _i, [_vp, _vp, C.POINTER(RspPrepareReply), C.POINTER(RspAccepts), _vp]),
    ("smr_rsp_handle_reconstruct", _i, [_vp, _vp, _vp, _vp, C.POINTER(RspShards), _vp]),
    ("smr_rsp_handle_reconstruct_reply", _i, [_vp, _vp, C.POINTER(RspShards), _vp]),
    ("smr_rsp_handle_heartbeat", _i, [_vp, _vp, C.POINTER(RspHeartbeat), _vp, C.POINTER(RspHeartbeat), _vp]),
    ("smr_rsp_bcast_heartbeat", _i, [_vp, _vp, C.POINTER(RspHeartbeat), _vp]),
    ("smr_rsp_dump", _i, [_vp, C.POINTER(RspDumpBufs)]),
    ("smr_rsp_exec_poll", _i, [_vp, _vp, _vp, _vp, _u64, C.POINTER(_u64)]),
    ("smr_rsp_pstore_create", _i, [_u32, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    ("smr_rsp_pstore_destroy", None, [_vp]),
    ("smr_rsp_pstore_put", _i, [_vp, _vp, _vp, _vp, _vp, _u64, _vp, _u32, _vp]),
    ("smr_rsp_pstore_follow", _i, [_vp, _vp, _u32, C.POINTER(_vp), _vp, _vp, _vp]),
    ("smr_rsp_pstore_follow_many", _i, [_u32, C.POINTER(_vp), C.POINTER(_vp), _vp, _i, _vp]),
    ("smr_rsp_pstore_put_follow_all", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _u32, _u32, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    ("smr_rsp_pstore_get_data", _i, [_vp, _u32, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp]),
    ("smr_rsp_pstore_extract", _i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_rsp_pstore_ingest", _i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_rsp_pstore_emit_accepts", _i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp]),
    ("smr_rsp_pstore_dump", _i, [_vp, _i, _vp, _vp, _vp]),
    ("smr_rsp_pstore_read_row", _i, [_vp, _i, _u32, _vp]),
    ("smr_rsp_pstore_layout", _i, [_vp, _i, C.POINTER(_vp), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    ("smr_rsp_pstore_voted_alias", _i, [_vp, C.POINTER(_vp), _vp]),
    ("smr_rsp_pstore_counters", _i, [_vp, _vp]),
    ("smr_rsp_pstore_debug_delivered", _i, [_vp, _vp]),
    ("smr_craft_pstore_create", _i, [_u32, _u32, _u32, _u32, _u32, C.POINTER(_vp)]),
    ("smr_craft_pstore_put", _i, [_vp, _vp, _vp, _vp, _u64, _vp, _u32, _vp]),
    ("smr_craft_pstore_follow", _i, [_vp, _vp, _u32, _vp, _vp, _vp]),
    ("smr_craft_pstore_follow_many", _i, [_u32, _vp, _vp, _vp, _vp]),
    ("smr_craft_pstore_put_follow_all", _i, [_vp, _vp, _vp, _vp, _u64, _vp, _u32, _u32, _vp, _vp, _vp]),
    ("smr_wire_reqbatch", C.c_int64, [C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64]),
    ("smr_wire_prepare", C.c_int64, [_u64, _u64, _vp, _u64]),
    ("smr_wire_prepare_reply", C.c_int64, [_u64, _u64, _u64, _u64, _i, _u64, _vp, _u64, _u64, _vp, _u64]),
    ("smr_wire_accept", C.c_int64, [_u64, _u64, _vp, _u64, _vp, _u64]),
    ("smr_wire_accept_reply", C.c_int64, [_u64, _u64, _vp, _u64]),
    ("smr_wal_prepare_bal", C.c_int64, [_u64, _u64, _vp, _u64]),
    ("smr_wal_accept_data", C.c_int64, [_u64, _u64, _vp, _u64, _vp, _u64]),
    ("smr_wal_commit_slot", C.c_int64, [_u64, _vp, _u64]),
    ("smr_wire_decode", C.c_int64, [_vp, _u64, C.POINTER(WireMsg)]),
    ("smr_wire_read_query", C.c_int64, [_vp, _u64, _vp, _u64]),
    ("smr_wire_read_query_reply", C.c_int64, [_u64, _u64, _u32, _vp, _vp, _vp, _vp, _i, _vp, _u64]),
    ("smr_wire_heartbeat", C.c_int64, [_u64, _u64, _u64, _u64, _vp, _u64]),
    ("smr_wire_commit_notice", C.c_int64, [_u64, _u64, _vp, _u64]),
    ("smr_wire_read_query_replies", C.c_int64, [_vp, _u64, _u32, _vp, _vp, _vp, _vp]),
    ("smr_wire_raft_append_entries", C.c_int64, [_u64, _u64, _u64, C.c_uint32, _vp, _vp, _vp, _vp, _u64, _u64, _vp, _u64]),
    ("smr_wire_raft_append_entries_reply", C.c_int64, [_u64, _u64, _i, _u64, _u64, _vp, _u64]),
    ("smr_wire_raft_request_vote", C.c_int64, [_u64, _u64, _u64, _vp, _u64]),
    ("smr_wire_raft_request_vote_reply", C.c_int64, [_u64, _i, _vp, _u64]),
    ("smr_wal_raft_metadata", C.c_int64, [_u64, _u8, _vp, _u64]),
    ("smr_wire_raft_decode", C.c_int64, [_vp, _u64, C.POINTER(WireRaftMsg), _vp, C.c_uint32]),
    ("smr_wire_rscodeword", C.c_int64, [_u8, _u8, _u64, _u64, _u32, _vp, _u64, _vp, _u64]),
    ("smr_wire_rsp_prepare", C.c_int64, [_u64, _u64, _vp, _u64]),
    ("smr_wire_rsp_prepare_reply", C.c_int64, [_u64, _u64, _u64, _u64, _i, _u64, _vp, _u64, _vp, _u64]),
    ("smr_wire_rsp_accept", C.c_int64, [_u64, _u64, _vp, _u64, _vp, _u64]),
    ("smr_wire_rsp_accept_reply", C.c_int64, [_u64, _u64, _vp, _u64]),
    ("smr_wire_rsp_reconstruct", C.c_int64, [_u32, _vp, _vp, _u64]),
    ("smr_wire_rsp_reconstruct_reply", C.c_int64, [_u32, _vp, _vp, _vp, _vp, _vp, _u64]),
    ("smr_wire_rsp_heartbeat", C.c_int64, [_u64, _u64, _u64, _u64, _vp, _u64]),
    ("smr_wal_rsp_accept_data", C.c_int64, [_u64, _u64, _vp, _u64, _vp, _u64]),
    ("smr_wire_rsp_decode", C.c_int64, [_vp, _u64, C.POINTER(WireRspMsg), _vp, _vp, _vp, _u32]),
    ("smr_wire_ep_msg", C.c_int64, [_u8, _u8, _u64, _u64, _u64, _vp, _u32, _vp, _u64, _vp, _u64]),
    ("smr_wal_ep_slot", C.c_int64, [_u8, _u8, _u64, _u64, _u64, _vp, _u32, _vp, _u64, _vp, _u64]),
    ("smr_wire_ep_decode", C.c_int64, [_vp, _u64, C.POINTER(WireEpMsg), _vp, _u32]),
    ("smr_wire_ingest_scratch_bytes", _u64, [_u32]),
    ("smr_wire_ingest_mp", _i, [_vp, _u64, _vp, _vp, _vp, _u32, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _vp, _vp, _vp]),
    ("smr_wire_ingest_mp_conn", _i, [_vp, _u64, _vp, _vp, _vp, _u32, _vp, _u64, _vp, _u32, _vp, _u32, _vp, _vp, _vp, _vp]),
    ("smr_wire_ingest_raft_replies", _i, [_vp, _u64, _vp, _vp, _vp, _vp, _u32, _u32, _u8, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    ("smr_wire_ingest_rsp_accept_replies", _i, [_vp, _u64, _vp, _vp, _vp, _vp, _u32, _u32, _u8, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _vp]),
    ("smr_wire_emit_mp_accept_replies", _i, [_vp, _u64, _vp, _vp, _vp]),
    ("smr_wire_emit_raft_replies", _i, [_vp, _vp, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    ("smr_wire_emit_ep_pre_accept_replies", _i, [_vp, _u8, _vp, _vp, _vp, _vp, _u32, _u8, _vp, _vp, _vp]),
    ("smr_wire_ingest_ep_pre_accept_replies", _i, [_vp, _u64, _vp, _vp, _vp, _vp, _u32, _u32, _u8, _u8, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp,
                                                   _vp]),
    ("smr_batcher_create", _i, [_u32, _u32, C.POINTER(_vp)]),
    ("smr_batcher_destroy", None, [_vp]),
    ("smr_batcher_submit", _i, [_vp, _u32, _u64, _u64, _u8, C.c_char_p, _u32, C.c_char_p, _u32]),
    ("smr_batcher_pending", _i, [_vp, C.POINTER(_u64)]),
    ("smr_batcher_tick", C.c_int64, [_vp, _vp, _vp, _vp, _u32, _vp, _u64]),
    ("smr_mp_replica_log_view", _i, [_vp, _u8, C.POINTER(QreadLog)]),
    ("smr_kv_create", _i, [_u32, _u32, C.POINTER(_vp)]),
    ("smr_kv_destroy", None, [_vp]),
    ("smr_kv_execute", _i, [_vp, _u32, _vp, _vp, _vp, _vp, _vp]),
    ("smr_kv_table", _i, [_vp, C.POINTER(_vp)]),
    ("smr_kv_dump", _i, [_vp, _vp]),
    ("smr_qread_create", _i, [C.POINTER(QreadCfg), C.POINTER(_vp)]),
    ("smr_qread_destroy", None, [_vp]),
    ("smr_qread_refresh_highest_slot", _i, [_vp, _vp, _vp, _vp]),
    ("smr_qread_handle_read_query", _i, [_vp, _vp, _vp, _vp, _vp, C.POINTER(QreadLog), C.POINTER(QreadReplies), _vp, _vp]),
    ("smr_qread_issue", _i, [_vp, _u32, _vp, C.POINTER(QreadReplies), _vp]),
    ("smr_qread_handle_replies", _i, [_vp, _u32, C.POINTER(QreadReplies), _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_qread_dump", _i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_skv_create", _i, [_u32, _u32, _u64, C.POINTER(_vp)]),
    ("smr_skv_destroy", None, [_vp]),
    ("smr_skv_execute", _i, [_vp, _u32, _vp, _vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("smr_skv_heap", _i, [_vp, C.POINTER(_vp), C.POINTER(_u64)]),
    ("smr_skv_read", _i, [_vp, _u32, _u32, _u32, _vp]),
    ("smr_skv_stats", _i, [_vp, _vp, _vp, _vp]),
    ("smr_hb_create", _i, [C.POINTER(HbCfg), _u64, C.POINTER(_vp)]),
    ("smr_hb_destroy", None, [_vp]),
    ("smr_hb_set_sending", _i, [_vp, _vp, _vp]),
    ("smr_hb_kickoff_hear_timer", _i, [_vp, _vp, _u64, _vp, _vp]),
    ("smr_hb_poll", _i, [_vp, _u64, _vp, _vp, _vp]),
    ("smr_hb_clear_reply_cnts", _i, [_vp, _vp, _vp]),
    ("smr_hb_update_bcast_cnts", _i, [_vp, _vp, _vp, _vp]),
    ("smr_hb_update_heard_cnt", _i, [_vp, _vp, _vp]),
    ("smr_hb_dump", _i, [_vp] + [_vp] * 8),
    ("smr_wallog_create", _i, [C.POINTER(_vp)]),
    ("smr_wallog_destroy", None, [_vp]),
    ("smr_wallog_len", C.c_int64, [_vp]),
    ("smr_wallog_bytes", C.c_int64, [_vp, _vp, _u64]),
    ("smr_wallog_write", _i, [_vp, _u64, C.c_char_p, _u64, _u64, C.POINTER(_u8), C.POINTER(_u64)]),
    ("smr_wallog_append", _i, [_vp, _u64, C.c_char_p, _u64, C.POINTER(_u64)]),
    ("smr_wallog_read", _i, [_vp, _u64, _u64, _vp, _u64, C.POINTER(C.c_int64), C.POINTER(_u64)]),
    ("smr_wallog_truncate", _i, [_vp, _u64, _u64, C.POINTER(_u8), C.POINTER(_u64)]),
    ("smr_wallog_discard", _i, [_vp, _u64, _u64, _u64, C.POINTER(_u8), C.POINTER(_u64)]),
    ("smr_lease_create", _i, [C.POINTER(LeaseCfg), C.POINTER(_vp)]),
    ("smr_lease_destroy", None, [_vp]),
    ("smr_lease_step", _i, [_vp, _u64] + [_vp] * 8),
    ("smr_lease_attempt_refresh", _i, [_vp, _u64, _vp, _vp, _vp, _vp]),
    ("smr_lease_sets", _i, [_vp, _vp, _vp, _vp, _vp]),
    ("smr_lease_dump", _i, [_vp] * 6),
    ("smr_repnothing_create", _i, [C.POINTER(_vp)]),
    ("smr_repnothing_destroy", None, [_vp]),
    ("smr_repnothing_submit_batch", _i, [_vp, C.c_uint32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    ("smr_repnothing_poll_reply", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(C.c_uint8),
                                       C.POINTER(C.c_int), _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("smr_repnothing_stats", _i, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
]

_lib = None


def load():
    """dlopen the engine; raises if it has not been built (see summerset_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SummersetError(SMR_ERR_STATE, "libsummerset_hip.so is not built: run "
                                 "`python -c 'import __graft_entry__ as g; g.build()'` "
                                 "(there is no CPU fallback)")
        lib = C.CDLL(os.environ.get("SUMMERSET_HIP_LIB", LIB_PATH))   # override: kernel-variant experiments only
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)        # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def stream_ptr(stream):
    """The hipStream_t a call is launched on: an explicit handle (int), or torch's current stream.
    One helper for every mirror class, so the binding layer has a single path (and one test)."""
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream
    return int(stream)


def check(rc):
    if rc != 0:
        raise SummersetError(rc, load().smr_last_error().decode("utf-8", "replace"))
    return rc
