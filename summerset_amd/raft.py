"""Host-side handle of the batched Raft replica (G groups, one replica id).

Mirrors `RaftReplica` (src/protocols/raft/mod.rs:237-330): the leader half --
`handle_req_batch` log append, `handle_msg_append_entries_reply`
(raft/messages.rs:222-388) -- the follower's `handle_msg_append_entries`
(messages.rs:13-218) and the election handlers (`become_a_candidate`,
`handle_msg_request_vote`, `handle_msg_request_vote_reply`).  Thin: every
method is one C-ABI call; messages are device tensors with one entry per group.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RaftAppendEntries, RaftAppendReply, RaftCfg, RaftDumpBufs, check, stream_ptr

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

    def handle_req_batch(self, n_new, stream=None):
        """append n_new[g] entries of the current term to every group's log"""
        check(self._L.smr_raft_leader_append(self._h, _ptr(n_new), stream_ptr(stream)))

    def handle_req_batch_emit(self, n_new, stream=None, out=None):
        """handle_req_batch + [R, G] first slot of the entries sent to each peer (-1 = nothing); `out`: an earlier result to refill"""
        import torch
        first = out if out is not None else torch.zeros((self.R, self.G), dtype=torch.int32, device=n_new.device)
        check(self._L.smr_raft_leader_append_emit(self._h, _ptr(n_new), _ptr(first), stream_ptr(stream)))
        return first

    def gather_entries(self, first, max_entries, stream=None, out=None):
        """the AppendEntries for one peer out of my log: first[g] = first slot to send (row of handle_req_batch_emit); `out`: an
        earlier message of the same `max_entries` to refill (the kernel writes every field of every group)"""
        import torch
        dev, G, K = first.device, self.G, int(max_entries)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=dev)
        m = out if out is not None else dict(
            flags=z(G, torch.uint8), leader=z(G, torch.uint8), term=z(G, torch.int64), prev_slot=z(G, torch.int32),
            prev_term=z(G, torch.int64), n_entries=z(G, torch.int32), entry_term=z((K, G), torch.int64),
            leader_commit=z(G, torch.int32), last_snap=z(G, torch.int32))
        msg = RaftAppendEntries(_ptr(m["flags"]), _ptr(m["leader"]), _ptr(m["term"]), _ptr(m["prev_slot"]),
                                _ptr(m["prev_term"]), _ptr(m["n_entries"]), _ptr(m["entry_term"]), K,
                                _ptr(m["leader_commit"]), _ptr(m["last_snap"]))
        check(self._L.smr_raft_leader_gather_entries(self._h, _ptr(first), C.byref(msg), stream_ptr(stream)))
        return m

    def replicate_many(self, followers, first, msgs, replies, entry_masks=None, stream=None):
        """`gather_entries(first[k], K, out=msgs[k])` + `followers[k].handle_msg_append_entries(**msgs[k], entry_mask=entry_masks[k],
        out=replies[k])` for every k in ONE launch (`smr_raft_cluster_replicate`): followers = replica objects on my device, first[k] =
        int32 [G] (a row of `handle_req_batch_emit`), msgs[k] = a message dict as `gather_entries` returns it (refilled), replies[k] =
        the reply tensors to fill (e.g. rows of my [R, G] reply arrays), entry_masks[k] = uint8 [K, G] (CRaft).  Returns msgs."""
        n = len(followers)
        hs = (C.c_void_p * n)(*[f._h for f in followers])
        fs = (C.c_void_p * n)(*[_ptr(first[k]) for k in range(n)])
        ms, rs = (RaftAppendEntries * n)(), (RaftAppendReply * n)()
        for k in range(n):
            m, r = msgs[k], replies[k]
            ms[k] = RaftAppendEntries(_ptr(m["flags"]), _ptr(m["leader"]), _ptr(m["term"]), _ptr(m["prev_slot"]), _ptr(m["prev_term"]),
                                      _ptr(m["n_entries"]), _ptr(m["entry_term"]), int(m["entry_term"].shape[0]), _ptr(m["leader_commit"]),
                                      _ptr(m["last_snap"]), _ptr(entry_masks[k]) if entry_masks is not None else None)
            rs[k] = RaftAppendReply(*[_ptr(r[x]) for x in ("flags", "term", "end_slot", "conflict_term", "conflict_slot")])
        check(self._L.smr_raft_cluster_replicate(self._h, n, hs, fs, ms, rs, stream_ptr(stream)))
        return msgs

    def cluster_tick(self, n_new, first, followers, msgs, replies, reply_term, end_slot, flags, conflict_term=None, conflict_slot=None,
                     order=None, entry_masks=None, stream=None):
        """a co-located cluster's steady tick in ONE launch (`smr_raft_cluster_tick`): `handle_req_batch_emit(n_new, out=first)` +
        `replicate_many(followers, [first[id of follower k]], msgs, replies, entry_masks)` + `handle_msg_append_entries_reply(
        reply_term, end_slot, flags, conflict_term, conflict_slot, order)`; replies[k] = follower k's rows of the [R, G] reply
        arrays.  Returns msgs."""
        n = len(followers)
        hs = (C.c_void_p * n)(*[f._h for f in followers])
        fs = (C.c_void_p * n)(*[_ptr(first[f.me]) for f in followers])
        ms, rs = (RaftAppendEntries * n)(), (RaftAppendReply * n)()
        for k in range(n):
            m, r = msgs[k], replies[k]
            ms[k] = RaftAppendEntries(_ptr(m["flags"]), _ptr(m["leader"]), _ptr(m["term"]), _ptr(m["prev_slot"]), _ptr(m["prev_term"]),
                                      _ptr(m["n_entries"]), _ptr(m["entry_term"]), int(m["entry_term"].shape[0]), _ptr(m["leader_commit"]),
                                      _ptr(m["last_snap"]), _ptr(entry_masks[k]) if entry_masks is not None else None)
            rs[k] = RaftAppendReply(*[_ptr(r[x]) for x in ("flags", "term", "end_slot", "conflict_term", "conflict_slot")])
        check(self._L.smr_raft_cluster_tick(self._h, _ptr(n_new), _ptr(first), n, hs, fs, ms, rs, _ptr(reply_term), _ptr(end_slot),
                                            _ptr(conflict_term), _ptr(conflict_slot), _ptr(flags), _ptr(order), stream_ptr(stream)))
        return msgs

    def new_message(self, max_entries, device):
        """the tensors of one AppendEntries message (what `gather_entries` allocates when it is given no `out`)"""
        import torch
        G, K = self.G, int(max_entries)
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        return dict(flags=z(G, torch.uint8), leader=z(G, torch.uint8), term=z(G, torch.int64), prev_slot=z(G, torch.int32),
                    prev_term=z(G, torch.int64), n_entries=z(G, torch.int32), entry_term=z((K, G), torch.int64),
                    leader_commit=z(G, torch.int32), last_snap=z(G, torch.int32))

    def handle_msg_append_entries_reply(self, reply_term, end_slot, flags, conflict_term=None, conflict_slot=None,
                                        order=None, stream=None):
        """one AppendEntriesReply per (peer, group); device tensors shaped [R, G]"""
        check(self._L.smr_raft_leader_handle_replies(self._h, _ptr(reply_term), _ptr(end_slot), _ptr(conflict_term),
                                                     _ptr(conflict_slot), _ptr(flags), _ptr(order),
                                                     stream_ptr(stream)))

    def run_ticks(self, ticks, stream=None):
        """a batch of ticks as ONE call (`smr_raft_leader_run_ticks`: one launch per <= 16 ticks): ticks[t] = dict(n_new=[G] or
        None, reply_term / end_slot / flags = [R, G] (or flags None: no replies), conflict_term / conflict_slot / order
        optional) -- tick t = handle_req_batch(n_new) then handle_msg_append_entries_reply(...), as the two calls would do it.
        The tensors must stay untouched until the stream's work is done (they are kept referenced until the next call)."""
        from ._lib import RaftTick
        arr = (RaftTick * len(ticks))()
        for t, x in enumerate(ticks):
            for n, _ in RaftTick._fields_:
                setattr(arr[t], n, _ptr(x.get(n)))
        check(self._L.smr_raft_leader_run_ticks(self._h, arr, len(ticks), stream_ptr(stream)))
        self._held_ticks = list(ticks)

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

    # ---- follower side and elections (device tensors, one entry per group) ----
    def preset(self, role, leader, term, voted_for=0xFF):
        check(self._L.smr_raft_replica_preset(self._h, role, leader, term, voted_for))

    def handle_msg_append_entries(self, flags, leader, term, prev_slot, prev_term, n_entries, entry_term,
                                  leader_commit, last_snap, entry_mask=None, stream=None, out=None):
        """returns the AppendEntriesReply tensors dict(flags, term, end_slot, conflict_term, conflict_slot); entry_mask
        [K, G] uint8 (CRaft replicas only): avail_shards_map of every sent entry's codeword; `out`: the reply tensors to fill (e.g.
        rows of the leader's [R, G] reply arrays: the kernel writes every group's reply)"""
        import torch
        dev, G = flags.device, self.G
        r = out if out is not None else dict(
            flags=torch.zeros(G, dtype=torch.uint8, device=dev), term=torch.zeros(G, dtype=torch.int64, device=dev),
            end_slot=torch.zeros(G, dtype=torch.int32, device=dev),
            conflict_term=torch.zeros(G, dtype=torch.int64, device=dev),
            conflict_slot=torch.zeros(G, dtype=torch.int32, device=dev))
        m = RaftAppendEntries(_ptr(flags), _ptr(leader), _ptr(term), _ptr(prev_slot), _ptr(prev_term), _ptr(n_entries),
                              _ptr(entry_term), int(entry_term.shape[0]), _ptr(leader_commit), _ptr(last_snap), _ptr(entry_mask))
        rr = RaftAppendReply(*[_ptr(r[k]) for k in ("flags", "term", "end_slot", "conflict_term", "conflict_slot")])
        check(self._L.smr_raft_replica_handle_append_entries(self._h, C.byref(m), C.byref(rr), stream_ptr(stream)))
        return r

    def become_a_candidate(self, timeout_src, stream=None):
        import torch
        dev, G = timeout_src.device, self.G
        r = dict(flags=torch.zeros(G, dtype=torch.uint8, device=dev), term=torch.zeros(G, dtype=torch.int64, device=dev),
                 last_slot=torch.zeros(G, dtype=torch.int32, device=dev),
                 last_term=torch.zeros(G, dtype=torch.int64, device=dev))
        check(self._L.smr_raft_replica_become_candidate(self._h, _ptr(timeout_src), _ptr(r["flags"]), _ptr(r["term"]),
                                                        _ptr(r["last_slot"]), _ptr(r["last_term"]),
                                                        stream_ptr(stream)))
        return r

    def handle_msg_request_vote(self, flags, candidate, term, last_slot, last_term, stream=None):
        import torch
        dev, G = flags.device, self.G
        r = dict(flags=torch.zeros(G, dtype=torch.uint8, device=dev), term=torch.zeros(G, dtype=torch.int64, device=dev))
        check(self._L.smr_raft_replica_handle_request_vote(self._h, _ptr(flags), _ptr(candidate), _ptr(term),
                                                           _ptr(last_slot), _ptr(last_term), _ptr(r["flags"]),
                                                           _ptr(r["term"]), stream_ptr(stream)))
        return r

    def handle_msg_request_vote_reply(self, term, flags, order=None, stream=None):
        """one RequestVoteReply per (peer, group), tensors [R, G]"""
        import torch
        dev, G, R = flags.device, self.G, self.R
        r = dict(hb_prev_slot=torch.zeros((R, G), dtype=torch.int32, device=dev),
                 elected=torch.zeros(G, dtype=torch.uint8, device=dev))
        check(self._L.smr_raft_replica_handle_vote_replies(self._h, _ptr(term), _ptr(flags), _ptr(order),
                                                           _ptr(r["hb_prev_slot"]), _ptr(r["elected"]),
                                                           stream_ptr(stream)))
        return r

    def ring_guard_hits(self):
        """entries of AppendEntries messages the follower path skipped because they had left the W-entry term ring
        (`smr_raft_ring_guard_hits`: a harness rule, not the reference's; size `window` so that it stays 0)"""
        n = C.c_uint64()
        check(self._L.smr_raft_ring_guard_hits(self._h, C.byref(n)))
        return int(n.value)

    def dump_votes(self):
        G = self.G
        r = dict(voted_for=np.zeros(G, np.uint8), votes=np.zeros(G, np.uint8), n_exec=np.zeros(G, np.uint32),
                 n_trunc=np.zeros(G, np.uint32))
        check(self._L.smr_raft_replica_dump_votes(self._h, *[r[k].ctypes.data_as(C.c_void_p)
                                                              for k in ("voted_for", "votes", "n_exec", "n_trunc")]))
        return r


class CRaftLeaderGroup(RaftLeaderGroup):
    """The leader of `CRaftReplica` (src/protocols/craft/, a fork of raft/): Raft's match-index quorum with the
    `majority + fault_tolerance` commit rule (craft/messages.rs:301-313), the full-copy fall-back mode
    (`switch_assignment_mode`, craft/leadership.rs:80-141) that `bcast_heartbeats` enters when the Heartbeater's reply
    counters (server/heartbeat.rs:240-296) say `fault_tolerance` or more peers are gone (craft/leadership.rs:283-288),
    and the shard assignment of a new entry's RS codeword (craft/request.rs:71-100; the shards themselves:
    `rscoding.RSCodeword`).  Leader side: every entry of the log was created by this leader.  Follower side
    (`preset(role=0, ...)`): `handle_msg_append_entries` follows the fork's handler (craft/messages.rs:14-254) with the
    entries' shard bitmaps, `handle_msg_reconstruct` answers a Reconstruct (craft/messages.rs:622-663)."""

    def __init__(self, n_groups, population=5, leader_id=0, window=64, term=1, fault_tolerance=1, repeat_threshold=3):
        super().__init__(n_groups, population, leader_id, window, term, commit_extra=0)
        self.fault_tolerance = int(fault_tolerance)
        check(self._L.smr_raft_craft_enable(self._h, self.fault_tolerance, int(repeat_threshold)))

    def bcast_heartbeats(self, device, stream=None):
        """the send tick: {hb_flags, prev_slot, prev_term} [R, G], {leader_commit, last_snap} [G]"""
        import torch
        R, G = self.R, self.G
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        m = dict(hb_flags=z((R, G), torch.uint8), prev_slot=z((R, G), torch.int32), prev_term=z((R, G), torch.int64),
                 leader_commit=z(G, torch.int32), last_snap=z(G, torch.int32))
        check(self._L.smr_raft_craft_bcast_heartbeats(self._h, _ptr(m["hb_flags"]), _ptr(m["prev_slot"]), _ptr(m["prev_term"]),
                                                      _ptr(m["leader_commit"]), _ptr(m["last_snap"]), stream_ptr(stream)))
        return m

    def switch_assignment_mode(self, to_full_copy, stream=None):
        """to_full_copy[g] (uint8): 0 / 1, anything else = no call for that group"""
        check(self._L.smr_raft_craft_switch_assignment_mode(self._h, _ptr(to_full_copy), stream_ptr(stream)))

    def assignment(self, device, stream=None):
        """shard masks of a new entry: persist [G] (the leader's WAL entry), send [R, G] (AppendEntries per peer)"""
        import torch
        persist = torch.zeros(self.G, dtype=torch.int32, device=device)
        send = torch.zeros((self.R, self.G), dtype=torch.int32, device=device)
        check(self._L.smr_raft_craft_assignment(self._h, _ptr(persist), _ptr(send), stream_ptr(stream)))
        return persist, send

    def dump_craft(self):
        R, G = self.R, self.G
        out = dict(full_copy_mode=np.zeros(G, np.uint8), peer_alive=np.zeros(G, np.uint8), hb_replied=np.zeros((R, G), np.uint64),
                   hb_seen=np.zeros((R, G), np.uint64), hb_repeat=np.zeros((R, G), np.uint8))
        check(self._L.smr_raft_craft_dump(self._h, *[out[k].ctypes.data_as(C.c_void_p) for k in
                                                     ("full_copy_mode", "peer_alive", "hb_replied", "hb_seen", "hb_repeat")]))
        return out


def _craft_follower_methods():
    def handle_msg_reconstruct(self, n, slot, term, stream=None):
        """Reconstruct { slots }: n [G], slot [K, G] int32, term [K, G] int64 -> dict(n [G], has [K, G], mask [K, G])"""
        import torch
        dev, G, K = n.device, self.G, int(slot.shape[0])
        r = dict(n=torch.zeros(G, dtype=torch.int32, device=dev), has=torch.zeros((K, G), dtype=torch.uint8, device=dev),
                 mask=torch.zeros((K, G), dtype=torch.uint8, device=dev))
        check(self._L.smr_raft_craft_handle_reconstruct(self._h, _ptr(n), _ptr(slot), _ptr(term), K, _ptr(r["n"]), _ptr(r["has"]),
                                                        _ptr(r["mask"]), stream_ptr(stream)))
        return r

    def dump_masks(self):
        d = dict(mask=np.zeros((self.W, self.G), np.uint8), counters=np.zeros(2, np.uint64))
        check(self._L.smr_raft_craft_dump_masks(self._h, d["mask"].ctypes.data_as(C.c_void_p), d["counters"].ctypes.data_as(C.c_void_p)))
        return d
    def poll_reconstructs(self, device, max_slots=16, stream=None):
        """the Reconstruct { slots } the reply handler queued since the last poll: dict(n [G], slot / term [K, G])"""
        import torch
        G, K = self.G, int(max_slots)
        r = dict(n=torch.zeros(G, dtype=torch.int32, device=device), slot=torch.zeros((K, G), dtype=torch.int32, device=device),
                 term=torch.zeros((K, G), dtype=torch.int64, device=device))
        check(self._L.smr_raft_craft_poll_reconstructs(self._h, K, _ptr(r["n"]), _ptr(r["slot"]), _ptr(r["term"]), stream_ptr(stream)))
        return r

    def handle_msg_reconstruct_reply(self, peer, n, slot, mask, stream=None):
        check(self._L.smr_raft_craft_handle_reconstruct_reply(self._h, _ptr(peer), _ptr(n), _ptr(slot), _ptr(mask), int(slot.shape[0]),
                                                              stream_ptr(stream)))
    CRaftLeaderGroup.poll_reconstructs = poll_reconstructs
    CRaftLeaderGroup.handle_msg_reconstruct_reply = handle_msg_reconstruct_reply
    CRaftLeaderGroup.handle_msg_reconstruct = handle_msg_reconstruct
    CRaftLeaderGroup.dump_masks = dump_masks


_craft_follower_methods()
