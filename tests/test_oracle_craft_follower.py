"""The CRaft follower of oracle/raft_oracle.c (orc_craft_handle_append_entries, orc_craft_handle_reconstruct) against traces
worked out by hand from craft/messages.rs:14-254 and :622-663, R = 5: majority 3, data shards 0-2, parity shards 3-4.
An entry's codeword is its shard availability bitmap."""
import numpy as np

G = 2
ALL = 0b11111


def _follower(oracle, W=16):
    o = oracle.CRaftOracle(G, R=5, W=W, leader_id=2, term=1, fault_tolerance=1)
    o.preset(0, 0, 1)                                            # follower of replica 0 in term 1
    return o


def _ae(o, prev_slot, prev_term, terms, masks, leader_commit=0, term=1, leader=0, last_snap=0):
    K = max(len(terms), 1)
    et = np.zeros((K, G), np.uint64); em = np.zeros((K, G), np.uint8)
    for k, (t, m) in enumerate(zip(terms, masks)):
        et[k] = t; em[k] = m
    f = lambda v, dt: np.full(G, v, dt)
    return o.handle_append_entries(f(1, np.uint8), f(leader, np.uint8), f(term, np.uint64), f(prev_slot, np.uint32), f(prev_term, np.uint64),
                                   f(len(terms), np.uint32), et, f(leader_commit, np.uint32), f(last_snap, np.uint32), entry_mask=em)


def _masks(o, n):
    return [int(o.dump_masks()["mask"][s, 0]) for s in range(n)]


def test_one_shard_per_entry_is_stored_and_not_executed(oracle):
    o = _follower(oracle)
    r = _ae(o, 0, 0, [1, 1], [0b00100, 0b00100], leader_commit=2)     # my shard (2) of two entries; the leader has committed both
    assert int(r["flags"][0]) == 1 and int(r["end_slot"][0]) == 2
    d = o.dump()
    # entries.len() is 0 after the drain (the raft quirk the fork keeps): new_commit = min(2, prev_slot + 0) = 0
    assert (d["log_len"] == 3).all() and (d["last_commit"] == 0).all()
    assert _masks(o, 3) == [ALL, 0b00100, 0b00100]                     # slot 0: the dummy entry
    # a heartbeat later carries the commit index: 1 shard < majority -> postponed (:197-208), last_commit stays
    r = _ae(o, 2, 1, [], [], leader_commit=2)
    assert int(r["flags"][0]) == 1 and int(r["end_slot"][0]) == 2
    assert (o.dump()["last_commit"] == 0).all() and list(o.dump_masks()["counters"]) == [0, G]
    assert (o.dump_votes()["n_exec"] == 0).all()


def test_resent_entries_are_absorbed_and_executed(oracle):
    o = _follower(oracle)
    _ae(o, 0, 0, [1, 1], [0b00100, 0b00100])
    # the leader fell back to full-copy mode and re-sends from slot 1 with the data shards 0-2 (craft/messages.rs:420-460),
    # plus a new slot 3: same terms -> no conflict; my bitmaps differ and hold < 3 data shards -> absorb (:133-146); the walk
    # stops at slot 3 (beyond my log): first_new = 3, one entry appended
    r = _ae(o, 0, 0, [1, 1, 1], [0b00111, 0b00011, 0b00111], leader_commit=3)
    assert int(r["flags"][0]) == 1 and int(r["end_slot"][0]) == 3
    assert _masks(o, 4) == [ALL, 0b00111, 0b00111, 0b00111]
    # entries.len() is 2 after the drain: new_commit = min(3, 0 + 2) = 2; both hold 3 data shards: executed as they are
    assert (o.dump()["last_commit"] == 2).all() and (o.dump_votes()["n_exec"] == 2).all()
    assert list(o.dump_masks()["counters"]) == [0, 0]


def test_a_message_that_brings_nothing_new_is_appended_again(oracle):
    """the quirk the fork inherits from raft/messages.rs:143-167: when every sent entry is already there the walk never
    breaks, first_new stays prev_slot + 1 and ALL entries are drained and pushed (their shards were absorbed first)"""
    o = _follower(oracle)
    _ae(o, 0, 0, [1, 1], [0b00100, 0b00100])
    r = _ae(o, 0, 0, [1, 1], [0b00111, 0b00011], leader_commit=2)
    assert int(r["flags"][0]) == 1 and int(r["end_slot"][0]) == 2
    assert (o.dump()["log_len"] == 5).all() and _masks(o, 5) == [ALL, 0b00111, 0b00111, 0b00111, 0b00011]
    assert (o.dump()["last_commit"] == 0).all()                        # entries.len() == 0 after the drain: min(2, 0 + 0)


def test_parity_shards_need_reconstruct_data(oracle):
    o = _follower(oracle)
    _ae(o, 0, 0, [1], [0b10100])                                       # shards 2 and 4
    _ae(o, 0, 0, [1, 1], [0b01000, 0b00100], leader_commit=1)          # + shard 3 for slot 1 (and a new slot 2)
    # 3 shards >= majority but 1 data shard < majority: reconstruct_data fills the data shards, then it executes (:209-229)
    assert _masks(o, 3) == [ALL, 0b11111, 0b00100]
    assert (o.dump()["last_commit"] == 1).all() and list(o.dump_masks()["counters"]) == [G, 0]


def test_no_absorb_once_the_data_shards_are_there(oracle):
    o = _follower(oracle)
    _ae(o, 0, 0, [1], [0b00111])
    _ae(o, 0, 0, [1, 1], [0b11000, 0b00100])                           # avail_data_shards() == majority already: left alone (:133)
    assert _masks(o, 3) == [ALL, 0b00111, 0b00100]


def test_execution_stops_at_the_first_entry_without_enough_shards(oracle):
    o = _follower(oracle)
    _ae(o, 0, 0, [1, 1, 1], [0b00111, 0b00100, 0b00111])
    _ae(o, 3, 1, [], [], leader_commit=3)                              # heartbeat: new_commit = min(3, 3 + 0) = 3
    assert (o.dump()["last_commit"] == 1).all()                        # slot 2 has one shard: break; slot 3 waits behind it
    assert (o.dump_votes()["n_exec"] == 1).all() and list(o.dump_masks()["counters"]) == [0, G]
    _ae(o, 1, 1, [1, 1, 1], [0b00011, 0b00111, 0b00100], leader_commit=3)   # slot 2's other data shards arrive; slot 4 is new
    # absorbed at slots 2 and 3, slot 4 appended; entries.len() is 2 after the drain: new_commit = min(3, 1 + 2) = 3
    assert _masks(o, 5) == [ALL, 0b00111, 0b00111, 0b00111, 0b00100]
    assert (o.dump()["last_commit"] == 3).all() and (o.dump_votes()["n_exec"] == 3).all()


def test_heartbeats_are_checked_too_and_a_failed_check_still_records_the_leader(oracle):
    o = _follower(oracle)
    _ae(o, 0, 0, [1], [0b00100])
    r = _ae(o, 5, 1, [], [], leader=3)                                 # a heartbeat whose prev_slot I do not have: raft would ignore the check
    assert int(r["flags"][0]) == 3 and int(r["end_slot"][0]) == 5 and int(r["conflict_term"][0]) == 0 and int(r["conflict_slot"][0]) == 5
    assert (o.dump()["leader"] == 3).all()                             # craft/messages.rs:81-84
    r = _ae(o, 1, 7, [], [], leader=4)                                 # prev_term mismatch: conflict hint = first slot of my term-1 run
    assert int(r["flags"][0]) == 3 and int(r["conflict_term"][0]) == 1 and int(r["conflict_slot"][0]) == 1
    r = _ae(o, 1, 1, [], [], term=0, leader=1)                         # stale term: fails, and the leader is NOT recorded
    assert int(r["flags"][0]) == 3 and (o.dump()["leader"] == 4).all()


def test_conflicting_entry_truncates_and_takes_the_new_shards(oracle):
    o = _follower(oracle)
    _ae(o, 0, 0, [1, 1], [0b00111, 0b00111])
    _ae(o, 0, 0, [1, 2, 2], [0b00111, 0b00100, 0b00100], term=2)       # slot 2 now of term 2: truncate there, append 2 and 3
    d = o.dump()
    assert (d["log_len"] == 4).all() and (o.dump_votes()["n_trunc"] == 1).all()
    assert _masks(o, 4) == [ALL, 0b00111, 0b00100, 0b00100]


def test_reconstruct_answers_what_it_holds_under_the_asked_term(oracle):
    o = _follower(oracle)
    _ae(o, 0, 0, [1, 1], [0b00100, 0b01100])
    n = np.full(G, 4, np.uint32)
    slot = np.array([[1] * G, [2] * G, [2] * G, [9] * G], np.uint32)
    term = np.array([[1] * G, [1] * G, [3] * G, [1] * G], np.uint64)   # slot 2 under term 3: not mine; slot 9: beyond my log
    r = o.handle_reconstruct(n, slot, term)
    assert (r["n"] == 2).all() and r["has"][:, 0].tolist() == [1, 1, 0, 0] and r["mask"][:, 0].tolist() == [0b00100, 0b01100, 0, 0]
    n[:] = 0
    assert (o.handle_reconstruct(n, slot, term)["n"] == 0).all()       # nothing asked: no ReconstructReply (:649)


def test_a_leader_that_was_a_follower_gates_on_shards_and_asks_for_them(oracle):
    """craft/messages.rs:315-358 and :665-745: replica 2 holds slots 1-3 as replica 0's follower left them (one shard, the data
    shards, two shards), is elected for term 2, appends slot 4 and gets it replicated: the commit index reaches 4, nothing can
    execute before slot 1 has majority shards; the peers' ReconstructReplies bring them"""
    o = _follower(oracle)
    _ae(o, 0, 0, [1, 1, 1], [0b00100, 0b00111, 0b10100])
    f = lambda v, dt: np.full(G, v, dt)
    r = o.become_candidate(f(0, np.uint8))                              # the timer about leader 0 fires
    assert (r["flags"] == 1).all() and (r["term"] == 2).all()
    vt = np.zeros((5, G), np.uint64); vf = np.zeros((5, G), np.uint8)
    vt[[1, 3]] = 2; vf[[1, 3]] = 3                                      # two votes + my own = majority
    o.handle_vote_replies(vt, vf, None)
    assert (o.dump()["role"] == 2).all()
    o.append(f(1, np.uint32))                                           # slot 4, term 2, every shard (I made it)
    rt = np.zeros((5, G), np.uint64); es = np.zeros((5, G), np.uint32); fl = np.zeros((5, G), np.uint8)
    rt[[1, 3]] = 2; es[[1, 3]] = 4; fl[[1, 3]] = 1
    o.handle_replies(rt, es, fl, None, None, None)                      # 3 holders of slot 4 < majority + f = 4
    assert (o.take_reconstructs()["n"] == 0).all()
    rt[:] = 0; es[:] = 0; fl[:] = 0
    rt[4] = 2; es[4] = 4; fl[4] = 1
    o.handle_replies(rt, es, fl, None, None, None)                      # the fourth: new_commit = 4
    assert (o.dump()["last_commit"] == 0).all()                         # slot 1 has one shard: nothing executes behind it (:318-325)
    q = o.take_reconstructs()
    assert (q["n"] == 2).all() and q["slot"][:2, 0].tolist() == [1, 3] and q["term"][:2, 0].tolist() == [1, 1]
    assert _masks(o, 5) == [ALL, 0b00100, 0b00111, 0b10100, ALL]
    o.handle_replies(rt, es, fl, None, None, None)                      # the same reply again: asked only once (last_recon)
    assert (o.take_reconstructs()["n"] == 0).all()
    # ReconstructReply of peer 1: its shard (1) of slots 1 and 3
    n = f(2, np.uint32)
    slot = np.array([[1] * G, [3] * G], np.uint32); mask = np.array([[0b00010] * G, [0b00010] * G], np.uint8)
    o.handle_reconstruct_reply(f(1, np.uint8), n, slot, mask)
    assert _masks(o, 5) == [ALL, 0b00110, 0b00111, 0b10110, ALL] and (o.dump()["last_commit"] == 0).all()   # slot 1: 2 shards
    # ReconstructReply of peer 3: shard 3 of slot 1 -> 3 shards, 2 of them data: reconstruct_data, then everything up to the
    # shadow commit index (the 3rd largest match index = 4) executes: slot 3 needs reconstruct_data too
    o.handle_reconstruct_reply(f(3, np.uint8), f(1, np.uint32), np.array([[1] * G], np.uint32), np.array([[0b01000] * G], np.uint8))
    assert _masks(o, 5) == [ALL, 0b01111, 0b00111, 0b10111, ALL]
    assert (o.dump()["last_commit"] == 4).all() and o.total_commits() == 4 * G
    assert list(o.dump_masks()["counters"]) == [2 * G, 0]
