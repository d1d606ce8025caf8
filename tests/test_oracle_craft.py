"""Pins the CRaft leader variant of oracle/raft_oracle.c (orc_craft_*) by traces worked out by hand from the
reference's rules: craft/messages.rs:256-404 (reply handler: heard-heartbeat count, no stale-success test, the
`majority + fault_tolerance` / full-copy commit rule), craft/leadership.rs:80-141,249-291 (mode switch, heartbeat
tick + fall-back test), server/heartbeat.rs:117-131,240-296 (reply counters), craft/request.rs:71-100 (shard
assignment).  CPU only."""
import numpy as np
import pytest

G = 3


@pytest.fixture()
def orc(oracle):
    return oracle.CRaftOracle(G, R=5, W=64, leader_id=0, term=1, fault_tolerance=1, repeat_threshold=2)


def _reply(o, peers_end, term=1, conflict=None):
    """one handle_replies call: peers_end = {peer: end_slot} for every group alike"""
    R = 5
    rt = np.zeros((R, G), np.uint64); es = np.zeros((R, G), np.uint32); fl = np.zeros((R, G), np.uint8)
    ct = np.zeros((R, G), np.uint64); cs = np.zeros((R, G), np.uint32)
    for p, e in peers_end.items():
        rt[p] = term; es[p] = e; fl[p] = 1
        if conflict and p in conflict:
            fl[p] = 3; ct[p], cs[p] = conflict[p]
    o.handle_replies(rt, es, fl, ct, cs, None)


def test_commit_needs_majority_plus_fault_tolerance(orc):
    orc.append(np.full(G, 2, np.uint32))                  # slots 1, 2 of term 1
    _reply(orc, {1: 2, 2: 2})                             # leader + 2 = 3 = majority: not enough with f = 1
    assert (orc.dump()["last_commit"] == 0).all()
    _reply(orc, {3: 1})                                   # slot 1 now on 4 replicas = majority + f
    assert (orc.dump()["last_commit"] == 1).all()
    _reply(orc, {3: 2})
    d = orc.dump()
    assert (d["last_commit"] == 2).all() and (d["last_snap"] == 0).all()
    _reply(orc, {4: 2})                                   # every server holds 1..2: safe to snapshot
    assert (orc.dump()["last_snap"] == 2).all()
    assert orc.total_commits() == 2 * G


def test_full_copy_mode_commits_at_majority(orc):
    orc.switch_assignment_mode(np.array([1, 0, 7], np.uint8))   # group 0 falls back, 1 stays (already 1-shard), 2 no call
    assert orc.dump_craft()["full_copy_mode"].tolist() == [1, 0, 0]
    orc.append(np.full(G, 1, np.uint32))
    _reply(orc, {1: 1, 2: 1})
    assert orc.dump()["last_commit"].tolist() == [1, 0, 0]
    orc.switch_assignment_mode(np.array([0, 1, 1], np.uint8))   # and back (the method allows it; vanilla CRaft never calls it)
    assert orc.dump_craft()["full_copy_mode"].tolist() == [0, 1, 1]
    _reply(orc, {1: 1})                                   # a repeated success reply re-runs the commit scan
    assert orc.dump()["last_commit"].tolist() == [1, 1, 1]


def test_assignment_masks(orc):
    persist, send = orc.assignment()
    assert persist.tolist() == [1, 1, 1]                  # the leader (id 0) keeps its own shard
    assert send[:, 0].tolist() == [0, 2, 4, 8, 16]        # shard p to peer p
    orc.switch_assignment_mode(np.array([1, 9, 9], np.uint8))
    persist, send = orc.assignment()
    assert persist.tolist() == [7, 1, 1]                  # the data shards 0..majority
    assert send[:, 0].tolist() == [0, 7, 7, 7, 7] and send[:, 1].tolist() == [0, 2, 4, 8, 16]


def test_stale_success_reply_moves_next_and_match_back(orc):
    """raft returns on `next_slot > end_slot + 1` (raft/messages.rs:245-247); the fork only debug_asserts it
    (craft/messages.rs:279), so a release build takes the reply: next / match go back, try_next and the commit stay"""
    orc.append(np.full(G, 3, np.uint32))
    _reply(orc, {1: 3, 2: 3, 3: 3})
    assert (orc.dump()["last_commit"] == 3).all()
    _reply(orc, {1: 1})
    d = orc.dump()
    assert (d["next_slot"][1] == 2).all() and (d["match_slot"][1] == 1).all() and (d["try_next_slot"][1] == 4).all()
    assert (d["last_commit"] == 3).all()


def test_reply_counters_and_fallback(orc):
    c = orc.dump_craft()
    assert c["hb_replied"][:, 0].tolist() == [0, 1, 1, 1, 1] and c["peer_alive"].tolist() == [31] * G
    hb = orc.bcast_heartbeats()                           # tick 1: replied 1 > seen 0 for every peer
    assert hb["hb_flags"][:, 0].tolist() == [0, 1, 1, 1, 1] and (hb["prev_slot"] == 0).all() and (hb["prev_term"] == 0).all()
    c = orc.dump_craft()
    assert c["hb_seen"][:, 0].tolist() == [0, 1, 1, 1, 1] and (c["hb_repeat"] == 0).all()
    for tick in (2, 3, 4):                                # peers 1..3 answer every tick, peer 4 never does
        _reply(orc, {1: 0, 2: 0, 3: 0})
        orc.bcast_heartbeats()
        c = orc.dump_craft()
        if tick < 4:
            assert c["hb_repeat"][4].tolist() == [tick - 1] * G and c["peer_alive"].tolist() == [31] * G
            assert c["full_copy_mode"].tolist() == [0] * G
    # repetition 3 > repeat_threshold 2: peer 4 speculated dead, counter reset; 5 - 4 alive >= f = 1: fall back
    assert c["hb_repeat"][4].tolist() == [0] * G and c["peer_alive"].tolist() == [15] * G
    assert c["full_copy_mode"].tolist() == [1] * G
    assert c["hb_replied"][1].tolist() == [4] * G and c["hb_seen"][1].tolist() == [4] * G
    _reply(orc, {4: 0})                                   # peer 4 is heard again: alive, but the mode stays (heard_heartbeat's
    c = orc.dump_craft()                                  # switch back is commented out, craft/leadership.rs:304-309)
    assert c["peer_alive"].tolist() == [31] * G and c["full_copy_mode"].tolist() == [1] * G
    assert c["hb_replied"][4].tolist() == [2] * G


def test_heartbeat_prev_slot_follows_try_next(orc):
    orc.append(np.full(G, 2, np.uint32))                  # try_next of every peer -> 3 after the sends
    hb = orc.bcast_heartbeats()
    assert (hb["prev_slot"][1:] == 2).all() and (hb["prev_term"][1:] == 1).all() and (hb["leader_commit"] == 0).all()
    _reply(orc, {2: 2}, conflict={2: (1, 1)})             # conflict: next 1 -> try_next 1 ... resend up to end_slot
    d = orc.dump()
    assert (d["next_slot"][2] == 1).all()
    # a higher term in a reply: step down, no heard count, and a follower's Heartbeater does not tick
    before = orc.dump_craft()["hb_replied"].copy()
    _reply(orc, {3: 0}, term=5)
    assert (orc.dump()["role"] == 0).all() and np.array_equal(orc.dump_craft()["hb_replied"], before)
    hb = orc.bcast_heartbeats()
    assert (hb["hb_flags"] == 0).all()
    assert np.array_equal(orc.dump_craft()["hb_repeat"], np.zeros((5, G), np.uint8))


def test_golden_final_state(oracle):
    """the frozen run of tests/test_zz_craft_gpu.py, oracle alone, ends in the committed state (tests/golden/late_golden.npz)"""
    import os
    import test_zz_craft_gpu as t
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "late_golden.npz"))
    out = t._run(None, oracle, **t.GOLDEN_RUN)
    assert out and all(np.array_equal(v, gold["craft_" + k]) for k, v in out.items())
