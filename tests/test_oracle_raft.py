"""Raft follower / election handlers of the CPU oracle against hand-derived traces of the
reference code (src/protocols/raft/messages.rs, leadership.rs, durability.rs): every expected
value below was worked out from the cited lines, including the reference's quirks."""
import numpy as np

FOLLOWER, CANDIDATE, LEADER, NO = 0, 1, 2, 0xFF


def _mk(oracle, log_terms, role=FOLLOWER, leader=0, term=None, voted=NO, me=2, W=16):
    """one group whose log is [dummy 0] + log_terms (built through the leader append path)"""
    o = oracle.RaftOracle(1, 5, W, leader_id=me, term=1)
    for t in log_terms:
        o.preset(LEADER, me, t)
        o.append(np.ones(1, np.uint32))
    o.preset(role, leader, term if term is not None else (log_terms[-1] if log_terms else 1), voted)
    return o


def _ae(o, leader, term, prev_slot, prev_term, ents, commit=0, snap=0, K=4):
    et = np.zeros((K, 1), np.uint64)
    et[:len(ents), 0] = ents
    a = lambda v, t: np.array([v], t)
    return o.handle_append_entries(a(1, np.uint8), a(leader, np.uint8), a(term, np.uint64), a(prev_slot, np.uint32),
                                   a(prev_term, np.uint64), a(len(ents), np.uint32), et, a(commit, np.uint32),
                                   a(snap, np.uint32))


def _terms(o):
    d = o.dump()
    return [int(d["entry_term"][s % o.W, 0]) for s in range(int(d["start_slot"][0]), int(d["log_len"][0]))]


def test_consistent_append_replies_after_last_entry(oracle):
    o = _mk(oracle, [1, 1])                                  # log terms [0,1,1]
    r = _ae(o, 0, 1, 2, 1, [1, 1], commit=2)
    assert _terms(o) == [0, 1, 1, 1, 1]
    assert (int(r["flags"][0]), int(r["term"][0]), int(r["end_slot"][0])) == (1, 1, 4)      # durability.rs:113-123
    # messages.rs:184-186 after the drain: entries.len() == 0 -> new_commit = min(2, prev_slot + 0) = 2
    assert int(o.dump()["last_commit"][0]) == 2 and int(o.dump_votes()["n_exec"][0]) == 2


def test_heartbeat_gets_a_reply_with_prev_slot(oracle):
    o = _mk(oracle, [1])
    r = _ae(o, 0, 1, 1, 1, [])                               # messages.rs:172-181: end_slot = first_new - 1 = prev_slot
    assert (int(r["flags"][0]), int(r["end_slot"][0])) == (1, 1) and _terms(o) == [0, 1]
    # an empty message is never checked against my log (:46 `!entries.is_empty() &&`): prev beyond the log
    r = _ae(o, 0, 1, 7, 3, [])
    assert (int(r["flags"][0]), int(r["end_slot"][0])) == (1, 7)


def test_prev_mismatch_conflict_hint(oracle):
    o = _mk(oracle, [1, 2, 2, 2], term=3)                    # [0,1,2,2,2]
    r = _ae(o, 0, 3, 4, 3, [3])                              # my term at 4 is 2, not 3
    assert int(r["flags"][0]) == 3 and int(r["end_slot"][0]) == 5          # :70-74 end_slot = prev_slot + len
    assert (int(r["conflict_term"][0]), int(r["conflict_slot"][0])) == (2, 2)   # first slot of term 2 (:60-68)
    assert _terms(o) == [0, 1, 2, 2, 2]
    r = _ae(o, 0, 3, 9, 3, [3])                              # prev beyond my log: conflict_term 0, slot = prev
    assert (int(r["flags"][0]), int(r["conflict_term"][0]), int(r["conflict_slot"][0])) == (3, 0, 9)


def test_stale_term_is_refused_with_my_term(oracle):
    o = _mk(oracle, [1, 1], term=5)
    r = _ae(o, 0, 4, 2, 1, [4])
    assert (int(r["flags"][0]), int(r["term"][0])) == (3, 5) and _terms(o) == [0, 1, 1]


def test_conflicting_suffix_is_truncated(oracle):
    o = _mk(oracle, [1, 1, 1], term=2)                       # [0,1,1,1]
    r = _ae(o, 0, 2, 1, 1, [2, 2])                           # slots 2,3 hold term 1 -> truncate at 2, append both
    assert _terms(o) == [0, 1, 2, 2] and int(o.dump_votes()["n_trunc"][0]) == 1
    assert (int(r["flags"][0]), int(r["end_slot"][0])) == (1, 3)


def test_resent_entries_are_appended_again(oracle):
    # messages.rs:99-167: when no entry differs and none is beyond my log, first_new stays prev_slot + 1
    # and every entry is pushed again (the reference does not skip what it already holds)
    o = _mk(oracle, [1, 1, 1])
    r = _ae(o, 0, 1, 1, 1, [1, 1])
    assert _terms(o) == [0, 1, 1, 1, 1, 1]
    assert (int(r["flags"][0]), int(r["end_slot"][0])) == (1, 3)           # the reply still names slot_e = prev + n


def test_commit_learning_is_capped_below_the_new_entries(oracle):
    o = _mk(oracle, [1, 1])                                  # [0,1,1]
    _ae(o, 0, 1, 1, 1, [1, 1, 1], commit=5)                  # overlap 1 (slot 2 matches), new from slot 3
    assert _terms(o) == [0, 1, 1, 1, 1]
    assert int(o.dump()["last_commit"][0]) == 2              # min(5, prev_slot + skipped = 1 + 1)
    o2 = _mk(oracle, [1, 1, 1])
    _ae(o2, 0, 1, 3, 1, [1], commit=3)
    assert int(o2.dump()["last_commit"][0]) == 3
    _ae(o2, 0, 1, 1, 1, [1, 1], commit=4)                    # resend: skipped = 0 -> new_commit = min(4, 1) = 1 < 3
    assert int(o2.dump()["last_commit"][0]) == 1             # last_commit moves BACK (messages.rs:207)


def test_leader_and_candidate_roles(oracle):
    o = _mk(oracle, [1], role=LEADER, leader=2, term=1)
    r = _ae(o, 0, 1, 1, 1, [1])                              # same term, I am leader: ignored (:32-42)
    assert int(r["flags"][0]) == 0 and _terms(o) == [0, 1]
    r = _ae(o, 0, 2, 1, 1, [2])                              # higher term: step down, message dropped (check_term true)
    d = o.dump()
    assert int(r["flags"][0]) == 0 and (int(d["role"][0]), int(d["curr_term"][0]), int(d["leader"][0])) == (FOLLOWER, 2, 0)
    o = _mk(oracle, [1], role=CANDIDATE, leader=NO, term=2, voted=2)
    r = _ae(o, 3, 2, 1, 1, [2])                              # equal term while Candidate: the :33-39 hack
    d = o.dump()
    assert (int(d["role"][0]), int(d["curr_term"][0]), int(d["leader"][0])) == (FOLLOWER, 2, 3)
    assert int(r["flags"][0]) == 1 and _terms(o) == [0, 1, 2] and int(o.dump_votes()["voted_for"][0]) == NO


def test_become_a_candidate(oracle):
    o = _mk(oracle, [1, 3], leader=0, term=3)
    r = o.become_candidate(np.array([4], np.uint8))          # timer about 4, my leader is 0: ignored (:80-85)
    assert int(r["flags"][0]) == 0
    r = o.become_candidate(np.array([0], np.uint8))
    assert (int(r["flags"][0]), int(r["term"][0]), int(r["last_slot"][0]), int(r["last_term"][0])) == (1, 4, 2, 3)
    d, v = o.dump(), o.dump_votes()
    assert (int(d["role"][0]), int(d["curr_term"][0]), int(v["voted_for"][0]), int(v["votes"][0])) == (CANDIDATE, 4, 2, 0b100)
    assert int(o.become_candidate(np.array([0], np.uint8))["flags"][0]) == 0     # not a Follower any more


def _rv(o, cand, term, last_slot, last_term):
    a = lambda v, t: np.array([v], t)
    r = o.handle_request_vote(a(1, np.uint8), a(cand, np.uint8), a(term, np.uint64), a(last_slot, np.uint32),
                              a(last_term, np.uint64))
    return int(r["flags"][0]), int(r["term"][0])


def test_request_vote_rules(oracle):
    o = _mk(oracle, [1, 2], term=2)                          # my last: slot 2, term 2
    assert _rv(o, 1, 1, 9, 9) == (1, 2)                      # smaller term: refused with my term (:408-422)
    assert _rv(o, 1, 2, 2, 2) == (3, 2) and int(o.dump_votes()["voted_for"][0]) == 1
    assert _rv(o, 3, 2, 5, 2) == (0, 0)                      # already voted for 1 in this term: NO reply at all
    assert _rv(o, 1, 2, 2, 2) == (3, 2)                      # the same candidate again: granted again
    assert _rv(o, 4, 3, 0, 1) == (0, 0)                      # new term (vote cleared) but its log is behind: no reply
    d, v = o.dump(), o.dump_votes()
    assert (int(d["curr_term"][0]), int(d["leader"][0]), int(v["voted_for"][0])) == (3, 4, NO)
    # the laxer second clause (:429-430): last_term == curr_term and last_slot + 1 >= my log end
    o = _mk(oracle, [5, 5], term=3)                          # my last term 5 > candidate's 3
    assert _rv(o, 1, 3, 2, 3) == (3, 3)


def test_vote_replies_elect_at_quorum(oracle):
    o = _mk(oracle, [1, 1], leader=0, term=1)
    o.become_candidate(np.array([0], np.uint8))              # term 2, votes {2}
    term = np.full((5, 1), 2, np.uint64)
    flags = np.zeros((5, 1), np.uint8)
    flags[0] = 1
    r = o.handle_vote_replies(term, flags)
    assert int(r["elected"][0]) == 0 and int(o.dump_votes()["votes"][0]) == 0b101
    flags[:] = 0
    flags[4] = 1
    r = o.handle_vote_replies(term, flags, granted=np.zeros((5, 1), np.uint8))   # `granted` is not looked at (:503)
    d = o.dump()
    assert int(r["elected"][0]) == 1 and int(d["role"][0]) == LEADER
    # bcast_heartbeats (leadership.rs:156) uses try_next_slot as it stood (3 from the set-up's appends):
    # prev_slot = min(try_next - 1, last slot) = 2; the re-initialisation comes after (:159-168)
    assert [int(r["hb_prev_slot"][p, 0]) for p in (0, 1, 3, 4)] == [2, 2, 2, 2]
    assert [int(d["next_slot"][p, 0]) for p in (0, 1, 3, 4)] == [3, 3, 3, 3]
    assert [int(d["match_slot"][p, 0]) for p in (0, 1, 3, 4)] == [0, 0, 0, 0]
    flags[:] = 0
    flags[1] = 1
    term[1] = 3                                              # a later-term reply deposes the fresh leader
    o.handle_vote_replies(term, flags)
    d = o.dump()
    assert (int(d["role"][0]), int(d["curr_term"][0]), int(d["leader"][0])) == (FOLLOWER, 3, 1)
