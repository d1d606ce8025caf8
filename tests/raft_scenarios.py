"""Seeded message generators for the Raft follower / election handlers, driven by the CPU
oracle's current state so that the messages land in every branch (consistent append, stale
term, prev mismatch, overlap with and without differing terms, duplicates, commit learning)."""
import numpy as np

NO = 0xFF


def append_entries_round(rng, dump, G, K, me, W):
    """one AppendEntries message per group, crafted against the replica's current log"""
    log_len, start, term, et = dump["log_len"], dump["start_slot"], dump["curr_term"], dump["entry_term"]
    kind = rng.integers(0, 10, G)
    flags = (rng.random(G) < 0.9).astype(np.uint8)
    leader = rng.integers(0, 5, G).astype(np.uint8)
    leader[leader == me] = (me + 1) % 5
    m_term = term.copy()
    m_term[kind == 0] -= np.minimum(m_term[kind == 0], 1)                    # stale term
    m_term[kind == 1] += 1                                                   # newer term
    back = np.where(kind >= 6, rng.integers(0, 4, G), 0)                     # overlap the tail
    prev_slot = np.maximum(log_len.astype(np.int64) - 1 - back, start.astype(np.int64)).astype(np.uint32)
    prev_slot[kind == 2] += 3                                                # beyond my log
    g = np.arange(G)
    real_prev = np.where(prev_slot < log_len, et[prev_slot % W, g], 0)
    prev_term = real_prev.copy()
    prev_term[kind == 3] += 1                                                # prev_term mismatch
    n_entries = rng.integers(0, K + 1, G).astype(np.uint32)
    entry_term = np.zeros((K, G), np.uint64)
    for k in range(K):
        slot = prev_slot.astype(np.int64) + 1 + k
        existing = np.where(slot < log_len, et[slot % W, g], m_term)
        differ = (kind == 7) & (rng.random(G) < 0.5)                         # conflicting suffix: truncate
        entry_term[k] = np.where(differ, existing + 1, np.where(kind == 8, existing, np.maximum(existing, 1)))
        entry_term[k] = np.where(slot >= log_len, np.maximum(m_term, 1), entry_term[k])
    leader_commit = (log_len + rng.integers(-3, 4, G)).clip(0).astype(np.uint32)
    last_snap = rng.integers(0, 3, G).astype(np.uint32)
    return dict(flags=flags, leader=leader, term=m_term.astype(np.uint64), prev_slot=prev_slot,
                prev_term=prev_term.astype(np.uint64), n_entries=n_entries, entry_term=np.ascontiguousarray(entry_term),
                leader_commit=leader_commit, last_snap=last_snap)


def timeout_round(rng, dump, G, me):
    src = np.where(rng.random(G) < 0.5, dump["leader"], NO).astype(np.uint8)
    src[rng.random(G) < 0.1] = (me + 2) % 5                                  # a timer about somebody who is not my leader
    return src


def request_vote_round(rng, dump, G, me, W):
    log_len, term, et = dump["log_len"], dump["curr_term"], dump["entry_term"]
    g = np.arange(G)
    flags = (rng.random(G) < 0.85).astype(np.uint8)
    cand = rng.integers(0, 5, G).astype(np.uint8)
    cand[cand == me] = (me + 1) % 5
    m_term = (term.astype(np.int64) + rng.integers(-1, 3, G)).clip(0).astype(np.uint64)
    my_last = et[(log_len - 1) % W, g]
    last_term = (my_last.astype(np.int64) + rng.integers(-1, 2, G)).clip(0).astype(np.uint64)
    last_slot = (log_len.astype(np.int64) - 1 + rng.integers(-2, 3, G)).clip(0).astype(np.uint32)
    return dict(flags=flags, candidate=cand, term=m_term, last_slot=last_slot, last_term=last_term)


def vote_reply_round(rng, dump, G, R, me, ctl):
    term = dump["curr_term"]
    flags = (rng.random((R, G)) < 0.6).astype(np.uint8)
    flags[me] = 0
    t = np.broadcast_to(term, (R, G)).copy()
    t[rng.random((R, G)) < 0.1] += 1                                         # a peer already in a later term
    t[rng.random((R, G)) < 0.1] -= 1
    return dict(term=np.ascontiguousarray(t.astype(np.uint64)), flags=np.ascontiguousarray(flags), order=ctl)
