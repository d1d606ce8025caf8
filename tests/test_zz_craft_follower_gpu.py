"""The CRaft follower kernels (raft_append_entries_kernel<true>, craft_reconstruct_kernel; summerset_amd/csrc/raft_engine.hip,
through the C-ABI): the hand-derived traces of tests/test_oracle_craft_follower.py on the engine, and seeded rounds of crafted
AppendEntries (stale / newer terms, overlapping and conflicting suffixes, heartbeats, random shard bitmaps) + Reconstructs with
every reply and the full state -- log, shard bitmaps, counters -- compared with the oracle after every call.  Bit-exact."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import raft_scenarios as sc  # noqa: E402
import test_oracle_craft_follower as tr  # noqa: E402

pytestmark = pytest.mark.gpu


def _t(a, cuda):
    import torch
    return torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else (a.view(np.int32) if a.dtype == np.uint32 else a)).to(cuda)


class _Eng:
    """CRaftLeaderGroup behind the oracle's numpy interface"""

    def __init__(self, cuda, G, R, W, me, term, ft):
        from summerset_amd import CRaftLeaderGroup
        self.e, self.cuda, self.G, self.R, self.W = CRaftLeaderGroup(G, R, leader_id=me, window=W, term=term, fault_tolerance=ft), cuda, G, R, W

    def preset(self, *a): self.e.preset(*a)
    def append(self, n_new): self.e.handle_req_batch(_t(n_new, self.cuda))
    def dump(self): return self.e.dump()
    def dump_votes(self): return self.e.dump_votes()
    def dump_masks(self): return self.e.dump_masks()

    def handle_append_entries(self, flags, leader, term, prev_slot, prev_term, n_entries, entry_term, leader_commit, last_snap, entry_mask=None):
        if entry_mask is None:
            entry_mask = np.full(entry_term.shape, (1 << self.R) - 1, np.uint8)
        r = self.e.handle_msg_append_entries(*[_t(x, self.cuda) for x in (flags, leader, term, prev_slot, prev_term, n_entries, entry_term,
                                                                          leader_commit, last_snap)], entry_mask=_t(entry_mask, self.cuda))
        like = dict(flags=np.uint8, term=np.uint64, end_slot=np.uint32, conflict_term=np.uint64, conflict_slot=np.uint32)
        return {k: r[k].cpu().numpy().view(v) for k, v in like.items()}

    def become_candidate(self, src):
        r = self.e.become_a_candidate(_t(src, self.cuda))
        like = dict(flags=np.uint8, term=np.uint64, last_slot=np.uint32, last_term=np.uint64)
        return {k: r[k].cpu().numpy().view(v) for k, v in like.items()}

    def handle_vote_replies(self, term, flags, order):
        return self.e.handle_msg_request_vote_reply(_t(term, self.cuda), _t(flags, self.cuda), None if order is None else _t(order, self.cuda))

    def handle_replies(self, rt, es, fl, ct, cs, order):
        f = lambda a: None if a is None else _t(a, self.cuda)
        self.e.handle_msg_append_entries_reply(_t(rt, self.cuda), _t(es, self.cuda), _t(fl, self.cuda), f(ct), f(cs), f(order))

    def total_commits(self): return self.e.total_commits()

    def take_reconstructs(self, K=16):
        r = self.e.poll_reconstructs(self.cuda, K)
        return dict(n=r["n"].cpu().numpy().view(np.uint32), slot=r["slot"].cpu().numpy().view(np.uint32), term=r["term"].cpu().numpy().view(np.uint64))

    def handle_reconstruct_reply(self, peer, n, slot, mask):
        self.e.handle_msg_reconstruct_reply(_t(peer, self.cuda), _t(n, self.cuda), _t(slot, self.cuda), _t(mask, self.cuda))

    def handle_reconstruct(self, n, slot, term):
        r = self.e.handle_msg_reconstruct(_t(n, self.cuda), _t(slot, self.cuda), _t(term, self.cuda))
        return dict(n=r["n"].cpu().numpy().view(np.uint32), has=r["has"].cpu().numpy(), mask=r["mask"].cpu().numpy())


class _AsOracle:
    def __init__(self, cuda):
        self.cuda = cuda

    def CRaftOracle(self, G, R=5, W=64, leader_id=0, term=1, fault_tolerance=1, repeat_threshold=3):
        return _Eng(self.cuda, G, R, W, leader_id, term, fault_tolerance)


TRACES = [getattr(tr, n) for n in dir(tr) if n.startswith("test_")]


@pytest.mark.parametrize("trace", TRACES, ids=lambda f: f.__name__[5:])
def test_trace_on_the_engine(cuda, trace):
    trace(_AsOracle(cuda))


@pytest.mark.parametrize("G,W,seed", [(777, 64, 1), (4096, 32, 2)])
def test_crafted_rounds_match_the_oracle(cuda, oracle, G, W, seed):
    R, me, K = 5, 2, 6
    rng = np.random.default_rng(seed)
    eng = _Eng(cuda, G, R, W, me, 1, 1)
    orc = oracle.CRaftOracle(G, R, W, leader_id=me, term=1, fault_tolerance=1)
    for _ in range(3):                                       # a few leader appends (every shard), then everybody follows replica 0
        n_new = rng.integers(0, 4, G).astype(np.uint32)
        eng.append(n_new); orc.append(n_new)
    eng.preset(0, 0, 1); orc.preset(0, 0, 1)
    g = np.arange(G)
    for step in range(40):
        d = orc.dump()
        if step % 4 != 3:
            m = sc.append_entries_round(rng, d, G, K, me, W)
            # shard bitmaps: one's own shard, the data shards, two or three random shards, nothing at all
            kind = rng.integers(0, 5, (K, G))
            em = np.select([kind == 0, kind == 1, kind == 2, kind == 3], [1 << me, 0b00111, rng.integers(0, 32, (K, G)) | rng.integers(0, 32, (K, G)),
                                                                         rng.integers(0, 32, (K, G))], 0).astype(np.uint8)
            m["entry_mask"] = np.ascontiguousarray(em)
            ro, re_ = orc.handle_append_entries(**m), eng.handle_append_entries(**m)
            for k in ro:
                assert np.array_equal(ro[k], re_[k]), (step, k)
        else:
            n = rng.integers(0, K + 1, G).astype(np.uint32)
            slot = (d["log_len"][None, :].astype(np.int64) - rng.integers(0, 6, (K, G))).clip(0).astype(np.uint32)
            et = d["entry_term"][slot % W, g[None, :]]
            term = np.where(rng.random((K, G)) < 0.8, et, et + 1).astype(np.uint64)
            ro, re_ = orc.handle_reconstruct(n, slot, np.ascontiguousarray(term)), eng.handle_reconstruct(n, slot, np.ascontiguousarray(term))
            for k in ro:
                assert np.array_equal(ro[k], re_[k]), (step, k)
        a, b = eng.dump(), orc.dump()
        for k in a:
            assert np.array_equal(a[k], b[k]), (step, k)
        a, b = eng.dump_masks(), orc.dump_masks()
        assert np.array_equal(a["mask"], b["mask"]) and list(a["counters"]) == list(b["counters"]), step
        a, b = eng.dump_votes(), orc.dump_votes()
        for k in a:
            assert np.array_equal(a[k].astype(np.uint64), b[k].astype(np.uint64)), (step, k)
    c = orc.dump_masks()["counters"]
    assert c[0] > 0 and c[1] > 0                              # reconstruct_data and postponed executions both happened
    # ---- everybody is elected and leads a log it did not create: the shard gate, Reconstruct queues, ReconstructReplies ----
    def same(where):
        a, b = eng.dump(), orc.dump()
        for k in a:
            assert np.array_equal(a[k], b[k]), (where, k)
        a, b = eng.dump_masks(), orc.dump_masks()
        assert np.array_equal(a["mask"], b["mask"]) and list(a["counters"]) == list(b["counters"]), where
        assert eng.total_commits() == orc.total_commits(), where
    src = orc.dump()["leader"].astype(np.uint8)
    src[src == me] = 0
    ro, re_ = orc.become_candidate(src), eng.become_candidate(src)
    for k in ro:
        assert np.array_equal(ro[k], re_[k]), ("candidate", k)
    t_now = orc.dump()["curr_term"]
    vt = np.zeros((R, G), np.uint64); vf = np.zeros((R, G), np.uint8)
    for p in (0, 1, 3):
        vt[p] = t_now; vf[p] = 3
    orc.handle_vote_replies(vt, vf, None); eng.handle_vote_replies(vt, vf, None)
    same("elected")
    assert (orc.dump()["role"] == 2).mean() > 0.9
    asked = 0
    for step in range(12):
        n_new = rng.integers(0, 3, G).astype(np.uint32)
        orc.append(n_new); eng.append(n_new)
        d = orc.dump()
        rt = np.zeros((R, G), np.uint64); es = np.zeros((R, G), np.uint32); fl = np.zeros((R, G), np.uint8)
        for p in range(R):
            if p == me:
                continue
            on = rng.random(G) < 0.8
            rt[p] = d["curr_term"]; fl[p] = on
            es[p] = (d["log_len"].astype(np.int64) - 1 - rng.integers(0, 3, G)).clip(0)
        orc.handle_replies(rt, es, fl, None, None, None); eng.handle_replies(rt, es, fl, None, None, None)
        same(("replies", step))
        qo, qe = orc.take_reconstructs(8), eng.take_reconstructs(8)
        for k in qo:
            assert np.array_equal(qo[k], qe[k]), ("queue", step, k)
        asked += int(qo["n"].sum())
        # the peers answer some of what was asked (random shards), one peer per call
        for p in (0, 3):
            n = np.where(rng.random(G) < 0.7, qo["n"], 0).astype(np.uint32)
            mask = rng.integers(0, 32, qo["slot"].shape).astype(np.uint8)
            peer = np.full(G, p, np.uint8)
            peer[rng.random(G) < 0.1] = 0xFF
            orc.handle_reconstruct_reply(peer, n, qo["slot"], mask); eng.handle_reconstruct_reply(peer, n, qo["slot"], mask)
            same(("reconstruct_reply", step, p))
    assert asked > 0 and orc.total_commits() > 0
    assert orc.dump_votes()["n_exec"].sum() > 0 and orc.dump_votes()["n_trunc"].sum() > 0
