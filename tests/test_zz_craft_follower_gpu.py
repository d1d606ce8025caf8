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
    assert orc.dump_votes()["n_exec"].sum() > 0 and orc.dump_votes()["n_trunc"].sum() > 0
