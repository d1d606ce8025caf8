"""CRaft leader variant: the HIP kernels `raft_replies_kernel<true>`, `craft_heartbeat_kernel`,
`craft_mode_kernel` through the C-ABI vs the literal restatement (oracle/raft_oracle.c, orc_craft_*) on the same seeded
streams, bit-exact after every call: replies with loss / lag (stale success replies move next / match back in the fork),
conflicts and step-downs, heartbeat ticks with peers that go silent for a while (reply counters -> peer_alive ->
full-copy fall-back), explicit mode switches, and the shard assignment tied to the RS encode of a new entry."""
import numpy as np
import pytest

from summerset_amd import stream
from test_raft_gpu import _replies

pytestmark = pytest.mark.gpu


def _silent(G, R, t):
    """[R, G] bool: peer p of group g does not answer at tick t (one peer of every third group for ticks 6..19, a
    second one of every seventh group for ticks 10..15, everybody of every 11th group at tick 12)"""
    g = np.arange(G)[None, :]
    p = np.arange(R)[:, None]
    s = (g % 3 == 0) & (p == 1 + g % (R - 1)) & (t >= 6) & (t < 20)
    s |= (g % 7 == 0) & (p == 1 + (g + 2) % (R - 1)) & (t >= 10) & (t < 16)
    s |= (g % 11 == 0) & (t == 12)
    return s


def _same(eng, orc, where):
    a, b = eng.dump(), orc.dump()
    for k in b:
        assert np.array_equal(a[k], b[k]), (where, k, np.nonzero(a[k] != b[k]))
    a, b = eng.dump_craft(), orc.dump_craft()
    for k in b:
        assert np.array_equal(a[k], b[k]), (where, k, np.nonzero(a[k] != b[k]))
    assert eng.total_commits() == orc.total_commits()


def _run(cuda, oracle, G, R, W, T, ft, thr, higher_p=0.0, seed=5):
    """cuda None: the oracle alone (tests/golden/make_golden.py and the CPU test that pins its final state)"""
    orc = oracle.CRaftOracle(G, R, W, 0, 1, ft, thr)
    eng = None
    if cuda is not None:
        import torch
        from summerset_amd import CRaftLeaderGroup
        eng = CRaftLeaderGroup(G, R, 0, W, term=1, fault_tolerance=ft, repeat_threshold=thr)
        dev = lambda a: torch.from_numpy(a).to(cuda)
        _same(eng, orc, "start")
    for t in range(T):
        n_new = (stream._key(seed, 9, t, np.arange(G, dtype=np.uint64)) % np.uint64(3)).astype(np.uint32)
        orc.append(n_new)
        po, so = orc.assignment()
        if eng:
            eng.handle_req_batch(dev(n_new))
            pe, se = eng.assignment(cuda)
            assert np.array_equal(pe.cpu().numpy().view(np.uint32), po) and np.array_equal(se.cpu().numpy().view(np.uint32), so), t
        d = orc.dump()
        term, es, fl, ct, cs, order = _replies(seed, t, G, R, d["log_len"], d["curr_term"], higher_p=higher_p)
        fl[_silent(G, R, t)] = 0
        orc.handle_replies(term, es, fl, ct, cs, order)
        if eng:
            eng.handle_msg_append_entries_reply(dev(term), dev(es), dev(fl), dev(ct), dev(cs), dev(order))
            _same(eng, orc, ("replies", t))
        if t % 2 == 1:                                        # the send tick
            ho = orc.bcast_heartbeats()
            if eng:
                he = eng.bcast_heartbeats(cuda)
                for k, v in ho.items():
                    e = he[k].cpu().numpy()
                    assert np.array_equal(e.view(v.dtype), v), ("heartbeat", t, k)
                _same(eng, orc, ("heartbeat", t))
        if t == T // 2:                                       # someone switches a few groups by hand, both ways
            to = np.full(G, 0xFF, np.uint8); to[::5] = 1; to[2::5] = 0
            orc.switch_assignment_mode(to)
            if eng:
                eng.switch_assignment_mode(dev(to))
                _same(eng, orc, ("switch", t))
    c = orc.dump_craft()
    assert orc.total_commits() > 0
    c.update(orc.dump())
    return c


GOLDEN_RUN = dict(G=96, R=5, W=32, T=36, ft=1, thr=2, higher_p=0.002, seed=31)   # tests/golden/late_golden.npz, "craft_*"


def test_craft_leader_fallback_and_commit_rule(cuda, oracle):
    c = _run(cuda, oracle, G=700, R=5, W=64, T=40, ft=1, thr=2)
    # the stream reaches both modes and speculated deaths that were taken back
    assert 0 < int(c["full_copy_mode"].sum()) < 700 and (c["peer_alive"] == 31).any()


def test_craft_leader_step_down_and_other_populations(cuda, oracle):
    _run(cuda, oracle, G=300, R=5, W=32, T=30, ft=1, thr=1, higher_p=0.002)
    _run(cuda, oracle, G=200, R=3, W=64, T=30, ft=1, thr=2)
    _run(cuda, oracle, G=200, R=7, W=64, T=30, ft=2, thr=3)
    _run(cuda, oracle, G=130, R=5, W=64, T=24, ft=0, thr=2)


def test_craft_entry_shards_follow_the_assignment(cuda, oracle):
    """new entries end to end, one group per codeword: the batches' RS(3,2) codewords from the encode kernel, handed
    out by the masks of craft/request.rs:86-100 -- in 1-shard mode peer p gets shard p (and the shards of any 3
    holders rebuild the batch, craft/messages.rs:335), in full-copy mode every peer gets the 3 data shards and reads
    the batch without decoding"""
    import torch
    from summerset_amd import CRaftLeaderGroup
    from summerset_amd.rscoding import RSCodewordBatch
    G, L = 8, 3000
    eng = CRaftLeaderGroup(G, 5, 0, 64, term=1, fault_tolerance=1, repeat_threshold=2)
    to = np.full(G, 0xFF, np.uint8); to[1::2] = 1
    eng.switch_assignment_mode(torch.from_numpy(to).to(cuda))
    persist, send = eng.assignment(cuda)
    persist, send = persist.cpu().numpy(), send.cpu().numpy()
    data = np.random.default_rng(3).integers(0, 256, (G, L), dtype=np.uint8)
    cw = RSCodewordBatch.from_data(torch.from_numpy(data).to(cuda), 3, 2)
    cw.compute_parity()
    sl = cw.shard_len
    for g in range(G):                                        # the kernel's parity is the oracle's
        par = oracle.rs_encode(3, 2, data[g])
        for k in range(2):
            assert np.array_equal(cw.shard(3 + k)[g].cpu().numpy(), par[k]), (g, k)
    for g in range(G):
        full = g % 2 == 1
        assert persist[g] == (7 if full else 1)
        assert send[:, g].tolist() == ([0, 7, 7, 7, 7] if full else [0, 2, 4, 8, 16])
    # 1-shard groups: what peers 1, 3, 4 hold (shards 1, 3, 4) rebuilds every batch
    got = RSCodewordBatch.from_null(G, 3, 2, device=cuda)
    for p in (1, 3, 4):
        got.absorb_other(cw.subset_copy(int(send[p, 0])))
    assert got.avail == 0b11010 and got.avail_data_shards() < 3
    got.reconstruct_data()
    assert np.array_equal(got.get_data().cpu().numpy(), data)
    # full-copy groups: one peer's share is already the batch
    one = cw.subset_copy(int(send[2, 1]))
    assert one.avail == 7 and np.array_equal(one.get_data().cpu().numpy(), data) and sl * 3 >= L


def test_final_state_is_the_golden_one(cuda, oracle):
    """the oracle (and the engine, equal to it after every call) ends the frozen run in the committed state"""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "late_golden.npz"))
    c = _run(cuda, oracle, **GOLDEN_RUN)
    for k, v in c.items():
        assert np.array_equal(v, gold["craft_" + k]), k
