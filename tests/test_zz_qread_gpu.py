"""MultiPaxos near quorum reads: the HIP kernels `qr_{refresh,read_query,issue,replies}_kernel` through
the C-ABI vs the literal restatement (oracle/qr_oracle.c) on the same seeded streams, bit-exact after every call.
Five replicas of every group each run the responder on their own view of one log (different commit progress, different
highest-slot tables after message loss); the issuer's tally takes their replies with loss, in random delivery orders,
with stable-leader short cuts, repeated replies and (injected) conflicting values."""
import numpy as np
import pytest

from summerset_amd import stream

pytestmark = pytest.mark.gpu

NO = 0xFFFFFFFF


def _u(seed, tag, t, *shape_idx):
    return stream._key(seed, tag, t, *shape_idx)


def _t(a, cuda):
    import torch
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _same_dump(eng, orc, where):
    a, b = eng.dump(), orc.dump()
    for k in b:
        assert np.array_equal(a[k], b[k]), (where, k, np.nonzero(a[k] != b[k]))


def _np(d):
    return {k: v.cpu().numpy().view(np.uint32) if v.dtype.itemsize == 4 else v.cpu().numpy() for k, v in d.items()}


def _run(cuda, oracle, G, R, K, B, Q, W, T, seed):
    """cuda None: the oracles alone (tests/golden/make_golden.py and the CPU test that pins their final state)"""
    engs = None
    if cuda is not None:
        from summerset_amd import QuorumReadGroup
        engs = [QuorumReadGroup(G, R, r, K, B, Q) for r in range(R)]
    orcs = [oracle.QrOracle(G, R, r, K, B, Q) for r in range(R)]
    g = np.arange(G, dtype=np.uint64)
    i_ = np.arange(B, dtype=np.uint64)[:, None]
    r_ = np.arange(R, dtype=np.uint64)[:, None]
    log_len = np.zeros(G, np.uint32)                            # one shared log: slot s of group g writes token below
    token = np.zeros((W, G), np.uint32)
    commit = np.zeros((R, G), np.uint32)                        # replica r has committed slots < commit[r]
    kv = np.zeros((K, G), np.uint32)                            # the state machine of a replica that executed everything
    n_done = 0
    for t in range(T):
        # ---- a new batch per group (some groups none): every replica that hears of it refreshes its table
        has = (_u(seed, 1, t, g) % np.uint64(4)) != 0
        slot = np.where(has, log_len, NO).astype(np.uint32)
        put_keys = (_u(seed, 2, t, i_, g[None, :]) % np.uint64(K + 2)).astype(np.uint8)
        put_keys[put_keys >= K] = 0xFF                           # Gets / non-Puts of the batch
        tok = (1 + t * G + np.arange(G)).astype(np.uint32)
        for s_, gg in zip(slot[has], np.nonzero(has)[0]):
            token[s_ % W, gg] = tok[gg]
            for i in range(B):
                if put_keys[i, gg] != 0xFF:
                    kv[put_keys[i, gg], gg] = tok[gg]
        hears = (_u(seed, 3, t, r_, g[None, :]) % np.uint64(8)) != 0       # a follower may miss the Accept
        hears[0] = True
        for r in range(R):
            sl = np.where(hears[r], slot, NO).astype(np.uint32)
            orcs[r].refresh_highest_slot(sl, put_keys)
            if engs:
                engs[r].refresh_highest_slot(_t(sl, cuda), _t(put_keys, cuda))
        log_len = (log_len + has).astype(np.uint32)
        adv = (_u(seed, 4, t, r_, g[None, :]) % np.uint64(3)).astype(np.uint32)
        commit = np.minimum(commit + adv, log_len[None, :]).astype(np.uint32)
        start = np.where(log_len > W, log_len - W, 0).astype(np.uint32)
        # ---- replica `iss` issues query q = t % Q: its own inspect, then the others' replies
        iss, q = int(t % R), int(t % Q)
        keys = (_u(seed, 5, t, i_, g[None, :]) % np.uint64(K)).astype(np.uint8)
        n = (_u(seed, 6, t, g) % np.uint64(B + 1)).astype(np.uint8)          # 0 = the group issues nothing this tick
        stable = ((_u(seed, 7, t, g) % np.uint64(16)) == 0).astype(np.uint8)  # group has a stable leased leader (replica 0)
        rep = dict(state=np.zeros((R, B, G), np.uint8), slot=np.zeros((R, B, G), np.uint32), val=np.zeros((R, B, G), np.uint32))
        flags = np.zeros((R, G), np.uint8)
        for r in range(R):
            status = np.full((W, G), 2, np.uint8)
            sl_idx = np.arange(W)[:, None]
            # slot of ring row w for group g: the newest slot congruent to w below log_len
            abs_slot = (log_len[None, :].astype(np.int64) - 1 - ((log_len[None, :].astype(np.int64) - 1 - sl_idx) % W))
            status[(abs_slot >= 0) & (abs_slot < commit[r][None, :])] = 3
            log = dict(start_slot=start, log_end=log_len.copy(), status=status, token=token)
            st_in = stable if r == 0 and iss != 0 else None
            o_out, o_fl = orcs[r].handle_read_query(keys, n, log, st_in, kv if st_in is not None else None)
            if engs:
                dlog = {k: _t(v, cuda) for k, v in log.items()}
                e_out, e_fl = engs[r].handle_msg_read_query(_t(keys, cuda), _t(n, cuda), dlog, None if st_in is None else _t(st_in, cuda),
                                                            None if st_in is None else _t(kv, cuda))
                e_np = _np(e_out)
                for k in o_out:
                    assert np.array_equal(e_np[k], o_out[k]), ("read_query", t, r, k)
                assert np.array_equal(e_fl.cpu().numpy(), o_fl), ("from_leader", t, r)
            if r == iss:
                orcs[r].issue(q, n, o_out)
                if engs:
                    engs[r].issue(q, _t(n, cuda), e_out)
                    _same_dump(engs[r], orcs[r], ("issue", t))
            else:
                for k in rep:
                    rep[k][r] = o_out[k]
                lost = (_u(seed, 8, t, np.uint64(r), g) % np.uint64(5)) == 0
                flags[r] = np.where((n > 0) & ~lost, 1 | (o_fl << 1), 0)
        # an occasional corrupted committed value, to reach the conflicting-value error
        bad = (_u(seed, 9, t, r_, g[None, :]) % np.uint64(40)) == 0
        rep["val"][:, 0, :] = np.where(bad & (rep["state"][:, 0, :] == 2), rep["val"][:, 0, :] ^ 0x4000, rep["val"][:, 0, :])
        order = stream.random_ackctl(seed, t, 1, G, R, 0.0)[0]
        halves = [flags.copy(), flags.copy()]
        split = (_u(seed, 10, t, r_, g[None, :]) % np.uint64(2)) == 0       # deliver in two calls; some replies twice
        halves[0][split] = 0
        twice = (_u(seed, 11, t, r_, g[None, :]) % np.uint64(6)) == 0
        halves[1][~split & ~twice] = 0
        for hi, fl in enumerate(halves):
            oo = orcs[iss].handle_replies(q, rep, fl, np.ascontiguousarray(order))
            n_done += int(oo[2].sum())
            if engs:
                ee = engs[iss].handle_msg_read_query_reply(q, {k: _t(v, cuda) for k, v in rep.items()}, _t(fl, cuda), _t(np.ascontiguousarray(order), cuda))
                for name, a, b in zip(("outcome", "out_val", "done"), ee, oo):
                    a = a.cpu().numpy()
                    assert np.array_equal(a.view(b.dtype) if a.dtype.itemsize == 4 else a, b), (name, t, hi)
                _same_dump(engs[iss], orcs[iss], ("replies", t, hi))
    c = sum(o.dump()["counters"] for o in orcs)
    assert n_done > 0 and all(int(x) > 0 for x in c), c           # values, retries, not-founds and conflicts all occurred
    final = {"counters": c}
    for r in range(R):
        for k, v in orcs[r].dump().items():
            if k != "counters":
                final["r%d_%s" % (r, k)] = v
    return final


GOLDEN_RUN = dict(G=64, R=5, K=9, B=3, Q=2, W=16, T=30, seed=77)   # tests/golden/late_golden.npz, "qr_*"


def test_quorum_reads_match_oracle(cuda, oracle):
    _run(cuda, oracle, G=600, R=5, K=12, B=3, Q=2, W=16, T=40, seed=21)


def test_quorum_reads_other_shapes(cuda, oracle):
    _run(cuda, oracle, G=130, R=3, K=5, B=1, Q=1, W=8, T=40, seed=22)
    _run(cuda, oracle, G=257, R=7, K=30, B=5, Q=3, W=32, T=30, seed=23)


def test_responder_reads_the_multipaxos_engines_log_in_place(cuda, oracle):
    """quorum reads on top of the MultiPaxos cluster engine: after some ticks of the bench-like stream (loss, a leader
    change) every replica answers ReadQueries straight from its device-resident log (`smr_mp_replica_log_view`: wave-tiled
    rings, Status inside the meta words); the oracle answers from the dumped state"""
    import test_mp_gpu as t
    from summerset_amd import MultiPaxosCluster, QuorumReadGroup
    G, R, S, W, K, B = 200, 5, 4, 32, 10, 3
    cap = W + 4
    eng = MultiPaxosCluster(G, R, W, win_reserve=W // 8, outbox_cap=cap)
    eng.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=12, drop_p=0.15, timeout_frac=0.02, hb_every=3, rand_rows=S + 4, max_drop=2)
    rng = np.random.default_rng(8)
    qe = [QuorumReadGroup(G, R, r, K, B, 1) for r in range(R)]
    qo = [oracle.QrOracle(G, R, r, K, B, 1) for r in range(R)]
    n_val = 0
    for tick in range(12):
        eng.tick(**t._to_dev(st.tick(tick), cuda))
        if tick % 3 != 2:
            continue
        for r in range(R):
            d = eng.dump(r)
            # highest-slot entries anywhere around the replica's log: below start_slot, inside, at and past the end
            for _ in range(4):
                slot = rng.integers(0, int(d["log_len"].max()) + 3, G).astype(np.uint32)
                pk = rng.integers(0, K + 3, (B, G)).astype(np.uint8); pk[pk >= K] = 0xFF
                qo[r].refresh_highest_slot(slot, pk)
                qe[r].refresh_highest_slot(_t(slot, cuda), _t(pk, cuda))
            keys = rng.integers(0, K, (B, G)).astype(np.uint8)
            n = rng.integers(0, B + 1, G).astype(np.uint8)
            log = dict(start_slot=d["start_slot"], log_end=d["log_len"], status=d["s_status"], token=d["s_reqs"])
            o_out, _ = qo[r].handle_read_query(keys, n, log)
            e_out, _ = qe[r].handle_msg_read_query(_t(keys, cuda), _t(n, cuda), eng.replica_log_view(r))
            e_np = _np(e_out)
            for k in o_out:
                assert np.array_equal(e_np[k], o_out[k]), (tick, r, k, np.nonzero(e_np[k] != o_out[k]))
            n_val += int((o_out["state"] == 2).sum())
    assert n_val > 100


def test_final_state_is_the_golden_one(cuda, oracle):
    """the oracles (and the engines, equal to them after every call) end the frozen run in the committed state"""
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "late_golden.npz"))
    for k, v in _run(cuda, oracle, **GOLDEN_RUN).items():
        assert np.array_equal(v, gold["qr_" + k]), k
