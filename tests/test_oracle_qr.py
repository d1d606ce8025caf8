"""Pins oracle/qr_oracle.c (MultiPaxos near quorum reads) by traces worked out by hand from the reference's rules:
multipaxos/quorumread.rs:8-26 (highest-slot table), :30-73 (inspect), :75-188 (responder), :190-346 (the issuer's merge,
read quorum and answers), request.rs:55-101 (bookkeeping).  CPU only."""
import numpy as np
import pytest

G, R, K, B, Q, W = 2, 5, 8, 3, 2, 8
NONE, SLOT, VALUE = 0, 1, 2
PENDING, NOT_FOUND, RETRY, GOT = 0, 1, 2, 3
NO = 0xFFFFFFFF


@pytest.fixture()
def orc(oracle):
    return oracle.QrOracle(G, R, me=0, K=K, B=B, Q=Q)


def _log(start=0, length=6, committed=(), tokens=None):
    status = np.full((W, G), 2, np.uint8)                         # Accepting
    for s in committed:
        status[s % W] = 3
    token = np.zeros((W, G), np.uint32)
    for s in range(W):
        token[s] = 100 + s if tokens is None else tokens.get(s, 0)
    return dict(start_slot=np.full(G, start, np.uint32), log_end=np.full(G, start + length, np.uint32), status=status, token=token)


def _keys(*ks):
    a = np.zeros((B, G), np.uint8)
    for i, k in enumerate(ks):
        a[i] = k
    return a, np.full(G, len(ks), np.uint8)


def _puts(*ks):
    a = np.full((B, G), 0xFF, np.uint8)
    for i, k in enumerate(ks):
        a[i] = k
    return a


def _rep(rows):
    """rows = {peer: [(state, slot, val), ...]} for every group alike -> replies [R, B, G], flags [R, G]"""
    d = dict(state=np.zeros((R, B, G), np.uint8), slot=np.zeros((R, B, G), np.uint32), val=np.zeros((R, B, G), np.uint32))
    fl = np.zeros((R, G), np.uint8)
    for p, (lst, leader) in rows.items():
        fl[p] = 1 | (2 if leader else 0)
        for i, (st, sl, v) in enumerate(lst):
            d["state"][p, i] = st; d["slot"][p, i] = sl; d["val"][p, i] = v
    return d, fl


def test_highest_slot_and_inspect(orc):
    orc.refresh_highest_slot(np.array([3, NO], np.uint32), _puts(1, 2))     # group 1 has no batch
    orc.refresh_highest_slot(np.array([5, 2], np.uint32), _puts(2))
    orc.refresh_highest_slot(np.array([4, 1], np.uint32), _puts(2, 0xFF, 1))  # lower slots never lower an entry
    hs = orc.dump()["highest_slot"]
    assert hs[1].tolist() == [4, 1] and hs[2].tolist() == [5, 2] and (hs[0] == NO).all()
    keys, n = _keys(1, 2, 7)
    out, fl = orc.handle_read_query(keys, n, _log(length=6, committed=(4, 2)))
    assert fl.tolist() == [0, 0]
    # group 0: key 1 at slot 4 committed -> value 104; key 2 at slot 5 not committed; key 7 never seen
    assert out["state"][:, 0].tolist() == [VALUE, SLOT, NONE] and out["slot"][:, 0].tolist() == [4, 5, 0] and out["val"][:, 0].tolist() == [104, 0, 0]
    # group 1: key 1 at slot 1 (Accepting), key 2 at slot 2 committed
    assert out["state"][:, 1].tolist() == [SLOT, VALUE, NONE] and out["val"][:, 1].tolist() == [0, 102, 0]
    # a slot outside [start_slot, start_slot + len) counts as not committed (:38-43)
    out, _ = orc.handle_read_query(keys, n, _log(start=5, length=1, committed=(4, 5, 2)))
    assert out["state"][:, 0].tolist() == [SLOT, VALUE, NONE] and out["state"][:, 1].tolist() == [SLOT, SLOT, NONE]
    out, _ = orc.handle_read_query(keys, n, _log(start=0, length=4, committed=(4, 5, 2)))
    assert out["state"][:, 0].tolist() == [SLOT, SLOT, NONE]


def test_stable_leader_answers_from_the_state_machine(orc):
    orc.refresh_highest_slot(np.array([3, 3], np.uint32), _puts(1))
    kv = np.zeros((K, G), np.uint32); kv[1] = [55, 0]
    keys, n = _keys(1, 6)
    out, fl = orc.handle_read_query(keys, n, _log(), stable_leader=np.array([1, 1], np.uint8), kv=kv)
    assert fl.tolist() == [1, 1]
    assert out["state"][:, 0].tolist() == [VALUE, NONE, NONE] and out["slot"][:, 0].tolist() == [0, 0, 0] and out["val"][0].tolist() == [55, 0]
    assert out["state"][:, 1].tolist() == [NONE, NONE, NONE]
    out, fl = orc.handle_read_query(keys, np.array([2, 0], np.uint8), _log(), stable_leader=np.array([0, 1], np.uint8), kv=kv)
    assert fl.tolist() == [0, 0] and out["state"][:, 0].tolist() == [SLOT, NONE, NONE]    # n = 0: no message, no reply


def _issue(orc, own):
    d = dict(state=np.zeros((B, G), np.uint8), slot=np.zeros((B, G), np.uint32), val=np.zeros((B, G), np.uint32))
    for i, (st, sl, v) in enumerate(own):
        d["state"][i] = st; d["slot"][i] = sl; d["val"][i] = v
    orc.issue(0, np.full(G, len(own), np.uint8), d)


def test_read_quorum_merge_and_answers(orc):
    # three reads; I know: read 0 nothing, read 1 slot 4 uncommitted, read 2 slot 3 = 30
    _issue(orc, [(NONE, 0, 0), (SLOT, 4, 0), (VALUE, 3, 30)])
    d = orc.dump()
    assert d["live"][0].tolist() == [1, 1] and d["rq_acks"][0].tolist() == [1, 1] and d["n"][0].tolist() == [3, 3]
    # peer 1: read 0 committed 2 = 20 (nothing known before: only the slot is kept, :231-233), read 1 slot 4 = 40
    # (>= an uncommitted 4: taken), read 2 an older committed slot (ignored)
    rep, fl = _rep({1: ([(VALUE, 2, 20), (VALUE, 4, 40), (VALUE, 1, 10)], False)})
    outcome, val, done = orc.handle_replies(0, rep, fl)
    assert done.tolist() == [0, 0] and (outcome == PENDING).all()
    d = orc.dump()
    assert d["rq_acks"][0].tolist() == [3, 3]
    assert d["mx_state"][0, :, 0].tolist() == [SLOT, VALUE, VALUE] and d["mx_slot"][0, :, 0].tolist() == [2, 4, 3]
    assert d["mx_val"][0, :, 0].tolist() == [0, 40, 30]
    # the same peer again: already counted (:211), nothing changes
    orc.handle_replies(0, *_rep({1: ([(VALUE, 9, 90), (VALUE, 9, 90), (VALUE, 9, 90)], False)}))
    assert np.array_equal(orc.dump()["mx_slot"], d["mx_slot"])
    # peer 3: read 0 committed 2 = 20 again (now taken, slot >= 2), read 1 a higher uncommitted slot (drops the value),
    # read 2 a higher committed slot; with me + 1 + 3 = quorum of 3 the clients are answered
    outcome, val, done = orc.handle_replies(0, *_rep({3: ([(VALUE, 2, 20), (SLOT, 6, 0), (VALUE, 5, 50)], False)}))
    assert done.tolist() == [1, 1]
    assert outcome[:, 0].tolist() == [GOT, RETRY, GOT] and val[:, 0].tolist() == [20, 0, 50]
    d = orc.dump()
    assert d["live"][0].tolist() == [0, 0] and d["counters"].tolist() == [4, 2, 0, 0]
    # a late reply finds no bookkeeping (:205)
    outcome, val, done = orc.handle_replies(0, *_rep({4: ([(VALUE, 7, 70)] * 3, False)}))
    assert done.tolist() == [0, 0] and (outcome == PENDING).all()


def test_not_found_and_leader_shortcut_and_conflict(orc):
    _issue(orc, [(NONE, 0, 0), (VALUE, 3, 30)])
    rep, fl = _rep({2: ([(NONE, 0, 0), (VALUE, 3, 31)], False)})        # same slot, another value: logged_err (:243-250)
    outcome, val, done = orc.handle_replies(0, rep, fl)
    d = orc.dump()
    assert done.tolist() == [0, 0] and d["rq_acks"][0].tolist() == [1, 1] and d["counters"][3] == G   # not counted as replied
    rep, fl = _rep({1: ([(NONE, 0, 0), (SLOT, 3, 0)], False), 4: ([(NONE, 0, 0), (NONE, 0, 0)], False)})
    outcome, val, done = orc.handle_replies(0, rep, fl)                  # two replies in one call: quorum at the second
    assert done.tolist() == [1, 1] and outcome[:, 0].tolist() == [NOT_FOUND, GOT, PENDING] and val[:, 0].tolist() == [0, 30, 0]
    # a stable leader's reply replaces everything and answers at once, whatever was known (:206-210)
    _issue(orc, [(SLOT, 9, 0), (VALUE, 3, 30)])
    outcome, val, done = orc.handle_replies(0, *_rep({2: ([(VALUE, 0, 77), (NONE, 0, 0)], True)}))
    assert done.tolist() == [1, 1] and outcome[:, 0].tolist() == [GOT, NOT_FOUND, PENDING] and val[:, 0].tolist() == [77, 0, 0]


def test_delivery_order_decides_which_replies_count(orc):
    _issue(orc, [(SLOT, 1, 0)])
    rows = {1: ([(VALUE, 1, 11)], False), 2: ([(SLOT, 2, 0)], False), 3: ([(VALUE, 2, 22)], False)}
    rep, fl = _rep(rows)
    # identity order: peers 1, 2 make the quorum: (2, None) -> retry; peer 3's committed value comes too late
    outcome, val, done = orc.handle_replies(0, rep, fl)
    assert outcome[0].tolist() == [RETRY, RETRY]
    _issue(orc, [(SLOT, 1, 0)])
    order = np.full(G, 3 | (2 << 3) | (1 << 6) | (0 << 9) | (4 << 12), np.uint32)   # peers 3, 2 first
    outcome, val, done = orc.handle_replies(0, rep, fl, order)
    assert outcome[0].tolist() == [GOT, GOT] and val[0].tolist() == [22, 22]


def test_reads_never_miss_an_acknowledged_write(oracle):
    """the restatement against the optimisation's own safety argument (Charapko et al., HotStorage '19, cited at
    request.rs:59-61) on logs that are consistent the way MultiPaxos keeps them: one value per slot, a slot is chosen once
    a majority holds it, a replica marks a slot committed only if it is chosen and it holds it, the leader (replica 0)
    holds everything.  Whatever a read quorum answers -- any issuer, any 2 other repliers, any delivery order -- a read
    that returns a value returns the value of the highest slot any member of the quorum has seen for the key, and that
    slot is not below the highest write the leader has acknowledged (committed); a read that says "not found" implies no
    acknowledged write to the key.  Reads that cannot tell are sent to the slow path, never answered wrong."""
    rng = np.random.default_rng(11)
    G, R, K, B, W, S = 400, 5, 6, 3, 32, 20
    orcs = [oracle.QrOracle(G, R, r, K, B, 1) for r in range(R)]
    key_of = rng.integers(0, K + 2, (S, G)); key_of[key_of >= K] = 0xFF        # slot s of group g puts key_of (0xFF: no Put)
    token = (1000 + np.arange(S)[:, None] * G + np.arange(G)[None, :]).astype(np.uint32)
    holds = rng.random((R, S, G)) < 0.7; holds[0] = True                        # who accepted which slot
    chosen = holds.sum(0) >= 3
    committed = holds & chosen[None] & (rng.random((R, S, G)) < 0.6)
    for s in range(S):
        pk = np.full((B, G), 0xFF, np.uint8); pk[0] = key_of[s]
        for r in range(R):
            orcs[r].refresh_highest_slot(np.where(holds[r, s], s, NO).astype(np.uint32), pk)
    tok_ring = np.zeros((W, G), np.uint32); tok_ring[:S] = token
    checked = {GOT: 0, RETRY: 0, NOT_FOUND: 0}
    for trial in range(12):
        keys = rng.integers(0, K, (B, G)).astype(np.uint8)
        n = np.full(G, B, np.uint8)
        views = []
        for r in range(R):
            status = np.full((W, G), 2, np.uint8); status[:S][committed[r]] = 3
            log = dict(start_slot=np.zeros(G, np.uint32), log_end=np.full(G, S, np.uint32), status=status, token=tok_ring)
            views.append(orcs[r].handle_read_query(keys, n, log)[0])
        iss = trial % R
        orcs[iss].issue(0, n, views[iss])
        rep = {k: np.stack([views[r][k] for r in range(R)]) for k in ("state", "slot", "val")}
        order = np.array([sum(int(p) << (3 * i) for i, p in enumerate(rng.permutation(R))) for _ in range(G)], np.uint32)
        fl = np.ones((R, G), np.uint8); fl[iss] = 0
        outcome, val, done = orcs[iss].handle_replies(0, rep, fl, order)
        assert done.all()
        for g in range(G):
            quorum = [iss] + [p for p in [(int(order[g]) >> (3 * i)) & 7 for i in range(R)] if p != iss][:2]
            for i in range(B):
                k = keys[i, g]
                writes = np.nonzero(key_of[:, g] == k)[0]
                acked = [s for s in writes if committed[0, s, g]]
                seen = [s for s in writes if any(holds[r, s, g] for r in quorum)]
                o = outcome[i, g]
                checked[int(o)] += 1
                if o == GOT:
                    assert val[i, g] == token[max(seen), g] and (not acked or max(seen) >= max(acked))
                elif o == NOT_FOUND:
                    assert not seen and not acked
                else:
                    assert o == RETRY and seen
    assert all(v > 100 for v in checked.values()), checked


def test_golden_final_state(oracle):
    """the frozen run of tests/test_zz_qread_gpu.py, oracle alone, ends in the committed state (tests/golden/late_golden.npz)"""
    import os
    import test_zz_qread_gpu as t
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "late_golden.npz"))
    out = t._run(None, oracle, **t.GOLDEN_RUN)
    assert out and all(np.array_equal(v, gold["qr_" + k]) for k, v in out.items())
