"""MultiPaxos lock-step parity: HIP engine (through the C-ABI) vs the CPU oracle
on identical seeded streams, compared after EVERY tick on the full canonical
state of all replicas (bit-exact)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _to_dev(t, cuda):
    import torch
    return {k: (torch.from_numpy(v).to(cuda) if isinstance(v, np.ndarray) else v) for k, v in t.items()}


def _compare(eng, orc, R, tick, check_slots=True):
    from oracle.oracle import MP_SCALARS, MP_SLOTS
    for r in range(R):
        a, b = eng.dump(r), orc.dump(r)
        assert np.array_equal(a["overflow"], b["overflow"]), "tick %d rep %d overflow flags differ" % (tick, r)
        live = b["overflow"] == 0
        for n in MP_SCALARS:
            if not np.array_equal(a[n][live], b[n][live]):
                bad = np.nonzero((a[n] != b[n]) & live)[0][:5]
                raise AssertionError("tick %d rep %d %s differs at groups %s: hip %s oracle %s"
                                     % (tick, r, n, bad, a[n][bad], b[n][bad]))
        assert np.array_equal(a["peer_exec_bar"][:, live], b["peer_exec_bar"][:, live]), (tick, r)
        if check_slots:
            for n, _ in MP_SLOTS:
                if not np.array_equal(a[n][:, live], b[n][:, live]):
                    w, g = np.nonzero((a[n] != b[n]) & live[None, :])
                    raise AssertionError("tick %d rep %d %s differs at (w,g) %s: hip %s oracle %s"
                                         % (tick, r, n, list(zip(w[:5], g[:5])), a[n][w[:5], g[:5]], b[n][w[:5], g[:5]]))


def _acks_as_records(eng, cuda, G, R, cap, t):
    """between R2 and R3: every replica's ack matrix taken out as AcceptReply records (smr_mp_collect_acks), the matrix
    zeroed, the records -- shuffled, with records that answer nothing mixed in -- put back (smr_mp_deliver_acks); a
    second collect must give the same set, and the tick must still end in the oracle's state"""
    import torch
    from summerset_amd.multipaxos import ACK_DTYPE
    rng = np.random.default_rng(1000 + t)
    room = cap * G * R
    for r in range(R):
        out = torch.zeros(room * ACK_DTYPE.itemsize, dtype=torch.uint8, device=cuda)
        n = torch.zeros(1, dtype=torch.int64, device=cuda)
        eng.collect_acks(r, out, n)
        n0 = int(n.item())
        assert n0 <= room
        rec = out.cpu().numpy().view(ACK_DTYPE)[:n0].copy()
        assert (rec["peer"] != r).all() and (rec["peer"] < R).all() and (rec["group"] < G).all()
        junk = np.zeros(7, ACK_DTYPE)                          # must all be ignored and counted
        if n0:
            junk[:] = rec[rng.integers(0, n0, 7)]
        junk["ballot"][0] += 1 << 8                            # an AcceptReply of another ballot
        junk["slot"][1] += 100000                              # of a slot I sent no Accept for
        junk["group"][2] = G                                   # out of range
        junk["peer"][3] = R
        junk["peer"][4] = r                                    # my own id
        junk["ballot"][5] = 0
        junk["slot"][6] = (1 << 30) - 1
        mixed = np.concatenate([rec, junk])
        mixed = mixed[rng.permutation(len(mixed))]
        dev = torch.from_numpy(mixed.view(np.uint8).reshape(-1).copy()).to(cuda)
        dropped = torch.zeros(1, dtype=torch.int64, device=cuda)
        eng.clear_acks(r)
        eng.collect_acks(r, out, n)
        assert int(n.item()) == 0, "cleared matrix still holds acknowledgements"
        eng.deliver_acks(r, dev, len(mixed), dropped)
        eng.collect_acks(r, out, n)
        assert int(dropped.item()) == len(junk), (t, r, int(dropped.item()))
        assert int(n.item()) == n0, (t, r, int(n.item()), n0)
        back = out.cpu().numpy().view(ACK_DTYPE)[:n0]
        assert np.array_equal(np.sort(back, order=list(ACK_DTYPE.names)), np.sort(rec, order=list(ACK_DTYPE.names))), (t, r)


def _run(cuda, oracle, G, R, S, W, n_ticks, drop_p, timeout_frac, hb_every, preset, commit_extra=0, seed=None,
         every=1, timeout_rep=1, straggler_ticks=0, per_round=False, fused=0, rotate=False, no_array_when_quiet=False):
    from summerset_amd import MultiPaxosCluster, stream
    cap = W + 4
    eng = MultiPaxosCluster(G, R, W, outbox_cap=cap, commit_extra=commit_extra, commit_list_cap=G * (S * 4 * max(fused, 1) + W) + 64,
                            straggler_ticks=straggler_ticks)
    if rotate:                              # rows of the bulk launches by role (smr_mp_set_role_rotation): same results
        eng.set_role_rotation(True)
    orc = oracle.MpOracle(G, R, W, cap=cap, commit_extra=commit_extra)
    if preset:
        eng.preset_leader(0)
        orc.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=n_ticks, drop_p=drop_p, timeout_frac=timeout_frac,
                                 hb_every=hb_every, seed=seed or stream.DEFAULT_SEED, timeout_rep=timeout_rep)
    if not preset:
        # natural bootstrap: replica 0 times out first (leader None)
        t0 = dict(timeout_rep=np.zeros(G, np.uint8), timeout_src=np.full(G, 0xFF, np.uint8),
                  ackctl=np.full((cap, G), stream.CTL_IDENTITY, np.uint32), heartbeat=False)
        orc.tick(**t0)
        eng.tick(**_to_dev(t0, cuda))
        _compare(eng, orc, R, -1)
    total = 0
    pending = []                            # fused: ticks collected for one smr_mp_run_ticks call
    og_acc = [[np.zeros(0, np.uint32)] * 2 for _ in range(R)]
    for t in range(n_ticks):
        inp = st.tick(t)
        orc.tick(**inp)
        if fused:                           # batches of `fused` ticks through the fused tick kernel (one launch each)
            if no_array_when_quiet and not (inp["timeout_rep"] != 0xFF).any():   # a host knows when no timer fired: no array at all,
                inp = dict(inp, timeout_rep=None, timeout_src=None)              # which is what lets the engine call a stretch quiet
            pending.append(_to_dev(inp, cuda))
            for r in range(R):
                g_, s_ = orc.take_commits(r)
                og_acc[r] = [np.concatenate([og_acc[r][0], g_]), np.concatenate([og_acc[r][1], s_])]
            if len(pending) < fused and t != n_ticks - 1:
                continue
            eng.run_ticks(pending)
            pending = []
        elif per_round:                     # the four rounds as separate C-ABI calls
            d = _to_dev(inp, cuda)
            eng.round_local(d["timeout_rep"], d["timeout_src"], d["req_target"], d["req_cnt"], d["req_val"])
            eng.round_deliver()
            if per_round == "records":
                _acks_as_records(eng, cuda, G, R, cap, t)
            elif callable(per_round):       # a test's own step between R2 and R3 (tests/test_zz_wire_ingest_gpu.py)
                per_round(eng, cuda, G, R, cap, t)
            eng.round_replies(d["ackctl"], publish_heartbeat=inp["heartbeat"])
            if inp["heartbeat"]:
                eng.round_heartbeat()
            eng.end_tick()
        else:
            eng.tick(**_to_dev(inp, cuda))
        if fused or t % every == 0 or t == n_ticks - 1:
            _compare(eng, orc, R, t)
        # ordered committed-slot list of the (possibly several) leaders
        for r in range(R):
            og, os_ = og_acc[r] if fused else orc.take_commits(r)
            og_acc[r] = [np.zeros(0, np.uint32)] * 2
            eg, es = eng.poll_commits(r)
            assert len(og) == len(eg), "tick %d rep %d commit count %d vs %d" % (t, r, len(eg), len(og))
            ko = np.lexsort((np.arange(len(og)), og))
            ke = np.argsort(eg, kind="stable")
            assert np.array_equal(og[ko], eg[ke]) and np.array_equal(os_[ko], es[ke]), (t, r)
            total += len(og)
    for r in range(R):
        assert eng.counters(r)["commits"] == orc.total_commits(r)
    assert total > 0
    return eng, orc


def test_steady_state_no_loss(cuda, oracle):
    _run(cuda, oracle, G=300, R=5, S=1, W=32, n_ticks=40, drop_p=0.0, timeout_frac=0.0, hb_every=4, preset=True)


def test_steady_state_drops_s4(cuda, oracle):
    _run(cuda, oracle, G=1000, R=5, S=4, W=64, n_ticks=60, drop_p=0.1, timeout_frac=0.0, hb_every=3, preset=True)


@pytest.mark.parametrize("straggler_ticks", [0, 1, 0xFF])
def test_leader_change(cuda, oracle, straggler_ticks):
    # every group sees a HearTimeout on replica 1 at some tick; groups in a leader change run on the
    # engine's side stream for the default 4 ticks / 1 tick / never -- same results
    _run(cuda, oracle, G=512, R=5, S=2, W=64, n_ticks=48, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True,
         straggler_ticks=straggler_ticks)


def test_leader_change_round_by_round(cuda, oracle):
    _run(cuda, oracle, G=300, R=5, S=2, W=64, n_ticks=40, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True,
         per_round=True)


def test_accept_replies_as_records(cuda, oracle):
    """smr_mp_collect_acks / smr_mp_clear_acks / smr_mp_deliver_acks: the acknowledgements of every tick -- steady
    appends (regular outbox), losses, leader changes with their long re-Accept outboxes -- go through the record form
    and the cluster still matches the oracle after every tick"""
    _run(cuda, oracle, G=200, R=5, S=2, W=64, n_ticks=36, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True,
         per_round="records")
    _run(cuda, oracle, G=70, R=3, S=4, W=32, n_ticks=12, drop_p=0.0, timeout_frac=0.0, hb_every=3, preset=True,
         per_round="records")


def test_leader_change_straggler_list_overflow(cuda, oracle):
    # 4096 groups time out within 6 ticks: more than the side stream's list takes per tick, the rest
    # must be handled by the bulk launches
    _run(cuda, oracle, G=4096, R=5, S=2, W=64, n_ticks=12, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True,
         every=3)


def test_natural_bootstrap_noop_slot(cuda, oracle):
    # leader None -> replica 0 steps up on an empty log: trigger == log end pushes a no-op (A.8 #3)
    _run(cuda, oracle, G=128, R=5, S=1, W=64, n_ticks=30, drop_p=0.05, timeout_frac=0.3, hb_every=4, preset=False)


def test_three_replicas(cuda, oracle):
    _run(cuda, oracle, G=257, R=3, S=2, W=32, n_ticks=40, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True)


def test_rspaxos_threshold(cuda, oracle):
    # RSPaxos commit rule: majority + fault_tolerance acks (rspaxos/messages.rs:438-439)
    eng, orc = _run(cuda, oracle, G=400, R=5, S=2, W=64, n_ticks=40, drop_p=0.15, timeout_frac=0.0, hb_every=4,
                    preset=True, commit_extra=1)
    d = eng.dump(0)
    committed = d["s_status"] >= 3
    acks = np.unpackbits(d["s_acks"][:, :, None], axis=2).sum(axis=2)
    assert (acks[committed & (d["s_flags"] & 1 == 1)] == 4).all()      # mask froze at exactly majority + f


def test_window_backpressure_and_overflow_flags(cuda, oracle):
    # tiny window, no heartbeats for long stretches: leader must refuse batches identically
    eng, orc = _run(cuda, oracle, G=64, R=5, S=3, W=16, n_ticks=40, drop_p=0.0, timeout_frac=0.0, hb_every=8,
                    preset=True)
    assert eng.counters(0)["rejects"] > 0


def test_window_overflow_while_the_r3_rest_rides_in_the_next_r1(cuda, oracle):
    """in a quiet stretch (no HearTimeout for 2 x 16 + ttl ticks) smr_mp_run_ticks defers the rest of a tick's R3 into the
    next tick's R1 launch (mp_rest_then_local), where the group's `overflow` flag is re-read behind the block's own R3 only.  A
    tiny window with heartbeats far apart makes groups refuse batches and freeze in exactly such stretches; flags and every
    unfrozen group's state must stay the oracle's, batch after batch.  (What makes the fused launch safe: without Prepare traffic
    the rest of R3 has no path that freezes a group -- the window and outbox checks sit in R1 / R2 and in the PrepareReply
    handlers -- so no replica's R1 can miss a flag another replica's deferred R3 would have set.)"""
    for W, S, hb in ((16, 3, 8), (16, 5, 16), (32, 4, 12)):
        eng, _ = _run(cuda, oracle, G=96, R=5, S=S, W=W, n_ticks=96, drop_p=0.05, timeout_frac=0.0, hb_every=hb, preset=True, fused=8,
                      straggler_ticks=4, every=8, no_array_when_quiet=True)
        assert eng.counters(0)["rejects"] > 0


def test_config2_4096_groups(cuda, oracle):
    # BASELINE.json configs[1]: 4096 groups x 5 replicas, 10 % ack loss, 1 % leader timeouts
    _run(cuda, oracle, G=4096, R=5, S=1, W=64, n_ticks=128, drop_p=0.1, timeout_frac=0.01, hb_every=4,
         preset=True, every=8)


def _run_bench_shape(cuda, oracle, G, frac, span, n_ticks, straggler_ticks=0, every=6, log=None, fused=0, rotate=False):
    """bench.py's shape (S=32, W=512, pooled tick inputs, <= 2 acks lost per slot) with leader
    changes: logs of 100+ slots go through the long-outbox / cooperative paths the small shapes
    above never reach."""
    from summerset_amd import MultiPaxosCluster, stream
    R, S, W, H = 5, 32, 512, 4
    cap = W + 4
    eng = MultiPaxosCluster(G, R, W, win_reserve=W // 8, outbox_cap=cap, straggler_ticks=straggler_ticks)
    if rotate:
        eng.set_role_rotation(True)
    orc = oracle.MpOracle(G, R, W, win_reserve=W // 8, cap=cap, record_commits=False)
    eng.preset_leader(0)
    orc.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=n_ticks, drop_p=0.1, timeout_frac=frac, hb_every=H,
                                 rand_rows=S + 4, max_drop=2, timeout_span=span)
    pool = [st.tick(t) for t in range(4)]
    batch = []
    for t in range(n_ticks):
        inp = dict(pool[t % 4])
        inp.update(st.tick_events(t))
        inp["heartbeat"] = st.heartbeat(t)
        orc.tick(**inp)
        if not (inp["timeout_rep"] != 0xFF).any():     # a host knows when no timer fired: no array at all
            inp["timeout_rep"] = inp["timeout_src"] = None
        if fused:                                      # `fused` ticks per smr_mp_run_ticks call
            batch.append(_to_dev(inp, cuda))
            if len(batch) == fused or t == n_ticks - 1 or t % every == every - 1:
                eng.run_ticks(batch)
                batch = []
        else:
            eng.tick(**_to_dev(inp, cuda))
        if t % every == every - 1 or t == n_ticks - 1:
            _compare(eng, orc, R, t)
            if log:
                log("tick %d ok; rejects %s" % (t, [eng.counters(r)["rejects"] for r in range(R)]))
    for r in range(R):
        assert eng.counters(r)["commits"] == orc.total_commits(r)
    return eng, orc


@pytest.mark.parametrize("straggler_ticks", [0, 4])
def test_bench_shape_leader_changes(cuda, oracle, straggler_ticks):
    _run_bench_shape(cuda, oracle, G=1024, frac=0.25, span=10, n_ticks=30, straggler_ticks=straggler_ticks)


# ---- the fused tick kernel (smr_mp_run_ticks): batches of ticks in one launch, same results as tick by tick ----
@pytest.mark.parametrize("fused", [1, 3, 16, 20])
def test_fused_ticks_leader_change(cuda, oracle, fused):
    _run(cuda, oracle, G=512, R=5, S=2, W=64, n_ticks=48, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=fused)


def test_fused_ticks_other_shapes(cuda, oracle):
    _run(cuda, oracle, G=1000, R=5, S=4, W=64, n_ticks=60, drop_p=0.1, timeout_frac=0.0, hb_every=3, preset=True, fused=7)
    _run(cuda, oracle, G=128, R=5, S=1, W=64, n_ticks=30, drop_p=0.05, timeout_frac=0.3, hb_every=4, preset=False, fused=4)
    _run(cuda, oracle, G=257, R=3, S=2, W=32, n_ticks=40, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True, fused=5)
    _run(cuda, oracle, G=100, R=7, S=2, W=64, n_ticks=30, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=6)
    _run(cuda, oracle, G=400, R=5, S=2, W=64, n_ticks=40, drop_p=0.15, timeout_frac=0.0, hb_every=4, preset=True, commit_extra=1, fused=16)
    eng, _ = _run(cuda, oracle, G=64, R=5, S=3, W=16, n_ticks=40, drop_p=0.0, timeout_frac=0.0, hb_every=8, preset=True, fused=9)
    assert eng.counters(0)["rejects"] > 0


def test_fused_ticks_bench_shape(cuda, oracle):
    _run_bench_shape(cuda, oracle, G=1024, frac=0.25, span=10, n_ticks=30, fused=16)


# ---- smr_mp_run_ticks with the straggler list on: the list's groups run the batch's ticks in one side-stream launch ----
@pytest.mark.parametrize("fused,sticks", [(16, 8), (5, 2), (20, 3), (1, 4)])
def test_batched_ticks_with_the_straggler_list(cuda, oracle, fused, sticks):
    _run(cuda, oracle, G=512, R=5, S=2, W=64, n_ticks=48, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=fused,
         straggler_ticks=sticks)
    _run(cuda, oracle, G=300, R=5, S=1, W=64, n_ticks=40, drop_p=0.05, timeout_frac=0.3, hb_every=4, preset=True, fused=fused,
         straggler_ticks=sticks)


def test_batched_ticks_with_r2_split_into_fast_path_and_rest(cuda, oracle, monkeypatch):
    """SMR_MP_SPLIT_R2 (round 5; off by default: it did not pay, profiles/r8i): beside the side stream R2's bulk launch is the fast
    path alone (mp_round_deliver: 79 VGPRs, no scratch) + mp_round_deliver_rest for the lanes it could not finish.  Leader changes
    all over the run with a short ttl, so that groups come back to the bulk kernels mid-change and the rest launch has work."""
    monkeypatch.setenv("SMR_MP_SPLIT_R2", "1")
    _run(cuda, oracle, G=512, R=5, S=2, W=64, n_ticks=48, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=8, straggler_ticks=1)
    _run(cuda, oracle, G=300, R=5, S=3, W=64, n_ticks=40, drop_p=0.05, timeout_frac=0.5, hb_every=4, preset=True, fused=5, straggler_ticks=2)
    _run(cuda, oracle, G=257, R=3, S=2, W=32, n_ticks=40, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True, fused=16, straggler_ticks=1)
    _run_bench_shape(cuda, oracle, G=1024, frac=0.25, span=10, n_ticks=30, fused=8, straggler_ticks=1)


def test_batched_ticks_with_the_straggler_list_other_shapes(cuda, oracle):
    _run(cuda, oracle, G=257, R=3, S=2, W=32, n_ticks=40, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True, fused=5, straggler_ticks=4)
    _run(cuda, oracle, G=100, R=7, S=2, W=64, n_ticks=30, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=6, straggler_ticks=8)
    _run(cuda, oracle, G=128, R=5, S=1, W=64, n_ticks=30, drop_p=0.05, timeout_frac=0.3, hb_every=4, preset=False, fused=4, straggler_ticks=2)
    # more groups with a timeout in one batch than the list holds (1024): the rest stay with the bulk kernels
    _run(cuda, oracle, G=3000, R=5, S=2, W=64, n_ticks=20, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=16, straggler_ticks=8)


def test_batched_ticks_with_the_straggler_list_bench_shape(cuda, oracle):
    _run_bench_shape(cuda, oracle, G=1024, frac=0.25, span=10, n_ticks=30, fused=16, straggler_ticks=8)
    _run_bench_shape(cuda, oracle, G=1024, frac=0.25, span=10, n_ticks=30, fused=7, straggler_ticks=4, every=3)


def test_more_batches_per_tick_than_r1_prefetches(cuda, oracle):
    """S = 40 > R1_PF = 32: the tail of a tick's client batches goes through the loop that loads its tokens itself"""
    _run(cuda, oracle, G=300, R=5, S=40, W=256, n_ticks=16, drop_p=0.05, timeout_frac=0.0, hb_every=4, preset=True)
    _run(cuda, oracle, G=200, R=5, S=33, W=256, n_ticks=12, drop_p=0.1, timeout_frac=0.3, hb_every=3, preset=True, straggler_ticks=2)


def test_batches_and_single_ticks_share_the_list(cuda, oracle):
    """a batch leaves ttl behind that the per-tick mark pass picks up, and the other way round"""
    from summerset_amd import MultiPaxosCluster, stream
    G, R, S, W = 256, 5, 2, 64
    cap = W + 4
    eng = MultiPaxosCluster(G, R, W, outbox_cap=cap, commit_list_cap=G * (S * 64 + W) + 64, straggler_ticks=6)
    orc = oracle.MpOracle(G, R, W, cap=cap)
    eng.preset_leader(0); orc.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=36, drop_p=0.1, timeout_frac=1.0, hb_every=4)
    t = 0
    for chunk in (3, 1, 1, 5, 1, 16, 1, 1, 7):
        ins = []
        for _ in range(chunk):
            inp = st.tick(t); t += 1
            orc.tick(**inp)
            ins.append(_to_dev(inp, cuda))
        if chunk == 1:
            eng.tick(**ins[0])
        else:
            eng.run_ticks(ins)
        _compare(eng, orc, R, t)
    for r in range(R):
        assert eng.counters(r)["commits"] == orc.total_commits(r)


def test_role_rotation_is_bit_exact(cuda, oracle):
    """smr_mp_set_role_rotation: row y of the bulk round launches runs replica (y + leader[g]) mod R of group g -- leaders in row
    0, followers behind -- instead of replica y.  Leader changes all over the run (so the rotation differs from group to group
    and from tick to tick), loss, heartbeats; one call per tick and batches; 3 / 5 / 7 replicas; the bench's own shape"""
    _run(cuda, oracle, G=300, R=5, S=2, W=64, n_ticks=40, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, straggler_ticks=2, rotate=True)
    _run(cuda, oracle, G=257, R=3, S=2, W=32, n_ticks=40, drop_p=0.2, timeout_frac=0.5, hb_every=2, preset=True, fused=5, straggler_ticks=4, rotate=True)
    _run(cuda, oracle, G=100, R=7, S=2, W=64, n_ticks=30, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True, fused=6, straggler_ticks=8, rotate=True)
    _run(cuda, oracle, G=128, R=5, S=1, W=64, n_ticks=30, drop_p=0.05, timeout_frac=0.3, hb_every=4, preset=False, straggler_ticks=2, rotate=True)
    _run_bench_shape(cuda, oracle, G=1024, frac=0.25, span=10, n_ticks=30, fused=8, straggler_ticks=4, every=3, rotate=True)
    from summerset_amd import MultiPaxosCluster, SummersetError
    with pytest.raises(SummersetError):                                           # it rides on the straggler mark pass
        MultiPaxosCluster(64, 5, 64, outbox_cap=68).set_role_rotation(True)


def run_rest_rides_in_next_r1(cuda, oracle, G, S, W, n_ticks, drop_p):
    """`smr_mp_run_ticks` with the list on and NO timer for more than 2 x 16 + ttl ticks: the rest of a tick's R3 then rides in
    the next tick's R1 launch (`mp_rest_then_local`, round 4).  Uncapped i.i.d. reply loss, so that the tally's closed form
    fails for many lanes and that rest has real work -- rows short of their quorum, commit-bar runs that stop inside the
    batch of rows -- in every tick; leader changes in the first ticks leave groups led by other replicas.  Full state against
    the oracle at every batch end."""
    from summerset_amd import MultiPaxosCluster, stream
    R, H, cap = 5, 4, W + 4
    eng = MultiPaxosCluster(G, R, W, win_reserve=W // 8, outbox_cap=cap, straggler_ticks=4)
    orc = oracle.MpOracle(G, R, W, win_reserve=W // 8, cap=cap, record_commits=False)
    eng.preset_leader(0); orc.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=n_ticks, drop_p=drop_p, timeout_frac=0.3, hb_every=H, timeout_span=5)
    batch, deferred_possible = [], 0
    for t in range(n_ticks):
        inp = st.tick(t)
        orc.tick(**inp)
        fired = bool((inp["timeout_rep"] != 0xFF).any())
        assert fired == (t < 5) or not fired
        if not fired:                                              # a host knows when no timer fired: no array at all
            inp["timeout_rep"] = inp["timeout_src"] = None
        batch.append(_to_dev(inp, cuda))
        if len(batch) == 8 or t == n_ticks - 1:
            eng.run_ticks(batch)
            batch = []
            _compare(eng, orc, R, t)
            deferred_possible += t >= 5 + 2 * 16 + 4
    assert deferred_possible >= 3                                  # batches that ran the fused launch
    for r in range(R):
        assert eng.counters(r)["commits"] == orc.total_commits(r)
    assert (orc.dump(1)["leader"] != 0).any()


def run_next_appends_ride_in_the_tally(cuda, oracle, G, S, W, n_ticks, frac, batch=8, H=4, R=5, win_reserve=None, straggler_ticks=4,
                                       rotate=False, max_drop=2, drop_p=0.1, expect_rejects=False):
    """`smr_mp_run_ticks` with the list on (round 6): the quorum-tally launch of tick t also runs the leader's steady-state
    `handle_req_batch` calls of tick t + 1 for every group whose tick t its closed form completes (`MpNextLocal`,
    `quorum_tally_block`), and that tick's R1 launch skips them (`r1_done`).  Quorum-preserving loss and no commit list, as in
    bench.py, so that the closed form is the common case; leader changes (other replicas lead afterwards, batches addressed to
    a deposed leader are redirected), heartbeat ticks (no fold across the heartbeat round), batch ends (the next batch's
    inputs are not known) and, with a small window, back-pressure (the fold's window test fails and R1 refuses the batch).
    Full state against the oracle at every batch end; the counter says that the path ran.  The engine only does it with
    SMR_MP_FOLD_R1 in the environment (measured a wash, off by default): both ways here."""
    from summerset_amd import MultiPaxosCluster, stream
    cap = W + 4
    wr = W // 8 if win_reserve is None else win_reserve
    folded = {}
    for no_fold in (False, True):
        if not no_fold:
            os.environ["SMR_MP_FOLD_R1"] = "1"
        try:
            eng = MultiPaxosCluster(G, R, W, win_reserve=wr, outbox_cap=cap, straggler_ticks=straggler_ticks)
        finally:
            os.environ.pop("SMR_MP_FOLD_R1", None)
        if rotate:
            eng.set_role_rotation(True)
        orc = oracle.MpOracle(G, R, W, win_reserve=wr, cap=cap, record_commits=False)
        eng.preset_leader(0); orc.preset_leader(0)
        st = stream.MultiPaxosStream(G, R, S, cap=cap, n_ticks=n_ticks, drop_p=drop_p, timeout_frac=frac, hb_every=H,
                                     rand_rows=S + 4, max_drop=max_drop, timeout_span=max(n_ticks // 2, 1))
        pending = []
        for t in range(n_ticks):
            inp = st.tick(t)
            orc.tick(**inp)
            if not (inp["timeout_rep"] != 0xFF).any():
                inp = dict(inp, timeout_rep=None, timeout_src=None)
            pending.append(_to_dev(inp, cuda))
            if len(pending) == batch or t == n_ticks - 1:
                eng.run_ticks(pending)
                pending = []
                _compare(eng, orc, R, t)
        for r in range(R):
            assert eng.counters(r)["commits"] == orc.total_commits(r)
        folded[no_fold] = sum(eng.debug_folded_batches(r) for r in range(R))
        if expect_rejects:
            assert sum(eng.counters(r)["rejects"] for r in range(R)) > 0
    assert folded[False] > 0 and folded[True] == 0, folded
    return folded[False]


def test_next_ticks_appends_ride_in_the_tally(cuda, oracle):
    n = run_next_appends_ride_in_the_tally(cuda, oracle, G=1024, S=32, W=512, n_ticks=40, frac=0.25)
    assert n > 1024 * 32 * 10                                       # most groups, 6 of 8 ticks
    run_next_appends_ride_in_the_tally(cuda, oracle, G=300, S=40, W=512, n_ticks=24, frac=0.1)        # more batches than the tally prefetches
    run_next_appends_ride_in_the_tally(cuda, oracle, G=200, S=6, W=32, n_ticks=48, frac=0.1, H=12, win_reserve=2, expect_rejects=True)
    run_next_appends_ride_in_the_tally(cuda, oracle, G=257, S=4, W=64, n_ticks=40, frac=0.5, R=3, max_drop=1, batch=5, H=3)
    run_next_appends_ride_in_the_tally(cuda, oracle, G=130, S=3, W=64, n_ticks=36, frac=0.3, rotate=True, batch=16)
    run_next_appends_ride_in_the_tally(cuda, oracle, G=130, S=2, W=64, n_ticks=36, frac=0.2, R=7, max_drop=3)
    run_next_appends_ride_in_the_tally(cuda, oracle, G=600, S=3, W=64, n_ticks=48, frac=0.2, drop_p=0.3, max_drop=None)   # uncapped loss: rows short of the quorum


def test_batched_ticks_rest_rides_in_next_r1(cuda, oracle):
    run_rest_rides_in_next_r1(cuda, oracle, G=600, S=3, W=64, n_ticks=72, drop_p=0.3)
    run_rest_rides_in_next_r1(cuda, oracle, G=256, S=8, W=128, n_ticks=64, drop_p=0.15)
