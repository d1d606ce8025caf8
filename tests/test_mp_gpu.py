"""MultiPaxos lock-step parity: HIP engine (through the C-ABI) vs the CPU oracle
on identical seeded streams, compared after EVERY tick on the full canonical
state of all replicas (bit-exact)."""
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


def _run(cuda, oracle, G, R, S, W, n_ticks, drop_p, timeout_frac, hb_every, preset, commit_extra=0, seed=None,
         every=1, timeout_rep=1):
    from summerset_amd import MultiPaxosCluster, stream
    cap = W + 4
    eng = MultiPaxosCluster(G, R, W, outbox_cap=cap, commit_extra=commit_extra, commit_list_cap=G * S * 4 + 64)
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
    for t in range(n_ticks):
        inp = st.tick(t)
        orc.tick(**inp)
        eng.tick(**_to_dev(inp, cuda))
        if t % every == 0 or t == n_ticks - 1:
            _compare(eng, orc, R, t)
        # ordered committed-slot list of the (possibly several) leaders
        for r in range(R):
            og, os_ = orc.take_commits(r)
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


def test_leader_change(cuda, oracle):
    # every group sees a HearTimeout on replica 1 at some tick
    _run(cuda, oracle, G=512, R=5, S=2, W=64, n_ticks=48, drop_p=0.1, timeout_frac=1.0, hb_every=4, preset=True)


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


def test_config2_4096_groups(cuda, oracle):
    # BASELINE.json configs[1]: 4096 groups x 5 replicas, 10 % ack loss, 1 % leader timeouts
    _run(cuda, oracle, G=4096, R=5, S=1, W=64, n_ticks=128, drop_p=0.1, timeout_frac=0.01, hb_every=4,
         preset=True, every=8)
