"""Pins the CPU MultiPaxos oracle with hand-derived traces and the invariants the
reference states (multipaxos/mod.rs:465-468; TLA+ "one value per slot").  CPU only."""
import os

import numpy as np

from summerset_amd import stream

GOLD = os.path.join(os.path.dirname(__file__), "golden", "mp_golden.npz")
ID = stream.CTL_IDENTITY
NONE = 0xFF


def _ctl(order, drop=0, R=5):
    w = 0
    ids = list(order) + [i for i in range(8) if i not in order]
    for i, r in enumerate(ids):
        w |= r << (3 * i)
    return w | (drop << 24)


def test_hand_trace_single_slot(oracle):
    """One group, preset leader 0, one batch: follow every field by hand."""
    m = oracle.MpOracle(1, 5, 16)
    m.preset_leader(0)
    cap = m.cap
    ctl = np.full((cap, 1), _ctl([3, 1, 4, 2, 0]), np.uint32)          # replies arrive 3,1,4,2
    m.tick(req_target=np.zeros(1, np.uint8), req_cnt=np.ones(1, np.uint32),
           req_val=np.full((1, 1), 77, np.uint32), ackctl=ctl)
    L = m.dump(0)
    assert L["bal_prepared"][0] == 0x101 and L["leader"][0] == 0
    assert L["log_len"][0] == 1 and L["accept_bar"][0] == 1
    # self ack, then peers 3 and 1 reach quorum 3; later acks are ignored (A.8 #1: mask freezes)
    assert L["s_acks"][0, 0] == 0b01011
    assert L["s_status"][0, 0] == 4 and L["commit_bar"][0] == 1 and L["exec_bar"][0] == 1
    assert L["s_bal"][0, 0] == 0x101 and L["s_reqs"][0, 0] == 77
    assert L["s_vbal"][0, 0] == 0x101 and L["s_vreqs"][0, 0] == 77 and L["s_flags"][0, 0] == 0b101
    F = m.dump(2)
    assert F["s_status"][0, 0] == 2 and F["accept_bar"][0] == 1 and F["commit_bar"][0] == 0   # not yet told
    assert F["s_flags"][0, 0] == 0b010 and F["s_src"][0, 0] == 0
    assert m.total_commits(0) == 1
    g, s = m.take_commits(0)
    assert g.tolist() == [0] and s.tolist() == [0]
    # heartbeat: followers learn the commit (advance_commit_bar), execute, everything trims
    m.tick(ackctl=ctl, heartbeat=True)
    F = m.dump(2)
    assert F["commit_bar"][0] == 1 and F["exec_bar"][0] == 1 and F["s_status"][0, 0] == 4 and F["start_slot"][0] == 0
    m.tick(ackctl=ctl, heartbeat=True)
    assert m.dump(2)["start_slot"][0] == 1 and m.dump(0)["start_slot"][0] == 1


def test_hand_trace_lost_quorum(oracle):
    """Three of four peer acks lost: slot stays Accepting with 2 bits, commit_bar stalls, later slot commits."""
    m = oracle.MpOracle(1, 5, 16)
    m.preset_leader(0)
    ctl = np.full((m.cap, 1), ID, np.uint32)
    ctl[0, 0] = _ctl([0, 1, 2, 3, 4], drop=0b11100)                       # only peer 1 gets through for entry 0
    m.tick(req_target=np.zeros(1, np.uint8), req_cnt=np.full(1, 2, np.uint32),
           req_val=np.array([[5], [6]], np.uint32), ackctl=ctl)
    L = m.dump(0)
    assert L["s_status"][0, 0] == 2 and L["s_acks"][0, 0] == 0b00011
    assert L["s_status"][1, 0] == 3 and L["s_acks"][1, 0] == 0b00111       # committed but not executed
    assert L["commit_bar"][0] == 0 and L["exec_bar"][0] == 0 and m.total_commits(0) == 1


def test_bootstrap_noop_pins_exec_bar(oracle):
    """Leader None + HearTimeout on an empty log: slot 0 is a no-op that goes straight to Executed at
    commit time (durability.rs:171-172) WITHOUT moving exec_bar (execution.rs:70 only fires for a
    finished command) -- reproduced as-is."""
    m = oracle.MpOracle(1, 5, 16)
    ctl = np.full((m.cap, 1), ID, np.uint32)
    m.tick(timeout_rep=np.zeros(1, np.uint8), timeout_src=np.full(1, NONE, np.uint8), ackctl=ctl)
    L = m.dump(0)
    assert L["bal_prepared"][0] == 0x101 and L["bal_prep_sent"][0] == 0x101       # make_greater_ballot(0) on id 0
    assert L["s_status"][0, 0] == 2 and L["s_packs"][0, 0] == 0b00111              # quorum of PrepareReplies
    assert L["s_reqs"][0, 0] == 0 and L["s_ltrig"][0, 0] == 0 and L["s_lendp"][0, 0] == 0
    for r in range(1, 5):
        assert m.dump(r)["leader"][0] == 0 and m.dump(r)["bal_max_seen"][0] == 0x101
    m.tick(req_target=np.zeros(1, np.uint8), req_cnt=np.ones(1, np.uint32), req_val=np.full((1, 1), 9, np.uint32),
           ackctl=ctl)
    L = m.dump(0)
    assert L["s_status"][0, 0] == 4 and L["s_status"][1, 0] == 4 and L["commit_bar"][0] == 2
    assert L["exec_bar"][0] == 0                                                   # pinned by the no-op


def test_leader_change_adopts_highest_voted(oracle):
    m = oracle.MpOracle(1, 5, 16)
    m.preset_leader(0)
    ctl = np.full((m.cap, 1), _ctl([0, 1, 2, 3, 4], drop=0b11110), np.uint32)      # all acks lost: nothing commits
    m.tick(req_target=np.zeros(1, np.uint8), req_cnt=np.ones(1, np.uint32), req_val=np.full((1, 1), 42, np.uint32),
           ackctl=ctl)
    ok = np.full((m.cap, 1), ID, np.uint32)
    m.tick(timeout_rep=np.ones(1, np.uint8), timeout_src=np.zeros(1, np.uint8), ackctl=ok)
    N = m.dump(1)
    assert N["leader"][0] == 1 and N["bal_prepared"][0] == 0x202 and N["bal_prep_sent"][0] == 0x202
    assert N["s_status"][0, 0] == 2 and N["s_reqs"][0, 0] == 42 and N["s_pmax"][0, 0] == 0x101
    assert m.dump(0)["leader"][0] == 1                                             # old leader stepped down
    m.tick(ackctl=ok)                                                              # re-Accept round
    N = m.dump(1)
    assert N["s_status"][0, 0] == 4 and N["s_bal"][0, 0] == 0x202 and N["commit_bar"][0] == 1
    assert m.total_commits(1) == 1 and m.total_commits(0) == 0


def test_invariants_random_streams(oracle):
    G, R, S, W, T = 64, 5, 3, 64, 60
    m = oracle.MpOracle(G, R, W)
    m.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=m.cap, n_ticks=T, drop_p=0.2, timeout_frac=0.6, hb_every=3, seed=11)
    chosen = {}
    for t in range(T):
        m.tick(**st.tick(t))
        dumps = [m.dump(r) for r in range(R)]
        for d in dumps:
            live = d["overflow"] == 0
            # exec_bar <= commit_bar <= accept_bar <= start_slot + len (mod.rs:465-468)
            assert (d["exec_bar"] <= d["commit_bar"])[live].all()
            assert (d["commit_bar"] <= d["accept_bar"])[live].all()
            assert (d["accept_bar"] <= d["log_len"])[live].all()
            assert (d["start_slot"] <= d["exec_bar"])[live].all()
        # at most one value is ever committed per (group, slot), across replicas and time
        for d in dumps:
            w, g = np.nonzero(d["s_status"] >= 3)
            for wi, gi in zip(w, g):
                lo, hi = int(d["start_slot"][gi]), int(d["log_len"][gi])
                slot = next(s for s in range(lo, hi) if s % W == wi)
                v = int(d["s_reqs"][wi, gi])
                assert chosen.setdefault((gi, slot), v) == v
    assert len(chosen) > G * 10


def test_golden_final_state(oracle):
    g = np.load(GOLD)
    G, R, S, W, T = [int(x) for x in g["params"]]
    m = oracle.MpOracle(G, R, W, cap=W + 4)
    m.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=W + 4, n_ticks=T, drop_p=0.1, timeout_frac=0.5, hb_every=4)
    for t in range(T):
        m.tick(**st.tick(t))
    for r in range(R):
        d = m.dump(r)
        for k, v in d.items():
            assert np.array_equal(v, g["r%d_%s" % (r, k)]), (r, k)
        assert m.total_commits(r) == int(g["r%d_commits" % r][0])


def test_safety_properties_under_leader_changes_and_loss(oracle):
    """The protocol's own invariants on the restatement (nothing here compares with the engine): a slot
    that any replica holds Committed / Executed has ONE value across all replicas and over time, bars are
    ordered, and a leader only commits what a quorum accepted at its ballot."""
    from summerset_amd import stream
    G, R, S, W, T = 200, 5, 3, 64, 60
    o = oracle.MpOracle(G, R, W, cap=W + 4, record_commits=False)
    o.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, S, cap=W + 4, n_ticks=T, drop_p=0.15, timeout_frac=0.9, hb_every=3, timeout_rep=1)
    st2 = stream.MultiPaxosStream(G, R, S, cap=W + 4, n_ticks=T, drop_p=0.15, timeout_frac=0.5, hb_every=3, timeout_rep=3,
                                  seed=77)
    chosen = {}                                            # (g, slot) -> value, once committed anywhere
    for t in range(T):
        inp = st.tick(t)
        if t >= T // 2:                                    # a second wave of timeouts, on another replica
            ev = st2.tick_events(t - T // 2)
            hit = ev["timeout_rep"] != 0xFF
            inp["timeout_rep"] = np.where(hit, ev["timeout_rep"], inp["timeout_rep"]).astype(np.uint8)
            inp["timeout_src"] = np.where(hit, inp["req_target"], inp["timeout_src"]).astype(np.uint8)
        o.tick(**inp)
        d = [o.dump(r) for r in range(R)]
        live = d[0]["overflow"] == 0
        for r in range(R):
            x = d[r]
            assert (x["exec_bar"][live] <= x["commit_bar"][live]).all()
            assert (x["commit_bar"][live] <= x["log_len"][live]).all() and (x["start_slot"][live] <= x["exec_bar"][live]).all()
            st_, val = x["s_status"], x["s_reqs"]
            for g in np.nonzero(live)[0]:
                lo, hi = int(x["start_slot"][g]), int(x["log_len"][g])
                for s in range(lo, hi):
                    if st_[s % W, g] >= 3:                 # Committed or Executed
                        v = int(val[s % W, g])
                        assert chosen.setdefault((int(g), s), v) == v, ("two values chosen", g, s, r, t)
    assert len(chosen) > G * 20
    # (no liveness claim: with uncapped loss a slot that misses its quorum stalls its group until the next
    #  leader change -- the reference never retransmits an Accept -- but most groups do move on)
    d0 = o.dump(0)
    assert (np.stack([o.dump(r)["commit_bar"] for r in range(R)]).max(axis=0)[d0["overflow"] == 0] > 20).mean() > 0.5
