"""Batched Heartbeater kernels (summerset_amd/csrc/heartbeater.hip, through the C-ABI) against the oracle: the hand-derived
traces of tests/test_oracle_hb.py on the engine, and a seeded random stream of every call over thousands of groups, full
state and every returned event compared after each call -- bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALL, NONE = 0xFE, 0xFF


class _Np:
    """the engine behind the oracle's numpy interface"""

    def __init__(self, eng, dev):
        import torch
        self.e, self.dev, self.torch = eng, dev, torch

    def _t(self, a, dt=np.uint8):
        a = np.ascontiguousarray(a, dt)
        return self.torch.from_numpy(a.view(np.int32) if dt == np.uint32 else a).to(self.dev)

    def set_sending(self, x): self.e.set_sending(self._t(x))
    def kickoff_hear_timer(self, p, now, draw): self.e.kickoff_hear_timer(self._t(p), now, self._t(draw, np.uint32))
    def poll(self, now):
        t, s = self.e.poll(now)
        return t.cpu().numpy(), s.cpu().numpy()
    def clear_reply_cnts(self, p): self.e.clear_reply_cnts(self._t(p))
    def update_bcast_cnts(self, f): return self.e.update_bcast_cnts(self._t(f)).cpu().numpy()
    def update_heard_cnt(self, p): self.e.update_heard_cnt(self._t(p))
    def dump(self): return self.e.dump()


def test_traces_on_the_engine(cuda, oracle):
    import test_oracle_hb as tr
    from summerset_amd import Heartbeater, SummersetError

    class Fake:                                                 # the traces construct `oracle.HbOracle(...)`
        @staticmethod
        def HbOracle(G, R=5, me=0, a=1200, b=2000, c=20, now_ms=0):
            try:
                return _Np(Heartbeater(G, R, me, a, b, c, now_ms), cuda)
            except SummersetError as e:
                raise ValueError(str(e))
    for name in ("test_configuration_checks", "test_hear_timers_by_hand", "test_send_ticker_skips_missed_ticks",
                 "test_reply_counters_death_and_revival", "test_reference_timer_timeout", "test_reference_timer_restart",
                 "test_reference_timer_set_backwards"):
        getattr(tr, name)(Fake)


@pytest.mark.parametrize("G,R,me", [(3000, 5, 0), (700, 3, 2), (257, 8, 5)])
def test_random_calls_match_oracle(cuda, oracle, G, R, me):
    from summerset_amd import Heartbeater
    cfg = (150, 400, 30)
    eng, orc = _Np(Heartbeater(G, R, me, *cfg, now_ms=7), cuda), oracle.HbOracle(G, R, me, *cfg, now_ms=7)
    rng = np.random.default_rng(G + R)
    now, events = 7, 0

    def sel():
        x = rng.integers(0, R, G).astype(np.uint8)
        u = rng.random(G)
        x[u < 0.2] = ALL
        x[u > 0.7] = NONE
        return x
    for step in range(120):
        now += int(rng.integers(0, 90))
        k = step % 6
        if k == 0:
            p, d = sel(), rng.integers(0, 2**32, (R, G), dtype=np.uint32)
            eng.kickoff_hear_timer(p, now, d); orc.kickoff_hear_timer(p, now, d)
        elif k == 1:
            s = rng.choice(np.array([0, 1, NONE], np.uint8), G)
            eng.set_sending(s); orc.set_sending(s)
        elif k == 2:
            p = sel()
            p[p == ALL] = NONE
            eng.update_heard_cnt(p); orc.update_heard_cnt(p)
        elif k == 3:
            f = (rng.random(G) < 0.8).astype(np.uint8)
            a, b = eng.update_bcast_cnts(f), orc.update_bcast_cnts(f)
            assert np.array_equal(a, b), step
        elif k == 4 and step % 18 == 4:
            p = sel()
            eng.clear_reply_cnts(p); orc.clear_reply_cnts(p)
        (ta, sa), (tb, sb) = eng.poll(now), orc.poll(now)
        assert np.array_equal(ta, tb) and np.array_equal(sa, sb), step
        events += int(tb.sum()) + int(sb.sum())
        a, b = eng.dump(), orc.dump()
        for name in b:
            assert np.array_equal(a[name], b[name]), (step, name)
    assert events > G and (orc.dump()["alive"] != (1 << R) - 1).any()


def test_timeouts_feed_the_multipaxos_engine(cuda, oracle):
    """replica 1's Heartbeater decides WHEN, the MultiPaxos engine does the leader change -- the event arrays never leave the
    device; the oracle gets the same arrays and the clusters agree afterwards"""
    import torch
    from summerset_amd import Heartbeater, MultiPaxosCluster, stream
    from summerset_amd.heartbeater import hear_timeouts_for_engine
    G, R, W = 192, 5, 32
    cap = W + 4
    hb = Heartbeater(G, R, 1, 150, 400, 30, now_ms=0)
    draw = torch.from_numpy(np.random.default_rng(3).integers(0, 2**31, (R, G)).astype(np.int32)).to(cuda)
    hb.kickoff_hear_timer(torch.full((G,), 0, dtype=torch.uint8, device=cuda), 0, draw)       # replica 1 listens for leader 0
    eng, orc = MultiPaxosCluster(G, R, W, outbox_cap=cap), oracle.MpOracle(G, R, W, cap=cap)
    eng.preset_leader(0); orc.preset_leader(0)
    st = stream.MultiPaxosStream(G, R, 1, cap=cap, n_ticks=12, drop_p=0.0, timeout_frac=0.0, hb_every=3)
    fired = 0
    for t in range(12):
        now = 40 * (t + 1)                                      # 40 ms per tick: the 150-400 ms timers fire from tick 3 on
        tm, _ = hb.poll(now)
        rep, src = hear_timeouts_for_engine(tm, 1)
        fired += int((rep != 0xFF).sum())
        inp = st.tick(t)
        inp["timeout_rep"], inp["timeout_src"] = rep.cpu().numpy(), src.cpu().numpy()
        inp["req_target"] = np.where(orc.dump(1)["leader"] == 1, 1, 0).astype(np.uint8)   # clients follow the leader they see
        orc.tick(**inp)
        dev = {k: (torch.from_numpy(v).to(cuda) if isinstance(v, np.ndarray) else v) for k, v in inp.items()}
        dev["timeout_rep"], dev["timeout_src"] = rep, src       # the engine takes the Heartbeater's device arrays as they are
        eng.tick(**dev)
    assert fired == G                                           # every group's timer fired exactly once (never re-armed)
    for r in range(R):
        a, b = eng.dump(r), orc.dump(r)
        for k in b:
            assert np.array_equal(a[k], b[k]), (r, k)
    assert (orc.dump(1)["leader"] == 1).all()
