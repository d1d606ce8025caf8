"""The Heartbeater restatement (oracle/hb_oracle.c) against hand-derived traces of src/server/heartbeat.rs: the
configuration checks of new_and_setup, hear timers (random timeout inside [min, max], one event per explosion, a re-armed
timer's queued event is dropped), the send ticker (first tick at once, then the period grid, late ticks skipped), and
the reply counters with the peer-death / revival rule.  The reference has no tests for this module; its hear timers are
utils::Timer objects, and the reference's own timer tests (src/utils/timer.rs:184-330: timer_timeout, timer_restart,
set_backwards) are restated on them at the end of this file -- that much of the module is pinned."""
import numpy as np
import pytest

ALL, NONE = 0xFE, 0xFF


def _u8(G, v):
    return np.full(G, v, np.uint8)


def test_configuration_checks(oracle):
    H = oracle.HbOracle
    H(1, 5, 0, 100, 200, 1)                                    # the smallest legal timeouts
    with pytest.raises(ValueError):
        H(1, 5, 0, 99, 300, 20)                                # :69-74 min < 100 ms
    with pytest.raises(ValueError):
        H(1, 5, 0, 1200, 1299, 20)                             # :75-81 max < min + 100 ms
    with pytest.raises(ValueError):
        H(1, 5, 0, 1200, 2000, 0)                              # :82 send_interval < 1 ms
    with pytest.raises(ValueError):
        H(1, 5, 0, 1200, 2000, 2001)                           # :83 send_interval > max


def test_hear_timers_by_hand(oracle):
    G, R, me = 3, 5, 1
    h = oracle.HbOracle(G, R, me, 1200, 2000, 20, now_ms=1000)
    d = h.dump()
    assert (d["alive"] == 0b11111).all() and (d["cnt0"][[0, 2, 3, 4]] == 1).all() and (d["deadline"] == 0).all()
    draw = np.zeros((R, G), np.uint32)
    draw[0] = [0, 800, 801]                                    # min + 0, min + 800 = max, min + (801 mod 801) = min
    draw[2] = [5, 5, 5]
    h.kickoff_hear_timer(np.array([ALL, 0, NONE], np.uint8), 1000, draw)
    d = h.dump()
    assert d["deadline"][0].tolist() == [2200, 3000, 0] and d["deadline"][2].tolist() == [2205, 0, 0]
    assert d["deadline"][me].tolist() == [0, 0, 0]             # no timer for myself (:101-103)
    t, s = h.poll(2199)
    assert not t.any() and not s.any()                         # nothing due, ticker off
    t, _ = h.poll(2200)                                        # group 0 armed ALL its timers: draws 0 / 5 / 0 / 0 for peers 0 / 2 / 3 / 4
    assert t[:, 0].tolist() == [1, 0, 0, 1, 1] and not t[:, 1:].any()   # HearTimeout { peer } for 0, 3, 4 at 2200, once each
    assert not h.poll(2201)[0].any() and h.dump()["exploded"][0, 0] == 1
    t, _ = h.poll(2300)
    assert t[:, 0].tolist() == [0, 0, 1, 0, 0] and not t[:, 1:].any()   # peer 2's 2205
    # a timer re-armed before its deadline never fires the old deadline
    h.kickoff_hear_timer(_u8(G, 0), 2500, draw)
    assert h.dump()["deadline"][0].tolist() == [3700, 4500, 3700] and h.dump()["exploded"][0, 0] == 0
    assert not h.poll(3000)[0].any()                           # group 1's old 3000 deadline is gone
    t, _ = h.poll(4600)
    assert t[0].tolist() == [1, 1, 1]
    # kickoff for my own id is a no-op (:194-195)
    h.kickoff_hear_timer(_u8(G, me), 5000, draw)
    assert (h.dump()["deadline"][me] == 0).all()


def test_send_ticker_skips_missed_ticks(oracle):
    h = oracle.HbOracle(2, 5, 0, 1200, 2000, 20, now_ms=100)
    assert not h.poll(100)[1].any()                            # is_sending starts false (:121)
    h.set_sending(np.array([1, NONE], np.uint8))
    _, s = h.poll(100)
    assert s.tolist() == [1, 0]                                # the interval's first tick completes immediately
    assert not h.poll(119)[1].any()
    assert h.poll(120)[1].tolist() == [1, 0]
    assert h.poll(175)[1].tolist() == [1, 0]                   # ticks of 140 and 160 were missed: ONE event, and the
    assert not h.poll(179)[1].any()                            # next one back on the grid at 180 (Skip)
    assert h.poll(180)[1].tolist() == [1, 0]
    h.set_sending(np.array([0, 1], np.uint8))
    assert h.poll(400)[1].tolist() == [0, 1]


def test_reply_counters_death_and_revival(oracle):
    G, R, me = 1, 5, 0
    h = oracle.HbOracle(G, R, me, 1200, 2000, 400)             # repeat_threshold = 1200 / 400 = 3
    one = np.ones(G, np.uint8)
    assert h.update_bcast_cnts(one).tolist() == [0]            # (1, 0, 0) -> .1 = 1, repetition 0
    d = h.dump()
    assert (d["cnt1"][1:] == 1).all() and (d["rep"] == 0).all()
    h.update_heard_cnt(_u8(G, 2))                              # peer 2 keeps answering
    for k in range(3):
        assert h.update_bcast_cnts(one).tolist() == [0]        # repetition 1, 2, 3 for the silent peers: not > 3 yet
        h.update_heard_cnt(_u8(G, 2))
    assert h.dump()["rep"][:, 0].tolist() == [0, 3, 0, 3, 3]
    assert h.update_bcast_cnts(one).tolist() == [1]            # 4 > 3: peers 1, 3, 4 speculated dead, repetition reset
    d = h.dump()
    assert d["alive"].tolist() == [0b00101] and d["rep"][:, 0].tolist() == [0, 0, 0, 0, 0]
    h.update_heard_cnt(_u8(G, 3))                              # heard again: back alive at once (:291-294)
    assert h.dump()["alive"].tolist() == [0b01101]
    # four more broadcasts, nobody answers: peer 2 (last heard before the death round) goes 1, 2, 3, 4 > 3 and dies in
    # the fourth; peer 3 spends the first one catching .1 up, so it is at 3; the dead peers 1 and 4 wrap to 0 silently
    assert [h.update_bcast_cnts(one).tolist() for _ in range(4)] == [[0], [0], [0], [1]]
    assert h.dump()["alive"].tolist() == [0b01001] and h.dump()["rep"][:, 0].tolist() == [0, 0, 0, 3, 0]
    assert h.update_bcast_cnts(one).tolist() == [1]            # ... and peer 3 follows one broadcast later
    assert h.dump()["alive"].tolist() == [0b00001]
    h.clear_reply_cnts(_u8(G, ALL))
    d = h.dump()
    assert (d["cnt0"][1:] == 1).all() and (d["cnt1"] == 0).all() and (d["rep"] == 0).all()
    assert h.update_bcast_cnts(np.zeros(G, np.uint8)).tolist() == [0] and (h.dump()["cnt1"] == 0).all()   # flag clear: no call


# ---- the reference's own timer tests (src/utils/timer.rs:184-330) on the Heartbeater's hear timers ----------------------
# A hear timer IS a utils::Timer (heartbeat.rs:95-110) that kickoff_hear_timer cancels and kicks off with the drawn
# duration (:174-185); poll() at a time delivers HearTimeout iff the timer has exploded by then.  The tests' durations
# are reached with min = 100 ms and the draw = duration - 100 (timeout = min + draw mod (max - min + 1)).
def _timer(oracle):
    h = oracle.HbOracle(1, 3, 0, 100, 2000, 20, now_ms=0)

    def kickoff(now, dur):
        draw = np.zeros((3, 1), np.uint32)
        draw[1, 0] = dur - 100
        h.kickoff_hear_timer(np.array([1], np.uint8), now, draw)

    def fired(now):
        return bool(h.poll(now)[0][1, 0])
    return kickoff, fired


def test_reference_timer_timeout(oracle):                       # timer.rs:184-231 (the kickoff parts; extend: lease tests)
    kickoff, fired = _timer(oracle)
    kickoff(0, 300)
    assert not fired(0) and not fired(299)                      # assert!(!timer.exploded())
    assert fired(300)                                           # finish - start >= 300 ms
    assert not fired(301)                                       # one timeout per explosion
    kickoff(300, 300)                                           # twice
    assert not fired(599) and fired(600)


def test_reference_timer_restart(oracle):                       # timer.rs:233-258
    kickoff, fired = _timer(oracle)
    kickoff(0, 400)
    kickoff(100, 400)                                           # 100 ms later: the deadline moves to 500
    assert not fired(400) and not fired(499)
    assert fired(500)                                           # >= 500 ms and < 800 ms after the start


def test_reference_timer_set_backwards(oracle):                 # timer.rs:308-330
    kickoff, fired = _timer(oracle)
    kickoff(0, 600)
    kickoff(100, 200)                                           # a shorter duration moves the deadline back to 300
    assert not fired(299)
    assert fired(300)                                           # >= 300 ms and < 600 ms
    assert not fired(600)                                       # the long setting is gone
