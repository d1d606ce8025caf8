"""A closed-loop RSPaxos run (tests/rsp_cluster.py) with everything the protocol has: steady appends, lost
Accepts and replies, a HearTimeout that makes replica 1 (then 2) step up in part of the groups -- Prepare
phase with shard merging, re-Accepts, reconstruction reads -- and periodic heartbeats."""
import numpy as np

import rsp_cluster as rc


def run(reps, G, ticks, seed, loss=0.0, changes=True, on_tick=None, view=None):
    """view = (g0, n): the scenario is the one of G groups, but `reps` hold only its groups [g0, g0 + n) and get that
    slice of every input (groups are independent: the slice must behave as it does inside the whole population)"""
    R = len(reps)
    cut = (lambda a: a) if view is None else (lambda a: np.ascontiguousarray(a[view[0]:view[0] + view[1]]))
    for r in reps:
        r.preset_leader(0)
    rng = np.random.default_rng(seed)
    target = np.zeros(G, np.uint8)
    g = np.arange(G)
    log = []
    for t in range(ticks):
        val = (1 + t * G + g).astype(np.uint32)
        val[rng.random(G) < 0.1] = rc.NULL                       # no batch in this group this tick
        to = None
        if changes and t == ticks // 3:                          # replica 1 suspects leader 0 in the even groups
            to = [np.full(G, rc.NO_REP, np.uint8) for _ in range(R)]
            to[1] = np.where(g % 2 == 0, 0, rc.NO_REP).astype(np.uint8)
        if changes and t == ticks // 3 + 1:
            target = np.where(g % 2 == 0, 1, target).astype(np.uint8)
        if changes and t == 2 * ticks // 3:                      # replica 2 suspects whoever it follows in every fourth group
            to = [np.full(G, rc.NO_REP, np.uint8) for _ in range(R)]
            to[2] = np.where(g % 4 == 0, 1, np.where(g % 4 == 1, 0, rc.NO_REP)).astype(np.uint8)
        if changes and t == 2 * ticks // 3 + 1:
            target = np.where(g % 4 <= 1, 2, target).astype(np.uint8)
        drop = None
        if loss:
            kinds = ("accept", "accept_reply", "prepare", "prepare_reply", "recon", "recon_reply", "hb")
            drop = {(k, s, q): rng.random(G) < loss for k in kinds for s in range(R) for q in range(R) if s != q}
        out = rc.tick(reps, cut(val), cut(target), timeouts=None if to is None else [cut(x) for x in to],
                      drop=None if drop is None else {k: cut(v) for k, v in drop.items()}, heartbeat=(t % 3 == 2))
        log.append((t, out))
        if on_tick:
            on_tick(t)
    return log
