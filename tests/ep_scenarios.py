"""Seeded message generators for the EPaxos handlers, driven by the CPU oracle's current state
so that the messages land in every branch (fast path, slow path, undecided, stale ballots,
padding with null instances, suspected peers)."""
import numpy as np

N, NO_KEY = 0xFFFFFFFF, 0xFF


def _rand_deps(rng, lens, R, G, p_some=0.5):
    """[R, G] DepSet pointing at existing columns (or None)"""
    d = np.full((R, G), N, np.uint32)
    for r in range(R):
        has = (rng.random(G) < p_some) & (lens[r] > 0)
        col = (rng.random(G) * np.maximum(lens[r], 1)).astype(np.uint32)
        d[r] = np.where(has, col, N)
    return d


def propose_round(rng, G, n_keys):
    key = rng.integers(0, n_keys, G).astype(np.uint8)
    key[rng.random(G) < 0.2] = NO_KEY
    exploded = np.where(rng.random(G) < 0.1, rng.integers(0, 32, G), 0).astype(np.uint8)
    return key, exploded


def acceptor_round(rng, dump, G, R, me, n_keys, W):
    """one PreAccept / Accept per group from a random peer, mostly for its next column"""
    lens = dump["len"]
    flags = (rng.random(G) < 0.85).astype(np.uint8)
    peer = rng.integers(0, R, G).astype(np.uint8)
    peer[peer == me] = (me + 1) % R
    g = np.arange(G)
    nxt = lens[peer, g]
    kind = rng.integers(0, 10, G)
    col = nxt.copy()
    col = np.where(kind == 0, nxt + rng.integers(1, 4, G), col)                  # leaves a hole: null padding
    col = np.where((kind == 1) & (nxt > 0), nxt - 1, col).astype(np.uint32)      # an existing column again
    ballot = (peer.astype(np.uint64) + 1)
    ballot = np.where(kind == 2, ballot + (1 << 8), ballot)                      # a higher ballot (explicit prepare)
    ballot = np.where(kind == 1, np.where(rng.random(G) < 0.5, 0, ballot), ballot).astype(np.uint64)   # maybe stale
    seq = rng.integers(1, 6, G).astype(np.uint64)
    deps = _rand_deps(rng, lens, R, G)
    key = rng.integers(0, n_keys, G).astype(np.uint8)
    key[rng.random(G) < 0.05] = NO_KEY
    return dict(flags=flags, peer=peer, col=col, ballot=ballot, seq=seq, deps=np.ascontiguousarray(deps), key=key)


def pre_accept_replies_round(rng, dump, msg, G, R, me, ctl):
    """replies to the instance just proposed (msg = the propose output): mostly agreeing, ~10 % carry
    an extra dependency or a larger seq"""
    col = msg["col"].copy()
    flags = (rng.random((R, G)) < 0.75).astype(np.uint8)
    flags[me] = 0
    ballot = np.full((R, G), me + 1, np.uint64)
    ballot[rng.random((R, G)) < 0.05] = 0                                        # "failure suspected" re-evaluation
    ballot[rng.random((R, G)) < 0.03] = 77                                       # not my ballot
    seq = np.broadcast_to(msg["seq"], (R, G)).copy()
    deps = np.broadcast_to(msg["deps"], (R, R, G)).copy()
    extra = rng.random((R, G)) < 0.1
    seq[extra] += 1
    for p in range(R):
        r = rng.integers(0, R, G)
        c = rng.integers(0, 4, G).astype(np.uint32)
        sel = extra[p]
        cur = deps[p, r, np.arange(G)]
        deps[p, r, np.arange(G)] = np.where(sel, np.where(cur == N, c, cur + 1), cur)
    exploded = np.where(rng.random(G) < 0.2, rng.integers(0, 32, G), 0).astype(np.uint8)
    return dict(col=col, ballot=np.ascontiguousarray(ballot), seq=np.ascontiguousarray(seq.astype(np.uint64)),
                deps=np.ascontiguousarray(deps.astype(np.uint32)), flags=np.ascontiguousarray(flags), order=ctl,
                exploded=exploded)


def accept_replies_round(rng, col, G, R, me, ctl):
    flags = (rng.random((R, G)) < 0.6).astype(np.uint8)
    flags[me] = 0
    ballot = np.full((R, G), me + 1, np.uint64)
    ballot[rng.random((R, G)) < 0.05] = 9
    return dict(col=col, ballot=np.ascontiguousarray(ballot), flags=np.ascontiguousarray(flags), order=ctl)


def commit_round(rng, dump, G, R, me, n_keys, W):
    """one CommitNotice per group from a random peer: mostly for the column at its row's commit bar (the bar
    moves and execution is attempted), sometimes one beyond (a hole), sometimes an older column again;
    dependencies mostly below the other rows' commit bars (the attempt can run), sometimes beyond"""
    lens, bars = dump["len"], dump["commit_bars"]
    flags = (rng.random(G) < 0.9).astype(np.uint8)
    peer = rng.integers(0, R, G).astype(np.uint8)
    peer[peer == me] = (me + 1) % R
    g = np.arange(G)
    bar = bars[peer, g]
    kind = rng.integers(0, 10, G)
    col = bar.copy()
    col = np.where(kind == 0, bar + 1, col)
    col = np.where((kind == 1) & (bar > 0), bar - 1, col).astype(np.uint32)
    ballot = (peer.astype(np.uint64) + 1)
    seq = rng.integers(1, 6, G).astype(np.uint64)
    deps = np.full((R, G), N, np.uint32)
    for r in range(R):
        below = (rng.random(G) * np.maximum(bars[r], 1)).astype(np.uint32)
        beyond = bars[r] + rng.integers(0, 2, G).astype(np.uint32)
        pick = rng.random(G)
        d = np.where(pick < 0.45, N, np.where((pick < 0.92) & (bars[r] > 0), below, beyond))
        deps[r] = d.astype(np.uint32)
    key = rng.integers(0, n_keys, G).astype(np.uint8)
    key[rng.random(G) < 0.05] = NO_KEY
    return dict(flags=flags, peer=peer, col=col, ballot=ballot, seq=seq, deps=np.ascontiguousarray(deps), key=key)
