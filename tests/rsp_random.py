"""Seeded random handler calls for ONE RSPaxos replica object, built from the oracle's current state so that they
land in every branch (stale / equal / higher ballots, holes, slots outside the ring, replies to instances in every
status, a leader that is sent Accepts and Prepares, reconstruction rows for unknown slots).  Not a legal protocol
run -- a differential test: both implementations must do the same thing with any input."""
import numpy as np

NULL, NO_REP = 0xFFFFFFFF, 0xFF


def _pick(rng, G, *choices):
    """per group one of the arrays / scalars in `choices`, uniformly"""
    k = rng.integers(0, len(choices), G)
    out = np.zeros(G, np.int64)
    for i, c in enumerate(choices):
        out = np.where(k == i, np.asarray(c, np.int64), out)
    return out


def calls(rng, d, G, R, me, W):
    """yield (method, kwargs) for one step; d = the oracle's dump before the step"""
    ln, cb, bms, bps, bpd = (d[k].astype(np.int64) for k in ("len", "commit_bar", "bal_max_seen", "bal_prep_sent", "bal_prepared"))
    g = np.arange(G)
    peer = rng.integers(0, R, G).astype(np.uint8)
    peer[peer == me] = (me + 1) % R
    u32 = lambda a: np.ascontiguousarray(np.clip(a, 0, 2 ** 31 - 1).astype(np.uint32))
    u64 = lambda a: np.ascontiguousarray(np.maximum(a, 0).astype(np.uint64))
    fl = lambda p: (rng.random(G) < p).astype(np.uint8)
    higher = (((bms >> 8) + 1) << 8) | (peer.astype(np.int64) + 1)
    higher = np.where(rng.random(G) < 0.15, higher, bms)           # a higher ballot (a new leader) only now and then
    kind = rng.choice(9, p=[0.2, 0.1, 0.25, 0.05, 0.05, 0.1, 0.05, 0.1, 0.1])
    if kind == 0:
        v = rng.integers(1, 1 << 20, G).astype(np.uint32)
        v[rng.random(G) < 0.3] = NULL
        yield "req_batch", dict(val=v)
    elif kind == 1:
        slot = _pick(rng, G, ln, ln + rng.integers(1, 4, G), np.maximum(ln - 1, 0), np.maximum(ln - W - 1, 0), rng.integers(0, 3 * W, G))
        ballot = _pick(rng, G, bms, bms, higher, np.maximum(bms - 256, 0), 0)
        yield "accept", dict(flags=fl(0.9), peer=peer, slot=u32(slot), ballot=u64(ballot), val=rng.integers(0, 1 << 20, G).astype(np.uint32),
                             mask=_pick(rng, G, 1 << me, 1 << me, rng.integers(0, 32, G)).astype(np.uint8))
    elif kind == 2:
        slot = _pick(rng, G, cb, cb, cb, np.maximum(ln - 1, 0), np.maximum(ln - 1, 0), rng.integers(0, 2 * W, G))
        ballot = np.stack([_pick(rng, G, bpd, bpd, bpd, bpd, bpd, bps, 7) for _ in range(R)])
        flags = (rng.random((R, G)) < 0.8).astype(np.uint8)
        flags[me] = 0
        yield "accept_replies", dict(slot=u32(slot), ballot=u64(ballot), flags=np.ascontiguousarray(flags))
    elif kind == 3:
        src = _pick(rng, G, NO_REP, NO_REP, d["leader"], peer).astype(np.uint8)
        yield "become_leader", dict(src=src)
    elif kind == 4:
        trig = _pick(rng, G, cb, ln, ln + 2, np.maximum(ln - W, 0), rng.integers(0, 2 * W, G))
        ballot = _pick(rng, G, higher, higher, bms, np.maximum(bms - 256, 0))
        yield "prepare", dict(flags=fl(0.9), peer=peer, trig=u32(trig), ballot=u64(ballot))
    elif kind == 5:
        # a batch shaped like the answer to my own Prepare (first Preparing slot's bookkeeping), sometimes off
        st, lt, le = d["s_status"], d["s_ltrig"].astype(np.int64), d["s_lendp"].astype(np.int64)
        trig = np.zeros(G, np.int64); endp = np.zeros(G, np.int64)
        for w in range(W):
            hit = (st[w] == 1) & (d["s_flags"][w] & 1 == 1) & (trig == 0)
            trig = np.where(hit, lt[w], trig); endp = np.where(hit, le[w], endp)
        trig = _pick(rng, G, trig, trig, trig, rng.integers(0, 2 * W, G))
        endp = np.maximum(endp, trig)
        n = np.minimum(endp - trig + 1 + _pick(rng, G, 0, 0, 0, 2), W)
        n = np.where(rng.random(G) < 0.15, rng.integers(0, W, G), n)
        vbal = np.where(rng.random((W, G)) < 0.5, _pick(rng, G, (1 << 8) | 1, (1 << 8) | 1, (2 << 8) | 2)[None, :], 0)
        vval = np.where(vbal > 0, rng.integers(0, 50, (W, G)), NULL)
        vmask = np.where(vbal > 0, 1 << peer.astype(np.int64)[None, :], 0)
        yield "prepare_replies", dict(peer=peer, pr_n=u32(n), pr_trig=u32(trig), pr_endp=u32(endp), pr_ballot=u64(_pick(rng, G, bps, bps, bps, bms, 5)),
                                      pr_vbal=u64(vbal), pr_vval=np.ascontiguousarray(vval.astype(np.uint32)), pr_vmask=np.ascontiguousarray(vmask.astype(np.uint8)))
    elif kind == 6:
        n = rng.integers(0, 5, G)
        slots = np.stack([_pick(rng, G, cb, np.maximum(ln - 1, 0), ln + 1, rng.integers(0, 2 * W, G)) for _ in range(W)])
        yield "reconstruct", dict(flags=fl(0.9), rc_n=u32(n), rc_slot=u32(slots))
    elif kind == 7:
        n = rng.integers(0, 4, G)
        slots = np.stack([_pick(rng, G, cb, cb + 1, np.maximum(cb - 1, 0), rng.integers(0, 2 * W, G)) for _ in range(W)])
        bal = np.stack([_pick(rng, G, bms, bms, 0, bms + 256) for _ in range(W)])
        yield "reconstruct_reply", dict(flags=fl(0.9), rr_n=u32(n), rr_slot=u32(slots), rr_bal=u64(bal),
                                        rr_val=rng.integers(0, 50, (W, G)).astype(np.uint32), rr_mask=rng.integers(0, 32, (W, G)).astype(np.uint8))
    else:
        ballot = _pick(rng, G, bms, bms, higher, np.maximum(bms - 256, 0))
        commit = _pick(rng, G, cb + 1, cb + 2, ln, ln + 2, np.maximum(cb - 1, 0))
        yield "heartbeat", dict(flags=fl(0.9), peer=peer, ballot=u64(ballot), commit_bar=u32(commit),
                                exec_bar=u32(_pick(rng, G, d["exec_bar"], d["exec_bar"], 0, d["exec_bar"].astype(np.int64) + 1)),
                                snap_bar=u32(_pick(rng, G, 0, d["snap_bar"], d["exec_bar"])))
        yield "bcast_heartbeat", dict(flags=(d["leader"] == me).astype(np.uint8))
