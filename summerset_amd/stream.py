"""Seeded synthetic client-op / fault streams.

Host-side generator of the per-tick inputs both the HIP engine and the CPU
oracle consume, so parity runs are driven by IDENTICAL streams.  Pure numpy,
keyed SplitMix64: every value is a function of (seed, tick, entry, group, ...)
only, so any tick can be regenerated independently.  Default seed 0x5EED5EED.
"""
import numpy as np

DEFAULT_SEED = 0x5EED5EED
CTL_IDENTITY = 0x00FAC688
NO_REPLICA = 0xFF
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """SplitMix64 finaliser on a uint64 array (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _key(seed, *parts):
    """hash a tuple of broadcastable integer arrays into uint64"""
    h = splitmix64(np.uint64(seed))
    for p in parts:
        with np.errstate(over="ignore"):
            h = splitmix64(h ^ (np.asarray(p).astype(np.uint64) * np.uint64(0xD1B54A32D192ED03)))
    return h


def random_ackctl(seed, tick, n_entries, G, R, drop_p, cap=None, max_drop=None, group_base=0):
    """ackctl[cap][G]: per outbox entry a random peer order + loss mask.

    Rows >= n_entries (never reached by a tick that emits <= n_entries
    messages) hold the identity order with no loss.  `max_drop` caps how many
    replica ids may be marked lost per entry: the reference never retransmits
    an Accept (TCP is reliable), so an entry that loses its quorum stalls its
    group's commit_bar until the next leader change; with max_drop = R - thresh
    every entry stays committable whoever leads.
    """
    cap = n_entries if cap is None else cap
    out = np.full((cap, G), CTL_IDENTITY, np.uint32)
    if n_entries == 0:
        return out
    j = np.arange(n_entries, dtype=np.uint64)[:, None, None]
    g = (np.arange(G, dtype=np.uint64) + np.uint64(group_base))[None, :, None]   # keyed by the GLOBAL group id
    r = np.arange(R, dtype=np.uint64)[None, None, :]
    keys = _key(seed, 0xA11C, tick, j, g, r)
    order = np.argsort(keys, axis=2, kind="stable").astype(np.uint32)        # [n, G, R] replica ids
    word = np.zeros((n_entries, G), np.uint32)
    for i in range(R):
        word |= order[:, :, i] << np.uint32(3 * i)
    for i in range(R, 8):
        word |= np.uint32(i) << np.uint32(3 * i)                              # ids >= R are skipped
    if drop_p > 0:
        u = (_key(seed, 0xD209, tick, j, g, r) >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        bits = (u < drop_p).astype(np.uint32)
        if max_drop is not None:
            bits = np.where(np.cumsum(bits, axis=2) > max_drop, 0, bits).astype(np.uint32)
        mask = np.zeros((n_entries, G), np.uint32)
        for i in range(R):
            mask |= bits[:, :, i] << np.uint32(i)
        word |= mask << np.uint32(24)
    out[:n_entries] = word
    return out


class MultiPaxosStream:
    """S new batches per group per tick to the believed
    leader; every reply row gets a seeded peer order, each peer ack is lost
    with probability drop_p; a fraction of the groups sees one HearTimeout on
    `timeout_rep` (about peer 0) at a seeded tick, after which the clients
    follow the new leader."""

    def __init__(self, G, R=5, S=1, cap=None, n_ticks=1024, seed=DEFAULT_SEED, drop_p=0.1, timeout_frac=0.01,
                 timeout_rep=1, hb_every=4, rand_rows=None, max_drop=None, timeout_span=None, group_base=0):
        self.G, self.R, self.S, self.n_ticks, self.seed = G, R, S, n_ticks, seed
        self.cap = cap if cap is not None else S + 8
        self.drop_p, self.hb_every, self.timeout_rep, self.max_drop = drop_p, hb_every, timeout_rep, max_drop
        self.rand_rows = min(self.cap, rand_rows if rand_rows is not None else self.cap)
        # group_base: this stream covers groups [group_base, group_base + G) of a larger job -- every
        # value is keyed by the global group id, so a sharded job sees exactly the unsharded stream
        self.group_base = int(group_base)
        g = np.arange(G, dtype=np.uint64) + np.uint64(group_base)
        h = _key(seed, 0x7130, g)
        u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
        # the seeded tick of a group's timeout, drawn from [0, timeout_span) (default: the whole run)
        span = n_ticks if timeout_span is None else timeout_span
        t_at = (_key(seed, 0x7131, g) % np.uint64(max(span, 1))).astype(np.int64)
        self.timeout_tick = np.where(u < timeout_frac, t_at, -1)

    def heartbeat(self, t):
        return (t % self.hb_every) == self.hb_every - 1

    def tick_events(self, t):
        """the tick's small per-group arrays: HearTimeout events and where the clients send"""
        hit = self.timeout_tick == t
        timeout_rep = np.where(hit, self.timeout_rep, NO_REPLICA).astype(np.uint8)
        timeout_src = np.where(hit, 0, NO_REPLICA).astype(np.uint8)
        moved = (self.timeout_tick >= 0) & (t > self.timeout_tick)
        req_target = np.where(moved, self.timeout_rep, 0).astype(np.uint8)
        return dict(timeout_rep=timeout_rep, timeout_src=timeout_src, req_target=req_target)

    def tick(self, t):
        G, S = self.G, self.S
        out = self.tick_events(t)
        k = np.arange(S, dtype=np.uint64)[:, None]
        g = (np.arange(G, dtype=np.uint64) + np.uint64(self.group_base))[None, :]
        # opaque non-zero batch tokens (consensus kernels never read payload bytes)
        req_val = ((_key(self.seed, 0x70CE, t, k, g) & np.uint64(0x7FFFFFFF)) | np.uint64(1)).astype(np.uint32)
        out.update(req_cnt=np.full(G, S, np.uint32), req_val=np.ascontiguousarray(req_val),
                   ackctl=random_ackctl(self.seed, t, self.rand_rows, G, self.R, self.drop_p, cap=self.cap,
                                        max_drop=self.max_drop, group_base=self.group_base),
                   heartbeat=self.heartbeat(t))
        return out
