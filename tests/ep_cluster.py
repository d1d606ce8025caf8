"""Closed-loop EPaxos cluster out of R per-replica handler objects (backend-agnostic: the CPU
oracle's EpOracle or the HIP EPaxosReplicaGroup behind a numpy adapter).  One tick: every replica
proposes (or not) one instance per group; PreAccepts go to all peers; their replies come back; the
command leader decides fast / slow path; slow-path Accepts and their replies; CommitNotices to all
peers.  Message order: senders ascending, each receiver handles one sender's message at a time."""
import numpy as np

N, NO_KEY = 0xFFFFFFFF, 0xFF


class NumpyEngine:
    """EPaxosReplicaGroup (device tensors) behind the EpOracle-style numpy interface"""

    def __init__(self, eng, cuda):
        import torch
        self.e, self.cuda, self.torch = eng, cuda, torch
        self.W, self.n_keys = eng.W, eng.K
        self._polls = []                                         # with execution on: the submissions of every call so far

    def _after_call(self):
        if getattr(self.e, "execute", False):
            self._polls.append(self.e.exec_poll())               # the engine keeps only the last call's list

    def _t(self, a):
        if a is None:
            return None
        v = a.view(np.int64) if a.dtype == np.uint64 else (a.view(np.int32) if a.dtype == np.uint32 else a)
        return self.torch.from_numpy(np.ascontiguousarray(v)).to(self.cuda)

    @staticmethod
    def _n(d, like):
        return {k: d[k].cpu().numpy().view(v) for k, v in like.items()}

    def propose(self, key, exploded=None):
        o = self.e.handle_req_batch(self._t(key), self._t(exploded))
        self._after_call()
        return self._n(o, dict(flags=np.uint8, col=np.uint32, seq=np.uint64, deps=np.uint32))

    def handle_pre_accept(self, **m):
        o = self.e.handle_msg_pre_accept({k: self._t(v) for k, v in m.items()})
        self._after_call()
        return self._n(o, dict(flags=np.uint8, ballot=np.uint64, seq=np.uint64, deps=np.uint32))

    def handle_accept(self, **m):
        o = self.e.handle_msg_accept({k: self._t(v) for k, v in m.items()})
        self._after_call()
        return self._n(o, dict(flags=np.uint8, ballot=np.uint64))

    def handle_commit_notice(self, **m):
        self.e.handle_msg_commit_notice({k: self._t(v) for k, v in m.items()})
        self._after_call()

    def handle_pre_accept_replies(self, col, ballot, seq, deps, flags, order=None, exploded=None):
        o = self.e.handle_msg_pre_accept_reply(self._t(col), self._t(ballot), self._t(seq), self._t(deps), self._t(flags),
                                               self._t(order), self._t(exploded))
        self._after_call()
        return self._n(o, dict(decision=np.uint8, seq=np.uint64, deps=np.uint32))

    def handle_accept_replies(self, col, ballot, flags, order=None):
        o = self.e.handle_msg_accept_reply(self._t(col), self._t(ballot), self._t(flags), self._t(order))
        self._after_call()
        return self._n(o, dict(committed=np.uint8))

    def dump(self):
        return self.e.dump()

    def exec_dump(self):
        return self.e.exec_dump()

    def take_submissions(self):
        """like EpOracle.take_submissions: everything since the last call, group-major, submission order per group"""
        g = np.concatenate([p[0] for p in self._polls]) if self._polls else np.zeros(0, np.uint32)
        r = np.concatenate([p[1] for p in self._polls]) if self._polls else np.zeros(0, np.uint8)
        c = np.concatenate([p[2] for p in self._polls]) if self._polls else np.zeros(0, np.uint32)
        self._polls = []
        k = np.argsort(g, kind="stable")
        return g[k], r[k], c[k]


def tick(reps, keys, drop=None):
    """reps[r]: backend of replica r; keys[r][G]: proposal of replica r (0xFF none);
    drop[(s, q)] (optional): bool [G] -- the PreAccept from s to q is lost (with its reply).
    Returns per-leader decisions."""
    R = len(reps)
    G = keys.shape[1]
    u8 = lambda v: np.full(G, v, np.uint8)
    pa = [reps[r].propose(np.ascontiguousarray(keys[r])) for r in range(R)]
    # PreAccepts, sender ascending at every receiver
    rep = {}
    for q in range(R):
        for s in range(R):
            if s == q:
                continue
            fl = pa[s]["flags"].copy()
            if drop is not None and (s, q) in drop:
                fl[drop[(s, q)]] = 0
            rep[(q, s)] = reps[q].handle_pre_accept(flags=fl, peer=u8(s), col=pa[s]["col"],
                                                    ballot=np.full(G, s + 1, np.uint64), seq=pa[s]["seq"],
                                                    deps=np.ascontiguousarray(pa[s]["deps"]), key=np.ascontiguousarray(keys[s]))
    out = []
    for s in range(R):
        ballot = np.zeros((R, G), np.uint64); seq = np.zeros((R, G), np.uint64)
        deps = np.full((R, R, G), N, np.uint32); flags = np.zeros((R, G), np.uint8)
        for q in range(R):
            if q == s:
                continue
            r_ = rep[(q, s)]
            flags[q] = r_["flags"]; ballot[q] = r_["ballot"]; seq[q] = r_["seq"]; deps[q] = r_["deps"]
        dec = reps[s].handle_pre_accept_replies(pa[s]["col"], ballot, seq, deps, flags)
        # slow path: Accept round for the instances that went Accepting
        slow = (dec["decision"] == 2).astype(np.uint8)
        aflags = np.zeros((R, G), np.uint8); aballot = np.zeros((R, G), np.uint64)
        if slow.any():
            for q in range(R):
                if q == s:
                    continue
                ar = reps[q].handle_accept(flags=slow, peer=u8(s), col=pa[s]["col"], ballot=np.full(G, s + 1, np.uint64),
                                           seq=dec["seq"], deps=np.ascontiguousarray(dec["deps"]),
                                           key=np.ascontiguousarray(keys[s]))
                aflags[q] = ar["flags"]; aballot[q] = ar["ballot"]
        acc = reps[s].handle_accept_replies(pa[s]["col"], aballot, aflags)
        committed = ((dec["decision"] == 3) | (acc["committed"] == 1)).astype(np.uint8)
        # the committed (seq, deps): the fast-path class resp. the Accept's
        for q in range(R):
            if q == s:
                continue
            reps[q].handle_commit_notice(flags=committed, peer=u8(s), col=pa[s]["col"], ballot=np.full(G, s + 1, np.uint64),
                                         seq=dec["seq"], deps=np.ascontiguousarray(dec["deps"]),
                                         key=np.ascontiguousarray(keys[s]))
        out.append(dict(col=pa[s]["col"], proposed=pa[s]["flags"], decision=dec["decision"], committed=committed,
                        seq=dec["seq"], deps=dec["deps"]))
    return out


def zipf_keys(rng, R, G, n_keys, p_propose=0.9):
    z = 1.0 / np.arange(1, n_keys + 1) ** 0.99
    z /= z.sum()
    k = rng.choice(n_keys, (R, G), p=z).astype(np.uint8)
    k[rng.random((R, G)) >= p_propose] = NO_KEY
    return k
