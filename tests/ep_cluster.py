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

    def _m(self, flags, peer, col, ballot, seq, deps, key, row):
        m = dict(flags=flags, peer=peer, col=col, ballot=ballot, seq=seq, deps=deps, key=key, row=row)
        return {k: self._t(v) for k, v in m.items() if v is not None}

    def handle_pre_accept(self, flags, peer, col, ballot, seq, deps, key, row=None):
        o = self.e.handle_msg_pre_accept(self._m(flags, peer, col, ballot, seq, deps, key, row))
        self._after_call()
        return self._n(o, dict(flags=np.uint8, ballot=np.uint64, seq=np.uint64, deps=np.uint32))

    def handle_accept(self, flags, peer, col, ballot, seq, deps, key, row=None):
        o = self.e.handle_msg_accept(self._m(flags, peer, col, ballot, seq, deps, key, row))
        self._after_call()
        return self._n(o, dict(flags=np.uint8, ballot=np.uint64))

    def handle_commit_notice(self, flags, peer, col, ballot, seq, deps, key, row=None):
        self.e.handle_msg_commit_notice(self._m(flags, peer, col, ballot, seq, deps, key, row))
        self._after_call()

    def handle_pre_accept_replies(self, col, ballot, seq, deps, flags, order=None, exploded=None, row=None):
        o = self.e.handle_msg_pre_accept_reply(self._t(col), self._t(ballot), self._t(seq), self._t(deps), self._t(flags),
                                               self._t(order), self._t(exploded), self._t(row))
        self._after_call()
        return self._n(o, dict(decision=np.uint8, seq=np.uint64, deps=np.uint32))

    def handle_accept_replies(self, col, ballot, flags, order=None, row=None):
        o = self.e.handle_msg_accept_reply(self._t(col), self._t(ballot), self._t(flags), self._t(order), self._t(row))
        self._after_call()
        return self._n(o, dict(committed=np.uint8))

    def heartbeat_timeout(self, src, exploded=None):
        o = self.e.heartbeat_timeout(self._t(src), self._t(exploded))
        self._after_call()
        return self._n(o, dict(n=np.uint32, col=np.uint32, ballot=np.uint64))

    def handle_exp_prepare(self, flags, peer, row, col, new_ballot):
        o = self.e.handle_msg_exp_prepare(dict(flags=self._t(flags), peer=self._t(peer), row=self._t(row), col=self._t(col),
                                               new_ballot=self._t(new_ballot)))
        self._after_call()
        r = self._n(o, dict(flags=np.uint8, voted_bal=np.uint64, voted_status=np.uint8, voted_seq=np.uint64, voted_deps=np.uint32,
                            voted_key=np.uint8))
        return dict(flags=r["flags"], voted_bal=r["voted_bal"], status=r["voted_status"], seq=r["voted_seq"], deps=r["voted_deps"],
                    key=r["voted_key"])

    def handle_exp_prepare_replies(self, row, col, new_ballot, voted_bal, voted_status, voted_seq, voted_deps, voted_key, flags,
                                   order=None):
        rep = dict(flags=self._t(flags), voted_bal=self._t(voted_bal), voted_status=self._t(voted_status), voted_seq=self._t(voted_seq),
                   voted_deps=self._t(voted_deps), voted_key=self._t(voted_key))
        o = self.e.handle_msg_exp_prepare_reply(self._t(row), self._t(col), self._t(new_ballot), rep, self._t(order))
        self._after_call()
        return self._n(o, dict(decision=np.uint8, ballot=np.uint64, seq=np.uint64, deps=np.uint32, key=np.uint8))

    def xp_dump(self):
        return self.e.xp_dump()

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


def tick(reps, keys, drop=None, via=None, phase_major=False):
    """reps[r]: backend of replica r; keys[r][G]: proposal of replica r (0xFF none);
    drop[(s, q)] (optional): bool [G] -- the PreAccept from s to q is lost (with its reply).
    via (optional): via(s, col, ballot, seq, deps, flags) -> (ballot, seq, deps, flags) -- the PreAcceptReplies on their way
    to command leader s (tests/test_zz_reply_ingest_gpu.py sends them as frames through the device parser).
    phase_major: the leaders' part of the tick phase by phase instead of leader by leader (every leader's replies, then every
    Accept round, every AcceptReply tally, every CommitNotice): the order smr_ep_cluster_set_mode(c, 2) runs.
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
    dec, slowf, aflags, aballot, committed = [None] * R, [None] * R, [None] * R, [None] * R, [None] * R

    def replies(s):
        ballot = np.zeros((R, G), np.uint64); seq = np.zeros((R, G), np.uint64)
        deps = np.full((R, R, G), N, np.uint32); flags = np.zeros((R, G), np.uint8)
        for q in range(R):
            if q == s:
                continue
            r_ = rep[(q, s)]
            flags[q] = r_["flags"]; ballot[q] = r_["ballot"]; seq[q] = r_["seq"]; deps[q] = r_["deps"]
        if via is not None:
            ballot, seq, deps, flags = via(s, pa[s]["col"], ballot, seq, deps, flags)
        dec[s] = reps[s].handle_pre_accept_replies(pa[s]["col"], ballot, seq, deps, flags)
        slowf[s] = (dec[s]["decision"] == 2).astype(np.uint8)

    def accepts(s):                                          # slow path: Accept round for the instances that went Accepting
        aflags[s] = np.zeros((R, G), np.uint8); aballot[s] = np.zeros((R, G), np.uint64)
        if slowf[s].any():
            for q in range(R):
                if q == s:
                    continue
                ar = reps[q].handle_accept(flags=slowf[s], peer=u8(s), col=pa[s]["col"], ballot=np.full(G, s + 1, np.uint64),
                                           seq=dec[s]["seq"], deps=np.ascontiguousarray(dec[s]["deps"]),
                                           key=np.ascontiguousarray(keys[s]))
                aflags[s][q] = ar["flags"]; aballot[s][q] = ar["ballot"]

    def accept_replies(s):
        acc = reps[s].handle_accept_replies(pa[s]["col"], aballot[s], aflags[s])
        committed[s] = ((dec[s]["decision"] == 3) | (acc["committed"] == 1)).astype(np.uint8)

    def commits(s):                                          # the committed (seq, deps): the fast-path class resp. the Accept's
        for q in range(R):
            if q == s:
                continue
            reps[q].handle_commit_notice(flags=committed[s], peer=u8(s), col=pa[s]["col"], ballot=np.full(G, s + 1, np.uint64),
                                         seq=dec[s]["seq"], deps=np.ascontiguousarray(dec[s]["deps"]),
                                         key=np.ascontiguousarray(keys[s]))

    phases = (replies, accepts, accept_replies, commits)
    if phase_major:                                          # (smr_ep_cluster_set_mode bit 1: another legal delivery order)
        for ph in phases:
            for s in range(R):
                ph(s)
    else:
        for s in range(R):
            for ph in phases:
                ph(s)
    return [dict(col=pa[s]["col"], proposed=pa[s]["flags"], decision=dec[s]["decision"], committed=committed[s],
                 seq=dec[s]["seq"], deps=dec[s]["deps"]) for s in range(R)]


def zipf_keys(rng, R, G, n_keys, p_propose=0.9):
    z = 1.0 / np.arange(1, n_keys + 1) ** 0.99
    z /= z.sum()
    k = rng.choice(n_keys, (R, G), p=z).astype(np.uint8)
    k[rng.random((R, G)) >= p_propose] = NO_KEY
    return k


# ---- explicit prepare: a command leader dies mid-instance, the others recover its row ------------------------------------
def crash_tick(reps, dead, keys, rng, G):
    """replica `dead` proposes keys[g]; per group the run is cut at a seeded point: (0) PreAccepts reach a random subset of
    the peers and nothing comes back; (1) all replies come back and the leader decides, a slow-path Accept reaches a random
    subset; (2) the instance commits at the leader and the CommitNotice reaches a random subset.  After this call `dead`
    is never driven again.  Returns the cut per group."""
    R = len(reps)
    u8 = lambda v: np.full(G, v, np.uint8)
    cut = rng.integers(0, 3, G)
    pa = reps[dead].propose(np.ascontiguousarray(keys))
    reach = {q: rng.random(G) < 0.6 for q in range(R) if q != dead}
    rep = {}
    for q in range(R):
        if q == dead:
            continue
        fl = pa["flags"].copy()
        fl[(cut == 0) & ~reach[q]] = 0
        rep[q] = reps[q].handle_pre_accept(flags=fl, peer=u8(dead), col=pa["col"], ballot=np.full(G, dead + 1, np.uint64),
                                           seq=pa["seq"], deps=np.ascontiguousarray(pa["deps"]), key=np.ascontiguousarray(keys))
    ballot = np.zeros((R, G), np.uint64); seq = np.zeros((R, G), np.uint64)
    deps = np.full((R, R, G), N, np.uint32); flags = np.zeros((R, G), np.uint8)
    for q in rep:
        flags[q] = rep[q]["flags"] * (cut > 0); ballot[q] = rep[q]["ballot"]; seq[q] = rep[q]["seq"]; deps[q] = rep[q]["deps"]
    dec = reps[dead].handle_pre_accept_replies(pa["col"], ballot, seq, deps, flags)
    slow = (dec["decision"] == 2)
    aflags = np.zeros((R, G), np.uint8); aballot = np.zeros((R, G), np.uint64)
    for q in range(R):
        if q == dead:
            continue
        fl = (slow & ((cut == 2) | reach[q])).astype(np.uint8)
        ar = reps[q].handle_accept(flags=fl, peer=u8(dead), col=pa["col"], ballot=np.full(G, dead + 1, np.uint64), seq=dec["seq"],
                                   deps=np.ascontiguousarray(dec["deps"]), key=np.ascontiguousarray(keys))
        aflags[q] = ar["flags"] * (cut == 2); aballot[q] = ar["ballot"]
    acc = reps[dead].handle_accept_replies(pa["col"], aballot, aflags)
    committed = ((dec["decision"] == 3) | (acc["committed"] == 1)) & (cut == 2)
    for q in range(R):
        if q == dead:
            continue
        reps[q].handle_commit_notice(flags=(committed & reach[q]).astype(np.uint8), peer=u8(dead), col=pa["col"],
                                     ballot=np.full(G, dead + 1, np.uint64), seq=dec["seq"], deps=np.ascontiguousarray(dec["deps"]),
                                     key=np.ascontiguousarray(keys))
    return cut


def recover_row(reps, who, dead, live, G, rng=None, loss=0.0, trace=None):
    """replica `who` times out on `dead` and recovers its row with the help of the replicas in `live` (messages to and from
    each of them lost with probability `loss`).  Every output of every call goes to `trace` (a list) for comparison
    between backends.  Returns how many instances reached which decision."""
    R = len(reps)
    u8 = lambda v: np.full(G, v, np.uint8)
    rec = (lambda *x: trace.append(x)) if trace is not None else (lambda *x: None)
    lost = (lambda: (rng.random(G) < loss)) if (rng is not None and loss > 0) else (lambda: np.zeros(G, bool))
    exploded = u8(1 << dead)
    hb = reps[who].heartbeat_timeout(u8(dead), exploded)
    rec("hb", hb["n"].copy(), hb["col"].copy(), hb["ballot"].copy())
    tally = {1: 0, 2: 0, 3: 0}
    row = u8(dead)
    for k in range(int(hb["n"].max()) if G else 0):
        on = (hb["n"] > k)
        col, nb = np.ascontiguousarray(hb["col"][k]), np.ascontiguousarray(hb["ballot"][k])
        nbal = np.zeros((R, G), np.uint64); vb = np.zeros((R, G), np.uint64); vs = np.zeros((R, G), np.uint8)
        vq = np.zeros((R, G), np.uint64); vd = np.full((R, R, G), N, np.uint32); vk = np.full((R, G), NO_KEY, np.uint8)
        fl = np.zeros((R, G), np.uint8)
        for q in live:
            if q == who:
                continue
            r_ = reps[q].handle_exp_prepare((on & ~lost()).astype(np.uint8), u8(who), row, col, nb)
            rec("xp", q, k, *[r_[x].copy() for x in ("flags", "voted_bal", "status", "seq", "deps", "key")])
            fl[q] = r_["flags"] * ~lost(); nbal[q] = nb; vb[q] = r_["voted_bal"]; vs[q] = r_["status"]; vq[q] = r_["seq"]
            vd[q] = r_["deps"]; vk[q] = r_["key"]
        dec = reps[who].handle_exp_prepare_replies(row, col, nbal, vb, vs, vq, vd, vk, fl)
        rec("dec", k, *[dec[x].copy() for x in ("decision", "ballot", "seq", "deps", "key")])
        for st in (1, 2, 3):
            tally[st] += int((dec["decision"] == st).sum())
        seq, deps, key, bal = dec["seq"], np.ascontiguousarray(dec["deps"]), dec["key"], dec["ballot"]
        # PreAccepting: a PreAccept round under the new ballot; the instance avoids the fast path
        pre = dec["decision"] == 1
        if pre.any():
            ballot = np.zeros((R, G), np.uint64); rs = np.zeros((R, G), np.uint64)
            rd = np.full((R, R, G), N, np.uint32); rf = np.zeros((R, G), np.uint8)
            for q in live:
                if q == who:
                    continue
                r_ = reps[q].handle_pre_accept(flags=(pre & ~lost()).astype(np.uint8), peer=u8(who), col=col, ballot=bal, seq=seq,
                                               deps=deps, key=key, row=row)
                rf[q] = r_["flags"] * ~lost(); ballot[q] = r_["ballot"]; rs[q] = r_["seq"]; rd[q] = r_["deps"]
            d2 = reps[who].handle_pre_accept_replies(col, ballot, rs, rd, rf, exploded=exploded, row=row)
            rec("pre", k, *[d2[x].copy() for x in ("decision", "seq", "deps")])
            assert not ((d2["decision"] == 3) & pre).any()         # no fast path after an explicit prepare
            took = pre & (d2["decision"] == 2)
            seq = np.where(took, d2["seq"], seq); deps = np.where(took[None, :], d2["deps"], deps)
            acc_now = (dec["decision"] == 2) | took
        else:
            acc_now = dec["decision"] == 2
        committed = dec["decision"] == 3
        if acc_now.any():
            aflags = np.zeros((R, G), np.uint8); aballot = np.zeros((R, G), np.uint64)
            for q in live:
                if q == who:
                    continue
                ar = reps[q].handle_accept(flags=(acc_now & ~lost()).astype(np.uint8), peer=u8(who), col=col, ballot=bal, seq=seq,
                                           deps=np.ascontiguousarray(deps), key=key, row=row)
                aflags[q] = ar["flags"] * ~lost(); aballot[q] = ar["ballot"]
            acc = reps[who].handle_accept_replies(col, aballot, aflags, row=row)
            rec("acc", k, acc["committed"].copy())
            committed = committed | (acc["committed"] == 1)
        for q in live:
            if q == who:
                continue
            reps[q].handle_commit_notice(flags=(committed & ~lost()).astype(np.uint8), peer=u8(who), col=col, ballot=bal, seq=seq,
                                         deps=np.ascontiguousarray(deps), key=key, row=row)
    return tally


def check_agreement(reps, live, dead, G):
    """safety: wherever two live replicas hold a column of the dead row as committed, they hold the same (seq, deps, key)"""
    dumps = {q: reps[q].dump() for q in live}
    W = reps[live[0]].W
    n_committed = 0
    for a in live:
        for b in live:
            if b <= a:
                continue
            da, db = dumps[a], dumps[b]
            both = (da["status"][dead] >= 3) & (db["status"][dead] >= 3) & (da["len"][dead][None, :] == db["len"][dead][None, :])
            assert (da["seq"][dead][both] == db["seq"][dead][both]).all()
            assert (da["key"][dead][both] == db["key"][dead][both]).all()
            assert (da["deps"][dead][both] == db["deps"][dead][both]).all()
            n_committed += int(both.sum())
    return n_committed
