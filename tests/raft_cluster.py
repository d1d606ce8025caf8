"""Closed-loop Raft cluster out of R per-replica handler objects (backend-agnostic: RaftOracle or the
HIP RaftLeaderGroup behind a numpy adapter).  One tick: HearTimeouts -> RequestVote round -> vote
replies; client batches at whoever leads; the AppendEntries those appends produce (one combined
message per leader and follower) -> replies -> the leader's match-index quorum.  No message loss."""
import numpy as np

NO, NONE32 = 0xFF, 0xFFFFFFFF
FOLLOWER, CANDIDATE, LEADER = 0, 1, 2


class NumpyRaft:
    """RaftLeaderGroup (device tensors) behind the RaftOracle-style numpy interface"""

    def __init__(self, eng, cuda):
        import torch
        self.e, self.cuda, self.torch = eng, cuda, torch
        self.G, self.R = eng.G, eng.R

    def _t(self, a):
        if a is None:
            return None
        v = a.view(np.int64) if a.dtype == np.uint64 else (a.view(np.int32) if a.dtype == np.uint32 else a)
        return self.torch.from_numpy(np.ascontiguousarray(v)).to(self.cuda)

    @staticmethod
    def _n(d, like):
        return {k: d[k].cpu().numpy().view(v) for k, v in like.items()}

    def preset(self, *a):
        self.e.preset(*a)

    def become_candidate(self, src):
        return self._n(self.e.become_a_candidate(self._t(src)), dict(flags=np.uint8, term=np.uint64, last_slot=np.uint32,
                                                                  last_term=np.uint64))

    def handle_request_vote(self, flags, candidate, term, last_slot, last_term):
        o = self.e.handle_msg_request_vote(self._t(flags), self._t(candidate), self._t(term), self._t(last_slot),
                                           self._t(last_term))
        return self._n(o, dict(flags=np.uint8, term=np.uint64))

    def handle_vote_replies(self, term, flags, order=None):
        o = self.e.handle_msg_request_vote_reply(self._t(term), self._t(flags), self._t(order))
        return self._n(o, dict(hb_prev_slot=np.uint32, elected=np.uint8))

    def append_emit(self, n_new):
        return self.e.handle_req_batch_emit(self._t(n_new)).cpu().numpy().view(np.uint32)

    def gather_entries(self, first, K):
        o = self.e.gather_entries(self._t(np.ascontiguousarray(first)), K)
        return self._n(o, dict(flags=np.uint8, leader=np.uint8, term=np.uint64, prev_slot=np.uint32, prev_term=np.uint64,
                               n_entries=np.uint32, entry_term=np.uint64, leader_commit=np.uint32, last_snap=np.uint32))

    def handle_append_entries(self, **m):
        o = self.e.handle_msg_append_entries(**{k: self._t(v) for k, v in m.items()})
        return self._n(o, dict(flags=np.uint8, term=np.uint64, end_slot=np.uint32, conflict_term=np.uint64,
                               conflict_slot=np.uint32))

    def replicate_many(self, followers, firsts, K):
        """my AppendEntries for `followers` (NumpyRaft objects) and their handlers in one launch; returns [(message, reply)] as numpy"""
        torch, dev, G = self.torch, self.cuda, self.G
        msgs = [self.e.new_message(K, dev) for _ in followers]
        z = lambda dt: torch.zeros(G, dtype=dt, device=dev)
        reps = [dict(flags=z(torch.uint8), term=z(torch.int64), end_slot=z(torch.int32), conflict_term=z(torch.int64), conflict_slot=z(torch.int32))
                for _ in followers]
        self.e.replicate_many([f.e for f in followers], [self._t(np.ascontiguousarray(f)) for f in firsts], msgs, reps)
        ml = dict(flags=np.uint8, leader=np.uint8, term=np.uint64, prev_slot=np.uint32, prev_term=np.uint64, n_entries=np.uint32,
                  entry_term=np.uint64, leader_commit=np.uint32, last_snap=np.uint32)
        rl = dict(flags=np.uint8, term=np.uint64, end_slot=np.uint32, conflict_term=np.uint64, conflict_slot=np.uint32)
        return [(self._n(m, ml), self._n(r, rl)) for m, r in zip(msgs, reps)]

    def cluster_tick(self, n_new, followers, K):
        """my append + my AppendEntries for `followers` + their handlers + my reply handler in ONE launch (`smr_raft_cluster_tick`);
        returns (first [R][G], [(message, reply)]) as numpy"""
        torch, dev, G, R = self.torch, self.cuda, self.G, self.R
        msgs = [self.e.new_message(K, dev) for _ in followers]
        z = lambda dt: torch.zeros((R, G), dtype=dt, device=dev)
        arr = dict(flags=z(torch.uint8), term=z(torch.int64), end_slot=z(torch.int32), conflict_term=z(torch.int64), conflict_slot=z(torch.int32))
        first = torch.zeros((R, G), dtype=torch.int32, device=dev)
        reps = [{k: v[f.e.me] for k, v in arr.items()} for f in followers]
        self.e.cluster_tick(self._t(np.ascontiguousarray(n_new)), first, [f.e for f in followers], msgs, reps, arr["term"], arr["end_slot"], arr["flags"],
                            arr["conflict_term"], arr["conflict_slot"])
        ml = dict(flags=np.uint8, leader=np.uint8, term=np.uint64, prev_slot=np.uint32, prev_term=np.uint64, n_entries=np.uint32,
                  entry_term=np.uint64, leader_commit=np.uint32, last_snap=np.uint32)
        rl = dict(flags=np.uint8, term=np.uint64, end_slot=np.uint32, conflict_term=np.uint64, conflict_slot=np.uint32)
        return first.cpu().numpy().view(np.uint32), [(self._n(m, ml), self._n(r, rl)) for m, r in zip(msgs, reps)]

    def handle_replies(self, reply_term, end_slot, flags, conflict_term=None, conflict_slot=None, order=None):
        self.e.handle_msg_append_entries_reply(self._t(reply_term), self._t(end_slot), self._t(flags), self._t(conflict_term),
                                               self._t(conflict_slot), self._t(order))

    def dump(self):
        return self.e.dump()

    def dump_votes(self):
        return self.e.dump_votes()


def tick(reps, timeouts, n_new, K, via=None, sender_major=False, one_launch=False, seen=None, sender_ticks=False):
    """timeouts[r][G]: HearTimeout source at replica r (0xFF none); n_new[r][G]: client batches handed to
    replica r (those that do not lead redirect them).  Returns nothing; state lives in the replicas.
    sender_major: the replication step goes sender by sender (every follower handles sender 0's message, then sender 1's ..)
    instead of receiver by receiver -- the order in which `one_launch` (NumpyRaft.replicate_many: a sender's messages and their
    handlers in one launch) can stand for the calls; seen (a list): the (sender, receiver, message, reply) tuples of the step.
    sender_ticks: a sender's append, replication and replies before the next sender's append -- the order in which `one_launch="tick"`
    (NumpyRaft.cluster_tick: all three in one launch) can stand for the calls.
    via (optional): via(s, rt, es, fl, ct, cs) -> the same five [R][G] arrays -- the AppendEntriesReplies on their way to
    leader s (tests/test_zz_reply_ingest_gpu.py sends them as frames through the device parser)."""
    R = len(reps)
    G = timeouts.shape[1]
    u8 = lambda v: np.full(G, v, np.uint8)
    # elections
    rv = [reps[r].become_candidate(np.ascontiguousarray(timeouts[r])) for r in range(R)]
    vote = {}
    for q in range(R):
        for c in range(R):
            if c == q:
                continue
            vote[(q, c)] = reps[q].handle_request_vote(rv[c]["flags"], u8(c), rv[c]["term"], rv[c]["last_slot"],
                                                       rv[c]["last_term"])
    for c in range(R):
        term = np.zeros((R, G), np.uint64); flags = np.zeros((R, G), np.uint8)
        for q in range(R):
            if q != c:
                term[q] = vote[(q, c)]["term"]; flags[q] = vote[(q, c)]["flags"] & 1
        reps[c].handle_vote_replies(term, flags)
    # replication
    if sender_ticks:                    # sender by sender, each its WHOLE tick: append, AppendEntries + handlers, replies
        for s in range(R):
            qs = [q for q in range(R) if q != s]
            if one_launch == "tick":
                _, out = reps[s].cluster_tick(n_new[s], [reps[q] for q in qs], K)
            else:
                f = reps[s].append_emit(np.ascontiguousarray(n_new[s]))
                out = []
                for q in qs:
                    m = reps[s].gather_entries(f[q], K)
                    out.append((m, reps[q].handle_append_entries(**m)))
                rt = np.zeros((R, G), np.uint64); es = np.zeros((R, G), np.uint32); fl = np.zeros((R, G), np.uint8)
                ct = np.zeros((R, G), np.uint64); cs = np.zeros((R, G), np.uint32)
                for q, (m, r_) in zip(qs, out):
                    rt[q] = r_["term"]; es[q] = r_["end_slot"]; fl[q] = r_["flags"]; ct[q] = r_["conflict_term"]; cs[q] = r_["conflict_slot"]
                reps[s].handle_replies(rt, es, fl, ct, cs)
            if seen is not None:
                for q, (m, r_) in zip(qs, out):
                    seen.append((s, q, m, r_))
        return
    first = [reps[r].append_emit(np.ascontiguousarray(n_new[r])) for r in range(R)]
    rep = {}
    if sender_major:
        for s in range(R):
            qs = [q for q in range(R) if q != s]
            if one_launch:
                out = reps[s].replicate_many([reps[q] for q in qs], [first[s][q] for q in qs], K)
            else:
                out = []
                for q in qs:
                    m = reps[s].gather_entries(first[s][q], K)
                    out.append((m, reps[q].handle_append_entries(**m)))
            for q, (m, r) in zip(qs, out):
                rep[(q, s)] = r
                if seen is not None:
                    seen.append((s, q, m, r))
    for q in range(R) if not sender_major else ():
        for s in range(R):
            if s == q:
                continue
            m = reps[s].gather_entries(first[s][q], K)
            rep[(q, s)] = reps[q].handle_append_entries(**m)
    for s in range(R):
        rt = np.zeros((R, G), np.uint64); es = np.zeros((R, G), np.uint32); fl = np.zeros((R, G), np.uint8)
        ct = np.zeros((R, G), np.uint64); cs = np.zeros((R, G), np.uint32)
        for q in range(R):
            if q == s:
                continue
            r_ = rep[(q, s)]
            rt[q] = r_["term"]; es[q] = r_["end_slot"]; fl[q] = r_["flags"]; ct[q] = r_["conflict_term"]; cs[q] = r_["conflict_slot"]
        if via is not None:
            rt, es, fl, ct, cs = via(s, rt, es, fl, ct, cs)
        reps[s].handle_replies(rt, es, fl, ct, cs)
