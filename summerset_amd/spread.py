"""Layout L2 for the per-replica engines: the replicas of a group live on DIFFERENT ranks (replica r
on rank r mod world), so every message crosses ranks -- the exchange step the co-located layout L1 (shard.py) does
not have.

SPMD: every rank runs the same lock-step schedule (rsp_cluster.tick) over the same list of R replica objects and
calls the same handlers in the same order.  `SpreadReplica` wraps one replica: on the rank that owns it the call
runs the handler (the HIP engine, or in tests the oracle) and its outputs -- the messages some other replica
consumes next -- are packed into ONE buffer and broadcast (RCCL over xGMI with the nccl backend, gloo on CPU); on
the other ranks the call just receives that buffer.  After a call every rank holds the same message arrays, which
is the property the quorum kernels need ("the same ack matrix visible at every rank").  One collective per handler
call, sized by the message (a [W][G] list is one buffer, not W x G sends).  First version: a broadcast delivers to
every rank what only the destination replica's rank needs; grouping a step's messages into one all-to-all is the
next refinement.  Not measured yet (needs more than one GPU)."""
import numpy as np

_G, _WG = "G", "WG"
_ACC = dict(a_n=("u4", _G), a_slot=("u4", _WG), a_val=("u4", _WG), a_ballot=("u8", _G))
RSP_OUT = {
    "req_batch": _ACC, "prepare_replies": _ACC,
    "accept": dict(r_ballot=("u8", _G), r_slot=("u4", _G)),
    "accept_replies": dict(committed=("u1", _G)),
    "become_leader": dict(hb_flags=("u1", _G), hb_ballot=("u8", _G), hb_commit=("u4", _G), hb_exec=("u4", _G), hb_snap=("u4", _G),
                          p_flags=("u1", _G), p_trig=("u4", _G), p_ballot=("u8", _G), rc_n=("u4", _G), rc_slot=("u4", _WG)),
    "prepare": dict(pr_n=("u4", _G), pr_trig=("u4", _G), pr_endp=("u4", _G), pr_ballot=("u8", _G), pr_vbal=("u8", _WG),
                    pr_vval=("u4", _WG), pr_vmask=("u1", _WG)),
    "reconstruct": dict(rr_n=("u4", _G), rr_slot=("u4", _WG), rr_bal=("u8", _WG), rr_val=("u4", _WG), rr_mask=("u1", _WG)),
    "reconstruct_reply": None,
    "heartbeat": dict(reply=("u1", _G), ballot=("u8", _G), commit_bar=("u4", _G), exec_bar=("u4", _G), snap_bar=("u4", _G)),
    "bcast_heartbeat": dict(ballot=("u8", _G), commit_bar=("u4", _G), exec_bar=("u4", _G), snap_bar=("u4", _G)),
    "is_leader": dict(leader=("u1", _G)),
}


def owner_of(replica, world):
    return replica % world


class SpreadReplica:
    """replica `rid` of every group, owned by rank owner_of(rid, world); `local` is the real object on that rank
    (numpy interface of rsp_cluster), None elsewhere"""

    def __init__(self, rid, local, G, W, rank, world, device="cpu"):
        self.me, self.local, self.G, self.W, self.rank, self.world, self.device = rid, local, G, W, rank, world, device
        self.owner = owner_of(rid, world)
        assert (local is not None) == (self.owner == rank)
        self.bytes_exchanged = 0

    def _shape(self, kind):
        return (self.G,) if kind == _G else (self.W, self.G)

    def _exchange(self, spec, out):
        """owner: pack `out` (dict of numpy arrays) and broadcast; others: receive and unpack"""
        import torch
        import torch.distributed as dist
        sizes = [int(np.prod(self._shape(k))) * np.dtype(dt).itemsize for dt, k in spec.values()]
        buf = np.zeros(sum(sizes), np.uint8)
        if out is not None:
            o = 0
            for (name, (dt, kind)), n in zip(spec.items(), sizes):
                a = np.ascontiguousarray(out[name], np.dtype(dt))
                assert a.shape == self._shape(kind), (name, a.shape)
                buf[o:o + n] = a.view(np.uint8).reshape(-1)
                o += n
        if self.world > 1:
            t = torch.from_numpy(buf).to(self.device)
            dist.broadcast(t, src=self.owner)
            buf = t.cpu().numpy()
            self.bytes_exchanged += buf.size
        res, o = {}, 0
        for (name, (dt, kind)), n in zip(spec.items(), sizes):
            res[name] = buf[o:o + n].view(np.dtype(dt)).reshape(self._shape(kind)).copy()
            o += n
        return res

    def __getattr__(self, name):
        if name not in RSP_OUT:
            raise AttributeError(name)
        spec = RSP_OUT[name]

        def call(*a, **kw):
            out = None
            if self.local is not None:
                out = getattr(self.local, name)(*a, **kw)
                if name == "is_leader":
                    out = dict(leader=out)
            if spec is None:
                return None
            res = self._exchange(spec, out)
            return res["leader"] if name == "is_leader" else res
        return call

    def preset_leader(self, leader):
        if self.local is not None:
            self.local.preset_leader(leader)

    def dump(self):
        if self.local is None:
            raise RuntimeError("replica %d lives on rank %d" % (self.me, self.owner))
        return self.local.dump()


# ---- near quorum reads over L2 (multipaxos/quorumread.rs; summerset_amd/quorumread.py) ---------------------------
def read_quorum_step(replicas, rank, world, issuer, q, keys, n, logs, flags, order=None, stable=None, kv=None, device="cpu"):
    """One ReadQuery round of replica `issuer` with the replicas of every group spread over the ranks (replica r on rank
    r mod world).  `replicas[r]` is the local object of replica r (numpy interface: handle_read_query / issue /
    handle_replies, i.e. the oracle in tests or an adapter over QuorumReadGroup) on its owner's rank, None elsewhere;
    `logs[r]` likewise the log view of replica r.  The exchange: every rank answers for its replicas, ONE all-gather of
    the packed replies (state u8 + slot u32 + val u32 + from_leader per (replica, read, group)) gives every rank the
    [R][B][G] reply arrays, the issuer's rank tallies, ONE broadcast returns the clients' answers.  `flags[R][G]` says
    which replies arrive (loss); returns (outcome, out_val, done) on every rank.  `device`: where the two collectives'
    buffers live -- "cpu" under gloo; with the nccl (RCCL) backend pass the rank's cuda device (the packed buffers are
    staged through it; this host-staged prototype is not the device-resident path, see spread_mp.py)."""
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl" and str(device) == "cpu":
        raise ValueError("read_quorum_step: the nccl backend needs device= the rank's cuda device")
    R = len(replicas)
    mine = [r for r in range(R) if owner_of(r, world) == rank]
    B, G = keys.shape
    per = B * G * 9 + G                                            # bytes of one replica's packed reply
    slots = (R + world - 1) // world                               # replicas per rank, padded
    buf = np.zeros(slots * per, np.uint8)
    own = None
    for i, r in enumerate(mine):
        st = stable if (stable is not None and r != issuer) else None
        out, fl = replicas[r].handle_read_query(keys, n, logs[r], st, kv if st is not None else None)
        if r == issuer:
            own = out
        o = i * per
        buf[o:o + B * G] = out["state"].ravel(); o += B * G
        buf[o:o + 4 * B * G] = out["slot"].astype(np.uint32).ravel().view(np.uint8); o += 4 * B * G
        buf[o:o + 4 * B * G] = out["val"].astype(np.uint32).ravel().view(np.uint8); o += 4 * B * G
        buf[o:o + G] = fl
    gathered = [torch.zeros(slots * per, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(buf).to(device))
    rep = dict(state=np.zeros((R, B, G), np.uint8), slot=np.zeros((R, B, G), np.uint32), val=np.zeros((R, B, G), np.uint32))
    from_leader = np.zeros((R, G), np.uint8)
    for rk in range(world):
        a = gathered[rk].cpu().numpy()
        for i, r in enumerate([x for x in range(R) if owner_of(x, world) == rk]):
            o = i * per
            rep["state"][r] = a[o:o + B * G].reshape(B, G); o += B * G
            rep["slot"][r] = a[o:o + 4 * B * G].view(np.uint32).reshape(B, G); o += 4 * B * G
            rep["val"][r] = a[o:o + 4 * B * G].view(np.uint32).reshape(B, G); o += 4 * B * G
            from_leader[r] = a[o:o + G]
    res = np.zeros(B * G * 5 + G, np.uint8)
    if owner_of(issuer, world) == rank:
        replicas[issuer].issue(q, n, own)
        fl = (flags & 1) | (from_leader << 1) * (flags & 1)
        fl[issuer] = 0
        outcome, out_val, done = replicas[issuer].handle_replies(q, rep, fl.astype(np.uint8), order)
        res[:B * G] = outcome.ravel()
        res[B * G:5 * B * G] = out_val.astype(np.uint32).ravel().view(np.uint8)
        res[5 * B * G:] = done
    t = torch.from_numpy(res).to(device)
    dist.broadcast(t, src=owner_of(issuer, world))
    res = t.cpu().numpy()
    return (res[:B * G].reshape(B, G).copy(), res[B * G:5 * B * G].view(np.uint32).reshape(B, G).copy(), res[5 * B * G:].copy())


def read_quorum_step_device(replicas, rank, world, issuer, q, keys, n, logs, flags, order=None, stable=None, kv=None):
    """`read_quorum_step` device-resident, for `QuorumReadGroup` objects: every argument a device tensor (keys uint8 [B, G],
    n uint8 [G], flags uint8 [R, G], order int32 [G], stable uint8 [G], kv int32 [K, G]; logs[r] as
    `handle_msg_read_query` takes it), every intermediate one too.  The replies of this rank's replicas are byte-viewed
    into ONE send tensor (val, slot int32 [B, G], state uint8 [B, G], from_leader uint8 [G] per replica, padded to 8
    bytes), ONE all_gather on device tensors gives every rank all of them, the [R, B, G] arrays the tally kernel takes are
    views of the gathered buffer restacked by replica id, the issuer's rank tallies, ONE broadcast of (out_val int32, outcome
    uint8 [B, G], done uint8 [G]) returns the clients' answers -- no `.cpu()`, no numpy.  Returns (outcome, out_val, done)
    as device tensors on every rank."""
    import torch
    import torch.distributed as dist
    R = len(replicas)
    mine = [r for r in range(R) if owner_of(r, world) == rank]
    B, G = keys.shape
    dev = keys.device
    per = (9 * B * G + G + 7) & ~7                                 # bytes of one replica's packed reply
    slots = (R + world - 1) // world                               # replicas per rank, padded
    pad = torch.zeros(per - 9 * B * G - G, dtype=torch.uint8, device=dev)
    parts, own = [], None
    for r in mine:
        st = stable if (stable is not None and r != issuer) else None
        out, fl = replicas[r].handle_msg_read_query(keys, n, logs[r], st, kv if st is not None else None)
        if r == issuer:
            own = out
        parts += [out["val"].contiguous().view(torch.uint8).reshape(-1), out["slot"].contiguous().view(torch.uint8).reshape(-1),
                  out["state"].reshape(-1), fl, pad]
    if len(mine) < slots:
        parts.append(torch.zeros((slots - len(mine)) * per, dtype=torch.uint8, device=dev))
    send = torch.cat(parts)
    big = torch.empty(world * slots * per, dtype=torch.uint8, device=dev)
    if world > 1:
        dist.all_gather(list(big.view(world, slots * per).unbind(0)), send)
    else:
        big.copy_(send)

    def field(r, off, nbytes, dt, shape):                          # replica r's field: a view of the gathered buffer
        o = (owner_of(r, world) * slots + r // world) * per + off
        return big[o:o + nbytes].view(dt).reshape(shape)
    rep = dict(val=torch.stack([field(r, 0, 4 * B * G, torch.int32, (B, G)) for r in range(R)]),
               slot=torch.stack([field(r, 4 * B * G, 4 * B * G, torch.int32, (B, G)) for r in range(R)]),
               state=torch.stack([field(r, 8 * B * G, B * G, torch.uint8, (B, G)) for r in range(R)]))
    from_leader = torch.stack([field(r, 9 * B * G, G, torch.uint8, (G,)) for r in range(R)])
    res = torch.zeros((5 * B * G + G + 7) & ~7, dtype=torch.uint8, device=dev)
    if owner_of(issuer, world) == rank:
        replicas[issuer].issue(q, n, own)
        got = flags & 1
        fl = (got | ((from_leader << 1) * got)).to(torch.uint8)
        fl[issuer] = 0
        outcome, out_val, done = replicas[issuer].handle_msg_read_query_reply(q, rep, fl.contiguous(), order)
        res = torch.cat([out_val.contiguous().view(torch.uint8).reshape(-1), outcome.reshape(-1), done,
                         torch.zeros(res.numel() - 5 * B * G - G, dtype=torch.uint8, device=dev)])
    if world > 1:
        dist.broadcast(res, src=owner_of(issuer, world))
    return (res[4 * B * G:5 * B * G].reshape(B, G), res[:4 * B * G].view(torch.int32).reshape(B, G), res[5 * B * G:5 * B * G + G])


# ---- MultiPaxos cluster engine: the AcceptReply exchange of L2 (multipaxos.MultiPaxosCluster.collect_acks / deliver_acks) ----
def split_acks_by_owner(rec, total_groups, world):
    """ACK_DTYPE records with GLOBAL group ids -> one array per rank, by the rank that owns the group's leader-side state
    (shard.group_range blocks); group ids are rebased to the owner's block"""
    from . import shard
    out = []
    for k in range(world):
        lo, hi = shard.group_range(total_groups, world, k)
        part = rec[(rec["group"] >= lo) & (rec["group"] < hi)].copy()
        part["group"] -= lo
        out.append(part)
    return out


def mp_exchange_acks(outgoing, rank, world, device="cpu"):
    """The follower->leader half of a tick's all-to-all: `outgoing[k]` = the AcceptReply records
    (multipaxos.ACK_DTYPE, 24 B each) the replicas hosted on this rank produced for leaders whose state lives on rank k
    (smr_mp_collect_acks + split_acks_by_owner).  One all-gather of the world x world count matrix, then ONE collective
    for the records -- all_to_all_single with the exact split sizes over RCCL (nccl backend); an all-gather of padded
    buffers where the backend has no all-to-all (gloo on CPU).  Returns the records addressed to this rank, sender-major,
    ready for smr_mp_deliver_acks before the quorum kernel (R3)."""
    import torch
    import torch.distributed as dist
    from .multipaxos import ACK_DTYPE
    B = ACK_DTYPE.itemsize
    assert len(outgoing) == world
    if world == 1:
        return outgoing[0].copy()
    cnt = torch.tensor([len(o) for o in outgoing], dtype=torch.int64, device=device)
    allcnt = [torch.zeros(world, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(allcnt, cnt)
    counts = torch.stack(allcnt).cpu().numpy()                       # counts[src][dst]
    send = np.concatenate([o.view(np.uint8).reshape(-1) for o in outgoing]) if counts[rank].sum() else np.zeros(0, np.uint8)
    if dist.get_backend() == "nccl":
        t_in = torch.from_numpy(send.copy()).to(device)
        t_out = torch.empty(int(counts[:, rank].sum()) * B, dtype=torch.uint8, device=device)
        dist.all_to_all_single(t_out, t_in, output_split_sizes=[int(c) * B for c in counts[:, rank]],
                               input_split_sizes=[int(c) * B for c in counts[rank]])
        got = t_out.cpu().numpy()
    else:
        width = int(counts.sum(axis=1).max()) * B                     # every rank's whole send buffer, padded
        buf = np.zeros(max(width, 1), np.uint8)
        buf[:send.size] = send
        gathered = [torch.zeros(max(width, 1), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(gathered, torch.from_numpy(buf))
        parts = []
        for src in range(world):
            off = int(counts[src][:rank].sum()) * B
            parts.append(gathered[src].numpy()[off:off + int(counts[src][rank]) * B])
        got = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    return got.view(ACK_DTYPE).copy()
