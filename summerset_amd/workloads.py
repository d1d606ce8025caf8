"""The launch shapes bench.py TIMES, built in one place so the parity tests at BASELINE size run exactly them.

VERDICT r3 weak #1: bench.py timed `smr_mp_run_ticks` in batches of 8 with the straggler list on while the 65 536-group
parity test ran one `smr_mp_tick` per tick with the list off.  Both now take their cluster, their stream and their
launch mode from here: `headline_cluster` / `headline_stream` / `drive_headline` for the MultiPaxos headline
(reference: multipaxos/request.rs:156-224, messages.rs:295-443), `config4_*` for the RSPaxos one-launch tick with the
fused encode + fan-out (rspaxos/request.rs:71-142).
"""
import os

import numpy as np

# what the headline line is quoted on (BASELINE.json: 65 536 groups x 5 replicas; bench.py's defaults)
HEADLINE = dict(R=5, S=32, W=512, H=4, drop_p=0.1, max_drop=2, straggler_ticks=4, batch=8)
MAX_TICKS_PER_CALL = 16                      # smr_mp_run_ticks takes at most this many tick descriptors per call


def headline_cluster(G, W=HEADLINE["W"], R=HEADLINE["R"], straggler_ticks=HEADLINE["straggler_ticks"], role_rotation=False):
    """the MultiPaxos cluster of the headline line: ring of W slots, W // 8 of them held back for re-Accept rounds,
    outboxes of W + 4 entries, replica 0 preset as every group's leader"""
    from .multipaxos import MultiPaxosCluster
    eng = MultiPaxosCluster(G, R, W, win_reserve=W // 8, outbox_cap=W + 4, straggler_ticks=straggler_ticks)
    if role_rotation and straggler_ticks:
        eng.set_role_rotation(True)
    eng.preset_leader(0)
    return eng


def headline_stream(G, n_ticks, timeout_frac, timeout_span, S=HEADLINE["S"], W=HEADLINE["W"], R=HEADLINE["R"], H=HEADLINE["H"],
                    drop_p=HEADLINE["drop_p"], group_base=0):
    """the synthetic client-op / loss / timeout stream of the headline line, keyed by the GLOBAL group id"""
    from . import stream
    return stream.MultiPaxosStream(G, R, S, cap=W + 4, n_ticks=n_ticks, drop_p=drop_p, timeout_frac=timeout_frac, hb_every=H,
                                   rand_rows=S + 4, max_drop=HEADLINE["max_drop"], timeout_span=timeout_span, group_base=group_base)


def batches(t0, t1, batch):
    """[t0, t1) cut into the chunks one smr_mp_run_ticks call takes"""
    step = min(batch, MAX_TICKS_PER_CALL)
    return [list(range(b0, min(b0 + step, t1))) for b0 in range(t0, t1, step)]


def drive_headline(eng, tick_args, t0, t1, batch=HEADLINE["batch"], before_call=None):
    """ticks [t0, t1) the way the headline line runs them: batch > 0 -- `smr_mp_run_ticks` over chunks of `batch` ticks (the
    bulk kernels tick by tick, the straggler list's groups through the chunk in one side-stream launch); batch == 0 -- one
    `smr_mp_tick` per tick.  tick_args(t) -> the tick's arguments; before_call(i, n_calls, ticks) runs ahead of call i."""
    if batch:
        chunks = batches(t0, t1, batch)
        for i, ch in enumerate(chunks):
            if before_call:
                before_call(i, len(chunks), ch)
            eng.run_ticks([tick_args(t) for t in ch])
        return len(chunks)
    for i, t in enumerate(range(t0, t1)):
        if before_call:
            before_call(i, t1 - t0, [t])
        eng.tick(**tick_args(t))
    return t1 - t0


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 4: RSPaxos, 16 384 groups x 5 replicas, 4 KiB values, RS(3,2) -- the one-launch steady tick with the
# one-pass from_data + encode + shard fan-out in front of it (bench.py's `rspaxos` leg)
# ---------------------------------------------------------------------------------------------------------------------
CONFIG4 = dict(G=16384, R=5, W=64, L=4113, H=4, ft=1, n_buffers=4, loss_p=0.3)   # L = bincode(ReqBatch of one 4 KiB Put)


def config4_cluster(G=CONFIG4["G"], W=CONFIG4["W"], ft=CONFIG4["ft"], leader=0, one_launch=True):
    """five RSPaxos replica engines (f = ft) with `leader` preset, and the steady loop over them"""
    from . import rsp_cluster
    from .rspaxos import RSPaxosReplicaGroup
    R = CONFIG4["R"]
    reps = [RSPaxosReplicaGroup(G, R, me=r, window=W, fault_tolerance=ft) for r in range(R)]
    for r in reps:
        r.preset_leader(leader)
    return reps, rsp_cluster.SteadyLoop(reps, leader=leader, one_launch=one_launch)


def config4_loss(rng, G, p=CONFIG4["loss_p"], leader=0):
    """<= 1 of a slot's 4 AcceptReplies lost (the threshold is 4 of 5 at f = 1 and nothing is retransmitted): a fraction p
    of the groups lose the reply of ONE follower.  (kind, from, to) -> uint8 [G], the keys `SteadyLoop.tick(lost=)` takes"""
    R = CONFIG4["R"]
    who = rng.integers(1, R, G)
    hit = rng.random(G) < p
    return {("accept_reply", q, leader): (hit & (who == q)).astype(np.uint8) for q in range(R) if q != leader}


def config4_tokens(G, j):
    """the batch tokens of tick j (one client batch per group), int32 [G]"""
    return ((1 + np.arange(G, dtype=np.int64) + j * G) & 0x3FFFFFFF).astype(np.int32)


def config4_tick(loop, k, src, val, lost, heartbeat):
    """one tick of the leg: the encode pass out of source buffer k -- from_data + RS(3,2) with every shard written once,
    straight into its holder's store (slot k of the rotating stores) -- then the tick's handlers in one launch.
    Returns (the leader's committed flags, the tick's codewords as a view of the stores)."""
    cw = loop.encode_stores(src, slot=k)
    return loop.tick(val, lost=lost, heartbeat=heartbeat), cw


# ---------------------------------------------------------------------------------------------------------------------
# config 4 with the shard bytes in the product's payload store (bench.py's `rspaxos_payload` leg; held against the oracles
# at 16 384 groups x L = 4113 by tests/test_baseline_configs_gpu.py::test_config3_payload_store_16384_groups)
# ---------------------------------------------------------------------------------------------------------------------
PAYLOAD_W = 16                                    # the stores' ring (two planes x W x 5 shards x G x 1376 B per replica: 3.6 GB at 16 384 groups)


def config4_payload_cluster(G=CONFIG4["G"], W=PAYLOAD_W, ft=CONFIG4["ft"], L=CONFIG4["L"], leader=0):
    """config 4's engines + one-launch steady loop, and one payload store per replica"""
    from .rsp_payload import RSPaxosPayloadStore
    reps, loop = config4_cluster(G, W, ft, leader=leader, one_launch=True)
    return reps, loop, [RSPaxosPayloadStore(G, CONFIG4["R"], W, max_data_len=L) for _ in range(CONFIG4["R"])]


def config4_payload_tick(reps, loop, stores, slot, src, val, lost, heartbeat, ones, engine=True, bytes_=True, one_call=True):
    """one tick: the engines' handlers in one launch, then the tick's bytes -- the leader's put of the serialized batches `src`
    (uint8 [G, L]) into the rows of `slot` (int32 [G]: in the steady state every group appends every tick, slot = tick) and one
    follow per replica, the followers' out of the leader's REQS plane.  Returns the leader's committed flags (None without engine)."""
    from .rsp_payload import REQS
    s = loop.s
    committed = loop.tick(val, lost=lost, heartbeat=heartbeat) if engine else None
    if bytes_:
        others = [q for q in range(len(reps)) if q != s]                  # the followers consumed ONE Accept broadcast: one call for all of them
        if one_call:                                                      # round 6: the three calls below as one, four launches instead of five
            stores[s].put_follow_all(reps[s], dict(a_n=ones, a_slot=slot, a_val=val), src, [stores[q] for q in others], [reps[q] for q in others])
        else:
            stores[s].put(dict(a_n=ones, a_slot=slot, a_val=val), src)
            stores[s].follow(reps[s])
            stores[s].follow_many([stores[q] for q in others], [reps[q] for q in others], (stores[s], REQS))
    return committed


# ---------------------------------------------------------------------------------------------------------------------
# CRaft with its shard bytes in the payload store, config 4's shape on the Raft fork (bench.py's `craft_payload` leg; held
# against the CRaft oracles and the oracle's encoder at 16 384 groups x L = 4113 by
# tests/test_baseline_configs_gpu.py::test_craft_payload_store_16384_groups)
# ---------------------------------------------------------------------------------------------------------------------
CRAFT_PAYLOAD = dict(G=16384, R=5, W=32, L=4113, ft=1)


def craft_payload_cluster(G=CRAFT_PAYLOAD["G"], W=CRAFT_PAYLOAD["W"], L=CRAFT_PAYLOAD["L"], ft=CRAFT_PAYLOAD["ft"], device=None):
    """replica 0 leads term 1, replicas 1 .. 4 follow; one CRaft payload store per replica; the tick's reusable device buffers"""
    import torch
    from .raft import CRaftLeaderGroup
    from .rsp_payload import CRaftPayloadStore
    R = CRAFT_PAYLOAD["R"]
    reps = [CRaftLeaderGroup(G, R, leader_id=r, window=W, term=1, fault_tolerance=ft) for r in range(R)]
    for r in range(1, R):
        reps[r].preset(0, 0, 1)
    stores = [CRaftPayloadStore(G, R, W, max_data_len=L) for _ in range(R)]
    _, send = reps[0].assignment(device)                                  # balanced assignment: every follower is sent its own shard
    z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
    bufs = dict(ones=torch.ones(G, dtype=torch.int32, device=device),
                em=[send[q].to(torch.uint8).reshape(1, G).contiguous() for q in range(R)],
                rt=z((R, G), torch.int64), es=z((R, G), torch.int32), fl=z((R, G), torch.uint8), ct=z((R, G), torch.int64), cs=z((R, G), torch.int32),
                first=z((R, G), torch.int32), msg=[None] * R)            # every message and reply lives in buffers of the loop's own:
    return reps, stores, bufs                                            # a tick is its handlers' launches and nothing else


def craft_payload_tick(reps, stores, bufs, slot, src, lens=None, bytes_=True, one_launch=True, one_call=True, one_tick_launch=None,
                       replies_first=False):
    """one tick: the leader appends one batch per group (slot [G] int32 = where: in the steady state log_len before the call = 1 +
    the tick's number) and `put`s the serialized batches `src` (uint8 [G, L]); its follow; per follower the AppendEntries out of the
    leader's log + `handle_msg_append_entries` -- all four in ONE launch (`smr_raft_cluster_replicate`; one_launch=False: the eight
    calls it stands for); ONE follow_many for the four followers; the replies' match-index quorum at the leader.  Returns the
    AppendEntries messages (for a checker).
    one_tick_launch (round 6, with one_launch and one_call): the engines' three launches -- append, replicate, replies -- as ONE
    (`smr_raft_cluster_tick`), the tick's bytes behind it; replies_first: the separate calls in that order (its reference)."""
    from .rsp_payload import CRaftPayloadStore
    R = len(reps)
    if one_tick_launch is None:                                          # (SMR_RAFT_CLUSTER_TICK=0: the three launches, for A/B runs)
        one_tick_launch = os.environ.get("SMR_RAFT_CLUSTER_TICK", "1") != "0"
    if one_tick_launch and one_launch and one_call:
        qs = list(range(1, R))
        for q in qs:
            if bufs["msg"][q] is None:
                bufs["msg"][q] = reps[0].new_message(1, bufs["first"].device)
        rep_rows = [dict(flags=bufs["fl"][q], term=bufs["rt"][q], end_slot=bufs["es"][q], conflict_term=bufs["ct"][q], conflict_slot=bufs["cs"][q])
                    for q in qs]
        reps[0].cluster_tick(bufs["ones"], bufs["first"], [reps[q] for q in qs], [bufs["msg"][q] for q in qs], rep_rows, bufs["rt"], bufs["es"],
                             bufs["fl"], entry_masks=[bufs["em"][q] for q in qs])
        if bytes_:
            stores[0].put_follow_all(reps[0], slot, src, stores[1:], reps[1:], lens=lens)
        return {q: bufs["msg"][q] for q in qs}
    first = reps[0].handle_req_batch_emit(bufs["ones"], out=bufs["first"])
    if bytes_ and not one_call:
        stores[0].put(reps[0], slot, src, lens)
        stores[0].follow(reps[0])
    msgs = {}
    reply = lambda q: dict(flags=bufs["fl"][q], term=bufs["rt"][q], end_slot=bufs["es"][q], conflict_term=bufs["ct"][q],
                           conflict_slot=bufs["cs"][q])                           # the reply straight into the leader's [R, G] arrays
    if one_launch:
        qs = list(range(1, R))
        for q in qs:
            if bufs["msg"][q] is None:
                bufs["msg"][q] = reps[0].new_message(1, first.device)
        reps[0].replicate_many([reps[q] for q in qs], [first[q] for q in qs], [bufs["msg"][q] for q in qs], [reply(q) for q in qs],
                               entry_masks=[bufs["em"][q] for q in qs])
        msgs = {q: bufs["msg"][q] for q in qs}
    for q in range(1, R) if not one_launch else ():
        m = bufs["msg"][q] = reps[0].gather_entries(first[q], 1, out=bufs["msg"][q])
        reps[q].handle_msg_append_entries(**m, entry_mask=bufs["em"][q], out=reply(q))
        msgs[q] = m
    if replies_first:
        reps[0].handle_msg_append_entries_reply(bufs["rt"], bufs["es"], bufs["fl"])
    if bytes_ and one_call:            # round 6: put + the leader's follow + the followers' follow_many as ONE call, four launches --
        stores[0].put_follow_all(reps[0], slot, src, stores[1:], reps[1:], lens=lens)   # behind the followers' handlers (their masks)
    elif bytes_:
        CRaftPayloadStore.follow_many(stores[1:], reps[1:], source=stores[0])
    if not replies_first:
        reps[0].handle_msg_append_entries_reply(bufs["rt"], bufs["es"], bufs["fl"])
    return msgs


# ---- serialized request batches behind a token (the payload stores' device tests and their emulator runs) --------------------
def payload_batch_len(tok, L):
    """length of the batch a token stands for: a function of the token alone, in 1 .. L"""
    return (1 + (tok.astype(np.uint64) * np.uint64(7919)) % np.uint64(L)).astype(np.uint32)


def payload_batch_bytes(tok, L):
    """the serialized request batch behind a token: a function of the token alone; [n, L] (bytes past the length are junk the
    store must never read into a shard)"""
    t = tok.astype(np.uint64)[:, None]
    i = np.arange(L, dtype=np.uint64)[None, :]
    return (((t * np.uint64(2654435761) + i * np.uint64(40503)) >> np.uint64(7)) & np.uint64(0xFF)).astype(np.uint8)
