"""One parity test per BASELINE.json line, at that line's size, named after it.

Groups are independent and every synthetic stream is keyed by the GLOBAL group id, so the engine runs the whole
population on the device while the CPU oracle runs SLICES of it -- started at the slice's group offset, fed the slice
of the same inputs -- and the engine's state of those groups must be the oracle's, bit for bit.  The slices are drawn
from a seeded generator (64-aligned: a dump of a slice is one contiguous piece of the wave-tiled arrays) and always
include the first and the last tile.  The small-shape scenario tests (tests/test_{mp,raft,ep}_gpu.py,
test_zz_rsp_gpu.py) compare every group after every tick; these prove the same at the sizes the numbers are quoted on.

Every body is a helper taking its sizes, so tests/test_hostsim.py reruns them small on the emulator build."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1200)]


def _slices(G, width, n, seed):
    """n 64-aligned slices of `width` groups: the first tile, the last tile, the rest seeded"""
    width = min(width, G)
    last = (G - width) // 64 * 64
    rng = np.random.default_rng(seed)
    starts = {0, last}
    while len(starts) < min(n, last // 64 + 1):
        starts.add(int(rng.integers(0, last // 64 + 1)) * 64)
    return [(s, min(width, G - s)) for s in sorted(starts)]


def _to_dev(t, cuda):
    import torch
    return {k: (torch.from_numpy(v).to(cuda) if isinstance(v, np.ndarray) else v) for k, v in t.items()}


# ---------------------------------------------------------------------------------------------------------------
# configs[1]: "MultiPaxos, 4 096 groups x 5 replicas, 1xMI355X, bit-exact vs CPU"  -- every group, every 8th tick
# ---------------------------------------------------------------------------------------------------------------
def test_config1_multipaxos_4096_groups_bit_exact(cuda, oracle):
    import test_mp_gpu as t
    t._run(cuda, oracle, G=4096, R=5, S=1, W=64, n_ticks=128, drop_p=0.1, timeout_frac=0.01, hb_every=4, preset=True, every=8)


# ---------------------------------------------------------------------------------------------------------------
# the headline metric's configuration: "65 536 groups, 5 replicas" at bench.py's shape (S = 32, W = 512, heartbeat
# every 4th tick, 10 % ack loss capped at 2 per slot, 1 % of the groups change leader)
# ---------------------------------------------------------------------------------------------------------------
def run_multipaxos_slices(cuda, oracle, G, S, W, n_ticks, frac, span, width, n_slices, every=4, straggler_ticks=0, batch=0):
    """cluster, stream and launch mode from summerset_amd/workloads.py -- the helpers bench.py's headline run calls.
    batch = 0: one smr_mp_tick per tick; batch > 0: smr_mp_run_ticks over chunks of `batch` ticks, which with
    straggler_ticks > 0 is the TIMED shape (bulk kernels tick by tick + mp_mark_batch + mp_straggler_batch on the side
    stream); the state is then compared at chunk ends (`every` counts chunks).  Returns (groups whose leader changed in
    the slices, commits, the longest straggler list any mark pass wanted, the list's capacity)."""
    from oracle.oracle import MP_SCALARS, MP_SLOTS
    from summerset_amd import workloads
    R = 5
    eng = workloads.headline_cluster(G, W=W, R=R, straggler_ticks=straggler_ticks)
    st = workloads.headline_stream(G, n_ticks, frac, span, S=S, W=W, R=R)
    sl = _slices(G, width, n_slices, seed=G + S)
    orcs, sts, pools = [], [], []
    for g0, n in sl:
        o = oracle.MpOracle(n, R, W, win_reserve=W // 8, cap=W + 4, record_commits=False)
        o.preset_leader(0)
        orcs.append(o)
        sts.append(workloads.headline_stream(n, n_ticks, frac, span, S=S, W=W, R=R, group_base=g0))
        pools.append([sts[-1].tick(t) for t in range(4)])
    pool = [_to_dev({k: v for k, v in st.tick(t).items() if k in ("req_cnt", "req_val", "ackctl")}, cuda) for t in range(4)]
    # the slice streams ARE the global stream (keyed by global group id): checked on the inputs themselves
    for (g0, n), p in zip(sl, pools):
        full = st.tick(1)
        assert np.array_equal(full["ackctl"][:, g0:g0 + n], p[1]["ackctl"]) and np.array_equal(full["req_val"][:, g0:g0 + n], p[1]["req_val"])
    events = [_to_dev(st.tick_events(t), cuda) for t in range(n_ticks)]
    fired = [bool((st.tick_events(t)["timeout_rep"] != 0xFF).any()) for t in range(n_ticks)]

    def tick_args(t):                                    # bench.py's tick_args: no timeout arrays on a tick without a timer
        e = events[t]
        return dict(timeout_rep=e["timeout_rep"] if fired[t] else None, timeout_src=e["timeout_src"] if fired[t] else None,
                    req_target=e["req_target"], heartbeat=st.heartbeat(t), **pool[t % 4])

    def compare(t):
        for (g0, n), o in zip(sl, orcs):
            for r in range(R):
                a, b = eng.dump(r, g0, n), o.dump(r)
                assert np.array_equal(a["overflow"], b["overflow"]), (t, g0, r)
                live = b["overflow"] == 0
                for name in MP_SCALARS:
                    assert np.array_equal(a[name][live], b[name][live]), "tick %d slice %d rep %d %s: groups %s" % (
                        t, g0, r, name, g0 + np.nonzero((a[name] != b[name]) & live)[0][:5])
                assert np.array_equal(a["peer_exec_bar"][:, live], b["peer_exec_bar"][:, live]), (t, g0, r)
                for name, _ in MP_SLOTS:
                    assert np.array_equal(a[name][:, live], b[name][:, live]), (t, g0, r, name)

    chunks = workloads.batches(0, n_ticks, batch) if batch else [[t] for t in range(n_ticks)]
    want_max, cap = 0, 0
    for i, ch in enumerate(chunks):
        workloads.drive_headline(eng, tick_args, ch[0], ch[-1] + 1, batch=batch)
        if straggler_ticks:
            cap, want = eng.straggler_stats()
            want_max = max(want_max, want)
        for t in ch:
            for o, s_, p in zip(orcs, sts, pools):
                inp = dict(p[t % 4])
                inp.update(s_.tick_events(t))
                inp["heartbeat"] = s_.heartbeat(t)
                o.tick(**inp)
        if i % every == every - 1 or i == len(chunks) - 1:
            compare(ch[-1])
    changed = 0
    for (g0, n), o in zip(sl, orcs):
        changed += int((o.dump(1)["leader"] != 0).sum())
        assert int(o.dump(0)["commit_bar"].min()) > 0
    total = sum(eng.counters(r)["commits"] for r in range(R))
    assert total > 0
    return changed, total, want_max, cap


def test_headline_multipaxos_65536_groups_s32(cuda, oracle):
    """8 slices x 512 groups of the 65 536: full state of all five replicas (every slot of the 512-slot rings) against
    the oracle every 4th tick, 24 ticks with the leader changes of 1 % of the groups inside them"""
    changed, total, _, _ = run_multipaxos_slices(cuda, oracle, G=65536, S=32, W=512, n_ticks=24, frac=0.01, span=12, width=512, n_slices=8)
    assert total > 65536 * 32 * 16                       # >= 16 of the 24 ticks' slots committed (loss is quorum-preserving)


def test_headline_multipaxos_65536_groups_s32_many_leader_changes(cuda, oracle):
    """the same population with a quarter of the groups changing leader inside 10 ticks: the cooperative rare path at
    full occupancy (long re-Accept outboxes, Prepare batches) next to the bulk path"""
    changed, _, _, _ = run_multipaxos_slices(cuda, oracle, G=65536, S=32, W=512, n_ticks=20, frac=0.25, span=10, width=256, n_slices=6)
    assert changed > 100


# --- the launch shape bench.py TIMES: smr_mp_run_ticks in batches of 8, straggler list on (ttl 4) ----
def test_headline_multipaxos_65536_groups_s32_bench_launch(cuda, oracle):
    """bench.py's default headline run as it is launched -- workloads.headline_cluster(straggler_ticks=4) driven by
    workloads.drive_headline(batch=8): per batch one mp_mark_batch, one mp_straggler_batch on the side stream and the bulk
    round kernels tick by tick -- with 1 % of the 65 536 groups changing leader inside 24 ticks (~27 per tick, the driver
    command's rate).  Full state of all five replicas of 8 x 512 groups against the oracle after every batch; the
    straggler list (1024 groups) must never have been full."""
    changed, total, want, cap = run_multipaxos_slices(cuda, oracle, G=65536, S=32, W=512, n_ticks=24, frac=0.01, span=24, width=512,
                                                      n_slices=8, every=1, straggler_ticks=4, batch=8)
    assert total > 65536 * 32 * 16
    assert cap == 1024 and 0 < want <= cap, "straggler list: %d groups wanted, capacity %d" % (want, cap)


def test_headline_multipaxos_65536_groups_s32_bench_launch_many_leader_changes(cuda, oracle):
    """the same launch shape with a quarter of the groups changing leader inside 16 ticks (~1000 per tick): every batch's
    mark pass wants far more groups than the list holds, so the 192 x 6 side blocks run full and the overflow groups'
    leader changes go through the bulk kernels' cooperative jobs -- both must be the oracle's, bit for bit"""
    changed, total, want, cap = run_multipaxos_slices(cuda, oracle, G=65536, S=32, W=512, n_ticks=24, frac=0.25, span=16, width=256,
                                                      n_slices=6, every=1, straggler_ticks=4, batch=8)
    assert changed > 100
    assert want > cap == 1024, "this workload is meant to overflow the list (%d wanted)" % want


# ---------------------------------------------------------------------------------------------------------------
# configs[2]: "Raft, 65 536 groups x 5 replicas, AppendEntries ack-matrix quorum kernel" -- every group, every tick
# ---------------------------------------------------------------------------------------------------------------
def test_config2_raft_65536_groups(cuda, oracle):
    import test_raft_gpu as t
    t._run(cuda, oracle, G=65536, R=5, W=64, T=12)


# ---------------------------------------------------------------------------------------------------------------
# configs[3]: "RSPaxos, 16 384 groups x 5 replicas, 4 KiB values, RS(3,2) GF(2^8) encode"
# ---------------------------------------------------------------------------------------------------------------
def _cut(d, G, g0, n):
    """the groups [g0, g0 + n) of every per-group array of a dump (the one axis of length G; counters have none)"""
    out = {}
    for k, v in d.items():
        ax = [i for i, m in enumerate(getattr(v, "shape", ())) if m == G]
        if len(ax) == 1:
            out[k] = np.take(v, np.arange(g0, g0 + n), axis=ax[0])
    return out


def run_rspaxos_slices(cuda, oracle, G, W, T, ft, loss, width, n_slices):
    """five RSPaxos replica engines of G groups in the closed loop of summerset_amd/rsp_cluster.py (appends, loss, two
    leader changes with shard merging / re-Accepts / reconstruction reads, heartbeats); oracle clusters run slices of
    the same scenario (tests/rsp_scenarios.run(view=...)).  The slice's clusters make fewer handler calls than the
    whole population's (a call happens when ANY group has the message), so this also checks that a handler call is a
    no-op for the groups whose flag is clear.  Compared: every replica's full state in the middle and at the end of
    the run, and what every replica executed, in order."""
    import rsp_cluster as rc
    import rsp_scenarios as sc
    from summerset_amd import RSPaxosReplicaGroup
    R = 5
    sl = _slices(G, width, n_slices, seed=G + ft)
    at = (T // 2, T - 1)
    engs = [rc.NumpyEngine(RSPaxosReplicaGroup(G, R, me=r, window=W, fault_tolerance=ft), cuda) for r in range(R)]
    snaps, execd = {}, [[] for _ in range(R)]

    def on_eng(t):
        for r in range(R):
            execd[r].append(engs[r].take_executed())
        if t in at:
            full = [e.dump() for e in engs]
            snaps[t] = [[_cut(full[r], G, g0, n) for r in range(R)] for g0, n in sl]
    sc.run(engs, G, T, seed=G + ft, loss=loss, on_tick=on_eng)
    counters = np.zeros(3, np.int64)
    for i, (g0, n) in enumerate(sl):
        orcs = [oracle.RspOracle(n, R, me=r, W=W, fault_tolerance=ft) for r in range(R)]
        osn, oex = {}, [[] for _ in range(R)]

        def on_orc(t):
            for r in range(R):
                oex[r].append(orcs[r].take_executed())
            if t in at:
                osn[t] = [o.dump() for o in orcs]
        sc.run(orcs, G, T, seed=G + ft, loss=loss, view=(g0, n), on_tick=on_orc)
        for t in at:
            for r in range(R):
                a, b = snaps[t][i][r], _cut(osn[t][r], n, 0, n)
                assert set(a) == set(b) and len(b) > 8
                for name in b:
                    assert np.array_equal(a[name], b[name]), (t, g0, r, name, [x[:4] for x in np.nonzero(a[name] != b[name])])
        for r in range(R):
            for t in range(T):
                eg, es, ev = execd[r][t]
                m = (eg >= g0) & (eg < g0 + n)
                og, os_, ov = oex[r][t]
                assert np.array_equal(eg[m] - g0, og) and np.array_equal(es[m], os_) and np.array_equal(ev[m], ov), (t, g0, r, "executed")
        counters += np.asarray(orcs[0].dump()["counters"][:3], np.int64)
    assert counters[0] > 0 and counters[1] > 0


def run_rspaxos_one_launch(cuda, oracle, G, W, L, T, ft):
    """bench.py's config-4 leg as it is launched (summerset_amd/workloads.py: config4_cluster / config4_loss / config4_tokens
    / config4_tick): per tick ONE pass `rs_from_data_xtime` (from_data + RS(3,2) encode, every shard written once into its holder's store) and ONE
    launch `smr_rsp_cluster_steady_tick` -- against five ORACLES of the WHOLE population in the numpy-staged closed loop
    (rsp_cluster.tick) and the oracle's RS encoder on every codeword of every tick.  Compared: the leader's committed flags
    every tick, the codeword bytes and every replica's shard store byte for byte every tick, every replica's full state
    at the end."""
    import torch
    from summerset_amd import RSCodewordBatch, rsp_cluster as rc, workloads
    c4 = workloads.CONFIG4
    R, NB, H = c4["R"], c4["n_buffers"], c4["H"]
    reps, loop = workloads.config4_cluster(G, W, ft, one_launch=True)
    orcs = [oracle.RspOracle(G, R, me=r, W=W, fault_tolerance=ft) for r in range(R)]
    for o in orcs:
        o.preset_leader(0)
    rng = np.random.default_rng(0x5EED5EED)
    data = [rng.integers(0, 256, (G, L), dtype=np.uint8) for _ in range(NB)]
    srcs = [torch.from_numpy(d).to(cuda) for d in data]
    sl_ = -(-L // 3)
    want_par = [oracle.rs_encode_batch(3, 2, d, L, L, G).reshape(G, 2, sl_) for d in data]   # the oracle's encoder, every codeword
    padded = []
    for d in data:                                       # from_data geometry (rscoding.rs:165-220): zero-padded to 3 shard_len, split
        x = np.zeros((G, 3 * sl_), np.uint8)
        x[:, :L] = d
        padded.append(x.reshape(G, 3, sl_))
    masks = [workloads.config4_loss(rng, G) for _ in range(NB)]
    dmasks = [{k: torch.from_numpy(v).to(cuda) for k, v in m.items()} for m in masks]
    total = 0
    for t in range(T):
        k, hb = t % NB, t % H == H - 1
        val = workloads.config4_tokens(G, t)
        got, cw = workloads.config4_tick(loop, k, srcs[k], torch.from_numpy(val).to(cuda), dmasks[k], hb)
        got = got.cpu().numpy()
        log = rc.tick(orcs, val.view(np.uint32), np.zeros(G, np.uint8), drop={k_: v.astype(bool) for k_, v in masks[k].items()}, heartbeat=hb)
        want = [e for e in log if e["kind"] == "commit"]
        assert len(want) == 1 and np.array_equal(got, want[0]["committed"]), t
        total += int(got.sum())
        stores = loop.stores.cpu().numpy()               # [R, G, shard_len]: what replica q holds of this tick's codewords
        assert cw.buf is None and cw.stores.data_ptr() == loop.stores.data_ptr() and cw.shard_len == sl_   # the leader's codeword IS the stores
        for q in range(R):
            exp = padded[k][:, q] if q < 3 else want_par[k][:, q - 3]
            assert np.array_equal(stores[q], exp), (t, q, "shard store")
        if t < NB:                                       # the shard-major batch as a codeword: parity verifies, the data is the batch
            assert bool(cw.verify_parity().all())
            assert np.array_equal(cw.get_data().cpu().numpy(), data[k])
    for r in range(R):
        a, b = reps[r].dump(), orcs[r].dump()
        assert len(b) > 8
        for n in b:
            assert np.array_equal(a[n], b[n]), (r, n)
    return total


def test_config3_rspaxos_one_launch_tick_16384_groups(cuda, oracle):
    """the launches bench.py TIMES for config 4 -- `smr_rsp_cluster_steady_tick` + `rs_from_data_xtime`
    with the fan-out -- at 16 384 groups x L = 4113, every group and every byte against the oracles, 12 ticks (three
    rotations of the four buffer pairs, three heartbeat ticks), ~30 % of the slots losing one of their four replies"""
    total = run_rspaxos_one_launch(cuda, oracle, G=16384, W=64, L=4113, T=12, ft=1)
    assert total > 16384 * 10


def run_rspaxos_payload(cuda, oracle, G, W, L, T, ft):
    """bench.py's `rspaxos_payload` leg as it is launched (summerset_amd/workloads.py: config4_payload_cluster / config4_payload_tick):
    per tick the engines' one-launch tick, the leader's `smr_rsp_pstore_put` and one `smr_rsp_pstore_follow` per replica --
    against five ORACLES of the whole population (committed flags every tick, every replica's state at the end) and the
    oracle's RS encoder on every codeword of every tick: the leader's row (all five shards), every follower's shard in both
    planes, tokens / masks / lengths of the tick's cells, and the batch read back through `get_data`, byte for byte.  T > W:
    the ring wraps and every row is re-keyed."""
    import torch
    from summerset_amd import rsp_cluster as rc, workloads
    from summerset_amd.rsp_payload import REQS, VOTED
    c4 = workloads.CONFIG4
    R, NB, H = c4["R"], c4["n_buffers"], c4["H"]
    reps, loop, stores = workloads.config4_payload_cluster(G, W, ft, L)
    orcs = [oracle.RspOracle(G, R, me=r, W=W, fault_tolerance=ft) for r in range(R)]
    for o in orcs:
        o.preset_leader(0)
    rng = np.random.default_rng(0x5EED5EED)
    data = [rng.integers(0, 256, (G, L), dtype=np.uint8) for _ in range(NB)]
    srcs = [torch.from_numpy(d).to(cuda) for d in data]
    sl_ = -(-L // 3)
    want = []
    for d in data:                                       # the oracle's codeword of every batch: from_data geometry + compute_parity
        x = np.zeros((G, 3 * sl_), np.uint8)
        x[:, :L] = d
        want.append(np.concatenate([x.reshape(G, 3, sl_), oracle.rs_encode_batch(3, 2, d, L, L, G).reshape(G, 2, sl_)], axis=1))   # [G, 5, sl]
    masks = [workloads.config4_loss(rng, G) for _ in range(NB)]
    dmasks = [{k: torch.from_numpy(v).to(cuda) for k, v in m.items()} for m in masks]
    ones = torch.ones(G, dtype=torch.int32, device=cuda)
    shard = [torch.full((G,), 1 << q, dtype=torch.uint8, device=cuda) for q in range(R)]
    every = torch.full((G,), (1 << R) - 1, dtype=torch.uint8, device=cuda)
    total = 0
    for t in range(T):
        k, hb = t % NB, t % H == H - 1
        val = workloads.config4_tokens(G, t)
        slot = torch.full((G,), t, dtype=torch.int32, device=cuda)
        got = workloads.config4_payload_tick(reps, loop, stores, slot, srcs[k], torch.from_numpy(val).to(cuda), dmasks[k], hb, ones).cpu().numpy()
        log = rc.tick(orcs, val.view(np.uint32), np.zeros(G, np.uint8), drop={k_: v.astype(bool) for k_, v in masks[k].items()}, heartbeat=hb)
        cm = [e for e in log if e["kind"] == "commit"]
        assert len(cm) == 1 and np.array_equal(got, cm[0]["committed"]), t
        total += int(got.sum())
        # the tick's cells through `extract` (the header a message would carry + the bytes)
        for q in range(R):
            for plane in (REQS, VOTED):
                m = stores[q].extract(slot, every, plane)
                held = (1 << R) - 1 if (q == 0 and plane == REQS) else 1 << q        # the leader's codeword; a vote / a follower: its own shard
                assert (m["mask"].cpu().numpy() == held).all() and (m["dlen"].cpu().numpy() == L).all(), (t, q, plane)
                assert np.array_equal(m["tok"].cpu().numpy(), val), (t, q, plane)
                for s_ in range(R):
                    if (held >> s_) & 1:
                        assert np.array_equal(m["buf"][s_, :, :sl_].cpu().numpy(), want[k][:, s_]), (t, q, plane, s_)
        out, ln, ok = stores[0].get_data(slot, expect=torch.from_numpy(val).to(cuda))
        assert bool(ok.all()) and (ln.cpu().numpy() == L).all() and np.array_equal(out[:, :L].cpu().numpy(), data[k]), t
    for r in range(R):
        a, b = reps[r].dump(), orcs[r].dump()
        for n in b:
            assert np.array_equal(a[n], b[n]), (r, n)
        c = stores[r].counters()
        assert c["unsatisfied"] == 0 and c["rebuilt"] == 0 and c["copied"] == (1 if r == 0 else 2) * G * T, (r, c)
        # once the ring wrapped: both planes of a follower's row; the leader's put re-keys its REQS row itself, and the vote that was an
        # alias of that row's shard goes with it
        assert c["rekeyed"] == (0 if r == 0 else 2) * G * max(T - W, 0), (r, c)
        from summerset_amd.rsp_payload import VOTED
        assert np.array_equal(stores[r].voted_alias(), stores[r].dump(VOTED)["avail"]), r      # no vote was stored a second time
    return total


def test_config3_payload_store_16384_groups(cuda, oracle):
    """the launches bench.py's `rspaxos_payload` leg TIMES, at its size (16 384 groups x L = 4113, window 16), 20 ticks"""
    total = run_rspaxos_payload(cuda, oracle, G=16384, W=16, L=4113, T=20, ft=1)
    assert total > 16384 * 17


def run_craft_payload(cuda, oracle, G, W, L, T, width=256, n_slices=4):
    """the launches bench.py's `craft_payload` leg TIMES (summerset_amd/workloads.py: craft_payload_cluster / craft_payload_tick), T ticks
    with ragged batch lengths.  Engines: every follower's replies and every replica's final log, bitmaps and counters against CRaft
    oracles on 64-aligned slices fed the same AppendEntries.  Bytes: after the run EVERY cell of EVERY replica's store against its
    engine's avail_shards_map, and the rows of the last ticks -- all five shards at the leader, one's own at every follower, all
    groups -- against the oracle's encoder; the last entries read back (`get_data`) as the batches that were appended."""
    import torch
    from summerset_amd import workloads
    R, d = 5, 3
    reps, stores, bufs = workloads.craft_payload_cluster(G, W, L, device=cuda)
    sl = _slices(G, width, n_slices, seed=G + L)
    orcs = [[oracle.CRaftOracle(n, R, W, leader_id=r, term=1, fault_tolerance=1) for r in range(R)] for _, n in sl]
    for oc in orcs:
        for r in range(1, R):
            oc[r].preset(0, 0, 1)
    rng = np.random.default_rng(G * 3 + L)
    u = lambda t, dt: t.cpu().numpy().view(dt)
    keep = {}                                                              # tick -> (data, lens) of the last ticks
    for j in range(T):
        data = rng.integers(0, 256, (G, L), dtype=np.uint8)
        lens = rng.integers(1, L + 1, G).astype(np.uint32)
        lens[rng.random(G) < 0.3] = L
        keep = {k: v for k, v in keep.items() if k >= j - 2}
        keep[j] = (data, lens)
        slot = torch.full((G,), 1 + j, dtype=torch.int32, device=cuda)
        msgs = workloads.craft_payload_tick(reps, stores, bufs, slot, torch.from_numpy(data).to(cuda), torch.from_numpy(lens.view(np.int32)).to(cuda))
        for (g0, n), oc in zip(sl, orcs):
            oc[0].append(np.ones(n, np.uint32))
            rt, es, fl = np.zeros((R, n), np.uint64), np.zeros((R, n), np.uint32), np.zeros((R, n), np.uint8)
            for q in range(1, R):
                m = msgs[q]
                cut = lambda a, dt: np.ascontiguousarray(u(a, dt)[..., g0:g0 + n])
                r = oc[q].handle_append_entries(cut(m["flags"], np.uint8), cut(m["leader"], np.uint8), cut(m["term"], np.uint64), cut(m["prev_slot"], np.uint32),
                                                cut(m["prev_term"], np.uint64), cut(m["n_entries"], np.uint32), cut(m["entry_term"], np.uint64),
                                                cut(m["leader_commit"], np.uint32), cut(m["last_snap"], np.uint32), entry_mask=cut(bufs["em"][q], np.uint8))
                assert np.array_equal(r["term"], u(bufs["rt"][q], np.uint64)[g0:g0 + n]) and np.array_equal(r["end_slot"], u(bufs["es"][q], np.uint32)[g0:g0 + n]) \
                    and np.array_equal(r["flags"], u(bufs["fl"][q], np.uint8)[g0:g0 + n]), (j, g0, q)
                rt[q], es[q], fl[q] = r["term"], r["end_slot"], r["flags"]
            oc[0].handle_replies(rt, es, fl, None, None, None)
    dumps = [(e.dump(), e.dump_masks()) for e in reps]
    for (g0, n), oc in zip(sl, orcs):
        for r in range(R):
            a, b = dumps[r][0], oc[r].dump()
            for k in b:
                assert np.array_equal(a[k][..., g0:g0 + n], b[k]), (g0, r, k)
            am, bm = dumps[r][1], oc[r].dump_masks()
            assert np.array_equal(am["mask"][:, g0:g0 + n], bm["mask"]), (g0, r, "masks")
        assert oc[0].total_commits() > 0
    # ---- the bytes: every cell of every store = its engine's bitmap ...
    shard_len = lambda ln: -(-ln // d)
    for r in range(R):
        dmp, masks, sd = dumps[r][0], dumps[r][1]["mask"], stores[r].dump()
        ln = dmp["log_len"].astype(np.int64)
        assert (ln == T + 1).all() and (dmp["start_slot"] <= 1).all() if T + 1 <= W else True
        for s in range(max(1, T + 1 - W), T + 1):
            w = s % W
            assert np.array_equal(sd["avail"][w], masks[w]), (r, s, "avail")
            assert (masks[w] == (31 if r == 0 else 1 << r)).all(), (r, s, "balanced assignment: the leader every shard, a follower its own")
        assert stores[r].counters()["unsatisfied"] == 0
    # ... and the last ticks' rows byte for byte the oracle's codewords, all groups
    total = 0
    for j, (data, lens) in sorted(keep.items()):
        s = 1 + j
        rows = [stores[r].read_row(s) for r in range(R)]                  # [R][R shards][G][group_stride]
        for ln_ in np.unique(lens):
            gs = np.nonzero(lens == ln_)[0]
            if len(gs) > 64:
                gs = gs[:: max(1, len(gs) // 64)]                        # (the full-length batches: a sample of them; ragged lengths are all distinct groups)
            sl_ = shard_len(int(ln_))
            pad = np.zeros((len(gs), d * sl_), np.uint8)
            pad[:, :ln_] = data[gs, :ln_]
            cw = np.concatenate([pad.reshape(len(gs), d, sl_), np.stack([np.asarray(oracle.rs_encode(d, R - d, data[g, :ln_]), np.uint8).reshape(R - d, sl_) for g in gs])], axis=1)
            for k in range(R):
                assert np.array_equal(rows[0][k][gs, :sl_], cw[:, k]), (j, "leader", k, int(ln_))
            for q in range(1, R):
                assert np.array_equal(rows[q][q][gs, :sl_], cw[:, q]), (j, "follower", q, int(ln_))
            total += len(gs)
        out, got_len, ok = stores[0].get_data(torch.full((G,), s, dtype=torch.int32, device=cuda))
        out, got_len, ok = out.cpu().numpy(), got_len.cpu().numpy(), ok.cpu().numpy()
        assert ok.all() and np.array_equal(got_len.view(np.uint32), lens)
        full = lens == L
        assert np.array_equal(out[full], data[full])
        g_ = int(np.nonzero(~full)[0][0])
        assert np.array_equal(out[g_, :lens[g_]], data[g_, :lens[g_]])
    return total


def test_craft_payload_store_16384_groups(cuda, oracle):
    """the launches bench.py's `craft_payload` leg TIMES, at its size (16 384 groups x L = 4113, window 32), 10 ticks"""
    assert run_craft_payload(cuda, oracle, G=16384, W=32, L=4113, T=10) > 2000


def test_config3_rspaxos_16384_groups_rs32_4k_values(cuda, oracle):
    import torch
    from summerset_amd import RSCodewordBatch
    # (i) the consensus path at 16 384 groups, f = 1
    run_rspaxos_slices(cuda, oracle, G=16384, W=32, T=21, ft=1, loss=0.05, width=512, n_slices=6)
    # (ii) the tick's payload: 16 384 request batches of one 4 KiB Put each -- bincode(ReqBatch) L = 4113 (SURVEY
    # Appendix C: 4110 + len(client varint) + len(id varint), both < 251 here... client 1 B, id 2 B) -- RS(3,2),
    # EVERY codeword against the oracle, then erase two shards of every codeword and rebuild
    n, L = 16384, 4113
    rng = np.random.default_rng(0xC0F4)
    data = rng.integers(0, 256, (n, L), dtype=np.uint8)
    cw = RSCodewordBatch.from_data(torch.from_numpy(data).to(cuda), 3, 2)
    cw.compute_parity()
    sl_ = cw.shard_len
    assert sl_ == 1371
    par = cw.buf[:, 3 * sl_:5 * sl_].cpu().numpy().reshape(n, 2, sl_)
    want = oracle.rs_encode_batch(3, 2, data, L, L, n).reshape(n, 2, sl_)
    assert np.array_equal(par, want)
    assert bool(cw.verify_parity().all())
    keep = cw.buf.clone()
    for pat in ((0, 1), (2, 4), (1, 3)):
        cw.erase(pat)
        cw.reconstruct_all()
        assert torch.equal(cw.buf, keep), pat
    cw.erase((0, 2))
    cw.reconstruct_data()
    assert np.array_equal(cw.get_data().cpu().numpy(), data)


# ---------------------------------------------------------------------------------------------------------------
# configs[4]: "EPaxos, 65 536 groups x 5 replicas, dependency-graph + fast-quorum kernel": every replica proposes one
# instance per group per tick on Zipf(0.99) keys of 64, lost PreAccepts, execution on
# ---------------------------------------------------------------------------------------------------------------
def run_epaxos_slices(cuda, oracle, G, W, K, T, width, n_slices, execute=True, loss=0.1):
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup
    R = 5
    sl = _slices(G, width, n_slices, seed=G + K)
    engs = [ec.NumpyEngine(EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=execute), cuda) for r in range(R)]
    orcs = [[oracle.EpOracle(n, R, me=r, W=W, n_keys=K, execute=execute) for r in range(R)] for _, n in sl]
    rng = np.random.default_rng(G + W)
    fast = slow = 0
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < loss for s in range(R) for q in range(R) if s != q}
        oe = ec.tick(engs, keys, drop)
        for (g0, n), oc in zip(sl, orcs):
            oo = ec.tick(oc, np.ascontiguousarray(keys[:, g0:g0 + n]), {k: v[g0:g0 + n] for k, v in drop.items()})
            for s in range(R):
                for k in oo[s]:
                    assert np.array_equal(oe[s][k][..., g0:g0 + n], oo[s][k]), (t, g0, s, k)
                fast += int((oo[s]["decision"] == 3).sum())
                slow += int((oo[s]["decision"] == 2).sum())
    full = [(e.dump(), e.exec_dump() if execute else {}) for e in engs]
    for (g0, n), oc in zip(sl, orcs):
        for r in range(R):
            for a, b in ((_cut(full[r][0], G, g0, n), _cut(oc[r].dump(), n, 0, n)),
                         (_cut(full[r][1], G, g0, n), _cut(oc[r].exec_dump(), n, 0, n) if execute else {})):
                assert set(a) == set(b)
                for name in b:
                    assert np.array_equal(a[name], b[name]), (g0, r, name)
    assert fast > 0 and slow > 0
    return fast, slow


def test_config4_epaxos_65536_groups_all_replicas_propose(cuda, oracle):
    run_epaxos_slices(cuda, oracle, G=65536, W=16, K=64, T=8, width=512, n_slices=6)


def run_epaxos_cluster_slices(cuda, oracle, G, W, K, T, width, n_slices, phase_major, loss=0.1):
    """BASELINE config 5 through `smr_ep_cluster_tick` -- the whole tick ONE launch -- at full size, against five oracles per
    slice wired into tests/ep_cluster.py's loop in the same order: every leader's outputs every tick, every replica's protocol
    and execution state at the end"""
    import torch
    import ep_cluster as ec
    from summerset_amd import EPaxosReplicaGroup, ep_cluster
    R = 5
    sl = _slices(G, width, n_slices, seed=G + K + 1)
    reps = [EPaxosReplicaGroup(G, R, me=r, window=W, n_keys=K, execute=True) for r in range(R)]
    job = ep_cluster.EPaxosCluster(reps, phase_major=phase_major)
    orcs = [[oracle.EpOracle(n, R, me=r, W=W, n_keys=K, execute=True) for r in range(R)] for _, n in sl]
    rng = np.random.default_rng(G + W + 1)
    dv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)   # noqa: E731
    fast = slow = 0
    for t in range(T):
        keys = ec.zipf_keys(rng, R, G, K)
        drop = {(s, q): rng.random(G) < loss for s in range(R) for q in range(R) if s != q}
        oe = job.tick([dv(keys[r]) for r in range(R)], {k: dv(v) for k, v in drop.items()})
        oe = [{k: v.cpu().numpy() for k, v in o.items()} for o in oe]
        for (g0, n), oc in zip(sl, orcs):
            oo = ec.tick(oc, np.ascontiguousarray(keys[:, g0:g0 + n]), {k: v[g0:g0 + n] for k, v in drop.items()}, phase_major=phase_major)
            for s in range(R):
                for k in oo[s]:
                    assert np.array_equal(oe[s][k][..., g0:g0 + n].view(oo[s][k].dtype), oo[s][k]), (t, g0, s, k)
                fast += int((oo[s]["decision"] == 3).sum())
                slow += int((oo[s]["decision"] == 2).sum())
    full = [(e.dump(), e.exec_dump()) for e in reps]
    for (g0, n), oc in zip(sl, orcs):
        for r in range(R):
            for a, b in ((_cut(full[r][0], G, g0, n), _cut(oc[r].dump(), n, 0, n)), (_cut(full[r][1], G, g0, n), _cut(oc[r].exec_dump(), n, 0, n))):
                assert set(a) == set(b)
                for name in b:
                    assert np.array_equal(a[name], b[name]), (g0, r, name)
    job.close()
    assert fast > 0 and slow > 0


@pytest.mark.parametrize("phase_major", [False, True])
def test_config5_epaxos_65536_groups_one_launch(cuda, oracle, phase_major):
    """the one-launch cluster tick at BASELINE config 5's size, in the loops' order and with the leaders' steps phase by phase"""
    run_epaxos_cluster_slices(cuda, oracle, G=65536, W=16, K=64, T=8, width=512, n_slices=4, phase_major=phase_major)
