"""Spread layout (L2) of the MultiPaxos cluster engine, all ranks of the job inside one process: the image kernels
(smr_mp_image_pack / _unpack), the live masks and the exchange plan of summerset_amd/spread_mp.py -- against the
co-located engine on the same streams, every replica's full state after every tick.  The gpu-marked test runs on the
device; tests/test_hostsim.py reruns it on the emulator build; tests/test_spread_mp_gloo.py is the two-process job."""
import numpy as np
import pytest


def _to_dev(t, dev):
    import torch
    return {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in t.items()}


def run_spread_vs_colocated(dev, G, R, S, W, world, n_ticks, drop_p, timeout_frac, hb_every=3, ovf_cap=8192, make=None, compare_every=1, rounds=None):
    """`make(world)` -> object with preset_leader / tick(inputs, heartbeat) and .ranks (default: spread_mp.in_process).
    rounds: smr_mp_spread_set_concurrent's argument -- 2 the blocks' rounds in one launch (the default), 1 side by side on streams
    (round 4), 0 one after the other"""
    from oracle.oracle import MP_SCALARS, MP_SLOTS
    from summerset_amd import MultiPaxosCluster, shard, spread_mp, stream
    cap = W + 4
    ref = MultiPaxosCluster(G, R, W, outbox_cap=cap)
    ref.preset_leader(0)
    job = spread_mp.in_process(G, R, W, world, dev, S, ovf_cap=ovf_cap, outbox_cap=cap) if make is None else make(world)
    job.preset_leader(0)
    if rounds is not None:
        from summerset_amd._lib import check
        for rk in job.ranks:
            check(rk._L.smr_mp_spread_set_concurrent(rk._spread, int(rounds)))
    kw = dict(cap=cap, n_ticks=n_ticks, drop_p=drop_p, timeout_frac=timeout_frac, hb_every=hb_every)
    st = stream.MultiPaxosStream(G, R, S, **kw)
    bst = {b: stream.MultiPaxosStream(hi - lo, R, S, group_base=lo, **kw) for b, (lo, hi) in
           ((b, shard.group_range(G, world, b)) for b in range(world)) if hi > lo}
    for t in range(n_ticks):
        inp = st.tick(t)
        ref.tick(**_to_dev(inp, dev))
        job.tick({b: _to_dev({k: v for k, v in s_.tick(t).items() if k != "heartbeat"}, dev) for b, s_ in bst.items()}, heartbeat=inp["heartbeat"])
        if t % compare_every and t != n_ticks - 1:
            continue
        for rk in job.ranks:
            for b, (cl, live, lo, hi) in rk.blocks.items():
                assert lo % 64 == 0, "test shapes keep the blocks 64-aligned (smr_mp_dump_range)"
                for r in live:
                    a, x = cl.dump(r), ref.dump(r, lo, hi - lo)
                    assert not a["overflow"].any() and not x["overflow"].any()
                    for name in list(MP_SCALARS) + ["peer_exec_bar"] + [n for n, _ in MP_SLOTS]:
                        assert np.array_equal(a[name], x[name]), "tick %d rank %d block %d replica %d: %s differs" % (t, rk.rank, b, r, name)
    total = sum(rk.commits() for rk in job.ranks)
    assert total == sum(ref.counters(r)["commits"] for r in range(R)) and total > 0
    assert sum(rk.dropped_overflow_entries() for rk in job.ranks) == 0
    return job


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3, 8])
def test_spread_job_is_the_colocated_one(cuda, world):
    # every group changes leader inside the run: Prepares, PrepareReply batches, long re-Accept outboxes (entries >= 64:
    # ack byte cells), step-up heartbeats -- all through the images' overflow lists
    job = run_spread_vs_colocated(cuda, G=64 * world * 2, R=5, S=2, W=64, world=world, n_ticks=30, drop_p=0.1, timeout_frac=1.0)
    assert all(rk.bytes_sent > 0 for rk in job.ranks)


@pytest.mark.gpu
@pytest.mark.parametrize("rounds", [1, 0])
def test_spread_job_with_the_blocks_rounds_launched_one_by_one(cuda, rounds):
    """round 5 launches the rounds of a rank's blocks as ONE kernel (blockIdx.z = block); round 4's ways stay selectable"""
    run_spread_vs_colocated(cuda, G=64 * 3 * 2, R=5, S=2, W=64, world=3, n_ticks=24, drop_p=0.1, timeout_frac=1.0, rounds=rounds)


@pytest.mark.gpu
def test_spread_steady_state_and_three_replicas(cuda):
    run_spread_vs_colocated(cuda, G=512, R=5, S=4, W=64, world=4, n_ticks=24, drop_p=0.1, timeout_frac=0.0)
    run_spread_vs_colocated(cuda, G=256, R=3, S=2, W=32, world=2, n_ticks=24, drop_p=0.2, timeout_frac=0.5, hb_every=2)
