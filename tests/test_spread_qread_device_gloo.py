"""Layout L2 of the quorum reads, device-resident (summerset_amd.spread.read_quorum_step_device): a two-process gloo job with
the EMULATOR BUILD of the quorum-read engine (QuorumReadGroup: the shipped kernels compiled for the host) on every rank,
every message a tensor from handler to collective to handler -- against the single-process run of the CPU oracle on the same
rounds (tests/test_spread_qread_gloo.py holds the rounds), on every rank."""
import os
import sys

import numpy as np

from test_spread_qread_gloo import B, G, K, R, ROOT, _free_port, _inputs, _single


def _t(a):
    import torch
    if a is None:
        return None
    if a.dtype == np.uint32:
        a = a.view(np.int32)
    return torch.from_numpy(np.ascontiguousarray(a))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import hostsim
    from summerset_amd import QuorumReadGroup, spread
    dist.init_process_group("gloo", rank=rank, world_size=world)
    save = {}
    with hostsim.patched():
        reps = [QuorumReadGroup(G, R, r, K, B, 1) if spread.owner_of(r, world) == rank else None for r in range(R)]
        for t, (refresh, logs, keys, n, flags, order, stable, kv) in enumerate(_inputs()):
            for r in range(R):
                if reps[r] is not None:
                    reps[r].refresh_highest_slot(_t(refresh[r][0]), _t(refresh[r][1]))
            lg = [{k: _t(v) for k, v in logs[r].items()} if reps[r] is not None else None for r in range(R)]
            outcome, val, done = spread.read_quorum_step_device(reps, rank, world, t % R, 0, _t(keys), _t(n), lg, _t(flags), _t(order),
                                                                _t(stable), _t(kv))
            save["o%d" % t], save["v%d" % t], save["d%d" % t] = outcome.numpy().copy(), val.numpy().view(np.uint32).copy(), done.numpy().copy()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **save)
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_device_resident_quorum_reads(tmp_path, oracle):
    import torch.multiprocessing as mp
    import hostsim
    hostsim.build()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    want = _single(oracle)
    answered = 0
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        for t, (o, v, d) in enumerate(want):
            assert np.array_equal(got["o%d" % t], o) and np.array_equal(got["v%d" % t], v) and np.array_equal(got["d%d" % t], d), (rank, t)
            answered += int(d.sum())
    assert answered > 100


def run_one_rank(oracle, dev):
    """world 1 (no process group): the same function, nothing leaves the rank"""
    from summerset_amd import QuorumReadGroup, spread
    want = _single(oracle)
    d_ = lambda a: None if a is None else _t(a).to(dev)
    reps = [QuorumReadGroup(G, R, r, K, B, 1) for r in range(R)]
    for t, (refresh, logs, keys, n, flags, order, stable, kv) in enumerate(_inputs()):
        for r in range(R):
            reps[r].refresh_highest_slot(d_(refresh[r][0]), d_(refresh[r][1]))
        lg = [{k: d_(v) for k, v in logs[r].items()} for r in range(R)]
        o, v, d = spread.read_quorum_step_device(reps, 0, 1, t % R, 0, d_(keys), d_(n), lg, d_(flags), d_(order), d_(stable), d_(kv))
        assert np.array_equal(o.cpu().numpy(), want[t][0]) and np.array_equal(v.cpu().numpy().view(np.uint32), want[t][1]), t
        assert np.array_equal(d.cpu().numpy(), want[t][2]), t


def test_one_rank_is_the_plain_round(oracle):
    import hostsim
    hostsim.build()
    with hostsim.patched():
        run_one_rank(oracle, "cpu")
