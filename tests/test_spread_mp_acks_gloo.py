"""The AcceptReply exchange of layout L2 for the MultiPaxos cluster engine, on CPU: world_size 2 over gloo.  Every rank
holds records its followers produced for groups all over the job (global group ids); split_acks_by_owner +
mp_exchange_acks must hand every rank exactly the records of its own block of groups, rebased, sender-major -- what
smr_mp_deliver_acks then puts into the leaders' ack matrices.  (The record <-> matrix kernels themselves:
tests/test_mp_gpu.py::test_accept_replies_as_records, on the emulator in tests/test_hostsim.py.)"""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOTAL, R, ROUNDS = 101, 5, 6


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _produced(rank, t):
    """what rank `rank` holds after R2 of tick t: a seeded, ragged bag of records (tick 3: nothing at all on rank 1)"""
    from summerset_amd.multipaxos import ACK_DTYPE
    rng = np.random.default_rng(100 * t + rank)
    n = 0 if (t == 3 and rank == 1) else int(rng.integers(1, 400))
    rec = np.zeros(n, ACK_DTYPE)
    rec["group"] = rng.integers(0, TOTAL, n)
    rec["slot"] = rng.integers(0, 50, n)
    rec["ballot"] = (rng.integers(1, 4, n).astype(np.uint64) << np.uint64(8)) | np.uint64(1)
    rec["peer"] = rng.integers(1, R, n)
    return rec


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from summerset_amd import spread
    dist.init_process_group("gloo", rank=rank, world_size=world)
    save = {}
    for t in range(ROUNDS):
        out = spread.split_acks_by_owner(_produced(rank, t), TOTAL, world)
        save["t%d" % t] = spread.mp_exchange_acks(out, rank, world).view(np.uint8)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **save)
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT)
    from summerset_amd import shard
    from summerset_amd.multipaxos import ACK_DTYPE
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    total = 0
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        lo, hi = shard.group_range(TOTAL, world, rank)
        for t in range(ROUNDS):
            want = []
            for src in range(world):                                  # sender-major, each sender's own order kept
                rec = _produced(src, t)
                rec = rec[(rec["group"] >= lo) & (rec["group"] < hi)].copy()
                rec["group"] -= lo
                want.append(rec)
            want = np.concatenate(want)
            have = got["t%d" % t].view(ACK_DTYPE)
            assert np.array_equal(have, want), (rank, t, len(have), len(want))
            total += len(have)
    assert total > 500


def test_single_rank_is_identity():
    sys.path.insert(0, ROOT)
    from summerset_amd import spread
    rec = _produced(0, 0)
    parts = spread.split_acks_by_owner(rec, TOTAL, 1)
    assert len(parts) == 1 and np.array_equal(spread.mp_exchange_acks(parts, 0, 1), rec)
