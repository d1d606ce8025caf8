"""Layout L2 for the quorum reads on CPU: world_size 2 over gloo.  The five replicas of every group are spread over the
ranks; a ReadQuery round = every rank answers for its replicas, one all-gather of the replies, the issuer's rank tallies,
one broadcast of the answers (summerset_amd.spread.read_quorum_step).  The CPU oracle's objects stand in for the HIP
engine (which needs a GPU).  The spread job must give the answers of the single-process one, on every rank."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, R, K, B, W, ROUNDS = 40, 5, 7, 3, 16, 10


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs():
    """per round: what every replica's table learns, every replica's log view, the query, loss and delivery order"""
    rng = np.random.default_rng(17)
    rounds = []
    for t in range(ROUNDS):
        refresh = [(rng.integers(0, 12, G).astype(np.uint32), np.where(rng.random((B, G)) < 0.7, rng.integers(0, K, (B, G)), 0xFF).astype(np.uint8))
                   for _ in range(R)]
        logs = []
        for r in range(R):
            status = rng.integers(2, 5, (W, G)).astype(np.uint8)
            logs.append(dict(start_slot=np.zeros(G, np.uint32), log_end=rng.integers(4, 13, G).astype(np.uint32), status=status,
                             token=(1000 * (np.arange(W)[:, None] + 1) + np.arange(G)[None, :]).astype(np.uint32)))
        keys = rng.integers(0, K, (B, G)).astype(np.uint8)
        n = rng.integers(0, B + 1, G).astype(np.uint8)
        flags = (rng.random((R, G)) < 0.85).astype(np.uint8)
        order = np.array([sum(int(p) << (3 * i) for i, p in enumerate(rng.permutation(R))) for _ in range(G)], np.uint32)
        stable = (rng.random(G) < 0.1).astype(np.uint8)
        kv = rng.integers(0, 50, (K, G)).astype(np.uint32)
        rounds.append((refresh, logs, keys, n, flags, order, stable, kv))
    return rounds


def _single(O):
    reps = [O.QrOracle(G, R, r, K, B, 1) for r in range(R)]
    res = []
    for t, (refresh, logs, keys, n, flags, order, stable, kv) in enumerate(_inputs()):
        iss = t % R
        rep = dict(state=np.zeros((R, B, G), np.uint8), slot=np.zeros((R, B, G), np.uint32), val=np.zeros((R, B, G), np.uint32))
        fl = np.zeros((R, G), np.uint8)
        for r in range(R):
            reps[r].refresh_highest_slot(*refresh[r])
            st = stable if r != iss else None
            out, f = reps[r].handle_read_query(keys, n, logs[r], st, kv if st is not None else None)
            if r == iss:
                reps[r].issue(0, n, out)
            else:
                for k in rep:
                    rep[k][r] = out[k]
                fl[r] = (flags[r] & 1) | ((f << 1) * (flags[r] & 1))
        res.append(reps[iss].handle_replies(0, rep, fl, order))
    return res


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from oracle import oracle as O
    from summerset_amd import spread
    dist.init_process_group("gloo", rank=rank, world_size=world)
    reps = [O.QrOracle(G, R, r, K, B, 1) if spread.owner_of(r, world) == rank else None for r in range(R)]
    save = {}
    for t, (refresh, logs, keys, n, flags, order, stable, kv) in enumerate(_inputs()):
        for r in range(R):
            if reps[r] is not None:
                reps[r].refresh_highest_slot(*refresh[r])
        lg = [logs[r] if reps[r] is not None else None for r in range(R)]
        outcome, val, done = spread.read_quorum_step(reps, rank, world, t % R, 0, keys, n, lg, flags, order, stable, kv)
        save["o%d" % t], save["v%d" % t], save["d%d" % t] = outcome, val, done
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **save)
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = _single(oracle)
    answered = 0
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npz" % rank))
        for t, (o, v, d) in enumerate(want):
            assert np.array_equal(got["o%d" % t], o) and np.array_equal(got["v%d" % t], v) and np.array_equal(got["d%d" % t], d), (rank, t)
            answered += int(d.sum())
    assert answered > 100
