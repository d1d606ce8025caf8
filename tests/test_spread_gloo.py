"""Layout L2 on CPU: world_size 2 over gloo.  The five replicas of every group are spread over the
ranks (replica r on rank r mod 2), both ranks run the same lock-step RSPaxos schedule (summerset_amd/rsp_cluster.py),
and every handler's outputs -- the messages -- are exchanged by summerset_amd.spread.SpreadReplica.  The CPU oracle's
replica objects stand in for the HIP engine (which needs a GPU).  The spread job must be the single-process one:
same commit log on every rank, same final state of every replica."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, R, W, TICKS, FT, SEED, LOSS = 24, 5, 32, 15, 1, 9, 0.1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _log_digest(log):
    """the commit events of a run, reduced to comparable arrays"""
    out = []
    for t, events in log:
        for e in events:
            if e["kind"] == "commit":
                out.append(np.concatenate([[t, e["s"]], e["slot"], e["val"], e["committed"]]).astype(np.int64))
    return np.stack(out)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import rsp_scenarios as sc
    from oracle import oracle as O
    from summerset_amd import spread
    dist.init_process_group("gloo", rank=rank, world_size=world)
    reps = []
    for r in range(R):
        mine = spread.owner_of(r, world) == rank
        reps.append(spread.SpreadReplica(r, O.RspOracle(G, R, me=r, W=W, fault_tolerance=FT) if mine else None, G, W, rank, world))
    log = sc.run(reps, G, TICKS, seed=SEED, loss=LOSS)
    save = dict(log=_log_digest(log), exchanged=np.array([sum(x.bytes_exchanged for x in reps)]))
    for r in range(R):
        if reps[r].local is not None:
            for k, v in reps[r].dump().items():
                save["rep%d_%s" % (r, k)] = v
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **save)
    dist.barrier()
    dist.destroy_process_group()


def test_owner_map():
    from summerset_amd import spread
    assert [spread.owner_of(r, 2) for r in range(5)] == [0, 1, 0, 1, 0]
    assert [spread.owner_of(r, 8) for r in range(5)] == [0, 1, 2, 3, 4]
    assert [spread.owner_of(r, 1) for r in range(5)] == [0] * 5


def test_world_one_is_the_plain_cluster(oracle):
    """without a process group the wrapper only adds a pack / unpack of every message"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rsp_scenarios as sc
    from summerset_amd import spread
    plain = [oracle.RspOracle(G, R, me=r, W=W, fault_tolerance=FT) for r in range(R)]
    wrapped = [spread.SpreadReplica(r, oracle.RspOracle(G, R, me=r, W=W, fault_tolerance=FT), G, W, 0, 1) for r in range(R)]
    a, b = sc.run(plain, G, TICKS, seed=SEED, loss=LOSS), sc.run(wrapped, G, TICKS, seed=SEED, loss=LOSS)
    assert np.array_equal(_log_digest(a), _log_digest(b))
    for r in range(R):
        x, y = plain[r].dump(), wrapped[r].dump()
        for k in x:
            assert np.array_equal(x[k], y[k]), (r, k)


def test_two_ranks_over_gloo(oracle, tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rsp_scenarios as sc
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(2)]
    ref = [oracle.RspOracle(G, R, me=r, W=W, fault_tolerance=FT) for r in range(R)]
    ref_log = _log_digest(sc.run(ref, G, TICKS, seed=SEED, loss=LOSS))
    for k in range(2):
        assert np.array_equal(ranks[k]["log"], ref_log)          # every rank saw the same commits as the single process
        assert int(ranks[k]["exchanged"][0]) > 0
    assert (ref_log[:, 2 + 2 * G:] > 0).any()
    for r in range(R):
        d, have = ref[r].dump(), ranks[r % 2]
        for n in d:
            assert np.array_equal(have["rep%d_%s" % (r, n)], d[n]), (r, n)
        assert ("rep%d_leader" % r) not in ranks[1 - r % 2].files   # and only its owner holds it
