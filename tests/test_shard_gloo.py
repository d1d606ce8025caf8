"""The N > 1 path of bench.py on CPU: world_size 2 over gloo.

Layout L1: the job's groups are block-partitioned over the
ranks, every rank steps its own shard (here: the CPU oracle stands in for the
HIP engine, which needs a GPU), and the metric is reduced at the end -- MAX of
the elapsed time, SUM of the committed slots.  The sharded job must be the
unsharded one: same streams (keyed by global group id), same commits."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G_TOTAL, R, S, W, TICKS = 96, 5, 3, 64, 20


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream(n_groups, base):
    from summerset_amd import stream
    return stream.MultiPaxosStream(n_groups, R, S, cap=W + 4, n_ticks=TICKS, drop_p=0.1, timeout_frac=0.3,
                                   hb_every=4, max_drop=2, group_base=base)


def _run_shard(lo, hi):
    from oracle import oracle as O
    m = O.MpOracle(hi - lo, R, W, cap=W + 4, record_commits=False)
    m.preset_leader(0)
    st = _stream(hi - lo, lo)
    for t in range(TICKS):
        m.tick(**st.tick(t))
    commits = sum(m.total_commits(r) for r in range(R))
    return commits, m.dump(0)["commit_bar"].copy(), m.dump(1)["leader"].copy()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import torch.distributed as dist
    from summerset_amd import shard
    r, _, w = shard.env_world()
    dist.init_process_group("gloo", rank=r, world_size=w)
    lo, hi = shard.group_range(G_TOTAL, w, r)
    commits, cbar, leader = _run_shard(lo, hi)
    elapsed, total = shard.reduce_metric(1.0 + r, commits)          # rank 1 is "slower": MAX must pick 2.0
    np.savez(os.path.join(out_dir, "rank%d.npz" % r), lo=lo, hi=hi, commits=commits, cbar=cbar, leader=leader,
             elapsed=elapsed, total=total)
    dist.barrier()
    dist.destroy_process_group()


def test_group_range_partitions():
    from summerset_amd import shard
    for total, world in ((96, 2), (65536, 8), (10, 3), (3, 8), (0, 4)):
        edges = [shard.group_range(total, world, k) for k in range(world)]
        assert edges[0][0] == 0 and edges[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
        sizes = [hi - lo for lo, hi in edges]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard.group_range(8, 2, 2)
    assert shard.reduce_metric(0.5, 7) == (0.5, 7)                   # no process group: identity


def test_sharded_stream_is_the_global_stream():
    full = _stream(G_TOTAL, 0)
    a, b = _stream(40, 0), _stream(G_TOTAL - 40, 40)
    assert np.array_equal(full.timeout_tick, np.concatenate([a.timeout_tick, b.timeout_tick]))
    for t in (0, 7):
        f, x, y = full.tick(t), a.tick(t), b.tick(t)
        for k in ("req_val", "ackctl"):
            assert np.array_equal(f[k], np.concatenate([x[k], y[k]], axis=1)), k
        for k in ("timeout_rep", "req_target", "req_cnt"):
            assert np.array_equal(f[k], np.concatenate([x[k], y[k]])), k


def test_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(str(tmp_path / ("rank%d.npz" % k))) for k in range(2))
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 48, 48, 96)
    # both ranks hold the reduced metric
    assert float(r0["elapsed"]) == float(r1["elapsed"]) == 2.0
    assert int(r0["total"]) == int(r1["total"]) == int(r0["commits"]) + int(r1["commits"])
    # ... and the sharded job is the unsharded one
    sys.path.insert(0, ROOT)
    commits, cbar, leader = _run_shard(0, G_TOTAL)
    assert commits == int(r0["total"]) and commits > 0
    assert np.array_equal(cbar, np.concatenate([r0["cbar"], r1["cbar"]]))
    assert np.array_equal(leader, np.concatenate([r0["leader"], r1["leader"]]))


def _l2_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a = bench._l2_outcome_of_all_ranks(dist, world, False)              # nobody failed
    with open(os.path.join(out_dir, "l2_rank%d.txt" % rank), "w") as f:
        f.write("%d" % int(a))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_settle_the_l2_outcome_through_the_store(tmp_path):
    """bench.py at N > 1: a rank whose own L2 pass was fine must learn that a peer's failed before it walks into the
    closing barrier -- through the process group's store.  Two gloo ranks: nobody failed -> False on both."""
    import torch.multiprocessing as mp
    mp.spawn(_l2_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for k in range(2):
        assert open(str(tmp_path / ("l2_rank%d.txt" % k))).read().split()[0] == "0"


def _l2_fail_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bad = bench._l2_outcome_of_all_ranks(dist, world, rank == 1, wait_s=20.0)   # rank 1's pass "failed"
    with open(os.path.join(out_dir, "l2f_rank%d.txt" % rank), "w") as f:
        f.write("%d" % int(bad))
    dist.barrier()
    dist.destroy_process_group()


def test_one_failed_l2_pass_is_seen_by_every_rank(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_l2_fail_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert [open(str(tmp_path / ("l2f_rank%d.txt" % k))).read() for k in range(2)] == ["1", "1"]


def test_bench_launch_plan():
    """`python bench.py --gpus N` is how the driver asks for N ranks at round end (through torchrun) and how a user does
    without one: the flag and the launcher's environment must agree, and without a launcher the script spawns the ranks"""
    from summerset_amd import shard
    assert shard.resolve_world(1, {}) == ("run", 0, 0, 1)
    assert shard.resolve_world(8, {}) == ("spawn", 8)
    assert shard.resolve_world(4, {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}) == ("run", 3, 3, 4)
    assert shard.resolve_world(1, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}) == ("run", 0, 0, 1)
    with pytest.raises(ValueError):
        shard.resolve_world(8, {"WORLD_SIZE": "1"})           # round 1's bug shape: --gpus parsed, never honoured
    with pytest.raises(ValueError):
        shard.resolve_world(1, {"WORLD_SIZE": "2"})
    with pytest.raises(ValueError):
        shard.resolve_world(0, {})
    cmd = shard.launch_command(8, "bench.py", ["--gpus", "8", "--steps", "20"], port=29511, python="python")
    assert cmd == ["python", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", "29511", "bench.py", "--gpus", "8", "--steps", "20"]


def test_bench_self_spawns_the_ranks():
    """the real thing on CPU: `python bench.py --gpus 2 --launch-check` starts two ranks (gloo: there is no GPU here),
    they count themselves with one all-reduce and rank 0 prints one line"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == 2 and d["backend"] == "gloo"
    # a launcher whose world is not --gpus is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "launcher started 1 ranks" in bad.stderr


def test_bench_help_prints():
    """argparse expands % in help strings: a stray one makes `bench.py --help` a traceback"""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--layout" in out.stdout and "spread-epaxos" in out.stdout, out.stderr[-400:]
