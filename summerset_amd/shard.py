"""Multi-GPU layout L1: groups are independent, so a job of
`total_groups` replica groups is block-partitioned over the ranks -- rank k owns
groups [k * total / N, (k + 1) * total / N) with ALL their replicas -- and runs
with no data-path collective.  The only communication is the end-of-run
reduction of the metric: MAX of the elapsed time, SUM of the committed slots.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on the GPU box,
"gloo" in the CPU tests).
"""
import os


def env_world():
    """(rank, local_rank, world_size) as torchrun exports them; (0, 0, 1) when run alone"""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def resolve_world(gpus_arg, env=None):
    """What `bench.py --gpus N` has to do, from the flag and the environment:
      ("run", rank, local, world)   -- this process is one rank (torchrun's env is present, or N == 1)
      ("spawn", N)                  -- plain `python bench.py --gpus N` with N > 1: nothing launched the ranks yet,
                                       so the script launches them itself (launch_command) and relays rank 0's line
    A torchrun world that disagrees with --gpus is an error, never silently one of the two."""
    env = os.environ if env is None else env
    n = int(gpus_arg)
    if n < 1:
        raise ValueError("--gpus must be >= 1")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != n:
            raise ValueError("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (n, world))
        return ("run", int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), world)
    if n == 1:
        return ("run", 0, 0, 1)
    return ("spawn", n)


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n, script, argv, port=None, python=None):
    """the command line the driver itself uses for N > 1: one rank per GPU of ONE node, rendezvous on 127.0.0.1"""
    import sys
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), script] + list(argv)


def group_range(total_groups, world, rank):
    """block partition: the first (total % world) ranks own one group more"""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    q, r = divmod(int(total_groups), int(world))
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def reduce_metric(elapsed_s, commits, device=None):
    """whole-job (max elapsed over ranks, total committed slots); identity without a process group"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), int(commits)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    c = torch.tensor([commits], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c.item())


def count_ranks(device=None):
    """how many ranks really took part (one all-reduce of ones over the job's backend); 1 without a process group"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    one = torch.ones(1, dtype=torch.int64, device=device)
    dist.all_reduce(one, op=dist.ReduceOp.SUM)
    return int(one.item())
