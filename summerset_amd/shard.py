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
