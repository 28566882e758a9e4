"""Multi-GPU plumbing for the match_trie path: one process per GPU, independent rid shards, no data-path
collective.  torch.distributed is imported only when WORLD_SIZE > 1 and only for the
barrier and two scalar reductions of the benchmark (gloo: CPU scalars; nothing moves over xGMI)."""
import os


def rank_info():
    """-> (rank, local_rank, world_size) from the torchrun environment (1 process if unset)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_dist():
    """Initialise the process group if launched with WORLD_SIZE > 1; returns torch.distributed or None."""
    _, _, world = rank_info()
    if world <= 1:
        return None
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    return dist


def barrier(dist):
    if dist is not None:
        dist.barrier()


def reduce_timing(dist, elapsed_s, units):
    """-> (max elapsed over ranks, total units over ranks)."""
    if dist is None:
        return float(elapsed_s), float(units)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    u = torch.tensor([float(units)], dtype=torch.float64)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t[0]), float(u[0])
