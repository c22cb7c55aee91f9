"""Multi-GPU plumbing: ensembles (independent IQ streams) are the unit of sharding -- every piece of persistent state
(frequency offsets, sync state, CIF history ring) is per ensemble, so ranks never exchange data-path bytes.
torch.distributed (RCCL on GPUs, gloo in the CPU tests) is used only for the timing barrier, the max-over-ranks of the
elapsed time and, in tests, to check that the shards tile the ensemble set."""


def shard_range(n_units, rank, world):
    """contiguous block partition of units 0..n_units-1: returns (first, count) for `rank`; sizes differ by at most 1"""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n_units, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def barrier(dist=None):
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, dist=None, device=None):
    """max of a python float over all ranks (identity when not distributed)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def all_ranks(value, dist=None, device=None):
    """the python float of every rank, in rank order (a list of one when not distributed): the per-rank step times of a scaling run,
    which show a straggler that the max over ranks hides"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    import torch
    world = dist.get_world_size()
    t = torch.zeros(world, dtype=torch.float64, device=device or "cpu")
    t[dist.get_rank()] = float(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.cpu()]
