"""Multi-GPU inference shards independent images (or the reference's no-halo
tiles, /root/reference utils/predict.py:44-70) across one process per GPU.
Weights are replicated, nothing is exchanged on the data path; the only
collectives are the barrier and the max/sum reductions a benchmark or a driver
uses to report whole-job figures."""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """Round-robin assignment: item i goes to rank i % world."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, n_items, world))


def tile_ranges(size, patch):
    """Tile ranges along one axis as the reference's patch_predict cuts them:
    non-overlapping, the last one shifted back to end at the border."""
    out, start = [], 0
    n = -(-size // patch)
    for _ in range(n):
        end = start + patch
        if end > size:
            end, start = size, size - patch
        out.append((start, end))
        start = end
    return out


def _tensor(value, dtype, device):
    return torch.tensor([value], dtype=dtype, device=device)


def max_over_ranks(value, device="cpu"):
    """Largest `value` of any rank (the job's elapsed time is its slowest rank's)."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = _tensor(value, torch.float64, device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = _tensor(value, torch.float64, device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
