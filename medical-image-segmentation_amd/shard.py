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


class GradBuckets:
    """Data-parallel training (tf.distribute mirrored strategy at
    /root/reference model_train.py:272-301): every rank holds the whole model and its own slice of
    the batch; the gradients, one flat fp32 buffer, are summed over ranks.

    The backward pass fills the buffer from its END towards its start (the last layers' gradients
    are ready first).  `ready(lo)` announces that elements [lo, end of the not-yet-reduced part)
    are final; once at least `bucket_bytes` have piled up they go out as ONE all-reduce, issued
    asynchronously so that RCCL moves them over xGMI while the remaining layers' backward kernels
    run.  `flush()` sends what is left, `wait()` blocks until every bucket has landed.  Few large
    buckets suit the per-link-bound ring of point-to-point xGMI better than many small ones.

    Backend "nccl" (RCCL) reduces the device buffer in place.  With "gloo" (CPU rehearsals and
    single-GPU tests) device memory is staged through the host synchronously.
    """

    def __init__(self, flat, bucket_bytes=25 << 20, group=None):
        self.flat = flat
        self.bucket = max(1, int(bucket_bytes) // flat.element_size())
        self.group = group
        self.hi = flat.numel()
        self.pending = []
        self.launched = []          # (lo, hi) of each bucket, for tests and tuning
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def begin(self):
        self.hi = self.flat.numel()
        self.pending, self.launched = [], []

    def _send(self, lo, hi):
        if self.world <= 1 or hi <= lo:
            return
        part = self.flat[lo:hi]
        self.launched.append((lo, hi))
        if part.is_cuda and dist.get_backend(self.group) != "nccl":
            host = part.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            part.copy_(host)
            return
        self.pending.append(dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, lo):
        """Elements [lo, self.hi) will not change any more in this backward pass."""
        lo = max(0, min(lo, self.hi))
        if self.hi - lo >= self.bucket:
            self._send(lo, self.hi)
            self.hi = lo

    def flush(self):
        self._send(0, self.hi)
        self.hi = 0

    def wait(self):
        for w in self.pending:
            w.wait()
        self.pending = []
