"""The N>1 path on CPU: two gloo ranks shard a list of images round-robin, every
image is processed exactly once, and the whole-job figures (max elapsed time,
summed image count) are what bench.py reports."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_items, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import miseg_amd
    from miseg_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard.shard_indices(n_items, rank, world)
    shard.barrier()
    elapsed = shard.max_over_ranks(1.0 + rank)
    total = shard.sum_over_ranks(len(mine))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        torch.save({"elapsed": elapsed, "total": total, "gathered": gathered}, os.path.join(out_dir, "r.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [7, 8, 1])
def test_two_rank_sharding(tmp_path, n_items):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    r = torch.load(os.path.join(str(tmp_path), "r.pt"))
    assert r["elapsed"] == 2.0                    # the slower rank
    assert r["total"] == n_items
    flat = sorted(i for part in r["gathered"] for i in part)
    assert flat == list(range(n_items))           # every image exactly once
    assert abs(len(r["gathered"][0]) - len(r["gathered"][1])) <= 1


def _predict_files_worker(rank, world, images, out_dir):
    """What a rank of `predict.py --gpu 0,1` does before it touches a GPU: reads RANK / WORLD_SIZE
    and picks its files of the sorted list."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    from utils.predict import files_of_rank
    files, r, w = files_of_rank(images)
    assert (r, w) == (rank, world)
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("\n".join("%s_prediction.png" % os.path.splitext(os.path.basename(x))[0] for x in files))


@pytest.mark.parametrize("n_files,world", [(5, 2), (4, 2), (1, 2), (7, 3)])
def test_prediction_files_are_dealt_over_ranks_exactly_once(tmp_path, n_files, world):
    """Multi-GPU inference shards the sorted file list of /root/reference utils/predict.py:125-131,152
    over one process per GPU with no collective: the union of the names the ranks would write is
    every file exactly once."""
    from PIL import Image
    import numpy as np
    images = os.path.join(str(tmp_path), "images")
    os.makedirs(images)
    for i in range(n_files):
        Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(os.path.join(images, "slide%02d.png" % i))
    mp.spawn(_predict_files_worker, args=(world, images, str(tmp_path)), nprocs=world, join=True)
    names = []
    for r in range(world):
        with open(os.path.join(str(tmp_path), "rank%d.txt" % r)) as f:
            names += [n for n in f.read().split("\n") if n]
    assert sorted(names) == ["slide%02d_prediction.png" % i for i in range(n_files)]
    assert len(set(names)) == n_files


def test_tile_ranges_match_reference_patch_predict():
    import miseg_amd
    from miseg_amd import shard
    assert shard.tile_ranges(1920, 512) == [(0, 512), (512, 1024), (1024, 1536), (1408, 1920)]
    assert shard.tile_ranges(2560, 2560) == [(0, 2560)]
    with pytest.raises(ValueError):
        shard.shard_indices(4, 2, 2)


def _bucket_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import miseg_amd  # noqa: F401
    from miseg_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 10000
    flat = torch.zeros(n)
    b = shard.GradBuckets(flat, bucket_bytes=4 * 3000)
    launched = []
    for step in range(2):                          # two backward passes reuse the object
        flat.zero_()
        b.begin()
        # "layers" fill the buffer from the end towards the start, as a backward pass does
        bounds = [10000, 9000, 8500, 6000, 5900, 2500, 100, 0]
        for hi, lo in zip(bounds[:-1], bounds[1:]):
            flat[lo:hi] = torch.arange(lo, hi, dtype=torch.float32) * (rank + 1) + step
            b.ready(lo)
        b.flush()
        b.wait()
        launched.append(list(b.launched))
    if rank == 0:
        torch.save({"flat": flat, "launched": launched}, os.path.join(out_dir, "b.pt"))
    dist.destroy_process_group()


def test_gradient_buckets_two_ranks(tmp_path):
    """Bucketed gradient all-reduce: every element summed over ranks exactly once, in few large
    buckets, whatever the layer boundaries."""
    world = 2
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = torch.load(os.path.join(str(tmp_path), "b.pt"))
    want = torch.arange(10000, dtype=torch.float32) * 3 + 2      # ranks 1x + 2x, step 1 on both
    assert torch.equal(r["flat"], want)
    for launched in r["launched"]:
        assert launched == [(6000, 10000), (2500, 6000), (0, 2500)]
        covered = sorted(launched)
        assert covered[0][0] == 0 and covered[-1][1] == 10000
        assert all(a[1] == b_[0] for a, b_ in zip(covered[:-1], covered[1:]))


def test_gradient_buckets_single_process_is_a_no_op():
    import miseg_amd  # noqa: F401
    from miseg_amd import shard
    flat = torch.ones(100)
    b = shard.GradBuckets(flat, bucket_bytes=64)
    b.begin()
    b.ready(50)
    b.flush()
    b.wait()
    assert b.launched == [] and torch.equal(flat, torch.ones(100))
