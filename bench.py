#!/usr/bin/env python3
"""Benchmark of the segmentation-network hot path on MI355X.

A step is one forward pass (the replacement of `model(batch, training=False)`,
/root/reference utils/predict.py:159) of ResNet-34 U-Net, 3 classes, over one
batch of synthetic 1920x2560x3 images per GPU with the inputs already resident
in HBM; images are independent, so ranks share nothing and no collective is on
the data path (weak scaling).  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak of MI355X
HBM_PEAK_GBS = 8000.0
# forward GFLOP per 1920x2560 image of the layer graphs as the reference builds them (2 x MAC over every
# convolution and transpose convolution; linear in H x W).  The engine runs some layers in an algebraically
# equal form with fewer multiply-adds (FPN's final_stage and finest pyramid level), so the executed
# figure can be lower than the graph's.
GRAPH_GFLOP_1920x2560 = {("resnet34", "U-Net"): 1770.6, ("resnet18", "U-Net"): 1408.2,
                         ("densenet169", "Linknet"): 781.9, ("densenet169", "FPN"): 3666.2}


def synthetic_images(n, h, w, seed=1234):
    """uint8 uniform noise -> per-image min-max to [0,1] (utils/data.py:64-80)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8).astype(np.float32)
    lo = img.reshape(n, -1).min(axis=1)[:, None, None, None]
    hi = img.reshape(n, -1).max(axis=1)[:, None, None, None]
    return (img - lo) / (hi - lo)


def build_weights(graph, seed=1234):
    from miseg_amd import weights as W
    return W.init_weights(graph, seed=seed, randomize_bn=True)


def cpu_baseline(encoder, decoder, classes, h, w, weights, budget_s=30.0):
    """Time the CPU oracle (torch-CPU back-end, fp32) on one image of the same
    workload, on this host's cores.  Test infrastructure used as a reported
    baseline only."""
    from oracle import nets, ops_torch
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
    torch.set_num_threads(threads)
    x = synthetic_images(1, h, w, seed=99)
    small = synthetic_images(1, 256, 256, seed=98)
    with torch.no_grad():
        nets.forward(ops_torch, encoder, decoder, weights, small)      # warm-up
        t0 = time.perf_counter()
        n_done = 0
        while True:
            nets.forward(ops_torch, encoder, decoder, weights, x)
            n_done += 1
            el = time.perf_counter() - t0
            if el > budget_s * 0.5 or n_done >= 3:
                break
        # BASELINE config 1: the reference's own CPU-runnable case (predict.py --gpu -1, 512x512x3, batch 1)
        x512 = synthetic_images(1, 512, 512, seed=97)
        nets.forward(ops_torch, encoder, decoder, weights, x512)
        t1 = time.perf_counter()
        n512 = 0
        while n512 < 10 and time.perf_counter() - t1 < 5.0:
            nets.forward(ops_torch, encoder, decoder, weights, x512)
            n512 += 1
        el512 = time.perf_counter() - t1
    return {"value": n_done / el, "unit": "images/s", "cores": threads, "kind": "port",
            "sample": "%d forward pass(es) of 1 image %dx%dx3, oracle/nets.py on torch-CPU fp32 "
                      "(TensorFlow reference not installable offline)" % (n_done, h, w),
            "config1_512x512": {"value": n512 / el512, "unit": "images/s",
                                "sample": "%d forward passes of 1 image 512x512x3, same network and cores" % n512}}


PMC_FILES = {
    # (images per plan, H, W, encoder, decoder) the counter passes were collected on -> reduced file
    (4, 1920, 2560, "resnet34", "U-Net"): "r03_pmc_traffic.json",
    (8, 1920, 2560, "densenet169", "Linknet"): "r03_pmc_traffic_d169_linknet.json",
    (4, 1920, 2560, "densenet169", "FPN"): "r03_pmc_traffic_d169_fpn.json",
    (16, 1920, 2560, "resnet18", "U-Net-train"): "r03_pmc_traffic_train_r18_unet.json",
}


def all_reduce_scalar(dist, t, op=None):
    """In-place all-reduce of a small device tensor: RCCL takes it as it is, gloo (rehearsals) gets a host copy."""
    op = dist.ReduceOp.SUM if op is None else op
    if dist.get_backend() == "nccl":
        dist.all_reduce(t, op=op)
    else:
        host = t.cpu()
        dist.all_reduce(host, op=op)
        t.copy_(host)
    return t


def pmc_traffic(kernel, batch, h, w, encoder, decoder):
    """HBM bytes per launch of `kernel` from the committed counter passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE over the same command and configuration, reduced by scripts/pmc_traffic.py); None for a
    configuration no pass was collected on."""
    name = PMC_FILES.get((batch, h, w, encoder, decoder))
    path = os.path.join(ROOT, "profiles", name) if name else None
    if not path or not os.path.isfile(path):
        return None
    with open(path) as f:
        kernels = json.load(f)["kernels"]
    k = kernels.get(kernel.split("(")[0])
    if k:
        return k["hbm_bytes_per_launch"]
    # a name without template arguments stands for all its instantiations: mean over their launches
    fam = [v for n, v in kernels.items() if n.split("<")[0] == kernel.split("<")[0]]
    if not fam:
        return None
    return sum(v["hbm_bytes_per_launch"] * v["dispatches"] for v in fam) / sum(v["dispatches"] for v in fam)


def training_leg(args, dev, rank, world, dist, h, w, augment=False, global_batch=0):
    """Second headline configuration: ResNet-18 U-Net training steps (forward on batch statistics,
    cce+Dice loss, backward, bucketed gradient all-reduce over RCCL overlapped with the backward
    pass, Adam) on a fixed synthetic batch per GPU -- the loop behind `model.fit`,
    /root/reference model_train.py:284-329 with --loss dice (:172-173).
    global_batch = 0: `--train-batch` images per GPU per step (weak scaling).  global_batch = G: the G images of
    a step are split evenly over the ranks, G / world each (strong scaling: BASELINE config 3's "batch 16,
    data-parallel 1/2/4/8 GPUs", the split tf.distribute makes of the global batch)."""
    from miseg_amd import netspec, trainer
    args = argparse.Namespace(**vars(args))
    if global_batch:
        if global_batch % world:
            raise SystemExit("--train-global-batch %d does not divide over %d ranks" % (global_batch, world))
        args.train_batch = global_batch // world
    graph = netspec.build_graph(args.train_encoder, args.decoder, (h, w, 3), args.classes)
    tr = trainer.Trainer(graph, build_weights(graph), args.train_batch, device=str(dev), loss="cce_dice",
                         world_size=world)
    gen = torch.Generator(device=dev).manual_seed(77 + rank)
    x = torch.rand((args.train_batch, h, w, 3), device=dev, generator=gen)
    lab = torch.randint(0, args.classes, (args.train_batch, h, w), device=dev, generator=gen)
    gt = torch.nn.functional.one_hot(lab, args.classes).float()
    del lab
    if augment:
        # the reference's augment_dataset (utils/data.py:240-295: flips, brightness, contrast, hue, saturation,
        # drawn per image) on the device, inside the timed step
        from miseg_amd import imageops
        arng = np.random.default_rng(4321 + rank)

        def one_step():
            xa, ga = imageops.augment(x, gt, imageops.random_augment_params(args.train_batch, arng))
            tr.train_step(xa, ga)
    else:
        def one_step():
            tr.train_step(x, gt)
    for _ in range(args.train_warmup):
        one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.train_steps):
        one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_scalar(dist, t, dist.ReduceOp.MAX)
        elapsed = float(t.item())
    peak = torch.cuda.max_memory_allocated(dev) / 2 ** 30
    # every rank runs the profiled step: it all-reduces its gradient buckets like any other step
    prof = tr.profile_step(x, gt)
    roof = train_roofline(prof, 1e3 * elapsed / args.train_steps,
                          (args.train_batch, h, w, args.train_encoder, args.decoder + "-train")) if rank == 0 else None
    del tr, x, gt
    torch.cuda.empty_cache()
    enc_names = {"resnet18": "ResNet-18", "resnet34": "ResNet-34", "densenet169": "DenseNet-169"}
    return {"metric": "images/sec at %dx%dx3 %s %s training" % (h, w, enc_names.get(args.train_encoder, args.train_encoder),
                                                                 args.decoder),
            "value": args.train_batch * world * args.train_steps / elapsed, "unit": "images/s",
            "ms_per_step": 1e3 * elapsed / args.train_steps, "steps": args.train_steps,
            "warmup": args.train_warmup, "scaling": "strong" if global_batch else "weak",
            "global_batch": args.train_batch * world, "n_gpus": world,
            "dtype": "bf16 activations/gradients, fp32 accumulation and master weights",
            "config": {"workload": "%s %s training %dx%dx3, %d classes, cce+Dice loss, Adam, batch %d "
                                   "per GPU per step%s%s" % (args.train_encoder, args.decoder, h, w,
                                                           args.classes, args.train_batch,
                                                           " (global batch %d split over %d GPUs)" % (global_batch, world)
                                                           if global_batch else "",
                                                           ", on-device augmentation (flips, brightness, contrast, "
                                                           "hue, saturation) inside the step" if augment else ""),
                       "parallelism": "data parallel, bucketed RCCL all-reduce overlapped with backward"
                       if world > 1 else "single GPU"},
            "peak_hbm_gib": peak, "roofline": roof}


def train_roofline(prof, ms_per_step, traffic_key=None):
    """`roofline` of the training leg from one step timed launch by launch (HIP events on each
    launch's own stream, Trainer.profile_step).  The dominant kernel is the one with the largest
    summed duration; convolution kernels are priced against the bf16 MFMA peak with their
    algorithmic FLOP (forward, data gradient and weight gradient: 2 x MAC each), the others against
    the HBM peak with their algorithmic bytes (operands read once, results written once).
    Weight gradients run on a second stream beside the data-gradient chain, so the launch durations
    include that contention and their sum exceeds the step time."""
    dom_name = max(prof, key=lambda k: prof[k]["ms"])
    dom = prof[dom_name]
    conv = [g for g in prof.values() if g["flops"] > 0]
    conv_ms, conv_flops = sum(g["ms"] for g in conv), sum(g["flops"] for g in conv)
    other = [g for g in prof.values() if g["flops"] == 0]
    other_ms, other_bytes = sum(g["ms"] for g in other), sum(g["bytes"] for g in other)
    total_bytes = sum(g["bytes"] for g in prof.values())
    if dom["flops"] > 0:
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        head = {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_BF16_PEAK_TFLOPS, "gflop_per_launch": dom["flops"] / dom["launches"] / 1e9}
    else:
        ach = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        head = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"]}
    head.update({
        "kernel": dom_name, "launches_per_step": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
        "timing": "HIP events around every launch of one step, on the launch's own stream",
        "traffic": pmc_traffic(dom_name, *traffic_key) if traffic_key else None,
        "all_conv_launches": {"achieved": conv_flops / (conv_ms * 1e-3) / 1e12,
                              "frac": conv_flops / (conv_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                              "gflop_per_step": conv_flops / 1e9, "ms_per_step": conv_ms,
                              "note": "forward + data-gradient + weight-gradient convolutions, summed launch durations"},
        "bandwidth_bound_launches": {"achieved_gbs": other_bytes / (other_ms * 1e-3) / 1e9,
                                     "frac": other_bytes / (other_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     "gbytes_per_step": other_bytes / 1e9, "ms_per_step": other_ms},
        "whole_step": {"algorithmic_gbytes": total_bytes / 1e9, "gflop": conv_flops / 1e9, "ms": ms_per_step,
                       "hbm_frac": total_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "mfma_frac": conv_flops / (ms_per_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS},
        "kernels": {k: {"launches": g["launches"], "ms": round(g["ms"], 3),
                        "tflops": round(g["flops"] / (g["ms"] * 1e-3) / 1e12, 1) if g["flops"] else None,
                        "gbs": round(g["bytes"] / (g["ms"] * 1e-3) / 1e9, 1)}
                    for k, g in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:12]},
    })
    return head


def whole_pass(encoder, decoder, h, w, ips, gflop_exec, gbytes_img, nops):
    """The whole forward pass against both peaks.  `mfma_frac` prices the multiply-adds the kernels execute;
    `mfma_frac_graph` prices the layer graph as the reference states it (the figure a layer-by-layer run would
    execute), for the networks whose graph figure is tabulated."""
    out = {"gflop_per_image_executed": gflop_exec, "algorithmic_gbytes_per_image": gbytes_img,
           "mfma_frac": ips * gflop_exec / 1e3 / MFMA_BF16_PEAK_TFLOPS,
           "mfma_frac_uses": "gflop_per_image_executed",
           "hbm_frac": ips * gbytes_img / HBM_PEAK_GBS, "launches_per_pass": int(nops)}
    g = GRAPH_GFLOP_1920x2560.get((encoder, decoder))
    if g is not None:
        g = g * (h * w) / (1920.0 * 2560.0)
        out["gflop_per_image_graph"] = g
        out["mfma_frac_graph"] = ips * g / 1e3 / MFMA_BF16_PEAK_TFLOPS
    return out


def extra_config(encoder, decoder, classes, h, w, batch, streams, dev, storage, steps=20, warmup=3, repeats=3):
    """One more inference configuration of BASELINE.json (the DenseNet-169 models), measured the
    same way as the headline one on this rank's GPU: images/s over `steps` steps of `streams`
    concurrent plans of `batch` images, the dominant kernel's roofline from per-launch HIP events,
    and the whole pass against the HBM peak with the algorithmic bytes of its convolution layers
    (operands read once, results written once)."""
    from miseg_amd import engine, netspec
    graph = netspec.build_graph(encoder, decoder, (h, w, 3), classes)
    weights = build_weights(graph)
    engs = [engine.Engine(graph, weights, batch=batch, device=str(dev), storage=storage) for _ in range(streams)]
    strs = [torch.cuda.Stream(dev) for _ in range(streams)]
    for i, e in enumerate(engs):
        e.x_in.copy_(torch.from_numpy(synthetic_images(batch, h, w, seed=4321 + i)).to(dev))
    torch.cuda.synchronize()

    def step():
        for e, st in zip(engs, strs):
            with torch.cuda.stream(st):
                e.run()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    rates = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        rates.append(batch * streams * steps / (time.perf_counter() - t0))
    elapsed = batch * streams * steps / float(np.median(rates))
    eng = engs[0]
    nops = eng.num_ops
    op_ms = np.zeros(nops)
    for _ in range(3):
        for i in range(nops):
            op_ms[i] += eng.time_ops(i, i + 1)
    op_ms /= 3
    macs = np.array([eng.op_macs(i) for i in range(nops)], dtype=np.float64)
    kernels = [eng.op_kernel(i) for i in range(nops)]
    groups = {}
    for i in range(nops):
        g = groups.setdefault(kernels[i], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
        g["launches"] += 1
        g["ms"] += op_ms[i]
        g["flops"] += 2.0 * macs[i]
        g["bytes"] += eng.op_bytes[i]
    dom_name = max(groups, key=lambda k: groups[k]["ms"])
    dom = groups[dom_name]
    ips = batch * streams * steps / elapsed
    conv = macs > 0
    gbytes_img = sum(eng.op_bytes) / batch / 1e9
    gflop_img = 2.0 * macs.sum() / batch / 1e9
    enc_names = {"resnet18": "ResNet-18", "resnet34": "ResNet-34", "densenet169": "DenseNet-169"}
    out = {
        "metric": "images/sec at %dx%dx3 %s %s inference" % (h, w, enc_names.get(encoder, encoder), decoder),
        "value": ips, "unit": "images/s", "steps": steps, "warmup": warmup,
        "repeats": {"count": repeats, "min": min(rates), "median": float(np.median(rates)), "max": max(rates),
                    "note": "`value` is the median of the repeats of `steps` steps each"},
        "ms_per_step": 1e3 * elapsed / steps, "dtype": storage,
        "config": {"workload": "%s %s inference %dx%dx3, %d classes, random-init weights, %d concurrent plan(s) of %d "
                               "images, %s storage" % (encoder, decoder, h, w, classes, streams, batch, storage)},
        "roofline": {
            "bound": "mfma", "kernel": dom_name, "launches_per_step": dom["launches"],
            "avg_launch_ms": dom["ms"] / dom["launches"],
            "gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
            "achieved": dom["flops"] / (dom["ms"] * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": dom["flops"] / (dom["ms"] * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
            "share_of_pass": dom["ms"] / op_ms.sum(),
            "traffic": pmc_traffic(dom_name, batch, h, w, encoder, decoder),
            "timing": "HIP events on the launch stream, one plan at a time (averaged over 3 passes)",
            "all_conv_launches": {"achieved": 2.0 * macs.sum() / (op_ms[conv].sum() * 1e-3) / 1e12,
                                  "frac": 2.0 * macs.sum() / (op_ms[conv].sum() * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                  "ms_per_pass": float(op_ms[conv].sum()),
                                  "non_conv_ms_per_pass": float(op_ms[~conv].sum())},
            "whole_pass": whole_pass(encoder, decoder, h, w, ips, gflop_img, gbytes_img, nops),
        },
    }
    del engs, eng
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="images per plan (one launch sequence)")
    ap.add_argument("--streams", type=int, default=2,
                    help="independent plans per GPU, each on its own HIP stream; a step runs every plan once "
                         "(images per GPU per step = batch x streams)")
    ap.add_argument("--height", type=int, default=1920)
    ap.add_argument("--width", type=int, default=2560)
    ap.add_argument("--encoder", default="resnet34")
    ap.add_argument("--decoder", default="U-Net")
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training leg")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the DenseNet-169 LinkNet / FPN inference lines (the `configs` array)")
    ap.add_argument("--train-encoder", default="resnet18")
    ap.add_argument("--train-batch", type=int, default=16, help="training images per GPU per step")
    ap.add_argument("--train-global-batch", type=int, default=16,
                    help="with more than one GPU, a second training leg (`train_strong`) splits this many images "
                         "per step evenly over the GPUs (strong scaling: BASELINE config 3); 0 skips it.  On one "
                         "GPU it coincides with --train-batch 16 and is not run twice")
    ap.add_argument("--train-steps", type=int, default=10)
    ap.add_argument("--train-warmup", type=int, default=2)
    ap.add_argument("--per-op", action="store_true", help="print a per-kernel table to stderr")
    ap.add_argument("--storage", default="fp16", choices=["fp16", "bf16"],
                    help="16-bit storage format of activations and weights (same kernels, rate and bytes; fp16 is "
                         "the engine's default because its masks match the fp32 reference on >= 99.9 %% of pixels)")
    ap.add_argument("--residual", default="auto", choices=["auto", "16", "32"],
                    help="ResNet residual stream: 16-bit storage format or fp32 (auto: 16-bit with fp16 storage, "
                         "fp32 with bf16 storage; see Engine)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d "
                             "--master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # MISEG_BENCH_REHEARSAL=1: the ranks share the GPUs that are there (rank % device count) and talk over gloo --
    # RCCL refuses two ranks on one device.  For walking the N > 1 code path on a one-GPU box; its numbers mean
    # nothing and the line says so.
    rehearsal = bool(os.environ.get("MISEG_BENCH_REHEARSAL")) and world > 1
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import miseg_amd
    from miseg_amd import engine, netspec

    h, w = args.height, args.width
    graph = netspec.build_graph(args.encoder, args.decoder, (h, w, 3), args.classes)
    weights = build_weights(graph)
    # Images are independent: each plan owns its buffers and runs on its own stream, so the tail of
    # one plan's kernel overlaps the head of another's.
    engs = [engine.Engine(graph, weights, batch=args.batch, device=str(dev), residual_bf16={"auto": None, "16": True, "32": False}[args.residual],
                          storage=args.storage)
            for _ in range(args.streams)]
    streams = [torch.cuda.Stream(dev) for _ in range(args.streams)]
    eng = engs[0]
    for i, e in enumerate(engs):
        x = torch.from_numpy(synthetic_images(args.batch, h, w, seed=1234 + rank * args.streams + i)).to(dev)
        e.x_in.copy_(x)
    torch.cuda.synchronize()

    def step():
        for e, s in zip(engs, streams):
            with torch.cuda.stream(s):
                e.run()

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_scalar(dist, t, dist.ReduceOp.MAX)
        elapsed = float(t.item())

    train = None
    train5 = None
    train_strong = None
    ranks_seen = None
    if dist is not None:
        # what RCCL itself saw: the sum of one 1 per rank
        t = torch.ones(1, dtype=torch.float32, device=dev)
        all_reduce_scalar(dist, t)
        ranks_seen = int(t.item())
    if not args.no_train:
        train = training_leg(args, dev, rank, world, dist, h, w)
        if world > 1 and args.train_global_batch and args.train_global_batch != args.train_batch * world:
            train_strong = training_leg(args, dev, rank, world, dist, h, w, global_batch=args.train_global_batch)
        if not args.no_extra and (args.encoder, args.decoder) == ("resnet34", "U-Net"):
            # BASELINE.json config 5: DenseNet-169 FPN training (Dice loss, data parallel), 4 images per GPU
            a5 = argparse.Namespace(**vars(args))
            a5.train_encoder, a5.decoder, a5.train_batch, a5.train_steps, a5.train_warmup = "densenet169", "FPN", 4, 3, 1
            train5 = training_leg(a5, dev, rank, world, dist, h, w, augment=True)

    result = None
    if rank == 0:
        # per-kernel timing with HIP events on the launch stream
        nops = eng.num_ops
        reps = 5
        op_ms = np.zeros(nops)
        for _ in range(reps):
            for i in range(nops):
                op_ms[i] += eng.time_ops(i, i + 1)
        op_ms /= reps
        macs = np.array([eng.op_macs(i) for i in range(nops)], dtype=np.float64)
        kernels = [eng.op_kernel(i) for i in range(nops)]
        is_conv = macs > 0
        conv_ms = op_ms[is_conv].sum()
        conv_flops = 2.0 * macs.sum()
        # group launches by device kernel (template instantiation), as rocprofv3 --stats does
        groups = {}
        for i in range(nops):
            g = groups.setdefault(kernels[i], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            g["launches"] += 1
            g["ms"] += op_ms[i]
            g["flops"] += 2.0 * macs[i]
            g["bytes"] += eng.op_bytes[i]
        dom_name = max(groups, key=lambda k: groups[k]["ms"])
        dom = groups[dom_name]
        dom_tflops = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        if args.per_op:
            for i in range(nops):
                tf = 2.0 * macs[i] / (op_ms[i] * 1e-3) / 1e12 if macs[i] else 0.0
                print("%3d %-32s %-36s %8.3f ms %8.2f GMAC %8.1f TFLOP/s" %
                      (i, eng.op_names[i], kernels[i], op_ms[i], macs[i] / 1e9, tf), file=sys.stderr)
            for k, g in sorted(groups.items(), key=lambda kv: -kv[1]["ms"]):
                print("%-40s launches %3d  total %8.3f ms  avg %8.4f ms  %8.1f TFLOP/s" %
                      (k, g["launches"], g["ms"], g["ms"] / g["launches"],
                       g["flops"] / (g["ms"] * 1e-3) / 1e12), file=sys.stderr)
            print("sum of per-op times %.3f ms (conv %.3f ms)" % (op_ms.sum(), conv_ms), file=sys.stderr)
        per_step = args.batch * args.streams
        images = per_step * world * args.steps
        enc_names = {"resnet18": "ResNet-18", "resnet34": "ResNet-34", "densenet169": "DenseNet-169"}
        result = {
            "metric": "images/sec at %dx%dx3 %s %s inference" % (h, w, enc_names.get(args.encoder, args.encoder),
                                                                  args.decoder),
            "value": images / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.storage,
            "data": "synthetic",
            "config": {"workload": "%s %s inference %dx%dx3, %d classes, random-init weights, "
                                   "%d images per GPU per step (%d concurrent plans of %d images)%s" %
                                   (args.encoder, args.decoder, h, w, args.classes, per_step, args.streams,
                                    args.batch, ", %s storage, %s residual stream" % (
                                        args.storage, "16-bit" if eng.residual_bf16 else "fp32")),
                       "images_per_step_per_gpu": per_step,
                       "parallelism": "independent images per GPU, no collective"},
            "roofline": {
                "bound": "mfma",
                "kernel": dom_name,
                "launches_per_step": dom["launches"],
                "avg_launch_ms": dom["ms"] / dom["launches"],
                "gflop_per_launch": dom["flops"] / dom["launches"] / 1e9,
                "achieved": dom_tflops,
                "timing": "HIP events on the launch stream, one plan at a time (averaged over 5 passes)",
                "peak": MFMA_BF16_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": dom_tflops / MFMA_BF16_PEAK_TFLOPS,
                "traffic": pmc_traffic(dom_name, args.batch, h, w, args.encoder, args.decoder),
                "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, KiB -> bytes), "
                                "from the committed PMC passes named in PMC_FILES",
                "algorithmic_bytes_per_launch": dom["bytes"] / dom["launches"],
                "all_conv_launches": {
                    "achieved": conv_flops / (conv_ms * 1e-3) / 1e12,
                    "frac": conv_flops / (conv_ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    "gflop_per_step": conv_flops / 1e9, "ms_per_step": conv_ms,
                    "non_conv_ms_per_step": float(op_ms[~is_conv].sum())},
            },
        }
        if dist is not None:
            result["world"] = {"world_size": dist.get_world_size(), "ranks_in_rccl_allreduce": ranks_seen,
                               "backend": dist.get_backend()}
            if rehearsal:
                result["world"]["rehearsal"] = "ranks share GPUs over gloo: the N > 1 code path only, not a measurement"
        result["roofline"]["whole_pass"] = whole_pass(args.encoder, args.decoder, h, w, images / elapsed / world,
                                                      conv_flops / args.batch / 1e9, sum(eng.op_bytes) / args.batch / 1e9,
                                                      nops)
        if train is not None:
            result["train"] = train
        if train_strong is not None:
            result["train_strong"] = train_strong
        if train5 is not None:
            result["train_configs"] = [train5]
        if not args.no_extra and (args.encoder, args.decoder) == ("resnet34", "U-Net"):
            # BASELINE.json configs 4 and 5's models, on this rank's GPU (their own lines; `value` above
            # is the headline configuration only)
            del engs, eng
            torch.cuda.empty_cache()
            result["configs"] = [
                extra_config("densenet169", "Linknet", args.classes, h, w, 8, 2, dev, args.storage),
                extra_config("densenet169", "FPN", args.classes, h, w, 4, 2, dev, args.storage),
            ]
            if args.storage == "fp16":
                # the headline configuration once more on bf16 storage (same kernels and bytes; fp32 residual stream)
                result["configs"].append(extra_config(args.encoder, args.decoder, args.classes, h, w, args.batch,
                                                      args.streams, dev, "bf16"))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.encoder, args.decoder, args.classes, h, w, weights)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
