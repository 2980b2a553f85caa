"""Inference throughput with S concurrent plans (one per HIP stream) of B images each.
usage: stream_bench.py --streams S --batch B [--steps K]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import miseg_amd
from miseg_amd import engine, netspec
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--batch", type=int, default=2)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--encoder", default="resnet34")
ap.add_argument("--decoder", default="U-Net")
a = ap.parse_args()
dev = torch.device("cuda:0")
h, w = 1920, 2560
graph = netspec.build_graph(a.encoder, a.decoder, (h, w, 3), 3)
weights = bench.build_weights(graph)
engs = [engine.Engine(graph, weights, batch=a.batch, device=str(dev)) for _ in range(a.streams)]
streams = [torch.cuda.Stream(dev) for _ in range(a.streams)]
x = torch.from_numpy(bench.synthetic_images(a.batch, h, w, seed=1234)).to(dev)
for e in engs:
    e.x_in.copy_(x)
torch.cuda.synchronize()
def step():
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.run()
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("streams %d batch %d: %.2f ms/step  %.1f images/s" % (a.streams, a.batch, 1e3 * dt / a.steps,
      a.streams * a.batch * a.steps / dt))
