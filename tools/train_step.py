#!/usr/bin/env python3
"""Time one TRAINING step of the stem — tcn0(gcn0(x)) in .train(), loss.backward() (train_sttran.py:185-191 restricted
to the stem) — on synthetic clips; prints one JSON line.  Optional data-parallel run under torchrun (one flat-bucket
gradient all-reduce per step).  Secondary measurement: bench.py's headline stays the eval forward.

    python tools/train_step.py [--clips 256] [--math bf16x3] [--steps 60]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench
import stgcn_amd
from stgcn_amd import dist as sd

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--frames", type=int, default=180)
ap.add_argument("--math", default="bf16x3"); ap.add_argument("--steps", type=int, default=60); ap.add_argument("--warmup", type=int, default=15)
a = ap.parse_args()
local = int(os.environ.get("LOCAL_RANK", "0"))
backend = os.environ.get("STGCN_DIST_BACKEND", "nccl")   # gloo + fewer GPUs than ranks: rehearsal on a 1-GPU box
dev = torch.device("cuda", local if backend == "nccl" else local % max(torch.cuda.device_count(), 1))
torch.cuda.set_device(dev)
rank, world = sd.init(backend, dev)
gcn, tcn = bench.build_stem(22, "SHRE", a.math)
gcn, tcn = gcn.to(dev).train(), tcn.to(dev).train()
x = bench.synthetic_clips(a.clips, a.frames, 22, seed=rank).to(dev)
G = torch.randn(a.clips, 128, a.frames, 22, device=dev)


def step():
    for p in list(gcn.parameters()) + list(tcn.parameters()):
        p.grad = None
    z = tcn(gcn(x))
    z.backward(G)                       # dL/dz handed in (a loss head would produce it)
    sd.all_reduce_grads([gcn, tcn])


for _ in range(a.warmup):
    step()
torch.cuda.synchronize(dev); sd.barrier(); torch.cuda.synchronize(dev)
t0 = time.perf_counter()
for _ in range(a.steps):
    step()
torch.cuda.synchronize(dev); sd.barrier(); torch.cuda.synchronize(dev)
el = sd.max_over_ranks(time.perf_counter() - t0, dev)
if rank == 0:
    fl = 3 * (2 * 128 * 128 * 9 * a.frames * 22)    # forward conv + dgrad + wgrad of the temporal conv (dominant terms)
    print(json.dumps({"metric": "clips/sec ST-GCN stem training step (forward + backward)", "value": round(a.clips * world * a.steps / el, 1),
                      "unit": "clips/s", "n_gpus": world, "ms_per_step": round(el / a.steps * 1e3, 3), "math": a.math,
                      "clips_per_gpu": a.clips, "T": a.frames, "V": 22,
                      "algorithmic_TFLOPs": round(fl * a.clips * world * a.steps / el / 1e12, 1)}), flush=True)
if world > 1:
    torch.distributed.destroy_process_group()
