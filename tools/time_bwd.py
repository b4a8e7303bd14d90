#!/usr/bin/env python3
"""Time the training-mode graph-conv backward (stem class) alone, per STGCN_ABLATE mask, in one process.

    [STGCN_LIB=.../libstgcn_hip_abl.so] python tools/time_bwd.py [--clips 256] [--masks 0,1,2,4,8,16]

Gather-kernel masks (diagnostic library only): 1 global loads, 2 h FMAs, 4 Gram MFMAs, 8 h atomics, 16 u computation.
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from stgcn_amd import functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--frames", type=int, default=180)
ap.add_argument("--masks", default="0"); ap.add_argument("--iters", type=int, default=20); ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
gcn, tcn = bench.build_stem(22, "SHRE", "bf16x3")
gcn = gcn.to(dev).train()
x = bench.synthetic_clips(a.clips, a.frames, 22, 0).to(dev)
st = gcn._staged(dev)
bn, d = gcn.bn, gcn.down[1]
y, P, zm, zd, stats = F.agcn_forward_train(x, st["A_eff"], st["Wa"], st["ba"], st["Wb"], st["bb"], st["Wd"], st["bd"], st["Wdown"], st["bdown"],
                                           (bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var),
                                           (d.weight.detach(), d.bias.detach(), d.running_mean, d.running_var), 0.1, 1e-5, save=True)
dy = torch.randn_like(y)


def run():
    return F.agcn_backward_train(x, st["A_eff"], st["Wa"], st["ba"], st["Wb"], st["bb"], st["Wd"], st["bd"], st["Wdown"], st["bdown"], P, None, None,
                                 bn.weight.detach(), bn.bias.detach(), d.weight.detach(), d.bias.detach(), stats, dy, y=y)


masks = [int(m) for m in a.masks.split(",")]
res = {m: [] for m in masks}
for _ in range(5):
    run()
for _ in range(a.rounds):
    for m in masks:
        os.environ["STGCN_ABLATE"] = str(m)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record(); torch.cuda.synchronize()
        res[m].append(e0.elapsed_time(e1) / a.iters * 1e3)
for m in masks:
    print(f"mask {m:3d}: {min(res[m]):8.1f} us  (rounds {[round(v, 1) for v in res[m]]})")
