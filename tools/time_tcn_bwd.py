#!/usr/bin/env python3
"""Time the training-mode Unit2D backward (128->128, K=9) alone; A/B of env toggles in one process.
    STGCN_LIB=.../libstgcn_hip_abl.so python tools/time_tcn_bwd.py   (STGCN_ABLATE=1: wgrad without the XCD-aware block numbering)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
from stgcn_amd import functional as F
ap = argparse.ArgumentParser(); ap.add_argument("--env", default="STGCN_ABLATE"); ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--need-dx", type=int, default=0); ap.add_argument("--val", default="1")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(a.clips, 128, 180, 22, device=dev).relu_()
W = torch.randn(128, 128, 9, device=dev) * 0.03
b = torch.randn(128, device=dev) * 0.1
bn = (torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1, torch.zeros(128, device=dev), torch.ones(128, device=dev))
y, z, mean, inv = F.tcn_forward_train(x, W, b, bn, math=F.MATH_BF16X3, save=True)
dy = torch.randn_like(y)
run = lambda: F.tcn_backward_train(x, W, z, bn[0], bn[1], mean, inv, dy, math=F.MATH_BF16X3, need_dx=bool(a.need_dx))
for _ in range(3): run()
res = {0: [], 1: []}
for _ in range(4):
    for on in (0, 1):  # 1 -> STGCN_ABLATE set to --val
        if on: os.environ[a.env] = a.val
        else: os.environ.pop(a.env, None)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): run()
        e1.record(); torch.cuda.synchronize()
        res[on].append(e0.elapsed_time(e1) / 10 * 1e3)
print(f"{a.env} unset: {min(res[0]):.1f} us   set: {min(res[1]):.1f} us   (backward without dx = stats + apply + wgrad)")
