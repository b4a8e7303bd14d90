#!/usr/bin/env python3
"""In-kernel clock stamps of the one-wave-per-SIMD weight-gradient kernel (diagnostic library only).

    STGCN_LIB=.../libstgcn_hip_abl.so python tools/stamps_wgrad.py [--groups]

Per wave, summed over its units: slot 9 = the unit's work (MFMA groups + fillers), 10 = wait at the unit's barrier;
with --groups (STGCN_ABLATE=64) slots 0..8 = the nine MFMA groups (each stamp drains the LDS queue: perturbs the schedule).
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
from stgcn_amd import functional as F
ap = argparse.ArgumentParser(); ap.add_argument("--groups", action="store_true"); ap.add_argument("--clips", type=int, default=256)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(a.clips, 128, 180, 22, device=dev).relu_()
W = torch.randn(128, 128, 9, device=dev) * 0.03
b = torch.randn(128, device=dev) * 0.1
bn = (torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.1, torch.zeros(128, device=dev), torch.ones(128, device=dev))
y, z, mean, inv = F.tcn_forward_train(x, W, b, bn, math=F.MATH_BF16X3, save=True)
dy = torch.randn_like(y)
run = lambda: F.tcn_backward_train(x, W, z, bn[0], bn[1], mean, inv, dy, math=F.MATH_BF16X3, need_dx=False)
for _ in range(3): run()
torch.cuda.synchronize()
buf = torch.zeros(8 * 4 * 12, dtype=torch.int64, device=dev)
os.environ["STGCN_DBG_PTR"] = hex(buf.data_ptr())
if a.groups: os.environ["STGCN_ABLATE"] = "64"
run(); torch.cuda.synchronize()
t = buf.cpu().view(8, 4, 12).double()
units = 2 * (90 + 4)
work, wait = t[:, :, 9], t[:, :, 10]
print(f"per unit (s_memtime ticks; {units} units per wave): work mean {work.mean() / units:.0f}  barrier wait mean {wait.mean() / units:.0f} "
      f"({100 * wait.mean() / (work.mean() + wait.mean()):.1f} %)   per-wave wait min {wait.min() / units:.0f} max {wait.max() / units:.0f}")
for w in range(4):
    print(f"  wave {w}: work {work[:, w].mean() / units:.0f}  wait {wait[:, w].mean() / units:.0f}")
if a.groups:
    for s in range(9):
        print(f"  group {s} (k-step {s // 3}, taps {3 * (s % 3)}..{3 * (s % 3) + 2}): {t[:, :, s].mean() / units:.0f} per unit")
