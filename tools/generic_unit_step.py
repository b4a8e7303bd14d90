#!/usr/bin/env python3
"""Training step (forward + backward with dx) of the two deeper TCN_GCN_unit shapes of the ST-TR family
(model/ST_TR/ST_TR_new.py:355-372: unit_agcn + Unit2D(k=9) [+ the unit's residual]) assembled from the drop-in modules, eager,
for timing and for rocprofv3 (no graph capture: counters are collected per dispatch).
    python tools/generic_unit_step.py [--steps 20] [--only 0|1]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
from stgcn_amd import unit_agcn, Unit2D, set_math_mode

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--only", type=int, default=-1); ap.add_argument("--clips", type=int, default=64); ap.add_argument("--frames", type=int, default=90)
ap.add_argument("--all", action="store_true", help="every unit shape of the ST-TR backbone (ST_TR_new.py:10-16), each at the frame count it sees")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = torch.rand(3, 22, 22) * (torch.rand(3, 22, 22) < 0.15)
res = {}
shapes = [(64, 64, 1, a.frames), (64, 128, 2, a.frames)]
if a.all:
    shapes += [(128, 128, 1, a.frames // 2), (128, 256, 2, a.frames // 2), (256, 256, 1, (a.frames // 2 + 1) // 2)]
for i, (cin, cout, stride, frames) in enumerate(shapes):
    if a.only >= 0 and a.only != i:
        continue
    gcn = unit_agcn(cin, cout, A.clone()).to(dev).train()
    tcn = Unit2D(cout, cout, kernel_size=9, stride=stride).to(dev).train()
    set_math_mode(tcn, "bf16x3")
    with torch.no_grad():
        gcn.bn.weight.fill_(1.0)
    x = torch.randn(a.clips, cin, frames, 22, device=dev).requires_grad_(True)

    params = list(gcn.parameters()) + list(tcn.parameters())
    gy = None

    def step():
        # optimizer.zero_grad() of a training loop (set_to_none, torch's default): without it every backward ends in one
        # accumulate-add kernel per parameter (~25 launches of 4.5 us that no training loop runs)
        for p in params:
            p.grad = None
        x.grad = None
        y = tcn(gcn(x))
        y.backward(gy if gy is not None else torch.ones_like(y))

    gy = torch.ones_like(tcn(gcn(x)).detach())

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    res[f"unit({cin},{cout},stride {stride})" + (f" T={frames}" if a.all else "")] = round((time.perf_counter() - t0) / a.steps * 1e3, 3)
print(json.dumps({"what": "TCN_GCN_unit-shaped training step, ms", "clips": a.clips, "T": a.frames, "V": 22, **res}))
