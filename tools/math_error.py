#!/usr/bin/env python3
"""Actual error of each arithmetic mode of the fused stem against the fp64 oracle (log the bf16 mode's
max error before considering a cheaper two-term variant).   python tools/math_error.py   (needs a GPU; oracle = checker)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_gpu_parity import _random_stem
from stgcn_amd import enable_stem_fusion, set_math_mode
from oracle import stgcn_oracle as so

dev = torch.device("cuda:0")
print("mode      max|err|/max|ref|  rms(err)/rms(ref)   (fused stem, 8 clips, T=180, V=22, three seeded random stems)")
for math in ("f32", "bf16x3", "bf16"):
    worst, rms = 0.0, 0.0
    for seed in (1, 2, 3):
        gcn, tcn, gp, tp, gen = _random_stem(22, None, seed, dev)
        set_math_mode(tcn, math)
        enable_stem_fusion(gcn, tcn)
        x = torch.randn(8, 3, 180, 22, generator=gen)
        ref = so.stem_forward(x.double(), gp.to(torch.float64), tp.to(torch.float64))
        with torch.no_grad():
            z = tcn(gcn(x.to(dev))).double().cpu()
        worst = max(worst, ((z - ref).abs().max() / ref.abs().max()).item())
        rms = max(rms, ((z - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    print(f"{math:8s}  {worst:.3e}          {rms:.3e}")
