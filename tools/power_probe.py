#!/usr/bin/env python3
"""Is the fused stem kernel clock/power-limited?  Times it on random data, on zero clips and on zero clips +
zero temporal-conv weights (same instruction stream; the matrix cores draw less power on zeros, so a
power-limited kernel gets faster, an issue/latency-limited one does not).  HIP events, fused kernel only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench
import stgcn_amd
from stgcn_amd import functional as F

dev = torch.device("cuda:0")
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = bench.synthetic_clips(clips, 180, 22, 0).to(dev)


def run(xx, zero_w):
    gcn, tcn = bench.build_stem(22, "SHRE", "bf16x3")
    if zero_w:
        with torch.no_grad():
            tcn.conv.weight.zero_()
    gcn, tcn = gcn.to(dev).eval(), tcn.to(dev).eval()
    stgcn_amd.enable_stem_fusion(gcn, tcn)
    with torch.no_grad():
        for _ in range(20):
            tcn(gcn(xx))
        t = F.KernelTimer(); F.kernel_timer = t
        for _ in range(200):
            tcn(gcn(xx))
        torch.cuda.synchronize()
        F.kernel_timer = None
    return t.mean_ms("stem_tail")


for name, xx, zw in [("random clips, random weights", x, False), ("zero clips, random weights", torch.zeros_like(x), False),
                     ("zero clips, zero conv weights", torch.zeros_like(x), True), ("random clips, zero conv weights", x, True),
                     ("random clips, random weights", x, False)]:
    print(f"{name:34s}: fused kernel {run(xx, zw):.4f} ms", flush=True)
