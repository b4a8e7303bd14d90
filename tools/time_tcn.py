#!/usr/bin/env python3
"""Time the stand-alone temporal conv (Unit2D eval forward, bf16x3) per STGCN_ABLATE mask in one process.
    STGCN_LIB=.../libstgcn_hip_abl.so python tools/time_tcn.py [--cin 128 --cout 128 --clips 256 --frames 180] [--masks 0,8192]
(diagnostic library: mask 8192 = the eight-wave kernel K3v4 instead of K3v6)"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch, stgcn_amd
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=128); ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--frames", type=int, default=180); ap.add_argument("--joints", type=int, default=22)
ap.add_argument("--masks", default="0,8192"); ap.add_argument("--math", default="bf16x3")
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
m = stgcn_amd.Unit2D(a.cin, a.cout, kernel_size=9).to(dev).eval()
stgcn_amd.set_math_mode(m, a.math)
x = torch.randn(a.clips, a.cin, a.frames, a.joints, device=dev).relu_()
masks = [int(v) for v in a.masks.split(",")]
res = {k: [] for k in masks}; outs = {}
with torch.no_grad():
    for _ in range(3): m(x)
    for _ in range(4):
        for k in masks:
            os.environ["STGCN_ABLATE"] = str(k)
            outs[k] = m(x); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): m(x)
            e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) / 30 * 1e3)
for k in masks:
    print(f"mask {k:5d}: min {min(res[k]):8.1f} us  (rounds {[round(v, 1) for v in res[k]]})")
if len(masks) > 1:
    print("max |difference| between the first two:", float((outs[masks[0]] - outs[masks[1]]).abs().max()))
