#!/usr/bin/env python3
"""Per-wave phase clocks of the generic expansion kernel (diagnostic library):
    STGCN_LIB=.../libstgcn_hip_abl.so python tools/stamps_k2g.py [--cin 64 --cout 64 --clips 256 --frames 180]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
from stgcn_amd import unit_agcn
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=64); ap.add_argument("--cout", type=int, default=64)
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--frames", type=int, default=180)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
A = torch.rand(3, 22, 22) * (torch.rand(3, 22, 22) < 0.15)
gcn = unit_agcn(a.cin, a.cout, A).to(dev).eval()
x = torch.randn(a.clips, a.cin, a.frames, 22, device=dev)
buf = torch.zeros(8 * 8 * 8, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3): gcn(x)
    torch.cuda.synchronize()
    os.environ["STGCN_DBG_PTR"] = hex(buf.data_ptr())
    gcn(x); torch.cuda.synchronize()
t = buf.cpu().view(8, 8, 8).double()
names = ["x rows -> LDS (+ next chunk's loads issued)", "barrier waits", "aggregation u = x.P", "weight fragments (load + wait)",
         "expansion MFMAs", "epilogue (stores)", "whole kernel"]
print(f"generic expansion, {a.cin} -> {a.cout} channels, T={a.frames}, {a.clips} clips: mean over 8 workgroups x 8 waves, shader-clock ticks")
for i, nm in enumerate(names):
    print(f"  {nm:46s} {t[:, :, i].mean():10.0f}   ({100 * t[:, :, i].mean() / t[:, :, 6].mean():5.1f} %)")
