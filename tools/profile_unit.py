#!/usr/bin/env python3
"""Eval forward of one deeper TCN_GCN_unit's pieces (unit_agcn(cin,cout) + Unit2D(cout,cout,9)) for rocprofv3.
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/profile_unit.py --cin 64 --cout 64"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch, stgcn_amd
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=64); ap.add_argument("--cout", type=int, default=64); ap.add_argument("--stride", type=int, default=1)
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--frames", type=int, default=180); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
A = torch.rand(3, 22, 22) * 0.1
gcn = stgcn_amd.unit_agcn(a.cin, a.cout, A).to(dev).eval()
tcn = stgcn_amd.Unit2D(a.cout, a.cout, kernel_size=9, stride=a.stride).to(dev).eval()
stgcn_amd.set_math_mode(tcn, "bf16x3")
x = torch.randn(a.clips, a.cin, a.frames, 22, device=dev)
with torch.no_grad():
    for _ in range(a.iters):
        y = tcn(gcn(x))
torch.cuda.synchronize()
print("ok", tuple(y.shape))
