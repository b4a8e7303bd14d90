#!/usr/bin/env python3
"""Time the adjacency kernel of a deeper unit_agcn layer (generic C_in); with the diagnostic library STGCN_ABLATE=2048
selects the VALU kernel for a same-process A/B.
    STGCN_LIB=.../libstgcn_hip_abl.so python tools/time_generic_attention.py [--cin 64 --cout 64 --clips 256 --frames 180]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
from stgcn_amd import functional as F
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=64); ap.add_argument("--cout", type=int, default=64)
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--frames", type=int, default=180); ap.add_argument("--joints", type=int, default=22)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
ic = a.cout // 4
x = torch.randn(a.clips, a.cin, a.frames, a.joints, device=dev)
A = torch.rand(3, a.joints, a.joints, device=dev) * 0.1
Wa, Wb = torch.randn(3, ic, a.cin, device=dev) * 0.2, torch.randn(3, ic, a.cin, device=dev) * 0.2
ba, bb = torch.randn(3, ic, device=dev) * 0.1, torch.randn(3, ic, device=dev) * 0.1
run = lambda: F.agcn_attention(x, A, Wa, ba, Wb, bb)
res = {}
for m in ("0", "2048"):
    os.environ["STGCN_ABLATE"] = m
    for _ in range(3): P = run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    res[m] = (e0.elapsed_time(e1) / 20 * 1e3, P)
print(f"matrix cores: {res['0'][0]:.1f} us   VALU kernel (diagnostic library only): {res['2048'][0]:.1f} us   max |dP| {float((res['0'][1]-res['2048'][1]).abs().max()):.2e}")
