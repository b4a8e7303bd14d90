#!/usr/bin/env python3
"""Per-wave phase clocks of the generic attention kernel (diagnostic library):
    STGCN_LIB=.../libstgcn_hip_abl.so python tools/stamps_k1g.py [--cin 256 --cout 256 --clips 64 --frames 23]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
from stgcn_amd import functional as F
ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=256); ap.add_argument("--cout", type=int, default=256)
ap.add_argument("--clips", type=int, default=64); ap.add_argument("--frames", type=int, default=23)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
ic = a.cout // 4
x = torch.randn(a.clips, a.cin, a.frames, 22, device=dev)
A = torch.rand(3, 22, 22, device=dev) * 0.1
Wa, Wb = torch.randn(3, ic, a.cin, device=dev) * 0.2, torch.randn(3, ic, a.cin, device=dev) * 0.2
ba, bb = torch.randn(3, ic, device=dev) * 0.1, torch.randn(3, ic, device=dev) * 0.1
for _ in range(3): F.agcn_attention(x, A, Wa, ba, Wb, bb)
buf = torch.zeros(8 * 8 * 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
os.environ["STGCN_DBG_PTR"] = hex(buf.data_ptr())
F.agcn_attention(x, A, Wa, ba, Wb, bb); torch.cuda.synchronize()
t = buf.cpu().view(8, 8, 8).double()
names = ["staging", "barrier waits", "embeddings", "Gram", "fragments: issue of the next phase's + wait for this one's (+ tail)", "whole kernel"]
print(f"generic attention, Cin={a.cin} inter_c={ic} T={a.frames} clips={a.clips}: mean over 8 workgroups x 8 waves, shader-clock ticks")
for i, nm in enumerate(names):
    print(f"  {nm:40s} {t[:, :, i].mean():10.0f}   ({100 * t[:, :, i].mean() / t[:, :, 5].mean():5.1f} %)")
