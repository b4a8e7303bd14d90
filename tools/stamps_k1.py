#!/usr/bin/env python3
"""Cycle stamps of the attention kernel's phases (workgroup 0, diagnostic library only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch, bench, stgcn_amd
dev = torch.device("cuda:0")
x = bench.synthetic_clips(256, 180, 22, 0).to(dev)
gcn, tcn = bench.build_stem(22, "SHRE", "bf16x3"); gcn, tcn = gcn.to(dev).eval(), tcn.to(dev).eval()
stgcn_amd.enable_stem_fusion(gcn, tcn)
buf = torch.zeros(2048, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3): tcn(gcn(x))
    torch.cuda.synchronize(); os.environ["STGCN_DBG_PTR"] = hex(buf.data_ptr())
    gcn(x); torch.cuda.synchronize()
t = buf.cpu()[1024:1032].tolist()
names = ["M matrices", "x -> LDS (first fill)", "Gram accumulate", "Gram reduce", "S = M.G", "soft-max + P store", "feature pass"]
for i, nm in enumerate(names): print(f"  {nm:24s} {t[i+1]-t[i]:8d} cycles")
print("  total", t[7]-t[0])
