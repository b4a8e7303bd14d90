#!/usr/bin/env python3
"""In-kernel cycle stamps of the v4 stem kernel (diagnostic library only).

    STGCN_LIB=.../libstgcn_hip_abl.so python tools/stamps.py [--math bf16x3] [--abl MASK]

Slots per wave (shader cycles summed over the wave's tiles): 0 = chunk-0 production + barrier, 1 = stage compute
(arrival at the stage barrier), 2 = stage barrier wait (vmcnt(0) + s_barrier), 3 = epilogue, 4 = next-tile fix-up.
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--math", default="bf16x3"); ap.add_argument("--abl", default="0")
ap.add_argument("--clips", type=int, default=256)
a = ap.parse_args()
import stgcn_amd
dev = torch.device("cuda:0")
x = bench.synthetic_clips(a.clips, 180, 22, 0).to(dev)
gcn, tcn = bench.build_stem(22, "SHRE", a.math)
gcn, tcn = gcn.to(dev).eval(), tcn.to(dev).eval()
stgcn_amd.enable_stem_fusion(gcn, tcn)
buf = torch.zeros(8 * 8 * 8, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3): tcn(gcn(x))
    torch.cuda.synchronize()
    os.environ["STGCN_ABLATE"] = a.abl
    os.environ["STGCN_DBG_PTR"] = hex(buf.data_ptr())
    tcn(gcn(x)); torch.cuda.synchronize()
t = buf.cpu().view(8, 8, 8).double()
names = ["chunk0+bar", "main loop (KF6: incl. barrier waits; KF4: stage compute)", "stage barrier wait", "epilogue", "post-epilogue (features + bar)"]
tot = t[:, :, :5].sum(-1)
print(f"math={a.math} abl={a.abl}: per-wave total stamped cycles: mean {tot.mean():.0f}")
for i, nm in enumerate(names):
    v = t[:, :, i]
    print(f"  {nm:20s} mean {v.mean():10.0f}  ({100 * v.mean() / tot.mean():5.1f} %)  min {v.min():10.0f} max {v.max():10.0f}")
print(f"  of the post-epilogue part: barrier wait behind the epilogue {t[:, :, 5].mean():.0f}, feature phase {t[:, :, 6].mean():.0f}")
print("per-wave (workgroup 0): stage compute / barrier wait")
for w in range(8):
    print(f"  wave {w}: compute {t[0, w, 1]:10.0f}  wait {t[0, w, 2]:10.0f}  epilogue {t[0, w, 3]:8.0f}  post-epi bar {t[0, w, 5]:8.0f}  features {t[0, w, 6]:8.0f}  chunk0 {t[0, w, 0]:8.0f}")
