#!/usr/bin/env python3
"""Eval-mode forward of the deeper layers' unit_agcn (K1g attention + K2g expansion) at the DESIGN section 7 shapes, HIP events:
    python tools/time_generic_forward.py [--clips 256]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
from stgcn_amd import unit_agcn

ap = argparse.ArgumentParser(); ap.add_argument("--clips", type=int, default=256); ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0"); torch.manual_seed(0)
A = torch.rand(3, 22, 22) * (torch.rand(3, 22, 22) < 0.15)
res = {}
for cin, cout, T in ((64, 64, 180), (128, 128, 90), (256, 256, 45)):
    gcn = unit_agcn(cin, cout, A.clone()).to(dev).eval()
    x = torch.randn(a.clips, cin, T, 22, device=dev)
    with torch.no_grad():
        for _ in range(5):
            gcn(x)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            gcn(x)
        e1.record(); torch.cuda.synchronize()
    # FLOP bound of the pair: embeddings + Gram (K1g) + aggregation + expansion (K2g), fp32 matrix peak 157.3 TF
    ic, P = cout // 4, T * 22
    fl = a.clips * (2 * 3 * 2 * ic * cin * P + 2 * 3 * 22 * 22 * ic * T + 2 * 3 * cin * T * 22 * 22 + 2 * cout * 3 * cin * P)
    ms = e0.elapsed_time(e1) / a.reps
    res[f"unit_agcn({cin},{cout}) T={T}"] = {"ms": round(ms, 3), "flop_bound_ms": round(fl / 157.3e12 * 1e3, 3), "ratio": round(ms / (fl / 157.3e12 * 1e3), 2)}
print(json.dumps({"what": "eval forward of the deeper layers' unit_agcn (attention + expansion)", "clips": a.clips, **res}))
