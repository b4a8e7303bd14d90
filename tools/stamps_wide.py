#!/usr/bin/env python3
"""Per-phase cycle stamps of KF6 on the two-hand graph (diagnostic library):  STGCN_LIB=.../libstgcn_hip_abl.so python tools/stamps_wide.py"""
import os, sys
sys.path.insert(0, "st-gcn-altformer_amd"); sys.path.insert(0, ".")
import torch, bench, stgcn_amd
dev = torch.device("cuda:0")
T, V, N = 200, 46, 256
x = bench.synthetic_clips(N, T, V, 0).to(dev)
gcn, tcn = bench.build_stem(V, "LMDHG", "bf16x3")
gcn, tcn = gcn.to(dev).eval(), tcn.to(dev).eval()
stgcn_amd.enable_stem_fusion(gcn, tcn)
buf = torch.zeros(8 * 8 * 8, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3): tcn(gcn(x))
    torch.cuda.synchronize()
    os.environ["STGCN_DBG_PTR"] = hex(buf.data_ptr())
    tcn(gcn(x)); torch.cuda.synchronize()
t = buf.cpu().view(8, 8, 8).double()[:, :4, :]
tiles = N * 37 / 256
for i, nm in enumerate(["chunk0", "main loop total", "pair-end wait+barrier", "epilogue", "pairs 1,2,5,6,7,8", "pair 3", "pair 4", "pair 0"]):
    print(f"{nm:26s} {t[:, :, i].mean() / tiles:10.0f} cycles/tile")
