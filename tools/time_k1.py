#!/usr/bin/env python3
"""Time the attention kernel (K1) alone, with and without the feature pass (HIP events, 20 launches each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
from ctypes import c_int, c_size_t, c_uint, c_void_p
import torch
import bench
from stgcn_amd import _capi, functional as F

dev = torch.device("cuda:0")
for clips in (256, 1024):
    x = bench.synthetic_clips(clips, 180, 22, 0).to(dev)
    gcn, tcn = bench.build_stem(22, "SHRE", "bf16x3")
    gcn = gcn.to(dev).eval()
    st = gcn._staged(dev)
    lib = _capi.lib()
    def run(flags):
        need = lib.stgcn_stem_ws_bytes(clips, 3, 128, 180, 22, 9, 3, flags)
        ws = torch.empty(need // 4 + 1, device=dev)
        p = lambda t: c_void_p(t.data_ptr())
        s = c_void_p(torch.cuda.current_stream().cuda_stream)
        call = lambda: _capi.call("stgcn_stem_attention", p(x), p(st["A_eff"]), p(st["Wa"]), p(st["ba"]), p(st["Wb"]),
                                  p(st["bb"]), p(ws), c_size_t(ws.numel() * 4), c_int(clips), c_int(3), c_int(128), c_int(180),
                                  c_int(22), c_int(32), c_int(3), c_int(9), c_uint(flags), s)
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / 20
    print(f"clips={clips}: K1 without features {run(0) * 1e3:.1f} us, with features {run(1) * 1e3:.1f} us")
