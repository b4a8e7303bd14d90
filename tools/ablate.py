#!/usr/bin/env python3
"""Price the phases of the fused stem kernel with the diagnostic library (build.py --ablation).

    STGCN_LIB=st-gcn-altformer_amd/stgcn_amd/libstgcn_hip_abl.so python tools/ablate.py [--clips 256]

For every arithmetic mode and every STGCN_ABLATE mask it times the fused kernel alone (HIP events on the
launching stream, interleaved rounds in one process).  Masks of the eight-wave kernels (KF4, f32): 1 producer, 2 MFMAs,
4 epilogue stores, 8 B-operand LDS reads, 16 weight loads — outputs are wrong when one of these is set.  Kernel
selection masks (results stay correct): 256 = KF4 instead of KF6 (the same-process yardstick of DESIGN.md section 3),
8192 = K3v4 instead of K3v6 (tools/time_tcn.py), 2048 / 4096 = the plain-FMA generic adjacency / expansion kernels.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--frames", type=int, default=180)
    ap.add_argument("--graph", default="SHRE")
    ap.add_argument("--maths", default="f32,bf16x3,bf16")
    ap.add_argument("--masks", default="0,1,2,4,3,7,8,16,24,25")   # bit 32 + (n << 8): stagger by n x s_sleep(127)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    import stgcn_amd
    from stgcn_amd import functional as F
    dev = torch.device("cuda:0")
    V = 22 if args.graph == "SHRE" else 46
    x = bench.synthetic_clips(args.clips, args.frames, V, 0).to(dev)
    masks = [int(m) for m in args.masks.split(",")]
    for math in args.maths.split(","):
        gcn, tcn = bench.build_stem(V, args.graph, math)
        gcn, tcn = gcn.to(dev).eval(), tcn.to(dev).eval()
        stgcn_amd.enable_stem_fusion(gcn, tcn)
        res = {m: [] for m in masks}
        with torch.no_grad():
            for _ in range(3):
                tcn(gcn(x))
            for _ in range(args.rounds):
                for m in masks:
                    os.environ["STGCN_ABLATE"] = str(m)
                    t = F.KernelTimer()
                    F.kernel_timer = t
                    for _ in range(args.iters):
                        tcn(gcn(x))
                    res[m].append(t.mean_ms("stem_tail"))
                    F.kernel_timer = None
        os.environ["STGCN_ABLATE"] = "0"
        base = min(res[0])
        print(f"== {math}: clips={args.clips} T={args.frames} V={V}")
        for m in masks:
            print(f"  mask {m:3d}: min {min(res[m]):8.4f} ms  med {sorted(res[m])[len(res[m]) // 2]:8.4f} ms  "
                  f"({min(res[m]) / base:5.2f}x of full)")


if __name__ == "__main__":
    main()
