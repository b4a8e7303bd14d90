#!/usr/bin/env python3
"""Benchmark of the ST-GCN stem forward, tcn0(gcn0(x)), on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # N>1, launched by the driver

A step = one pass of the stem (attention kernel + fused graph-conv/temporal-conv kernel, called
through the drop-in nn.Modules) over one batch of synthetic clips already resident in HBM.  Clips are
independent, so ranks own disjoint shards (weak scaling: --clips-per-gpu per rank) and the data path
has no collective; with N>1 each step ends with one tiny RCCL all-reduce of (clip count, output
checksum sample) — the DP form of the reference's accuracy reduction (train_sttran.py:105-109).

Rank 0 prints ONE JSON line: clips/s (whole job), ms/step, `roofline` for the dominant kernel
(HIP-event timed on the launching stream) and `cpu_baseline` (the CPU oracle on the host cores).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "st-gcn-altformer_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np   # noqa: E402
import torch         # noqa: E402

HBM_PEAK = 8.0e12                       # B/s, MI355X spec (MI355X_MICROARCH.md); measured copy ~6.29e12
MFMA_PEAK = {"f32": 157.3e12, "f32_valu": 157.3e12, "bf16": 2.5e15, "bf16x3": 2.5e15}   # dense FLOP/s


def build_stem(V, graph_name, math, seed=1234):
    """gcn0/tcn0 exactly as ST_GCN_AltFormer.py:33-50 builds them; weights seeded + randomised."""
    import stgcn_amd
    from stgcn_amd.graphs import LMDHGGraph, SHREGraph
    torch.manual_seed(seed)
    G = SHREGraph if graph_name == "SHRE" else LMDHGGraph
    A = torch.from_numpy(G("spatial").A.astype(np.float32))
    gcn = stgcn_amd.unit_agcn(3, 128, A)
    tcn = stgcn_amd.Unit2D(128, 128, kernel_size=9)
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        gcn.PA.data = torch.randn(3, V, V, generator=gen) * 0.05
        for bn in (gcn.bn, gcn.down[1], tcn.bn):
            C = bn.num_features
            bn.weight.copy_(torch.rand(C, generator=gen) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=gen) * 0.2)
            bn.running_mean.copy_(torch.randn(C, generator=gen) * 0.3)
            bn.running_var.copy_(torch.rand(C, generator=gen) * 1.5 + 0.25)
        for cv in list(gcn.conv_a) + list(gcn.conv_b) + list(gcn.conv_d) + [gcn.down[0], tcn.conv]:
            cv.bias.copy_(torch.randn(cv.bias.shape, generator=gen) * 0.1)
    stgcn_amd.set_math_mode(tcn, math)
    return gcn, tcn


def synthetic_clips(n, T, V, seed):
    gen = torch.Generator().manual_seed(seed)
    skel = torch.randn(n, T, V, 3, generator=gen)                 # loader layout (N,T,V,3)
    return skel.permute(0, 3, 1, 2).contiguous()                  # ST_GCN_AltFormer.py:64-68


def cpu_baseline(gcn, tcn, T, V, clips, reps):
    """The CPU oracle (torch CPU fp32 restatement) on this host's cores; bounded sample."""
    from oracle import stgcn_oracle as so
    gp = so.agcn_params_from_state({k: v.cpu() for k, v in gcn.state_dict().items()}, gcn.A.cpu())
    tp = so.tcn_params_from_state({k: v.cpu() for k, v in tcn.state_dict().items()})
    x = synthetic_clips(clips, T, V, 0)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("STGCN_CPU_THREADS", "16"))))   # the GPU box's CPU share is 16
    torch.set_num_threads(cores)
    times = []
    with torch.no_grad():
        for i in range(2 + reps):
            t0 = time.perf_counter()
            so.stem_forward(x, gp, tp)
            dt = time.perf_counter() - t0
            if i >= 2:
                times.append(dt)
    med = statistics.median(times)
    return {"value": round(clips / med, 2), "unit": "clips/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{clips} clips (T={T},V={V}) x median of {reps} after 2 warm-ups, {med * 1e3:.0f} ms/pass, "
                      f"oracle/stgcn_oracle.py fp32 torch-CPU"}


def profiled_traffic(kernel_prefix, default_config):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary (profiles/*_counters.json,
    FETCH_SIZE x2-corrected + WRITE_SIZE, separate passes; tools/collect_profiles.sh) — only for the default
    workload those profiles were taken on; None otherwise."""
    if not default_config:
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")), reverse=True):
        try:
            for name, m in json.load(open(path)).items():
                if name.startswith(kernel_prefix) and "hbm_traffic_bytes_per_launch" in m:
                    return int(m["hbm_traffic_bytes_per_launch"])
        except (OSError, ValueError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200,
                    help="timed steps; the default is long enough (~0.2 s) for the GPU clocks to settle: 20-step runs measure the "
                         "ramp after idle and read ~12 %% low (DESIGN.md section 5)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--clips-per-gpu", type=int, default=256, help="256 = BASELINE configs[1]; 1024 = configs[4] at 8 GPUs")
    ap.add_argument("--frames", type=int, default=180)
    ap.add_argument("--graph", choices=["SHRE", "LMDHG"], default="SHRE")
    ap.add_argument("--math", choices=["f32", "bf16x3", "bf16", "f32_valu"], default=os.environ.get("STGCN_MATH", "bf16x3"))
    ap.add_argument("--no-fuse", action="store_true", help="two-stage path (intermediate through HBM)")
    ap.add_argument("--layout", choices=["nctv", "ntvc"], default="nctv",
                    help="nctv = the reference's call (contiguous (N,3,T,V) in, (N,C,T,V) out; the headline); ntvc = SURVEY "
                         "§8(f)-1 layout fusion: the loader's (N,T,V,3) batch read in place, (N,T,V,C) written")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=32)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    import stgcn_amd
    from stgcn_amd import functional as F
    stgcn_amd.lib()                                              # fail loudly before touching the GPU
    from stgcn_amd import dist as sd
    # one rank per GPU; STGCN_DIST_BACKEND=gloo + fewer GPUs than ranks is a rehearsal mode for 1-GPU boxes only
    backend = os.environ.get("STGCN_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    sd.init(backend, dev)                                        # no-op at world 1

    T, V = args.frames, 22 if args.graph == "SHRE" else 46
    n_local = args.clips_per_gpu
    gcn, tcn = build_stem(V, args.graph, args.math)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(gcn, tcn, T, V, args.cpu_clips, 5)
    gcn, tcn = gcn.to(dev).eval(), tcn.to(dev).eval()
    if not args.no_fuse:
        stgcn_amd.enable_stem_fusion(gcn, tcn)
    x = synthetic_clips(n_local, T, V, seed=rank).to(dev)        # this rank's shard, resident in HBM before timing
    if args.layout == "ntvc":
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)   # (N,T,V,3) in memory, viewed (N,3,T,V)
        stgcn_amd.set_output_layout(tcn, "channels_last")
    stats, pending = None, None

    def step():
        nonlocal stats, pending
        with torch.no_grad():
            out = tcn(gcn(x))
        if world > 1:   # tiny, latency-bound; RCCL over xGMI on its own stream, beside the next step's kernels
            stats, pending = sd.all_reduce_stats_async(sd.step_stats(out, n_local))
        return out

    def fence():
        torch.cuda.synchronize(dev)
        sd.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        out = step()
    timer = F.KernelTimer()
    F.kernel_timer = timer
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    F.kernel_timer = None
    kern_ms = timer.mean_ms("stem_tail")
    elapsed = sd.max_over_ranks(elapsed, dev)
    assert torch.isfinite(out).all()
    if pending is not None:
        pending.wait()
    if stats is not None:
        assert int(stats[0].item()) == n_local * world, "all-reduced clip count disagrees with the sharding"

    if rank == 0:
        clips_total = n_local * world * args.steps
        value = clips_total / elapsed
        bytes_clip = 4 * T * V * (3 + 128)                       # fused stem: x in + activation out
        flops_clip = 2 * 128 * 128 * 9 * T * V + 2 * T * V * (3 * 3 * V + 128 * 13)   # temporal conv + graph conv
        roof = None
        if kern_ms:
            achieved = flops_clip * n_local / (kern_ms * 1e-3)
            peak = MFMA_PEAK[args.math]
            from stgcn_amd import _capi
            v4 = args.math in ("bf16x3", "bf16") and bool(_capi.lib().stgcn_stem_features_used(3, 128, T, V, 9, 3, F._flags(
                {"bf16x3": F.MATH_BF16X3, "bf16": F.MATH_BF16}.get(args.math, 0), False)))
            kname = "stem_mfma_f32_kernel" if args.math == "f32" else ("stem_bf16_v4_kernel" if v4 else "stem_mfma_bf16_kernel")
            peak_f32 = MFMA_PEAK["f32"]
            default_cfg = n_local == 256 and T == 180 and V == 22 and not args.no_fuse and args.math in ("bf16x3", "f32")
            roof = {"bound": "mfma", "kernel": kname,
                    "issued_over_algorithmic_flops": 3 if args.math == "bf16x3" else 1,
                    "achieved": round(achieved / 1e12, 3), "peak": round(peak / 1e12, 1), "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": profiled_traffic(kname, default_cfg),
                    "achieved_over_f32_matrix_peak": round(achieved / peak_f32, 3),
                    "kernel_ms": round(kern_ms, 4), "launches_timed": timer.count("stem_tail"),
                    "algorithmic_flops_per_launch": flops_clip * n_local,
                    "algorithmic_bytes_per_launch": bytes_clip * n_local,
                    "hbm_GBps_of_kernel": round(bytes_clip * n_local / (kern_ms * 1e-3) / 1e9, 1)}
        line = {
            "metric": "clips/sec ST-GCN forward", "value": round(value, 1), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f32_valu": "f32", "bf16": "bf16",
                      "bf16x3": "bf16x3 (fp32 operands split hi+lo bf16, 3 MFMAs, fp32 accumulate)"}[args.math],
            "data": "synthetic randn clips (N,3,T,V), seeded random-init weights",
            "config": {"workload": f"SHREC'17-shape stem forward: V={V}, T={T}, {n_local} clips/GPU "
                                   f"(BASELINE configs[1] batch at 1 GPU; weak-scaled)",
                       "clips_per_gpu": n_local, "global_clips": n_local * world, "T": T, "V": V,
                       "math": args.math, "fused": not args.no_fuse, "layout": args.layout, "parallelism": f"dp{world}",
                       "parity": "1e-4 rel fp32 vs CPU oracle (tests/test_gpu_parity.py)"},
            "hbm_frac": round(value / world * bytes_clip / HBM_PEAK, 5),
            "mfma_frac": round(value / world * flops_clip / MFMA_PEAK[args.math], 4),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
