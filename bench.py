#!/usr/bin/env python3
"""Benchmark of the ST-GCN stem forward, tcn0(gcn0(x)), on N MI355X (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W]            # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # N>1, launched by the driver

A step = one pass of the stem (attention kernel + fused graph-conv/temporal-conv kernel, called
through the drop-in nn.Modules) over this rank's synthetic clips, already resident in HBM.  Clips are
independent, so ranks own disjoint shards and the data path has no collective; with N>1 each step ends
with one tiny RCCL all-reduce of (clip count, output checksum sample) — the DP form of the reference's
accuracy reduction (train_sttran.py:105-109).

Two workloads; the line's `value` is the one BASELINE.json quotes for that GPU count, the other kind is measured in the
same process and reported beside it (`other_scaling`), so both scaling kinds are on every line:
  N = 1 default          BASELINE configs[1]: --clips-per-gpu (256) clips, "scaling": "weak";
  N > 1 default          BASELINE configs[4]: ONE batch of 8192 clips sharded over the ranks
                         (stgcn_amd.dist.shard_bounds), "scaling": "strong"; a rank walks its shard in sub-batches of
                         at most --sub-batch clips (512: features + output of one sub-batch stay inside the
                         256 MB Infinity Cache), so a step is ceil(shard / sub-batch) stem passes;
  --weak                 force the weak-scaling workload at N > 1 (256 clips on every rank);
  --global-clips G       force the strong-scaling workload with G clips (also at N = 1: 8192 = configs[4]'s base point).

Rank 0 prints ONE JSON line.  Besides the contract's fields:
  roofline      the dominant kernel, HIP-event timed on the launching stream over the timed region;
                `traffic` (HBM bytes per launch from the committed rocprofv3 PMC summary) is quoted only when
                the kernels' sources are byte-identical to the ones that profile was taken on (`traffic_source`);
  steady_state  a second, longer timed block in the same process (the driver's 20 steps run while the
                clocks still ramp after idle; this block shows where the figure settles);
  alt           a short block with the exact-fp32 arithmetic (STGCN_MATH_F32) and its own roofline;
  train         secondary: one training step of the stem (forward + backward in .train()), short block;
  cpu_baseline  the stem on torch's library CPU ops (oneDNN conv etc., oracle/stgcn_cpu_ops.py: the op mix the
                reference itself runs), rank 0 at N=1 only; `cpu_oracle` = the einsum oracle timed the same way.
Order inside the process: headline block, steady-state block, fp32 block, training block, CPU legs (the GPU is never left
idle behind a CPU leg before a GPU measurement).
"""
import argparse
import hashlib
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
MFMA_PEAK = {"f32": 157.3e12, "f32_valu": 157.3e12, "bf16": 2.5e15, "bf16x3": 2.5e15, "f16mx": 2.5e15}   # dense FLOP/s (16-bit rate)
DTYPE_NAME = {"f32": "f32", "f32_valu": "f32", "bf16": "bf16",
              "bf16x3": "bf16x3 (fp32 operands split hi+lo bf16, 3 MFMAs, fp32 accumulate)",
              "f16mx": "f16mx (fp32 operands as fp16 + two e4m3 residual products on the scaled MFMA, fp32 accumulate)"}


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


def _host_cores():
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    return max(1, min(cores, int(os.environ.get("STGCN_CPU_THREADS", "16"))))   # the GPU box's CPU share is 16


def cpu_legs(gcn_state, tcn_state, A, T, V, clips, reps):
    """Bounded CPU timings on this host's cores: (library-op baseline, einsum oracle).  Both compute the same
    stem (tests/test_oracle_golden.py::test_cpu_ops_*); the first is what the reference's own CPU forward costs."""
    from oracle import stgcn_cpu_ops as co
    from oracle import stgcn_oracle as so
    gp = so.agcn_params_from_state(gcn_state, A)
    tp = so.tcn_params_from_state(tcn_state)
    x = synthetic_clips(clips, T, V, 0)
    torch.set_num_threads(_host_cores())

    def timed(fn, what):
        times = []
        with torch.no_grad():
            for i in range(2 + reps):
                t0 = time.perf_counter()
                fn(x, gp, tp)
                dt = time.perf_counter() - t0
                if i >= 2:
                    times.append(dt)
        med = statistics.median(times)
        return {"value": round(clips / med, 2), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{clips} clips (T={T},V={V}) x median of {reps} after 2 warm-ups, {med * 1e3:.0f} ms/pass, {what}"}

    base = timed(co.stem_forward_ops, "oracle/stgcn_cpu_ops.py: fp32 torch-CPU library ops (F.conv2d / F.batch_norm / matmul, "
                                      "the reference's op mix)")
    orc = timed(so.stem_forward, "oracle/stgcn_oracle.py: fp32 torch-CPU einsum restatement (the parity checker)")
    return base, orc


def csrc_digest():
    """Content hash of every kernel source + the ABI header: identifies the binary a profile belongs to (the GPU box
    has no .git).  tools/summarize_profiles.py stores the same digest next to the counters it condenses."""
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "st-gcn-altformer_amd", "csrc")
    files = sorted(os.path.join(cs, f) for f in os.listdir(cs) if f.endswith((".hip", ".h")))
    for path in files + [os.path.join(ROOT, "include", "stgcn_hip.h")]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic(kernel_prefix, workload_key):
    """(bytes per launch, source) of the dominant kernel from the newest committed rocprofv3 PMC summary
    (profiles/*_counters.json: FETCH_SIZE x2-corrected + WRITE_SIZE, separate passes, tools/collect_profiles.sh) —
    only when that profile was taken on this workload AND on these kernel sources; (None, reason) otherwise."""
    import glob
    digest = csrc_digest()
    reason = "no PMC summary under profiles/"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        meta = doc.get("_meta", {})
        if meta.get("workload") != workload_key:
            reason = f"newest PMC summaries are of another workload ({meta.get('workload')!r})"
            continue
        if meta.get("csrc_digest") != digest:
            reason = (f"{os.path.basename(path)} was taken on csrc {meta.get('csrc_digest')}, this binary is {digest}: "
                      "not quoted")
            continue
        for name, m in doc.items():
            if name.startswith(kernel_prefix) and "hbm_traffic_bytes_per_launch" in m:
                return int(m["hbm_traffic_bytes_per_launch"]), f"profiles/{os.path.basename(path)}@csrc:{digest}"
    return None, reason


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--clips-per-gpu", type=int, default=256, help="weak-scaling shard size; 256 = BASELINE configs[1]")
    ap.add_argument("--global-clips", type=int, default=None,
                    help="strong scaling: ONE batch of this many clips sharded over the ranks (8192 = BASELINE configs[4]); "
                         "default: 8192 when --gpus > 1 (the config BASELINE.json quotes for the scaling curve), weak at 1 GPU")
    ap.add_argument("--weak", action="store_true", help="weak scaling (--clips-per-gpu on every rank) also at --gpus > 1")
    ap.add_argument("--other-steps", type=int, default=10,
                    help="timed steps of the block that measures the OTHER scaling kind in the same process (0 = skip)")
    ap.add_argument("--sub-batch", type=int, default=512, help="largest number of clips per stem pass in --global-clips mode")
    ap.add_argument("--frames", type=int, default=180)
    ap.add_argument("--graph", choices=["SHRE", "LMDHG"], default="SHRE")
    ap.add_argument("--math", choices=["f32", "bf16x3", "bf16", "f32_valu", "f16mx"], default=os.environ.get("STGCN_MATH", "bf16x3"))
    ap.add_argument("--no-fuse", action="store_true", help="two-stage path (intermediate through HBM)")
    ap.add_argument("--layout", choices=["nctv", "ntvc"], default="nctv",
                    help="nctv = the reference's call (contiguous (N,3,T,V) in, (N,C,T,V) out; the headline); ntvc = SURVEY "
                         "§8(f)-1 layout fusion: the loader's (N,T,V,3) batch read in place, (N,T,V,C) written")
    ap.add_argument("--steady-steps", type=int, default=300, help="length of the second timed block (0 = skip)")
    ap.add_argument("--alt-steps", type=int, default=40, help="timed steps of the exact-fp32 block (0 = skip)")
    ap.add_argument("--train-steps", type=int, default=30,
                    help="timed steps of the secondary training block: forward + backward of the stem in .train() (0 = skip)")
    ap.add_argument("--deeper-steps", type=int, default=20,
                    help="timed steps of the tertiary block: training step of the two deeper TCN_GCN_unit shapes (0 = skip; 1 GPU only)")
    ap.add_argument("--no-extras", action="store_true", help="headline block only (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-clips", type=int, default=64)
    args = ap.parse_args()
    if args.no_extras:
        args.steady_steps = args.alt_steps = args.train_steps = args.other_steps = args.deeper_steps = 0
        args.no_cpu_baseline = True

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
    if args.weak and args.global_clips:
        raise SystemExit("--weak and --global-clips exclude each other")
    if args.global_clips is None:
        args.global_clips = 0 if (args.weak or world == 1) else 8192
    strong = args.global_clips > 0
    gcn, tcn = build_stem(V, args.graph, args.math)
    cpu_state = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_state = ({k: v.clone() for k, v in gcn.state_dict().items()},
                     {k: v.clone() for k, v in tcn.state_dict().items()}, gcn.A.clone())
    gcn, tcn = gcn.to(dev).eval(), tcn.to(dev).eval()
    if not args.no_fuse:
        stgcn_amd.enable_stem_fusion(gcn, tcn)
    if args.layout == "ntvc":
        stgcn_amd.set_output_layout(tcn, "channels_last")

    def make_workload(strong_, global_clips):
        """This rank's clips, resident in HBM before any timing, cut into sub-batches (one, in the weak workload)."""
        if strong_:
            lo, hi = sd.shard_bounds(global_clips, rank, world)
            n_loc, n_glob, seed0 = hi - lo, global_clips, 1000
        else:
            n_loc, n_glob, lo, seed0 = args.clips_per_gpu, args.clips_per_gpu * world, 0, rank
        sub_ = min(args.sub_batch, n_loc) if strong_ else n_loc
        pieces = []
        for s_ in range(0, n_loc, max(sub_, 1)):
            n = min(sub_, n_loc - s_)
            # strong: seeded by the sub-batch's position in the global batch (the batch is the same at every world size)
            xs = synthetic_clips(n, T, V, seed=seed0 + (lo + s_) if strong_ else seed0).to(dev)
            if args.layout == "ntvc":
                xs = xs.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)   # (N,T,V,3) in memory, viewed (N,3,T,V)
            pieces.append(xs)
        return {"strong": strong_, "shard": pieces, "n_local": n_loc, "n_global": n_glob, "sub": sub_}

    wl = make_workload(strong, args.global_clips)
    shard, n_local, n_global, sub = wl["shard"], wl["n_local"], wl["n_global"], wl["sub"]
    launches_per_step = len(shard)
    stats, pending = None, None
    current = wl

    def step():
        nonlocal stats, pending
        out = None
        with torch.no_grad():
            for xs in current["shard"]:
                out = tcn(gcn(xs))
        if world > 1:   # tiny, latency-bound; RCCL over xGMI on its own stream, beside the next step's kernels
            stats, pending = sd.all_reduce_stats_async(sd.step_stats(out, current["n_local"]))
        return out

    def fence():
        torch.cuda.synchronize(dev)
        sd.barrier()
        torch.cuda.synchronize(dev)

    def timed_block(steps, warm):
        """`warm` untimed steps, then exactly `steps` steps between two fences; (seconds max over ranks, kernel ms)."""
        out = None
        for _ in range(warm):
            out = step()
        timer = F.KernelTimer()
        F.kernel_timer = timer
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        fence()
        dt = time.perf_counter() - t0
        F.kernel_timer = None
        return sd.max_over_ranks(dt, dev), timer.mean_ms("stem_tail"), timer.count("stem_tail"), out

    elapsed, kern_ms, n_launch, out = timed_block(args.steps, args.warmup)
    assert torch.isfinite(out).all()
    if pending is not None:
        pending.wait()
    if stats is not None:
        assert int(round(stats[0].item())) == n_global, "all-reduced clip count disagrees with the sharding"

    bytes_clip = 4 * T * V * (3 + 128)                       # fused stem: x in + activation out
    flops_clip = 2 * 128 * 128 * 9 * T * V + 2 * T * V * (3 * 3 * V + 128 * 13)   # temporal conv + graph conv
    clips_per_launch = n_local / launches_per_step

    def kernel_name(math):
        from stgcn_amd import _capi
        if args.no_fuse:
            return "tcn_bf16_v6_kernel" if math in ("bf16x3", "bf16", "f16mx") else "tcn_mfma_f32_kernel"
        fl = F._flags({"f32": F.MATH_F32, "bf16x3": F.MATH_BF16X3, "bf16": F.MATH_BF16, "f32_valu": F.MATH_F32_VALU,
                       "f16mx": F.MATH_F16MX}[math], False)
        return _capi.lib().stgcn_stem_kernel_name(3, 128, T, V, 9, 3, fl).decode() or "?"

    def workload_key(math):
        return f"{int(clips_per_launch)}x{T}x{V} {math} {'two-stage' if args.no_fuse else 'fused'} {args.layout}"

    def roofline(math, kms, launches, with_traffic):
        if not kms:
            return None
        achieved = flops_clip * clips_per_launch / (kms * 1e-3)
        peak = MFMA_PEAK[math]
        kname = kernel_name(math)
        roof = {"bound": "mfma", "kernel": kname,
                # matrix-core cycles issued per algorithmic product, in units of one 16-bit MFMA: three bf16 terms, or fp16 +
                # two scaled-e4m3 products at twice the 16-bit rate (+ the quarter-full tap-8 products: 10,752 / 4,608 cycles)
                "issued_over_algorithmic_flops": 3 if math == "bf16x3" else (2.33 if math == "f16mx" else 1),
                "achieved": round(achieved / 1e12, 3), "peak": round(peak / 1e12, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4),
                "achieved_over_f32_matrix_peak": round(achieved / MFMA_PEAK["f32"], 3),
                "kernel_ms": round(kms, 4), "launches_timed": launches,
                "algorithmic_flops_per_launch": int(flops_clip * clips_per_launch),
                "algorithmic_bytes_per_launch": int(bytes_clip * clips_per_launch),
                "hbm_GBps_of_kernel": round(bytes_clip * clips_per_launch / (kms * 1e-3) / 1e9, 1)}
        if with_traffic:
            roof["traffic"], roof["traffic_source"] = profiled_traffic(kname, workload_key(math))
        return roof

    line = None
    if rank == 0:
        value = n_global * args.steps / elapsed
        if strong:
            workload = (f"BASELINE configs[4]: ONE synthetic batch (N={n_global},3,{T},{V}) DP-sharded over {world} GPU(s), "
                        f"{n_local} clips on this rank in sub-batches of <= {sub}")
        else:
            named = {(256, 180, 22): "BASELINE configs[1] batch at 1 GPU", (512, 500, 22): "BASELINE configs[2] (DHG, long clips)",
                     (256, 200, 46): "BASELINE configs[3] (LMDHG two-hand graph)"}.get((n_local, T, V), "not a BASELINE config")
            workload = (f"{'SHREC' if V == 22 else 'LMDHG'}-graph stem forward: V={V}, T={T}, {n_local} clips/GPU "
                        f"({named}; weak-scaled)")
        line = {
            "metric": "clips/sec ST-GCN forward", "value": round(value, 1), "unit": "clips/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": DTYPE_NAME[args.math],
            "data": "synthetic randn clips (N,3,T,V), seeded random-init weights",
            "config": {"workload": workload, "clips_per_gpu": n_local, "global_clips": n_global, "T": T, "V": V,
                       "stem_passes_per_step": launches_per_step, "workload_key": workload_key(args.math),
                       "csrc_digest": csrc_digest(),
                       "math": args.math, "fused": not args.no_fuse, "layout": args.layout, "parallelism": f"dp{world}",
                       "parity": "1e-4 rel fp32 vs CPU oracle (tests/test_gpu_parity.py)"},
            "hbm_frac": round(value / world * bytes_clip / HBM_PEAK, 5),
            "mfma_frac": round(value / world * flops_clip / MFMA_PEAK[args.math], 4),
            "roofline": roofline(args.math, kern_ms, n_launch, True),
        }

    # ---- second, longer block: where the figure settles once the clocks have ramped -------------------------
    if args.steady_steps > 0:
        e2, k2, n2, _ = timed_block(args.steady_steps, 0)
        if rank == 0:
            line["steady_state"] = {"steps": args.steady_steps, "ms_per_step": round(e2 / args.steady_steps * 1e3, 4),
                                    "value": round(n_global * args.steady_steps / e2, 1),
                                    "kernel_ms": None if k2 is None else round(k2, 4),
                                    "roofline_frac": None if not k2 else round(
                                        flops_clip * clips_per_launch / (k2 * 1e-3) / MFMA_PEAK[args.math], 4)}

    # ---- the OTHER scaling kind, same process: strong (configs[4], 8192 clips over the ranks) beside a weak headline and
    #      vice versa, so that every line carries both and the 1-GPU line holds the base point of the strong curve --------
    if args.other_steps > 0 and T == 180 and V == 22:
        ow = make_workload(not strong, 8192)
        current = ow
        eo, ko, no_, outo = timed_block(args.other_steps, 3)
        assert torch.isfinite(outo).all()
        if pending is not None:
            pending.wait()
        if stats is not None:
            assert int(round(stats[0].item())) == ow["n_global"], "all-reduced clip count disagrees with the sharding"
        current = wl
        if rank == 0:
            line["other_scaling"] = {
                "scaling": "strong" if ow["strong"] else "weak",
                "workload": (f"BASELINE configs[4]: ONE batch of {ow['n_global']} clips over {world} GPU(s), sub-batches <= {ow['sub']}"
                             if ow["strong"] else f"{ow['n_local']} clips on every one of {world} GPU(s)"),
                "global_clips": ow["n_global"], "clips_per_gpu": ow["n_local"], "stem_passes_per_step": len(ow["shard"]),
                "steps": args.other_steps, "warmup": 3, "ms_per_step": round(eo / args.other_steps * 1e3, 4),
                "value": round(ow["n_global"] * args.other_steps / eo, 1), "unit": "clips/s",
                "kernel_ms": None if ko is None else round(ko, 4)}
        del ow

    # ---- exact-fp32 arithmetic, same workload, short block -----------------------------------------------------
    if args.alt_steps > 0 and args.math != "f32":
        stgcn_amd.set_math_mode(tcn, "f32")
        e3, k3, n3, out3 = timed_block(args.alt_steps, 5)
        assert torch.isfinite(out3).all()
        stgcn_amd.set_math_mode(tcn, args.math)
        if rank == 0:
            v3 = n_global * args.alt_steps / e3
            line["alt"] = {"dtype": "f32", "steps": args.alt_steps, "warmup": 5,
                           "ms_per_step": round(e3 / args.alt_steps * 1e3, 4), "value": round(v3, 1),
                           "mfma_frac": round(v3 / world * flops_clip / MFMA_PEAK["f32"], 4),
                           "roofline": roofline("f32", k3, n3, False)}

    # ---- secondary: one TRAINING step of the stem (train_sttran.py:185-191 restricted to the stem) --------
    if args.train_steps > 0 and not strong:
        g2, t2 = build_stem(V, args.graph, args.math)
        g2, t2 = g2.to(dev).train(), t2.to(dev).train()
        xt = shard[0]
        Gz = torch.randn(xt.shape[0], 128, T, V, device=dev)     # dL/dz handed in (a loss head would produce it)

        def train_step():
            for p_ in list(g2.parameters()) + list(t2.parameters()):
                p_.grad = None
            t2(g2(xt)).backward(Gz)
            sd.all_reduce_grads([g2, t2])                        # bucketed RCCL all-reduce; no-op at world 1

        for _ in range(5):
            train_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.train_steps):
            train_step()
        fence()
        e4 = sd.max_over_ranks(time.perf_counter() - t0, dev)
        if rank == 0:
            line["train"] = {"what": "forward + backward of tcn0(gcn0(x)) in .train() (batch-statistics BatchNorm, every "
                                     "parameter gradient; dL/dz synthetic), same clips",
                             "steps": args.train_steps, "warmup": 5, "ms_per_step": round(e4 / args.train_steps * 1e3, 4),
                             "value": round(n_global * args.train_steps / e4, 1), "unit": "clips/s"}
        del g2, t2, Gz

    # ---- tertiary: the deeper layers (model/ST_TR/ST_TR_new.py:355-372) — unit_agcn + Unit2D(k=9) of a
    #      TCN_GCN_unit, forward + backward WITH the input gradient, 64 clips x 90 frames (the shape DESIGN section 7 tracks) ------
    if args.deeper_steps > 0 and world == 1 and V == 22:
        from stgcn_amd import unit_agcn as _agcn, Unit2D as _u2d
        deeper = {}
        gA = torch.Generator().manual_seed(7)
        A3 = torch.rand(3, V, V, generator=gA) * (torch.rand(3, V, V, generator=gA) < 0.15)
        for cin_, cout_, st_ in ((64, 64, 1), (64, 128, 2)):
            torch.manual_seed(11)
            g3 = _agcn(cin_, cout_, A3.clone()).to(dev).train()
            t3 = _u2d(cout_, cout_, kernel_size=9, stride=st_).to(dev).train()
            stgcn_amd.set_math_mode(t3, args.math if args.math in ("bf16x3", "bf16", "f32") else "bf16x3")
            with torch.no_grad():
                g3.bn.weight.fill_(1.0)
            x3 = torch.randn(64, cin_, 90, V, device=dev).requires_grad_(True)
            gy3 = torch.ones_like(t3(g3(x3)).detach())
            ps3 = list(g3.parameters()) + list(t3.parameters())

            def deep_step():
                for p_ in ps3:
                    p_.grad = None
                x3.grad = None
                t3(g3(x3)).backward(gy3)

            for _ in range(3):
                deep_step()
            fence()
            t0 = time.perf_counter()
            for _ in range(args.deeper_steps):
                deep_step()
            fence()
            deeper[f"unit({cin_},{cout_},stride {st_})_ms"] = round((time.perf_counter() - t0) / args.deeper_steps * 1e3, 3)
            del g3, t3, x3, gy3, ps3
        if rank == 0:
            line["deeper_layers"] = {"what": "training step (forward + backward with dx) of unit_agcn + Unit2D(k=9) of a "
                                             "TCN_GCN_unit, 64 clips x 90 frames x 22 joints", "steps": args.deeper_steps,
                                     "warmup": 3, **deeper}

    if rank == 0:
        if cpu_state is not None:
            line["cpu_baseline"], line["cpu_oracle"] = cpu_legs(*cpu_state, T, V, args.cpu_clips, 12)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
