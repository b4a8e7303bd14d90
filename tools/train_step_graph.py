#!/usr/bin/env python3
"""The stem's training step captured ONCE as a HIP graph (torch.cuda.CUDAGraph) and replayed: every kernel of forward and
backward goes through the C ABI on the capturing stream, workspaces come from torch's (graph-private) allocator, nothing
synchronises the host.  Prints eager vs replay ms per step and checks that the replayed gradients equal the eager ones.

    python tools/train_step_graph.py [--clips 256] [--steps 100]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "st-gcn-altformer_amd")); sys.path.insert(0, ROOT)
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--clips", type=int, default=256); ap.add_argument("--frames", type=int, default=180)
ap.add_argument("--math", default="bf16x3"); ap.add_argument("--steps", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda:0")
gcn, tcn = bench.build_stem(22, "SHRE", a.math)
gcn, tcn = gcn.to(dev).train(), tcn.to(dev).train()
params = list(gcn.parameters()) + list(tcn.parameters())
x = bench.synthetic_clips(a.clips, a.frames, 22, seed=0).to(dev)
G = torch.randn(a.clips, 128, a.frames, 22, device=dev)


def step(set_to_none=True):
    if set_to_none:
        for p in params:
            p.grad = None
    z = tcn(gcn(x))
    z.backward(G)


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(10):
    step()
eager_ms = timed(step, a.steps)
# reference gradients from a state we can restore: the BatchNorm buffers move every step
state = [b.clone() for m in (gcn, tcn) for b in m.buffers()]
step()
ref = [p.grad.clone() for p in params]
for b, s in zip([b for m in (gcn, tcn) for b in m.buffers()], state):
    b.copy_(s)
# capture (static gradients: zero them in place, accumulate inside the graph)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
for b, s in zip([b for m in (gcn, tcn) for b in m.buffers()], state):
    b.copy_(s)
for p in params:
    p.grad = None
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step(set_to_none=False)
static_grads = [p.grad for p in params]
for b, s in zip([b for m in (gcn, tcn) for b in m.buffers()], state):
    b.copy_(s)
graph.replay()
torch.cuda.synchronize()
worst = max(((g - r).abs().max() / r.abs().max().clamp_min(1e-30)).item() for g, r in zip(static_grads, ref))
replay_ms = timed(graph.replay, a.steps)
print(json.dumps({"what": "stem training step, eager vs HIP-graph replay", "clips": a.clips, "eager_ms_per_step": round(eager_ms, 3),
                  "graph_ms_per_step": round(replay_ms, 3), "max_rel_grad_diff_replay_vs_eager": worst}))
