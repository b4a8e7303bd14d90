"""Data-parallel harness for the stem: one process per GPU, clips sharded over ranks.

The forward of unit_agcn/Unit2D never mixes clips in eval mode (Gram, soft-max, aggregation and both
convs are per clip; BatchNorm uses running statistics), so rank r simply owns clips
[r*N/R, (r+1)*N/R) with the 0.6 MB of weights replicated, and the data path needs NO collective.
The only exchange is a tiny all-reduce of per-rank reductions — clip count, output checksum and, when
logits exist, correct-count / class histogram — the data-parallel form of the reference's accuracy
reduction (SHREC/ST_TS/train_sttran.py:105-109) and the replacement for nn.DataParallel's
scatter/replicate/gather (:84).  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" on CPU (tests).
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> Tuple[int, int]:
    """Initialise torch.distributed from the torchrun environment; returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


def shard_bounds(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n_total clips for `rank` (remainder spread over low ranks)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(batch: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    lo, hi = shard_bounds(batch.shape[0], rank, world)
    return batch[lo:hi]


def step_stats(out: torch.Tensor, n_local: int, logits: Optional[torch.Tensor] = None,
               labels: Optional[torch.Tensor] = None, pred: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-rank reductions to all-reduce: [clips, sum(probe), sum(probe^2), correct].

    ``correct`` = number of clips whose ``argmax(logits)`` equals the label — ``get_acc`` of
    SHREC/ST_TS/train_sttran.py:105-109 without the device-to-host copy: on the GPU one HIP launch
    (stgcn_step_stats) produces all four numbers, the class indices optionally land in ``pred`` (int64).
    Tie-breaking is numpy's (lowest index).  CPU tensors (gloo tests) use the torch formulation."""
    if out.is_cuda:
        from ctypes import c_float, c_int, c_long, c_void_p
        from . import _capi
        if not out.is_contiguous() or out.dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("step_stats: `out` must be a contiguous fp32 / bf16 (N,C,T,V) tensor")
        if logits is not None:
            if logits.dtype != torch.float32 or not logits.is_contiguous() or logits.dim() != 2 or not logits.is_cuda:
                raise ValueError("step_stats: logits must be a contiguous fp32 (N,classes) GPU tensor")
            for t, nm in ((labels, "labels"), (pred, "pred")):
                if t is not None and (t.dtype != torch.int64 or not t.is_contiguous() or t.numel() != logits.shape[0]
                                      or t.device != logits.device):
                    raise ValueError(f"step_stats: {nm} must be a contiguous int64 tensor with one entry per row of logits")
        ptr = lambda t: c_void_p(0 if t is None else t.data_ptr())
        stats = torch.empty(4, device=out.device, dtype=torch.float32)
        plane = out[0, 0].numel()
        with torch.cuda.device(out.device):
            _capi.call("stgcn_step_stats", c_void_p(out.data_ptr()), c_int(out.dtype == torch.bfloat16),
                       c_void_p(stats.data_ptr()), c_int(out.shape[0]), c_int(out.shape[1]), c_long(plane),
                       c_float(float(n_local)), ptr(logits), ptr(labels if logits is not None else None),
                       ptr(pred if logits is not None else None), c_int(0 if logits is None else logits.shape[0]),
                       c_int(0 if logits is None else logits.shape[1]),
                       c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
        return stats
    probe = out.reshape(out.shape[0], out.shape[1], -1)[:, :, 0].float()
    correct = out.new_zeros((), dtype=torch.float32)
    if logits is not None:
        am = logits.argmax(dim=1)
        if pred is not None:
            pred.copy_(am)
        if labels is not None:
            correct = (am == labels).sum().float()                    # get_acc, train_sttran.py:105-109
    return torch.stack([out.new_tensor(float(n_local), dtype=torch.float32), probe.sum(), probe.square().sum(),
                        correct])


def all_reduce_stats(stats: torch.Tensor) -> torch.Tensor:
    """Sum over ranks (in place when distributed); tiny and latency-bound by design."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats)
    return stats


def all_reduce_stats_async(stats: torch.Tensor):
    """Same exchange without making the compute stream wait for it: returns (stats, work); call ``work.wait()`` (or
    synchronise the device) before reading ``stats``.  The collective runs on RCCL's own stream beside the next step's
    kernels — a synchronous all-reduce would put its latency (tens of microseconds over xGMI) into every 0.8 ms step."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return stats, dist.all_reduce(stats, async_op=True)
    return stats, None


def max_over_ranks(seconds: float, device: torch.device) -> float:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([seconds], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return seconds


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def _trainable(modules) -> List[torch.nn.Parameter]:
    if isinstance(modules, torch.nn.Module):
        modules = [modules]
    seen, out = set(), []
    for m in modules:
        for p in m.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                out.append(p)
    return out


def _bucketize(params: List[torch.nn.Parameter], bucket_bytes: int, min_buckets: int) -> List[List[torch.nn.Parameter]]:
    """Reverse registration order (the order autograd finishes gradients in), cut every `bucket_bytes`; the cap shrinks
    so that at least `min_buckets` buckets exist (a second collective can then run while the first one's copy-out does)."""
    total = sum(p.numel() * p.element_size() for p in params)
    if min_buckets > 1 and total > 0:
        bucket_bytes = max(1, min(bucket_bytes, -(-total // min_buckets)))
    buckets, cur, cur_b = [], [], 0
    for p in reversed(params):
        nb = p.numel() * p.element_size()
        if cur and cur_b + nb > bucket_bytes:
            buckets.append(cur)
            cur, cur_b = [], 0
        cur.append(p)
        cur_b += nb
    if cur:
        buckets.append(cur)
    return buckets


def all_reduce_grads(modules, average: bool = True, bucket_bytes: int = 32 << 20, min_buckets: int = 2) -> int:
    """Data-parallel gradient exchange after ``loss.backward()`` (replaces nn.DataParallel's per-forward parameter
    broadcast + gradient reduce-to-GPU0, train_sttran.py:84).

    Every parameter with ``requires_grad`` takes part on every rank — a parameter whose ``.grad`` is None locally (an
    unused branch, a first step that skipped backward) contributes zeros and receives the sum — so all ranks always
    exchange the same layout (a bucket built from the locally present gradients only would differ across ranks and
    hang or mis-sum).  Parameters are packed per BUCKET (reverse registration order, <= ``bucket_bytes`` each, at
    least ``min_buckets``), each bucket is one ASYNC all-reduce, and bucket k is unpacked while bucket k+1 is still on
    the wire; nothing the size of the whole model is ever concatenated.  Sync BatchNorm is NOT applied: like the
    reference's DataParallel each replica normalises with its own batch statistics.
    Returns the number of elements exchanged (0 without a process group).  For overlap with the backward itself use
    :class:`GradReducer`."""
    params = _trainable(modules)
    if not params or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    inflight = []
    for bucket in _bucketize(params, bucket_bytes, min_buckets):
        p0 = bucket[0]
        flat = torch.zeros(sum(p.numel() for p in bucket), device=p0.device, dtype=p0.dtype)
        off = 0
        for p in bucket:
            n = p.numel()
            if p.grad is not None:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        inflight.append((bucket, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))
    done = 0
    for bucket, flat, work in inflight:
        work.wait()
        if average:
            flat /= world
        off = 0
        for p in bucket:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
        done += off
    return done


class GradReducer:
    """Bucketed gradient exchange that overlaps the backward (the §8(f)-4 shape: the full model is 32.2 M parameters =
    129 MB fp32; the stem alone is 0.6 MB).

    Construction flattens the parameters' gradients into a few persistent bucket buffers — ``p.grad`` becomes a VIEW
    of its bucket, so autograd accumulates straight into the exchange buffer and nothing is packed or unpacked per
    step.  A post-accumulate hook per parameter counts a bucket down; the moment its last gradient lands, the bucket's
    collective is issued ``async_op`` on the communication stream while autograd keeps producing the earlier layers'
    gradients.  ``finish()`` (call it after ``loss.backward()``, before ``optimizer.step()``) launches whatever did not
    fire (parameters unused this step: their slots hold zeros, so every rank still exchanges identical layouts),
    waits, and averages.

    ``mode="rs_ag"`` issues reduce-scatter + all-gather per bucket instead of one all-reduce: the same bytes as a ring
    all-reduce, but as the two halves, so that a sharded optimizer step can later sit between them and each of the 7
    xGMI links carries 1/world of the bucket per half; backends without reduce-scatter (gloo) fall back to all-reduce.
    BatchNorm statistics stay per replica, like the reference's DataParallel.
    """

    def __init__(self, modules, bucket_bytes: int = 32 << 20, min_buckets: int = 2, average: bool = True,
                 mode: str = "all_reduce"):
        if mode not in ("all_reduce", "rs_ag"):
            raise ValueError(f"unknown mode {mode!r}")
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.average = average
        self.mode = mode
        self.params = _trainable(modules)
        self.buckets = []
        self._slot = {}
        self._handles = []
        for plist in _bucketize(self.params, bucket_bytes, min_buckets):
            p0 = plist[0]
            n = sum(p.numel() for p in plist)
            padded = -(-n // self.world) * self.world            # reduce-scatter wants equal shards
            flat = torch.zeros(padded, device=p0.device, dtype=p0.dtype)
            b = {"params": plist, "flat": flat, "n": n, "pending": len(plist), "work": None, "launched": False}
            off = 0
            for p in plist:
                view = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
                self._slot[id(p)] = (b, view)
                off += p.numel()
            self.buckets.append(b)
        for p in self.params:
            self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- hooks ------------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        b, view = self._slot[id(p)]
        if p.grad is not None and p.grad.data_ptr() != view.data_ptr():   # zero_grad(set_to_none=True) dropped the view
            view.copy_(p.grad)
            p.grad = view
        b["pending"] -= 1
        if b["pending"] == 0 and not b["launched"]:
            self._launch(b)

    def _launch(self, b):
        b["launched"] = True
        if self.world == 1:
            return
        flat = b["flat"]
        if self.mode == "rs_ag":
            shard = flat.numel() // self.world
            mine = flat[self.rank * shard:(self.rank + 1) * shard]
            try:
                dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM)       # in place: my shard of the sum
                b["work"] = dist.all_gather_into_tensor(flat, mine, async_op=True)
                return
            except (RuntimeError, NotImplementedError):
                self.mode = "all_reduce"                                            # e.g. gloo: no reduce-scatter
        b["work"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    # -- per step ---------------------------------------------------------------------------------------------
    def finish(self) -> int:
        """Complete this step's exchange; returns the number of gradient elements exchanged."""
        total = 0
        for b in self.buckets:
            if not b["launched"]:
                for p in b["params"]:                     # unused parameters whose view was dropped: restore zeros
                    _, view = self._slot[id(p)]
                    if p.grad is None or p.grad.data_ptr() != view.data_ptr():
                        if p.grad is not None:
                            view.copy_(p.grad)
                        p.grad = view
                self._launch(b)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            if self.average and self.world > 1:
                b["flat"] /= self.world
            b["pending"], b["launched"] = len(b["params"]), False
            total += b["n"]
        return total if self.world > 1 else 0

    def zero_grad(self) -> None:
        """Zero the bucket buffers (use instead of optimizer.zero_grad(set_to_none=True), which would drop the views;
        if it is called anyway the next backward re-attaches them at the price of one copy)."""
        for b in self.buckets:
            b["flat"].zero_()
        for p in self.params:
            p.grad = self._slot[id(p)][1]

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
