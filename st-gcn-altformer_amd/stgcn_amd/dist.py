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
from typing import Optional, Tuple

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
               labels: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Per-rank reductions to all-reduce: [clips, sum(probe), sum(probe^2), correct]."""
    if out.is_cuda and logits is None and out.is_contiguous() and out.dtype in (torch.float32, torch.bfloat16):
        # one HIP launch on the current stream (stgcn_step_stats) instead of half a dozen tiny torch kernels
        from ctypes import c_float, c_int, c_long, c_void_p
        from . import _capi
        stats = torch.empty(4, device=out.device, dtype=torch.float32)
        plane = out[0, 0].numel()
        with torch.cuda.device(out.device):
            _capi.call("stgcn_step_stats", c_void_p(out.data_ptr()), c_int(out.dtype == torch.bfloat16),
                       c_void_p(stats.data_ptr()), c_int(out.shape[0]), c_int(out.shape[1]), c_long(plane),
                       c_float(float(n_local)), c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
        return stats
    probe = out.reshape(out.shape[0], out.shape[1], -1)[:, :, 0].float()
    correct = out.new_zeros((), dtype=torch.float32)
    if logits is not None and labels is not None:
        correct = (logits.argmax(dim=1) == labels).sum().float()      # get_acc, train_sttran.py:105-109
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


def all_reduce_grads(modules, average: bool = True) -> int:
    """Data-parallel gradient exchange for training: every ``.grad`` below ``modules`` is packed
    into ONE flat bucket, summed over the ranks with a single all-reduce and unpacked (divided by the world size when
    ``average``).  The stem's 152,620 parameters are 0.6 MB — one latency-bound RCCL call per step instead of
    nn.DataParallel's per-forward parameter broadcast + gradient reduce-to-GPU0 (train_sttran.py:84).  Sync BatchNorm
    is NOT applied: like the reference's DataParallel each replica normalises with its own batch statistics.
    Returns the number of elements exchanged (0 without a process group)."""
    if isinstance(modules, torch.nn.Module):
        modules = [modules]
    grads = [p.grad for m in modules for p in m.parameters() if p.grad is not None]
    if not grads or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    bucket = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
    if average:
        bucket /= dist.get_world_size()
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(bucket[off:off + n].view_as(g))
        off += n
    return off
