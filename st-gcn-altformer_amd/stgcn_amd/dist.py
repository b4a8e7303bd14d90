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
    Tie-breaking is numpy's (lowest index).  ``out`` is probed in place through its clip / channel strides, so the
    channels-last view that ``set_output_layout(model, "channels_last")`` makes the stem return is as good as a
    contiguous tensor.  CPU tensors (gloo tests) use the torch formulation."""
    if out.is_cuda:
        from ctypes import c_float, c_int, c_long, c_void_p
        from . import _capi
        if out.dim() != 4 or out.dtype not in (torch.float32, torch.bfloat16) or out.stride(0) <= 0 or out.stride(1) <= 0:
            raise ValueError("step_stats: `out` must be an fp32 / bf16 (N,C,T,V) tensor with positive clip / channel strides")
        if logits is not None:
            if logits.dtype != torch.float32 or not logits.is_contiguous() or logits.dim() != 2 or not logits.is_cuda:
                raise ValueError("step_stats: logits must be a contiguous fp32 (N,classes) GPU tensor")
            for t, nm in ((labels, "labels"), (pred, "pred")):
                if t is not None and (t.dtype != torch.int64 or not t.is_contiguous() or t.numel() != logits.shape[0]
                                      or t.device != logits.device):
                    raise ValueError(f"step_stats: {nm} must be a contiguous int64 tensor with one entry per row of logits")
        ptr = lambda t: c_void_p(0 if t is None else t.data_ptr())
        stats = torch.empty(4, device=out.device, dtype=torch.float32)
        with torch.cuda.device(out.device):
            _capi.call("stgcn_step_stats", c_void_p(out.data_ptr()), c_int(out.dtype == torch.bfloat16),
                       c_void_p(stats.data_ptr()), c_int(out.shape[0]), c_int(out.shape[1]), c_long(out.stride(0)),
                       c_long(out.stride(1)), c_float(float(n_local)), ptr(logits), ptr(labels if logits is not None else None),
                       ptr(pred if logits is not None else None), c_int(0 if logits is None else logits.shape[0]),
                       c_int(0 if logits is None else logits.shape[1]),
                       c_void_p(torch.cuda.current_stream(out.device).cuda_stream))
        return stats
    probe = out[:, :, 0, 0].float()
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


def all_reduce_grads(modules, average: bool = True, bucket_bytes: int = 32 << 20, min_buckets: int = 2,
                     keep_none: bool = True) -> int:
    """Data-parallel gradient exchange after ``loss.backward()`` (replaces nn.DataParallel's per-forward parameter
    broadcast + gradient reduce-to-GPU0, train_sttran.py:84).

    Every parameter with ``requires_grad`` takes part on every rank — a parameter whose ``.grad`` is None locally (an
    unused branch, a first step that skipped backward) contributes zeros and receives the sum — so all ranks always
    exchange the same layout (a bucket built from the locally present gradients only would differ across ranks and
    hang or mis-sum).  Parameters are packed per BUCKET (reverse registration order, <= ``bucket_bytes`` each, at
    least ``min_buckets``), each bucket is one ASYNC all-reduce, and bucket k is unpacked while bucket k+1 is still on
    the wire; nothing the size of the whole model is ever concatenated.  Sync BatchNorm is NOT applied: like the
    reference's DataParallel each replica normalises with its own batch statistics.

    ``keep_none`` (default): a parameter whose gradient is None on EVERY rank keeps ``.grad = None`` afterwards, as under
    the reference's DataParallel (its reduce-add only sees gradients that exist) — the optimizer then skips it, where a
    zero gradient would still let weight decay / momentum move it.  One "present" flag per parameter rides at the end
    of the last bucket (no extra collective); reading the flags costs one device-to-host copy per call.  With
    ``keep_none=False`` such parameters receive zeros and nothing is read back.
    Returns the number of gradient elements exchanged (0 without a process group).  For overlap with the backward
    itself use :class:`GradReducer`."""
    params = _trainable(modules)
    if not params or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    world = dist.get_world_size()
    inflight = []
    buckets = _bucketize(params, bucket_bytes, min_buckets)
    order = [p for bucket in buckets for p in bucket]
    for k, bucket in enumerate(buckets):
        p0 = bucket[0]
        n_el = sum(p.numel() for p in bucket)
        extra = len(order) if (keep_none and k == len(buckets) - 1) else 0
        flat = torch.zeros(n_el + extra, device=p0.device, dtype=p0.dtype)
        off = 0
        for p in bucket:
            n = p.numel()
            if p.grad is not None:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if extra:
            flat[off:] = torch.tensor([0.0 if p.grad is None else 1.0 for p in order], dtype=p0.dtype).to(p0.device)
        inflight.append((bucket, flat, n_el, dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)))
    present = None
    if keep_none:
        inflight[-1][3].wait()
        present = dict(zip(map(id, order), (inflight[-1][1][inflight[-1][2]:] > 0).tolist()))
    done = 0
    for bucket, flat, n_el, work in inflight:
        work.wait()
        if average:
            flat[:n_el] /= world
        off = 0
        for p in bucket:
            n = p.numel()
            if present is not None and not present[id(p)]:
                pass                                           # None on every rank: stays None
            elif p.grad is None:
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
    step.  A post-accumulate hook per parameter marks it ready; the moment a bucket's last gradient lands — and every
    bucket before it has gone out: the issue order is the same on every rank by construction — its collective is
    issued ``async_op`` on the communication stream while autograd keeps producing the earlier layers' gradients.  ``finish()`` (call it after ``loss.backward()``, before ``optimizer.step()``) launches whatever did not
    fire, waits, and averages.

    Contract per step (= from one ``finish()`` to the next):

    * A parameter that received NO gradient this step travels as zeros (every rank exchanges identical layouts).  Its
      slot is zeroed in ``finish()`` unless ``p.grad`` still IS the bucket view, i.e. unless the caller kept the view
      alive and is responsible for its content (``red.zero_grad()`` zeroes it; ``optimizer.zero_grad(set_to_none=True)``
      drops the view, and what the slot held — the previous step's averaged gradient — is discarded, not re-sent).
    * ``unused="zeros"`` (default): such a parameter ends the step with the (averaged) sum of the other ranks'
      gradients, zeros if no rank had one — note that an optimizer with weight decay / momentum still moves a
      parameter with a zero gradient.  ``unused="none"``: one extra tiny all-reduce of per-parameter "present" flags
      in ``finish()`` (plus a device-to-host read) and parameters without a gradient on ANY rank end with
      ``p.grad = None``, as under the reference's DataParallel.
    * ONE ``backward()`` per ``finish()``, or gradient accumulation under ``with red.no_sync():`` for all but the last
      backward (hooks then only keep the views attached; nothing is launched before the last backward).  A second
      gradient for a parameter whose bucket is already on the wire raises ``RuntimeError`` — the collective would race
      with autograd's write into the bucket.

    ``mode="rs_ag"`` issues reduce-scatter + all-gather per bucket instead of one all-reduce: the same bytes as a ring
    all-reduce, but as the two halves, so that a sharded optimizer step can later sit between them and each of the 7
    xGMI links carries 1/world of the bucket per half.  Whether the backend has ``reduce_scatter_tensor`` is probed ONCE
    at construction (a 1-element-per-rank collective, symmetric on all ranks); a backend that answers "not supported"
    (gloo) is run as all-reduce and ``self.mode`` says so; any other error propagates — and so does every error of the
    real collectives later (no silent change of mode in the middle of training).
    BatchNorm statistics stay per replica, like the reference's DataParallel.
    """

    def __init__(self, modules, bucket_bytes: int = 32 << 20, min_buckets: int = 2, average: bool = True,
                 mode: str = "all_reduce", unused: str = "zeros"):
        if mode not in ("all_reduce", "rs_ag"):
            raise ValueError(f"unknown mode {mode!r}")
        if unused not in ("zeros", "none"):
            raise ValueError(f"unknown unused-parameter policy {unused!r}")
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.average = average
        self.mode = mode
        self.unused = unused
        self.params = _trainable(modules)
        self.buckets = []
        self._slot = {}
        self._handles = []
        self._next = 0                    # first bucket not yet launched this step (buckets launch in order)
        self._sync = True                 # False inside no_sync()
        self._seen = set()                # parameters that received a gradient since the last finish() (any backward)
        for plist in _bucketize(self.params, bucket_bytes, min_buckets):
            p0 = plist[0]
            n = sum(p.numel() for p in plist)
            padded = -(-n // self.world) * self.world            # reduce-scatter wants equal shards
            flat = torch.zeros(padded, device=p0.device, dtype=p0.dtype)
            b = {"params": plist, "flat": flat, "n": n, "ready": set(), "work": None, "launched": False}
            off = 0
            for p in plist:
                view = flat[off:off + p.numel()].view_as(p)
                if p.grad is not None:
                    view.copy_(p.grad)
                p.grad = view
                self._slot[id(p)] = (b, view)
                off += p.numel()
            self.buckets.append(b)
        if self.mode == "rs_ag" and self.world > 1 and not self._probe_reduce_scatter(self.params[0].device):
            self.mode = "all_reduce"
        self._flags = None
        for p in self.params:
            self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _probe_reduce_scatter(self, device) -> bool:
        probe = torch.ones(self.world, device=device)
        try:
            dist.reduce_scatter_tensor(probe[self.rank:self.rank + 1], probe, op=dist.ReduceOp.SUM)
        except (RuntimeError, NotImplementedError) as e:
            msg = str(e).lower()
            if "not support" in msg or "not implemented" in msg or "unsupported" in msg or "does not support" in msg:
                return False
            raise
        return True

    # -- hooks ------------------------------------------------------------------------------------------------
    def _attach(self, p, view):
        """p.grad -> the bucket view, keeping what autograd just produced (zero_grad(set_to_none=True) drops the view:
        autograd then writes a fresh tensor)."""
        if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view

    def _on_grad(self, p):
        b, view = self._slot[id(p)]
        if b["launched"]:
            raise RuntimeError(
                "GradReducer: a parameter received a second gradient while its bucket's collective is already in flight "
                "(two backward() calls before finish()); wrap every backward but the last in `with reducer.no_sync():`")
        self._attach(p, view)
        self._seen.add(id(p))
        if not self._sync:
            return
        b["ready"].add(id(p))
        # Buckets go out strictly in bucket order: collectives are matched across ranks by the ORDER they are issued in,
        # and a parameter unused on one rank only would otherwise let that rank issue a later bucket first (equal sizes
        # would then be summed into each other without any error).  A complete bucket behind an incomplete one waits;
        # finish() sends the rest, in the same order on every rank.
        while self._next < len(self.buckets):
            nb = self.buckets[self._next]
            if len(nb["ready"]) != len(nb["params"]):
                break
            self._launch(nb)
            self._next += 1

    def no_sync(self):
        """Context manager for gradient accumulation: backward() calls inside only accumulate into the bucket views;
        the first backward outside (followed by finish()) exchanges the accumulated sum."""
        red = self

        class _NoSync:
            def __enter__(self_inner):
                self_inner.prev, red._sync = red._sync, False

            def __exit__(self_inner, *exc):
                red._sync = self_inner.prev
                return False
        return _NoSync()

    def _launch(self, b):
        b["launched"] = True
        if self.world == 1:
            return
        flat = b["flat"]
        if self.mode == "rs_ag":
            shard = flat.numel() // self.world
            mine = flat[self.rank * shard:(self.rank + 1) * shard]
            dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM)           # in place: my shard of the sum
            b["work"] = dist.all_gather_into_tensor(flat, mine, async_op=True)
            return
        b["work"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    # -- per step ---------------------------------------------------------------------------------------------
    def finish(self) -> int:
        """Complete this step's exchange; returns the number of gradient elements exchanged."""
        if not self._sync:
            raise RuntimeError("GradReducer.finish() inside no_sync(): leave the context before the last backward()")
        total = 0
        for b in self.buckets:
            if not b["launched"]:
                for p in b["params"]:
                    _, view = self._slot[id(p)]
                    if id(p) in self._seen:                   # (a gradient from a no_sync backward: already in the view)
                        self._attach(p, view)
                    elif p.grad is None or p.grad.data_ptr() != view.data_ptr():
                        # no gradient this step and the view was dropped (set_to_none) or replaced by hand: the slot still
                        # holds the PREVIOUS step's result — it must travel as zeros (or as the replacement's values)
                        if p.grad is None:
                            view.zero_()
                        else:
                            view.copy_(p.grad)
                        p.grad = view
                    # else: the caller kept the view alive (red.zero_grad() zeroed it, or it holds what they put there)
                self._launch(b)
        present = None
        if self.unused == "none" and self.world > 1:
            dev = self.params[0].device
            flags = torch.tensor([1.0 if id(p) in self._seen else 0.0 for p in self.params]).to(dev)
            dist.all_reduce(flags, op=dist.ReduceOp.SUM)
            present = (flags > 0).tolist()
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            if self.average and self.world > 1:
                b["flat"] /= self.world
            b["ready"], b["launched"] = set(), False
            total += b["n"]
        if present is not None:
            for p, here in zip(self.params, present):
                if not here:
                    p.grad = None                             # no gradient on any rank: the optimizer skips it
        self._seen = set()
        self._next = 0
        return total if self.world > 1 else 0

    def zero_grad(self) -> None:
        """Zero the bucket buffers and re-attach the views (use instead of optimizer.zero_grad(set_to_none=True), which
        drops the views; if that is called anyway the next backward re-attaches them at the price of one copy)."""
        for b in self.buckets:
            if b["launched"]:
                raise RuntimeError("GradReducer.zero_grad() between backward() and finish(): a collective is in flight")
            b["flat"].zero_()
        for p in self.params:
            p.grad = self._slot[id(p)][1]

    def remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []
