"""Tensor-level wrappers over the C ABI: torch supplies device memory and the stream, nothing else.

Every function takes CUDA(HIP) tensors, validates dtype/contiguity/device, passes raw device
pointers plus ``torch.cuda.current_stream()`` to libstgcn_hip.so and returns freshly allocated
outputs.  Nothing here computes on the host and nothing falls back to torch ops.
"""
from __future__ import annotations

from ctypes import c_float, c_int, c_size_t, c_uint, c_void_p
from typing import Optional, Tuple

import torch

from . import _capi
from ._capi import MATH_BF16, MATH_BF16X3, MATH_F16MX, MATH_F32, MATH_F32_VALU, OUT_BF16  # noqa: F401 (re-export)

BN_EPS = 1e-5


def _dev_ptr(t: Optional[torch.Tensor], name: str, device=None, dtype=torch.float32) -> c_void_p:
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); this path has no CPU implementation")
    if device is not None and t.device != device:
        raise ValueError(f"{name} is on {t.device}, expected {device}")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return c_void_p(t.data_ptr())


def _stream(device) -> c_void_p:
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _flags(math: int, out_bf16: bool) -> int:
    # (MATH_F16MX = MATH_BF16X3 | STEM_F16MX: the extra bit only means something to the stgcn_stem_* entry points)
    return (math & (_capi.MATH_MASK | _capi.STEM_F16MX)) | (OUT_BF16 if out_bf16 else 0)


def bn_fold(weight, bias, running_mean, running_var, conv_bias=None, eps: float = BN_EPS):
    """Eval-mode BatchNorm as per-channel (scale, shift); see stgcn_bn_fold."""
    dev = weight.device
    C = weight.numel()
    scale = torch.empty(C, device=dev, dtype=torch.float32)
    shift = torch.empty(C, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _capi.call("stgcn_bn_fold", _dev_ptr(weight, "weight", dev), _dev_ptr(bias, "bias", dev),
                   _dev_ptr(running_mean, "running_mean", dev), _dev_ptr(running_var, "running_var", dev),
                   _dev_ptr(conv_bias, "conv_bias", dev), c_float(eps), _dev_ptr(scale, "scale"),
                   _dev_ptr(shift, "shift"), c_int(C), _stream(dev))
    return scale, shift


def agcn_attention(x, A_eff, Wa, ba, Wb, bb) -> torch.Tensor:
    """P (N,S,V,V) of unit_agcn.py:81-85.  Wa/Wb (S,inter_c,Cin), ba/bb (S,inter_c), A_eff (S,V,V)."""
    dev = x.device
    N, Cin, T, V = x.shape
    S, inter_c, _ = Wa.shape
    P = torch.empty(N, S, V, V, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _capi.call("stgcn_agcn_attention", _dev_ptr(x, "x", dev), _dev_ptr(A_eff, "A_eff", dev),
                   _dev_ptr(Wa, "Wa", dev), _dev_ptr(ba, "ba", dev), _dev_ptr(Wb, "Wb", dev),
                   _dev_ptr(bb, "bb", dev), _dev_ptr(P, "P"), c_int(N), c_int(Cin), c_int(T), c_int(V),
                   c_int(inter_c), c_int(S), _stream(dev))
    return P


def agcn_forward(x, A_eff, Wa, ba, Wb, bb, Wd, bd, Wdown, bdown, bn_scale, bn_shift, down_scale,
                 down_shift) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval forward of unit_agcn.  Returns (y (N,Cout,T,V), P (N,S,V,V)).

    Wd (S,Cout,Cin), bd (S,Cout); Wdown (Cout,Cin)/bdown/down_scale/down_shift or all None for
    the identity residual (Cin == Cout).
    """
    dev = x.device
    N, Cin, T, V = x.shape
    S, inter_c, _ = Wa.shape
    Cout = Wd.shape[1]
    y = torch.empty(N, Cout, T, V, device=dev, dtype=torch.float32)
    P = torch.empty(N, S, V, V, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _capi.call("stgcn_agcn_forward", _dev_ptr(x, "x", dev), _dev_ptr(A_eff, "A_eff", dev),
                   _dev_ptr(Wa, "Wa", dev), _dev_ptr(ba, "ba", dev), _dev_ptr(Wb, "Wb", dev),
                   _dev_ptr(bb, "bb", dev), _dev_ptr(Wd, "Wd", dev), _dev_ptr(bd, "bd", dev),
                   _dev_ptr(Wdown, "Wdown", dev), _dev_ptr(bdown, "bdown", dev),
                   _dev_ptr(bn_scale, "bn_scale", dev), _dev_ptr(bn_shift, "bn_shift", dev),
                   _dev_ptr(down_scale, "down_scale", dev), _dev_ptr(down_shift, "down_shift", dev),
                   _dev_ptr(P, "P"), _dev_ptr(y, "y"), c_int(N), c_int(Cin), c_int(Cout), c_int(T), c_int(V),
                   c_int(inter_c), c_int(S), _stream(dev))
    return y, P


def tcn_supported(Cin, Cout, T, V, K, stride, math=MATH_F32) -> bool:
    return bool(_capi.lib().stgcn_tcn_supported(Cin, Cout, T, V, K, stride, _flags(math, False)))


def tcn_out_frames(T: int, K: int, stride: int) -> int:
    pad = int((K - 1) / 2)
    return (T + 2 * pad - K) // stride + 1


def tcn_pack(W, scale, math=MATH_F32) -> torch.Tensor:
    """Pack conv.weight (Cout,Cin,K) * scale into the kernel's operand order; returns a byte buffer."""
    dev = W.device
    Cout, Cin, K = W.shape
    fl = _flags(math, False)
    nbytes = _capi.lib().stgcn_tcn_packed_bytes(Cin, Cout, K, fl)
    Wp = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _capi.call("stgcn_tcn_pack", _dev_ptr(W, "W", dev), _dev_ptr(scale, "scale", dev),
                   c_void_p(Wp.data_ptr()), c_int(Cin), c_int(Cout), c_int(K), c_uint(fl), _stream(dev))
    return Wp


def tcn_forward_packed(x, Wp, shift, Cout, K, stride=1, math=MATH_F32, out_bf16=False, along_v=False) -> torch.Tensor:
    """``along_v``: Unit2D(dim=3) (model/net.py:28-36) — the convolution runs along the joint axis of x (N,Cin,T,V), read in
    place; MATH_F32_VALU packing only; returns (N,Cout,T,V_out)."""
    dev = x.device
    N, Cin, T, V = x.shape
    Lout = tcn_out_frames(V if along_v else T, K, stride)
    if Lout < 1:
        raise ValueError(f"1-D conv: length {V if along_v else T}, K={K}, stride={stride} leaves no output position")
    if along_v and math != MATH_F32_VALU:
        raise ValueError("tcn_forward_packed(along_v=True) needs the MATH_F32_VALU packing")
    shape = (N, Cout, T, Lout) if along_v else (N, Cout, Lout, V)
    y = torch.empty(*shape, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    fl = _flags(math, out_bf16) | (_capi.CONV_ALONG_V if along_v else 0)
    with torch.cuda.device(dev):
        _capi.call("stgcn_tcn_forward_packed", _dev_ptr(x, "x", dev), c_void_p(Wp.data_ptr()),
                   _dev_ptr(shift, "shift", dev), c_void_p(y.data_ptr()), c_int(N), c_int(Cin), c_int(Cout),
                   c_int(T), c_int(V), c_int(K), c_int(stride), c_uint(fl), _stream(dev))
    return y


def tcn_forward(x, W, scale, shift, stride=1, math=MATH_F32, out_bf16=False) -> torch.Tensor:
    """One-shot temporal conv block: packs into a scratch buffer, then runs (stgcn_tcn_forward)."""
    dev = x.device
    N, Cin, T, V = x.shape
    Cout, _, K = W.shape
    Tout = tcn_out_frames(T, K, stride)
    if Tout < 1:
        raise ValueError(f"temporal conv: T={T}, K={K}, stride={stride} leaves no output frame")
    fl = _flags(math, out_bf16)
    nbytes = _capi.lib().stgcn_tcn_packed_bytes(Cin, Cout, K, fl)
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    y = torch.empty(N, Cout, Tout, V, device=dev, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    with torch.cuda.device(dev):
        _capi.call("stgcn_tcn_forward", _dev_ptr(x, "x", dev), _dev_ptr(W, "W", dev),
                   _dev_ptr(scale, "scale", dev), _dev_ptr(shift, "shift", dev), c_void_p(y.data_ptr()),
                   c_int(N), c_int(Cin), c_int(Cout), c_int(T), c_int(V), c_int(K), c_int(stride),
                   c_void_p(ws.data_ptr()), c_size_t(nbytes), c_uint(fl), _stream(dev))
    return y


def stem_supported(Cin, C, T, V, K, S, math=MATH_F32) -> bool:
    return bool(_capi.lib().stgcn_stem_supported(Cin, C, T, V, K, S, _flags(math, False)))


def stem_prepare(Wd, bd, Wdown, bdown, bn_scale, bn_shift, down_scale, down_shift, Wt, t_scale,
                 math=MATH_F32) -> torch.Tensor:
    """Fold + pack everything the fused stem kernel reads besides x/P; returns the prep blob."""
    dev = Wd.device
    S, C, Cin = Wd.shape
    K = Wt.shape[2]
    fl = _flags(math, False)
    nbytes = _capi.lib().stgcn_stem_prep_bytes(Cin, C, K, S, fl)
    prep = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _capi.call("stgcn_stem_prepare", _dev_ptr(Wd, "Wd", dev), _dev_ptr(bd, "bd", dev),
                   _dev_ptr(Wdown, "Wdown", dev), _dev_ptr(bdown, "bdown", dev),
                   _dev_ptr(bn_scale, "bn_scale", dev), _dev_ptr(bn_shift, "bn_shift", dev),
                   _dev_ptr(down_scale, "down_scale", dev), _dev_ptr(down_shift, "down_shift", dev),
                   _dev_ptr(Wt, "Wt", dev), _dev_ptr(t_scale, "t_scale", dev), c_void_p(prep.data_ptr()),
                   c_int(Cin), c_int(C), c_int(K), c_int(S), c_uint(fl), _stream(dev))
    return prep


def _is_channels_last(x: torch.Tensor) -> bool:
    """(N,C,T,V)-shaped tensor whose memory is (N,T,V,C) — what ``x.permute(0,3,1,2)`` of the loader's batch is."""
    return x.dim() == 4 and not x.is_contiguous() and x.permute(0, 2, 3, 1).is_contiguous()


def stem_forward(x, A_eff, Wa, ba, Wb, bb, prep, t_shift, C, K, math=MATH_F32, out_bf16=False,
                 out: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
                 channels_last_out: bool = False):
    """Fused tcn0(gcn0(x)).  Returns (out (N,C,T,V), P (N,S,V,V) — a view of the workspace).

    x is (N,Cin,T,V), contiguous or channels-last strided (the permuted (N,T,V,Cin) batch of
    ST_GCN_AltFormer.py:62-68 — read in place, STGCN_IN_NTVC).  With ``channels_last_out`` the result is
    still shaped (N,C,T,V) but laid out (N,T,V,C) (STGCN_OUT_NTVC): the rearranges of model_ST.py:152 /
    model_TS.py:161 are views of it."""
    dev = x.device
    N, Cin, T, V = x.shape
    S, inter_c, _ = Wa.shape
    odt = torch.bfloat16 if out_bf16 else torch.float32
    fl = _flags(math, out_bf16)
    if _is_channels_last(x):
        fl |= _capi.IN_NTVC
        x = x.permute(0, 2, 3, 1)              # the contiguous (N,T,V,Cin) tensor behind the view
    oshape = (N, T, V, C) if channels_last_out else (N, C, T, V)
    if channels_last_out:
        fl |= _capi.OUT_NTVC
    if out is None:
        out = torch.empty(oshape, device=dev, dtype=odt)
    elif tuple(out.shape) != oshape or out.dtype != odt or not out.is_contiguous() or out.device != dev:
        raise ValueError("stem_forward: `out` has the wrong shape/dtype/device")
    need = _capi.lib().stgcn_stem_ws_bytes(N, Cin, C, T, V, K, S, fl)
    if ws is None or ws.numel() * ws.element_size() < need or ws.device != dev:
        ws = torch.empty((need + 3) // 4, device=dev, dtype=torch.float32)
    ws_bytes = c_size_t(ws.numel() * ws.element_size())
    st = _stream(dev)
    with torch.cuda.device(dev):
        _capi.call("stgcn_stem_attention", _dev_ptr(x, "x", dev), _dev_ptr(A_eff, "A_eff", dev),
                   _dev_ptr(Wa, "Wa", dev), _dev_ptr(ba, "ba", dev), _dev_ptr(Wb, "Wb", dev),
                   _dev_ptr(bb, "bb", dev), _dev_ptr(ws, "ws", dev), ws_bytes, c_int(N), c_int(Cin), c_int(C),
                   c_int(T), c_int(V), c_int(inter_c), c_int(S), c_int(K), c_uint(fl), st)
        timer = kernel_timer
        if timer is not None:
            timer.start("stem_tail", dev)
        _capi.call("stgcn_stem_tail_prepared", _dev_ptr(x, "x", dev), _dev_ptr(ws, "ws", dev), ws_bytes,
                   c_void_p(prep.data_ptr()), _dev_ptr(t_shift, "t_shift", dev), c_void_p(out.data_ptr()),
                   c_int(N), c_int(Cin), c_int(C), c_int(T), c_int(V), c_int(S), c_int(K), c_uint(fl), st)
        if timer is not None:
            timer.stop("stem_tail", dev)
    if channels_last_out:
        out = out.permute(0, 3, 1, 2)          # (N,C,T,V) view, torch.channels_last strides
    return out, ws[:N * S * V * V].view(N, S, V, V)


def agcn_forward_train(x, A_eff, Wa, ba, Wb, bb, Wd, bd, Wdown, bdown, bn, down_bn, momentum=0.1, eps=BN_EPS,
                       save=False, frozen=False):
    """Training-mode forward of unit_agcn (batch-statistics BatchNorm; running buffers of `bn` / `down_bn` are
    updated in place like torch does).  bn / down_bn: (weight, bias, running_mean, running_var) tensors.
    Returns (y, P); with ``save`` (y, P, zm, zd, stats): stats = batch mean / invstd of both BatchNorms (4*Cout) and, on
    the stem class's moments path, the 63 feature moments behind them (STGCN_AGCN_SAVE_STATS_FLOATS);
    zm, zd = the pre-BatchNorm branches, kept only with ``save="branches"`` (else None: the stem shape class then runs
    the moments path, which never writes them, and its backward works from y, dy and the moments).
    ``frozen``: BatchNorm on its RUNNING statistics (eval mode under autograd): nothing is updated; stats = running mean /
    invstd, for agcn_backward_train(..., frozen=True)."""
    dev = x.device
    N, Cin, T, V = x.shape
    S, inter_c, _ = Wa.shape
    Cout = Wd.shape[1]
    y = torch.empty(N, Cout, T, V, device=dev, dtype=torch.float32)
    P = torch.empty(N, S, V, V, device=dev, dtype=torch.float32)
    branches = save == "branches"            # keep zm / zd (the materialising path); save=True keeps the statistics only
    nbytes = _capi.lib().stgcn_agcn_train_ws_bytes(N, Cin, Cout, T, V, S, 1 if (branches or down_bn is None or frozen) else 0)
    ws = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.float64)
    d = down_bn if down_bn is not None else (None, None, None, None)
    zm = torch.empty_like(y) if branches else None
    zd = torch.empty_like(y) if branches and down_bn is not None else None
    stats = torch.empty(4 * Cout + 128, device=dev, dtype=torch.float32) if save else None
    with torch.cuda.device(dev):
        _capi.call("stgcn_agcn_forward_train", _dev_ptr(x, "x", dev), _dev_ptr(A_eff, "A_eff", dev),
                   _dev_ptr(Wa, "Wa", dev), _dev_ptr(ba, "ba", dev), _dev_ptr(Wb, "Wb", dev), _dev_ptr(bb, "bb", dev),
                   _dev_ptr(Wd, "Wd", dev), _dev_ptr(bd, "bd", dev), _dev_ptr(Wdown, "Wdown", dev),
                   _dev_ptr(bdown, "bdown", dev), *[_dev_ptr(t, "bn", dev) for t in bn],
                   *[_dev_ptr(t, "down_bn", dev) for t in d], c_float(momentum), c_float(eps), _dev_ptr(P, "P"),
                   c_void_p(ws.data_ptr()), c_size_t(ws.numel() * 8), _dev_ptr(y, "y"), _dev_ptr(zm, "save_zm"),
                   _dev_ptr(zd, "save_zd"), _dev_ptr(stats, "save_stats"), c_int(N), c_int(Cin), c_int(Cout),
                   c_int(T), c_int(V), c_int(inter_c), c_int(S), c_uint(_capi.BN_FROZEN if frozen else 0), _stream(dev))
    return (y, P, zm, zd, stats) if save else (y, P)


def agcn_backward_supported(N, Cin, Cout, T, V, S) -> bool:
    """Any shape the attention kernels cover (V <= 64) has a HIP backward: the fused kernel for the stem class, the GEMM
    chain otherwise."""
    return _capi.lib().stgcn_agcn_backward_ws_bytes(N, Cin, Cout, T, V, S, 3) > 0


def agcn_backward_train(x, A_eff, Wa, ba, Wb, bb, Wd, bd, Wdown, bdown, P, zm, zd, bn_weight, bn_bias, dbn_weight,
                        dbn_bias, stats, dy, need_dx=False, y=None, frozen=False):
    """Gradients of the training-mode unit_agcn forward.  zm / zd: the saved pre-BatchNorm branches, or None.  y: the
    forward's output — with it (and zm = zd = None, stats from a moments-path forward) the stem class runs its
    one-pass moment form; otherwise the GEMM chain, which rebuilds missing branches in the call's workspace.
    Wdown / bdown / dbn_* None = identity residual (Cin == Cout).  ``frozen``: the forward ran on running statistics
    (stats = running mean / invstd): they are constants of the backward (GEMM chain).
    Returns a dict keyed dWa, dba, dWb, dbb, dWd, dbd, dgamma, dbeta, dPA, plus dWdown, dbdown, ddgamma, ddbeta with a
    down branch and dx with ``need_dx`` (the input gradient, model/ST_TR/ST_TR_new.py:355-372)."""
    dev = x.device
    N, Cin, T, V = x.shape
    S, inter_c, _ = Wa.shape
    Cout = Wd.shape[1]
    has_down = Wdown is not None
    # bit 1: size for the generic path whenever the call may take it (dx, identity residual, non-stem shapes)
    nbytes = _capi.lib().stgcn_agcn_backward_ws_bytes(N, Cin, Cout, T, V, S, (1 if zm is None else 0) | 2)
    if nbytes == 0:
        raise NotImplementedError(f"unit_agcn backward: shape Cin={Cin} S={S} Cout={Cout} V={V} is not covered by the HIP path")
    ws = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.float64)
    f = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    g = dict(dWa=f(S, inter_c, Cin), dba=f(S, inter_c), dWb=f(S, inter_c, Cin), dbb=f(S, inter_c), dWd=f(S, Cout, Cin),
             dbd=f(S, Cout), dgamma=f(Cout), dbeta=f(Cout), dPA=f(S, V, V))
    if has_down:
        g.update(dWdown=f(Cout, Cin), dbdown=f(Cout), ddgamma=f(Cout), ddbeta=f(Cout))
    if need_dx:
        g["dx"] = torch.empty_like(x)
    o = lambda k: _dev_ptr(g.get(k), k)
    with torch.cuda.device(dev):
        _capi.call("stgcn_agcn_backward_train", _dev_ptr(x, "x", dev), _dev_ptr(A_eff, "A_eff", dev), _dev_ptr(Wa, "Wa", dev),
                   _dev_ptr(ba, "ba", dev), _dev_ptr(Wb, "Wb", dev), _dev_ptr(bb, "bb", dev), _dev_ptr(Wd, "Wd", dev),
                   _dev_ptr(bd, "bd", dev), _dev_ptr(Wdown, "Wdown", dev), _dev_ptr(bdown, "bdown", dev),
                   _dev_ptr(P, "P", dev), _dev_ptr(zm, "zm", dev), _dev_ptr(zd, "zd", dev),
                   _dev_ptr(bn_weight, "bn_weight", dev), _dev_ptr(bn_bias, "bn_bias", dev),
                   _dev_ptr(dbn_weight, "dbn_weight", dev), _dev_ptr(dbn_bias, "dbn_bias", dev),
                   _dev_ptr(stats, "stats", dev), _dev_ptr(y, "y", dev), _dev_ptr(dy, "dy", dev),
                   *[o(k) for k in ("dWa", "dba", "dWb", "dbb", "dWd", "dbd", "dWdown", "dbdown", "dgamma",
                                    "dbeta", "ddgamma", "ddbeta", "dPA", "dx")],
                   c_void_p(ws.data_ptr()), c_size_t(ws.numel() * 8), c_int(N), c_int(Cin), c_int(Cout), c_int(T),
                   c_int(V), c_int(inter_c), c_int(S), c_uint(_capi.BN_FROZEN if frozen else 0), _stream(dev))
    return g


def tcn_forward_train(x, W, conv_bias, bn, stride=1, math=MATH_F32, momentum=0.1, eps=BN_EPS, save=False, frozen=False):
    """Training-mode forward of Unit2D(dim=2, dropout=0): raw conv -> batch statistics -> normalise -> ReLU.

    ``save=True`` also returns what the backward needs: (y, z = conv_t(x)+b, batch mean, batch invstd).
    ``frozen``: normalise with the RUNNING statistics instead (eval mode under autograd; nothing is updated, mean / invstd
    returned are the running ones)."""
    dev = x.device
    N, Cin, T, V = x.shape
    Cout, _, K = W.shape
    Tout = tcn_out_frames(T, K, stride)
    if Tout < 1:
        raise ValueError(f"temporal conv: T={T}, K={K}, stride={stride} leaves no output frame")
    fl = _flags(math, False) | (_capi.BN_FROZEN if frozen else 0)
    nbytes = _capi.lib().stgcn_tcn_train_ws_bytes(N, Cin, Cout, T, V, K, stride, fl)
    ws = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.float64)
    y = torch.empty(N, Cout, Tout, V, device=dev, dtype=torch.float32)
    z = torch.empty_like(y) if save else None
    mean = torch.empty(Cout, device=dev, dtype=torch.float32) if save else None
    invstd = torch.empty(Cout, device=dev, dtype=torch.float32) if save else None
    with torch.cuda.device(dev):
        _capi.call("stgcn_tcn_forward_train", _dev_ptr(x, "x", dev), _dev_ptr(W, "W", dev),
                   _dev_ptr(conv_bias, "conv_bias", dev), *[_dev_ptr(t, "bn", dev) for t in bn], c_float(momentum),
                   c_float(eps), c_void_p(ws.data_ptr()), c_size_t(ws.numel() * 8), _dev_ptr(y, "y"),
                   _dev_ptr(z, "save_z"), _dev_ptr(mean, "save_mean"), _dev_ptr(invstd, "save_invstd"), c_int(N),
                   c_int(Cin), c_int(Cout), c_int(T), c_int(V), c_int(K), c_int(stride), c_uint(fl), _stream(dev))
    return (y, z, mean, invstd) if save else y


def tcn_backward_train(x, W, z, bn_weight, bn_bias, mean, invstd, dy, stride=1, math=MATH_F32, need_dx=True,
                       has_bias=True, frozen=False):
    """Backward of the training-mode Unit2D forward: returns (dx | None, dW (Cout,Cin,K), dbias | None, dgamma, dbeta)."""
    dev = x.device
    N, Cin, T, V = x.shape
    Cout, _, K = W.shape
    fl = _flags(math, False) | (_capi.BN_FROZEN if frozen else 0)   # frozen: mean / invstd are constants (running statistics)
    nbytes = _capi.lib().stgcn_tcn_backward_ws_bytes(N, Cin, Cout, T, V, K, stride, fl)
    ws = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.float64)
    dx = torch.empty_like(x) if need_dx else None
    dW = torch.empty_like(W)
    dbias = torch.empty(Cout, device=dev, dtype=torch.float32) if has_bias else None
    dgamma = torch.empty(Cout, device=dev, dtype=torch.float32)
    dbeta = torch.empty(Cout, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _capi.call("stgcn_tcn_backward_train", _dev_ptr(x, "x", dev), _dev_ptr(W, "W", dev), _dev_ptr(z, "z", dev),
                   _dev_ptr(bn_weight, "bn_weight", dev), _dev_ptr(bn_bias, "bn_bias", dev), _dev_ptr(mean, "mean", dev),
                   _dev_ptr(invstd, "invstd", dev), _dev_ptr(dy, "dy", dev), _dev_ptr(dx, "dx"), _dev_ptr(dW, "dW"),
                   _dev_ptr(dbias, "dbias"), _dev_ptr(dgamma, "dgamma"), _dev_ptr(dbeta, "dbeta"),
                   c_void_p(ws.data_ptr()), c_size_t(ws.numel() * 8), c_int(N), c_int(Cin), c_int(Cout), c_int(T), c_int(V),
                   c_int(K), c_int(stride), c_uint(fl), _stream(dev))
    return dx, dW, dbias, dgamma, dbeta


def patch_embed(z, weight, bias, pos=None, order="ST"):
    """First patch embedding of the transformer heads applied to the stem output z (N,C,T,V) — contiguous, or
    channels-last strided as the fused stem writes it with ``set_output_layout(..., "channels_last")``:

      order "ST": ``rearrange(z,'b c f p -> (b f) p c')`` -> ``nn.Linear(C,E)`` -> ``+= Spatial_pos_embed``
                  (model/AltFormer/model_ST.py:152-155)  -> (N*T, V, E)
      order "TS": ``rearrange(z,'b c f p -> (b p) f c')`` -> ``nn.Linear(C,E)`` -> ``+= Temporal_pos_embed``
                  (model/AltFormer/model_TS.py:161-163)  -> (N*V, T, E)

    One strided GEMM per clip (stgcn_patch_embed): the rearrange is folded into the addressing.  ``weight`` (E,C),
    ``bias`` (E), ``pos`` (1,V,E) / (1,T,E) or None."""
    dev = z.device
    N, C, T, V = z.shape
    E = weight.shape[0]
    fl = 0
    if _is_channels_last(z):
        fl |= _capi.IN_NTVC
        z = z.permute(0, 2, 3, 1)
    if order == "TS":
        fl |= _capi.EMBED_TS
    elif order != "ST":
        raise ValueError(f"order must be 'ST' or 'TS' (got {order!r})")
    if pos is not None:
        want = (T if order == "TS" else V) * E
        if pos.numel() != want:
            raise ValueError(f"pos has {pos.numel()} elements, expected {want}")
        pos = pos.reshape(-1, E)
    out = torch.empty((N * V, T, E) if order == "TS" else (N * T, V, E), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _capi.call("stgcn_patch_embed", _dev_ptr(z, "z", dev), _dev_ptr(weight, "weight", dev), _dev_ptr(bias, "bias", dev),
                   _dev_ptr(pos, "pos", dev), _dev_ptr(out, "out"), c_int(N), c_int(C), c_int(E), c_int(T), c_int(V),
                   c_uint(fl), _stream(dev))
    return out


class KernelTimer:
    """HIP-event bracket around individual kernel launches on the launching stream.

    Assign an instance to ``functional.kernel_timer`` to have the wrappers record an event pair per
    launch (no synchronisation while recording); ``mean_ms(name)`` synchronises and averages.
    """

    def __init__(self):
        self.pairs = {}
        self._open = {}

    def start(self, name, dev):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(dev))
        self._open[name] = ev

    def stop(self, name, dev):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(dev))
        self.pairs.setdefault(name, []).append((self._open.pop(name), ev))

    def reset(self):
        self.pairs.clear()
        self._open.clear()

    def count(self, name):
        return len(self.pairs.get(name, []))

    def mean_ms(self, name):
        pairs = self.pairs.get(name, [])
        if not pairs:
            return None
        pairs[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in pairs) / len(pairs)


kernel_timer: Optional[KernelTimer] = None
