"""nn.Module mirror of the reference's ST-GCN stem, running on libstgcn_hip.so.

``unit_agcn`` <- model/unit_agcn.py:31-93, ``Unit2D`` <- model/net.py:7-57 of the reference: same
constructor arguments, sub-module / parameter names (so existing ``.pth`` files load strictly),
initialisation and error behaviour.  ``forward`` hands raw device pointers to the HIP kernels;
there is no torch-op implementation of the math in this package.

Eval-mode parameters are folded/packed once and cached; the cache is keyed on the version
counter AND the storage address of every parameter and buffer, so an optimizer step,
``load_state_dict``, a manual in-place edit or a ``param.data = other`` swap invalidates it.
"""
from __future__ import annotations

import math
import warnings
import os
from typing import Optional

import torch
import torch.nn as nn

from . import functional as F
from ._capi import MATH_BF16, MATH_BF16X3, MATH_F16MX, MATH_F32, MATH_F32_VALU

_MATH_NAMES = {"f32": MATH_F32, "bf16x3": MATH_BF16X3, "bf16": MATH_BF16, "f32_valu": MATH_F32_VALU, "f16mx": MATH_F16MX}


def _default_math() -> int:
    """Arithmetic of the temporal-conv contraction unless overridden per module (env STGCN_MATH).

    'bf16x3' (default) keeps the fp32 contract (1e-4 relative, tests/test_gpu_parity.py) at 3/16 of the
    fp32-MFMA cost; 'f32' is the bit-exact fp32 fma chain; 'bf16' rounds operands to bf16 (1e-2).
    """
    return _MATH_NAMES[os.environ.get("STGCN_MATH", "bf16x3").lower()]


def _identity(x):
    return x


def _versions(mod: nn.Module):
    """Cache key over every parameter and buffer: (version counter, storage address).  The counter sees in-place edits
    (an optimizer step, load_state_dict, ``p.data.mul_()``); the address sees ``param.data = new_tensor`` — an EMA or
    weight swap done that way leaves ``_version`` where it was (round-2 review) but moves ``data_ptr()``."""
    return tuple((t._version, t.data_ptr()) for t in list(mod.parameters()) + list(mod.buffers()))


def _wants_grad(mod: nn.Module, x: torch.Tensor) -> bool:
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mod.parameters()))


def _check_input(mod: nn.Module, x: torch.Tensor, backward_ok: bool = False, bn_training: Optional[bool] = None):
    """Refuse what the HIP path does not implement instead of answering wrongly.

    ``bn_training``: whether the module's BatchNorm(s) run on batch statistics (``self.bn.training`` — the reference's
    nn.BatchNorm2d decides per sub-module, so ``model.train()`` followed by ``bn.eval()`` freezes the statistics there
    and must do so here).  Autograd exists for both modes: batch statistics (the training scripts' case,
    train_sttran.py:176-191) and running statistics (eval mode with gradients enabled — frozen-BatchNorm fine-tuning,
    saliency; STGCN_BN_FROZEN); only Unit2D(dim=3) has no backward."""
    if not x.is_cuda:
        raise RuntimeError(
            f"{type(mod).__name__}: input is on {x.device}; the HIP path runs on the GPU only "
            "(there is deliberately no CPU fallback)")
    if x.dtype != torch.float32:
        raise TypeError(f"{type(mod).__name__}: input must be float32 (got {x.dtype})")
    if x.dim() != 4:
        raise ValueError(f"{type(mod).__name__}: expected (N,C,T,V), got {tuple(x.shape)}")
    if bn_training is None:
        bn_training = mod.training
    if _wants_grad(mod, x):
        if not backward_ok:
            raise NotImplementedError(
                f"{type(mod).__name__}: no HIP backward for this configuration (Unit2D(dim=3)); the dim=2 temporal block "
                "and unit_agcn have one (SURVEY.md §8f rank 2)")


# ----------------------------------------------------------------------------------------
class FusedStemOutput(torch.Tensor):
    """What ``unit_agcn.forward`` returns while stem fusion is enabled: the result of ``tcn0(gcn0(x))`` already
    computed by the fused kernel, typed so that ONLY its ``Unit2D`` can take it.

    ``ST_GCN_AltFormer.forward`` does ``x = self.gcn0(x); x = self.tcn0(x)`` (ST_GCN_AltFormer.py:70-72); with fusion
    the first call has nothing of its own to return (gcn0's activation never leaves the chip).  Handing out the final
    tensor as a plain Tensor would be a trap: a forward hook, a residual branch or a cast between the two modules would
    read tcn0's output as if it were gcn0's, or strip a marker and run the temporal conv twice.  So every torch
    operation on this type raises; shape / dtype / device queries work; ``Unit2D.forward`` of the paired module unwraps
    it.  Whoever needs gcn0's real activation calls ``disable_stem_fusion(gcn)`` (two-stage path)."""

    _PASSIVE = None

    @staticmethod
    def wrap(t: torch.Tensor, consumer: "Unit2D") -> "FusedStemOutput":
        with torch._C.DisableTorchFunctionSubclass():
            out = t.as_subclass(FusedStemOutput)
        out._stgcn_consumer = consumer
        return out

    def unwrap(self) -> torch.Tensor:
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor)

    @classmethod
    def _passive(cls):
        if cls._PASSIVE is None:
            T = torch.Tensor
            cls._PASSIVE = {T.shape.__get__, T.dtype.__get__, T.device.__get__, T.is_cuda.__get__, T.ndim.__get__,
                            T.requires_grad.__get__, T.grad_fn.__get__, T.is_leaf.__get__, T.layout.__get__,
                            T.dim, T.size, T.stride, T.numel, T.is_contiguous, T.data_ptr, T.element_size,
                            T.storage_offset, T.is_floating_point, T.__repr__, T.__len__, T._version.__get__}
        return cls._PASSIVE

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if func in cls._passive():
            with torch._C.DisableTorchFunctionSubclass():
                if func is torch.Tensor.__repr__:
                    a = args[0]
                    return f"FusedStemOutput(shape={tuple(a.shape)}, dtype={a.dtype}, device={a.device})"
                return func(*args, **(kwargs or {}))
        raise RuntimeError(
            f"{getattr(func, '__name__', func)}: this tensor is the fused ST-GCN stem's deferred result — unit_agcn returned "
            "it for its paired Unit2D only (stgcn_amd.enable_stem_fusion), and gcn0's own activation was never written. "
            "Pass it straight to that Unit2D, or call stgcn_amd.disable_stem_fusion(gcn) to get the two-stage path "
            "(hooks, residual branches and feature extraction on gcn0 need it)")


# ----------------------------------------------------------------------------------------
class _Unit2DTrainFn(torch.autograd.Function):
    """relu(BatchNorm(conv_t(x) + b)) with the HIP forward and backward (model/net.py:47-57): batch statistics in
    .train(), running statistics (``frozen``: constants of the backward) in .eval() under autograd."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn_weight, bn_bias, running_mean, running_var, stride, mode, momentum, eps, frozen=False):
        Cout, Cin, K = weight.shape[0], weight.shape[1], weight.shape[2]
        W = weight.detach().reshape(Cout, Cin, K).contiguous()
        bnw = bn_weight.detach() if bn_weight is not None else None
        if bn_weight is None or bn_bias is None:
            raise NotImplementedError("Unit2D: the HIP training path needs an affine BatchNorm")
        y, z, mean, invstd = F.tcn_forward_train(x.detach(), W, None if bias is None else bias.detach(),
                                                 (bnw, bn_bias.detach(), running_mean, running_var), stride, mode,
                                                 momentum, eps, save=True, frozen=frozen)
        ctx.save_for_backward(x.detach(), W, z, bnw, bn_bias.detach(), mean, invstd)
        ctx.meta = (stride, mode, bias is not None, tuple(weight.shape), frozen)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, z, bnw, bnb, mean, invstd = ctx.saved_tensors
        stride, mode, has_bias, wshape, frozen = ctx.meta
        dx, dW, db, dgamma, dbeta = F.tcn_backward_train(x, W, z, bnw, bnb, mean, invstd, dy.contiguous(), stride, mode,
                                                         need_dx=ctx.needs_input_grad[0], has_bias=has_bias, frozen=frozen)
        return (dx, dW.view(wshape), db if has_bias else None, dgamma, dbeta, None, None, None, None, None, None, None)


class _AgcnTrainFn(torch.autograd.Function):
    """unit_agcn.forward in .train() with the HIP forward and backward: every parameter gradient and, when the input
    requires it (the deeper TCN_GCN_unit layers, model/ST_TR/ST_TR_new.py:355-372), dx."""

    @staticmethod
    def forward(ctx, mod, x, *params):               # `params` = mod._train_params(): graph edges only, values via _staged
        st = mod._staged(x.device)
        bn = mod.bn
        has_down = mod._has_down()
        d = mod.down[1] if has_down else None
        xd = x.detach()
        frozen = not mod._bn_training()              # .eval() under autograd: running statistics, constants of the backward
        # The stem shape class (3 input channels, 3 subsets, batch statistics) derives its BatchNorm statistics from feature
        # moments and writes neither pre-BatchNorm branch; its backward works from y — which the consumer keeps anyway — dy
        # and those moments.  Every other call materialises the branches anyway: keep them for the backward (the deeper
        # TCN_GCN_unit layers spent 0.13 - 0.42 ms per step rebuilding them with the expansion kernel).
        moments = has_down and not frozen and x.shape[1] == 3 and mod.num_subset == 3
        y, P, zm, zd, stats = F.agcn_forward_train(
            xd, st["A_eff"], st["Wa"], st["ba"], st["Wb"], st["bb"], st["Wd"], st["bd"], st["Wdown"], st["bdown"],
            (bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var),
            (d.weight.detach(), d.bias.detach(), d.running_mean, d.running_var) if has_down else None,
            bn.momentum, bn.eps, save=True if moments else "branches", frozen=frozen)
        saved = [xd, st["A_eff"], st["Wa"], st["ba"], st["Wb"], st["bb"], st["Wd"], st["bd"], P, bn.weight.detach(),
                 bn.bias.detach(), stats, y]
        if has_down:
            saved += [st["Wdown"], st["bdown"], d.weight.detach(), d.bias.detach()]
        ctx.branches = zm is not None
        if ctx.branches:
            saved += [zm] + ([zd] if has_down else [])
        ctx.save_for_backward(*saved)
        ctx.S = mod.num_subset
        ctx.has_down = has_down
        ctx.frozen = frozen
        mod.last_attention = P
        return y

    @staticmethod
    def backward(ctx, dy):
        t = ctx.saved_tensors
        x, A_eff, Wa, ba, Wb, bb, Wd, bd, P, bnw, bnb, stats, y = t[:13]
        Wdown, bdown, dbnw, dbnb = t[13:17] if ctx.has_down else (None, None, None, None)
        rest = t[17:] if ctx.has_down else t[13:]
        zm = rest[0] if ctx.branches else None
        zd = rest[1] if ctx.branches and ctx.has_down else None
        need_dx = ctx.needs_input_grad[1]
        g = F.agcn_backward_train(x, A_eff, Wa, ba, Wb, bb, Wd, bd, Wdown, bdown, P, zm, zd, bnw, bnb, dbnw, dbnb, stats,
                                  dy.contiguous(), need_dx=need_dx, y=y, frozen=ctx.frozen)
        S = ctx.S
        out = [g["dPA"]]
        for w, b in (("dWa", "dba"), ("dWb", "dbb"), ("dWd", "dbd")):
            for i in range(S):
                out += [g[w][i].unsqueeze(-1).unsqueeze(-1), g[b][i]]
        if ctx.has_down:
            out += [g["dWdown"].unsqueeze(-1).unsqueeze(-1), g["dbdown"], g["ddgamma"], g["ddbeta"]]
        out += [g["dgamma"], g["dbeta"]]
        return (None, g.get("dx"), *out)


def conv_init(module):
    """He-normal on the weight only (model/net.py:60-65)."""
    n = module.out_channels
    for k in module.kernel_size:
        n = n * k
    module.weight.data.normal_(0, math.sqrt(2. / n))


def import_class(name):
    """'graph.SHRE' -> class; prints the name like the reference does (model/net.py:68-74)."""
    print("name", name)
    components = name.split('.')
    mod = __import__(components[0])
    for comp in components[1:]:
        mod = getattr(mod, comp)
    return mod


def _agcn_conv_init(conv):          # model/unit_agcn.py:12-14
    nn.init.kaiming_normal_(conv.weight, mode='fan_out')
    nn.init.constant_(conv.bias, 0)


def _bn_init(bn, scale):            # model/unit_agcn.py:17-19
    nn.init.constant_(bn.weight, scale)
    nn.init.constant_(bn.bias, 0)


def _conv_branch_init(conv, branches):   # model/unit_agcn.py:22-28
    n, k1, k2 = conv.weight.size(0), conv.weight.size(1), conv.weight.size(2)
    nn.init.normal_(conv.weight, 0, math.sqrt(2. / (n * k1 * k2 * branches)))
    nn.init.constant_(conv.bias, 0)


# ----------------------------------------------------------------------------------------
class unit_agcn(nn.Module):
    """Adaptive graph convolution; signature and state_dict of model/unit_agcn.py:31-71.

    Note on ``A``: like the reference, ``PA`` is created *on A's storage* and then filled with
    1e-6 (model/unit_agcn.py:37-39), so the caller's tensor and ``self.A`` both read 1e-6 after
    construction and the skeleton adjacency is effectively learned from scratch through ``PA``.
    This is reproduced on purpose (results must match the reference for the same checkpoint);
    assign ``module.A = graph_tensor`` afterwards to use the true graph as the fixed term.
    ``use_local_bn`` / ``mask_learning`` are accepted and ignored, as in the reference.
    """

    def __init__(self, in_channels, out_channels, A, coff_embedding=4, num_subset=3, use_local_bn=False,
                 mask_learning=False):
        super().__init__()
        inter_channels = out_channels // coff_embedding
        self.inter_c = inter_channels
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.PA = nn.Parameter(A)
        nn.init.constant_(self.PA, 1e-6)
        self.A = A                      # plain attribute: not a buffer, not in state_dict, never moved by .cuda()
        self.num_subset = num_subset

        self.conv_a = nn.ModuleList()
        self.conv_b = nn.ModuleList()
        self.conv_d = nn.ModuleList()
        for _ in range(self.num_subset):
            self.conv_a.append(nn.Conv2d(in_channels, inter_channels, 1))
            self.conv_b.append(nn.Conv2d(in_channels, inter_channels, 1))
            self.conv_d.append(nn.Conv2d(in_channels, out_channels, 1))

        if in_channels != out_channels:
            self.down = nn.Sequential(nn.Conv2d(in_channels, out_channels, 1), nn.BatchNorm2d(out_channels))
        else:
            self.down = _identity

        self.bn = nn.BatchNorm2d(out_channels)
        self.soft = nn.Softmax(-2)      # kept for attribute parity; the soft-max runs inside the HIP kernel
        self.relu = nn.ReLU()

        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                _agcn_conv_init(m)
            elif isinstance(m, nn.BatchNorm2d):
                _bn_init(m, 1)
        _bn_init(self.bn, 1e-6)
        for i in range(self.num_subset):
            _conv_branch_init(self.conv_d[i], self.num_subset)

        self._cache = None
        self._fused_tcn: Optional["Unit2D"] = None
        self.last_attention: Optional[torch.Tensor] = None   # P (N,S,V,V) of the latest forward

    # -- parameter staging ---------------------------------------------------------------
    def _has_down(self) -> bool:
        return isinstance(self.down, nn.Sequential)

    def _staged(self, device):
        """Stacked / folded device tensors for the C ABI, cached until any parameter changes."""
        key = (device, _versions(self), self.A._version, id(self.A), self.A.data_ptr())
        if self._cache is not None and self._cache["key"] == key:
            return self._cache
        with torch.no_grad():
            S = self.num_subset
            st = {"key": key}
            # the 18 embedding / expansion tensors as ONE concatenation (one launch; six stacks were six), viewed per group:
            # a stack of equally shaped tensors is the concatenation of their flattened values
            groups = [(self.conv_a, "Wa", "ba"), (self.conv_b, "Wb", "bb"), (self.conv_d, "Wd", "bd")]
            pieces, views, off = [], [], 0
            w0 = self.conv_a[0].weight
            zkey = (w0.device, w0.dtype)
            if getattr(self, "_zpad_key", None) != zkey:     # every group starts 256-byte aligned, like an allocation
                object.__setattr__(self, "_zpad", torch.zeros(64, device=w0.device, dtype=w0.dtype))
                object.__setattr__(self, "_zpad_key", zkey)
            for convs, wn, bn_ in groups:
                c0 = convs[0]
                for name, shape, ts in ((wn, (len(convs), c0.out_channels, c0.in_channels), [c.weight for c in convs]),
                                        (bn_, (len(convs), c0.out_channels), [c.bias for c in convs])):
                    n = math.prod(shape)
                    pieces += [t.reshape(-1) for t in ts]
                    views.append((name, shape, off, n))
                    pad = -n % 64
                    if pad:
                        pieces.append(self._zpad[:pad])
                    off += n + pad
            flat = torch.cat(pieces).to(device=device, dtype=torch.float32)
            for name, shape, o, n in views:
                st[name] = flat[o:o + n].view(shape)
            # A = self.A.cuda(dev) + self.PA  (model/unit_agcn.py:75-76).  The constant A is uploaded once per device and
            # version — not per call, and not per restage either: in training every step restages (PA moved), and a
            # host-to-device copy there would synchronise the host each step and forbid capturing the step in a HIP graph
            akey = (device, self.A._version, id(self.A), self.A.data_ptr())
            if getattr(self, "_A_dev_key", None) != akey:
                object.__setattr__(self, "_A_dev", self.A.to(device=device, dtype=torch.float32).contiguous())
                object.__setattr__(self, "_A_dev_key", akey)
            st["A_eff"] = (self._A_dev + self.PA.to(device)).contiguous()
            # folded running-statistics BatchNorms: what the inference kernels take — made on first use (_folded), not per
            # training step
            st["bn_scale"] = st["bn_shift"] = st["down_scale"] = st["down_shift"] = None
            if self._has_down():
                dc, dbn = self.down[0], self.down[1]
                st["Wdown"] = dc.weight.reshape(dc.out_channels, dc.in_channels).to(
                    device=device, dtype=torch.float32).contiguous()
                st["bdown"] = dc.bias.to(device=device, dtype=torch.float32).contiguous()
            else:
                st["Wdown"] = st["bdown"] = None
            assert S == st["Wd"].shape[0]
        self._cache = st
        return st

    def _folded(self, st):
        """scale / shift of the running-statistics BatchNorms (eval forward), cached in the staged set."""
        if st["bn_scale"] is None:
            with torch.no_grad():
                bn = self.bn
                st["bn_scale"], st["bn_shift"] = F.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, None, bn.eps)
                if self._has_down():
                    dbn = self.down[1]
                    st["down_scale"], st["down_shift"] = F.bn_fold(dbn.weight, dbn.bias, dbn.running_mean, dbn.running_var,
                                                                   None, dbn.eps)
        return st

    def _bn_training(self) -> bool:
        """Batch statistics or running statistics: decided by the BatchNorm sub-modules themselves, as nn.BatchNorm2d
        does in the reference (``self.bn`` at model/unit_agcn.py:60,91 and ``self.down[1]`` at :54).  Mixed modes (one
        frozen, one not) are refused."""
        main = self.bn.training
        if self._has_down() and self.down[1].training != main:
            raise NotImplementedError("unit_agcn: self.bn and self.down[1] are in different modes (one .train(), one .eval()); "
                                      "the HIP path normalises both with the same kind of statistics")
        return main

    def _fusable(self, x) -> bool:
        t = self._fused_tcn
        if t is None or getattr(self, "_is_replica", False) or t.bn.training or not self._has_down():
            return False
        if t.dim != 2 or t.stride != 1 or t.conv.in_channels != self.out_channels \
                or t.conv.out_channels != self.out_channels:
            return False
        if t.conv.weight.device != x.device:
            return False
        _, C, T, V = x.shape
        return F.stem_supported(C, self.out_channels, T, V, t.kernel_size, self.num_subset, t.math_mode)

    def forward(self, x):
        if isinstance(x, FusedStemOutput):
            x.sum()                                  # raises the explanatory error
        bn_training = self._bn_training()
        _check_input(self, x, backward_ok=True, bn_training=bn_training)   # (_forward_train refuses shapes without a backward)
        if x.shape[1] != self.in_channels:
            raise RuntimeError(f"unit_agcn: expected {self.in_channels} input channels, got {x.shape[1]}")
        if x.shape[3] != self.PA.shape[-1]:
            raise RuntimeError(f"unit_agcn: input has {x.shape[3]} joints, adjacency has {self.PA.shape[-1]}")
        st = self._staged(x.device)
        wants = _wants_grad(self, x)
        if wants and not bn_training and self._fused_tcn is not None and not getattr(self, "_warned_fusion_bypass", False):
            object.__setattr__(self, "_warned_fusion_bypass", True)
            warnings.warn("unit_agcn: eval-mode call with gradients enabled — taking the differentiable two-kernel path, not "
                          "the fused inference kernel; wrap inference in torch.no_grad() (as train_sttran.py:207-210 does)",
                          stacklevel=2)
        if not bn_training and not wants and self._fusable(x):
            if not F._is_channels_last(x):     # the permuted (N,T,V,C) batch of ST_GCN_AltFormer.py:62-68 is read in place
                x = x.contiguous()
            t = self._fused_tcn
            ts = t._staged(x.device)
            pkey = (st["key"], ts["key"], t.math_mode)
            if st.get("stem_key") != pkey:
                self._folded(st)
                st["stem_prep"] = F.stem_prepare(st["Wd"], st["bd"], st["Wdown"], st["bdown"], st["bn_scale"],
                                                 st["bn_shift"], st["down_scale"], st["down_shift"], ts["W"],
                                                 ts["scale"], t.math_mode)
                st["stem_key"] = pkey
            out, P = F.stem_forward(x, st["A_eff"], st["Wa"], st["ba"], st["Wb"], st["bb"], st["stem_prep"],
                                    ts["shift"], self.out_channels, t.kernel_size, t.math_mode, t.out_bf16,
                                    channels_last_out=t.channels_last_out)
            self.last_attention = P
            return FusedStemOutput.wrap(out, t)   # only `t` (Unit2D.forward) can take it; any other use raises
        x = x.contiguous()
        if bn_training or wants:             # (eval mode under autograd: the same kernels on the running statistics)
            return self._forward_train(x, st, frozen=not bn_training)
        self._folded(st)
        y, P = F.agcn_forward(x, st["A_eff"], st["Wa"], st["ba"], st["Wb"], st["bb"], st["Wd"], st["bd"],
                              st["Wdown"], st["bdown"], st["bn_scale"], st["bn_shift"], st["down_scale"],
                              st["down_shift"])
        self.last_attention = P
        return y


    def _train_params(self):
        """Parameters in the order _AgcnTrainFn.backward returns their gradients."""
        ps = [self.PA]
        for convs in (self.conv_a, self.conv_b, self.conv_d):
            for c in convs:
                ps += [c.weight, c.bias]
        if self._has_down():
            ps += [self.down[0].weight, self.down[0].bias, self.down[1].weight, self.down[1].bias]
        return ps + [self.bn.weight, self.bn.bias]

    def _forward_train(self, x, st, frozen=False):
        """Batch-statistics BatchNorm forward (model/unit_agcn.py:91-92 with self.training); updates running buffers.
        ``frozen`` (eval mode under autograd): running statistics, nothing updated."""
        bn = self.bn
        if not frozen and (bn.momentum is None or not bn.track_running_stats):
            raise NotImplementedError("unit_agcn: training-mode BatchNorm needs momentum and running statistics")
        if frozen and not bn.track_running_stats:
            raise NotImplementedError("unit_agcn: eval-mode BatchNorm without running statistics is not covered")
        if _wants_grad(self, x):                   # autograd: HIP forward + HIP backward (fused kernel or GEMM chain)
            N, C, T, V = x.shape
            if not F.agcn_backward_supported(N, C, self.out_channels, T, V, self.num_subset) \
                    or self.inter_c > max(self.out_channels // 4, 1):
                raise NotImplementedError(
                    "unit_agcn: the HIP backward covers V <= 64 joints and coff_embedding >= 4; this call is outside it")
            y = _AgcnTrainFn.apply(self, x, *self._train_params())
            with torch.no_grad():                  # (one launch for both counters)
                if frozen:
                    pass
                elif self._has_down():
                    torch._foreach_add_([bn.num_batches_tracked, self.down[1].num_batches_tracked], 1)
                else:
                    bn.num_batches_tracked += 1
            return y
        down_bn = None
        if self._has_down():
            d = self.down[1]
            down_bn = (d.weight, d.bias, d.running_mean, d.running_var)
        y, P = F.agcn_forward_train(x, st["A_eff"], st["Wa"], st["ba"], st["Wb"], st["bb"], st["Wd"], st["bd"],
                                    st["Wdown"], st["bdown"], (bn.weight, bn.bias, bn.running_mean, bn.running_var),
                                    down_bn, bn.momentum, bn.eps)
        with torch.no_grad():
            bn.num_batches_tracked += 1
            if down_bn is not None:
                self.down[1].num_batches_tracked += 1
        self.last_attention = P
        return y


# ----------------------------------------------------------------------------------------
class Unit2D(nn.Module):
    """Dropout -> Conv2d((k,1)) -> BatchNorm2d -> ReLU; signature of model/net.py:7-45.

    ``dim=3`` (conv along the joint axis, never used by the reference's models) runs the same
    kernel on the (T,V)-transposed tensor.  ``math_mode`` picks the contraction arithmetic
    (``stgcn_amd.MATH_F32`` default, env ``STGCN_MATH``); shapes the matrix-core kernel does not
    cover run on the fp32 VALU kernel.
    """

    def __init__(self, D_in, D_out, kernel_size, stride=1, dim=2, dropout=0, bias=True):
        super().__init__()
        pad = int((kernel_size - 1) / 2)
        if dim == 2:
            self.conv = nn.Conv2d(D_in, D_out, kernel_size=(kernel_size, 1), padding=(pad, 0),
                                  stride=(stride, 1), bias=bias)
        elif dim == 3:
            self.conv = nn.Conv2d(D_in, D_out, kernel_size=(1, kernel_size), padding=(0, pad),
                                  stride=(1, stride), bias=bias)
        else:
            raise ValueError()
        self.bn = nn.BatchNorm2d(D_out)
        self.relu = nn.ReLU()
        self.dropout = nn.Dropout(dropout, inplace=False)
        conv_init(self.conv)

        self.dim = dim
        self.kernel_size = kernel_size
        self.stride = stride
        self.math_mode = _default_math()
        self.out_bf16 = False
        self.channels_last_out = False       # set_output_layout(): (N,C,T,V) result laid out (N,T,V,C)
        self._cache = None

    def _staged(self, device):
        key = (device, _versions(self), self.math_mode)
        if self._cache is not None and self._cache["key"] == key:
            return self._cache
        with torch.no_grad():
            c, bn = self.conv, self.bn
            W = c.weight.reshape(c.out_channels, c.in_channels, self.kernel_size).to(
                device=device, dtype=torch.float32).contiguous()
            scale, shift = F.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, c.bias, bn.eps)
            st = {"key": key, "W": W, "scale": scale, "shift": shift, "packed": {}}
        self._cache = st
        return st

    def _packed(self, st, math_mode):
        if math_mode not in st["packed"]:
            st["packed"][math_mode] = F.tcn_pack(st["W"], st["scale"], math_mode)
        return st["packed"][math_mode]

    def forward(self, x):
        if isinstance(x, FusedStemOutput):
            if getattr(x, "_stgcn_consumer", None) is not self:
                raise RuntimeError("Unit2D: received the fused stem result of ANOTHER Unit2D (enable_stem_fusion pairs one "
                                   "unit_agcn with one Unit2D); call disable_stem_fusion on that unit_agcn")
            return x.unwrap()             # tcn0(gcn0(x)), computed by the fused stem kernel in unit_agcn.forward
        bn_training = self.bn.training    # nn.BatchNorm2d's own flag, like the reference's self.bn(...) (model/net.py:52)
        _check_input(self, x, backward_ok=self.dim == 2, bn_training=bn_training)
        if x.shape[1] != self.conv.in_channels:
            raise RuntimeError(f"Unit2D: expected {self.conv.in_channels} input channels, got {x.shape[1]}")
        x = x.contiguous()
        N, Cin, T, V = x.shape
        mode = self.math_mode
        if self.dim == 3:                    # conv along the joints (model/net.py:28-36), read in place by the plain-FMA kernel
            mode = MATH_F32_VALU
        elif mode != MATH_F32_VALU and not F.tcn_supported(Cin, self.conv.out_channels, T, V, self.kernel_size,
                                                           self.stride, mode):
            mode = MATH_F32_VALU
        if self.dropout.p > 0 and self.dropout.training:
            x = self.dropout(x)              # torch's RNG-driven op (the stem always uses p = 0, model/net.py:45)
        wants = _wants_grad(self, x)
        if bn_training or wants:             # (eval mode under autograd: the same kernels on the running statistics)
            bn = self.bn
            if self.dim == 3:                # batch statistics over a joint-axis conv (no model builds it): the frame-axis
                x = x.transpose(2, 3).contiguous()   # kernels on the transposed copy; inference reads x in place (below)
                mode = self.math_mode
                if mode != MATH_F32_VALU and not F.tcn_supported(Cin, self.conv.out_channels, V, T, self.kernel_size,
                                                                 self.stride, mode):
                    mode = MATH_F32_VALU
            if bn_training and (bn.momentum is None or not bn.track_running_stats):
                raise NotImplementedError("Unit2D: training-mode BatchNorm needs momentum and running statistics")
            if not bn_training and not bn.track_running_stats:
                raise NotImplementedError("Unit2D: eval-mode BatchNorm without running statistics is not covered")
            c = self.conv
            if wants:                        # autograd: HIP forward + HIP backward (tcn_backward.hip)
                y = _Unit2DTrainFn.apply(x, c.weight, c.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                         self.stride, mode, bn.momentum if bn.momentum is not None else 0.0, bn.eps,
                                         not bn_training)
            else:
                W = c.weight.detach().reshape(c.out_channels, c.in_channels, self.kernel_size).contiguous()
                y = F.tcn_forward_train(x, W, c.bias, (bn.weight, bn.bias, bn.running_mean, bn.running_var), self.stride,
                                        mode, bn.momentum, bn.eps)
            if bn_training:
                with torch.no_grad():
                    bn.num_batches_tracked += 1
            if self.dim == 3:
                y = y.transpose(2, 3).contiguous()
        else:
            st = self._staged(x.device)
            y = F.tcn_forward_packed(x, self._packed(st, mode), st["shift"], self.conv.out_channels,
                                     self.kernel_size, self.stride, mode, self.out_bf16, along_v=self.dim == 3)
        if self.channels_last_out:           # (only the fused stem writes this layout natively)
            y = y.contiguous(memory_format=torch.channels_last)
        return y


# ----------------------------------------------------------------------------------------
def enable_stem_fusion(gcn: unit_agcn, tcn: Unit2D) -> None:
    """Make ``tcn(gcn(x))`` run as ONE fused kernel pair (attention + fused stem).

    Both modules stay where they are (state_dict keys and the caller's forward are untouched):
    ``gcn.forward`` computes the whole stem and returns it as a ``FusedStemOutput`` — a tensor type on
    which every operation raises — and ``tcn.forward`` unwraps it.  Anything else that touches gcn0's
    return value (a forward hook, a residual branch as in TCN_GCN_unit, a cast) therefore fails loudly
    instead of reading tcn0's activation as gcn0's.  Falls back to the two-stage path whenever the fused
    kernel does not cover the shape, under nn.DataParallel replicas, or with batch-statistics BatchNorm.
    """
    object.__setattr__(gcn, "_fused_tcn", tcn)


def disable_stem_fusion(gcn: unit_agcn) -> None:
    object.__setattr__(gcn, "_fused_tcn", None)


def set_output_layout(module: nn.Module, layout: str) -> None:
    """'channels_last': every Unit2D below returns its (N,C,T,V) result laid out (N,T,V,C) in memory (torch.channels_last
    strides), so `rearrange(x, 'b c f p -> (b f) p c')` (model_ST.py:152) / `'b c f p -> (b p) f c'` (model_TS.py:161)
    are views.  The fused stem kernel writes that layout directly; 'contiguous' restores the default."""
    if layout not in ("channels_last", "contiguous"):
        raise ValueError(f"unknown layout {layout!r}")
    for sub in module.modules():
        if isinstance(sub, Unit2D):
            sub.channels_last_out = layout == "channels_last"


def set_math_mode(module: nn.Module, mode) -> None:
    """Set the temporal-conv arithmetic ('f32' | 'bf16x3' | 'bf16' | 'f32_valu') on every Unit2D below."""
    m = _MATH_NAMES[mode] if isinstance(mode, str) else int(mode)
    for sub in module.modules():
        if isinstance(sub, Unit2D):
            sub.math_mode = m
