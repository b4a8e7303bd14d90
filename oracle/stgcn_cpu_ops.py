"""CPU baseline of the ST-GCN stem built on torch's LIBRARY ops (oneDNN conv, native batch-norm).

TEST INFRASTRUCTURE ONLY — same rule as ``stgcn_oracle.py``: imported by ``tests/`` and by the
``cpu_baseline`` leg of ``bench.py``, never by the product.

Why a second restatement: ``stgcn_oracle.py`` is written with einsum only (so that it is dtype-parametric
and independent of nn.Conv2d/nn.BatchNorm2d) and is therefore a slow yardstick — timing it as "the CPU"
overstates the GPU/CPU ratio.  The reference itself runs ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``matmul`` /
``Softmax`` (``model/unit_agcn.py:47-62,81-93``, ``model/net.py:21-54``); this file restates the same forward
with the same op mix through ``torch.nn.functional``, so that its wall time is what the reference's own CPU
forward costs on the box (SURVEY.md §6 measured the imported reference at 107 clips/s on 8 vCPUs; this file
is held to the oracle's values in ``tests/test_oracle_golden.py`` and to that speed class in bench.py).

Takes the same ``AgcnParams`` / ``TcnParams`` containers as the oracle.  fp32, eval mode (running statistics).
"""
from __future__ import annotations

import torch
import torch.nn.functional as Fn

from .stgcn_oracle import BN_EPS, AgcnParams, BNParams, TcnParams


def _conv1x1(x, w, b):
    return Fn.conv2d(x, w.reshape(w.shape[0], w.shape[1], 1, 1), b)


def _bn_eval(x, p: BNParams):
    return Fn.batch_norm(x, p.running_mean, p.running_var, p.weight, p.bias, False, 0.0, BN_EPS)


def agcn_forward_ops(x: torch.Tensor, p: AgcnParams) -> torch.Tensor:
    """Eval forward of unit_agcn with library ops; op for op the mix of model/unit_agcn.py:73-93."""
    N, C, T, V = x.shape
    A = p.A + p.PA                                                       # :75-76
    y = None
    for i in range(p.num_subset):                                        # :80
        a = _conv1x1(x, p.conv_a_w[i], p.conv_a_b[i])                    # :81 embedding, (N,inter,T,V)
        a = a.permute(0, 3, 1, 2).reshape(N, V, p.inter_c * T)           #     joints leading, (channel,time) flattened
        b = _conv1x1(x, p.conv_b_w[i], p.conv_b_b[i]).reshape(N, p.inter_c * T, V)     # :83
        att = torch.softmax(torch.matmul(a, b) / a.shape[-1], dim=-2) + A[i]           # :84-85
        u = torch.matmul(x.reshape(N, C * T, V), att).reshape(N, C, T, V)              # :87-88
        z = _conv1x1(u, p.conv_d_w[i], p.conv_d_b[i])                    # :88
        y = z if y is None else z + y                                    # :89
    y = _bn_eval(y, p.bn)                                                # :91
    if p.down_w is not None:
        y = y + _bn_eval(_conv1x1(x, p.down_w, p.down_b), p.down_bn)     # :51-55, :92
    else:
        y = y + x                                                        # :57-58
    return torch.relu(y)                                                 # :93


def tcn_forward_ops(x: torch.Tensor, p: TcnParams) -> torch.Tensor:
    """Eval forward of Unit2D(dim=2, dropout=0) with library ops (model/net.py:47-57)."""
    K = p.conv_w.shape[2]
    y = Fn.conv2d(x, p.conv_w.unsqueeze(-1), p.conv_b, stride=(p.stride, 1), padding=(int((K - 1) / 2), 0))
    return torch.relu(_bn_eval(y, p.bn))


def stem_forward_ops(x: torch.Tensor, g: AgcnParams, t: TcnParams) -> torch.Tensor:
    """tcn0(gcn0(x)) (model/AltFormer/ST_GCN_AltFormer.py:70-72) on library ops."""
    return tcn_forward_ops(agcn_forward_ops(x, g), t)
