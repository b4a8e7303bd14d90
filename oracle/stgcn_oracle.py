"""CPU oracle for the ST-GCN stem (adaptive graph conv -> temporal conv block).

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The shipped path (``st-gcn-altformer_amd/``)
never imports anything from ``oracle/`` and raises when the HIP library is
missing.

What it restates (reference = zjtggssg/ST-GCN-AltFormer, paths relative to the
reference root):

* ``agcn_forward``   follows ``model/unit_agcn.py:73-93`` (forward of
  ``unit_agcn``): per-subset 1x1 embeddings (:81-83), Gram over (channel,time)
  scaled by 1/(inter_c*T) and soft-maxed over dim -2 (:84), plus the adjacency
  (:85, with ``A = self.A + self.PA`` from :75-76), per-frame aggregation and
  1x1 expansion (:87-89), BN, residual ``down`` branch and ReLU (:91-93).
* ``tcn_forward``    follows ``model/net.py:47-57`` (forward of ``Unit2D``):
  dropout (identity at p=0 / eval) -> Conv2d (k,1), pad ((k-1)//2, 0),
  stride (s,1) (:21-27) -> BatchNorm2d -> ReLU.
* ``batch_norm``     is torch ``BatchNorm2d`` semantics as used at
  ``model/unit_agcn.py:54,60`` and ``model/net.py:40``: eps 1e-5, momentum 0.1,
  biased variance for normalisation, unbiased for the running update.

Parity pinning: the reference ships no tests or golden vectors,
so this restatement is pinned by fixtures generated *from the imported
reference itself* in the build container: ``tests/golden/make_golden.py``
writes ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py`` checks this
file against them.

Everything is plain ``torch`` on CPU, written with einsum/matmul only (no
nn.Conv2d / nn.BatchNorm2d), and is dtype-parametric: run it in float32 to
mirror the reference's arithmetic, or float64 to get a tighter yardstick.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

BN_EPS = 1e-5          # torch.nn.BatchNorm2d default, never overridden by the reference
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------
# parameter containers (plain tensors; keys mirror the reference state_dict)
# ----------------------------------------------------------------------------
@dataclass
class BNParams:
    weight: torch.Tensor
    bias: torch.Tensor
    running_mean: torch.Tensor
    running_var: torch.Tensor

    def to(self, dtype):
        return BNParams(*(t.to(dtype) for t in
                          (self.weight, self.bias, self.running_mean, self.running_var)))


@dataclass
class AgcnParams:
    """Everything ``unit_agcn.forward`` reads (model/unit_agcn.py:35-62)."""
    A: torch.Tensor                     # (S,V,V) the non-learned term, self.A
    PA: torch.Tensor                    # (S,V,V) learned term
    conv_a_w: List[torch.Tensor]        # S x (inter_c, C)
    conv_a_b: List[torch.Tensor]        # S x (inter_c,)
    conv_b_w: List[torch.Tensor]
    conv_b_b: List[torch.Tensor]
    conv_d_w: List[torch.Tensor]        # S x (Cout, C)
    conv_d_b: List[torch.Tensor]
    bn: BNParams
    down_w: Optional[torch.Tensor] = None   # (Cout, C) or None when C == Cout
    down_b: Optional[torch.Tensor] = None
    down_bn: Optional[BNParams] = None

    @property
    def num_subset(self) -> int:
        return len(self.conv_d_w)

    @property
    def inter_c(self) -> int:
        return self.conv_a_w[0].shape[0]

    def to(self, dtype):
        cv = lambda ts: [t.to(dtype) for t in ts]
        return AgcnParams(
            A=self.A.to(dtype), PA=self.PA.to(dtype),
            conv_a_w=cv(self.conv_a_w), conv_a_b=cv(self.conv_a_b),
            conv_b_w=cv(self.conv_b_w), conv_b_b=cv(self.conv_b_b),
            conv_d_w=cv(self.conv_d_w), conv_d_b=cv(self.conv_d_b),
            bn=self.bn.to(dtype),
            down_w=None if self.down_w is None else self.down_w.to(dtype),
            down_b=None if self.down_b is None else self.down_b.to(dtype),
            down_bn=None if self.down_bn is None else self.down_bn.to(dtype))


@dataclass
class TcnParams:
    """Everything ``Unit2D.forward`` reads (model/net.py:21-45)."""
    conv_w: torch.Tensor                # (Cout, Cin, K) temporal taps (the trailing 1 squeezed)
    conv_b: Optional[torch.Tensor]      # (Cout,) or None when bias=False
    bn: BNParams
    stride: int = 1

    def to(self, dtype):
        return TcnParams(self.conv_w.to(dtype),
                         None if self.conv_b is None else self.conv_b.to(dtype),
                         self.bn.to(dtype), self.stride)


def agcn_params_from_state(sd: Dict[str, torch.Tensor], A: torch.Tensor,
                           prefix: str = "") -> AgcnParams:
    """Build AgcnParams from a reference-style state_dict (keys of SURVEY §8b)."""
    g = lambda k: torch.as_tensor(sd[prefix + k]).detach().clone()
    S = 0
    while f"{prefix}conv_d.{S}.weight" in sd:
        S += 1
    sq = lambda w: w.reshape(w.shape[0], w.shape[1])
    bn = lambda p: BNParams(g(p + ".weight"), g(p + ".bias"),
                            g(p + ".running_mean"), g(p + ".running_var"))
    has_down = f"{prefix}down.0.weight" in sd
    return AgcnParams(
        A=torch.as_tensor(A).detach().clone(), PA=g("PA"),
        conv_a_w=[sq(g(f"conv_a.{i}.weight")) for i in range(S)],
        conv_a_b=[g(f"conv_a.{i}.bias") for i in range(S)],
        conv_b_w=[sq(g(f"conv_b.{i}.weight")) for i in range(S)],
        conv_b_b=[g(f"conv_b.{i}.bias") for i in range(S)],
        conv_d_w=[sq(g(f"conv_d.{i}.weight")) for i in range(S)],
        conv_d_b=[g(f"conv_d.{i}.bias") for i in range(S)],
        bn=bn("bn"),
        down_w=sq(g("down.0.weight")) if has_down else None,
        down_b=g("down.0.bias") if has_down else None,
        down_bn=bn("down.1") if has_down else None)


def tcn_params_from_state(sd: Dict[str, torch.Tensor], stride: int = 1,
                          prefix: str = "") -> TcnParams:
    w = torch.as_tensor(sd[prefix + "conv.weight"]).detach().clone()
    assert w.shape[3] == 1, "oracle covers Unit2D(dim=2) only (the only form any reference model builds)"
    b = sd.get(prefix + "conv.bias")
    return TcnParams(
        conv_w=w[..., 0],
        conv_b=None if b is None else torch.as_tensor(b).detach().clone(),
        bn=BNParams(*(torch.as_tensor(sd[prefix + "bn." + k]).detach().clone()
                      for k in ("weight", "bias", "running_mean", "running_var"))),
        stride=stride)


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def pointwise_conv(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]) -> torch.Tensor:
    """1x1 Conv2d on (N,C,T,V): channel mixing only (model/unit_agcn.py:47-49,53)."""
    y = torch.einsum("oc,nctv->notv", w, x)
    if b is not None:
        y = y + b.view(1, -1, 1, 1)
    return y


def batch_norm(x: torch.Tensor, p: BNParams, training: bool,
               stats_out: Optional[dict] = None) -> torch.Tensor:
    """BatchNorm2d over (N,T,V) per channel.

    eval : (x - running_mean) / sqrt(running_var + eps) * weight + bias
    train: same with the batch mean / biased batch variance; the running
           buffers that torch would write (momentum 0.1, unbiased variance)
           are returned through ``stats_out`` instead of being mutated.
    """
    if training:
        n = x.shape[0] * x.shape[2] * x.shape[3]
        mean = x.mean(dim=(0, 2, 3))
        var_b = x.var(dim=(0, 2, 3), unbiased=False)
        if stats_out is not None:
            var_u = var_b * (n / max(n - 1, 1))
            stats_out["batch_mean"] = mean
            stats_out["batch_var"] = var_b
            stats_out["running_mean"] = (1 - BN_MOMENTUM) * p.running_mean + BN_MOMENTUM * mean
            stats_out["running_var"] = (1 - BN_MOMENTUM) * p.running_var + BN_MOMENTUM * var_u
    else:
        mean, var_b = p.running_mean, p.running_var
    inv = torch.rsqrt(var_b + BN_EPS)
    return (x - mean.view(1, -1, 1, 1)) * (inv * p.weight).view(1, -1, 1, 1) + p.bias.view(1, -1, 1, 1)


def adjacency_attention(x: torch.Tensor, wa, ba, wb, bb, inter_c: int) -> torch.Tensor:
    """Soft-maxed Gram matrix of one subset (model/unit_agcn.py:81-84).

    S[n,v,w] = sum_{c<inter_c, t<T} a[n,c,t,v] * b[n,c,t,w] / (inter_c*T),
    softmax over v (dim -2 of the (N,V,V) matrix), i.e. every column sums to 1.
    """
    T = x.shape[2]
    a = pointwise_conv(x, wa, ba)
    b = pointwise_conv(x, wb, bb)
    S = torch.einsum("nctv,nctw->nvw", a, b) / float(inter_c * T)
    return torch.softmax(S, dim=-2)


def agcn_forward(x: torch.Tensor, p: AgcnParams, training: bool = False,
                 aux: Optional[dict] = None) -> torch.Tensor:
    """Forward of unit_agcn on x (N,C,T,V).  ``aux`` receives P (N,S,V,V) and BN stats."""
    N, C, T, V = x.shape
    A_eff = p.A.to(x.dtype) + p.PA.to(x.dtype)                      # :75-76
    y = None
    Ps = []
    for i in range(p.num_subset):                                   # :80
        P = adjacency_attention(x, p.conv_a_w[i], p.conv_a_b[i],
                                p.conv_b_w[i], p.conv_b_b[i], p.inter_c) + A_eff[i]   # :81-85
        Ps.append(P)
        u = torch.einsum("nctv,nvw->nctw", x, P)                    # :87-88 (x viewed (N,C*T,V) @ P)
        z = pointwise_conv(u, p.conv_d_w[i], p.conv_d_b[i])         # :88
        y = z if y is None else z + y                               # :89
    st_main, st_down = {}, {}
    y = batch_norm(y, p.bn, training, st_main)                      # :91
    if p.down_w is not None:                                        # :51-55
        r = batch_norm(pointwise_conv(x, p.down_w, p.down_b), p.down_bn, training, st_down)
    else:                                                           # :57-58 identity
        r = x
    if aux is not None:
        aux["P"] = torch.stack(Ps, dim=1)
        aux["bn"] = st_main
        aux["down_bn"] = st_down
    return torch.relu(y + r)                                        # :92-93


def temporal_conv(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: int) -> torch.Tensor:
    """Conv2d with kernel (K,1), padding ((K-1)//2, 0), stride (s,1) (model/net.py:17-27).

    out[n,o,t,v] = b[o] + sum_{c,k} w[o,c,k] * x[n,c,t*s + k - pad, v], zero outside [0,T).
    """
    N, C, T, V = x.shape
    Cout, Cin, K = w.shape
    assert Cin == C
    pad = int((K - 1) / 2)
    T_out = (T + 2 * pad - K) // stride + 1
    xp = torch.zeros(N, C, T + 2 * pad, V, dtype=x.dtype)
    xp[:, :, pad:pad + T] = x
    out = torch.zeros(N, Cout, T_out, V, dtype=x.dtype)
    for k in range(K):
        sl = xp[:, :, k:k + (T_out - 1) * stride + 1:stride]       # (N,C,T_out,V)
        out = out + torch.einsum("oc,nctv->notv", w[:, :, k], sl)
    if b is not None:
        out = out + b.view(1, -1, 1, 1)
    return out


def tcn_forward(x: torch.Tensor, p: TcnParams, training: bool = False,
                aux: Optional[dict] = None) -> torch.Tensor:
    """Forward of Unit2D(dim=2, dropout=0) on x (N,Cin,T,V) (model/net.py:47-57)."""
    st = {}
    y = batch_norm(temporal_conv(x, p.conv_w, p.conv_b, p.stride), p.bn, training, st)
    if aux is not None:
        aux["bn"] = st
    return torch.relu(y)


def stem_forward(x: torch.Tensor, g: AgcnParams, t: TcnParams, training: bool = False,
                 aux: Optional[dict] = None) -> torch.Tensor:
    """tcn0(gcn0(x)) as at model/AltFormer/ST_GCN_AltFormer.py:70-72."""
    ga, ta = {}, {}
    y = agcn_forward(x, g, training, ga)
    z = tcn_forward(y, t, training, ta)
    if aux is not None:
        aux["gcn"] = ga
        aux["tcn"] = ta
        aux["gcn_out"] = y
    return z


def caller_layout(skel: torch.Tensor) -> torch.Tensor:
    """(N,T,V,3) skeleton batch -> contiguous (N,3,T,V) (ST_GCN_AltFormer.py:64-68)."""
    return skel.permute(0, 3, 1, 2).contiguous()
