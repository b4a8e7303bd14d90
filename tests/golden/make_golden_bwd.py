#!/usr/bin/env python3
"""Gradient fixtures FROM THE IMPORTED REFERENCE (build container only; see make_golden.py for the rules: the
reference's modules are imported as they lie, run on CPU, only data is written — inputs, parameters, the cotangent and
the gradients torch.autograd derives through the reference's own forward in .train()).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bwd.py

The cotangent dL/dz is zero where the reference output lies within 1e-4*max of the ReLU kink (gradients are
discontinuous there; see tests/test_gpu_parity.py::_kink_free_cotangent), so the fixtures are reproducible by an
fp32 implementation whose forward differs in the last bits.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import make_golden as mg                         # noqa: E402  (imports the reference's unit_agcn / Unit2D)


def cotangent(z, g):
    G = torch.randn(z.shape, generator=g)
    return G * (z.detach() > 1e-4 * z.detach().abs().max()).float()


def grads_np(mod, prefix):
    return {"grad." + prefix + k: p.grad.detach().numpy().copy() for k, p in mod.named_parameters() if p.grad is not None}


def stem_bwd_case(name, graph, N, T, seed):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    A_arg = mg.spatial_A(graph)
    gcn = mg.unit_agcn(3, 128, A_arg)
    tcn = mg.Unit2D(128, 128, kernel_size=9)
    mg.randomise_gcn(gcn, g, A_arg.clone())
    mg.randomise_tcn(tcn, g)
    V = A_arg.shape[-1]
    x = torch.randn(N, 3, T, V, generator=g)
    out = {"x": x.numpy(), "A_fixed": gcn.A.numpy().copy()}
    out.update(mg.sd_np(gcn, "gcn."))
    out.update(mg.sd_np(tcn, "tcn."))
    gcn.train(); tcn.train()
    with mg.cuda_is_identity():
        h = gcn(x)
        z = tcn(h)
    G = cotangent(z, g)
    (z * G).sum().backward()
    out["z"] = z.detach().numpy()
    out["G"] = G.numpy()
    out.update(grads_np(gcn, "gcn."))
    out.update(grads_np(tcn, "tcn."))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if k.startswith("grad.")})


def tcn_bwd_case(name, cin, cout, K, stride, N, T, V, seed):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    tcn = mg.Unit2D(cin, cout, kernel_size=K, stride=stride)
    mg.randomise_tcn(tcn, g)
    x = torch.randn(N, cin, T, V, generator=g).requires_grad_(True)
    out = {"x": x.detach().numpy()}
    out.update(mg.sd_np(tcn, "tcn."))
    tcn.train()
    z = tcn(x)
    G = cotangent(z, g)
    (z * G).sum().backward()
    out["z"] = z.detach().numpy()
    out["G"] = G.numpy()
    out["grad.x"] = x.grad.numpy()
    out.update(grads_np(tcn, "tcn."))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def gcn_bwd_case(name, graph, cin, cout, N, T, seed):
    """unit_agcn(cin, cout) alone, as the deeper TCN_GCN_unit layers use it (model/ST_TR/ST_TR_new.py:355-372): the
    input requires a gradient; cin == cout gives the identity residual (unit_agcn.py:57-58)."""
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    A_arg = mg.spatial_A(graph)
    gcn = mg.unit_agcn(cin, cout, A_arg)
    mg.randomise_gcn(gcn, g, A_arg.clone())
    V = A_arg.shape[-1]
    x = torch.randn(N, cin, T, V, generator=g).requires_grad_(True)
    out = {"x": x.detach().numpy(), "A_fixed": gcn.A.numpy().copy()}
    out.update(mg.sd_np(gcn, "gcn."))
    gcn.train()
    with mg.cuda_is_identity():
        y = gcn(x)
    G = cotangent(y, g)
    (y * G).sum().backward()
    out["y"] = y.detach().numpy()
    out["G"] = G.numpy()
    out["grad.x"] = x.grad.numpy()
    out.update(grads_np(gcn, "gcn."))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


if __name__ == "__main__":
    gcn_bwd_case("bwd_gcn_shre_64_64_identity", "graph.SHRE", 64, 64, N=2, T=12, seed=43)
    gcn_bwd_case("bwd_gcn_shre_64_128", "graph.SHRE", 64, 128, N=2, T=12, seed=44)
    stem_bwd_case("bwd_stem_shre_T20", "graph.SHRE", N=2, T=20, seed=41)
    tcn_bwd_case("bwd_tcn_64_128_k9_s2", 64, 128, 9, 2, N=2, T=21, V=22, seed=42)
