#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE IMPORTED REFERENCE.

Runs only in the build container (needs /root/reference, which never travels to
the GPU box).  The reference's modules are imported as they lie, executed on
CPU, and only *data* (inputs, parameters, outputs) is written here as .npz
(numpy arrays, no pickle).  Nothing of the reference's source is copied.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Harness-side shims (the reference tree is untouched):
  * ``model/unit_agcn.py:75`` calls ``self.A.cuda(x.get_device())`` which raises
    on a CPU tensor (device index -1) -> ``torch.Tensor.cuda`` is made the
    identity for the duration of a call.

Parameter randomisation: default init sets ``bn.weight = 1e-6`` (unit_agcn.py:69)
which hides the main branch, and all biases 0; so every case overwrites PA,
biases, BN affine and BN running statistics with seeded random values (the
"as constructed" case keeps the defaults on purpose).
"""
import contextlib
import os
import sys

import numpy as np
import torch

REF = os.environ.get("STGCN_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from model.unit_agcn import unit_agcn            # noqa: E402  (reference)
from model.net import Unit2D, import_class       # noqa: E402  (reference)


@contextlib.contextmanager
def cuda_is_identity():
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig


def spatial_A(graph_name):
    with contextlib.redirect_stdout(open(os.devnull, "w")):
        G = import_class(graph_name)
    return torch.from_numpy(G(labeling_mode="spatial").A.astype(np.float32))


def randomise_bn(bn, g):
    C = bn.num_features
    with torch.no_grad():
        bn.weight.copy_(torch.rand(C, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(C, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(C, generator=g) * 0.3)
        bn.running_var.copy_(torch.rand(C, generator=g) * 1.5 + 0.25)


def randomise_gcn(gcn, g, fixed_A):
    """Break the PA/A storage alias the way ``.cuda()`` does in the trainers, then randomise."""
    with torch.no_grad():
        gcn.PA.data = torch.randn(gcn.PA.shape, generator=g) * 0.05      # new storage: self.A keeps its value
        gcn.A = fixed_A.clone()
        for convs in (gcn.conv_a, gcn.conv_b, gcn.conv_d):
            for c in convs:
                c.bias.copy_(torch.randn(c.bias.shape, generator=g) * 0.1)
        # make the attention logits O(1) so the softmax is not trivially uniform
        for convs in (gcn.conv_a, gcn.conv_b):
            for c in convs:
                c.weight.mul_(3.0)
        randomise_bn(gcn.bn, g)
        if isinstance(gcn.down, torch.nn.Sequential):
            gcn.down[0].bias.copy_(torch.randn(gcn.down[0].bias.shape, generator=g) * 0.1)
            randomise_bn(gcn.down[1], g)


def randomise_tcn(tcn, g):
    with torch.no_grad():
        if tcn.conv.bias is not None:
            tcn.conv.bias.copy_(torch.randn(tcn.conv.bias.shape, generator=g) * 0.1)
        randomise_bn(tcn.bn, g)


def sd_np(mod, prefix):
    return {prefix + k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}


def run(mod, x, train):
    """Returns (output, state_dict after the call) without disturbing ``mod``."""
    import copy
    m = copy.deepcopy(mod)
    if hasattr(mod, "A"):
        m.A = mod.A.clone()
    m.train(train)
    soft = []
    hook = None
    if hasattr(m, "soft"):
        hook = m.soft.register_forward_hook(lambda _m, _i, o: soft.append(o.detach().clone()))
    with cuda_is_identity(), torch.no_grad():
        y = m(x)
    if hook is not None:
        hook.remove()
    return y, m, soft


def sample_idx(numel, count, seed):
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(numel, size=min(count, numel), replace=False)).astype(np.int64)


def gcn_case(name, graph, cin, cout, N, T, seed, fixed="quirk", randomise=True, sample=None):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    A_true = spatial_A(graph)
    A_arg = A_true.clone()
    gcn = unit_agcn(cin, cout, A_arg)                 # overwrites A_arg with 1e-6 (aliasing, unit_agcn.py:37-39)
    A_after_ctor = A_arg.clone()
    fixed_A = A_after_ctor if fixed == "quirk" else A_true
    if randomise:
        randomise_gcn(gcn, g, fixed_A)
    V = A_true.shape[-1]
    x = torch.randn(N, cin, T, V, generator=g)
    out = {"x": x.numpy(), "A_true": A_true.numpy(), "A_after_ctor": A_after_ctor.numpy(),
           "A_fixed": gcn.A.detach().numpy().copy(), "cin": cin, "cout": cout}
    out.update(sd_np(gcn, "gcn."))
    for train in (False, True):
        y, m, soft = run(gcn, x, train)
        tag = "train" if train else "eval"
        A_eff = (gcn.A + gcn.PA).detach()
        P = torch.stack([s + A_eff[i] for i, s in enumerate(soft)], dim=1)
        out[f"P_{tag}"] = P.numpy()
        yn = y.numpy()
        if sample:
            idx = sample_idx(yn.size, sample, seed)
            out[f"y_{tag}_idx"] = idx
            out[f"y_{tag}_val"] = yn.reshape(-1)[idx]
            out[f"y_{tag}_absmax"] = np.abs(yn).max()
            out[f"y_{tag}_sum"] = yn.astype(np.float64).sum()
        else:
            out[f"y_{tag}"] = yn
        if train:
            for k, v in m.state_dict().items():
                if "running" in k or "num_batches" in k:
                    out["after_train.gcn." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def tcn_case(name, cin, cout, K, stride, N, T, V, seed, bias=True):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    tcn = Unit2D(cin, cout, kernel_size=K, stride=stride, bias=bias)
    randomise_tcn(tcn, g)
    x = torch.randn(N, cin, T, V, generator=g)
    out = {"x": x.numpy(), "stride": stride, "K": K}
    out.update(sd_np(tcn, "tcn."))
    for train in (False, True):
        y, m, _ = run(tcn, x, train)
        tag = "train" if train else "eval"
        out[f"y_{tag}"] = y.numpy()
        if train:
            for k, v in m.state_dict().items():
                if "running" in k or "num_batches" in k:
                    out["after_train.tcn." + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def stem_case(name, graph, N, T, seed, sample):
    """tcn0(gcn0(x)) exactly as ST_GCN_AltFormer.py:43-50,64-72 builds and calls them."""
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1)
    A_true = spatial_A(graph)
    A_arg = A_true.clone()
    gcn = unit_agcn(3, 128, A_arg)
    tcn = Unit2D(128, 128, kernel_size=9)
    randomise_gcn(gcn, g, A_arg.clone())
    randomise_tcn(tcn, g)
    V = A_true.shape[-1]
    skel = torch.randn(N, T, V, 3, generator=g)                    # loader layout (N,T,V,3)
    x = skel.permute(0, 3, 1, 2).contiguous()
    out = {"skeleton": skel.numpy(), "A_fixed": gcn.A.numpy().copy()}
    out.update(sd_np(gcn, "gcn."))
    out.update(sd_np(tcn, "tcn."))
    for train in (False, True):
        tag = "train" if train else "eval"
        y, mg, soft = run(gcn, x, train)
        z, mt, _ = run(tcn, y, train)
        A_eff = (gcn.A + gcn.PA).detach()
        out[f"P_{tag}"] = torch.stack([s + A_eff[i] for i, s in enumerate(soft)], dim=1).numpy()
        for nm, arr in (("y", y.numpy()), ("z", z.numpy())):
            idx = sample_idx(arr.size, sample, seed + (7 if nm == "z" else 0))
            out[f"{nm}_{tag}_idx"] = idx
            out[f"{nm}_{tag}_val"] = arr.reshape(-1)[idx]
            out[f"{nm}_{tag}_absmax"] = np.abs(arr).max()
            out[f"{nm}_{tag}_sum"] = arr.astype(np.float64).sum()
            out[f"{nm}_{tag}_sumsq"] = (arr.astype(np.float64) ** 2).sum()
        if train:
            for pre, m in (("gcn.", mg), ("tcn.", mt)):
                for k, v in m.state_dict().items():
                    if "running" in k:
                        out["after_train." + pre + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def graph_fixture():
    out = {}
    for gname in ("graph.SHRE", "graph.LMDHG"):
        with contextlib.redirect_stdout(open(os.devnull, "w")):
            G = import_class(gname)
        for mode in ("uniform", "distance*", "distance", "spatial", "DAD", "DLD"):
            out[f"{gname.split('.')[1]}/{mode}"] = G(labeling_mode=mode).A
    np.savez_compressed(os.path.join(HERE, "graphs.npz"), **out)
    print("wrote graphs")


def init_stats_fixture():
    """Shapes/keys and init statistics of freshly constructed reference modules."""
    torch.manual_seed(0)
    A = spatial_A("graph.SHRE")
    gcn = unit_agcn(3, 128, A)
    tcn = Unit2D(128, 128, kernel_size=9)
    out = {}
    for pre, m in (("gcn.", gcn), ("tcn.", tcn)):
        for k, v in m.state_dict().items():
            out["shape:" + pre + k] = np.array(v.shape, dtype=np.int64)
            v = v.float()
            out["std:" + pre + k] = np.array(v.std().item() if v.numel() > 1 else 0.0)
            out["mean:" + pre + k] = np.array(v.mean().item())
    np.savez_compressed(os.path.join(HERE, "init_stats.npz"), **out)
    print("wrote init_stats")


if __name__ == "__main__":
    graph_fixture()
    init_stats_fixture()
    gcn_case("gcn_shre_3_128_quirk", "graph.SHRE", 3, 128, N=2, T=20, seed=11, fixed="quirk")
    gcn_case("gcn_shre_3_128_trueA", "graph.SHRE", 3, 128, N=2, T=20, seed=12, fixed="true")
    gcn_case("gcn_shre_3_128_default_init", "graph.SHRE", 3, 128, N=2, T=12, seed=13, randomise=False)
    gcn_case("gcn_lmdhg_3_128", "graph.LMDHG", 3, 128, N=2, T=16, seed=14, fixed="true")
    gcn_case("gcn_shre_64_64_identity", "graph.SHRE", 64, 64, N=2, T=12, seed=15, fixed="true")
    gcn_case("gcn_shre_64_128", "graph.SHRE", 64, 128, N=2, T=12, seed=16, fixed="quirk")
    gcn_case("gcn_shre_3_128_T180", "graph.SHRE", 3, 128, N=2, T=180, seed=17, fixed="quirk", sample=20000)
    tcn_case("tcn_128_128_k9", 128, 128, 9, 1, N=2, T=20, V=22, seed=21)
    tcn_case("tcn_64_128_k9_s2", 64, 128, 9, 2, N=2, T=21, V=22, seed=22)
    tcn_case("tcn_64_128_k1_s2", 64, 128, 1, 2, N=2, T=20, V=22, seed=23)
    tcn_case("tcn_128_128_k9_v46", 128, 128, 9, 1, N=1, T=16, V=46, seed=24)
    tcn_case("tcn_32_64_k5_nobias", 32, 64, 5, 1, N=2, T=9, V=22, seed=25, bias=False)
    stem_case("stem_shre_T180", "graph.SHRE", N=2, T=180, seed=31, sample=40000)
    stem_case("stem_lmdhg_T200", "graph.LMDHG", N=1, T=200, seed=32, sample=40000)
    stem_case("stem_shre_T500", "graph.SHRE", N=1, T=500, seed=33, sample=40000)
