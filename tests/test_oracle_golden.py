"""Pin the CPU oracle (oracle/) against the golden vectors captured from the imported reference.

These fixtures were produced by tests/golden/make_golden.py running the
reference's own ``unit_agcn`` / ``Unit2D`` / ``Graph`` on CPU.  The oracle is a
restatement with different op ordering (einsum vs conv), so agreement is to
fp32 rounding, gated at 2e-5 of max|ref| (the product's gate is 1e-4).
"""
import numpy as np
import pytest
import torch

from _util import gather_flat, load_golden, parity_gate, sub_state
from oracle import graph_oracle as go
from oracle import stgcn_oracle as so

TIGHT = 2e-5

GCN_CASES = ["gcn_shre_3_128_quirk", "gcn_shre_3_128_trueA", "gcn_shre_3_128_default_init",
             "gcn_lmdhg_3_128", "gcn_shre_64_64_identity", "gcn_shre_64_128"]
TCN_CASES = ["tcn_128_128_k9", "tcn_64_128_k9_s2", "tcn_64_128_k1_s2", "tcn_128_128_k9_v46",
             "tcn_32_64_k5_nobias"]
STEM_CASES = ["stem_shre_T180", "stem_lmdhg_T200", "stem_shre_T500"]


def test_graphs_match_reference():
    g = load_golden("graphs")
    for key, ref in g.items():
        name, mode = key.split("/")
        mine = go.labeling(name, mode)
        assert mine.shape == ref.shape, key
        np.testing.assert_allclose(mine, ref, rtol=0, atol=1e-15, err_msg=key)


def test_spatial_graph_properties():
    for name, V, nb in (("SHRE", 22, 21), ("LMDHG", 46, 50)):
        A = go.labeling(name, "spatial")
        assert A.shape == (3, V, V)
        assert np.count_nonzero(A[0]) == V and np.count_nonzero(A[1]) == nb and np.count_nonzero(A[2]) == nb
        # column-normalised: non-empty columns sum to one
        for k in (1, 2):
            cs = A[k].sum(0)
            assert np.all((np.abs(cs - 1) < 1e-12) | (cs == 0))


def test_aliasing_quirk_recorded():
    """unit_agcn.__init__ overwrites the caller's A with 1e-6 (model/unit_agcn.py:37-39)."""
    g = load_golden("gcn_shre_3_128_quirk")
    assert np.all(g["A_after_ctor"] == np.float32(1e-6))
    assert g["A_true"].sum() == pytest.approx(64.0, abs=1e-4) or g["A_true"].sum() > 22


@pytest.mark.parametrize("case", GCN_CASES)
@pytest.mark.parametrize("train", [False, True])
def test_agcn_oracle_vs_reference(case, train):
    g = load_golden(case)
    p = so.agcn_params_from_state(sub_state(g, "gcn."), torch.from_numpy(g["A_fixed"]))
    x = torch.from_numpy(g["x"])
    aux = {}
    y = so.agcn_forward(x, p, training=train, aux=aux)
    tag = "train" if train else "eval"
    parity_gate(aux["P"], g[f"P_{tag}"], TIGHT, f"{case} P {tag}")
    parity_gate(y, g[f"y_{tag}"], TIGHT, f"{case} y {tag}")
    if train:
        parity_gate(aux["bn"]["running_mean"], g["after_train.gcn.bn.running_mean"], TIGHT, "bn rm")
        parity_gate(aux["bn"]["running_var"], g["after_train.gcn.bn.running_var"], TIGHT, "bn rv")
        if p.down_w is not None:
            parity_gate(aux["down_bn"]["running_mean"], g["after_train.gcn.down.1.running_mean"], TIGHT, "dbn rm")
            parity_gate(aux["down_bn"]["running_var"], g["after_train.gcn.down.1.running_var"], TIGHT, "dbn rv")


def test_agcn_oracle_T180_samples():
    g = load_golden("gcn_shre_3_128_T180")
    p = so.agcn_params_from_state(sub_state(g, "gcn."), torch.from_numpy(g["A_fixed"]))
    x = torch.from_numpy(g["x"])
    for train in (False, True):
        tag = "train" if train else "eval"
        aux = {}
        y = so.agcn_forward(x, p, training=train, aux=aux)
        parity_gate(aux["P"], g[f"P_{tag}"], TIGHT, "P")
        assert y.abs().max().item() == pytest.approx(float(g[f"y_{tag}_absmax"]), rel=1e-5)
        err = (gather_flat(y, g[f"y_{tag}_idx"]).double() - torch.from_numpy(g[f"y_{tag}_val"]).double()).abs().max()
        assert err <= TIGHT * float(g[f"y_{tag}_absmax"])


@pytest.mark.parametrize("case", TCN_CASES)
@pytest.mark.parametrize("train", [False, True])
def test_tcn_oracle_vs_reference(case, train):
    g = load_golden(case)
    p = so.tcn_params_from_state(sub_state(g, "tcn."), stride=int(g["stride"]))
    aux = {}
    y = so.tcn_forward(torch.from_numpy(g["x"]), p, training=train, aux=aux)
    parity_gate(y, g["y_train" if train else "y_eval"], TIGHT, case)
    if train:
        parity_gate(aux["bn"]["running_mean"], g["after_train.tcn.bn.running_mean"], TIGHT, "rm")
        parity_gate(aux["bn"]["running_var"], g["after_train.tcn.bn.running_var"], TIGHT, "rv")


@pytest.mark.parametrize("case", STEM_CASES)
def test_stem_oracle_vs_reference(case):
    g = load_golden(case)
    gp = so.agcn_params_from_state(sub_state(g, "gcn."), torch.from_numpy(g["A_fixed"]))
    tp = so.tcn_params_from_state(sub_state(g, "tcn."))
    x = so.caller_layout(torch.from_numpy(g["skeleton"]))
    for train in (False, True):
        tag = "train" if train else "eval"
        aux = {}
        z = so.stem_forward(x, gp, tp, training=train, aux=aux)
        parity_gate(aux["gcn"]["P"], g[f"P_{tag}"], TIGHT, "P")
        for nm, arr in (("y", aux["gcn_out"]), ("z", z)):
            scale = float(g[f"{nm}_{tag}_absmax"])
            err = (gather_flat(arr, g[f"{nm}_{tag}_idx"]).double()
                   - torch.from_numpy(g[f"{nm}_{tag}_val"]).double()).abs().max().item()
            assert err <= TIGHT * scale, f"{case} {nm} {tag}: {err:.3e} vs scale {scale:.3e}"
            assert float(arr.double().sum()) == pytest.approx(float(g[f"{nm}_{tag}_sum"]), rel=1e-4, abs=1e-2 * scale)


def test_float64_oracle_close_to_float32_reference():
    """The fp64 run of the oracle is the tighter yardstick used by the GPU tests."""
    g = load_golden("tcn_128_128_k9")
    p = so.tcn_params_from_state(sub_state(g, "tcn.")).to(torch.float64)
    y = so.tcn_forward(torch.from_numpy(g["x"]).double(), p)
    parity_gate(y, g["y_eval"], TIGHT, "fp64 oracle vs fp32 reference")


# ---- backward: autograd through the oracle against gradient fixtures generated from the reference -----------------
def _oracle_leaf_names(gp=None, tp=None):
    """oracle leaf tensors keyed by the reference's parameter names (state_dict keys)."""
    m = {}
    if gp is not None:
        m.update({"gcn.PA": gp.PA, "gcn.down.0.weight": gp.down_w, "gcn.down.0.bias": gp.down_b,
                  "gcn.down.1.weight": gp.down_bn.weight, "gcn.down.1.bias": gp.down_bn.bias,
                  "gcn.bn.weight": gp.bn.weight, "gcn.bn.bias": gp.bn.bias})
        for i in range(gp.num_subset):
            m.update({f"gcn.conv_a.{i}.weight": gp.conv_a_w[i], f"gcn.conv_a.{i}.bias": gp.conv_a_b[i],
                      f"gcn.conv_b.{i}.weight": gp.conv_b_w[i], f"gcn.conv_b.{i}.bias": gp.conv_b_b[i],
                      f"gcn.conv_d.{i}.weight": gp.conv_d_w[i], f"gcn.conv_d.{i}.bias": gp.conv_d_b[i]})
    if tp is not None:
        m.update({"tcn.conv.weight": tp.conv_w, "tcn.conv.bias": tp.conv_b, "tcn.bn.weight": tp.bn.weight,
                  "tcn.bn.bias": tp.bn.bias})
    return m


def _zero_by_structure(name):
    """Gradients that are exactly zero in exact arithmetic: biases in front of a batch-statistics BatchNorm and
    conv_a's bias (constant shift of a soft-maxed column); held against their weight's gradient scale."""
    return name.endswith(".bias") and any(t in name for t in ("conv_a", "conv_d", "down.0", "tcn.conv"))


def _check_grads(got, g, rel):
    bad = []
    for name, val in got.items():
        ref = torch.from_numpy(g["grad." + name]).double().reshape(val.shape)
        scale = ref.abs().max().item()
        if _zero_by_structure(name):
            scale = max(scale, float(np.abs(g["grad." + name.replace(".bias", ".weight")]).max()))
        err = (val.double() - ref).abs().max().item()
        if err > rel * scale:
            bad.append(f"{name}: {err:.3e} > {rel:g}*{scale:.3e}")
    assert not bad, "; ".join(bad)


def test_oracle_autograd_vs_reference_stem_gradients():
    g = load_golden("bwd_stem_shre_T20")
    gp = so.agcn_params_from_state(sub_state(g, "gcn."), torch.from_numpy(g["A_fixed"])).to(torch.float64)
    tp = so.tcn_params_from_state(sub_state(g, "tcn.")).to(torch.float64)
    leaves = _oracle_leaf_names(gp, tp)
    for t in leaves.values():
        t.requires_grad_(True)
    z = so.stem_forward(torch.from_numpy(g["x"]).double(), gp, tp, training=True)
    parity_gate(z.detach(), g["z"], 2e-5, "train-mode stem forward")
    names = sorted(leaves)
    grads = torch.autograd.grad((z * torch.from_numpy(g["G"]).double()).sum(), [leaves[k] for k in names])
    _check_grads(dict(zip(names, grads)), g, 1e-3)     # (inner ReLU kinks: fp32 reference vs fp64 oracle)


def test_oracle_autograd_vs_reference_strided_tcn_gradients():
    g = load_golden("bwd_tcn_64_128_k9_s2")
    tp = so.tcn_params_from_state(sub_state(g, "tcn."), stride=2).to(torch.float64)
    leaves = _oracle_leaf_names(None, tp)
    x = torch.from_numpy(g["x"]).double().requires_grad_(True)
    leaves["x"] = x
    for t in leaves.values():
        t.requires_grad_(True)
    z = so.tcn_forward(x, tp, training=True)
    parity_gate(z.detach(), g["z"], 2e-5, "train-mode strided tcn forward")
    names = sorted(leaves)
    grads = torch.autograd.grad((z * torch.from_numpy(g["G"]).double()).sum(), [leaves[k] for k in names])
    _check_grads(dict(zip(names, grads)), g, 1e-4)


# ---------------------------------------------------------------------------------------
# the library-op CPU baseline (oracle/stgcn_cpu_ops.py, what bench.py times as `cpu_baseline`)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", GCN_CASES)
def test_cpu_ops_agcn_vs_reference(case):
    from oracle import stgcn_cpu_ops as co
    g = load_golden(case)
    p = so.agcn_params_from_state(sub_state(g, "gcn."), torch.from_numpy(g["A_fixed"]))
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        y = co.agcn_forward_ops(x, p)
    parity_gate(y, g["y_eval"], TIGHT, f"{case} library ops vs reference")
    parity_gate(y, so.agcn_forward(x, p), TIGHT, f"{case} library ops vs oracle")


@pytest.mark.parametrize("case,stride", [("tcn_128_128_k9", 1), ("tcn_64_128_k9_s2", 2), ("tcn_64_128_k1_s2", 2),
                                         ("tcn_128_128_k9_v46", 1), ("tcn_32_64_k5_nobias", 1)])
def test_cpu_ops_tcn_vs_reference(case, stride):
    from oracle import stgcn_cpu_ops as co
    g = load_golden(case)
    p = so.tcn_params_from_state(sub_state(g, "tcn."), stride=stride)
    x = torch.from_numpy(g["x"])
    with torch.no_grad():
        y = co.tcn_forward_ops(x, p)
    parity_gate(y, g["y_eval"], TIGHT, f"{case} library ops vs reference")
    parity_gate(y, so.tcn_forward(x, p), TIGHT, f"{case} library ops vs oracle")


def test_cpu_ops_stem_equals_oracle():
    """The thing bench.py times as the CPU baseline computes the same stem as the oracle (seeded, T=40)."""
    from oracle import stgcn_cpu_ops as co
    g = load_golden("stem_shre_T180")
    gp = so.agcn_params_from_state(sub_state(g, "gcn."), torch.from_numpy(g["A_fixed"]))
    tp = so.tcn_params_from_state(sub_state(g, "tcn."))
    x = so.caller_layout(torch.from_numpy(g["skeleton"]))[:, :, :40].contiguous()
    with torch.no_grad():
        parity_gate(co.stem_forward_ops(x, gp, tp), so.stem_forward(x, gp, tp), TIGHT, "stem library ops vs oracle")
