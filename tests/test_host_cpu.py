"""CPU-only tests of the host side: C-ABI surface, graph builders, module mirror (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from _util import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "stgcn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(stgcn_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from stgcn_amd import build as _b  # noqa: F401  (module import check)
    from stgcn_amd.build import build
    path = build()
    assert os.path.exists(path)
    handle = ctypes.CDLL(path)
    names = declared_symbols()
    assert "stgcn_stem_forward_prepared" in names and "stgcn_agcn_forward" in names
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/stgcn_hip.h but not exported"


def test_binding_covers_header_and_abi_version():
    from stgcn_amd import _capi
    assert sorted(_capi.PROTOTYPES) == declared_symbols()
    handle = _capi.lib()
    assert handle.stgcn_version() == _capi.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "stgcn_hip.h")).read()
    assert f"#define STGCN_ABI_VERSION {_capi.ABI_VERSION}" in hdr
    # size / support queries are pure host functions: callable without a GPU
    assert handle.stgcn_tcn_packed_bytes(128, 128, 9, 0) >= 128 * 128 * 9 * 4
    assert handle.stgcn_tcn_supported(128, 128, 180, 22, 9, 1, 0) == 1
    assert handle.stgcn_tcn_supported(30, 128, 180, 22, 9, 1, 0) == 0
    assert handle.stgcn_stem_supported(3, 128, 180, 22, 9, 3, 0) == 1
    assert handle.stgcn_stem_supported(3, 128, 200, 46, 9, 3, 0) == 1
    assert handle.stgcn_stem_supported(64, 128, 180, 22, 9, 3, 0) == 0


def test_null_arguments_are_rejected_with_message():
    from stgcn_amd import _capi
    handle = _capi.lib()
    rc = handle.stgcn_tcn_pack(None, None, None, 128, 128, 9, 0, None)
    assert rc == -1
    assert b"NULL" in handle.stgcn_last_error()
    with pytest.raises(_capi.StgcnError) as e:
        _capi.call("stgcn_bn_fold", None, None, None, None, None, 1e-5, None, None, 4, None)
    assert e.value.code == -1


def test_product_graphs_match_reference_fixture():
    from stgcn_amd.graphs import LMDHGGraph, SHREGraph
    g = load_golden("graphs")
    for key, ref in g.items():
        name, mode = key.split("/")
        G = SHREGraph if name == "SHRE" else LMDHGGraph
        np.testing.assert_allclose(G(mode).A, ref, rtol=0, atol=1e-15, err_msg=key)
    with pytest.raises(ValueError):
        SHREGraph("nonsense")


def test_graph_dropin_names(capsys):
    from model.net import import_class
    G = import_class("graph.SHRE")
    assert "name graph.SHRE" in capsys.readouterr().out          # the reference prints it (net.py:69)
    g = G(labeling_mode="spatial")
    assert g.A.shape == (3, 22, 22) and g.A.dtype == np.float64 and g.num_node == 22
    assert len(g.inward) == 21 and g.outward[0] == (g.inward[0][1], g.inward[0][0])
    L = import_class("graph.LMDHG")(labeling_mode="spatial")
    assert L.A.shape == (3, 46, 46) and len(L.inward) == 50
    import graph.tools as tools
    assert np.array_equal(tools.edge2mat([(0, 1)], 3), np.array([[0, 0, 0], [1, 0, 0], [0, 0, 0.]]))


def test_module_mirror_keys_shapes_and_init():
    from model.net import Unit2D
    from model.unit_agcn import unit_agcn
    from stgcn_amd.graphs import SHREGraph
    ref = load_golden("init_stats")
    torch.manual_seed(0)
    A = torch.from_numpy(SHREGraph("spatial").A.astype(np.float32))
    gcn = unit_agcn(3, 128, A)
    tcn = Unit2D(128, 128, kernel_size=9)
    ours = {"gcn." + k: v for k, v in gcn.state_dict().items()}
    ours.update({"tcn." + k: v for k, v in tcn.state_dict().items()})
    ref_keys = sorted(k[len("shape:"):] for k in ref if k.startswith("shape:"))
    assert sorted(ours) == ref_keys                               # strict-load compatibility
    for k in ref_keys:
        assert tuple(ours[k].shape) == tuple(ref["shape:" + k]), k
        if ours[k].numel() >= 384 and ours[k].dtype.is_floating_point:   # random-normal inits: same std
            assert ours[k].float().std().item() == pytest.approx(float(ref["std:" + k]), rel=0.15), k
        elif ours[k].dtype.is_floating_point:
            assert ours[k].float().mean().item() == pytest.approx(float(ref["mean:" + k]), abs=1e-7), k
    # the aliasing quirk of unit_agcn.py:37-39 is reproduced: caller's A now reads 1e-6
    assert torch.all(A == 1e-6) and torch.all(gcn.A == 1e-6) and gcn.A.data_ptr() == gcn.PA.data_ptr()
    assert "A" not in gcn.state_dict()
    assert gcn.inter_c == 32 and gcn.num_subset == 3
    assert gcn.bn.weight.detach().unique().item() == pytest.approx(1e-6)
    # identity residual when Cin == Cout
    g2 = unit_agcn(64, 64, torch.zeros(3, 22, 22))
    assert not isinstance(g2.down, torch.nn.Module) and g2.down(5) == 5
    assert not any(k.startswith("down") for k in g2.state_dict())


def test_module_mirror_loads_reference_state_dicts_strictly():
    from stgcn_amd import Unit2D, unit_agcn
    from _util import sub_state
    g = load_golden("stem_shre_T180")
    gcn = unit_agcn(3, 128, torch.zeros(3, 22, 22))
    tcn = Unit2D(128, 128, kernel_size=9)
    gcn.load_state_dict(sub_state(g, "gcn."), strict=True)
    tcn.load_state_dict(sub_state(g, "tcn."), strict=True)
    # DataParallel-style prefix also works through the usual wrapper
    wrapped = torch.nn.Sequential()
    wrapped.add_module("module", gcn)
    wrapped.load_state_dict({"module." + k: v for k, v in sub_state(g, "gcn.").items()}, strict=True)


def test_unit2d_signature_and_errors():
    from stgcn_amd import Unit2D
    m = Unit2D(64, 128, kernel_size=9, stride=2, dropout=0.5, bias=False)
    assert m.conv.kernel_size == (9, 1) and m.conv.padding == (4, 0) and m.conv.stride == (2, 1)
    assert m.conv.bias is None and m.dropout.p == 0.5
    m3 = Unit2D(8, 8, kernel_size=4, dim=3)
    assert m3.conv.kernel_size == (1, 4) and m3.conv.padding == (0, 1)
    with pytest.raises(ValueError):
        Unit2D(8, 8, kernel_size=3, dim=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.eval()(torch.zeros(1, 64, 8, 22))


def test_no_product_import_of_the_oracle():
    """The shipped package must never import anything from oracle/."""
    pkg = os.path.join(ROOT, "st-gcn-altformer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("# oracle", ""), f"{f} mentions the oracle"


def test_fused_stem_output_type_refuses_everything_but_its_unit2d():
    """Host logic of the fusion guard (no GPU needed): the deferred result can be inspected and handed to its paired
    Unit2D, nothing else — any torch operation raises with an explanation."""
    from stgcn_amd import FusedStemOutput, Unit2D
    tcn, other = Unit2D(4, 4, 3), Unit2D(4, 4, 3)
    payload = torch.randn(2, 4, 5, 6)
    w = FusedStemOutput.wrap(payload, tcn)
    assert isinstance(w, torch.Tensor) and tuple(w.shape) == (2, 4, 5, 6) and w.dtype == torch.float32
    assert w.dim() == 4 and w.size(1) == 4 and w.data_ptr() == payload.data_ptr() and "FusedStemOutput" in repr(w)
    for misuse in (lambda: w + 1, lambda: w.clone(), lambda: w.half(), lambda: torch.relu(w), lambda: w.detach(),
                   lambda: w[0], lambda: w.numpy(), lambda: torch.cat([w, w]), lambda: other(w)):
        with pytest.raises(RuntimeError):
            misuse()
    with pytest.raises(RuntimeError, match="disable_stem_fusion"):
        w.mean()
    out = tcn(w)                                           # the paired module unwraps; no kernel runs for it
    assert type(out) is torch.Tensor and out.data_ptr() == payload.data_ptr()
    assert torch.equal(out + 0, payload)


def test_input_checks_order_and_messages():
    from stgcn_amd import Unit2D, unit_agcn
    m = Unit2D(8, 8, kernel_size=3).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):     # device first: a CPU tensor is never a grad question
        m(torch.zeros(1, 8, 4, 4))
    g = unit_agcn(8, 8, torch.rand(3, 4, 4))
    g.bn.eval()
    assert g._bn_training() is False and g.training             # statistics mode follows the BatchNorm sub-module
    g2 = unit_agcn(3, 8, torch.rand(3, 4, 4))
    g2.down[1].eval()
    with pytest.raises(NotImplementedError, match="different modes"):
        g2._bn_training()
