"""INTEGRATION.md §1 executed verbatim against the reference's own caller files (container only).

The reference never travels to the GPU box, so these tests skip wherever ``/root/reference`` is absent.  They run in
subprocesses: the recipe is about ``sys.path`` order and the ``model`` namespace package, which an already-imported
``model`` in the pytest process would mask.  Round-2 review: the recipe used to fail with ``No module named
'model.AltFormer'`` because the shim's ``model/`` was a regular package shadowing the reference's namespace package.

  * process A (reference only): builds the REFERENCE's ST_GCN_AltFormer (constructor call of
    SHREC/ST_TS/train_sttran.py:75-82) and a TCN_GCN_unit (model/ST_TR/ST_TR_new.py:285-372) and saves their state_dicts;
  * process B (shim first on sys.path, the python block of INTEGRATION.md §1 as it stands): the reference's caller files
    import, their stems are ``stgcn_amd.modules`` classes, and the reference's state_dicts load with ``strict=True``.
No forward runs (no GPU here; the HIP modules refuse CPU tensors).
"""
import os
import re
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SHIM = os.path.join(ROOT, "st-gcn-altformer_amd")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "model", "AltFormer", "ST_GCN_AltFormer.py")),
                                reason="the reference tree is only present in the build container")

TIMM_STUB = textwrap.dedent('''
    import sys, types, torch
    sys.dont_write_bytecode = True
    class DropPath(torch.nn.Module):                 # timm is not installed; only these three names are imported
        def __init__(self, drop_prob=0.0):           # (model_ST.py:10, model_TS.py:12; SURVEY.md 8c caveat 2)
            super().__init__()
        def forward(self, x):
            return x
    _l = types.ModuleType("timm.models.layers")
    _l.DropPath, _l.trunc_normal_ = DropPath, torch.nn.init.trunc_normal_
    _l.to_2tuple = lambda v: v if isinstance(v, tuple) else (v, v)
    _t, _m = types.ModuleType("timm"), types.ModuleType("timm.models")
    _t.models, _m.layers = _m, _l
    sys.modules.update({"timm": _t, "timm.models": _m, "timm.models.layers": _l})
''')

BUILD = textwrap.dedent('''
    import torch
    from model.AltFormer.ST_GCN_AltFormer import ST_GCN_AltFormer
    from model.ST_TR.ST_TR_new import TCN_GCN_unit
    import numpy as np
    torch.manual_seed(7)
    net = ST_GCN_AltFormer(channel=3, backbone_in_c=128, num_frame=180, num_joints=22, num_class=14, style='ST',
                           graph='graph.SHRE', graph_args=dict([('labeling_mode', 'spatial')]))   # train_sttran.py:75-82
    A = torch.from_numpy(net.graph.A.astype(np.float32))
    kw = dict(attention=False, only_attention=False, tcn_attention=False, only_temporal_attention=False, relative=False,
              device=0, attention_3=False, dv=0.25, dk=0.25, Nh=8, num=4, dim_block1=10, dim_block2=30, dim_block3=75,
              num_point=22, weight_matrix=2, more_channels=False, drop_connect=True, starting_ch=64, all_layers=False,
              adjacency=False, data_normalization=True, visualization=False, skip_conn=True)
    units = [TCN_GCN_unit(64, 64, A.clone(), **kw), TCN_GCN_unit(64, 128, A.clone(), stride=2, **kw)]
''')


def _run(code, tmp_path, name):
    f = tmp_path / name
    f.write_text(code)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, str(f)], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, f"{name} failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    return r.stdout


def _integration_block():
    """The first python block of INTEGRATION.md §1, with its two placeholder paths filled in."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text.split("## 1.", 1)[1]
    block = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "/path/to/this/repo/st-gcn-altformer_amd" in block and "/path/to/ST-GCN-AltFormer" in block, \
        "INTEGRATION.md §1 no longer carries the two placeholder paths this test fills in"
    return block.replace("/path/to/this/repo/st-gcn-altformer_amd", SHIM).replace("/path/to/ST-GCN-AltFormer", REF)


def test_integration_recipe_imports_the_reference_callers(tmp_path):
    ref_sd = tmp_path / "ref_sd.pt"
    # A: the reference alone
    _run(TIMM_STUB + f"sys.path.insert(0, {REF!r})\n" + BUILD + textwrap.dedent(f'''
        assert type(net.gcn0).__module__ == "model.unit_agcn" and type(net.tcn0).__module__ == "model.net"
        torch.save({{"net": net.state_dict(), "u0": units[0].state_dict(), "u1": units[1].state_dict()}}, {str(ref_sd)!r})
    '''), tmp_path, "a_reference.py")
    # B: INTEGRATION.md §1 as written, then the reference's own caller files
    out = _run(TIMM_STUB + _integration_block() + BUILD + textwrap.dedent(f'''
        import model, graph, stgcn_amd
        assert list(model.__path__)[0].startswith({SHIM!r}) and any(p.startswith({REF!r}) for p in model.__path__), model.__path__
        assert ST_GCN_AltFormer.__module__ == "model.AltFormer.ST_GCN_AltFormer"
        assert sys.modules[ST_GCN_AltFormer.__module__].__file__.startswith({REF!r})          # the reference's own file
        assert sys.modules["model.net"].__file__.startswith({SHIM!r})
        assert sys.modules["model.unit_agcn"].__file__.startswith({SHIM!r})
        assert graph.__file__.startswith({SHIM!r})
        for m in (net.gcn0, net.tcn0, units[0].gcn1, units[0].tcn1, units[1].gcn1, units[1].tcn1, units[1].down1):
            assert type(m).__module__ == "stgcn_amd.modules", type(m)
        assert type(net.graph).__module__ == "stgcn_amd.graphs"
        sd = torch.load({str(ref_sd)!r}, weights_only=True)
        for mod, key in ((net, "net"), (units[0], "u0"), (units[1], "u1")):
            assert list(mod.state_dict().keys()) == list(sd[key].keys()), key            # same names, same order
            assert all(mod.state_dict()[k].shape == v.shape for k, v in sd[key].items()), key
            mod.load_state_dict(sd[key], strict=True)
            assert all(torch.equal(mod.state_dict()[k], v) for k, v in sd[key].items())
        # the quirk travels with the drop-in: the caller's A reads 1e-6 after construction (model/unit_agcn.py:37-39)
        assert float(net.A.max()) == float(net.A.min()) == float(torch.tensor(1e-6))
        # the HIP modules refuse CPU tensors instead of computing something else
        try:
            net.gcn0(torch.zeros(1, 3, 8, 22))
        except RuntimeError as e:
            assert "GPU" in str(e)
        else:
            raise AssertionError("CPU input accepted")
        print("RECIPE-OK", len(sd["net"]))
    '''), tmp_path, "b_dropin.py")
    assert "RECIPE-OK" in out


def test_shim_exports_every_name_the_reference_imports():
    """Every ``from model.net import …`` / ``from model.unit_agcn import …`` line of the reference resolves in the shim
    (ST_TR_new.py:8 imports ``conv_init`` from model.unit_agcn — the kaiming fan_out one, not model/net.py's)."""
    wanted = {"model.net": set(), "model.unit_agcn": set()}
    for dirpath, _, files in os.walk(REF):
        for fn in files:
            if not fn.endswith(".py"):
                continue
            for line in open(os.path.join(dirpath, fn), errors="replace"):
                m = re.match(r"\s*from (model\.net|model\.unit_agcn) import (.+)", line)
                if m:
                    wanted[m.group(1)] |= {n.strip() for n in m.group(2).split("#")[0].split(",") if n.strip()}
    assert wanted["model.net"] and wanted["model.unit_agcn"]
    import importlib
    for modname, names in wanted.items():
        mod = importlib.import_module(modname)
        assert mod.__file__.startswith(SHIM)
        for n in names:
            assert hasattr(mod, n), f"{modname}.{n} missing from the drop-in"
    import torch
    from model.unit_agcn import conv_init as agcn_ci
    c = torch.nn.Conv1d(4, 6, 1)
    agcn_ci(c)
    assert float(c.bias.detach().abs().max()) == 0.0          # model/unit_agcn.py:12-14 zeroes the bias; model/net.py:60-65 does not
