"""Shared helpers for the tests: golden loading and the parity gate."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def sub_state(g, prefix):
    """state_dict (torch tensors) of the keys that start with ``prefix``."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith(prefix)}


# arithmetic modes of the temporal-conv contraction and their gates:
#   f32 / f32_valu / bf16x3 : the fp32 contract of north_star — 1e-4 relative (both criteria below)
#   bf16                    : operands rounded to bf16; documented looser bound 1e-2*max|ref| (max-norm only)
#   f16mx                   : fused stem only (STGCN_STEM_F16MX, opt-in): fp16 x fp16 + two scaled-e4m3 residual products —
#                             inside north_star's 1e-4*max|ref| (measured 2e-5, tools/math_error_2term.py) but NOT inside the
#                             mixed allclose criterion (its error on near-zero outputs is ~2e-5*max|ref| > atol): max-norm only
MATH_GATES = {"f32": (1e-4, True), "f32_valu": (1e-4, True), "bf16x3": (1e-4, True), "bf16": (1e-2, False), "f16mx": (1e-4, False)}


def parity_gate(out, ref, rel=1e-4, what="", strict=True):
    """SURVEY §8(d) gate for fp32: max|out-ref| <= rel*max|ref| and allclose(rtol=rel, atol=rel/10*max|ref|)."""
    out = torch.as_tensor(out).double().cpu()
    ref = torch.as_tensor(ref).double().cpu()
    assert out.shape == ref.shape, f"{what}: shape {tuple(out.shape)} vs {tuple(ref.shape)}"
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    scale = ref.abs().max().item()
    err = (out - ref).abs().max().item()
    assert err <= rel * max(scale, 1e-30), f"{what}: max abs err {err:.3e} > {rel:g} * max|ref| ({scale:.3e})"
    if strict:
        assert torch.allclose(out, ref, rtol=rel, atol=rel * 0.1 * scale), f"{what}: allclose(rtol={rel}) failed"
    return err / max(scale, 1e-30)


def gather_flat(t, idx):
    return torch.as_tensor(t).reshape(-1)[torch.as_tensor(idx)]
