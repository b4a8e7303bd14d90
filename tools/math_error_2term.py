#!/usr/bin/env python3
"""Would a TWO-MFMA arithmetic keep north_star's 1e-4?  (measure it instead of inferring it.)

CPU emulation, no GPU needed:   python tools/math_error_2term.py > profiles/r03_math_error_2term.txt

The temporal conv of the fused stem multiplies the folded weights W' = scale*W (fp32) with the graph-conv output y (fp32,
post-ReLU) on the 16-bit matrix cores.  A 16-bit x 16-bit product is exact in the fp32 accumulator, so the arithmetic
error of a mode is the error of the operand representation: each operand is a sum of one or more 16-bit terms and the mode
multiplies a subset of the term pairs.  This script applies exactly those roundings (torch's RNE casts to bfloat16 /
float16, gradual underflow; '_ftz' variants flush fp16 subnormals as a pessimistic bound for the lo terms, whose
magnitude is ~2^-12 of the operand) and evaluates the conv in fp64, so that nothing but the operand rounding is measured;
the fp64 oracle is the reference, the three seeded stems / 8 clips / T=180, V=22 are those of tools/math_error.py, and
max|err|/max|ref| is the quantity the parity gate bounds by 1e-4.

  terms  mode                         what is multiplied
  1      bf16 / fp16                  W1*y1
  2      *_wsplit                     (Wh+Wl)*y1          = 2 MFMAs
  2      *_ysplit                     W1*(yh+yl)          = 2 MFMAs
  3      bf16x3 / fp16x3              Wh*yh + Wh*yl + Wl*yh   (the shipped BF16X3, and its fp16 analogue)
  "2"    *_hi+mx<fmt>                 Wh*yh on the 16-bit MFMA + the two residual terms Wl*yh, Wh*yl with BOTH operands
                                      of each in block-scaled 8-bit floats (OCP MX: 32 consecutive channels share a
                                      power-of-two scale, elements saturate) — the v_mfma_scale_f32_*_f8f6f4 path, which
                                      issues fp8 at twice the bf16 rate: 1 + 2 * 1/2 = 2 MFMA units instead of 3.  A residual
                                      term is ~2^-12 of the product, so 3 mantissa bits on its operands leave ~2^-16.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "st-gcn-altformer_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as TF

from oracle import stgcn_oracle as so

torch.set_num_threads(8)
FP16_TINY = 2.0 ** -14      # smallest normal fp16


def split(v32, dt, ftz=False):
    """fp32 tensor -> (hi, lo) as fp64 tensors holding exactly the 16-bit values."""
    hi = v32.to(dt)
    lo = (v32 - hi.float()).to(dt)
    hi, lo = hi.double(), lo.double()
    if ftz:
        hi = torch.where(hi.abs() < FP16_TINY, torch.zeros_like(hi), hi)
        lo = torch.where(lo.abs() < FP16_TINY, torch.zeros_like(lo), lo)
    return hi, lo


def random_stem(V, seed):
    """The CPU half of tests/test_gpu_parity.py::_random_stem (same generator sequence, no GPU modules)."""
    import stgcn_amd
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed)
    A = torch.rand(3, V, V, generator=gen) * (torch.rand(3, V, V, generator=gen) < 0.15)
    gcn = stgcn_amd.unit_agcn(3, 128, A.clone())
    tcn = stgcn_amd.Unit2D(128, 128, kernel_size=9)
    with torch.no_grad():
        gcn.PA.data = torch.randn(3, V, V, generator=gen) * 0.05
        for m in list(gcn.modules()) + list(tcn.modules()):
            if isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
                m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.3)
                m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 1.5 + 0.25)
            if isinstance(m, torch.nn.Conv2d):
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
        for cv in list(gcn.conv_a) + list(gcn.conv_b):
            cv.weight.mul_(3.0)
    gcn.A = A.clone()
    return gcn, tcn, gen


def mx_quant(v, dim, dt, emax, fmax, blk=32):
    """OCP-MX style block quantisation along `dim`: blocks of 32 share the scale 2^(floor(log2 max|v|) - emax), elements
    are rounded to `dt` (saturating at its largest finite value); returned as fp64."""
    v = v.double().movedim(dim, -1)
    K = v.shape[-1]
    pad = (-K) % blk
    if pad:
        v = TF.pad(v, (0, pad))
    b = v.reshape(*v.shape[:-1], -1, blk)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    sc = torch.pow(2.0, torch.floor(torch.log2(amax)) - emax)
    q = (b / sc).clamp(-fmax, fmax).float().to(dt).double() * sc
    return q.reshape(*v.shape)[..., :K].movedim(-1, dim)


MX = [("e4m3", torch.float8_e4m3fn, 8, 448.0), ("e5m2", torch.float8_e5m2, 15, 57344.0)]


def conv64(W, y):                # (Cout,Cin,9) x (N,Cin,T,V) fp64 -> (N,Cout,T,V)
    return TF.conv2d(y, W.unsqueeze(-1), padding=(4, 0))


MODES = []
for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
    MODES += [(name, dt, "1", False), (name + "_wsplit", dt, "w", False), (name + "_ysplit", dt, "y", False),
              (name + "x3", dt, "3", False)]
MODES += [("fp16_wsplit_ftz", torch.float16, "w", True), ("fp16_ysplit_ftz", torch.float16, "y", True),
          ("fp16x3_ftz", torch.float16, "3", True)]


def main():
    worst = {m[0]: 0.0 for m in MODES}
    rms = {m[0]: 0.0 for m in MODES}
    for h in ("fp16", "bf16"):
        for f in MX:
            worst[f"{h}_hi+mx{f[0]}"] = rms[f"{h}_hi+mx{f[0]}"] = 0.0
    rng = {"W'": [float("inf"), 0.0], "y": [float("inf"), 0.0]}
    for seed in (1, 2, 3):
        gcn, tcn, gen = random_stem(22, seed)
        gp = so.agcn_params_from_state(gcn.state_dict(), gcn.A)
        tp = so.tcn_params_from_state(tcn.state_dict())
        x = torch.randn(8, 3, 180, 22, generator=gen)
        ref = so.stem_forward(x.double(), gp.to(torch.float64), tp.to(torch.float64))
        y32 = so.agcn_forward(x.double(), gp.to(torch.float64))
        y32 = (y32[0] if isinstance(y32, tuple) else y32).float()                 # what the producer hands to the conv, as fp32
        bn = tcn.bn
        scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
        shift = (bn.bias - bn.running_mean * scale + tcn.conv.bias * scale).detach().double()
        W32 = (tcn.conv.weight.detach().squeeze(-1) * scale.view(-1, 1, 1)).float()   # folded weights, as stgcn_stem_prepare packs them
        nzw, nzy = W32[W32 != 0].abs(), y32[y32 != 0].abs()
        rng["W'"] = [min(rng["W'"][0], nzw.min().item()), max(rng["W'"][1], nzw.max().item())]
        rng["y"] = [min(rng["y"][0], nzy.min().item()), max(rng["y"][1], nzy.max().item())]
        chk = torch.relu(conv64(W32.double(), y32.double()) + shift.view(1, -1, 1, 1))
        assert ((chk - ref).abs().max() / ref.abs().max()).item() < 5e-6, "the emulation's own fp64 conv disagrees with the oracle"
        for name, dt, kind, ftz in MODES:
            wh, wl = split(W32, dt, ftz)
            yh, yl = split(y32, dt, ftz)
            if kind == "1":
                acc = conv64(wh, yh)
            elif kind == "w":
                acc = conv64(wh + wl, yh)
            elif kind == "y":
                acc = conv64(wh, yh + yl)
            else:
                acc = conv64(wh, yh + yl) + conv64(wl, yh)
            z = torch.relu(acc + shift.view(1, -1, 1, 1))
            worst[name] = max(worst[name], ((z - ref).abs().max() / ref.abs().max()).item())
            rms[name] = max(rms[name], ((z - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
        for hname, hdt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
            wh = W32.to(hdt)
            yh = y32.to(hdt)
            wl, yl = W32 - wh.float(), y32 - yh.float()
            for fname, fdt, emax, fmax in MX:
                q = lambda t, d: mx_quant(t, d, fdt, emax, fmax)       # blocks along the channel axis of either operand
                acc = conv64(wh.double(), yh.double()) + conv64(q(wl, 1), q(yh.float(), 1)) + conv64(q(wh.float(), 1), q(yl, 1))
                z = torch.relu(acc + shift.view(1, -1, 1, 1))
                name = f"{hname}_hi+mx{fname}"
                e = ((z - ref).abs().max() / ref.abs().max()).item()
                assert e == e, "NaN in the block-scaled emulation"
                worst[name] = max(worst.get(name, 0.0), e)
                rms[name] = max(rms.get(name, 0.0), ((z - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item())
    print("CPU emulation of the operand roundings, conv evaluated in fp64 (tools/math_error_2term.py); gate = 1e-4")
    print("fused stem, 8 clips, T=180, V=22, three seeded random stems (tools/math_error.py's); worst over the stems")
    print(f"operand magnitudes (non-zero): |W'| in [{rng[chr(87) + chr(39)][0]:.2e}, {rng[chr(87) + chr(39)][1]:.2e}], |y| in [{rng['y'][0]:.2e}, {rng['y'][1]:.2e}]"
          f"   (fp16: normal >= 6.1e-5, max 65504)")
    print(f"{'mode':18s} {'MFMAs':5s}  max|err|/max|ref|  rms(err)/rms(ref)   verdict")
    rows = [(name, {"1": "1", "w": "2", "y": "2", "3": "3"}[kind]) for name, dt, kind, ftz in MODES]
    rows += [(f"{h}_hi+mx{f[0]}", "1+2/2") for h in ("fp16", "bf16") for f in MX]
    for name, n in rows:
        v = "inside 1e-4" if worst[name] <= 1e-4 else ("inside 1e-2 only" if worst[name] <= 1e-2 else "outside")
        if worst[name] <= 5e-5:
            v += " with 2x margin"
        print(f"{name:18s} {n:>5s}  {worst[name]:.3e}          {rms[name]:.3e}           {v}")
    print("""
Reading.  No TWO-term 16-bit product keeps 1e-4: whichever operand stays a single fp16 carries a 2^-12 relative rounding,
which the K = 1152 contraction averages down to 1.9e-4 (activations split) / 3.5e-4 (weights split) of max|ref| — the
estimate of DESIGN.md section 3 (3.3e-4) was the right size; bf16 single operands are 8-10x worse.  Flushing fp16
subnormals does not matter for the two-term forms (the rounded operand dominates) but ruins fp16x3 (2.3e-4: the residuals of
small operands are subnormal), so an fp16 three-term mode would have to rely on gradual underflow in the MFMA.
What does keep the contract below three bf16-rate MFMAs is the last group: the leading term on the fp16 matrix cores and the two
2^-12-sized residual terms in block-scaled e4m3 (2.6e-5, 4x inside the gate): 2 MFMA units of issue instead of 3.  It is NOT
a drop-in arithmetic switch for KF6 — K = 128 per scaled MFMA, 8-bit operand images and scale bytes change the weight packing,
the producer and the LDS images — so it is recorded here as the candidate for the next rebuild of the main loop, not built.""")


if __name__ == "__main__":
    main()
