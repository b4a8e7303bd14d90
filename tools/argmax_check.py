#!/usr/bin/env python3
"""north_star's "bit-exact for class-index argmax" clause, container half (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tools/argmax_check.py [gpurun_out/argmax] > profiles/r03_argmax_check.txt

The GPU test ``test_whole_model_stem_output_and_argmax_handoff`` wrote the HIP stem's output for the fixture's 8 clips
(z_gpu_<math>.npy).  This script re-creates the reference ``ST_GCN_AltFormer`` from the fixture's seeds (imported from
/root/reference with the shims of tests/golden/make_golden_model.py), checks that it reproduces the fixture's logits
from the reference's own stem, then feeds the GPU-produced z through the reference's transformer heads on CPU and
asserts the class indices are identical for both heads; it reports the largest logit deviation next to the smallest
top-1/top-2 margin.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.environ.get("STGCN_REFERENCE", "/root/reference"))


def main():
    import make_golden as mg
    import make_golden_model as mm
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "argmax")
    with np.load(os.path.join(ROOT, "tests", "golden", "model_altformer_shre.npz")) as f:
        fix = {k: f[k] for k in f.files}
    ok = True
    for style in ("ST", "TS"):
        model, skel = mm.build_reference_model(style)
        assert np.array_equal(skel.numpy(), fix["skeleton"]), "fixture and regenerated clips differ"
        with mg.cuda_is_identity(), torch.no_grad():
            ref_logits = model(skel).numpy()
        dev0 = np.abs(ref_logits - fix[f"logits_{style}"]).max()
        print(f"[{style}] reference model re-created from seed {mm.MODEL_SEED}: max |logits - fixture| = {dev0:.3e}")
        assert dev0 <= 1e-5 and np.array_equal(ref_logits.argmax(1), fix[f"argmax_{style}"])
        margin = fix[f"margin_{style}"]
        for math in ("f32", "bf16x3", "f16mx"):
            path = os.path.join(src, f"z_gpu_{math}.npy")
            if not os.path.exists(path):
                print(f"[{style}] {math}: {path} missing (run the -m gpu suite first)")
                ok = False
                continue
            z = torch.from_numpy(np.load(path))
            logits = mm.heads_logits(model, z).numpy()
            same = np.array_equal(logits.argmax(1), fix[f"argmax_{style}"])
            dev = np.abs(logits - ref_logits).max()
            print(f"[{style}] {math:7s}: argmax GPU-stem {logits.argmax(1).tolist()} vs reference "
                  f"{fix[f'argmax_' + style].tolist()} -> {'IDENTICAL' if same else 'DIFFERENT'}; "
                  f"max |logit deviation| = {dev:.3e}, smallest top-1/top-2 margin = {margin.min():.3e} "
                  f"(ratio {dev / margin.min():.2e})")
            ok &= bool(same)
    print("ARGMAX PARITY:", "PASS" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
