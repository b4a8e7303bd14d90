"""Build libstgcn_hip.so (gfx950) in-tree with hipcc.

    python st-gcn-altformer_amd/stgcn_amd/build.py [--force] [--keep-temps]

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it travels
with the repo snapshot to the GPU box; it is git-ignored.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(HERE, "..", "csrc"))
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "..", "include"))
LIB = os.path.join(HERE, "libstgcn_hip.so")
SOURCES = ["capi.hip", "agcn_attention.hip", "agcn_expand.hip", "tcn_conv.hip", "tcn_bf16.hip", "stem_bf16_v4.hip", "stem_bf16_v6.hip", "stem_bf16_v6w.hip", "stem_f16mx.hip", "tcn_bf16_v6.hip", "train_bn.hip", "tcn_backward.hip", "tcn_wgrad_v6.hip", "agcn_backward.hip", "agcn_train.hip", "gemm_f32.hip", "agcn_backward_generic.hip"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(INCLUDE, "stgcn_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, keep_temps: bool = False, ablation: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library; returns its path.

    ``ablation=True`` builds the diagnostic library libstgcn_hip_abl.so (-DSTGCN_ABLATION: phases of the
    big kernels can be switched off with env STGCN_ABLATE to price them; never loaded by the package
    unless STGCN_LIB points at it).
    """
    if ablation:
        extra = os.environ.get("STGCN_EXTRA_DEFS", "").split()      # e.g. -DV6_NOFILL: one-off diagnostic variants
        return _build(LIB.replace(".so", "_abl.so"), "build_abl", ["-DSTGCN_ABLATION"] + extra, verbose, keep_temps)
    variant = os.environ.get("STGCN_VARIANT")                       # A/B builds: libstgcn_hip_<variant>.so with STGCN_EXTRA_DEFS,
    if variant:                                                     # no diagnostic code (load it with STGCN_LIB)
        return _build(LIB.replace(".so", f"_{variant}.so"), f"build_{variant}", os.environ.get("STGCN_EXTRA_DEFS", "").split(),
                      verbose, keep_temps)
    if not force and not _stale():
        return LIB
    return _build(LIB, "build", [], verbose, keep_temps)


def _build(lib_path, objsub, extra, verbose, keep_temps):
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, objsub)
    os.makedirs(objdir, exist_ok=True)
    common = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE,
              "-Wall", "-Wno-unused-function"] + extra
    if keep_temps:
        common += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = common + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose or keep_temps:
            sys.stdout.write(out)
    link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib_path] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return lib_path


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True, keep_temps="--keep-temps" in sys.argv,
                 ablation="--ablation" in sys.argv)
    print("built", path)
