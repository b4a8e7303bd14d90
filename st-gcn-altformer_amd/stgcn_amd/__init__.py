"""stgcn_amd — MI355X (gfx950) native ST-GCN stem: HIP kernels behind the reference's nn.Module API.

    from stgcn_amd import unit_agcn, Unit2D, enable_stem_fusion

or, as a drop-in for the reference's own import lines, put ``st-gcn-altformer_amd/`` on ``sys.path``
and keep ``from model.unit_agcn import unit_agcn`` / ``from model.net import Unit2D, import_class``.
"""
from ._capi import (ABI_VERSION, LIB_PATH, MATH_BF16, MATH_BF16X3, MATH_F16MX, MATH_F32, MATH_F32_VALU, OUT_BF16,
                    StgcnError, lib)
from .graphs import HandGraph, LMDHGGraph, SHREGraph
from .modules import (FusedStemOutput, Unit2D, conv_init, disable_stem_fusion, enable_stem_fusion, import_class,
                      set_math_mode, set_output_layout, unit_agcn)

__all__ = ["unit_agcn", "Unit2D", "FusedStemOutput", "conv_init", "import_class", "enable_stem_fusion", "disable_stem_fusion",
           "set_math_mode", "set_output_layout", "SHREGraph", "LMDHGGraph", "HandGraph", "lib", "StgcnError", "LIB_PATH",
           "ABI_VERSION", "MATH_F32", "MATH_BF16X3", "MATH_BF16", "MATH_F32_VALU", "MATH_F16MX", "OUT_BF16"]
