"""ctypes binding of libshiftnet_hip.so (the C ABI in include/shiftnet_hip.h).

There is deliberately NO fallback: if the shared object is missing or a symbol
is absent, importing the product path fails loudly (the HIP kernels ARE the
product; the CPU oracle under ``oracle/`` is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libshiftnet_hip.so")

SN_F32, SN_F16, SN_BF16 = 0, 1, 2

ABI_VERSION = 17     # == SN_ABI_VERSION of include/shiftnet_hip.h; a stale .so from before a struct / signature change fails the check in load()

SYMBOLS = [          # include/shiftnet_hip.h, production ABI
    "sn_abi_version", "sn_selftest_mfma", "sn_ingest", "sn_conv2d", "sn_conv_pool_blocks", "sn_upsample2_add", "sn_ca_mlp",
    "sn_cab_ca", "sn_cab_ca_scratch_floats", "sn_cab_fused_supported", "sn_cab_stats", "sn_cab_ca_lines", "sn_cab_fused", "sn_planar_pitch", "sn_nhwc_to_planar", "sn_dw5m_blocks",
    "sn_dw5m_gemm_gate", "sn_gsts_gather", "sn_temporal_roll", "sn_gsts_shiftconv", "sn_gsts_shiftconv_mfma", "sn_gsts_cab2_phase2", "sn_cab1_phase2",
    "sn_ingest_u8", "sn_egress_blocks", "sn_egress_u8", "sn_ssim_blocks", "sn_ssim_u8",
    "sn32_conv2d", "sn32_gsts_gather", "sn32_layernorm", "sn32_gate", "sn32_gate_sum", "sn32_chan_sum", "sn32_scale_residual", "sn32_ingest", "sn32_cab_ca", "sn32_dw_gate", "sn32_conv1x1_gate2", "sn32_gsts_shiftconv", "sn32_conv_csum_tiles",
    "sn_ln_gemm_gate", "sn_lngate_blocks", "sn_grp5_gemm_gate", "sn_grp5_blocks", "sn_gsts_cab2_phase1", "sn_cab1_phase1", "sn_phase1_pool_blocks", "sn_phase1_g1_store_bytes", "sn_p1r_plan", "sn_p1r_strip_begin",
]


class Phase1Weights(C.Structure):
    """sn_phase1_weights (device pointers): prep.pack_phase1r (csrc/sn_phase1r.hip)."""
    _fields_ = [("wfrag1", C.c_void_p), ("w3", C.c_void_p), ("wgrp", C.c_void_p), ("wfrag2", C.c_void_p)]


class SeFold(C.Structure):
    """sn_se_fold: CALayer2's MLP finished by the last workgroup of each frame of the phase-1 launch."""
    _fields_ = [("wa", C.c_void_p), ("wb", C.c_void_p), ("c", C.c_int), ("cr", C.c_int), ("ticket", C.c_void_p), ("ca", C.c_void_p), ("bad", C.c_void_p)]


class Phase1Opts(C.Structure):
    """sn_phase1_opts: the denoisers' inner CALayer2 (pass 1: g1_sums = 1; pass 2: g1_scale = its scale [T][C] f32); team: 0 = the library's
    choice of how many workgroups walk consecutive frames in lock step (measurement knob)."""
    _fields_ = [("g1_scale", C.c_void_p), ("g1_sums", C.c_int), ("team", C.c_int), ("g1_store", C.c_void_p)]


def cab_phase1(lib, src: "UnitSrc", hw_ptr, wt: "Phase1Weights", g2_ptr, pool_ptr, stream, se: "SeFold" = None, opt: "Phase1Opts" = None) -> int:
    """sn_gsts_cab2_phase1 (src.mode 1 / 2) or sn_cab1_phase1 (mode 0)."""
    sep = C.byref(se) if se is not None else None
    op = C.byref(opt) if opt is not None else None
    if src.mode:
        return lib.sn_gsts_cab2_phase1(C.byref(src), hw_ptr, C.byref(wt), g2_ptr, pool_ptr, sep, op, stream)
    return lib.sn_cab1_phase1(C.byref(src), C.byref(wt), g2_ptr, pool_ptr, sep, op, stream)


class ConvDesc(C.Structure):
    _fields_ = [
        ("inp", C.c_void_p * 3), ("n_in", C.c_int), ("cs_in", C.c_int),
        ("T", C.c_int), ("h_in", C.c_int), ("w_in", C.c_int), ("in_mode", C.c_int),
        ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("h_out", C.c_int), ("w_out", C.c_int),
        ("wfrag", C.c_void_p), ("mt", C.c_int), ("ks", C.c_int), ("bias", C.c_void_p),
        ("act", C.c_int), ("prelu", C.c_float), ("res", C.c_void_p), ("out", C.c_void_p),
        ("cs_out", C.c_int), ("out_mode", C.c_int), ("c_out", C.c_int), ("nchw_dtype", C.c_int),
        ("sc", C.c_void_p), ("sc_dtype", C.c_int), ("pool", C.c_void_p), ("oscale", C.c_void_p), ("oscale_stride", C.c_int), ("res2", C.c_void_p),
        ("flags", C.c_int),
    ]


SN_CONV_TILE_KERNEL = 1


class Conv32Desc(C.Structure):
    _fields_ = [
        ("inp", C.c_void_p * 3), ("c_in", C.c_int * 3), ("cs_in", C.c_int * 3), ("n_in", C.c_int),
        ("T", C.c_int), ("h_in", C.c_int), ("w_in", C.c_int), ("in_mode", C.c_int),
        ("k", C.c_int), ("stride", C.c_int), ("pad", C.c_int), ("groups", C.c_int),
        ("h_out", C.c_int), ("w_out", C.c_int), ("c_out", C.c_int),
        ("w", C.c_void_p), ("bias", C.c_void_p), ("act", C.c_int), ("prelu", C.c_float),
        ("oscale", C.c_void_p), ("oscale_stride", C.c_int), ("res", C.c_void_p), ("cs_res", C.c_int),
        ("out", C.c_void_p), ("cs_out", C.c_int), ("out_mode", C.c_int), ("nchw_dtype", C.c_int), ("sc", C.c_void_p),
        ("wsplit", C.c_void_p), ("iscale", C.c_void_p), ("iscale_stride", C.c_int), ("rscale", C.c_void_p), ("rscale_stride", C.c_int),
        ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("csum", C.c_void_p), ("csum_cpad", C.c_int),
    ]


class UnitSrc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("T", C.c_int), ("h", C.c_int), ("w", C.c_int), ("C", C.c_int),
                ("mode", C.c_int), ("wrap", C.c_int), ("halo", C.c_void_p), ("t0", C.c_int), ("nt", C.c_int)]


class ShiftNetLibError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the library once; raise (never degrade) if it or any declared symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm ships its own libamdhip64; it must be mapped BEFORE our library so that both bind to the same HIP
    # runtime (a second runtime in the process owns different streams/contexts and every launch fails).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ShiftNetLibError(
            f"{LIB_PATH} not found: build it with `python shift-net_amd/build.py` (hipcc, gfx950). "
            "There is no CPU fallback for the Shift-Net HIP path.")
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise ShiftNetLibError(f"{LIB_PATH} lacks symbols {missing}")
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    lib.sn_abi_version.restype = ci
    lib.sn_selftest_mfma.argtypes = [vp, vp, vp, vp]
    lib.sn_ingest.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci, vp]
    lib.sn_conv2d.argtypes = [C.POINTER(ConvDesc), vp]
    lib.sn_conv_pool_blocks.argtypes = [C.POINTER(ConvDesc)]
    lib.sn_upsample2_add.argtypes = [vp, vp, vp, ci, ci, ci, ci, vp]
    lib.sn_ca_mlp.argtypes = [vp, ci, ci, ci, ci, cf, vp, vp, vp, ci, vp, vp]
    lib.sn_planar_pitch.argtypes = [ci]
    lib.sn_nhwc_to_planar.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    lib.sn_dw5m_blocks.argtypes = [ci, ci]
    lib.sn_dw5m_gemm_gate.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.sn_cab_ca.argtypes = [vp, ci, ci, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp]
    lib.sn_cab_ca_scratch_floats.argtypes = [ci]
    lib.sn_cab_fused_supported.argtypes = [C.POINTER(ConvDesc), C.POINTER(ConvDesc)]
    lib.sn_cab_stats.argtypes = [C.POINTER(ConvDesc), ci, vp]
    lib.sn_cab_ca_lines.argtypes = [vp, ci, ci, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp]
    lib.sn_cab_fused.argtypes = [C.POINTER(ConvDesc), C.POINTER(ConvDesc), ci, vp]
    lib.sn32_cab_ca.argtypes = [vp, ci, ci, vp, ci, ci, ci, ci, vp, vp, vp, vp, vp, ci, vp]
    lib.sn_gsts_gather.argtypes = [C.POINTER(UnitSrc), vp, vp, vp]
    lib.sn_temporal_roll.argtypes = [C.POINTER(UnitSrc), vp, vp]
    lib.sn_gsts_shiftconv.argtypes = [C.POINTER(UnitSrc), vp, vp, vp, vp]
    lib.sn_gsts_shiftconv_mfma.argtypes = [C.POINTER(UnitSrc), vp, vp, vp, vp]
    lib.sn_ln_gemm_gate.argtypes = [C.POINTER(UnitSrc), vp, vp, vp, vp, vp, vp, ci, vp]
    lib.sn_lngate_blocks.argtypes = [ci, ci]
    lib.sn_grp5_gemm_gate.argtypes = [vp, vp, vp, vp, vp, vp, ci, ci, ci, ci, vp]
    lib.sn_grp5_blocks.argtypes = [ci, ci]
    lib.sn_phase1_pool_blocks.argtypes = [ci, ci, ci]
    lib.sn_phase1_g1_store_bytes.argtypes = [ci, ci, ci, ci, C.POINTER(C.c_longlong)]
    lib.sn_p1r_plan.argtypes = [ci, ci, ci, ci, ci, C.POINTER(ci * 7)]
    lib.sn_p1r_strip_begin.argtypes = [C.POINTER(ci * 7), ci, ci]
    lib.sn_gsts_cab2_phase1.argtypes = [C.POINTER(UnitSrc), vp, C.POINTER(Phase1Weights), vp, vp, C.POINTER(SeFold), C.POINTER(Phase1Opts), vp]
    lib.sn_cab1_phase1.argtypes = [C.POINTER(UnitSrc), C.POINTER(Phase1Weights), vp, vp, C.POINTER(SeFold), C.POINTER(Phase1Opts), vp]
    lib.sn_gsts_cab2_phase2.argtypes = [C.POINTER(UnitSrc), vp, vp, vp, vp, vp, vp]
    lib.sn_cab1_phase2.argtypes = [C.POINTER(UnitSrc), vp, vp, vp, vp, vp, vp]
    lib.sn_ingest_u8.argtypes = [vp, vp, ci, ci, ci, ci, vp]
    lib.sn_egress_blocks.argtypes = []
    lib.sn_egress_u8.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, vp]
    lib.sn_ssim_blocks.argtypes = []
    lib.sn_ssim_u8.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, vp]
    ll = C.c_longlong
    lib.sn32_conv2d.argtypes = [C.POINTER(Conv32Desc), vp]
    lib.sn32_gsts_gather.argtypes = [C.POINTER(UnitSrc), vp, vp, vp, vp]
    lib.sn32_layernorm.argtypes = [vp, ci, ci, vp, vp, vp, ci, ll, vp]
    lib.sn32_gate.argtypes = [vp, ci, ci, vp, ll, vp]
    lib.sn32_gate_sum.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci, vp, vp]
    lib.sn32_dw_gate.argtypes = [vp, ci, vp, ci, ci, vp, ci, ci, ci, ci, vp, vp]
    lib.sn32_conv_csum_tiles.argtypes = [ci, ci]
    lib.sn32_gsts_shiftconv.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.sn32_conv1x1_gate2.argtypes = [vp, ci, ci, vp, ci, ci, vp, ci, ci, vp, vp]
    lib.sn32_chan_sum.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp]
    lib.sn32_scale_residual.argtypes = [vp, vp, vp, ci, vp, ci, ci, ci, vp]
    lib.sn32_ingest.argtypes = [vp, ci, vp, vp, ci, ci, ci, ci, vp]
    for s in SYMBOLS:
        getattr(lib, s).restype = ci
    if lib.sn_abi_version() != ABI_VERSION:
        raise ShiftNetLibError(f"ABI version mismatch: {LIB_PATH} reports {lib.sn_abi_version()}, shiftnet_amd/lib.py expects {ABI_VERSION} "
                               "(stale build: run `python shift-net_amd/build.py`)")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise ShiftNetLibError(f"{what} failed with code {rc}")
