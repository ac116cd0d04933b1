"""ctypes binding of libhat_mi355x.so (the C ABI declared in include/hat_mi355x.h).

There is NO fallback: if the library is missing, cannot be loaded, or an entry point is absent,
every use raises.  Build it with `python -m super_resolution_amd.build` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HAT_MI355X_LIB") or os.path.join(_HERE, "libhat_mi355x.so")  # env override: A/B builds

ABI_VERSION = 2   # == HAT_ABI_VERSION of include/hat_mi355x.h: bump both whenever a descriptor or packing changes
HAT_F32, HAT_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_LRELU = 0, 1, 2
X_NHWC_T, X_NHWC_F32, X_NCHW_F32_MEAN = 0, 1, 2
O_NHWC_T, O_NHWC_F32, O_PIXSHUF_T, O_NCHW_F32 = 0, 1, 2, 3

_ERRORS = {-1: "HAT_EINVAL (bad argument)", -2: "HAT_ELDS (tile does not fit in 160 KiB LDS)",
           -3: "HAT_EUNSUPPORTED (shape not instantiated)"}


class HatConvDesc(C.Structure):
    """Mirror of `struct HatConvDesc` (include/hat_mi355x.h) — field order and types must match."""
    _fields_ = [
        ("x", C.c_void_p), ("x0", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("out", C.c_void_p), ("r1", C.c_void_p), ("r2", C.c_void_p), ("r2scale", C.c_void_p), ("colsum", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
        ("Cin", C.c_int32), ("ldx", C.c_int32), ("x_mode", C.c_int32),
        ("c_split", C.c_int32), ("ldx0", C.c_int32),
        ("ksize", C.c_int32), ("Kpad", C.c_int32),
        ("nt", C.c_int32), ("n_slices", C.c_int32),
        ("w_bstride", C.c_int64),
        ("n_store", C.c_int32),
        ("ldo", C.c_int32), ("out_mode", C.c_int32), ("act", C.c_int32),
        ("ldr1", C.c_int32), ("ldr2", C.c_int32), ("r2scale_bstride", C.c_int32),
        ("ps_r", C.c_int32),
        ("in_scale", C.c_float), ("out_scale", C.c_float),
        ("mean", C.c_float * 4),
        ("dtype", C.c_int32),
        ("ld_ln", C.c_int32), ("ln_ones", C.c_int32),
        ("ln_g", C.c_void_p), ("ln_b", C.c_void_p), ("ln_out", C.c_void_p),
        ("gap_out", C.c_void_p), ("n16_out", C.c_void_p), ("gap_c", C.c_int32), ("reserved0", C.c_int32),
    ]


class HatMlpDesc(C.Structure):
    """Mirror of `struct HatMlpDesc` (include/hat_mi355x.h)."""
    _fields_ = [
        ("x", C.c_void_p), ("w1f", C.c_void_p), ("b1", C.c_void_p), ("w2f", C.c_void_p), ("b2", C.c_void_p),
        ("r1", C.c_void_p), ("out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("hidden", C.c_int32),
        ("ldx", C.c_int32), ("ldr1", C.c_int32), ("ldo", C.c_int32),
        ("out_f32", C.c_int32), ("dtype", C.c_int32),
    ]


class HatFfnDesc(C.Structure):
    """Mirror of `struct HatFfnDesc` (include/hat_mi355x.h)."""
    _fields_ = [
        ("t_in", C.c_void_p), ("t_out", C.c_void_p), ("ln_g", C.c_void_p), ("ln_b", C.c_void_p),
        ("w1f", C.c_void_p), ("b1", C.c_void_p), ("dww", C.c_void_p), ("dwb", C.c_void_p),
        ("w2f", C.c_void_p), ("b2", C.c_void_p), ("ln1_g", C.c_void_p), ("ln1_b", C.c_void_p),
        ("n_out", C.c_void_p), ("gap_out", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
        ("chunks", C.c_int32), ("ldn", C.c_int32), ("gap_c", C.c_int32), ("dtype", C.c_int32),
        ("ldm_in", C.c_int32), ("m_in", C.c_void_p), ("n16_out", C.c_void_p),
    ]


class HatHabTailDesc(C.Structure):
    """Mirror of `struct HatHabTailDesc` (include/hat_mi355x.h)."""
    _fields_ = [("ffn", HatFfnDesc), ("n", C.c_void_p), ("y16", C.c_void_p), ("c1", C.c_void_p), ("w_aggr", C.c_void_p),
                ("wf", C.c_void_p), ("bias_b", C.c_void_p), ("ldn_in", C.c_int32), ("ldr2", C.c_int32), ("r2", C.c_void_p),
                ("r2scale", C.c_void_p), ("r2scale_bstride", C.c_int32), ("reserved1", C.c_int32)]


class HatCabFoldDesc(C.Structure):
    """Mirror of `struct HatCabFoldDesc` (include/hat_mi355x.h)."""
    _fields_ = [
        ("c1", C.c_void_p), ("c1_colsum", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("wk", C.c_void_p),
        ("bias_in", C.c_void_p), ("scale", C.c_void_p), ("wf", C.c_void_p), ("bias_out", C.c_void_p), ("tmp", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("mid", C.c_int32), ("ld1", C.c_int32),
        ("tiles", C.c_int32), ("ldcs", C.c_int32), ("k", C.c_int32), ("ld_scale", C.c_int32), ("dtype", C.c_int32),
        ("conv_scale", C.c_float), ("stats", C.c_void_p), ("w2f", C.c_void_p),
    ]


class HatAggrCabDesc(C.Structure):
    """Mirror of `struct HatAggrCabDesc` (include/hat_mi355x.h)."""
    _fields_ = [("lin", HatConvDesc), ("c1", C.c_void_p), ("wf", C.c_void_p), ("bias_b", C.c_void_p)]


# name -> (restype, argtypes); every symbol declared in include/hat_mi355x.h
SIGNATURES = {
    "hat_abi_version": (C.c_int, []),
    "hat_target_arch": (C.c_char_p, []),
    "hat_conv_tiles": (C.c_int, [C.POINTER(HatConvDesc), C.POINTER(C.c_int32)]),
    "hat_ocab_mlp": (C.c_int, [C.POINTER(HatMlpDesc), C.c_void_p]),
    "hat_ocab_qkv": (C.c_int, [C.POINTER(HatMlpDesc), C.c_void_p]),
    "hat_conv_occupancy": (C.c_int, [C.POINTER(HatConvDesc), C.POINTER(C.c_int32)]),
    "hat_conv_plan": (C.c_int, [C.POINTER(HatConvDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int64)]),
    "hat_conv": (C.c_int, [C.POINTER(HatConvDesc), C.c_void_p]),
    "hat_linear": (C.c_int, [C.POINTER(HatConvDesc), C.c_void_p]),
    "hat_conv3x3_small_groups": (C.c_int, [C.POINTER(HatConvDesc), C.POINTER(C.c_int32)]),
    "hat_conv3x3_small": (C.c_int, [C.POINTER(HatConvDesc), C.c_void_p]),
    "hat_cab_fold": (C.c_int, [C.POINTER(HatCabFoldDesc), C.c_void_p]),
    "hat_aggr_cab": (C.c_int, [C.POINTER(HatAggrCabDesc), C.c_void_p]),
    "hat_ffn_tiles": (C.c_int, [C.POINTER(HatFfnDesc), C.POINTER(C.c_int32)]),
    "hat_ffn": (C.c_int, [C.POINTER(HatFfnDesc), C.c_void_p]),
    "hat_ffn2": (C.c_int, [C.POINTER(HatFfnDesc), C.c_void_p]),
    "hat_hab_tail": (C.c_int, [C.POINTER(HatHabTailDesc), C.c_void_p]),
    "hat_hab_tail3": (C.c_int, [C.POINTER(HatHabTailDesc), C.c_void_p]),
    "hat_plan_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "hat_plan_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hat_plan_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hat_plan_free": (None, [C.c_void_p]),
    "hat_layernorm_blocks": (C.c_int, []),
    "hat_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64,
                                C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hat_rect_sum": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int64, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "hat_add_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "hat_esc_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hat_esc_conv13": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hat_eca_scale": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_int32, C.c_float,
                                C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "hat_dwconv_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hat_ocab_keybias": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p] + [C.c_int32] * 9 + [C.c_void_p]),
    "hat_ocab_attention_kb": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 12 + [C.c_void_p]),
    "hat_sgfn_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hat_ocab_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "hat_ocab_attention_log2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "hat_cab_squeeze_units": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "hat_cab_squeeze": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "hat_conv3x3_to_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "hat_window_attention": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                       C.c_int32, C.c_void_p]),
}

_lib = None
_lock = threading.Lock()


def load():
    """Load the library (once) and bind every entry point; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP kernels are not built. Run `python -m super_resolution_amd.build` "
                "(needs hipcc). There is no CPU fallback for the HAT forward pass.")
        # Load PyTorch's HIP runtime FIRST.  torch links "libamdhip64.so" (its bundled copy, found through
        # its own RPATH) while this library links "libamdhip64.so.7": if ours were loaded first the process
        # would end up with two HIP runtimes and our launches would target the one torch never initialised
        # (hipErrorNoDevice).  With torch's copy resident, our NEEDED entry resolves to it by SONAME.
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name, None)
            if fn is None:
                raise RuntimeError(f"{LIB_PATH} does not export `{name}` (stale build?)")
            fn.restype = res
            fn.argtypes = args
        if lib.hat_abi_version() != ABI_VERSION:
            raise RuntimeError(f"ABI version mismatch: library {lib.hat_abi_version()}, binding {ABI_VERSION} (stale or overridden "
                               f"{LIB_PATH}? rebuild with `python -m super_resolution_amd.build`)")
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = _ERRORS.get(rc, f"hipError_t {rc}")
        raise RuntimeError(f"{what} failed: {msg}")
