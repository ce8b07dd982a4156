"""ctypes binding of libwoft_hip.so (the C ABI declared in include/woft_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, this raises.
`import torch` happens first so that the process-wide HIP runtime (libamdhip64.so.7) is the one
PyTorch-ROCm loaded; the kernels then run on torch's streams and torch-allocated buffers.
"""
import ctypes as C
import os
from pathlib import Path

import torch  # noqa: F401  (must precede the CDLL so both share one HIP runtime)

LIB_PATH = Path(os.environ.get("WOFT_HIP_LIB") or Path(__file__).resolve().parent / "lib" / "libwoft_hip.so")  # (env: A/B of builds)

EPI_LINEAR, EPI_RELU, EPI_SIGMOID, EPI_TANH, EPI_RELU_RES_RELU, EPI_GRU_ZR, EPI_GRU_Q, EPI_CTX, EPI_WH_MEAN, EPI_FLOWHEAD = range(10)

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ConvParams(C.Structure):
    _fields_ = [
        ("in0", vp), ("in1", vp), ("cs0", i32), ("cs1", i32), ("c_split", i32),
        ("n_img", i32), ("h", i32), ("w", i32), ("ho", i32), ("wo", i32),
        ("taps_y", i32), ("taps_x", i32), ("stride", i32), ("pad_y", i32), ("pad_x", i32),
        ("cin_pad", i32), ("flat", i32), ("wgt", vp), ("wgt_hi", vp), ("wgt_lo", vp), ("precision", i32),
        ("bias", vp), ("alpha", f32),
        ("cout", i32), ("cout_pad", i32), ("out", vp), ("ldo", i64), ("co_off", i32),
        ("out_w", i32), ("out_pitch", i32), ("epi", i32), ("split", i32),
        ("e0", vp), ("e1", vp), ("lde0", i32), ("lde1", i32), ("out1", vp), ("ldo1", i32),
        ("stat_sum", vp), ("stat_sq", vp), ("tile_m", i32), ("tile_n", i32), ("halo", i32),
        ("in_norm", i32), ("in_mean", vp), ("in_rstd", vp), ("bias_map", vp), ("ld_bias_map", i32),
        ("out_index", vp),
        ("wh0_lookup", vp), ("wh0_ld", i32), ("wh0_mean", vp), ("wh0_w", vp), ("wh0_bias", vp), ("wh0_index", vp),
        ("wgt_frag", vp), ("wgt_mx", vp),
    ]


class LookupParams(C.Structure):
    _fields_ = [
        ("vol", vp * 4), ("ht", i32 * 4), ("wt", i32 * 4), ("plane", i64 * 4),
        ("levels", i32), ("radius", i32), ("coords", vp), ("n_pix", i64), ("out", vp), ("ldo", i32),
        ("vol_bf16", i32), ("ablate", i32),
    ]


class LookupOtfParams(C.Structure):
    _fields_ = [
        ("f1", vp), ("f2", vp * 4), ("h", i32 * 4), ("w", i32 * 4),
        ("levels", i32), ("radius", i32), ("terms", i32), ("hf", i32), ("wf", i32), ("k", i32),
        ("alpha", f32), ("coords", vp), ("out", vp), ("need", vp), ("ldo", i32), ("ablate", i32),
        ("fh_part", vp), ("fh_bias", vp), ("fh_delta", vp), ("fh_flow4", vp), ("fh_flow_cat", vp),
        ("fh_planes", i32), ("fh_ld", i32), ("fh_ld_delta", i32), ("fh_ld_cat", i32),
    ]


_SIGS = {
    "woft_abi_version": (i32, []),
    "woft_sizeof": (i32, [i32]),
    "woft_set_tuning": (i32, [i32, i32]),
    "woft_conv2d": (i32, [C.POINTER(ConvParams), vp]),
    "woft_conv2d_pair": (i32, [C.POINTER(ConvParams), C.POINTER(ConvParams), vp]),
    "woft_upload_u8": (i32, [vp, vp, vp, i64, i32, vp]),
    "woft_split_bf16": (i32, [vp, i64, vp, vp, vp]),
    "woft_split_bf16_lines": (i32, [vp, i64, vp, vp]),
    "woft_flow_to_tc": (i32, [vp, vp, i32, i32, vp, vp, i32, vp]),
    "woft_corr_lookup_otf": (i32, [C.POINTER(LookupOtfParams), vp]),
    "woft_conv3x3_narrow": (i32, [vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, i64, i32, vp]),
    "woft_flow_head_update": (i32, [vp, i32, i32, i32, i32, vp, vp, vp, i64, vp, vp, vp, i32, vp]),
    "woft_flow_head_gather": (i32, [vp, i32, i32, i32, i32, vp, vp, i64, vp, vp, vp, i32, vp]),
    "woft_corr_gemm_bf16": (i32, [vp, vp, i64, i64, i64, i64, i32, f32, vp, i64, i32, i32, vp]),
    "woft_inorm_finalize": (i32, [vp, vp, i32, i32, i32, i32, i64, f32, vp, vp, vp, vp]),
    "woft_inorm_ws_bytes": (i64, []),
    "woft_inorm_apply": (i32, [vp, vp, vp, vp, vp, vp, i32, vp, i64, i32, i32, vp]),
    "woft_preprocess_bgr_u8": (i32, [vp, i32, i32, vp, i32, i32, i32, i32, vp]),
    "woft_avgpool2_nhwc": (i32, [vp, i32, i32, i32, vp, vp]),
    "woft_feature_pyramid": (i32, [vp, i32, i32, i32, i32, vp, vp, i32, vp]),
    "woft_corr_lookup": (i32, [C.POINTER(LookupParams), vp]),
    "woft_tile_rows": (i32, [vp, i32, i32, i32, vp, vp]),
    "woft_coords_update": (i32, [vp, vp, i32, i32, i64, vp, vp, i32, vp]),
    "woft_coords_init": (i32, [vp, i32, i32, vp, vp, i32, vp]),
    "woft_colsum": (i32, [vp, i64, i32, vp, i32, vp, vp]),
    "woft_wh_pack": (i32, [vp, i32, vp, i32, vp, f32, i64, i32, vp, vp, vp]),
    "woft_wh_conv0": (i32, [vp, i32, vp, i64, i32, vp, vp, vp, vp, vp]),
    "woft_wh_reduce": (i32, [vp, i32, i32, vp, f32, i64, vp, vp]),
    "woft_wh_needed": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp]),
    "woft_convex_upsample": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]),
    "woft_convex_weights_at": (i32, [vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "woft_upflow8": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, i32, vp]),
    "woft_warp_perspective_u8": (i32, [vp, i32, i32, i32, C.POINTER(C.c_double), vp, vp, i32, vp]),
    "woft_resize_linear_u8": (i32, [vp, i32, i32, i32, vp, i32, i32, f32, f32, vp]),
    "woft_tc_select_ws_bytes": (i64, [i64]),
    "woft_tc_select": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, vp, vp, vp, i32, vp, vp]),
    "woft_tc_flags": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]),
    "woft_hfit_ws_bytes": (i64, []),
    "woft_hfit": (i32, [vp, vp, vp, i32, vp, i32, f32, i32, vp, vp, vp, vp]),
    "woft_hfit_step": (i32, [vp, vp, vp, i32, vp, i32, vp, vp, vp, vp, vp]),
    "woft_inlier_frac": (i32, [vp, vp, i32, vp, vp, f32, vp, vp]),
}

EXPORTS = tuple(_SIGS)
_lib = None


class WoftHipError(RuntimeError):
    pass


def load():
    """Load libwoft_hip.so (once).  Raises if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise WoftHipError(f"{LIB_PATH} is missing: build it with `python -m woft_amd.build` "
                           "(hipcc --offload-arch=gfx950); woft_amd has no CPU fallback")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)           # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.woft_sizeof(0) != C.sizeof(ConvParams) or lib.woft_sizeof(1) != C.sizeof(LookupParams) \
            or lib.woft_sizeof(2) != C.sizeof(LookupOtfParams):
        raise WoftHipError("ctypes mirror of woft_conv_params / woft_lookup_params is out of sync with the library")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise WoftHipError(f"{what} failed with code {rc}")


def ptr(t):
    """Device (or host) address of a tensor's first element; None -> NULL."""
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None) if os.environ.get("WOFT_RAW_STREAM", "1") != "0" else None
_CUR_DEV = torch.cuda.current_device


def stream_ptr():
    """HIP stream handle of torch's CURRENT stream on this process's device (every launch is enqueued there).
    torch.cuda.current_stream() builds a Stream object through several Python layers (7.7 us, x 200 launches per frame =
    1.5 ms of host time, most of what stands between a frame's device->host read and the next frame's first launches);
    the raw getter is one C call."""
    if _RAW_STREAM is None:
        return torch.cuda.current_stream().cuda_stream
    return _RAW_STREAM(_CUR_DEV())                       # (current_device(): one cheap C call; follows set_device / device guards)
