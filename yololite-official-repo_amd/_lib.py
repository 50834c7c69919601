"""ctypes binding of the C ABI in include/yololite_hip.h (libyololite_hip.so, built for gfx950 by
csrc/build.py).  There is NO CPU fallback: if the library is missing or fails to load, importing
this module raises -- the product path must fail loudly rather than run somewhere else."""
from __future__ import annotations

import ctypes as C
import os
import sys
import warnings


def _default_hw_queues():
    """The executor overlaps two batch chunks on two internal HIP streams.  ROCm maps streams onto
    GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin at creation; once RCCL (or a host with many
    torch streams) has created its own streams, a chunk stream can share a hardware queue with the caller's
    stream, which serialises the chunks and their fork/join barriers (measured: -25 % images/s as soon as
    init_process_group("nccl") has run).  8 queues avoid the aliasing.  The variable is read when the HIP
    runtime initialises (first device call), so it is set HERE -- on import of the package, whatever the host
    program is (bench.py, the CLIs, a serving process) -- unless the user chose a value; if the runtime is
    already up the default cannot take effect any more and that is said loudly."""
    if "GPU_MAX_HW_QUEUES" in os.environ or os.environ.get("YOLOLITE_NO_ENV_DEFAULTS"):
        return          # the user's value, or an explicit opt-out: the host application owns its environment
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    t = sys.modules.get("torch")
    try:
        late = t is not None and t.cuda.is_initialized()
    except Exception:   # pragma: no cover
        late = False
    if late:
        warnings.warn("yololite_amd: the HIP runtime was initialised before this package was imported, so "
                      "GPU_MAX_HW_QUEUES=8 cannot be applied; export it in the environment (multi-stream overlap "
                      "of the executor degrades by ~25 % next to RCCL streams otherwise)", RuntimeWarning)


_default_hw_queues()

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLOLITE_HIP_LIB selects another build of the same ABI (kernel A/B runs); default: the in-tree library
LIB_PATH = os.environ.get("YOLOLITE_HIP_LIB") or os.path.join(_HERE, "libyololite_hip.so")

YL_ABI_VERSION = 5
YL_MAX_LEVELS = 8
YL_OK = 0
ACT = {"none": 0, "relu": 1, "relu6": 2, "silu": 3, "gelu": 4, "relu_lab": 5}
OP_STEM, OP_CONV, OP_DW, OP_STEMBLOCK, OP_SE = 0, 1, 2, 3, 4
OP_POOL, OP_COPY, OP_LN, OP_GRN, OP_NHWC4 = 5, 6, 7, 8, 9
POST_MAIN, POST_FALLBACK, POST_EVAL = 0, 1, 2
CENTER = {"v8": 0, "simple": 1}
WH = {"softplus": 0, "v8": 1, "exp": 2}
NMS_TORCHVISION, NMS_GREEDY = 0, 1

# "dev_select" bits (developer A/B, bitwise kernel-equivalence tests; see yl_get_option in the header)
DEV_DW_TILE_OFF, DEV_PWS_OFF, DEV_S2C_OFF, DEV_DWC_ALL, DEV_DWT_OFF = 1, 2, 4, 8, 16
DEV_DWT_NOSPLIT = 1 << 10
DEV_WINO_V1 = 1 << 11            # Winograd: yl_conv_wino_kernel (every position in one wave) instead of yl_conv_wino2_kernel
DEV_DWL_OFF = 1 << 14             # depthwise 3x3 -> wide 1x1: yl_conv_dwk_kernel instead of yl_conv_dwl_kernel (window in LDS)
DEV_DWL_ALL = 1 << 15             # ... yl_conv_dwl_kernel on every grid (partial windows, few items: the bitwise test)
DEV_DPW_OFF = 1 << 16             # fused head launch: yl_conv_dpp_kernel (taps from L1/L2) instead of yl_conv_dpw_kernel (window in LDS)
DEV_K3W_OFF = 1 << 17             # small-channel dense 3x3: Winograd / direct kernels instead of yl_conv_k3w_kernel
DEV_WINO_SHAPE_SHIFT = 12         # yl_conv_wino2_kernel item shape (2 bits): 0 auto, 1 (4,4), 2 (2,7), 3 two m-tiles

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class yl_layer(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "op", "in_slot", "out_slot", "res_slot", "up_slot", "head_level", "cin", "cout",
        "k", "stride", "pad_t", "pad_l", "act", "in_shift", "dw_k", "dw_stride", "dw_pad_t", "dw_pad_l", "dw_act")] + [
        ("w", _fp), ("b", _fp), ("dw_w", _fp), ("dw_b", _fp),
        ("c2", C.c_int32), ("act2", C.c_int32), ("c3", C.c_int32), ("act3", C.c_int32),
        ("w2", _fp), ("b2", _fp), ("w3", _fp), ("b3", _fp), ("scale_slot", C.c_int32), ("reserved0", C.c_int32),
        ("lab_scale", C.c_float), ("lab_bias", C.c_float), ("eps", C.c_float), ("out_ch_off", C.c_int32)]


class yl_model_desc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("img_size", C.c_int32), ("in_channels", C.c_int32),
                ("num_classes", C.c_int32), ("num_masks", C.c_int32), ("proto_slot", C.c_int32),
                ("num_levels", C.c_int32),
                ("level_size", C.c_int32 * YL_MAX_LEVELS), ("level_anchors", C.c_int32 * YL_MAX_LEVELS),
                ("num_slots", C.c_int32), ("slot_h", _ip), ("slot_w", _ip), ("slot_c", _ip),
                ("num_layers", C.c_int32), ("layers", C.POINTER(yl_layer))]


class yl_post_cfg(C.Structure):
    _fields_ = [("mode", C.c_int32), ("conf_thr", C.c_float), ("iou_thr", C.c_float),
                ("per_class_cap", C.c_int32), ("topk", C.c_int32), ("max_out", C.c_int32),
                ("center_mode", C.c_int32), ("wh_mode", C.c_int32), ("backmap_dev", C.c_void_p),
                ("fallback_nms", C.c_int32)]


# every symbol include/yololite_hip.h declares: (name, restype, argtypes)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
SYMBOLS = [
    ("yl_create", C.c_int32, [C.POINTER(yl_model_desc), C.c_int32, C.POINTER(_vp)]),
    ("yl_clone", C.c_int32, [_vp, C.POINTER(_vp)]),
    ("yl_streams_overlap", C.c_int32, [C.c_int32, _vp, _vp, C.POINTER(C.c_int32)]),
    ("yl_destroy", None, [_vp]),
    ("yl_strerror", C.c_char_p, [C.c_int32]),
    ("yl_last_error", C.c_char_p, [_vp]),
    ("yl_abi_version", C.c_int32, []),
    ("yl_forward", C.c_int32, [_vp, _vp, C.c_int32, _vpp, _vp]),
    ("yl_forward_timed", C.c_int32, [_vp, _vp, C.c_int32, _vpp, _vp, _fp]),
    ("yl_last_timing", C.c_int32, [_vp, _fp, _fp]),
    ("yl_activation_bytes", C.c_int64, [_vp]),
    ("yl_read_slot", C.c_int32, [_vp, C.c_int32, C.c_int32, _vp, _vp]),
    ("yl_set_option", C.c_int32, [_vp, C.c_char_p, C.c_int32]),
    ("yl_get_option", C.c_int32, [_vp, C.c_char_p, _ip]),
    ("yl_query_fused_block", C.c_int32, [C.c_int32] * 7),
    ("yl_query_dw_prologue", C.c_int32, [C.c_int32] * 6),
    ("yl_forward_decoded", C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    ("yl_preprocess", C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, _vp]),
    ("yl_decode", C.c_int32, [_vp, _vpp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    ("yl_postprocess", C.c_int32, [_vp, _vpp, C.c_int32, C.POINTER(yl_post_cfg), _vp, _vp, _vp, _vp]),
    ("yl_predict", C.c_int32, [_vp, _vp, C.c_int32, C.POINTER(yl_post_cfg), _vp, _vp, _vp, _vp]),
    ("yl_allgather_dets", C.c_int32, [_vp, _vp, _vp, C.c_int64, _vp, _vp]),
    ("yl_masks", C.c_int32, [_vp, _vpp, C.c_int32, _vp, _vp, C.c_int32, C.c_float, _vp, _vp]),
    ("yl_masks_image", C.c_int32, [_vp, _vpp, C.c_int32, _vp, _vp, _vp, C.c_int32, C.c_float, _vp, _vp, _vp, C.c_int32,
                                   C.c_int32, C.c_int32, _vp, _vp]),
    ("yl_nms", C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_float, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    ("yl_eval_match", C.c_int32, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_double, _vp, _vp, _vp, _vp]),
    ("yl_eval_sweep", C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp]),
    ("yl_eval_confusion", C.c_int32, [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                      _vp, _vp, _vp]),
    ("yl_track_create", C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32,
                                    C.POINTER(_vp)]),
    ("yl_track_destroy", None, [_vp]),
    ("yl_track_reset", C.c_int32, [_vp, C.c_int32, _vp]),
    ("yl_track_update", C.c_int32, [_vp, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("yl_track_grow", C.c_int32, [_vp, C.c_int32]),
    ("yl_track_stats", C.c_int32, [_vp, _ip, _ip]),
]

_lib = None


class YoloLiteHipError(RuntimeError):
    pass


def load():
    """Load libyololite_hip.so once.  torch is imported first so that the HIP runtime the library
    binds to (libamdhip64.so.7) is the one PyTorch-ROCm already mapped into the process -- device
    pointers of torch tensors are then valid for our kernels."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YoloLiteHipError(
            f"{LIB_PATH} not found: build it with `python yololite-official-repo_amd/csrc/build.py` "
            "(or __graft_entry__.build()). There is no CPU fallback.")
    import torch  # noqa: F401  (maps PyTorch's libamdhip64 first)
    try:
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover
        raise YoloLiteHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    if lib.yl_abi_version() != YL_ABI_VERSION:
        raise YoloLiteHipError("libyololite_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(status: int, ctx=None, what: str = ""):
    if status == YL_OK:
        return
    lib = load()
    msg = lib.yl_strerror(status).decode()
    detail = lib.yl_last_error(ctx).decode() if ctx else ""
    raise YoloLiteHipError(f"{what or 'yololite_hip'}: {msg} ({status}) {detail}".strip())
