"""Host-side "compiler": reference checkpoint (state_dict + meta)  ->  fused layer program for the
HIP executor (include/yololite_hip.h: yl_layer / yl_model_desc).

What it mirrors in the reference:
  * build_model_from_meta            /root/reference/tools/infer.py:34-77   (which keys of `meta` are read)
  * YOLOLiteMS / YOLOLiteMS_CPU      /root/reference/scripts/model/model_v2.py:77-247, 250-383
      (FPN channel / depth arithmetic :106-107,:277-278; level order and anchors :138-152,:303-314;
       forward order P5 -> P4 -> P3, heads P3,P4,P5(,P6) :352-377)
  * the timm backbones behind timm.create_model(..., features_only=True) (model_v2.py:94-100,266-272),
    given here as DATA (arch strings) -- see BACKBONES; recollection of timm's definitions, the
    tables can be corrected without touching any kernel.

What it adds (MI355X-side design, no reference counterpart): BatchNorm folding, fusion of
depthwise -> pointwise pairs into one MFMA kernel with a depthwise prologue, residual /
nearest-upsample-add / activation epilogues, one GEMM for the three head output convs.
"""
from __future__ import annotations

import functools
import math
import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

def _make_divisible(v, divisor=8, round_limit=0.9):
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


# --------------------------------------------------------------------------------------------------
# backbone tables (timm arch-string notation):  type_r<repeat>_a<dw_start k>_k<kernel>_s<stride>_e<expand>_c<out>
BACKBONES: Dict[str, dict] = {
    "mobilenetv4_conv_small": dict(
        arch=[["cn_r1_k3_s2_e1_c32", "cn_r1_k1_s1_e1_c32"],
              ["cn_r1_k3_s2_e1_c96", "cn_r1_k1_s1_e1_c64"],
              ["uir_r1_a5_k5_s2_e3_c96", "uir_r4_a0_k3_s1_e2_c96", "uir_r1_a3_k0_s1_e4_c96"],
              ["uir_r1_a3_k3_s2_e6_c128", "uir_r1_a5_k5_s1_e4_c128", "uir_r1_a0_k5_s1_e4_c128",
               "uir_r1_a0_k5_s1_e3_c128", "uir_r2_a0_k3_s1_e4_c128"],
              ["cn_r1_k1_s1_e1_c960"]],
        cmult=1.0, dmult=1.0, act="relu", eps=1e-5, same=False, fix_first_last=False, stem=32),
    "tf_efficientnet_lite0": dict(
        arch=[["ds_r1_k3_s1_e1_c16"], ["ir_r2_k3_s2_e6_c24"], ["ir_r2_k5_s2_e6_c40"], ["ir_r3_k3_s2_e6_c80"],
              ["ir_r3_k5_s1_e6_c112"], ["ir_r4_k5_s2_e6_c192"], ["ir_r1_k3_s1_e6_c320"]],
        cmult=1.0, dmult=1.0, act="relu6", eps=1e-3, same=True, fix_first_last=True, stem=32),
    # tiny test vehicle (not a timm model): every block flavour at a few thousand parameters
    "oracle_tiny": dict(
        arch=[["cn_r1_k3_s2_e1_c8"], ["cn_r1_k3_s2_e1_c12", "cn_r1_k1_s1_e1_c8"],
              ["uir_r1_a5_k5_s2_e3_c16", "uir_r1_a0_k3_s1_e2_c16", "uir_r1_a3_k0_s1_e4_c16"],
              ["uir_r1_a3_k3_s2_e4_c24", "uir_r1_a0_k5_s1_e2_c24"], ["cn_r1_k1_s1_e1_c32"]],
        cmult=1.0, dmult=1.0, act="relu", eps=1e-5, same=False, fix_first_last=False, stem=16),
}
# timm _gen_efficientnetv2_base (tf_ variants: TF-SAME padding, BN eps 1e-3; SiLU; channel rounding with round_limit 0):
# ConvBnAct stage with residual, fused-MBConv `er` (EdgeResidual), MBConv `ir` with SqueezeExcite (se0.25 of the block
# input).  configs/v2_models/yololite_{n,s,m}.yaml; published checksums BENCHMARK.md:356-357.
BACKBONES["tf_efficientnetv2_b0"] = dict(
    arch=[["cn_r1_k3_s1_e1_c16_skip"], ["er_r2_k3_s2_e4_c32"], ["er_r2_k3_s2_e4_c48"], ["ir_r3_k3_s2_e4_c96_se0.25"],
          ["ir_r5_k3_s1_e6_c112_se0.25"], ["ir_r8_k3_s2_e6_c192_se0.25"]],
    cmult=1.0, dmult=1.0, act="silu", eps=1e-3, same=True, fix_first_last=False, stem=32, round_limit=0.0)
for _n, _c, _d in (("1", 1.0, 1.1), ("2", 1.1, 1.2), ("3", 1.2, 1.4)):
    # the stem is rounded like every other width (timm: no fix_stem for this family): 32 for b0-b2, 40 for b3
    BACKBONES["tf_efficientnetv2_b" + _n] = dict(BACKBONES["tf_efficientnetv2_b0"], cmult=_c, dmult=_d,
                                                 stem=_make_divisible(32 * _c, 8, 0.0))
# tiny efficientnetv2-style test vehicle (not a timm model; oracle/backbones.py: ORACLE_TINY_V2)
BACKBONES["oracle_tiny_v2"] = dict(
    arch=[["cn_r2_k3_s1_e1_c8_skip"], ["er_r2_k3_s2_e2_c12"], ["er_r1_k3_s2_e4_c16"], ["ir_r2_k3_s2_e4_c24_se0.25"],
          ["ir_r2_k3_s1_e3_c24_se0.25"], ["ir_r2_k3_s2_e4_c32_se0.25"]],
    cmult=1.0, dmult=1.0, act="silu", eps=1e-3, same=True, fix_first_last=False, stem=16, round_limit=0.0)
BACKBONES["mobilenetv4_conv_small_050"] = dict(BACKBONES["mobilenetv4_conv_small"], cmult=0.5)
BACKBONES["oracle_tiny_tf"] = dict(BACKBONES["oracle_tiny"], act="relu6", eps=1e-3, same=True)
for _n, _c, _d in (("1", 1.0, 1.1), ("2", 1.1, 1.2), ("3", 1.2, 1.4), ("4", 1.4, 1.8)):
    BACKBONES["tf_efficientnet_lite" + _n] = dict(BACKBONES["tf_efficientnet_lite0"], cmult=_c, dmult=_d)


def _parse(s: str) -> dict:
    parts = s.split("_")
    d = {"type": parts[0], "r": 1, "e": 1.0, "a": 0, "skip": False, "se": 0.0}
    for p in parts[1:]:
        if p == "skip":
            d["skip"] = True
        elif p.startswith("se"):
            d["se"] = float(p[2:])
        else:
            d[p[0]] = float(p[1:]) if p[0] == "e" else int(p[1:])
    return d


# --------------------------------------------------------------------------------------------------
@dataclass
class Layer:
    op: int
    in_slot: int
    out_slot: int
    cin: int
    cout: int
    k: int
    stride: int
    pad_t: int
    pad_l: int
    act: int
    w: np.ndarray
    b: Optional[np.ndarray]
    in_shift: int = 0           # conv reads its input nearest-upsampled by 2**in_shift
    res_slot: int = -1
    up_slot: int = -1
    head_level: int = -1
    scale_slot: int = -1        # squeeze-excite gate [1,1,cin] multiplying this 1x1 conv's input (an OP_SE output)
    dw_k: int = 0
    dw_stride: int = 1
    dw_pad_t: int = 0
    dw_pad_l: int = 0
    dw_act: int = 0
    dw_w: Optional[np.ndarray] = None
    dw_b: Optional[np.ndarray] = None
    c2: int = 0                 # OP_STEMBLOCK: second (3x3 s2) and optional third (1x1) conv
    act2: int = 0
    c3: int = 0
    act3: int = 0
    w2: Optional[np.ndarray] = None
    b2: Optional[np.ndarray] = None
    w3: Optional[np.ndarray] = None
    b3: Optional[np.ndarray] = None
    lab_scale: float = 1.0      # act == relu_lab: the two scalars of timm's LearnableAffineBlock (hgnetv2)
    lab_bias: float = 0.0
    eps: float = 0.0            # OP_LN / OP_GRN
    out_ch_off: int = 0         # OP_COPY: first channel of out_slot written
    name: str = ""
    macs: int = 0               # per image
    bytes_in: int = 0           # algorithmic HBM bytes per image (read)
    bytes_out: int = 0          # (written)


@dataclass
class Program:
    img_size: int
    num_classes: int
    level_size: List[int]
    level_anchors: List[int]
    strides: List[int]
    slots: List[Tuple[int, int, int]] = field(default_factory=list)       # (h, w, c)
    layers: List[Layer] = field(default_factory=list)
    used_keys: set = field(default_factory=set)
    known_keys: set = field(default_factory=set)
    feature_slots: Dict[str, int] = field(default_factory=dict)           # c3/c4/c5/p3/... for tests
    num_masks: int = 0          # build-defined instance-mask branch (0 = detector)
    proto_slot: int = -1

    @property
    def macs(self):
        return sum(l.macs for l in self.layers)


class SynthStateDict(dict):
    """Seeded synthetic weights, fabricated key by key while the program is built (there are no
    checkpoints or downloads in this environment).  Distributions follow SURVEY 8(d): kaiming-normal
    (fan_in-scaled, see make()) convs, BatchNorm gamma~U(.5,1.5) beta~N(0,.1) mean~N(0,.1) var~U(.5,1.5), detection bias
    init (obj=-ln 99, cls=-ln C, box=0; model_v2.py:7-14) plus N(0,head_noise) so scores straddle the
    thresholds.  Keys and shapes equal a reference checkpoint's state_dict."""

    def __init__(self, seed: int = 0, num_classes: int = 80, head_noise: float = 0.5):
        super().__init__()
        self.seed = int(seed)
        self.C, self.head_noise = num_classes, head_noise

    def make(self, key: str, shape):
        if key in self:
            return
        # one generator per key: the values do not depend on the order in which the builder asks for them
        r = np.random.RandomState((zlib.crc32(key.encode()) ^ (self.seed * 2654435761)) & 0x7FFFFFFF)
        if ".out." in key:
            if key.endswith(".weight"):
                v = r.randn(*shape) * (math.sqrt(1.0 / shape[1]) * (1.0 + self.head_noise))
            else:
                base = {"obj": -math.log(99.0), "cls": (-math.log(self.C) if self.C > 1 else 0.0), "box": 0.0, "mc": 0.0}
                v = base[key.split(".")[-2]] + r.randn(*shape) * self.head_noise
        elif key.endswith("grn.weight"):
            v = r.randn(*shape) * 0.2                   # timm initialises GRN at zero; trained values are small
        elif key.endswith("lab.scale"):
            v = 1.0 + r.randn(*shape) * 0.1
        elif len(shape) == 2:                           # nn.Linear weight [out, in] (convnext mlp)
            v = r.randn(*shape) * math.sqrt(1.0 / shape[1])
        elif key.endswith("se.conv_expand.bias"):
            v = 1.5 + r.randn(*shape) * 0.5             # gates around 0.8: activations keep their scale through 20 SE blocks
        elif key.endswith("running_var"):
            v = r.rand(*shape) + 0.5
        elif key.endswith("running_mean") or key.endswith(".bias"):
            v = r.randn(*shape) * 0.1
        elif len(shape) == 1:                       # BatchNorm gamma
            v = r.rand(*shape) + 0.5
        else:                                       # conv weight [cout, cin/groups, k, k]
            # fan_in scaling (not SURVEY's fan_out): keeps activations O(1) through ~60 layers so the
            # head logits are not saturated and fp32 tolerances stay meaningful
            v = r.randn(*shape) * math.sqrt(1.0 / (shape[1] * shape[2] * shape[3]))
        self[key] = np.ascontiguousarray(v, np.float32)


def synth_state_dict(meta: dict, seed: int = 0, head_noise: float = 0.5) -> Dict[str, np.ndarray]:
    """state_dict (numpy) for the model `meta` describes, with seeded synthetic weights."""
    sd = SynthStateDict(seed, int(meta.get("num_classes") or 80), head_noise)
    build_program(meta, sd)
    return dict(sd)


# target (mean, std) of the raw head outputs after calibrate_head(): objectness fires on ~3 % of the candidates at
# conf 0.4, every class is equally likely to carry a candidate's maximum (80 iid rows: max ~ mean + 2.4 std -> a
# class confidence of ~0.9), boxes are ~3 strides wide so that neighbouring survivors overlap around the NMS
# threshold
HEAD_TARGETS = {"obj": (-3.8, 2.0), "cls": (-4.0, 2.5), "tx": (0.0, 1.0), "ty": (0.0, 1.0), "tw": (3.0, 0.7),
                "th": (3.0, 0.7), "mc": (0.0, 1.0)}


def calibrate_head(state_dict: Dict[str, np.ndarray], meta: dict, levels: Sequence[np.ndarray],
                   targets: Optional[dict] = None) -> Dict[str, np.ndarray]:
    """Synthetic weights have no training behind them: with the init biases of model_v2.py:7-14 a random model
    detects nothing, and noise on ONE objectness row / a few class rows makes it fire everywhere or nowhere, in a
    handful of classes (the head inputs are post-ReLU with seed-dependent channel means).  This gives the head
    the statistics of a model that detects -- for EVERY seed -- by an exact re-parametrisation of the output
    convs: `levels` are the raw head outputs [B,A,S,S,5+C(+NM)] of any forward pass with `state_dict` (HIP or
    oracle) on representative inputs; every output row e of every level gets  w' = w*s/sd_e,
    b' = (b - mean_e)*s/sd_e + t  so that its logits have mean t and std s (HEAD_TARGETS) on those inputs.
    Returns a new state_dict (same keys); both the HIP model and the oracle then load the SAME weights."""
    tg = dict(HEAD_TARGETS, **(targets or {}))
    cfg = meta.get("config", {}) or {}
    tcfg, mcfg = cfg.get("training", {}) or {}, cfg.get("model", {}) or {}
    names = (["2"] if tcfg.get("use_p2") else []) + ["3", "4", "5"] + (["6"] if tcfg.get("use_p6") else [])
    C = int(meta.get("num_classes") or mcfg.get("num_classes") or 80)
    NM = int(mcfg.get("num_masks", 32)) if mcfg.get("seg") else 0
    out = {k: np.array(v, copy=True) for k, v in state_dict.items()}
    assert len(levels) == len(names)
    for k, lv in zip(names, levels):
        lv = np.asarray(lv, np.float64)
        A, E = lv.shape[1], lv.shape[-1]
        assert E == 5 + C + NM
        mu = lv.transpose(1, 4, 0, 2, 3).reshape(A, E, -1).mean(-1)
        sd_ = lv.transpose(1, 4, 0, 2, 3).reshape(A, E, -1).std(-1) + 1e-12
        for a in range(A):
            rows = [("box", 4 * a + j, j, tg[n]) for j, n in enumerate(("tx", "ty", "tw", "th"))]
            rows += [("obj", a, 4, tg["obj"])]
            rows += [("cls", C * a + c, 5 + c, tg["cls"]) for c in range(C)]
            rows += [("mc", NM * a + q, 5 + C + q, tg["mc"]) for q in range(NM)]
            for conv, r, e, (t, s) in rows:
                w, b = out[f"head{k}.out.{conv}.weight"], out[f"head{k}.out.{conv}.bias"]
                g = s / sd_[a, e]
                b[r] = np.float32((np.float64(b[r]) - mu[a, e]) * g + t)
                w[r] = (w[r].astype(np.float64) * g).astype(np.float32)
    return out


def make_meta(arch: str, backbone: str, num_classes: int = 80, img_size: int = 640, fpn_channels: int = 128,
              depth_multiple: float = 1.0, width_multiple: float = 1.0, head_depth: int = 1, use_p6: bool = False,
              use_p2: bool = False, anchors: int = 1, names=None, seg: bool = False, num_masks: int = 32,
              proto_channels: int = 64) -> dict:
    """A `meta` dict shaped like the one the reference stores in checkpoints (tools/train.py:62-75)."""
    nl = 3 + int(use_p6) + int(use_p2)
    return dict(metric_key="map50", metric_value=-1.0, names=names, num_classes=num_classes, img_size=img_size,
                arch=arch, backbone=backbone, num_anchors_per_level=(anchors,) * nl,
                config=dict(model=dict(arch=arch, backbone=backbone, num_classes=num_classes, fpn_channels=fpn_channels,
                                       depth_multiple=depth_multiple, width_multiple=width_multiple,
                                       head_depth=head_depth, **(dict(seg=True, num_masks=num_masks,
                                                                      proto_channels=proto_channels) if seg else {})),
                            training=dict(img_size=img_size, use_p6=use_p6, use_p2=use_p2)))


# /root/reference/configs/models/*.yaml
MODEL_ZOO = {
    "edge_n": dict(arch="YOLOLiteMS_CPU", backbone="mobilenetv4_conv_small_050", depth_multiple=0.65,
                   width_multiple=0.60, fpn_channels=160, head_depth=1),
    "edge_s": dict(arch="YOLOLiteMS_CPU", backbone="mobilenetv4_conv_small", depth_multiple=0.90,
                   width_multiple=0.75, fpn_channels=256, head_depth=2),
    "edge_m": dict(arch="YOLOLiteMS_CPU", backbone="mobilenetv4_conv_small", depth_multiple=0.95,
                   width_multiple=0.85, fpn_channels=288, head_depth=2),
    "edge_l": dict(arch="YOLOLiteMS_CPU", backbone="mobilenetv4_conv_small", depth_multiple=1.05,
                   width_multiple=1.00, fpn_channels=320, head_depth=3),
    "yololite_n": dict(arch="YOLOLiteMS", backbone="tf_efficientnet_lite0", depth_multiple=1.0,
                       width_multiple=1.0, fpn_channels=196, head_depth=1),
    "yololite_s": dict(arch="YOLOLiteMS", backbone="tf_efficientnet_lite1", depth_multiple=1.0,
                       width_multiple=1.0, fpn_channels=256, head_depth=1),
    "yololite_m": dict(arch="YOLOLiteMS", backbone="tf_efficientnet_lite2", depth_multiple=1.0,
                       width_multiple=1.0, fpn_channels=328, head_depth=2),
    "yololite_l": dict(arch="YOLOLiteMS", backbone="tf_efficientnet_lite3", depth_multiple=1.0,
                       width_multiple=1.0, fpn_channels=512, head_depth=3),
    "yololite_xl": dict(arch="YOLOLiteMS", backbone="tf_efficientnet_lite4", depth_multiple=1.5,
                        width_multiple=1.0, fpn_channels=512, head_depth=3),
    # /root/reference/configs/v2_models/*.yaml (the yololite_n / yololite_m whose parameters and MACs BENCHMARK.md:356-357
    # publishes: 8.923 M / 11.473 G and 17.916 M / 27.239 G)
    "yololite_n_v2": dict(arch="YOLOLiteMS", backbone="tf_efficientnetv2_b0", depth_multiple=1.0,
                          width_multiple=1.0, fpn_channels=196, head_depth=1),
    "yololite_s_v2": dict(arch="YOLOLiteMS", backbone="tf_efficientnetv2_b1", depth_multiple=1.0,
                          width_multiple=1.0, fpn_channels=256, head_depth=2),
    "yololite_m_v2": dict(arch="YOLOLiteMS", backbone="tf_efficientnetv2_b2", depth_multiple=1.0,
                          width_multiple=1.0, fpn_channels=328, head_depth=2),
    # round 5: the two remaining yamls -- configs/models/edge_xl.yaml (hgnetv2_b0), configs/v2_models/yololite_l.yaml
    # (convnextv2_tiny)
    "edge_xl": dict(arch="YOLOLiteMS_CPU", backbone="hgnetv2_b0", depth_multiple=1.0, width_multiple=1.0,
                    fpn_channels=256, head_depth=3),
    "yololite_l_v2": dict(arch="YOLOLiteMS", backbone="convnextv2_tiny", depth_multiple=1.0, width_multiple=1.0,
                          fpn_channels=512, head_depth=3),
}


def zoo_meta(name: str, num_classes: int = 80, img_size: int = 640, **kw) -> dict:
    """meta for one of the reference's configs/models/*.yaml (kw: use_p6, use_p2, anchors, seg, ...)."""
    return make_meta(num_classes=num_classes, img_size=img_size, **MODEL_ZOO[name], **kw)


_ACT = {"none": 0, "relu": 1, "relu6": 2, "silu": 3, "gelu": 4, "relu_lab": 5}
_OP_STEM, _OP_CONV, _OP_DW, _OP_STEMBLOCK, _OP_SE = 0, 1, 2, 3, 4
_OP_POOL, _OP_COPY, _OP_LN, _OP_GRN, _OP_NHWC4 = 5, 6, 7, 8, 9
DW_PROLOGUE_LDS_MAX = 32 * 1024      # bytes; mirrors YL_DW_LDS_MAX in csrc/yl_api.hip


def _query_fused_block(c_in, c_mid, c_out, dk, ds, oh, ow) -> int:
    """yl_query_fused_block: the LIBRARY says whether a fused inverted-residual block of this shape is instantiated
    (1 = yl_ir_kernel, 2 = yl_uib_kernel, 0 = no) -- host-side code of the .so, no device needed.  (the
    hand-copied mirrors of the kernels' shape tables that used to live here are gone.)"""
    return _query_cached(int(c_in), int(c_mid), int(c_out), int(dk), int(ds), int(oh), int(ow))


def _query_dw_prologue(c_in, c_out, dk, ds, oh, ow) -> int:
    """yl_query_dw_prologue: 2 = the streamed-tap depthwise -> 1x1 kernel takes the pair, 1 = the generic prologue kernels, 0 = no"""
    return _query_cached("dw", int(c_in), int(c_out), int(dk), int(ds), int(oh), int(ow))


@functools.lru_cache(maxsize=None)
def _query_cached(*shape) -> int:
    from . import _lib
    try:
        lib = _lib.load()
    except Exception as e:      # program inspection on a box without the built library: say what is missing
        raise _lib.YoloLiteHipError(
            "build_program() asks libyololite_hip.so which fused block shapes are instantiated (yl_query_fused_block, host "
            f"code, no GPU needed) and the library could not be loaded: {e}.  Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc cross-compiles without a GPU)") from e
    if shape and shape[0] == "dw":
        return int(lib.yl_query_dw_prologue(*shape[1:]))
    return int(lib.yl_query_fused_block(*shape))


class _Builder:
    def __init__(self, sd: Dict[str, np.ndarray], prog: Program, fuse_dw, fuse_stem=True, fuse_uib=True, fuse_ir=True,
                 fuse_uir=True, fuse_lat=True, fuse_chain=True, fuse_dws=True):
        self.sd, self.p, self.fuse_dw, self.fuse_stem, self.fuse_uib = sd, prog, fuse_dw, fuse_stem, fuse_uib
        self.fuse_ir, self.fuse_uir, self.fuse_lat, self.fuse_chain = fuse_ir, fuse_uir, fuse_lat, fuse_chain
        self.fuse_dws = fuse_dws

    # ---- state-dict access
    def get(self, key: str, shape: Tuple[int, ...]) -> np.ndarray:
        self.p.known_keys.add(key)
        if isinstance(self.sd, SynthStateDict):
            self.sd.make(key, shape)
        if key not in self.sd:
            raise KeyError(key)
        self.p.used_keys.add(key)
        a = np.asarray(self.sd[key], dtype=np.float64)
        # a Linear weight [out, in] stored as a 1x1 Conv2d weight [out, in, 1, 1] (timm's ConvNeXt blocks with conv_mlp=True:
        # the nano / pico / femto / atto variants, as recalled -- ADVICE r05) is the same matrix
        if a.ndim == 4 and len(shape) == 2 and a.shape[2:] == (1, 1) and tuple(a.shape[:2]) == tuple(shape):
            a = a.reshape(shape)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{key}: checkpoint shape {tuple(a.shape)} != expected {tuple(shape)}")
        return a

    def fold(self, conv: str, bn: Optional[str], eps: float, bias: bool, wshape: Tuple[int, ...]):
        """conv weight (+bias) with eval-mode BatchNorm folded in (float64 arithmetic, fp32 result)."""
        w = self.get(conv + ".weight", wshape)
        co = (wshape[0],)
        b = self.get(conv + ".bias", co) if bias else np.zeros(w.shape[0])
        if bn is not None:
            g, beta = self.get(bn + ".weight", co), self.get(bn + ".bias", co)
            mu, var = self.get(bn + ".running_mean", co), self.get(bn + ".running_var", co)
            self.p.known_keys.add(bn + ".num_batches_tracked")
            sc = g / np.sqrt(var + eps)
            w = w * sc[:, None, None, None]
            b = (b - mu) * sc + beta
        return np.ascontiguousarray(w, np.float32), np.ascontiguousarray(b, np.float32)

    # ---- tensors
    def slot(self, h, w, c) -> int:
        self.p.slots.append((int(h), int(w), int(c)))
        return len(self.p.slots) - 1

    def dims(self, s):
        return self.p.slots[s]

    @staticmethod
    def geom(h, k, s, same):
        if same and s > 1:
            out = -(-h // s)
            total = max((out - 1) * s + k - h, 0)
            return out, total // 2
        p = k // 2
        return (h + 2 * p - k) // s + 1, p

    # ---- layer emitters
    def stem(self, conv, bn, eps, act, cout, k, s, same):
        S = self.p.img_size
        oh, pad = self.geom(S, k, s, same)
        w, b = self.fold(conv, bn, eps, False, (cout, 3, k, k))
        o = self.slot(oh, oh, cout)
        self.p.layers.append(Layer(_OP_STEM, -1, o, 3, cout, k, s, pad, pad, _ACT[act], w, b, name=conv,
                                   macs=oh * oh * cout * 3 * k * k, bytes_in=4 * 3 * S * S,
                                   bytes_out=4 * oh * oh * cout))
        return o

    def stemblock(self, prefix, eps, act, c1, c2, c3):
        """stem 3x3 s2 -> blocks.0.0 (3x3 s2) -> optional blocks.0.1 (1x1) as ONE launch (symmetric padding)."""
        S = self.p.img_size
        sh, pad = self.geom(S, 3, 2, False)
        oh, _ = self.geom(sh, 3, 2, False)
        w1, b1 = self.fold(prefix + "conv_stem", prefix + "bn1", eps, False, (c1, 3, 3, 3))
        w2, b2 = self.fold(prefix + "blocks.0.0.conv", prefix + "blocks.0.0.bn1", eps, False, (c2, c1, 3, 3))
        w3 = b3 = None
        if c3:
            w3, b3 = self.fold(prefix + "blocks.0.1.conv", prefix + "blocks.0.1.bn1", eps, False, (c3, c2, 1, 1))
        cout = c3 or c2
        o = self.slot(oh, oh, cout)
        macs = sh * sh * c1 * 27 + oh * oh * c2 * c1 * 9 + oh * oh * c3 * c2
        self.p.layers.append(Layer(_OP_STEMBLOCK, -1, o, 3, c1, 3, 2, pad, pad, _ACT[act], w1, b1,
                                   c2=c2, act2=_ACT[act], c3=c3, act3=_ACT[act], w2=w2, b2=b2, w3=w3, b3=b3,
                                   name=prefix + "conv_stem+blocks.0", macs=macs, bytes_in=4 * 3 * S * S,
                                   bytes_out=4 * oh * oh * cout))
        return o

    def stemdw(self, prefix, eps, act, c1, c3, same):
        """stem 3x3 s2 -> blocks.0.0 as a DepthwiseSeparable block (depthwise 3x3 s1 + 1x1, no residual) as ONE launch
        (yl_stemdw_kernel, round 6): timm's EfficientNet-Lite entry, TF-SAME or symmetric stem padding."""
        S = self.p.img_size
        sh, pad = self.geom(S, 3, 2, same)
        w1, b1 = self.fold(prefix + "conv_stem", prefix + "bn1", eps, False, (c1, 3, 3, 3))
        w2, b2 = self.fold(prefix + "blocks.0.0.conv_dw", prefix + "blocks.0.0.bn1", eps, False, (c1, 1, 3, 3))
        w3, b3 = self.fold(prefix + "blocks.0.0.conv_pw", prefix + "blocks.0.0.bn2", eps, False, (c3, c1, 1, 1))
        o = self.slot(sh, sh, c3)
        macs = sh * sh * c1 * 27 + sh * sh * (c1 * 9 + c1 * c3)
        self.p.layers.append(Layer(_OP_STEMBLOCK, -1, o, 3, c1, 3, 2, pad, pad, _ACT[act], w1, b1,
                                   c2=c1, act2=_ACT[act], c3=c3, act3=_ACT["none"], w2=w2, b2=b2, w3=w3, b3=b3,
                                   dw_k=3, dw_stride=1, dw_pad_t=1, dw_pad_l=1,
                                   name=prefix + "conv_stem+blocks.0.0", macs=macs, bytes_in=4 * 3 * S * S,
                                   bytes_out=4 * sh * sh * c3))
        return o

    def uib_fusable(self, x, cmid, cout, dk):
        """per-wave fused block (yl_uib_kernel; option fuse_uib): stride 1, grids multiples of 4"""
        h, w, c1 = self.dims(x)
        if not self.fuse_uib or (h & 3) or (w & 3):
            return False
        return _query_fused_block(c1, cmid, cout, dk, 1, h, w) in (1, 2)

    def ir_fusable(self, x, cmid, cout, dk, ds, oh, ow):
        """workgroup-level-halo fused block (yl_ir_kernel).  fuse_dw == "dw3": only 3x3 depthwise convs are fused
        anywhere in the program, this kernel included."""
        if not self.fuse_ir or (self.fuse_dw == "dw3" and dk != 3):
            return False
        return _query_fused_block(self.dims(x)[2], cmid, cout, dk, ds, oh, ow) == 1

    def uib(self, x, pre, eps, act, cmid, cout, dk, res,
            keys=("pw_exp.conv", "pw_exp.bn", "dw_mid.conv", "dw_mid.bn", "pw_proj.conv", "pw_proj.bn"), ds=1, same=False):
        """pw_exp(+BN+act) -> dw_mid dk x dk (stride ds) (+BN+act) -> pw_proj(+BN)(+res) as ONE launch.  `keys`: parameter
        names of the three conv / BN pairs (MobileNetV4 UIB by default; EfficientNet-style InvertedResidual passes its own)."""
        h, w, c1 = self.dims(x)
        oh, pad = self.geom(h, dk, ds, same)
        ow, _ = self.geom(w, dk, ds, same)
        w2, b2 = self.fold(pre + keys[0], pre + keys[1], eps, False, (cmid, c1, 1, 1))
        dww, dwb = self.fold(pre + keys[2], pre + keys[3], eps, False, (cmid, 1, dk, dk))
        wp, bp = self.fold(pre + keys[4], pre + keys[5], eps, False, (cout, cmid, 1, 1))
        o = self.slot(oh, ow, cout)
        L = Layer(_OP_CONV, x, o, cmid, cout, 1, 1, 0, 0, _ACT["none"], wp, bp, res_slot=res,
                  dw_k=dk, dw_stride=ds, dw_pad_t=pad, dw_pad_l=pad, dw_act=_ACT[act], dw_w=dww, dw_b=dwb,
                  c2=c1, act2=_ACT[act], w2=w2, b2=b2, name=pre + ("ir" if keys[0] == "conv_pw" else "uib"),
                  macs=h * w * c1 * cmid + oh * ow * (cmid * dk * dk + cmid * cout),
                  bytes_in=4 * (h * w * c1 + (oh * ow * cout if res >= 0 else 0)), bytes_out=4 * oh * ow * cout)
        self.p.layers.append(L)
        return o

    def lateral_smooth(self, x, lat, sm, F_, up):
        """lateral{k} (1x1, bias, no BN / act, + nearest-upsampled `up`) -> smooth{k}.block.0 (dw3, no bias) ->
        block.1 (1x1) + block.2 (BN) + ReLU as ONE fused launch (model_v2.py:23-39,359-361)."""
        h, w, c1 = self.dims(x)
        w2, b2 = self.fold(lat, None, 0.0, True, (F_, c1, 1, 1))
        dww, _ = self.fold(sm + "0", None, 1e-5, False, (F_, 1, 3, 3))
        wp, bp = self.fold(sm + "1", sm + "2", 1e-5, False, (F_, F_, 1, 1))
        o = self.slot(h, w, F_)
        L = Layer(_OP_CONV, x, o, F_, F_, 1, 1, 0, 0, _ACT["relu"], wp, bp, up_slot=up,
                  dw_k=3, dw_stride=1, dw_pad_t=1, dw_pad_l=1, dw_act=_ACT["none"], dw_w=dww, dw_b=None,
                  c2=c1, act2=_ACT["none"], w2=w2, b2=b2, name=lat + "+" + sm + "1",
                  macs=h * w * (c1 * F_ + F_ * 9 + F_ * F_),
                  bytes_in=4 * (h * w * c1 + (self.dims(up)[0] * self.dims(up)[1] * F_ if up >= 0 else 0)),
                  bytes_out=4 * h * w * F_)
        self.p.layers.append(L)
        return o

    def dw(self, x, conv, bn, eps, act, k, s, same):
        """stand-alone depthwise conv (+BN+act) as its own layer"""
        h, wd, c = self.dims(x)
        w, b = self.fold(conv, bn, eps, False, (c, 1, k, k))
        oh, pad = self.geom(h, k, s, same)
        ow, _ = self.geom(wd, k, s, same)
        y = self.slot(oh, ow, c)
        self.p.layers.append(Layer(_OP_DW, x, y, c, c, k, s, pad, pad, _ACT[act], w, b, name=conv, macs=oh * ow * c * k * k,
                                   bytes_in=4 * h * wd * c, bytes_out=4 * oh * ow * c))
        return y

    def se(self, x, pre, rd, act):
        """timm SqueezeExcite gate of tensor x: [1,1,C] slot = sigmoid(conv_expand(act(conv_reduce(mean_hw(x)))))"""
        h, wd, c = self.dims(x)
        w1, b1 = self.get(pre + "conv_reduce.weight", (rd, c, 1, 1)), self.get(pre + "conv_reduce.bias", (rd,))
        w2, b2 = self.get(pre + "conv_expand.weight", (c, rd, 1, 1)), self.get(pre + "conv_expand.bias", (c,))
        g = self.slot(1, 1, c)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        self.p.layers.append(Layer(_OP_SE, x, g, c, rd, 1, 1, 0, 0, _ACT[act], f32(w1), f32(b1), c2=c, w2=f32(w2), b2=f32(b2),
                                   name=pre + "gate", macs=2 * c * rd, bytes_in=4 * h * wd * c, bytes_out=4 * c))
        return g

    def conv(self, x, conv, bn, eps, act, cout, k=1, s=1, same=False, bias=False, res=-1, up=-1, head_level=-1,
             dw=None, out_hw=None, wb=None, name=None, in_shift=0, chain=None, scale=-1):
        """dense conv; dw = dict(conv=..., bn=..., eps=..., act=..., k=..., s=..., bias=False) is a depthwise
        conv applied to x first (fused as a prologue when enabled, otherwise emitted as its own layer)."""
        h, wd, cin = self.dims(x)
        h, wd = h << in_shift, wd << in_shift           # the conv sees the nearest-upsampled tensor
        pro = None
        if dw is not None:
            dww, dwb = self.fold(dw["conv"], dw.get("bn"), dw.get("eps", 1e-5), dw.get("bias", False),
                                 (cin, 1, dw["k"], dw["k"]))
            doh, dpad = self.geom(h, dw["k"], dw["s"], same)
            dow, _ = self.geom(wd, dw["k"], dw["s"], same)
            dmacs = doh * dow * cin * dw["k"] ** 2
            fuse = (dw["k"] == 3) if self.fuse_dw == "dw3" else bool(self.fuse_dw)      # "auto" -> all
            # the prologue keeps the depthwise taps + bias of all Cin channels in LDS (yl_conv.hip) -- or, for the wide
            # EfficientNet-Lite conv_dw -> conv_pwl pairs, streams them with the 1x1 weights (yl_conv_dws_kernel, round 5): the
            # library says which (yl_query_dw_prologue)
            if fuse and (dw["k"] ** 2 + 1) * cin * 4 > DW_PROLOGUE_LDS_MAX:
                fuse = (self.fuse_dws and k == 1 and s == 1 and head_level < 0 and scale < 0 and not chain
                        and _query_dw_prologue(cin, cout, dw["k"], dw["s"], doh, dow) == 2)
            if fuse and k == 1 and s == 1:
                pro = dict(k=dw["k"], s=dw["s"], pad=dpad, act=_ACT[dw["act"]], w=dww,
                           b=dwb if (dw.get("bn") or dw.get("bias")) else None, macs=dmacs)
                oh, ow = doh, dow
            else:
                y = self.slot(doh, dow, cin)
                self.p.layers.append(Layer(_OP_DW, x, y, cin, cin, dw["k"], dw["s"], dpad, dpad, _ACT[dw["act"]],
                                           dww, dwb, name=dw["conv"], macs=dmacs, bytes_in=4 * h * wd * cin,
                                           bytes_out=4 * doh * dow * cin))
                x, h, wd = y, doh, dow
        if wb is None:
            w, b = self.fold(conv, bn, eps, bias, (cout, cin, k, k))
        else:
            w, b = wb
        if pro is None:
            oh, pad = self.geom(h, k, s, same)
            ow, _ = self.geom(wd, k, s, same)
        else:
            pad = 0
        if head_level >= 0:
            o = -1
        else:
            o = self.slot(oh, ow, chain["cout"] if chain else cout)
        L = Layer(_OP_CONV, x, o, cin, cout, k, s, pad, pad, _ACT[act], w, b, in_shift=in_shift, res_slot=res, up_slot=up,
                  head_level=head_level, scale_slot=scale, name=name or conv, macs=oh * ow * cout * cin * k * k,
                  bytes_in=4 * h * wd * cin, bytes_out=4 * oh * ow * cout)
        if chain:           # a 1x1 conv (+BN+act) chained in the same launch: chain = dict(conv, bn, eps, act, cout)
            w3, b3 = self.fold(chain["conv"], chain["bn"], chain["eps"], False, (chain["cout"], cout, 1, 1))
            L.c3, L.act3, L.w3, L.b3 = chain["cout"], _ACT[chain["act"]], w3, b3
            L.name = (name or conv) + "+" + chain["conv"].split(".")[-2] + ".conv"
            L.macs += oh * ow * cout * chain["cout"]
            L.bytes_out = 4 * oh * ow * chain["cout"]
        if res >= 0 or up >= 0:
            L.bytes_in += 4 * oh * ow * cout if res >= 0 else 4 * self.dims(up)[0] * self.dims(up)[1] * cout
        if pro is not None:
            L.dw_k, L.dw_stride, L.dw_pad_t, L.dw_pad_l, L.dw_act = pro["k"], pro["s"], pro["pad"], pro["pad"], pro["act"]
            L.dw_w, L.dw_b = pro["w"], pro["b"]
            L.macs += pro["macs"]
        self.p.layers.append(L)
        return o


# timm hgnetv2 (models/hgnet.py) and convnextv2 (models/convnext.py) as DATA, like BACKBONES (recollection; see
# oracle/backbones.py for the restated modules and the published-parameter checksums):
#   name -> (stem (mid, out), stages [(in, mid, out, blocks, downsample, light, kernel, layer_num)], use_lab)
HGNET: Dict[str, tuple] = {
    "hgnetv2_b0": ((16, 16), [(16, 16, 64, 1, False, False, 3, 3), (64, 32, 256, 1, True, False, 3, 3),
                              (256, 64, 512, 2, True, True, 5, 3), (512, 128, 1024, 1, True, True, 5, 3)], True),
    "oracle_tiny_hg": ((8, 8), [(8, 8, 16, 1, False, False, 3, 2), (16, 8, 32, 1, True, False, 3, 3),
                                (32, 8, 48, 2, True, True, 5, 2), (48, 16, 64, 1, True, True, 5, 3)], True),
}
#   name -> (depths, dims, depthwise kernel)
CONVNEXT: Dict[str, tuple] = {
    "convnextv2_tiny": ((3, 3, 9, 3), (96, 192, 384, 768), 7),
    "convnextv2_nano": ((2, 2, 8, 2), (80, 160, 320, 640), 7),
    "convnextv2_pico": ((2, 2, 6, 2), (64, 128, 256, 512), 7),
    "convnextv2_femto": ((2, 2, 6, 2), (48, 96, 192, 384), 7),
    "convnextv2_atto": ((2, 2, 6, 2), (40, 80, 160, 320), 7),
    "oracle_tiny_cnx": ((1, 2, 2, 1), (8, 16, 24, 32), 7),
}


class _Ops:
    """Emitters for the ABI-v5 op kinds on top of a _Builder (plain layers: nothing is fused across them)."""

    def __init__(self, b: "_Builder"):
        self.b = b

    def raw_conv(self, x, w, bias, k, s, pad, oh, ow, act, name, groups=1, lab=None, res=-1, scale=-1):
        b = self.b
        h, wd, cin = b.dims(x)
        cout = w.shape[0]
        o = b.slot(oh, ow, cout)
        f32 = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
        if groups == 1:
            L = Layer(_OP_CONV, x, o, cin, cout, k, s, pad, pad, _ACT[act], f32(w), f32(bias), res_slot=res, scale_slot=scale,
                      name=name, macs=oh * ow * cout * cin * k * k, bytes_in=4 * h * wd * cin, bytes_out=4 * oh * ow * cout)
        else:
            assert groups == cin == cout
            L = Layer(_OP_DW, x, o, cin, cout, k, s, pad, pad, _ACT[act], f32(w), f32(bias), res_slot=res, name=name,
                      macs=oh * ow * cout * k * k, bytes_in=4 * h * wd * cin, bytes_out=4 * oh * ow * cout)
        if lab is not None:
            L.lab_scale, L.lab_bias = float(lab[0]), float(lab[1])
        if res >= 0:
            L.bytes_in += 4 * oh * ow * cout
        b.p.layers.append(L)
        return o

    def plain(self, op, x, out_hwc, name, **kw):
        b = self.b
        h, wd, c = b.dims(x) if x >= 0 else (b.p.img_size, b.p.img_size, 3)
        o = kw.pop("out", None)
        if o is None:
            o = b.slot(*out_hwc)
        oh, ow, oc = b.dims(o)
        L = Layer(op, x, o, c if x >= 0 else 3, kw.pop("cout", c), kw.pop("k", 1), kw.pop("s", 1), 0, 0, 0, kw.pop("w", None),
                  kw.pop("bias", None), name=name, macs=0, bytes_in=4 * h * wd * c, bytes_out=4 * oh * ow * (c if op == _OP_COPY else oc))
        for kk, v in kw.items():
            setattr(L, kk, v)
        b.p.layers.append(L)
        return o

    def cat(self, xs, name):
        """torch.cat(xs, dim=1): one channel-slice copy per input into a fresh slot"""
        b = self.b
        h, w, _ = b.dims(xs[0])
        total = sum(b.dims(x)[2] for x in xs)
        o = b.slot(h, w, total)
        off = 0
        for i, x in enumerate(xs):
            self.plain(_OP_COPY, x, None, f"{name}[{i}]", out=o, out_ch_off=off)
            off += b.dims(x)[2]
        return o


def _hgnet_backbone(b: "_Builder", name: str, prefix: str):
    """timm HighPerfGpuNet features (hgnetv2): see oracle/backbones.py HgFeatureBackbone for the restated modules."""
    (smid, sout), stages, use_lab = HGNET[name]
    ops = _Ops(b)
    S = b.p.img_size

    def cba(x, key, cout, k, s=1, dw=False, use_act=True, same_size=False, res=-1):
        """ConvBNAct: conv (no bias, pad (k-1)//2) + BN(eps 1e-5) + ReLU + LAB.  same_size: the input was zero-extended by one
        row / column at the bottom / right (StemV2's F.pad): a k = 2 conv then keeps the size"""
        h, wd, cin = b.dims(x)
        w, bias = b.fold(key + ".conv", key + ".bn", 1e-5, False, (cout, 1 if dw else cin, k, k))
        pad = (k - 1) // 2
        oh, ow = ((h, wd) if same_size else ((h + 2 * pad - k) // s + 1, (wd + 2 * pad - k) // s + 1))
        lab, act = None, ("relu" if use_act else "none")
        if use_act and use_lab:
            lab = (float(b.get(key + ".lab.scale", (1,))[0]), float(b.get(key + ".lab.bias", (1,))[0]))
            act = "relu_lab"
        return ops.raw_conv(x, w, bias, k, s, pad, oh, ow, act, key + ".conv", groups=(cin if dw else 1), lab=lab, res=res)

    # ---- StemV2
    sh = (S + 2 - 3) // 2 + 1
    key = prefix + "stem.stem1"
    w1, b1 = b.fold(key + ".conv", key + ".bn", 1e-5, False, (smid, 3, 3, 3))
    lab, act = None, "relu"
    if use_lab:
        lab = (float(b.get(key + ".lab.scale", (1,))[0]), float(b.get(key + ".lab.bias", (1,))[0]))
        act = "relu_lab"
    if smid in (16, 32):                 # the MFMA stem kernel's shapes (reads NCHW directly)
        x = b.slot(sh, sh, smid)
        L = Layer(_OP_STEM, -1, x, 3, smid, 3, 2, 1, 1, _ACT[act], w1, b1, name=key + ".conv", macs=sh * sh * smid * 27,
                  bytes_in=4 * 3 * S * S, bytes_out=4 * sh * sh * smid)
        if lab:
            L.lab_scale, L.lab_bias = lab
        b.p.layers.append(L)
    else:                                # NHWC copy of the input with a zero fourth channel, then the generic conv
        x0 = ops.plain(_OP_NHWC4, -1, (S, S, 4), prefix + "input.nhwc4", cout=4)
        w4 = np.zeros((smid, 4, 3, 3), np.float32)
        w4[:, :3] = w1
        x = ops.raw_conv(x0, w4, b1, 3, 2, 1, sh, sh, act, key + ".conv", lab=lab)
    x2 = cba(x, prefix + "stem.stem2a", smid // 2, 2, same_size=True)
    x2 = cba(x2, prefix + "stem.stem2b", smid, 2, same_size=True)
    x1 = ops.plain(_OP_POOL, x, (sh, sh, smid), prefix + "stem.pool", k=2, s=1)
    x = ops.cat([x1, x2], prefix + "stem.cat")
    x = cba(x, prefix + "stem.stem3", smid, 3, 2)
    x = cba(x, prefix + "stem.stem4", sout, 1)
    red, feats = 4, []
    for si, (cin, mid, cout, nb, ds, light, k, ln) in enumerate(stages):
        sp = f"{prefix}stages_{si}."
        if ds:
            x = cba(x, sp + "downsample", cin, 3, 2, dw=True, use_act=False)
            red *= 2
        for bi in range(nb):
            bp = f"{sp}blocks.{bi}."
            ident, parts, y = x, [x], x
            for li in range(ln):
                if light:
                    y = cba(y, f"{bp}layers.{li}.conv1", mid, 1, use_act=False)
                    y = cba(y, f"{bp}layers.{li}.conv2", mid, k, dw=True)
                else:
                    y = cba(y, f"{bp}layers.{li}", mid, 3)
                parts.append(y)
            c = ops.cat(parts, bp + "cat")
            a = cba(c, bp + "aggregation.0", cout // 2, 1)
            x = cba(a, bp + "aggregation.1", cout, 1, res=(ident if bi > 0 else -1))
        feats.append((x, cout, red))
    return feats


def _convnext_backbone(b: "_Builder", name: str, prefix: str):
    """timm ConvNeXt features with GRN (convnextv2): see oracle/backbones.py ConvNeXtFeatureBackbone."""
    depths, dims, dk = CONVNEXT[name]
    ops = _Ops(b)
    S = b.p.img_size
    f32 = lambda a: np.ascontiguousarray(a, np.float32)

    def ln(x, key):
        h, w, c = b.dims(x)
        return ops.plain(_OP_LN, x, (h, w, c), key, w=f32(b.get(key + ".weight", (c,))), bias=f32(b.get(key + ".bias", (c,))),
                         eps=1e-6)

    x0 = ops.plain(_OP_NHWC4, -1, (S, S, 4), prefix + "input.nhwc4", cout=4)
    w = b.get(prefix + "stem_0.weight", (dims[0], 3, 4, 4))
    w4 = np.zeros((dims[0], 4, 4, 4), np.float32)
    w4[:, :3] = w
    x = ops.raw_conv(x0, w4, b.get(prefix + "stem_0.bias", (dims[0],)), 4, 4, 0, S // 4, S // 4, "none", prefix + "stem_0")
    x = ln(x, prefix + "stem_1")
    red, prev, feats = 4, dims[0], []
    for si, (depth, c) in enumerate(zip(depths, dims)):
        sp = f"{prefix}stages_{si}."
        h, wd, _ = b.dims(x)
        if si > 0:
            y = ln(x, sp + "downsample.0")
            x = ops.raw_conv(y, b.get(sp + "downsample.1.weight", (c, prev, 2, 2)), b.get(sp + "downsample.1.bias", (c,)),
                             2, 2, 0, h // 2, wd // 2, "none", sp + "downsample.1")
            red *= 2
            h, wd = h // 2, wd // 2
        for bi in range(depth):
            bp = f"{sp}blocks.{bi}."
            y = ops.raw_conv(x, b.get(bp + "conv_dw.weight", (c, 1, dk, dk)), b.get(bp + "conv_dw.bias", (c,)), dk, 1, dk // 2,
                             h, wd, "none", bp + "conv_dw", groups=c)
            y = ln(y, bp + "norm")
            w1 = b.get(bp + "mlp.fc1.weight", (4 * c, c)).reshape(4 * c, c, 1, 1)
            hid = ops.raw_conv(y, w1, b.get(bp + "mlp.fc1.bias", (4 * c,)), 1, 1, 0, h, wd, "gelu", bp + "mlp.fc1")
            gw, gb = b.get(bp + "mlp.grn.weight", (4 * c,)), b.get(bp + "mlp.grn.bias", (4 * c,))
            g = ops.plain(_OP_GRN, hid, (1, 1, 4 * c), bp + "mlp.grn", w=f32(gw), eps=1e-6)
            w2 = b.get(bp + "mlp.fc2.weight", (c, 4 * c))
            # GRN: y = x + (beta + gamma * (x * n)) = x * (1 + gamma * n) + beta; the gate multiplies fc2's input and beta
            # goes through fc2 once, on the host (float64): b2' = b2 + W2 . beta
            b2 = b.get(bp + "mlp.fc2.bias", (c,)) + w2 @ gb
            x = ops.raw_conv(hid, w2.reshape(c, 4 * c, 1, 1), b2, 1, 1, 0, h, wd, "none", bp + "mlp.fc2", res=x, scale=g)
        prev = c
        feats.append((x, c, red))
    return feats


def _backbone(b: _Builder, name: str, prefix: str = "backbone.") -> List[Tuple[int, int, int]]:
    """Emit the feature extractor; returns [(slot, channels, reduction)] for every feature tap."""
    if name in HGNET:
        return _hgnet_backbone(b, name, prefix)
    if name in CONVNEXT:
        return _convnext_backbone(b, name, prefix)
    if name not in BACKBONES:
        raise ValueError(f"backbone '{name}' has no layer table (known: {sorted(BACKBONES)})")
    spec = BACKBONES[name]
    arch, act, eps, same = spec["arch"], spec["act"], spec["eps"], spec["same"]
    ns = len(arch)
    s0 = [_parse(t) for t in arch[0]]
    rlim = spec.get("round_limit", 0.9)          # channel rounding of the OUTPUT widths (expansions always use 0.9)
    c_s0 = [_make_divisible(d["c"] * spec["cmult"], 8, rlim) for d in s0]
    fused_entry = (b.fuse_stem and not same and act in ("relu", "relu6") and spec["stem"] in (16, 32) and len(s0) in (1, 2)
                   and s0[0]["type"] == "cn" and s0[0]["k"] == 3 and s0[0]["s"] == 2 and s0[0]["r"] == 1
                   and (len(s0) == 1 or (s0[1]["type"] == "cn" and s0[1]["k"] == 1 and s0[1]["s"] == 1 and s0[1]["r"] == 1))
                   and all(c <= 32 and c % 4 == 0 for c in c_s0))
    # EfficientNet-Lite entry (round 6): stem -> DepthwiseSeparable blocks.0.0 (3x3 s1, no residual: the widths differ) as one
    # launch; further repeats of stage 0 (depth multipliers > 1) follow as ordinary layers
    s0_rep = s0[0]["r"] if (spec["dmult"] == 1.0 or (spec["fix_first_last"])) else int(math.ceil(s0[0]["r"] * spec["dmult"]))
    fused_dw_entry = (b.fuse_stem and not fused_entry and act in ("relu", "relu6") and spec["stem"] == 32 and len(s0) == 1
                      and s0[0]["type"] == "ds" and s0[0]["k"] == 3 and s0[0]["s"] == 1 and s0_rep == 1
                      and 4 <= c_s0[0] <= 32 and c_s0[0] % 4 == 0 and c_s0[0] != spec["stem"]
                      and b.p.img_size % 16 == 0)
    feats = []
    if fused_dw_entry:
        x = b.stemdw(prefix, eps, act, spec["stem"], c_s0[0], same)
        cin, red = c_s0[0], 2
        if _parse(arch[1][0])["s"] > 1:
            feats.append((x, cin, red))
    elif fused_entry:
        x = b.stemblock(prefix, eps, act, spec["stem"], c_s0[0], c_s0[1] if len(s0) == 2 else 0)
        cin, red = c_s0[-1], 4
        feats.append((-1, spec["stem"], 2))               # stem tap: never materialised, never consumed
        if _parse(arch[1][0])["s"] > 1:
            feats.append((x, cin, red))
    else:
        x = b.stem(prefix + "conv_stem", prefix + "bn1", eps, act, spec["stem"], 3, 2, same)
        cin, red = spec["stem"], 2
        if s0[0]["s"] > 1:
            feats.append((x, cin, red))
    for si, stage in enumerate(arch):
        if (fused_entry or fused_dw_entry) and si == 0:
            continue
        bi = 0
        skip_next = False
        for sidx, bstr in enumerate(stage):
            if skip_next:                      # this entry went out as the chained 1x1 of the conv before it
                skip_next = False
                bi += 1
                continue
            d = _parse(bstr)
            rep = d["r"]
            if spec["dmult"] != 1.0 and not (spec["fix_first_last"] and si in (0, ns - 1)):
                rep = int(math.ceil(rep * spec["dmult"]))
            cout = _make_divisible(d["c"] * spec["cmult"], 8, rlim)
            for r in range(rep):
                s = d["s"] if r == 0 else 1
                pre = f"{prefix}blocks.{si}.{bi}."
                skip = (cin == cout and s == 1)
                if d["type"] == "cn":
                    chain = None
                    if (d["k"] > 1 and rep == 1 and sidx + 1 < len(stage) and b.fuse_dw is not False and act in ("relu", "relu6")
                            and b.fuse_chain):
                        d2 = _parse(stage[sidx + 1])
                        c2o = _make_divisible(d2["c"] * spec["cmult"], 8, rlim)
                        if (d2["type"] == "cn" and d2["k"] == 1 and d2["s"] == 1 and d2["r"] == 1 and cout <= 96 and cout % 4 == 0
                                and c2o <= 32 and c2o % 4 == 0):
                            # dense k x k conv followed by a 1x1 conv (blocks.1.0 / blocks.1.1 of mobilenetv4_conv_small):
                            # the 1x1 is chained in the epilogue of the k x k launch (yl_conv_mfma_kernel)
                            chain = dict(conv=f"{prefix}blocks.{si}.{bi + 1}.conv", bn=f"{prefix}blocks.{si}.{bi + 1}.bn1",
                                         eps=eps, act=act, cout=c2o)
                    x = b.conv(x, pre + "conv", pre + "bn1", eps, act, cout, d["k"], s, same, chain=chain,
                               res=(x if (d["skip"] and skip) else -1))
                    if chain:
                        skip_next = True
                        cout = chain["cout"]
                elif d["type"] == "er":
                    # fused MBConv (timm EdgeResidual): k x k expansion conv (strided, +BN+act) -> 1x1 projection (+BN) (+res)
                    mid = _make_divisible(cin * d["e"], 8)
                    y = b.conv(x, pre + "conv_exp", pre + "bn1", eps, act, mid, d["k"], s, same)
                    x = b.conv(y, pre + "conv_pwl", pre + "bn2", eps, "none", cout, res=(x if skip else -1))
                elif d["type"] == "ir" and d["se"]:
                    # MBConv with squeeze-excite: 1x1 expand -> depthwise (own launch: the gate needs its whole output) ->
                    # gate (fixed-order mean + 2 FCs) -> 1x1 projection reading the depthwise output times the gate
                    mid = _make_divisible(cin * d["e"], 8)
                    y = b.conv(x, pre + "conv_pw", pre + "bn1", eps, act, mid, same=same)
                    z = b.dw(y, pre + "conv_dw", pre + "bn2", eps, act, d["k"], s, same)
                    g = b.se(z, pre + "se.", int(round(mid * (d["se"] / d["e"]))), act)
                    x = b.conv(z, pre + "conv_pwl", pre + "bn3", eps, "none", cout, res=(x if skip else -1), scale=g)
                elif d["type"] == "uir" and not d["a"] and d["k"] and not same and b.fuse_uir and \
                        b.ir_fusable(x, _make_divisible(cin * d["e"], 8), cout, d["k"], s,
                                     b.geom(b.dims(x)[0], d["k"], s, same)[0], b.geom(b.dims(x)[1], d["k"], s, same)[0]):
                    # MobileNetV4 UIB blocks without a start depthwise, at the shapes yl_ir_kernel is instantiated for
                    # (the 40x40 stage of edge_n / edge_m): expand -> dw -> project in one launch
                    x = b.uib(x, pre, eps, act, _make_divisible(cin * d["e"], 8), cout, d["k"], x if skip else -1, ds=s, same=same)
                elif d["type"] == "uir" and not d["a"] and d["k"] and s == 1 and not same and \
                        b.uib_fusable(x, _make_divisible(cin * d["e"], 8), cout, d["k"]):
                    x = b.uib(x, pre, eps, act, _make_divisible(cin * d["e"], 8), cout, d["k"], x if skip else -1)
                elif d["type"] == "uir":
                    mid = _make_divisible(cin * d["e"], 8)
                    dws = None
                    if d["a"]:
                        dws = dict(conv=pre + "dw_start.conv", bn=pre + "dw_start.bn", eps=eps, act="none",
                                   k=d["a"], s=(s if not d["k"] else 1))
                    y = b.conv(x, pre + "pw_exp.conv", pre + "pw_exp.bn", eps, act, mid, same=same, dw=dws)
                    dwm = None
                    if d["k"]:
                        dwm = dict(conv=pre + "dw_mid.conv", bn=pre + "dw_mid.bn", eps=eps, act=act, k=d["k"], s=s)
                    x = b.conv(y, pre + "pw_proj.conv", pre + "pw_proj.bn", eps, "none", cout, same=same, dw=dwm,
                               res=(x if skip else -1))
                elif d["type"] == "ds":
                    dw = dict(conv=pre + "conv_dw", bn=pre + "bn1", eps=eps, act=act, k=d["k"], s=s)
                    x = b.conv(x, pre + "conv_pw", pre + "bn2", eps, "none", cout, same=same, dw=dw,
                               res=(x if skip else -1))
                elif d["type"] == "ir" and b.ir_fusable(x, _make_divisible(cin * d["e"], 8), cout, d["k"], s,
                                                        b.geom(b.dims(x)[0], d["k"], s, same)[0], b.geom(b.dims(x)[1], d["k"], s, same)[0]):
                    # the early EfficientNet blocks (6x expanded tensor at 320x320 ... 80x80): one launch, the expanded
                    # tensor only as 16-channel slabs of a workgroup's halo region in LDS (yl_ir_kernel, round 3)
                    x = b.uib(x, pre, eps, act, _make_divisible(cin * d["e"], 8), cout, d["k"], x if skip else -1,
                              keys=("conv_pw", "bn1", "conv_dw", "bn2", "conv_pwl", "bn3"), ds=s, same=same)
                elif d["type"] == "ir" and s == 1 and b.uib_fusable(x, _make_divisible(cin * d["e"], 8), cout, d["k"]):
                    # (fuse_uib only; stride 1: TF-SAME padding is the symmetric k // 2 the fused kernel applies.  Measured
                    # on yololite_m B=32: 144-channel dw3 blocks @160x160 0.39 -> 0.35 ms, 288-channel dw5 blocks @80x80
                    # 0.28 -> 0.49 ms: the expansion recomputed on a 5x5 halo costs more than the 0.24 GB it saves)
                    x = b.uib(x, pre, eps, act, _make_divisible(cin * d["e"], 8), cout, d["k"], x if skip else -1,
                              keys=("conv_pw", "bn1", "conv_dw", "bn2", "conv_pwl", "bn3"))
                elif d["type"] == "ir":
                    mid = _make_divisible(cin * d["e"], 8)
                    y = b.conv(x, pre + "conv_pw", pre + "bn1", eps, act, mid, same=same)
                    dw = dict(conv=pre + "conv_dw", bn=pre + "bn2", eps=eps, act=act, k=d["k"], s=s)
                    x = b.conv(y, pre + "conv_pwl", pre + "bn3", eps, "none", cout, same=same, dw=dw,
                               res=(x if skip else -1))
                else:
                    raise ValueError(bstr)
                red *= s
                cin = cout
                bi += 1
        nxt = _parse(arch[si + 1][0])["s"] if si + 1 < ns else 2
        if nxt > 1:
            feats.append((x, cin, red))
    return feats


def build_program(meta: dict, state_dict: Dict[str, "np.ndarray"], fuse_dw="auto",
                  img_size: Optional[int] = None, fuse_stem: bool = True, fuse_uib: bool = False,
                  fuse_ir: Optional[bool] = None, fuse_uir: bool = True, fuse_lat: bool = True,
                  fuse_chain: bool = True, fuse_dws: bool = True) -> Program:
    """meta: the checkpoint's `meta` dict (tools/train.py:62-75); reads the keys
    build_model_from_meta reads (tools/infer.py:35-50).
    fuse_dw: True = every depthwise conv becomes the prologue of the following 1x1 conv, False = none,
    "dw3" = only 3x3; "auto" = measured best policy (currently: all, LDS-halo kernel).
    fuse_ir: EfficientNet-style inverted-residual blocks of the shapes yl_ir_kernel is instantiated for as ONE launch
    (None = on); fuse_uir: likewise the MobileNetV4 UIB blocks without a start depthwise; fuse_lat: FPN lateral + first
    depthwise smooth block as one launch; fuse_dws: wide EfficientNet-Lite conv_dw -> conv_pwl pairs through the streamed-tap
    kernel (yl_query_dw_prologue == 2); fuse_chain: a 1x1 conv chained in the epilogue of the dense k x k conv before
    it.  All of them are arguments (resolved once per model: YOLOLiteHIP.__init__), never process environment: two
    programs built in one process cannot silently differ."""
    if fuse_ir is None:
        fuse_ir = True
    cfg = meta.get("config", {}) or {}
    mcfg = cfg.get("model", {}) or {}
    tcfg = cfg.get("training", {}) or {}
    arch = (meta.get("arch") or mcfg.get("arch") or "YOLOLiteMS").lower()
    backbone = (meta.get("backbone") or mcfg.get("backbone") or "resnet18")
    C = int(meta.get("num_classes") or mcfg.get("num_classes") or 80)
    apl = tuple(meta.get("num_anchors_per_level") or (1, 1, 1))
    fpn = int(mcfg.get("fpn_channels", 128))
    depth_multiple = float(mcfg.get("depth_multiple", 1.0))
    width_multiple = float(mcfg.get("width_multiple", 1.0))
    head_depth = int(mcfg.get("head_depth", 1))
    use_p6 = cfg["training"]["use_p6"]                     # KeyError like the reference (tools/infer.py:49-50)
    use_p2 = cfg["training"]["use_p2"]
    S = int(img_size or tcfg.get("img_size", meta.get("img_size", 640)))
    if arch not in ("yololitems", "yololitems_cpu"):
        raise ValueError(f"Okänd arch i meta/config: {arch}")
    cpu_arch = arch == "yololitems_cpu"
    # build-defined instance-segmentation branch (the reference has no mask code): config.model.seg
    seg = bool(mcfg.get("seg", False))
    NM = int(mcfg.get("num_masks", 32)) if seg else 0
    Cp = int(mcfg.get("proto_channels", 64))

    if isinstance(state_dict, SynthStateDict):
        sd = state_dict
    else:
        sd = {k: (v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)) for k, v in state_dict.items()}
    prog = Program(img_size=S, num_classes=C, level_size=[], level_anchors=[], strides=[])
    on = fuse_dw is not False
    b = _Builder(sd, prog, fuse_dw, fuse_stem, bool(fuse_uib) and on, bool(fuse_ir) and on, bool(fuse_uir) and on,
                 bool(fuse_lat) and on, bool(fuse_chain) and on, bool(fuse_dws) and on)

    feats = _backbone(b, backbone)
    take = 4 if use_p2 else 3
    feats = feats[-take:]
    F_ = int(fpn * width_multiple)
    d = max(1, round(2 * depth_multiple))
    pyr = (["p2"] if use_p2 else []) + ["p3", "p4", "p5"]
    levels = pyr + (["p6"] if use_p6 else [])
    if len(apl) >= 3:
        a3, a4, a5 = (int(v) for v in apl[:3])
        amap = dict(p2=a3, p3=a3, p4=a4, p5=a5, p6=a5)
    else:
        a = int(apl[0]) if len(apl) else 1
        amap = dict(p2=a, p3=a, p4=a, p5=a, p6=a)
    for n_, (slot, ch, red) in zip(pyr, feats):
        prog.feature_slots["c" + n_[1]] = slot

    def smooth_step(x, key, i):
        """layer i of a smooth block: CPU arch [dw3 -> pw -> BN -> ReLU] (model_v2.py:23-39); GPU arch
        [conv3 -> BN -> SiLU] (model_v2.py:15-22)."""
        if cpu_arch:
            p = f"{key}.block."
            return b.conv(x, f"{p}{4 * i + 1}", f"{p}{4 * i + 2}", 1e-5, "relu", F_,
                          dw=dict(conv=f"{p}{4 * i}", act="none", k=3, s=1))
        return b.conv(x, f"{key}.{3 * i}", f"{key}.{3 * i + 1}", 1e-5, "silu", F_, k=3)

    def smooth(x, key):
        for i in range(d):
            x = smooth_step(x, key, i)
        return x

    # top-down pass, P5 first (model_v2.py:359-361): every level's lateral adds the SMOOTHED coarser level, so
    # lateral / smooth form one dependent chain.  The heads below are emitted LEVEL-BATCHED (trunk layer t of
    # every level side by side, then the output convs side by side): the executor launches such runs of
    # independent, identically shaped layers as ONE kernel (YlConvMulti), so the 20x20 / 40x40 heads ride along
    # with the 80x80 head.
    P = {}
    prev = None
    for n_ in reversed(pyr):
        k = n_[1]
        cslot = prog.feature_slots["c" + k]
        ch, cw = b.dims(cslot)[0], b.dims(cslot)[1]
        if cpu_arch and b.fuse_lat and b.ir_fusable(cslot, F_, F_, 3, 1, ch, cw):
            # lateral 1x1 (+bias, + upsample-add of the smoothed coarser level) and the first depthwise smooth block as
            # ONE launch (yl_ir_kernel: the lateral is the "expansion", its output -- which has no other consumer --
            # lives only as 16-channel slabs of a workgroup's halo region in LDS)
            x = b.lateral_smooth(cslot, f"lateral{k}", f"smooth{k}.block.", F_, P[prev] if prev else -1)
            for i in range(1, d):
                x = smooth_step(x, f"smooth{k}", i)
            P[n_] = x
        else:
            lat = b.conv(cslot, f"lateral{k}", None, 0.0, "none", F_, bias=True, up=(P[prev] if prev else -1))
            P[n_] = smooth(lat, f"smooth{k}")
        prev = n_
    if use_p6:                                             # model_v2.py:373-375
        x6 = b.conv(P["p5"], "p6_down", "p6_bn", 1e-5, "relu" if cpu_arch else "silu", F_, k=3, s=2)
        P["p6"] = smooth(x6, "smooth6")
    else:                                                  # parameters exist in every reference checkpoint
        for i in range(d):
            ks = ([f"smooth6.block.{4 * i}.weight", f"smooth6.block.{4 * i + 1}.weight"] +
                  [f"smooth6.block.{4 * i + 2}.{s}" for s in ("weight", "bias", "running_mean", "running_var",
                                                               "num_batches_tracked")]) if cpu_arch else \
                 ([f"smooth6.{3 * i}.weight"] + [f"smooth6.{3 * i + 1}.{s}" for s in
                                                  ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")])
            prog.known_keys.update(ks)
        prog.known_keys.update(["p6_down.weight"] + [f"p6_bn.{s}" for s in
                                                    ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")])
    for n_ in levels:
        prog.feature_slots[n_] = P[n_]
    if seg:
        # prototypes on P3: conv3x3+BN+SiLU -> nearest x2 (folded into the next conv's addressing) -> conv3x3 -> conv1x1
        y = b.conv(P["p3"], "proto.cv1.0", "proto.cv1.1", 1e-5, "silu", Cp, k=3)
        y = b.conv(y, "proto.cv2.0", "proto.cv2.1", 1e-5, "silu", Cp, k=3, in_shift=1)
        prog.proto_slot = b.conv(y, "proto.cv3.0", "proto.cv3.1", 1e-5, "silu", NM, k=1)
        prog.num_masks = NM

    # heads (model_v2.py:42-53,340-350): trunk, then box/obj/cls 1x1 convs fused into ONE GEMM per anchor;
    # level-batched emission (see above): trunk layer t of every level, then output conv a of every level
    X = {n_: P[n_] for n_ in levels}
    for t in range(head_depth):
        for n_ in levels:
            p = f"head{n_[1]}.trunk.{t}.block."
            X[n_] = b.conv(X[n_], p + "1", p + "2", 1e-5, "relu", F_, dw=dict(conv=p + "0", act="none", k=3, s=1))
    Amax = max(amap[n_] for n_ in levels)
    for a in range(Amax):
        for li, n_ in enumerate(levels):
            k = n_[1]
            A = amap[n_]
            if a >= A:
                continue
            wbox, bbox_ = b.get(f"head{k}.out.box.weight", (4 * A, F_, 1, 1)), b.get(f"head{k}.out.box.bias", (4 * A,))
            wobj, bobj = b.get(f"head{k}.out.obj.weight", (A, F_, 1, 1)), b.get(f"head{k}.out.obj.bias", (A,))
            wcls, bcls = b.get(f"head{k}.out.cls.weight", (A * C, F_, 1, 1)), b.get(f"head{k}.out.cls.bias", (A * C,))
            # conv channels are anchor-major (view(B,A,4,S,S))
            w = np.concatenate([wbox[4 * a:4 * a + 4], wobj[a:a + 1], wcls[C * a:C * (a + 1)]], 0)
            bb = np.concatenate([bbox_[4 * a:4 * a + 4], bobj[a:a + 1], bcls[C * a:C * (a + 1)]], 0)
            if seg:
                wmc, bmc = b.get(f"head{k}.out.mc.weight", (A * NM, F_, 1, 1)), b.get(f"head{k}.out.mc.bias", (A * NM,))
                w = np.concatenate([w, wmc[NM * a:NM * (a + 1)]], 0)
                bb = np.concatenate([bb, bmc[NM * a:NM * (a + 1)]], 0)
            b.conv(X[n_], None, None, 0.0, "none", 5 + C + NM, head_level=li,
                   wb=(np.ascontiguousarray(w, np.float32), np.ascontiguousarray(bb, np.float32)),
                   name=f"head{k}.out[a={a}]")
    for n_ in levels:
        prog.level_size.append(b.dims(X[n_])[0])
        prog.level_anchors.append(amap[n_])
    reds = [r for (_, _, r) in feats]
    prog.strides = reds + ([reds[-1] * 2] if use_p6 else [])
    return prog
