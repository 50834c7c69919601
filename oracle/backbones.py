"""ORACLE (test infrastructure only) -- CPU restatement of the timm feature extractors.

The reference gets its backbone from a third-party package:

    timm.create_model(backbone, features_only=True, pretrained=..., out_indices=...)
    (/root/reference/scripts/model/model_v2.py:94,98-100,266,270-272)

``timm`` (pinned only as ``timm>=0.9``, /root/reference/requirements.txt:3) is not
vendored in the reference and not installed in this image, so the arithmetic below is a
restatement of timm's published model definitions FROM RECOLLECTION:

* ``mobilenetv4_conv_small`` / ``mobilenetv4_conv_small_050``  (timm ``_gen_mobilenet_v4``)
* ``tf_efficientnet_lite0..4``                                 (timm ``_gen_efficientnet_lite``)
* ``tf_efficientnetv2_b0..b3``                                 (timm ``_gen_efficientnetv2_base``: ConvBnAct stage,
  fused-MBConv ``er`` = EdgeResidual, MBConv ``ir`` with SqueezeExcite, SiLU, TF-SAME padding, BN eps 1e-3,
  channel rounding with round_limit 0) -- the backbones of /root/reference/configs/v2_models/yololite_{n,s,m}.yaml

PARITY UNPINNED: no reference test or golden vector covers the backbone.  The only
checksums are the published parameter counts (/root/reference/BENCHMARK.md:353-357): edge_n 0.553 M (this
restatement: 0.5524 M at C=3), edge_m 2.950 M (2.949 M), and for the efficientnetv2 family yololite_n 8.923 M
and yololite_m 17.916 M (tests/test_oracle_golden.py::test_param_checksums_of_the_published_models).

Module/parameter names follow timm's state_dict layout (``conv_stem``, ``bn1``,
``blocks.<stage>.<idx>.<conv|bn1|dw_start.conv|...>``) so that reference checkpoints
(``backbone.*`` keys) load without renaming.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------- helpers
def make_divisible(v: float, divisor: int = 8, min_value: Optional[int] = None,
                   round_limit: float = 0.9) -> int:
    min_value = min_value or divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < round_limit * v:
        new_v += divisor
    return new_v


def round_channels(ch: int, multiplier: float = 1.0, divisor: int = 8, round_limit: float = 0.9) -> int:
    if not multiplier:
        return ch
    return make_divisible(ch * multiplier, divisor, round_limit=round_limit)


def _act(name: str) -> nn.Module:
    if name == "relu":
        return nn.ReLU()
    if name == "relu6":
        return nn.ReLU6()
    if name == "silu":
        return nn.SiLU()
    if name in ("none", "", None):
        return nn.Identity()
    raise ValueError(name)


class BatchNormAct2d(nn.BatchNorm2d):
    """BatchNorm2d followed by an activation (timm keeps the act inside the norm layer,
    so the state_dict has only the BN tensors)."""

    def __init__(self, ch: int, eps: float, act: str):
        super().__init__(ch, eps=eps)
        self.act = _act(act)

    def forward(self, x):
        return self.act(super().forward(x))


class Conv2dSame(nn.Conv2d):
    """TF 'SAME' padding: asymmetric, computed from the input size (extra pixel goes
    to the bottom/right)."""

    def forward(self, x):
        ih, iw = x.shape[-2:]
        kh, kw = self.kernel_size
        sh, sw = self.stride
        ph = max((math.ceil(ih / sh) - 1) * sh + (kh - 1) + 1 - ih, 0)
        pw = max((math.ceil(iw / sw) - 1) * sw + (kw - 1) + 1 - iw, 0)
        if ph > 0 or pw > 0:
            x = F.pad(x, (pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
        return F.conv2d(x, self.weight, self.bias, self.stride, (0, 0), self.dilation, self.groups)


def _conv(cin, cout, k, s, groups=1, same=False):
    if same and s > 1:
        return Conv2dSame(cin, cout, k, s, padding=0, groups=groups, bias=False)
    return nn.Conv2d(cin, cout, k, s, padding=k // 2, groups=groups, bias=False)


# ----------------------------------------------------------------------------- blocks
class ConvBnAct(nn.Module):                      # timm 'cn' (``_skip``: residual when shapes allow, efficientnetv2 stage 0)
    def __init__(self, cin, cout, k, s, act, eps, same, skip=False):
        super().__init__()
        self.has_skip = bool(skip) and s == 1 and cin == cout
        self.conv = _conv(cin, cout, k, s, same=same)
        self.bn1 = BatchNormAct2d(cout, eps, act)

    def forward(self, x):
        y = self.bn1(self.conv(x))
        return x + y if self.has_skip else y


class SqueezeExcite(nn.Module):                  # timm efficientnet SqueezeExcite (act = the block's act, gate = sigmoid)
    def __init__(self, ch, rd, act):
        super().__init__()
        self.conv_reduce = nn.Conv2d(ch, rd, 1, bias=True)
        self.act1 = _act(act)
        self.conv_expand = nn.Conv2d(rd, ch, 1, bias=True)

    def forward(self, x):
        g = x.mean((2, 3), keepdim=True)
        g = self.conv_expand(self.act1(self.conv_reduce(g)))
        return x * torch.sigmoid(g)


class EdgeResidual(nn.Module):                   # timm 'er' (fused MBConv): k x k expand (strided) -> 1x1 project
    def __init__(self, cin, cout, k, s, exp_ratio, act, eps, same):
        super().__init__()
        self.has_skip = (cin == cout and s == 1)
        mid = make_divisible(cin * exp_ratio, 8)
        self.conv_exp = _conv(cin, mid, k, s, same=same)
        self.bn1 = BatchNormAct2d(mid, eps, act)
        self.conv_pwl = _conv(mid, cout, 1, 1, same=same)
        self.bn2 = BatchNormAct2d(cout, eps, "none")

    def forward(self, x):
        y = self.bn2(self.conv_pwl(self.bn1(self.conv_exp(x))))
        return x + y if self.has_skip else y


class _ConvNorm(nn.Module):                      # timm ConvNormAct used inside UIB
    def __init__(self, cin, cout, k, s, groups, act, eps, same):
        super().__init__()
        self.conv = _conv(cin, cout, k, s, groups=groups, same=same)
        self.bn = BatchNormAct2d(cout, eps, act)

    def forward(self, x):
        return self.bn(self.conv(x))


class UniversalInvertedResidual(nn.Module):      # timm 'uir' (MobileNetV4 UIB)
    def __init__(self, cin, cout, dw_start_k, dw_mid_k, s, exp_ratio, act, eps, same):
        super().__init__()
        self.has_skip = (cin == cout and s == 1)
        if dw_start_k:
            s0 = s if not dw_mid_k else 1
            self.dw_start = _ConvNorm(cin, cin, dw_start_k, s0, cin, "none", eps, same)
        else:
            self.dw_start = nn.Identity()
        mid = make_divisible(cin * exp_ratio, 8)
        self.pw_exp = _ConvNorm(cin, mid, 1, 1, 1, act, eps, same)
        if dw_mid_k:
            self.dw_mid = _ConvNorm(mid, mid, dw_mid_k, s, mid, act, eps, same)
        else:
            self.dw_mid = nn.Identity()
        self.pw_proj = _ConvNorm(mid, cout, 1, 1, 1, "none", eps, same)

    def forward(self, x):
        y = self.pw_proj(self.dw_mid(self.pw_exp(self.dw_start(x))))
        return x + y if self.has_skip else y


class DepthwiseSeparableConv(nn.Module):         # timm 'ds'
    def __init__(self, cin, cout, k, s, act, eps, same):
        super().__init__()
        self.has_skip = (cin == cout and s == 1)
        self.conv_dw = _conv(cin, cin, k, s, groups=cin, same=same)
        self.bn1 = BatchNormAct2d(cin, eps, act)
        self.conv_pw = _conv(cin, cout, 1, 1, same=same)
        self.bn2 = BatchNormAct2d(cout, eps, "none")

    def forward(self, x):
        y = self.bn2(self.conv_pw(self.bn1(self.conv_dw(x))))
        return x + y if self.has_skip else y


def se_channels(mid: int, se_ratio: float, exp_ratio: float) -> int:
    """timm EfficientNetBuilder (se_from_exp=False): the ratio refers to the block INPUT, so it is divided by the
    expansion ratio before SqueezeExcite applies it to the expanded width; rd_round_fn = python round()."""
    return int(round(mid * (se_ratio / exp_ratio)))


class InvertedResidual(nn.Module):               # timm 'ir' (+ SqueezeExcite between the depthwise conv and the projection)
    def __init__(self, cin, cout, k, s, exp_ratio, act, eps, same, se_ratio=0.0):
        super().__init__()
        self.has_skip = (cin == cout and s == 1)
        mid = make_divisible(cin * exp_ratio, 8)
        self.conv_pw = _conv(cin, mid, 1, 1, same=same)
        self.bn1 = BatchNormAct2d(mid, eps, act)
        self.conv_dw = _conv(mid, mid, k, s, groups=mid, same=same)
        self.bn2 = BatchNormAct2d(mid, eps, act)
        self.se = SqueezeExcite(mid, se_channels(mid, se_ratio, exp_ratio), act) if se_ratio else nn.Identity()
        self.conv_pwl = _conv(mid, cout, 1, 1, same=same)
        self.bn3 = BatchNormAct2d(cout, eps, "none")

    def forward(self, x):
        y = self.bn3(self.conv_pwl(self.se(self.bn2(self.conv_dw(self.bn1(self.conv_pw(x)))))))
        return x + y if self.has_skip else y


# ----------------------------------------------------------------------------- arch strings
def _parse(block: str) -> dict:
    """'uir_r4_a0_k3_s1_e2_c96' -> {'type':'uir','r':4,'a':0,'k':3,'s':1,'e':2.0,'c':96}"""
    parts = block.split("_")
    d = {"type": parts[0], "r": 1, "e": 1.0, "a": 0, "skip": False, "se": 0.0}
    for p in parts[1:]:
        if p == "skip":
            d["skip"] = True
        elif p.startswith("se"):
            d["se"] = float(p[2:])
        else:
            key, val = p[0], p[1:]
            d[key] = float(val) if key == "e" else int(val)
    return d


MNV4_CONV_SMALL = [
    ["cn_r1_k3_s2_e1_c32", "cn_r1_k1_s1_e1_c32"],
    ["cn_r1_k3_s2_e1_c96", "cn_r1_k1_s1_e1_c64"],
    ["uir_r1_a5_k5_s2_e3_c96", "uir_r4_a0_k3_s1_e2_c96", "uir_r1_a3_k0_s1_e4_c96"],
    ["uir_r1_a3_k3_s2_e6_c128", "uir_r1_a5_k5_s1_e4_c128", "uir_r1_a0_k5_s1_e4_c128",
     "uir_r1_a0_k5_s1_e3_c128", "uir_r2_a0_k3_s1_e4_c128"],
    ["cn_r1_k1_s1_e1_c960"],
]

EFFNET_LITE = [
    ["ds_r1_k3_s1_e1_c16"],
    ["ir_r2_k3_s2_e6_c24"],
    ["ir_r2_k5_s2_e6_c40"],
    ["ir_r3_k3_s2_e6_c80"],
    ["ir_r3_k5_s1_e6_c112"],
    ["ir_r4_k5_s2_e6_c192"],
    ["ir_r1_k3_s1_e6_c320"],
]

EFFNETV2_BASE = [
    ["cn_r1_k3_s1_e1_c16_skip"],
    ["er_r2_k3_s2_e4_c32"],
    ["er_r2_k3_s2_e4_c48"],
    ["ir_r3_k3_s2_e4_c96_se0.25"],
    ["ir_r5_k3_s1_e6_c112_se0.25"],
    ["ir_r8_k3_s2_e6_c192_se0.25"],
]

# tiny efficientnetv2-style net (NOT a timm model): ConvBnAct with skip, fused MBConv (strided + residual), MBConv + SE
ORACLE_TINY_V2 = [
    ["cn_r2_k3_s1_e1_c8_skip"],
    ["er_r2_k3_s2_e2_c12"],
    ["er_r1_k3_s2_e4_c16"],
    ["ir_r2_k3_s2_e4_c24_se0.25"],
    ["ir_r2_k3_s1_e3_c24_se0.25"],
    ["ir_r2_k3_s2_e4_c32_se0.25"],
]

# Tiny MobileNetV4-style net exercising every block flavour (cn k3/k1, uir with dw_start only,
# dw_mid only, both, strided, residual).  NOT a timm model: it exists so that fixtures which
# carry a full state_dict stay a few tens of KB (tests/golden/make_fixtures.py).
ORACLE_TINY = [
    ["cn_r1_k3_s2_e1_c8"],
    ["cn_r1_k3_s2_e1_c12", "cn_r1_k1_s1_e1_c8"],
    ["uir_r1_a5_k5_s2_e3_c16", "uir_r1_a0_k3_s1_e2_c16", "uir_r1_a3_k0_s1_e4_c16"],
    ["uir_r1_a3_k3_s2_e4_c24", "uir_r1_a0_k5_s1_e2_c24"],
    ["cn_r1_k1_s1_e1_c32"],
]

# name -> (arch, channel multiplier, depth multiplier, act, bn eps, tf-same padding, fix first/last depth, stem
#          [, round_limit of the channel rounding: 0.9 default, 0.0 for efficientnetv2_base])
_ZOO = {
    "tf_efficientnetv2_b0":       (EFFNETV2_BASE, 1.0, 1.0, "silu", 1e-3, True, False, 32, 0.0),
    "tf_efficientnetv2_b1":       (EFFNETV2_BASE, 1.0, 1.1, "silu", 1e-3, True, False, 32, 0.0),
    "tf_efficientnetv2_b2":       (EFFNETV2_BASE, 1.1, 1.2, "silu", 1e-3, True, False, 32, 0.0),
    "tf_efficientnetv2_b3":       (EFFNETV2_BASE, 1.2, 1.4, "silu", 1e-3, True, False, 32, 0.0),
    "oracle_tiny_v2":             (ORACLE_TINY_V2, 1.0, 1.0, "silu", 1e-3, True, False, 16, 0.0),
    "mobilenetv4_conv_small":     (MNV4_CONV_SMALL, 1.0, 1.0, "relu", 1e-5, False, False, 32),
    "mobilenetv4_conv_small_050": (MNV4_CONV_SMALL, 0.5, 1.0, "relu", 1e-5, False, False, 32),
    "tf_efficientnet_lite0":      (EFFNET_LITE, 1.0, 1.0, "relu6", 1e-3, True, True, 32),
    "tf_efficientnet_lite1":      (EFFNET_LITE, 1.0, 1.1, "relu6", 1e-3, True, True, 32),
    "tf_efficientnet_lite2":      (EFFNET_LITE, 1.1, 1.2, "relu6", 1e-3, True, True, 32),
    "tf_efficientnet_lite3":      (EFFNET_LITE, 1.2, 1.4, "relu6", 1e-3, True, True, 32),
    "tf_efficientnet_lite4":      (EFFNET_LITE, 1.4, 1.8, "relu6", 1e-3, True, True, 32),
    "oracle_tiny":                (ORACLE_TINY, 1.0, 1.0, "relu", 1e-5, False, False, 16),
    "oracle_tiny_tf":             (ORACLE_TINY, 1.0, 1.0, "relu6", 1e-3, True, False, 16),
}


class FeatureBackbone(nn.Module):
    """features_only network: returns the list of feature maps selected by out_indices."""

    def __init__(self, name: str, out_indices: Optional[Sequence[int]] = None):
        super().__init__()
        arch, cmult, dmult, act, eps, same, fix_fl, stem = _ZOO[name][:8]
        rlim = _ZOO[name][8] if len(_ZOO[name]) > 8 else 0.9
        if arch is EFFNETV2_BASE:
            # timm's EfficientNet rounds the stem with round_chs_fn unless fix_stem is passed; _gen_efficientnetv2_base does
            # not pass it (b0-b2: 32, b3: 40).  _gen_efficientnet_lite passes fix_stem=True, _gen_mobilenet_v4 fixes it below 1.0x
            stem = round_channels(stem, cmult, round_limit=rlim)
        self.conv_stem = _conv(3, stem, 3, 2, same=same)
        self.bn1 = BatchNormAct2d(stem, eps, act)

        stages: List[nn.Sequential] = []
        cin, red = stem, 2
        # timm taps the stem only when the first block strides (MobileNetV4: yes, EfficientNet-Lite: no)
        self._stem_tap = _parse(arch[0][0])["s"] > 1
        info = [dict(num_chs=stem, reduction=2, module="bn1")] if self._stem_tap else []
        taps = []                                   # stage index feeding each later feature
        n_stage = len(arch)
        for si, stage in enumerate(arch):
            blocks = []
            for bi, bstr in enumerate(stage):
                d = _parse(bstr)
                rep = d["r"]
                if dmult != 1.0 and not (fix_fl and si in (0, n_stage - 1)):
                    rep = int(math.ceil(rep * dmult))
                cout = round_channels(d["c"], cmult, round_limit=rlim)
                for r in range(rep):
                    s = d["s"] if r == 0 else 1
                    if d["type"] == "cn":
                        blk = ConvBnAct(cin, cout, d["k"], s, act, eps, same, skip=d["skip"])
                    elif d["type"] == "er":
                        blk = EdgeResidual(cin, cout, d["k"], s, d["e"], act, eps, same)
                    elif d["type"] == "uir":
                        blk = UniversalInvertedResidual(cin, cout, d["a"], d["k"], s, d["e"], act, eps, same)
                    elif d["type"] == "ds":
                        blk = DepthwiseSeparableConv(cin, cout, d["k"], s, act, eps, same)
                    elif d["type"] == "ir":
                        blk = InvertedResidual(cin, cout, d["k"], s, d["e"], act, eps, same, se_ratio=d["se"])
                    else:
                        raise ValueError(bstr)
                    red *= s
                    cin = cout
                    blocks.append(blk)
            stages.append(nn.Sequential(*blocks))
            # a feature is tapped at the end of a stage when the next stage strides, or at the very end
            nxt_stride = _parse(arch[si + 1][0])["s"] if si + 1 < n_stage else 2
            if nxt_stride > 1:
                info.append(dict(num_chs=cin, reduction=red, module=f"blocks.{si}"))
                taps.append(si)
        self.blocks = nn.Sequential(*stages)
        self._taps = taps
        self._all_info = info
        self.out_indices = tuple(out_indices) if out_indices is not None else tuple(range(len(info)))
        self.feature_info = list(info)              # indexable like timm's FeatureInfo

    def forward(self, x):
        feats = []
        x = self.bn1(self.conv_stem(x))
        if self._stem_tap:
            feats.append(x)
        for si, stage in enumerate(self.blocks):
            x = stage(x)
            if si in self._taps:
                feats.append(x)
        return [feats[i] for i in self.out_indices]


def create_model(name: str, features_only: bool = True, pretrained: bool = False,
                 out_indices: Optional[Sequence[int]] = None, **_):
    """Signature-compatible stand-in for ``timm.create_model`` (features_only models only).
    ``pretrained`` is accepted and ignored: there is no network and no weight file."""
    if not features_only:
        raise NotImplementedError("oracle restates features_only backbones only")
    if name not in _ZOO:
        raise ValueError(f"oracle backbone '{name}' not restated")
    return FeatureBackbone(name, out_indices)
