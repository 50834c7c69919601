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


# ============================================================================= hgnetv2 (timm models/hgnet.py)
# /root/reference/configs/models/edge_xl.yaml:4 (`backbone: hgnetv2_b0`).  RECOLLECTION of timm's HighPerfGpuNet:
# StemV2 (3x3 s2 -> [2x2 -> 2x2 | max-pool 2x2 s1] on a bottom/right zero-padded map -> concat -> 3x3 s2 -> 1x1),
# four HighPerfGpuStages (optional depthwise 3x3 s2 downsample without activation; HighPerfGpuBlocks = `layer_num`
# 3x3 ConvBNAct -- or "light" pairs 1x1 (no act) + depthwise k x k -- whose outputs are concatenated with the block
# input and aggregated by two 1x1 ConvBNAct ('se' aggregation: total -> out/2 -> out); blocks after the first of a
# stage are residual).  ConvBNAct = conv (no bias, symmetric padding (k-1)//2) + BatchNorm (eps 1e-5) + ReLU +
# LearnableAffineBlock (two scalars: scale * x + bias; `use_lab=True` for b0..b3).  Module names follow timm's
# FeatureListNet with flatten_sequential=True (`stem.*`, `stages_<i>.*`).  Checksum: timm / PaddleClas publish 6.0 M
# parameters for the classifier (tests/test_oracle_golden.py adds the 1024 -> 2048 -> 1000 head to this trunk).
class LearnableAffineBlock(nn.Module):
    def __init__(self):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor([1.0]))
        self.bias = nn.Parameter(torch.tensor([0.0]))

    def forward(self, x):
        return self.scale * x + self.bias


class HgConvBNAct(nn.Module):
    def __init__(self, cin, cout, k, stride=1, groups=1, use_act=True, use_lab=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, padding=(k - 1) // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.act = nn.ReLU() if use_act else nn.Identity()
        self.lab = LearnableAffineBlock() if (use_act and use_lab) else nn.Identity()

    def forward(self, x):
        return self.lab(self.act(self.bn(self.conv(x))))


class HgLightConvBNAct(nn.Module):
    def __init__(self, cin, cout, k, use_lab=False):
        super().__init__()
        self.conv1 = HgConvBNAct(cin, cout, 1, use_act=False, use_lab=use_lab)
        self.conv2 = HgConvBNAct(cout, cout, k, groups=cout, use_act=True, use_lab=use_lab)

    def forward(self, x):
        return self.conv2(self.conv1(x))


class HgStemV2(nn.Module):
    def __init__(self, cin, mid, cout, use_lab=False):
        super().__init__()
        self.stem1 = HgConvBNAct(cin, mid, 3, 2, use_lab=use_lab)
        self.stem2a = HgConvBNAct(mid, mid // 2, 2, 1, use_lab=use_lab)
        self.stem2b = HgConvBNAct(mid // 2, mid, 2, 1, use_lab=use_lab)
        self.stem3 = HgConvBNAct(mid * 2, mid, 3, 2, use_lab=use_lab)
        self.stem4 = HgConvBNAct(mid, cout, 1, 1, use_lab=use_lab)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=1, ceil_mode=True)

    def forward(self, x):
        x = self.stem1(x)
        x = F.pad(x, (0, 1, 0, 1))
        x2 = self.stem2a(x)
        x2 = F.pad(x2, (0, 1, 0, 1))
        x2 = self.stem2b(x2)
        x1 = self.pool(x)
        x = torch.cat([x1, x2], dim=1)
        return self.stem4(self.stem3(x))


class HgBlock(nn.Module):
    def __init__(self, cin, mid, cout, layer_num, k=3, residual=False, light=False, use_lab=False):
        super().__init__()
        self.residual = residual
        self.layers = nn.ModuleList()
        for i in range(layer_num):
            c = cin if i == 0 else mid
            self.layers.append(HgLightConvBNAct(c, mid, k, use_lab) if light else HgConvBNAct(c, mid, k, 1, use_lab=use_lab))
        total = cin + layer_num * mid
        self.aggregation = nn.Sequential(HgConvBNAct(total, cout // 2, 1, use_lab=use_lab),
                                         HgConvBNAct(cout // 2, cout, 1, use_lab=use_lab))

    def forward(self, x):
        identity = x
        outs = [x]
        for layer in self.layers:
            x = layer(x)
            outs.append(x)
        x = self.aggregation(torch.cat(outs, dim=1))
        return x + identity if self.residual else x


class HgStage(nn.Module):
    def __init__(self, cin, mid, cout, block_num, layer_num, downsample, light, k, use_lab):
        super().__init__()
        self.downsample = (HgConvBNAct(cin, cin, 3, 2, groups=cin, use_act=False, use_lab=use_lab) if downsample
                           else nn.Identity())
        self.blocks = nn.Sequential(*[HgBlock(cin if i == 0 else cout, mid, cout, layer_num, k, residual=i > 0,
                                              light=light, use_lab=use_lab) for i in range(block_num)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


# name -> (stem [mid, out], stages [(in, mid, out, blocks, downsample, light, kernel, layer_num)], use_lab)
_HGNET = {
    "hgnetv2_b0": ((16, 16), [(16, 16, 64, 1, False, False, 3, 3), (64, 32, 256, 1, True, False, 3, 3),
                              (256, 64, 512, 2, True, True, 5, 3), (512, 128, 1024, 1, True, True, 5, 3)], True),
    # tiny test vehicle (NOT a timm model): every block flavour (plain / light, residual second block, no downsample)
    "oracle_tiny_hg": ((8, 8), [(8, 8, 16, 1, False, False, 3, 2), (16, 8, 32, 1, True, False, 3, 3),
                                (32, 8, 48, 2, True, True, 5, 2), (48, 16, 64, 1, True, True, 5, 3)], True),
}


class HgFeatureBackbone(nn.Module):
    def __init__(self, name, out_indices=None):
        super().__init__()
        (smid, sout), stages, use_lab = _HGNET[name]
        self.stem = HgStemV2(3, smid, sout, use_lab)
        red, info = 4, []
        for i, (cin, mid, cout, nb, ds, light, k, ln) in enumerate(stages):
            setattr(self, f"stages_{i}", HgStage(cin, mid, cout, nb, ln, ds, light, k, use_lab))
            red *= 2 if ds else 1
            info.append(dict(num_chs=cout, reduction=red, module=f"stages.{i}"))
        self._n = len(stages)
        self.out_indices = tuple(out_indices) if out_indices is not None else tuple(range(len(info)))
        self.feature_info = info

    def forward(self, x):
        x = self.stem(x)
        feats = []
        for i in range(self._n):
            x = getattr(self, f"stages_{i}")(x)
            feats.append(x)
        return [feats[i] for i in self.out_indices]


# ============================================================================= convnextv2 (timm models/convnext.py)
# /root/reference/configs/v2_models/yololite_l.yaml:4 (`backbone: convnextv2_tiny`).  RECOLLECTION of timm's ConvNeXt with
# use_grn=True, ls_init_value=None, conv_mlp=False: stem = conv 4x4 s4 (bias) + LayerNorm2d; stage i>0 starts with
# LayerNorm2d + conv 2x2 s2 (bias); block = depthwise 7x7 pad 3 (bias) -> LayerNorm over C (eps 1e-6) -> Linear C->4C ->
# GELU (erf) -> GlobalResponseNorm -> Linear 4C->C -> + shortcut.  GRN (channels last): g = ||x||_2 over (H, W) per image and
# channel, n = g / (mean_C(g) + 1e-6), y = x + (bias + weight * (x * n)).  Module names follow FeatureListNet with
# flatten_sequential=True (`stem_0`, `stem_1`, `stages_<i>.*`).  Checksum: timm publishes 28.64 M parameters for the
# classifier (tests/test_oracle_golden.py adds the LayerNorm + 768 -> 1000 head to this trunk).
class LayerNorm2d(nn.LayerNorm):
    def __init__(self, ch, eps=1e-6):
        super().__init__(ch, eps=eps)

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        x = F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2)


class GlobalResponseNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.zeros(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):                                  # channels last [B, H, W, C]
        x_g = x.norm(p=2, dim=(1, 2), keepdim=True)
        x_n = x_g / (x_g.mean(dim=-1, keepdim=True) + self.eps)
        return x + torch.addcmul(self.bias.view(1, 1, 1, -1), self.weight.view(1, 1, 1, -1), x * x_n)


class GrnMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.grn = GlobalResponseNorm(hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.grn(self.act(self.fc1(x))))


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim, k=7):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, k, 1, padding=k // 2, groups=dim, bias=True)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = GrnMlp(dim, 4 * dim)

    def forward(self, x):
        y = self.conv_dw(x).permute(0, 2, 3, 1)
        y = self.mlp(self.norm(y)).permute(0, 3, 1, 2)
        return y + x


class ConvNeXtStage(nn.Module):
    def __init__(self, cin, cout, stride, depth, k=7):
        super().__init__()
        if cin != cout or stride > 1:
            self.downsample = nn.Sequential(LayerNorm2d(cin), nn.Conv2d(cin, cout, stride, stride, bias=True))
        else:
            self.downsample = nn.Identity()
        self.blocks = nn.Sequential(*[ConvNeXtBlock(cout, k) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


# name -> (depths, dims, depthwise kernel)
_CONVNEXT = {
    "convnextv2_tiny": ((3, 3, 9, 3), (96, 192, 384, 768), 7),
    "convnextv2_nano": ((2, 2, 8, 2), (80, 160, 320, 640), 7),
    "convnextv2_pico": ((2, 2, 6, 2), (64, 128, 256, 512), 7),
    "convnextv2_femto": ((2, 2, 6, 2), (48, 96, 192, 384), 7),
    "convnextv2_atto": ((2, 2, 6, 2), (40, 80, 160, 320), 7),
    "oracle_tiny_cnx": ((1, 2, 2, 1), (8, 16, 24, 32), 7),          # tiny test vehicle (NOT a timm model)
}


class ConvNeXtFeatureBackbone(nn.Module):
    def __init__(self, name, out_indices=None):
        super().__init__()
        depths, dims, k = _CONVNEXT[name]
        self.stem_0 = nn.Conv2d(3, dims[0], 4, 4, bias=True)
        self.stem_1 = LayerNorm2d(dims[0])
        red, prev, info = 4, dims[0], []
        for i, (d, c) in enumerate(zip(depths, dims)):
            s = 2 if i > 0 else 1
            setattr(self, f"stages_{i}", ConvNeXtStage(prev, c, s, d, k))
            red *= s
            prev = c
            info.append(dict(num_chs=c, reduction=red, module=f"stages.{i}"))
        self._n = len(depths)
        self.out_indices = tuple(out_indices) if out_indices is not None else tuple(range(len(info)))
        self.feature_info = info

    def forward(self, x):
        x = self.stem_1(self.stem_0(x))
        feats = []
        for i in range(self._n):
            x = getattr(self, f"stages_{i}")(x)
            feats.append(x)
        return [feats[i] for i in self.out_indices]


def create_model(name: str, features_only: bool = True, pretrained: bool = False,
                 out_indices: Optional[Sequence[int]] = None, **_):
    """Signature-compatible stand-in for ``timm.create_model`` (features_only models only).
    ``pretrained`` is accepted and ignored: there is no network and no weight file."""
    if not features_only:
        raise NotImplementedError("oracle restates features_only backbones only")
    if name in _HGNET:
        return HgFeatureBackbone(name, out_indices)
    if name in _CONVNEXT:
        return ConvNeXtFeatureBackbone(name, out_indices)
    if name not in _ZOO:
        raise ValueError(f"oracle backbone '{name}' not restated")
    return FeatureBackbone(name, out_indices)
