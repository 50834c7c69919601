"""ORACLE (test infrastructure only) -- CPU restatement of the reference detector:
backbone -> FPN laterals -> nearest-upsample-add -> smooth blocks -> decoupled heads -> layout.

Follows /root/reference/scripts/model/model_v2.py:
  * conv_block            :15-22   (conv3x3 no-bias -> BN -> SiLU) x n
  * DWConvBlock           :23-39   (dw3x3 no-bias -> pw1x1 no-bias -> BN -> ReLU) x n, nothing between dw and pw
  * make_head             :42-53   trunk = head_depth x DWConvBlock; out.box/out.obj/out.cls 1x1 convs with bias
  * init_detect_bias      :7-14
  * _flatten_level_outputs:57-64
  * _pick_out_indices     :69-74
  * YOLOLiteMS            :77-247  (dense smooth blocks, SiLU)
  * YOLOLiteMS_CPU        :250-383 (depthwise smooth blocks, ReLU)
State-dict keys equal the reference's so a reference checkpoint loads directly.
Pinned against the reference classes imported under stubs (tests/golden/make_fixtures.py).
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import backbones


def dense_smooth(ch: int, n: int) -> nn.Sequential:
    """model_v2.py:15-22 -- plain Sequential, indices 3i (conv), 3i+1 (bn), 3i+2 (silu)."""
    mods: List[nn.Module] = []
    for _ in range(n):
        mods += [nn.Conv2d(ch, ch, 3, padding=1, bias=False), nn.BatchNorm2d(ch), nn.SiLU()]
    return nn.Sequential(*mods)


class DWSmooth(nn.Module):
    """model_v2.py:23-39 -- `.block` Sequential, indices 4i (dw), 4i+1 (pw), 4i+2 (bn), 4i+3 (relu)."""

    def __init__(self, ch: int, n: int = 1):
        super().__init__()
        mods: List[nn.Module] = []
        for _ in range(n):
            mods += [nn.Conv2d(ch, ch, 3, padding=1, groups=ch, bias=False),
                     nn.Conv2d(ch, ch, 1, bias=False),
                     nn.BatchNorm2d(ch), nn.ReLU()]
        self.block = nn.Sequential(*mods)

    def forward(self, x):
        return self.block(x)


class ProtoNet(nn.Module):
    """BUILD-DEFINED mask prototype head (the reference has no mask code; SURVEY 8a row a15, parity unpinned):
    conv3x3(F->Cp)+BN+SiLU -> nearest x2 -> conv3x3(Cp->Cp)+BN+SiLU -> conv1x1(Cp->NM)+BN+SiLU on P3."""

    def __init__(self, F_: int, Cp: int, NM: int):
        super().__init__()
        def cba(ci, co, k):
            return nn.Sequential(nn.Conv2d(ci, co, k, padding=k // 2, bias=False), nn.BatchNorm2d(co), nn.SiLU())
        self.cv1, self.cv2, self.cv3 = cba(F_, Cp, 3), cba(Cp, Cp, 3), cba(Cp, NM, 1)

    def forward(self, p3):
        y = self.cv1(p3)
        y = F.interpolate(y, scale_factor=2, mode="nearest")
        return self.cv3(self.cv2(y))


def build_head(A: int, head_depth: int, C: int, F_: int, NM: int = 0) -> nn.ModuleDict:
    """model_v2.py:42-53 + bias init :7-14 (+ build-defined mask-coefficient conv `mc` when NM > 0)."""
    trunk = nn.Sequential(*[DWSmooth(F_, 1) for _ in range(head_depth)])
    out = nn.ModuleDict(dict(box=nn.Conv2d(F_, 4 * A, 1), obj=nn.Conv2d(F_, A, 1), cls=nn.Conv2d(F_, A * C, 1)))
    if NM > 0:
        out["mc"] = nn.Conv2d(F_, A * NM, 1)
    with torch.no_grad():
        out["obj"].bias.fill_(-math.log((1 - 0.01) / 0.01))
        out["cls"].bias.fill_(-math.log(C) if C > 1 else 0.0)
        out["box"].bias.zero_()
    return nn.ModuleDict(dict(trunk=trunk, out=out))


class DetectorOracle(nn.Module):
    """One class for both reference architectures; ``arch`` selects the smooth-block flavour."""

    def __init__(self, arch: str = "YOLOLiteMS_CPU", backbone: str = "mobilenetv4_conv_small_050",
                 num_classes: int = 3, fpn_channels: int = 96,
                 num_anchors_per_level: Sequence[int] = (1, 1, 1),
                 depth_multiple: float = 1.0, width_multiple: float = 1.0, head_depth: int = 1,
                 use_p6: bool = False, use_p2: bool = False, backbone_module: nn.Module = None,
                 seg: bool = False, num_masks: int = 32, proto_channels: int = 64):
        super().__init__()
        arch_l = arch.lower()
        if arch_l not in ("yololitems", "yololitems_cpu"):
            raise ValueError(f"unknown arch {arch}")
        dw = arch_l == "yololitems_cpu"
        take = 4 if use_p2 else 3
        probe = backbone_module if backbone_module is not None else backbones.create_model(backbone)
        n = len(probe.feature_info)
        idx = list(range(n - take, n))                                  # :69-74
        self.reductions = [probe.feature_info[i]["reduction"] for i in idx]
        chs = [probe.feature_info[i]["num_chs"] for i in idx]
        self.backbone = backbone_module if backbone_module is not None else \
            backbones.create_model(backbone, out_indices=idx)
        self.use_p6, self.use_p2 = bool(use_p6), bool(use_p2)
        Fc = int(fpn_channels * width_multiple)                         # :277 / :106
        d = max(1, round(2 * depth_multiple))                           # :278 / :107
        self.fpn_channels, self.smooth_depth = Fc, d
        mk = (lambda: DWSmooth(Fc, d)) if dw else (lambda: dense_smooth(Fc, d))

        self.pyr = (["p2"] if use_p2 else []) + ["p3", "p4", "p5"]      # levels fed by the backbone
        for name, c in zip(self.pyr, chs):
            k = name[1]
            setattr(self, f"lateral{k}", nn.Conv2d(c, Fc, 1))           # bias=True, no BN, no act
            setattr(self, f"smooth{k}", mk())
        # P6 parameters always exist in the reference state_dict (:130-133 / :297-300)
        self.p6_down = nn.Conv2d(Fc, Fc, 3, 2, 1, bias=False)
        self.p6_bn = nn.BatchNorm2d(Fc)
        self.p6_act = nn.ReLU() if dw else nn.SiLU()
        self.smooth6 = mk()

        self.levels = self.pyr + (["p6"] if use_p6 else [])
        if len(num_anchors_per_level) >= 3:
            a3, a4, a5 = (int(v) for v in num_anchors_per_level[:3])
            amap = dict(p2=a3, p3=a3, p4=a4, p5=a5, p6=a5)
        else:
            a = int(num_anchors_per_level[0]) if len(num_anchors_per_level) else 1
            amap = dict(p2=a, p3=a, p4=a, p5=a, p6=a)
        self.num_anchors_per_level = tuple(amap[l] for l in self.levels)
        self.num_classes = int(num_classes)
        self.export_concat = False
        self.num_masks = int(num_masks) if seg else 0
        for l in self.levels:
            setattr(self, f"head{l[1]}", build_head(amap[l], head_depth, self.num_classes, Fc, self.num_masks))
        if seg:
            self.proto = ProtoNet(Fc, int(proto_channels), self.num_masks)
        self.fpn_strides = list(self.reductions) + ([self.reductions[-1] * 2] if use_p6 else [])

    # -- pieces ---------------------------------------------------------------------------
    def _head(self, p, hd, A):                                          # :340-350
        t = hd["trunk"](p)
        B, _, S, _ = t.shape
        parts = [hd["out"]["box"](t).view(B, A, 4, S, S),
                 hd["out"]["obj"](t).view(B, A, 1, S, S),
                 hd["out"]["cls"](t).view(B, A, self.num_classes, S, S)]
        if self.num_masks:
            parts.append(hd["out"]["mc"](t).view(B, A, self.num_masks, S, S))
        return torch.cat(parts, dim=2).permute(0, 1, 3, 4, 2).contiguous()

    def neck(self, feats: List[torch.Tensor]) -> Dict[str, torch.Tensor]:
        """top-down pass, P5 first (:359-361)."""
        c = dict(zip(self.pyr, feats))
        p = {}
        p["p5"] = self.smooth5(self.lateral5(c["p5"]))
        prev = "p5"
        for name in reversed(self.pyr[:-1]):
            k = name[1]
            lat = getattr(self, f"lateral{k}")(c[name])
            up = F.interpolate(p[prev], size=lat.shape[-2:], mode="nearest")
            p[name] = getattr(self, f"smooth{k}")(up + lat)
            prev = name
        if self.use_p6:
            p["p6"] = self.smooth6(self.p6_act(self.p6_bn(self.p6_down(p["p5"]))))
        return p

    def forward(self, x):
        feats = self.backbone(x)
        p = self.neck(feats)
        outs = [self._head(p[l], getattr(self, f"head{l[1]}"), A)
                for l, A in zip(self.levels, self.num_anchors_per_level)]
        if self.num_masks:                                              # build-defined: (levels, prototypes [B,NM,PH,PW])
            return outs, self.proto(p["p3"])
        if self.export_concat:                                          # :57-64
            return torch.cat([o.reshape(o.shape[0], -1, o.shape[-1]) for o in outs], dim=1)
        return outs

    def get_strides(self):
        return list(self.fpn_strides)

    def get_num_anchors_per_level(self):
        return tuple(self.num_anchors_per_level)


def build_from_meta(meta: dict) -> DetectorOracle:
    """/root/reference/tools/infer.py:34-77 (same keys, same defaults, same errors)."""
    cfg = meta.get("config", {}) or {}
    mcfg = cfg.get("model", {}) or {}
    arch = (meta.get("arch") or mcfg.get("arch") or "YOLOLiteMS")
    return DetectorOracle(
        arch=arch,
        backbone=(meta.get("backbone") or mcfg.get("backbone") or "resnet18"),
        num_classes=int(meta.get("num_classes") or mcfg.get("num_classes") or 80),
        fpn_channels=int(mcfg.get("fpn_channels", 128)),
        num_anchors_per_level=tuple(meta.get("num_anchors_per_level") or (1, 1, 1)),
        depth_multiple=float(mcfg.get("depth_multiple", 1.0)),
        width_multiple=float(mcfg.get("width_multiple", 1.0)),
        head_depth=int(mcfg.get("head_depth", 1)),
        use_p6=cfg["training"]["use_p6"], use_p2=cfg["training"]["use_p2"],
        seg=bool(mcfg.get("seg", False)), num_masks=int(mcfg.get("num_masks", 32)),
        proto_channels=int(mcfg.get("proto_channels", 64)))


# ------------------------------------------------------------------ synthetic weights
@torch.no_grad()
def randomize_(model: nn.Module, seed: int = 0, head_noise: float = 0.5) -> nn.Module:
    """Seeded synthetic weights: kaiming-normal convs, randomised BN statistics, detection-bias
    init plus N(0, head_noise) on the head output convs so scores straddle the thresholds."""
    g = torch.Generator().manual_seed(seed)
    for name, m in model.named_modules():
        if isinstance(m, nn.Conv2d):
            fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1] // m.groups
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * math.sqrt(2.0 / max(fan_out, 1)))
            if m.bias is not None and ".out." not in name:
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    for name, m in model.named_modules():
        if isinstance(m, nn.Conv2d) and ".out." in name:
            m.weight.add_(torch.randn(m.weight.shape, generator=g) * head_noise * 0.2)
            m.bias.add_(torch.randn(m.bias.shape, generator=g) * head_noise)
    return model


MODEL_ZOO = {
    # /root/reference/configs/models/{edge_n,edge_m,yololite_m}.yaml
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
    # /root/reference/configs/v2_models/yololite_{n,s,m}.yaml (tf_efficientnetv2_b0 / b1 / b2)
    "yololite_n_v2": dict(arch="YOLOLiteMS", backbone="tf_efficientnetv2_b0", depth_multiple=1.0,
                          width_multiple=1.0, fpn_channels=196, head_depth=1),
    "yololite_s_v2": dict(arch="YOLOLiteMS", backbone="tf_efficientnetv2_b1", depth_multiple=1.0,
                          width_multiple=1.0, fpn_channels=256, head_depth=2),
    "yololite_m_v2": dict(arch="YOLOLiteMS", backbone="tf_efficientnetv2_b2", depth_multiple=1.0,
                          width_multiple=1.0, fpn_channels=328, head_depth=2),
    # /root/reference/configs/models/edge_xl.yaml (hgnetv2_b0), configs/v2_models/yololite_l.yaml (convnextv2_tiny)
    "edge_xl": dict(arch="YOLOLiteMS_CPU", backbone="hgnetv2_b0", depth_multiple=1.0, width_multiple=1.0,
                    fpn_channels=256, head_depth=3),
    "yololite_l_v2": dict(arch="YOLOLiteMS", backbone="convnextv2_tiny", depth_multiple=1.0, width_multiple=1.0,
                          fpn_channels=512, head_depth=3),
}


def make_meta(model_name: str, num_classes: int = 80, img_size: int = 640, use_p6=False, use_p2=False) -> dict:
    """A ``meta`` dict shaped like the one tools/train.py:62-75 stores in checkpoints."""
    m = dict(MODEL_ZOO[model_name])
    return dict(metric_key="map50", metric_value=-1.0, names=None, num_classes=num_classes,
                img_size=img_size, arch=m["arch"], backbone=m["backbone"],
                num_anchors_per_level=(1, 1, 1) + ((1,) if use_p6 else ()) + ((1,) if use_p2 else ()),
                config=dict(model=dict(m, num_classes=num_classes),
                            training=dict(img_size=img_size, use_p6=use_p6, use_p2=use_p2)))
