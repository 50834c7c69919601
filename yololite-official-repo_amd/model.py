"""HIP-backed detector with the reference's model interface.

Mirrors (same names, argument meaning and error behaviour):
  * build_model_from_meta(meta)                       /root/reference/tools/infer.py:34-77
  * load_model_names_imgsize_from_ckpt(weights, dev)  /root/reference/tools/infer.py:80-102
  * model(x) -> list of [B,A,S,S,5+C]; export_concat; get_strides(); get_num_anchors_per_level()
                                                      /root/reference/scripts/model/model_v2.py:352-383
All device arithmetic runs in libyololite_hip.so (hand-written gfx950 kernels); torch tensors are
used only as owners of device memory.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .program import Program, build_program


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class HipContext:
    """Owns one yl_ctx.  Post-processing-only contexts are created with no layers."""

    def __init__(self, img_size: int, num_classes: int, level_size: Sequence[int], level_anchors: Sequence[int],
                 program: Optional[Program] = None, device: int = 0, num_masks: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.YoloLiteHipError("no HIP device visible (torch.cuda.is_available() is False); "
                                        "this package has no CPU execution path")
        self.device = torch.device("cuda", device)
        self.img_size, self.C = int(img_size), int(num_classes)
        self.level_size, self.level_anchors = [int(s) for s in level_size], [int(a) for a in level_anchors]
        self.L = len(self.level_size)
        self.N = sum(a * s * s for a, s in zip(self.level_anchors, self.level_size))
        d = _lib.yl_model_desc()
        self.NM = int(program.num_masks) if program is not None else int(num_masks)
        self.proto_slot = int(program.proto_slot) if program is not None else -1
        self.E = 5 + self.C + self.NM
        d.abi_version, d.img_size, d.in_channels = _lib.YL_ABI_VERSION, self.img_size, 3
        d.num_classes, d.num_levels = self.C, self.L
        d.num_masks, d.proto_slot = self.NM, self.proto_slot
        for i in range(self.L):
            d.level_size[i], d.level_anchors[i] = self.level_size[i], self.level_anchors[i]
        keep = []                                        # host arrays must outlive yl_create
        if program is not None and program.layers:
            sh = np.ascontiguousarray([s[0] for s in program.slots], np.int32)
            sw = np.ascontiguousarray([s[1] for s in program.slots], np.int32)
            sc = np.ascontiguousarray([s[2] for s in program.slots], np.int32)
            keep += [sh, sw, sc]
            d.num_slots = len(program.slots)
            d.slot_h = sh.ctypes.data_as(_lib._ip)
            d.slot_w = sw.ctypes.data_as(_lib._ip)
            d.slot_c = sc.ctypes.data_as(_lib._ip)
            arr = (_lib.yl_layer * len(program.layers))()

            def fp(a):
                if a is None:
                    return None
                a = np.ascontiguousarray(a, np.float32)
                keep.append(a)
                return a.ctypes.data_as(_lib._fp)

            for i, l in enumerate(program.layers):
                y = arr[i]
                y.op, y.in_slot, y.out_slot, y.res_slot, y.up_slot = l.op, l.in_slot, l.out_slot, l.res_slot, l.up_slot
                y.head_level, y.cin, y.cout = l.head_level, l.cin, l.cout
                y.k, y.stride, y.pad_t, y.pad_l, y.act, y.in_shift = l.k, l.stride, l.pad_t, l.pad_l, l.act, l.in_shift
                y.dw_k, y.dw_stride, y.dw_pad_t, y.dw_pad_l, y.dw_act = l.dw_k, l.dw_stride, l.dw_pad_t, l.dw_pad_l, l.dw_act
                y.w, y.b, y.dw_w, y.dw_b = fp(l.w), fp(l.b), fp(l.dw_w), fp(l.dw_b)
                y.c2, y.act2, y.c3, y.act3 = l.c2, l.act2, l.c3, l.act3
                y.w2, y.b2, y.w3, y.b3 = fp(l.w2), fp(l.b2), fp(l.w3), fp(l.b3)
                y.scale_slot, y.reserved0 = l.scale_slot, 0
                y.lab_scale, y.lab_bias, y.eps, y.out_ch_off = l.lab_scale, l.lab_bias, l.eps, l.out_ch_off
            d.num_layers = len(program.layers)
            d.layers = arr
            keep.append(arr)
        self.num_layers = int(d.num_layers)
        self.proto_shape = tuple(program.slots[self.proto_slot]) if self.proto_slot >= 0 else None
        h = C.c_void_p()
        st = self.lib.yl_create(C.byref(d), device, C.byref(h))
        self.handle = h
        if st != _lib.YL_OK:
            msg = self.lib.yl_last_error(h).decode() if h else ""
            if h:
                self.lib.yl_destroy(h)
            self.handle = None
            raise _lib.YoloLiteHipError(f"yl_create: {self.lib.yl_strerror(st).decode()} ({st}) {msg}")

    def __del__(self):
        h = getattr(self, "handle", None)
        if h:
            try:
                self.lib.yl_destroy(h)
            except Exception:
                pass
            self.handle = None

    # ---- helpers
    def _ptr_array(self, tensors: Sequence[torch.Tensor]):
        arr = (C.c_void_p * self.L)()
        for i, t in enumerate(tensors):
            arr[i] = t.data_ptr()
        return arr

    def _check_levels(self, levels: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        if len(levels) != self.L:
            raise ValueError(f"expected {self.L} levels, got {len(levels)}")
        out = []
        B = levels[0].shape[0]
        for t, s, a in zip(levels, self.level_size, self.level_anchors):
            if t.dim() == 4:
                t = t.unsqueeze(1)
            if tuple(t.shape) != (B, a, s, s, self.E):
                raise ValueError(f"level shape {tuple(t.shape)} != {(B, a, s, s, self.E)}")
            if t.dtype != torch.float32 or t.device != self.device:
                t = t.to(device=self.device, dtype=torch.float32)
            out.append(t.contiguous())
        return out

    def clone(self) -> "HipContext":
        """yl_clone: another context of the same model on the same device -- shares the packed weights, copies the current
        options, owns its arenas / workspaces / streams / graphs.  One context per batch in flight (serving.ServingPipeline)."""
        o = object.__new__(HipContext)
        for k in ("lib", "device", "img_size", "C", "level_size", "level_anchors", "L", "N", "NM", "proto_slot", "E",
                  "num_layers", "proto_shape"):
            setattr(o, k, getattr(self, k))
        h = C.c_void_p()
        st = self.lib.yl_clone(self.handle, C.byref(h))
        o.handle = h
        if st != _lib.YL_OK:
            if h:
                self.lib.yl_destroy(h)
            o.handle = None
            raise _lib.YoloLiteHipError(f"yl_clone: {self.lib.yl_strerror(st).decode()} ({st})")
        return o

    def set_option(self, name: str, value: int):
        """yl_set_option.  A value that is already set is not sent again: the library drops its cached hipGraphs on
        every option write."""
        if self.get_option(name) == int(value):
            return
        _lib.check(self.lib.yl_set_option(self.handle, name.encode(), int(value)), self.handle, "yl_set_option")

    def get_option(self, name: str, default: Optional[int] = None) -> int:
        """yl_get_option: the library's CURRENT value (its default when never written, clamped to the option's range).
        `default` is returned for a name the library does not know (else that raises)."""
        v = C.c_int32()
        st = self.lib.yl_get_option(self.handle, name.encode(), C.byref(v))
        if st != _lib.YL_OK:
            if default is not None:
                return default
            _lib.check(st, self.handle, f"yl_get_option({name})")
        return int(v.value)

    # ---- forward
    def forward(self, x: torch.Tensor, timed: bool = False):
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.img_size or x.shape[3] != self.img_size:
            raise ValueError(f"input must be [B,3,{self.img_size},{self.img_size}], got {tuple(x.shape)}")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        outs = [torch.empty((B, a, s, s, self.E), device=self.device, dtype=torch.float32)
                for s, a in zip(self.level_size, self.level_anchors)]
        arr = self._ptr_array(outs)
        sp = _stream_ptr(self.device)
        if timed:
            ms = (C.c_float * self.num_layers)()
            _lib.check(self.lib.yl_forward_timed(self.handle, x.data_ptr(), B, arr, sp, ms), self.handle, "yl_forward_timed")
            return outs, list(ms)
        _lib.check(self.lib.yl_forward(self.handle, x.data_ptr(), B, arr, sp), self.handle, "yl_forward")
        return outs

    def last_timing(self):
        """(infer_ms, post_ms) of the last predict() under option "time_split" (HIP events on the launch stream)."""
        a, b = C.c_float(), C.c_float()
        _lib.check(self.lib.yl_last_timing(self.handle, C.byref(a), C.byref(b)), self.handle, "yl_last_timing")
        return float(a.value), float(b.value)

    def activation_bytes(self) -> int:
        return int(self.lib.yl_activation_bytes(self.handle))

    def read_slot(self, slot: int, B: int, shape) -> torch.Tensor:
        h, w, c = shape
        t = torch.empty((B, h, w, c), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.yl_read_slot(self.handle, slot, B, t.data_ptr(), _stream_ptr(self.device)), self.handle,
                   "yl_read_slot")
        return t

    # ---- decode / post-processing
    def decode(self, levels, center_mode="v8", wh_mode="softplus"):
        lv = self._check_levels(levels)
        B = lv[0].shape[0]
        box = torch.empty((B, self.N, 4), device=self.device, dtype=torch.float32)
        obj = torch.empty((B, self.N, 1), device=self.device, dtype=torch.float32)
        cls = torch.empty((B, self.N, self.C), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.yl_decode(self.handle, self._ptr_array(lv), B, _lib.CENTER[center_mode], _lib.WH[wh_mode],
                                      box.data_ptr(), obj.data_ptr(), cls.data_ptr() if self.C else None,
                                      _stream_ptr(self.device)), self.handle, "yl_decode")
        return {"box": box, "obj": obj, "cls": cls}

    def forward_decoded(self, x: torch.Tensor, center_mode="v8", wh_mode="softplus"):
        """yl_forward_decoded: network input -> the decoded triple {"box" [B,N,4], "obj" [B,N,1], "cls" [B,N,C]} in one
        call (the reference's exported wire format, export/export_onnx.py:283-296)."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        B = x.shape[0]
        box = torch.empty((B, self.N, 4), device=self.device, dtype=torch.float32)
        obj = torch.empty((B, self.N, 1), device=self.device, dtype=torch.float32)
        cls = torch.empty((B, self.N, self.C), device=self.device, dtype=torch.float32)
        _lib.check(self.lib.yl_forward_decoded(self.handle, x.data_ptr(), B, _lib.CENTER[center_mode], _lib.WH[wh_mode],
                                               box.data_ptr(), obj.data_ptr(), cls.data_ptr() if self.C else None,
                                               _stream_ptr(self.device)), self.handle, "yl_forward_decoded")
        return {"box": box, "obj": obj, "cls": cls}

    def make_cfg(self, mode, conf, iou, per_class_cap, topk, max_out, center_mode="v8", wh_mode="softplus",
                 backmap: Optional[torch.Tensor] = None, fallback_nms: int = _lib.NMS_TORCHVISION):
        cfg = _lib.yl_post_cfg()
        cfg.mode, cfg.conf_thr, cfg.iou_thr = mode, float(conf), float(iou)
        cfg.per_class_cap, cfg.topk, cfg.max_out = int(per_class_cap or 0), int(topk or 0), int(max_out)
        cfg.center_mode, cfg.wh_mode = _lib.CENTER[center_mode], _lib.WH[wh_mode]
        cfg.backmap_dev = backmap.data_ptr() if backmap is not None else None
        cfg.fallback_nms = int(fallback_nms)
        return cfg

    def default_max_out(self, mode, per_class_cap, topk):
        if mode == _lib.POST_FALLBACK and topk and topk > 0:
            return min(self.N, max(int(topk), 1))
        if per_class_cap and per_class_cap > 0:
            return min(self.N, max(1, self.C) * int(per_class_cap))
        return self.N

    def postprocess(self, levels, mode, conf, iou, per_class_cap=300, topk=0, max_out=None, center_mode="v8",
                    wh_mode="softplus", backmap=None, want_idx=False, fallback_nms: int = _lib.NMS_TORCHVISION):
        """Returns dets [B,max_out,6] (x1,y1,x2,y2,score,class), counts [B] (int32, device) and
        optionally the candidate index of every detection."""
        lv = self._check_levels(levels)
        B = lv[0].shape[0]
        if max_out is None:
            max_out = self.default_max_out(mode, per_class_cap, topk)
        if backmap is not None:
            backmap = backmap.to(device=self.device, dtype=torch.float32).contiguous()
        cfg = self.make_cfg(mode, conf, iou, per_class_cap, topk, max_out, center_mode, wh_mode, backmap, fallback_nms)
        dets = torch.empty((B, max_out, 6), device=self.device, dtype=torch.float32)
        counts = torch.empty((B,), device=self.device, dtype=torch.int32)
        idx = torch.empty((B, max_out), device=self.device, dtype=torch.int32) if want_idx else None
        _lib.check(self.lib.yl_postprocess(self.handle, self._ptr_array(lv), B, C.byref(cfg), dets.data_ptr(),
                                           counts.data_ptr(), idx.data_ptr() if want_idx else None,
                                           _stream_ptr(self.device)), self.handle, "yl_postprocess")
        return (dets, counts, idx) if want_idx else (dets, counts)

    def prototypes(self, B: int) -> torch.Tensor:
        """mask prototypes of the last forward/predict as [B,NM,PH,PW] (the device tensor is NHWC)."""
        if self.proto_slot < 0:
            raise _lib.YoloLiteHipError("model has no mask branch")
        return self.read_slot(self.proto_slot, B, self.proto_shape).permute(0, 3, 1, 2)

    def masks(self, counts: torch.Tensor, keep_idx: torch.Tensor, max_out: int, thr: float = 0.5,
              levels: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        """uint8 [B,max_out,PH,PW] instance masks of the detections of the last predict()/postprocess()
        (build-defined semantics, see include/yololite_hip.h: yl_masks)."""
        B = counts.shape[0]
        ph, pw, _ = self.proto_shape
        out = torch.zeros((B, max_out, ph, pw), device=self.device, dtype=torch.uint8)
        arr = None
        if levels is not None:
            arr = self._ptr_array(self._check_levels(levels))
        _lib.check(self.lib.yl_masks(self.handle, arr, B, counts.data_ptr(), keep_idx.data_ptr(), int(max_out),
                                     float(thr), out.data_ptr(), _stream_ptr(self.device)), self.handle, "yl_masks")
        return out

    def masks_image(self, dets: torch.Tensor, counts: torch.Tensor, keep_idx: torch.Tensor, backmap=None, out_hw=None,
                    thr: float = 0.5, packed: bool = False, levels: Optional[Sequence[torch.Tensor]] = None,
                    arena: Optional[torch.Tensor] = None):
        """Image-resolution instance masks of the detections of the last predict() (yl_masks_image, build-defined).
        backmap: the [B,5] rows given to predict() (then `dets` are already in original-image coordinates and
        out_hw[b] = (h0, w0)); None: masks on the S x S network-input grid.  Returns a list of B device tensors:
        uint8 [Ni, h_b, w_b], or with packed=True uint32 (as an int32 view: torch has no uint32 arithmetic)
        [Ni, h_b, ceil(w_b/32)] (bit k of word j = pixel 32j+k).

        arena: a preallocated uint8 device tensor of at least B*max_out*h*row bytes (all images one output size) makes
        the call ASYNCHRONOUS -- fixed-capacity layout, image b's masks start at b*max_out*h*row, no host read of
        `counts`, no allocation; returns ONE view [B, max_out, h, row] whose first counts[b] entries of image b are
        written (serving loops / the benchmark: the masks travel with the packed detections)."""
        B, max_out = int(dets.shape[0]), int(dets.shape[1])
        if out_hw is None:
            if backmap is not None:
                bmh = np.asarray(backmap.cpu() if torch.is_tensor(backmap) else backmap, dtype=np.float64).reshape(B, 5)
                out_hw = np.stack([bmh[:, 4], bmh[:, 3]], 1)
            else:
                out_hw = np.full((B, 2), self.img_size)
        hw = np.ascontiguousarray(np.asarray(out_hw).reshape(B, 2), np.int32)
        rowb = (((hw[:, 1].astype(np.int64) + 31) // 32) * 4) if packed else hw[:, 1].astype(np.int64)
        bm_d = None
        if backmap is not None:
            bm_d = (backmap if torch.is_tensor(backmap) else torch.as_tensor(np.asarray(backmap, np.float32)))
            bm_d = bm_d.to(device=self.device, dtype=torch.float32).contiguous()
        arr = self._ptr_array(self._check_levels(levels)) if levels is not None else None
        if arena is not None:
            if not (hw == hw[0]).all():
                raise ValueError("masks_image(arena=...) needs one output size for the whole batch")
            h, row = int(hw[0, 0]), int(rowb[0])
            per = max_out * h * row
            if arena.dtype != torch.uint8 or not arena.is_contiguous() or arena.numel() < B * per or arena.data_ptr() % 16:
                raise ValueError(f"arena must be a contiguous 16-byte aligned uint8 tensor of >= {B * per} bytes")
            key = (B, max_out, h, int(hw[0, 1]), bool(packed))
            cached = getattr(self, "_mask_geom", None)
            if cached is None or cached[0] != key:
                offs = (np.arange(B, dtype=np.int64) * per)
                cached = (key, torch.from_numpy(hw).to(self.device), torch.from_numpy(offs).to(self.device))
                self._mask_geom = cached
            _lib.check(self.lib.yl_masks_image(self.handle, arr, B, dets.data_ptr(), counts.data_ptr(), keep_idx.data_ptr(),
                                               max_out, float(thr), bm_d.data_ptr() if bm_d is not None else None,
                                               cached[1].data_ptr(), cached[2].data_ptr(), h, int(hw[0, 1]),
                                               1 if packed else 0, arena.data_ptr(), _stream_ptr(self.device)),
                       self.handle, "yl_masks_image")
            v = arena[:B * per]
            return v.view(torch.int32).view(B, max_out, h, row // 4) if packed else v.view(B, max_out, h, row)
        cn = np.minimum(counts.cpu().numpy().astype(np.int64), max_out)
        sizes = cn * hw[:, 0].astype(np.int64) * rowb
        sizes = (sizes + 15) & ~15
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        total = int(sizes.sum())
        buf = torch.empty((max(total, 16),), device=self.device, dtype=torch.uint8)
        hw_d = torch.from_numpy(hw).to(self.device)
        off_d = torch.from_numpy(offs).to(self.device)
        _lib.check(self.lib.yl_masks_image(self.handle, arr, B, dets.data_ptr(), counts.data_ptr(), keep_idx.data_ptr(),
                                           max_out, float(thr), bm_d.data_ptr() if bm_d is not None else None,
                                           hw_d.data_ptr(), off_d.data_ptr(), int(hw[:, 0].max()), int(hw[:, 1].max()),
                                           1 if packed else 0, buf.data_ptr(), _stream_ptr(self.device)),
                   self.handle, "yl_masks_image")
        out = []
        for b in range(B):
            n, h, w = int(cn[b]), int(hw[b, 0]), int(hw[b, 1])
            seg = buf[int(offs[b]):int(offs[b]) + n * h * int(rowb[b])]
            out.append(seg.view(torch.int32).view(n, h, (w + 31) // 32) if packed else seg.view(n, h, w))
        return out

    def predict(self, x: torch.Tensor, mode, conf, iou, per_class_cap=300, topk=0, max_out=None, backmap=None,
                out: Optional[tuple] = None, want_idx: bool = False, center_mode="v8", wh_mode="softplus",
                fallback_nms: int = _lib.NMS_TORCHVISION):
        """Fused forward + post-processing on the context's own level buffers (no raw output copy)."""
        B = x.shape[0]
        if max_out is None:
            max_out = self.default_max_out(mode, per_class_cap, topk)
        if backmap is not None:
            backmap = backmap.to(device=self.device, dtype=torch.float32).contiguous()
        cfg = self.make_cfg(mode, conf, iou, per_class_cap, topk, max_out, center_mode, wh_mode, backmap, fallback_nms)
        if out is None:
            dets = torch.empty((B, max_out, 6), device=self.device, dtype=torch.float32)
            counts = torch.empty((B,), device=self.device, dtype=torch.int32)
        else:
            dets, counts = out
        idx = torch.empty((B, max_out), device=self.device, dtype=torch.int32) if want_idx else None
        _lib.check(self.lib.yl_predict(self.handle, x.data_ptr(), B, C.byref(cfg), dets.data_ptr(), counts.data_ptr(),
                                       idx.data_ptr() if want_idx else None, _stream_ptr(self.device)),
                   self.handle, "yl_predict")
        return (dets, counts, idx) if want_idx else (dets, counts)

    def nms(self, boxes: torch.Tensor, scores: torch.Tensor, iou_thr: float, max_det: int = 300,
            impl: int = _lib.NMS_TORCHVISION) -> torch.Tensor:
        boxes = boxes.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1, 4)
        scores = scores.to(device=self.device, dtype=torch.float32).contiguous().reshape(-1)
        n = boxes.shape[0]
        keep = torch.empty((max(max_det, 1),), device=self.device, dtype=torch.int32)
        cnt = torch.zeros((1,), device=self.device, dtype=torch.int32)
        _lib.check(self.lib.yl_nms(self.handle, boxes.data_ptr(), scores.data_ptr(), n, float(iou_thr), impl,
                                   int(max_det), keep.data_ptr(), cnt.data_ptr(), _stream_ptr(self.device)),
                   self.handle, "yl_nms")
        return keep[:int(cnt.item())].to(torch.int64)


class YOLOLiteHIP:
    """Stands where the reference's nn.Module stands (YOLOLiteMS / YOLOLiteMS_CPU): built from a
    checkpoint's meta, weights attached with load_state_dict(), moved with .to(device), called with
    a [B,3,S,S] float tensor, returns the list of level tensors."""

    def __init__(self, meta: dict, fuse_dw="auto", fuse_stem: bool = True, fuse_uib: bool = False, fuse_ir=None,
                 fuse_uir: bool = True, fuse_lat: bool = True, fuse_chain: bool = True, fuse_dws: bool = True):
        self.meta = meta
        self.fuse_dw, self.fuse_stem, self.fuse_uib, self.fuse_ir = fuse_dw, fuse_stem, fuse_uib, fuse_ir
        # every fusion policy of the host compiler is resolved HERE, once per model (no process environment)
        self._fuse_kw = dict(fuse_dw=fuse_dw, fuse_stem=fuse_stem, fuse_uib=fuse_uib, fuse_ir=fuse_ir, fuse_uir=fuse_uir,
                             fuse_lat=fuse_lat, fuse_chain=fuse_chain, fuse_dws=fuse_dws)
        self.export_concat = False
        # yl_set_option values applied to EVERY context of this model (one per input size), e.g. the pip API's serving
        # options; set with set_context_options() so that contexts that already exist get them too
        self.context_options: Dict[str, int] = {}
        self._ctxs: Dict[int, tuple] = {}
        self.program: Optional[Program] = None
        self.ctx: Optional[HipContext] = None
        self._sd = None
        self._device_index = 0
        # validate arch / required config keys now, like the reference constructor would
        cfg = meta.get("config", {}) or {}
        _ = cfg["training"]["use_p6"], cfg["training"]["use_p2"]
        arch = (meta.get("arch") or (cfg.get("model", {}) or {}).get("arch") or "YOLOLiteMS").lower()
        if arch not in ("yololitems", "yololitems_cpu"):
            raise ValueError(f"Okänd arch i meta/config: {arch}")

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = False):
        """Returns (missing_keys, unexpected_keys) like nn.Module.load_state_dict(strict=False).
        Unlike the reference, a key the forward pass needs cannot be left at its random init:
        missing weights raise."""
        try:
            self.program = build_program(self.meta, state_dict, **self._fuse_kw)
        except KeyError as e:
            raise RuntimeError(f"checkpoint lacks a weight the forward pass needs: {e.args[0]}") from None
        self._sd = state_dict
        unexpected = [k for k in state_dict if k not in self.program.known_keys]
        missing: List[str] = []
        if strict and unexpected:
            raise RuntimeError(f"unexpected keys in state_dict: {unexpected[:8]}")
        return missing, unexpected

    def to(self, device):
        dev = torch.device(device) if not isinstance(device, torch.device) else device
        if dev.type != "cuda":
            raise _lib.YoloLiteHipError("YOLOLiteHIP runs on a HIP device only (no CPU path)")
        if self.program is None:
            raise RuntimeError("load_state_dict() first")
        self._device_index = dev.index or 0
        self._ctxs = {}
        self.ctx = self._ctx_for(self.program.img_size)
        return self

    def set_context_options(self, **opts: int):
        """Options of the executor (yl_set_option) for every context of this model, present and future: a checkpoint run
        at two input sizes must not get two different kernel selections (the pip API used to set its serving
        options on the img_size context only)."""
        self.context_options.update({k: int(v) for k, v in opts.items()})
        for _, ctx in self._ctxs.values():
            for k, v in opts.items():
                ctx.set_option(k, int(v))

    def _ctx_for(self, img_size: int) -> HipContext:
        """The reference module is input-size agnostic (tools/infer.py --img_size); the HIP program is
        planned per size, so contexts are cached by input size."""
        if img_size not in self._ctxs:
            p = self.program if img_size == self.program.img_size else \
                build_program(self.meta, self._sd, img_size=img_size, **self._fuse_kw)
            ctx = HipContext(p.img_size, p.num_classes, p.level_size, p.level_anchors, p, self._device_index)
            for k, v in self.context_options.items():       # one place for EVERY context of this model
                ctx.set_option(k, v)
            self._ctxs[img_size] = (p, ctx)
        return self._ctxs[img_size][1]

    def eval(self):
        return self

    def get_strides(self):
        return list(self.program.strides)

    def get_num_anchors_per_level(self):
        return tuple(self.program.level_anchors)

    @property
    def num_classes(self):
        return self.program.num_classes

    def __call__(self, x: torch.Tensor):
        if self.ctx is None:
            raise RuntimeError("model.to('cuda') first")
        ctx = self._ctx_for(int(x.shape[-1]))
        outs = ctx.forward(x)
        if ctx.NM:                                  # build-defined seg model: (levels, prototypes [B,NM,PH,PW])
            return outs, ctx.prototypes(x.shape[0])
        if self.export_concat:                      # model_v2.py:57-64
            B = outs[0].shape[0]
            return torch.cat([o.view(B, -1, o.shape[-1]) for o in outs], dim=1)
        return outs

    forward = __call__

    def forward_decoded(self, x: torch.Tensor, center_mode="v8", wh_mode="softplus"):
        """The reference's exported "decoded" outputs (export/export_onnx.py:283-296) straight from the input."""
        if self.ctx is None:
            raise RuntimeError("model.to('cuda') first")
        return self._ctx_for(int(x.shape[-1])).forward_decoded(x, center_mode, wh_mode)


def build_model_from_meta(meta: dict, fuse_dw="auto", fuse_stem: bool = True, fuse_uib: bool = False,
                          fuse_ir=None, **fuse_kw) -> YOLOLiteHIP:
    """tools/infer.py:34-77.  (fuse_*: launch-fusion policies of the host compiler, see program.build_program)"""
    return YOLOLiteHIP(meta, fuse_dw=fuse_dw, fuse_stem=fuse_stem, fuse_uib=fuse_uib, fuse_ir=fuse_ir, **fuse_kw)


def load_model_names_imgsize_from_ckpt(weights: str, device):
    """tools/infer.py:80-102: checkpoint {"state_dict","meta"} -> (model on device, names, img_size)."""
    ckpt = torch.load(weights, map_location="cpu", weights_only=False)
    if not (isinstance(ckpt, dict) and "state_dict" in ckpt and "meta" in ckpt):
        raise RuntimeError("Checkpoint saknar 'state_dict'/'meta'. Spara vikter via save_checkpoint_state(...).")
    meta = ckpt["meta"] or {}
    model = build_model_from_meta(meta)
    missing, unexpected = model.load_state_dict(ckpt["state_dict"], strict=False)
    if missing:
        print(f"[load_state_dict] missing keys: {len(missing)}")
    if unexpected:
        print(f"[load_state_dict] unexpected keys: {len(unexpected)}")
    model.to(device).eval()
    names = meta.get("names") or [str(i) for i in range(int(meta.get("num_classes", 80)))]
    meta_img_size = int(meta.get("img_size", 640))
    return model, names, meta_img_size
