#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz|*.json by RUNNING THE REFERENCE's own Python
(/root/reference, read-only) in this container.  The reference never travels to the GPU box;
only the data files written here do.

Three of the reference's imports are third-party packages that are absent from this image
(timm, torchvision, cv2).  They are provided as stub modules:

  timm.create_model      -> oracle.backbones.create_model (the build's restated backbone; the
                            backbone is therefore NOT pinned by these fixtures)
  torchvision.ops.nms    -> oracle.postproc.nms_torchvision (restated; NOT pinned)
  torchvision.ops.box_iou-> small torch implementation (unused by the hot path)
  cv2                    -> numpy stand-ins for imread / resize(same size) / copyMakeBorder /
                            cvtColor; drawing calls are no-ops

Everything else that runs is the reference's code: scripts/model/model_v2.py (neck, heads,
layout), scripts/helpers/utils_ms.py (decode), scripts/helpers/helpers.py
(_decode_batch_to_coco_dets), tools/infer.py (decode_anchorfree_like_train with its own
pure-torch NMS, nms(), build_model_from_meta, load_model_names_imgsize_from_ckpt and the whole
main() flow).

Run:  cd /root/repo && python tests/golden/make_fixtures.py
"""
import io
import json
import os
import sys
import tempfile
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
# YL_FIXTURE_OUT: write somewhere else (tests/test_oracle_golden.py re-runs the recipe into a temp dir and
# compares with the committed files)
OUT = os.environ.get("YL_FIXTURE_OUT") or os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from oracle import backbones as obb          # noqa: E402
from oracle import model as omodel           # noqa: E402
from oracle import postproc as opost         # noqa: E402

torch.set_num_threads(1)                      # deterministic reduction order for stored outputs


# ------------------------------------------------------------------------------ stubs
def install_stubs(image_store):
    timm = types.ModuleType("timm")
    timm.create_model = obb.create_model
    sys.modules["timm"] = timm

    tv = types.ModuleType("torchvision")
    tvo = types.ModuleType("torchvision.ops")

    def _nms(boxes, scores, thr):
        k = opost.nms_torchvision(boxes.detach().cpu().numpy(), scores.detach().cpu().numpy(), float(thr))
        return torch.from_numpy(k).to(boxes.device)

    def _box_iou(a, b):
        area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])
        lt = torch.max(a[:, None, :2], b[None, :, :2]); rb = torch.min(a[:, None, 2:], b[None, :, 2:])
        wh = (rb - lt).clamp(min=0); inter = wh[..., 0] * wh[..., 1]
        return inter / (area(a)[:, None] + area(b)[None] - inter)

    tvo.nms, tvo.box_iou = _nms, _box_iou
    tv.ops = tvo
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.ops"] = tvo

    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT, cv2.COLOR_BGR2RGB = 1, 0, 4
    cv2.FONT_HERSHEY_SIMPLEX, cv2.LINE_AA = 0, 16
    cv2.imread = lambda p: image_store[p].copy()

    def _resize(im, size, interpolation=None):
        w, h = size
        assert im.shape[0] == h and im.shape[1] == w, "cv2 stub only supports identity resize"
        return im.copy()

    def _border(im, top, bottom, left, right, btype, value=(0, 0, 0)):
        out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, im.shape[2]), im.dtype)
        out[...] = np.asarray(value, im.dtype)
        out[top:top + im.shape[0], left:left + im.shape[1]] = im
        return out

    cv2.resize, cv2.copyMakeBorder = _resize, _border
    cv2.cvtColor = lambda im, code: im[..., ::-1].copy()
    cv2.getTextSize = lambda *a, **k: ((10, 10), 2)
    cv2.rectangle = cv2.putText = lambda *a, **k: None
    cv2.imwrite = lambda *a, **k: True
    sys.modules["cv2"] = cv2


def to_np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def main():
    images = {}
    install_stubs(images)
    os.chdir(REF)                              # the reference appends os.getcwd() to sys.path
    sys.path.append(REF)
    from scripts.model import model_v2 as ref_model
    from scripts.helpers import utils_ms as ref_decode
    from scripts.helpers import helpers as ref_helpers
    # by file path: `import tools.infer` would resolve to THIS repository's tools/ package (sys.path[0] = REPO)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_tools_infer", os.path.join(REF, "tools", "infer.py"))
    ref_infer = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_infer)

    # ---------------------------------------------------------------- A. neck + head + layout
    cases = [
        dict(tag="cpu_d1_h1", cls="YOLOLiteMS_CPU", kw=dict(fpn_channels=16, depth_multiple=0.5, width_multiple=1.0,
             head_depth=1, num_classes=3, num_anchors_per_level=(1, 1, 1), use_p6=False, use_p2=False)),
        dict(tag="cpu_d2_h2_p6_a2", cls="YOLOLiteMS_CPU", kw=dict(fpn_channels=24, depth_multiple=1.0, width_multiple=0.85,
             head_depth=2, num_classes=5, num_anchors_per_level=(2, 2, 2, 2), use_p6=True, use_p2=False)),
        dict(tag="gpu_d2_h1", cls="YOLOLiteMS", kw=dict(fpn_channels=20, depth_multiple=1.0, width_multiple=1.0,
             head_depth=1, num_classes=1, num_anchors_per_level=(1, 1, 1), use_p6=False, use_p2=False)),
        dict(tag="gpu_d1_h2_p2", cls="YOLOLiteMS", kw=dict(fpn_channels=16, depth_multiple=0.5, width_multiple=1.0,
             head_depth=2, num_classes=4, num_anchors_per_level=(1, 1, 1, 1), use_p6=False, use_p2=True)),
    ]
    blob = {}
    for i, c in enumerate(cases):
        m = getattr(ref_model, c["cls"])(backbone="oracle_tiny", **c["kw"]).eval()
        omodel.randomize_(m, seed=100 + i)
        x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(7 + i))
        with torch.no_grad():
            outs = m(x)
            m.export_concat = True
            cat = m(x)
        blob[f"{c['tag']}/x"] = x.numpy()
        for k, v in to_np(m.state_dict()).items():
            blob[f"{c['tag']}/sd/{k}"] = v
        for j, o in enumerate(outs):
            blob[f"{c['tag']}/out{j}"] = o.numpy()
        blob[f"{c['tag']}/concat"] = cat.numpy()
        blob[f"{c['tag']}/strides"] = np.asarray(m.get_strides())
        blob[f"{c['tag']}/anchors"] = np.asarray(m.get_num_anchors_per_level())
    np.savez_compressed(os.path.join(OUT, "neck_head.npz"), **blob)
    with open(os.path.join(OUT, "neck_head_cases.json"), "w") as f:
        json.dump([dict(tag=c["tag"], cls=c["cls"], kw={k: (list(v) if isinstance(v, tuple) else v)
                                                       for k, v in c["kw"].items()}) for c in cases], f, indent=1)

    # ---------------------------------------------------------------- B. decode KATs
    g = torch.Generator().manual_seed(11)
    blob = {}
    zero = [torch.zeros(1, 1, s, s, 8) for s in (80, 40, 20)]
    d = ref_decode.decode_preds_anchorfree(zero, img_size=640)
    blob["zero/box_first"] = d["box"][0, :3].numpy()
    blob["zero/box_last"] = d["box"][0, -3:].numpy()
    lv = [torch.randn(2, 1, s, s, 8, generator=g) * 3.0 for s in (16, 8, 4)]
    lv[0][0, 0, 0, 0, 2] = 25.0          # softplus threshold branch (x > 20 -> x)
    lv[0][0, 0, 0, 1, 3] = 20.0          # exactly at the threshold
    lv[1][1, 0, 7, 7, 2:4] = 30.0        # clamps to img_size-1
    lv[2][0, 0, 0, 0, 0:2] = -40.0       # sigmoid underflow
    for j, t in enumerate(lv):
        blob[f"rand/level{j}"] = t.numpy()
    for cm in ("v8", "simple"):
        for wm in ("softplus", "v8", "exp"):
            d = ref_decode.decode_preds_anchorfree(lv, img_size=128, center_mode=cm, wh_mode=wm)
            for k in ("box", "obj", "cls"):
                blob[f"rand/{cm}_{wm}/{k}"] = d[k].numpy()
    lv2 = [torch.randn(1, 2, s, s, 6, generator=g) for s in (6, 3)]   # A=2, C=1, non-pow2 stride (96/6, 96/3)
    for j, t in enumerate(lv2):
        blob[f"a2/level{j}"] = t.numpy()
    d = ref_decode.decode_preds_anchorfree(lv2, img_size=96)
    for k in ("box", "obj", "cls"):
        blob[f"a2/{k}"] = d[k].numpy()
    np.savez_compressed(os.path.join(OUT, "decode.npz"), **blob)

    # ---------------------------------------------------------------- C. eval + fallback pipelines
    blob = {}
    meta_cases = []

    def levels_for(seed, B, C, sizes, scale, A=1):
        gg = torch.Generator().manual_seed(seed)
        return [torch.randn(B, A, s, s, 5 + C, generator=gg) * scale for s in sizes]

    pipe_cases = [
        dict(tag="c3", seed=21, B=2, C=3, sizes=(16, 8, 4), scale=2.0, img=128, conf=0.4, iou=0.5),
        dict(tag="c3_lowconf", seed=22, B=2, C=3, sizes=(16, 8, 4), scale=2.0, img=128, conf=0.001, iou=0.65),
        dict(tag="c1", seed=23, B=2, C=1, sizes=(16, 8, 4), scale=2.0, img=128, conf=0.3, iou=0.5),
        dict(tag="c80", seed=24, B=1, C=80, sizes=(20, 10, 5), scale=2.5, img=160, conf=0.25, iou=0.45),
        dict(tag="empty", seed=25, B=2, C=3, sizes=(8, 4, 2), scale=0.1, img=64, conf=0.9, iou=0.5),
        dict(tag="cap", seed=26, B=1, C=2, sizes=(40, 20, 10), scale=3.0, img=320, conf=0.05, iou=0.95),
    ]
    for c in pipe_cases:
        lv = levels_for(c["seed"], c["B"], c["C"], c["sizes"], c["scale"])
        if c["tag"] == "c3":                  # exact score ties + duplicate boxes (IoU == 1) + zero-area box
            lv[0][0, 0, 1, 1] = lv[0][0, 0, 0, 0]
            lv[0][0, 0, 3, 3, 2:4] = -60.0    # softplus -> 0 => zero-area box
            lv[0][0, 0, 3, 3, 4:] = 6.0
        for j, t in enumerate(lv):
            blob[f"{c['tag']}/level{j}"] = t.numpy()
        # eval pipeline: reference function, stubbed torchvision nms
        dets = ref_helpers._decode_batch_to_coco_dets(lv, c["img"], conf_th=c["conf"], iou_th=c["iou"], add_one=True)
        for b, dl in enumerate(dets):
            blob[f"{c['tag']}/eval/{b}/bbox"] = np.asarray([d_["bbox"] for d_ in dl], np.float64).reshape(-1, 4)
            blob[f"{c['tag']}/eval/{b}/score"] = np.asarray([d_["score"] for d_ in dl], np.float64)
            blob[f"{c['tag']}/eval/{b}/cat"] = np.asarray([d_["category_id"] for d_ in dl], np.int64)
        # fallback pipeline: reference function with the reference's OWN pure-torch NMS
        # (make `from torchvision.ops import nms` fail inside tools/infer.py:nms)
        saved = {k: sys.modules.pop(k) for k in ("torchvision", "torchvision.ops")}
        try:
            fb = ref_infer.decode_anchorfree_like_train(lv, c["img"], conf_th=c["conf"], iou_th=c["iou"], topk=300)
        finally:
            sys.modules.update(saved)
        for b in range(c["B"]):
            blob[f"{c['tag']}/fallback/{b}/boxes"] = fb["boxes"][b].numpy().reshape(-1, 4)
            blob[f"{c['tag']}/fallback/{b}/scores"] = fb["scores"][b].numpy()
            blob[f"{c['tag']}/fallback/{b}/classes"] = fb["classes"][b].numpy()
        meta_cases.append(c)
    # nms() wrapper of tools/infer.py with the stubbed torchvision (cap) and without (greedy)
    gg = torch.Generator().manual_seed(31)
    ctr = torch.rand(500, 2, generator=gg) * 200
    wh = torch.rand(500, 2, generator=gg) * 40 + 2
    bx = torch.cat([ctr - wh / 2, ctr + wh / 2], 1)
    sc = torch.rand(500, generator=gg)
    blob["nms/boxes"], blob["nms/scores"] = bx.numpy(), sc.numpy()
    blob["nms/keep_tv_cap300_iou09"] = ref_infer.nms(bx, sc, 0.9, 300).numpy()
    blob["nms/keep_tv_cap20_iou05"] = ref_infer.nms(bx, sc, 0.5, 20).numpy()
    saved = {k: sys.modules.pop(k) for k in ("torchvision", "torchvision.ops")}
    try:
        blob["nms/keep_greedy_iou05"] = ref_infer.nms(bx, sc, 0.5, 300).numpy()
        blob["nms/keep_greedy_iou03"] = ref_infer.nms(bx, sc, 0.3, 300).numpy()
    finally:
        sys.modules.update(saved)
    np.savez_compressed(os.path.join(OUT, "pipelines.npz"), **blob)
    with open(os.path.join(OUT, "pipelines_cases.json"), "w") as f:
        json.dump([{k: (list(v) if isinstance(v, tuple) else v) for k, v in c.items()} for c in meta_cases], f, indent=1)

    # ---------------------------------------------------------------- D. the whole tools/infer.py main() flow
    # checkpoint in the reference format (tools/train.py:62-75) for a tiny model, synthetic 96x96 BGR image
    # (same size as img_size so the cv2.resize stub is an identity), conf 0.4 / iou 0.5 defaults.
    meta = dict(metric_key="map50", metric_value=0.0, names=["a", "b", "c"], num_classes=3, img_size=96,
                arch="YOLOLiteMS_CPU", backbone="oracle_tiny", num_anchors_per_level=(1, 1, 1),
                config=dict(model=dict(arch="YOLOLiteMS_CPU", backbone="oracle_tiny", fpn_channels=16,
                                       depth_multiple=0.5, width_multiple=1.0, head_depth=1, num_classes=3),
                            training=dict(img_size=96, use_p6=False, use_p2=False)))
    m = ref_infer.build_model_from_meta(meta).eval()
    omodel.randomize_(m, seed=5, head_noise=2.0)
    with torch.no_grad():                      # lift objectness so that a useful number of boxes pass conf 0.4
        for k in ("head3", "head4", "head5"):
            getattr(m, k)["out"]["obj"].bias.add_(5.0)
            getattr(m, k)["out"]["cls"].bias.add_(1.5)
    tmp = tempfile.mkdtemp(prefix="ylfix_")
    ck = os.path.join(tmp, "tiny.pt")
    torch.save({"state_dict": {k: v.cpu() for k, v in m.state_dict().items()}, "meta": meta}, ck)
    rng = np.random.RandomState(1234)
    img_sq = rng.randint(0, 256, size=(96, 96, 3)).astype(np.uint8)
    img_wide = rng.randint(0, 256, size=(48, 96, 3)).astype(np.uint8)      # letterboxed: pad top/bottom 24
    images[os.path.join(tmp, "sq.png")] = img_sq
    images[os.path.join(tmp, "wide.png")] = img_wide
    for p in images:
        open(p, "wb").close()                  # Path(p).exists() must hold
    os.chdir(tmp)
    argv = sys.argv
    results = {}
    for name in ("sq", "wide"):
        sys.argv = ["infer.py", "--weights", ck, "--img", os.path.join(tmp, f"{name}.png"), "--device", "cpu"]
        buf = io.StringIO()
        so = sys.stdout
        sys.stdout = buf
        try:
            ref_infer.main()
        finally:
            sys.stdout = so
        run_dirs = sorted(os.listdir(os.path.join(tmp, "runs", "infer")), key=int)
        with open(os.path.join(tmp, "runs", "infer", run_dirs[-1], "json", f"{name}.json")) as f:
            results[name] = json.load(f)["detections"]
    sys.argv = argv
    blob = {f"sd/{k}": v for k, v in to_np(m.state_dict()).items()}
    blob["img_sq"], blob["img_wide"] = img_sq, img_wide
    for name, dets in results.items():
        blob[f"{name}/bbox_xyxy"] = np.asarray([d_["bbox_xyxy"] for d_ in dets], np.float64).reshape(-1, 4)
        blob[f"{name}/score"] = np.asarray([d_["score"] for d_ in dets], np.float64)
        blob[f"{name}/class_id"] = np.asarray([d_["class_id"] for d_ in dets], np.int64)
        print(f"[main-flow] {name}: {len(dets)} detections")
    np.savez_compressed(os.path.join(OUT, "infer_main.npz"), **blob)
    with open(os.path.join(OUT, "infer_main_meta.json"), "w") as f:
        mm = dict(meta); mm["num_anchors_per_level"] = list(mm["num_anchors_per_level"])
        json.dump(mm, f, indent=1)
    for fn in sorted(os.listdir(OUT)):
        print(f"{fn:28s} {os.path.getsize(os.path.join(OUT, fn)):9d} B")


if __name__ == "__main__":
    main()
