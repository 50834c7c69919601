#!/usr/bin/env python3
"""Headline benchmark: images/sec + p50 ms/frame of the YoloLite inference hot path
(backbone -> FPN -> heads -> decode -> class-wise NMS), edge_n 640x640 batch 64 per GPU
(BASELINE.json configs[1]; N>1: the same per GPU = configs[4], weak scaling, one RCCL all-gather of
the packed detections per step).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path over one batch that is already resident in HBM:
yl_predict (36 fused conv launches per batch chunk, decode inside the head-output convs, NMS) [+ all-gather].  Synthetic data, seeded
synthetic weights (no checkpoints exist in this environment).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# The executor overlaps two batch chunks on two HIP streams.  ROCm maps streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4) round-robin at creation; once RCCL has created its own streams the chunk stream can end up
# sharing a hardware queue with the caller's stream, which serialises the chunks AND their fork/join barriers
# (measured: 30.5 k -> 22.9 k images/s as soon as init_process_group("nccl") has run).  8 queues avoid the
# aliasing; must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PEAK_16BIT_MFMA_TFLOPS = 2500.0     # dense bf16 / fp16 MFMA peak (same guide); the reduced-precision modes' roof
RIDGE = PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)   # 19.7 FLOP/B


def synth_images(B, S, seed=1234):
    """uniform u8 image normalised with the ImageNet mean/std (what a letterboxed frame looks like)."""
    rng = np.random.RandomState(seed)
    u8 = rng.randint(0, 256, size=(B, S, S, 3)).astype(np.float32) / 255.0
    im = (u8 - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    return torch.from_numpy(np.ascontiguousarray(im.transpose(0, 3, 1, 2)))


STRESS_SEEDS = {"edge_n": 2, "edge_m": 9, "yololite_m": 10}
MAX_OUT = 1024                     # result rows per image: sized so that the workload drops nothing (asserted)


def build_workload(model_name="edge_n", S=640, B=64, seed=1, seg=False, dev="cuda:0", rank=0, stress=False,
                   fuse_dw="auto", fuse_stem=True, fuse_uib=False):
    """The benchmark's model + input, shared with tests/test_bench_config.py (the -m gpu parity tests of exactly
    this configuration).  Seeded synthetic weights (program.synth_state_dict) whose detection head is then
    CALIBRATED (program.calibrate_head: exact per-row rescaling of the head's output convs from the statistics of
    one HIP forward pass over 8 calibration images) so that every seed detects: O(100-600) detections per image
    spread over >= 40 of the 80 classes at conf 0.4.  stress=True is round 1's workload instead (uncalibrated
    N(0,2) head noise, a seed that happens to fire: ~2000 survivors in 10 classes -- an NMS stress case, labelled as
    such in the JSON)."""
    import yololite_amd as ya
    from yololite_amd.program import calibrate_head, synth_state_dict, zoo_meta
    meta = zoo_meta(model_name, 80, S, seg=bool(seg))
    kw = dict(fuse_dw=fuse_dw, fuse_stem=fuse_stem, fuse_uib=fuse_uib)
    if stress:
        seed = STRESS_SEEDS.get(model_name, 2)
        sd = synth_state_dict(meta, seed=seed, head_noise=2.0)
    else:
        sd0 = synth_state_dict(meta, seed=seed)
        m0 = ya.build_model_from_meta(meta, **kw)
        m0.load_state_dict(sd0)
        m0.to(dev)
        out = m0(synth_images(8, S, seed=99).to(dev))
        lv = out[0] if seg else out
        sd = calibrate_head(sd0, meta, [t.cpu().numpy() for t in lv])
        del m0, out, lv
    model = ya.build_model_from_meta(meta, **kw)
    model.load_state_dict(sd)
    model.to(dev)
    if os.environ.get("YL_DEV_SELECT"):                     # developer A/B runs of whole test / bench commands (0 in production)
        model._ctx_for(S).set_option("dev_select", int(os.environ["YL_DEV_SELECT"], 0))
    return dict(meta=meta, sd=sd, model=model, ctx=model._ctx_for(S), prog=model.program, seed=seed,
                x=synth_images(B, S, seed=1234 + rank).to(dev))


def _cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _ms_stats(v):
    v = np.asarray(v, np.float64)
    return {"mean": round(float(v.mean()), 3), "std": round(float(v.std()), 3), "p50": round(float(np.percentile(v, 50)), 3),
            "p90": round(float(np.percentile(v, 90)), 3), "p95": round(float(np.percentile(v, 95)), 3)}


def cpu_baseline(meta, sd, S, conf, iou, budget_s=20.0, B=64):
    """The oracle (CPU restatement of the reference path, PyTorch-CPU fp32 + numpy NMS; checker code used only as the
    timed baseline) on this host's cores, protocol of BASELINE.md section 3 = the reference's own harnesses:
      * config 1: batch-1 tools/infer.py-shaped flow (letterbox + normalise -> forward -> decode + per-class NMS ->
        back-map) on a 480x640 BGR frame, 10 warm-up runs (export/infer_onnx.py:99,136-139), perf_counter around
        pre / infer / post (:152-244), mean / std / p50 / p90 / p95 and img/s = 1000 / mean(total_ms) (:273-296);
      * throughput comparator: the SAME batch-B workload the GPU runs (forward + post-processing on normalised
        tensors), 1 warm-up batch, ms/img = sum ms / sum images (scripts/helpers/evaluate.py:253-303).
    Bounded: about budget_s seconds in total."""
    from oracle import model as omodel, postproc as opost, preproc as opre
    m = omodel.build_from_meta(meta).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    # thread count: measured on the GPU box (2 x EPYC 9575F, 256 logical CPUs): 8-16 threads are the
    # fastest for this small-channel network (158 ms / 8 images), 128 threads are 5x slower
    n = min(16, os.cpu_count() or 1)
    torch.set_num_threads(n)
    img = np.random.RandomState(7).randint(0, 256, size=(480, 640, 3)).astype(np.uint8)
    pre, inf, post = [], [], []

    def once(rec):
        t0 = time.perf_counter()
        x, (padx, pady, scale, w0, h0) = opre.preprocess(img, S)
        xt = torch.from_numpy(x[None])
        t1 = time.perf_counter()
        lv = m(xt)
        t2 = time.perf_counter()
        r = opost.pipeline_main(lv, S, conf, iou)
        opost.backmap(r["boxes"][0], padx, pady, scale, w0, h0)
        t3 = time.perf_counter()
        if rec:
            pre.append((t1 - t0) * 1e3); inf.append((t2 - t1) * 1e3); post.append((t3 - t2) * 1e3)
    with torch.no_grad():
        for _ in range(10):
            once(False)
        t_start = time.perf_counter()
        while len(pre) < 200 and (time.perf_counter() - t_start < 0.35 * budget_s or len(pre) < 5):
            once(True)
        tot = np.asarray(pre) + np.asarray(inf) + np.asarray(post)
        b1 = {"pre_ms": _ms_stats(pre), "infer_ms": _ms_stats(inf), "post_ms": _ms_stats(post), "total_ms": _ms_stats(tot),
              "images_per_sec": round(1000.0 / float(tot.mean()), 2), "runs": len(pre), "warmup": 10,
              "flow": "480x640 BGR u8 -> letterbox+normalise -> forward -> decode + per-class NMS -> back-map"}
        x = synth_images(B, S)
        opost.pipeline_main(m(x), S, conf, iou)           # warm-up batch
        t0 = time.perf_counter()
        done = 0
        while True:
            opost.pipeline_main(m(x), S, conf, iou)
            done += B
            el = time.perf_counter() - t0
            if el >= 0.4 * budget_s or done >= 10 * B:
                break
    return {"value": round(done / el, 2), "unit": "images/sec", "cores": n, "kind": "port",
            "sample": f"{done} images (batches of {B}) of the same {S}x{S} workload, forward+decode+NMS, {el:.1f} s wall "
                      f"(+ {len(pre)} batch-1 runs), torch {n} threads of {os.cpu_count()} logical CPUs",
            "cpu_model": _cpu_model_string(), "logical_cpus": os.cpu_count(), "batch1_infer_flow": b1}


def gpu_batch1_flow(meta, sd, S, conf, iou, dev, budget_s=4.0):
    """The GPU side of the reference's only published protocol (export/infer_onnx.py:136-296, BENCHMARK.md:336-339):
    batch 1 through the pip API -- YoloLite(checkpoint).predict(frame) on the SAME 480x640 BGR uint8 frame the CPU
    baseline's batch-1 flow uses: host packing + H2D + letterbox/normalise kernel (pre), forward (infer), decode + NMS +
    back-map (post), detections copied to the host.  10 warm-up runs, then up to 200 runs / budget_s seconds; pre_ms is wall
    clock around a stream sync, infer_ms / post_ms are HIP-event intervals (yl_last_timing), total = their sum as in the
    reference's harness; wall_ms is perf_counter around the whole predict() call (includes the D2H copy of the rows)."""
    import tempfile
    from yololite_amd.api import YoloLite
    m = dict(meta)
    m["names"] = [f"c{i}" for i in range(int(meta.get("num_classes") or 80))]
    with tempfile.TemporaryDirectory() as td:
        ck = os.path.join(td, "bench_model.pt")
        torch.save({"state_dict": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, "meta": m}, ck)
        yl = YoloLite(ck, device=str(dev))
    img = np.random.RandomState(7).randint(0, 256, size=(480, 640, 3)).astype(np.uint8)
    for _ in range(10):
        r = yl.predict(img, conf=conf, iou=iou)
    rows = {k: [] for k in ("pre_ms", "infer_ms", "post_ms", "total_ms", "wall_ms")}
    t_start = time.perf_counter()
    while len(rows["wall_ms"]) < 200 and (time.perf_counter() - t_start < budget_s or len(rows["wall_ms"]) < 20):
        t0 = time.perf_counter()
        r = yl.predict(img, conf=conf, iou=iou)
        rows["wall_ms"].append((time.perf_counter() - t0) * 1e3)
        for k in ("pre_ms", "infer_ms", "post_ms", "total_ms"):
            rows[k].append(r[0]["speed"][k])
    out = {k: _ms_stats(v) for k, v in rows.items()}
    out.update(images_per_sec=round(1000.0 / float(np.mean(rows["total_ms"])), 1),
               images_per_sec_wall=round(1000.0 / float(np.mean(rows["wall_ms"])), 1), runs=len(rows["wall_ms"]), warmup=10,
               detections=int(len(r[0]["scores"])),
               flow="480x640 BGR u8 on the host -> YoloLite.predict (pack + H2D + letterbox/normalise kernel -> forward -> "
                    "decode + per-class NMS + back-map -> rows on the host), batch 1, one chunk, launches replayed from a hipGraph "
                    "(time_split: forward and post-processing as two replays with HIP events between them), split_k 1")
    return out


def synth_coco(n_img, n_cls=80, gt_per_img=8, det_per_img=100, seed=5):
    """COCO-style annotation / detection lists of an evaluation pass (conf 0.001 keeps ~100 dets/img)."""
    r = np.random.RandomState(seed)
    anns, dets = [], []
    for img in range(1, n_img + 1):
        g = np.c_[r.uniform(0, 500, (gt_per_img, 2)), r.uniform(10, 140, (gt_per_img, 2))]
        gc = r.randint(1, n_cls + 1, gt_per_img)
        for k in range(gt_per_img):
            anns.append({"image_id": img, "category_id": int(gc[k]), "bbox": [float(v) for v in g[k]]})
        j = r.randint(gt_per_img, size=det_per_img)
        near = r.rand(det_per_img) < 0.5
        b = np.where(near[:, None], g[j] * (1 + r.normal(0, 0.08, (det_per_img, 4))),
                     np.c_[r.uniform(0, 500, (det_per_img, 2)), r.uniform(10, 140, (det_per_img, 2))]).astype(np.float32)
        c = np.where(near, gc[j], r.randint(1, n_cls + 1, det_per_img))
        sc = r.rand(det_per_img).astype(np.float32)
        for k in range(det_per_img):
            dets.append({"image_id": img, "category_id": int(c[k]), "bbox": [float(v) for v in b[k]],
                         "score": float(sc[k])})
    return anns, dets


def bench_eval_consumers(args):
    """--workload eval: the evaluate-path consumers -- P/R/F1 curves (201 steps) and the
    detection confusion matrix over one validation pass of N images x 100 detections.  Separate JSON line;
    the headline workload is unaffected."""
    from yololite_amd import evalops
    n_img = args.batch * 32                               # 2048 images at the default batch
    anns, dets = synth_coco(n_img)
    names = [f"c{i}" for i in range(80)]

    def step():
        s = evalops.build_curves_from_coco([], anns, dets, None, iou=0.5, steps=201)
        cm = evalops.confusion_matrix_counts(anns, dets, 80, 0.5, s["best_conf"])
        return s, cm
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    evalops.DEVICE_MS.update(total=0.0, launches=0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / args.steps
    dev_ms = evalops.DEVICE_MS["total"] / args.steps
    out = {"metric": "eval detections/sec (P/R/F1 sweep + confusion matrix)", "value": round(len(dets) / el, 1),
           "device_kernels_ms_per_step": round(dev_ms, 3), "device_kernel_launches_per_step": evalops.DEVICE_MS["launches"] // args.steps,
           "device_kernels_share_of_step": round(dev_ms / (el * 1e3), 4),
           "detections_per_sec_device_kernels_only": round(len(dets) / (dev_ms * 1e-3), 1) if dev_ms > 0 else None,
           "unit": "detections/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(el * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64/f32", "data": "synthetic",
           "config": {"workload": f"{n_img} images x 100 detections x 8 ground truths, 80 classes, host list "
                                  f"-> array conversion included"}}
    if not args.no_cpu_baseline:
        from oracle import evalcons as oeval              # checker code, used only as the timed baseline
        m = n_img                                         # the whole pass: ~10 s of python loops
        a2 = [a for a in anns if a["image_id"] <= m]
        d2 = [d for d in dets if d["image_id"] <= m]
        t0 = time.perf_counter()
        s = oeval.build_curves_from_coco([], a2, d2, None, iou=0.5, steps=201)
        oeval.confusion_matrix_counts(a2, d2, 80, 0.5, s["best_conf"])
        cel = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(len(d2) / cel, 1), "unit": "detections/sec", "cores": 1, "kind": "port",
                               "sample": f"{m} images ({len(d2)} detections) of the same lists, {cel:.1f} s wall, "
                                         f"python loops as in the reference"}
    print(json.dumps(out))


def bench_tracker(args):
    """--workload track: Kalman-SORT tracker bank, S = batch streams x 48 objects,
    one yl_track_update launch per frame on detections resident on the device.  Separate JSON line."""
    from yololite_amd.tracker import TrackerBank
    S, n_obj, F, max_out = args.batch, 48, 60, 64
    r = np.random.RandomState(3)
    pos = r.uniform(40, 600, (S, n_obj, 2)); vel = r.uniform(-4, 4, (S, n_obj, 2)); wh = r.uniform(20, 60, (S, n_obj, 2))
    cls = r.randint(0, 8, (S, n_obj)).astype(np.float32)
    frames = []
    for f in range(F):
        c = pos + vel * f + r.normal(0, 0.6, (S, n_obj, 2))
        d = np.zeros((S, max_out, 6), np.float32)
        d[:, :n_obj, :4] = np.concatenate([c - wh / 2, c + wh / 2], -1)
        d[:, :n_obj, 4] = r.uniform(0.4, 0.99, (S, n_obj)); d[:, :n_obj, 5] = cls
        frames.append(d)
    dev = torch.device("cuda", torch.cuda.current_device())
    fr_dev = [torch.from_numpy(d).to(dev) for d in frames]
    cnt = torch.full((S,), n_obj, dtype=torch.int32, device=dev)
    bank = TrackerBank(S, max_tracks=256, device=dev)

    def run():
        bank.reset(-1)
        for d in fr_dev:
            bank.update(d, cnt)
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / (args.steps * F)
    out = {"metric": "tracker stream-frames/sec", "value": round(S / el, 1), "unit": "stream-frames/sec", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{S} streams x {n_obj} objects x {F} frames, detections resident in HBM, "
                                  f"one launch per frame (ms_per_step = one frame of all streams)"}}
    if not args.no_cpu_baseline:
        from oracle import tracker as otrack              # checker code, used only as the timed baseline
        ns = S
        t0 = time.perf_counter()
        for s in range(ns):
            trk = otrack.SortOracle()
            for d in frames:
                trk.update(d[s, :n_obj, :4], d[s, :n_obj, 4], d[s, :n_obj, 5].astype(np.int32))
        cel = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(ns * F / cel, 1), "unit": "stream-frames/sec", "cores": 1, "kind": "port",
                               "sample": f"{ns} of the {S} streams x {F} frames, {cel:.1f} s wall, numpy as in the reference"}
    print(json.dumps(out))


def wino_eligible(L):
    """the predicate of yl_create (yl_api.hip: the layers that get a Winograd weight image)"""
    return (L.op == 1 and L.k == 3 and L.stride == 1 and L.dw_k == 0 and L.c2 == 0 and L.c3 == 0 and L.pad_t == 1 and L.pad_l == 1
            and L.in_shift <= 1 and L.cin >= 16 and L.cout >= 16 and L.cout % 4 == 0
            and L.head_level < 0 and L.up_slot < 0 and L.scale_slot < 0)


def wino_layers(prog, mode):
    """indices of the layers option "winograd" = mode runs as Winograd F(2x2,3x3): 1 (the library default) = every
    eligible layer, 2 = only the >= 64-channel layers on the largest grid they occur on (the finest level's smooth block)"""
    el = [i for i, L in enumerate(prog.layers) if wino_eligible(L)]
    if mode == 1 or not el:
        return set(el) if mode else set()
    if mode != 2:
        return set()
    el = [i for i in el if prog.layers[i].cin >= 64 and prog.layers[i].cout >= 64]
    if not el:
        return set()
    hw = lambda i: prog.slots[prog.layers[i].out_slot][0] * prog.slots[prog.layers[i].out_slot][1]
    top = max(hw(i) for i in el)
    return {i for i in el if hw(i) >= top}


def kernel_family(L, winograd=False, out_hw=None, batch=1):
    """Kernel the dispatcher (yl_launch_conv_multi, yl_conv.hip / yl_convc.hip) picks for a fused layer of the program:
    a label for the roofline object, the rocprofv3 summaries under profiles/ carry the exact instantiation.
    winograd: this layer runs as Winograd (see wino_layers); out_hw / batch: output grid and images per launch (the
    round-5 kernels with 8 x 8-pixel windows take a layer only where the grid fills them, yl_convc.hip)."""
    if L.op == 3:
        return "yl_stemdw_kernel" if L.dw_k == 3 else "yl_stemblock_kernel"
    if L.op == 4:
        return "yl_se_gate_kernel"
    if L.op != 1:
        return "yl_stem_mfma_kernel" if L.op == 0 else "yl_dw_tile_kernel"
    nt, kb = -(-L.cout // 16), -(-L.cin // 16)
    oh, ow = out_hw if out_hw else (0, 0)
    if winograd and wino_eligible(L):
        th, tw = (oh + 1) // 2, (ow + 1) // 2
        if kb >= 4 and nt >= 3 and th * tw * 100 >= -(-th // 4) * -(-tw // 4) * 16 * 65:
            return "yl_conv_wino2_kernel"
        return "yl_conv_wino_kernel"
    if L.dw_k == 0:
        if L.k == 1:
            return "yl_conv_pwt_kernel"
        streamed = L.k == 3 and ((nt % 7 == 0 and 9 * kb * 7 > 96) or (nt == 4 and 9 * kb * 4 > 96))
        return "yl_conv_kxk_kernel" if streamed else "yl_conv_mfma_kernel"
    if nt <= 6:
        return "yl_conv_dwt_kernel"
    if L.dw_k == 3 and kb >= 12 and nt > 8 and (nt % 7 == 0 or nt % 8 == 0):
        wins = batch * -(-oh // 8) * -(-ow // 8)
        if L.dw_stride == 1 and L.dw_pad_t == 1 and nt in (16, 21) and batch * oh * ow * 10 >= wins * 64 * 8 and wins >= 768:
            return "yl_conv_dwl_kernel"
        return "yl_conv_dwk_kernel"
    return "yl_conv_dwh_kernel"


def csrc_digest():
    """SHA-256 over the kernel sources (csrc/*.hip, *.h and the C header), the key that ties a committed PMC pass to
    the code it was taken on: tools/pmc_summary.py stores it in profiles/rNN_pmc_traffic*.json and roofline.traffic
    is null as soon as the sources differ (git is not available on the GPU box, file contents are)."""
    import hashlib
    h = hashlib.sha256()
    cs = os.path.join(ROOT, "yololite-official-repo_amd", "csrc")
    files = sorted(f for f in os.listdir(cs) if f.endswith((".hip", ".h")))
    for f in [os.path.join(cs, f) for f in files] + [os.path.join(ROOT, "include", "yololite_hip.h")]:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


PMC_FILES = {("edge_n", 0, 64): "pmc_traffic.json", ("yololite_m", 0, 32): "pmc_traffic_yololite_m_b32.json",
             ("edge_m", 1, 32): "pmc_traffic_edge_m_seg_b32.json", ("yololite_m_v2", 0, 32): "pmc_traffic_yololite_m_v2_b32.json"}


def pmc_traffic(model_name, seg, B, kname, stem_pattern):
    """HBM-side bytes per launch of kernel `kname` from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    separate runs, tools/profile_round.sh; bench.py cannot run the profiler on itself).  Only a pass taken on EXACTLY
    these kernel sources counts (csrc_sha256 in the file == csrc_digest()); FETCH_SIZE is corrected with the factor
    measured on independent kernels of known byte counts (profiles/rNN_fetch_calibration.json: 0.50-0.53)."""
    tail = PMC_FILES.get((model_name, int(seg), B))
    if not tail or not kname:
        return None, "no committed PMC pass for this configuration"
    cand = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_" + tail)), reverse=True)
    for f in cand:
        try:
            with open(os.path.join(ROOT, "profiles", f)) as fh:
                j = json.load(fh)
        except (OSError, ValueError):
            continue
        if j.get("csrc_sha256") != csrc_digest():
            continue
        hit = [v for k, v in j["kernels"].items() if kname in k]
        if not hit:
            continue
        ffac = 0.5
        for cf in sorted((c for c in os.listdir(os.path.join(ROOT, "profiles")) if c.endswith("_fetch_calibration.json")), reverse=True):
            try:
                with open(os.path.join(ROOT, "profiles", cf)) as fh:
                    ffac = json.load(fh)["kernels"]["calib_x3s8" if stem_pattern else "calib_x4"]["counter_over_actual"]
                break
            except (OSError, KeyError, ValueError):
                continue
        v = hit[0]
        return (round((v["FETCH_SIZE_KB_median"] / ffac + v["WRITE_SIZE_KB_median"]) * 1024.0),
                f"profiles/{f} (csrc_sha256 {j['csrc_sha256']}): median over the full-batch dispatches of FETCH_SIZE / {ffac} + "
                f"WRITE_SIZE, eager launches; factor from independent known-byte-count kernels")
    return None, f"no PMC pass under profiles/ was taken on these kernel sources (csrc_sha256 {csrc_digest()})"


def traffic_per_step(model_name, seg, B, prog, S, max_out):
    """HBM-side bytes of ONE yl_predict step from the committed counter passes of exactly these kernel sources
    (profiles/rNN_pmc_traffic*.json: `per_step` = sum over the step's dispatches of FETCH_SIZE and WRITE_SIZE, taken on
    predict-only runs: tools/profile_round.sh) against the compulsory bytes of the step: the fp32 input batch, the packed
    detections and the weights once.  null when no pass matches the sources."""
    tail = PMC_FILES.get((model_name, int(seg), B))
    comp = 4.0 * 3 * S * S * B + B * (max_out * 6 * 4 + 4) + 4.0 * sum(
        sum(int(np.asarray(a).size) for a in (l.w, l.b, l.dw_w, l.dw_b, l.w2, l.b2, l.w3, l.b3) if a is not None) for l in prog.layers)
    out = {"compulsory_bytes": round(comp), "hbm_bytes": None, "ratio": None, "source": None}
    if not tail:
        out["source"] = "no committed PMC pass for this configuration"
        return out
    for f in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_" + tail)), reverse=True):
        try:
            with open(os.path.join(ROOT, "profiles", f)) as fh:
                j = json.load(fh)
        except (OSError, ValueError):
            continue
        ps = j.get("per_step")
        if j.get("csrc_sha256") != csrc_digest() or not ps:
            continue
        ffac = 0.5
        t = ps["fetch_kb"] * 1024.0 / ffac + ps["write_kb"] * 1024.0
        out.update(hbm_bytes=round(t), ratio=round(t / comp, 2),
                   source=f"profiles/{f} (csrc_sha256 {j['csrc_sha256']}): sum over the {ps['dispatches_per_step']} dispatches of a "
                          f"step (mean of {ps['steps']} eager one-stream steps) of FETCH_SIZE / {ffac} + WRITE_SIZE")
        return out
    out["source"] = f"no PMC pass under profiles/ was taken on these kernel sources (csrc_sha256 {csrc_digest()})"
    return out


def measure_predict(args, model_name, B, seg, dev, rank, world, gather, min_seconds, max_blocks=60):
    """One configuration of the hot path: build the workload, time per-layer durations eagerly (HIP events around every
    launch), then time BLOCKS of exactly args.steps steps each -- every block bracketed by barrier +
    torch.cuda.synchronize() on both sides, max over ranks -- until >= min_seconds of timed steps (the first block alone
    is ~40 ms at the headline configuration: too short against the DVFS ramp).  Returns the JSON fields."""
    import torch.distributed as dist
    from yololite_amd import _lib, dist as ydist
    S = args.img
    wl = build_workload(model_name, S, B, seed=(args.seed if args.seed >= 0 else 1), seg=bool(seg), dev=dev, rank=rank,
                        stress=bool(args.stress),
                        fuse_dw=(args.fuse_dw if args.fuse_dw in ("auto", "dw3") else bool(int(args.fuse_dw))),
                        fuse_stem=bool(args.fuse_stem), fuse_uib=bool(args.fuse_uib))
    meta, sd, ctx, prog, x = wl["meta"], wl["sd"], wl["ctx"], wl["prog"], wl["x"]
    seed = wl["seed"]
    if args.tile_m:
        ctx.set_option("tile_m", args.tile_m)
    K = max(1, int(args.in_flight))                          # batches in flight (serving.ServingPipeline lanes)
    streams = args.streams if args.streams > 0 else (2 if K == 1 else 1)
    ctx.set_option("streams", streams)
    if args.bf16:
        ctx.set_option("mfma_bf16", 1)
    if args.f16:
        ctx.set_option("mfma_f16", 1)
    if args.store_f16:
        ctx.set_option("store_f16", 1)
    ctx.set_option("lanes", args.lanes)
    ctx.set_option("fuse_decode", args.fuse_decode)
    ctx.set_option("batch_levels", args.batch_levels)
    ctx.set_option("hybrid", args.hybrid)
    ctx.set_option("nms_groups", args.nms_groups)
    ctx.set_option("winograd", args.winograd)
    for kv in args.opt:                                      # developer A/B: any other context option, name=value
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    max_out = MAX_OUT                                    # packed result rows per image
    # One set of buffers per LANE (batch in flight): its own resident input batch (different images per lane), result
    # rows, gather slots (equal shards: yl_predict writes into the gather buffer, one collective, no pack/unpack kernels)
    # and mask arena -- what a double-buffered serving loop holds.
    xs = [x] + [synth_images(B, S, seed=1234 + rank + 1000 * k).to(dev) for k in range(1, K)]
    gats = [ydist.DetGatherer(B, max_out, dev) for _ in range(K)] if gather else None
    lane_dets = [None if gather else torch.empty((B, max_out, 6), device=dev, dtype=torch.float32) for _ in range(K)]
    lane_counts = [None if gather else torch.empty((B,), device=dev, dtype=torch.int32) for _ in range(K)]
    lane_arena = [(torch.empty((B * max_out * S * ((S + 31) // 32) * 4,), device=dev, dtype=torch.uint8) if seg else None)
                  for _ in range(K)]
    gat = gats[0] if gats else None
    dets, counts = lane_dets[0], lane_counts[0]

    def lane_work(c, k):
        """ONE step = the complete hot path over one batch on context c (lane k)"""
        if seg:
            _, _, idx = c.predict(xs[k], _lib.POST_MAIN, args.conf, args.iou, per_class_cap=300, max_out=max_out,
                                  out=(lane_dets[k], lane_counts[k]), want_idx=True)
            # image-resolution (640 x 640) masks, bit-packed rows, into a fixed-capacity arena: asynchronous like the
            # detections themselves (no host read of the counts inside the step)
            c.masks_image(lane_dets[k], lane_counts[k], idx, packed=True, arena=lane_arena[k])
            return lane_dets[k], lane_counts[k]
        if gats is not None:     # results go straight into the gather slot; its all-gather overlaps the following steps
            c.predict(xs[k], _lib.POST_MAIN, args.conf, args.iou, per_class_cap=300, max_out=max_out,
                      out=(gats[k].dets, gats[k].counts))
            return gats[k].gather()
        c.predict(xs[k], _lib.POST_MAIN, args.conf, args.iou, per_class_cap=300, max_out=max_out,
                  out=(lane_dets[k], lane_counts[k]))
        return lane_dets[k], lane_counts[k]

    def step():                  # plain call on the model's own context (per-layer timing phase, serial comparison)
        return lane_work(ctx, 0)

    # ---- per-layer durations (HIP events on the launch stream), eager launches
    ctx.set_option("graph", 0)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    reps = args.layer_reps
    if reps > 0:
        for _ in range(3):                                   # the eager timed path itself, untimed: clocks / caches settle
            ctx.forward(x, timed=True)
        lay = np.median(np.stack([np.asarray(ctx.forward(x, timed=True)[1]) for _ in range(reps)]), axis=0)   # median: a
        # host hiccup between two eager launches must not crown a 30 us layer "dominant kernel"
    else:                                                    # counter passes (tools/profile_round.sh): predict steps only
        lay = np.full(len(prog.layers), 1e-3)

    ctx.set_option("graph", args.graph)
    # ---- the serial loop (ONE batch in flight: calls back to back on one context, two chunk streams) next to the
    # pipelined one, same process, same clocks: reported as `one_batch_in_flight` so that the gain is visible
    serial = None
    if K > 1:
        # same protocol as the headline below: blocks of exactly args.steps steps, barrier + synchronize on
        # both sides, max over ranks, repeated to >= min_seconds, the MEDIAN block reported; under --gpus N the exchange
        # of every step is inside the block as well
        ctx.set_option("streams", args.streams if args.streams > 0 else 2)
        for _ in range(max(args.warmup, 1)):
            step()
        if gat is not None:
            gat.flush()
        torch.cuda.synchronize()

        def serial_block():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            if gat is not None:
                gat.flush()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t.item())
            return el
        s_els = [serial_block()]
        for _ in range(int(min(max_blocks - 1, max(0, np.ceil((min_seconds - s_els[0]) / max(s_els[0], 1e-6)))))):
            s_els.append(serial_block())
        s_med = float(np.sort(s_els)[len(s_els) // 2])
        serial = {"value": round(world * B * args.steps / s_med, 1), "unit": "images/sec",
                  "ms_per_step": round(s_med / args.steps * 1e3, 4),
                  "blocks": len(s_els), "steps_each": args.steps, "seconds_timed": round(float(np.sum(s_els)), 3),
                  "streams": int(ctx.get_option("streams")),
                  "what": "the round-1..4 loop under the headline's protocol (median block): the same steps issued back to "
                          "back on ONE context (every call joins its chunk streams before the next call starts)"}
        ctx.set_option("streams", streams)
    from yololite_amd.serving import ServingPipeline
    pipe = ServingPipeline(ctx, lanes=K, streams_per_lane=streams, graph=bool(args.graph), timing=True)
    for _ in range(max(args.warmup, 1) * K):
        pipe.run(lane_work)
    pipe.flush()
    for g in (gats or []):
        g.flush()

    def block():
        """exactly args.steps steps between barrier + synchronize on both sides; max over ranks.  Step i runs on lane
        i % K; every step's work -- and every exchange -- has completed inside the timed region."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        pipe.events.clear()
        ev0 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for i in range(args.steps):
            pipe.run(lane_work)
        pipe.flush()
        for g in (gats or []):
            g.flush()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el_local = time.perf_counter() - t0
        el = el_local
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        done = np.sort([ev0.elapsed_time(e1) for _, e1 in pipe.events])
        # throughput-equivalent time per batch: with K lanes the completions arrive in groups of K (a median of the
        # intervals between CONSECUTIVE completions picks the short gap inside a group -- VERDICT r05 weak #4), so the
        # interval is taken over windows of K consecutive completions and divided by K; K = 1: the plain difference.
        # Second list: per-batch latency (start-to-done on its lane)
        return el, el_local, list((done[K:] - done[:-K]) / K), [e0.elapsed_time(e1) for e0, e1 in pipe.events]

    els, step_ms, el_locals, lat_ms = [], [], [], []
    el, ell, sm, lm = block()
    els.append(el); el_locals.append(ell); step_ms += sm; lat_ms += lm
    # number of further blocks: the same on every rank (derived from the all-reduced first block)
    more = int(min(max_blocks - 1, max(0, np.ceil((min_seconds - el) / max(el, 1e-6)))))
    for _ in range(more):
        el, ell, sm, lm = block()
        els.append(el); el_locals.append(ell); step_ms += sm; lat_ms += lm
    els = np.asarray(els)
    rates = world * B * args.steps / els
    k_med = int(np.argsort(els)[len(els) // 2])          # the median block: value and ms_per_step come from ONE block
    if gat is not None:
        counts = gat.flush()[1][rank if world > 1 else 0]
        drows = gat.flush()[0][rank if world > 1 else 0]
    else:
        drows = dets
    ndet = float(counts.float().mean().item())
    dropped = int((counts.to(torch.int64) - max_out).clamp(min=0).sum().item())
    cn_host = counts.cpu().numpy()
    ncls = float(np.mean([len(np.unique(drows[b, :min(int(cn_host[b]), max_out), 5].cpu().numpy())) for b in range(min(B, 8))]))
    # the workload must exercise NMS: a synthetic model without detections would time an idle post-processing;
    # and the result must be the reference's result for this input: nothing dropped by the packed-row capacity
    assert ndet >= 50.0 or os.environ.get("YL_BENCH_ALLOW_EMPTY") == "1", \
        f"benchmark workload produced {ndet:.1f} detections / image at conf {args.conf}: pick another --seed"
    assert args.stress or dropped == 0, f"{dropped} detections dropped by max_out={max_out}"
    extra = {}
    if world > 1 or gat is not None:
        extra["rccl_ranks"] = dist.get_world_size() if dist.is_initialized() else 1
        if dist.is_initialized():
            t = torch.tensor([el_locals[k_med] / args.steps * 1e3], device=dev, dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
            dist.all_gather(allt, t)
            extra["ms_per_step_per_rank"] = [round(float(v.item()), 4) for v in allt]
            # the exchange alone: synchronous all-gathers of the packed [dets | counts] row on an idle GPU
            loc, out_ = gat._loc[0], gat._out[0].view(-1)
            for _ in range(3):
                dist.all_gather_into_tensor(out_, loc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                dist.all_gather_into_tensor(out_, loc)
            e1.record()
            torch.cuda.synchronize()
            extra["allgather_ms"] = round(e0.elapsed_time(e1) / 20.0, 4)
            extra["allgather_bytes_per_rank"] = int(loc.numel() * 4)
    if rank != 0:
        return None

    wl_set = wino_layers(prog, args.winograd)

    def roofline_of(k):
        """roofline object of fused layer k: executed FLOPs (Winograd layers: 16/36 of the direct MACs) or algorithmic
        bytes per launch / median eager launch duration against the roof its arithmetic intensity puts it under"""
        L = prog.layers[k]
        flops = 2.0 * L.macs * B
        fam = kernel_family(L, k in wl_set, tuple(prog.slots[L.out_slot][:2]), B)
        wino = "yl_conv_wino" in fam
        flops_direct = flops
        if wino:                  # Winograd F(2x2,3x3) executes 16 multiplications where the direct conv has 36
            flops = flops * 16.0 / 36.0
        byts = float(L.bytes_in + L.bytes_out) * B
        lowp = bool(args.bf16 or args.f16 or args.store_f16)
        if args.store_f16:        # fp16 activation tensors: half the bytes, except the fp32 network input of an entry layer
            entry = L.op in (0, 3, 9)                                      # OP_STEM, OP_STEMBLOCK, OP_NHWC4
            byts = float((L.bytes_in if entry else 0.5 * L.bytes_in) + 0.5 * L.bytes_out) * B
        # reduced-precision modes run on the 16-bit matrix pipe: dense peak 2.5 PFLOP/s (MI355X_MICROARCH.md), ridge 312 FLOP/B
        peak_tf = PEAK_16BIT_MFMA_TFLOPS if lowp else PEAK_FP32_MFMA_TFLOPS
        ai = flops / byts
        dur = lay[k] * 1e-3
        if ai >= peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9):
            roof = {"bound": "mfma", "achieved": round(flops / dur / 1e12, 3), "peak": peak_tf, "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": round(byts / dur / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        roof["traffic"], roof["traffic_source"] = pmc_traffic(model_name, seg, B, fam, L.op == 3)
        roof["kernel"] = f"layer {k} {L.name} ({fam}, cin={L.cin} cout={L.cout} k={L.k} dw={L.dw_k})"
        if wino:
            roof["direct_conv_equivalent_tflops"] = round(flops_direct / dur / 1e12, 3)
        roof["avg_launch_ms"] = round(float(lay[k]), 4)
        roof["share_of_forward"] = round(float(lay[k] / lay.sum()), 4)
        roof["algorithmic_flops_per_launch"] = flops
        roof["algorithmic_bytes_per_launch"] = byts
        return roof

    # dominant kernel = the fused layer with the largest median duration; worst = the layer furthest below its roof among
    # those that take >= 3 % of the forward pass
    k = int(np.argmax(lay))
    L = prog.layers[k]
    roof = roofline_of(k)
    # (head-output layers are left out: under yl_predict they do not run as the launches the eager per-layer pass times --
    # detectors: fused with the trunk and the decode (yl_conv_dpp_kernel) or decode in the epilogue; seg models: the split
    # det-rows + mask-coefficient launches -- VERDICT r04 item 6)
    heavy = [i for i in range(len(lay)) if lay[i] >= 0.03 * lay.sum() and prog.layers[i].macs > 0 and prog.layers[i].head_level < 0]
    roof_worst = min((roofline_of(i) for i in heavy), key=lambda r: r["frac"]) if heavy else None
    if roof_worst is not None:
        roof_worst["candidates"] = "fused layers with >= 3 % of the eager forward pass, head-output layers excluded (yl_predict runs them in another form)"
    # executed multiplications of the whole forward: Winograd layers count 16/36 of their direct-conv MACs
    net_macs = sum(l.macs * (16.0 / 36.0 if i in wl_set else 1.0) for i, l in enumerate(prog.layers))
    net_flops = 2.0 * net_macs * B
    fwd_ms = float(lay.sum())
    # p50 per batch = median over windows of K consecutive completions / K (see block()); it is the per-rank
    # throughput-equivalent batch time, so p50_ms_per_frame x (value / n_gpus) must be 1000 ms: held to 5 %
    p50_batch = float(np.median(step_ms))
    ident = p50_batch / B * float(rates[k_med]) / world / 1e3
    # (held when the block is long enough for the median to mean something: with 3-step blocks every window touches the
    # pipeline fill at the block's start)
    assert 0.95 <= ident <= 1.05 or args.steps < 10 or len(step_ms) < 30 or os.environ.get("YL_BENCH_NO_P50_ASSERT") == "1", \
        f"p50_ms_per_frame {p50_batch / B:.5f} x images/s per GPU {float(rates[k_med]) / world:.1f} = {ident:.3f} s, expected 1 +- 5 %"
    out = {
        "metric": "images/sec", "value": round(float(rates[k_med]), 1), "unit": "images/sec", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(els[k_med]) / args.steps * 1e3, 4),
        "p50_ms_per_frame": round(p50_batch / B, 5),
        "p50_ms_per_batch": round(p50_batch, 4),
        "p50_identity": round(p50_batch / B * float(rates[k_med]) / world / 1e3, 4),
        "p50_batch_latency_ms": round(float(np.median(lat_ms)), 4),
        "in_flight": {"batches": K, "streams_per_context": streams,
                      "what": "serving.ServingPipeline: step i = the complete yl_predict of one batch on context i % K "
                              "(yl_clone: shared weights, own arenas / graphs), each lane on its own HIP stream, its own "
                              "resident input batch and output rows; p50_ms_per_batch = median over windows of K consecutive "
                              "completions of (interval / K) = throughput-equivalent time per batch (p50_ms_per_frame = that / B; "
                              "p50_identity = p50_ms_per_frame x images/s per GPU / 1000, asserted 1 +- 5 %), p50_batch_latency_ms = median start-to-done time of a "
                              "batch on its lane (HIP events)"},
        "one_batch_in_flight": serial,
        "blocks": {"n": int(len(els)), "steps_each": args.steps, "seconds_timed": round(float(els.sum()), 3),
                   "images_per_sec_min": round(float(rates.min()), 1), "images_per_sec_median": round(float(rates[k_med]), 1),
                   "images_per_sec_max": round(float(rates.max()), 1), "images_per_sec_first": round(float(rates[0]), 1)},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": ("f16 operands, f16 activation tensors in HBM / f32 accumulate, f32 weights, levels and prototypes (reduced-precision mode, not the headline)"
                  if args.store_f16 else
                  ("bf16" if args.bf16 else "f16") + " operands / f32 accumulate and storage (reduced-precision mode, not the headline)")
                 if (args.bf16 or args.f16 or args.store_f16) else "f32",
        "options": {"winograd": int(args.winograd),
                    "winograd_layers": [prog.layers[i].name for i in sorted(wl_set)]},
        "data": "synthetic",
        "config": {"workload": f"{model_name} {'detector+instance-seg head' if seg else 'detector'} {S}x{S} C=80 batch={B}/GPU, forward+decode+per-class NMS{'+masks' if seg else ''} "
                               f"(conf {args.conf}, iou {args.iou}), input resident in HBM"
                               + (", + RCCL all-gather of packed dets" if world > 1 else ""),
                   "global_batch": B * world, "img_size": S, "parallelism": f"dp{world} (batch sharded, weights replicated)",
                   "hipgraph": bool(args.graph), "streams": streams, "batches_in_flight": K, "weights_seed": seed,
                   "head": "NMS stress: uncalibrated N(0,2) head noise (round-1 workload)" if args.stress else
                           "calibrated (program.calibrate_head)",
                   "mean_dets_per_image": round(ndet, 1), "mean_classes_per_image": round(ncls, 1),
                   "max_out": max_out, "dets_dropped": dropped},
        "roofline": roof,
        "roofline_worst": roof_worst,
        "traffic_per_step": traffic_per_step(model_name, seg, B, prog, S, max_out),
        "network": {"conv_gflop_per_image": round(2.0 * prog.macs / 1e9, 4), "launches": len(prog.layers),
                    "activation_mb": round(ctx.activation_bytes() / 1e6, 1),
                    "forward_ms_sum_of_layers": round(fwd_ms, 4),
                    "forward_tflops_executed": round(net_flops / (fwd_ms * 1e-3) / 1e12, 2),
                    "forward_frac_of_fp32_mfma_peak": round(net_flops / (fwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                    "step_tflops_executed": round(net_flops / (float(els[k_med]) / args.steps) / 1e12, 2),
                    "step_frac_of_fp32_mfma_peak": round(net_flops / (float(els[k_med]) / args.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
    }
    out.update(extra)
    if args.layers:
        print(f"# layer table: {model_name}{'+seg' if seg else ''} B={B} {S}x{S}, eager launches, median of "
              f"the timed forwards; {len(prog.layers)} launches, sum {fwd_ms:.4f} ms", file=sys.stderr)
        for i, (l, ms) in enumerate(zip(prog.layers, lay)):
            print(f"{i:3d} {l.name:42s} cin{l.cin:4d} cout{l.cout:4d} k{l.k} dw{l.dw_k} {ms:8.4f} ms "
                  f"{2.0 * l.macs * B / (ms * 1e-3) / 1e12:7.2f} TF {(l.bytes_in + l.bytes_out) * B / (ms * 1e-3) / 1e9:8.1f} GB/s",
                  file=sys.stderr)
    out["_cpu"] = (meta, sd)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--model", default="edge_n")
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--conf", type=float, default=0.4)
    ap.add_argument("--iou", type=float, default=0.5)
    ap.add_argument("--graph", type=int, default=1, help="replay the forward from a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fuse-dw", default="auto", help="auto | 1 | 0: fuse depthwise convs into the following 1x1 conv")
    ap.add_argument("--fuse-stem", type=int, default=1, help="fused stem+blocks.0 entry kernel")
    ap.add_argument("--fuse-uib", type=int, default=0, help="whole inverted-residual blocks as one launch")
    ap.add_argument("--seg", type=int, default=0, help="add the build-defined instance-seg branch (BASELINE config 4)")
    ap.add_argument("--streams", type=int, default=0, help="internal streams a context splits its batch over (0 = auto: 2 with "
                    "one batch in flight, 1 with more)")
    ap.add_argument("--in-flight", type=int, default=2, help="batches in flight: step i runs on context i %% K of a "
                    "serving.ServingPipeline (1 = calls back to back on one context: the loop of rounds 1-4)")
    ap.add_argument("--tile-m", type=int, default=0, help="conv M-tile hint (0 auto, 1/2 force m-tiles per wave)")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer timing table (stderr)")
    ap.add_argument("--seed", type=int, default=-1, help="synthetic weight seed (-1: 1).  The detection head is calibrated "
                    "(program.calibrate_head), so every seed detects")
    ap.add_argument("--stress", type=int, default=0, help="1: round-1 NMS stress workload instead (uncalibrated head noise, "
                    "~2000 survivors in 10 classes; rows beyond max_out are dropped) -- labelled in the JSON")
    ap.add_argument("--nms-groups", type=int, default=0, help="NMS workgroups per image (0 = auto: 1 at detector thresholds, 4 at conf < 0.05)")
    ap.add_argument("--hybrid", type=int, default=0, help="full-batch launches for the high-resolution layers, chunks only for the low-resolution run")
    ap.add_argument("--batch-levels", type=int, default=1, help="smooth / head layers of all pyramid levels as one launch")
    ap.add_argument("--fuse-decode", type=int, default=1, help="decode inside the head-output conv epilogue")
    ap.add_argument("--opt", action="append", default=[], help="extra context option name=value (developer A/B), repeatable")
    ap.add_argument("--lanes", type=int, default=0, help="side-stream lane for the coarse-level neck/head layers")
    ap.add_argument("--bf16", type=int, default=0, help="1: bf16-MFMA compute mode (f4; NOT the headline: reduced precision)")
    ap.add_argument("--f16", type=int, default=0, help="1: fp16-MFMA compute mode (the reference's fp16 autocast; NOT the headline)")
    ap.add_argument("--store-f16", type=int, default=0, help="1: fp16 operands AND fp16 activation tensors in HBM (option store_f16: the "
                    "storage side of the reference's fp16 autocast; NOT the headline)")
    ap.add_argument("--winograd", type=int, default=1, help="dense 3x3 stride-1 convs as Winograd F(2x2,3x3), 2.25x fewer MACs: "
                    "1 (library default) = every eligible layer, 2 = only the >= 64-channel ones on the largest grid (the finest "
                    "level's smooth block), 0 = direct convolution everywhere.  Score error vs the oracle measured equal for "
                    "all three (profiles/r04_winograd_margin*.json)")
    ap.add_argument("--workload", default="predict", help="predict (headline) | eval (evaluate-path consumers, f3) | track (tracker bank, f4)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="repeat the block of --steps timed steps until this much "
                    "step time has been measured (value = the median block)")
    ap.add_argument("--layer-reps", type=int, default=15, help="eager per-layer timing passes (HIP events around every launch); 0 = "
                    "none (counter passes: only yl_predict steps dispatch kernels; roofline fields are then meaningless)")
    ap.add_argument("--other-configs", type=int, default=-1, help="append BASELINE configs 3 and 4 (yololite_m B=32, edge_m+seg "
                    "B=32) as `other_configs`; -1 = only for the default single-GPU headline run")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    # one rank per GPU: the rank count comes from the launcher (torch.distributed.run); a bare `python bench.py --gpus 8`
    # would measure ONE GPU under an 8-GPU label
    assert args.gpus == world, (f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 as `python -m torch.distributed.run "
                                f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`")
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback exists)"
    torch.cuda.set_device(local)
    if args.workload in ("eval", "track"):
        import yololite_amd  # noqa: F401
        if rank == 0:
            (bench_eval_consumers if args.workload == "eval" else bench_tracker)(args)
        return
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    force_coll = os.environ.get("YL_BENCH_FORCE_COLLECTIVE") == "1"      # test hook: collective path at world 1
    init_only = os.environ.get("YL_BENCH_FORCE_COLLECTIVE") == "2"       # probe: process group initialised, no exchange
    if world > 1 or force_coll or init_only:
        dist.init_process_group("nccl", device_id=dev)
    import yololite_amd as ya  # noqa: F401

    out = measure_predict(args, args.model, args.batch, args.seg, dev, rank, world, gather=(world > 1 or force_coll),
                          min_seconds=args.min_seconds)
    headline = (args.model == "edge_n" and args.batch == 64 and not args.seg and not args.bf16 and not args.f16 and not args.store_f16 and args.winograd == 1
                and not args.stress and args.img == 640)
    want_other = args.other_configs == 1 or (args.other_configs == -1 and headline and world == 1 and not force_coll)
    if rank == 0:
        meta, sd = out.pop("_cpu")
        if want_other:
            # BASELINE configs 3 and 4 (and the published efficientnetv2 yololite_m) through the same harness, library
            # defaults, each a few seconds
            others = {}
            for name, (m, b, sg) in {"yololite_m_b32": ("yololite_m", 32, 0), "edge_m_seg_b32": ("edge_m", 32, 1),
                                     "yololite_m_v2_b32": ("yololite_m_v2", 32, 0)}.items():
                o = measure_predict(args, m, b, sg, dev, 0, 1, gather=False, min_seconds=min(args.min_seconds, 0.5), max_blocks=8)
                o.pop("_cpu")
                others[name] = {k: o[k] for k in ("value", "unit", "steps", "ms_per_step", "p50_ms_per_frame", "blocks", "dtype",
                                                  "options", "config", "roofline", "roofline_worst", "traffic_per_step", "network")}
                torch.cuda.empty_cache()
            out["other_configs"] = others
        if world == 1 and not args.no_cpu_baseline:
            # the reference's published protocol is batch 1 (export/infer_onnx.py, BENCHMARK.md:336-339): GPU side first,
            # then the CPU baseline (which carries the same flow as batch1_infer_flow)
            out["gpu_batch1_flow"] = gpu_batch1_flow(meta, sd, args.img, args.conf, args.iou, dev)
            out["cpu_baseline"] = cpu_baseline(meta, sd, args.img, args.conf, args.iou)
    if world > 1 or force_coll or init_only:
        dist.destroy_process_group()
    if rank == 0:
        # last thing on stdout (RCCL prints its version banner there), flushed
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
