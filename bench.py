#!/usr/bin/env python3
"""Headline benchmark: images/sec + p50 ms/frame of the YoloLite inference hot path
(backbone -> FPN -> heads -> decode -> class-wise NMS), edge_n 640x640 batch 64 per GPU
(BASELINE.json configs[1]; N>1: the same per GPU = configs[4], weak scaling, one RCCL all-gather of
the packed detections per step).

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path over one batch that is already resident in HBM:
yl_predict (42 fused conv launches + decode + NMS) [+ all-gather].  Synthetic data, seeded
synthetic weights (no checkpoints exist in this environment).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
RIDGE = PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)   # 19.7 FLOP/B


def synth_images(B, S, seed=1234):
    """uniform u8 image normalised with the ImageNet mean/std (what a letterboxed frame looks like)."""
    rng = np.random.RandomState(seed)
    u8 = rng.randint(0, 256, size=(B, S, S, 3)).astype(np.float32) / 255.0
    im = (u8 - np.array([0.485, 0.456, 0.406], np.float32)) / np.array([0.229, 0.224, 0.225], np.float32)
    return torch.from_numpy(np.ascontiguousarray(im.transpose(0, 3, 1, 2)))


def cpu_baseline(meta, sd, S, conf, iou, budget_s=12.0, bs=8):
    """The oracle (CPU restatement of the reference path, PyTorch-CPU fp32 + numpy NMS) timed on this
    host's cores on a bounded sample of the same workload.  Checker code used only as a baseline."""
    from oracle import model as omodel, postproc as opost
    m = omodel.build_from_meta(meta).eval()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    x = synth_images(bs, S)
    # thread count: measured on the GPU box (2 x EPYC 9575F, 256 logical CPUs): 8-16 threads are the
    # fastest for this small-channel network (158 ms / 8 images), 128 threads are 5x slower
    n = min(16, os.cpu_count() or 1)
    torch.set_num_threads(n)
    with torch.no_grad():
        opost.pipeline_main(m(x), S, conf, iou)          # warm-up
        t0 = time.perf_counter()
        done = 0
        while True:
            opost.pipeline_main(m(x), S, conf, iou)
            done += bs
            el = time.perf_counter() - t0
            if el >= budget_s or done >= 40 * bs:
                break
    return {"value": round(done / el, 2), "unit": "images/sec", "cores": n, "kind": "port",
            "sample": f"{done} images (batches of {bs}) of the same edge_n 640x640 workload, forward+decode+NMS, "
                      f"{el:.1f} s wall, torch {n} threads of {os.cpu_count()} logical CPUs"}


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
    ap.add_argument("--streams", type=int, default=2, help="internal streams the batch is split over")
    ap.add_argument("--tile-m", type=int, default=0, help="conv M-tile hint (0 auto, 1/2 force m-tiles per wave)")
    ap.add_argument("--layers", action="store_true", help="also print the per-layer timing table (stderr)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback exists)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    force_coll = os.environ.get("YL_BENCH_FORCE_COLLECTIVE") == "1"      # test hook: collective path at world 1
    if world > 1 or force_coll:
        dist.init_process_group("nccl", device_id=dev)

    import yololite_amd as ya
    from yololite_amd import _lib, dist as ydist
    from yololite_amd.program import synth_state_dict, zoo_meta

    B, S = args.batch, args.img
    meta = zoo_meta(args.model, 80, S, seg=bool(args.seg))
    sd = synth_state_dict(meta, seed=0, head_noise=2.0)
    model = ya.build_model_from_meta(meta, fuse_dw=(args.fuse_dw if args.fuse_dw in ("auto", "dw3") else bool(int(args.fuse_dw))),
                                     fuse_stem=bool(args.fuse_stem), fuse_uib=bool(args.fuse_uib))
    model.load_state_dict(sd)
    model.to(dev)
    ctx, prog = model._ctx_for(S), model.program
    if args.tile_m:
        ctx.set_option("tile_m", args.tile_m)
    ctx.set_option("streams", args.streams)
    x = synth_images(B, S, seed=1234 + rank).to(dev)
    max_out = 300                                        # packed result rows per image
    dets = torch.empty((B, max_out, 6), device=dev, dtype=torch.float32)
    counts = torch.empty((B,), device=dev, dtype=torch.int32)

    def step():
        if args.seg:
            _, _, idx = ctx.predict(x, _lib.POST_MAIN, args.conf, args.iou, per_class_cap=300, max_out=max_out,
                                    out=(dets, counts), want_idx=True)
            ctx.masks(counts, idx, max_out)
            return dets, counts
        ctx.predict(x, _lib.POST_MAIN, args.conf, args.iou, per_class_cap=300, max_out=max_out, out=(dets, counts))
        if world > 1 or force_coll:
            return ydist.allgather_dets(dets, counts, B * world, force=force_coll)
        return dets, counts

    # ---- per-layer durations (HIP events on the launch stream), eager launches
    ctx.set_option("graph", 0)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    reps = 5
    lay = np.zeros(len(prog.layers))
    for _ in range(reps):
        _, ms = ctx.forward(x, timed=True)
        lay += np.asarray(ms)
    lay /= reps

    ctx.set_option("graph", args.graph)
    for _ in range(max(args.warmup, 1)):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    step_ms = np.asarray([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)])
    ndet = float(counts.float().mean().item())

    if rank == 0:
        value = world * B * args.steps / el
        # dominant kernel = the fused layer with the largest average duration
        k = int(np.argmax(lay))
        L = prog.layers[k]
        flops = 2.0 * L.macs * B
        byts = float(L.bytes_in + L.bytes_out) * B
        ai = flops / byts
        dur = lay[k] * 1e-3
        if ai >= RIDGE:
            roof = {"bound": "mfma", "achieved": round(flops / dur / 1e12, 3), "peak": PEAK_FP32_MFMA_TFLOPS,
                    "unit": "TFLOP/s"}
        else:
            roof = {"bound": "hbm", "achieved": round(byts / dur / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s"}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        roof["traffic"] = None
        roof["kernel"] = f"layer {k} {L.name} (yl_conv_mfma_kernel, cin={L.cin} cout={L.cout} k={L.k} dw={L.dw_k})" \
            if L.op == 1 else f"layer {k} {L.name}"
        roof["avg_launch_ms"] = round(float(lay[k]), 4)
        roof["algorithmic_flops_per_launch"] = flops
        roof["algorithmic_bytes_per_launch"] = byts
        net_flops = 2.0 * prog.macs * B
        fwd_ms = float(lay.sum())
        out = {
            "metric": "images/sec", "value": round(value, 1), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 4),
            "p50_ms_per_frame": round(float(np.median(step_ms)) / B, 5),
            "p50_ms_per_batch": round(float(np.median(step_ms)), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.model} {'detector+instance-seg head' if args.seg else 'detector'} 640x640 C=80 batch={B}/GPU, forward+decode+per-class NMS{'+masks' if args.seg else ''} "
                                   f"(conf {args.conf}, iou {args.iou}), input resident in HBM"
                                   + (", + RCCL all-gather of packed dets" if world > 1 else ""),
                       "global_batch": B * world, "img_size": S, "parallelism": f"dp{world} (batch sharded, weights replicated)",
                       "hipgraph": bool(args.graph), "streams": args.streams, "mean_dets_per_image": round(ndet, 1)},
            "roofline": roof,
            "network": {"conv_gflop_per_image": round(2.0 * prog.macs / 1e9, 4), "launches": len(prog.layers),
                        "forward_ms_sum_of_layers": round(fwd_ms, 4),
                        "forward_tflops": round(net_flops / (fwd_ms * 1e-3) / 1e12, 2),
                        "forward_frac_of_fp32_mfma_peak": round(net_flops / (fwd_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)},
        }
        if args.layers:
            for i, (l, ms) in enumerate(zip(prog.layers, lay)):
                print(f"{i:3d} {l.name:42s} cin{l.cin:4d} cout{l.cout:4d} k{l.k} dw{l.dw_k} {ms:8.4f} ms "
                      f"{2.0 * l.macs * B / (ms * 1e-3) / 1e12:7.2f} TF {(l.bytes_in + l.bytes_out) * B / (ms * 1e-3) / 1e9:8.1f} GB/s",
                      file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(meta, sd, S, args.conf, args.iou)
        print(json.dumps(out))
    if world > 1 or force_coll:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
