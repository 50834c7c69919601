#!/usr/bin/env python3
"""Probe for the hardware-queue aliasing: throughput of the two-chunk-stream executor in a host
process that is NOT bench.py -- torch imported first, optionally an RCCL process group initialised BEFORE the package
is imported -- relying only on the package's own GPU_MAX_HW_QUEUES default (yololite_amd._lib).  Prints one JSON line.

    python tools/rccl_queue_probe.py --rccl 0|1 [--batch 64] [--steps 60]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rccl", type=int, default=0)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--port", type=int, default=29761)
    a = ap.parse_args()
    import torch                                   # the host imports torch first, like any serving process
    assert not torch.cuda.is_initialized()
    import yololite_amd  # noqa: F401               (sets GPU_MAX_HW_QUEUES unless the caller exported one)
    queues = os.environ.get("GPU_MAX_HW_QUEUES")
    torch.cuda.set_device(0)
    if a.rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(a.port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        t = torch.ones(8, device="cuda")
        dist.all_reduce(t)                         # the communicator and its streams exist before the context does
        torch.cuda.synchronize()
    import bench
    from yololite_amd import _lib
    wl = bench.build_workload("edge_n", 640, a.batch, seed=1, dev="cuda:0")
    ctx, x = wl["ctx"], wl["x"]
    ctx.set_option("graph", 1)
    dets = torch.empty((a.batch, bench.MAX_OUT, 6), device="cuda", dtype=torch.float32)
    counts = torch.empty((a.batch,), device="cuda", dtype=torch.int32)

    def step():
        ctx.predict(x, _lib.POST_MAIN, 0.4, 0.5, per_class_cap=300, max_out=bench.MAX_OUT, out=(dets, counts))
    for _ in range(30):
        step()
    rates = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        rates.append(a.batch * a.steps / (time.perf_counter() - t0))
    rates.sort()
    print(json.dumps({"rccl": a.rccl, "GPU_MAX_HW_QUEUES": queues, "images_per_sec": round(rates[len(rates) // 2], 1),
                      "min": round(rates[0], 1), "max": round(rates[-1], 1)}), flush=True)
    if a.rccl:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
