"""Serving loop with K batches in flight (`ServingPipeline`).

The reference serves one image at a time from one model object (tools/infer.py:435-516) or one batch at a time in
evaluate (scripts/helpers/evaluate.py:421-429); nothing overlaps.  On MI355X a single yl_predict call of edge_n at
B = 64 is a dependent chain of ~30 launches whose 40x40 / 20x20 stages leave most of the 256 CUs idle, and the call ends
with a join -- back-to-back calls on one context therefore never overlap the latency-bound tail of batch i with the
compute-bound head of batch i+1.  `ServingPipeline` keeps `lanes` contexts of the same model (yl_clone: shared weights,
own arenas / workspaces / graphs) and `lanes` HIP streams; `submit(x)` runs the COMPLETE yl_predict of that batch on lane
i % lanes and returns the lane's previous result (like dist.DetGatherer.gather()); `flush()` drains.  Every batch is
processed by exactly the kernels of a plain call, so results are bitwise those of `ctx.predict`
(tests/test_gpu_parity.py::test_serving_pipeline_*; at the benchmark's full sizes and its exact schedule -- 2 lanes x 1 stream x
hipGraph replay -- tests/test_bench_config.py::test_bench_schedule_two_lanes_graph_full_size_parity).

Measured (edge_n 640x640 B=64, hipGraph replay; tools/pipeline_probe.py, output kept as profiles/r06_pipeline_probe.txt):
one context, two chunk streams, calls back to back against 2 / 3 / 4 lanes x 1 stream and 2 lanes x 2 chunk streams --
2 lanes of un-chunked launches are the optimum on every benchmarked configuration.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import ctypes as C

import torch

from . import _lib
from .model import HipContext


class ServingPipeline:
    def __init__(self, ctx: HipContext, lanes: int = 2, streams_per_lane: int = 1, graph: bool = True, timing: bool = False):
        """ctx: the model's context (model._ctx_for(img_size)).  EVERY lane is a clone of it (yl_clone: the packed weights are
        shared, options copied as they are now), lane 0 included: the caller's context keeps its own `streams` / `graph`
        settings and arenas, so plain ctx.predict calls behave the same before, during and after the pipeline's life
        (ADVICE r05).  streams_per_lane: internal chunk streams of every lane ("streams" option; 1 = un-chunked full-batch
        launches, measured best with >= 2 lanes).  timing: keep (start, done) HIP-event pairs of every submission (`events`)."""
        if lanes < 1:
            raise ValueError("lanes >= 1")
        self.lanes = int(lanes)
        self.ctxs: List[HipContext] = [ctx.clone() for _ in range(self.lanes)]
        for c in self.ctxs:
            c.set_option("streams", int(streams_per_lane))
            c.set_option("graph", 1 if graph else 0)
        self.device = ctx.device
        self.streams = self._pick_streams(ctx, self.lanes)
        self._res: List[Optional[tuple]] = [None] * self.lanes
        self._done = [torch.cuda.Event() for _ in range(self.lanes)]
        self._i = 0
        self.timing = bool(timing)
        self.events: List[tuple] = []

    @staticmethod
    def _pick_streams(ctx: HipContext, n: int):
        """n HIP streams whose kernels really run beside each other and beside the submitting (current) stream.  ROCm gives a
        stream its hardware queue at creation, round-robin over every stream the process has created; two lanes on one queue
        serialise (measured: 46.3 k -> 39 k images/s on edge_n B=64, depending only on the number of streams created before
        the pipeline).  yl_streams_overlap measures it; candidates that alias a chosen stream are passed over (they stay in
        torch's stream pool)."""
        dev = ctx.device
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        cur = torch.cuda.current_stream(dev)

        def overlap(a, b) -> bool:
            r = C.c_int32(0)
            _lib.check(ctx.lib.yl_streams_overlap(idx, C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream), C.byref(r)),
                       ctx.handle, "yl_streams_overlap")
            return bool(r.value)

        picked: List[torch.cuda.Stream] = []
        last = None
        for _ in range(24):
            s = torch.cuda.Stream(device=dev)
            last = s
            if s.cuda_stream == cur.cuda_stream or any(s.cuda_stream == q.cuda_stream for q in picked):
                continue
            if all(overlap(s, q) for q in picked + [cur]):
                picked.append(s)
                if len(picked) == n:
                    return picked
        while len(picked) < n:                  # no clean set among the candidates: take what there is (still correct, slower)
            picked.append(last if last is not None and all(last is not q for q in picked) else torch.cuda.Stream(device=dev))
            last = None
        return picked

    def run(self, fn, inputs=()):
        """Generic form: `fn(ctx, lane)` enqueues ONE batch's work (yl_predict and whatever follows it on the same context:
        yl_masks_image, the all-gather of its packed result, ...) with the lane's stream current; `inputs` are tensors the
        work reads (ordered behind the current stream, kept alive for the lane's stream).  Returns the lane's previous
        result (whatever its fn returned), ready on the current stream, or None."""
        k = self._i % self.lanes
        self._i += 1
        cur = torch.cuda.current_stream(self.device)
        prev = self._hand_back(k, cur)
        s = self.streams[k]
        s.wait_stream(cur)        # inputs (and the caller's reads of the lane's previous outputs) are ordered before this batch
        with torch.cuda.stream(s):
            if self.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(s)
            r = fn(self.ctxs[k], k)
            if self.timing:
                e1.record(s)
                self.events.append((e0, e1))
            self._done[k].record(s)
        for t in inputs:
            t.record_stream(s)
        self._res[k] = (r,)
        return prev

    def _hand_back(self, k, cur):
        if self._res[k] is None:
            return None
        cur.wait_event(self._done[k])              # the lane's previous batch: its results become visible to the caller now
        (r,) = self._res[k]
        self._res[k] = None
        for t in (r if isinstance(r, (tuple, list)) else (r,)):
            if isinstance(t, torch.Tensor):
                t.record_stream(cur)
        return r

    def submit(self, x: torch.Tensor, mode=_lib.POST_MAIN, conf: float = 0.4, iou: float = 0.5, per_class_cap: int = 300,
               max_out: Optional[int] = None, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, **kw):
        """Enqueue one batch (x resident on the device, produced on the CURRENT stream) on the next lane: the complete
        yl_predict of that batch.  Returns the result that lane produced `lanes` submissions ago -- (dets, counts[, idx]),
        ready on the current stream -- or None.  `out`: per-call output buffers; they must stay untouched until this
        batch's result has been handed back."""
        return self.run(lambda c, k: c.predict(x, mode, conf, iou, per_class_cap=per_class_cap, max_out=max_out, out=out, **kw),
                        inputs=(x,))

    def flush(self) -> List[tuple]:
        """Wait (on the current stream) for every batch in flight; returns their results in submission order."""
        cur = torch.cuda.current_stream(self.device)
        outs = []
        n = min(self._i, self.lanes)
        for j in range(n):
            r = self._hand_back((self._i - n + j) % self.lanes, cur)
            if r is not None:
                outs.append(r)
        return outs
