#!/usr/bin/env python3
"""Developer aid: concurrency picture of the pipelined serving loop from a rocprofv3 --kernel-trace CSV of bench.py
(two batches in flight on two contexts / streams).  For the last `win` ms of the trace: wall span, union-busy time, time with
>= 2 kernels in flight, and per kernel name: calls, mean duration, share of the summed durations, mean number of OTHER
kernels in flight while it ran.  (Tracing perturbs the overlap; use for the structure, not for the absolute step time.)
    python tools/overlap_stats.py kernel_trace.csv [win_ms]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
win = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
ks = [k for k in ks if "yl_" in k[2]]
tend = max(k[1] for k in ks)
# skip the tail (flush) and take a window before it
t1 = tend - int(2e6)
t0 = t1 - int(win * 1e6)
sel = [k for k in ks if k[0] >= t0 and k[1] <= t1]
ev = []
for s, e, n, q in sel:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = two = 0; depth = 0; last = t0
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: two += t - last
    depth += d; last = t
print(f"window {win:.1f} ms: kernels {len(sel)}, union busy {busy / 1e6:.3f} ms ({busy / (t1 - t0):.1%}), >=2 in flight {two / 1e6:.3f} ms ({two / (t1 - t0):.1%}), queues {sorted(set(k[3] for k in sel))}")
st = collections.defaultdict(lambda: [0, 0, 0.0])
tot = 0
import bisect
starts = [k[0] for k in sel]
for s, e, n, q in sel:
    ov = 0
    for s2, e2, n2, q2 in sel[max(0, bisect.bisect_left(starts, s) - 40): bisect.bisect_right(starts, e)]:
        if (s2, e2, n2, q2) != (s, e, n, q) and s2 < e and e2 > s:
            ov += min(e, e2) - max(s, s2)
    a = st[n.replace("void ", "").split("(")[0]]
    a[0] += 1; a[1] += e - s; a[2] += ov
    tot += e - s
nst = sum(1 for k in sel if "stemblock" in k[2])
print(f"stem-block launches (= batches) in the window: {nst}; window / batch = {(t1 - t0) / 1e6 / max(nst, 1):.4f} ms; sum of durations / batch = {tot / 1e6 / max(nst, 1):.4f} ms")
for n, (c, d, ov) in sorted(st.items(), key=lambda kv: -kv[1][1]):
    print(f"{n[:58]:58s} n {c:5d} per-batch {c / max(nst, 1):5.1f}  mean {d / c / 1e3:8.1f} us  share {d / tot:6.1%}  others-in-flight {ov / d:5.2f}")
