#!/usr/bin/env python3
"""Developer aid: timeline of ONE steady-state step from a rocprofv3 kernel trace (csv) of bench.py.
    python tools/timeline.py trace.csv [step_index_from_end]
CAUTION: tracing changes how the two chunk streams overlap (traced: chunk 1's entry kernel starts ~0.6 ms after chunk
0's; three plan orderings derived from that picture were all slower than the unordered plan when timed WITHOUT the
tracer, round 3) -- use it for per-kernel durations at the chunk batch size, not for the overlap structure.
Prints every kernel of the step (start / duration in us relative to the step's first kernel, queue), the busy union
and the sum of durations."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
ks.sort()
# a step = from one entry kernel (stem block / stem) on a queue to the next one on the SAME queue (both queues' kernels)
ent = [i for i, k in enumerate(ks) if "stem" in k[2]]
q0 = ks[ent[0]][3]
starts = [i for i in ent if ks[i][3] == q0]
i0, i1 = starts[-back - 1], starts[-back]
step = ks[i0:i1]
t0 = step[0][0]
busy, cur_s, cur_e = 0, None, None
for s, e, n, q in step:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
for s, e, n, q in step:
    nm = n.replace("void ", "").split("(")[0][:46]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {nm}")
span = max(e for _, e, _, _ in step) - t0
print(f"kernels {len(step)}  span {span / 1e3:.1f} us  busy-union {busy / 1e3:.1f} us  sum {sum(e - s for s, e, _, _ in step) / 1e3:.1f} us  next step starts at {(ks[i1][0] - t0) / 1e3:.1f} us")
