#!/usr/bin/env python3
"""Timeline of ONE benchmark step from a rocprofv3 --kernel-trace CSV: per kernel start/end relative to the step start,
stream/queue, overlap statistics.   python tools/trace_step.py <kernel_trace.csv> [step_index_from_end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ev.sort()
# steps are delimited by the stem block launches of chunk 0: find stemblock kernels
stems = [i for i, e in enumerate(ev) if "stemblock" in e[2]]
nchunk = 2 if len(stems) >= 4 and (ev[stems[1]][0] - ev[stems[0]][0]) < (ev[stems[2]][0] - ev[stems[1]][0]) else 1
starts = stems[::nchunk]
i0, i1 = starts[-back - 1], starts[-back]
seg = ev[i0:i1]
t0 = seg[0][0]
tend = max(e[1] for e in seg)
print(f"step: {len(seg)} kernels, span {(tend - t0) / 1e3:.1f} us, next step starts at {(ev[i1][0] - t0) / 1e3:.1f} us")
busy = 0; cur_end = t0; sum_dur = 0
for s, e, n, q in seg:
    sum_dur += e - s
    if s > cur_end: busy += 0; gap = s - cur_end
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e
print(f"sum of kernel durations {sum_dur / 1e3:.1f} us, union busy {busy / 1e3:.1f} us, idle inside step {(tend - t0 - busy) / 1e3:.1f} us")
for s, e, n, q in seg:
    short = n.replace("void ", "").replace("(YlConvMulti)", "").replace("(YlConvP)", "")[:44]
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q}  {short}")
