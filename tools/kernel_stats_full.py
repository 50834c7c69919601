#!/usr/bin/env python3
"""Per-kernel duration statistics of the FULL-BATCH dispatches of a rocprofv3 --kernel-trace run:
the benchmark's workload builder also dispatches every kernel once or twice on the 8-image head-calibration batch, which
drags rocprofv3's own --stats averages down; here dispatches shorter than half the kernel's median are dropped, so that
bench.py's roofline.avg_launch_ms can be reproduced from the committed file.
    python tools/kernel_stats_full.py <..._kernel_trace.csv>  >  profiles/rNN_kernel_stats_full_batch_<cfg>.csv"""
import collections
import csv
import statistics
import sys

rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Name", "FullBatchCalls", "DroppedCalls", "MedianNs", "AverageNs", "MinNs", "MaxNs", "TotalNs"])
out = []
for k, v in rows.items():
    med = statistics.median(v)
    full = [x for x in v if x >= 0.5 * med]
    out.append((sum(full), [k, len(full), len(v) - len(full), int(statistics.median(full)), round(sum(full) / len(full), 1), min(full),
                            max(full), sum(full)]))
for _, r in sorted(out, key=lambda t: -t[0]):
    w.writerow(r)
