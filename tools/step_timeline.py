#!/usr/bin/env python3
"""One training step's timeline from a rocprofv3 kernel trace (gpurun_out/tl/*kernel_trace.csv): per stream (HSA queue) the busy time,
the idle gaps between consecutive kernels on the caller's stream, and the kernels in order.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python tools/profile_workload.py ; python tools/step_timeline.py"""
import collections
import csv
import glob

import os
f = sorted(glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True))[-1]
FIRST = tuple((os.environ.get("TXE_TL_FIRST") or "gat_prepare_multi,gat_prepare_kernel").split(","))   # the step's first launch
rows = [dict(name=r["Kernel_Name"].replace("void ", "").replace("txe::", "").split("(")[0][:60], q=r["Queue_Id"], s=int(r["Start_Timestamp"]),
             e=int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r["s"])
busy = collections.defaultdict(int)
for r in rows:
    busy[r["q"]] += r["e"] - r["s"]
main = max(busy, key=busy.get)
# the last full step: from the last `gat_prepare_multi_kernel` (first launch of a step) but one to the last one
starts = [i for i, r in enumerate(rows) if r["name"].startswith(FIRST)]
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = step[0]["s"]
print(f"step wall {(rows[b]['s'] - t0) / 1e3:.1f} us, {len(step)} launches ({sum(1 for r in step if r['q'] == main)} on the caller's stream)")
prev_e = None
gap_tot = 0
for r in step:
    tag = "main" if r["q"] == main else " 2nd"
    gap = ""
    if r["q"] == main:
        if prev_e is not None:
            g = r["s"] - prev_e
            gap_tot += max(g, 0)
            gap = f"gap {g / 1e3:6.1f}"
        prev_e = r["e"]
    print(f"{(r['s'] - t0) / 1e3:8.1f} +{(r['e'] - r['s']) / 1e3:7.1f} us  {tag}  {gap:12s} {r['name']}")
print(f"idle on the caller's stream inside the step: {gap_tot / 1e3:.1f} us; its kernels: {sum(r['e'] - r['s'] for r in step if r['q'] == main) / 1e3:.1f} us")
