#!/usr/bin/env python3
"""Summarise rocprofv3 outputs of tools/collect_profiles.sh (per workload: kernel stats CSV + FETCH_SIZE / WRITE_SIZE / MFMA counter CSVs)
into profiles/<tag>_<workload>_{summary.md,traffic.json,kernel_stats.csv}.

    python tools/summarize_profiles.py <dir with <workload>_kt_kernel_stats.csv, <workload>_fetch_counter_collection.csv, ...> <tag>
"""
import collections
import csv
import json
import os
import sys

src_dir, tag = sys.argv[1], sys.argv[2]
import shutil
import subprocess

try:
    COMMIT = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(__file__))).stdout.strip()
except Exception:
    COMMIT = ''


def summarize(src, wl, tag):
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    os.makedirs(out_dir, exist_ok=True)
    CALIB_BYTES = 256 * 1024 * 1024 * 4


    def short(name):
        name = name.replace("void ", "").replace("txe::", "")
        return name.split("(")[0][:70]


    lines = [f"# rocprofv3 summary `{tag}` -- workload `{wl}` (commit {COMMIT})", ""]
    ks = os.path.join(src, f"{wl}_kt_kernel_stats.csv")
    if os.path.exists(ks):
        rows = list(csv.DictReader(open(ks)))
        # which HIP stream (HSA queue) a kernel runs on, from the kernel trace: the queue with the most kernel time is the caller's
        # stream; launches on another queue run UNDER main-stream kernels (second-stream overlap) -- their durations are
        # stretched by sharing the machine and they are off the critical path
        queue_of = {}
        kt = os.path.join(src, f"{wl}_kt_kernel_trace.csv")
        if os.path.exists(kt):
            busy, per = collections.defaultdict(float), collections.defaultdict(lambda: collections.defaultdict(float))
            for r in csv.DictReader(open(kt)):
                d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                busy[r["Queue_Id"]] += d
                per[r["Kernel_Name"]][r["Queue_Id"]] += d
            main_q = max(busy, key=busy.get) if busy else None
            for k, qs in per.items():
                on_main = qs.get(main_q, 0.0)
                tot = sum(qs.values())
                queue_of[k] = "main" if on_main >= 0.999 * tot else ("2nd" if on_main <= 0.001 * tot else f"main {100 * on_main / tot:.0f}% / 2nd")
        lines += ["## kernel-trace --stats (" + ("python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline" if wl == "pgat" else f"TXE_PROF_WORKLOAD={wl} python tools/profile_workload.py") + ")", "",
                  "stream: `main` = the caller's stream (critical path); `2nd` = the library's second stream -- those launches run concurrently with",
                  "main-stream kernels, their durations are stretched by sharing the machine and do not add to the step time.", "",
                  "| kernel | calls | total ms | avg us | % | stream |", "|---|---|---|---|---|---|"]
        for r in rows[:40]:
            lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.1f} | {float(r['Percentage']):.2f} | {queue_of.get(r['Name'], '')} |")
        lines.append("")

    traffic = {}
    for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        f = os.path.join(src, f"{wl}_{kind}_counter_collection.csv")
        if not os.path.exists(f):
            continue
        per = collections.defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            per[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = r["Kernel_Name"]
        # calibration: the single big elementwise copy
        calib = [(v, d) for d, v in per.items() if "elementwise" in names[d] or "copy" in names[d].lower()]
        cv, cd = max(calib) if calib else (0.0, None)
        factor = CALIB_BYTES / (cv * 1024.0) if cv > 0 else 1.0
        agg = collections.defaultdict(lambda: [0, 0.0])
        for d, v in per.items():
            if d == cd:
                continue
            a = agg[short(names[d])]
            a[0] += 1
            a[1] += v * 1024.0
        traffic[kind] = dict(raw_unit_bytes=1024, calibration_kernel=short(names[cd]) if cd else None, calibration_counter=cv,
                             correction_factor=factor,
                             kernels={k: dict(launches=n, avg_raw_bytes=b / n, avg_corrected_bytes=b / n * factor) for k, (n, b) in agg.items()})
        lines += [f"## --pmc {cname} (tools/profile_workload.py)", "",
                  f"calibration: 1 GiB device copy reported {cv:.0f} x 1 KiB -> correction factor x{factor:.3f}", "",
                  "| kernel | launches | avg raw MB | avg corrected MB |", "|---|---|---|---|"]
        for k, (n, b) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if any(t in k for t in ("txe", "gemm", "gat", "readout", "cl_", "gcn", "gcl", "rank", "adam", "colsum", "gather", "segsum")):
                lines.append(f"| `{k}` | {n} | {b/n/1e6:.2f} | {b/n*factor/1e6:.2f} |")
        lines.append("")

    # MFMA pipe utilisation from counters: SQ_VALU_MFMA_BUSY_CYCLES (64 cycles per v_mfma_f32_32x32x2_f32, summed over the chip's
    # 1,024 SIMDs) against GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs), collected in their own pass
    f = os.path.join(src, f"{wl}_mfma_counter_collection.csv")
    if os.path.exists(f):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
        lines += ["## --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (tools/profile_workload.py)", "",
                  "utilisation = MFMA busy cycles / (GUI-active cycles per XCD x 1,024 SIMDs); the fp32 MFMA count is busy / 64", "",
                  "| kernel | launches | MFMAs per launch (M) | GUI-active cycles per XCD (k) | MFMA pipe utilisation |", "|---|---|---|---|---|"]
        for k, c in sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
            mf, gu, n = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("GRBM_GUI_ACTIVE", 0.0), cnt[k]
            if mf <= 0 or gu <= 0 or n == 0:
                continue
            lines.append(f"| `{k}` | {n} | {mf / 64 / n / 1e6:.3f} | {gu / 8 / n / 1e3:.1f} | {mf / (gu / 8 * 1024):.3f} |")
        lines.append("")

    open(os.path.join(out_dir, f"{tag}_{wl}_summary.md"), "w").write("\n".join(lines) + "\n")
    if traffic:
        traffic["collected_at_commit"] = COMMIT
        json.dump(traffic, open(os.path.join(out_dir, f"{tag}_{wl}_traffic.json"), "w"), indent=1)
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(out_dir, f"{tag}_{wl}_kernel_stats.csv"))
    print("\n".join(lines[:45]))


out_dir_made = False
for wl in ("pgat", "pgcn", "pgat2", "infer"):
    if os.path.exists(os.path.join(src_dir, f"{wl}_kt_kernel_stats.csv")) or os.path.exists(os.path.join(src_dir, f"{wl}_fetch_counter_collection.csv")):
        summarize(src_dir, wl, tag)
