#!/usr/bin/env python3
"""print the step time and the per-kernel averages matching the given patterns from a bench.py JSON line
    python bench.py --no-cpu-baseline --no-extra > b.json; python tools/bench_kernels.py b.json gemm cl_bwd"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
pat = sys.argv[2:]
print("step", round(d["ms_per_step"], 4), "ms; with a fresh batch", round(d.get("step_incl_batch_build_ms", 0), 4), "ms")
for r in d.get("roofline_all", []):
    if not pat or any(p in r["kernel"] for p in pat):
        print(f"  {r['kernel'][:64]:64s} {r.get('avg_us', 0):8.1f} us  frac {r.get('frac', 0):.3f}  {r.get('stream', '')}")
