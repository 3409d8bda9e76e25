#!/bin/bash
# ad-hoc PMC probe of the training step:  gpurun -- 'bash tools/pmc_probe.sh "VALUBusy SALUBusy" "MemUnitBusy MemUnitStalled"'
# one rocprofv3 pass per quoted counter group over tools/profile_workload.py; prints per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/pmc_probe; rm -rf $O; mkdir -p $O
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/p$i -o p -- python tools/profile_workload.py > $O/p$i.log 2>&1
  f=$(find $O/p$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
per = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("void ", "").replace("txe::", "").split("(")[0][:48]
    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for v in per.values() for c in v})
print("kernel".ljust(50), *[n[:18].rjust(18) for n in names])
for k, v in sorted(per.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    if any(t in k for t in ("gemm", "gat_", "cl_", "adam", "reduce", "bil_", "rowdot", "nce")):
        print(k.ljust(50), *[f"{sum(v[n]) / max(len(v[n]), 1):18.3f}" for n in names])
PY
done
