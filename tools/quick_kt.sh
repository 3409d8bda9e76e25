#!/bin/bash
# one rocprofv3 kernel-trace pass over a short bench run; prints the per-kernel stats (quick A/B view of a single kernel's duration)
#   gpurun -- 'bash tools/quick_kt.sh [pgat|pgcn|pgat2|infer] [grep pattern]'
set -u
W=${1:-pgat}
PAT=${2:-.}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/quick_kt
rm -rf "$O"; mkdir -p "$O"
export TXE_PROF_WORKLOAD=$W
if [ "$W" = pgat ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o kt -- python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > "$O/log" 2>&1
else
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O" -o kt -- python tools/profile_workload.py > "$O/log" 2>&1
fi
f=$(find "$O" -name "*kernel_stats.csv" | head -1)
python - "$f" "$PAT" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print(f'{int(r["Calls"]):5d} x {float(r["AverageNs"]) / 1e3:8.1f} us  {r["Name"][:110]}')
PY
