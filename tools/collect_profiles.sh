#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_<workload>_*: for each workload (pgat = BASELINE configs[1], pgcn = configs[4], pgat2 = configs[3]'s
# model, infer = configs[2]'s MAG-Full all-candidate inference) a kernel-trace + stats pass, then FETCH_SIZE and WRITE_SIZE in SEPARATE
# --pmc passes (TCC slots; never combined with sys/hip/hsa tracing) and a fourth pass with the MFMA-busy / GUI-active counters, all over
# tools/profile_workload.py (pgat's kernel trace is taken over the bench command itself).
#   gpurun -- 'bash tools/collect_profiles.sh r02 [workloads...]'      then, back in the build container:
#   python tools/summarize_profiles.py gpurun_out/prof_r02 r02
set -u
TAG=${1:-r02}
shift || true
WLS=${@:-pgat pgcn pgat2 infer}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/prof_$TAG
mkdir -p "$O"
for W in $WLS; do
  export TXE_PROF_WORKLOAD=$W
  if [ "$W" = pgat ]; then
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${W}_kt" -o kt -- python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > "$O/${W}_kt.log" 2>&1
  else
    rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${W}_kt" -o kt -- python tools/profile_workload.py > "$O/${W}_kt.log" 2>&1
  fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/${W}_fetch" -o fetch -- python tools/profile_workload.py > "$O/${W}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/${W}_write" -o write -- python tools/profile_workload.py > "$O/${W}_write.log" 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/${W}_mfma" -o mfma -- python tools/profile_workload.py > "$O/${W}_mfma.log" 2>&1
  for k in kt fetch write mfma; do
    find "$O/${W}_$k" -name "*.csv" | while read f; do mv "$f" "$O/${W}_$(basename "$f")"; done
    rm -rf "$O/${W}_$k"
  done
done
ls "$O" | head -60
grep -h '"metric"' "$O"/pgat_kt.log 2>/dev/null | cut -c1-200
