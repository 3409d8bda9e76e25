#!/bin/bash
# rocprofv3 passes behind profiles/<tag>_*: kernel-trace stats of the bench command, then FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (TCC slots; never combined with sys/hip/hsa tracing) over tools/profile_workload.py, and a fourth pass with
# the MFMA-busy / GUI-active counters (MFMA pipe utilisation of the GEMMs).
#   gpurun -- 'bash tools/collect_profiles.sh r01'      then, back in the build container:
#   python tools/summarize_profiles.py gpurun_out/prof_r01 r01
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/prof_$TAG
rm -rf "$O" && mkdir -p "$O"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/kt" -o kt -- python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > "$O/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$O/fetch" -o fetch -- python tools/profile_workload.py > "$O/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$O/write" -o write -- python tools/profile_workload.py > "$O/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$O/mfma" -o mfma -- python tools/profile_workload.py > "$O/mfma.log" 2>&1
find "$O" -mindepth 2 -name "*.csv" -exec mv {} "$O/" \;
ls -la "$O" | head -30
grep -h '"metric"' "$O/kt.log" | cut -c1-200
