#!/bin/bash
# The GPU test suite under every A/B switch of the library (each selects another route to the same numbers):
#   gpurun --timeout 3000 -- 'bash tools/test_switches.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for sw in TXE_NO_BALANCED_SPLITS TXE_NO_PERSIST_GEMM TXE_NO_PERSIST_SLICES TXE_NO_X_DROPPED TXE_NO_BN160 TXE_NO_BN160_SPLIT TXE_NO_SIDE_STREAM TXE_NO_FUSED_BWD \
          TXE_NO_TABLE_SWEEP TXE_NO_MULTI_PREPARE TXE_NO_FUSED_LOGITS TXE_NO_FOLD TXE_NO_DEDUP TXE_NO_DXPOS TXE_NO_TN_LDS TXE_BN160_STRICT TXE_NO_QUERY_RUNS; do
  echo -n "$sw=1: "; env $sw=1 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1
done
echo -n "TXE_TORCH_EVENTS=1: "; TXE_TORCH_EVENTS=1 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1
echo -n "TXE_FWD_NPW=1: "; TXE_FWD_NPW=1 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1
for v in 0 1; do echo -n "TXE_PREFETCH_V=$v: "; TXE_PREFETCH_V=$v python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1; done
