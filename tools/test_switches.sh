#!/bin/bash
# The GPU test suite on each of the library's alternative routes (tests/conftest.py: TXE_TEST_ROUTE sets the module attribute):
#   gpurun --timeout 1200 -- 'bash tools/test_switches.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in no_fold no_fused_bwd no_fused_logits no_side_stream no_query_runs no_tail_chain no_match_fold no_fold_edot no_split_gemm no_ego_walk no_walk_plan no_virtual_x; do
  echo -n "TXE_TEST_ROUTE=$r: "; TXE_TEST_ROUTE=$r python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1
done
