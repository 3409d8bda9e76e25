import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle"), os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# TXE_TEST_ROUTE=<name> runs the whole suite on one of the library's alternative routes (tools/test_switches.sh): the names map to the
# module attributes the parity tests monkeypatch one at a time.  Read HERE, by the test harness -- the library reads no environment.
_ROUTES = {"no_fold": ("model_zoo", "_NO_FOLD"), "no_fused_bwd": ("ops", "_NO_FUSED_BWD"), "no_fused_logits": ("ops", "_NO_FUSED_LOGITS"),
           "no_side_stream": ("ops", "_NO_SIDE_STREAM"), "no_table_sweep": ("ops", "_NO_TABLE_SWEEP"), "no_query_runs": ("ops", "_NO_QUERY_RUNS"), "no_tail_chain": ("ops", "_NO_TAIL_CHAIN"), "no_match_fold": ("ops", "_NO_MATCH_FOLD"), "no_fold_edot": ("ops", "_NO_FOLD_EDOT"),
           "no_split_gemm": ("ops", "_NO_SPLIT_GEMM"), "no_ego_walk": ("ops", "_NO_EGO_WALK"),
           "no_walk_plan": ("ops", "_NO_WALK_PLAN"),
           "no_virtual_x": ("ops", "_NO_VIRTUAL_X")}
# tests OF a route that the setting removes (they look into the folded output layer's buffers)
_ROUTE_TESTS = {
    "no_fold": ("test_fused_backward_sweep_equals_unfused_chain", "test_collapsed_output_layer_equals_unfused_path",
                "test_deferred_node_output_behaves_like_the_tensor", "test_empty_and_single_node_batches",
                "test_fused_stack_intermediates_match_reference_goldens"),
}


def _apply_test_route():
    name = os.environ.get("TXE_TEST_ROUTE", "")
    if not name:
        return
    import importlib
    mod, attr = _ROUTES[name]
    setattr(importlib.import_module("taxoexpan_amd." + mod), attr, True)


def pytest_collection_modifyitems(config, items):
    _apply_test_route()
    route = os.environ.get("TXE_TEST_ROUTE", "")
    if route in _ROUTE_TESTS:
        skip_route = pytest.mark.skip(reason=f"TXE_TEST_ROUTE={route} switches the tested route off")
        for item in items:
            if item.originalname in _ROUTE_TESTS[route] or item.name.split("[")[0] in _ROUTE_TESTS[route]:
                item.add_marker(skip_route)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
