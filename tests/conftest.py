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


# tests OF a route that an A/B switch removes (they look into the folded output layer's buffers / assert the table route was taken)
_ROUTE_TESTS = {
    "TXE_NO_FOLD": ("test_fused_backward_sweep_equals_unfused_chain", "test_collapsed_output_layer_equals_unfused_path",
                    "test_deferred_node_output_behaves_like_the_tensor", "test_empty_and_single_node_batches",
                    "test_fused_stack_intermediates_match_reference_goldens"),
    "TXE_NO_DEDUP": ("test_eval_encode_on_table_rows_equals_materialised_features",),
}


def pytest_collection_modifyitems(config, items):
    for env, names in _ROUTE_TESTS.items():
        if os.environ.get(env, "0") == "1":
            skip_route = pytest.mark.skip(reason=f"{env}=1 switches the tested route off")
            for item in items:
                if item.originalname in names or item.name.split("[")[0] in names:
                    item.add_marker(skip_route)
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
