"""`dgl` name-compatibility for the reference's loaders: `import taxoexpan_amd.dgl_compat as dgl` gives
dgl.DGLGraph / dgl.batch (dataset.py:429-435, data_loaders.py:25, test_fast.py:102) on our graph container.

The generic DGL message-passing entry points (apply_edges / update_all / edge_softmax / mean_nodes) are NOT
re-implemented one primitive at a time: their only callers in the reference are the model_zoo classes, which
taxoexpan_amd.model_zoo replaces with fused HIP kernels.  Calling them raises, loudly, instead of silently running
some slow generic path.
"""
from .graph import BatchedDGLGraph, DGLGraph, batch  # noqa: F401


def _unsupported(name):
    raise NotImplementedError(
        f"dgl.{name}: generic DGL message passing is not provided on MI355X -- use taxoexpan_amd.model_zoo "
        "(GATLayer/GCNLayer/PGAT/PGCN/readouts), which fuse these primitives into HIP kernels")


def apply_edges(g, func):
    _unsupported("DGLGraph.apply_edges")


def update_all(g, message_func, reduce_func):
    _unsupported("DGLGraph.update_all")


def mean_nodes(g, feat, weight=None):
    _unsupported("mean_nodes")


def sum_nodes(g, feat, weight=None):
    _unsupported("sum_nodes")
