"""taxoexpan_amd -- MI355X-native implementation of TaxoExpan's propagation / readout / match path.

    from taxoexpan_amd import model_zoo            # drop-in for the reference's model/model_zoo.py
    from taxoexpan_amd.graph import DGLGraph, batch  # the DGL-0.4 graph surface the loaders use
The compute lives in taxoexpan_amd/csrc/libtxe.so (C ABI: include/txe.h).  There is no CPU fallback.
"""
from . import graph, model_zoo, ops  # noqa: F401
from .model import TaxoExpan, encode_graph  # noqa: F401

__all__ = ["graph", "model_zoo", "ops", "TaxoExpan", "encode_graph"]
