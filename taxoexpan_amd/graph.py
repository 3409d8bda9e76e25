"""Batched egonet graph container with the DGL-0.4 `BatchedDGLGraph` surface the reference touches.

DGL does not exist for this stack, so the "DGLGraph-batched input contract"  is served by this class:
  construction   dgl.DGLGraph(), add_nodes(n, data), add_edges(u, v), nodes()      dataset.py:429-435
  batching       dgl.batch(list) -> batch_size, batch_num_nodes, batch_num_edges  data_loaders.py:25, model_zoo.py:249
  frames         g.ndata / g.edata get / set / pop                                model_zoo.py:40-42,86-88,212,241
  queries        number_of_nodes(), number_of_edges(), in_degrees(), edges()
As in DGL 0.4 the *structure* lives on the host (int64 numpy COO in edge-id order); feature frames hold torch
tensors on any device.  What is new: `csr(device)` builds -- once, cached -- the two int32 CSR views the HIP kernels
read (destination-sorted for forward, source-sorted for the atomic-free backward) plus the per-graph node offsets.
A vectorised constructor (`from_egonet_shapes`) builds whole batches without per-egonet Python objects.
"""
from collections import namedtuple

import numpy as np
import torch

from . import _lib

CSR = namedtuple("CSR", "n_nodes n_edges n_graphs rowptr_in col_src eid_in rowptr_out col_dst pos_out graph_off")


class Frame(dict):
    """ndata / edata: dict of tensors with pop(), like a DGL frame for every use the reference makes."""


class DGLGraph:
    def __init__(self):
        self._n = 0
        self._src = np.zeros(0, dtype=np.int64)
        self._dst = np.zeros(0, dtype=np.int64)
        self.ndata = Frame()
        self.edata = Frame()
        self._csr_cache = {}

    # ---- construction (dataset.py:429-435) -------------------------------------------------------------------
    def add_nodes(self, num, data=None):
        if self._n != 0 and data:
            raise ValueError("node features can only be attached by the first add_nodes call")
        self._n += int(num)
        self._csr_cache.clear()
        if data:
            for k, v in data.items():
                self.ndata[k] = v

    def add_edges(self, u, v):
        u = np.asarray(u.cpu() if torch.is_tensor(u) else u, dtype=np.int64).reshape(-1)
        v = np.asarray(v.cpu() if torch.is_tensor(v) else v, dtype=np.int64).reshape(-1)
        if u.size == 0 or v.size == 0:
            return
        if u.size == 1 and v.size > 1:
            u = np.broadcast_to(u, v.shape)
        if v.size == 1 and u.size > 1:
            v = np.broadcast_to(v, u.shape)
        if u.size != v.size:
            raise ValueError("add_edges: length mismatch")
        if u.max(initial=-1) >= self._n or v.max(initial=-1) >= self._n or u.min(initial=0) < 0 or v.min(initial=0) < 0:
            raise ValueError("add_edges: node id out of range")
        self._src = np.concatenate([self._src, u])
        self._dst = np.concatenate([self._dst, v])
        self._csr_cache.clear()

    # ---- queries ---------------------------------------------------------------------------------------------
    def nodes(self):
        return torch.arange(self._n)

    def number_of_nodes(self):
        return self._n

    def number_of_edges(self):
        return int(self._src.size)

    def edges(self):
        return torch.from_numpy(self._src.copy()), torch.from_numpy(self._dst.copy())

    def in_degrees(self):
        return torch.from_numpy(np.bincount(self._dst, minlength=self._n).astype(np.int64))

    def to_networkx(self):
        import networkx as nx
        g = nx.DiGraph()
        g.add_nodes_from(range(self._n))
        g.add_edges_from(zip(self._src.tolist(), self._dst.tolist()))
        return g

    # ---- per-graph layout (a single graph is a batch of one) ---------------------------------------------------
    @property
    def batch_size(self):
        return 1

    @property
    def batch_num_nodes(self):
        return [self._n]

    @property
    def batch_num_edges(self):
        return [int(self._src.size)]

    def _graph_offsets(self):
        return np.concatenate([[0], np.cumsum(np.asarray(self.batch_num_nodes, dtype=np.int64))])

    # ---- device structure -------------------------------------------------------------------------------------
    def csr(self, device, method="auto"):
        """int32 CSR views on `device` (cached).  method: 'host' = numpy stable argsort + upload,
        'device' = txe_build_csr (hipCUB radix sort) from the uploaded COO, 'auto' = device for large graphs."""
        device = torch.device(device)
        key = str(device)
        hit = self._csr_cache.get(key)
        if hit is not None:
            return hit
        n, e = self._n, int(self._src.size)
        if n >= 2 ** 31 - 1 or e >= 2 ** 31 - 1:
            raise ValueError("graph too large for int32 indices")
        goff = torch.from_numpy(self._graph_offsets().astype(np.int32)).to(device)
        if method == "auto":
            method = "device" if (device.type == "cuda" and e >= 200000) else "host"
        if method == "device":
            src = torch.from_numpy(self._src.astype(np.int32)).to(device)
            dst = torch.from_numpy(self._dst.astype(np.int32)).to(device)
            out = build_csr_device(src, dst, n)
            csr = CSR(n, e, len(self.batch_num_nodes), *out, goff)
        else:
            order_in = np.argsort(self._dst, kind="stable")
            rowptr_in = np.concatenate([[0], np.cumsum(np.bincount(self._dst, minlength=n))])
            order_out = np.argsort(self._src, kind="stable")
            rowptr_out = np.concatenate([[0], np.cumsum(np.bincount(self._src, minlength=n))])
            inv_in = np.empty(e, dtype=np.int64)
            inv_in[order_in] = np.arange(e)
            up = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.int32)).to(device)
            csr = CSR(n, e, len(self.batch_num_nodes), up(rowptr_in), up(self._src[order_in]), up(order_in), up(rowptr_out),
                      up(self._dst[order_out]), up(inv_in[order_out]), goff)
        self._csr_cache[key] = csr
        return csr

    # ---- DGL message passing with user functions: its only callers in the reference are the model_zoo classes, which
    # taxoexpan_amd.model_zoo replaces with fused HIP kernels; a generic (slow) route is deliberately not provided.
    def apply_edges(self, func):
        raise NotImplementedError("DGLGraph.apply_edges with user functions is not provided on MI355X -- use the taxoexpan_amd.model_zoo "
                                  "layers (GATLayer / GCNLayer / PGAT / PGCN / readouts), which fuse these primitives into HIP kernels")

    def update_all(self, message_func, reduce_func):
        raise NotImplementedError("DGLGraph.update_all with user functions is not provided on MI355X -- use the taxoexpan_amd.model_zoo "
                                  "layers (GATLayer / GCNLayer / PGAT / PGCN / readouts), which fuse these primitives into HIP kernels")


class BatchedDGLGraph(DGLGraph):
    """dgl.batch result: nodes / edges of the member graphs concatenated in list order with id offsets."""

    def __init__(self, graphs=()):
        super().__init__()
        graphs = list(graphs)
        self._batch_num_nodes = [g.number_of_nodes() for g in graphs]
        self._batch_num_edges = [g.number_of_edges() for g in graphs]
        if graphs:
            offs = np.concatenate([[0], np.cumsum(self._batch_num_nodes)])
            self._n = int(offs[-1])
            self._src = np.concatenate([g._src + o for g, o in zip(graphs, offs[:-1])])
            self._dst = np.concatenate([g._dst + o for g, o in zip(graphs, offs[:-1])])
            for k in graphs[0].ndata.keys():
                self.ndata[k] = torch.cat([g.ndata[k] for g in graphs], 0)

    @property
    def batch_size(self):
        return len(self._batch_num_nodes)

    @property
    def batch_num_nodes(self):
        return self._batch_num_nodes

    @property
    def batch_num_edges(self):
        return self._batch_num_edges

    @classmethod
    def from_egonet_shapes(cls, k, m, ndata=None):
        """Vectorised batch of egonets with k[i] grand-parents, the anchor, m[i] siblings each, in the node / edge
        order of dataset.py:404-437 (parents -> anchor, anchor -> siblings, then self loops) -- no per-egonet objects."""
        k = np.asarray(k, dtype=np.int64)
        m = np.asarray(m, dtype=np.int64)
        n = k + 1 + m
        g = cls()
        noff = np.concatenate([[0], np.cumsum(n)])
        ecount = 2 * n - 1
        eoff = np.concatenate([[0], np.cumsum(ecount)])
        N, E = int(noff[-1]), int(eoff[-1])
        src = np.empty(E, dtype=np.int64)
        dst = np.empty(E, dtype=np.int64)
        gid_e = np.repeat(np.arange(len(n)), ecount)
        local = np.arange(E) - eoff[gid_e]                # edge index inside its egonet
        kk, nn, base = k[gid_e], n[gid_e], noff[gid_e]
        is_par = local < kk
        is_chd = (~is_par) & (local < nn - 1)
        is_self = ~(is_par | is_chd)
        src[is_par] = (base + local)[is_par]
        dst[is_par] = (base + kk)[is_par]
        src[is_chd] = (base + kk)[is_chd]
        dst[is_chd] = (base + local + 1)[is_chd]
        self_id = (base + local - (nn - 1))[is_self]
        src[is_self] = self_id
        dst[is_self] = self_id
        g._n, g._src, g._dst = N, src, dst
        g._batch_num_nodes = n.tolist()
        g._batch_num_edges = ecount.tolist()
        gid_n = np.repeat(np.arange(len(n)), n)
        lnode = np.arange(N) - noff[gid_n]
        pos = np.where(lnode < k[gid_n], 0, np.where(lnode == k[gid_n], 1, 2)).astype(np.int64)
        g.ndata["pos"] = torch.from_numpy(pos)
        if ndata:
            for key, val in ndata.items():
                g.ndata[key] = val
        return g


def batch(graph_list):
    """dgl.batch (data_loaders.py:25, test_fast.py:102,161,172)."""
    return BatchedDGLGraph(graph_list)


def build_csr_device(src, dst, n_nodes):
    """txe_build_csr on device int32 COO tensors.  Returns (rowptr_in, col_src, eid_in, rowptr_out, col_dst, pos_out)."""
    assert src.is_cuda and src.dtype == torch.int32 and dst.dtype == torch.int32
    e = int(src.numel())
    dev = src.device
    i32 = lambda k: torch.empty(k, dtype=torch.int32, device=dev)
    rowptr_in, rowptr_out = i32(n_nodes + 1), i32(n_nodes + 1)
    col_src, eid_in, col_dst, pos_out = i32(max(e, 1)), i32(max(e, 1)), i32(max(e, 1)), i32(max(e, 1))
    wsb = _lib.call("txe_build_csr_ws_bytes", n_nodes, e)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    with _lib.on_device(dev):
        _lib.call("txe_build_csr", _lib.ptr(src), _lib.ptr(dst), n_nodes, e, _lib.ptr(rowptr_in), _lib.ptr(col_src),
                  _lib.ptr(eid_in), _lib.ptr(rowptr_out), _lib.ptr(col_dst), _lib.ptr(pos_out), _lib.ptr(ws), wsb,
                  _lib.stream_ptr())
    return rowptr_in, col_src[:e], eid_in[:e], rowptr_out, col_dst[:e], pos_out[:e]


class DeviceBatchedGraph(BatchedDGLGraph):
    """A batch of egonets whose node table and CSR views were built ON DEVICE (txe_egonet_*), never as host objects.
    Same surface as BatchedDGLGraph; the host COO (`_src/_dst`, edges(), in_degrees()) is materialised lazily on demand."""

    def __init__(self, csr, node_off, ids, pos):
        DGLGraph.__init__(self)
        self._n = csr.n_nodes
        self._csr_dev = csr
        self._csr_cache[str(csr.rowptr_in.device)] = csr
        self._node_off = node_off
        self._host_ready = False
        self.ndata["_id"] = ids
        self.ndata["pos"] = pos

    def _materialise_host(self):
        if self._host_ready:
            return
        c = self._csr_dev
        e = c.n_edges
        eid = c.eid_in[:e].cpu().numpy().astype(np.int64)
        src = np.empty(e, dtype=np.int64)
        dst = np.empty(e, dtype=np.int64)
        src[eid] = c.col_src[:e].cpu().numpy()
        rp = c.rowptr_in.cpu().numpy().astype(np.int64)
        dst[eid] = np.repeat(np.arange(c.n_nodes), np.diff(rp))
        self.__dict__["_src_host"], self.__dict__["_dst_host"] = src, dst
        off = self._node_off.cpu().numpy().astype(np.int64)
        self._batch_num_nodes = np.diff(off).tolist()
        self._batch_num_edges = (2 * np.diff(off) - 1).tolist()
        self._host_ready = True

    @property
    def _src(self):
        self._materialise_host()
        return self.__dict__["_src_host"]

    @_src.setter
    def _src(self, v):
        pass

    @property
    def _dst(self):
        self._materialise_host()
        return self.__dict__["_dst_host"]

    @_dst.setter
    def _dst(self, v):
        pass

    @property
    def batch_size(self):
        return self._csr_dev.n_graphs

    @property
    def batch_num_nodes(self):
        self._materialise_host()
        return self._batch_num_nodes

    @property
    def batch_num_edges(self):
        self._materialise_host()
        return self._batch_num_edges

    def number_of_edges(self):
        return self._csr_dev.n_edges

    def csr(self, device, method="auto"):
        device = torch.device(device)
        if device == self._csr_dev.rowptr_in.device:
            return self._csr_dev
        return super().csr(device, method)


class DeviceTaxonomy:
    """parent / child CSR (int32) and the node feature table of a taxonomy, resident on one GPU"""

    def __init__(self, par_ptr, par_idx, chd_ptr, chd_idx, features, device):
        i32 = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.int32).to(device)
        self.par_ptr, self.par_idx, self.chd_ptr, self.chd_idx = i32(par_ptr), i32(par_idx), i32(chd_ptr), i32(chd_idx)
        self.features = features.to(device) if features is not None else None
        self.device = torch.device(device)


class _EgonetJob:
    """a batch of egonets whose node counts are being computed on the device (device_egonet_begin)"""
    __slots__ = ("dtax", "anchors", "exclude", "G", "expand_factor", "seed", "index_base", "node_off", "ws", "n_host", "ready")


def device_egonet_begin(dtax, anchors, exclude=None, expand_factor=50, seed=0, index_base=0):
    """First half of device_egonet_batch: the per-egonet node counts and their prefix sum on the current stream, and an ASYNCHRONOUS
    read-back of the batch's node count into pinned memory.  Nothing waits here: a loader that begins batch i+1 before the consumer
    enqueues step i finds the count ready when it finishes the batch (data_loaders.DeviceBatchLoader)."""
    dev = dtax.device
    to_dev = lambda a: None if a is None else (a.to(device=dev, dtype=torch.int32) if torch.is_tensor(a)
                                               else torch.as_tensor(np.asarray(a), dtype=torch.int32).to(dev))
    job = _EgonetJob()
    job.dtax, job.anchors, job.exclude = dtax, to_dev(anchors), to_dev(exclude)
    job.G = G = int(job.anchors.numel())
    job.expand_factor, job.seed, job.index_base = expand_factor, seed, int(index_base)
    job.node_off = torch.empty(max(G + 1, 1), dtype=torch.int32, device=dev)
    with _lib.on_device(dev):
        wsb = _lib.call("txe_egonet_ws_bytes", G)
        job.ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        _lib.call("txe_egonet_offsets", _lib.ptr(dtax.par_ptr), _lib.ptr(dtax.chd_ptr), _lib.ptr(dtax.chd_idx), _lib.ptr(job.anchors),
                  _lib.ptr(job.exclude), G, expand_factor, seed, job.index_base, _lib.ptr(job.node_off), _lib.ptr(job.ws), wsb, _lib.stream_ptr())
        job.n_host = _count_slot(dev)
        job.n_host.copy_(job.node_off[G:G + 1], non_blocking=True)
        job.ready = torch.cuda.Event()
        job.ready.record()
    return job


_COUNT_SLOTS = {}


def _count_slot(dev):
    """a pinned int32 for a job's read-back: slots are handed back by device_egonet_finish; a job that is never finished keeps its
    slot (a few bytes), it is never shared"""
    free = _COUNT_SLOTS.setdefault(str(dev), [])
    return free.pop() if free else torch.empty(1, dtype=torch.int32).pin_memory()


def device_egonet_finish(job, with_features=True):
    """Second half of device_egonet_batch: waits for the node count (the one host synchronisation of batch construction: the sizes of
    the output arrays), then fills the node table and both CSR views on the current stream."""
    dtax, dev, G = job.dtax, job.dtax.device, job.G
    job.ready.synchronize()
    N = int(job.n_host[0])
    _COUNT_SLOTS.setdefault(str(dev), []).append(job.n_host)
    job.n_host = None
    E = 2 * N - G
    i32 = lambda k: torch.empty(max(k, 1), dtype=torch.int32, device=dev)
    node_off = job.node_off
    ids, pos = i32(N), i32(N)
    rowptr_in, rowptr_out = i32(N + 1), i32(N + 1)
    col_src, eid_in, col_dst, pos_out = i32(E), i32(E), i32(E), i32(E)
    with _lib.on_device(dev):
        _lib.call("txe_egonet_fill", _lib.ptr(dtax.par_ptr), _lib.ptr(dtax.par_idx), _lib.ptr(dtax.chd_ptr), _lib.ptr(dtax.chd_idx),
                  _lib.ptr(job.anchors), _lib.ptr(job.exclude), G, job.expand_factor, job.seed, job.index_base, _lib.ptr(node_off), _lib.ptr(ids),
                  _lib.ptr(pos), _lib.ptr(rowptr_in), _lib.ptr(col_src), _lib.ptr(eid_in), _lib.ptr(rowptr_out), _lib.ptr(col_dst),
                  _lib.ptr(pos_out), _lib.stream_ptr())
    csr = CSR(N, E, G, rowptr_in[:N + 1], col_src[:E], eid_in[:E], rowptr_out[:N + 1], col_dst[:E], pos_out[:E], node_off[:G + 1])
    g = DeviceBatchedGraph(csr, node_off[:G + 1], ids[:N], pos[:N])
    if with_features and dtax.features is not None:
        if with_features == "lazy":          # x = features[_id] kept symbolic: eval-mode encoders project the table once (ops.GatheredRows)
            from .ops import GatheredRows
            g.ndata["x"] = GatheredRows(dtax.features, ids[:N])
        else:
            g.ndata["x"] = dtax.features.index_select(0, ids[:N].long())
    return g


def device_egonet_batch(dtax, anchors, exclude=None, expand_factor=50, seed=0, with_features=True, index_base=0):
    """Batched egonets of `anchors` built on the GPU (dataset.py:404-437 + dgl.batch).  anchors / exclude: int arrays or
    int32 device tensors.  Returns a DeviceBatchedGraph with ndata '_id', 'pos' (int32, device) and 'x' (features gathered;
    with_features="lazy": an ops.GatheredRows over the taxonomy's feature table).  index_base: position of anchors[0] in the caller's whole
    anchor list -- chunks of one list (test_fast.py's `-b`) then sample the same siblings as the single batch."""
    return device_egonet_finish(device_egonet_begin(dtax, anchors, exclude, expand_factor, seed, index_base), with_features)
