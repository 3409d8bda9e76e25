"""data_loader/data_loaders.py of the reference, on the array-based dataset: the two collate functions (same names, same
outputs) and MaskedGraphDataLoader (same constructor arguments).  `data_path` is a raw dataset directory holding
`<name>.terms/.taxo/.terms.embed` or the `<name>.txe.npz` cache written by taxoexpan_amd.dataset.MAGDataset; the
reference's DGL pickles cannot be read without DGL (dataset.py says so when handed one)."""
import os
import itertools
import logging

import numpy as np
import torch
import torch.utils.data

from . import dataset as _ds
from .graph import batch

BATCH_GRAPH_NODE_LIMIT = 100000      # data_loaders.py:7: nodes per batched graph in large-batch mode


def _flatten(samples):
    """instances (tuples of [egonet, query feature, label] triplets) -> three parallel lists"""
    triplets = list(itertools.chain.from_iterable(samples))
    return [t[0] for t in triplets], [t[1] for t in triplets], [t[2] for t in triplets]


def collate_graph_and_node_small_batch(samples):
    """data_loaders.py:9-28: list of instances -> (batched graph, [B, d] query features, [B] labels)"""
    graphs, node_features, labels = _flatten(samples)
    return batch(graphs), torch.stack(node_features), torch.tensor(labels)


def collate_graph_and_node_large_batch(samples):
    """data_loaders.py:31-72: as above but cut into several batched graphs once a running node count passes
    BATCH_GRAPH_NODE_LIMIT (the egonet that crosses the limit stays in the batch it closes)"""
    graphs, node_features, labels = _flatten(samples)
    out_g, out_f, out_l = [], [], []
    start, nodes = 0, 0
    for i, g in enumerate(graphs):
        nodes += g.number_of_nodes()
        if nodes > BATCH_GRAPH_NODE_LIMIT and i > start:
            out_g.append(batch(graphs[start:i + 1]))
            out_f.append(torch.stack(node_features[start:i + 1]))
            out_l.append(torch.tensor(labels[start:i + 1]))
            start, nodes = i + 1, 0
    if start < len(graphs):
        out_g.append(batch(graphs[start:]))
        out_f.append(torch.stack(node_features[start:]))
        out_l.append(torch.tensor(labels[start:]))
    return out_g, out_f, out_l


def _open_dataset(data_path):
    if os.path.isdir(data_path):
        names = [f[:-len(".terms")] for f in os.listdir(data_path) if f.endswith(".terms")]
        if len(names) != 1:
            raise ValueError(f"{data_path}: expected exactly one <name>.terms file, found {sorted(names)}")
        name = names[0]
        cache = os.path.join(data_path, f"{name}.txe.npz")
        # (the optional existing partition, dataset.py:160-170: a changed split must invalidate the cache too)
        raw = [os.path.join(data_path, f) for f in (f"{name}.terms", f"{name}.taxo", f"{name}.terms.embed", f"{name}.terms.train",
                                                    f"{name}.terms.validation", f"{name}.terms.test")]
        # the cache written by the first raw load is used while it is newer than the raw files (the reference's pickle flow,
        # generate_dataset_binary.py): train / validation / test loaders and every rank parse the text files once, not 3 x world times
        if os.path.exists(cache) and all(os.path.getmtime(cache) >= os.path.getmtime(f) for f in raw if os.path.exists(f)):
            try:
                return _ds.MAGDataset(name=name, path=cache, raw=False)
            except Exception as exc:                  # unreadable cache: say so, fall back to the raw files (which rewrites it)
                logging.getLogger(__name__).warning("ignoring unreadable dataset cache %s (%r): re-reading the raw files", cache, exc)
        return _ds.MAGDataset(name=name, path=data_path, raw=True)
    return _ds.MAGDataset(name="", path=data_path, raw=False)


_COLLATE = {"small_batch": collate_graph_and_node_small_batch, "large_batch": collate_graph_and_node_large_batch}


class MaskedGraphDataLoader(torch.utils.data.DataLoader):
    """data_loaders.py:75-117 (same constructor arguments; `data_path` is a raw directory or a .txe.npz cache)"""

    def __init__(self, mode, data_path, sampling_mode=1, batch_size=10, batch_type="small_batch", negative_size=20, expand_factor=50,
                 shuffle=True, num_workers=8, cache_refresh_time=64, normalize_embed=False, test_topk=-1):
        if batch_type not in _COLLATE:
            raise AssertionError("batch_type arg must be either small_batch or large_batch")
        if mode not in ("train", "validation", "test"):
            raise AssertionError("mode must be one of train, validation, and test")
        self.mode, self.sampling_mode, self.batch_size_, self.batch_type = mode, sampling_mode, batch_size, batch_type
        self.negative_size, self.expand_factor, self.shuffle = negative_size, expand_factor, shuffle
        self.cache_refresh_time, self.normalize_embed = cache_refresh_time, normalize_embed
        self.dataset_ = _ds.MaskedGraphDataset(_open_dataset(data_path), mode=mode, sampling_mode=sampling_mode,
                                               negative_size=negative_size, expand_factor=expand_factor,
                                               cache_refresh_time=cache_refresh_time, normalize_embed=normalize_embed,
                                               test_topk=test_topk)
        collate = _COLLATE[batch_type]
        super().__init__(dataset=self.dataset_, batch_size=batch_size, shuffle=shuffle, collate_fn=collate, num_workers=num_workers,
                         pin_memory=torch.cuda.is_available())
        self.n_samples = len(self.dataset_)

    def __str__(self):
        return "\n\t".join([f"MaskedGraphDataLoader mode: {self.mode}", f"sampling_mode: {self.sampling_mode}",
                            f"batch_size: {self.batch_size_}", f"negative_size: {self.negative_size}",
                            f"expand_factor: {self.expand_factor}", f"cache_refresh_time: {self.cache_refresh_time}",
                            f"normalize_embed: {self.normalize_embed}"])


_PINNED = {}


def begin_device_batch(dtax, anchors, exclude, query_ids, expand_factor=50, seed=0, stream=None, repeated_queries=False):
    """First half of build_device_batch: the index arrays go up in one pinned copy and the egonets' node counts are computed, on
    `stream` (default: the current stream); nothing is waited for.  Returns the job finish_device_batch completes.
    repeated_queries: the batch's query features come back as ops.RepeatedRows -- the runs of equal consecutive query ids are found
    here, on the host, and only the distinct rows are gathered (BIM / LBM then project U rows instead of one per pair)."""
    from .graph import device_egonet_begin
    dev = dtax.device
    main = torch.cuda.current_stream(dev)
    side = stream if stream is not None else main
    # (the side stream does NOT wait for the caller's stream -- that would put the construction behind the running step again: the
    #  taxonomy arrays and the feature table must be complete before the first call; DeviceBatchLoader synchronises once when it is made)
    # the three index arrays travel in ONE pinned buffer and one copy (each small pageable upload costs ~60 us of host time); two
    # buffers per device take turns, each guarded by the event of its last upload (a loader keeps two batches in flight)
    B = len(anchors)
    ring = _PINNED.setdefault(str(dev), dict(k=0, slots=[None, None]))     # per device; the buffers grow to the largest batch seen
    ring["k"] ^= 1
    slot = ring["slots"][ring["k"]]
    if slot is None or slot[0].numel() < 4 * B + 1:               # [anchors | exclude | query ids | run offsets (<= B + 1)]
        if slot is not None and slot[1] is not None:
            slot[1].synchronize()                                 # (the smaller buffer's last upload has left it)
        cap = max(4 * B + 1, 2 * slot[0].numel() if slot is not None else 0)
        slot = ring["slots"][ring["k"]] = [torch.empty(cap, dtype=torch.int32).pin_memory(), None]
    host, uploaded = slot
    if uploaded is not None:
        uploaded.synchronize()
    hv = host.numpy()
    hv[:B] = anchors
    hv[B:2 * B] = exclude if exclude is not None else -1
    n_runs = 0
    if repeated_queries:                                          # distinct consecutive query ids + the first pair of every run
        q = np.asarray(query_ids).reshape(-1)
        start = np.flatnonzero(np.concatenate([[True], q[1:] != q[:-1]])) if B else np.zeros(0, dtype=np.int64)
        n_runs = len(start)
        hv[2 * B:2 * B + n_runs] = q[start]
        hv[3 * B:3 * B + n_runs] = start
        hv[3 * B + n_runs] = B
    else:
        hv[2 * B:3 * B] = query_ids
    with torch.cuda.stream(side):
        packed = host[:4 * B + 1].to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        job = device_egonet_begin(dtax, packed[:B], packed[B:2 * B] if exclude is not None else None, expand_factor=expand_factor, seed=seed)
        packed.record_stream(side)
    return dict(job=job, packed=packed, B=B, side=side, dev=dev, n_runs=n_runs if repeated_queries else None)


def finish_device_batch(pending, features):
    """Second half of build_device_batch: waits for the node count of the batch begun earlier (the one host synchronisation of the
    construction; behind work enqueued a whole step ago it returns at once), fills the node table and both CSR views and gathers the
    node / query features on the builder's stream; the CURRENT stream is made to wait for the finished batch, and every tensor of the
    batch is marked as used on it (caching-allocator safety).  Returns dict(g, x, pos, qf, n_nodes, n_edges)."""
    from .graph import device_egonet_finish
    side, dev, packed, B = pending["side"], pending["dev"], pending["packed"], pending["B"]
    main = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side):
        g = device_egonet_finish(pending["job"], with_features=True)
        x = g.ndata.pop("x")
        U = pending["n_runs"]
        if U is None:
            qt = qf = features.index_select(0, packed[2 * B:3 * B])
        else:                                                     # only the distinct query rows are gathered
            from .ops import RepeatedRows
            qt = features.index_select(0, packed[2 * B:2 * B + U])
            qf = RepeatedRows(qt, packed[3 * B:3 * B + U + 1], B)
        from . import ops
        # the batch's walk plan (a view of its graphs like the CSR orders; the backward sweep of a four-head stack stages from it)
        plan = ops.walk_plan(g.csr(dev)) if not (ops._NO_EGO_WALK or ops._NO_WALK_PLAN) else None
    if side is not main:
        main.wait_stream(side)
        csr = g.csr(dev)
        packed.record_stream(main)
        for t in (x, qt, g.ndata["_id"], g.ndata["pos"], csr.rowptr_in, csr.col_src, csr.eid_in, csr.rowptr_out, csr.col_dst, csr.pos_out,
                  csr.graph_off) + ((plan,) if plan is not None else ()):
            t.record_stream(main)
    return dict(g=g, x=x, pos=g.ndata["pos"], qf=qf, n_nodes=g.number_of_nodes(), n_edges=g.number_of_edges())


def build_device_batch(dtax, anchors, exclude, query_ids, features, expand_factor=50, seed=0, stream=None, repeated_queries=False):
    """One training batch built ON the device (data_loaders.py:9-28 + dataset.py:404-437 without host egonet objects):
    graph.device_egonet_batch for the anchors, node features and query features gathered from the resident table.
    stream: build on that side stream -- the one host synchronisation of the construction (the array sizes) then waits for the
    builder's own few microseconds of work only, not for the training step still running on the caller's stream.  A loop that
    calls begin_device_batch for batch i+1 BEFORE it enqueues step i does not wait at all (DeviceBatchLoader does).
    Returns dict(g, x, pos, qf, n_nodes, n_edges)."""
    return finish_device_batch(begin_device_batch(dtax, anchors, exclude, query_ids, expand_factor, seed, stream, repeated_queries), features)


class DeviceBatchLoader:
    """`for batch in loader` over a MaskedGraphDataset in 'train' / 'validation' mode with the batches built on the GPU: the sampler
    (dataset.sample_anchors: the reference's positive pointer and negative sampling, host Python like data_loader/dataset.py:334-381)
    hands anchors to begin_device_batch / finish_device_batch.  Each batch is built on a side stream while the previous step runs,
    in two halves around the consumer's enqueue of that step: batch b+1 is BEGUN (sampled, uploaded, node counts launched) before
    batch b is handed out and FINISHED (arrays sized from the count, filled, features gathered) at the next `next()`, so the one host
    synchronisation of the construction finds its value already there.
    repeated_queries (default): the query features are an ops.RepeatedRows -- the sampler pairs one query with 1 + negative_size
    consecutive anchors, only the distinct rows are gathered and BIM / LBM project those (every other matcher of model_zoo densifies it;
    `.dense()` gives the reference's stacked [B, in_dim] tensor); False: the stacked tensor itself.  Yields (graph, node features, query features, labels) -- MaskedGraphDataLoader's small-batch
    tuple with the node features popped, all on `device`."""

    def __init__(self, dataset, batch_size, device, shuffle=True, seed=0, drop_last=False, repeated_queries=True):
        self.repeated_queries = bool(repeated_queries)
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.seed, self.drop_last = shuffle, int(seed), drop_last
        self.dtax = dataset.device_taxonomy(self.device)
        self.features = self.dtax.features
        self._side = torch.cuda.Stream(device=self.device)
        torch.cuda.current_stream(self.device).synchronize()      # the resident taxonomy / feature table are complete from here on
        self._epoch = 0

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        import random
        order = list(range(len(self.dataset)))
        if self.shuffle:
            random.Random(self.seed + self._epoch).shuffle(order)
        self._epoch += 1
        def begin(b):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            query, anchor, label, exclude = self.dataset.sample_anchors(idx)
            return label, begin_device_batch(self.dtax, anchor, exclude, query, expand_factor=self.dataset.expand_factor,
                                             seed=self.seed + 7919 * self._epoch + b, stream=self._side, repeated_queries=self.repeated_queries)
        # two batches in flight: batch b+1 is begun (sampled, uploaded, node counts launched) before the consumer gets batch b, so the
        # count's read-back has a whole step's enqueue to arrive and `finish` never waits
        nxt = begin(0) if len(self) else None
        for b in range(len(self)):
            label, pending = nxt
            batch = finish_device_batch(pending, self.features)
            nxt = begin(b + 1) if b + 1 < len(self) else None
            yield batch["g"], batch["x"], batch["qf"], torch.as_tensor(label).to(self.device, non_blocking=True)
