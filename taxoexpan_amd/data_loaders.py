"""data_loader/data_loaders.py of the reference, on the array-based dataset: the two collate functions (same names, same
outputs) and MaskedGraphDataLoader (same constructor arguments).  `data_path` is a raw dataset directory holding
`<name>.terms/.taxo/.terms.embed` or the `<name>.txe.npz` cache written by taxoexpan_amd.dataset.MAGDataset; the
reference's DGL pickles cannot be read without DGL (dataset.py says so when handed one)."""
import os
import itertools
import logging

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
