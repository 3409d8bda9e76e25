"""On-disk formats and the sampling side of the egonet path, without DGL / gensim / networkx.

Mirrors the reference's data_loader interface for the path -- same class names, constructor arguments and attributes:
  MAGDataset(name, path, embed_suffix, raw, existing_partition)       data_loader/dataset.py:40-205
      raw `.terms` / `.taxo` / `.terms.embed` files (README.md:21-51), seeded leaf split (dataset.py:171-181)
  MaskedGraphDataset(graph_dataset, mode, sampling_mode, ...)           data_loader/dataset.py:208-437
      node_list / node2parents / node2masks / all_positions, the negative-sampling queue, __getitem__ -> triplets
  load_new_taxons(path, normalize)                                      infer.py:23-38
What differs by design: the taxonomy lives in CSR arrays (parents ascending by node id, children in file order -- the
orders the reference's nx.DiGraph round trip produces), an instance is first a list of node ids (`sample()`), and whole
batches go to the GPU as arrays (`batch_arrays` -> graph.device_egonet_batch / BatchedDGLGraph.from_egonet_shapes) instead
of per-egonet graph objects.  Python's `random` module is consumed call for call like the reference consumes it, so a
seeded run emits the same instances (tests/test_dataset.py against traces captured from the unmodified reference).

The reference's `*.pickle.bin` cache embeds a DGL-0.4 graph object and cannot be read without DGL; this module caches
to `<name>.txe.npz` instead and says so when handed a pickle.

Restated reference code, acknowledged: `_get_at_most_k_negatives` / `_get_exactly_k_negatives` follow
data_loader/dataset.py:334-381 statement for statement (about 35 lines, including its corner-case alert message) -- the
`random` stream has to be consumed call for call for the trace tests to hold, which leaves no freedom in how they are written.
Everything else in this file is organised differently (CSR adjacency, array batches, device taxonomy).
"""
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

from .graph import BatchedDGLGraph, DGLGraph


# ---- raw formats ------------------------------------------------------------------------------------------------------
def read_terms(path):
    """`<taxon_id>\\t<name>` per line (README.md:25-31) -> (ids, names) in file order"""
    ids, names = [], []
    with open(path, "r") as fin:
        for line in fin:
            line = line.strip()
            if not line:
                continue
            segs = line.split("\t")
            if len(segs) != 2:
                raise AssertionError(f"Wrong number of segmentations {line}")          # dataset.py:117
            ids.append(segs[0])
            names.append(segs[1])
    return ids, names


def read_taxo(path, index):
    """`<parent_id>\\t<child_id>` per line (README.md:33-40) -> list of distinct (parent, child) node-id pairs, first
    occurrence order; unknown ids raise KeyError like dataset.py:130-131"""
    seen, edges = set(), []
    with open(path, "r") as fin:
        for line in fin:
            line = line.strip()
            if not line:
                continue
            segs = line.split("\t")
            if len(segs) != 2:
                raise AssertionError(f"Wrong number of segmentations {line}")
            e = (index[segs[0]], index[segs[1]])
            if e not in seen:
                seen.add(e)
                edges.append(e)
    return edges


def read_embed(path):
    """word2vec text format (README.md:42-51): header `count dim`, then `<taxon_id> v0 v1 ...` -> {id: row}, dim"""
    rows = {}
    with open(path, "r") as fin:
        header = fin.readline().split()
        count, dim = int(header[0]), int(header[1])
        for line in fin:
            segs = line.rstrip().split(" ")
            if len(segs) < dim + 1:
                continue
            rows[segs[0]] = np.asarray([float(t) for t in segs[1:dim + 1]], dtype=np.float32)
    if len(rows) != count:
        raise ValueError(f"{path}: header announces {count} vectors, found {len(rows)}")
    return rows, dim


def load_new_taxons(path, normalize=False):
    """infer.py:23-38: `<name with spaces>\\t<v0 v1 ...>` per line -> (vocab with '_' for ' ', float64 array).  The
    reference 'normalises' by the ROW SUM (infer.py:35-36), not the L2 norm; kept."""
    vocab, nf = [], []
    with open(path, "r") as fin:
        for line in fin:
            line = line.strip()
            if line:
                segs = line.split("\t")
                vocab.append("_".join(segs[0].split(" ")))
                nf.append([float(t) for t in segs[1].split(" ")])
    nf = np.array(nf)
    if normalize:
        nf = nf / nf.sum(axis=1)[:, np.newaxis]
    return vocab, nf


def _csr(n, edges, key, order_by_other):
    """CSR of `edges` grouped by column `key` (0 = by parent -> children, 1 = by child -> parents).  Within a group the
    other endpoint keeps edge order, or ascends by node id when order_by_other."""
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    k, o = e[:, key], e[:, 1 - key]
    perm = np.lexsort((o, k)) if order_by_other else np.argsort(k, kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(k, minlength=n))]).astype(np.int64)
    return ptr, o[perm]


class MAGDataset:
    """data_loader/dataset.py:40-205.  Attributes: name, vocab, g_full (host graph container with ndata['x']),
    train_node_ids / validation_node_ids / test_node_ids; plus the CSR arrays the GPU path consumes."""

    def __init__(self, name, path, embed_suffix="", raw=True, existing_partition=False):
        self.name = name
        self.embed_suffix = embed_suffix
        self.existing_partition = existing_partition
        self.g_full = DGLGraph()
        self.vocab = []
        self.train_node_ids, self.validation_node_ids, self.test_node_ids = [], [], []
        if raw:
            self._load_dataset_raw(path)
        else:
            self._load_dataset_cached(path)

    # -- raw ----------------------------------------------------------------------------------------------------------
    def _load_dataset_raw(self, dir_path):
        stem = self.name if self.embed_suffix == "" else f"{self.name}.{self.embed_suffix}"
        embed_file = os.path.join(dir_path, f"{self.name}.terms.embed" if self.embed_suffix == "" else
                                  f"{self.name}.terms.{self.embed_suffix}.embed")
        tx_ids, names = read_terms(os.path.join(dir_path, f"{self.name}.terms"))
        index = {}
        for t in tx_ids:
            if t in index:
                raise ValueError(f"duplicate taxon id {t} in {self.name}.terms")
            index[t] = len(index)
        pairs = read_taxo(os.path.join(dir_path, f"{self.name}.taxo"), index)
        rows, dim = read_embed(embed_file)
        n = len(tx_ids)
        if len(rows) != n:                                   # dataset.py:159 allocates embeddings.vectors.shape rows
            raise ValueError(f"{embed_file}: {len(rows)} vectors for {n} terms")
        self.vocab = [names[i] + "@@@" + str(i) for i in range(n)]                       # dataset.py:149
        # edge ids of g_full follow nx.DiGraph.edges(): by parent in node order, children in first-occurrence order (:152-156)
        by_parent = sorted(range(len(pairs)), key=lambda i: pairs[i][0])                 # stable
        pairs = [pairs[i] for i in by_parent]
        feats = np.zeros((n, dim), dtype=np.float64)
        for t, i in index.items():
            feats[i] = rows[t]                               # KeyError for a term without a vector, like dataset.py:161
        self._finish(n, pairs, torch.FloatTensor(feats))
        if self.existing_partition:                          # dataset.py:139-144,166-169
            rd = lambda suffix: [index[t] for t in self._load_node_list(os.path.join(dir_path, f"{self.name}.terms.{suffix}"))]
            self.train_node_ids, self.validation_node_ids, self.test_node_ids = rd("train"), rd("validation"), rd("test")
        else:                                                # dataset.py:171-181: seeded shuffle of the leaves, 10 % / 10 %
            leaves = [i for i in range(n) if self.chd_ptr[i + 1] == self.chd_ptr[i]]
            random.seed(47)
            random.shuffle(leaves)
            n_val = int(len(leaves) * 0.1)
            n_test = int(len(leaves) * 0.1)
            self.validation_node_ids = leaves[:n_val]
            self.test_node_ids = leaves[n_val:n_val + n_test]
            held = set(self.validation_node_ids) | set(self.test_node_ids)
            self.train_node_ids = [i for i in range(n) if i not in held]
        self.save(os.path.join(dir_path, f"{stem}.txe.npz"))

    def _finish(self, n, pairs, features):
        self.n_nodes = n
        self.edges = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        self.chd_ptr, self.chd_idx = _csr(n, self.edges, 0, False)
        self.par_ptr, self.par_idx = _csr(n, self.edges, 1, True)
        self.g_full = DGLGraph()
        self.g_full.add_nodes(n, {"x": features})
        self.g_full.add_edges(self.edges[:, 0], self.edges[:, 1])

    @staticmethod
    def _load_node_list(file_path):
        with open(file_path, "r") as fin:
            return [line.strip() for line in fin if line.strip()]

    # -- cache ----------------------------------------------------------------------------------------------------------
    def save(self, path):
        """written through a temporary file + os.replace: concurrent loaders (one per rank / DataLoader) never see a torn archive"""
        tmp = f"{path}.{os.getpid()}.tmp.npz"
        np.savez(tmp, name=self.name, vocab=np.asarray(self.vocab, dtype=object), edges=self.edges,
                 features=self.g_full.ndata["x"].numpy(), train=np.asarray(self.train_node_ids, dtype=np.int64),
                 validation=np.asarray(self.validation_node_ids, dtype=np.int64), test=np.asarray(self.test_node_ids, dtype=np.int64))
        os.replace(tmp, path)

    def _load_dataset_cached(self, path):
        if not str(path).endswith(".npz"):
            raise ValueError(f"{path}: the reference's pickled datasets embed a DGL-0.4 graph object and cannot be read without "
                             "DGL; load the raw .terms/.taxo/.embed directory once (raw=True), which writes <name>.txe.npz")
        d = np.load(path, allow_pickle=True)
        self.name = str(d["name"])
        self.vocab = d["vocab"].tolist()
        self._finish(len(self.vocab), d["edges"], torch.from_numpy(d["features"]))
        self.train_node_ids, self.validation_node_ids, self.test_node_ids = (d[k].tolist() for k in ("train", "validation", "test"))


class KeyedRows:
    """The slice of gensim's KeyedVectors the path touches (dataset.py:227-229 builds `self.kv`; test_fast.py:88,121,194 and
    infer.py:37-38,97,145 read `kv[str(key)]`; dataset.py:323 calls `kv.distances`): string key -> feature row (numpy), without gensim."""

    def __init__(self, vector_size):
        self.vector_size = int(vector_size)
        self.index2word, self.vocab = [], {}
        self.vectors = np.zeros((0, self.vector_size), dtype=np.float32)

    def add(self, entities, weights, replace=False):
        weights = np.asarray(weights, dtype=np.float32).reshape(-1, self.vector_size)
        if len(entities) != weights.shape[0]:
            raise ValueError("KeyedRows.add: one weight row per entity")
        new = [i for i, e in enumerate(entities) if e not in self.vocab]
        if replace:
            for i, e in enumerate(entities):
                if e in self.vocab:
                    self.vectors[self.vocab[e]] = weights[i]
        base = len(self.index2word)
        for j, i in enumerate(new):
            self.vocab[entities[i]] = base + j
            self.index2word.append(entities[i])
        self.vectors = weights[new] if base == 0 and len(new) == len(entities) else np.concatenate([self.vectors, weights[new]], 0)

    def __contains__(self, key):
        return key in self.vocab

    def __len__(self):
        return len(self.index2word)

    def __getitem__(self, key):
        if isinstance(key, (list, tuple)):
            return np.stack([self[k] for k in key])
        return self.vectors[self.vocab[key]]          # KeyError for an unknown key, like gensim

    get_vector = __getitem__

    def distances(self, key, other_keys=()):
        """cosine distances 1 - cos(key, other) to `other_keys` (all rows when empty) -- gensim's KeyedVectors.distances"""
        v = self[key] if isinstance(key, str) else np.asarray(key, dtype=np.float32)
        m = self.vectors if len(other_keys) == 0 else self.vectors[[self.vocab[k] for k in other_keys]]
        return 1.0 - (m @ v) / (np.linalg.norm(m, axis=1) * np.linalg.norm(v))


class _NodeList(list):
    """`graph.nodes` as an attribute AND as nx's `graph.nodes()` call (infer.py:80 iterates `test_dataset.graph.nodes()`)"""

    def __call__(self):
        return self


class _Adjacency:
    """mutable successor / predecessor lists of the masked taxonomy (the reference's nx subgraph copy, dataset.py:231-239):
    parents ascend by node id, children keep g_full's edge order"""

    def __init__(self, ds, keep):
        keep_mask = np.zeros(ds.n_nodes, dtype=bool)
        keep_mask[np.asarray(keep, dtype=np.int64)] = True
        self.nodes = _NodeList(i for i in range(ds.n_nodes) if keep_mask[i])
        self.succ = {i: [int(c) for c in ds.chd_idx[ds.chd_ptr[i]:ds.chd_ptr[i + 1]] if keep_mask[c]] for i in self.nodes}
        self.pred = {i: [int(p) for p in ds.par_idx[ds.par_ptr[i]:ds.par_ptr[i + 1]] if keep_mask[p]] for i in self.nodes}

    def descendants(self, node):
        seen, stack = set(), [node]
        while stack:
            for c in self.succ[stack.pop()]:
                if c not in seen:
                    seen.add(c)
                    stack.append(c)
        return seen

    def drop_in_edges(self, node):
        removed = len(self.pred[node])
        for p in self.pred[node]:
            self.succ[p].remove(node)
        self.pred[node] = []
        return removed

    def edges(self):
        return [(u, v) for u in self.nodes for v in self.succ[u]]


class MaskedGraphDataset(torch.utils.data.Dataset):
    """data_loader/dataset.py:208-437.  One instance = a list of (anchor egonet, query feature, label) triplets."""

    def __init__(self, graph_dataset, mode="train", sampling_mode=1, negative_size=32, expand_factor=64, cache_refresh_time=128,
                 normalize_embed=False, test_topk=-1):
        assert mode in ["train", "validation", "test"], "mode in MaskedGraphDataset must be one of train, validation, and test"
        assert sampling_mode in [0, 1, 2, 3], "sampling_mode in MaskedGraphDataset must be in [0,1,2,3]"
        if mode == "test":
            assert sampling_mode == 0, "!!! During testing, sampling_mode must be 0, in order to emit all positive true parents"
        self.mode, self.sampling_mode, self.negative_size = mode, sampling_mode, negative_size
        self.expand_factor, self.cache_refresh_time = expand_factor, cache_refresh_time
        self.normalize_embed, self.test_topk = normalize_embed, test_topk
        self.node_features = graph_dataset.g_full.ndata["x"]
        if normalize_embed:
            self.node_features = F.normalize(self.node_features, p=2, dim=1)
        self.vocab = graph_dataset.vocab
        self.kv = KeyedRows(vector_size=self.node_features.shape[1])           # dataset.py:227-229
        self.kv.add([str(i) for i in range(len(self.vocab))], self.node_features.numpy())
        held = {"train": [], "validation": graph_dataset.validation_node_ids, "test": graph_dataset.test_node_ids}[mode]
        self.node_list = graph_dataset.train_node_ids if mode == "train" else held
        self.graph = _Adjacency(graph_dataset, list(graph_dataset.train_node_ids) + list(held))
        roots = [v for v in self.graph.nodes if not self.graph.pred[v]]
        interested = set(self.node_list) - set(roots)                      # set order decides node_list order (dataset.py:243-244)
        self.node_list = list(interested)
        self.node2parents, self.node2positive_pointer, self.node2masks = {}, {}, {}
        self.all_positions = set(graph_dataset.train_node_ids)
        for v in self.graph.nodes:
            parents = list(self.graph.pred[v])
            self.node2parents[v] = parents
            self.node2positive_pointer[v] = 0
            if v in interested:
                self.node2masks[v] = set(list(self.graph.descendants(v)) + parents + [v] + roots)
        # held-out nodes lose their in-edges AFTER parents / masks are recorded: no leakage into egonets (dataset.py:262-272)
        for v in held:
            self.graph.drop_in_edges(v)
        self.cache, self.cache_counter = {}, {}
        self.pointer = 0
        self.queue = (graph_dataset.train_node_ids * 5).copy()

    def __str__(self):
        return f"MaskedGraphDataset mode:{self.mode}"

    def __len__(self):
        return len(self.node_list)

    # ---- sampling (random-module call sequence of dataset.py:293-402) ------------------------------------------------------
    def sample(self, idx):
        """instance idx as (query node, [(anchor, label, egonet node ids, k parents), ...])"""
        q = self.node_list[idx]
        out = []
        if self.sampling_mode == 0:
            for p in self.node2parents[q]:
                out.append((p, 1) + self._egonet(q, p, 1))
        elif self.sampling_mode == 1:
            ptr = self.node2positive_pointer[q]
            p = self.node2parents[q][ptr]
            out.append((p, 1) + self._egonet(q, p, 1))
            self.node2positive_pointer[q] = (ptr + 1) % len(self.node2parents[q])
        if self.mode in ("train", "validation"):
            negatives = self._get_negative_anchors(q, self.negative_size)
        elif self.test_topk == -1:
            negatives = [a for a in self.all_positions if a not in self.node2masks[q]]
        else:                                                 # nearest test_topk unmasked positions by cosine distance (:307-311)
            pool = [a for a in self.all_positions if a not in self.node2masks[q]]
            x = self.node_features.numpy()
            m, v = x[np.asarray(pool, dtype=np.int64)], x[q]
            dist = 1.0 - (m @ v) / (np.linalg.norm(m, axis=1) * np.linalg.norm(v))
            negatives = [pool[i] for i in np.argsort(dist, kind="stable")[:self.test_topk]]
        for a in negatives:
            out.append((a, 0) + self._egonet(q, a, 0))
        return q, out

    def _get_negative_anchors(self, query_node, negative_size):
        if self.sampling_mode == 0:
            return self._get_at_most_k_negatives(query_node, negative_size)
        elif self.sampling_mode == 1:
            return self._get_exactly_k_negatives(query_node, negative_size)

    def _get_at_most_k_negatives(self, query_node, negative_size):
        if self.pointer == 0:
            random.shuffle(self.queue)
        masks = self.node2masks[query_node]
        negatives = [a for a in self.queue[self.pointer:self.pointer + negative_size] if a not in masks]
        if not negatives:                 # the reference spins forever on this window (dataset.py:342-345); fail instead
            raise RuntimeError(f"no unmasked negative for query {query_node} in the current queue window")
        self.pointer += negative_size
        if self.pointer >= len(self.queue):
            self.pointer = 0
        return negatives

    def _get_exactly_k_negatives(self, query_node, negative_size):
        if self.pointer == 0:
            random.shuffle(self.queue)
        masks = self.node2masks[query_node]
        negatives, tries = [], 0
        while len(negatives) != negative_size:
            lack = negative_size - len(negatives)
            negatives.extend(a for a in self.queue[self.pointer:self.pointer + lack] if a not in masks)
            self.pointer += lack
            if self.pointer >= len(self.queue):
                self.pointer = 0
                random.shuffle(self.queue)
            tries += 1
            if tries > 10:                # corner case (dataset.py:370-375): trim / pad from the head of the queue
                print(f"Alert in _get_exactly_k_negatives, query_node: {query_node}, current negative size: {len(negatives)}")
                if len(negatives) > negative_size:
                    negatives = negatives[:negative_size]
                else:
                    negatives.extend(self.queue[:negative_size - len(negatives)])
        return negatives

    def _egonet(self, query_node, anchor, instance_mode):
        """(node ids [parents, anchor, siblings], k); negatives are cached for cache_refresh_time uses (dataset.py:390-400)"""
        if instance_mode == 0 and anchor in self.cache and self.cache_counter[anchor] < self.cache_refresh_time:
            self.cache_counter[anchor] += 1
            return self.cache[anchor]
        ego = self._build_egonet(query_node, anchor, instance_mode)
        if instance_mode == 0:
            self.cache[anchor] = ego
            self.cache_counter[anchor] = 0
        return ego

    def _build_egonet(self, query_node, anchor, instance_mode):
        parents = self.graph.pred[anchor]
        children = self.graph.succ[anchor]
        if len(children) > self.expand_factor:                # with replacement (dataset.py:419,424)
            children = random.choices(children, k=self.expand_factor)
        if instance_mode == 1:
            children = [c for c in children if c != query_node]
        return parents + [anchor] + list(children), len(parents)

    def _get_subgraph(self, query_node, anchor_node, instance_mode):
        """dataset.py:404-437 as a graph object (infer.py:82 calls this directly, bypassing the cache)"""
        return self._graph_of(*self._build_egonet(query_node, anchor_node, instance_mode))

    def _graph_of(self, ids, k):
        n = len(ids)
        g = DGLGraph()
        g.add_nodes(n, {"x": self.node_features[ids, :], "_id": torch.tensor(ids), "pos": torch.tensor([0] * k + [1] + [2] * (n - k - 1))})
        g.add_edges(list(range(k)), k)
        g.add_edges(k, list(range(k + 1, n)))
        g.add_edges(g.nodes(), g.nodes())
        return g

    def __getitem__(self, idx):
        q, inst = self.sample(idx)
        qf = self.node_features[q, :]
        return tuple([self._graph_of(ids, k), qf, label] for (_a, label, ids, k) in inst)

    # ---- array batches for the GPU path -------------------------------------------------------------------------------------
    def batch_arrays(self, indices):
        """instances `indices` as arrays: dict(query [B], label [B], k [B], m [B], ids [N]) -- what dgl.batch over the
        triplets would hold (data_loaders.py:24-28), ready for BatchedDGLGraph.from_egonet_shapes"""
        query, label, k, m, ids = [], [], [], [], []
        for idx in indices:
            q, inst = self.sample(idx)
            for (_a, lab, nodes, kk) in inst:
                query.append(q)
                label.append(lab)
                k.append(kk)
                m.append(len(nodes) - kk - 1)
                ids.extend(nodes)
        i64 = lambda a: np.asarray(a, dtype=np.int64)
        return dict(query=i64(query), label=i64(label), k=i64(k), m=i64(m), ids=i64(ids))

    def sample_anchors(self, indices):
        """instances `indices` as (query [B], anchor [B], label [B], exclude [B]) int64 arrays for graph.device_egonet_batch:
        the positive / negative anchors of `sample()` without building host egonets (siblings beyond expand_factor are then drawn
        on device).  exclude = the query for positive pairs (it must not appear among its parent's children, dataset.py:421-424)."""
        query, anchor, label = [], [], []
        for idx in indices:
            q = self.node_list[idx]
            if self.sampling_mode == 0:
                pos = list(self.node2parents[q])
            else:
                ptr = self.node2positive_pointer[q]
                pos = [self.node2parents[q][ptr]]
                self.node2positive_pointer[q] = (ptr + 1) % len(self.node2parents[q])
            if self.mode in ("train", "validation"):
                neg = self._get_negative_anchors(q, self.negative_size)
            else:
                neg = [a for a in self.all_positions if a not in self.node2masks[q]]
            for a in pos:
                query.append(q); anchor.append(a); label.append(1)
            for a in neg:
                query.append(q); anchor.append(a); label.append(0)
        i64 = lambda a: np.asarray(a, dtype=np.int64)
        query, anchor, label = i64(query), i64(anchor), i64(label)
        return query, anchor, label, np.where(label == 1, query, -1)

    def device_taxonomy(self, device):
        """the masked taxonomy (held-out in-edges removed) + features as device CSR arrays for graph.device_egonet_batch:
        all-candidate inference builds `_get_subgraph(-1, anchor, 0)` for every anchor (test_fast.py:93-97, infer.py:80-82)"""
        from .graph import DeviceTaxonomy
        n = self.node_features.shape[0]
        cnt = lambda adj: np.asarray([len(adj.get(i, ())) for i in range(n)], dtype=np.int64)
        flat = lambda adj: np.asarray([v for i in range(n) for v in adj.get(i, ())], dtype=np.int64)
        ptr = lambda c: np.concatenate([[0], np.cumsum(c)])
        return DeviceTaxonomy(ptr(cnt(self.graph.pred)), flat(self.graph.pred), ptr(cnt(self.graph.succ)), flat(self.graph.succ),
                              self.node_features, device)

    def collate(self, indices):
        """(batched graph, query features, labels) like collate_graph_and_node_small_batch (data_loaders.py:9-28)"""
        b = self.batch_arrays(indices)
        g = BatchedDGLGraph.from_egonet_shapes(b["k"], b["m"])
        ids = torch.from_numpy(b["ids"])
        g.ndata["_id"] = ids
        g.ndata["x"] = self.node_features[ids]
        return g, self.node_features[torch.from_numpy(b["query"])], torch.from_numpy(b["label"])
