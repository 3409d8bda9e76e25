"""TEST INFRASTRUCTURE ONLY -- the sliver of gensim.models.KeyedVectors the reference's data_loader/dataset.py touches
(gensim is not installed and cannot be), so that the UNMODIFIED dataset.py can be imported in the build container by
oracle/gen_dataset_golden.py.  Nothing in the product imports this.

PARITY-UNPINNED (gensim is an un-vendored third-party dependency, README.md:7-13 of the reference lists it unpinned):
  load_word2vec_format(path)   dataset.py:136   text word2vec: header "count dim", then "key v0 v1 ..." per line
  kv.vectors, kv[key]          dataset.py:137,161
  KeyedVectors(vector_size=d)  dataset.py:228
  kv.add(keys, vectors)        dataset.py:229
  kv.distances(key, others)    dataset.py:309   cosine distance 1 - cos(key, other) in the order of `others`
"""
import numpy as np


class KeyedVectors:
    def __init__(self, vector_size=0):
        self.vector_size = vector_size
        self.index = {}
        self.vectors = np.zeros((0, vector_size), dtype=np.float32)

    @classmethod
    def load_word2vec_format(cls, path):
        with open(path, "r") as fin:
            count, dim = (int(t) for t in fin.readline().split())
            kv = cls(dim)
            keys, rows = [], []
            for line in fin:
                segs = line.rstrip().split(" ")
                if len(segs) < dim + 1:
                    continue
                keys.append(segs[0])
                rows.append([float(t) for t in segs[1:dim + 1]])
        assert len(keys) == count
        kv.add(keys, np.asarray(rows, dtype=np.float32))
        return kv

    def add(self, keys, vectors):
        vectors = np.asarray(vectors, dtype=np.float32)
        base = self.vectors.shape[0]
        for i, k in enumerate(keys):
            self.index[k] = base + i
        self.vectors = np.concatenate([self.vectors.reshape(base, vectors.shape[1]), vectors], 0)

    def __getitem__(self, key):
        return self.vectors[self.index[key]]

    def distances(self, key, others=()):
        v = self[key]
        m = np.stack([self[o] for o in others]) if len(others) else self.vectors
        return 1.0 - (m @ v) / (np.linalg.norm(m, axis=1) * np.linalg.norm(v))
