"""Host side of the negative sampler: proposal distributions and the alias table the device
kernel (csrc/ops.hip: sample_negatives_kernel) draws from.  Extension -- the reference has full
softmax only (model.py:257,382-397).  Integer pipeline specified in oracle/rng.py."""
import numpy as np


def log_uniform_probs(V, rank=None):
    """P(item) = log((r+2)/(r+1)) / log(V+1) with r the item's frequency rank (default: r = id)."""
    r = np.arange(V, dtype=np.float64) if rank is None else np.asarray(rank, dtype=np.float64)
    return (np.log(r + 2.0) - np.log(r + 1.0)) / np.log(V + 1.0)


def unigram_probs(counts, power=0.75):
    c = np.asarray(counts, dtype=np.float64) ** power
    return c / c.sum()


def build_alias_table(probs):
    """Vose's alias method, float64, fixed processing order (so two hosts build the same table).
    Returns (thresh uint32[V], alias int32[V]): bucket j is kept iff a 32-bit uniform < thresh[j]."""
    p = np.asarray(probs, dtype=np.float64)
    V = p.shape[0]
    scaled = (p / p.sum()) * V
    alias = np.arange(V, dtype=np.int64)
    accept = np.ones(V, dtype=np.float64)
    small = np.nonzero(scaled < 1.0)[0].tolist()
    large = np.nonzero(scaled >= 1.0)[0].tolist()
    sc = scaled.tolist()
    while small and large:
        s = small.pop()
        l = large.pop()
        accept[s] = sc[s]
        alias[s] = l
        sc[l] = (sc[l] + sc[s]) - 1.0
        if sc[l] < 1.0:
            small.append(l)
        else:
            large.append(l)
    thresh = np.minimum(np.floor(accept * 4294967296.0), 4294967295.0).astype(np.uint64).astype(np.uint32)
    return thresh, alias.astype(np.int32)
