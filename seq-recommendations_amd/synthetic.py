"""Synthetic "MSNBC-shaped" sessions at catalogue scale.

The reference's own generator, ``sampler.MCSampler`` (sampler.py:43-134), is a first-order Markov
chain with a dense n x n transition matrix and a Python loop per item -- unusable at |items| = 1M.
This is its scalable analogue with the same structure (first-order dependence on the previous
item): first item ~ Zipf(s) over a fixed random permutation of the catalogue; next item = one of 8
fixed successors of the current item w.p. 0.8, else a fresh Zipf draw.  Session lengths are
``clip(1 + Geometric(0.2), 2, 50)`` items (MSNBC's mean session length is ~5.7), or a constant 50
("saturated").  Storage is CSR: flat int32 ids + int64 starts.
"""
import numpy as np


class SyntheticSessions:
    def __init__(self, V, seed=1234, zipf_s=1.05, n_succ=8, p_succ=0.8):
        self.V = V
        self.rng = np.random.default_rng(seed)
        self.perm = self.rng.permutation(V).astype(np.int32)          # rank -> item
        self.rank = np.empty(V, np.int64)
        self.rank[self.perm] = np.arange(V)
        w = 1.0 / np.power(np.arange(1, V + 1, dtype=np.float64), zipf_s)
        self.cdf = np.cumsum(w)
        self.cdf /= self.cdf[-1]
        self.n_succ, self.p_succ = n_succ, p_succ
        self.succ = self._zipf(V * n_succ).reshape(V, n_succ)

    def _zipf(self, n):
        r = np.searchsorted(self.cdf, self.rng.random(n), side="right")
        return self.perm[np.minimum(r, self.V - 1)]

    def lengths(self, n, saturated=False, max_items=50):
        if saturated:
            return np.full(n, max_items, np.int64)
        return np.clip(1 + self.rng.geometric(0.2, size=n), 2, max_items).astype(np.int64)

    def generate(self, n, saturated=False, max_items=50):
        """-> (flat int32 ids, starts int64[n+1])"""
        L = self.lengths(n, saturated, max_items)
        starts = np.zeros(n + 1, np.int64)
        np.cumsum(L, out=starts[1:])
        flat = np.empty(int(starts[-1]), np.int32)
        cur = self._zipf(n)
        flat[starts[:-1]] = cur
        for s in range(1, int(L.max())):
            act = np.nonzero(L > s)[0]
            if act.size == 0:
                break
            c = cur[act]
            nxt = self.succ[c, self.rng.integers(0, self.n_succ, size=act.size)]
            fresh = self.rng.random(act.size) >= self.p_succ
            nf = int(fresh.sum())
            if nf:
                nxt[fresh] = self._zipf(nf)
            cur[act] = nxt
            flat[starts[act] + s] = nxt
        return flat, starts

    def proposal_rank(self):
        """Frequency rank of every item (for the log-uniform negative-sampling proposal)."""
        return self.rank
