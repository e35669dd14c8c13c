"""Shared test helpers: random sessions, padded/oracle batches, parameter init."""
import numpy as np

from oracle import nn as onn


def make_sessions(rng, n, V, min_len=2, max_len=9):
    """n sessions of item ids (each yields len-1 transitions, preprocessor.py:75-78)."""
    return [rng.integers(0, V, size=int(rng.integers(min_len, max_len + 1))).tolist() for _ in range(n)]


def pad_batch(sessions, T=None):
    """Reference pairing + PRE-padding (preprocessor.py:16-20,67-94) in id form."""
    L = [len(s) - 1 for s in sessions]
    T = T or max(L)
    B = len(sessions)
    ids = np.zeros((B, T), np.int64)
    tgt = np.zeros((B, T), np.int64)
    mask = np.zeros((B, T), bool)
    for b, s in enumerate(sessions):
        n = L[b]
        if n <= 0:
            continue
        ids[b, T - n:] = s[:-1]
        tgt[b, T - n:] = s[1:]
        mask[b, T - n:] = True
    return {"ids": ids, "tgt": tgt, "mask": mask}


def init_params(rng, cfg, V, H, D=None, dtype=np.float64, scale=0.3):
    G = onn.N_GATES[cfg["cell"]]
    p = {}
    if cfg["input"] == "embed":
        p["E"] = rng.normal(0, scale, (V, D))
        p["W"] = rng.normal(0, scale, (D, G * H))
    else:
        p["Wk"] = rng.normal(0, scale, (V, G * H))
    p["U"] = rng.normal(0, scale, (H, G * H))
    if cfg.get("use_bias", True):
        p["b"] = rng.normal(0, scale, (G * H,))
    if cfg["output"] == "full":
        p["Wout"] = rng.normal(0, scale, (H, V))
    elif not cfg.get("tied", False):
        p["Eout"] = rng.normal(0, scale, (V, H))
    if cfg.get("out_bias", False):
        p["bout"] = rng.normal(0, scale, (V,))
    return {k: v.astype(dtype) for k, v in p.items()}


def dense_grad(g, shape):
    """Row-sparse (rows, vals) -> dense array."""
    if isinstance(g, tuple):
        out = np.zeros(shape, g[1].dtype)
        out[g[0]] = g[1]
        return out
    return g
