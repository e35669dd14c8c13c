#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the importable parts of the reference.

Only two reference modules parse under Python 3 and have no Keras/Theano
dependency: ``utils.py`` (count-model fitting + NLL metric definitions) and
``sampler.py`` (synthetic sequence generators).  They are imported from
/root/reference (never copied), called on fixed inputs, and the inputs and
outputs are written to ``tests/golden/reference_utils_sampler.json``.

The reference is Python 2; two py3 shims are needed and nothing else:
  * ``builtins.xrange = range``            (sampler.py:31,190-191)
  * ``matplotlib.use`` wrapped to drop the removed ``warn=`` kwarg (utils.py:3)

Run only in the build container (the reference does not exist on the GPU box);
the JSON it writes is the committed fixture.
"""
import builtins
import json
import os
import random
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_utils_sampler.json")


def _import_reference():
    builtins.xrange = range
    import matplotlib
    _use = matplotlib.use

    def use(backend, *a, **kw):
        kw.pop("warn", None)
        return _use(backend, *a, **kw)

    matplotlib.use = use
    sys.path.insert(0, REF)
    import utils as ref_utils
    import sampler as ref_sampler
    sys.path.remove(REF)
    return ref_utils, ref_sampler


def main():
    ru, rs = _import_reference()
    out = {"generator": "tests/golden/make_reference_vectors.py", "cases": {}}

    # --- the three example sequences of model.py:413-415 (n=4 classes) ------
    seqs = [[3, 1, 0, 2, 3, 2, 3, 1, 3, 2], [3, 1, 2, 2, 1, 1, 1, 2], [3, 1, 3, 3, 1]]
    n = 4
    c = {"seqs": seqs, "n": n}
    c["multinomial_k1"] = ru.multinomial_probabilities(seqs, n, k=1.0).tolist()
    c["multinomial_k1_unnorm"] = ru.multinomial_probabilities(seqs, n, k=1.0, normalize=False).tolist()
    A, g = ru.transition_matrix(seqs, n, k=10e-7, freq=False, end_state=False)
    c["transition_k1e-6"] = {"alpha": A.tolist(), "gamma": g.tolist()}
    A2, g2 = ru.transition_matrix(seqs, n, k=1.0, freq=True, end_state=True)
    c["transition_counts_end_state"] = {"alpha": A2.tolist(), "gamma": g2.tolist()}
    # MarkovModel.predict rule (model.py:159-167) applied with the reference's matrices
    preds = []
    for s in seqs:
        p = [float(g[s[0]])] + [float(A[i, j]) for i, j in zip(s[:-1], s[1:])]
        preds.append(p)
    c["markov_preds"] = preds
    c["compute_likelihood"] = float(ru.compute_likelihood(preds, count_first_prob=False))
    c["compute_likelihood_first"] = float(ru.compute_likelihood(preds, count_first_prob=True))
    tr, va = ru.compute_likelihood_cut(preds, 0.7)
    c["compute_likelihood_cut_0.7"] = [float(tr), float(va)]
    # orig_lengths branch (used by ValLossHistoryCut, model.py:111-112): padded rows
    T = max(len(p) for p in preds)
    padded = [[0.5] * (T - len(p)) + p for p in preds]
    lens = [len(p) for p in preds]
    tr2, va2 = ru.compute_likelihood_cut(padded, 0.7, orig_lengths=lens)
    c["compute_likelihood_cut_orig_lengths"] = {"padded": padded, "lengths": lens, "out": [float(tr2), float(va2)]}
    c["neg_log_likelihood"] = float(ru.neg_log_likelihood(np.array([0.5, 0.25, 0.125])))
    c["compute_seq_max_length"] = int(ru.compute_seq_max_length(seqs))
    c["chop_sequences_4"] = ru.chop_sequences(seqs, offset=4)
    out["cases"]["model_py_example"] = c

    # --- sampler.MCSampler with fixed alpha/gamma and seeded `random` ---------
    rng = np.random.RandomState(7)
    n = 6
    alpha = rng.rand(n, n + 1)
    alpha /= alpha.sum(axis=1, keepdims=True)
    gamma = rng.rand(n)
    gamma /= gamma.sum()
    random.seed(1234)
    mc = rs.MCSampler(alpha.copy(), gamma.copy(), beta=0.5, use_end_token=True)
    mc_seqs = [[int(x) for x in mc.gen_sequence()] for _ in range(8)]
    out["cases"]["mcsampler_end_token"] = {
        "alpha": alpha.tolist(), "gamma": gamma.tolist(), "beta": 0.5,
        "random_seed": 1234, "sequences": mc_seqs,
    }
    alpha2 = rng.rand(n, n)
    alpha2 /= alpha2.sum(axis=1, keepdims=True)
    random.seed(99)
    mc2 = rs.MCSampler(alpha2.copy(), gamma.copy(), beta=1.0, use_end_token=False)
    mc2_seqs = [[int(x) for x in mc2.gen_sequence(12)] for _ in range(4)]
    out["cases"]["mcsampler_fixed_len"] = {
        "alpha": alpha2.tolist(), "gamma": gamma.tolist(), "beta": 1.0,
        "random_seed": 99, "length": 12, "sequences": mc2_seqs,
    }

    # --- a seeded MCSampler DATASET at the MSNBC vocabulary size (SURVEY 8c KAT 9): inputs of the 3-epoch
    #     loss-trajectory parity test (tests/test_gpu_model.py)
    r17 = np.random.RandomState(17)
    n17 = 17
    a17 = r17.rand(n17, n17 + 1) ** 3                      # peaked rows: a learnable chain
    a17[:, n17] = 0.12 * a17[:, :n17].sum(axis=1)          # end-token mass -> mean length ~8
    a17 /= a17.sum(axis=1, keepdims=True)
    g17 = r17.rand(n17)
    g17 /= g17.sum()
    random.seed(2017)
    mc17 = rs.MCSampler(a17.copy(), g17.copy(), beta=0.8, use_end_token=True)
    data = []
    while len(data) < 650:
        sq = [int(v) for v in mc17.gen_sequence()]
        if 2 <= len(sq) <= 40:
            data.append(sq)
    out["cases"]["mcsampler_v17_dataset"] = {"alpha": a17.tolist(), "gamma": g17.tolist(), "beta": 0.8, "random_seed": 2017,
                                             "train": data[:500], "val": data[500:]}

    # --- randomized count-model / metric cases (incl. the MSNBC vocabulary size, 17) ------------------
    rnd = []
    for seed, n, nseq in ((11, 7, 9), (12, 17, 14), (13, 12, 6)):
        r = np.random.RandomState(seed)
        seqs = [[int(v) for v in r.randint(0, n, size=int(r.randint(2, 15)))] for _ in range(nseq)]
        c = {"seqs": seqs, "n": n, "variants": []}
        for k, freq, end_state in ((0.5, False, False), (2.0, False, True), (0.0 if n < 10 else 1e-3, True, False), (1.0, True, True)):
            A, g = ru.transition_matrix(seqs, n, k=k, freq=freq, end_state=end_state)
            c["variants"].append({"k": k, "freq": freq, "end_state": end_state, "alpha": np.asarray(A).tolist(),
                                  "gamma": np.asarray(g).tolist()})
        c["multinomial_k0.5"] = ru.multinomial_probabilities(seqs, n, k=0.5).tolist()
        A, g = ru.transition_matrix(seqs, n, k=0.5, freq=False, end_state=False)
        preds = [[float(g[s[0]])] + [float(A[i, j]) for i, j in zip(s[:-1], s[1:])] for s in seqs]
        c["markov_preds"] = preds
        c["compute_likelihood"] = [float(ru.compute_likelihood(preds, count_first_prob=False)),
                                   float(ru.compute_likelihood(preds, count_first_prob=True))]
        c["cut"] = {}
        for tp in (0.7, 0.5):
            tr, va = ru.compute_likelihood_cut(preds, tp)
            c["cut"][str(tp)] = [float(tr), float(va)]
        T = max(len(p) for p in preds)
        padded = [[0.25] * (T - len(p)) + p for p in preds]
        lens = [len(p) for p in preds]
        tr2, va2 = ru.compute_likelihood_cut(padded, 0.7, orig_lengths=lens)
        c["cut_orig_lengths"] = {"padded": padded, "lengths": lens, "out": [float(tr2), float(va2)]}
        c["unique_elements"] = int(ru.compute_unique_elements(seqs))
        c["seq_max_length"] = int(ru.compute_seq_max_length(seqs))
        rnd.append(c)
    out["cases"]["random_count_models"] = rnd

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
