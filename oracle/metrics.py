"""Metric and count-model definitions of the reference, restated  --  TEST INFRASTRUCTURE ONLY.

PINNED: every function here is checked against outputs of the reference's own
``utils.py`` (run in the build container) in ``tests/test_oracle_golden.py``.

  multinomial_probabilities   utils.py:79-99
  transition_matrix           utils.py:102-138
  neg_log_likelihood          utils.py:141-142
  compute_likelihood_cut      utils.py:145-163
  compute_likelihood          utils.py:166-178
  markov_predict              model.py:159-167
  multinomial_predict         model.py:138-143
  val_loss_history_cut        model.py:106-112 (ValLossHistoryCut.on_epoch_end)
  recall_at_k                 extension (the reference has no ranking metric)
"""
import numpy as np


def multinomial_probabilities(seqs, n, k=1.0, normalize=True):
    counts = np.zeros((1, n))
    for seq in seqs:
        for s in seq:
            counts[0, s] += 1
    z = counts.sum(axis=1)
    out = counts + k
    if normalize:
        out = out / (z + n * k)
    return out


def transition_matrix(seqs, n, k=0, freq=False, end_state=True):
    alpha = np.zeros((n, n + 1)) if end_state else np.zeros((n, n))
    gamma = np.zeros(n)
    for seq in seqs:
        if len(seq) > 1:
            for i, j in zip(seq[:-1], seq[1:]):
                alpha[i, j] += 1
            if end_state:
                alpha[seq[-1], n] += 1
        elif end_state:
            alpha[seq[0], n] += 1
        gamma[seq[0]] += 1
    sa = alpha + k
    sg = gamma + k
    if not freq:
        z = alpha.sum(axis=1).reshape((n, 1))
        sa = sa / (z + n * k)
        sg = sg / (gamma.sum() + n * k)
    return sa, sg


def neg_log_likelihood(probs):
    return -np.sum(np.log(probs))


def compute_likelihood_cut(predictions, train_percent, orig_lengths=None, count_first_prob=False):
    assert train_percent <= 1.0
    tr, va = [], []
    for i, pred in enumerate(predictions):
        sp = pred[:]
        if not count_first_prob:
            sp = sp[1:]
        if orig_lengths is not None:
            sp = pred[-int(orig_lengths[i]):]
        L = len(sp)
        n_tr = int(np.ceil(train_percent * L))
        n_va = int(np.floor((1.0 - train_percent) * L))
        if n_tr > 0:
            tr.append(neg_log_likelihood(sp[0:n_tr]) / n_tr)
        if n_va > 0:
            va.append(neg_log_likelihood(sp[-n_va:]) / n_va)
    return np.sum(tr) / len(tr), np.sum(va) / len(va)


def compute_likelihood(predictions, count_first_prob=False):
    eps = 1e-07
    lls = []
    for pred in predictions:
        sp = pred[:]
        if not count_first_prob:
            sp = sp[1:]
        sp = np.clip(sp, eps, 1.0 - eps)
        if len(sp) > 0:
            lls.append(neg_log_likelihood(sp) / len(sp))
    return np.mean(lls)


def multinomial_predict(model, seqs):
    return [[model[0, s] for s in seq] for seq in seqs]


def markov_predict(alpha, gamma, seqs):
    out = []
    for seq in seqs:
        p = [gamma[seq[0]]]
        if len(seq) > 1:
            for i, j in zip(seq[:-1], seq[1:]):
                p.append(alpha[i, j])
        out.append(p)
    return out


def val_loss_history_cut(y_pred, y_true, orig_lengths, eps=1e-7):
    """probability of the true class per step (max over classes of pred*onehot),
    clipped, then the last-30 % NLL of compute_likelihood_cut."""
    pt = np.max(np.multiply(y_pred, y_true), axis=2)
    pt = np.clip(pt, eps, 1.0 - eps)
    return compute_likelihood_cut(pt, 0.7, orig_lengths=orig_lengths)


def recall_at_k(scores, targets, k):
    """scores:(n,V) targets:(n,) -> fraction of rows whose target has fewer than k
    items scoring strictly higher (ties resolved in the target's favour)."""
    ts = scores[np.arange(scores.shape[0]), targets]
    rank = (scores > ts[:, None]).sum(axis=1)
    return float(np.mean(rank < k))
