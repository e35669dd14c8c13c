"""Metric definitions and count-model fitting used around the hot path (reference utils.py:79-202),
restated in numpy.  What "loss" means for the count baselines and for ValLossHistoryCut is defined
here; plotting helpers of the reference are out of scope.

Note the two different averages: these metrics are per-sequence means averaged over sequences
(utils.py:166-178), while the Keras training loss is a batch-level mean over real tokens.
"""
import numpy as np

epsilon = 10e-8


def chop_sequences(seqs, offset=200):
    return [s[:offset] for s in seqs]


def multinomial_probabilities(seqs, n, k=1.0, normalize=True):
    """Smoothed unigram model, shape (1, n)."""
    flat = np.fromiter((s for seq in seqs for s in seq), dtype=np.int64)
    counts = np.bincount(flat, minlength=n).astype(np.float64).reshape(1, n)
    total = counts.sum(axis=1)
    out = counts + k
    return out / (total + n * k) if normalize else out


def transition_matrix(seqs, n, k=0, freq=False, end_state=True):
    """First-order transition counts (+k smoothing); optional end-state column; initial-state vector."""
    alpha = np.zeros((n, n + 1 if end_state else n))
    gamma = np.zeros(n)
    for seq in seqs:
        if len(seq) > 1:
            a = np.asarray(seq[:-1], dtype=np.int64)
            b = np.asarray(seq[1:], dtype=np.int64)
            np.add.at(alpha, (a, b), 1.0)
            if end_state:
                alpha[seq[-1], n] += 1
        elif end_state:
            alpha[seq[0], n] += 1
        gamma[seq[0]] += 1
    sa, sg = alpha + k, gamma + k
    if not freq:
        sa = sa / (alpha.sum(axis=1).reshape((n, 1)) + n * k)
        sg = sg / (gamma.sum() + n * k)
    return sa, sg


def neg_log_likelihood(probs):
    return -np.sum(np.log(probs))


def compute_likelihood_cut(predictions, train_percent, orig_lengths=None, count_first_prob=False):
    """(mean NLL of the first ceil(p*L) steps, mean NLL of the last floor((1-p)*L) steps), each a
    mean over sequences of per-sequence means."""
    assert train_percent <= 1.0, "ERROR: train_percent should be <= 1.0"
    head, tail = [], []
    for i, pred in enumerate(predictions):
        p = pred[:] if count_first_prob else pred[1:]
        if orig_lengths is not None:
            p = pred[-int(orig_lengths[i]):]
        L = len(p)
        n_head = int(np.ceil(train_percent * L))
        n_tail = int(np.floor((1.0 - train_percent) * L))
        if n_head > 0:
            head.append(neg_log_likelihood(p[0:n_head]) / n_head)
        if n_tail > 0:
            tail.append(neg_log_likelihood(p[-n_tail:]) / n_tail)
    return np.sum(head) / len(head), np.sum(tail) / len(tail)


def compute_likelihood(predictions, count_first_prob=False):
    eps = 1e-07
    out = []
    for pred in predictions:
        p = np.clip(pred[:] if count_first_prob else pred[1:], eps, 1.0 - eps)
        if len(p) > 0:
            out.append(neg_log_likelihood(p) / len(p))
    return np.mean(out)


def compute_unique_elements(seqs):
    return len({s for seq in seqs for s in seq})


def compute_seq_max_length(seqs):
    return max((len(s) for s in seqs), default=0)


def sample_weights(alpha, sigma):
    return np.random.normal(alpha, sigma)


def recall_at_k(ranks, k=20):
    """Extension: fraction of tokens whose target has fewer than k items scoring above it."""
    ranks = np.asarray(ranks)
    return float(np.mean(ranks < k)) if ranks.size else 0.0
