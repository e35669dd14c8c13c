"""The two dataset pieces that sit directly in front of the hot path: the MSNBC text
format (datasets.py:199-227) and the history features ``xs`` of the full model (datasets.py:97-113).
The other parsers, the download script and the split helpers of the reference's datasets.py are
out of scope.  Same function names, arguments and return values as the reference."""
import numpy as np


def load_msnbc_data(eliminate_repeats=False, path="data/msnbc-data.txt"):
    """MSNBC.com anonymous web data: 8 header lines, then one session per line as whitespace-separated
    page-category tokens.  Tokens are mapped to ids in order of first appearance.
    eliminate_repeats drops a token equal to the one kept before it.  -> (seqs, vocab)"""
    vocab = {}
    seqs = []
    with open(path, "r") as f:
        for lineno, line in enumerate(f):
            if lineno < 8:
                continue
            seq = []
            last = None
            for tok in line.split():
                if tok not in vocab:
                    vocab[tok] = len(vocab)
                if eliminate_repeats and tok == last:
                    continue
                seq.append(vocab[tok])
                last = tok
            seqs.append(seq)
    return seqs, vocab


def build_xs(sequences, vocab, freq=False):
    """History features: xs[i][t] = multi-hot (freq: counts) over the vocabulary of sequence i's items
    0..t.  Returned as a list of (len_i, |vocab|) lists like the reference; the HBM-resident form is
    seqrec_history_features (engine.Engine.upload_device(history=True))."""
    V = len(vocab)
    out = []
    for seq in sequences:
        s = np.asarray(seq, dtype=np.int64)
        hot = np.zeros((len(s), V), dtype=np.int64)
        if len(s):
            hot[np.arange(len(s)), s] = 1
            hot = np.cumsum(hot, axis=0)
            if not freq:
                hot = np.minimum(hot, 1)
        out.append(hot.tolist())
    return out


def to_flat(sequences):
    """list of id lists -> (flat int32 ids, int64 starts[n+1]) for Engine.put_dataset / batching.pack_flat."""
    lens = np.fromiter((len(s) for s in sequences), dtype=np.int64, count=len(sequences))
    starts = np.zeros(len(sequences) + 1, dtype=np.int64)
    np.cumsum(lens, out=starts[1:])
    flat = np.fromiter((v for s in sequences for v in s), dtype=np.int32, count=int(starts[-1]))
    return flat, starts
