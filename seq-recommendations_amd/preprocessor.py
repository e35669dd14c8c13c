"""Batching step in front of the hot path: the reference's preprocessors (preprocessor.py:9-127)
without Keras -- same class names, arguments and output tensors, built with array ops instead of
Python lists of one-hot lists.

  x_i = seq[i], y_i = seq[i+1]  (a session of n items gives n-1 steps),  PRE-padded to seq_length.
"""
import numpy as np

from .keras_compat import pad_sequences


class Preprocessor:
    def __init__(self, vocab, pad_value=0., seq_length=None, sparse=False):
        self.vocab = vocab
        self.seq_length = seq_length
        self.pad_value = pad_value
        self.sparse = sparse

    def _pad_sequences(self, sequences, dtype="int32"):
        padded = pad_sequences(sequences, maxlen=self.seq_length, dtype=dtype, padding="pre", truncating="pre",
                               value=self.pad_value)
        self.seq_length = padded.shape[1]
        return padded

    def _pairs(self, sequences, width):
        """Padded (N, T, width) one-hot tensors of inputs seq[:-1] and targets seq[1:]."""
        n = len(sequences)
        lens = np.array([max(len(s) - 1, 0) for s in sequences], dtype=np.int64)
        T = self.seq_length if self.seq_length is not None else (int(lens.max()) if n else 0)
        self.seq_length = T
        x = np.full((n, T, width), self.pad_value, dtype=np.float64)
        y = np.full((n, T, width), self.pad_value, dtype=np.float64)
        for b, s in enumerate(sequences):
            s = np.asarray(s[-(T + 1):] if len(s) > T + 1 else s, dtype=np.int64)      # truncating='pre'
            L = len(s) - 1
            if L <= 0:
                continue
            t = np.arange(T - L, T)
            x[b, t] = 0.0
            y[b, t] = 0.0
            x[b, t, s[:-1]] = 1.0
            y[b, t, s[1:]] = 1.0
        return x, y

    def transform_data(self, sequences, xs=None, pad=True):
        pass


class BaselinePreprocessor(Preprocessor):
    """one-hot(seq[i]) [+ history features xs[i]] -> one-hot(seq[i+1]); empty sessions dropped."""

    def __init__(self, vocab, pad_value=0., seq_length=None):
        Preprocessor.__init__(self, vocab, pad_value, seq_length)

    def transform_data(self, sequences, xs=None, pad=True):
        V = len(self.vocab)
        keep = [i for i, s in enumerate(sequences) if len(s) > 1]
        seqs = [sequences[i] for i in keep]
        if not pad:
            eye = np.eye(V)
            xd, yd = [], []
            for j, s in zip(keep, seqs):
                xi = [eye[s[i]].tolist() + (list(xs[j][i]) if xs is not None else []) for i in range(len(s) - 1)]
                xd.append(xi)
                yd.append([eye[s[i + 1]].tolist() for i in range(len(s) - 1)])
            return xd, yd
        x, y = self._pairs(seqs, V)
        if xs is not None:
            T = self.seq_length
            f = np.full((len(seqs), T, V), self.pad_value, dtype=np.float64)
            for b, j in enumerate(keep):
                L = min(len(seqs[b]) - 1, T)
                f[b, T - L:] = np.asarray(xs[j], dtype=np.float64)[len(seqs[b]) - 1 - L:len(seqs[b]) - 1]
            x = np.concatenate([x, f], axis=2)
        return x, y


class FullModelPreprocessor(Preprocessor):
    """Returns (x, y, xs) padded tensors; ``sparse=True`` keeps ids as (N, T, 1) instead of one-hots."""

    def __init__(self, vocab, pad_value=0., seq_length=None, sparse=False):
        Preprocessor.__init__(self, vocab, pad_value, seq_length, sparse=sparse)

    def transform_data(self, sequences, xs, pad=True):
        V = len(self.vocab)
        if self.sparse:
            x = self._pad_sequences([[[v] for v in s[:-1]] for s in sequences], dtype=np.float64)
            y = self._pad_sequences([[[v] for v in s[1:]] for s in sequences], dtype=np.float64)
            x = x.reshape(len(sequences), self.seq_length, 1)
            y = y.reshape(len(sequences), self.seq_length, 1)
        else:
            x, y = self._pairs(sequences, V)
        c = self._pad_sequences([list(f[:-1]) for f in xs], dtype=np.float64)
        c = np.reshape(c, (len(c), self.seq_length, V))
        return x, y, c

    def gen_data(self, sequences, xs, with_xs=True, with_x=True, batch_size=100):
        """Endless batches of ``batch_size`` consecutive sessions (the reference's generator re-reads
        rows 0..batch_size-1 forever, preprocessor.py:105-114 -- a bug that is not reproduced)."""
        n = len(sequences)
        start = 0
        while True:
            idx = [(start + i) % n for i in range(batch_size)]
            start = (start + batch_size) % n
            x, y, c = self.transform_data([sequences[i] for i in idx], [xs[i] for i in idx])
            if with_xs and with_x:
                yield [x, c], y
            elif with_xs:
                yield c, y
            elif with_x:
                yield x, y
