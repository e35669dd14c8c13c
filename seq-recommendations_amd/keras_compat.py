"""The handful of Keras 2.0 collaborator types the reference's callers pass into the hot path
(experiments_methods.py:8-9,30,37,41; preprocessor.py:4-5), as plain Python -- no Keras.

Only behaviour that the reference's call sites rely on is reproduced:
  Adagrad(lr, epsilon, decay, clipnorm)       parameter holder; the arithmetic is in csrc/ops.hip
  EarlyStopping(monitor, min_delta, patience, verbose, mode)
  ModelCheckpoint(filepath, monitor, save_weights_only, save_best_only)
  Callback / History                          the callback protocol of Model.fit
  pad_sequences, to_categorical               used by preprocessor.py
  initializers                                glorot_uniform / glorot_normal / orthogonal / ...
"""
import numpy as np


# --------------------------------------------------------------------------- optimizers
class Adagrad:
    """keras.optimizers.Adagrad: a += g^2 ; p -= lr * g / (sqrt(a) + epsilon), after the
    global-norm clip ``g *= clipnorm / max(norm, clipnorm)`` over ALL trainable tensors."""

    def __init__(self, lr=0.01, epsilon=1e-08, decay=0.0, clipnorm=None, **kwargs):
        if decay:
            raise NotImplementedError("learning-rate decay is never used by the reference (decay=0.0)")
        self.lr = float(lr)
        self.epsilon = float(epsilon)
        self.decay = float(decay)
        self.clipnorm = None if clipnorm is None else float(clipnorm)


# --------------------------------------------------------------------------- callbacks
class Callback:
    def __init__(self):
        self.model = None
        self.params = {}

    def set_model(self, model):
        self.model = model

    def set_params(self, params):
        self.params = params

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass


class History(Callback):
    def on_train_begin(self, logs=None):
        self.epoch = []
        self.history = {}

    def on_epoch_end(self, epoch, logs=None):
        self.epoch.append(epoch)
        for k, v in (logs or {}).items():
            self.history.setdefault(k, []).append(v)


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", min_delta=0, patience=0, verbose=0, mode="auto"):
        Callback.__init__(self)
        self.monitor, self.patience, self.verbose = monitor, patience, verbose
        self.min_delta = abs(min_delta)
        if mode == "max" or (mode == "auto" and "acc" in monitor):
            self.better = lambda cur, best: cur - self.min_delta > best
            self.best0 = -np.inf
        else:
            self.better = lambda cur, best: cur + self.min_delta < best
            self.best0 = np.inf
        self.stopped_epoch = 0

    def on_train_begin(self, logs=None):
        self.wait = 0
        self.best = self.best0

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if self.better(cur, self.best):
            self.best = cur
            self.wait = 0
        else:
            if self.wait >= self.patience:
                self.stopped_epoch = epoch
                self.model.stop_training = True
            self.wait += 1


class ModelCheckpoint(Callback):
    """Saves the model's weights (never the graph) after an epoch; ``filepath`` may contain
    ``{epoch:02d}`` and any key of the epoch logs, as in experiments_methods.py:30."""

    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=False,
                 mode="auto", period=1):
        Callback.__init__(self)
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.period = save_best_only, period
        self.maximise = mode == "max" or (mode == "auto" and "acc" in monitor)
        self.best = -np.inf if self.maximise else np.inf
        self.since = 0
        self.saved = []

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        self.since += 1
        if self.since < self.period:
            return
        self.since = 0
        path = self.filepath.format(epoch=epoch + 1, **logs)
        if self.save_best_only:
            cur = logs.get(self.monitor)
            if cur is None:
                return
            if (cur > self.best) if self.maximise else (cur < self.best):
                self.best = cur
            else:
                return
        self.model.save_weights(path)
        self.saved.append(path)


# --------------------------------------------------------------------------- preprocessing helpers
class l2:
    """keras.regularizers.l2 (imported by model.py:19, experiments_server.py:7): l * sum(w^2)."""

    def __init__(self, l=0.01):
        self.l = float(l)

    def terms(self):
        """-> (means or None, strength) of the penalty strength * sum (w - means)^2"""
        return None, self.l

    def __call__(self, x):
        return float(self.l * np.sum(np.square(np.asarray(x, dtype=np.float64))))

    def get_config(self):
        return {"l2": self.l}


def to_categorical(y, num_classes=None):
    y = np.asarray(y, dtype=np.int64).ravel()
    if num_classes is None:
        num_classes = int(y.max()) + 1
    out = np.zeros((y.shape[0], num_classes))
    out[np.arange(y.shape[0]), y] = 1.0
    return out


def pad_sequences(sequences, maxlen=None, dtype="int32", padding="pre", truncating="pre", value=0.0):
    """keras.preprocessing.sequence.pad_sequences for lists of (lists of scalars | lists of vectors)."""
    n = len(sequences)
    lengths = [len(s) for s in sequences]
    if maxlen is None:
        maxlen = max(lengths) if lengths else 0
    sample = ()
    for s in sequences:
        if len(s):
            sample = np.asarray(s).shape[1:]
            break
    out = np.full((n, maxlen) + tuple(sample), value, dtype=dtype)
    for i, s in enumerate(sequences):
        if not len(s):
            continue
        t = np.asarray(s[-maxlen:] if truncating == "pre" else s[:maxlen], dtype=dtype)
        if padding == "pre":
            out[i, maxlen - len(t):] = t
        else:
            out[i, :len(t)] = t
    return out


# --------------------------------------------------------------------------- initializers
class RandomUniform:
    def __init__(self, minval=-0.05, maxval=0.05, seed=None):
        self.minval, self.maxval, self.seed = minval, maxval, seed

    def __call__(self, shape, dtype=None):
        rs = np.random.RandomState(self.seed) if self.seed is not None else np.random
        return rs.uniform(self.minval, self.maxval, size=shape)


def _fans(shape):
    return shape[0], shape[1]


def initialize(spec, shape):
    """Keras initializer by name / object -> float32 array.  Objects are called as
    ``spec(shape)`` (so the reference's ArrayInitializer, model.py:32-45, just returns its array)."""
    if callable(spec) and not isinstance(spec, str):
        return np.asarray(spec(shape), dtype=np.float32).reshape(shape)
    fi, fo = _fans(shape) if len(shape) == 2 else (shape[0], shape[0])
    if spec in ("glorot_uniform", None):
        lim = np.sqrt(6.0 / (fi + fo))
        w = np.random.uniform(-lim, lim, size=shape)
    elif spec == "glorot_normal":
        w = np.random.normal(0.0, np.sqrt(2.0 / (fi + fo)), size=shape)
    elif spec == "random_uniform":
        w = np.random.uniform(-0.05, 0.05, size=shape)
    elif spec == "orthogonal":
        a = np.random.normal(0.0, 1.0, (shape[0], shape[1]))
        u, _, v = np.linalg.svd(a, full_matrices=False)
        w = u if u.shape == tuple(shape) else v
    elif spec == "zeros":
        w = np.zeros(shape)
    elif spec == "ones":
        w = np.ones(shape)
    else:
        raise ValueError("unknown initializer %r" % (spec,))
    return np.asarray(w, dtype=np.float32)
