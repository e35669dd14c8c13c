"""The reference's model surface (model.py:23-45,94-258,322-403) on the HIP engine.

Same class names, constructor arguments, method names and return conventions as the
reference, so its callers (experiments_methods.py, the L5 scripts) work unchanged:

  RNNBaseline(timesteps, features, n_classes, ...)                     model.py:241-258
  RNNFullModel(timesteps, x_dim, y_dim, z_dim, ..., y_to_z, y_to_y, x_to_y, x_to_z, ...)   model.py:322-403
      -- the "ytoz" wiring (y_to_z only) is the hot path, with all three dropouts (y_to_z,
         z_to_z = Keras recurrent_dropout, z_to_y); the side branches are dense V x V terms for the
         reference's small vocabularies: y_to_y (row of a V x V kernel added to the logits, optionally
         frozen at a log-transition-count initialisation), x_to_y (history features through the same
         output Dense as z, OnlyNonZeroDiagonal re-applied after every update), x_to_z (history
         features concatenated into the cell input); kernel regularizers (GaussPriorRegularizer, l2) on
         the y_to_y and to_y Dense kernels.
  BaseRNNModel.compile_model / fit_model / fit_generator / predict / evaluate /
      save_model_weights / load_model_weights / get_layer_weights / set_layer_weights_trainable /
      set_layer_weights / get_model_weights / get_activations              model.py:170-238
  .model        a Keras-``Model``-shaped object (fit / predict / evaluate / get_layer / ...)
  ModelResults, ArrayInitializer, ValLossHistoryCut, MultinomialModel, MarkovModel

What differs, by necessity: no Keras/Theano underneath -- ``.model`` is ``SeqModel`` below, whose
``fit`` drives ``engine.Engine`` (HIP kernels through the C ABI).  Weight files are numpy ``.npz``
containers (``weight0..weightN`` keys, the reference's naming, model.py:206) because h5py is not
part of the image.
"""
import os

import numpy as np

from . import utils
from . import batching
from .keras_compat import Adagrad, Callback, History, initialize



def _lib_act():
    """Names of the cell activations the scans implement (_lib.ACT)."""
    from ._lib import ACT
    return ACT

class ModelResults:
    def __init__(self, train_loss=None, val_loss=None, epoch=None):
        self.val_loss = val_loss
        self.train_loss = train_loss
        self.epoch = epoch


class ArrayInitializer:
    """Initializer that returns a fixed array (model.py:32-45): the hook for injecting identical
    initial weights into the oracle and the GPU path."""

    def __init__(self, values=0):
        self.values = values

    def __call__(self, shape, dtype=None):
        return self.values

    def get_config(self):
        return {"value": self.values}


class GaussPriorRegularizer:
    """model.py:71-91: penalty sum(1 / (2 var) * (w - means)^2) on a Dense kernel; Keras adds it to the
    loss it reports and differentiates.  Evaluated on the device by seqrec_prior_grad."""

    def __init__(self, means, var):
        self.means = np.asarray(means, dtype=np.float32)
        self.var = var

    def terms(self):
        return self.means, 1.0 / (2.0 * float(self.var))

    def __call__(self, x):
        return float(np.sum(1.0 / (2.0 * float(self.var)) * np.square(np.asarray(x, dtype=np.float64) - self.means)))

    def get_config(self):
        return {"var": float(self.var), "means": self.means}


def gauss_prior(means, var):
    return GaussPriorRegularizer(means, var)


# ------------------------------------------------------------------------------------------------
# count baselines (model.py:127-167) -- tiny, numpy; kept so experiments_methods drops in whole
# ------------------------------------------------------------------------------------------------
class BaseModel:
    def __init__(self, n_classes, model_name="test_model"):
        self.n_classes = n_classes
        self.model_name = model_name
        self.model = None


class MultinomialModel(BaseModel):
    """Count baseline (reference model.py:127-143, same constructor keywords): smoothed unigram probabilities."""

    def __init__(self, n_classes, model_name="multinomial_model", k=1.0):
        super().__init__(n_classes, model_name)
        self.k = k
        self.model = np.zeros((1, n_classes))                    # row vector of item probabilities once fitted

    def fit_model(self, seqs, normalize=True):
        self.model = utils.multinomial_probabilities(seqs, self.n_classes, self.k, normalize)

    def predict(self, seqs):
        return [[self.model[0, s] for s in seq] for seq in seqs]


class MarkovModel(BaseModel):
    """Count baseline (reference model.py:146-167, same constructor keywords): first-order transition probabilities."""

    def __init__(self, n_classes, model_name="markov_model", order=1, k=1.0):
        super().__init__(n_classes, model_name)
        if order != 1:
            raise ValueError("MarkovModel: order %r is not available, the transition matrix is first-order" % (order,))
        self.order, self.k = order, k
        self.initial_probs = np.zeros(n_classes)
        self.model = np.zeros((n_classes, n_classes))

    def fit_model(self, seqs, freq=False):
        self.model, self.initial_probs = utils.transition_matrix(seqs, self.n_classes, self.k, freq=freq, end_state=False)

    def predict(self, seqs):
        out = []
        for seq in seqs:
            p = [self.initial_probs[seq[0]]]
            p.extend(self.model[i, j] for i, j in zip(seq[:-1], seq[1:]))
            out.append(p)
        return out


# ------------------------------------------------------------------------------------------------
# the Keras-Model-shaped object behind `.model`
# ------------------------------------------------------------------------------------------------
class Layer:
    """Named bundle of parameters with Keras' get_weights/set_weights/trainable protocol."""

    def __init__(self, owner, name, keys):
        self._owner, self.name, self.keys = owner, name, keys
        self._trainable = True

    @property
    def trainable(self):
        return self._trainable

    @trainable.setter
    def trainable(self, v):
        self._trainable = bool(v)
        self._owner._sync_trainable()

    def get_weights(self):
        return [self._owner._get(k) for k in self.keys]

    def set_weights(self, weights):
        if len(weights) != len(self.keys):
            raise ValueError("layer %s expects %d arrays, got %d" % (self.name, len(self.keys), len(weights)))
        for k, w in zip(self.keys, weights):
            self._owner._set(k, w)


class SeqModel:
    """Masking -> [Dropout] -> SimpleRNN|LSTM|GRU -> Dropout -> TimeDistributed(Dense) -> softmax.

    Host-side weights live in ``self.w`` (UNPADDED numpy, Keras layouts) until the first call that
    needs the GPU creates the engine; afterwards the engine's HBM copy is authoritative."""

    metrics_names = ["loss", "categorical_crossentropy"]

    def __init__(self, timesteps, in_dim, n_classes, z_dim, rnn_type, activation, rnn_name, out_name, use_bias=True,
                 out_bias=False, drop_in=0.0, drop_rec=0.0, drop_out=0.0, kernel_initializer="glorot_uniform",
                 device="cuda:0", y_dim=None, x_dim=0, y_to_z=True, x_to_z=False, y_to_y=False, x_to_y=False,
                 diag_b=True, ytoy_bias=False, y_to_y_w_initializer="glorot_uniform", y_to_y_regularizer=None,
                 toy_regularizer=None, frozen_keys=(), toy_reg_on_x=True):
        cell = {"simpleRNN": "simplernn", "LSTM": "lstm", "GRU": "gru"}.get(rnn_type)
        if cell is None:
            raise ValueError("rnn_type must be 'simpleRNN', 'LSTM' or 'GRU' (got %r)" % (rnn_type,))
        if activation not in _lib_act():
            # Keras 2.0's element-wise list is covered (relu, tanh, linear, sigmoid, hard_sigmoid, softplus, softsign, elu); `softmax`
            # as a CELL activation normalises over the hidden units -- not element-wise, and in no call of the reference
            raise NotImplementedError("activation %r: the HIP scans implement %s" % (activation, ", ".join(sorted(_lib_act()))))
        self.timesteps, self.in_dim, self.n_classes, self.z_dim = timesteps, in_dim, n_classes, z_dim
        self.cell, self.activation = cell, activation
        self.use_bias, self.out_bias = use_bias, out_bias
        self.drop_in, self.drop_rec, self.drop_out = float(drop_in), float(drop_rec), float(drop_out)
        self.device = device
        # RNNFullModel wiring (model.py:322-403): which inputs feed the cell (y_to_z / x_to_z) and
        # which terms are added to the logits (x_to_y through the same Dense as z; y_to_y directly)
        self.y_dim = in_dim if y_dim is None else y_dim
        self.x_dim = x_dim
        self.y_to_z, self.x_to_z, self.y_to_y, self.x_to_y = y_to_z, x_to_z, y_to_y, x_to_y
        self.diag_b, self.ytoy_bias = diag_b, ytoy_bias
        self.uses_y = y_to_y or y_to_z
        self.uses_x = x_to_y or x_to_z
        G = {"simplernn": 1, "gru": 3, "lstm": 4}[cell]
        H = z_dim
        w = {"Wk": initialize(kernel_initializer, (in_dim, G * H)), "U": self._recurrent_init(G, H)}
        if use_bias:
            b = np.zeros(G * H, np.float32)
            if cell == "lstm":
                b[H:2 * H] = 1.0                       # unit_forget_bias=True
            w["b"] = b
        if x_to_y:      # ONE Dense over concat([z, x]) (model.py:375-384): glorot over the joint fan-in
            k = initialize("glorot_uniform", (H + x_dim, n_classes))
            w["Wout"], w["Wxy"] = k[:H].copy(), k[H:].copy()
            if diag_b:
                w["Wxy"] = w["Wxy"] * np.eye(x_dim, n_classes, dtype=np.float32)
        else:
            w["Wout"] = initialize("glorot_uniform", (H, n_classes))
        if out_bias:
            w["bout"] = np.zeros(n_classes, np.float32)
        if y_to_y:
            w["Wyy"] = initialize(y_to_y_w_initializer, (self.y_dim, n_classes))
            if ytoy_bias:
                w["byy"] = np.zeros(n_classes, np.float32)
        self.w = w
        rk = ["Wk", "U"] + (["b"] if use_bias else [])
        ok = [("Wout+Wxy" if x_to_y else "Wout")] + (["bout"] if out_bias else [])
        self.layers = [Layer(self, rnn_name, rk), Layer(self, out_name, ok)]
        if y_to_y:
            self.layers.append(Layer(self, "y_to_y_output", ["Wyy"] + (["byy"] if ytoy_bias else [])))
        for r in (y_to_y_regularizer, toy_regularizer):
            if r is not None and not hasattr(r, "terms"):
                raise NotImplementedError("kernel regularizer %r: GaussPriorRegularizer and l2 are implemented" % (r,))
        self.y_to_y_regularizer = y_to_y_regularizer if y_to_y else None
        self.toy_regularizer = toy_regularizer
        self.toy_reg_on_x = bool(toy_reg_on_x) and x_to_y      # the penalty covers concat([Wout; Wxy]) (model.py:375-384)
        self.frozen_keys = set(frozen_keys)
        self.engine = None
        self.input_mode = None
        self.optimizer = None
        self.loss = None
        self.stop_training = False
        self.seed = int(np.random.randint(0, 2 ** 31 - 1))     # dropout stream seed (Keras draws one too)
        self._step = 0

    @staticmethod
    def _recurrent_init(G, H):
        a = np.random.normal(0.0, 1.0, (H, G * H))
        u, _, v = np.linalg.svd(a, full_matrices=False)
        q = u if u.shape == (H, G * H) else v
        return np.asarray(q, dtype=np.float32)

    # ---- weights -------------------------------------------------------------------------------
    def _get(self, k):
        if k == "Wout+Wxy":
            return np.concatenate([self._get("Wout"), self._get("Wxy")], axis=0)
        return self.engine.get_param(k) if self.engine is not None else self.w[k].copy()

    def _set(self, k, v):
        v = np.asarray(v, dtype=np.float32)
        if k == "Wout+Wxy":
            self._set("Wout", v[: self.z_dim])
            self._set("Wxy", v[self.z_dim:])
            return
        if v.shape != self.w[k].shape:
            raise ValueError("weight %s: shape %s expected, got %s" % (k, self.w[k].shape, v.shape))
        self.w[k] = v.copy()
        if self.engine is not None:
            self.engine.set_param(k, v)

    def _prior_names(self):
        names = []
        if self.y_to_y_regularizer is not None:
            names.append("Wyy")
        if self.toy_regularizer is not None:
            names += ["Wout"] + (["Wxy"] if self.toy_reg_on_x else [])
        return tuple(names)

    def _sync_trainable(self):
        if self.engine is not None:
            for l in self.layers:
                for k in l.keys:
                    for kk in k.split("+"):
                        self.engine.trainable[kk] = l.trainable and kk not in getattr(self, "frozen_keys", ())
            for kk in getattr(self, "frozen_keys", ()):
                if kk in self.engine.trainable:
                    self.engine.trainable[kk] = False

    def get_layer(self, name=None, index=None):
        if index is not None:
            return self.layers[index]
        for l in self.layers:
            if l.name == name:
                return l
        raise ValueError("No such layer: %s" % name)

    def get_weights(self):
        return [a for l in self.layers for a in l.get_weights()]

    def set_weights(self, weights):
        i = 0
        for l in self.layers:
            l.set_weights(weights[i:i + len(l.keys)])
            i += len(l.keys)

    @property
    def trainable_weights(self):
        return ["%s/%s" % (l.name, k) for l in self.layers if l.trainable for k in l.keys]

    @property
    def non_trainable_weights(self):
        return ["%s/%s" % (l.name, k) for l in self.layers if not l.trainable for k in l.keys]

    @staticmethod
    def _npz_path(filepath):
        """Weight files are numpy .npz containers: a path that claims HDF5 (.h5 / .hdf5, as the reference's callers
        pass) gets the true extension appended instead of holding npz bytes under an HDF5 name."""
        fp = str(filepath)
        return fp + ".npz" if fp.lower().endswith((".h5", ".hdf5")) else fp

    def save_weights(self, filepath):
        filepath = self._npz_path(filepath)
        d = os.path.dirname(filepath)
        if d and not os.path.exists(d):
            os.makedirs(d)
        with open(filepath, "wb") as f:
            np.savez(f, **{"weight%d" % i: a for i, a in enumerate(self.get_weights())})

    def load_weights(self, filepath, by_name=False):
        if not os.path.exists(filepath) and os.path.exists(self._npz_path(filepath)):
            filepath = self._npz_path(filepath)
        with np.load(filepath, allow_pickle=False) as z:
            self.set_weights([z["weight%d" % i] for i in range(len(z.files))])

    # ---- engine --------------------------------------------------------------------------------
    def _ensure_engine(self, input_mode):
        from . import engine as E
        if self.engine is not None and self.input_mode == input_mode:
            return self.engine
        if self.engine is not None:                 # input kind changed: carry the weights over
            for k in list(self.w):
                self.w[k] = self.engine.get_param(k)
        cfg = E.NetConfig(cell=self.cell, act=self.activation, H=self.z_dim, V_in=self.in_dim, V_out=self.n_classes,
                          input=input_mode, output="full", use_bias=self.use_bias, out_bias=self.out_bias,
                          drop_in=self.drop_in, drop_rec=self.drop_rec, drop_out=self.drop_out, seed=self.seed,
                          y_to_y=self.y_to_y, yy_bias=self.ytoy_bias, x_to_y=self.x_to_y, x_dim=self.x_dim,
                          diag_b=self.diag_b, priors=self._prior_names())
        self.engine = E.Engine(cfg, self.device)
        self.input_mode = input_mode
        for k, v in self.w.items():
            self.engine.set_param(k, v)
        if self.y_to_y_regularizer is not None:
            self.engine.set_prior("Wyy", *self.y_to_y_regularizer.terms())
        if self.toy_regularizer is not None:            # ONE penalty over the to_y kernel concat([Wout; Wxy])
            means, strength = self.toy_regularizer.terms()
            H = self.z_dim
            self.engine.set_prior("Wout", None if means is None else means[:H], strength)
            if self.toy_reg_on_x:
                self.engine.set_prior("Wxy", None if means is None else means[H:], strength)
        self._sync_trainable()
        return self.engine

    # ---- data ----------------------------------------------------------------------------------
    @staticmethod
    def _first(x):
        return x[0] if isinstance(x, (list, tuple)) else x

    def _prepare(self, x, y=None):
        """Reference tensors -> (mask, ids|None, feats|None, tgt|None, input_mode, xs|None).
        ``x`` is the Keras input list: [y_input] and/or [x_input] in that order (model.py:398-402)."""
        xl = list(x) if isinstance(x, (list, tuple)) else [x]
        want = int(self.uses_y) + int(self.uses_x)
        if len(xl) != want:
            raise ValueError("this model takes %d input array(s), got %d" % (want, len(xl)))
        y_in = np.asarray(xl[0]) if self.uses_y else None
        x_in = np.asarray(xl[-1]) if self.uses_x else None
        if y_in is not None and (y_in.ndim != 3 or y_in.shape[2] != self.y_dim):
            raise ValueError("expected y input of shape (N, T, %d), got %s" % (self.y_dim, y_in.shape))
        if x_in is not None and (x_in.ndim != 3 or x_in.shape[2] != self.x_dim):
            raise ValueError("expected x input of shape (N, T, %d), got %s" % (self.x_dim, x_in.shape))
        ids = None
        exact = False
        if y_in is not None:
            mask, ids, exact = batching.onehot_to_ids(y_in)
        else:
            mask = np.any(x_in != 0, axis=2)
        if self.y_to_y and not exact:
            raise NotImplementedError("the y_to_y branch needs one-hot y inputs")
        if self.y_to_z and not self.x_to_z:
            feats = None if exact else y_in
        elif self.x_to_z and not self.y_to_z:
            feats = x_in
        else:
            feats = np.concatenate([y_in, x_in], axis=2)          # concatenate([masked_y, masked_x]) (model.py:354)
        mode = "onehot" if feats is None else "dense"
        tgt = None
        if y is not None:
            y = np.asarray(y)
            tgt = y[:, :, 0].astype(np.int64) if y.shape[2] == 1 and self.n_classes != 1 else np.argmax(y, axis=2)
        return mask, (ids if exact else None), feats, tgt, mode, (x_in if self.x_to_y else None)

    def _batch(self, prep, idx):
        mask, ids, feats, tgt, _, xs = prep
        rb, tcol = batching.pack_padded(mask[idx], None if ids is None else ids[idx], None if tgt is None else tgt[idx],
                                        None if feats is None else feats[idx], None if xs is None else xs[idx])
        return rb, tcol

    # ---- Keras Model API -----------------------------------------------------------------------
    def compile(self, loss="categorical_crossentropy", optimizer=None, metrics=None):
        if loss not in ("categorical_crossentropy", "sparse_categorical_crossentropy"):
            raise NotImplementedError("loss %r" % (loss,))
        if optimizer is None or isinstance(optimizer, str):
            optimizer = Adagrad()
        if not isinstance(optimizer, Adagrad):
            raise NotImplementedError("only Adagrad (experiments_methods.py:41) is implemented on the device")
        self.loss, self.optimizer = loss, optimizer

    def _run_epoch_batches(self, prep, order, batch_size, train):
        import torch
        eng = self._ensure_engine(prep[4])
        N = len(order)
        opt = self.optimizer
        tot = torch.zeros(1, device=eng.dev)
        for s in range(0, N, batch_size):
            idx = order[s:s + batch_size]
            rb, _ = self._batch(prep, idx)
            d = eng.upload(rb)
            if train:
                l = eng.train_step(d, lr=opt.lr, eps=opt.epsilon, clipnorm=opt.clipnorm, step=self._step)
                self._step += 1
            else:
                l = eng.eval_loss(d)
            tot += l * float(len(idx))
        out = float(tot.item()) / max(N, 1)
        eng.check_status()                 # the epoch's host sync: raise if a kernel reported a failure (Engine.check_status)
        return out

    def fit(self, x, y, validation_data=None, epochs=10, batch_size=100, verbose=1, callbacks=None, shuffle=True):
        if self.optimizer is None:
            raise RuntimeError("You must compile a model before training/testing.")
        prep = self._prepare(x, y)
        vprep = None
        if validation_data is not None:
            vprep = self._prepare(validation_data[0], validation_data[1])
        hist = History()
        cbs = [hist] + list(callbacks or [])
        for cb in cbs:
            cb.set_model(self)
            cb.set_params({"epochs": epochs, "batch_size": batch_size, "verbose": verbose})
        self.stop_training = False
        logs0 = {}
        for cb in cbs:
            cb.on_train_begin(logs0)
        N = prep[0].shape[0]
        index = np.arange(N)
        for epoch in range(epochs):
            for cb in cbs:
                cb.on_epoch_begin(epoch, {})
            if shuffle:
                np.random.shuffle(index)
            logs = {"loss": self._run_epoch_batches(prep, index, batch_size, True)}
            if vprep is not None:
                logs["val_loss"] = self._run_epoch_batches(vprep, np.arange(vprep[0].shape[0]), batch_size, False)
            for k, v in logs0.items():          # e.g. "my_loss" seeded by ValLossHistoryCut.on_train_begin
                logs.setdefault(k, v)
            for cb in cbs:
                cb.on_epoch_end(epoch, logs)
            if verbose:
                print("Epoch %d/%d - " % (epoch + 1, epochs) + " - ".join("%s: %.4f" % kv for kv in logs.items()))
            if self.stop_training:
                break
        for cb in cbs:
            cb.on_train_end({})
        return hist

    def fit_generator(self, generator, steps_per_epoch, epochs=1, verbose=1, callbacks=None, validation_data=None,
                      validation_steps=None):
        """Keras fit_generator: each step pulls ONE (x, y) batch from the generator."""
        import torch
        if self.optimizer is None:
            raise RuntimeError("You must compile a model before training/testing.")
        hist = History()
        cbs = [hist] + list(callbacks or [])
        for cb in cbs:
            cb.set_model(self)
            cb.on_train_begin({})
        opt = self.optimizer
        self.stop_training = False
        for epoch in range(epochs):
            tot = n = 0.0
            for _ in range(steps_per_epoch):
                xb, yb = next(generator)[:2]
                prep = self._prepare(xb, yb)
                eng = self._ensure_engine(prep[4])
                nb = prep[0].shape[0]
                rb, _ = self._batch(prep, np.arange(nb))
                l = eng.train_step(eng.upload(rb), lr=opt.lr, eps=opt.epsilon, clipnorm=opt.clipnorm, step=self._step)
                self._step += 1
                tot += float(l.item()) * nb
                n += nb
            logs = {"loss": tot / max(n, 1)}
            if steps_per_epoch:
                eng.check_status()
            if validation_data is not None:
                if hasattr(validation_data, "__next__"):
                    vt = vn = 0.0
                    for _ in range(validation_steps or 1):
                        xv, yv = next(validation_data)[:2]
                        vt += self.evaluate(xv, yv, batch_size=len(self._first(xv)))[0] * len(self._first(xv))
                        vn += len(self._first(xv))
                    logs["val_loss"] = vt / max(vn, 1)
                else:
                    logs["val_loss"] = self.evaluate(validation_data[0], validation_data[1])[0]
            for cb in cbs:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in cbs:
            cb.on_train_end({})
        return hist

    def evaluate(self, x, y, batch_size=32, verbose=0, sample_weight=None):
        prep = self._prepare(x, y)
        loss = self._run_epoch_batches(prep, np.arange(prep[0].shape[0]), batch_size, False)
        return [loss, loss]

    def predict(self, x, batch_size=32, verbose=0):
        """(N, T, n_classes) softmax outputs, pad positions included like Keras' Model.predict: the masked scan carries
        its state through a pad step and the per-step Dense still runs there, so a pad step shows
        softmax(Wout . h_carried + bout) -- the previous step's output, or softmax(bout) before the first real step.
        With the y_to_y / x_to_y branches the previous step's output also contained that step's Wyy row / feature term,
        which a pad step (all-zero unmasked inputs) does not have: those rows are recomputed from the carried state
        (+ the y_to_y bias), as the reference graph does (model.py:375-397)."""
        prep = self._prepare(x)
        mask = prep[0]
        eng = self._ensure_engine(prep[4])
        N, T = mask.shape
        V = self.n_classes
        out = np.zeros((N, T, V), np.float32)
        for s in range(0, N, batch_size):
            idx = np.arange(s, min(N, s + batch_size))
            rb, tcol = self._batch(prep, idx)
            if rb.n_tok == 0:
                continue
            pr = eng.predict_rows(eng.upload(rb)).cpu().numpy()
            out[idx[rb.tok_b], tcol] = pr
        if (self.y_to_y or self.x_to_y) and not mask.all():
            import torch
            hc = self.hidden(x, batch_size=batch_size)                     # carried state at every position (zeros before the first step)
            pi, pt = np.nonzero(~mask)
            Hp = np.zeros((len(pi), eng.Hp), np.float32)
            Hp[:, : self.z_dim] = hc[pi, pt]
            out[pi, pt] = eng.probs_from_hidden(torch.from_numpy(Hp).to(eng.dev)).cpu().numpy()
            return out
        b = self._get("bout") if self.out_bias else np.zeros(V, np.float32)
        p0 = np.exp(b - b.max())
        p0 = (p0 / p0.sum()).astype(np.float32)
        prev = np.broadcast_to(p0, (N, V)).copy()
        for t in range(T):
            m = mask[:, t]
            out[~m, t] = prev[~m]
            prev = out[:, t].copy()
        return out

    def hidden(self, x, batch_size=32):
        prep = self._prepare(x)
        mask = prep[0]
        eng = self._ensure_engine(prep[4])
        N, T = mask.shape
        out = np.zeros((N, T, self.z_dim), np.float32)
        for s in range(0, N, batch_size):
            idx = np.arange(s, min(N, s + batch_size))
            rb, tcol = self._batch(prep, idx)
            if rb.n_tok == 0:
                continue
            h = eng.hidden_rows(eng.upload(rb)).cpu().numpy()[:, :self.z_dim]
            out[idx[rb.tok_b], tcol] = h
        for t in range(1, T):
            m = mask[:, t]
            out[~m, t] = out[~m, t - 1]
        return out


# ------------------------------------------------------------------------------------------------
# the reference's wrappers
# ------------------------------------------------------------------------------------------------
class BaseRNNModel(BaseModel):
    def __init__(self, n_classes, model_name="test_model", rnn_type="simpleRNN"):
        BaseModel.__init__(self, n_classes, model_name)
        self.rnn_type = rnn_type

    def compile_model(self, loss="categorical_crossentropy", metrics=[], optimizer="adam"):
        self.model.compile(loss=loss, optimizer=optimizer, metrics=[loss] + list(metrics))

    def fit_model(self, x_train, y_train, validation_data=None, n_epochs=10, batch_size=100, verbose=1, callbacks=None):
        return self.model.fit(x_train, y_train, validation_data=validation_data, epochs=n_epochs,
                              batch_size=batch_size, verbose=verbose, callbacks=callbacks)

    def fit_generator(self, train_gen, steps_per_epoch, validation_steps, epochs, verbose, callbacks, validation_data):
        return self.model.fit_generator(train_gen, validation_data=validation_data, callbacks=callbacks,
                                        steps_per_epoch=steps_per_epoch, validation_steps=validation_steps,
                                        epochs=epochs, verbose=verbose)

    def predict(self, x_test, batch_size=10, verbose=1):
        return self.model.predict(x_test, batch_size=batch_size, verbose=verbose)

    def evaluate(self, x_test, y_test, batch_size=10, verbose=0):
        scores = self.model.evaluate(x_test, y_test, verbose=verbose, batch_size=batch_size)
        return self.model.metrics_names, scores

    def save_model_weights(self, directory):
        if not os.path.exists(directory):
            os.makedirs(directory)
        self.model.save_weights(directory + self.model_name + ".h5")

    def load_model_weights(self, filepath):
        self.model.load_weights(filepath, by_name=False)

    def get_layer_weights(self, layer):
        if isinstance(layer, str):
            return self.model.get_layer(layer).get_weights()
        return self.model.layers[layer].get_weights()

    def set_layer_weights_trainable(self, name, trainable=True):
        self.model.get_layer(name).trainable = trainable

    def set_layer_weights(self, name, weights):
        self.model.get_layer(name).set_weights(weights)

    def get_model_weights(self):
        return self.model.trainable_weights, self.model.non_trainable_weights

    def get_activations(self, layer, inputs, input_layers):
        """Outputs of a named layer in test phase: the recurrent layer -> (N,T,z_dim) states,
        the output layer -> (N,T,n_classes) probabilities."""
        x = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        if layer == self.model.layers[0].name:
            return self.model.hidden(x)
        if layer == self.model.layers[1].name:
            return self.model.predict(x)
        raise ValueError("No such layer: %s" % layer)


class RNNBaseline(BaseRNNModel):
    def __init__(self, timesteps, features, n_classes, model_name="baseline_model", rnn_type="simpleRNN",
                 out_activation="softmax", z_activation="relu", z_dim=20, z_to_y_drop=0.0):
        BaseRNNModel.__init__(self, n_classes, model_name=model_name, rnn_type=rnn_type)
        if out_activation != "softmax":
            raise NotImplementedError("out_activation %r" % (out_activation,))
        rnn_name = {"simpleRNN": "rnn", "LSTM": "lstm", "GRU": "gru"}.get(rnn_type, "rnn")
        self.model = SeqModel(timesteps, features, n_classes, z_dim, rnn_type, z_activation, rnn_name, "output",
                              use_bias=True, out_bias=True, drop_out=z_to_y_drop)


class NoRecurrenceModel(BaseRNNModel):
    """model.py:264-319: softmax(B x_t + A y_{t-1} (+ c)) -- the same logit terms as RNNFullModel's
    y_to_y / x_to_y branches with no recurrent state.  Runs on the same engine with the cell's
    weights frozen at zero (h == 0, so the z -> y Dense contributes nothing and receives no
    gradient).  ``embed_y`` (z = W y_{t-1} + c through Dense(z_dim), then the y -> y Dense on z,
    model.py:276-288) is a linear-activation cell whose recurrent kernel is frozen at zero."""

    def __init__(self, timesteps, x_dim, y_dim, model_name="y_to_y_model", y_to_y_activation="linear",
                 x_to_y_activation="linear", y_to_y_w_initializer=None, out_activation="softmax", mask_value=0.0,
                 y_bias=False, xy_bias=False, y_to_y_regularizer=None, z_dim=10, z_bias=True, connect_x=True,
                 connect_y=True, embed_y=False, diag_b=True):
        BaseRNNModel.__init__(self, y_dim, model_name=model_name, rnn_type=None)
        if not (connect_x or connect_y):
            raise ValueError("ERROR: the model needs an input! either x or y should be added.")
        if mask_value != 0.0:
            raise NotImplementedError("non-zero mask values")
        if out_activation != "softmax" or y_to_y_activation != "linear" or x_to_y_activation != "linear":
            raise NotImplementedError("output activations other than linear->softmax")
        if y_to_y_w_initializer is None:
            y_to_y_w_initializer = "random_uniform"
        if embed_y and connect_y:
            # y_to_z_output = Dense(z_dim)(y): the cell's input kernel (+ bias); y_output = Dense(y_dim)(z): the
            # output Dense, which carries the y_to_y initializer / regularizer / bias of the reference
            if y_bias and xy_bias and connect_x:
                raise NotImplementedError("embed_y with BOTH y_bias and xy_bias (two biases on the same logits)")
            m = SeqModel(timesteps, y_dim, y_dim, z_dim, "simpleRNN", "linear", "y_to_z_output", "y_output",
                         use_bias=z_bias, out_bias=y_bias or (xy_bias and connect_x), y_dim=y_dim, x_dim=x_dim, y_to_z=True,
                         x_to_z=False, y_to_y=False, x_to_y=connect_x, diag_b=diag_b, toy_regularizer=y_to_y_regularizer,
                         frozen_keys=("U",), toy_reg_on_x=False)       # the x_to_y Dense is a separate, unpenalised layer
            m.w["U"] = np.zeros_like(m.w["U"])
            m.w["Wout"] = initialize(y_to_y_w_initializer, (z_dim, y_dim))
            # the reference's three Dense layers, each with its own weight list
            m.layers = [Layer(m, "y_to_z_output", ["Wk"] + (["b"] if z_bias else [])),
                        Layer(m, "y_output", ["Wout"] + (["bout"] if y_bias else []))]
            if connect_x:
                k = initialize("glorot_uniform", (x_dim, y_dim))
                m.w["Wxy"] = k * np.eye(x_dim, y_dim, dtype=np.float32) if diag_b else k
                m.layers.append(Layer(m, "x_to_y_output", ["Wxy"] + (["bout"] if (xy_bias and not y_bias) else [])))
            self.model = m
            return
        m = SeqModel(timesteps, y_dim if connect_y else x_dim, y_dim, 8, "simpleRNN", "relu", "unused_rnn", "x_to_y_output",
                     use_bias=False, out_bias=xy_bias, y_dim=y_dim, x_dim=x_dim, y_to_z=connect_y, x_to_z=not connect_y,
                     y_to_y=connect_y, x_to_y=connect_x, diag_b=diag_b, ytoy_bias=y_bias,
                     y_to_y_w_initializer=y_to_y_w_initializer, y_to_y_regularizer=y_to_y_regularizer,
                     frozen_keys=("Wout",))
        # no recurrent path: zero, frozen cell and zero z -> y kernel
        for k in ("Wk", "U", "Wout"):
            m.w[k] = np.zeros_like(m.w[k])
        m.layers[0].trainable = False
        if connect_y:
            m.get_layer("y_to_y_output").name = "y_output"
        self.model = m


class RNNFullModel(BaseRNNModel):
    def __init__(self, timesteps, x_dim, y_dim, z_dim=20, model_name="y_to_y_model", rnn_type="simpleRNN",
                 z_to_z_activation="relu", y_to_y_activation="linear", xz_to_y_activation="linear",
                 y_to_y_w_initializer=None, out_activation="softmax", ytoy_bias=False, toy_bias=False, z_bias=True,
                 y_to_y_regularizer=None, toy_regularizer=None, y_to_z=True, y_to_z_initializer="glorot_normal",
                 y_to_y=True, x_to_y=True, x_to_z=False, z_to_y_dropout=0.0, diag_b=True, y_to_z_dropout=0.0,
                 z_to_z_dropout=0.0):
        BaseRNNModel.__init__(self, y_dim, model_name=model_name, rnn_type=rnn_type)
        if not (y_to_z or x_to_z):
            raise ValueError("ERROR: the model needs an input into z's! either x or y should be added.")
        if out_activation != "softmax" or xz_to_y_activation != "linear" or y_to_y_activation != "linear":
            raise NotImplementedError("output activations other than linear->softmax")
        if y_to_y_w_initializer is None:
            y_to_y_w_initializer = "glorot_uniform"
        # the reference passes kernel_initializer only to the LSTM (model.py:345-352)
        kinit = y_to_z_initializer if rnn_type == "LSTM" else "glorot_uniform"
        in_dim = (y_dim if y_to_z else 0) + (x_dim if x_to_z else 0)
        self.model = SeqModel(timesteps, in_dim, y_dim, z_dim, rnn_type, z_to_z_activation, "z_to_z_output", "to_y_output",
                              use_bias=z_bias, out_bias=toy_bias, drop_in=max(y_to_z_dropout, 0.0),
                              drop_rec=z_to_z_dropout, drop_out=max(z_to_y_dropout, 0.0), kernel_initializer=kinit,
                              y_dim=y_dim, x_dim=x_dim, y_to_z=y_to_z, x_to_z=x_to_z, y_to_y=y_to_y, x_to_y=x_to_y,
                              diag_b=diag_b, ytoy_bias=ytoy_bias, y_to_y_w_initializer=y_to_y_w_initializer,
                              y_to_y_regularizer=y_to_y_regularizer, toy_regularizer=toy_regularizer)


class ValLossHistoryCut(Callback):
    """model.py:94-117: per epoch, predict on the validation set, take the probability of the true
    class per step, and score the last 30 % of every sequence (utils.compute_likelihood_cut)."""

    def __init__(self, val_data, orig_seqs_lengths):
        Callback.__init__(self)
        self.val_data = val_data
        self.orig_seqs_lengths = orig_seqs_lengths

    def on_train_begin(self, logs={}):
        self.val_lossses = []
        if "my_loss" not in logs:
            logs["my_loss"] = 0.0

    def on_epoch_end(self, epoch, logs={}):
        y_pred = self.model.predict(self.val_data[0])
        pt = np.max(np.multiply(y_pred, self.val_data[1]), axis=2)
        pt = np.clip(pt, 1e-7, 1.0 - 1e-7)
        _, val_neg_ll = utils.compute_likelihood_cut(pt, 0.7, orig_lengths=self.orig_seqs_lengths)
        self.val_lossses.append(val_neg_ll)
        logs["my_loss"] = val_neg_ll
