"""Catalogue-scale form of the reference's model surface (SURVEY.md 8a1 "build form for configs 2-5").

``model.RNNFullModel`` / ``RNNBaseline`` keep the reference's tensors -- one-hot ``(N, T, V)`` inputs and
targets, a dense softmax -- which exist only at the reference's vocabulary sizes (|items| ~ 17..10^3).
``SampledRNNModel`` is the same model for |items| = 10^5..10^7: sessions stay lists of item ids, the
input kernel is factorised into an embedding table E[V, D] and W[D, G*H], the loss is the sampled softmax
over K shared log-uniform negatives (optional log-Q correction, optional tied input/output table), and
the method names mirror ``BaseRNNModel`` (model.py:170-238): compile_model / fit_model / evaluate / predict.

    m = SampledRNNModel(n_items=1_000_000, z_dim=256, rnn_type="GRU", n_negatives=2000)
    m.compile_model(optimizer=Adagrad(lr=0.01, epsilon=1e-8, clipnorm=1.))
    hist = m.fit_model(train_sessions, validation_data=val_sessions, n_epochs=3, batch_size=512)
    recall = m.recall_at_k(test_sessions, k=20)
    items, scores = m.predict(test_sessions, k=20)            # top-k next items after each session's last step

Under ``torch.distributed`` (one process per GPU, backend nccl = RCCL) pass ``dist=torch.distributed``:
the item tables are row-sharded (distributed.ShardedEngine) and every rank trains on its own sessions
(every rank must then make the same number of fit / evaluate / recall / predict batches: they are collective).
"""
import numpy as np

from . import batching, sampling
from .keras_compat import Adagrad, History


class SampledRNNModel:
    metrics_names = ["loss"]

    def __init__(self, n_items, z_dim=256, embed_dim=None, model_name="sampled_rnn", rnn_type="GRU", z_activation="relu",
                 n_negatives=2000, tied=False, logq=True, z_bias=True, item_counts=None, seed=0, device="cuda:0", dist=None):
        from . import engine as E
        cell = {"simpleRNN": "simplernn", "LSTM": "lstm", "GRU": "gru"}.get(rnn_type)
        if cell is None:
            raise ValueError("rnn_type must be 'simpleRNN', 'LSTM' or 'GRU' (got %r)" % (rnn_type,))
        self.n_classes, self.model_name, self.rnn_type = n_items, model_name, rnn_type
        D = z_dim if (embed_dim is None or tied) else embed_dim
        cfg = E.NetConfig(cell=cell, act=z_activation, H=z_dim, V_in=n_items, V_out=n_items, input="embed", D=D, output="sampled",
                          K=n_negatives, tied=tied, use_bias=z_bias, logq=logq, seed=seed)
        self.dist = dist
        if dist is not None:
            from .distributed import ShardedEngine
            self.engine = ShardedEngine(cfg, device, dist)
            self.rank, self.world = dist.get_rank(), dist.get_world_size()
        else:
            self.engine = E.Engine(cfg, device)
            self.rank, self.world = 0, 1
        self.cfg = cfg
        self._init_weights(seed)
        # negative-sampling proposal: log-uniform over the frequency ranking (item_counts) or over the ids
        rank = None
        if item_counts is not None:
            order = np.argsort(-np.asarray(item_counts), kind="stable")
            rank = np.empty(n_items, np.int64)
            rank[order] = np.arange(n_items)
        probs = sampling.log_uniform_probs(n_items, rank)
        if dist is not None:
            pl = probs[self.rank::self.world]
            pl = pl / pl.sum()
            th, al = sampling.build_alias_table(pl)
            self.engine.set_sampler(th, al, (np.log(pl) - np.log(self.world)).astype(np.float32))
        else:
            th, al = sampling.build_alias_table(probs)
            self.engine.set_sampler(th, al, np.log(probs).astype(np.float32))
        self.optimizer = None
        self.stop_training = False
        self._step = 0

    def _init_weights(self, seed):
        """E, Eout ~ U(-0.01, 0.01) (cf. tune_params.py:80), W glorot-uniform, U orthogonal per gate, b = 0
        (LSTM forget bias 1) -- the Keras defaults of the reference's layers."""
        import torch
        eng, c = self.engine, self.cfg
        g = torch.Generator(device=eng.dev)
        g.manual_seed(seed + 1 + self.rank)
        H, D, G = c.H, c.D, eng.G
        rs = np.random.default_rng(seed)
        with torch.no_grad():
            eng.P["E"].uniform_(-0.01, 0.01, generator=g)
            if "Eout" in eng.P:
                eng.P["Eout"].uniform_(-0.01, 0.01, generator=g)
        lim = float(np.sqrt(6.0 / (D + G * H)))
        eng.set_param("W", rs.uniform(-lim, lim, (D, G * H)).astype(np.float32))
        eng.set_param("U", np.concatenate([np.linalg.qr(rs.normal(size=(H, H)))[0] for _ in range(G)], axis=1).astype(np.float32))
        if "b" in eng.P:
            b = np.zeros(G * H, np.float32)
            if c.cell == "lstm":
                b[H:2 * H] = 1.0
            eng.set_param("b", b)

    # ---- BaseRNNModel surface (model.py:170-238) on sessions of item ids ---------------------------------
    def compile_model(self, loss="sampled_softmax", metrics=None, optimizer=None):
        if optimizer is None or isinstance(optimizer, str):
            optimizer = Adagrad()
        if not isinstance(optimizer, Adagrad):
            raise NotImplementedError("only Adagrad (experiments_methods.py:41) is implemented on the device")
        self.optimizer = optimizer

    @staticmethod
    def _flat(sessions):
        if isinstance(sessions, tuple) and len(sessions) == 2:          # already (flat ids, starts)
            return np.asarray(sessions[0]), np.asarray(sessions[1], dtype=np.int64)
        lens = np.fromiter((len(s) for s in sessions), dtype=np.int64, count=len(sessions))
        starts = np.zeros(len(sessions) + 1, dtype=np.int64)
        np.cumsum(lens, out=starts[1:])
        return np.fromiter((v for s in sessions for v in s), dtype=np.int32, count=int(starts[-1])), starts

    def _park(self, flat, starts):
        """Sessions parked in HBM once (single-GPU engine): batches are then built on the GPU."""
        return self.engine.put_dataset(flat, starts) if self.dist is None else None

    def _epoch(self, flat, starts, order, batch_size, train, ds=None):
        import torch
        eng, opt = self.engine, self.optimizer
        tot = torch.zeros(1, device=eng.dev)
        cnt = 0
        sels = [order[s:s + batch_size] for s in range(0, len(order), batch_size)]
        planner = None
        if ds is None and self.dist is not None:
            # row-sharded engine: a window of batches is routed at once, its count exchange begun half a window ahead (no host wait)
            from .distributed import WindowPlanner
            planner = WindowPlanner(eng, lambda j: batching.pack_flat(flat, starts, sels[j]) if j < len(sels) else None, 32)
        for i, sel in enumerate(sels):
            if ds is not None:
                d = eng.upload_device(ds, sel, defer=True)       # the gather rides in the training step's prologue launch
            elif planner is not None:
                d = planner.get(i)
            else:
                d = eng.upload(batching.pack_flat(flat, starts, sel))
            if d["n"] == 0:
                continue
            if train:
                l = eng.train_step(d, lr=opt.lr, eps=opt.epsilon, clipnorm=opt.clipnorm, step=self._step)
            else:
                l = eng.eval_loss(d, step=self._step)
            self._step += 1
            tot += l * float(len(sel))
            cnt += len(sel)
        out = float(tot.item()) / max(cnt, 1)
        eng.check_status()                 # the epoch's host sync: raise if a kernel reported a failure (Engine.check_status)
        return out

    def fit_model(self, x_train, y_train=None, validation_data=None, n_epochs=10, batch_size=512, verbose=1, callbacks=None,
                  shuffle=True):
        """x_train: list of sessions (item-id lists) or (flat ids, starts).  A session of n items yields the
        n-1 pairs x = s[i], y = s[i+1] (preprocessor.py:75-78); y_train is implied and ignored.  Epoch losses
        are batch-size-weighted means of the batches' token-mean losses, like Keras' History."""
        if self.optimizer is None:
            raise RuntimeError("You must compile a model before training/testing.")
        flat, starts = self._flat(x_train)
        val = None if validation_data is None else self._flat(validation_data)
        hist = History()
        cbs = [hist] + list(callbacks or [])
        for cb in cbs:
            cb.set_model(self)
            cb.on_train_begin({})
        N = len(starts) - 1
        index = np.arange(N)
        self.stop_training = False
        ds_tr = self._park(flat, starts)
        ds_va = None if val is None else self._park(val[0], val[1])
        for epoch in range(n_epochs):
            if shuffle:
                np.random.shuffle(index)
            logs = {"loss": self._epoch(flat, starts, index, batch_size, True, ds_tr)}
            if val is not None:
                logs["val_loss"] = self._epoch(val[0], val[1], np.arange(len(val[1]) - 1), batch_size, False, ds_va)
            for cb in cbs:
                cb.on_epoch_end(epoch, logs)
            if verbose:
                print("Epoch %d/%d - " % (epoch + 1, n_epochs) + " - ".join("%s: %.4f" % kv for kv in logs.items()))
            if self.stop_training:
                break
        for cb in cbs:
            cb.on_train_end({})
        return hist

    def evaluate(self, x_test, y_test=None, batch_size=512, verbose=0):
        flat, starts = self._flat(x_test)
        return self.metrics_names, [self._epoch(flat, starts, np.arange(len(starts) - 1), batch_size, False, self._park(flat, starts))]

    def recall_at_k(self, sessions, k=20, batch_size=512):
        """Share of transitions whose true next item is among the k best-scored items (rank counting over
        the whole catalogue; no N x V matrix)."""
        flat, starts = self._flat(sessions)
        hits = n = 0
        for s in range(0, len(starts) - 1, batch_size):
            d = self.engine.upload(batching.pack_flat(flat, starts, np.arange(s, min(len(starts) - 1, s + batch_size))))
            if d["n"] == 0:
                continue
            rk = self.engine.rank_counts(d)
            hits += int((rk < k).sum().item())
            n += d["n"]
        return hits / max(n, 1)

    def predict(self, sessions, k=20, batch_size=512, verbose=0):
        """Top-k next items after each session's LAST item: -> (ids int32 [N, k], scores float32 [N, k]); a
        session contributes its items s[0..n-1] as inputs (one dummy target is appended internally).  The
        large-vocabulary form of BaseRNNModel.predict (model.py:186-190)."""
        out_i = np.full((len(sessions), k), -1, np.int32)
        out_v = np.full((len(sessions), k), -np.inf, np.float32)
        for s in range(0, len(sessions), batch_size):
            chunk = [list(x) + [0] for x in sessions[s:s + batch_size]]           # every item becomes an input step
            rb = batching.pack_sessions(chunk)
            if rb.n_tok == 0:
                continue
            d = self.engine.upload(rb)
            last = np.array([int(rb.step_off[l - 1] + b) for b, l in enumerate(rb.lengths)], dtype=np.int32)
            ids, val = self.engine.topk_rows(d, k=k, rows=last)
            out_i[s + rb.order] = ids.cpu().numpy()
            out_v[s + rb.order] = val.cpu().numpy()
        return out_i, out_v

    # ---- weights ---------------------------------------------------------------------------------------
    def get_model_weights(self):
        return {k: self.engine.get_param(k) for k in self.engine.P}

    def set_model_weights(self, weights):
        for k, v in weights.items():
            self.engine.set_param(k, v)

    def save_model_weights(self, filepath):
        np.savez(filepath, **self.get_model_weights())

    def load_model_weights(self, filepath):
        with np.load(filepath, allow_pickle=False) as z:
            self.set_model_weights({k: z[k] for k in z.files})
