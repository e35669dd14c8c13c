"""numpy restatement of the reference's RNN hot path  --  TEST INFRASTRUCTURE ONLY.

Follows, layer by layer, the Keras 2.0.x / Theano graph that the reference
builds and trains (all [K2] = behaviour of the un-vendored dependency; see
``oracle/__init__.py`` -- PARITY UNPINNED for this file):

  graph        model.py:241-258 (RNNBaseline), model.py:322-403 (RNNFullModel,
               y_to_z-only "ytoz" wiring)
  Masking      model.py:246,335-336     step is real iff any(x[b,t,:] != 0)
  RNN scan     model.py:248-255,344-369 masked steps carry state and output
  cells        Keras 2.0 SimpleRNN / LSTM (gate order i,f,c,o; hard_sigmoid
               recurrent activation; activation = z_to_z_activation, "relu" in
               every shipped call: experiments_methods.py:190) and, as the
               build's extension, Keras 2.0 GRU (order z,r,h; reset gate applied
               BEFORE the recurrent matmul)
  Dropout      model.py:256,346,351,357,363,368,372 (masks are explicit inputs here)
  Dense+softmax model.py:257,381-384,397
  loss         Theano categorical_crossentropy (renormalise, clip to
               [1e-7, 1-1e-7]) under Keras' masked token mean (model.py:175-177)
  optimizer    Adagrad(lr, epsilon=1e-8, decay=0, clipnorm=1.)
               experiments_methods.py:41,70,162,218

Everything works on the PADDED (B,T) view the reference uses (pre-padding,
preprocessor.py:16-20), computing only the real rows of each step -- which is
exactly what the mask ``switch`` evaluates to.

Extensions (no reference counterpart; this file is their specification):
factorised input (E then W), sampled softmax with shared negatives, tied
input/output table, GRU.
"""
import numpy as np

EPS_CE = 1e-7


# ----------------------------------------------------------------------------
# activations
# ----------------------------------------------------------------------------
def hard_sigmoid(x):
    return np.clip(x * x.dtype.type(0.2) + x.dtype.type(0.5), 0, 1)


def d_hard_sigmoid(pre):
    # Theano clip gradient: 1 inside [lo, hi], 0 outside; 0.2x+0.5 in [0,1] <=> |x| <= 2.5.
    # (The boundary is a measure-zero set; product and oracle both use the open interval.)
    return ((pre > -2.5) & (pre < 2.5)).astype(pre.dtype) * pre.dtype.type(0.2)


def act_fwd(name, pre):
    if name == "relu":
        return np.maximum(pre, 0)
    if name == "tanh":
        return np.tanh(pre)
    if name == "linear":
        return pre
    if name == "sigmoid":
        return 1.0 / (1.0 + np.exp(-pre))
    if name == "hard_sigmoid":
        return hard_sigmoid(pre)
    if name == "softplus":                       # Keras / Theano T.nnet.softplus
        return np.where(pre > 20, pre, np.log1p(np.exp(np.minimum(pre, 20))))
    if name == "softsign":
        return pre / (1 + np.abs(pre))
    if name == "elu":                            # alpha = 1
        return np.where(pre > 0, pre, np.expm1(np.minimum(pre, 0)))
    raise ValueError(name)


def act_bwd(name, pre, y):
    """d act / d pre as a function of the pre-activation and the output."""
    if name == "relu":
        return (pre > 0).astype(pre.dtype)   # relu'(0) := 0 (product and oracle agree)
    if name == "tanh":
        return 1 - y * y
    if name == "linear":
        return np.ones_like(pre)
    if name == "sigmoid":
        return y * (1 - y)
    if name == "hard_sigmoid":
        return d_hard_sigmoid(pre)
    if name == "softplus":
        return 1.0 / (1.0 + np.exp(-pre))
    if name == "softsign":
        return 1.0 / np.square(1 + np.abs(pre))
    if name == "elu":
        return np.where(pre > 0, np.ones_like(pre), np.exp(np.minimum(pre, 0)))
    raise ValueError(name)


N_GATES = {"simplernn": 1, "gru": 3, "lstm": 4}


# ----------------------------------------------------------------------------
# one recurrent step on the active rows
# ----------------------------------------------------------------------------
def cell_fwd(cell, act, xw, h, c, U, rm):
    """xw:(n,G*H) input projection incl. bias, h,c:(n,H) previous state,
    U:(H,G*H), rm: None or (G,n,H) recurrent-dropout multipliers.
    Returns h_new, c_new, cache."""
    H = h.shape[1]
    if cell == "simplernn":
        hm = h if rm is None else h * rm[0]
        pre = xw + hm @ U
        y = act_fwd(act, pre)
        return y, c, (pre, y, hm)
    if cell == "lstm":
        if rm is None:
            hms = [h, h, h, h]
            pre = xw + h @ U
        else:
            hms = [h * rm[g] for g in range(4)]
            pre = xw + np.concatenate([hms[g] @ U[:, g * H:(g + 1) * H] for g in range(4)], axis=1)
        i = hard_sigmoid(pre[:, 0:H])
        f = hard_sigmoid(pre[:, H:2 * H])
        g = act_fwd(act, pre[:, 2 * H:3 * H])
        o = hard_sigmoid(pre[:, 3 * H:4 * H])
        c_new = f * c + i * g
        ac = act_fwd(act, c_new)
        h_new = o * ac
        return h_new, c_new, (pre, i, f, g, o, c, c_new, ac, hms)
    if cell == "gru":
        hz = h if rm is None else h * rm[0]
        hr = h if rm is None else h * rm[1]
        pre_z = xw[:, 0:H] + hz @ U[:, 0:H]
        pre_r = xw[:, H:2 * H] + hr @ U[:, H:2 * H]
        z = hard_sigmoid(pre_z)
        r = hard_sigmoid(pre_r)
        rh = r * h if rm is None else r * h * rm[2]
        pre_h = xw[:, 2 * H:3 * H] + rh @ U[:, 2 * H:3 * H]
        hh = act_fwd(act, pre_h)
        h_new = z * h + (1 - z) * hh
        return h_new, c, (pre_z, pre_r, pre_h, z, r, hh, h, hz, hr, rh)
    raise ValueError(cell)


def cell_bwd(cell, act, dh, dc, cache, U, rm):
    """Returns dh_prev, dc_prev, dxw, dU for one step on the active rows."""
    H = dh.shape[1]
    if cell == "simplernn":
        pre, y, hm = cache
        dpre = dh * act_bwd(act, pre, y)
        dU = hm.T @ dpre
        dhp = dpre @ U.T
        if rm is not None:
            dhp = dhp * rm[0]
        return dhp, dc, dpre, dU
    if cell == "lstm":
        pre, i, f, g, o, c_prev, c_new, ac, hms = cache
        do = dh * ac
        dct = dc + dh * o * act_bwd(act, c_new, ac)
        di = dct * g
        df = dct * c_prev
        dg = dct * i
        dc_prev = dct * f
        dpre = np.concatenate([
            di * d_hard_sigmoid(pre[:, 0:H]),
            df * d_hard_sigmoid(pre[:, H:2 * H]),
            dg * act_bwd(act, pre[:, 2 * H:3 * H], g),
            do * d_hard_sigmoid(pre[:, 3 * H:4 * H])], axis=1)
        if rm is None:
            dU = hms[0].T @ dpre
            dhp = dpre @ U.T
        else:
            dU = np.concatenate([hms[k].T @ dpre[:, k * H:(k + 1) * H] for k in range(4)], axis=1)
            dhp = sum((dpre[:, k * H:(k + 1) * H] @ U[:, k * H:(k + 1) * H].T) * rm[k] for k in range(4))
        return dhp, dc_prev, dpre, dU
    if cell == "gru":
        pre_z, pre_r, pre_h, z, r, hh, h, hz, hr, rh = cache
        dz = dh * (h - hh)
        dhh = dh * (1 - z)
        dhp = dh * z
        dpre_h = dhh * act_bwd(act, pre_h, hh)
        drh = dpre_h @ U[:, 2 * H:3 * H].T
        if rm is not None:
            drh = drh * rm[2]
        dr = drh * h
        dhp = dhp + drh * r
        dpre_z = dz * d_hard_sigmoid(pre_z)
        dpre_r = dr * d_hard_sigmoid(pre_r)
        bz = dpre_z @ U[:, 0:H].T
        br = dpre_r @ U[:, H:2 * H].T
        if rm is not None:
            bz = bz * rm[0]
            br = br * rm[1]
        dhp = dhp + bz + br
        dU = np.concatenate([hz.T @ dpre_z, hr.T @ dpre_r, rh.T @ dpre_h], axis=1)
        return dhp, dc, np.concatenate([dpre_z, dpre_r, dpre_h], axis=1), dU
    raise ValueError(cell)


def rnn_forward(cell, act, xw, mask, U, rec_masks=None):
    """Masked scan (Keras/Theano K.rnn): xw:(B,T,G*H), mask:(B,T) bool.
    Masked steps carry state and repeat the previous output; the pre-scan
    output and the initial states are zero."""
    B, T, GH = xw.shape
    H = U.shape[0]
    h = np.zeros((B, H), xw.dtype)
    c = np.zeros((B, H), xw.dtype)
    hs = np.zeros((B, T, H), xw.dtype)
    caches = []
    for t in range(T):
        idx = np.nonzero(mask[:, t])[0]
        if idx.size:
            rm = None if rec_masks is None else rec_masks[:, idx]
            hn, cn, cache = cell_fwd(cell, act, xw[idx, t], h[idx], c[idx], U, rm)
            h[idx] = hn
            c[idx] = cn
            caches.append((idx, cache))
        else:
            caches.append(None)
        hs[:, t] = h
    return hs, caches


def rnn_backward(cell, act, dhs, U, caches, rec_masks=None):
    B, T, H = dhs.shape
    GH = U.shape[1]
    dh = np.zeros((B, H), dhs.dtype)
    dc = np.zeros((B, H), dhs.dtype)
    dxw = np.zeros((B, T, GH), dhs.dtype)
    dU = np.zeros_like(U)
    for t in range(T - 1, -1, -1):
        dh = dh + dhs[:, t]
        if caches[t] is not None:
            idx, cache = caches[t]
            rm = None if rec_masks is None else rec_masks[:, idx]
            dhp, dcp, dxw_t, dU_t = cell_bwd(cell, act, dh[idx], dc[idx], cache, U, rm)
            dh[idx] = dhp
            dc[idx] = dcp
            dxw[idx, t] = dxw_t
            dU += dU_t
    return dxw, dU


# ----------------------------------------------------------------------------
# output layers + loss
# ----------------------------------------------------------------------------
def _ce_bounds(dtype):
    lo = dtype.type(EPS_CE)
    hi = dtype.type(1.0 - EPS_CE)
    return lo, hi


def full_softmax_ce(logits, tgt, denom):
    """Keras softmax + Theano categorical_crossentropy on n token rows.
    Returns (sum of ce, dlogits for loss = sum(ce)/denom, probs)."""
    dt = logits.dtype
    m = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - m)
    p = e / e.sum(axis=1, keepdims=True)
    q = p / p.sum(axis=1, keepdims=True)          # Theano-backend renormalisation
    n = logits.shape[0]
    pt = q[np.arange(n), tgt]
    lo, hi = _ce_bounds(dt)
    ce = -np.log(np.clip(pt, lo, hi))
    active = ((pt >= lo) & (pt <= hi)).astype(dt)
    dlog = p.copy()
    dlog[np.arange(n), tgt] -= 1
    dlog *= (active / dt.type(denom))[:, None]
    return ce.sum(dtype=np.float64), dlog, p


def sampled_softmax_ce(h, tgt, neg, Eout, bout, logq, denom):
    """Sampled softmax over {target} U negatives (shared per batch), accidental
    hits (negative == target) removed, optional log-Q correction.
    Returns (sum ce, dh, dlt (n,), dln (n,K), lt, ln)."""
    dt = h.dtype
    Et = Eout[tgt]
    En = Eout[neg]
    lt = np.einsum("ij,ij->i", h, Et)
    ln = h @ En.T
    if bout is not None:
        lt = lt + bout[tgt]
        ln = ln + bout[neg][None, :]
    if logq is not None:
        lt = lt - logq[tgt].astype(dt)
        ln = ln - logq[neg].astype(dt)[None, :]
    hit = neg[None, :] == tgt[:, None]
    ln = np.where(hit, -np.inf, ln).astype(dt)
    m = np.maximum(lt, ln.max(axis=1))
    et = np.exp(lt - m)
    en = np.exp(ln - m[:, None])
    s = et + en.sum(axis=1)
    pt = et / s
    pn = en / s[:, None]
    lo, hi = _ce_bounds(dt)
    ce = -np.log(np.clip(pt, lo, hi))
    active = ((pt >= lo) & (pt <= hi)).astype(dt) / dt.type(denom)
    dlt = (pt - 1) * active
    dln = pn * active[:, None]
    dh = dlt[:, None] * Et + dln @ En
    return ce.sum(dtype=np.float64), dh, dlt, dln, lt, ln


def merge_rows(rows, vals):
    """Sum duplicate row contributions: returns (sorted unique rows, summed vals)."""
    rows = np.asarray(rows, dtype=np.int64)
    u, inv = np.unique(rows, return_inverse=True)
    out = np.zeros((u.shape[0],) + vals.shape[1:], vals.dtype)
    np.add.at(out, inv, vals)
    return u, out


# ----------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------
class OracleNet:
    """cfg keys: cell ('simplernn'|'lstm'|'gru'), act, H, input ('onehot'|'embed'|'dense'),
    output ('full'|'sampled'), tied (bool), use_bias (cell bias), out_bias (bool).

    params (numpy arrays, all of one float dtype):
      input 'onehot'/'dense': Wk (V|F, G*H)      'embed': E (V,D), W (D,G*H)
      U (H,G*H); b (G*H,) if use_bias
      output 'full': Wout (H,V), bout (V,) if out_bias
      output 'sampled': Eout (V,H) (absent if tied -> E), bout (V,) if out_bias
    """

    def __init__(self, cfg, params):
        self.cfg = dict(cfg)
        self.p = params
        self.dtype = params["U"].dtype
        self.G = N_GATES[cfg["cell"]]

    # -- forward -----------------------------------------------------------
    def forward(self, batch, drop=None, negatives=None, logq=None):
        """batch: dict(ids (B,T) int | x (B,T,F) float, tgt (B,T) int, mask (B,T) bool).
        drop: None or dict(in_scale, out_mask, rec_masks) of explicit multipliers.
        Returns dict with loss (float64), n_tok, and what ``predict`` needs."""
        cfg, p, dt = self.cfg, self.p, self.dtype
        drop = drop or {}
        mask = batch["mask"]
        B, T = mask.shape
        st = {"batch": batch, "drop": drop, "negatives": negatives, "logq": logq}
        b = p["b"] if cfg.get("use_bias", True) else None
        if cfg["input"] == "onehot":
            ids = np.where(mask, batch["ids"], 0)
            xw = p["Wk"][ids]
            if drop.get("in_scale") is not None:
                xw = xw * drop["in_scale"][:, :, None]
        elif cfg["input"] == "embed":
            ids = np.where(mask, batch["ids"], 0)
            x = p["E"][ids]
            if drop.get("in_scale") is not None:
                x = x * drop["in_scale"]
            st["x"] = x
            xw = x @ p["W"]
        else:
            x = batch["x"].astype(dt)
            if drop.get("in_scale") is not None:
                x = x * drop["in_scale"]
            st["x"] = x
            xw = x @ p["Wk"]
        if b is not None:
            xw = xw + b
        xw = xw * mask[:, :, None].astype(dt)     # pad rows never enter the scan
        hs, caches = rnn_forward(cfg["cell"], cfg["act"], xw, mask, p["U"], drop.get("rec_masks"))
        st["hs"], st["caches"] = hs, caches
        hd = hs if drop.get("out_mask") is None else hs * drop["out_mask"]
        st["hd"] = hd
        bi, ti = np.nonzero(mask)
        st["bi"], st["ti"] = bi, ti
        n_tok = bi.shape[0]
        tgt = batch["tgt"][bi, ti].astype(np.int64) if "tgt" in batch else None
        hrows = hd[bi, ti]
        out = {"n_tok": n_tok, "hs": hs}
        if cfg["output"] == "full":
            logits = hrows @ p["Wout"]
            if cfg.get("out_bias", False):
                logits = logits + p["bout"]
            # RNNFullModel side branches (model.py:375-392): history features straight into the output
            # layer (x_to_y; same Dense as z, kernel rows [z_dim:]) and the direct y_{t-1} -> y_t term
            # (y_to_y: TimeDistributed(Dense) on the UNmasked one-hot input = a row of Wyy)
            if "Wxy" in p:
                logits = logits + batch["xs"][bi, ti].astype(dt) @ p["Wxy"]
            if "Wyy" in p:
                logits = logits + p["Wyy"][batch["ids"][bi, ti]]
                if "byy" in p:
                    logits = logits + p["byy"]
            if tgt is not None:
                ce, dlog, pr = full_softmax_ce(logits, tgt, max(n_tok, 1))
                st["dlog"] = dlog
                out["loss"] = ce / max(n_tok, 1)
            else:
                m = logits.max(axis=1, keepdims=True)
                e = np.exp(logits - m)
                pr = e / e.sum(axis=1, keepdims=True)
            out["probs_rows"] = pr
        else:
            Eout = p["E"] if cfg.get("tied", False) else p["Eout"]
            bout = p["bout"] if cfg.get("out_bias", False) else None
            ce, dh, dlt, dln, lt, ln = sampled_softmax_ce(hrows, tgt, negatives, Eout, bout, logq, max(n_tok, 1))
            st.update(dh_rows=dh, dlt=dlt, dln=dln, hrows=hrows, tgt=tgt)
            out["loss"] = ce / max(n_tok, 1)
            out["lt"], out["ln"] = lt, ln
        self.st = st
        return out

    def predict_dense(self, batch):
        """Keras ``Model.predict``: (B,T,V) probabilities, pad positions included
        (softmax of the carried/zero state)."""
        cfg, p = self.cfg, self.p
        self.forward({k: v for k, v in batch.items() if k != "tgt"})
        hs = self.st["hs"]
        logits = hs @ p["Wout"]
        if cfg.get("out_bias", False):
            logits = logits + p["bout"]
        if "Wxy" in p:
            logits = logits + batch["xs"].astype(self.dtype) @ p["Wxy"]
        if "Wyy" in p:                       # unmasked one-hot input: pad rows contribute only the bias
            logits = logits + p["Wyy"][batch["ids"]] * batch["mask"][:, :, None]
            if "byy" in p:
                logits = logits + p["byy"]
        m = logits.max(axis=2, keepdims=True)
        e = np.exp(logits - m)
        return e / e.sum(axis=2, keepdims=True)

    # -- backward ----------------------------------------------------------
    def backward(self):
        """Gradients of the masked-token-mean loss.  Dense tensors come back as
        arrays; table tensors ('E', 'Eout', 'Wk' in onehot mode, 'bout' in sampled
        mode) as (unique rows, summed row grads)."""
        cfg, p, st, dt = self.cfg, self.p, self.st, self.dtype
        batch, drop = st["batch"], st["drop"]
        mask = batch["mask"]
        B, T = mask.shape
        H = p["U"].shape[0]
        bi, ti = st["bi"], st["ti"]
        g = {}
        sparse = {}
        dhd = np.zeros((B, T, H), dt)
        if cfg["output"] == "full":
            dlog = st["dlog"]
            hrows = st["hd"][bi, ti]
            g["Wout"] = hrows.T @ dlog
            if cfg.get("out_bias", False):
                g["bout"] = dlog.sum(axis=0)
            if "Wxy" in p:
                g["Wxy"] = batch["xs"][bi, ti].astype(dt).T @ dlog
            if "Wyy" in p:
                sparse["Wyy"] = (batch["ids"][bi, ti].astype(np.int64), dlog)
                if "byy" in p:
                    g["byy"] = dlog.sum(axis=0)
            dhd[bi, ti] = dlog @ p["Wout"].T
        else:
            neg = st["negatives"]
            tgt = st["tgt"]
            dhd[bi, ti] = st["dh_rows"]
            rows = np.concatenate([tgt, neg.astype(np.int64)])
            vals = np.concatenate([st["dlt"][:, None] * st["hrows"], st["dln"].T @ st["hrows"]], axis=0)
            sparse["Eout"] = (rows, vals)
            if cfg.get("out_bias", False):
                sparse["bout"] = (rows, np.concatenate([st["dlt"], st["dln"].sum(axis=0)])[:, None])
        dhs = dhd if drop.get("out_mask") is None else dhd * drop["out_mask"]
        dxw, dU = rnn_backward(cfg["cell"], cfg["act"], dhs, p["U"], st["caches"], drop.get("rec_masks"))
        g["U"] = dU
        dxw_rows = dxw[bi, ti]
        if cfg.get("use_bias", True):
            g["b"] = dxw_rows.sum(axis=0)
        if cfg["input"] == "onehot":
            vals = dxw_rows
            if drop.get("in_scale") is not None:
                vals = vals * drop["in_scale"][bi, ti][:, None]
            sparse["Wk"] = (batch["ids"][bi, ti].astype(np.int64), vals)
        elif cfg["input"] == "embed":
            xr = st["x"][bi, ti]
            g["W"] = xr.T @ dxw_rows
            dx = dxw_rows @ p["W"].T
            if drop.get("in_scale") is not None:
                dx = dx * drop["in_scale"][bi, ti]
            sparse["E"] = (batch["ids"][bi, ti].astype(np.int64), dx)
        else:
            xr = st["x"][bi, ti]
            g["Wk"] = xr.T @ dxw_rows
        if cfg.get("tied", False) and "Eout" in sparse:
            r0, v0 = sparse.pop("Eout")
            r1, v1 = sparse["E"]
            sparse["E"] = (np.concatenate([r1, r0]), np.concatenate([v1, v0], axis=0))
        for k, (r, v) in sparse.items():
            g[k] = merge_rows(r, v)
        if "bout" in g and isinstance(g["bout"], tuple):
            g["bout"] = (g["bout"][0], g["bout"][1][:, 0])
        return g


# ----------------------------------------------------------------------------
# optimizer: global-norm clip + Adagrad (Keras 2.0 ``optimizers.Adagrad``)
# ----------------------------------------------------------------------------
def grad_sqnorm(grads):
    s = 0.0
    for v in grads.values():
        a = v[1] if isinstance(v, tuple) else v
        s += float(np.sum(np.square(a, dtype=np.float64)))
    return s


def clip_scale(sqnorm, clipnorm):
    """Keras clip_norm: g * c / norm if norm >= c else g."""
    if clipnorm is None or clipnorm <= 0:
        return 1.0
    n = np.sqrt(sqnorm)
    return float(clipnorm / n) if n >= clipnorm else 1.0


def prior_penalty(w, means, strength):
    """Kernel regularizer of the y_to_y / to_y Dense layers.  GaussPriorRegularizer (model.py:71-91):
    ``K.sum(1/(2 var) * K.square(x - means))`` -> strength = 1/(2 var); keras ``l2(l)``: means = 0,
    strength = l.  Keras adds the value to the loss it reports (train and validation) and to the
    objective it differentiates.  -> (penalty, d penalty / d w)"""
    d = np.asarray(w, dtype=np.float64) - (0.0 if means is None else np.asarray(means, dtype=np.float64))
    return float(strength * np.sum(d * d)), (2.0 * strength * d).astype(np.asarray(w).dtype)


def add_priors(params, grads, priors):
    """grads (from OracleNet.backward) += penalty gradients; row-sparse entries become dense.
    priors: {name: (means or None, strength)}.  -> total penalty"""
    total = 0.0
    for k, (means, strength) in priors.items():
        pen, g = prior_penalty(params[k], means, strength)
        total += pen
        cur = grads.get(k)
        if isinstance(cur, tuple):
            dense = np.zeros(params[k].shape, dtype=params[k].dtype)
            np.add.at(dense, cur[0], cur[1])
            cur = dense
        grads[k] = g if cur is None else cur + g
    return total


def adagrad_step(params, accum, grads, lr=0.01, eps=1e-8, clipnorm=1.0, frozen=()):
    """In-place update.  ``accum`` holds one zero-initialised accumulator per param.
    Row-sparse grads touch only their rows -- identical to the dense rule, since a
    zero gradient leaves both accumulator and parameter unchanged."""
    scale = clip_scale(grad_sqnorm({k: v for k, v in grads.items() if k not in frozen}), clipnorm)
    for k, v in grads.items():
        if k in frozen:
            continue
        p, a = params[k], accum[k]
        dt = p.dtype
        if isinstance(v, tuple):
            rows, gv = v
            gv = (gv * dt.type(scale)).astype(dt)
            a[rows] = a[rows] + gv * gv
            p[rows] = p[rows] - dt.type(lr) * gv / (np.sqrt(a[rows]) + dt.type(eps))
        else:
            gv = (v * dt.type(scale)).astype(dt)
            a += gv * gv
            p -= dt.type(lr) * gv / (np.sqrt(a) + dt.type(eps))
    return scale
