"""Property / known-answer tests that hold the [K2] restatement (oracle.nn) together.
SURVEY.md section 8c items (1)-(8).  No reference outputs exist for these (parity unpinned)."""
import numpy as np
import pytest
import torch

from oracle import nn as onn
from helpers import make_sessions, pad_batch, init_params, dense_grad

CELLS = ["simplernn", "lstm", "gru"]


def cfg_of(cell, inp="onehot", out="full", act="tanh", **kw):
    c = dict(cell=cell, act=act, input=inp, output=out, use_bias=True, out_bias=False, tied=False)
    c.update(kw)
    return c


def loss_of(cfg, params, batch, **fw):
    return onn.OracleNet(cfg, params).forward(batch, **fw)["loss"]


# (1) zero weights => h == 0 => loss = ln V exactly
@pytest.mark.parametrize("cell", CELLS)
def test_zero_weights_loss_is_lnV(cell):
    rng = np.random.default_rng(0)
    V, H = 17, 8
    cfg = cfg_of(cell, act="relu", out_bias=True)
    p = {k: np.zeros_like(v) for k, v in init_params(rng, cfg, V, H).items()}
    batch = pad_batch(make_sessions(rng, 6, V))
    assert abs(loss_of(cfg, p, batch) - np.log(17)) < 1e-12
    assert abs(np.log(17) - 2.833213344) < 1e-9


# (2) pad-invariance + (3) row-permutation invariance, loss and grads
@pytest.mark.parametrize("cell", CELLS)
def test_pad_and_permutation_invariance(cell):
    rng = np.random.default_rng(1)
    V, H = 11, 6
    cfg = cfg_of(cell, act="relu")
    p = init_params(rng, cfg, V, H)
    sess = make_sessions(rng, 7, V)
    net = onn.OracleNet(cfg, p)
    l0 = net.forward(pad_batch(sess))["loss"]
    g0 = net.backward()
    l1 = net.forward(pad_batch(sess, T=20))["loss"]
    g1 = net.backward()
    perm = rng.permutation(len(sess))
    l2 = net.forward(pad_batch([sess[i] for i in perm]))["loss"]
    g2 = net.backward()
    assert abs(l0 - l1) < 1e-12 and abs(l0 - l2) < 1e-12
    for k in g0:
        a = dense_grad(g0[k], p[k].shape)
        np.testing.assert_allclose(dense_grad(g1[k], p[k].shape), a, atol=1e-12)
        np.testing.assert_allclose(dense_grad(g2[k], p[k].shape), a, atol=1e-12)


# Keras' literal dense formulation: every (b,t) computed, switch on the mask.
def keras_dense_forward(cfg, p, batch):
    ids, mask = batch["ids"], batch["mask"]
    B, T = mask.shape
    V = p["Wk"].shape[0]
    H = p["U"].shape[0]
    x = np.zeros((B, T, V))
    x[np.arange(B)[:, None], np.arange(T)[None, :], ids] = 1.0
    x *= mask[:, :, None]                      # pad rows are all-zero one-hots
    m = np.any(x != 0, axis=2)                 # Masking(0.0)
    xw = x @ p["Wk"] + p["b"]
    h = np.zeros((B, H)); c = np.zeros((B, H)); out_prev = np.zeros((B, H))
    outs = []
    for t in range(T):
        hn, cn, _ = onn.cell_fwd(cfg["cell"], cfg["act"], xw[:, t], h, c, p["U"], None)
        mt = m[:, t][:, None]
        out_prev = np.where(mt, hn, out_prev)
        h = np.where(mt, hn, h)
        c = np.where(mt, cn, c)
        outs.append(out_prev)
    return np.stack(outs, axis=1)


@pytest.mark.parametrize("cell", CELLS)
def test_active_rows_scan_equals_keras_dense_switch(cell):
    rng = np.random.default_rng(2)
    V, H = 9, 5
    cfg = cfg_of(cell, act="relu")
    p = init_params(rng, cfg, V, H)
    batch = pad_batch(make_sessions(rng, 5, V), T=12)
    net = onn.OracleNet(cfg, p)
    net.forward(batch)
    np.testing.assert_allclose(net.st["hs"], keras_dense_forward(cfg, p, batch), atol=1e-13)


# (4) hand-computed single steps, incl. hard_sigmoid saturation and relu kinks
def test_single_step_hand_values():
    H = 1
    one = np.ones((1, 1))
    # SimpleRNN relu: h = relu(xw + h_prev*U)
    h, _, _ = onn.cell_fwd("simplernn", "relu", one * -0.5, one * 2.0, None, one * 0.5, None)
    assert h[0, 0] == 0.5
    # LSTM: pre_i=3 (saturated ->1), pre_f=-3 (->0), pre_c=2 relu->2, pre_o=0 ->0.5
    xw = np.array([[3.0, -3.0, 2.0, 0.0]])
    h, c, _ = onn.cell_fwd("lstm", "relu", xw, np.zeros((1, 1)), np.full((1, 1), 7.0), np.zeros((1, 4)), None)
    assert c[0, 0] == 2.0 and h[0, 0] == 1.0
    # GRU reset-before-matmul: z=hs(0)=.5, r=hs(-2.5)=0 -> candidate ignores h_prev entirely
    xw = np.array([[0.0, -2.5, 1.0]])
    U = np.array([[0.0, 0.0, 100.0]])
    h, _, _ = onn.cell_fwd("gru", "tanh", xw, np.full((1, 1), 3.0), None, U, None)
    assert abs(h[0, 0] - (0.5 * 3.0 + 0.5 * np.tanh(1.0))) < 1e-15
    assert onn.hard_sigmoid(np.array([2.5, -2.5, 0.0, 1.0])).tolist() == [1.0, 0.0, 0.5, 0.7]


# (5) finite differences, fp64
def fd_check(cfg, p, batch, fw, names, rng, n_probe=6, h=1e-6, tol=2e-6):
    net = onn.OracleNet(cfg, p)
    net.forward(batch, **fw)
    g = net.backward()
    for name in names:
        gd = dense_grad(g[name], p[name].shape)
        for _ in range(n_probe):
            idx = tuple(int(rng.integers(0, s)) for s in p[name].shape)
            old = p[name][idx]
            p[name][idx] = old + h
            lp = loss_of(cfg, p, batch, **fw)
            p[name][idx] = old - h
            lm = loss_of(cfg, p, batch, **fw)
            p[name][idx] = old
            fd = (lp - lm) / (2 * h)
            assert abs(fd - gd[idx]) <= tol * max(1.0, abs(fd)), (name, idx, fd, gd[idx])


@pytest.mark.parametrize("cell", CELLS)
@pytest.mark.parametrize("inp,out", [("onehot", "full"), ("embed", "sampled"), ("embed", "full")])
def test_finite_differences(cell, inp, out):
    rng = np.random.default_rng(3)
    V, H, D = 13, 5, 4
    # tanh keeps probes away from relu kinks; hard_sigmoid kinks are hit with prob ~0
    cfg = cfg_of(cell, inp, out, act="tanh", out_bias=True)
    p = init_params(rng, cfg, V, H, D)
    batch = pad_batch(make_sessions(rng, 6, V))
    fw = {}
    if out == "sampled":
        fw = dict(negatives=rng.integers(0, V, size=7).astype(np.int32), logq=np.log(np.full(V, 1.0 / V)))
    fd_check(cfg, p, batch, fw, list(p.keys()), rng)


@pytest.mark.parametrize("act", ["sigmoid", "softplus", "softsign", "elu", "hard_sigmoid"])
@pytest.mark.parametrize("cell", CELLS)
def test_finite_differences_other_activations(cell, act):
    """Keras 2.0's other element-wise cell activations (the reference passes any name through `z_to_z_activation`,
    model.py:324,346,351): the oracle's derivatives against fp64 finite differences, every cell."""
    rng = np.random.default_rng(30)
    V, H, D = 11, 5, 4
    cfg = cfg_of(cell, "embed", "full", act=act)
    p = init_params(rng, cfg, V, H, D)
    fd_check(cfg, p, pad_batch(make_sessions(rng, 5, V)), {}, list(p.keys()), rng, n_probe=4)


def test_finite_differences_tied_and_dropout():
    rng = np.random.default_rng(4)
    V, H = 12, 5
    cfg = cfg_of("gru", "embed", "sampled", act="tanh", tied=True)
    p = init_params(rng, cfg, V, H, H)
    batch = pad_batch(make_sessions(rng, 5, V))
    B, T = batch["mask"].shape
    drop = dict(in_scale=(rng.random((B, T, H)) > 0.3) / 0.7,
                out_mask=(rng.random((B, T, H)) > 0.3) / 0.7,
                rec_masks=(rng.random((3, B, H)) > 0.3) / 0.7)
    fw = dict(negatives=rng.integers(0, V, size=6).astype(np.int32), drop=drop)
    fd_check(cfg, p, batch, fw, list(p.keys()), rng)
    for cell in ("lstm", "simplernn"):
        cfg = cfg_of(cell, "onehot", "full", act="tanh")
        p = init_params(rng, cfg, V, H)
        G = onn.N_GATES[cell]
        drop = dict(in_scale=(rng.random((B, T)) > 0.3) / 0.7, out_mask=(rng.random((B, T, H)) > 0.3) / 0.7,
                    rec_masks=(rng.random((G, B, H)) > 0.3) / 0.7)
        fd_check(cfg, p, batch, dict(drop=drop), list(p.keys()), rng)


# torch-autograd twin of the forward (hand-written cells, NOT nn.GRU/nn.LSTM)
def torch_loss(cfg, p, batch):
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p.items()}
    ids, tgt, mask = (torch.tensor(batch[k]) for k in ("ids", "tgt", "mask"))
    B, T = mask.shape
    H = p["U"].shape[0]
    hs_ = lambda x: torch.clamp(0.2 * x + 0.5, 0, 1)
    act = torch.relu if cfg["act"] == "relu" else torch.tanh
    xw = tp["Wk"][ids] + tp["b"]
    h = torch.zeros(B, H, dtype=torch.float64); c = torch.zeros(B, H, dtype=torch.float64)
    U = tp["U"]
    tot = 0.0
    for t in range(T):
        x = xw[:, t]
        if cfg["cell"] == "simplernn":
            hn, cn = act(x + h @ U), c
        elif cfg["cell"] == "lstm":
            pre = x + h @ U
            i, f, g, o = hs_(pre[:, :H]), hs_(pre[:, H:2*H]), act(pre[:, 2*H:3*H]), hs_(pre[:, 3*H:])
            cn = f * c + i * g
            hn = o * act(cn)
        else:
            z = hs_(x[:, :H] + h @ U[:, :H]); r = hs_(x[:, H:2*H] + h @ U[:, H:2*H])
            hh = act(x[:, 2*H:] + (r * h) @ U[:, 2*H:])
            hn, cn = z * h + (1 - z) * hh, c
        m = mask[:, t][:, None]
        h = torch.where(m, hn, h); c = torch.where(m, cn, c)
        logp = torch.log_softmax(h @ tp["Wout"], dim=1)
        tot = tot - (logp[torch.arange(B), tgt[:, t]] * mask[:, t]).sum()
    loss = tot / mask.sum()
    loss.backward()
    return loss.item(), {k: v.grad.numpy() for k, v in tp.items()}


@pytest.mark.parametrize("cell", CELLS)
@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_backward_matches_torch_autograd_twin(cell, act):
    rng = np.random.default_rng(5)
    V, H = 10, 7
    cfg = cfg_of(cell, act=act)
    p = init_params(rng, cfg, V, H)
    batch = pad_batch(make_sessions(rng, 8, V))
    net = onn.OracleNet(cfg, p)
    l = net.forward(batch)["loss"]
    g = net.backward()
    tl, tg = torch_loss(cfg, p, batch)
    assert abs(l - tl) < 1e-12
    for k in p:
        np.testing.assert_allclose(dense_grad(g[k], p[k].shape), tg[k], atol=1e-11)


def test_simplernn_tanh_matches_torch_nn_rnn():
    """An INDEPENDENT implementation for the one cell both libraries define identically: the oracle's
    SimpleRNN with tanh on pre-padded batches against torch.nn.RNN on packed (ragged) sequences --
    Masking + pre-padding == every session run from a zero state over its real steps (SURVEY 3.2
    items 1-2).  Loss (masked token mean of softmax CE) and all weight gradients, fp64."""
    rng = np.random.default_rng(15)
    V, H = 9, 6
    cfg = cfg_of("simplernn", act="tanh")
    p = init_params(rng, cfg, V, H)
    sessions = make_sessions(rng, 7, V, 2, 9)
    batch = pad_batch(sessions)
    net = onn.OracleNet(cfg, p)
    loss = net.forward(batch)["loss"]
    g = net.backward()
    rnn = torch.nn.RNN(V, H, nonlinearity="tanh", batch_first=True).double()
    with torch.no_grad():
        rnn.weight_ih_l0.copy_(torch.tensor(p["Wk"].T))
        rnn.weight_hh_l0.copy_(torch.tensor(p["U"].T))
        rnn.bias_ih_l0.copy_(torch.tensor(p["b"]))
        rnn.bias_hh_l0.zero_()
    Wout = torch.tensor(p["Wout"], dtype=torch.float64, requires_grad=True)
    lens = [len(s) - 1 for s in sessions]
    T = max(lens)
    x = torch.zeros(len(sessions), T, V, dtype=torch.float64)
    tgt = torch.zeros(len(sessions), T, dtype=torch.long)
    for b, sq in enumerate(sessions):                       # POST-padded for pack_padded_sequence
        for t in range(lens[b]):
            x[b, t, sq[t]] = 1.0
            tgt[b, t] = sq[t + 1]
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, torch.tensor(lens), batch_first=True, enforce_sorted=False)
    out, _ = rnn(packed)
    hs, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True, total_length=T)
    logp = torch.log_softmax(hs @ Wout, dim=2)
    m = torch.zeros(len(sessions), T, dtype=torch.float64)
    for b in range(len(sessions)):
        m[b, : lens[b]] = 1.0
    tl = -(logp.gather(2, tgt[:, :, None])[:, :, 0] * m).sum() / m.sum()
    tl.backward()
    assert abs(loss - tl.item()) < 1e-12
    np.testing.assert_allclose(dense_grad(g["Wk"], p["Wk"].shape), rnn.weight_ih_l0.grad.numpy().T, atol=1e-11)
    np.testing.assert_allclose(g["U"], rnn.weight_hh_l0.grad.numpy().T, atol=1e-11)
    np.testing.assert_allclose(g["b"], rnn.bias_ih_l0.grad.numpy(), atol=1e-11)
    np.testing.assert_allclose(g["Wout"], Wout.grad.numpy(), atol=1e-11)


# (6) clipnorm + Adagrad
def test_clip_and_adagrad_first_step():
    assert onn.clip_scale(0.25, 1.0) == 1.0
    assert onn.clip_scale(1.0, 1.0) == 1.0           # norm == c: g*c/norm == g
    assert abs(onn.clip_scale(4.0, 1.0) - 0.5) < 1e-15
    p = {"w": np.array([1.0, -2.0, 3.0])}
    a = {"w": np.zeros(3)}
    g = {"w": np.array([0.3, -0.4, 0.0])}            # norm .5 < 1: no clip
    onn.adagrad_step(p, a, g, lr=0.01, eps=1e-8, clipnorm=1.0)
    exp = np.array([1.0, -2.0, 3.0]) - 0.01 * g["w"] / (np.abs(g["w"]) + 1e-8)
    np.testing.assert_allclose(p["w"], exp, atol=1e-15)
    np.testing.assert_allclose(a["w"], g["w"] ** 2)
    # sparse rows == dense with zero rows
    rng = np.random.default_rng(6)
    W = rng.normal(size=(6, 3)); A = np.abs(rng.normal(size=(6, 3)))
    gd = np.zeros((6, 3)); gd[[1, 4]] = rng.normal(size=(2, 3)) * 3
    p1, a1 = {"W": W.copy()}, {"W": A.copy()}
    p2, a2 = {"W": W.copy()}, {"W": A.copy()}
    s1 = onn.adagrad_step(p1, a1, {"W": gd})
    s2 = onn.adagrad_step(p2, a2, {"W": (np.array([1, 4]), gd[[1, 4]])})
    assert s1 == s2 and s1 < 1.0
    np.testing.assert_array_equal(p1["W"], p2["W"]); np.testing.assert_array_equal(a1["W"], a2["W"])


# (7) one-hot dense GEMM == row gather, bit for bit in fp64
def test_onehot_gemm_equals_gather():
    rng = np.random.default_rng(7)
    V, GH = 23, 12
    Wk = rng.normal(size=(V, GH))
    ids = rng.integers(0, V, size=40)
    x = np.zeros((40, V)); x[np.arange(40), ids] = 1.0
    np.testing.assert_array_equal(x @ Wk, Wk[ids])


# (8) sampled softmax with negatives = every item (hits masked) == full softmax
@pytest.mark.parametrize("cell", ["gru", "lstm"])
def test_sampled_all_items_equals_full(cell):
    rng = np.random.default_rng(8)
    V, H, D = 14, 6, 5
    cf = cfg_of(cell, "embed", "full", act="relu", out_bias=True)
    cs = cfg_of(cell, "embed", "sampled", act="relu", out_bias=True)
    pf = init_params(rng, cf, V, H, D)
    ps = {k: v.copy() for k, v in pf.items() if k != "Wout"}
    ps["Eout"] = pf["Wout"].T.copy()
    batch = pad_batch(make_sessions(rng, 6, V))
    nf, ns = onn.OracleNet(cf, pf), onn.OracleNet(cs, ps)
    lf = nf.forward(batch)["loss"]
    ls = ns.forward(batch, negatives=np.arange(V, dtype=np.int32))["loss"]
    assert abs(lf - ls) < 1e-12
    gf, gs = nf.backward(), ns.backward()
    np.testing.assert_allclose(dense_grad(gs["Eout"], (V, H)), gf["Wout"].T, atol=1e-12)
    np.testing.assert_allclose(dense_grad(gs["bout"], (V,)), gf["bout"], atol=1e-12)
    for k in ("U", "W", "b"):
        np.testing.assert_allclose(gs[k], gf[k], atol=1e-12)
    np.testing.assert_allclose(dense_grad(gs["E"], (V, D)), dense_grad(gf["E"], (V, D)), atol=1e-12)


def test_ce_clip_saturates_and_kills_gradient():
    logits = np.array([[40.0, 0.0, 0.0], [0.0, 40.0, 0.0]], dtype=np.float32)
    ce, dlog, p = onn.full_softmax_ce(logits, np.array([1, 1]), 2)
    # token 0: p_t ~ 4e-18 < 1e-7 -> ce = -log(1e-7), gradient 0; token 1: p_t > 1-1e-7 -> clipped too
    assert abs(ce - (-np.log(np.float32(1e-7)) - np.log(np.float32(1 - 1e-7)))) < 1e-5
    assert np.all(dlog == 0)


def test_predict_dense_pad_positions_are_softmax_of_bias():
    rng = np.random.default_rng(9)
    V, H = 6, 4
    cfg = cfg_of("lstm", act="relu", out_bias=True)
    p = init_params(rng, cfg, V, H)
    batch = pad_batch([[1, 2, 3], [4, 5, 0, 1, 2]])
    pr = onn.OracleNet(cfg, p).predict_dense(batch)
    assert pr.shape == (2, 4, V)
    sb = np.exp(p["bout"] - p["bout"].max()); sb /= sb.sum()
    np.testing.assert_allclose(pr[0, 0], sb, atol=1e-14)       # leading pad: h = 0
    np.testing.assert_allclose(pr.sum(axis=2), 1.0, atol=1e-13)


def test_finite_differences_side_branches():
    """y_to_y (direct one-hot term), x_to_y (history features into the output Dense) and x_to_z
    (history features concatenated into the RNN input, i.e. dense input)."""
    rng = np.random.default_rng(21)
    V, H = 9, 5
    batch = pad_batch(make_sessions(rng, 6, V))
    B, T = batch["mask"].shape
    xs = rng.random((B, T, V)) * batch["mask"][:, :, None]
    batch["xs"] = xs
    cfg = cfg_of("lstm", "onehot", "full", act="tanh")
    p = init_params(rng, cfg, V, H)
    p["Wyy"] = rng.normal(0, 0.3, (V, V))
    p["byy"] = rng.normal(0, 0.3, (V,))
    p["Wxy"] = rng.normal(0, 0.3, (V, V))
    fd_check(cfg, p, batch, {}, list(p.keys()), rng)
    # x_to_z: the RNN input is [one-hot(y), xs] -> dense mode with F = 2V
    cfgd = cfg_of("gru", "dense", "full", act="tanh")
    pd = init_params(rng, cfgd, 2 * V, H)
    pd["Wout"] = rng.normal(0, 0.3, (H, V))
    oh = np.zeros((B, T, V)); oh[np.arange(B)[:, None], np.arange(T)[None, :], batch["ids"]] = 1.0
    bd = dict(batch, x=np.concatenate([oh * batch["mask"][:, :, None], xs], axis=2))
    fd_check(cfgd, pd, bd, {}, list(pd.keys()), rng)
