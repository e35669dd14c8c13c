"""Op-level parity of the HIP kernels (through the C ABI) against numpy / the oracle.
All tests need a real MI355X."""
import importlib

import numpy as np
import pytest
import torch

from oracle import nn as onn
from oracle import rng as orng

pytestmark = pytest.mark.gpu

L = importlib.import_module("seq-recommendations_amd._lib")
B_ = importlib.import_module("seq-recommendations_amd.batching")
ptr, call = L.ptr, L.call


_KEEP = []


def dev(a):
    """numpy -> device tensor, kept alive until the end of the test: ptr() only captures the
    address, and a temporary freed mid-argument-list would be recycled by the caching allocator."""
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _KEEP.append(t)
    return t


@pytest.fixture(autouse=True)
def _keepalive():
    yield
    torch.cuda.synchronize()
    _KEEP.clear()


def st():
    return torch.cuda.current_stream().cuda_stream


def gemm(a_kc, b_kc, M, N, K, A, lda, Bm, ldb, Cm, ldc, bias=None, acc=0, splitk=1):
    ws = torch.empty(max(1, splitk * M * N), device="cuda") if splitk > 1 else None
    call("seqrec_gemm_f32", a_kc, b_kc, M, N, K, ptr(A), lda, ptr(Bm), ldb, ptr(Cm), ldc, ptr(bias), acc, splitk, ptr(ws), st())


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (64, 64, 16), (100, 17, 64), (300, 200, 70), (513, 257, 129),
                                   (2603, 768, 256), (1500, 2000, 256)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 0), (1, 1), (0, 0), (0, 1)])
def test_gemm_layouts(M, N, K, a_kc, b_kc):
    rng = np.random.default_rng(M * 7 + N * 3 + K + a_kc * 2 + b_kc)
    A = rng.normal(size=(M, K)).astype(np.float32)
    Bm = rng.normal(size=(K, N)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    ref = A.astype(np.float64) @ Bm.astype(np.float64) + bias
    Ad = dev(A if a_kc else A.T)
    Bd = dev(Bm.T if b_kc else Bm)
    C = torch.full((M, N), float("nan"), device="cuda")
    gemm(a_kc, b_kc, M, N, K, Ad, K if a_kc else M, Bd, K if b_kc else N, C, N, bias=dev(bias))
    err = np.abs(C.cpu().numpy() - ref).max()
    assert err <= 2e-5 * max(1.0, np.sqrt(K)), err


@pytest.mark.parametrize("tile", [1, 2, 3])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 0), (1, 1), (0, 0), (0, 1)])
def test_gemm_lds_dma_kernels_every_tile_every_layout(tile, a_kc, b_kc):
    """The LDS-DMA GEMM (gemm.hip v2): every workgroup tile x every operand layout on shapes with ragged M / N edges,
    a K tail (K % 16 != 0), split-K, bias and accumulate, against float64 numpy; integer operands make it bit-exact."""
    lib = L.load()
    try:
        lib.seqrec_debug_gemm_tile(tile, 0)
        for (M, N, K, splitk) in [(64, 64, 16, 1), (130, 68, 100, 1), (300, 200, 72, 1), (516, 260, 132, 3), (2604, 768, 256, 1),
                                  (1500, 2000, 256, 1), (384, 256, 2604, 5)]:
            rng = np.random.default_rng(M + N + K + tile)
            A = rng.integers(-4, 5, size=(M, K)).astype(np.float32)
            Bm = rng.integers(-4, 5, size=(K, N)).astype(np.float32)
            bias = rng.integers(-3, 4, size=N).astype(np.float32)
            C0 = rng.integers(-9, 10, size=(M, N)).astype(np.float32)
            C = dev(C0)
            gemm(a_kc, b_kc, M, N, K, dev(A if a_kc else A.T.copy()), K if a_kc else M, dev(Bm.T.copy() if b_kc else Bm), K if b_kc else N,
                 C, N, bias=dev(bias), acc=1, splitk=splitk)
            ref = (A.astype(np.float64) @ Bm.astype(np.float64) + bias + C0).astype(np.float32)
            np.testing.assert_array_equal(C.cpu().numpy(), ref, err_msg=str((M, N, K, splitk)))
    finally:
        lib.seqrec_debug_gemm_tile(0, 0)


def test_gemm_is_exact_fp32_fma_chain_on_integers():
    # small integers: every product and partial sum is exact -> bit-exact result; asymmetric B
    # catches a transposed C write, A = I catches a wrong operand map.
    rng = np.random.default_rng(0)
    M, N, K = 96, 80, 48
    A = rng.integers(-8, 9, size=(M, K)).astype(np.float32)
    Bm = (np.arange(K)[:, None] * 3 + np.arange(N)[None, :] * 7 % 11 - 5).astype(np.float32)
    for a_kc, b_kc in [(1, 0), (0, 1), (1, 1), (0, 0)]:
        C = torch.zeros((M, N), device="cuda")
        gemm(a_kc, b_kc, M, N, K, dev(A if a_kc else A.T), K if a_kc else M, dev(Bm.T if b_kc else Bm), K if b_kc else N, C, N)
        np.testing.assert_array_equal(C.cpu().numpy(), A @ Bm)
    I = np.eye(K, dtype=np.float32)
    C = torch.zeros((K, N), device="cuda")
    gemm(1, 0, K, N, K, dev(I), K, dev(Bm), N, C, N)
    np.testing.assert_array_equal(C.cpu().numpy(), Bm)


@pytest.mark.parametrize("splitk", [2, 7, 32])
def test_gemm_splitk_accumulate_and_strided_views(splitk):
    rng = np.random.default_rng(splitk)
    M, N, K = 256, 512, 2603
    A = rng.normal(size=(K, M)).astype(np.float32)          # stored K x M (a_kcontig = 0)
    big = rng.normal(size=(K, 768)).astype(np.float32)
    Bv = big[:, 256:]                                        # view: ldb = 768, N = 512
    C0 = rng.normal(size=(M, 768)).astype(np.float32)
    Cd = dev(C0)
    bigd = dev(big)
    gemm(0, 0, M, N, K, dev(A), M, bigd[:, 256:], 768, Cd[:, 256:], 768, acc=1, splitk=splitk)
    ref = C0.copy().astype(np.float64)
    ref[:, 256:] += A.T.astype(np.float64) @ Bv.astype(np.float64)
    out = Cd.cpu().numpy()
    np.testing.assert_array_equal(out[:, :256], C0[:, :256])
    assert np.abs(out - ref).max() < 2e-3


def test_gather_rows_bit_exact_and_variants():
    rng = np.random.default_rng(1)
    V, W, n = 5000, 256, 3001
    tab = rng.normal(size=(V, W)).astype(np.float32)
    ids = rng.integers(0, V, size=n).astype(np.int32)
    ids[::97] = -1
    out = torch.empty((n, W), device="cuda")
    call("seqrec_gather_rows", ptr(dev(tab)), ptr(dev(ids)), ptr(out), n, W, None, None, 0, st())
    ref = np.where(ids[:, None] >= 0, tab[np.maximum(ids, 0)], 0).astype(np.float32)
    np.testing.assert_array_equal(out.cpu().numpy(), ref)
    sc = rng.normal(size=n).astype(np.float32)
    bias = rng.normal(size=W).astype(np.float32)
    call("seqrec_gather_rows", ptr(dev(tab)), ptr(dev(ids)), ptr(out), n, W, ptr(dev(sc)), ptr(dev(bias)), 1, st())
    ref2 = ref + (ref * sc[:, None] + bias)
    np.testing.assert_allclose(out.cpu().numpy(), ref2, rtol=1e-6, atol=1e-6)
    # odd width (scalar path) and empty input
    tab3 = rng.normal(size=(40, 17)).astype(np.float32)
    ids3 = rng.integers(0, 40, size=9).astype(np.int32)
    out3 = torch.empty((9, 17), device="cuda")
    call("seqrec_gather_rows", ptr(dev(tab3)), ptr(dev(ids3)), ptr(out3), 9, 17, None, None, 0, st())
    np.testing.assert_array_equal(out3.cpu().numpy(), tab3[ids3])
    call("seqrec_gather_rows", ptr(dev(tab3)), ptr(dev(ids3)), ptr(out3), 0, 17, None, None, 0, st())


def test_bad_arguments_are_rejected_before_launch():
    lib = L.load()
    assert lib.seqrec_gather_rows(None, None, None, 5, 8, None, None, 0, None) == -1
    assert lib.seqrec_rnn_fwd(0, 0, 100, 100, 3, 4, None, None, None, None, None, None, None) == -2
    assert lib.seqrec_gemm_f32(1, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 1, None, None) == -1


def packed_scan_inputs(rng, cell, H, B, maxlen, V=50):
    sess = [rng.integers(0, V, size=int(rng.integers(2, maxlen + 2))).tolist() for _ in range(B)]
    rb = B_.pack_sessions(sess)
    G = onn.N_GATES[cell]
    XW = (rng.normal(size=(rb.n_tok, G * H)) * 0.7).astype(np.float32)
    U = (rng.normal(size=(H, G * H)) * (0.6 / np.sqrt(H))).astype(np.float32)
    return rb, XW, U


def oracle_scan(cell, act, rb, XW, U, dH=None):
    """Run oracle.nn's masked scan on the padded view of the packed batch."""
    B, T, n = rb.B, rb.T, rb.n_tok
    G = onn.N_GATES[cell]
    H = U.shape[0]
    step_off = rb.step_off
    tok_t = rb.tok_s.astype(np.int64)
    tok_bs = np.arange(n) - step_off[tok_t]
    # post-padded is fine for the oracle: masked steps only carry state
    xw = np.zeros((B, T, G * H), np.float64)
    mask = np.zeros((B, T), bool)
    xw[tok_bs, tok_t] = XW
    mask[tok_bs, tok_t] = True
    hs, caches = onn.rnn_forward(cell, act, xw, mask, U.astype(np.float64))
    out = {"H": hs[tok_bs, tok_t]}
    if dH is not None:
        dhs = np.zeros((B, T, H))
        dhs[tok_bs, tok_t] = dH
        dxw, dU = onn.rnn_backward(cell, act, dhs, U.astype(np.float64), caches)
        out["dPre"] = dxw[tok_bs, tok_t]
        out["dU"] = dU
    return out


@pytest.mark.parametrize("cell", ["simplernn", "lstm", "gru"])
@pytest.mark.parametrize("H,B,maxlen,act", [(64, 5, 6, "relu"), (64, 37, 12, "tanh"), (128, 100, 9, "relu"),
                                            (256, 70, 20, "relu"), (512, 33, 7, "tanh"), (256, 512, 49, "tanh"),
                                            (128, 40, 10, "linear")])
def test_rnn_scan_forward_backward_vs_oracle(cell, H, B, maxlen, act):
    rng = np.random.default_rng(H + B + maxlen)
    rb, XW, U = packed_scan_inputs(rng, cell, H, B, maxlen)
    n, G = rb.n_tok, onn.N_GATES[cell]
    dH = (rng.normal(size=(n, H)) * 0.5).astype(np.float32)
    ref = oracle_scan(cell, act, rb, XW, U, dH)
    so = dev(rb.step_off)
    Hout = torch.full((n, H), float("nan"), device="cuda")
    gates = torch.full((n, G * H), float("nan"), device="cuda")
    aux = torch.full((n, H), float("nan"), device="cuda")
    up = torch.empty(int(L.load().seqrec_rnn_upack_floats(L.CELL[cell], H)), device="cuda")
    XWd, Ud = dev(XW), dev(U)
    call("seqrec_rnn_pack_u", L.CELL[cell], H, ptr(Ud), ptr(up), st())
    call("seqrec_rnn_fwd", L.CELL[cell], L.ACT[act], H, H, rb.T, rb.B, ptr(so), ptr(XWd), ptr(Hout), ptr(gates),
         ptr(aux), ptr(up), st())
    got = Hout.cpu().numpy()
    scale = max(1.0, np.abs(ref["H"]).max())
    assert np.abs(got - ref["H"]).max() <= 3e-5 * scale, np.abs(got - ref["H"]).max()
    dPre = torch.full((n, G * H), float("nan"), device="cuda")
    call("seqrec_rnn_bwd", L.CELL[cell], L.ACT[act], H, H, rb.T, rb.B, ptr(so), ptr(dev(dH)), ptr(Hout), ptr(gates), ptr(aux),
         ptr(dPre), ptr(up), st())
    gp = dPre.cpu().numpy()
    s2 = max(1.0, np.abs(ref["dPre"]).max())
    # relu / hard_sigmoid kinks: a pre-activation within rounding of a kink may pick the other
    # branch in fp32 -- allow a vanishing fraction of outliers, everything else tight.
    bad = np.abs(gp - ref["dPre"]) > 1e-4 * s2
    assert bad.mean() < 2e-4, (bad.mean(), np.abs(gp - ref["dPre"]).max())


def test_rnn_scan_h_real_padding_keeps_padded_units_zero():
    rng = np.random.default_rng(5)
    H, Hr = 128, 100
    rb, XW, U = packed_scan_inputs(rng, "lstm", H, 20, 8)
    n = rb.n_tok
    Hout = torch.empty((n, H), device="cuda"); gates = torch.empty((n, 4 * H), device="cuda"); aux = torch.empty((n, H), device="cuda")
    up = torch.empty(8 * H * H, device="cuda")
    call("seqrec_rnn_pack_u", 1, H, ptr(dev(U)), ptr(up), st())
    call("seqrec_rnn_fwd", 1, 1, H, Hr, rb.T, rb.B, ptr(dev(rb.step_off)), ptr(dev(XW)), ptr(Hout), ptr(gates), ptr(aux), ptr(up), st())
    h = Hout.cpu().numpy()
    assert np.all(h[:, Hr:] == 0) and np.abs(h[:, :Hr]).max() > 0


@pytest.mark.parametrize("n,V", [(1, 3), (77, 17), (300, 1000), (64, 4099)])
def test_full_softmax_ce_vs_oracle(n, V):
    rng = np.random.default_rng(n + V)
    logits = (rng.normal(size=(n, V)) * 3).astype(np.float32)
    logits[0, :] = 0
    if n > 2:
        logits[1, 0] = 60.0      # forces the clip branch when the target is another class
    tgt = rng.integers(0, V, size=n).astype(np.int32)
    if n > 2:
        tgt[1] = 1 % V
    ce, dlog, p = onn.full_softmax_ce(logits.astype(np.float64), tgt.astype(np.int64), n)
    ce32, dlog32, _ = onn.full_softmax_ce(logits.copy(), tgt.astype(np.int64), n)
    Vp = (V + 3) // 4 * 4
    buf = torch.zeros((n, Vp), device="cuda"); buf[:, :V] = dev(logits)
    lr = torch.empty(n, device="cuda"); pr = torch.empty((n, V), device="cuda")
    call("seqrec_full_softmax_ce", ptr(buf), Vp, ptr(dev(tgt)), n, V, 1.0 / n, ptr(lr), ptr(pr), st())
    np.testing.assert_allclose(pr.cpu().numpy(), p, atol=2e-6)
    assert abs(lr.cpu().numpy().astype(np.float64).sum() - ce32) <= 2e-5 * max(1.0, abs(ce32))
    np.testing.assert_allclose(buf[:, :V].cpu().numpy(), dlog32, atol=3e-6 / n + 1e-7)
    # prediction-only mode leaves the logits alone
    buf2 = torch.zeros((n, Vp), device="cuda"); buf2[:, :V] = dev(logits)
    call("seqrec_full_softmax_ce", ptr(buf2), Vp, None, n, V, 0.0, None, ptr(pr), st())
    np.testing.assert_array_equal(buf2[:, :V].cpu().numpy(), logits)


@pytest.mark.parametrize("n,K,H,V", [(5, 7, 64, 30), (333, 2000, 256, 5000), (64, 100, 128, 100), (40, 600, 64, 900),
                                     (33, 3000, 64, 4000), (17, 5000, 64, 6000)])
@pytest.mark.parametrize("bias,lq", [(False, False), (True, True)])
def test_sampled_softmax_ce_vs_oracle(n, K, H, V, bias, lq):
    rng = np.random.default_rng(n + K)
    h = (rng.normal(size=(n, H)) * 0.5).astype(np.float32)
    E = (rng.normal(size=(V, H)) * 0.3).astype(np.float32)
    bout = rng.normal(size=V).astype(np.float32) if bias else None
    logq = np.log(orng.log_uniform_probs(V)).astype(np.float32) if lq else None
    tgt = rng.integers(0, V, size=n).astype(np.int32)
    neg = rng.integers(0, V, size=K).astype(np.int32)
    neg[:3] = tgt[0]           # accidental hits
    ce, dh, dlt, dln, lt, ln = onn.sampled_softmax_ce(h, tgt.astype(np.int64), neg, E, bout, logq, n)
    lnraw = h @ E[neg].T
    lnd = dev(lnraw.astype(np.float32))
    lr = torch.empty(n, device="cuda"); dltd = torch.empty(n, device="cuda")
    for pre_gathered in (False, True):       # per-candidate log-Q vector gathered by the caller: identical results
        lnd = dev(lnraw.astype(np.float32))
        cand = dev(logq[neg]) if (lq and pre_gathered) else None
        call("seqrec_sampled_softmax_ce", ptr(lnd), K, ptr(dev(h)), H, ptr(dev(E)), ptr(dev(bout)) if bias else None,
             ptr(dev(logq)) if lq else None, ptr(cand), ptr(dev(tgt)), ptr(dev(neg)), n, K, 1.0 / n, ptr(lr), ptr(dltd), st())
        assert abs(lr.cpu().numpy().astype(np.float64).sum() - ce) <= 3e-5 * max(1.0, abs(ce))
        np.testing.assert_allclose(dltd.cpu().numpy(), dlt, atol=2e-6)
        np.testing.assert_allclose(lnd.cpu().numpy(), dln, atol=2e-6)
        assert np.all(lnd.cpu().numpy()[0, :3] == 0)


def test_colsum_reduce_mul_fill():
    rng = np.random.default_rng(2)
    X = rng.normal(size=(2603, 768)).astype(np.float32)
    out = torch.ones(768, device="cuda")
    ws = torch.empty(64 * 768, device="cuda")
    call("seqrec_colsum", ptr(dev(X)), 2603, 768, 768, ptr(out), 1, ptr(ws), st())
    np.testing.assert_allclose(out.cpu().numpy(), 1 + X.astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)
    call("seqrec_colsum", ptr(dev(X)), 3, 17, 768, ptr(out), 0, ptr(ws), st())
    np.testing.assert_allclose(out.cpu().numpy()[:17], X[:3, :17].sum(0), rtol=1e-6, atol=1e-6)
    s = torch.zeros(1, device="cuda")
    v = rng.normal(size=100001).astype(np.float32)
    call("seqrec_reduce_sum", ptr(dev(v)), v.size, ptr(s), 0, st())
    assert abs(s.item() - v.astype(np.float64).sum()) < 1e-2
    a = torch.empty(1000, device="cuda"); call("seqrec_fill_f32", ptr(a), 2.5, 1000, st())
    i = torch.empty(1000, dtype=torch.int32, device="cuda"); call("seqrec_fill_i32", ptr(i), 2 ** 31 - 1, 1000, st())
    assert torch.all(a == 2.5) and torch.all(i == 2 ** 31 - 1)
    m = dev(rng.normal(size=1000).astype(np.float32))
    call("seqrec_mul", ptr(a), ptr(m), ptr(a), 1000, st())
    np.testing.assert_array_equal(a.cpu().numpy(), 2.5 * m.cpu().numpy())


def test_sparse_rows_path_equals_dense_adagrad_with_duplicates():
    rng = np.random.default_rng(3)
    V, W, n1, n2 = 3000, 256, 700, 300
    P0 = rng.normal(size=(V, W)).astype(np.float32)
    A0 = np.abs(rng.normal(size=(V, W))).astype(np.float32)
    rows1 = rng.integers(0, 50, size=n1).astype(np.int32)        # heavy duplicates (Zipf head)
    rows2 = rng.integers(0, V, size=n2).astype(np.int32)
    v1 = rng.normal(size=(n1, W)).astype(np.float32); s1 = rng.normal(size=n1).astype(np.float32)
    v2 = rng.normal(size=(n2, W)).astype(np.float32)
    Pd, Ad = dev(P0), dev(A0)
    gt = torch.zeros((V, W), device="cuda")
    slot = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
    r1, r2 = dev(rows1), dev(rows2)
    call("seqrec_rows_scatter_add", ptr(gt), ptr(slot), ptr(r1), ptr(dev(v1)), W, ptr(dev(s1)), n1, W, 0, st())
    call("seqrec_rows_scatter_add", ptr(gt), ptr(slot), ptr(r2), ptr(dev(v2)), W, None, n2, W, n1, st())
    gref = np.zeros((V, W), np.float64)
    np.add.at(gref, rows1, v1.astype(np.float64) * s1[:, None])
    np.add.at(gref, rows2, v2.astype(np.float64))
    np.testing.assert_allclose(gt.cpu().numpy(), gref, atol=2e-4)
    sq = torch.zeros(1, device="cuda")
    call("seqrec_rows_sqnorm", ptr(gt), ptr(slot), ptr(r1), n1, W, 0, ptr(sq), st())
    call("seqrec_rows_sqnorm", ptr(gt), ptr(slot), ptr(r2), n2, W, n1, ptr(sq), st())
    assert abs(sq.item() - (gref ** 2).sum()) <= 1e-4 * (gref ** 2).sum()
    scale = torch.empty(1, device="cuda")
    call("seqrec_clip_scale", ptr(sq), 1.0, ptr(scale), st())
    sc = 1.0 / np.sqrt((gref ** 2).sum())
    assert abs(scale.item() - sc) < 1e-6 * sc + 1e-9
    call("seqrec_rows_adagrad", ptr(Pd), ptr(Ad), ptr(gt), ptr(slot), ptr(r1), n1, W, 0, 0.01, 1e-8, ptr(scale), st())
    call("seqrec_rows_adagrad", ptr(Pd), ptr(Ad), ptr(gt), ptr(slot), ptr(r2), n2, W, n1, 0.01, 1e-8, ptr(scale), st())
    g = gref * sc
    Aref = A0 + g * g
    Pref = P0 - 0.01 * g / (np.sqrt(Aref) + 1e-8)
    np.testing.assert_allclose(Ad.cpu().numpy(), Aref, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(Pd.cpu().numpy(), Pref, rtol=2e-6, atol=2e-7)
    assert torch.all(gt == 0) and torch.all(slot == 2 ** 31 - 1)     # cleared for the next step
    untouched = np.setdiff1d(np.arange(V), np.concatenate([rows1, rows2]))
    np.testing.assert_array_equal(Pd.cpu().numpy()[untouched], P0[untouched])


def test_index_affine_and_filler_rows():
    """seqrec_index_affine_i32 (bit-exact int plumbing, negative positions skipped) and the
    rows[i] < 0 filler convention of the row-sparse kernels (single and multi-list forms)."""
    rng = np.random.default_rng(31)
    n = 5000
    src = rng.integers(0, 2 ** 20, size=n).astype(np.int32)
    dpos = rng.permutation(3 * n)[:n].astype(np.int32)
    dpos[::7] = -1
    dst = torch.full((3 * n,), -5, dtype=torch.int32, device="cuda")
    call("seqrec_index_affine_i32", ptr(dst), ptr(dev(dpos)), ptr(dev(src)), None, n, 3, 11, st())
    ref = np.full(3 * n, -5, np.int32)
    ok = dpos >= 0
    ref[dpos[ok]] = src[ok] * 3 + 11
    np.testing.assert_array_equal(dst.cpu().numpy(), ref)
    spos = rng.integers(0, n, size=777).astype(np.int32)
    out = torch.zeros(777, dtype=torch.int32, device="cuda")
    # ids parked in a FLOAT buffer (small ints = denormal bit patterns) must come back bit-exact
    fbuf = dev(src).view(torch.float32)
    call("seqrec_index_affine_i32", ptr(out), None, ptr(fbuf), ptr(dev(spos)), 777, 1, 0, st())
    np.testing.assert_array_equal(out.cpu().numpy(), src[spos])
    # filler rows
    V, W, m = 500, 64, 300
    rows = rng.integers(0, V, size=m).astype(np.int32)
    rows[::5] = -1
    vals = rng.normal(size=(m, W)).astype(np.float32)
    keep = rows >= 0
    gref = np.zeros((V, W), np.float64)
    np.add.at(gref, rows[keep], vals[keep].astype(np.float64))
    for multi in (False, True):
        P0 = rng.normal(size=(V, W)).astype(np.float32)
        Pd, Ad = dev(P0), torch.zeros((V, W), device="cuda")
        gt = torch.zeros((V, W), device="cuda")
        slot = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
        rd, vd = dev(rows), dev(vals)
        sq = torch.zeros(1, device="cuda"); scale = torch.ones(1, device="cuda")
        if multi:
            arr, cnt = L.rows_jobs([dict(table=Pd, accum=Ad, gtab=gt, slot=slot, rows=rd, vals=vd, ldv=W, row_scale=None, n=m,
                                         width=W, base=0)])
            call("seqrec_rows_scatter_add_multi", arr, cnt, st())
            call("seqrec_rows_sqnorm_multi", arr, cnt, ptr(sq), st())
        else:
            call("seqrec_rows_scatter_add", ptr(gt), ptr(slot), ptr(rd), ptr(vd), W, None, m, W, 0, st())
            call("seqrec_rows_sqnorm", ptr(gt), ptr(slot), ptr(rd), m, W, 0, ptr(sq), st())
        np.testing.assert_allclose(gt.cpu().numpy(), gref, atol=1e-4)
        assert abs(sq.item() - (gref ** 2).sum()) <= 1e-4 * (gref ** 2).sum()
        if multi:
            call("seqrec_rows_adagrad_multi", arr, cnt, 0.01, 1e-8, ptr(scale), st())
        else:
            call("seqrec_rows_adagrad", ptr(Pd), ptr(Ad), ptr(gt), ptr(slot), ptr(rd), m, W, 0, 0.01, 1e-8, ptr(scale), st())
        Pref = P0 - 0.01 * gref / (np.sqrt(gref * gref) + 1e-8)
        np.testing.assert_allclose(Pd.cpu().numpy(), Pref, rtol=1e-5, atol=1e-6)
        assert torch.all(gt == 0) and torch.all(slot == 2 ** 31 - 1)


@pytest.mark.parametrize("W", [64, 128, 256, 192, 512, 768])
def test_scatter_add_multi_with_hot_rows(W):
    """seqrec_rows_scatter_add_multi on Zipf-like rows (one row takes a tenth of the contributions, as the most popular
    item of a c3 batch does): widths 64 / 128 / 256 take the form that combines a workgroup's contributions per row in
    LDS before the atomics, multiples of 256 (c4: 512) the same in 256-wide column blocks, 192 the plain form; two lists share one table (bases 0 and n1), the first has row scales, a
    wider value stride and filler rows.  Gradient table against a float64 sum, owner slots exactly."""
    rng = np.random.default_rng(77 + W)
    V, n1, n2, ldv = 3000, 2501, 1999, W + 8
    def zipf(n):
        r = np.minimum((V * rng.random(n) ** 6).astype(np.int64), V - 1)
        r[rng.random(n) < 0.1] = 7
        return r.astype(np.int32)
    r1, r2 = zipf(n1), zipf(n2)
    r1[::11] = -1
    v1 = rng.normal(size=(n1, ldv)).astype(np.float32); v2 = rng.normal(size=(n2, W)).astype(np.float32)
    s1 = rng.normal(size=n1).astype(np.float32)
    gref = np.zeros((V, W), np.float64)
    k1 = r1 >= 0
    np.add.at(gref, r1[k1], v1[k1, :W].astype(np.float64) * s1[k1, None].astype(np.float64))
    np.add.at(gref, r2, v2.astype(np.float64))
    sref = np.full(V, 2 ** 31 - 1, np.int64)
    np.minimum.at(sref, r1[k1], np.nonzero(k1)[0])
    np.minimum.at(sref, r2, n1 + np.arange(n2))
    gt = torch.zeros((V, W), device="cuda")
    slot = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
    keep = [dev(r1), dev(v1), dev(s1), dev(r2), dev(v2)]
    arr, cnt = L.rows_jobs([dict(table=gt, accum=gt, gtab=gt, slot=slot, rows=keep[0], vals=keep[1], ldv=ldv, row_scale=keep[2],
                                 n=n1, width=W, base=0),
                            dict(table=gt, accum=gt, gtab=gt, slot=slot, rows=keep[3], vals=keep[4], ldv=W, row_scale=None,
                                 n=n2, width=W, base=n1)])
    call("seqrec_rows_scatter_add_multi", arr, cnt, st())
    torch.cuda.synchronize()
    np.testing.assert_allclose(gt.cpu().numpy(), gref, rtol=2e-5, atol=2e-4)
    np.testing.assert_array_equal(slot.cpu().numpy().astype(np.int64), sref)


@pytest.mark.parametrize("W", [256, 192, 512])
def test_gemm_slabs_feed_the_scatter_like_the_reduced_product(W):
    """seqrec_gemm_f32_slabs leaves the split-K partial products in the workspace; a scatter list with n_slabs / slab_stride
    adds them per row in slab order.  Slabs summed in that order == the product seqrec_gemm_f32 writes with the same split
    (its reduce launch adds in the same order), bit for bit; the scatter of the slabs == the scatter of that product to
    rounding of the atomics' order only (rows distinct here: exactly)."""
    import ctypes
    rng = np.random.default_rng(5 + W)
    n, K, sk, V = 700, 1000, 4, 5000          # dEneg-shaped: C[n, W] = A^T[n, K] . B[K, W], split over K
    A = dev(rng.normal(size=(K, n)).astype(np.float32)); B = dev(rng.normal(size=(K, W)).astype(np.float32))
    Cref = torch.zeros((n, W), device="cuda"); ws = torch.zeros(sk * n * W, device="cuda"); ws2 = torch.zeros(sk * n * W, device="cuda")
    call("seqrec_gemm_f32", 0, 0, n, W, K, ptr(A), n, ptr(B), W, ptr(Cref), W, None, 0, sk, ptr(ws), st())
    ns = ctypes.c_int(0)
    call("seqrec_gemm_f32_slabs", 0, 0, n, W, K, ptr(A), n, ptr(B), W, sk, ptr(ws2), ctypes.addressof(ns), st())
    torch.cuda.synchronize()
    assert ns.value == sk
    slabs = ws2.view(sk, n, W)
    acc = slabs[0].clone()
    for z in range(1, sk):
        acc += slabs[z]
    assert torch.equal(acc, Cref)
    rows = dev(rng.permutation(V)[:n].astype(np.int32))
    out = []
    for vals, nsl, stride in ((Cref, 0, 0), (ws2, sk, n * W)):
        gt = torch.zeros((V, W), device="cuda")
        slot = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
        arr, cnt = L.rows_jobs([dict(table=gt, accum=gt, gtab=gt, slot=slot, rows=rows, vals=vals, ldv=W, row_scale=None, n=n,
                                     width=W, base=0, n_slabs=nsl, slab_stride=stride)])
        call("seqrec_rows_scatter_add_multi", arr, cnt, st())
        torch.cuda.synchronize()
        out.append((gt, slot))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    # one slab (split 1) is the whole product
    call("seqrec_gemm_f32_slabs", 0, 0, n, W, K, ptr(A), n, ptr(B), W, 1, ptr(ws2), ctypes.addressof(ns), st())
    call("seqrec_gemm_f32", 0, 0, n, W, K, ptr(A), n, ptr(B), W, ptr(Cref), W, None, 0, 1, None, st())
    torch.cuda.synchronize()
    assert ns.value == 1 and torch.equal(ws2[: n * W].view(n, W), Cref)
    # the deterministic merge does not read slabs
    arr, cnt = L.rows_jobs([dict(table=gt, accum=gt, gtab=gt, slot=slot, rows=rows, vals=ws2, ldv=W, row_scale=None, n=n,
                                 width=W, base=0, n_slabs=2, slab_stride=n * W)])
    lib = L.load()
    nbytes = int(lib.seqrec_rows_merge_workspace_bytes(n, W))
    mws = torch.zeros((nbytes + 3) // 4, device="cuda")
    assert lib.seqrec_rows_merge_sorted(arr, cnt, ptr(mws), nbytes, None) == -3


def test_grouped_gemm_takes_six_problems_of_different_shapes():
    """seqrec_gemm_f32_grouped / _grouped_slabs with SIX problems that share only the layout, K and the split count -- shapes from
    a [1, N] row to a 2000-row product with a ragged edge, one A operand gathered along K through an index (with -1 entries) --
    against float64 products; the slab form leaves problem i at sum_{j<i} n_slabs M_j N_j, slab s at + s M_i N_i (the layout the
    norm launch and the row scatter rely on when dEneg rides in the weight-gradient launch); a seventh problem is refused."""
    import ctypes
    rng = np.random.default_rng(77)
    n, sk = 2555, 4
    shapes = [(256, 512), (256, 256), (256, 768), (1, 768), (2000, 256), (70, 33)]
    Bm = dev(rng.normal(size=(n, 768)).astype(np.float32) * 0.1)
    idx = rng.integers(0, 900, n).astype(np.int32); idx[rng.random(n) < 0.05] = -1
    table = dev(rng.normal(size=(900, 256)).astype(np.float32))
    items, refs = [], []
    for i, (M, N) in enumerate(shapes):
        C = torch.full((M, N), 3.0, device="cuda")
        if i == 0:          # gathered: A row k = table[idx[k]] (zero row for -1)
            A = table
            Ah = table.cpu().numpy().astype(np.float64)[np.maximum(idx, 0)] * (idx >= 0)[:, None]
            items.append((M, N, n, A, 256, Bm, 768, C, N, dev(idx)))
        else:
            A = dev(rng.normal(size=(n, M)).astype(np.float32))
            Ah = A.cpu().numpy().astype(np.float64)
            items.append((M, N, n, A, M, Bm, 768, C, N))
        refs.append(Ah.T @ Bm.cpu().numpy().astype(np.float64)[:, :N])
    descs = L.gemm_descs(items)
    ws = torch.zeros(sk * sum(M * N for M, N in shapes), device="cuda")
    call("seqrec_gemm_f32_grouped", 6, 0, 0, descs, sk, ptr(ws), st())
    torch.cuda.synchronize()
    for it, ref in zip(items, refs):
        got = it[7].cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    ns = ctypes.c_int(0)
    ws.zero_()
    call("seqrec_gemm_f32_grouped_slabs", 6, 0, 0, descs, sk, ptr(ws), ctypes.addressof(ns), st())
    torch.cuda.synchronize()
    off = 0
    for (M, N), ref in zip(shapes, refs):
        got = ws[off:off + ns.value * M * N].view(ns.value, M, N).double().sum(0).cpu().numpy()
        assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
        off += ns.value * M * N
    seven = L.gemm_descs(items + [items[1]])
    assert L.load().seqrec_gemm_f32_grouped(7, 0, 0, seven, sk, ptr(ws), st()) == -1


def test_norm_launch_finishes_grouped_slab_products():
    """seqrec_gemm_f32_grouped_slabs + seqrec_opt_sqnorm_slabs == seqrec_gemm_f32_grouped + seqrec_opt_sqnorm: the products
    written by the norm launch are bit-identical to the reducing form (same slab order), two of them column blocks of ONE
    tensor (dU_zr | dU_h), one a [1, N] row (db); the norm agrees to rounding; a plain dense tensor, a scatter list and
    the batch loss ride along in both."""
    import ctypes
    rng = np.random.default_rng(91)
    n, H, sk = 1500, 64, 4
    A1 = dev(rng.normal(size=(n, H)).astype(np.float32)); A2 = dev(rng.normal(size=(n, H)).astype(np.float32))
    ones = torch.ones((n, 4), device="cuda")
    D = dev(rng.normal(size=(n, 3 * H)).astype(np.float32) * 0.1)
    extra = dev(rng.normal(size=777).astype(np.float32))
    V, W, m = 400, 64, 300
    rows = dev(rng.permutation(V)[:m].astype(np.int32)); vals = dev(rng.normal(size=(m, W)).astype(np.float32))
    lrows = dev(rng.random(n).astype(np.float32))
    res = []
    for slabs in (False, True):
        U = torch.full((H, 3 * H), 5.0, device="cuda"); b = torch.full((3 * H,), 5.0, device="cuda")
        items = [(H, 2 * H, n, A1, H, D, 3 * H, U, 3 * H), (H, H, n, A2, H, D[:, 2 * H:], 3 * H, U[:, 2 * H:], 3 * H),
                 (1, 3 * H, n, ones, 4, D, 3 * H, b, 3 * H)]
        descs = L.gemm_descs(items)
        ws = torch.zeros(sk * sum(i[0] * i[1] for i in items), device="cuda")
        gt = torch.zeros((V, W), device="cuda"); slot = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
        arr, cnt = L.rows_jobs([dict(table=gt, accum=gt, gtab=gt, slot=slot, rows=rows, vals=vals, ldv=W, row_scale=None, n=m,
                                     width=W, base=0)])
        call("seqrec_rows_scatter_add_multi", arr, cnt, st())
        sq = torch.zeros(1, device="cuda"); lo = torch.zeros(2, device="cuda")
        if slabs:
            ns = ctypes.c_int(0)
            call("seqrec_gemm_f32_grouped_slabs", 3, 0, 0, descs, sk, ptr(ws), ctypes.addressof(ns), st())
            assert ns.value == sk
            call("seqrec_opt_sqnorm_slabs", 1, L.ptr_array([extra]), L.i64_array([extra.numel()]), 3, descs, ns.value, ptr(ws), arr, cnt,
                 ptr(sq), ptr(lrows), n, ptr(lo), st())
        else:
            call("seqrec_gemm_f32_grouped", 3, 0, 0, descs, sk, ptr(ws), st())
            call("seqrec_opt_sqnorm", 3, L.ptr_array([extra, U, b]), L.i64_array([extra.numel(), U.numel(), b.numel()]), arr, cnt,
                 ptr(sq), ptr(lrows), n, ptr(lo), st())
        torch.cuda.synchronize()
        res.append((U, b, sq.item(), lo.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][3], res[1][3])
    assert abs(res[0][2] - res[1][2]) <= 2e-6 * res[0][2]
    ref = float((res[0][0].double() ** 2).sum() + (res[0][1].double() ** 2).sum() + (extra.double() ** 2).sum() + (vals.double() ** 2).sum())
    assert abs(res[1][2] - ref) <= 1e-5 * ref
    # argument checks: a product listed with a bias or accumulate is refused, so is a missing workspace
    lib = L.load()
    bad = L.gemm_descs(items); bad[0].accumulate = 1
    assert lib.seqrec_opt_sqnorm_slabs(0, None, None, 3, bad, sk, ptr(ws), None, 0, ptr(sq), None, 0, None, None) == -3
    assert lib.seqrec_opt_sqnorm_slabs(0, None, None, 3, descs, sk, None, None, 0, ptr(sq), None, 0, None, None) == -1


def test_fused_optimizer_launches_equal_the_separate_kernels():
    """seqrec_opt_sqnorm / seqrec_opt_apply (two launches for the whole clipnorm + Adagrad step) against
    the five separate kernels on the same data: dense tensors of odd sizes + two scatter lists with
    duplicates, filler rows and a shared table."""
    rng = np.random.default_rng(77)
    V, W = 4000, 256
    sizes = [196608, 7, 768, 50001]
    n1, n2 = 900, 333
    rows1 = rng.integers(0, 60, size=n1).astype(np.int32); rows1[::11] = -1
    rows2 = rng.integers(0, V, size=n2).astype(np.int32)
    v1 = rng.normal(size=(n1, W)).astype(np.float32) * 0.01
    v2 = rng.normal(size=(n2, W)).astype(np.float32) * 0.01
    P0 = rng.normal(size=(V, W)).astype(np.float32); A0 = np.abs(rng.normal(size=(V, W))).astype(np.float32)
    dp = [rng.normal(size=s).astype(np.float32) for s in sizes]
    da = [np.abs(rng.normal(size=s)).astype(np.float32) for s in sizes]
    dg = [(rng.normal(size=s) * 0.003).astype(np.float32) for s in sizes]
    res = []
    for fused in (False, True):
        Pd, Ad = dev(P0.copy()), dev(A0.copy())
        gt = torch.zeros((V, W), device="cuda")
        slot = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
        r1, r2, d1, d2 = dev(rows1), dev(rows2), dev(v1), dev(v2)
        jobs = [dict(table=Pd, accum=Ad, gtab=gt, slot=slot, rows=r1, vals=d1, ldv=W, row_scale=None, n=n1, width=W, base=0),
                dict(table=Pd, accum=Ad, gtab=gt, slot=slot, rows=r2, vals=d2, ldv=W, row_scale=None, n=n2, width=W, base=n1)]
        arr, cnt = L.rows_jobs(jobs)
        call("seqrec_rows_scatter_add_multi", arr, cnt, st())
        ps, as_, gs = [dev(x.copy()) for x in dp], [dev(x.copy()) for x in da], [dev(x) for x in dg]
        pp, ap, gp, nn = L.ptr_array(ps), L.ptr_array(as_), L.ptr_array(gs), L.i64_array(sizes)
        sq = torch.zeros(2, device="cuda"); sq[1] = 123.0
        scale = torch.zeros(1, device="cuda")
        if fused:
            call("seqrec_opt_sqnorm", len(sizes), gp, nn, arr, cnt, ptr(sq[0:1]), None, 0, None, st())
            call("seqrec_opt_apply", len(sizes), pp, ap, gp, nn, arr, cnt, ptr(sq[0:1]), 0.05, 0.01, 1e-8, ptr(scale), ptr(sq[1:2]), None, None, None, st())
            assert sq[1].item() == 0.0                       # the other norm slot was cleared for the next step
        else:
            call("seqrec_sqnorm_multi", len(sizes), gp, nn, ptr(sq[0:1]), st())
            call("seqrec_rows_sqnorm_multi", arr, cnt, ptr(sq[0:1]), st())
            call("seqrec_clip_scale", ptr(sq[0:1]), 0.05, ptr(scale), st())
            call("seqrec_adagrad_dense_multi", len(sizes), pp, ap, gp, nn, 0.01, 1e-8, ptr(scale), st())
            call("seqrec_rows_adagrad_multi", arr, cnt, 0.01, 1e-8, ptr(scale), st())
        assert torch.all(gt == 0) and torch.all(slot == 2 ** 31 - 1)
        res.append((sq[0].item(), scale.item(), Pd.cpu().numpy(), Ad.cpu().numpy(), [x.cpu().numpy() for x in ps],
                    [x.cpu().numpy() for x in as_]))
    a, b = res
    assert abs(a[0] - b[0]) <= 1e-5 * a[0] and abs(a[1] - b[1]) <= 1e-6 * a[1] and a[1] < 1.0      # clip engaged
    np.testing.assert_allclose(b[2], a[2], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(b[3], a[3], rtol=1e-5, atol=1e-7)
    for x, y in zip(a[4] + a[5], b[4] + b[5]):
        np.testing.assert_allclose(y, x, rtol=1e-5, atol=1e-7)


def test_dense_adagrad_and_norm():
    rng = np.random.default_rng(4)
    n = 200003
    p = rng.normal(size=n).astype(np.float32); a = np.abs(rng.normal(size=n)).astype(np.float32)
    g = (rng.normal(size=n) * 0.001).astype(np.float32)
    sq = torch.zeros(1, device="cuda"); scale = torch.empty(1, device="cuda")
    gd, pd_, ad = dev(g), dev(p), dev(a)
    call("seqrec_sqnorm", ptr(gd), n, ptr(sq), st())
    call("seqrec_clip_scale", ptr(sq), 1.0, ptr(scale), st())
    assert scale.item() == 1.0 and abs(sq.item() - (g.astype(np.float64) ** 2).sum()) < 1e-6
    call("seqrec_adagrad_dense", ptr(pd_), ptr(ad), ptr(gd), n, 0.01, 1e-8, ptr(scale), st())
    ar = a + g * g
    np.testing.assert_allclose(ad.cpu().numpy(), ar, rtol=1e-6)
    np.testing.assert_allclose(pd_.cpu().numpy(), p - np.float32(0.01) * g / (np.sqrt(ar) + np.float32(1e-8)), rtol=2e-6, atol=1e-7)


def test_counter_rng_bit_exact_vs_oracle():
    V, K = 100003, 2000
    probs = orng.log_uniform_probs(V)
    th, al = orng.build_alias_table(probs)
    thd = torch.from_numpy(th.view(np.int32).copy()).cuda(); ald = dev(al)
    out = torch.empty(K, dtype=torch.int32, device="cuda")
    for seed, step in [(0, 0), (7, 3), (2 ** 40 + 5, 123456)]:
        call("seqrec_sample_negatives", seed, step, K, ptr(thd), ptr(ald), V, ptr(out), st())
        np.testing.assert_array_equal(out.cpu().numpy(), orng.sample_negatives(seed, step, K, th, al))
    # the draws follow the proposal (head items dominate)
    assert (out.cpu().numpy() < 1000).mean() > 0.4
    # fused draw + row gather + log-Q gather: same ids, bit-identical rows
    rng = np.random.default_rng(8)
    for W in (256, 6):
        tab = rng.normal(size=(V, W)).astype(np.float32)
        lq = np.log(probs).astype(np.float32)
        neg2 = torch.empty(K, dtype=torch.int32, device="cuda"); rows = torch.empty((K, W), device="cuda"); lqo = torch.empty(K, device="cuda")
        call("seqrec_sample_gather", 7, 3, K, ptr(thd), ptr(ald), V, ptr(dev(tab)), W, ptr(dev(lq)), ptr(neg2), ptr(rows), ptr(lqo), st())
        ids = orng.sample_negatives(7, 3, K, th, al)
        np.testing.assert_array_equal(neg2.cpu().numpy(), ids)
        np.testing.assert_array_equal(rows.cpu().numpy(), tab[ids])
        np.testing.assert_array_equal(lqo.cpu().numpy(), lq[ids])
        # the same draw + gathers riding in the launch that re-packs U (seqrec_rnn_pack_u_sample): both outputs as the two calls
        for cell, H in ((2, 256), (0, 64), (1, 128)):
            G = {0: 1, 1: 4, 2: 3}[cell]
            U = dev(rng.normal(size=(H, G * H)).astype(np.float32))
            nf = int(L.load().seqrec_rnn_upack_floats(cell, H))
            up_a = torch.zeros(nf, device="cuda"); up_b = torch.full((nf,), 7.0, device="cuda")
            neg3 = torch.empty(K, dtype=torch.int32, device="cuda"); rows3 = torch.empty((K, W), device="cuda"); lq3 = torch.empty(K, device="cuda")
            call("seqrec_rnn_pack_u_stepwise", cell, H, ptr(U), ptr(up_a), st())
            call("seqrec_rnn_pack_u_sample", cell, H, ptr(U), ptr(up_b), 7, 3, K, ptr(thd), ptr(ald), V, ptr(dev(tab)), W, ptr(dev(lq)),
                 ptr(neg3), ptr(rows3), ptr(lq3), st())
            torch.cuda.synchronize()
            assert torch.equal(up_a, up_b)
            assert torch.equal(neg3, neg2) and torch.equal(rows3, rows) and torch.equal(lq3, lqo)
    rk = np.arange(50, dtype=np.int64) * 977 + 13
    m = torch.zeros((50, 12), device="cuda")
    call("seqrec_dropout_mask", 11, 35, ptr(dev(rk)), 50, 10, 12, 0.3, ptr(m), st())
    ref = orng.dropout_mask(11, 35, rk, 10, 0.3)
    np.testing.assert_array_equal(m.cpu().numpy()[:, :10], ref)
    assert np.all(m.cpu().numpy()[:, 10:] == 0) and 0.55 < (ref > 0).mean() < 0.85


def test_rank_count_vs_numpy():
    rng = np.random.default_rng(6)
    n, H, V = 300, 128, 7001
    h = rng.normal(size=(n, H)).astype(np.float32)
    E = rng.normal(size=(V, H)).astype(np.float32)
    b = rng.normal(size=V).astype(np.float32)
    tgt = rng.integers(0, V, size=n).astype(np.int32)
    rank = torch.zeros(n, dtype=torch.int32, device="cuda"); thr = torch.empty(n, device="cuda")
    call("seqrec_rank_count", ptr(dev(h)), H, ptr(dev(E)), ptr(dev(b)), ptr(dev(tgt)), n, V, ptr(rank), ptr(thr), st())
    sc = h.astype(np.float64) @ E.T.astype(np.float64) + b
    ts = sc[np.arange(n), tgt]
    ref = (sc > ts[:, None]).sum(1)
    got = rank.cpu().numpy()
    assert np.abs(got - ref).max() <= 2 and (got == ref).mean() > 0.97     # fp32 near-ties only


@pytest.mark.parametrize("cell", ["gru", "lstm", "simplernn"])
@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("H,B,maxlen,act", [(64, 5, 6, "relu"), (64, 37, 12, "tanh"), (128, 100, 9, "relu"),
                                            (256, 70, 20, "relu"), (256, 512, 49, "tanh"), (128, 40, 10, "linear"),
                                            (512, 33, 7, "relu"),
                                            # > 8 row blocks: the wide BPTT tile of the GRU (16 x 64, full K per wave)
                                            (128, 300, 8, "relu"), (512, 160, 5, "relu"), (256, 200, 6, "linear")])
def test_stepwise_scan_vs_oracle_and_persistent(cell, H, B, maxlen, act, graph):
    """The launch-per-step scan (rnn_step.hip) against the oracle, and close to the persistent scan
    (same fp32 MFMA chains, only the accumulation split differs)."""
    rng = np.random.default_rng(H * 3 + B + maxlen)
    rb, XW, U = packed_scan_inputs(rng, cell, H, B, maxlen)
    n, G = rb.n_tok, onn.N_GATES[cell]
    ci = L.CELL[cell]
    dH = (rng.normal(size=(n, H)) * 0.5).astype(np.float32)
    ref = oracle_scan(cell, act, rb, XW, U, dH)
    Hout = torch.full((n, H), float("nan"), device="cuda"); gates = torch.full((n, G * H), float("nan"), device="cuda")
    aux = torch.full((n, H), float("nan"), device="cuda")
    up = torch.empty(int(L.load().seqrec_rnn_upack_floats(ci, H)), device="cuda")
    XWd, Ud = dev(XW), dev(U)
    call("seqrec_rnn_pack_u_stepwise", ci, H, ptr(Ud), ptr(up), st())
    so = rb.step_off
    sod = dev(rb.step_off)
    for rep in range(1 + graph):          # graph mode: the second call replays the captured launches
        call("seqrec_rnn_fwd_stepwise", ci, L.ACT[act], H, H, rb.T, rb.B, ptr(sod), so.ctypes.data, ptr(XWd), ptr(Hout), ptr(gates),
             ptr(aux), ptr(up), None, graph, st())
    got = Hout.cpu().numpy()
    scale = max(1.0, np.abs(ref["H"]).max())
    assert np.abs(got - ref["H"]).max() <= 3e-5 * scale
    dPre = torch.full((n, G * H), float("nan"), device="cuda")
    ws = torch.full((2 * n * H,), float("nan"), device="cuda")
    dHd = dev(dH)
    for rep in range(1 + graph):
        call("seqrec_rnn_bwd_stepwise", ci, L.ACT[act], H, H, rb.T, rb.B, ptr(sod), so.ctypes.data, n, ptr(dHd), ptr(Hout), ptr(gates),
             ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, graph, st())
    gp = dPre.cpu().numpy()
    s2 = max(1.0, np.abs(ref["dPre"]).max())
    bad = np.abs(gp - ref["dPre"]) > 1e-4 * s2
    assert np.isfinite(gp).all() and bad.mean() < 2e-4, (bad.mean(), np.abs(gp - ref["dPre"]).max())
    # persistent scan on the same inputs
    H2 = torch.empty((n, H), device="cuda"); g2 = torch.empty((n, G * H), device="cuda"); a2 = torch.empty((n, H), device="cuda")
    call("seqrec_rnn_pack_u", ci, H, ptr(Ud), ptr(up), st())
    call("seqrec_rnn_fwd", ci, L.ACT[act], H, H, rb.T, rb.B, ptr(dev(rb.step_off)), ptr(XWd), ptr(H2), ptr(g2), ptr(a2), ptr(up), st())
    assert np.abs(H2.cpu().numpy() - got).max() <= 2e-5 * scale
    if graph:
        assert L.load().seqrec_graph_cache_clear() == 0


def _run_scan_both_ways(lib, cell, act, H, H_real, rb, XWd, up, dHd, rmask, graph=0):
    """forward + BPTT through the step-wise entry points -> (Hout, gates, aux, dPre) as numpy"""
    n, G, ci = rb.n_tok, onn.N_GATES[cell], L.CELL[cell]
    so = rb.step_off
    Hout = torch.full((n, H), float("nan"), device="cuda"); gates = torch.full((n, G * H), float("nan"), device="cuda")
    aux = torch.full((n, H), float("nan"), device="cuda")
    call("seqrec_rnn_fwd_stepwise", ci, L.ACT[act], H, H_real, rb.T, rb.B, None, so.ctypes.data, ptr(XWd), ptr(Hout), ptr(gates),
         ptr(aux), ptr(up), ptr(rmask), graph, st())
    dPre = torch.full((n, G * H), float("nan"), device="cuda")
    ws = torch.full((2 * n * H,), float("nan"), device="cuda")
    call("seqrec_rnn_bwd_stepwise", ci, L.ACT[act], H, H_real, rb.T, rb.B, None, so.ctypes.data, n, ptr(dHd), ptr(Hout), ptr(gates),
         ptr(aux), ptr(dPre), ptr(up), ptr(ws), ptr(rmask), graph, st())
    torch.cuda.synchronize()
    return Hout.cpu().numpy(), gates.cpu().numpy(), aux.cpu().numpy(), dPre.cpu().numpy()


@pytest.mark.parametrize("cell,rd", [("gru", 0), ("gru", 1), ("lstm", 0), ("lstm", 1), ("simplernn", 0), ("simplernn", 1)])
@pytest.mark.parametrize("H,B,maxlen,act", [(256, 512, 40, "tanh"), (256, 700, 9, "relu"), (128, 333, 25, "tanh"), (64, 40, 12, "linear"),
                                            (512, 100, 17, "tanh"), (512, 512, 12, "relu"), (256, 5, 60, "tanh"), (256, 16, 1, "tanh"),
                                            (128, 1100, 6, "tanh")])       # 69 row blocks: more than one launch's flag buffer holds (64)
def test_cluster_scan_equals_stepwise_scan(cell, rd, H, B, maxlen, act):
    """The one-launch cluster form of the scans (rnn_cluster.hip GRU, rnn_cluster2.hip LSTM / SimpleRNN: in-kernel exchange
    between the column-slice workgroups of a row block; with and without recurrent dropout) against the launch-per-product
    form on the same inputs: same arithmetic element for element, so Hout / gates / aux agree bit for bit and dPre to the
    last bits; run twice (flag epochs advance between calls).  The LSTM forward with recurrent dropout at H = 512 has no
    cluster form (its masks do not fit the registers next to the kernel slices): both runs are step-wise there."""
    lib = L.load()
    rng = np.random.default_rng(H + B + maxlen)
    rb, XW, U = packed_scan_inputs(rng, cell, H, B, maxlen)
    n, G = rb.n_tok, onn.N_GATES[cell]
    ci = L.CELL[cell]
    up = torch.empty(int(lib.seqrec_rnn_upack_floats(ci, H)), device="cuda")
    XWd, Ud = dev(XW), dev(U)
    call("seqrec_rnn_pack_u_stepwise", ci, H, ptr(Ud), ptr(up), st())
    dHd = dev((rng.normal(size=(n, H)) * 0.5).astype(np.float32))
    rmask = dev(((rng.random((G, rb.B, H)) > 0.3) / 0.7).astype(np.float32)) if rd else None
    out = {}
    try:
        for mode in (0, 1, 1):
            lib.seqrec_debug_scan_cluster(mode)
            got = _run_scan_both_ways(lib, cell, act, H, H - 3, rb, XWd, up, dHd, rmask)
            if mode == 0:
                out = got
            else:
                for name, x, y in zip(("Hout", "gates", "aux", "dPre"), out, got):
                    if cell == "simplernn" and name in ("gates", "aux"):
                        continue                                   # unused by this cell
                    assert np.isfinite(y).all(), name
                    if name == "dPre":
                        # same products and sums, but the element-wise epilogues are separate code (FMA contraction may
                        # differ), and steps with more than 8 row blocks use the wide tile in the step-wise GRU BPTT (one K
                        # chain per wave instead of four partial chains): last-bit differences
                        assert np.abs(x - y).max() <= 2e-5 * max(1.0, np.abs(x).max()), name
                    else:
                        np.testing.assert_array_equal(x, y, err_msg=name)
        assert lib.seqrec_cluster_scan_errors(st()) == 0          # no bounded spin ran out
    finally:
        lib.seqrec_debug_scan_cluster(-1)


@pytest.mark.parametrize("cell", ["lstm", "simplernn", "gru"])
def test_cluster_scan_at_c4_size_vs_oracle(cell):
    """The default (cluster) form straight against the oracle at BASELINE config 4's cell size: H = 512, 512 sessions,
    up to 49 steps (LSTM: the reference's own cell, model.py:349-352)."""
    lib = L.load()
    H, B, maxlen, act = 512, 512, 49, "tanh"
    rng = np.random.default_rng(11)
    rb, XW, U = packed_scan_inputs(rng, cell, H, B, maxlen)
    n = rb.n_tok
    dH = (rng.normal(size=(n, H)) * 0.5).astype(np.float32)
    ref = oracle_scan(cell, act, rb, XW, U, dH)
    up = torch.empty(int(lib.seqrec_rnn_upack_floats(L.CELL[cell], H)), device="cuda")
    call("seqrec_rnn_pack_u_stepwise", L.CELL[cell], H, ptr(dev(U)), ptr(up), st())
    Hout, _, _, dPre = _run_scan_both_ways(lib, cell, act, H, H, rb, dev(XW), up, dev(dH), None)
    assert np.abs(Hout - ref["H"]).max() <= 3e-5 * max(1.0, np.abs(ref["H"]).max())
    bad = np.abs(dPre - ref["dPre"]) > 1e-4 * max(1.0, np.abs(ref["dPre"]).max())
    assert np.isfinite(dPre).all() and bad.mean() < 2e-4, (bad.mean(), np.abs(dPre - ref["dPre"]).max())
    assert lib.seqrec_cluster_scan_errors(st()) == 0


def test_graph_replays_queued_back_to_back_keep_their_own_arguments():
    """Two DIFFERENT batches with the same number of steps (= the same captured launch sequence) through use_graph = 1,
    enqueued back to back WITHOUT a host sync behind a long-running kernel -- the second call rewrites graph nodes while the
    first replay is still queued.  Both results must equal the eager issue bit for bit (rnn_step.hip keeps a ring of
    executables per sequence and rewrites one only after the event behind its last launch has completed).  Three rounds, so
    that the ring wraps."""
    lib = L.load()
    H, act, cell = 256, "tanh", "lstm"
    ci, G = L.CELL[cell], 4
    rng = np.random.default_rng(21)
    up = torch.empty(int(lib.seqrec_rnn_upack_floats(ci, H)), device="cuda")
    U = (rng.normal(size=(H, G * H)) * (0.6 / np.sqrt(H))).astype(np.float32)
    call("seqrec_rnn_pack_u_stepwise", ci, H, ptr(dev(U)), ptr(up), st())
    batches = []
    for Bn in (70, 33, 120, 20, 64, 97):
        sess = [rng.integers(0, 50, size=int(rng.integers(2, 14))).tolist() for _ in range(Bn - 1)] + [list(range(14))]   # same T = 13
        rb = B_.pack_sessions(sess)
        assert rb.T == 13
        n = rb.n_tok
        batches.append(dict(rb=rb, XW=dev((rng.normal(size=(n, G * H)) * 0.7).astype(np.float32)),
                            dH=dev((rng.normal(size=(n, H)) * 0.5).astype(np.float32))))

    def run(b, graph):
        rb, n = b["rb"], b["rb"].n_tok
        so = rb.step_off
        o = dict(Hout=torch.full((n, H), float("nan"), device="cuda"), gates=torch.full((n, G * H), float("nan"), device="cuda"),
                 aux=torch.full((n, H), float("nan"), device="cuda"), dPre=torch.full((n, G * H), float("nan"), device="cuda"),
                 ws=torch.full((2 * n * H,), float("nan"), device="cuda"))
        call("seqrec_rnn_fwd_stepwise", ci, L.ACT[act], H, H, rb.T, rb.B, None, so.ctypes.data, ptr(b["XW"]), ptr(o["Hout"]),
             ptr(o["gates"]), ptr(o["aux"]), ptr(up), None, graph, st())
        call("seqrec_rnn_bwd_stepwise", ci, L.ACT[act], H, H, rb.T, rb.B, None, so.ctypes.data, n, ptr(b["dH"]), ptr(o["Hout"]),
             ptr(o["gates"]), ptr(o["aux"]), ptr(o["dPre"]), ptr(up), ptr(o["ws"]), None, graph, st())
        return o
    try:
        lib.seqrec_debug_scan_cluster(0)             # the step-wise form is the one that is captured
        eager = [run(b, 0) for b in batches]
        torch.cuda.synchronize()
        big = torch.randn(8192, 8192, device="cuda")
        for _ in range(3):
            big = (big @ big) * 1e-4                 # ~10 ms of queued work: the replays below pile up behind it
        replay = [run(b, 1) for b in batches]        # no sync in between
        torch.cuda.synchronize()
        for e, r in zip(eager, replay):
            for k in ("Hout", "gates", "aux", "dPre"):
                np.testing.assert_array_equal(e[k].cpu().numpy(), r[k].cpu().numpy(), err_msg=k)
        assert lib.seqrec_release_stream(st()) == 0  # drops the captured graphs of this stream (and its cluster flags)
        assert lib.seqrec_graph_cache_clear() == 0
    finally:
        lib.seqrec_debug_scan_cluster(-1)


def test_cluster_scan_timeout_is_counted_and_poisons_the_output():
    """A wait of the cluster kernels that runs out (forced: 1 poll) is counted, and the wave writes NaN into the output it
    owns instead of leaving a plausible-looking stale value (include/seqrec_hip.h: seqrec_cluster_scan_errors)."""
    lib = L.load()
    rng = np.random.default_rng(3)
    cell, H, act = "lstm", 256, "tanh"
    rb, XW, U = packed_scan_inputs(rng, cell, H, 200, 30)
    up = torch.empty(int(lib.seqrec_rnn_upack_floats(L.CELL[cell], H)), device="cuda")
    call("seqrec_rnn_pack_u_stepwise", L.CELL[cell], H, ptr(dev(U)), ptr(up), st())
    dHd = dev(np.zeros((rb.n_tok, H), np.float32))
    assert lib.seqrec_cluster_scan_errors_reset(st()) == 0
    try:
        lib.seqrec_debug_cluster_spin_limit(1)
        Hout, _, _, dPre = _run_scan_both_ways(lib, cell, act, H, H, rb, dev(XW), up, dHd, None)
        assert lib.seqrec_cluster_scan_errors(st()) > 0
        assert np.isnan(Hout).any() or np.isnan(dPre).any()       # whichever direction lost a wait
    finally:
        lib.seqrec_debug_cluster_spin_limit(0)
        lib.seqrec_cluster_scan_errors_reset(st())
    Hout, _, _, _ = _run_scan_both_ways(lib, cell, act, H, H, rb, dev(XW), up, dHd, None)
    assert np.isfinite(Hout).all() and lib.seqrec_cluster_scan_errors(st()) == 0


def test_opt_apply_refuses_a_degenerate_scale_and_reports_it():
    """seqrec_opt_apply with a squared norm of inf (what one overflowing gradient value produces: clip scale 0, the whole
    step a silent no-op), NaN, a negative number, or a zero / non-finite divisor: SEQREC_STATUS_* is set and NOTHING is
    written (weights, accumulators); a healthy call leaves the status word alone."""
    rng = np.random.default_rng(8)
    n = 5000
    p0 = rng.normal(size=n).astype(np.float32); g0 = rng.normal(size=n).astype(np.float32)
    for sqv, div, want in ((float("inf"), None, 1), (float("nan"), None, 1), (-1.0, None, 1), (4.0, 0.0, 2), (4.0, float("inf"), 2),
                           (4.0, float("nan"), 2), (4.0, None, 0), (4.0, 3.0, 0)):
        p, a, g = dev(p0.copy()), dev(np.zeros(n, np.float32)), dev(g0)
        sq, scale = dev(np.array([sqv], np.float32)), dev(np.zeros(1, np.float32))
        dv = None if div is None else dev(np.array([div], np.float32))
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        call("seqrec_opt_apply", 1, L.ptr_array([p]), L.ptr_array([a]), L.ptr_array([g]), L.i64_array([n]), None, 0, ptr(sq), 1.0, 0.01,
             1e-8, ptr(scale), None, ptr(dv), ptr(status), None, st())
        assert int(status.item()) == want, (sqv, div, int(status.item()))
        if want:
            np.testing.assert_array_equal(p.cpu().numpy(), p0)
            assert float(a.abs().sum().item()) == 0.0
        else:
            assert np.abs(p.cpu().numpy() - p0).max() > 0


def test_gather_rows_bounded_flags_an_index_outside_the_table():
    rng = np.random.default_rng(2)
    tab = rng.normal(size=(50, 64)).astype(np.float32)
    ids = np.array([3, -1, 49, 50, 7, 10 ** 6], np.int32)
    out = torch.full((6, 64), float("nan"), device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    call("seqrec_gather_rows_bounded", ptr(dev(tab)), 50, ptr(dev(ids)), ptr(out), 6, 64, None, None, 0, ptr(status), st())
    o = out.cpu().numpy()
    np.testing.assert_array_equal(o[[0, 2, 4]], tab[[3, 49, 7]])
    assert np.all(o[[1, 3, 5]] == 0) and int(status.item()) == 8
    status.zero_()
    call("seqrec_gather_rows_bounded", ptr(dev(tab)), 50, ptr(dev(ids[:3])), ptr(out), 3, 64, None, None, 0, ptr(status), st())
    assert int(status.item()) == 0


@pytest.mark.parametrize("cell,H,B,maxlen", [("gru", 256, 300, 30), ("gru", 128, 77, 12), ("lstm", 64, 50, 9)])
def test_bptt_with_its_input_gradient_in_parts(cell, H, B, maxlen):
    """seqrec_rnn_bwd_stepwise_parts (dHout = 3 split-K slabs + scale[q] * table[index[q]], index -1 = no term) against
    seqrec_rnn_bwd_stepwise on the materialised sum (same order of additions): the cluster form of the GRU adds the parts
    in its own loads, the step-wise form (forced for the GRU, the only one for the LSTM) sums them into the scratch array
    first -- dPre bit for bit in both."""
    import ctypes
    lib = L.load()
    rng = np.random.default_rng(H + B)
    rb, XW, U = packed_scan_inputs(rng, cell, H, B, maxlen)
    n, G = rb.n_tok, {"gru": 3, "lstm": 4}[cell]
    ci = L.CELL[cell]
    up = torch.empty(int(lib.seqrec_rnn_upack_floats(ci, H)), device="cuda")
    XWd, Ud = dev(XW), dev(U)
    call("seqrec_rnn_pack_u_stepwise", ci, H, ptr(Ud), ptr(up), st())
    so = rb.step_off
    ns, V = 3, 500
    slabs = dev((rng.normal(size=(ns, n + 5, H)) * 0.3).astype(np.float32))          # slab stride > n * H
    table = dev(rng.normal(size=(V, H + 8)).astype(np.float32)); idx = rng.integers(0, V, size=n).astype(np.int32)
    idx[::7] = -1
    idxd, scale = dev(idx), dev(rng.normal(size=n).astype(np.float32))
    total = (slabs[0, :n] + slabs[1, :n]) + slabs[2, :n]
    term = scale[:, None] * table[torch.from_numpy(np.maximum(idx, 0)).cuda().long(), :H]
    total = total + torch.where(torch.from_numpy(idx >= 0).cuda()[:, None], term, torch.zeros_like(term))
    parts = L.dh_parts(slabs, ns, (n + 5) * H, add_table=table, add_index=idxd, add_scale=scale, add_ld=H + 8)
    try:
        for mode in ((1, 0) if cell == "gru" else (-1,)):
            lib.seqrec_debug_scan_cluster(mode)
            Hout = torch.zeros((n, H), device="cuda"); gates = torch.zeros((n, G * H), device="cuda"); aux = torch.zeros((n, H), device="cuda")
            call("seqrec_rnn_fwd_stepwise", ci, L.ACT["tanh"], H, H, rb.T, rb.B, None, so.ctypes.data, ptr(XWd), ptr(Hout), ptr(gates),
                 ptr(aux), ptr(up), None, 0, st())
            res = []
            for use_parts in (False, True):
                dPre = torch.full((n, G * H), float("nan"), device="cuda"); ws = torch.zeros(2 * n * H, device="cuda")
                if use_parts:
                    scratch = torch.full((n, H), float("nan"), device="cuda")
                    call("seqrec_rnn_bwd_stepwise_parts", ci, L.ACT["tanh"], H, H, rb.T, rb.B, None, so.ctypes.data, n,
                         ctypes.addressof(parts), ptr(scratch), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, 0, st())
                else:
                    call("seqrec_rnn_bwd_stepwise", ci, L.ACT["tanh"], H, H, rb.T, rb.B, None, so.ctypes.data, n, ptr(total), ptr(Hout),
                         ptr(gates), ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, 0, st())
                torch.cuda.synchronize()
                res.append(dPre)
            assert torch.isfinite(res[1]).all()
            assert torch.equal(res[0], res[1]), (cell, mode)
        assert lib.seqrec_cluster_scan_errors(st()) == 0
        # argument checks
        bad = L.dh_parts(slabs, 0, (n + 5) * H)
        assert lib.seqrec_rnn_bwd_stepwise_parts(ci, 0, H, H, rb.T, rb.B, None, so.ctypes.data, n, ctypes.addressof(bad), ptr(total), ptr(Hout),
                                                 ptr(gates), ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, 0, None) == -1
        assert lib.seqrec_rnn_bwd_stepwise_parts(ci, 0, H, H, rb.T, rb.B, None, so.ctypes.data, n, ctypes.addressof(parts), None, ptr(Hout),
                                                 ptr(gates), ptr(aux), ptr(dPre), ptr(up), ptr(ws), None, 0, None) == -1
    finally:
        lib.seqrec_debug_scan_cluster(-1)


@pytest.mark.parametrize("width,V,sizes", [(256, 5000, (2603, 2000, 2603)), (1, 300, (700, 50)), (100, 64, (900,)),
                                           (512, 2000, (4000, 4000)), (2048, 40, (300, 17)), (64, 7, (5000, 3, 129)),
                                           (256, 5000, (6000, 5001)), (128, 3000, (8192,)), (128, 3000, (8193,))])
def test_sorted_merge_is_bitwise_reproducible_and_equals_the_atomic_scatter(width, V, sizes):
    """csrc/merge.hip (deterministic row-gradient merge) against numpy float64 sums and against the
    float-atomic scatter: same gradient table to rounding, same owner slots exactly, and two invocations on
    the same inputs agree bit for bit.  Rows are Zipf-skewed so that single rows collect hundreds of
    contributions (runs longer than one 64-position tile and longer than two: the partial chain), V = 7 puts
    thousands on one row; -1 rows are fillers; one list carries a per-contribution scale.  (A one-workgroup bitonic sort
    in LDS for up to 8 192 contributions was tried in place of the device radix sort: 292 against 234 us per call.)"""
    rng = np.random.default_rng(width + V)
    jobs, base = [], 0
    ref = np.zeros((V, width), np.float64)
    first = np.full(V, 2 ** 31 - 1, np.int64)
    for li, n in enumerate(sizes):
        rows = np.minimum(rng.zipf(1.3, size=n) - 1, V - 1).astype(np.int32)
        rows[rng.random(n) < 0.02] = -1
        ld = width + (4 if li == 1 else 0)                        # a strided value matrix
        vals = rng.normal(size=(n, ld)).astype(np.float32)
        scale = rng.normal(size=n).astype(np.float32) if li == 0 else None
        for i in np.nonzero(rows >= 0)[0]:
            ref[rows[i]] += vals[i, :width].astype(np.float64) * (1.0 if scale is None else float(scale[i]))
            first[rows[i]] = min(first[rows[i]], base + i)
        jobs.append(dict(rows=dev(rows), vals=dev(vals), ldv=ld, row_scale=None if scale is None else dev(scale), n=n,
                         width=width, base=base))
        base += n
    total = base
    nbytes = int(L.load().seqrec_rows_merge_workspace_bytes(total, width))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(2):
        gt = torch.zeros((V, width), device="cuda")
        sl = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
        js = [dict(j, table=None, accum=None, gtab=gt, slot=sl) for j in jobs[::-1]]      # any order: sorted by base inside
        arr, cnt = L.rows_jobs(js)
        ws.fill_(0xA5 if rep else 0)                                                     # stale workspace contents must not matter
        call("seqrec_rows_merge_sorted", arr, cnt, ptr(ws), nbytes, st())
        outs.append((gt.cpu().numpy(), sl.cpu().numpy()))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][1].astype(np.int64), first)
    scale_ = max(1.0, np.abs(ref).max())
    assert np.abs(outs[0][0] - ref).max() <= 2e-5 * scale_
    # the atomic path on the same lists
    gt = torch.zeros((V, width), device="cuda")
    sl = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
    arr, cnt = L.rows_jobs([dict(j, table=None, accum=None, gtab=gt, slot=sl) for j in jobs])
    call("seqrec_rows_scatter_add_multi", arr, cnt, st())
    np.testing.assert_array_equal(sl.cpu().numpy(), outs[0][1])
    assert np.abs(gt.cpu().numpy() - ref).max() <= 2e-5 * scale_
    # bad arguments: overlapping index ranges, mixed tables
    bad = [dict(j, table=None, accum=None, gtab=gt, slot=sl) for j in jobs]
    if len(bad) > 1:
        bad[1] = dict(bad[1], base=bad[0]["base"])
        arr, cnt = L.rows_jobs(bad)
        with pytest.raises(L.SeqrecError):
            call("seqrec_rows_merge_sorted", arr, cnt, ptr(ws), nbytes, st())
    arr, cnt = L.rows_jobs([dict(jobs[0], table=None, accum=None, gtab=gt, slot=sl)])
    with pytest.raises(L.SeqrecError):
        call("seqrec_rows_merge_sorted", arr, cnt, ptr(ws), 16, st())                    # workspace too small


def test_ordered_norm_equals_the_atomic_norm_and_is_reproducible():
    rng = np.random.default_rng(5)
    V, W = 3000, 256
    rows = np.minimum(rng.zipf(1.2, size=4000) - 1, V - 1).astype(np.int32)
    vals = rng.normal(size=(4000, W)).astype(np.float32)
    gt = torch.zeros((V, W), device="cuda")
    sl = torch.full((V,), 2 ** 31 - 1, dtype=torch.int32, device="cuda")
    job = dict(table=None, accum=None, gtab=gt, slot=sl, rows=dev(rows), vals=dev(vals), ldv=W, row_scale=None, n=4000, width=W, base=0)
    arr, cnt = L.rows_jobs([job])
    call("seqrec_rows_scatter_add_multi", arr, cnt, st())
    dense = [dev(rng.normal(size=s).astype(np.float32)) for s in ((256, 768), (768,), (3,))]
    gp = L.ptr_array(dense)
    nn = L.i64_array([t.numel() for t in dense])
    ref = float(sum((t.double() ** 2).sum().item() for t in dense) + (gt.double() ** 2).sum().item())
    npart = int(L.load().seqrec_opt_sqnorm_ordered_floats(3, 1, 4000))
    part = torch.empty(npart, device="cuda")
    got = []
    for rep in range(2):
        sq = torch.full((1,), float("nan"), device="cuda")
        part.fill_(float(rep))
        call("seqrec_opt_sqnorm_ordered", 3, gp, nn, arr, cnt, ptr(part), npart, ptr(sq), 0, None, 0, None, st())
        got.append(sq.cpu().numpy().copy())
    np.testing.assert_array_equal(got[0], got[1])
    assert abs(float(got[0][0]) - ref) <= 1e-5 * ref
    sq = torch.zeros(1, device="cuda")
    lrows = dev(rng.random(2603).astype(np.float32) * 12)
    lo = torch.full((2,), float("nan"), device="cuda")
    call("seqrec_opt_sqnorm", 3, gp, nn, arr, cnt, ptr(sq), ptr(lrows), 2603, ptr(lo), st())       # + the batch loss in the same launch
    assert abs(float(sq.item()) - ref) <= 1e-5 * ref
    want = float(lrows.double().sum().item())
    assert abs(float(lo[0].item()) - want) <= 1e-5 * want and abs(float(lo[1].item()) - want / 2603) <= 1e-5 * want / 2603
    lo2 = torch.full((2,), float("nan"), device="cuda")
    call("seqrec_loss_reduce", ptr(lrows), 2603, ptr(lo2), st())
    assert torch.equal(lo, lo2)                                                                     # same fixed order


@pytest.fixture(params=[0, 2, 3])
def forced_gemm_tile(request):
    """0 = the library's own tile choice; 2 / 3 force the 128x64 / 128x128 LDS-DMA tiles (grouped form: 128x64)."""
    lib = L.load()
    lib.seqrec_debug_gemm_tile(request.param, min(request.param, 2))
    yield request.param
    lib.seqrec_debug_gemm_tile(0, 0)


@pytest.mark.parametrize("splitk", [1, 5])
def test_gemm_fused_gathered_a_operand_and_row_add(splitk, forced_gemm_tile):
    """seqrec_gemm_f32_fused against numpy: (1) A rows read through an index (x.W with x = E[ids], -1 = zero row),
    bit-identical to the GEMM on the materialised gather; (2) the index along K for the stored-KxM form
    (Hout[prev]^T . dPre, ragged sizes); (3) the row add of the final write (dH += dlt * Eout[tgt]) with and
    without split-K."""
    import ctypes
    rng = np.random.default_rng(3 + splitk)
    V, D, N, n = 700, 100, 200, 333
    E = rng.normal(size=(V, D)).astype(np.float32)
    W = rng.normal(size=(D, N)).astype(np.float32)
    ids = rng.integers(0, V, size=n).astype(np.int32)
    ids[::17] = -1
    X = np.where(ids[:, None] >= 0, E[np.maximum(ids, 0)], 0).astype(np.float32)
    Ed, Wd, idd = dev(E), dev(W), dev(ids)
    ws = torch.empty(max(1, splitk * max(n, D) * N), device="cuda")

    def fused(a_kc, b_kc, M, N_, K, A, lda, B, ldb, C, ldc, f, bias=None, acc=0):
        call("seqrec_gemm_f32_fused", a_kc, b_kc, M, N_, K, ptr(A), lda, ptr(B), ldb, ptr(C), ldc, ptr(bias), acc, splitk, ptr(ws),
             ctypes.addressof(f), st())
    # (1) row gather, a_kcontig = 1
    bias = dev(rng.normal(size=N).astype(np.float32))
    C1 = torch.full((n, N), float("nan"), device="cuda")
    fused(1, 0, n, N, D, Ed, D, Wd, N, C1, N, L.gemm_fuse(a_index=idd), bias=bias)
    C0 = torch.full((n, N), float("nan"), device="cuda")
    gemm(1, 0, n, N, D, dev(X), D, Wd, N, C0, N, bias=bias, splitk=splitk)
    assert torch.equal(C1, C0)                                                  # same tiles, same order: bit-identical
    ref = X.astype(np.float64) @ W.astype(np.float64) + bias.cpu().numpy()
    assert np.abs(C1.cpu().numpy() - ref).max() < 2e-4
    # (2) gather along K, a_kcontig = 0: C[D, N] = X^T . G with X = E[ids] never materialised
    G = rng.normal(size=(n, N)).astype(np.float32)
    C2 = torch.full((D, N), float("nan"), device="cuda")
    fused(0, 0, D, N, n, Ed, D, dev(G), N, C2, N, L.gemm_fuse(a_index=idd))
    ref2 = X.astype(np.float64).T @ G.astype(np.float64)
    assert np.abs(C2.cpu().numpy() - ref2).max() < 5e-4
    # grouped form with one gathered and one plain problem
    C3 = torch.full((D, N), float("nan"), device="cuda")
    C4 = torch.full((D, N), float("nan"), device="cuda")
    descs = L.gemm_descs([(D, N, n, Ed, D, dev(G), N, C3, N, idd), (D, N, n, dev(X.T.copy().T), D, dev(G), N, C4, N)])
    call("seqrec_gemm_f32_grouped", 2, 0, 0, descs, splitk, ptr(ws.new_empty(2 * splitk * D * N)) if splitk > 1 else None, st())
    assert np.abs(C3.cpu().numpy() - ref2).max() < 5e-4 and np.abs(C4.cpu().numpy() - ref2).max() < 5e-4
    # (3) row add: C[m,:] = A.B + scale[m] * T[idx[m],:]
    T = rng.normal(size=(V, N + 8)).astype(np.float32)                            # add_ld > N
    tix = rng.integers(0, V, size=n).astype(np.int32)
    tix[5] = -1
    sc = rng.normal(size=n).astype(np.float32)
    C5 = torch.full((n, N), float("nan"), device="cuda")
    fused(1, 0, n, N, D, dev(X), D, Wd, N, C5, N, L.gemm_fuse(add_table=dev(T), add_index=dev(tix), add_scale=dev(sc), add_ld=N + 8))
    add = np.where(tix[:, None] >= 0, T[np.maximum(tix, 0), :N] * sc[:, None], 0)
    assert np.abs(C5.cpu().numpy() - (X.astype(np.float64) @ W.astype(np.float64) + add)).max() < 2e-4
    # bad: add table narrower than N
    with pytest.raises(L.SeqrecError):
        fused(1, 0, n, N, D, dev(X), D, Wd, N, C5, N, L.gemm_fuse(add_table=dev(T), add_index=dev(tix), add_ld=N - 1))
