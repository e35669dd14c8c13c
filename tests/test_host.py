"""CPU-side tests: the C-ABI library loads and exports every declared symbol, host batching,
the product's RNG/alias host code against the oracle's specification."""
import importlib
import os
import re

import numpy as np
import pytest

from helpers import make_sessions, pad_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_in_the_header():
    L = importlib.import_module("seq-recommendations_amd._lib")
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "seqrec_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(seqrec_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(L.EXPORTS) == declared
    assert lib.seqrec_abi_version() == L.ABI_VERSION == 5
    assert lib.seqrec_build_arch() == b"gfx950"
    # argument validation happens on the host, before any launch: callable without a GPU
    assert lib.seqrec_gather_rows(None, None, None, -1, 8, None, None, 0, None) == -1
    assert lib.seqrec_rnn_upack_floats(2, 256) == 2 * 3 * 256 * 256 + 256 * 256      # + the wide BPTT form of U_h^T


def test_engine_refuses_to_run_without_gpu_or_library():
    import torch
    E = importlib.import_module("seq-recommendations_amd.engine")
    if not torch.cuda.is_available():
        with pytest.raises(Exception):
            E.Engine(E.NetConfig())


def test_pack_sessions_layout_and_roundtrip():
    B = importlib.import_module("seq-recommendations_amd.batching")
    rng = np.random.default_rng(0)
    sess = make_sessions(rng, 50, 30, 1, 14) + [[], [5]]
    rb = B.pack_sessions(sess)
    L = np.array([max(len(s) - 1, 0) for s in sess])
    assert rb.n_sessions == 52 and rb.B == int((L > 0).sum()) and rb.T == L.max() and rb.n_tok == L.sum()
    assert np.all(np.diff(rb.lengths) <= 0)                      # sorted descending
    bt = np.diff(rb.step_off)
    assert np.all(np.diff(bt) <= 0) and bt[0] == rb.B
    for p in range(rb.n_tok):
        b, s = rb.tok_b[p], rb.tok_s[p]
        assert rb.ids[p] == sess[b][s] and rb.tgt[p] == sess[b][s + 1]
        if s == 0:
            assert rb.prev[p] == -1
        else:
            q = rb.prev[p]
            assert rb.tok_b[q] == b and rb.tok_s[q] == s - 1
    # padded view of the same sessions packs identically
    keep = [s for s in sess if len(s) > 1]
    pb = pad_batch(keep, T=20)
    rb2, tcol = B.pack_padded(pb["mask"], pb["ids"], pb["tgt"])
    rb3 = B.pack_sessions(keep)
    np.testing.assert_array_equal(rb2.ids, rb3.ids); np.testing.assert_array_equal(rb2.tgt, rb3.tgt)
    np.testing.assert_array_equal(rb2.step_off, rb3.step_off); np.testing.assert_array_equal(rb2.prev, rb3.prev)
    # flat storage
    starts = np.zeros(len(sess) + 1, np.int64); starts[1:] = np.cumsum([len(s) for s in sess])
    flat = np.array([v for s in sess for v in s], np.int64)
    sel = rng.permutation(len(sess))[:20]
    rb4 = B.pack_flat(flat, starts, sel)
    rb5 = B.pack_sessions([sess[i] for i in sel])
    np.testing.assert_array_equal(rb4.ids, rb5.ids); np.testing.assert_array_equal(rb4.step_off, rb5.step_off)


def test_pack_edge_cases():
    B = importlib.import_module("seq-recommendations_amd.batching")
    rb = B.pack_sessions([])
    assert rb.n_tok == 0 and rb.T == 0 and rb.B == 0 and list(rb.step_off) == [0]
    rb = B.pack_sessions([[3], [], [4]])
    assert rb.n_tok == 0 and rb.n_sessions == 3
    rb = B.pack_sessions([list(range(50))])
    assert rb.T == 49 and rb.B == 1 and list(rb.prev[:3]) == [-1, 0, 1]
    x = np.zeros((2, 4, 5)); x[0, 2, 1] = 1; x[0, 3, 4] = 1; x[1, 3, 0] = 1
    m, ids, exact = B.onehot_to_ids(x)
    assert exact and m.tolist() == [[False, False, True, True], [False, False, False, True]] and ids[0, 3] == 4
    x[1, 3, 2] = 0.5
    assert not B.onehot_to_ids(x)[2]


def test_msnbc_text_format_and_history_features(tmp_path):
    """datasets.load_msnbc_data (datasets.py:199-227: 8 header lines, whitespace-separated tokens, ids in
    order of first appearance, eliminate_repeats) and build_xs (datasets.py:97-113) against literal
    restatements of the reference loops."""
    import importlib
    DS = importlib.import_module("seq-recommendations_amd.datasets")
    rng = np.random.default_rng(5)
    lines = ["%% header %d" % i for i in range(8)]
    raw = []
    for _ in range(40):
        toks = [str(int(v)) for v in rng.integers(1, 18, size=int(rng.integers(0, 12)))]
        if len(toks) > 3:
            toks[2] = toks[1]                      # a repeat to eliminate
        raw.append(toks)
        lines.append(" ".join(toks) + " ")
    f = tmp_path / "msnbc.txt"
    f.write_text("\n".join(lines) + "\n")
    for elim in (False, True):
        seqs, vocab = DS.load_msnbc_data(eliminate_repeats=elim, path=str(f))
        ref_vocab, ref = {}, []
        for toks in raw:                           # the reference's loop, restated
            seq, prev = [], None
            for t in toks:
                if t not in ref_vocab:
                    ref_vocab[t] = len(ref_vocab)
                if elim and t != prev:
                    seq.append(ref_vocab[t]); prev = t
                elif not elim:
                    seq.append(ref_vocab[t])
            ref.append(seq)
        assert seqs == ref and vocab == ref_vocab and len(seqs) == 40
    seqs, vocab = DS.load_msnbc_data(path=str(f))
    for freq in (False, True):
        xs = DS.build_xs(seqs, vocab, freq=freq)
        for seq, x in zip(seqs, xs):
            cur = [0] * len(vocab)
            assert len(x) == len(seq)
            for t, v in enumerate(seq):
                cur[v] = cur[v] + 1 if freq else 1
                assert x[t] == cur
    flat, starts = DS.to_flat(seqs)
    assert flat.dtype == np.int32 and starts[-1] == len(flat) == sum(len(s) for s in seqs)
    B = importlib.import_module("seq-recommendations_amd.batching")
    sel = np.arange(len(seqs))
    a, b = B.pack_flat(flat, starts, sel), B.index_flat(starts, sel)
    for k in ("order", "lengths", "step_off", "prev", "tok_b", "tok_s", "tok_row"):
        np.testing.assert_array_equal(getattr(a, k), getattr(b, k))
    assert b.ids is None and (a.B, a.T, a.n_tok, a.n_sessions) == (b.B, b.T, b.n_tok, b.n_sessions)


def test_c_abi_rejects_bad_arguments_before_any_launch():
    """Error behaviour of the boundary: negative status (SEQREC_E_ARG = -1, SEQREC_E_SHAPE = -2) for null
    pointers / inconsistent sizes, detected on the host -- so this runs without a GPU."""
    L = importlib.import_module("seq-recommendations_amd._lib")
    lib = L.load()
    E_ARG, E_SHAPE = -1, -2
    one = 0x1000                       # a non-null address that is never dereferenced on the host
    assert lib.seqrec_gather_rows(None, None, None, 5, 8, None, None, 0, None) == E_ARG
    assert lib.seqrec_gemm_f32(1, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, 1, None, None) == E_ARG
    assert lib.seqrec_gemm_f32(1, 0, 4, 4, 4, one, 4, one, 4, one, 4, None, 0, 2, None, None) == E_ARG          # split-K without workspace
    assert lib.seqrec_gemm_f32_grouped(0, 0, 0, None, 1, None, None) == E_ARG
    assert lib.seqrec_rnn_fwd(0, 0, 100, 100, 3, 4, None, None, None, None, None, None, None) == E_SHAPE       # H not in {64..512}
    assert lib.seqrec_rnn_fwd_stepwise(2, 0, 256, 300, 3, 4, one, None, one, one, one, one, one, None, 0, None) == E_SHAPE   # H_real > H
    assert lib.seqrec_rnn_fwd_stepwise(2, 0, 256, 256, 3, 4, None, None, None, None, None, None, None, None, 0, None) == E_ARG
    assert lib.seqrec_rnn_pack_u_stepwise(1, 100, one, one, None) == E_SHAPE
    assert lib.seqrec_full_softmax_ce(one, 4, one, 3, 8, 1.0, one, None, None) == E_ARG                          # ld < V
    assert lib.seqrec_sampled_softmax_ce(None, 8, one, 64, one, None, None, None, one, one, 3, 8, 1.0, one, one, None) == E_ARG
    assert lib.seqrec_rows_scatter_add(one, one, one, one, 8, None, -1, 8, 0, None) == E_ARG
    assert lib.seqrec_rows_adagrad(None, one, one, one, one, 4, 8, 0, 0.01, 1e-8, one, None) == E_ARG
    assert lib.seqrec_opt_sqnorm(0, None, None, None, 0, one, None, 0, None, None) == E_ARG                                      # nothing to do is an error
    assert lib.seqrec_opt_sqnorm(9, one, one, None, 0, one, None, 0, None, None) == E_ARG                                        # > 8 dense tensors
    assert lib.seqrec_opt_apply(1, None, None, one, one, None, 0, one, 1.0, 0.01, 1e-8, one, None, None, None, None, None) == E_ARG
    assert lib.seqrec_gather_rows_bounded(one, 0, one, one, 4, 8, None, None, 0, None, None) == E_ARG                       # a bound of 0 rows
    assert lib.seqrec_sample_negatives(1, 0, 4, None, None, 10, one, None) == E_ARG
    assert lib.seqrec_sample_gather(1, 0, 4, one, one, 10, one, 0, None, one, one, None, None) == E_ARG           # width 0
    assert lib.seqrec_dropout_mask(1, 2, one, 4, 8, 4, 0.5, one, None) == E_ARG                                   # ld < width
    assert lib.seqrec_dropout_mask(1, 2, one, 4, 8, 8, 1.0, one, None) == E_ARG                                   # rate == 1
    assert lib.seqrec_pack_batch(None, one, one, one, 4, 3, one, one, one, None) == E_ARG
    assert lib.seqrec_pack_batch_host(one, one, one, one, 957, 3, one, one, one, one, one, None) == E_SHAPE      # B + T + 1 > 960
    assert lib.seqrec_pack_batch_host(one, one, None, one, 4, 3, one, one, one, one, one, None) == E_ARG
    assert lib.seqrec_history_features(one, one, one, one, 4, 3, 8, 4, 0, one, None) == E_ARG                     # ld < x_dim
    assert lib.seqrec_topk_finish(one, one, 4, 65, one, one, None) == E_ARG                                       # k > 64
    assert lib.seqrec_topk_merge(one, 4, 4, 8, 0, None, one, one, None) == E_ARG                                  # ld < width
    assert lib.seqrec_rank_count(None, 64, one, None, one, 4, 10, one, one, None) == E_ARG
    assert lib.seqrec_prior_grad(one, None, 4, 0.5, None, None, None) == E_ARG                                    # neither grad nor loss
    assert lib.seqrec_index_affine_i32(None, None, one, None, 4, 1, 0, None) == E_ARG
    # empty inputs are NOT errors
    assert lib.seqrec_gather_rows(None, None, None, 0, 8, None, None, 0, None) == 0
    assert lib.seqrec_rows_scatter_add(None, None, None, None, 8, None, 0, 8, 0, None) == 0
    assert lib.seqrec_rnn_fwd_stepwise(2, 0, 256, 256, 0, 0, None, None, None, None, None, None, None, None, 0, None) == 0


def test_packing_properties_hypothesis():
    """Property-based: for ANY ragged set of sessions (empty ones, single-item ones, up to 50 items) the three
    packers agree and the layout invariants hold -- every transition appears exactly once, time-major in
    length-sorted order, prev links walk each session backwards."""
    from hypothesis import given, settings, strategies as hst
    B = importlib.import_module("seq-recommendations_amd.batching")

    @settings(max_examples=120, deadline=None)
    @given(hst.lists(hst.lists(hst.integers(0, 40), min_size=0, max_size=50), min_size=0, max_size=40))
    def check(sessions):
        rb = B.pack_sessions(sessions)
        lens = [max(len(s) - 1, 0) for s in sessions]
        assert rb.n_tok == sum(lens) and rb.T == (max(lens) if lens else 0) and rb.B == sum(1 for l in lens if l > 0)
        assert rb.n_sessions == len(sessions)
        assert list(rb.lengths) == sorted((l for l in lens if l > 0), reverse=True)
        assert rb.step_off[0] == 0 and rb.step_off[-1] == rb.n_tok and np.all(np.diff(rb.step_off) >= 0)
        assert np.all(np.diff(np.diff(rb.step_off)) <= 0) if rb.T > 1 else True        # B_t is non-increasing
        seen = set()
        for p in range(rb.n_tok):
            b, t = int(rb.tok_b[p]), int(rb.tok_s[p])
            assert rb.ids[p] == sessions[b][t] and rb.tgt[p] == sessions[b][t + 1]
            assert p == rb.step_off[t] + rb.tok_row[p]
            assert rb.prev[p] == (-1 if t == 0 else rb.step_off[t - 1] + rb.tok_row[p])
            seen.add((b, t))
        assert len(seen) == rb.n_tok
        if sessions:
            starts = np.zeros(len(sessions) + 1, np.int64)
            np.cumsum([len(s) for s in sessions], out=starts[1:])
            flat = np.array([v for s in sessions for v in s], dtype=np.int64)
            rf = B.pack_flat(flat, starts, np.arange(len(sessions)))
            ri = B.index_flat(starts, np.arange(len(sessions)))
            for k in ("order", "lengths", "step_off", "prev", "tok_b", "tok_s", "tok_row"):
                np.testing.assert_array_equal(getattr(rf, k), getattr(rb, k))
                np.testing.assert_array_equal(getattr(ri, k), getattr(rb, k))
            np.testing.assert_array_equal(rf.ids, rb.ids)
            np.testing.assert_array_equal(rf.tgt, rb.tgt)

    check()


def test_device_status_word_decodes_to_a_raise_message():
    """Engine.check_status() raises SeqrecError with these texts when a kernel set a SEQREC_STATUS_* bit or a cluster scan
    counted a wait that ran out (the GPU side: tests/test_gpu_engine.py::test_device_side_failure_raises_instead_of_training_on)."""
    E = importlib.import_module("seq-recommendations_amd.engine")
    assert E.status_messages(0, 0) == []
    m = E.status_messages(1 | 8, 0)
    assert len(m) == 2 and "gradient norm" in m[0] and "row index" in m[1]
    assert "clip scale" in E.status_messages(4, 0)[0] and "divisor" in E.status_messages(2, 0)[0]
    m = E.status_messages(0, 3)
    assert len(m) == 1 and "3 in-kernel wait" in m[0] and "NaN-poisoned" in m[0]
    assert "unknown status bits" in E.status_messages(64, 0)[0]


def test_window_planner_begins_half_a_window_ahead_and_never_skips_a_batch():
    """distributed.WindowPlanner (host logic, no GPU): batches come back in step order, every window is begun (count exchange
    queued) half a window before it is needed and ended a quarter window before, a finite stream ends cleanly, and every rank
    would issue the same begin / end sequence (it depends on the step index alone)."""
    import importlib
    D = importlib.import_module("seq-recommendations_amd.distributed")

    class FakeEngine:
        def __init__(self):
            self.log = []

        def prepare_begin(self, rbs):
            self.log.append(("begin", rbs[0], rbs[-1], self.now))
            return list(rbs)

        def prepare_end(self, h):
            self.log.append(("end", h[0], h[-1], self.now))
            return [("batch", x) for x in h]

    eng = FakeEngine()
    n_total = 150
    pl = D.WindowPlanner(eng, lambda j: j if j < n_total else None, window=32)
    got = []
    for i in range(n_total):
        eng.now = i
        got.append(pl.get(i))
    assert got == [("batch", i) for i in range(n_total)]
    begins = [(a, b, at) for k, a, b, at in eng.log if k == "begin"]
    ends = [(a, b, at) for k, a, b, at in eng.log if k == "end"]
    assert [b[:2] for b in begins] == [(0, 31), (32, 63), (64, 95), (96, 127), (128, 149)] == [e[:2] for e in ends]
    assert begins[0][2] == 0 and ends[0][2] == 0                       # the first window: planned on the spot
    for (lo, _, at_b), (_, _, at_e) in zip(begins[1:], ends[1:]):
        assert at_b == lo - 16 and at_e == lo - 8, (lo, at_b, at_e)    # half a window ahead / a quarter ahead
    assert pl.windows == 5 and pl.done
