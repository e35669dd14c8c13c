"""Ragged session batches: the build's replacement for the reference's dense tensors.

The reference turns every session ``s`` into the pairs ``x_i = s[i], y_i = s[i+1]``
(``preprocessor.py:75-78``), one-hot encodes both, PRE-pads to the longest
session (``preprocessor.py:16-20``) and lets Keras' ``Masking(0.0)`` skip the pad
rows (``model.py:246,335-336``).  Under pre-padding that is the same as running
every session from a zero state over its real steps only,
which is what the HIP scan computes from this layout:

  * sessions sorted by number of transitions L_b, descending (stable);
  * time-major packed tokens: ``p = step_off[t] + b`` for ``b < B_t = #{L_b > t}``;
  * ``prev[p]`` = token of the same session at step t-1 (-1 at t = 0).

Pure numpy host code; nothing here touches the GPU.
"""
import numpy as np


class RaggedBatch:
    """Host-side packed batch.

    order[b]   original index (within the batch) of sorted session b
    lengths[b] transitions of sorted session b
    step_off   int32[T+1]
    ids, tgt   int32[N_tok]   (ids may be None when dense features are used)
    x          float32[N_tok, F] dense features or None
    xs         float32[N_tok, Fx] history features for the x_to_y branch or None
    prev       int32[N_tok]
    tok_b      int32[N_tok]   original batch index of the token's session
    tok_s      int32[N_tok]   step index t of the token
    tok_row    int32[N_tok]   sorted session row b of the token (p = step_off[t] + b)
    n_sessions number of sessions the caller handed over (incl. empty ones) -- Keras
               weights epoch losses by this batch size.
    """

    __slots__ = ("order", "lengths", "step_off", "ids", "tgt", "x", "prev", "tok_b", "tok_s", "tok_row", "xs",
                 "n_sessions", "B", "T", "n_tok")

    def ensure_tokens(self):
        """Per-token coordinates (prev / tok_b / tok_s / tok_row) of a batch packed by
        index_flat(..., lean=True), which skips them: only dropout and host-side tests need them."""
        if self.tok_s is None:
            so = self.step_off.astype(np.int64)
            bt = np.diff(so)
            tok_t = np.repeat(np.arange(self.T, dtype=np.int64), bt)
            tok_bs = np.arange(self.n_tok, dtype=np.int64) - so[tok_t]
            _finish(self, self.order.astype(np.int64), self.lengths, so, tok_t, tok_bs, self.B, self.T, self.n_tok,
                    self.n_sessions)
        return self


def _pack_index(lengths):
    """lengths (any order) -> order, sorted lengths, step_off, (tok_t, tok_bsorted)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    ls = lengths[order]
    B = int(np.count_nonzero(ls > 0))
    order, ls = order[:B], ls[:B]
    T = int(ls[0]) if B else 0
    # B_t = #{L > t}: ls is descending, so it is the first index with ls <= t
    bt = np.searchsorted(-ls, -np.arange(T, dtype=np.int64), side="left").astype(np.int64)
    step_off = np.zeros(T + 1, dtype=np.int64)
    np.cumsum(bt, out=step_off[1:])
    n_tok = int(step_off[-1])
    tok_t = np.repeat(np.arange(T, dtype=np.int64), bt)
    tok_bs = np.arange(n_tok, dtype=np.int64) - step_off[tok_t]
    return order, ls, step_off, tok_t, tok_bs, B, T, n_tok


def _finish(rb, order, ls, step_off, tok_t, tok_bs, B, T, n_tok, n_sessions):
    rb.order = order.astype(np.int32)
    rb.lengths = ls.astype(np.int32)
    rb.step_off = step_off.astype(np.int32)
    prev = np.full(n_tok, -1, dtype=np.int64)
    later = tok_t > 0
    prev[later] = step_off[tok_t[later] - 1] + tok_bs[later]
    rb.prev = prev.astype(np.int32)
    rb.tok_b = order[tok_bs].astype(np.int32)
    rb.tok_s = tok_t.astype(np.int32)
    rb.tok_row = tok_bs.astype(np.int32)
    rb.B, rb.T, rb.n_tok, rb.n_sessions = B, T, n_tok, n_sessions
    return rb


def pack_sessions(sessions):
    """sessions: list of item-id lists.  A session of n items yields n-1 (input, target) pairs."""
    n_sessions = len(sessions)
    lengths = np.fromiter((max(len(s) - 1, 0) for s in sessions), dtype=np.int64, count=n_sessions)
    order, ls, step_off, tok_t, tok_bs, B, T, n_tok = _pack_index(lengths)
    starts = np.zeros(n_sessions + 1, dtype=np.int64)
    np.cumsum(np.fromiter((len(s) for s in sessions), dtype=np.int64, count=n_sessions), out=starts[1:])
    flat = np.fromiter((v for s in sessions for v in s), dtype=np.int64, count=int(starts[-1]))
    src = starts[order[tok_bs]] + tok_t
    rb = RaggedBatch()
    rb.ids = flat[src].astype(np.int32)
    rb.tgt = flat[src + 1].astype(np.int32)
    rb.x = None
    rb.xs = None
    return _finish(rb, order, ls, step_off, tok_t, tok_bs, B, T, n_tok, n_sessions)


def pack_flat(flat, starts, sel):
    """Same as pack_sessions for sessions stored as one flat id array: session i is
    flat[starts[i]:starts[i+1]]; ``sel`` picks the batch's sessions."""
    sel = np.asarray(sel, dtype=np.int64)
    lens = starts[sel + 1] - starts[sel]
    lengths = np.maximum(lens - 1, 0)
    order, ls, step_off, tok_t, tok_bs, B, T, n_tok = _pack_index(lengths)
    src = starts[sel[order[tok_bs]]] + tok_t
    rb = RaggedBatch()
    rb.ids = flat[src].astype(np.int32)
    rb.tgt = flat[src + 1].astype(np.int32)
    rb.x = None
    rb.xs = None
    return _finish(rb, order, ls, step_off, tok_t, tok_bs, B, T, n_tok, len(sel))


def index_flat(starts, sel, lean=False):
    """Index-only packing for the device-side batcher (engine.Engine.upload_device): everything the
    host needs (order, lengths, step offsets, token coordinates) from the session LENGTHS alone; the
    item ids are gathered on the GPU by seqrec_pack_batch, so rb.ids / rb.tgt stay None.
    lean=True computes only what a training step needs on the host (order, lengths, step offsets:
    a sort of the batch's lengths); RaggedBatch.ensure_tokens() adds the per-token arrays on demand."""
    sel = np.asarray(sel, dtype=np.int64)
    lengths = np.maximum(starts[sel + 1] - starts[sel] - 1, 0)
    rb = RaggedBatch()
    rb.ids = rb.tgt = rb.x = rb.xs = None
    if lean:
        order = np.argsort(-lengths, kind="stable")
        ls = lengths[order]
        B = int(np.count_nonzero(ls > 0))
        T = int(ls[0]) if B else 0
        step_off = np.zeros(T + 1, dtype=np.int32)
        # B_t = #{L > t} for the descending ls
        np.cumsum(np.searchsorted(-ls[:B], -np.arange(T, dtype=np.int64), side="left"), out=step_off[1:])
        rb.order, rb.lengths, rb.step_off = order[:B].astype(np.int32), ls[:B].astype(np.int32), step_off
        rb.prev = rb.tok_b = rb.tok_s = rb.tok_row = None
        rb.B, rb.T, rb.n_tok, rb.n_sessions = B, T, int(step_off[-1]), len(sel)
        return rb
    order, ls, step_off, tok_t, tok_bs, B, T, n_tok = _pack_index(lengths)
    return _finish(rb, order, ls, step_off, tok_t, tok_bs, B, T, n_tok, len(sel))


def pack_padded(mask, ids=None, tgt=None, x=None, xs=None):
    """From the reference's padded view: mask (B,T) bool marks real steps (any position --
    masked steps simply carry state, so only the order of the real steps matters);
    ids/tgt (B,T) ints and/or x (B,T,F) float features."""
    mask = np.asarray(mask, dtype=bool)
    n_sessions, Tp = mask.shape
    lengths = mask.sum(axis=1)
    order, ls, step_off, tok_t, tok_bs, B, T, n_tok = _pack_index(lengths)
    # column (padded time index) of the s-th real step of each row
    cols = np.argsort(~mask, axis=1, kind="stable")       # real steps first, in time order
    rows = order[tok_bs]
    tcol = cols[rows, tok_t]
    rb = RaggedBatch()
    rb.ids = None if ids is None else np.asarray(ids)[rows, tcol].astype(np.int32)
    rb.tgt = None if tgt is None else np.asarray(tgt)[rows, tcol].astype(np.int32)
    rb.x = None if x is None else np.ascontiguousarray(np.asarray(x)[rows, tcol], dtype=np.float32)
    rb.xs = None if xs is None else np.ascontiguousarray(np.asarray(xs)[rows, tcol], dtype=np.float32)
    rb = _finish(rb, order, ls, step_off, tok_t, tok_bs, B, T, n_tok, n_sessions)
    return rb, tcol.astype(np.int64)


def onehot_to_ids(x):
    """(N,T,V) one-hot (pad rows all zero, preprocessor.py:77-78,88) -> mask, ids, exact?
    ``exact`` says whether every real row is a clean one-hot (single entry equal to 1)."""
    x = np.asarray(x)
    nz = x != 0
    mask = nz.any(axis=2)
    ids = np.argmax(x, axis=2)
    cnt = nz.sum(axis=2)
    peak = np.take_along_axis(x, ids[:, :, None], axis=2)[:, :, 0]
    exact = bool(np.all(cnt[mask] == 1) and np.all(peak[mask] == 1))
    return mask, ids.astype(np.int64), exact
