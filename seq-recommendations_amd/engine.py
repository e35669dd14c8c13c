"""Device engine: parameters in HBM + the kernel sequence of one training / evaluation step.

Counterpart of what Keras' ``Model.train_function`` / ``test_function`` /
``predict_function`` do for the graphs built in ``model.py:241-258`` and
``model.py:322-403``: every arithmetic op is a call into
libseqrec_hip.so (``include/seqrec_hip.h``); torch is used for device memory,
streams and (in ``distributed.py``) collectives only.

HBM layout
  * hidden size is zero-padded to Hp in {64,128,256,512}; gate g of a G-gate kernel
    occupies columns [g*Hp, g*Hp+H).  Padded units stay exactly zero (zero
    weights -> zero pre-activations -> zero state and zero gradients).
  * item tables (E, Eout, one-hot input kernel Wk, output bias) are row-major with
    one row per item; each has an Adagrad accumulator AND a gradient table of the
    same shape that is all-zero between steps, plus an int32 owner slot per row --
    the row-sparse update touches only the rows of the batch (exactly equivalent to
    the reference's dense Adagrad, experiments_methods.py:41).  Sized for 288 GB HBM:
    3x table bytes per table (c3: 2 tables x 3 x 0.95 GiB).
  * activations are token-major [N_tok, width] in the time-major packed order of
    ``batching.RaggedBatch``.
"""
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import CELL, ACT, N_GATES, ptr
from ._lib import call as _raw_call

INT32_MAX = 2 ** 31 - 1
import os as _os
# split-K is raised until about this many workgroups are in flight (4 per CU; measured on the c3
# backward shapes: 512 -> 1024 takes dEneg from 53 to 45 us and dH from 46 to 43 us)
SPLITK_TARGET_WGS = int(_os.environ.get("SEQREC_SPLITK_WGS", "512"))
SPLITK_MIN_K = int(_os.environ.get("SEQREC_SPLITK_MIN_K", "512"))
SPLITK_FILL_WGS = int(_os.environ.get("SEQREC_SPLITK_FILL", "1152"))
SPLITK_FILL_WGRAD = int(_os.environ.get("SEQREC_SPLITK_FILL_WGRAD", "1700"))
SPLITK_LONGK_WGS = int(_os.environ.get("SEQREC_SPLITK_LONGK", "1280"))

# Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg):
# _PROF = {"events": [(name, tag, start_event, end_event), ...]} while enabled, else None.
_PROF = None


def profile_start():
    global _PROF
    _PROF = {"events": []}


def profile_stop():
    """-> {name: (launches, total_ms)} ; synchronises."""
    global _PROF
    ev, _PROF = _PROF["events"], None
    torch.cuda.synchronize()
    out = {}
    for name, tag, a, b in ev:
        key = name if not tag else "%s[%s]" % (name, tag)
        n, ms = out.get(key, (0, 0.0))
        out[key] = (n + 1, ms + a.elapsed_time(b))
    return out


def call(name, *args, tag=None, prof_name=None):
    """prof_name: the per-call profile books this call under another entry point's name (a variant of the same step)."""
    if _PROF is None:
        return _raw_call(name, *args)
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    _raw_call(name, *args)
    b.record()
    _PROF["events"].append((prof_name or name, tag, a, b))


def _pad_h(H):
    for c in (64, 128, 256, 512):
        if H <= c:
            return c
    raise ValueError("hidden size %d > 512 is not supported by the gfx950 scan kernels" % H)


def _ceil4(n):
    return (n + 3) // 4 * 4


@dataclass
class NetConfig:
    cell: str = "lstm"            # 'simplernn' | 'lstm' | 'gru'
    act: str = "relu"             # z_to_z_activation (model.py:243,324)
    H: int = 64                   # z_dim
    V_in: int = 17                # input width: vocabulary (ids) or feature count (dense)
    V_out: int = 17               # n_classes
    input: str = "onehot"         # 'onehot': XW = Wk[id] + b   'embed': (E[id]).W + b   'dense': x.Wk + b
    D: int = 0                    # embedding width for input == 'embed'
    output: str = "full"          # 'full' softmax | 'sampled' softmax
    K: int = 0                    # shared negatives per step (sampled)
    tied: bool = False            # Eout is E (needs D == H)
    use_bias: bool = True         # z_bias
    out_bias: bool = False        # toy_bias / Dense bias of RNNBaseline
    drop_in: float = 0.0          # y_to_z_dropout
    drop_rec: float = 0.0         # z_to_z_dropout (recurrent)
    drop_out: float = 0.0         # z_to_y_dropout
    logq: bool = False            # subtract log Q(item) from sampled logits
    seed: int = 0                 # counter-RNG seed (negatives, dropout)
    scan: str = "auto"            # 'persistent' (rnn.hip) | 'stepwise' (rnn_step.hip) | 'auto'
    # RNNFullModel side branches (model.py:375-392; full softmax only, small vocabularies)
    y_to_y: bool = False          # logits += Wyy[id_t] (+ byy): direct term on the one-hot input
    yy_bias: bool = False
    x_to_y: bool = False          # logits += xs_t . Wxy   (history features, x_dim wide)
    x_dim: int = 0
    diag_b: bool = True           # OnlyNonZeroDiagonal on Wxy (model.py:48-66): re-applied after every update
    merge: str = "atomic"         # row-sparse gradient merge: 'atomic' (float atomics, sums reproducible to rounding) |
                                  # 'sorted' (stable sort by row + ordered segment sum, csrc/merge.hip: bitwise reproducible)
    priors: tuple = ()            # parameters that carry a kernel regularizer (Engine.set_prior); a table listed
                                  # here is updated densely (the penalty's gradient touches every row)

    @property
    def G(self):
        return N_GATES[self.cell]


def status_messages(bits, scan_errors):
    """What a device status word (SEQREC_STATUS_* bits) and a cluster-scan error count say, as text; [] = healthy."""
    msgs = [txt for b, txt in sorted(_lib.STATUS_BITS.items()) if bits & b]
    if bits & ~sum(_lib.STATUS_BITS):
        msgs.append("unknown status bits 0x%x" % (bits & ~sum(_lib.STATUS_BITS)))
    if scan_errors:
        msgs.append("%s in-kernel wait(s) of a cluster scan ran out (workgroups of a row block not co-resident, e.g. another "
                    "process holding the GPU): the scan's outputs are NaN-poisoned" % ("some" if scan_errors < 0 else scan_errors))
    return msgs


class PinnedRing:
    """Engine-owned staging memory for every host -> device upload that must not block the host (``non_blocking=True``).
    A copy from PAGEABLE memory may still be reading its source after the call returns; a numpy temporary whose last
    reference dies at the end of the caller's loop body is then free to be overwritten under the copy.  Here the bytes
    are first copied into a page-locked slot that the ring owns, the device copy is issued from the slot and an event is
    recorded behind it on the stream it was issued on; a slot is reused only after its event has completed (the host
    waits if the ring is full: 64 slots, i.e. 64 uploads in flight).  The slots are carved out of ONE page-locked arena
    (16 MB, allocated at the first upload: a hipHostMalloc per slot cost ~3 ms each inside the first 64 steps); an upload
    larger than a slot gets a page-locked buffer of its own."""
    SLOT_BYTES = 256 * 1024

    def __init__(self, device, slots=64):
        self.dev = device
        self.n = slots
        self.arena = None
        self.NBIG = 4                      # uploads larger than a slot (a window's routing blocks, ~3 MB) rotate through 4 buffers of their own
        self.big = [None] * self.NBIG
        self.big_events = [None] * self.NBIG
        self.big_pos = 0
        self.events = [None] * slots
        self.pos = 0

    def put(self, arr):
        """numpy array (any dtype) -> device tensor of the same dtype and shape, uploaded asynchronously on the current stream."""
        a = np.ascontiguousarray(arr)
        if a.size == 0:
            return torch.empty(a.shape, dtype=torch.from_numpy(np.zeros(0, a.dtype)).dtype, device=self.dev)

        def fill(dst):
            dst[:] = a.reshape(-1)
        return self.put_fill(a.size, a.dtype, fill).view(a.shape)

    def put_fill(self, count, dtype, fill):
        """The same without a staging copy on the caller's side: fill(numpy view of `count` elements of `dtype`) writes the
        upload straight into the page-locked slot (the native planner does: csrc/route.hip)."""
        tdt = torch.from_numpy(np.zeros(0, dtype)).dtype
        nbytes = int(count) * np.dtype(dtype).itemsize
        if nbytes <= self.SLOT_BYTES:
            i = self.pos
            self.pos = (i + 1) % self.n
            if self.events[i] is not None:
                self.events[i].synchronize()   # the copy issued from this slot `slots` uploads ago
            if self.arena is None:
                self.arena = torch.empty(self.n * self.SLOT_BYTES, dtype=torch.uint8).pin_memory()
            buf = self.arena[i * self.SLOT_BYTES: i * self.SLOT_BYTES + nbytes]
            evs = self.events
        else:
            i = self.big_pos
            self.big_pos = (i + 1) % self.NBIG
            if self.big_events[i] is not None:
                self.big_events[i].synchronize()
            if self.big[i] is None or self.big[i].numel() < nbytes:
                self.big[i] = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8).pin_memory()
            buf = self.big[i][:nbytes]
            evs = self.big_events
        host = buf.view(tdt)
        fill(host.numpy())
        out = host.to(self.dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        evs[i] = ev
        return out


class Engine:
    def __init__(self, cfg: NetConfig, device="cuda:0"):
        _lib.load()                                  # fail loudly without the HIP library
        if not torch.cuda.is_available():
            raise _lib.SeqrecError("no GPU visible: the seq-recommendations_amd engine needs an MI355X (no CPU fallback)")
        self.cfg = cfg
        self.dev = torch.device(device)
        torch.cuda.set_device(self.dev)
        c = cfg
        if c.cell not in CELL or c.act not in ACT:
            raise ValueError("unsupported cell/activation %r/%r" % (c.cell, c.act))
        if c.merge not in ("atomic", "sorted"):
            raise ValueError("merge must be 'atomic' or 'sorted', not %r" % (c.merge,))
        self.Hp = _pad_h(c.H)
        self.G = c.G
        self.GHp = self.G * self.Hp
        self.Dp = _ceil4(c.D) if c.input == "embed" else 0
        self.Fp = _ceil4(c.V_in) if c.input == "dense" else c.V_in
        self.Vp = _ceil4(c.V_out)
        if c.tied:
            if c.input != "embed" or c.output != "sampled" or c.D != c.H or c.V_in != c.V_out:
                raise ValueError("tied tables need input='embed', output='sampled', D == H, V_in == V_out")
            self.Dp = self.Hp
        f32 = dict(dtype=torch.float32, device=self.dev)
        z = lambda *s: torch.zeros(*s, **f32)
        P = {}
        if c.input == "embed":
            P["E"] = z(c.V_in, self.Dp)
            P["W"] = z(self.Dp, self.GHp)
        else:
            P["Wk"] = z(self.Fp, self.GHp)
        P["U"] = z(self.Hp, self.GHp)
        if c.use_bias:
            P["b"] = z(self.GHp)
        if (c.y_to_y or c.x_to_y) and c.output != "full":
            raise ValueError("the y_to_y / x_to_y branches are dense V x V terms: full softmax only")
        self.Fxp = _ceil4(c.x_dim) if c.x_to_y else 0
        if c.output == "full":
            P["Wout"] = z(self.Hp, self.Vp)
            if c.out_bias:
                P["bout"] = z(self.Vp)
            if c.y_to_y:
                P["Wyy"] = z(c.V_out, self.Vp)       # row per input item (y_dim == n_classes)
                if c.yy_bias:
                    P["byy"] = z(self.Vp)
            if c.x_to_y:
                P["Wxy"] = z(self.Fxp, self.Vp)
                if c.diag_b:
                    if c.x_dim != c.V_out:
                        raise ValueError("diag_b needs x_dim == y_dim")
                    m = torch.zeros(self.Fxp, self.Vp, **f32)
                    m[torch.arange(c.x_dim), torch.arange(c.x_dim)] = 1.0
                    self.diag_mask = m
        else:
            if not c.tied:
                P["Eout"] = z(c.V_out, self.Hp)
            if c.out_bias:
                P["bout"] = z(c.V_out)
        self.P = P
        self.table_params = set()
        if c.input == "embed":
            self.table_params.add("E")
        if c.input == "onehot":
            self.table_params.add("Wk")
        if c.y_to_y and "Wyy" not in c.priors:
            self.table_params.add("Wyy")
        for k in c.priors:
            if k not in ("Wyy", "Wout", "Wxy"):
                raise ValueError("kernel regularizers exist on the y_to_y and to_y Dense kernels only (model.py:383,389)")
        self.priors = {}          # name -> (padded means tensor or None, strength)
        self.reg_sum = z(1)
        if c.output == "sampled":
            if not c.tied:
                self.table_params.add("Eout")
            if c.out_bias:
                self.table_params.add("bout")
        self.trainable = {k: True for k in P}
        self.A = {k: torch.zeros_like(v) for k, v in P.items()}          # Adagrad accumulators
        self.Gd = {k: torch.zeros_like(v) for k, v in P.items() if k not in self.table_params}
        self.Gt = {k: torch.zeros_like(P[k]) for k in self.table_params}  # gradient tables (zero between steps)
        self.slot = {k: torch.full((P[k].shape[0],), INT32_MAX, dtype=torch.int32, device=self.dev)
                     for k in self.table_params}
        self.ws = {}
        self._merge_ws = {}
        self._views = {}
        self._cur_st = None
        self._side_stream = None
        self.last_slabs = {}
        self._fuse_prologue = _os.environ.get("SEQREC_FUSE_PROLOGUE", "1") != "0"   # A/B switch: U re-pack + negatives in one launch
        # A/B switch, OFF: dH reaches the BPTT as split-K slabs + row term (seqrec_rnn_bwd_stepwise_parts).  Measured: the reduce
        # launch (5 us) goes, but the cluster BPTT's per-step operand prefetch grows from 5 to 9 loads with a dependent index
        # hop and the call from 110 to 138 us -- 0.405 -> 0.427 ms per step
        self._slab_dh = _os.environ.get("SEQREC_SLAB_DH", "0") != "0"
        self._slab_wgrad = _os.environ.get("SEQREC_SLAB_WGRAD", "1") != "0"        # A/B switch: the weight gradients' split-K reduce rides in the norm launch
        self._slab_scatter = _os.environ.get("SEQREC_SLAB_SCATTER", "1") != "0"   # A/B switch: dX / dEneg reach the scatter as split-K slabs
        self._slab_min_k = int(_os.environ.get("SEQREC_SLAB_MIN_K", "256"))
        self._overlap = _os.environ.get("SEQREC_OVERLAP", "0") != "0"      # A/B switch: dEneg GEMM on a side stream under the BPTT (measured +-0.5 %: off)
        # dEneg = dlogits^T . H reduces over the tokens like dW / dU and has the same operand layout: it rides in THEIR grouped
        # launch (one launch of ~950 workgroups instead of two of ~500 that fill 1.5-2 slots per CU each; A/B switch)
        self._group_deneg = _os.environ.get("SEQREC_GROUP_DENEG", "1") != "0"
        # A/B switch, OFF: dH and dEneg (both products of dlogits) in ONE launch of two layout bodies (seqrec_gemm_f32_pair), instead of
        # dEneg riding in the weight-gradient launch.  Built, bit-identical, measured: the pair launch takes 58.4 us where dH alone takes 34
        # and dEneg 25 -- these launches are bound by the 64 x 64 tile's rate per CU, not by their ramp, so sharing a launch shares the
        # CUs' time: c3 0.4040 -> 0.4061 ms per step (profiles/r04_pair_probe.txt)
        self._pair_dh = _os.environ.get("SEQREC_PAIR_DH", "0") != "0"
        self.sq1 = z(1)                 # squared gradient norm (multi-launch path)
        self.sq2 = z(2)                 # two alternating slots of the fused optimizer launches
        self._sq_slots = (self.sq2[0:1], self.sq2[1:2])
        self._sq_par = 0
        self.sq = self.sq1
        self.scale = torch.ones(1, **f32)
        self.loss_out = z(2)            # [sum of the per-token CE, its token mean]: seqrec_loss_reduce / spare block of seqrec_opt_sqnorm
        self.loss_sum, self.loss_mean = self.loss_out[0:1], self.loss_out[1:2]
        self.upack = torch.empty(int(_lib.load().seqrec_rnn_upack_floats(CELL[c.cell], self.Hp)), **f32)
        self.upack_dirty = True
        self.stepwise = (c.scan == "stepwise") or (c.scan == "auto" and (self.Hp >= 128 or c.drop_rec > 0 or ACT[c.act] > 2))
        if ACT[c.act] > 2 and not self.stepwise:
            raise NotImplementedError("activation %r runs in the step-wise form of the scans (scan='stepwise' or 'auto')" % c.act)
        if c.drop_rec > 0 and not self.stepwise:
            raise NotImplementedError("recurrent (z_to_z) dropout needs scan='stepwise' (or 'auto')")
        import os
        # the scan's ~135 dependent launches per step ride a captured hipGraph whose nodes are rewritten with every batch's
        # exact geometry (csrc/rnn_step.hip issue_graph); SEQREC_SCAN_GRAPH=0 issues them eagerly
        self.use_graph = self.stepwise and os.environ.get("SEQREC_SCAN_GRAPH", "1") != "0"
        if os.environ.get("SEQREC_SCAN_CLUSTER") == "0":        # developer switch: the library itself reads no environment variable
            _lib.load().seqrec_debug_scan_cluster(0)
        self.sampler = None       # (thresh uint32-as-int32 tensor, alias int32 tensor, logq float tensor)
        self.step_count = 0
        # conditions only the device sees (SEQREC_STATUS_*: a gradient norm that is not finite, a clip scale of 0, an index
        # outside its table): kernels OR their bits into this word, check_status() raises on it at the next host sync
        self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.pinned = PinnedRing(self.dev)
        self.drop_seed = cfg.seed       # RNG stream of the dropout masks (distributed.ShardedEngine: one per rank)
        # the cell of the common training step in ONE C-ABI call (seqrec_train_cell); SEQREC_NATIVE_CELL=0: call by call
        self.native_cell = _os.environ.get("SEQREC_NATIVE_CELL", "1") != "0"
        self._plan, self._plan_keep, self._plan_keep_batch = None, None, None
        # batches of more tokens than this materialise E[ids] and h_{t-1} once instead of gathering them inside three GEMMs
        self.fuse_gather_max = int(_os.environ.get("SEQREC_FUSE_GATHER_MAX", "8192"))

    # ------------------------------------------------------------------ device-side failures
    def check_status(self):
        """Raise SeqrecError if a kernel reported a failure since the last check: a refused optimizer step (gradient norm /
        divisor / clip scale not usable -- the reference would carry NaNs or, at scale 0, silently train nothing), an index
        outside its table, or an in-kernel wait of a cluster scan that ran out (its outputs are NaN-poisoned).  Synchronises
        the current stream: called where the host waits anyway (end of an epoch, evaluation, parameter read-back,
        checkpoint) -- the counterpart of the reference's asserts / exceptions (model.py:136,149)."""
        bits = int(self.status.item())
        nerr = int(_lib.load().seqrec_cluster_scan_errors(self._stream()))
        msgs = status_messages(bits, nerr)
        if not msgs:
            return
        self.status.zero_()
        _lib.load().seqrec_cluster_scan_errors_reset(self._stream())
        if nerr:
            # the one-launch scan rests on its workgroups being co-resident (include/seqrec_hip.h, "Hidden state"): where that
            # failed once it may fail again -- the rest of the process scans step-wise (one launch per recurrent product, no
            # in-kernel wait), so a caller that catches this error and retries the step makes progress
            _lib.load().seqrec_debug_scan_cluster(0)
            msgs.append("the cluster form of the scans is now OFF for this process (step-wise form from here on)")
        raise _lib.SeqrecError("device-side failure at or before training step %d: %s" % (self.step_count, "; ".join(msgs)))

    # ------------------------------------------------------------------ utilities
    def _side(self):
        """Side stream (+ fork / join events) for work that is independent of the scan it runs under."""
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.dev)
            self._ev_fork = torch.cuda.Event()
            self._ev_join = torch.cuda.Event()
        return self._side_stream

    def _stream(self):
        """HIP stream handle of torch's current stream; remembered so that the helpers called later in
        the same operation (gemm) need not ask torch again."""
        self._cur_st = torch.cuda.current_stream(self.dev).cuda_stream
        return self._cur_st

    def buf(self, name, *shape, dtype=torch.float32):
        """Grow-only named workspace; the shaped views are cached (creating two tensor objects per
        buffer and step is a measurable share of the host time of a 0.7 ms step)."""
        key = (name, shape, dtype)
        v = self._views.get(key)
        if v is not None:
            return v
        n = 1
        for d in shape:
            n *= int(d)
        t = self.ws.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.dev)
            self.ws[name] = t
            for k in [k for k in self._views if k[0] == name]:      # views of the replaced storage are stale
                del self._views[k]
        v = t[:n].view(*shape) if shape else t[:1]
        if len(self._views) > 8192:
            self._views.clear()
        self._views[key] = v
        return v

    def autotune_scan(self, run_step, blocks=3, block_steps=16, warm=40, margin=0.015):
        """Pick how the scan's dependent launches are issued on THIS box.  Eager issue is paced by the host (~2.3-3 us per
        hipLaunchKernel): on a fast host core it beats the graph replay by ~3 % of a step, on a slow one it loses up to
        10 % (measured, profiles/README.md); the graph replay costs ~0.6 us of host time per launch and is GPU-paced.
        run_step() must run ONE training step on the next batch and return that batch's T.  `warm` steps per mode first
        (graphs of the common T values get built), then `blocks` alternating blocks of `block_steps` steps per mode, wall
        time per block normalised by the blocks' expected cost (a + b T).  Eager is chosen only if it is faster by
        `margin`.  Returns the decision record; sets self.use_graph."""
        import time
        if not self.stepwise:
            return {"scan_issue": "persistent"}
        for mode in (True, False):
            self.use_graph = mode
            for _ in range(warm):
                run_step()
        tot = {True: [0.0, 0.0], False: [0.0, 0.0]}
        for b in range(2 * blocks):
            mode = (b % 2 == 0)
            self.use_graph = mode
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cost = 0.0
            for _ in range(block_steps):
                cost += 0.30 + 0.0125 * float(run_step())          # expected ms of a c3-like step with T time steps
            torch.cuda.synchronize()
            tot[mode][0] += time.perf_counter() - t0
            tot[mode][1] += cost
        rate = {m: tot[m][0] / max(tot[m][1], 1e-9) for m in tot}
        self.use_graph = not (rate[False] < rate[True] * (1.0 - margin))
        return {"scan_issue": "graph" if self.use_graph else "eager", "eager_over_graph": round(rate[False] / rate[True], 4),
                "steps_per_mode": blocks * block_steps}

    def reserve(self, n_tok_max):
        """Size every grow-only workspace of a training step for batches of up to n_tok_max transitions,
        so that a loop over fresh batches (a new N_tok every step) never re-allocates in steady state."""
        c = self.cfg
        n, Hp, GHp = int(n_tok_max), self.Hp, self.GHp
        need = {"XW": n * GHp, "Hout": n * Hp, "gates": n * GHp, "aux": n * Hp, "loss_rows": n, "dHd": n * Hp,
                "dPre": n * GHp, "Hprev": n * Hp, "scan_ws": 2 * n * Hp, "thr": n,
                "colsum_ws": 64 * max(GHp, self.Vp, c.K if c.output == "sampled" else 1)}
        if c.input == "embed":
            need.update(X=n * self.Dp, dX=n * self.Dp)
        if c.output == "sampled":
            need.update(Eneg=c.K * Hp, lq_neg=c.K, ln=n * c.K, dlt=n, dEneg=c.K * Hp)
        else:
            need.update(logits=n * self.Vp, probs=n * c.V_out)
        ws = 0
        for m in range(128, n + 64, 64):
            m = min(m, n)
            use = [self._splitk(m, Hp, c.K if c.output == "sampled" else c.V_out, fill=True) * m * Hp]            # dH
            if c.output == "sampled":
                use.append(self._splitk(c.K, Hp, m) * c.K * Hp)                                          # dEneg
            else:
                use.append(self._splitk(Hp, c.V_out, m) * Hp * c.V_out)                                  # dWout
            kin = self.Dp if c.input == "embed" else (self.Fp if c.input == "dense" else 0)
            if kin:
                use.append(self._splitk(m, kin, GHp, fill=True) * m * kin)                                          # dX
            shapes = [(Hp, GHp), (kin, GHp), (1, GHp)]                                                   # grouped dU, dW, db
            tiles = sum(((a + 63) // 64) * ((b + 63) // 64) for a, b in shapes if a)
            sk = self._splitk_tiles(tiles, m, fill=SPLITK_FILL_WGRAD, long_k=True)
            use.append(sk * sum(a * b for a, b in shapes))
            ws = max(ws, max(use))
        need["gemm_ws"] = ws
        if self._slab_scatter and c.merge != "sorted":      # the slab forms of dX / dEneg keep their own buffers up to the scatter
            if c.input == "embed":
                need["dX_slabs"] = max(1, min(32, GHp // self._slab_min_k)) * n * self.Dp
            if c.output == "sampled":
                if self._slab_dh:
                    need["dH_slabs"] = max(self._splitk(m, Hp, c.K, fill=True) * m * Hp for m in range(64, n + 64, 64))
                need["dEneg_slabs"] = max(self._splitk(c.K, Hp, m) for m in range(64, n + 64, 64)) * c.K * Hp
        for name, sz in need.items():
            self.buf(name, int(max(sz, 1)))
        self.buf("neg", c.K if c.output == "sampled" else 1, dtype=torch.int32)

    ONES_LD = 4

    def _ones(self, n):
        """[n, 4] ones, used as the K x 1 operand of db = ones^T . dPre with lda = 4: 16-byte rows keep the grouped
        weight-gradient launch on the LDS-DMA GEMM kernels (columns 1-3 are rows >= M, never stored)."""
        t = self.ws.get("_ones")
        if t is None or t.numel() < n * self.ONES_LD:
            t = torch.ones(max(n, 4096) * self.ONES_LD, dtype=torch.float32, device=self.dev)
            self.ws["_ones"] = t
        return t[:n * self.ONES_LD]

    def gemm(self, a_kc, b_kc, M, N, K, A, lda, B, ldb, Cm, ldc, bias=None, accumulate=0, splitk=1, tag=None,
             ws_name="gemm_ws", fuse=None):
        """fuse: _lib.gemm_fuse(...) -- gathered A operand and/or the row add of the epilogue (seqrec_gemm_f32_fused)."""
        wsp = None
        if splitk > 1:
            wsp = self.buf(ws_name, splitk * M * N)
        st = self._cur_st if self._cur_st is not None else self._stream()
        if fuse is None:
            call("seqrec_gemm_f32", int(a_kc), int(b_kc), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(Cm), ldc, ptr(bias),
                 accumulate, splitk, ptr(wsp), st, tag=tag)
        else:
            import ctypes
            call("seqrec_gemm_f32_fused", int(a_kc), int(b_kc), M, N, K, ptr(A), lda, ptr(B), ldb, ptr(Cm), ldc, ptr(bias),
                 accumulate, splitk, ptr(wsp), ctypes.addressof(fuse), st, tag=tag)

    def gemm_slabs(self, a_kc, b_kc, M, N, K, A, lda, B, ldb, name, splitk, tag=None):
        """The product as split-K slabs left in the buffer `name` (seqrec_gemm_f32_slabs): no reduce launch, no C.
        Returns (buffer, n_slabs, slab_stride); the row scatter adds the slabs (seqrec_rows_job.n_slabs)."""
        import ctypes
        ws = self.buf(name, max(splitk, 1) * M * N)
        ns = ctypes.c_int(0)
        st = self._cur_st if self._cur_st is not None else self._stream()
        call("seqrec_gemm_f32_slabs", int(a_kc), int(b_kc), M, N, K, ptr(A), lda, ptr(B), ldb, int(max(splitk, 1)), ptr(ws),
             ctypes.addressof(ns), st, tag=tag)
        self.last_slabs[name] = (ws, int(ns.value), M, N)       # for inspection (tests rebuild the product from the slabs)
        return ws, int(ns.value), M * N

    @staticmethod
    def _splitk(M, N, K, fill=False):
        """Split K until about SPLITK_TARGET_WGS workgroups are in flight, never below SPLITK_MIN_K per slab: every split
        costs a slab round trip + a share of the reduce launch (tools/bench_gemm2.py: dH 3, dEneg 4, dX 1, dW+dU 5 at c3)."""
        return Engine._splitk_tiles(((M + 63) // 64) * ((N + 63) // 64), K, fill=fill)

    @staticmethod
    def _splitk_tiles(tiles, K, min_k=None, fill=False, long_k=False):
        """fill: products with 256 or more tiles that would not be split at all (c4: dH 320, dX 320, dW+dU 544 tiles of 64 x 64 on
        1 280 workgroup slots -- one thin round, 1.25-2.1 workgroups per CU) are split until ~SPLITK_FILL_WGS slots are taken
        (same-box A/B at c4, tools/ab_c4.sh: dH 142 -> 105 us, dX 71 -> 54, dW+dU 210 -> 176; dEneg -- 504 tiles, 8 MB per
        slab for the scatter to re-read -- is the one product that loses and does not ask for it).  fill may be a slot count
        of its own: the grouped weight-gradient launch asks for SPLITK_FILL_WGRAD (its slabs are summed by the norm launch,
        which streams them once: 3 slabs of 544 tiles, 178 -> 117 us), while more slabs of dX cost the scatter more than
        the GEMM gains (45 -> 64 us at 5 slabs)."""
        mk = min_k or SPLITK_MIN_K
        sk = SPLITK_TARGET_WGS // max(tiles, 1)
        if fill and tiles >= 256:
            sk = max(sk, (SPLITK_FILL_WGS if fill is True else int(fill)) // tiles)
        if long_k and K >= 8192:
            # a long reduction into few tiles (saturated c3: the weight gradients, 108 tiles x K = 25 088 tokens): 512 / tiles
            # splits leave two thirds of the 1 280 workgroup slots empty and 392 K steps per workgroup; fill the slots while a
            # slab still reduces >= 2 048 tokens (the norm launch streams the slabs once)
            sk = max(sk, min(SPLITK_LONGK_WGS // max(tiles, 1), K // 2048))
        return int(max(1, min(32, sk, K // mk)))

    # ------------------------------------------------------------------ parameters (Keras layouts)
    def _gate_pad(self, w, rows_p):
        """(rows, G*H) -> (rows_p, G*Hp) with each gate block placed at g*Hp."""
        c = self.cfg
        w = np.asarray(w, dtype=np.float32)
        out = np.zeros((rows_p, self.GHp), np.float32)
        for g in range(self.G):
            out[: w.shape[0], g * self.Hp: g * self.Hp + c.H] = w[:, g * c.H:(g + 1) * c.H]
        return out

    def _gate_unpad(self, t, rows):
        c = self.cfg
        a = t.detach().cpu().numpy()
        return np.concatenate([a[:rows, g * self.Hp: g * self.Hp + c.H] for g in range(self.G)], axis=1)

    def set_param(self, name, value, accum=False):
        """Load an UNPADDED array (Keras layout) into the padded device tensor."""
        c = self.cfg
        tgt = self.A if accum else self.P
        v = np.asarray(value, dtype=np.float32)
        if name in ("Wk", "W"):
            rows_p = tgt[name].shape[0]
            arr = self._gate_pad(v, rows_p)
        elif name == "U":
            arr = self._gate_pad(v, self.Hp)
        elif name == "b":
            arr = self._gate_pad(v[None, :], 1)[0]
        elif name == "E":
            arr = np.zeros(tuple(tgt[name].shape), np.float32)
            arr[:, : v.shape[1]] = v
        elif name == "Wout":
            arr = np.zeros((self.Hp, self.Vp), np.float32)
            arr[: c.H, : c.V_out] = v
        elif name == "Eout":
            arr = np.zeros((c.V_out, self.Hp), np.float32)
            arr[:, : c.H] = v
        elif name in ("bout", "byy"):
            arr = np.zeros(tuple(tgt[name].shape), np.float32)
            arr[: c.V_out] = v
        elif name in ("Wyy", "Wxy"):
            arr = np.zeros(tuple(tgt[name].shape), np.float32)
            arr[: v.shape[0], : c.V_out] = v
        else:
            raise KeyError(name)
        tgt[name].copy_(torch.from_numpy(arr))
        if name == "U" and not accum:
            self.upack_dirty = True

    def get_param(self, name, accum=False, src=None):
        c = self.cfg
        self.check_status()               # a host sync anyway: never hand out weights of a run that failed on the device
        if src is None:
            src = self.A if accum else self.P
        t = src[name]
        if name == "Wk":
            return self._gate_unpad(t, c.V_in)
        if name == "W":
            return self._gate_unpad(t, c.D)
        if name == "U":
            return self._gate_unpad(t, c.H)
        if name == "b":
            return self._gate_unpad(t[None, :], 1)[0]
        a = t.detach().cpu().numpy()
        if name == "E":
            return a[:, : (c.H if c.tied else c.D)].copy()
        if name == "Wout":
            return a[: c.H, : c.V_out].copy()
        if name == "Eout":
            return a[:, : c.H].copy()
        if name in ("bout", "byy"):
            return a[: c.V_out].copy()
        if name == "Wyy":
            return a[:, : c.V_out].copy()
        if name == "Wxy":
            return a[: c.x_dim, : c.V_out].copy()
        raise KeyError(name)

    def _row_cols(self, name):
        """Unpadded width of a row of item table `name`."""
        c = self.cfg
        if name == "E":
            return c.H if c.tied else c.D
        if name == "Eout":
            return c.H
        if name == "bout":
            return 1
        raise KeyError("%s is not an item table with plain rows (E, Eout, bout)" % name)

    def get_rows(self, name, rows, accum=False):
        """Rows `rows` (int array) of item table `name` -- or of its Adagrad accumulator -- as an unpadded host array: the
        read-back of what ONE step changed at catalogue scale, where get_param would move the whole table."""
        self.check_status()
        t = (self.A if accum else self.P)[name]
        idx = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(self.dev)
        w = self._row_cols(name)
        if t.dim() == 1:
            return t[idx].cpu().numpy()
        return t[idx][:, :w].cpu().numpy()

    def set_rows(self, name, rows, values, accum=False):
        """Overwrite rows of an item table (accum=True: of its accumulator) with unpadded host values; padded columns stay
        zero.  With set_param(..., accum=True) this loads a complete optimizer state (bench.py's re-synchronised parity leg)."""
        t = (self.A if accum else self.P)[name]
        idx = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int64)).to(self.dev)
        v = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float32)).to(self.dev)
        if t.dim() == 1:
            t[idx] = v.reshape(-1)
            return
        w = self._row_cols(name)
        if w == t.shape[1]:
            t[idx] = v
        else:
            pad = torch.zeros((idx.numel(), t.shape[1]), dtype=torch.float32, device=self.dev)
            pad[:, :w] = v
            t[idx] = pad

    def set_sampler(self, thresh, alias, logq=None):
        """Alias table of the negative-sampling proposal (built on the host)."""
        th = torch.from_numpy(np.asarray(thresh, dtype=np.uint32).view(np.int32).copy()).to(self.dev)
        al = torch.from_numpy(np.asarray(alias, dtype=np.int32).copy()).to(self.dev)
        lq = None if logq is None else torch.from_numpy(np.asarray(logq, dtype=np.float32).copy()).to(self.dev)
        self.sampler = (th, al, lq)
        self._plan = None

    def set_prior(self, name, means, strength):
        """Kernel regularizer strength * sum (w - means)^2 on parameter `name` (must be listed in
        NetConfig.priors): GaussPriorRegularizer(means, var) -> strength = 1 / (2 var) (model.py:80);
        keras l2(l) -> means None, strength = l."""
        if name not in self.cfg.priors:
            raise ValueError("%s is not declared in NetConfig.priors" % name)
        mt = None
        if means is not None:
            mt = torch.zeros_like(self.P[name])
            m = np.asarray(means, dtype=np.float32)
            mt[: m.shape[0], : m.shape[1]] = torch.from_numpy(m).to(self.dev)
        self.priors[name] = (mt, float(strength))

    def _apply_priors(self, with_grad):
        """reg_sum = sum of the penalties; with_grad: their gradients are added to the dense gradients."""
        st = self._stream()
        self.reg_sum.zero_()
        for name, (mt, strength) in self.priors.items():
            g = self.Gd[name] if (with_grad and self.trainable[name]) else None
            call("seqrec_prior_grad", ptr(self.P[name]), ptr(mt), self.P[name].numel(), strength, ptr(g), ptr(self.reg_sum), st)

    # ------------------------------------------------------------------ batch upload
    def upload(self, rb):
        """Host RaggedBatch -> device index arrays (one pinned-free H2D copy of a single int32 blob)."""
        c = self.cfg
        n = rb.n_tok
        parts = [rb.step_off.astype(np.int32), rb.prev.astype(np.int32)]
        if rb.ids is not None:
            parts.append(rb.ids.astype(np.int32))
        if rb.tgt is not None:
            parts.append(rb.tgt.astype(np.int32))
        # every non-blocking upload goes through the engine's page-locked ring (PinnedRing): safe by construction
        blob = self.pinned.put(np.concatenate(parts))
        o = 0
        d = {"n": n, "T": rb.T, "B": rb.B, "rb": rb}
        d["step_off"] = blob[o:o + rb.T + 1]; o += rb.T + 1
        d["prev"] = blob[o:o + n]; o += n
        if rb.ids is not None:
            d["ids"] = blob[o:o + n]; o += n
        if rb.tgt is not None:
            d["tgt"] = blob[o:o + n]; o += n
        if rb.x is not None:
            x = np.zeros((n, self.Fp), np.float32)
            x[:, : rb.x.shape[1]] = rb.x
            d["x"] = self.pinned.put(x)
        if getattr(rb, "xs", None) is not None and c.x_to_y:
            xs = np.zeros((n, self.Fxp), np.float32)
            xs[:, : rb.xs.shape[1]] = rb.xs
            d["xs"] = self.pinned.put(xs)
        d["blob"] = blob
        return d

    def put_dataset(self, flat, starts):
        """Park a whole dataset in HBM (flat item ids + session offsets) for upload_device()."""
        flat = np.ascontiguousarray(flat, dtype=np.int32)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        return {"flat": torch.from_numpy(flat).to(self.dev), "starts": torch.from_numpy(starts).to(self.dev),
                "starts_host": starts}

    def _materialise(self, d):
        """A batch built with upload_device(defer=True) whose gather has not been launched yet: launch it now (seqrec_pack_batch_host).
        The training step's one-call form launches it INSIDE its prologue instead (seqrec_rnn_pack_u_sample_batch)."""
        pend = d.pop("_pending", None)
        if pend is not None:
            ds, sess, so32 = pend
            call("seqrec_pack_batch_host", ptr(ds["flat"]), ptr(ds["starts"]), sess.ctypes.data, so32.ctypes.data, d["B"], d["T"],
                 ptr(d["sess"]), ptr(d["step_off"]), ptr(d["ids"]), ptr(d["tgt"]), ptr(d["prev"]), self._stream())
        return d

    def upload_device(self, ds, sel, history=False, freq=False, defer=False):
        """Device-side batcher: like upload(batching.pack_flat(flat, starts, sel)) but only
        the batch's session indices and step offsets (a few KB) cross PCIe; ids / targets / prev links
        -- and, with history=True, the x_to_y history features of datasets.build_xs -- are produced
        by seqrec_pack_batch / seqrec_history_features from the HBM-resident dataset.
        defer=True (training loops): the gather launch is left to the consumer -- the one-call training step issues it inside its
        prologue launch together with the U re-pack and the negatives (one launch instead of two); every other consumer calls
        _materialise() first, so the batch reads the same either way."""
        from .batching import index_flat
        c = self.cfg
        rb = index_flat(ds["starts_host"], sel, lean=True)
        n, T, B = rb.n_tok, rb.T, rb.B
        sess = np.asarray(sel, dtype=np.int64)[rb.order].astype(np.int32)
        out = torch.empty(3 * max(n, 1), dtype=torch.int32, device=self.dev)
        st = self._stream()
        so32 = np.ascontiguousarray(rb.step_off, dtype=np.int32)
        if B + T + 1 <= _lib.PACK_HOST_MAX:
            # offsets and session indices ride in the launch's kernel arguments: no copy in front of the launch
            blob = torch.empty(T + 1 + B, dtype=torch.int32, device=self.dev)
            d = {"n": n, "T": T, "B": B, "rb": rb, "blob": blob, "step_off": blob[: T + 1], "sess": blob[T + 1:],
                 "ids": out[:n], "tgt": out[n:2 * n], "prev": out[2 * n:3 * n], "_out": out}
            if defer and not history and n > 0 and B + T + 1 <= _lib.PACK_MERGED_MAX:
                d["_pending"] = (ds, sess, so32)
            else:
                call("seqrec_pack_batch_host", ptr(ds["flat"]), ptr(ds["starts"]), sess.ctypes.data, so32.ctypes.data, B, T,
                     ptr(d["sess"]), ptr(d["step_off"]), ptr(d["ids"]), ptr(d["tgt"]), ptr(d["prev"]), st)
        else:
            blob = self.pinned.put(np.concatenate([so32, sess]))
            d = {"n": n, "T": T, "B": B, "rb": rb, "blob": blob, "step_off": blob[: T + 1], "sess": blob[T + 1:],
                 "ids": out[:n], "tgt": out[n:2 * n], "prev": out[2 * n:3 * n], "_out": out}
            call("seqrec_pack_batch", ptr(ds["flat"]), ptr(ds["starts"]), ptr(d["sess"]), ptr(d["step_off"]), B, T, ptr(d["ids"]),
                 ptr(d["tgt"]), ptr(d["prev"]), st)
        if history:
            if not c.x_to_y:
                raise ValueError("history features feed the x_to_y branch (NetConfig.x_to_y)")
            xs = torch.empty((max(n, 1), self.Fxp), dtype=torch.float32, device=self.dev)
            call("seqrec_history_features", ptr(ds["flat"]), ptr(ds["starts"]), ptr(d["sess"]), ptr(d["step_off"]), B, T,
                 c.x_dim, self.Fxp, int(bool(freq)), ptr(xs), st)
            d["xs"] = xs[:n]
        return d

    def _drop_masks(self, d, step):
        """Inverted-dropout multipliers for this step (counter RNG; oracle/rng.py)."""
        c = self.cfg
        rb = d["rb"]
        n = d["n"]
        st = self._stream()
        out = {}
        if c.drop_in > 0 or c.drop_out > 0 or c.drop_rec > 0:
            rb.ensure_tokens()
        if c.drop_in > 0 or c.drop_out > 0:
            key = (rb.tok_b.astype(np.int64) << 16) + rb.tok_s.astype(np.int64)
        if n == 0:
            return out
        if c.drop_in > 0:
            sid = _lib.STREAM_DROP_IN + 16 * (step + 1)
            if c.input == "onehot":
                ids_host = rb.ids if rb.ids is not None else d["ids"].cpu().numpy()     # device-packed batch
                rk = torch.from_numpy(key * c.V_in + ids_host.astype(np.int64)).to(self.dev)
                m = self.buf("in_scale", n)
                call("seqrec_dropout_mask", self.drop_seed, sid, ptr(rk), n, 1, 1, float(c.drop_in), ptr(m), st)
            else:
                w = c.D if c.input == "embed" else c.V_in
                ld = self.Dp if c.input == "embed" else self.Fp
                rk = torch.from_numpy(key).to(self.dev)
                m = self.buf("in_mask", n, ld)
                m.zero_()
                call("seqrec_dropout_mask", self.drop_seed, sid, ptr(rk), n, w, ld, float(c.drop_in), ptr(m), st)
            out["in"] = m
            out["_rk_in"] = rk
        if c.drop_rec > 0:
            # Keras recurrent_dropout: one mask per gate and session, reused at every step.
            # row key = gate * 2^20 + ORIGINAL batch index of the session
            sid = _lib.STREAM_DROP_REC + 16 * (step + 1)
            G, B = self.G, d["B"]
            rk = (np.arange(G, dtype=np.int64)[:, None] * (1 << 20) + rb.order.astype(np.int64)[None, :]).reshape(-1)
            rkd = torch.from_numpy(rk).to(self.dev)
            m = self.buf("rec_mask", G * B, self.Hp)
            m.zero_()
            call("seqrec_dropout_mask", self.drop_seed, sid, ptr(rkd), G * B, c.H, self.Hp, float(c.drop_rec), ptr(m), st)
            out["rec"] = m
            out["_rk_rec"] = rkd
        if c.drop_out > 0:
            sid = _lib.STREAM_DROP_OUT + 16 * (step + 1)
            rk = torch.from_numpy(key).to(self.dev)
            m = self.buf("out_mask", n, self.Hp)
            m.zero_()
            call("seqrec_dropout_mask", self.drop_seed, sid, ptr(rk), n, c.H, self.Hp, float(c.drop_out), ptr(m), st)
            out["out"] = m
            out["_rk_out"] = rk
        return out

    def _job(self, name, rows, vals, ldv, row_scale, n, width, base, n_slabs=0, slab_stride=0):
        """One scatter list of this step for table `name` (see seqrec_rows_job)."""
        return dict(table=self.P[name], accum=self.A[name], gtab=self.Gt[name], slot=self.slot[name], rows=rows, vals=vals,
                    ldv=ldv, row_scale=row_scale, n=n, width=width, base=base, name=name, n_slabs=n_slabs, slab_stride=slab_stride)

    def _merge_sorted(self, jobs):
        """Deterministic merge (csrc/merge.hip): the scatter lists of each table are sorted by row (stable) and
        summed per row in increasing contribution order -- what seqrec_rows_scatter_add_multi does with float
        atomics, bitwise reproducible."""
        st = self._stream()
        by_table = {}
        for j in jobs:
            by_table.setdefault(j["name"], []).append(j)
        lib = _lib.load()
        for name, js in by_table.items():
            total = sum(int(j["n"]) for j in js)
            if total == 0:
                continue
            key = (total, int(js[0]["width"]))
            nbytes = self._merge_ws.get(key)
            if nbytes is None:
                nbytes = self._merge_ws[key] = int(lib.seqrec_rows_merge_workspace_bytes(total, key[1]))
                if nbytes <= 0:
                    raise _lib.SeqrecError("seqrec_rows_merge_workspace_bytes failed")
            ws = self.buf("merge_ws", (nbytes + 3) // 4)
            arr, cnt = _lib.rows_jobs(js)
            call("seqrec_rows_merge_sorted", arr, cnt, ptr(ws), nbytes, st)

    # ------------------------------------------------------------------ recurrent scan
    def _scan_fwd(self, d, XW, Hout, gates, aux, rmask=None):
        c, st = self.cfg, self._stream()
        Hp = self.Hp
        if self.upack_dirty:
            call("seqrec_rnn_pack_u_stepwise" if self.stepwise else "seqrec_rnn_pack_u", CELL[c.cell], Hp, ptr(self.P["U"]),
                 ptr(self.upack), st)
            self.upack_dirty = False
        if self.stepwise:
            so = d["rb"].step_off
            call("seqrec_rnn_fwd_stepwise", CELL[c.cell], ACT[c.act], Hp, c.H, d["T"], d["B"], None, so.ctypes.data, ptr(XW),
                 ptr(Hout), ptr(gates), ptr(aux), ptr(self.upack), ptr(rmask), int(self.use_graph), st)
        else:
            call("seqrec_rnn_fwd", CELL[c.cell], ACT[c.act], Hp, c.H, d["T"], d["B"], ptr(d["step_off"]), ptr(XW),
                 ptr(Hout), ptr(gates), ptr(aux), ptr(self.upack), st)

    def _scan_bwd(self, d, dHout, Hout, gates, aux, dPre, rmask=None, parts=None):
        """parts (_lib.dh_parts): dHout arrives as split-K slabs + a row term; dHout is then scratch for the scan forms that need
        the sum in one array (seqrec_rnn_bwd_stepwise_parts)."""
        c, st = self.cfg, self._stream()
        Hp = self.Hp
        if self.stepwise and parts is not None:
            import ctypes
            so = d["rb"].step_off
            cap = d["n"]
            wsp = self.buf("scan_ws", 2 * cap * Hp)
            call("seqrec_rnn_bwd_stepwise_parts", CELL[c.cell], ACT[c.act], Hp, c.H, d["T"], d["B"], None, so.ctypes.data, cap,
                 ctypes.addressof(parts), ptr(dHout), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(self.upack), ptr(wsp),
                 ptr(rmask), int(self.use_graph), st, prof_name="seqrec_rnn_bwd_stepwise")
        elif self.stepwise:
            so = d["rb"].step_off
            cap = d["n"]
            wsp = self.buf("scan_ws", 2 * cap * Hp)
            call("seqrec_rnn_bwd_stepwise", CELL[c.cell], ACT[c.act], Hp, c.H, d["T"], d["B"], None, so.ctypes.data, cap,
                 ptr(dHout), ptr(Hout), ptr(gates), ptr(aux), ptr(dPre), ptr(self.upack), ptr(wsp), ptr(rmask),
                 int(self.use_graph), st)
        else:
            call("seqrec_rnn_bwd", CELL[c.cell], ACT[c.act], Hp, c.H, d["T"], d["B"], ptr(d["step_off"]), ptr(dHout), ptr(Hout),
                 ptr(gates), ptr(aux), ptr(dPre), ptr(self.upack), st)

    # ------------------------------------------------------------------ forward
    def forward(self, d, train=False, step=0, want_probs=False, negatives=None, stop_at_hidden=False, defer_loss=False):
        """Runs the graph up to the loss.  Returns a dict of device tensors; in training
        mode the gradient w.r.t. the logits is left in place of the logits."""
        c, P = self.cfg, self.P
        st = self._stream()
        if "_pending" in d:
            self._materialise(d)
        n, T, B = d["n"], d["T"], d["B"]
        Hp, GHp = self.Hp, self.GHp
        r = {}
        if n == 0:
            self.loss_out.zero_()
            return r
        drops = self._drop_masks(d, step) if train else {}
        r["drops"] = drops
        pre_neg = None
        if (self.stepwise and self.upack_dirty and c.output != "full" and negatives is None and not stop_at_hidden
                and self.sampler is not None and self._fuse_prologue):
            # the re-pack of U and the draw + gather of the step's negatives both depend on the weights alone: one launch
            K = c.K
            Et = P["E"] if c.tied else P["Eout"]
            th, al, lq = self.sampler
            pre_neg = (self.buf("neg", K, dtype=torch.int32), self.buf("Eneg", K, Hp), self.buf("lq_neg", K) if c.logq else None)
            call("seqrec_rnn_pack_u_sample", CELL[c.cell], Hp, ptr(P["U"]), ptr(self.upack), int(c.seed), int(step), K, ptr(th),
                 ptr(al), c.V_out, ptr(Et), Hp, ptr(lq if c.logq else None), ptr(pre_neg[0]), ptr(pre_neg[1]), ptr(pre_neg[2]), st)
            self.upack_dirty = False
        XW = self.buf("XW", n, GHp)
        bias = P.get("b")
        if c.input == "onehot":
            call("seqrec_gather_rows", ptr(P["Wk"]), ptr(d["ids"]), ptr(XW), n, GHp, ptr(drops.get("in")), ptr(bias), 0, st)
        else:
            xidx = None
            if c.input == "embed" and "in" not in drops and n <= self.fuse_gather_max:
                # the embedding lookup is fused into the cell's input GEMM: A = E read THROUGH the ids (no X copy)
                X, xidx = P["E"], d["ids"]
                Wm, Kd = P["W"], self.Dp
            elif c.input == "embed":
                X = self.buf("X", n, self.Dp)
                call("seqrec_gather_rows", ptr(P["E"]), ptr(d["ids"]), ptr(X), n, self.Dp, None, None, 0, st, tag="E")
                Wm, Kd = P["W"], self.Dp
            else:
                X = d["x"]
                Wm, Kd = P["Wk"], self.Fp
            if "in" in drops:
                Xd = self.buf("Xd", n, Kd)
                call("seqrec_mul", ptr(X), ptr(drops["in"]), ptr(Xd), n * Kd, st)
                X = Xd
            r["X"], r["X_index"] = X, xidx
            self.gemm(1, 0, n, GHp, Kd, X, Kd, Wm, GHp, XW, GHp, bias=bias, tag="xw",
                      fuse=None if xidx is None else _lib.gemm_fuse(a_index=xidx))
        Hout = self.buf("Hout", n, Hp)
        gates = self.buf("gates", n, GHp)
        aux = self.buf("aux", n, Hp)
        self._scan_fwd(d, XW, Hout, gates, aux, drops.get("rec"))
        r.update(XW=XW, Hout=Hout, gates=gates, aux=aux)
        Hd = Hout
        if "out" in drops:
            Hd = self.buf("Hd", n, Hp)
            call("seqrec_mul", ptr(Hout), ptr(drops["out"]), ptr(Hd), n * Hp, st)
        r["Hd"] = Hd
        if stop_at_hidden:
            return r
        tgt = d.get("tgt")
        loss_rows = self.buf("loss_rows", n)
        inv = 1.0 / n
        if c.output == "full":
            Vp = self.Vp
            logits = self.buf("logits", n, Vp)
            if c.y_to_y and Vp != c.V_out:
                logits.zero_()                      # the row scatter of dlogits below is Vp wide
            self.gemm(1, 0, n, c.V_out, Hp, Hd, Hp, P["Wout"], Vp, logits, Vp, bias=P.get("bout"), tag="logits")
            if c.x_to_y:
                self.gemm(1, 0, n, c.V_out, self.Fxp, d["xs"], self.Fxp, P["Wxy"], Vp, logits, Vp, accumulate=1, tag="x_to_y")
            if c.y_to_y:
                call("seqrec_gather_rows", ptr(P["Wyy"]), ptr(d["ids"]), ptr(logits), n, Vp, None, ptr(P.get("byy")), 1, st)
            probs = self.buf("probs", n, c.V_out) if want_probs else None
            call("seqrec_full_softmax_ce", ptr(logits), Vp, ptr(tgt), n, c.V_out, inv, ptr(loss_rows), ptr(probs), st)
            r["dlogits"] = logits
            r["probs"] = probs
        else:
            K = c.K
            Et = P["E"] if c.tied else P["Eout"]
            th, al, lq = self.sampler
            Eneg = self.buf("Eneg", K, Hp)
            lq_neg = self.buf("lq_neg", K) if c.logq else None      # the candidates' log-Q once per step, not once per row
            if pre_neg is not None:
                neg = pre_neg[0]                                    # drawn and gathered in the step's opening launch
            elif negatives is None:
                neg = self.buf("neg", K, dtype=torch.int32)         # draw + row gather + log-Q gather: one launch
                call("seqrec_sample_gather", int(c.seed), int(step), K, ptr(th), ptr(al), c.V_out, ptr(Et), Hp,
                     ptr(lq if c.logq else None), ptr(neg), ptr(Eneg), ptr(lq_neg), st)
            else:
                neg = negatives
                call("seqrec_gather_rows", ptr(Et), ptr(neg), ptr(Eneg), K, Hp, None, None, 0, st)
                if c.logq:
                    call("seqrec_gather_rows", ptr(lq), ptr(neg), ptr(lq_neg), K, 1, None, None, 0, st)
            ln = self.buf("ln", n, K)
            self.gemm(1, 1, n, K, Hp, Hd, Hp, Eneg, Hp, ln, K, tag="logits")
            dlt = self.buf("dlt", n)
            call("seqrec_sampled_softmax_ce", ptr(ln), K, ptr(Hd), Hp, ptr(Et), ptr(P.get("bout")),
                 ptr(lq if c.logq else None), ptr(lq_neg), ptr(tgt), ptr(neg), n, K, inv, ptr(loss_rows), ptr(dlt), st)
            r.update(neg=neg, Eneg=Eneg, dln=ln, dlt=dlt)
        r["loss_rows"] = loss_rows if tgt is not None else None
        if tgt is not None and not defer_loss:
            call("seqrec_loss_reduce", ptr(loss_rows), n, ptr(self.loss_out), st)
        return r

    # ------------------------------------------------------------------ training step
    def train_step(self, d, lr=0.01, eps=1e-8, clipnorm=1.0, step=None, negatives=None, apply_update=True):
        """One full step on an uploaded batch: forward, masked-mean CE, BPTT, global-norm clip,
        Adagrad.  Returns the batch loss as a 1-element device tensor (no host sync)."""
        c, P = self.cfg, self.P
        self.last_slabs.clear()
        if step is None:
            step = self.step_count
        self.step_count = step + 1
        n, T, B = d["n"], d["T"], d["B"]
        if n == 0:
            return torch.zeros(1, device=self.dev)
        st = self._stream()
        Hp, GHp = self.Hp, self.GHp
        if n <= self.fuse_gather_max and self._native_cell_ok(negatives, apply_update):
            return self._train_step_native(d, lr, eps, clipnorm, step)
        # the batch loss is reduced by a spare workgroup of the gradient-norm launch (defer_loss) when that launch runs
        r = self.forward(d, train=True, step=step, negatives=negatives, defer_loss=apply_update)
        drops = r["drops"]
        lrows = r["loss_rows"]
        Hd = r["Hd"]
        Gd, Gt = self.Gd, self.Gt
        tr = self.trainable
        sparse_jobs = []     # scatter lists of this step (see _job)
        join_side = False    # a side-stream GEMM has to be joined before its consumer
        deneg_late, deneg_at = None, -1      # dEneg deferred into the weight-gradient launch: (table, rows, dln, Hd, K), its scatter-list slot
        dh_parts = None      # dH as split-K slabs + target-row term for the BPTT (seqrec_rnn_bwd_stepwise_parts)
        wgrad = []           # deferred weight-gradient GEMMs (M, N, K, A, lda, B, ldb, C, ldc), launched grouped
        wcover = set()       # dense tensors whose gradient those products tile completely
        wg_slabs = None      # (descs, count, n_slabs, workspace) when the products were left as split-K slabs for the norm launch
        dHd = self.buf("dHd", n, Hp)
        cs_ws = self.buf("colsum_ws", 64 * max(GHp, self.Vp, c.K if c.output == "sampled" else 1))
        if c.output == "full":
            Vp = self.Vp
            dl = r["dlogits"]
            if tr["Wout"]:
                self.gemm(0, 0, Hp, c.V_out, n, Hd, Hp, dl, Vp, Gd["Wout"], Vp, splitk=self._splitk(Hp, c.V_out, n), tag="dWout")
            if c.out_bias and tr["bout"]:
                call("seqrec_colsum", ptr(dl), n, c.V_out, Vp, ptr(Gd["bout"]), 0, ptr(cs_ws), st)
            if c.x_to_y and tr["Wxy"]:
                self.gemm(0, 0, self.Fxp, c.V_out, n, d["xs"], self.Fxp, dl, Vp, Gd["Wxy"], Vp,
                          splitk=self._splitk(self.Fxp, c.V_out, n), tag="dWxy")
            if c.y_to_y:
                if tr["Wyy"] and "Wyy" in self.table_params:
                    sparse_jobs.append(self._job("Wyy", d["ids"], dl, Vp, None, n, Vp, 0))
                elif tr["Wyy"]:            # regularized: dense gradient = scatter of the batch rows + prior term
                    Gd["Wyy"].zero_()
                    sl = self.buf("wyy_slot", c.V_out, dtype=torch.int32)
                    call("seqrec_rows_scatter_add", ptr(Gd["Wyy"]), ptr(sl), ptr(d["ids"]), ptr(dl), Vp, None, n, Vp, 0, st)
                if c.yy_bias and tr["byy"]:
                    call("seqrec_colsum", ptr(dl), n, c.V_out, Vp, ptr(Gd["byy"]), 0, ptr(cs_ws), st)
            self.gemm(1, 1, n, Hp, c.V_out, dl, Vp, P["Wout"], Vp, dHd, Hp, splitk=self._splitk(n, Hp, c.V_out, fill=True), tag="dH")
        else:
            K = c.K
            tname = "E" if c.tied else "Eout"
            Et = P[tname]
            dln, dlt, neg, Eneg = r["dln"], r["dlt"], r["neg"], r["Eneg"]
            # dH = dlogits . Eneg + dlt * Eout[tgt]: the target-row term rides in the GEMM's final write
            sk_h = self._splitk(n, Hp, K, fill=True)
            pair = tr[tname] and self._pair_ok(n, sk_h, self._splitk(K, Hp, n)) and apply_update and not self.priors
            if pair:
                # dH and dEneg = dlogits^T . H are both products of dlogits and neither fills the chip alone at these sizes: ONE launch
                # (seqrec_gemm_f32_pair: 46.7 us side by side against 54.2 back to back, tools/pair_probe.py); dEneg comes out as slabs
                import ctypes
                q = self._pair_plan(n, K, dln, Eneg, dHd, sk_h, Et, d["tgt"], dlt, Hd, self._splitk(K, Hp, n))
                call("seqrec_gemm_f32_pair", ctypes.addressof(q), st, tag="dH+dEneg")
                pair_ns = int(q.n_slabs1)
            elif self._slab_dh and self.stepwise and sk_h > 1 and "out" not in drops:
                # dH's only reader is the BPTT: its split-K slabs and the target-row term go there as parts (the cluster scan
                # adds them where it reads dHout -- no reduce launch; other scan forms sum them into dHd first)
                hs, ns_h, ss_h = self.gemm_slabs(1, 0, n, Hp, K, dln, K, Eneg, Hp, "dH_slabs", sk_h, tag="dH")
                dh_parts = _lib.dh_parts(hs, ns_h, ss_h, add_table=Et, add_index=d["tgt"], add_scale=dlt, add_ld=Hp)
            else:
                self.gemm(1, 0, n, Hp, K, dln, K, Eneg, Hp, dHd, Hp, splitk=sk_h, tag="dH",
                          fuse=_lib.gemm_fuse(add_table=Et, add_index=d["tgt"], add_scale=dlt, add_ld=Hp))
            if tr[tname]:
                dEneg = self.buf("dEneg", K, Hp)
                ns_neg = ss_neg = 0
                if pair:
                    dEneg, ns_neg, ss_neg = self.buf("dEneg_slabs", max(self._splitk(K, Hp, n), 1) * K * Hp), pair_ns, K * Hp
                    self.last_slabs["dEneg_slabs"] = (dEneg, pair_ns, K, Hp)
                elif self._overlap and _PROF is None:      # (the per-call profile times calls on the main stream)
                    # dEneg = dlogits^T . H needs nothing from the BPTT: it runs on a side stream UNDER the (latency-bound,
                    # one-launch) BPTT and is joined in front of the scatter that consumes it
                    side = self._side()
                    self._ev_fork.record(torch.cuda.current_stream(self.dev))
                    side.wait_event(self._ev_fork)
                    self._cur_st = side.cuda_stream
                    self.gemm(0, 0, K, Hp, n, dln, K, Hd, Hp, dEneg, Hp, splitk=self._splitk(K, Hp, n), tag="dEneg",
                              ws_name="gemm_ws_side")
                    self._cur_st = st
                    self._ev_join.record(side)
                    join_side = True
                elif (self._slab_scatter and c.merge != "sorted" and self._group_deneg and self._slab_wgrad and apply_update and not self.priors
                      and "rec" not in drops and c.input != "onehot" and n <= 4096
                      and ((K + 63) // 64) * ((Hp + 63) // 64) + ((self.Dp + 63) // 64 + (Hp + 63) // 64 + 1) * ((GHp + 63) // 64) <= 512):
                    # ... and it waits for the weight-gradient launch behind the BPTT (its reader, the scatter, comes later still);
                    # small shapes only: every product of a grouped launch takes the same number of splits, and a large dEneg
                    # (c4: 504 tiles, 8 MB per slab for the scatter to re-read) wants fewer than the weight gradients do; with
                    # many tokens (saturated c3: 25 k) dEneg alone runs at 73 % of the MFMA peak and the mixed launch at 36 %
                    deneg_late = (tname, neg, dln, Hd, K)
                    dEneg = None
                elif self._slab_scatter and c.merge != "sorted":
                    # dEneg's only reader is the row scatter: the split-K slabs go there as they are (no reduce launch)
                    dEneg, ns_neg, ss_neg = self.gemm_slabs(0, 0, K, Hp, n, dln, K, Hd, Hp, "dEneg_slabs", self._splitk(K, Hp, n), tag="dEneg")
                else:
                    self.gemm(0, 0, K, Hp, n, dln, K, Hd, Hp, dEneg, Hp, splitk=self._splitk(K, Hp, n), tag="dEneg")
                sparse_jobs.append(self._job(tname, d["tgt"], Hd, Hp, dlt, n, Hp, 0))
                if dEneg is None:
                    deneg_at = len(sparse_jobs)          # filled in behind the grouped launch
                    sparse_jobs.append(None)
                else:
                    sparse_jobs.append(self._job(tname, neg, dEneg, Hp, None, K, Hp, n, ns_neg, ss_neg))
            if c.out_bias and tr["bout"]:
                dbn = self.buf("dbn", K)
                call("seqrec_colsum", ptr(dln), n, K, K, ptr(dbn), 0, ptr(cs_ws), st)
                sparse_jobs.append(self._job("bout", d["tgt"], dlt, 1, None, n, 1, 0))
                sparse_jobs.append(self._job("bout", neg, dbn, 1, None, K, 1, n))
        dHout = dHd
        if "out" in drops:
            call("seqrec_mul", ptr(dHd), ptr(drops["out"]), ptr(dHd), n * Hp, st)
        dPre = self.buf("dPre", n, GHp)
        self._scan_bwd(d, dHout, r["Hout"], r["gates"], r["aux"], dPre, drops.get("rec"), parts=dh_parts)
        bias_in_group = c.use_bias and tr["b"] and "rec" not in drops
        if c.use_bias and tr["b"] and not bias_in_group:
            call("seqrec_colsum", ptr(dPre), n, GHp, GHp, ptr(Gd["b"]), 0, ptr(cs_ws), st)
        if tr["U"]:
            if "rec" in drops or n > self.fuse_gather_max:
                # many tokens (saturated batches: 25 k): the weight-gradient GEMM that gathers its A rows along K runs at ~45 % of the
                # plain one's rate per K step; a 26 MB copy of h_{t-1} (and of E[ids], above) costs ~10 us and serves it at full rate
                Hprev, hidx = self.buf("Hprev", n, Hp), None
                call("seqrec_gather_rows", ptr(r["Hout"]), ptr(d["prev"]), ptr(Hprev), n, Hp, None, None, 0, st, tag="Hprev")
            else:
                Hprev, hidx = r["Hout"], d["prev"]      # h_{t-1} rows are read through the prev links inside the GEMM
            sk = self._splitk(Hp, GHp, n)
            if "rec" in drops:
                # dU_g = (A_g * m_g)^T . dPre_g with the gate's time-invariant mask expanded to tokens
                B_ = d["B"]
                rowidx = self.buf("tok_row", n, dtype=torch.int32)
                rowidx.copy_(torch.from_numpy(d["rb"].tok_row).to(self.dev))
                mt = self.buf("mask_tok", n, Hp)
                Am = self.buf("A_masked", n, Hp)
                for g in range(self.G):
                    call("seqrec_gather_rows", ptr(drops["rec"][g * B_:(g + 1) * B_]), ptr(rowidx), ptr(mt), n, Hp, None, None, 0, st)
                    src = r["aux"] if (c.cell == "gru" and g == 2) else Hprev
                    call("seqrec_mul", ptr(src), ptr(mt), ptr(Am), n * Hp, st)
                    self.gemm(0, 0, Hp, Hp, n, Am, Hp, dPre[:, g * Hp:], GHp, Gd["U"][:, g * Hp:], GHp,
                              splitk=self._splitk(Hp, Hp, n), tag="dU")
            elif c.cell == "gru":
                wgrad.append((Hp, 2 * Hp, n, Hprev, Hp, dPre, GHp, Gd["U"], GHp, hidx))
                wgrad.append((Hp, Hp, n, r["aux"], Hp, dPre[:, 2 * Hp:], GHp, Gd["U"][:, 2 * Hp:], GHp))
                wcover.add("U")
            else:
                wgrad.append((Hp, GHp, n, Hprev, Hp, dPre, GHp, Gd["U"], GHp, hidx))
                wcover.add("U")
        if c.input == "onehot":
            if tr["Wk"]:
                sparse_jobs.append(self._job("Wk", d["ids"], dPre, GHp, drops.get("in"), n, GHp, 0))
        else:
            X = r["X"]
            Kd = X.shape[1]
            wname = "W" if c.input == "embed" else "Wk"
            if tr[wname]:
                wgrad.append((Kd, GHp, n, X, Kd, dPre, GHp, Gd[wname], GHp, r["X_index"]))
                wcover.add(wname)
            if c.input == "embed" and tr["E"]:
                ns_x = ss_x = 0
                if self._slab_scatter and c.merge != "sorted" and "in" not in drops:
                    # dX too is read by the scatter alone: split K until the chip is full (160 tiles of 64x64 at c3: 3 slabs of
                    # K = 256 measured best of 1 / 2 / 3 / 4 / 6), slabs unreduced
                    sk_x = self._splitk_tiles(((n + 63) // 64) * ((self.Dp + 63) // 64), GHp, min_k=self._slab_min_k, fill=True)
                    dX, ns_x, ss_x = self.gemm_slabs(1, 1, n, self.Dp, GHp, dPre, GHp, P["W"], GHp, "dX_slabs", sk_x, tag="dX")
                else:
                    dX = self.buf("dX", n, self.Dp)
                    self.gemm(1, 1, n, self.Dp, GHp, dPre, GHp, P["W"], GHp, dX, self.Dp, splitk=self._splitk(n, self.Dp, GHp, fill=True), tag="dX")
                    if "in" in drops:
                        call("seqrec_mul", ptr(dX), ptr(drops["in"]), ptr(dX), n * self.Dp, st)
                base_i = (n + c.K) if c.tied else 0
                sparse_jobs.append(self._job("E", d["ids"], dX, self.Dp, None, n, self.Dp, base_i, ns_x, ss_x))
        if bias_in_group:
            if 0 < len(wgrad) < 4:
                # db = ones^T . dPre rides in the same launch as the other token reductions (M = 1)
                wgrad.append((1, GHp, n, self._ones(n), self.ONES_LD, dPre, GHp, Gd["b"], GHp))
                wcover.add("b")
            else:
                call("seqrec_colsum", ptr(dPre), n, GHp, GHp, ptr(Gd["b"]), 0, ptr(cs_ws), st)
        if wgrad:
            # the weight gradients A^T . dPre all reduce over the tokens: one grouped split-K launch
            tiles = sum(((w_[0] + 63) // 64) * ((w_[1] + 63) // 64) for w_ in wgrad)
            sk = self._splitk_tiles(tiles, n, fill=SPLITK_FILL_WGRAD, long_k=True)
            slabs_ok = (self._slab_wgrad and apply_update and c.merge != "sorted" and not self.priors and len(sparse_jobs) <= 4
                        and len([k for k in Gd if tr[k]]) <= 8)
            ride = deneg_late is not None and slabs_ok and len(wgrad) < 6
            if ride:
                # dEneg = dlogits^T . H rides as the LAST problem (same layout, same reduction over the tokens); with it the launch
                # is sized for ~1 000 workgroups
                t_, rows_, dln_, Hd_, K_ = deneg_late
                tiles += ((K_ + 63) // 64) * ((Hp + 63) // 64)
                sk = max(sk, int(max(1, min(1024 // max(tiles, 1), n // SPLITK_MIN_K))))
                wgrad_all = wgrad + [(K_, Hp, n, dln_, K_, Hd_, Hp, Hd_, Hp)]
            else:
                wgrad_all = wgrad
            wsz = sum(sk * w_[0] * w_[1] for w_ in wgrad_all)
            wsp = self.buf("gemm_ws", wsz) if (sk > 1 or ride) else None
            if slabs_ok and (sk > 1 or ride):
                # the split-K slabs stay unreduced: the norm launch adds them, writes the gradients and takes their squares
                import ctypes
                descs, ns = _lib.gemm_descs(wgrad_all), ctypes.c_int(0)
                call("seqrec_gemm_f32_grouped_slabs", len(wgrad_all), 0, 0, descs, sk, ptr(wsp), ctypes.addressof(ns), st,
                     tag="dW+dU+dEneg" if ride else "dW+dU")
                wg_slabs = (descs, len(wgrad), int(ns.value), wsp)
                if ride:
                    nsv = int(ns.value)
                    off = nsv * sum(w_[0] * w_[1] for w_ in wgrad)
                    view = wsp[off:off + nsv * K_ * Hp]
                    sparse_jobs[deneg_at] = self._job(t_, rows_, view, Hp, None, K_, Hp, n, nsv, K_ * Hp)
                    self.last_slabs["dEneg_slabs"] = (view, nsv, K_, Hp)
                    deneg_late = None
            else:
                call("seqrec_gemm_f32_grouped", len(wgrad), 0, 0, _lib.gemm_descs(wgrad), sk, ptr(wsp), st, tag="dW+dU")
        if deneg_late is not None:           # no launch to ride in after all: on its own, as before
            t_, rows_, dln_, Hd_, K_ = deneg_late
            dEn, ns_neg, ss_neg = self.gemm_slabs(0, 0, K_, Hp, n, dln_, K_, Hd_, Hp, "dEneg_slabs", self._splitk(K_, Hp, n), tag="dEneg")
            sparse_jobs[deneg_at] = self._job(t_, rows_, dEn, Hp, None, K_, Hp, n, ns_neg, ss_neg)
        if join_side:
            torch.cuda.current_stream(self.dev).wait_event(self._ev_join)
        return self._finish_step(d, sparse_jobs, wg_slabs, wcover, lrows, lr, eps, clipnorm, apply_update)

    def _finish_step(self, d, sparse_jobs, wg_slabs, wcover, lrows, lr, eps, clipnorm, apply_update):
        """Behind the cell: row scatter, global-norm clip, Adagrad (train_step and its one-call form _train_step_native)."""
        c, P = self.cfg, self.P
        Gd, tr = self.Gd, self.trainable
        n = d["n"]
        st = self._stream()
        self.last_counts = {"wgrad": wg_slabs[2] if wg_slabs is not None else 1}      # split-K slab counts of this step (bench.py's byte models)
        for j in sparse_jobs:
            if j.get("n_slabs", 0) > 1:
                self.last_counts["dX" if j["name"] == "E" and j["rows"] is d.get("ids") else "dEneg"] = j["n_slabs"]
        # ---- row-sparse contributions: one launch for (up to 4) scatter lists
        groups = [sparse_jobs[i:i + 4] for i in range(0, len(sparse_jobs), 4)]
        packed = [_lib.rows_jobs(g) for g in groups]
        if c.merge == "sorted":
            self._merge_sorted(sparse_jobs)
        else:
            for arr, cnt in packed:
                call("seqrec_rows_scatter_add_multi", arr, cnt, st)
        if self.priors:
            self._apply_priors(True)
        if not apply_update:
            return sparse_jobs
        # ---- global-norm clip over every trainable tensor (Keras clipnorm), then Adagrad
        dk = [k for k in Gd if tr[k]]
        clip = float(clipnorm if clipnorm else 0.0)
        if len(packed) <= 1 and len(dk) <= 8 and (dk or packed):
            # two launches: norm of everything, then scale + dense + row-sparse update; the two norm slots
            # alternate so that the slot of the NEXT step is cleared by this step's update launch
            gp = _lib.ptr_array([Gd[k] for k in dk]) if dk else None
            nn = _lib.i64_array([Gd[k].numel() for k in dk]) if dk else None
            arr, cnt = packed[0] if packed else (None, 0)
            cur, nxt = self._sq_slots[self._sq_par], self._sq_slots[1 - self._sq_par]
            if c.merge == "sorted":      # the norm without float atomics too: per-block partials added in index order
                mx = max([j["n"] for j in sparse_jobs] + [0])
                npart = int(_lib.load().seqrec_opt_sqnorm_ordered_floats(len(dk), cnt, mx))
                call("seqrec_opt_sqnorm_ordered", len(dk), gp, nn, arr, cnt, ptr(self.buf("sq_partials", npart)), npart, ptr(cur), 0,
                     ptr(lrows), n, ptr(self.loss_out), st)
            elif wg_slabs is not None:
                plain = [k for k in dk if k not in wcover]
                gpp = _lib.ptr_array([Gd[k] for k in plain]) if plain else None
                nnp = _lib.i64_array([Gd[k].numel() for k in plain]) if plain else None
                call("seqrec_opt_sqnorm_slabs", len(plain), gpp, nnp, wg_slabs[1], wg_slabs[0], wg_slabs[2], ptr(wg_slabs[3]), arr, cnt,
                     ptr(cur), ptr(lrows), n, ptr(self.loss_out), st)
            else:
                call("seqrec_opt_sqnorm", len(dk), gp, nn, arr, cnt, ptr(cur), ptr(lrows), n, ptr(self.loss_out), st)
            call("seqrec_opt_apply", len(dk), _lib.ptr_array([P[k] for k in dk]) if dk else None,
                 _lib.ptr_array([self.A[k] for k in dk]) if dk else None, gp, nn, arr, cnt, ptr(cur), clip, lr, eps,
                 ptr(self.scale), ptr(nxt), None, ptr(self.status), None, st)
            self._sq_par ^= 1
            self.sq = cur
        else:
            # more than 8 dense tensors (RNNFullModel with every side branch and bias) or more than 4 scatter lists: the multi-launch
            # form, the dense tensors in groups of 8 (the kernels' argument arrays) -- both merges (round 4: merge='sorted' raised here)
            self.sq = self.sq1
            self.sq.zero_()
            call("seqrec_loss_reduce", ptr(lrows), n, ptr(self.loss_out), st)
            chunks = [dk[i:i + 8] for i in range(0, len(dk), 8)]
            arrs = [(len(ch), _lib.ptr_array([P[k] for k in ch]), _lib.ptr_array([self.A[k] for k in ch]), _lib.ptr_array([Gd[k] for k in ch]),
                     _lib.i64_array([Gd[k].numel() for k in ch])) for ch in chunks]
            if c.merge == "sorted":      # ordered partial sums, added call by call (stream order): no float atomics
                mx = max([j["n"] for j in sparse_jobs] + [0])
                lib = _lib.load()
                npart = int(max(lib.seqrec_opt_sqnorm_ordered_floats(min(len(dk), 8), 0, 0), lib.seqrec_opt_sqnorm_ordered_floats(0, 4, mx)))
                pbuf = self.buf("sq_partials", npart)
                for cnt_d, _, _, gp, nn in arrs:
                    call("seqrec_opt_sqnorm_ordered", cnt_d, gp, nn, None, 0, ptr(pbuf), npart, ptr(self.sq), 1, None, 0, None, st)
                for arr, cnt in packed:
                    call("seqrec_opt_sqnorm_ordered", 0, None, None, arr, cnt, ptr(pbuf), npart, ptr(self.sq), 1, None, 0, None, st)
            else:
                for cnt_d, _, _, gp, nn in arrs:
                    call("seqrec_sqnorm_multi", cnt_d, gp, nn, ptr(self.sq), st)
                for arr, cnt in packed:
                    call("seqrec_rows_sqnorm_multi", arr, cnt, ptr(self.sq), st)
            call("seqrec_clip_scale", ptr(self.sq), clip, ptr(self.scale), st)
            for cnt_d, pp, pa, gp, nn in arrs:
                call("seqrec_adagrad_dense_multi", cnt_d, pp, pa, gp, nn, lr, eps, ptr(self.scale), st)
            for arr, cnt in packed:
                call("seqrec_rows_adagrad_multi", arr, cnt, lr, eps, ptr(self.scale), st)
        if tr["U"]:
            self.upack_dirty = True
        if c.x_to_y and c.diag_b and tr["Wxy"]:
            # Keras applies a kernel constraint AFTER the optimizer update: w *= mask (model.py:64)
            call("seqrec_mul", ptr(P["Wxy"]), ptr(self.diag_mask), ptr(P["Wxy"]), P["Wxy"].numel(), st)
        if self.priors:
            return self.loss_mean + self.reg_sum
        return self.loss_mean            # a VIEW of the engine's loss slot: read or consume it before the next step is enqueued

    # ------------------------------------------------------------------ the cell in ONE host call (seqrec_train_cell)
    def _native_cell_ok(self, negatives, apply_update):
        """The step shapes seqrec_train_cell covers: embedding input, sampled softmax over the engine's own negatives, every tensor
        trainable, no dropout / bias table / regularizer, atomic merge, the default slab forms.  Everything else -- and the per-kernel
        profile of bench.py, which times call by call -- takes the call-by-call sequence below, which is the specification."""
        c = self.cfg
        return (self.native_cell and _PROF is None and apply_update and negatives is None and c.input == "embed" and c.output == "sampled"
                and self.stepwise and self.sampler is not None and c.merge == "atomic" and not self.priors and not c.out_bias
                and c.drop_in == 0 and c.drop_out == 0 and c.drop_rec == 0 and all(self.trainable.values()) and c.use_bias
                and self._slab_scatter and self._slab_wgrad and not self._slab_dh and not self._overlap and self._fuse_prologue
                and self.Dp % 4 == 0)

    def _pair_ok(self, n, sk_h, sk_e):
        """dH and dEneg in one launch (seqrec_gemm_f32_pair)?  When both are small enough to share one round of the chip (the
        library decides the same way and would issue them one after the other otherwise) and the slab forms are on."""
        c, Hp, K = self.cfg, self.Hp, self.cfg.K
        nt0 = ((n + 63) // 64) * ((Hp + 63) // 64)
        nt1 = ((K + 63) // 64) * ((Hp + 63) // 64)
        return (self._pair_dh and c.output == "sampled" and self._slab_scatter and c.merge != "sorted" and not self._slab_dh and not self._overlap
                and K % 4 == 0 and nt0 * sk_h + nt1 * sk_e <= 1536)

    def _pair_plan(self, n, K, dln, Eneg, dHd, sk_h, Et, tgt_idx, dlt, Hd, sk_e):
        q = _lib.GemmPair()
        Hp = self.Hp
        q.a_kc0, q.b_kc0, q.M0, q.N0, q.K0, q.A0, q.lda0, q.B0, q.ldb0 = 1, 0, n, Hp, K, ptr(dln), K, ptr(Eneg), Hp
        q.C0, q.ldc0, q.splitk0, q.ws0 = ptr(dHd), Hp, sk_h, ptr(self.buf("gemm_ws", max(sk_h, 1) * n * Hp))
        q.add_table, q.add_index, q.add_scale, q.add_ld = ptr(Et), ptr(tgt_idx), ptr(dlt), Hp
        q.a_kc1, q.b_kc1, q.M1, q.N1, q.K1, q.A1, q.lda1, q.B1, q.ldb1 = 0, 0, K, Hp, n, ptr(dln), K, ptr(Hd), Hp
        q.splitk1, q.ws1 = max(sk_e, 1), ptr(self.buf("dEneg_slabs", max(sk_e, 1) * K * Hp))
        return q

    def _cell_plan(self):
        """The persistent argument block of seqrec_train_cell: everything that does not change from step to step is written once
        (the parameter tensors never move; set_sampler drops the block)."""
        c, P = self.cfg, self.P
        pl = self._plan
        if pl is not None:
            return pl
        pl = _lib.CellPlan()
        pl.cell, pl.act, pl.Hp, pl.H_real, pl.G, pl.K, pl.Dp = CELL[c.cell], ACT[c.act], self.Hp, c.H, self.G, c.K, self.Dp
        pl.seed = int(c.seed)
        pl.U, pl.upack = ptr(P["U"]), ptr(self.upack)
        th, al, lq = self.sampler
        Et = P["E"] if c.tied else P["Eout"]
        pl.thresh, pl.alias, pl.V, pl.sample_table = ptr(th), ptr(al), c.V_out, ptr(Et)
        pl.sample_logq = ptr(lq if c.logq else None)
        self._plan_keep = (th, al, lq)
        self._plan = pl
        return pl

    def _train_step_native(self, d, lr, eps, clipnorm, step):
        """train_step for the shapes of _native_cell_ok with the cell issued by ONE C-ABI call: the same launches with the same
        arguments as the sequence in train_step (bit-identical buffers; tests/test_gpu_engine.py), ~12 ctypes calls and their Python
        glue less per step."""
        import ctypes
        c, P, Gd = self.cfg, self.P, self.Gd
        n, T, B = d["n"], d["T"], d["B"]
        Hp, GHp, Dp, K = self.Hp, self.GHp, self.Dp, c.K
        st = self._cur_st
        tname = "E" if c.tied else "Eout"
        Et = P[tname]
        th, al, lq = self.sampler
        neg = self.buf("neg", K, dtype=torch.int32)
        Eneg = self.buf("Eneg", K, Hp)
        lq_neg = self.buf("lq_neg", K) if c.logq else None
        XW, Hout, gates, aux = self.buf("XW", n, GHp), self.buf("Hout", n, Hp), self.buf("gates", n, GHp), self.buf("aux", n, Hp)
        ln, dlt, lrows = self.buf("ln", n, K), self.buf("dlt", n), self.buf("loss_rows", n)
        dHd, dPre = self.buf("dHd", n, Hp), self.buf("dPre", n, GHp)
        scan_ws = self.buf("scan_ws", 2 * n * Hp)
        pl = self._cell_plan()
        pl.stages, pl.n, pl.T, pl.B, pl.use_graph = 7, n, T, B, int(self.use_graph)
        so = d["rb"].step_off
        pl.step_off_host = so.ctypes.data
        pl.step = int(step)
        pend = d.pop("_pending", None)
        pl.batch = 0
        if self.upack_dirty:
            pl.pack_u, pl.sample = 1, 1
            self.upack_dirty = False
            if pend is not None:                # the batch's own gather rides in the prologue launch: three openers, one launch
                ds_, sess_, _ = pend
                pl.batch, pl.flat, pl.starts, pl.sess_host = 1, ptr(ds_["flat"]), ptr(ds_["starts"]), sess_.ctypes.data
                pl.sess_out, pl.step_off_out = ptr(d["sess"]), ptr(d["step_off"])
                pl.ids_out, pl.tgt_out, pl.prev_out = ptr(d["ids"]), ptr(d["tgt"]), ptr(d["prev"])
                self._plan_keep_batch = (ds_, sess_)
        else:                                   # U frozen since the last pack: the negatives on their own
            if pend is not None:
                d["_pending"] = pend
                self._materialise(d)
            pl.pack_u = pl.sample = 0
            call("seqrec_sample_gather", int(c.seed), int(step), K, ptr(th), ptr(al), c.V_out, ptr(Et), Hp, ptr(lq if c.logq else None),
                 ptr(neg), ptr(Eneg), ptr(lq_neg), st)
        pl.neg_out, pl.Eneg_out, pl.lq_neg_out = ptr(neg), ptr(Eneg), ptr(lq_neg)
        pl.x_table, pl.x_ld, pl.x_index, pl.W, pl.bias = ptr(P["E"]), Dp, ptr(d["ids"]), ptr(P["W"]), ptr(P["b"])
        pl.XW, pl.Hout, pl.gates, pl.aux = ptr(XW), ptr(Hout), ptr(gates), ptr(aux)
        pl.Eneg, pl.neg, pl.lq_neg = ptr(Eneg), ptr(neg), ptr(lq_neg)
        pl.ln, pl.dlt, pl.loss_rows, pl.inv_denom = ptr(ln), ptr(dlt), ptr(lrows), 1.0 / n
        pl.tgt_table, pl.tgt_ld, pl.tgt_index, pl.tgt_ids, pl.lq_tgt = ptr(Et), Hp, ptr(d["tgt"]), ptr(d["tgt"]), None
        pl.logq_table = ptr(lq if c.logq else None)
        # split policy: exactly train_step's
        sk_h = self._splitk(n, Hp, K, fill=True)
        shapes = [(Hp, 2 * Hp), (Hp, Hp)] if c.cell == "gru" else [(Hp, GHp)]
        shapes += [(Dp, GHp), (1, GHp)]
        tiles = sum(((a + 63) // 64) * ((b + 63) // 64) for a, b in shapes)
        sk_w = self._splitk_tiles(tiles, n, fill=SPLITK_FILL_WGRAD, long_k=True)
        sk_e = self._splitk(K, Hp, n)
        pair = self._pair_ok(n, sk_h, sk_e)
        ride = (not pair and self._group_deneg and n <= 4096 and len(shapes) < 6
                and ((K + 63) // 64) * ((Hp + 63) // 64) + ((Dp + 63) // 64 + (Hp + 63) // 64 + 1) * ((GHp + 63) // 64) <= 512)
        if ride:
            tiles_all = tiles + ((K + 63) // 64) * ((Hp + 63) // 64)
            sk_w = max(sk_w, int(max(1, min(1024 // max(tiles_all, 1), n // SPLITK_MIN_K))))
        slabs_w = sk_w > 1 or ride
        wsz = sum(sk_w * a * b for a, b in shapes) + (sk_w * K * Hp if ride else 0)
        wsp = self.buf("gemm_ws", max(wsz, sk_h * n * Hp, 1))
        dEs = None if ride else self.buf("dEneg_slabs", max(sk_e, 1) * K * Hp)
        sk_x = self._splitk_tiles(((n + 63) // 64) * ((Dp + 63) // 64), GHp, min_k=self._slab_min_k, fill=True)
        dXs = self.buf("dX_slabs", max(sk_x, 1) * n * Dp)
        pl.dHd, pl.gemm_ws, pl.sk_dh = ptr(dHd), ptr(wsp), sk_h
        pl.deneg_mode, pl.sk_deneg, pl.dEneg_slabs = (3 if pair else (2 if ride else 1)), sk_e, ptr(dEs)
        pl.sk_wgrad, pl.wgrad_slabs, pl.wgrad_ws = sk_w, int(slabs_w), ptr(wsp)
        pl.dPre, pl.scan_ws, pl.prev = ptr(dPre), ptr(scan_ws), ptr(d["prev"])
        pl.dU, pl.dW, pl.db, pl.ones = ptr(Gd["U"]), ptr(Gd["W"]), ptr(Gd["b"]), ptr(self._ones(n))
        pl.sk_dx, pl.dX_slabs = sk_x, ptr(dXs)
        _raw_call("seqrec_train_cell", ctypes.addressof(pl), st)
        # ---- the scatter lists, as train_step builds them
        jobs = [self._job(tname, d["tgt"], Hout, Hp, dlt, n, Hp, 0)]
        if ride:
            nsv, off = int(pl.ns_wgrad), int(pl.deneg_off)
            view = wsp[off:off + nsv * K * Hp]
            jobs.append(self._job(tname, neg, view, Hp, None, K, Hp, n, nsv, K * Hp))
            self.last_slabs["dEneg_slabs"] = (view, nsv, K, Hp)
        else:
            jobs.append(self._job(tname, neg, dEs, Hp, None, K, Hp, n, int(pl.ns_deneg), K * Hp))
            self.last_slabs["dEneg_slabs"] = (dEs, int(pl.ns_deneg), K, Hp)
        jobs.append(self._job("E", d["ids"], dXs, Dp, None, n, Dp, (n + K) if c.tied else 0, int(pl.ns_dx), n * Dp))
        self.last_slabs["dX_slabs"] = (dXs, int(pl.ns_dx), n, Dp)
        wg = wcover = None
        if slabs_w:
            wg = (ctypes.addressof(pl.descs_out), int(pl.n_descs), int(pl.ns_wgrad), wsp)
            wcover = {"U", "W", "b"}
        return self._finish_step(d, jobs, wg, wcover or set(), lrows, lr, eps, clipnorm, True)

    def grads(self, d, step=0, negatives=None):
        """Debug/test hook: loss and UNPADDED gradients of one batch, no update applied.
        Table gradients come back dense (only sensible for small tables)."""
        n = d["n"]
        count = self.step_count
        self.train_step(d, step=step, negatives=negatives, apply_update=False)
        self.step_count = count
        out = {}
        for k in self.P:
            if not self.trainable[k]:
                continue
            out[k] = self.get_param(k, src=self.Gt if k in self.table_params else self.Gd)
        for k in self.table_params:
            self.Gt[k].zero_()
            self.slot[k].fill_(INT32_MAX)
        reg = float(self.reg_sum.item()) if self.priors else 0.0
        return float(self.loss_mean.item()) + reg, out

    # ------------------------------------------------------------------ evaluation / prediction
    def eval_loss(self, d, negatives=None, step=0):
        """Masked-token-mean CE of one batch (Keras test_function); device tensor."""
        if d["n"] == 0:
            return torch.zeros(1, device=self.dev)
        self.forward(d, train=False, step=step, negatives=negatives)
        if self.priors:
            self._apply_priors(False)
            return self.loss_mean + self.reg_sum
        return self.loss_mean.clone()

    def predict_rows(self, d):
        """Softmax probabilities per real token, [N_tok, V_out] (full softmax only)."""
        if self.cfg.output != "full":
            raise ValueError("dense probabilities exist only for output='full'; use rank_counts/recall for sampled models")
        dd = dict(d)
        dd.pop("tgt", None)
        r = self.forward(dd, train=False, want_probs=True)
        return r["probs"]

    def probs_from_hidden(self, H):
        """softmax(H . Wout + bout (+ byy)) for m given hidden rows [m, Hp] (device tensor): the output Keras' TimeDistributed
        Dense produces at a PAD step, where the masked scan carries the previous state and the unmasked y_to_y / x_to_y
        inputs are all-zero rows (only their biases remain).  Full softmax only."""
        c, P = self.cfg, self.P
        if c.output != "full":
            raise ValueError("dense probabilities exist only for output='full'")
        m = H.shape[0]
        st = self._stream()
        logits = self.buf("pad_logits", m, self.Vp)
        self.gemm(1, 0, m, c.V_out, self.Hp, H, self.Hp, P["Wout"], self.Vp, logits, self.Vp, bias=P.get("bout"), tag="logits")
        if c.y_to_y and c.yy_bias:
            zero = torch.zeros(m, dtype=torch.int32, device=self.dev)
            call("seqrec_gather_rows", ptr(P["byy"]), ptr(zero), ptr(logits), m, self.Vp, None, None, 1, st)     # += byy
        probs = torch.empty((m, c.V_out), dtype=torch.float32, device=self.dev)
        call("seqrec_full_softmax_ce", ptr(logits), self.Vp, None, m, c.V_out, 0.0, None, ptr(probs), st)
        return probs

    def hidden_rows(self, d):
        r = self.forward(d, train=False, stop_at_hidden=True)
        return r["Hd"]

    def topk_rows(self, d, k=20, rows=None, chunk=65536):
        """Top-k next items per token for sampled / tied output tables (the large-vocabulary form of
        ``predict``): -> (item ids int32 [m,k], scores float32 [m,k]), best first.  ``rows`` (int tensor /
        array of packed token indices, e.g. each session's last step) restricts the tokens that are
        scored.  Scores one chunk of items at a time (GEMM + running top-64 merge): no m x V matrix."""
        c, P = self.cfg, self.P
        if c.output != "sampled":
            raise ValueError("topk_rows is the catalogue-scale form; full-softmax models have predict_rows")
        if not 1 <= k <= 64:
            raise ValueError("1 <= k <= 64")
        Hd = self.hidden_rows(d)
        st = self._stream()
        if rows is not None:
            idx = torch.as_tensor(np.asarray(rows) if not torch.is_tensor(rows) else rows, dtype=torch.int32).to(self.dev)
            Hs = torch.empty((idx.numel(), self.Hp), dtype=torch.float32, device=self.dev)
            call("seqrec_gather_rows", ptr(Hd), ptr(idx), ptr(Hs), idx.numel(), self.Hp, None, None, 0, st)
            Hd = Hs
        m = Hd.shape[0]
        Et = P["E"] if c.tied else P["Eout"]
        sv = torch.full((max(m, 1), 64), float("-inf"), dtype=torch.float32, device=self.dev)
        si = torch.full((max(m, 1), 64), -1, dtype=torch.int32, device=self.dev)
        V = c.V_out
        chunk = int(min(chunk, V))
        sc = self.buf("topk_scores", m, chunk)
        for c0 in range(0, V, chunk):
            w = min(chunk, V - c0)
            self.gemm(1, 1, m, w, self.Hp, Hd, self.Hp, Et[c0:c0 + w], self.Hp, sc, chunk, tag="topk")
            call("seqrec_topk_merge", ptr(sc), chunk, m, w, c0, ptr(P.get("bout")), ptr(sv), ptr(si), st)
        out_v = torch.empty((m, k), dtype=torch.float32, device=self.dev)
        out_i = torch.empty((m, k), dtype=torch.int32, device=self.dev)
        call("seqrec_topk_finish", ptr(sv), ptr(si), m, k, ptr(out_v), ptr(out_i), st)
        return out_i, out_v

    def rank_counts(self, d):
        """rank[i] = number of items scoring strictly above the target of token i (sampled/tied
        output tables; score = h . Eout[v] + bout[v]).  Recall@K = mean(rank < K)."""
        c, P = self.cfg, self.P
        n = d["n"]
        Hd = self.hidden_rows(d)
        Et = P["E"] if c.tied else P["Eout"]
        rank = torch.zeros(n, dtype=torch.int32, device=self.dev)
        thr = self.buf("thr", n)
        call("seqrec_rank_count", ptr(Hd), self.Hp, ptr(Et), ptr(P.get("bout")), ptr(d["tgt"]), n, c.V_out, ptr(rank),
             ptr(thr), self._stream())
        return rank
