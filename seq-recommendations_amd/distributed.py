"""Multi-GPU form of the hot path: one process per GPU, torch.distributed over RCCL/xGMI.

Partitioning -- the reference is single-process, so this is the build's design:
  * sessions are data-parallel: every rank trains on its own batch of B sessions per step
    (weak scaling; global batch = B * R);
  * the cell weights W, U, b are replicated; their gradients are summed with ONE all-reduce of a
    flat bucket (<= 8.4 MB at c4) -- small, so it is a single latency-bound collective;
  * the item tables E and Eout are ROW-SHARDED, row r on rank r mod R at local index r // R
    (interleaved, so the Zipf head spreads evenly).  Rows are moved point-to-point with
    all-to-all (every peer pair has its own xGMI link, nothing rides a ring):
      forward   owners gather the requested rows and send them      (A2A of [rows, width] fp32)
      backward  row gradients travel the same routes in reverse and are scatter-added into the
                owner's gradient table; the owner alone runs the sparse Adagrad for its rows;
  * negatives are stratified by owner: every owner draws K/R negatives per requesting rank from its
    own shard (shard-local alias table), so the negative exchange has fixed sizes and needs no id
    round trip;  Q(v) = Q_shard(v) / R is the proposal used for the log-Q correction;
  * the global gradient norm is  sum_ranks |owned table-row grads|^2 + |dense grads|^2  (one small
    all-reduce), so every rank applies the same Keras clip scale.

Collectives per training step when E and Eout rows have the same padded width (every BASELINE
config: D == H) -- the "unified" path: THREE.  Both tables live in one local allocation, so one
routing plan covers input rows, target rows and the stratified negatives:
      forward   ONE all-to-all: [requested E rows | requested Eout rows | K/R negative rows | their ids]
      backward  ONE all-to-all of the matching row gradients (same routes reversed), one scatter list
      update    ONE all-reduce of [flat dense gradients | squared norm of the owned row gradients]
No step contains a host synchronisation, and neither does batch upload: the routing of MANY batches
(per-peer request counts, requested local rows) is exchanged by ShardedEngine.prepare() in two collectives
and ONE device -> host copy for the whole group of batches (an epoch, or a window of the batch stream), all
index arithmetic is host numpy and reaches the device as one blob per batch; the global token count rides in
the update all-reduce and divides the gradients on the device (seqrec_opt_apply grad_div).  The dense all-reduce
is issued asynchronously on a side stream / second communicator as soon as the weight gradients exist, so it
runs under the row-gradient exchange, the scatter and the dX GEMM (BASELINE config 5: "grad all-reduce overlap");
only two scalars (row-gradient norm, token count) are reduced on the critical path.  With D != H the tables keep
separate widths and the step uses one exchange per table (seven collectives, per-batch planning).

``RowExchange`` is device-agnostic torch code (unit-tested with gloo on CPU, world size 2 and 3);
``ShardedEngine`` wires it to the HIP kernels.
"""
import numpy as np
import torch

from . import _lib
from ._lib import ptr
from .engine import Engine, call, INT32_MAX, SPLITK_TARGET_WGS, _ceil4, _pad_h


class HostStagedDist:
    """torch.distributed look-alike that stages device tensors through the host and a gloo group.
    Debug / test transport only: RCCL refuses two ranks on one device, so the multi-rank logic of
    ShardedEngine and bench.py is exercised on a ONE-GPU box with several processes sharing cuda:0
    (tests/dist_gpu_worker.py, SEQREC_BENCH_BACKEND=gloo-staged).  Never used by the product path.
    Every device <-> host hop goes through PAGE-LOCKED staging tensors (no pageable multi-MB copies while four
    processes share the GPU), and with verify=True every host -> device hop is read back and compared: a transport
    that corrupts data says so itself instead of showing up as a wrong training result."""

    def __init__(self, dist, verify=False):
        self.d = dist
        self.ReduceOp = dist.ReduceOp
        self.verify = verify

    def _down(self, t):
        """device tensor -> page-locked host tensor (synchronous)"""
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t.detach())
        return h

    def _up(self, dst, h, what):
        """host tensor -> device tensor (synchronous), optionally verified by a read-back"""
        if not h.is_pinned():
            p = torch.empty(h.shape, dtype=h.dtype, pin_memory=True)
            p.copy_(h)
            h = p
        dst.copy_(h)
        if self.verify:
            back = torch.empty(h.shape, dtype=h.dtype, pin_memory=True)
            back.copy_(dst)
            same = back.view(torch.uint8).reshape(-1) == h.view(torch.uint8).reshape(-1) if h.numel() else torch.ones(0, dtype=torch.bool)
            if not bool(same.all()):
                bad = int((~same).sum())
                raise RuntimeError("HostStagedDist: host -> device copy of %s (%d bytes) arrived with %d differing bytes: the TEST "
                                   "TRANSPORT corrupted data (not the engine)" % (what, h.numel() * h.element_size(), bad))

    def get_world_size(self, group=None):
        return self.d.get_world_size()

    def get_rank(self, group=None):
        return self.d.get_rank()

    def all_to_all_single(self, out, inp, output_split_sizes=None, input_split_sizes=None, group=None):
        o = torch.zeros(out.shape, dtype=out.dtype, pin_memory=True)
        self.d.all_to_all_single(o, self._down(inp).contiguous(), output_split_sizes=output_split_sizes,
                                 input_split_sizes=input_split_sizes)
        self._up(out, o, "all_to_all result")

    class _Done:
        def wait(self):
            return True

    def new_group(self, *a, **k):
        return None                      # one gloo group: the staged transport is synchronous anyway

    def all_reduce(self, t, op=None, group=None, async_op=False):
        c = self._down(t)
        self.d.all_reduce(c, op=self.d.ReduceOp.SUM if op is None else op)
        self._up(t, c, "all_reduce result")
        return HostStagedDist._Done() if async_op else None

    def all_gather(self, outs, t, group=None):
        cs = [torch.zeros(o.shape, dtype=o.dtype, pin_memory=True) for o in outs]
        self.d.all_gather(cs, self._down(t))
        for o, c in zip(outs, cs):
            self._up(o, c, "all_gather result")

    def broadcast(self, t, src=0, group=None):
        c = self._down(t)
        self.d.broadcast(c, src=src)
        self._up(t, c, "broadcast result")

    def barrier(self, group=None):
        self.d.barrier()

    def destroy_process_group(self):
        self.d.destroy_process_group()


class RowPlan:
    """Routing of one list of global row ids (fixed per batch): who owns what, in which order."""
    __slots__ = ("n", "send_counts", "recv_counts", "perm", "inv_perm", "recv_local", "m")


class SegPlan:
    """Routing of one batch's row requests with `extra` owner-chosen rows per peer (see plan_seg).

    requester-side buffer (n_tot rows): segment j = [rows I asked rank j for | extra rows from j]
    owner-side buffer     (m_tot rows): segment j = [rows rank j asked me for | extra rows for j]
      req_split / own_split   per-peer row counts of the two layouts (python lists)
      req_pos   int32[n]      row of request i in the requester-side buffer
      req_extra int32[R,extra]  rows of the extras there
      own_rows  int32[m_tot]  owner-local row index of every owner-side row (-1 at the extras)
      own_extra int32[R,extra]  rows of the extras in the owner-side buffer
      back_src  int32[n_tot]  request index of every requester-side row (-1 at the extras)"""
    __slots__ = ("n", "extra", "n_tot", "m_tot", "req_split", "own_split", "req_pos", "req_extra", "own_rows",
                 "own_extra", "back_src", "host", "got_pad", "n_global")


class RowExchange:
    def __init__(self, dist, group=None, device="cpu"):
        self.dist, self.group = dist, group
        self.R = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.dev = torch.device(device)

    # -- plan ------------------------------------------------------------------------------------
    def plan(self, ids):
        """ids: int tensor [n] of GLOBAL row ids this rank needs.  Collective (tiny): exchanges the
        per-peer counts and the requested local row indices.  Call at batch-upload time."""
        R, dist = self.R, self.dist
        ids = ids.to(self.dev).long()
        n = ids.numel()
        owner = ids % R
        perm = torch.argsort(owner, stable=True)
        send_counts = torch.bincount(owner, minlength=R)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        m = int(sum(rc))
        want = (ids[perm] // R).to(torch.int32)
        got = torch.empty(m, dtype=torch.int32, device=self.dev)
        dist.all_to_all_single(got, want, output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        p = RowPlan()
        p.n, p.m = n, m
        p.send_counts, p.recv_counts = sc, rc
        p.perm = perm.to(torch.int32)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(n, device=self.dev)
        p.inv_perm = inv.to(torch.int32)
        p.recv_local = got                      # local row index of every row peers asked me for
        return p

    def plan_seg(self, owner, want, extra=0):
        """Request i asks rank owner[i] for its local row want[i]; in addition every (owner, requester)
        pair moves `extra` rows chosen by the owner (stratified negatives).  Collective (tiny, at
        batch-upload time).  The same plan routes the gradients back (push_seg)."""
        R, dist, dev = self.R, self.dist, self.dev
        owner = owner.to(dev).long()
        want = want.to(dev).to(torch.int32)
        n = owner.numel()
        perm = torch.argsort(owner, stable=True)
        send_counts = torch.bincount(owner, minlength=R)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()
        m = int(sum(rc))
        got = torch.empty(m, dtype=torch.int32, device=dev)
        dist.all_to_all_single(got, want[perm].contiguous(), output_split_sizes=rc, input_split_sizes=sc, group=self.group)
        p = SegPlan()
        p.n, p.extra = n, extra
        p.req_split = [c + extra for c in sc]
        p.own_split = [c + extra for c in rc]
        p.n_tot, p.m_tot = n + R * extra, m + R * extra
        ar_r = torch.arange(R, device=dev)
        ar_e = torch.arange(extra, device=dev)
        # requester side
        req_pos = torch.empty(n, dtype=torch.long, device=dev)
        req_pos[perm] = torch.arange(n, device=dev) + extra * owner[perm]
        p.req_pos = req_pos.to(torch.int32)
        sc_end = torch.cumsum(send_counts, 0)
        p.req_extra = (sc_end[:, None] + extra * ar_r[:, None] + ar_e[None, :]).to(torch.int32)
        back = torch.full((p.n_tot,), -1, dtype=torch.int32, device=dev)
        back[req_pos] = torch.arange(n, device=dev, dtype=torch.int32)
        p.back_src = back
        # owner side
        seg = torch.repeat_interleave(ar_r, recv_counts)
        own_rows = torch.full((p.m_tot,), -1, dtype=torch.int32, device=dev)
        own_rows[torch.arange(m, device=dev) + extra * seg] = got
        p.own_rows = own_rows
        rc_end = torch.cumsum(recv_counts, 0)
        p.own_extra = (rc_end[:, None] + extra * ar_r[:, None] + ar_e[None, :]).to(torch.int32)
        return p

    def plan_seg_many(self, reqs, extra=0, device_fields=True, group="default", tokens=None):
        """plan_seg for M batches at once: reqs = [(owner int array [n_b], want int array [n_b]), ...] (host numpy).
        TWO collectives and ONE host synchronisation for all M batches instead of two + two per batch: the per-peer
        request counts of every batch travel in one all-to-all (then one device -> host copy), the requested local
        rows of every batch in a second one.  All index arithmetic is host numpy; each plan's tensors are created
        on self.dev.  Returns the same SegPlan objects plan_seg would build batch by batch.  device_fields=False leaves
        the index tensors to the caller (plan.host holds them as numpy, plan.got_pad the received rows + a -1 sentinel:
        own_rows = got_pad[host['own_src']]), so that ONE blob per batch crosses PCIe (ShardedEngine.prepare).
        tokens (optional, one int per batch): this rank's token count of every batch; the counts of all ranks ride in the
        same first collective and plan.n_global is their sum -- the step's gradient divisor, known on the host before the
        step is enqueued."""
        R, dist, dev = self.R, self.dist, self.dev
        grp = self.group if isinstance(group, str) else group          # planning may run on its own communicator
        M = len(reqs)
        owners = [np.asarray(o, dtype=np.int64) for o, _ in reqs]
        wants = [np.asarray(w, dtype=np.int32) for _, w in reqs]
        perms = [np.argsort(o, kind="stable") for o in owners]
        SC = np.stack([np.bincount(o, minlength=R) for o in owners]).astype(np.int64) if M else np.zeros((0, R), np.int64)
        # counts: row j of the send matrix goes to peer j -> I receive, from peer i, its counts towards me per batch
        tk = np.zeros(M, np.int64) if tokens is None else np.asarray(tokens, dtype=np.int64)
        sc_ext = np.concatenate([SC.T, np.tile(tk[None, :], (R, 1))], axis=1)            # [R, 2M]: counts towards peer j | my tokens
        sc_dev = torch.from_numpy(np.ascontiguousarray(sc_ext)).to(dev)
        rc_dev = torch.empty_like(sc_dev)
        dist.all_to_all_single(rc_dev, sc_dev, group=grp)
        rc_ext = rc_dev.cpu().numpy()                                                    # (the ONE host sync)
        RC = rc_ext[:, :M].T.copy()                                                      # [M, R]
        n_global = rc_ext[:, M:].sum(axis=0)                                             # [M] tokens of every rank, summed
        # wants: for peer j the concatenation over batches of the local rows I ask it for
        segs = [[wants[b][perms[b]][SC[b, :j].sum():SC[b, :j + 1].sum()] for b in range(M)] for j in range(R)]
        send = np.concatenate([x for j in range(R) for x in segs[j]]) if M else np.zeros(0, np.int32)
        in_split = [int(SC[:, j].sum()) for j in range(R)]
        out_split = [int(RC[:, i].sum()) for i in range(R)]
        got_all = torch.empty(int(sum(out_split)), dtype=torch.int32, device=dev)
        dist.all_to_all_single(got_all, torch.from_numpy(send.astype(np.int32)).to(dev), output_split_sizes=out_split,
                               input_split_sizes=in_split, group=grp)
        got_pad = torch.cat([got_all, torch.full((1,), -1, dtype=torch.int32, device=dev)])
        peer_off = np.concatenate([[0], np.cumsum(out_split)])[:-1]                      # start of peer i's block in got_all
        within = np.cumsum(RC, axis=0) - RC                                              # [M, R] offset of batch b inside peer i's block
        plans = []
        ar_e = np.arange(extra, dtype=np.int64)
        for b in range(M):
            n = owners[b].shape[0]
            sc, rc = SC[b], RC[b]
            m = int(rc.sum())
            p = SegPlan()
            p.n, p.extra = n, extra
            p.req_split = [int(c) + extra for c in sc]
            p.own_split = [int(c) + extra for c in rc]
            p.n_tot, p.m_tot = n + R * extra, m + R * extra
            perm = perms[b]
            req_pos = np.empty(n, np.int64)
            req_pos[perm] = np.arange(n) + extra * owners[b][perm]
            sc_end, rc_end = np.cumsum(sc), np.cumsum(rc)
            req_extra = sc_end[:, None] + extra * np.arange(R)[:, None] + ar_e[None, :]
            back = np.full(p.n_tot, -1, np.int64)
            back[req_pos] = np.arange(n)
            seg = np.repeat(np.arange(R), rc)
            own_pos = np.arange(m) + extra * seg                                         # rows of the requests in the owner-side buffer
            src = np.concatenate([peer_off[i] + within[b, i] + np.arange(rc[i]) for i in range(R)]) if m else np.zeros(0, np.int64)
            own_extra = rc_end[:, None] + extra * np.arange(R)[:, None] + ar_e[None, :]
            # owner-side rows as ONE gather index into [got_all | -1]: requests take their received local row, extras -1
            own_src = np.full(p.m_tot, got_all.numel(), np.int64)
            own_src[own_pos] = src
            p.host = dict(req_pos=req_pos, req_extra=req_extra, back=back, own_extra=own_extra, own_src=own_src)
            if device_fields:
                i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
                p.req_pos, p.req_extra, p.back_src, p.own_extra = i32(req_pos), i32(req_extra), i32(back), i32(own_extra)
                p.own_rows = got_pad[torch.from_numpy(own_src).to(dev)]
            else:
                p.req_pos = p.req_extra = p.back_src = p.own_extra = p.own_rows = None
            p.got_pad = got_pad
            p.n_global = int(n_global[b])
            plans.append(p)
        return plans

    def plan_unified(self, rbs, V_in, tied, Kr, nid, w, lq_host=None, group="default", put_fill=None):
        return self.plan_unified_end(self.plan_unified_begin(rbs, V_in, tied, Kr, nid, w, lq_host=lq_host, group=group, put_fill=put_fill))

    def plan_unified_begin(self, rbs, V_in, tied, Kr, nid, w, lq_host=None, group="default", put_fill=None):
        """The routing of M batches of the UNIFIED step (requests = [input rows ; target rows], `Kr + nid` owner-chosen rows per
        peer pair), planned by the native host routines of csrc/route.hip: same two collectives and one host sync as
        plan_seg_many, one pass per batch instead of ~60 numpy operations (0.7 ms of host time per batch became the
        bottleneck of a 0.55 ms step).  Returns per batch (blob, parts, plan): `blob` the batch's int32 index block -- the blocks
        of ALL M batches are written into ONE upload by put_fill(count, dtype, fill) (PinnedRing.put_fill; default: a plain tensor
        on self.dev) and cross PCIe in one copy; `parts` the (name, offset, length) list of its fields, `plan` a SegPlan with the
        split sizes fetch_seg / push_seg need.  Round 4: the host no longer touches the received request list -- the blob's
        `own_src` field is seqrec_exchange_pack's `kinds` (an index into plan.got_pad for requested rows, -1 / -2 at the id /
        negative rows) -- and the three small transfers go through page-locked memory (a pageable copy blocks the host until
        the DMA engine has served it, behind whatever the GPU is busy with)."""
        import ctypes
        R, dist, dev = self.R, self.dist, self.dev
        grp = self.group if isinstance(group, str) else group
        lib = _lib.load()
        M = len(rbs)
        extra = Kr + nid
        i32p, i64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
        P32 = lambda a: a.ctypes.data_as(i32p)
        P64 = lambda a: a.ctypes.data_as(i64p)
        ids = [np.ascontiguousarray(rb.ids, dtype=np.int32) for rb in rbs]
        tgt = [np.ascontiguousarray(rb.tgt, dtype=np.int32) for rb in rbs]
        SC = np.zeros((M, R), np.int64)
        for b in range(M):
            _lib.check(lib.seqrec_route_count_host(P32(ids[b]), P32(tgt[b]), rbs[b].n_tok, R, P64(SC[b])), "seqrec_route_count_host")
        tk = np.array([rb.n_tok for rb in rbs], np.int64)
        sc_ext = np.ascontiguousarray(np.concatenate([SC.T, np.tile(tk[None, :], (R, 1))], axis=1)) if M else np.zeros((R, 0), np.int64)

        def up(arr):                                   # host array -> device tensor, page-locked when the caller gave us its ring
            if put_fill is None or arr.size == 0:
                return torch.from_numpy(arr).to(dev)
            flat_ = arr.reshape(-1)

            def f(dst):
                dst[:] = flat_
            return put_fill(flat_.size, arr.dtype, f).view(arr.shape)
        sc_dev = up(sc_ext)
        rc_dev = torch.empty_like(sc_dev)
        dist.all_to_all_single(rc_dev, sc_dev, group=grp)
        # ---- first half ends here: the count exchange is queued, its result on the way to page-locked memory.  Nothing has waited.
        hv = ev = None
        if dev.type == "cuda":
            hb = getattr(self, "_rc_host", None)
            if hb is None or hb.numel() < rc_dev.numel():
                hb = self._rc_host = torch.empty(max(rc_dev.numel(), 4096), dtype=torch.int64).pin_memory()
            self._rc_slot = (getattr(self, "_rc_slot", 0) + 1) % 2       # two windows may be between begin and end
            half = hb.numel() // 2
            if rc_dev.numel() > half:
                hb = self._rc_host = torch.empty(4 * rc_dev.numel(), dtype=torch.int64).pin_memory()
                half = hb.numel() // 2
            hv = hb[self._rc_slot * half: self._rc_slot * half + rc_dev.numel()].view(rc_dev.shape)
            hv.copy_(rc_dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
        return dict(rbs=rbs, V_in=V_in, tied=tied, Kr=Kr, nid=nid, w=w, lq_host=lq_host, grp=grp, put_fill=put_fill, up=up, ids=ids, tgt=tgt, SC=SC,
                    rc_dev=rc_dev, hv=hv, ev=ev, M=M, extra=extra)

    def plan_unified_end(self, hd):
        """Second half of plan_unified: wait for the count exchange queued by plan_unified_begin (the ONE host sync -- none at all when a few
        training steps were enqueued in between), exchange the requested rows, write and upload the window's index blocks."""
        import ctypes
        import time as _time
        R, dist, dev = self.R, self.dist, self.dev
        lib = _lib.load()
        rbs, V_in, tied, Kr, nid, w, lq_host, grp, put_fill, up = (hd[k] for k in ("rbs", "V_in", "tied", "Kr", "nid", "w", "lq_host", "grp", "put_fill", "up"))
        ids, tgt, SC, rc_dev, M, extra = hd["ids"], hd["tgt"], hd["SC"], hd["rc_dev"], hd["M"], hd["extra"]
        i32p, i64p = ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int64)
        P32 = lambda a: a.ctypes.data_as(i32p)
        P64 = lambda a: a.ctypes.data_as(i64p)
        _t0 = _time.perf_counter()
        if hd["ev"] is not None:
            hd["ev"].synchronize()                                                       # (the ONE host sync)
            rc_ext = hd["hv"].numpy().copy()
        else:
            rc_ext = rc_dev.cpu().numpy()
        self.sync_wait_s = getattr(self, "sync_wait_s", 0.0) + (_time.perf_counter() - _t0)   # how long the host stood here (bench.py reports it)
        RC = np.ascontiguousarray(rc_ext[:, :M].T)                                       # [M, R]
        n_global = rc_ext[:, M:].sum(axis=0)
        in_split = SC.sum(axis=0)
        out_split = RC.sum(axis=0)
        base = (np.concatenate([[0], np.cumsum(in_split)])[:-1][None, :] + np.cumsum(SC, axis=0) - SC).astype(np.int64)   # [M, R]
        send = np.empty(int(in_split.sum()), np.int32)
        ranks = []
        for b in range(M):
            rr = np.empty(2 * rbs[b].n_tok, np.int32)
            _lib.check(lib.seqrec_route_fill_host(P32(ids[b]), P32(tgt[b]), rbs[b].n_tok, R, int(V_in), int(bool(tied)), P64(SC[b]),
                                                  P64(np.ascontiguousarray(base[b])), P32(send), P32(rr)), "seqrec_route_fill_host")
            ranks.append(rr)
        n_got = int(out_split.sum())
        got_pad = torch.empty(n_got + 1, dtype=torch.int32, device=dev)                 # [received rows | -1]
        got_pad[n_got:] = -1
        dist.all_to_all_single(got_pad[:n_got], up(send), output_split_sizes=[int(x) for x in out_split],
                               input_split_sizes=[int(x) for x in in_split], group=grp)
        got_off = (np.concatenate([[0], np.cumsum(out_split)])[:-1][None, :] + np.cumsum(RC, axis=0) - RC).astype(np.int64)   # [M, R]
        # ---- one upload for the whole window: batch b's block at word offs[b] (64-word aligned)
        metas, offs, cur = [], [], 0
        for b, rb in enumerate(rbs):
            n, T = rb.n_tok, rb.T
            m = int(RC[b].sum())
            n_tot, m_tot = 2 * n + R * extra, m + R * extra
            names = [("step_off", T + 1), ("prev", n), ("ids", n), ("tgt", n), ("neg_slots", R * Kr), ("id_rows", R * nid), ("take_in", n),
                     ("take_tgt", n), ("neg_rows", R * Kr), ("negid_idx", R * Kr), ("back_idx", n_tot), ("own_src", m_tot), ("ntok", 1)]
            if lq_host is not None:
                names.append(("lq_tgt", n))
            total = sum(c for _, c in names)
            metas.append((names, total, n_tot, m_tot))
            offs.append(cur)
            cur += (total + 63) // 64 * 64
        f32p = ctypes.POINTER(ctypes.c_float)

        def fill_all(dst):
            for b, rb in enumerate(rbs):
                names, total, _, _ = metas[b]
                n, T = rb.n_tok, rb.T
                so = np.ascontiguousarray(rb.step_off, dtype=np.int32)
                pv = np.ascontiguousarray(rb.prev, dtype=np.int32)
                lq = None if lq_host is None else np.ascontiguousarray(lq_host[tgt[b]], dtype=np.float32)
                scb, rcb, gob = np.ascontiguousarray(SC[b]), np.ascontiguousarray(RC[b]), np.ascontiguousarray(got_off[b])
                seg = dst[offs[b]: offs[b] + total]
                wrote = lib.seqrec_route_blob_host(P32(so), T, P32(pv), P32(ids[b]), P32(tgt[b]), n, R, Kr, nid, w, P64(scb), P64(rcb),
                                                   P32(ranks[b]), P64(gob), float(n_global[b]),
                                                   None if lq is None else lq.ctypes.data_as(f32p), seg.ctypes.data_as(i32p), total)
                if wrote != total:
                    raise _lib.SeqrecError("seqrec_route_blob_host wrote %d of %d words" % (wrote, total))
        if put_fill is not None and cur > 0:
            big = put_fill(cur, np.int32, fill_all)
        else:
            host = np.zeros(max(cur, 1), np.int32)
            fill_all(host)
            big = torch.from_numpy(host).to(dev)
        out = []
        for b, rb in enumerate(rbs):
            names, total, n_tot, m_tot = metas[b]
            blob = big[offs[b]: offs[b] + total]
            parts, o = [], 0
            for name, c in names:
                parts.append((name, o, c))
                o += c
            p = SegPlan()
            p.n, p.extra, p.n_tot, p.m_tot = 2 * rb.n_tok, extra, n_tot, m_tot
            p.req_split = [int(c) + extra for c in SC[b]]
            p.own_split = [int(c) + extra for c in RC[b]]
            p.req_pos = p.req_extra = p.back_src = p.own_extra = p.own_rows = p.host = None
            p.got_pad = got_pad
            p.n_global = int(n_global[b])
            out.append((blob, parts, p))
        return out

    def fetch_seg(self, plan, rows):
        """rows [m_tot, w] in the owner-side layout -> [n_tot, w] in the requester-side layout."""
        out = torch.empty((plan.n_tot,) + tuple(rows.shape[1:]), dtype=rows.dtype, device=self.dev)
        self.dist.all_to_all_single(out, rows, output_split_sizes=plan.req_split, input_split_sizes=plan.own_split,
                                    group=self.group)
        return out

    def push_seg(self, plan, grads):
        """grads [n_tot, w] in the requester-side layout -> [m_tot, w] in the owner-side layout."""
        out = torch.empty((plan.m_tot,) + tuple(grads.shape[1:]), dtype=grads.dtype, device=self.dev)
        self.dist.all_to_all_single(out, grads, output_split_sizes=plan.own_split, input_split_sizes=plan.req_split,
                                    group=self.group)
        return out

    # -- forward: fetch rows ------------------------------------------------------------------------
    def fetch(self, plan, gather_local, width, take):
        """gather_local(idx int32[m]) -> [m, width] rows of MY shard; returns [n, width] rows in the
        order of the ids given to plan().  take(src [n,width], idx int32[n]) -> src[idx]."""
        mine = gather_local(plan.recv_local)
        out = torch.empty((plan.n, width), dtype=mine.dtype, device=self.dev)
        self.dist.all_to_all_single(out, mine, output_split_sizes=plan.send_counts,
                                    input_split_sizes=plan.recv_counts, group=self.group)
        return take(out, plan.inv_perm)

    # -- backward: push row gradients to their owners -------------------------------------------------
    def push(self, plan, grads, take):
        """grads [n, width] in plan order -> ([m, width] contributions, their local rows int32[m])."""
        sorted_g = take(grads, plan.perm)
        out = torch.empty((plan.m, grads.shape[1]), dtype=grads.dtype, device=self.dev)
        self.dist.all_to_all_single(out, sorted_g, output_split_sizes=plan.recv_counts,
                                    input_split_sizes=plan.send_counts, group=self.group)
        return out, plan.recv_local

    # -- fixed-size exchange (stratified negatives) -----------------------------------------------------
    def swap_fixed(self, x):
        """x [R, k, ...]: slice j goes to rank j; returns [R, k, ...] with slice i from rank i."""
        out = torch.empty_like(x)
        self.dist.all_to_all_single(out, x.contiguous(), group=self.group)
        return out


def shard_rows(table, rank, R):
    """Rows r = rank, rank+R, ... of a global [V, w] array (numpy or torch)."""
    return table[rank::R]


def shard_size(V, rank, R):
    return (V - rank + R - 1) // R


class ShardedEngine(Engine):
    """Engine whose item tables hold only this rank's rows.  Single-rank groups degenerate to the
    plain engine arithmetic (same kernels, the exchanges become local copies)."""

    def __init__(self, cfg, device, dist, group=None):
        if cfg.input != "embed" or cfg.output != "sampled":
            raise ValueError("ShardedEngine shards item tables: it needs input='embed', output='sampled'")
        if cfg.out_bias and _ceil4(cfg.D) != _pad_h(cfg.H):
            raise NotImplementedError("ShardedEngine: the per-item output bias rides in the unified step (D == H after padding) only")
        if (cfg.drop_in or cfg.drop_out or cfg.drop_rec) and _ceil4(cfg.D) != _pad_h(cfg.H):
            raise NotImplementedError("ShardedEngine: dropout is wired into the unified step (D == H after padding) only")
        if cfg.merge != "atomic" and _ceil4(cfg.D) != _pad_h(cfg.H):
            raise NotImplementedError("ShardedEngine: the sorted (bitwise-reproducible) row-gradient merge covers the unified step (D == H after padding)")
        self.dist, self.group = dist, group
        self.R = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if cfg.K % self.R:
            raise ValueError("K=%d must be divisible by the world size %d (stratified negatives)" % (cfg.K, self.R))
        self.V_global = cfg.V_out
        import dataclasses
        local = dataclasses.replace(cfg, V_in=shard_size(cfg.V_in, self.rank, self.R),
                                    V_out=shard_size(cfg.V_out, self.rank, self.R))
        Engine.__init__(self, local, device)
        # hipGraph replay of the step-wise scan was measured for this engine (round 2: 1.27 against 1.19 ms per step in steady
        # state on one rank) and is OFF here: the loop is host-bound by its collectives, not by the scan's launches
        self.use_graph = False
        # dropout masks are keyed by (local batch row, step): every rank draws from its own stream
        self.drop_seed = cfg.seed + 1000003 * (self.rank + 1)
        self.gcfg = cfg
        self.ex = RowExchange(dist, group, self.dev)
        # dense gradients live in ONE flat buffer (+1 slot for the squared norm of the owned row
        # gradients): the step's only all-reduce runs in place on it
        names = sorted(self.Gd)
        tot = sum(self.Gd[k].numel() for k in names)
        self.gflat = torch.zeros(tot + 2, dtype=torch.float32, device=self.dev)
        o = 0
        for k in names:
            nk = self.Gd[k].numel()
            self.Gd[k] = self.gflat[o:o + nk].view_as(self.Gd[k])
            o += nk
        self.n_dense = tot
        self.sq = self.gflat[tot:tot + 1]
        self.ntok = self.gflat[tot + 1:tot + 2]      # split path: global token count of the step (after the all-reduce)
        # unified path: [row-gradient norm, slot A (all-reduced) | dense norm (fixed order) | row-gradient norm, slot B | unused]
        self.norms = torch.zeros(4, dtype=torch.float32, device=self.dev)
        self._norm_par = 0
        # the dense all-reduce runs on its own stream and (RCCL) its own communicator, under the row-gradient exchange
        self.side = torch.cuda.Stream(device=self.dev)
        self.dense_group = group
        if hasattr(dist, "new_group"):                      # also on one rank: the communicator path is the one N > 1 uses
            g2 = dist.new_group()
            if g2 is not None:
                self.dense_group = g2
        self._dense_work = None
        # batch routing is planned on a third stream / communicator: ShardedEngine.prepare() then waits only for its own
        # two tiny collectives, not for the training steps already queued, and its host arithmetic overlaps them
        self.plan_stream = torch.cuda.Stream(device=self.dev)
        self.plan_group = group
        if hasattr(dist, "new_group"):
            g3 = dist.new_group()
            if g3 is not None:
                self.plan_group = g3
        # unified item table: E rows then Eout rows in one allocation (same padded width)
        self.unified = self.Dp == self.Hp
        if self.unified:
            w, P = self.Hp, self.P
            nE = P["E"].shape[0]
            nO = 0 if cfg.tied else P["Eout"].shape[0]
            self.off_out = 0 if cfg.tied else nE
            f32 = dict(dtype=torch.float32, device=self.dev)
            self.TT = torch.zeros(nE + nO, w, **f32)
            self.TA = torch.zeros(nE + nO, w, **f32)
            self.TG = torch.zeros(nE + nO, w, **f32)
            self.TS = torch.full((nE + nO,), INT32_MAX, dtype=torch.int32, device=self.dev)
            for name, lo, hi in (("E", 0, nE),) + ((("Eout", nE, nE + nO),) if not cfg.tied else ()):
                P[name] = self.TT[lo:hi]
                self.A[name] = self.TA[lo:hi]
                self.Gt[name] = self.TG[lo:hi]
                self.slot[name] = self.TS[lo:hi]

    # ---- device-side failures: every rank raises, or none does ----------------------------------------
    def check_status(self):
        """Engine.check_status over the GROUP: the status word and the cluster scans' error count are rank-local (a bad
        index from a peer, an exchange wait that ran out on one GPU), but a rank that raises alone leaves the others inside
        the next all-to-all -- under RCCL a hang until the watchdog fires, not a failure.  Both are all-reduced first (OR of
        the status bits as a MAX over the bit columns, MAX of the error count; it is a host sync anyway), so every rank
        decodes the same condition and raises together; the message names the ranks that reported it.  Collective: call it
        on every rank at the same point (epoch end, evaluation, read-back -- where Engine does)."""
        from .engine import status_messages
        lib = _lib.load()
        st = self._stream()
        bits = self.status.to(torch.int64)
        nerr = int(lib.seqrec_cluster_scan_errors(st))
        nb = 32
        cols = ((bits >> torch.arange(nb, device=self.dev)) & 1).to(torch.int64)          # [32] one column per status bit
        mine = torch.cat([cols, torch.tensor([nerr if nerr >= 0 else 1 << 40], dtype=torch.int64, device=self.dev)])
        every = [torch.empty_like(mine) for _ in range(self.R)]
        if self.R > 1:
            self.dist.all_gather(every, mine, group=self.group)
        else:
            every = [mine]
        table = torch.stack(every).cpu().numpy()                                            # [R, 33]
        gbits = 0
        for b in range(nb):
            if table[:, b].any():
                gbits |= 1 << b
        gerr = int(table[:, nb].max())
        msgs = status_messages(gbits, gerr if gerr < (1 << 40) else -1)
        if not msgs:
            return
        who = [int(r) for r in np.nonzero(table.any(axis=1))[0]]
        self.status.zero_()
        lib.seqrec_cluster_scan_errors_reset(st)
        if gerr:                                     # every rank switches, so the replicas keep issuing the same launches
            lib.seqrec_debug_scan_cluster(0)
            msgs.append("the cluster form of the scans is now OFF for this process (step-wise form from here on)")
        raise _lib.SeqrecError("device-side failure at or before training step %d on rank(s) %s of %d (raised on every rank): %s"
                               % (self.step_count, who, self.R, "; ".join(msgs)))

    # ---- helpers -----------------------------------------------------------------------------------
    def _neg_ones(self, n):
        t = self.ws.get("_neg_ones")
        if t is None or t.numel() < n:
            t = self.ws["_neg_ones"] = torch.full((max(n, 4096),), -1.0, dtype=torch.float32, device=self.dev)
        return t

    def _lq_tgt(self, d):
        """The per-target vector the CE kernel subtracts: log-Q of the targets, minus their output bias when the model has one."""
        return self._lq_tgt_adj if self.cfg.out_bias else d.get("lq_tgt")

    def _take(self, src, idx, out=None):
        if out is None:
            out = torch.empty((idx.numel(), src.shape[1]), dtype=src.dtype, device=self.dev)
        # index lists of the row exchange crossed PCIe and a collective: bounded gather (an index outside `src` reads a zero
        # row and sets SEQREC_STATUS_BAD_INDEX, raised by check_status) -- a stale index cannot pull arbitrary memory into a gradient
        call("seqrec_gather_rows_bounded", ptr(src), src.shape[0], ptr(idx), ptr(out), idx.numel(), src.shape[1], None, None, 0,
             ptr(self.status), self._stream(), prof_name="seqrec_gather_rows")
        return out

    def _gather_from(self, table):
        return lambda idx: self._take(table, idx)

    def set_sampler(self, thresh, alias, logq=None):
        """Shard-LOCAL alias table (proposal restricted to this rank's rows) and the log of the
        EFFECTIVE proposal Q(v) = Q_shard(v) / R for the local rows.  The per-item log-Q values are
        then replicated (V floats) so that no step needs a log-Q exchange."""
        Engine.set_sampler(self, thresh, alias, logq)
        self.logq_global = None
        if logq is not None:
            R = self.R
            mine = self.sampler[2]
            nmax = shard_size(self.V_global, 0, R)
            pad = torch.zeros(nmax, dtype=torch.float32, device=self.dev)
            pad[: mine.numel()] = mine
            allq = [torch.empty_like(pad) for _ in range(R)]
            if R > 1:
                self.dist.all_gather(allq, pad, group=self.group)
            else:
                allq = [pad]
            g = torch.zeros(self.V_global, dtype=torch.float32, device=self.dev)
            for j in range(R):
                g[j::R] = allq[j][: shard_size(self.V_global, j, R)]
            self.logq_global = g
            self.logq_global_host = g.cpu().numpy()

    def upload(self, rb):
        return self.prepare([rb])[0]

    def prepare(self, rbs):
        """Host RaggedBatches -> device batches incl. their routing.  Collective.  For the unified tables (every
        BASELINE config) the routing of ALL the batches costs two collectives and one device -> host copy
        (RowExchange.plan_seg_many), every derived index array is host numpy and one int32 blob per batch crosses
        PCIe: call it with an epoch's -- or a window's -- worth of batches and no training step waits for the host."""
        c, R = self.cfg, self.R
        for rb in rbs:
            if rb.n_tok == 0:
                raise ValueError("ShardedEngine: every rank needs at least one transition per step (collectives are unconditional)")
        ds = [Engine.upload(self, rb) for rb in rbs] if not self.unified else None
        if not self.unified:
            for d in ds:
                n = d["n"]
                nt = torch.tensor([n], dtype=torch.float64, device=self.dev)
                self.dist.all_reduce(nt, group=self.group)
                d["n_total"] = float(nt.item())
                if c.logq and self.logq_global is not None:
                    d["lq_tgt"] = self.logq_global[d["tgt"].long()]         # fixed per batch
                d["arange"] = torch.arange(n, device=self.dev, dtype=torch.int32)
                d["plan_in"] = self.ex.plan(d["ids"])
                d["plan_tgt"] = self.ex.plan(d["tgt"])
            return ds
        return self.prepare_end(self.prepare_begin(rbs))

    def prepare_begin(self, rbs):
        """First half of prepare() for the unified tables: the per-peer request counts of the window's batches are exchanged (queued on
        the planning stream, on their way to page-locked memory) and NOTHING waits.  Collective; every rank calls prepare_begin /
        prepare_end at the same points of its loop.  prepare_end a few training steps later finds the counts already there: the loop
        never stands still for the planner (round 4: up to 2 ms per window on a box whose DMA engine is slow to serve the copy)."""
        if not self.unified:
            return {"rbs": rbs, "split": True}
        for rb in rbs:
            if rb.n_tok == 0:
                raise ValueError("ShardedEngine: every rank needs at least one transition per step (collectives are unconditional)")
        c, R = self.cfg, self.R
        w, Kr = self.Hp, c.K // R
        nid = -(-Kr // w)                                            # rows that carry the negatives' ids
        lq_host = self.logq_global_host if (c.logq and self.logq_global is not None) else None
        with torch.cuda.stream(self.plan_stream):
            hd = self.ex.plan_unified_begin(rbs, self.gcfg.V_in, c.tied, Kr, nid, w, lq_host=lq_host, group=self.plan_group,
                                            put_fill=self.pinned.put_fill)
        return {"rbs": rbs, "hd": hd}

    def prepare_end(self, h):
        """Second half of prepare(): -> the device batches of the window begun with prepare_begin."""
        rbs = h["rbs"]
        if h.get("split"):
            return self.prepare(rbs)
        main = torch.cuda.current_stream(self.dev)
        with torch.cuda.stream(self.plan_stream):
            ds = self._prepare_unified(rbs, h["hd"])
            ready = torch.cuda.Event()
            ready.record(self.plan_stream)
        seen = set()
        for d in ds:
            d["ready"] = ready
            for t in (d["blob"], d["plan"].got_pad):
                key = t.untyped_storage().data_ptr()
                if key not in seen:              # (one storage per window: the blocks of all its batches, the received list)
                    seen.add(key)
                    t.record_stream(main)        # allocated on the planning stream, consumed on the training stream
        return ds

    def _prepare_unified(self, rbs, hd):
        """Routing of a window of batches by the native planner (RowExchange.plan_unified_begin / _end, csrc/route.hip): two
        collectives, one host sync, and the window's int32 index blocks written straight into the engine's page-locked upload ring."""
        planned = self.ex.plan_unified_end(hd)
        ds = []
        for rb, (blob, parts, plan) in zip(rbs, planned):
            d = {"n": rb.n_tok, "T": rb.T, "B": rb.B, "rb": rb, "blob": blob, "plan": plan, "n_global": plan.n_global}
            for name, o, cnt in parts:
                d[name] = blob[o:o + cnt]
            for k in ("lq_tgt", "ntok"):
                if k in d:
                    d[k] = d[k].view(torch.float32)
            # owner-side row kinds for seqrec_exchange_pack: an index into the received request list (plan.got_pad) for the rows the
            # peers asked for, -1 at the id rows, -2 at the rows of my draws -- written by the native planner, no device work here
            d["send_idx"] = d.pop("own_src")
            ds.append(d)
        return ds

    # ---- one training step ----------------------------------------------------------------------------
    def train_step(self, d, lr=0.01, eps=1e-8, clipnorm=1.0, step=None, negatives=None, apply_update=True):
        if step is None:
            step = self.step_count
        self.step_count = step + 1
        self._wait_ready(d)
        if self.unified:
            return self._step_unified(d, lr, eps, clipnorm, step, apply_update)
        return self._step_split(d, lr, eps, clipnorm, step, apply_update)

    def _cell_and_loss_split(self, d, X, Etgt, Eneg, neg, dX, dEtgt, dEneg, train=True, reduce_dense=True):
        """Everything between the two exchanges: x.W, scan, sampled softmax CE, BPTT, dense weight
        gradients; writes the three row-gradient blocks."""
        c, P = self.cfg, self.P
        st = self._stream()
        n, Hp, GHp, Dp, K = d["n"], self.Hp, self.GHp, self.Dp, c.K
        # unified path: gradients are left as SUMS; the global token count is only known on the device after the update's
        # all-reduce and divides them there (seqrec_opt_apply grad_div) -- no host round trip per batch
        inv = 1.0 / d["n_total"] if "n_total" in d else 1.0
        Gd = self.Gd
        lq_neg = None
        if c.logq:
            lq_neg = self.buf("lq_neg", K)
            call("seqrec_gather_rows", ptr(self.logq_global), ptr(neg), ptr(lq_neg), K, 1, None, None, 0, st)
        XW = self.buf("XW", n, GHp)
        self.gemm(1, 0, n, GHp, Dp, X, Dp, P["W"], GHp, XW, GHp, bias=P.get("b"), tag="xw")
        Hout = self.buf("Hout", n, Hp); gates = self.buf("gates", n, GHp); aux = self.buf("aux", n, Hp)
        self._scan_fwd(d, XW, Hout, gates, aux)
        Hd = Hout
        ln = self.buf("ln", n, K)
        self.gemm(1, 1, n, K, Hp, Hd, Hp, Eneg, Hp, ln, K, tag="logits")
        # rows are remote: the softmax kernel takes the target rows and the candidates' log-Q as vectors
        dlt = self.buf("dlt", n)
        loss_rows = self.buf("loss_rows", n)
        call("seqrec_sampled_softmax_ce_rows", ptr(ln), K, ptr(Hd), Hp, ptr(Etgt), ptr(d.get("lq_tgt")), ptr(lq_neg),
             ptr(d["tgt"]), ptr(neg), n, K, inv, ptr(loss_rows), ptr(dlt), st)
        call("seqrec_loss_reduce", ptr(loss_rows), n, ptr(self.loss_out), st)
        if not train:
            return
        # -- backward
        ar = d["arange"]
        dHd = self.buf("dHd", n, Hp)
        self.gemm(1, 0, n, Hp, K, ln, K, Eneg, Hp, dHd, Hp, splitk=self._splitk(n, Hp, K, fill=True), tag="dH",
                  fuse=_lib.gemm_fuse(add_table=Etgt, add_index=ar, add_scale=dlt, add_ld=Hp))
        self.gemm(0, 0, K, Hp, n, ln, K, Hd, Hp, dEneg, Hp, splitk=self._splitk(K, Hp, n), tag="dEneg")
        call("seqrec_gather_rows", ptr(Hd), ptr(ar), ptr(dEtgt), n, Hp, ptr(dlt), None, 0, st)
        dPre = self.buf("dPre", n, GHp)
        self._scan_bwd(d, dHd, Hout, gates, aux, dPre)
        if c.cell == "gru":          # h_{t-1} = Hout read through the prev links inside the GEMM
            wgrad = [(Hp, 2 * Hp, n, Hout, Hp, dPre, GHp, Gd["U"], GHp, d["prev"]),
                     (Hp, Hp, n, aux, Hp, dPre[:, 2 * Hp:], GHp, Gd["U"][:, 2 * Hp:], GHp)]
        else:
            wgrad = [(Hp, GHp, n, Hout, Hp, dPre, GHp, Gd["U"], GHp, d["prev"])]
        wgrad.append((Dp, GHp, n, X, Dp, dPre, GHp, Gd["W"], GHp))
        if c.use_bias:                       # db = ones^T . dPre in the same grouped launch (M = 1)
            wgrad.append((1, GHp, n, self._ones(n), self.ONES_LD, dPre, GHp, Gd["b"], GHp))
        tiles = sum(((w_[0] + 63) // 64) * ((w_[1] + 63) // 64) for w_ in wgrad)
        sk = self._splitk_tiles(tiles, n, fill=True)
        wsp = self.buf("gemm_ws", sum(sk * w_[0] * w_[1] for w_ in wgrad)) if sk > 1 else None
        call("seqrec_gemm_f32_grouped", len(wgrad), 0, 0, _lib.gemm_descs(wgrad), sk, ptr(wsp), st, tag="dW+dU")
        if self.unified and reduce_dense:
            self._start_dense_allreduce()        # the dense gradients are final: reduce them under everything that follows
        self.gemm(1, 1, n, Dp, GHp, dPre, GHp, P["W"], GHp, dX, Dp, splitk=self._splitk(n, Dp, GHp, fill=True), tag="dX")

    def _cell_unified(self, d, recv, Eneg, neg, lq_neg, step, train=True, reduce_dense=True):
        """Everything between the two exchanges of the unified step -- the plain engine's kernel sequence (Engine.train_step) on
        rows that arrived by all-to-all: the input rows and the target rows are read THROUGH their positions in the receive
        buffer (gathered-A GEMMs, indexed CE, row-add epilogue of dH: no staging copy), dX and dEneg are left as split-K
        slabs for the gradient-routing launch, dropout as in the plain engine.  Returns what that launch needs."""
        c, P, Gd = self.cfg, self.P, self.Gd
        st = self._stream()
        n, Hp, GHp, Dp, K, w = d["n"], self.Hp, self.GHp, self.Dp, c.K, self.Hp
        from . import engine as _eng
        if (train and self.native_cell and self.stepwise and _eng._PROF is None and c.use_bias and c.drop_in == 0 and c.drop_out == 0 and c.drop_rec == 0
                and all(self.trainable.values()) and Dp % 4 == 0):
            return self._cell_unified_native(d, recv, Eneg, neg, lq_neg, reduce_dense)
        drops = self._drop_masks(d, step) if train else {}
        XW = self.buf("XW", n, GHp)
        xidx = d["take_in"]
        if "in" in drops:                    # input dropout acts on the rows themselves: materialise them once
            X = self.buf("X", n, Dp)
            self._take(recv, xidx, X)
            call("seqrec_mul", ptr(X), ptr(drops["in"]), ptr(X), n * Dp, st)
            xidx = None
            self.gemm(1, 0, n, GHp, Dp, X, Dp, P["W"], GHp, XW, GHp, bias=P.get("b"), tag="xw")
        else:
            X = recv
            self.gemm(1, 0, n, GHp, Dp, recv, w, P["W"], GHp, XW, GHp, bias=P.get("b"), tag="xw", fuse=_lib.gemm_fuse(a_index=xidx))
        Hout = self.buf("Hout", n, Hp); gates = self.buf("gates", n, GHp); aux = self.buf("aux", n, Hp)
        self._scan_fwd(d, XW, Hout, gates, aux, drops.get("rec"))
        Hd = Hout
        if "out" in drops:
            Hd = self.buf("Hd", n, Hp)
            call("seqrec_mul", ptr(Hout), ptr(drops["out"]), ptr(Hd), n * Hp, st)
        ln = self.buf("ln", n, K)
        self.gemm(1, 1, n, K, Hp, Hd, Hp, Eneg, Hp, ln, K, tag="logits")
        dlt = self.buf("dlt", n)
        loss_rows = self.buf("loss_rows", n)
        # gradients stay SUMS (inv_denom 1): the global token count divides them in the update (seqrec_opt_apply grad_div)
        call("seqrec_sampled_softmax_ce_rows_idx", ptr(ln), K, ptr(Hd), Hp, ptr(recv), w, ptr(d["take_tgt"]), ptr(self._lq_tgt(d)),
             ptr(lq_neg), ptr(d["tgt"]), ptr(neg), n, K, 1.0, ptr(loss_rows), ptr(dlt), st, prof_name="seqrec_sampled_softmax_ce")
        r = {"Hd": Hd, "dlt": dlt, "loss_rows": loss_rows, "Hout": Hout}
        if not train:
            call("seqrec_loss_reduce", ptr(loss_rows), n, ptr(self.loss_out), st)
            return r
        # -- backward
        dHd = self.buf("dHd", n, Hp)
        self.gemm(1, 0, n, Hp, K, ln, K, Eneg, Hp, dHd, Hp, splitk=self._splitk(n, Hp, K, fill=True), tag="dH",
                  fuse=_lib.gemm_fuse(add_table=recv, add_index=d["take_tgt"], add_scale=dlt, add_ld=w))
        r["dEneg"] = self.gemm_slabs(0, 0, K, Hp, n, ln, K, Hd, Hp, "dEneg_slabs", self._splitk(K, Hp, n), tag="dEneg")
        if "out" in drops:
            call("seqrec_mul", ptr(dHd), ptr(drops["out"]), ptr(dHd), n * Hp, st)
        dPre = self.buf("dPre", n, GHp)
        self._scan_bwd(d, dHd, Hout, gates, aux, dPre, drops.get("rec"))
        wgrad = []
        if "rec" in drops:
            # dU_g = (A_g * m_g)^T . dPre_g with the gate's time-invariant mask expanded to tokens (Engine.train_step)
            B_ = d["B"]
            d["rb"].ensure_tokens()
            rowidx = self.buf("tok_row", n, dtype=torch.int32)
            rowidx.copy_(self.pinned.put(d["rb"].tok_row.astype(np.int32)))
            Hprev = self.buf("Hprev", n, Hp)
            call("seqrec_gather_rows", ptr(Hout), ptr(d["prev"]), ptr(Hprev), n, Hp, None, None, 0, st)
            mt = self.buf("mask_tok", n, Hp); Am = self.buf("A_masked", n, Hp)
            for g in range(self.G):
                call("seqrec_gather_rows", ptr(drops["rec"][g * B_:(g + 1) * B_]), ptr(rowidx), ptr(mt), n, Hp, None, None, 0, st)
                src = aux if (c.cell == "gru" and g == 2) else Hprev
                call("seqrec_mul", ptr(src), ptr(mt), ptr(Am), n * Hp, st)
                self.gemm(0, 0, Hp, Hp, n, Am, Hp, dPre[:, g * Hp:], GHp, Gd["U"][:, g * Hp:], GHp, splitk=self._splitk(Hp, Hp, n), tag="dU")
        elif c.cell == "gru":            # h_{t-1} = Hout read through the prev links inside the GEMM
            wgrad += [(Hp, 2 * Hp, n, Hout, Hp, dPre, GHp, Gd["U"], GHp, d["prev"]),
                      (Hp, Hp, n, aux, Hp, dPre[:, 2 * Hp:], GHp, Gd["U"][:, 2 * Hp:], GHp)]
        else:
            wgrad.append((Hp, GHp, n, Hout, Hp, dPre, GHp, Gd["U"], GHp, d["prev"]))
        if xidx is None:
            wgrad.append((Dp, GHp, n, X, Dp, dPre, GHp, Gd["W"], GHp))
        else:                               # the input rows read through their positions in the receive buffer
            wgrad.append((Dp, GHp, n, recv, w, dPre, GHp, Gd["W"], GHp, xidx))
        if c.use_bias:                       # db = ones^T . dPre in the same grouped launch (M = 1)
            wgrad.append((1, GHp, n, self._ones(n), self.ONES_LD, dPre, GHp, Gd["b"], GHp))
        tiles = sum(((w_[0] + 63) // 64) * ((w_[1] + 63) // 64) for w_ in wgrad)
        sk = self._splitk_tiles(tiles, n, fill=True)
        wsp = self.buf("gemm_ws", sum(sk * w_[0] * w_[1] for w_ in wgrad)) if sk > 1 else None
        call("seqrec_gemm_f32_grouped", len(wgrad), 0, 0, _lib.gemm_descs(wgrad), sk, ptr(wsp), st, tag="dW+dU")
        if reduce_dense:
            self._start_dense_allreduce()        # the dense gradients are final: reduce them under everything that follows
        if "in" in drops:
            dX = self.buf("dX", n, Dp)
            self.gemm(1, 1, n, Dp, GHp, dPre, GHp, P["W"], GHp, dX, Dp, splitk=self._splitk(n, Dp, GHp, fill=True), tag="dX")
            call("seqrec_mul", ptr(dX), ptr(drops["in"]), ptr(dX), n * Dp, st)
            r["dX"] = (dX, 1, n * Dp)
        else:
            sk_x = self._splitk_tiles(((n + 63) // 64) * ((Dp + 63) // 64), GHp, min_k=self._slab_min_k, fill=True)
            r["dX"] = self.gemm_slabs(1, 1, n, Dp, GHp, dPre, GHp, P["W"], GHp, "dX_slabs", sk_x, tag="dX")
        return r

    def _cell_unified_native(self, d, recv, Eneg, neg, lq_neg, reduce_dense):
        """_cell_unified (training, no dropout) with its launches issued by seqrec_train_cell: stages 1 + 2 (forward ... weight
        gradients) in one call, the dense all-reduce started from here, stage 4 (dX slabs) in a second call.  Same launches, same
        arguments, same buffers as the call-by-call sequence above."""
        import ctypes
        from .engine import _raw_call
        from ._lib import CELL, ACT
        c, P, Gd = self.cfg, self.P, self.Gd
        n, Hp, GHp, Dp, K, w = d["n"], self.Hp, self.GHp, self.Dp, c.K, self.Hp
        st = self._cur_st
        pl = self._plan
        if pl is None:
            pl = self._plan = _lib.CellPlan()
            pl.cell, pl.act, pl.Hp, pl.H_real, pl.G, pl.K, pl.Dp = CELL[c.cell], ACT[c.act], Hp, c.H, self.G, K, Dp
            pl.U, pl.upack, pl.W, pl.bias = ptr(P["U"]), ptr(self.upack), ptr(P["W"]), ptr(P["b"])
            pl.dU, pl.dW, pl.db = ptr(Gd["U"]), ptr(Gd["W"]), ptr(Gd["b"])
            pl.inv_denom, pl.deneg_mode, pl.wgrad_slabs, pl.sample = 1.0, 1, 0, 0
        XW, Hout, gates, aux = self.buf("XW", n, GHp), self.buf("Hout", n, Hp), self.buf("gates", n, GHp), self.buf("aux", n, Hp)
        ln, dlt, lrows = self.buf("ln", n, K), self.buf("dlt", n), self.buf("loss_rows", n)
        dHd, dPre = self.buf("dHd", n, Hp), self.buf("dPre", n, GHp)
        scan_ws = self.buf("scan_ws", 2 * n * Hp)
        shapes = ([(Hp, 2 * Hp), (Hp, Hp)] if c.cell == "gru" else [(Hp, GHp)]) + [(Dp, GHp), (1, GHp)]
        tiles = sum(((a + 63) // 64) * ((b + 63) // 64) for a, b in shapes)
        sk_w = self._splitk_tiles(tiles, n, fill=True)
        sk_h = self._splitk(n, Hp, K, fill=True)
        sk_e = self._splitk(K, Hp, n)
        sk_x = self._splitk_tiles(((n + 63) // 64) * ((Dp + 63) // 64), GHp, min_k=self._slab_min_k, fill=True)
        wsp = self.buf("gemm_ws", max(sk_w * sum(a * b for a, b in shapes) if sk_w > 1 else 1, sk_h * n * Hp, 1))
        dEs = self.buf("dEneg_slabs", max(sk_e, 1) * K * Hp)
        dXs = self.buf("dX_slabs", max(sk_x, 1) * n * Dp)
        pl.stages, pl.n, pl.T, pl.B, pl.use_graph = 3, n, d["T"], d["B"], int(self.use_graph)
        pl.step_off_host = d["rb"].step_off.ctypes.data
        pl.pack_u = int(self.upack_dirty)
        self.upack_dirty = False
        pl.x_table, pl.x_ld, pl.x_index = ptr(recv), w, ptr(d["take_in"])
        pl.XW, pl.Hout, pl.gates, pl.aux = ptr(XW), ptr(Hout), ptr(gates), ptr(aux)
        pl.Eneg, pl.neg, pl.lq_neg = ptr(Eneg), ptr(neg), ptr(lq_neg)
        pl.ln, pl.dlt, pl.loss_rows = ptr(ln), ptr(dlt), ptr(lrows)
        pl.tgt_table, pl.tgt_ld, pl.tgt_index, pl.tgt_ids, pl.lq_tgt = ptr(recv), w, ptr(d["take_tgt"]), ptr(d["tgt"]), ptr(self._lq_tgt(d))
        pl.logq_table = None
        pl.dHd, pl.gemm_ws, pl.sk_dh, pl.sk_deneg, pl.dEneg_slabs = ptr(dHd), ptr(wsp), sk_h, sk_e, ptr(dEs)
        pl.deneg_mode = 3 if self._pair_ok(n, sk_h, sk_e) else 1      # dH and dEneg in one launch where neither fills the chip (engine.py)
        pl.sk_wgrad, pl.wgrad_ws = sk_w, ptr(wsp)
        pl.dPre, pl.scan_ws, pl.prev, pl.ones = ptr(dPre), ptr(scan_ws), ptr(d["prev"]), ptr(self._ones(n))
        pl.sk_dx, pl.dX_slabs = sk_x, ptr(dXs)
        _raw_call("seqrec_train_cell", ctypes.addressof(pl), st)
        if reduce_dense:
            self._start_dense_allreduce()        # the dense gradients are final: reduce them under everything that follows
        pl.stages = 4
        _raw_call("seqrec_train_cell", ctypes.addressof(pl), st)
        self.last_slabs["dEneg_slabs"] = (dEs, int(pl.ns_deneg), K, Hp)
        self.last_slabs["dX_slabs"] = (dXs, int(pl.ns_dx), n, Dp)
        return {"Hd": Hout, "dlt": dlt, "loss_rows": lrows, "Hout": Hout, "dEneg": (dEs, int(pl.ns_deneg), K * Hp),
                "dX": (dXs, int(pl.ns_dx), n * Dp)}

    def _dense_arrays(self):
        """(names, P / A / G pointer arrays, sizes, partial-sum floats) of the dense tensors: they never move, built once."""
        da = getattr(self, "_dense_arr", None)
        if da is None:
            dk = sorted(self.Gd)
            da = self._dense_arr = (dk, _lib.ptr_array([self.P[k] for k in dk]), _lib.ptr_array([self.A[k] for k in dk]),
                                    _lib.ptr_array([self.Gd[k] for k in dk]), _lib.i64_array([self.Gd[k].numel() for k in dk]),
                                    int(_lib.load().seqrec_opt_sqnorm_ordered_floats(len(dk), 0, 0)))
        return da

    def _start_dense_allreduce(self):
        """Unified path: the dense all-reduce AND the dense gradient norm run on the side stream (its own communicator under
        RCCL) while the main stream routes the row gradients; the norm is taken in a fixed order (per-block partials added
        in index order) from the all-reduced -- hence identical -- gradients, so it is bit-identical on every rank."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        main_st = self._cur_st                   # the engine's helpers launch on the remembered stream: restore it below
        with torch.cuda.stream(self.side):
            self.side.wait_event(ev)
            if getattr(self, "debug_capture", False):      # tests: this rank's OWN dense gradients, before the sum over the ranks
                self._local_dense = self.gflat[: self.n_dense].clone()
            work = self.dist.all_reduce(self.gflat[: self.n_dense], group=self.dense_group, async_op=True)
            if self.unified:
                if work is not None:
                    work.wait()                  # orders the side stream behind the collective (no host wait under RCCL)
                dk, _, _, gp, nn, npart = self._dense_arrays()
                call("seqrec_opt_sqnorm_ordered", len(dk), gp, nn, None, 0, ptr(self.buf("sq_partials_dense", npart)), npart,
                     ptr(self.norms[1:2]), 0, None, 0, None, self._stream())
                self._dense_work = None
            else:
                self._dense_work = work
        self._cur_st = main_st

    def _dense_update(self, lr, eps, clipnorm, rows_job=None, n_local=None):
        """After the owned row norms were added into self.sq.  Unified path (n_local given): the dense gradients are
        already being reduced on the side stream (_start_dense_allreduce); only [row-gradient norm, token count] is
        reduced here, then the dense norms go on top (identical on every rank) and ONE launch applies the Keras clip
        scale, the 1 / token-count division, the dense Adagrad and (rows_job = (jobs array, count)) the row-sparse
        Adagrad of the owned rows.  Split path: one all-reduce of [dense grads | sq]."""
        st = self._stream()
        P, Gd = self.P, self.Gd
        div = None
        if n_local is not None:
            call("seqrec_fill_f32", ptr(self.ntok), float(n_local), 1, st)
            self.dist.all_reduce(self.gflat[self.n_dense:], group=self.group)
            if self._dense_work is not None:
                self._dense_work.wait()
                self._dense_work = None
            torch.cuda.current_stream(self.dev).wait_stream(self.side)
            div = self.ntok
        else:
            self.dist.all_reduce(self.gflat[: self.n_dense + 1], group=self.group)
        dk = sorted(Gd)
        gp = _lib.ptr_array([Gd[k] for k in dk])
        nn = _lib.i64_array([Gd[k].numel() for k in dk])
        # the dense norm in a FIXED order (per-block partials added in index order, no float atomics): every rank adds the same
        # number to the (all-reduced, hence identical) row norm, so the clip scale -- and with it the replicated weights --
        # stay bit-identical across ranks; the float-atomic form differed in the last bit from rank to rank
        npart = int(_lib.load().seqrec_opt_sqnorm_ordered_floats(len(dk), 0, 0))
        call("seqrec_opt_sqnorm_ordered", len(dk), gp, nn, None, 0, ptr(self.buf("sq_partials", npart)), npart, ptr(self.sq), 1,
             None, 0, None, st)
        arr, cnt = rows_job if rows_job is not None else (None, 0)
        call("seqrec_opt_apply", len(dk), _lib.ptr_array([P[k] for k in dk]), _lib.ptr_array([self.A[k] for k in dk]), gp, nn,
             arr, cnt, ptr(self.sq), float(clipnorm if clipnorm else 0.0), lr, eps, ptr(self.scale), None, ptr(div),
             ptr(self.status), None, st)
        self.upack_dirty = True

    def grads(self, d, step=0, negatives=None):
        """Test hook (Engine.grads): this rank's loss and gradient CONTRIBUTIONS, normalised by the global token count
        like the trained quantities (the unified step keeps sums and divides on the device in the update)."""
        loss, out = Engine.grads(self, d, step=step, negatives=negatives)
        if self.unified:
            nt = float(d["n_global"])
            loss = float(self.loss_sum.item()) * self.R / nt
            out = {k: v / np.float32(nt) for k, v in out.items()}
        return loss, out

    def failure_report(self):
        """After check_status() raised: where the non-finite value sits.  For every buffer the last unified step wrote, the number
        of non-finite entries and the largest finite magnitude over the region the step used -- the first buffer of the step's
        dataflow with a non-zero count is where the failure entered (test / support aid; synchronises)."""
        if not self.unified or not hasattr(self, "_last"):
            return {}
        L, c = self._last, self.cfg
        n, K, w = L["n"], c.K, self.Hp

        def stat(t):
            t = t.detach().float().reshape(-1)
            fin = torch.isfinite(t)
            mx = float(t[fin].abs().max().item()) if bool(fin.any()) else 0.0
            return (int((~fin).sum().item()), mx)
        ws, out = self.ws, {}
        nsx = self.last_slabs.get("dX_slabs", (None, 1))[1]
        nsn = self.last_slabs.get("dEneg_slabs", (None, 1))[1]
        reg = {"sendbuf": ("sendbuf", 0, L["m_tot"] * w), "Eneg": ("Eneg", 0, K * w), "XW": ("XW", 0, n * self.GHp), "Hout": ("Hout", 0, n * w),
               "gates": ("gates", 0, n * self.GHp), "aux": ("aux", 0, n * w), "ln(dlogits)": ("ln", 0, n * K), "dlt": ("dlt", 0, n),
               "loss_rows": ("loss_rows", 0, n), "dHd": ("dHd", 0, n * w), "dPre": ("dPre", 0, n * self.GHp),
               "dX slabs": ("dX_slabs", 0, nsx * n * w), "dEneg slabs": ("dEneg_slabs", 0, nsn * K * w), "backbuf": ("backbuf", 0, L["n_tot"] * w)}
        for name, (buf, lo, hi) in reg.items():
            if buf in ws:
                out[name] = stat(ws[buf][lo:hi])
        out["gback"] = stat(L["gback"])
        for k in sorted(self.Gd):
            out["Gd[%s]" % k] = stat(self.Gd[k])
        loc = getattr(self, "_local_dense", None)
        if loc is not None:                  # the same tensors as this rank produced them (before the all-reduce)
            o = 0
            for k in sorted(self.Gd):
                nk = self.Gd[k].numel()
                out["local Gd[%s]" % k] = stat(loc[o:o + nk])
                if k == "U" and "Hout" in ws and "dPre" in ws and "prev" in L:
                    # recompute dU = Hout[prev]^T . dPre with torch and name the elements that differ
                    Hout = ws["Hout"][: n * w].view(n, w); dPre = ws["dPre"][: n * self.GHp].view(n, self.GHp)
                    prev = L["prev"].long()
                    A = torch.where((prev >= 0)[:, None], Hout[prev.clamp(min=0)], torch.zeros_like(Hout))
                    ref = A.double().t() @ dPre.double()
                    got = loc[o:o + nk].view(w, self.GHp).double()
                    diff = (got - ref).abs()
                    bad = torch.nonzero(diff > 1e-3 * (1.0 + ref.abs()))
                    out["local dU vs torch"] = {"n_bad": int(bad.shape[0]), "first_bad": bad[:6].tolist(),
                                                "rows": sorted(set((bad[:, 0] // 64).tolist()))[:8], "cols": sorted(set((bad[:, 1] // 64).tolist()))[:8],
                                                "max_bad_value": float(got[diff > 1e-3 * (1.0 + ref.abs())].abs().max().item()) if bad.shape[0] else 0.0}
                o += nk
        rows = L["send_idx"].long()
        rows = rows[rows >= 0]
        out["TG[touched rows]"] = stat(self.TG[rows])
        out["norms[rows A, dense, rows B]"] = [float(x) for x in self.norms[:3].tolist()]
        out["scale"] = float(self.scale.item())
        return out

    def _wait_ready(self, d):
        ev = d.get("ready")
        if ev is not None:
            torch.cuda.current_stream(self.dev).wait_event(ev)     # the batch's routing was produced on the planning stream

    def _global_tokens(self, d):
        """Device scalar: the step's token count summed over the ranks (evaluation paths; no host sync)."""
        nt = torch.full((1,), float(d["n"]), dtype=torch.float32, device=self.dev)
        self.dist.all_reduce(nt, group=self.group)
        return nt

    def _rows_in(self, d, step):
        """Forward exchange of the unified path (collective 1).  ONE launch packs the owner-side buffer (requested rows, this
        rank's stratified draws for every requester and their global ids: seqrec_exchange_pack), the all-to-all moves it, ONE
        launch lifts the K negative rows / ids / log-Q out of the receive buffer; input and target rows stay where they
        arrived.  -> (recv [n_tot, w], Eneg [K, w], neg int32[K], lq_neg or None, rows_eff int32[m_tot])."""
        c, R = self.cfg, self.R
        st = self._stream()
        w, K = self.Hp, c.K
        Kr = K // R
        nid = -(-Kr // w)
        plan = d["plan"]
        th, al, _ = self.sampler
        sendbuf = self.buf("sendbuf", plan.m_tot, w)
        rows_eff = self.buf("rows_eff", plan.m_tot, dtype=torch.int32)
        bias_out = bias_rows = None
        if c.out_bias:       # the per-item output bias (RNNBaseline's Dense bias, model.py:257) travels beside the rows: one float per row
            bias_out = self.buf("bias_send", plan.m_tot)
            bias_rows = self.buf("bias_rows", plan.m_tot, dtype=torch.int32)
        call("seqrec_exchange_pack", ptr(self.TT), self.TT.shape[0], w, ptr(d["send_idx"]), ptr(plan.got_pad), plan.got_pad.numel() - 1,
             plan.m_tot, int(c.seed), int(step) * R + self.rank,
             R * Kr, ptr(th), ptr(al), c.V_out, self.off_out, ptr(d["neg_slots"]), ptr(d["id_rows"]), R * nid, Kr, R, self.rank,
             ptr(sendbuf), ptr(rows_eff), ptr(self.status), ptr(self.P["bout"] if c.out_bias else None), ptr(bias_out), ptr(bias_rows), st)
        recv = self.ex.fetch_seg(plan, sendbuf)                                 # collective 1
        self._recv_bias = self.ex.fetch_seg(plan, bias_out) if c.out_bias else None      # (+ one all-to-all of n_tot floats with a bias)
        self._bias_rows = bias_rows
        Eneg = self.buf("Eneg", K, w)
        neg = self.buf("neg", K, dtype=torch.int32)
        lq_neg = self.buf("lq_neg", K) if (c.logq and self.logq_global is not None) else None
        call("seqrec_exchange_unpack", ptr(recv), w, ptr(d["neg_rows"]), ptr(d["negid_idx"]), K,
             ptr(self.logq_global if lq_neg is not None else None), self.V_global, ptr(Eneg), ptr(neg), ptr(lq_neg),
             ptr(self.status), st)
        if c.out_bias:
            # logit = h . e + bias - logQ: the CE kernels subtract one vector per candidate and one per target, so the bias enters as
            # (logQ - bias) -- gathered from the received bias values through the rows' positions, no kernel of its own
            rb_ = self._recv_bias
            none = self._neg_ones(max(K, d["n"]))
            adj_n = self.buf("lqb_neg", K)
            adj_t = self.buf("lqb_tgt", d["n"])
            for adj, base, idx, m in ((adj_n, lq_neg, d["neg_rows"], K), (adj_t, d.get("lq_tgt"), d["take_tgt"], d["n"])):
                if base is not None:
                    adj.copy_(base)
                else:
                    adj.zero_()
                call("seqrec_gather_rows_bounded", ptr(rb_), rb_.numel(), ptr(idx), ptr(adj), m, 1, ptr(none), None, 1, ptr(self.status), st,
                     prof_name="seqrec_gather_rows")
            lq_neg = adj_n
            self._lq_tgt_adj = adj_t
        else:
            self._lq_tgt_adj = None
        return recv, Eneg, neg, lq_neg, rows_eff

    def eval_loss(self, d, negatives=None, step=0):
        """Sampled-softmax CE of this rank's batch (no update), scaled like train_step's return value."""
        self._wait_ready(d)
        if not self.unified:
            X, Etgt, Eneg, neg, _ = self._split_rows(d, step)
            self._cell_and_loss_split(d, X, Etgt, Eneg, neg, None, None, None, train=False)
            return self.loss_sum * (self.R / d["n_total"])
        recv, Eneg, neg, lq_neg, _ = self._rows_in(d, step)
        self._cell_unified(d, recv, Eneg, neg, lq_neg, step, train=False)
        return self.loss_sum * (float(self.R) / d["n_global"])

    def rank_counts(self, d):
        """Global rank of every target of THIS rank's tokens (Recall@K = mean(rank < K)): hidden rows and
        target scores of all ranks are all-gathered, every rank counts against its own Eout shard
        (seqrec_rank_count_thr) and the partial counts are summed with one all-reduce."""
        c, R, st = self.cfg, self.R, self._stream()
        n, w = d["n"], self.Hp
        self._wait_ready(d)
        thr = self.buf("thr", n)
        if self.unified:
            recv = self._rows_in(d, 0)[0]
            Hd = self._hidden(d, recv)
            call("seqrec_target_score", ptr(Hd), w, ptr(recv), ptr(self._recv_bias), ptr(d["take_tgt"]), n, ptr(thr), st)
        else:                                # D != H: one exchange per table, rows materialised
            X, Etgt, _, _, _ = self._split_rows(d, 0, negatives=False)
            Hd = self._hidden(d, None, X=X)
            call("seqrec_target_score", ptr(Hd), w, ptr(Etgt), None, ptr(d["arange"]), n, ptr(thr), st)
        nm = torch.tensor([n], dtype=torch.int64, device=self.dev)
        self.dist.all_reduce(nm, op=self.dist.ReduceOp.MAX, group=self.group)
        nmax = int(nm.item())
        pay = torch.zeros((nmax, w + 2), dtype=torch.float32, device=self.dev)
        pay[:, w] = float("inf")                                     # padding rows never count
        pay[:n, :w] = Hd
        pay[:n, w] = thr
        pay[:n, w + 1] = d["tgt"].view(torch.float32)
        pay[n:, w + 1] = torch.full((nmax - n,), -1, dtype=torch.int32, device=self.dev).view(torch.float32)
        parts = [torch.empty_like(pay) for _ in range(R)]
        if R > 1:
            self.dist.all_gather(parts, pay, group=self.group)
        else:
            parts = [pay]
        allp = torch.cat(parts)
        Hall = allp[:, :w].contiguous()
        thr_all = allp[:, w].contiguous()
        tg = allp[:, w + 1].contiguous().view(torch.int32).long()
        tgt_local = torch.where((tg >= 0) & (tg % R == self.rank), tg // R, torch.full_like(tg, -1)).to(torch.int32)
        counts = torch.zeros(R * nmax, dtype=torch.int32, device=self.dev)
        Et = self.P["E" if c.tied else "Eout"]
        call("seqrec_rank_count_thr", ptr(Hall), w, ptr(Et), ptr(self.P.get("bout")), ptr(tgt_local), ptr(thr_all), R * nmax, Et.shape[0],
             ptr(counts), st)
        self.dist.all_reduce(counts, group=self.group)
        return counts[self.rank * nmax: self.rank * nmax + n].clone()

    def topk_rows(self, d, k=20, rows=None, chunk=65536):
        """Top-k next items (GLOBAL ids) for this rank's tokens over the whole row-sharded catalogue:
        hidden rows of all ranks are all-gathered, every rank keeps a running top-64 over its own shard
        (seqrec_topk_merge), the per-shard candidates return to the rows' home ranks with one all-to-all
        and the final top-k of the R x 64 candidates is taken there.  -> (ids int32 [m,k], scores [m,k])."""
        if not 1 <= k <= 64:
            raise ValueError("1 <= k <= 64")
        c, R, st = self.cfg, self.R, self._stream()
        n, w = d["n"], self.Hp
        self._wait_ready(d)
        Hd = self._hidden(d, self._rows_in(d, 0)[0]) if self.unified else self._hidden(d, None, X=self._split_rows(d, 0, negatives=False)[0])
        if rows is not None:
            idx = torch.as_tensor(np.asarray(rows) if not torch.is_tensor(rows) else rows, dtype=torch.int32).to(self.dev)
            Hd = self._take(Hd, idx)
        m = Hd.shape[0]
        mm = torch.tensor([m], dtype=torch.int64, device=self.dev)
        self.dist.all_reduce(mm, op=self.dist.ReduceOp.MAX, group=self.group)
        mmax = int(mm.item())
        pad = torch.zeros((mmax, w), dtype=torch.float32, device=self.dev)
        pad[:m] = Hd
        parts = [torch.empty_like(pad) for _ in range(R)]
        if R > 1:
            self.dist.all_gather(parts, pad, group=self.group)
        else:
            parts = [pad]
        Hall = torch.cat(parts)                                          # [R * mmax, w]
        rows_all = R * mmax
        Et = self.P["E" if c.tied else "Eout"]
        Vl = Et.shape[0]
        sv = torch.full((rows_all, 64), float("-inf"), dtype=torch.float32, device=self.dev)
        si = torch.full((rows_all, 64), -1, dtype=torch.int32, device=self.dev)
        ch = int(min(chunk, max(Vl, 1)))
        sc = self.buf("topk_scores", rows_all, ch)
        for c0 in range(0, Vl, ch):
            wd = min(ch, Vl - c0)
            self.gemm(1, 1, rows_all, wd, w, Hall, w, Et[c0:c0 + wd], w, sc, ch, tag="topk")
            call("seqrec_topk_merge", ptr(sc), ch, rows_all, wd, c0, ptr(self.P.get("bout")), ptr(sv), ptr(si), st)
        gi = torch.where(si >= 0, si * R + self.rank, si)                # local row -> global item id
        cand_v = self.ex.swap_fixed(sv.view(R, mmax, 64))                 # [R, mmax, 64]: candidates of MY rows from every shard
        cand_i = self.ex.swap_fixed(gi.view(R, mmax, 64))
        cv = cand_v.permute(1, 0, 2).reshape(mmax, R * 64).contiguous()
        ci = cand_i.permute(1, 0, 2).reshape(mmax, R * 64).contiguous()
        fv = torch.full((mmax, 64), float("-inf"), dtype=torch.float32, device=self.dev)
        fc = torch.full((mmax, 64), -1, dtype=torch.int32, device=self.dev)
        call("seqrec_topk_merge", ptr(cv), R * 64, mmax, R * 64, 0, None, ptr(fv), ptr(fc), st)
        out_v = torch.empty((mmax, k), dtype=torch.float32, device=self.dev)
        out_c = torch.empty((mmax, k), dtype=torch.int32, device=self.dev)
        call("seqrec_topk_finish", ptr(fv), ptr(fc), mmax, k, ptr(out_v), ptr(out_c), st)
        out_i = torch.gather(ci, 1, out_c.long().clamp_(min=0))          # candidate column -> global item id
        return out_i[:m].contiguous(), out_v[:m].contiguous()

    def _hidden(self, d, recv, X=None):
        """Hidden rows of this rank's tokens from the receive buffer of _rows_in (input rows read through their positions) or,
        split path, from the materialised input rows X."""
        P = self.P
        n, Hp, GHp, Dp = d["n"], self.Hp, self.GHp, self.Dp
        XW = self.buf("XW", n, GHp)
        if X is not None:
            self.gemm(1, 0, n, GHp, Dp, X, Dp, P["W"], GHp, XW, GHp, bias=P.get("b"), tag="xw")
        else:
            self.gemm(1, 0, n, GHp, Dp, recv, self.Hp, P["W"], GHp, XW, GHp, bias=P.get("b"), tag="xw",
                      fuse=_lib.gemm_fuse(a_index=d["take_in"]))
        Hout = self.buf("Hout", n, Hp); gates = self.buf("gates", n, GHp); aux = self.buf("aux", n, Hp)
        self._scan_fwd(d, XW, Hout, gates, aux)
        return Hout

    def _step_unified(self, d, lr, eps, clipnorm, step, apply_update):
        """Collectives: all-to-all (rows in), all-reduce (dense gradients, side stream, under what follows), all-to-all (row
        gradients out), all-reduce (ONE float: the owned rows' squared gradient norm).  Launches besides the cell's own:
        exchange_pack, exchange_unpack, exchange_grad_pack, the combined scatter, norm (+ batch loss) in a fixed order,
        the 2-term sum, the update."""
        c, R = self.cfg, self.R
        st = self._stream()
        n, w, K = d["n"], self.Hp, c.K
        plan = d["plan"]
        recv, Eneg, neg, lq_neg, rows_eff = self._rows_in(d, step)
        r = self._cell_unified(d, recv, Eneg, neg, lq_neg, step, reduce_dense=apply_update)
        # -- row gradients travel the same routes back: the routing launch reads dX / dEneg as split-K slabs and dlt * Hd in place
        backbuf = self.buf("backbuf", plan.n_tot, w)
        (dX, nsx, ssx), (dEn, nsn, ssn) = r["dX"], r["dEneg"]
        dbn = bias_grad = None
        if c.out_bias:       # d loss / d bias: dlt for a target, the column sum of dlogits for a negative; same routes back
            dbn = self.buf("dbn", K)
            call("seqrec_colsum", ptr(self.buf("ln", n, K)), n, K, K, ptr(dbn), 0, ptr(self.buf("colsum_ws", 64 * K)), st)
            bias_grad = self.buf("bias_back", plan.n_tot)
        call("seqrec_exchange_grad_pack", ptr(d["back_idx"]), plan.n_tot, n, K, w, ptr(dX), nsx, ssx, ptr(r["Hd"]), ptr(r["dlt"]),
             ptr(dEn), nsn, ssn, ptr(backbuf), ptr(dbn), ptr(bias_grad), st)
        gback = self.ex.push_seg(plan, backbuf)                                 # collective 2
        gbias = self.ex.push_seg(plan, bias_grad) if c.out_bias else None       # (+ one all-to-all of m_tot floats with a bias)
        jobs = [dict(table=self.TT, accum=self.TA, gtab=self.TG, slot=self.TS, rows=rows_eff, vals=gback, ldv=w, row_scale=None,
                     n=plan.m_tot, width=w, base=0, name="T")]
        if c.out_bias:       # the bias gradients: a width-1 scatter list over the same owner-side rows
            jobs.append(dict(table=self.P["bout"], accum=self.A["bout"], gtab=self.Gt["bout"], slot=self.slot["bout"], rows=self._bias_rows,
                             vals=gbias, ldv=1, row_scale=None, n=plan.m_tot, width=1, base=0, name="bout"))
        job, cnt = _lib.rows_jobs(jobs)
        if c.merge == "sorted":
            # the deterministic merge (round 4; csrc/merge.hip): the received row gradients sorted by row (stable) and summed per row in
            # arrival order -- with the slab sums of grad_pack (slab order) and the fixed-order norms, a rank's update repeats bit for
            # bit given the same received bytes; what the reference's dense Adagrad gives on one device (experiments_methods.py:41)
            self._merge_sorted(jobs)
        else:
            for j1 in jobs:          # one launch per list: lists of different widths would lose the combining form
                a1, c1 = _lib.rows_jobs([j1])
                call("seqrec_rows_scatter_add_multi", a1, c1, st)
        self._last = {"gback": gback, "send_idx": rows_eff, "n": n, "m_tot": plan.m_tot, "n_tot": plan.n_tot, "prev": d["prev"]}    # what a failure report names
        if not apply_update:
            return None
        # -- norm of the owned row gradients (+ this rank's batch loss), summed over the ranks.  Two slots alternate (norms[0] /
        # norms[2]): this step's update launch clears the slot of the NEXT step (zero_next), nobody touches it meanwhile.  Any
        # summation order will do here -- the all-reduce hands every rank the same sum.
        cur = self.norms[0:1] if self._norm_par == 0 else self.norms[2:3]
        nxt = self.norms[2:3] if self._norm_par == 0 else self.norms[0:1]
        self._norm_par ^= 1
        if c.merge == "sorted":
            npart = int(_lib.load().seqrec_opt_sqnorm_ordered_floats(0, cnt, plan.m_tot))
            call("seqrec_opt_sqnorm_ordered", 0, None, None, job, cnt, ptr(self.buf("sq_partials", npart)), npart, ptr(cur), 0,
                 ptr(r["loss_rows"]), n, ptr(self.loss_out), st)
        else:
            call("seqrec_opt_sqnorm", 0, None, None, job, cnt, ptr(cur), ptr(r["loss_rows"]), n, ptr(self.loss_out), st)
        self.dist.all_reduce(cur, group=self.group)                             # collective 3: one float
        torch.cuda.current_stream(self.dev).wait_stream(self.side)              # dense gradients reduced, their fixed-order norm in norms[1]
        dk, pp, pa, gp, nn, _ = self._dense_arrays()
        call("seqrec_opt_apply", len(dk), pp, pa, gp, nn, job, cnt, ptr(cur),
             float(clipnorm if clipnorm else 0.0), lr, eps, ptr(self.scale), ptr(nxt), ptr(d["ntok"]), ptr(self.status),
             ptr(self.norms[1:2]), st)
        self.sq = cur
        self.upack_dirty = True
        return self.loss_sum * (float(self.R) / d["n_global"])    # this rank's share, scaled so the mean over ranks is the global loss

    def _split_rows(self, d, step, negatives=True):
        """Forward exchanges of the split path (D != H: one exchange per table): input rows, target rows and -- for a loss --
        the stratified negatives with their ids.  -> (X [n, Dp], Etgt [n, Hp], Eneg [K, Hp], neg int32[K], negl)."""
        c, P, R = self.cfg, self.P, self.R
        st = self._stream()
        Hp, Dp, K = self.Hp, self.Dp, c.K
        Kr = K // R
        tname = "E" if c.tied else "Eout"
        X = self.ex.fetch(d["plan_in"], self._gather_from(P["E"]), Dp, self._take)
        Etgt = self.ex.fetch(d["plan_tgt"], self._gather_from(P[tname]), Hp, self._take)
        if not negatives:
            return X, Etgt, None, None, None
        th, al, _ = self.sampler
        negl = self.buf("negl", R * Kr, dtype=torch.int32)             # local rows I draw for every requester
        call("seqrec_sample_negatives", int(c.seed), int(step) * R + self.rank, R * Kr, ptr(th), ptr(al), c.V_out, ptr(negl), st)
        # rows and their global ids travel together: [R, Kr, Hp + 1] with the id bit-cast into the last column
        ids_out = (negl.long() * R + self.rank).to(torch.int32)
        pay = torch.empty((R * Kr, Hp + 1), dtype=torch.float32, device=self.dev)
        pay[:, :Hp] = self._take(P[tname], negl)
        pay[:, Hp] = ids_out.view(torch.float32)
        got = self.ex.swap_fixed(pay.view(R, Kr, Hp + 1)).view(K, Hp + 1)
        return X, Etgt, got[:, :Hp].contiguous(), got[:, Hp].contiguous().view(torch.int32), negl

    def _step_split(self, d, lr, eps, clipnorm, step, apply_update):
        """D != H: the two tables have different row widths, one exchange per table."""
        c, P, R = self.cfg, self.P, self.R
        st = self._stream()
        n, Hp, Dp, K = d["n"], self.Hp, self.Dp, c.K
        Kr = K // R
        tname = "E" if c.tied else "Eout"
        X, Etgt, Eneg, neg, negl = self._split_rows(d, step)
        dX = self.buf("dX", n, Dp); dEtgt = self.buf("dEtgt", n, Hp); dEneg = self.buf("dEneg", K, Hp)
        self._cell_and_loss_split(d, X, Etgt, Eneg, neg, dX, dEtgt, dEneg)
        Gt = self.Gt
        jobs = []
        g_in, r_in = self.ex.push(d["plan_in"], dX, self._take)
        g_tg, r_tg = self.ex.push(d["plan_tgt"], dEtgt, self._take)
        g_ng = self.ex.swap_fixed(dEneg.view(R, Kr, Hp)).view(R * Kr, Hp)        # grads for the rows I drew
        base = 0
        for (tab, rows, g) in ((tname, r_tg, g_tg), (tname, negl, g_ng), ("E", r_in, g_in)):
            m = rows.numel()
            if m:
                call("seqrec_rows_scatter_add", ptr(Gt[tab]), ptr(self.slot[tab]), ptr(rows), ptr(g), g.shape[1], None, m,
                     g.shape[1], base, st)
                jobs.append((tab, rows, m, g.shape[1], base, g))
            base += m
        if not apply_update:
            return jobs
        call("seqrec_fill_f32", ptr(self.sq), 0.0, 1, st)
        for (tab, rows, m, w, b, _) in jobs:
            call("seqrec_rows_sqnorm", ptr(Gt[tab]), ptr(self.slot[tab]), ptr(rows), m, w, b, ptr(self.sq), st)
        self._dense_update(lr, eps, clipnorm)
        for (tab, rows, m, w, b, _) in jobs:
            call("seqrec_rows_adagrad", ptr(P[tab]), ptr(self.A[tab]), ptr(Gt[tab]), ptr(self.slot[tab]), ptr(rows), m, w, b,
                 lr, eps, ptr(self.scale), st)
        return self.loss_sum * (self.R / d["n_total"])


class WindowPlanner:
    """Feeds a training loop over a row-sharded engine with routed batches, a WINDOW at a time, without ever making it wait for the
    planner: the next window's count exchange is BEGUN when half of the current one is left (ShardedEngine.prepare_begin: queued,
    nothing waits) and ENDED a quarter window later (prepare_end: the counts are there by then).  make_rb(i) -> the host RaggedBatch of
    step i (or None past the end); get(i) must be called with i = 0, 1, 2, ... on every rank alike (the planner's collectives are
    issued at the same loop positions everywhere).  pack_s / plan_s: host seconds spent building batches / planning."""

    def __init__(self, eng, make_rb, window=32):
        self.eng, self.make_rb, self.W = eng, make_rb, int(window)
        self.ready = {}            # step -> device batch
        self.next_lo = 0           # first step not yet planned or begun
        self.begun = None          # (lo, hi, handle) of the window whose counts are in flight
        self.pack_s = self.plan_s = 0.0
        self.windows = 0
        self.done = False

    def _begin(self):
        import time
        t0 = time.perf_counter()
        rbs = []
        for j in range(self.next_lo, self.next_lo + self.W):
            rb = self.make_rb(j)
            if rb is None:
                self.done = True
                break
            rbs.append(rb)
        t1 = time.perf_counter()
        self.pack_s += t1 - t0
        if rbs:
            self.begun = (self.next_lo, self.next_lo + len(rbs), self.eng.prepare_begin(rbs))
            self.next_lo += len(rbs)
        self.plan_s += time.perf_counter() - t1

    def _end(self):
        import time
        t0 = time.perf_counter()
        lo, hi, h = self.begun
        self.ready.update(zip(range(lo, hi), self.eng.prepare_end(h)))
        self.begun = None
        self.windows += 1
        self.plan_s += time.perf_counter() - t0

    def get(self, i):
        if i not in self.ready:                                   # start of the loop (or a consumer running ahead): plan now, blocking
            if self.begun is None and not self.done:
                self._begin()
            if self.begun is not None:
                self._end()
        elif self.begun is None and not self.done and (i + self.W // 2) >= self.next_lo:
            self._begin()                                          # half a window left: queue the next window's count exchange
        elif self.begun is not None and (i + self.W // 4) >= self.begun[0]:
            self._end()                                            # a quarter left: its counts arrived long ago
        return self.ready.pop(i)
