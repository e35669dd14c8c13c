/*
 * seqrec_hip.h  --  C ABI of libseqrec_hip.so (MI355X / gfx950).
 *
 * The reference (efikarra/seq-recommendations) has NO native boundary: its hot
 * path sits behind the Python class surface of model.py (BaseRNNModel,
 * model.py:170-238) and all arithmetic happens inside Keras 2.0.x / Theano.
 * This header therefore declares the boundary a maintainer would bind with
 * ctypes from model.py; each entry point names the Keras
 * op / reference call site whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed, e.g. torch tensor
 *     data_ptr()) unless its name ends in _host;
 *   - fp32 everywhere ("dtype": "f32"), row-major, sizes in elements;
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered and
 *     no entry point synchronises, except the three that say so
 *     (seqrec_cluster_scan_errors, seqrec_release_stream, seqrec_graph_cache_clear);
 *   - return value: 0 = ok, negative = SEQREC_E_* (bad argument / unsupported
 *     shape, detected on the host BEFORE anything is launched), positive =
 *     hipError_t of a failed launch.  No exceptions cross the boundary.
 *     Conditions only the DEVICE can see (a gradient norm that is not finite, an
 *     index outside its table, an in-kernel wait that ran out) are reported
 *     through a caller-owned status word (SEQREC_STATUS_* bits, uint32_t in
 *     device memory, zeroed by the caller) and through
 *     seqrec_cluster_scan_errors; the Python engine raises on either at its
 *     next host synchronisation (engine.Engine.check_status).
 *   - Hidden state.  Workspaces are the caller's, with two exceptions that
 *     live inside the library, both keyed by stream and both freed by
 *     seqrec_release_stream(stream):
 *       (1) the cluster form of the recurrent scans (seqrec_rnn_*_stepwise)
 *           keeps 16.25 KB of exchange flags per stream -- hipMalloc + a
 *           synchronous hipMemset on the stream's FIRST cluster scan, then
 *           epoch-numbered (no reset between calls);
 *       (2) use_graph != 0 keeps captured launch graphs (a ring of up to 4
 *           executables per kernel sequence and stream, <= 512 sequences,
 *           mutex-guarded; seqrec_graph_cache_clear drops them all).
 *     Two process-wide test switches exist (seqrec_debug_*): they select
 *     between equivalent kernel forms or force a failure path, are not
 *     thread-safe, and no product code calls them.
 *     Residency assumption of the cluster scans: a launch may hold more row
 *     blocks than can be resident at once (launch_sliced); forward progress
 *     then rests on the dispatcher handing workgroups out in blockIdx order
 *     per XCD, so that an XCD holds complete groups (which finish on their
 *     own) and at most one group that is still arriving.  That is observed
 *     behaviour of this runtime, not a documented guarantee, and other
 *     processes on the GPU can break it: EVERY in-kernel wait is bounded,
 *     counted (seqrec_cluster_scan_errors) and poisons its output; the Python
 *     engine raises on a non-zero count and switches the process to the
 *     step-wise form (seqrec_debug_scan_cluster(0)) for the retry.
 *     Environment: the library reads NO environment variable.  The developer
 *     switches of earlier rounds (SEQREC_GEMM_V2, _V2_TILE, _V2_GTILE, _BK32,
 *     _TILE_THR, SEQREC_CE_BLOCK, SEQREC_SCATTER_COMBINE, SEQREC_SCAN_CLUSTER,
 *     SEQREC_SCAN_XCD, SEQREC_SCAN_WIDE_RB) exist only in a -DSEQREC_TUNABLES
 *     build (tools/build_diag.py tunables), where each is read once per
 *     process at its first use; in the product build each is its default, a
 *     compile-time constant (csrc/common.h seqrec_env).
 *     Two entry points DO synchronise beyond the three named above: the first
 *     cluster scan on a stream (hipMalloc + hipMemset of its flag block) and
 *     use_graph != 0 when all four executables of a sequence's ring are still
 *     queued (hipEventSynchronize on the oldest).
 *     One stream per caller thread; calls on one stream must not be issued
 *     concurrently from two threads.
 *
 * Ragged session batch layout ("time-major packed", produced by
 * batching.pack_sessions; replaces the dense pre-padded (N,T,V) one-hot tensors
 * of preprocessor.py:67-94 + the Masking layer of model.py:246,335-336):
 *   sessions sorted by number of transitions L_b descending; B_t = #{b: L_b > t};
 *   step_off[t] = sum_{s<t} B_s (T+1 entries); token p = step_off[t] + b.
 */
#ifndef SEQREC_HIP_H
#define SEQREC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: seqrec_rows_job grew (n_slabs, slab_stride); new entry points seqrec_gemm_f32_slabs, seqrec_gemm_f32_grouped_slabs,
 *    seqrec_opt_sqnorm_slabs, seqrec_pack_batch_host, seqrec_rnn_pack_u_sample, seqrec_rnn_bwd_stepwise_parts */
/* 4: seqrec_opt_apply takes a status word; new entry points seqrec_gather_rows_bounded, seqrec_release_stream,
 *    seqrec_cluster_scan_errors_reset, seqrec_debug_cluster_spin_limit, seqrec_exchange_pack / _unpack / _grad_pack,
 *    seqrec_sampled_softmax_ce_rows_idx, seqrec_route_*_host; the packed layout of the step-wise LSTM forward
 *    kernel changed (seqrec_rnn_pack_u_stepwise and the scans of one library always agree) */
/* 5: seqrec_exchange_unpack bounds the ids it reads (logq_rows, status); seqrec_exchange_pack takes the received request list;
 *    seqrec_route_blob_host writes the row kinds itself (no got_sentinel); new entry point seqrec_train_cell */
#define SEQREC_ABI_VERSION 5

enum { SEQREC_OK = 0, SEQREC_E_ARG = -1, SEQREC_E_SHAPE = -2, SEQREC_E_UNSUPPORTED = -3 };
/* bits of a device status word (see Conventions): set with atomic OR by the kernels, never cleared by them */
enum {
    SEQREC_STATUS_BAD_NORM = 1,      /* seqrec_opt_apply: squared gradient norm negative, NaN or infinite -- update NOT applied */
    SEQREC_STATUS_BAD_DIVISOR = 2,   /* seqrec_opt_apply: *grad_div not a finite number > 0            -- update NOT applied */
    SEQREC_STATUS_BAD_SCALE = 4,     /* seqrec_opt_apply: clip scale 0 / not finite (a norm so large that clipnorm / norm
                                        underflows): the step would be a silent no-op                  -- update NOT applied */
    SEQREC_STATUS_BAD_INDEX = 8      /* seqrec_gather_rows_bounded: an index >= table_rows (read as a zero row) */
};

/* recurrent cell (Keras layer constructed at model.py:248-254,344-352; GRU = extension) */
enum { SEQREC_CELL_SIMPLERNN = 0, SEQREC_CELL_LSTM = 1, SEQREC_CELL_GRU = 2 };
/* activation = z_to_z_activation / z_activation (model.py:243,324); recurrent activation is
 * always Keras' hard_sigmoid */
enum { SEQREC_ACT_RELU = 0, SEQREC_ACT_TANH = 1, SEQREC_ACT_LINEAR = 2,
       /* ABI 5, seqrec_rnn_{fwd,bwd}_stepwise[_parts] only (step-wise form; the cluster form and seqrec_rnn_fwd / _bwd take 0-2): */
       SEQREC_ACT_SIGMOID = 3, SEQREC_ACT_HARD_SIGMOID = 4, SEQREC_ACT_SOFTPLUS = 5, SEQREC_ACT_SOFTSIGN = 6, SEQREC_ACT_ELU = 7 };

int seqrec_abi_version(void);
/* name of the gfx target the code objects were built for ("gfx950") */
const char* seqrec_build_arch(void);

/* ---- item-embedding lookup: the `onehot . W` product inside Keras' SimpleRNN/LSTM
 *      (model.py:248-255,345-369) and the Dense embed of NoRecurrenceModel (model.py:276-279).
 *      out[i,:] (+)= table[ids[i],:] * (row_scale ? row_scale[i] : 1) + (bias ? bias[:] : 0)
 *      ids[i] < 0 yields a zero row (used for "previous hidden state" of first steps).
 *      Algorithmic HBM bytes: 8 * width per row (4 read + 4 written). */
int seqrec_gather_rows(const float* table, const int32_t* ids, float* out, int64_t n, int width,
                       const float* row_scale, const float* bias, int accumulate, void* stream);
/*      the same with the table's row count: an index >= table_rows reads nothing (zero row, like a negative one) and sets
 *      SEQREC_STATUS_BAD_INDEX in *status (nullable).  For index lists that crossed PCIe or a collective (the row exchange
 *      of distributed.py): a stale or corrupted index cannot pull arbitrary memory into a gradient. */
int seqrec_gather_rows_bounded(const float* table, int64_t table_rows, const int32_t* ids, float* out, int64_t n, int width,
                               const float* row_scale, const float* bias, int accumulate, uint32_t* status, void* stream);

/* ---- dense GEMM on the fp32 MFMA path (exact fp32; v_mfma_f32_32x32x2_f32).
 *      C[M,N] (+)= opA(A)[M,K] . opB(B)[K,N] (+ bias[N]).  Replaces the BLAS calls Theano makes for
 *      the cell's x.W / TimeDistributed(Dense) (model.py:257,381-384) and their gradients.
 *      a_kcontig: A(m,k)=A[m*lda+k] (row-major MxK) else A(m,k)=A[k*lda+m] (stored KxM);
 *      b_kcontig: B(k,n)=B[n*ldb+k] (stored NxK)     else B(k,n)=B[k*ldb+n] (row-major KxN).
 *      splitk > 1 needs `workspace` of splitk*M*N floats.  accumulate != 0: C += result. */
int seqrec_gemm_f32(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                    const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, const float* bias, int accumulate,
                    int splitk, float* workspace, void* stream);
int64_t seqrec_gemm_workspace_floats(int64_t M, int64_t N, int splitk);
/*      the same product with its split-K partial sums LEFT in `workspace` (seqrec_gemm_workspace_floats floats): slab s
 *      = workspace + s * M * N, row stride N, *n_slabs of them (<= splitk; 1 = the whole product).  No reduce launch and no
 *      C: for products whose only reader adds the slabs itself -- the row scatter of dX / dEneg (seqrec_rows_job.n_slabs). */
int seqrec_gemm_f32_slabs(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                          const float* A, int64_t lda, const float* B, int64_t ldb,
                          int splitk, float* workspace, int* n_slabs, void* stream);
/*      the same GEMM with the gathers of the hot path fused in (no materialised copy, no extra launch):
 *      a_index  : the A operand is read THROUGH an index -- a_kcontig: row m of A is A[a_index[m]*lda + k]
 *                 (x.W with x = E[ids]: the embedding lookup of model.py:248-255 inside the cell GEMM);
 *                 !a_kcontig: slice k of A is A[a_index[k]*lda + m] (dU = Hout[prev]^T . dPre).  A negative
 *                 index is an all-zero row.  M (or K) counts index entries; 64x64 tiles are used.
 *      add_*    : C[m,:] += add_scale[m] * add_table[add_index[m]*add_ld + :] where the final C is written
 *                 (dH = dlogits . Eneg + dlt * Eout[tgt]); add_scale NULL = 1, add_index < 0 adds nothing.
 *      fuse == NULL is seqrec_gemm_f32. */
typedef struct seqrec_gemm_fuse {
    const int32_t* a_index;
    const float* add_table; const int32_t* add_index; const float* add_scale; int64_t add_ld;
} seqrec_gemm_fuse;
int seqrec_gemm_f32_fused(int a_kcontig, int b_kcontig, int64_t M, int64_t N, int64_t K,
                          const float* A, int64_t lda, const float* B, int64_t ldb,
                          float* C, int64_t ldc, const float* bias, int accumulate,
                          int splitk, float* workspace, const seqrec_gemm_fuse* fuse_host, void* stream);
/*      TWO independent products of different layouts in ONE launch (ABI 5): product 0 as seqrec_gemm_f32_fused would compute it
 *      (C0, split-K + reduce, the add_* row term in the final write), product 1 as seqrec_gemm_f32_slabs would leave it (n_slabs1
 *      slabs in ws1).  For dH = dlogits . Eneg beside dEneg = dlogits^T . H of the sampled softmax's backward: both read dlogits and
 *      neither fills the chip at MSNBC-shaped batch sizes.  Shapes the pair kernel does not cover (other layouts than (1,0) + (0,0),
 *      gathered operands, products that fill the chip alone) are issued one after the other with identical results; `together`
 *      reports which way it went.  The plan is HOST memory; n_slabs1 / together are outputs. */
typedef struct seqrec_gemm_pair {
    int32_t a_kc0, b_kc0, a_kc1, b_kc1;
    int64_t M0, N0, K0; const float* A0; int64_t lda0; const float* B0; int64_t ldb0; float* C0; int64_t ldc0; int32_t splitk0, reserved0_; float* ws0;
    const float* add_table; const int32_t* add_index; const float* add_scale; int64_t add_ld;
    int64_t M1, N1, K1; const float* A1; int64_t lda1; const float* B1; int64_t ldb1; int32_t splitk1, reserved1_; float* ws1;
    int32_t n_slabs1, together;
} seqrec_gemm_pair;
int seqrec_gemm_f32_pair(seqrec_gemm_pair* plan_host, void* stream);
/*      grouped form: up to 6 independent problems that share the layout flags, K and the split count
 *      in ONE launch (the weight-gradient GEMMs dW / dU all reduce over K = N_tok).
 *      workspace (splitk > 1): sum_i splitk * M_i * N_i floats. */
typedef struct seqrec_gemm_desc {
    int64_t M, N, K;
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    const float* bias; int32_t accumulate;
    const int32_t* a_index;           /* nullable: gathered A operand, as in seqrec_gemm_fuse */
} seqrec_gemm_desc;
int seqrec_gemm_f32_grouped(int count, int a_kcontig, int b_kcontig, const seqrec_gemm_desc* descs_host,
                            int splitk, float* workspace, void* stream);
/*      the grouped products with their split-K partial sums LEFT in `workspace` (problem i at float offset
 *      sum_{j<i} n_slabs * M_j * N_j, slab s at + s * M_i * N_i, row stride N_i; *n_slabs <= splitk of them): no reduce
 *      launch; C / ldc / bias / accumulate of the descs are not used here.  seqrec_opt_sqnorm_slabs finishes them. */
int seqrec_gemm_f32_grouped_slabs(int count, int a_kcontig, int b_kcontig, const seqrec_gemm_desc* descs_host,
                                  int splitk, float* workspace, int* n_slabs, void* stream);
/*      diagnostics (tests): force the workgroup tile of the LDS-DMA GEMM kernels -- tile 1 = 64x64, 2 = 128x64,
 *      3 = 128x128 (grouped form: 1 or 2); <= 0 restores the built-in choice.  Results never depend on it beyond
 *      the order of the split-K partial sums. */
void seqrec_debug_gemm_tile(int tile, int grouped_tile);
/*      diagnostics (tests ONLY; process-wide, not thread-safe): scan form of seqrec_rnn_fwd_stepwise / _bwd_stepwise --
 *      1 cluster (one launch, in-kernel exchange between the column-slice workgroups of a row block), 0 step-wise (one
 *      launch per recurrent product), -1 the built-in choice (cluster whenever the call qualifies; SEQREC_SCAN_CLUSTER=0
 *      disables it).  seqrec_debug_cluster_spin_limit(polls > 0) shortens the bounded waits of the cluster kernels so that a
 *      test can drive their failure path (0 restores the built-in 2^22 polls). */
void seqrec_debug_scan_cluster(int mode);
void seqrec_debug_cluster_spin_limit(int polls);
/*      every wait inside the cluster kernels is a BOUNDED spin.  A wave whose wait runs out counts it, writes NaN into the
 *      output element it owns (Hout / dPre of that step: nothing downstream looks plausible, and the gradient norm of the
 *      step is not finite, which seqrec_opt_apply refuses) and leaves.  scan_errors returns the count on `stream` since
 *      its first cluster scan or the last reset (0 in a healthy run; SYNCHRONISES the stream; -1 on a HIP error). */
int seqrec_cluster_scan_errors(void* stream);
int seqrec_cluster_scan_errors_reset(void* stream);
/*      frees the library's per-stream state (Conventions: hidden state); synchronises the stream first */
int seqrec_release_stream(void* stream);

/* ---- recurrent scan over the ragged batch (Keras K.rnn under Masking).
 *      H must be 64, 128, 256 or 512 (callers zero-pad); H_real <= H are the live units.
 *      XW   [N_tok, G*H]  input projection incl. bias (gate order i,f,c,o | z,r,h)
 *      U    [H, G*H]      recurrent kernel
 *      Hout [N_tok, H]    hidden state per token (the layer output, return_sequences=True)
 *      gates[N_tok, G*H]  post-activation gate values (stash for BPTT; unused for SimpleRNN)
 *      aux  [N_tok, H]    LSTM: cell state c_t;  GRU: r_t * h_{t-1};  SimpleRNN: unused
 *      upack: seqrec_rnn_upack_floats() floats written by seqrec_rnn_pack_u (U re-laid-out into
 *             per-wave MFMA B-fragment order, forward layouts then transposed backward layouts);
 *             re-pack after every update of U.
 *      step_off: DEVICE int32[T+1]; B = number of sessions (= step_off[1]) */
int64_t seqrec_rnn_upack_floats(int cell, int H);
int seqrec_rnn_pack_u(int cell, int H, const float* U, float* upack, void* stream);
int seqrec_rnn_fwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* step_off,
                   const float* XW, float* Hout, float* gates, float* aux,
                   const float* upack, void* stream);
/*      BPTT of the same scan (Theano autodiff through scan).
 *      dHout [N_tok,H]  in: dLoss/dHout;  dPre [N_tok,G*H] out: dLoss/d(pre-activations) = dLoss/dXW.
 *      dU, dW, db follow from dPre by seqrec_gemm_f32 / seqrec_colsum. */
int seqrec_rnn_bwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* step_off,
                   const float* dHout, const float* Hout, const float* gates, const float* aux,
                   float* dPre, const float* upack, void* stream);

/*      Step-wise entry points of the same scan: two forms behind one signature, chosen per call.
 *      CLUSTER form (rnn_cluster.hip, rnn_cluster2.hip; every cell, with or without rmask, T <= 159): ONE launch per
 *      call; the 16 session rows of a row block belong to a group of H/16 resident workgroups that keep their slices of
 *      the recurrent kernel in registers and exchange h (LSTM, SimpleRNN: one exchange per step; GRU: two) inside the
 *      kernel.  Taken when all workgroups of a launch can be resident (occupancy query; larger batches go in slices).
 *      STEP-WISE form (rnn_step.hip): one small whole-chip launch per recurrent GEMM (GRU: 2 per step; LSTM/SimpleRNN:
 *      1 forward, pointwise + GEMM backward) -- for T > 159, the activations beyond relu / tanh / linear, after
 *      seqrec_debug_scan_cluster(0), and as the form the cluster kernels are tested against.  Same buffers; Hout / gates / aux agree BIT FOR
 *      BIT between the forms, dPre to the last bits.
 *      upack: seqrec_rnn_upack_floats() floats written by seqrec_rnn_pack_u_stepwise.
 *      bwd workspace: 2 * N_tok * H floats (step-wise form only).
 *      rmask (nullable): recurrent-dropout multipliers [G][B][H] for the SORTED session rows
 *      (Keras recurrent_dropout, model.py:346,351: one mask per gate, fixed over time); with it the
 *      GRU aux stash holds r*h_prev WITHOUT the mask. */
int seqrec_rnn_pack_u_stepwise(int cell, int H, const float* U, float* upack, void* stream);
int seqrec_rnn_fwd_stepwise(int cell, int act, int H, int H_real, int T, int B,
                            const int32_t* step_off, const int32_t* step_off_host, const float* XW,
                            float* Hout, float* gates, float* aux, const float* upack,
                            const float* rmask, int use_graph, void* stream);
int seqrec_rnn_bwd_stepwise(int cell, int act, int H, int H_real, int T, int B,
                            const int32_t* step_off, const int32_t* step_off_host, int64_t n_tok,
                            const float* dHout, const float* Hout, const float* gates, const float* aux,
                            float* dPre, const float* upack, float* workspace, const float* rmask,
                            int use_graph, void* stream);
/*      the same BPTT with its input gradient given IN PARTS -- dHout[q, :] = sum_{s < n_slabs} (slabs + s * slab_stride)[q*H + :]
 *      + add_scale[q] * add_table[add_index[q] * add_ld + :] (add_table NULL: no such term; add_scale NULL = 1; add_index < 0
 *      adds nothing): the split-K slabs of dH = dlogits . Eneg (seqrec_gemm_f32_slabs) and the target-row term that the
 *      reduce launch of seqrec_gemm_f32_fused would add.  The cluster form of the GRU scan adds the parts where it reads
 *      dHout (no reduce launch); every other form first writes the sum to `dHout_scratch` (n_tok * H floats, required). */
typedef struct seqrec_dh_parts {
    const float* slabs; int32_t n_slabs; int32_t reserved_; int64_t slab_stride;
    const float* add_table; const int32_t* add_index; const float* add_scale; int64_t add_ld;
} seqrec_dh_parts;
int seqrec_rnn_bwd_stepwise_parts(int cell, int act, int H, int H_real, int T, int B,
                                  const int32_t* step_off, const int32_t* step_off_host, int64_t n_tok,
                                  const seqrec_dh_parts* parts, float* dHout_scratch, const float* Hout, const float* gates,
                                  const float* aux, float* dPre, const float* upack, float* workspace, const float* rmask,
                                  int use_graph, void* stream);
/*      step_off_host: int32[T+1] on the HOST (required): every launch gets its exact geometry as kernel arguments;
 *      step_off (device copy) is not read by these two entry points and may be NULL (kept for signature stability).
 *      use_graph != 0: the call's launch sequence is captured ONCE per distinct kernel sequence (cell, activation, H,
 *      direction, T) into a hipGraph (process-wide cache, <= 512 entries); every later call REWRITES the nodes'
 *      grids and arguments for its batch (hipGraphExecKernelNodeSetParams) and replays -- same kernels, same
 *      arguments as the eager form, ~0.6 us of host time per launch instead of ~3 us.  An executable is only
 *      rewritten after the event recorded behind its previous launch has completed (ring of 4 per sequence and
 *      stream), so calls may be enqueued back to back without host synchronisation.  seqrec_graph_cache_clear()
 *      drops the cache (waits for launches still queued). */
int seqrec_graph_cache_clear(void);

/* ---- softmax + Theano categorical_crossentropy under the Keras token-mean mask
 *      (model.py:175-177,257,397), fused with its gradient.
 *      logits [n,V] is overwritten by dlogits = (p - onehot) * active * inv_denom, where active = 0
 *      when the target probability is clipped (outside [1e-7, 1-1e-7]).  loss_rows [n] receives
 *      -log(clip(p_i[tgt_i])) per token (sum it with seqrec_reduce_sum).  probs (nullable) gets p.
 *      tgt == NULL: prediction only (probs required, logits untouched). */
int seqrec_full_softmax_ce(float* logits, int64_t ld, const int32_t* tgt, int64_t n, int V,
                           float inv_denom, float* loss_rows, float* probs, void* stream);
/*      sampled softmax over {target} U K shared negatives (extension).
 *      ln [n,K] = Hd . Eout[neg]^T is overwritten by dln.  The target logit is computed here from
 *      hd[n,H] and Eout[tgt] (row gather); bout (nullable, [V]) is added and logq (nullable, [V])
 *      subtracted for every candidate; accidental hits (neg == tgt) are removed.  dlt [n] out.
 *      cand_logq (nullable, [K]) = logq[neg[k]] gathered ONCE per step by the caller: the negatives are
 *      shared by all rows, so the kernel then reads one coalesced vector instead of doing K dependent
 *      item-table lookups per row (the target term still uses logq[tgt]). */
int seqrec_sampled_softmax_ce(float* ln, int64_t ld, const float* hd, int H, const float* Eout,
                              const float* bout, const float* logq, const float* cand_logq,
                              const int32_t* tgt, const int32_t* neg, int64_t n, int K, float inv_denom,
                              float* loss_rows, float* dlt, void* stream);
/*      same, for row-sharded tables (multi-GPU): Etgt [n,H] holds the target rows already fetched
 *      from their owners (row i for token i), lq_tgt [n] / lq_neg [K] the candidates' log-Q values
 *      (nullable); tgt / neg are GLOBAL item ids, used only for accidental-hit removal. */
int seqrec_sampled_softmax_ce_rows(float* ln, int64_t ld, const float* hd, int H, const float* Etgt,
                                   const float* lq_tgt, const float* lq_neg, const int32_t* tgt,
                                   const int32_t* neg, int64_t n, int K, float inv_denom,
                                   float* loss_rows, float* dlt, void* stream);
/*      the same with the target rows read THROUGH an index -- row i of the target table is table[tgt_row[i] * table_ld + :]
 *      (the received rows of the exchange buffer; no staging copy) */
int seqrec_sampled_softmax_ce_rows_idx(float* ln, int64_t ld, const float* hd, int H, const float* table, int64_t table_ld,
                                       const int32_t* tgt_row, const float* lq_tgt, const float* lq_neg, const int32_t* tgt,
                                       const int32_t* neg, int64_t n, int K, float inv_denom, float* loss_rows, float* dlt,
                                       void* stream);
/* ---- out[0] (+)= sum_i x[i], one workgroup, fixed summation order (deterministic) */
int seqrec_reduce_sum(const float* x, int64_t n, float* out, int accumulate, void* stream);

/* ---- column sum: out[j] (+)= sum_i X[i,j]   (bias gradients); two-stage, deterministic.
 *      workspace: 64*width floats. */
int seqrec_colsum(const float* X, int64_t n, int width, int64_t ld, float* out, int accumulate,
                  float* workspace, void* stream);
/* ---- y[i,:] = x[i,:] * m[i,:]  (Dropout layers, model.py:256,357,363,368,372); in place allowed */
int seqrec_mul(const float* x, const float* m, float* y, int64_t n, void* stream);
/* ---- fill n floats / n int32 */
int seqrec_fill_f32(float* x, float v, int64_t n, void* stream);
int seqrec_fill_i32(int32_t* x, int32_t v, int64_t n, void* stream);
/*      dst[dst_pos ? dst_pos[i] : i] = src[src_pos ? src_pos[i] : i] * mul + add   (int32; an entry with a
 *      negative position is skipped).  Index plumbing of the row exchange (distributed.py): places the
 *      per-step negative draws into the batch's routing lists and moves item ids through float
 *      buffers bit-exactly (no float arithmetic ever touches them). */
int seqrec_index_affine_i32(int32_t* dst, const int32_t* dst_pos, const int32_t* src, const int32_t* src_pos,
                            int64_t n, int32_t mul, int32_t add, void* stream);

/* ---- the cell of a sampled-softmax training step as ONE host call (ABI 5; step.hip).  What Keras' train_function does between
 *      the embedding lookup and the optimizer (model.py:179-182 -> Model.fit -> train_function) is a fixed
 *      sequence of the entry points above; issued one by one from Python it costs more host time than the GPU needs to run
 *      it once collectives share the step (distributed.py: 0.6 ms of host time for a 0.53 ms step).  seqrec_train_cell issues
 *      the SAME launches with the SAME arguments from one call -- results are bit-identical to the call-by-call sequence,
 *      which stays the specification (engine.Engine.train_step) and the form the per-kernel profile times.
 *      stages (bit mask):
 *        1 forward   [pack_u: seqrec_rnn_pack_u_stepwise, or with sample != 0 seqrec_rnn_pack_u_sample]  ->
 *                    XW = x . W + bias with the rows of x read through x_index (seqrec_gemm_f32_fused)  ->  seqrec_rnn_fwd_stepwise  ->
 *                    ln = Hout . Eneg^T  ->  seqrec_sampled_softmax_ce (logq_table / tgt_table by item id; lq_tgt NULL) or
 *                    seqrec_sampled_softmax_ce_rows_idx (target rows through tgt_index, lq_tgt per token)
 *        2 backward  dHd = dln . Eneg + dlt * tgt rows (seqrec_gemm_f32_fused, sk_dh splits)  ->  [deneg_mode 1: dEneg slabs; 3: dH and
 *                    dEneg together, seqrec_gemm_f32_pair]  ->
 *                    seqrec_rnn_bwd_stepwise  ->  the weight gradients dU (GRU: two column blocks), dW, db [, deneg_mode 2: dEneg] in one
 *                    grouped launch: wgrad_slabs != 0 leaves split-K slabs (seqrec_gemm_f32_grouped_slabs; descs_out / ns_wgrad
 *                    are what seqrec_opt_sqnorm_slabs needs), else reduced into dU / dW / db
 *        4 dX        dX = dPre . W^T as sk_dx split-K slabs (seqrec_gemm_f32_slabs) for the row scatter / the gradient routing
 *      All pointers are device pointers unless marked host; every workspace is the caller's.  Outputs ns_* / deneg_off /
 *      n_descs are written into the (host) plan. */
typedef struct seqrec_cell_plan {
    int32_t stages, cell, act, Hp, H_real, G, K, Dp, T, B, use_graph, reserved0_;
    int64_t n;
    const int32_t* step_off_host;                      /* host, T + 1 entries */
    /* prologue */
    int32_t pack_u, sample; uint64_t seed, step;
    const float* U; float* upack;
    const uint32_t* thresh; const int32_t* alias; int32_t V, reserved1_; const float* sample_table; const float* sample_logq;
    int32_t* neg_out; float* Eneg_out; float* lq_neg_out;
    /* batch != 0 (needs pack_u and sample): the batch itself is gathered in the prologue launch (seqrec_rnn_pack_u_sample_batch) */
    int32_t batch, reserved3_; const int32_t* flat; const int64_t* starts; const int32_t* sess_host;      /* sess_host: host, B entries */
    int32_t* sess_out; int32_t* step_off_out; int32_t* ids_out; int32_t* tgt_out; int32_t* prev_out;
    /* input projection */
    const float* x_table; int64_t x_ld; const int32_t* x_index; const float* W; const float* bias;
    float *XW, *Hout, *gates, *aux;
    /* output side */
    const float* Eneg; const int32_t* neg; const float* lq_neg;
    float *ln, *dlt, *loss_rows; float inv_denom; int32_t reserved2_;
    const float* tgt_table; int64_t tgt_ld; const int32_t* tgt_index; const int32_t* tgt_ids; const float* lq_tgt; const float* logq_table;
    /* backward */
    float* dHd; float* gemm_ws; int32_t sk_dh, deneg_mode, sk_deneg, sk_wgrad, wgrad_slabs, sk_dx;
    float* dEneg_slabs; float* dPre; float* scan_ws; const int32_t* prev;
    float *dU, *dW, *db; const float* ones; float* wgrad_ws; float* dX_slabs;
    /* outputs (host) */
    int32_t ns_deneg, ns_wgrad, ns_dx, n_descs; int64_t deneg_off;
    seqrec_gemm_desc descs_out[6];
} seqrec_cell_plan;
int seqrec_train_cell(seqrec_cell_plan* plan_host, void* stream);
int64_t seqrec_cell_plan_bytes(void);          /* sizeof(seqrec_cell_plan): bindings check their struct layout against it */

/* ---- row exchange of the multi-GPU step (distributed.py; SURVEY 8e: tables row-sharded, rows moved by all-to-all; no
 *      reference counterpart).  One launch on each side of the step's two collectives:
 *      exchange_pack (owner, before all-to-all #1): sendbuf[j,:] for the m_tot owner-side rows -- kinds[j] >= 0: table row
 *        kinds[j] (>= table_rows: zero row + SEQREC_STATUS_BAD_INDEX); -1: an id row; -2: a negative row.  got != NULL (ABI 5):
 *        kinds[j] >= 0 is an index into the request list `got` (got_len entries, received from the peers by all-to-all) and
 *        the table row is got[kinds[j]] -- the host never touches the received list (no per-batch index arithmetic on the device
 *        between the planning collective and this launch).  The rank's n_neg
 *        stratified draws (seqrec_sample_negatives(seed, step, n_neg, ...), per_peer per requester) are drawn here: draw i
 *        -> table row row_offset + id, copied to sendbuf[neg_slots[i]]; id row r of peer p (id_rows[p * n_id_rows / R + r])
 *        carries the draws' global ids id * id_mul + id_add bit-cast into the float buffer.  rows_eff[j] (m_tot) receives
 *        the table row every owner-side row stands for (-1: none) -- the scatter list of the returning gradients.
 *        bias_table / bias_out / bias_rows (nullable, together; ABI 5): the per-item OUTPUT BIAS of a sampled model
 *        (`Dense(n_classes)` of RNNBaseline, model.py:257) travels beside the rows -- bias_out[j] = bias_table[row - row_offset] for
 *        rows of the output table and for my draws, 0 elsewhere (one float per owner-side row, moved by an all-to-all with the
 *        rows' split sizes); bias_rows[j] = that bias row or -1: the scatter list of the bias gradients that return in the same
 *        positions (exchange_grad_pack bias_grad[j] = dlt of a target row, dbn[k] = sum_i dlogits[i,k] of negative k, else 0).
 *      exchange_unpack (requester, after it): Eneg[k,:] = recv[neg_rows[k],:], neg[k] = ((int32*)recv)[negid_idx[k]],
 *        lq_neg[k] = logq[neg[k]] (nullable; the ids were written by a peer: one outside [0, logq_rows) reads 0 and sets
 *        SEQREC_STATUS_BAD_INDEX in *status, ABI 5).
 *      exchange_grad_pack (requester, before all-to-all #2): out[j,:] for the n_tot requester-side rows, b = back_idx[j]:
 *        b < 0 zero; b < n: sum of the dx_slabs split-K slabs of dX row b; b < 2n: dlt[b-n] * Hd[b-n,:]; else the sum of the
 *        dn_slabs slabs of dEneg row b - 2n (seqrec_gemm_f32_slabs products: no reduce launch, no staging copy). */
int seqrec_exchange_pack(const float* table, int64_t table_rows, int width, const int32_t* kinds, const int32_t* got,
                         int64_t got_len, int64_t m_tot, uint64_t seed, uint64_t step, int n_neg, const uint32_t* thresh, const int32_t* alias, int V_local,
                         int32_t row_offset, const int32_t* neg_slots, const int32_t* id_rows, int n_id_rows, int per_peer,
                         int32_t id_mul, int32_t id_add, float* sendbuf, int32_t* rows_eff, uint32_t* status,
                         const float* bias_table, float* bias_out, int32_t* bias_rows, void* stream);
int seqrec_exchange_unpack(const float* recv, int width, const int32_t* neg_rows, const int32_t* negid_idx, int K,
                           const float* logq, int64_t logq_rows, float* Eneg, int32_t* neg, float* lq_neg, uint32_t* status,
                           void* stream);
int seqrec_exchange_grad_pack(const int32_t* back_idx, int64_t n_tot, int n, int K, int width, const float* dX, int dx_slabs,
                              int64_t dx_stride, const float* Hd, const float* dlt, const float* dEneg, int dn_slabs,
                              int64_t dn_stride, float* out, const float* dbn, float* bias_grad, void* stream);

/*      HOST routines (no device work, `_host` pointers throughout): the routing arithmetic of a batch of the unified step
 *      (distributed.py RowExchange.plan_unified) -- per-peer request counts; the per-peer lists of requested local rows and
 *      every request's position in owner-sorted order; the batch's whole int32 index blob (field order in csrc/route.hip),
 *      written in place into the caller's (page-locked) upload buffer.  route_blob returns the words written (< 0: error). */
int seqrec_route_count_host(const int32_t* ids_host, const int32_t* tgt_host, int64_t n, int R, int64_t* counts_host);
int seqrec_route_fill_host(const int32_t* ids_host, const int32_t* tgt_host, int64_t n, int R, int64_t V_in, int tied,
                           const int64_t* counts_host, const int64_t* base_host, int32_t* send_host, int32_t* req_rank_host);
int64_t seqrec_route_blob_host(const int32_t* step_off_host, int T, const int32_t* prev_host, const int32_t* ids_host,
                               const int32_t* tgt_host, int64_t n, int R, int Kr, int nid, int w, const int64_t* sc_host,
                               const int64_t* rc_host, const int32_t* req_rank_host, const int64_t* got_off_host,
                               float ntok, const float* lq_tgt_host, int32_t* blob_host, int64_t blob_len);

/* ---- row-sparse gradient path for the item tables (E, Eout, Wk, bout) -- the exact sparse
 *      equivalent of Keras' dense Adagrad (experiments_methods.py:41): a row with zero gradient is
 *      left untouched by the dense rule.  `gtab` is a table-shaped gradient accumulator that is all
 *      zero between steps; `slot` (int32[rows of table], all INT32_MAX between steps) elects one
 *      owner per touched row (the smallest contribution index).
 *      A contribution with rows[i] < 0 is a filler and is ignored by all three kernels.
 *      scatter_add: gtab[rows[i],:] += vals[i,:] * (row_scale ? row_scale[i] : 1)  (float atomics),
 *                   slot[rows[i]] = min(slot[rows[i]], base + i)
 *      sqnorm:      partial[blockIdx] = sum over owned rows of |gtab[row,:]|^2  (added to sq_accum)
 *      adagrad:     owner applies a += (s g)^2, p -= lr s g / (sqrt(a) + eps), then clears gtab row
 *                   and slot.  s = *scale (device scalar from seqrec_clip_scale). */
int seqrec_rows_scatter_add(float* gtab, int32_t* slot, const int32_t* rows, const float* vals,
                            int64_t ldv, const float* row_scale, int64_t n, int width, int32_t base,
                            void* stream);
int seqrec_rows_sqnorm(const float* gtab, const int32_t* slot, const int32_t* rows, int64_t n,
                       int width, int32_t base, float* sq_accum, void* stream);
int seqrec_rows_adagrad(float* table, float* accum, float* gtab, int32_t* slot, const int32_t* rows,
                        int64_t n, int width, int32_t base, float lr, float eps, const float* scale,
                        void* stream);

/*      multi-list forms: up to 4 scatter lists (possibly of different tables) per launch */
typedef struct seqrec_rows_job {
    float* table; float* accum; float* gtab; int32_t* slot;
    const int32_t* rows; const float* vals; int64_t ldv; const float* row_scale;
    int64_t n; int32_t width; int32_t base;
    int32_t n_slabs; int32_t reserved_; int64_t slab_stride;   /* n_slabs > 1: the values are the SUM over s < n_slabs of
                                                                  (vals + s * slab_stride)[i * ldv + :], added in that order
                                                                  (seqrec_gemm_f32_slabs); 0 or 1: vals alone.  Read by
                                                                  seqrec_rows_scatter_add_multi only (merge_sorted refuses > 1) */
} seqrec_rows_job;
int seqrec_rows_scatter_add_multi(const seqrec_rows_job* jobs_host, int count, void* stream);
/*      deterministic alternative to seqrec_rows_scatter_add_multi for the lists of ONE table (same gtab, slot
 *      and width; `base` ranges disjoint): contributions are keyed (row, base + i), sorted by row with a stable
 *      radix sort and summed per row in increasing index order with plain loads/stores -- no float atomics,
 *      bitwise reproducible (merge.hip).  Leaves gtab / slot exactly as the atomic form does
 *      (gtab[row] += sum, slot[row] = min(base + i)), so sqnorm / adagrad are shared.  The reference's dense
 *      Adagrad (experiments_methods.py:41) is deterministic; this is the path that matches that property. */
int64_t seqrec_rows_merge_workspace_bytes(int64_t n_total, int width);
int seqrec_rows_merge_sorted(const seqrec_rows_job* jobs_host, int count, void* workspace, int64_t workspace_bytes,
                             void* stream);
int seqrec_rows_sqnorm_multi(const seqrec_rows_job* jobs_host, int count, float* sq_accum, void* stream);
int seqrec_rows_adagrad_multi(const seqrec_rows_job* jobs_host, int count, float lr, float eps,
                              const float* scale, void* stream);

/* ---- dense tensors: sq_accum += |g|^2 ; scale = (norm >= clipnorm) ? clipnorm/norm : 1 (Keras
 *      clip_norm); Adagrad update (Keras optimizers.Adagrad, epsilon=1e-8, decay=0). */
int seqrec_sqnorm(const float* g, int64_t n, float* sq_accum, void* stream);
int seqrec_clip_scale(const float* sq_accum, float clipnorm, float* scale, void* stream);
int seqrec_adagrad_dense(float* p, float* a, const float* g, int64_t n, float lr, float eps,
                         const float* scale, void* stream);

/*      multi-tensor forms (<= 8 tensors per launch; host arrays of device pointers / sizes) */
int seqrec_sqnorm_multi(int count, const float* const* g, const int64_t* n, float* sq_accum, void* stream);
int seqrec_adagrad_dense_multi(int count, float* const* p, float* const* a, const float* const* g,
                               const int64_t* n, float lr, float eps, const float* scale, void* stream);

/* ---- fused optimizer step (clipnorm + Adagrad, experiments_methods.py:41) over up to 8 dense tensors and up
 *      to 4 scatter lists: ONE launch for the squared gradient norm of everything, ONE for the Keras clip
 *      scale + dense Adagrad + row-sparse Adagrad (same arithmetic as seqrec_sqnorm_multi /
 *      seqrec_rows_sqnorm_multi / seqrec_clip_scale / seqrec_adagrad_dense_multi / seqrec_rows_adagrad_multi).
 *      opt_sqnorm ADDS into *sq_accum (zero it first); opt_apply writes the scale to *scale_out and, if
 *      zero_next != NULL, stores 0 to *zero_next -- callers alternate two norm slots so that no step needs
 *      a separate clearing launch. */
int seqrec_opt_sqnorm(int n_dense, const float* const* grads, const int64_t* numel,
                      const seqrec_rows_job* jobs_host, int n_jobs, float* sq_accum,
                      const float* loss_rows, int64_t n_loss, float* loss_out, void* stream);
/*      opt_sqnorm_slabs: additionally up to 4 dense gradients that arrive as split-K slabs
 *      (seqrec_gemm_f32_grouped_slabs with the same descs, n_slabs and workspace; bias NULL, accumulate 0): the launch
 *      adds the slabs in slab order, WRITES every product to descs[i].C (row stride ldc) for seqrec_opt_apply and adds its
 *      squares to the norm -- the weight gradients' reduce launch rides in the norm launch.  Those tensors must NOT also be
 *      listed in `grads`; n_dense + n_products <= 8. */
int seqrec_opt_sqnorm_slabs(int n_dense, const float* const* grads, const int64_t* numel,
                            int n_products, const seqrec_gemm_desc* products_host, int n_slabs, const float* workspace,
                            const seqrec_rows_job* jobs_host, int n_jobs, float* sq_accum,
                            const float* loss_rows, int64_t n_loss, float* loss_out, void* stream);
/*      loss_out (nullable, 2 floats): the same launch also reduces the per-token losses of the CE kernels with one
 *      spare workgroup in a fixed order -- loss_out[0] = sum, loss_out[1] = sum / n_loss (the Keras token mean): no
 *      separate reduction launch in a training step.  seqrec_loss_reduce is that reduction on its own (evaluation). */
int seqrec_loss_reduce(const float* loss_rows, int64_t n, float* loss_out, void* stream);
/*      deterministic form of the norm (no float atomics): per-block partial sums into `partials`
 *      (seqrec_opt_sqnorm_ordered_floats() floats), added in index order by one block; *sq_out is overwritten
 *      (accumulate == 0) or added to (calls on one stream add in call order) */
int64_t seqrec_opt_sqnorm_ordered_floats(int n_dense, int n_jobs, int64_t max_job_rows);
int seqrec_opt_sqnorm_ordered(int n_dense, const float* const* grads, const int64_t* numel,
                              const seqrec_rows_job* jobs_host, int n_jobs, float* partials,
                              int64_t partials_floats, float* sq_out, int accumulate,
                              const float* loss_rows, int64_t n_loss, float* loss_out, void* stream);
int seqrec_opt_apply(int n_dense, float* const* params, float* const* accums, const float* const* grads,
                     const int64_t* numel, const seqrec_rows_job* jobs_host, int n_jobs, const float* sq,
                     float clipnorm, float lr, float eps, float* scale_out, float* zero_next,
                     const float* grad_div, uint32_t* status, const float* sq_extra, void* stream);
/*      sq_extra (nullable device scalar): a second part of the squared norm, added to *sq (multi-GPU: the all-reduced norm of
 *      the row gradients + the fixed-order norm of the dense gradients, kept apart until here).
 *      status (nullable device word): *sq not a finite number >= 0, *grad_div not a finite number > 0 or a clip scale that
 *      is 0 / not finite set SEQREC_STATUS_BAD_NORM / _BAD_DIVISOR / _BAD_SCALE and the launch changes NOTHING (weights,
 *      accumulators and gradient tables stay as they are; *scale_out still receives the scale): an overflowing or
 *      uninitialised gradient value can neither poison the weights nor turn the step into a silent no-op (scale 0).
 *      grad_div (nullable DEVICE scalar): the gradients in memory are sums still to be divided by *grad_div -- the
 *      global token count of a multi-GPU step, which only exists after the all-reduce; norm, clip scale and update
 *      then use g / *grad_div, so no step needs the count on the host.  *scale_out receives clip_scale / *grad_div. */

/* ---- counter RNG (specification: oracle/rng.py).  Alias-method draw of K negatives for
 *      training step `step`; inverted-dropout multipliers out[r*ld + j] (j < width) drawn with
 *      counter rowkey[r]*width + j (rowkey nullable -> r): 0 or 1/(1-rate). */
int seqrec_sample_negatives(uint64_t seed, uint64_t step, int K, const uint32_t* thresh,
                            const int32_t* alias, int V, int32_t* out, void* stream);
/*      the same draws, fused with the gathers every step does right after them: rows_out[k,:] =
 *      table[neg_out[k],:] and (logq_out != NULL) logq_out[k] = logq[neg_out[k]] -- one launch */
int seqrec_sample_gather(uint64_t seed, uint64_t step, int K, const uint32_t* thresh, const int32_t* alias,
                         int V, const float* table, int width, const float* logq, int32_t* neg_out,
                         float* rows_out, float* logq_out, void* stream);
/*      seqrec_rnn_pack_u_stepwise + seqrec_sample_gather in ONE launch (the two independent openers of a sampled-softmax
 *      training step): same arguments, same outputs as the two calls */
int seqrec_rnn_pack_u_sample(int cell, int H, const float* U, float* upack, uint64_t seed, uint64_t step, int K,
                             const uint32_t* thresh, const int32_t* alias, int V, const float* table, int width,
                             const float* logq, int32_t* neg_out, float* rows_out, float* logq_out, void* stream);
/*      ... + seqrec_pack_batch_host in the SAME launch (ABI 5): the three openers of a training step on a batch drawn from an
 *      HBM-resident data set -- batch gather, U re-pack, negatives -- as one launch; B + T + 1 <= 640 (kernel arguments) */
#define SEQREC_PACK_MERGED_MAX 640
int seqrec_rnn_pack_u_sample_batch(int cell, int H, const float* U, float* upack, uint64_t seed, uint64_t step, int K,
                                   const uint32_t* thresh, const int32_t* alias, int V, const float* table, int width,
                                   const float* logq, int32_t* neg_out, float* rows_out, float* logq_out,
                                   const int32_t* flat, const int64_t* starts, const int32_t* sess_host,
                                   const int32_t* step_off_host, int B, int T, int32_t* sess_out, int32_t* step_off_out,
                                   int32_t* ids, int32_t* tgt, int32_t* prev, void* stream);
int seqrec_dropout_mask(uint64_t seed, uint64_t stream_id, const int64_t* rowkey, int64_t n_rows,
                        int width, int64_t ld, double rate, float* out, void* stream);

/* ---- kernel regularizer of the y_to_y / to_y Dense layers: GaussPriorRegularizer (model.py:71-91,
 *      `K.sum(1/(2 var) * K.square(x - means))`) and keras.regularizers.l2 (means == NULL,
 *      strength = l).  grad (nullable) += 2 strength (w - means);  *loss_accum (nullable) += strength
 *      sum (w - means)^2 -- Keras adds the penalty to the reported loss, train and validation alike. */
int seqrec_prior_grad(const float* w, const float* means, int64_t n, float strength, float* grad,
                      float* loss_accum, void* stream);

/* ---- device-side ragged batcher: counterpart of the pairing x = s[i], y = s[i+1] of
 *      FullModelPreprocessor.transform_data (preprocessor.py:67-94) on ids, and of datasets.build_xs
 *      (datasets.py:97-113), for a dataset that lives in HBM as one flat id array
 *      (session i = flat[starts[i] .. starts[i+1])).  `sess[r]` = dataset index of the batch's r-th
 *      session in DESCENDING length order, `step_off` = device int32[T+1] packed-token offsets (the
 *      host computes both from the lengths alone; no item id crosses PCIe).
 *      pack_batch:        ids[p] = s[t], tgt[p] = s[t+1], prev[p] = token of the same session at t-1 (-1 at t = 0)
 *      history_features:  xs[p, v] = [v in s[0..t]]  (freq != 0: number of occurrences), row stride ld >= x_dim */
int seqrec_pack_batch(const int32_t* flat, const int64_t* starts, const int32_t* sess, const int32_t* step_off,
                      int B, int T, int32_t* ids, int32_t* tgt, int32_t* prev, void* stream);
/*      pack_batch_host: the same with `sess_host` / `step_off_host` read on the HOST and carried in the launch's
 *      kernel arguments (B + T + 1 <= SEQREC_PACK_HOST_MAX, else SEQREC_E_SHAPE): no host -> device copy precedes the
 *      launch.  The launch also writes both arrays to sess_out[B] / step_off_out[T+1] in HBM for later readers. */
#define SEQREC_PACK_HOST_MAX 960
int seqrec_pack_batch_host(const int32_t* flat, const int64_t* starts, const int32_t* sess_host,
                           const int32_t* step_off_host, int B, int T, int32_t* sess_out, int32_t* step_off_out,
                           int32_t* ids, int32_t* tgt, int32_t* prev, void* stream);
int seqrec_history_features(const int32_t* flat, const int64_t* starts, const int32_t* sess, const int32_t* step_off,
                            int B, int T, int x_dim, int64_t ld, int freq, float* xs, void* stream);

/* ---- Recall@K support (extension): rank[i] = #{v : score(i,v) > score(i,tgt_i)},
 *      score(i,v) = hd[i,:] . Eout[v,:] (+ bout[v]).  rank must be zeroed by the caller. */
int seqrec_rank_count(const float* hd, int H, const float* Eout, const float* bout,
                      const int32_t* tgt, int64_t n, int V, int32_t* rank, float* thr_workspace /* n floats */,
                      void* stream);

/*      thr[i] = hd[i,:] . Eout[tgt[i],:] (+ bout[tgt[i]])  -- the target score on its own */
int seqrec_target_score(const float* hd, int H, const float* Eout, const float* bout, const int32_t* tgt,
                        int64_t n, float* thr, void* stream);
/*      row-sharded tables: the caller supplies the target scores thr[i] (the target rows live on
 *      other ranks) and tgt_local[i] = the target's row in THIS shard or -1; rank[i] += the shard's
 *      count, so an all-reduce over the ranks gives the global rank. */
int seqrec_rank_count_thr(const float* hd, int H, const float* Eout, const float* bout,
                          const int32_t* tgt_local, const float* thr, int64_t n, int V, int32_t* rank,
                          void* stream);

/* ---- top-K prediction at catalogue scale (extension; the reference's predict returns a dense (N,T,V)
 *      tensor, model.py:186-190, which does not exist at |items| = 1M).  state_val / state_idx [n,64]
 *      hold a running top-64 per row (initialise to -inf / -1); topk_merge folds in one chunk
 *      scores[n, width] (+ bias[col0 + c]) of item columns col0 .. col0+width; topk_finish writes the k <= 64
 *      best, sorted, to out_val / out_idx [n,k]. */
int seqrec_topk_merge(const float* scores, int64_t ld, int64_t n, int width, int col0, const float* bias,
                      float* state_val, int32_t* state_idx, void* stream);
int seqrec_topk_finish(const float* state_val, const int32_t* state_idx, int64_t n, int k, float* out_val,
                       int32_t* out_idx, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEQREC_HIP_H */
