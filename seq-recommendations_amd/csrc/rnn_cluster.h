// Cluster form of the recurrent scans (rnn_cluster.hip: GRU; rnn_cluster2.hip: LSTM, SimpleRNN): internal interface used by
// the seqrec_rnn_*_stepwise entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct seqrec_dh_parts;
// One scan call in cluster form.  true: the call was taken (*rc set); false: not applicable (switched off, T out of range,
// the kernel's workgroups cannot all be resident, recurrent dropout on a shape whose masks do not fit the registers) -> the
// caller issues the step-wise plan.  rmask (nullable): recurrent-dropout multipliers [G][B][H] of the sorted session rows.
// parts (nullable, GRU BPTT only): dHout is given as split-K slabs + a row term; the kernel adds them where it reads dHout.
bool seqrec_cluster_fwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* step_off_host, const float* XW,
                        float* Hout, float* gates, float* aux, const float* upack, const float* rmask, hipStream_t st, int* rc);
bool seqrec_cluster_bwd(int cell, int act, int H, int H_real, int T, int B, const int32_t* step_off_host, const float* dHout,
                        const float* Hout, const float* gates, const float* aux, float* dPre, const float* upack,
                        const float* rmask, hipStream_t st, int* rc, const seqrec_dh_parts* parts = nullptr);
// frees what the cluster scans keep for `st` (flag buffer); the stream must be idle
void seqrec_cluster_release_stream(hipStream_t st);
