// Cluster form of the GRU scan (rnn_cluster.hip): internal interface used by the seqrec_rnn_*_stepwise entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// true: the call was taken (rc set); false: not applicable -> the caller issues the step-wise plan
bool seqrec_cluster_gru_fwd(int act, int H, int H_real, int T, const int32_t* step_off_host, const float* XW, float* Hout,
                            float* gates, float* aux, const float* upack, hipStream_t st, int* rc);
// parts (nullable): dHout is given as split-K slabs + a row term (seqrec_dh_parts); the kernel adds them where it reads dHout
struct seqrec_dh_parts;
bool seqrec_cluster_gru_bwd(int act, int H, int H_real, int T, const int32_t* step_off_host, const float* dHout, const float* Hout,
                            const float* gates, const float* aux, float* dPre, const float* upack, hipStream_t st, int* rc,
                            const seqrec_dh_parts* parts = nullptr);
