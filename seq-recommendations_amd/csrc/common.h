// Shared device/host helpers for libseqrec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/seqrec_hip.h"
#include <stdlib.h>

// Developer switches (A/B and tuning knobs of tools/) are read from the environment ONLY in a -DSEQREC_TUNABLES build
// (tools/build_diag.py tunables -> tools/diag/, selected with SEQREC_LIB); the product library reads no environment
// variable: every switch is its default, a compile-time constant (include/seqrec_hip.h, "Hidden state").
#ifdef SEQREC_TUNABLES
static inline long seqrec_env(const char* name, long dflt) { const char* v = getenv(name); return v ? atol(v) : dflt; }
#else
static inline constexpr long seqrec_env(const char*, long dflt) { return dflt; }
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SEQREC_LAUNCH_CHECK()                                  \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return (int)e__;                \
    } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Keras hard_sigmoid: clip(0.2 x + 0.5, 0, 1)
__device__ __forceinline__ float hard_sigmoid(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.0f), 1.0f); }
// derivative expressed through the gate value g = hard_sigmoid(pre): 0.2 on the open ramp
__device__ __forceinline__ float hard_sigmoid_grad(float g) { return (g > 0.0f && g < 1.0f) ? 0.2f : 0.0f; }

// The cell activation (Keras `activation=` of SimpleRNN / LSTM / GRU; model.py:324,346,351 pass any Keras name through).  relu / tanh /
// linear are template instances of every scan kernel; the rest of Keras 2.0's element-wise list -- sigmoid, hard_sigmoid, softplus,
// softsign, elu -- shares ONE instance (SEQREC_ACT_OTHER) of the step-wise kernels with the kind as a run-time value `rt`
// (SEQREC_ACT_SIGMOID ...): nobody's hot path, one third more kernels instead of 2.7x.
constexpr int SEQREC_ACT_OTHER = 3;
template <int ACT> __device__ __forceinline__ float act_fwd(float x, int rt = 0) {
    if (ACT == SEQREC_ACT_RELU) return fmaxf(x, 0.0f);
    if (ACT == SEQREC_ACT_TANH) return tanhf(x);
    if (ACT == SEQREC_ACT_OTHER) {
        switch (rt) {
            case SEQREC_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
            case SEQREC_ACT_HARD_SIGMOID: return hard_sigmoid(x);
            case SEQREC_ACT_SOFTPLUS: return x > 20.0f ? x : log1pf(expf(x));
            case SEQREC_ACT_SOFTSIGN: return x / (1.0f + fabsf(x));
            case SEQREC_ACT_ELU: return x > 0.0f ? x : expm1f(x);
        }
    }
    return x;
}
// derivative through the OUTPUT y = act(pre)   (relu'(0) := 0; every activation here has a derivative that is a function of y)
template <int ACT> __device__ __forceinline__ float act_grad(float y, int rt = 0) {
    if (ACT == SEQREC_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (ACT == SEQREC_ACT_TANH) return 1.0f - y * y;
    if (ACT == SEQREC_ACT_OTHER) {
        switch (rt) {
            case SEQREC_ACT_SIGMOID: return y * (1.0f - y);
            case SEQREC_ACT_HARD_SIGMOID: return hard_sigmoid_grad(y);
            case SEQREC_ACT_SOFTPLUS: return 1.0f - expf(-y);                       // sigmoid(x) with e^x = e^y - 1
            case SEQREC_ACT_SOFTSIGN: { const float u = 1.0f - fabsf(y); return u * u; }   // 1 / (1 + |x|)^2 with |x| = |y| / (1 - |y|)
            case SEQREC_ACT_ELU: return y > 0.0f ? 1.0f : y + 1.0f;                // e^x = y + 1 for x <= 0
        }
    }
    return 1.0f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// splitmix64 finaliser -- specification in oracle/rng.py
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ uint64_t key64(uint64_t seed, uint64_t stream) {
    return mix64(((seed + 1) * 0x9E3779B97F4A7C15ULL) ^ ((stream + 1) * 0xD1B54A32D192ED03ULL));
}
__host__ __device__ __forceinline__ uint64_t rand64(uint64_t key, uint64_t ctr) {
    return mix64(key + (ctr + 1) * 0x9E3779B97F4A7C15ULL);
}
