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

template <int ACT> __device__ __forceinline__ float act_fwd(float x) {
    if (ACT == SEQREC_ACT_RELU) return fmaxf(x, 0.0f);
    if (ACT == SEQREC_ACT_TANH) return tanhf(x);
    return x;
}
// derivative through the OUTPUT y = act(pre)   (relu'(0) := 0)
template <int ACT> __device__ __forceinline__ float act_grad(float y) {
    if (ACT == SEQREC_ACT_RELU) return y > 0.0f ? 1.0f : 0.0f;
    if (ACT == SEQREC_ACT_TANH) return 1.0f - y * y;
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
