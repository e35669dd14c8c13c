// helper library of tools/cumask_probe.py (built by the tool): CU-masked streams and an XCD histogram kernel
#include <hip/hip_runtime.h>
#include <stdint.h>
#define SEQREC_E_ARG (-1)
#define SEQREC_LAUNCH_CHECK() do { hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess) return (int)e__; } while (0)
static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

namespace {
__global__ void xcc_count_kernel(int* __restrict__ counts) {
    if (threadIdx.x == 0) atomicAdd(counts + (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) & 0xF), 1);
}
}  // namespace

// cu_mask_host: bit i of word i / 32 = CU i may run the stream's kernels (n_words * 32 >= the device's CU count)
extern "C" int seqrec_stream_create_masked(const uint32_t* cu_mask_host, int n_words, void** stream_out) {
    if (!cu_mask_host || n_words < 1 || !stream_out) return SEQREC_E_ARG;
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, cu_mask_host);
    if (e != hipSuccess) return (int)e;
    *stream_out = reinterpret_cast<void*>(st);
    return 0;
}
extern "C" int seqrec_stream_destroy(void* stream) {
    if (!stream) return SEQREC_E_ARG;
    return (int)hipStreamDestroy(as_stream(stream));
}
// diagnostics: counts8[x] += workgroups of an n_workgroups launch that ran on XCD x (how a CU mask maps to XCDs)
extern "C" int seqrec_debug_xcc_count(int n_workgroups, int32_t* counts8, void* stream) {
    if (n_workgroups < 1 || !counts8) return SEQREC_E_ARG;
    hipLaunchKernelGGL(xcc_count_kernel, dim3((unsigned)n_workgroups), dim3(64), 0, as_stream(stream), counts8);
    SEQREC_LAUNCH_CHECK();
    return 0;
}
