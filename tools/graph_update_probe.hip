// Probe (developer tool, GPU box): what does it cost to replay a captured chain of N dependent kernel launches when EVERY
// node's launch geometry and kernel arguments are rewritten before each replay (hipGraphExecKernelNodeSetParams) --
// against issuing the same N launches eagerly?  Decides whether the scan's per-batch launches can ride a graph without
// reading their geometry from a device table.
//   hipcc --offload-arch=gfx950 -O3 tools/graph_update_probe.hip -o tools/bin/graph_update_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
struct Args { int v[36]; float* p; };           // ~150 bytes, like StepArgs
__global__ void k(Args a) {
    if (a.v[0] < 0) a.p[blockIdx.x] = 1.f;       // never true: the kernel is an empty body with real arguments
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int N = 130, REP = 300;
    float* buf; CK(hipMalloc(&buf, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Args a = {}; a.p = buf;
    auto eager = [&](int rep) { for (int i = 0; i < N; ++i) { a.v[1] = i + rep; hipLaunchKernelGGL(k, dim3(256 + (i & 3) * 256), dim3(256), 0, st, a); } };
    eager(0); CK(hipStreamSynchronize(st));
    double t0 = now(); for (int r = 0; r < REP; ++r) eager(r); double th = now() - t0; CK(hipStreamSynchronize(st)); double tw = now() - t0;
    printf("eager:            host %.1f us / chain   wall %.1f us / chain   (%.2f us per launch)\n", th / REP * 1e6, tw / REP * 1e6, tw / REP / N * 1e6);
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal)); eager(0); CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    size_t nn = 0; CK(hipGraphGetNodes(g, nullptr, &nn)); std::vector<hipGraphNode_t> nodes(nn); CK(hipGraphGetNodes(g, nodes.data(), &nn));
    printf("graph nodes: %zu\n", nn);
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
    t0 = now(); for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ex, st)); th = now() - t0; CK(hipStreamSynchronize(st)); tw = now() - t0;
    printf("graph replay:     host %.1f us / chain   wall %.1f us / chain   (%.2f us per node)\n", th / REP * 1e6, tw / REP * 1e6, tw / REP / N * 1e6);
    // rewrite every node before each replay
    std::vector<Args> args(nn, a); std::vector<void*> ptrs(nn);
    t0 = now(); double tupd = 0;
    for (int r = 0; r < REP; ++r) {
        double u0 = now();
        for (size_t i = 0; i < nn; ++i) {
            hipKernelNodeParams p = {};
            args[i].v[1] = (int)i + r; ptrs[i] = &args[i];
            p.func = (void*)k; p.gridDim = dim3(256 + ((i + r) & 3) * 256); p.blockDim = dim3(256); p.sharedMemBytes = 0;
            p.kernelParams = &ptrs[i]; p.extra = nullptr;
            CK(hipGraphExecKernelNodeSetParams(ex, nodes[i], &p));
        }
        tupd += now() - u0;
        CK(hipGraphLaunch(ex, st));
    }
    th = now() - t0; CK(hipStreamSynchronize(st)); tw = now() - t0;
    printf("graph + rewrite:  host %.1f us / chain (%.1f us of it in SetParams = %.2f us per node)   wall %.1f us / chain\n",
           th / REP * 1e6, tupd / REP * 1e6, tupd / REP / nn * 1e6, tw / REP * 1e6);
    return 0;
}
