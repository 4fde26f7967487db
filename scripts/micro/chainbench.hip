// Microbenchmark: what does a chain of DEPENDENT kernel launches cost per link on this GPU, by launch form?
//   eager        hipLaunchKernelGGL back to back on one stream (the host enqueues every packet)
//   graph        the same chain captured once into a hipGraph and replayed (one host call per replay)
//   graph-noflush ... with kernels that carry no memory traffic at all (nothing dirty in L2 at the boundary)
// and by geometry (1 workgroup / 512 x 5 waves / 1024 x 5 waves) and by how much the kernel writes (0 / 37 MB: the
// end-of-kernel release has to write dirty L2 lines back before the next packet may start).
// Usage: ./chainbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_empty(float* out, int n) {
    if (n == 12345) out[blockIdx.x] = 1.0f;
}
// every thread copies `per` float4 from src to dst (coalesced) — a kernel that leaves grid * block * per * 16 bytes dirty
__global__ void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, int per) {
    const size_t base = (size_t)blockIdx.x * blockDim.x * per + threadIdx.x;
    for (int i = 0; i < per; ++i) dst[base + (size_t)i * blockDim.x] = src[base + (size_t)i * blockDim.x];
}

struct Case { const char* name; int grid, block, per; };

static float time_eager(const Case& c, float4* a, float4* b, int chain, int iters, hipStream_t s) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto one = [&](int i) {
        if (c.per == 0) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, s, (float*)a, 1);
        else hipLaunchKernelGGL(k_copy, dim3(c.grid), dim3(c.block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, c.per);
    };
    for (int i = 0; i < chain; ++i) one(i);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < iters; ++r) for (int i = 0; i < chain; ++i) one(i);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / (iters * chain);
}

static float time_graph(const Case& c, float4* a, float4* b, int chain, int iters, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ex;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < chain; ++i) {
        if (c.per == 0) hipLaunchKernelGGL(k_empty, dim3(c.grid), dim3(c.block), 0, s, (float*)a, 1);
        else hipLaunchKernelGGL(k_copy, dim3(c.grid), dim3(c.block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, c.per);
    }
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraphLaunch(ex, s); hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < iters; ++r) hipGraphLaunch(ex, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
    return ms * 1e3f / (iters * chain);
}

int main() {
    const size_t bytes = 512u << 20;
    float4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const Case cases[] = {
        {"1 x 64, no traffic", 1, 64, 0},
        {"512 x 320, no traffic", 512, 320, 0},
        {"1024 x 320, no traffic", 1024, 320, 0},
        {"512 x 320, copy 18.9 MB (37.7 MB moved)", 512, 320, 7},      // ~ the 32 768-env shard's bytes
        {"1024 x 320, copy 36.7 MB (73.4 MB moved)", 1024, 320, 7},    // ~ the headline's bytes
        {"2048 x 320, copy 36.7 MB", 2048, 320, 4},
        {"1024 x 640, copy 36.7 MB", 1024, 640, 4},
        {"512 x 640, copy 36.7 MB", 512, 640, 7},
        {"256 x 1024, copy 36.7 MB", 256, 1024, 9},
    };
    for (const Case& c : cases) {
        const float e = time_eager(c, a, b, 25, 40, s), g = time_graph(c, a, b, 25, 40, s);
        const double mb = 2.0 * c.grid * c.block * (double)c.per * 16 / 1e6;
        printf("%-44s eager %6.2f us  graph %6.2f us per link", c.name, e, g);
        if (c.per) printf("   (%.1f MB moved: %.2f / %.2f TB/s)", mb, mb / e, mb / g);
        printf("\n");
    }
    return 0;
}
