// mfma_valu_overlap.hip — can ONE wave hide VALU work under its own f32 MFMAs?  A wave issues v_mfma_f32_32x32x2_f32 (64 cycles on the
// matrix core) on four independent accumulators and, per MFMA, M independent v_fma_f32; one wave per SIMD (256 threads, one block
// per CU).  If the time per MFMA stays at the M = 0 value up to M ~ 12, the VALU work rides for free (the design question behind
// "activate one column pair in registers under the other pair's k-loop", eb_policy.hip).  Also: two waves per SIMD, one doing only
// MFMAs and one only VALU (W2 rows) — the co-resident-block case.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int M>
__global__ __launch_bounds__(256) void k_mix(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int v = 0; v < 16; ++v) acc[q][v] = threadIdx.x * 1e-3f + q;
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-4f + j;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < M; ++j) x[j] = __builtin_fmaf(x[j], a, b);
        }
    }
    const long long t1 = clock64();
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int v = 0; v < 16; ++v) s += acc[q][v];
    for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) reinterpret_cast<long long*>(out + 1048576)[blockIdx.x] = t1 - t0;
}
// waves 0-3 of a 512-thread block: MFMA only; waves 4-7: VALU only (V fmas per "slot") — two waves per SIMD with different work
template <int V>
__global__ __launch_bounds__(512) void k_two(float* out, int iters, float a, float b) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int v = 0; v < 16; ++v) acc[q][v] = threadIdx.x * 1e-3f + q;
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-4f + j;
    const bool mf = threadIdx.x < 256;
    const long long t0 = clock64();
    if (mf) {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q], 0, 0, 0);
    } else {
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < V; ++j) x[j] = __builtin_fmaf(x[j], a, b);
    }
    const long long t1 = clock64();
    float s = 0;
    for (int q = 0; q < 4; ++q) for (int v = 0; v < 16; ++v) s += acc[q][v];
    for (int j = 0; j < 16; ++j) s += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 255) == 0) reinterpret_cast<long long*>(out + 1048576)[blockIdx.x * 2 + (threadIdx.x >> 8)] = t1 - t0;
}
template <class F> void run(const char* name, F launch, float* out, int iters, int nrec, int per) {
    launch(); hipDeviceSynchronize();
    launch(); hipDeviceSynchronize();
    long long h[1024];
    hipMemcpy(h, out + 1048576, nrec * sizeof(long long), hipMemcpyDeviceToHost);
    for (int k = 0; k < per; ++k) {
        double s = 0; int n = 0;
        for (int i = k; i < nrec; i += per) { s += h[i]; ++n; }
        printf("%s%s: %.1f cycles per MFMA slot\n", name, per == 2 ? (k == 0 ? " [MFMA wave]" : " [VALU wave]") : "", s / n / iters / 4.0);
    }
}
int main() {
    float* out; hipMalloc(&out, (1048576 + 4096) * 4);
    const int iters = 4000;
#define MIX(M) run("one wave per SIMD, " #M " fma per MFMA", [&] { hipLaunchKernelGGL(k_mix<M>, dim3(256), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); }, out, iters, 256, 1)
    MIX(0); MIX(2); MIX(4); MIX(8); MIX(12); MIX(16);
#define TWO(V) run("two waves per SIMD, MFMA wave + VALU wave with " #V " fma per slot", [&] { hipLaunchKernelGGL(k_two<V>, dim3(256), dim3(512), 0, 0, out, iters, 1.0001f, 0.5f); }, out, iters, 512, 2)
    TWO(0); TWO(4); TWO(8); TWO(12); TWO(16);
    return 0;
}
