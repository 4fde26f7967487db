// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 and the shader clock it runs at (s_memtime / s_memrealtime).
// Usage: ./mfmabench [waves_per_simd=2] [iters=20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(int iters, float* out, long long* clk) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + (float)blockIdx.x * 1e-6f;
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    float s = 0;
    for (int k = 0; k < 16; ++k) s += a0[k] + a1[k] + a2[k] + a3[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 20000;
    const int blocks = 256 * wps;   // 4 waves per block -> wps waves per SIMD on 256 CUs
    float* out; long long* clk;
    hipMalloc(&out, sizeof(float) * blocks * 256); hipMalloc(&clk, sizeof(long long) * 2 * blocks);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, iters, out, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
        const double flop = 4096.0 * 4 * iters * 4.0 * blocks;   // per MFMA 32*32*2*2 flop, 4 per iteration, 4 waves per block
        printf("waves/SIMD %d: %.1f us, %.1f TFLOP/s; shader clocks %lld over %.2f us -> %.0f MHz; %.1f clocks per MFMA per SIMD\n",
               wps, ms * 1e3, flop / ms / 1e9, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / (4.0 * iters * wps));
    }
    return 0;
}
