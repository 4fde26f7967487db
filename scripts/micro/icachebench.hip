// icachebench.hip — does a kernel start on a cold instruction cache at EVERY launch?  One wave per CU runs N straight-line dependent
// VALU instructions (no loop: the code IS N instructions long) once or twice; the second pass of the same launch runs from a warm
// cache, the first pass of a later launch shows whether the cache survived the launch boundary.  Times from s_memrealtime inside the wave.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o icachebench icachebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// eight independent accumulators: a wave issues these back to back (~4-5 cycles each), i.e. a 64-byte line of code every ~35 cycles — a
// dependent chain (8-9 cycles per instruction) would leave any fetch latency hidden behind itself
template <int N> __device__ __forceinline__ float body(float x) {
    float a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
        asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                     "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(1.0001f), "v"(0.5f));
    return ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
}
template <int N> __global__ void straight(float* out, long long* t, int passes) {
    float x = threadIdx.x * 1e-3f;
    long long w[3];
    w[0] = wall_clock64();
    x = body<N>(x);
    w[1] = wall_clock64();
    if (passes > 1) {
        asm volatile("s_nop 0" ::: "memory");
        for (int p = 1; p < passes; ++p) { asm volatile("" : "+v"(x)); x = body<N>(x); }   // (the same code again only if the compiler keeps one copy: see the note printed)
    }
    w[2] = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = w[1] - w[0]; t[blockIdx.x * 2 + 1] = w[2] - w[1]; }
}
// one copy of the code, run `passes` times by a loop around a noinline function
template <int N> __device__ __noinline__ float body_fn(float x) { return body<N>(x); }
template <int N> __global__ void looped(float* out, long long* t, int passes) {
    float x = threadIdx.x * 1e-3f;
    long long first = 0, rest = 0;
    for (int p = 0; p < passes; ++p) {
        const long long a = wall_clock64();
        x = body_fn<N>(x);
        const long long b = wall_clock64();
        if (p == 0) first = b - a; else rest += b - a;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = first; t[blockIdx.x * 2 + 1] = passes > 1 ? rest / (passes - 1) : 0; }
}
template <int N> void run(float* out, long long* t, int waves_per_cu) {
    std::vector<long long> h(512 * 8);
    const int grid = 256 * waves_per_cu;
    double f = 0, r = 0;
    const int reps = 30;
    for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(looped<N>, dim3(grid), dim3(64), 0, 0, out, t, 3);
    hipDeviceSynchronize();
    for (int k = 0; k < reps; ++k) {
        hipLaunchKernelGGL(looped<N>, dim3(grid), dim3(64), 0, 0, out, t, 3);   // back to back: launch k starts right after launch k - 1
    }
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, grid * 16, hipMemcpyDeviceToHost);
    for (int b = 0; b < grid; ++b) { f += h[2 * b]; r += h[2 * b + 1]; }
    printf("%6d instructions (%4d KB), %d wave(s) per CU: first pass of a launch %.2f us, later passes of the same launch %.2f us -> cold start costs %.2f us (%.1f ns per 64-byte line)\n",
           N, N * 8 / 1024, waves_per_cu, f / grid / 100.0, r / grid / 100.0, (f - r) / grid / 100.0, (f - r) / grid * 10.0 / (N * 8 / 64.0));
}
int main() {
    float* out; long long* t;
    hipMalloc(&out, 256 * 8 * 64 * 4); hipMalloc(&t, 256 * 8 * 2 * 8);
    for (int w : {1, 4}) { run<512>(out, t, w); run<2048>(out, t, w); run<4096>(out, t, w); run<8192>(out, t, w); }
    return 0;
}
