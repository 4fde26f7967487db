// Microbenchmark (profiling aid, not product): VALU issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_cndmask / v_cmp
// with every SIMD holding W waves; reports the implied clock x lanes product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int N = 4096;   // instructions per chain set

__global__ void k_fma(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < N / 4; ++i) { x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b); }
    if (x0 + x1 + x2 + x3 == 12345.f) out[0] = x0;
}
__global__ void k_pk(float* out, float a, float b) {
    v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
    const v2f aa = {a, a}, bb = {b, b};
    for (int i = 0; i < N / 4; ++i) { x0 = __builtin_elementwise_fma(x0, aa, bb); x1 = __builtin_elementwise_fma(x1, aa, bb); x2 = __builtin_elementwise_fma(x2, aa, bb); x3 = __builtin_elementwise_fma(x3, aa, bb); }
    v2f s = x0 + x1 + x2 + x3;
    if (s.x + s.y == 12345.f) out[0] = s.x;
}
__global__ void k_dep(float* out, float a, float b) {   // one dependent chain
    float x0 = threadIdx.x;
    for (int i = 0; i < N; ++i) x0 = __builtin_fmaf(x0, a, b);
    if (x0 == 12345.f) out[0] = x0;
}
__global__ void k_divd(float* out, float a, float b) {   // x / c as (float)((double)x * (1/c)): cvt, v_mul_f64, cvt
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const double rc = 1.0 / (double)a;
    for (int i = 0; i < N / 4; ++i) { x0 = (float)((double)x0 * rc) + b; x1 = (float)((double)x1 * rc) + b; x2 = (float)((double)x2 * rc) + b; x3 = (float)((double)x3 * rc) + b; }
    if (x0 + x1 + x2 + x3 == 12345.f) out[0] = x0;
}
__global__ void k_divf(float* out, float a, float b) {   // 3-op exact form: mul, fma, fma
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    const float rc = 1.0f / a;
    for (int i = 0; i < N / 4; ++i) {
        float q0 = x0 * rc, q1 = x1 * rc, q2 = x2 * rc, q3 = x3 * rc;
        x0 = __builtin_fmaf(__builtin_fmaf(-q0, a, x0), rc, q0) + b; x1 = __builtin_fmaf(__builtin_fmaf(-q1, a, x1), rc, q1) + b;
        x2 = __builtin_fmaf(__builtin_fmaf(-q2, a, x2), rc, q2) + b; x3 = __builtin_fmaf(__builtin_fmaf(-q3, a, x3), rc, q3) + b;
    }
    if (x0 + x1 + x2 + x3 == 12345.f) out[0] = x0;
}
__global__ void k_sqrt(float* out, float a, float b) {
    float x0 = threadIdx.x + 1.f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    for (int i = 0; i < N / 4; ++i) { x0 = __builtin_sqrtf(x0 + a); x1 = __builtin_sqrtf(x1 + a); x2 = __builtin_sqrtf(x2 + a); x3 = __builtin_sqrtf(x3 + a); }
    if (x0 + x1 + x2 + x3 == 12345.f) out[0] = x0;
}

int main() {
    float* out; CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("CUs %d clock %d kHz\n", p.multiProcessorCount, p.clockRate);
    auto run = [&](const char* name, auto kern, int wavesPerSimd, double instr) {
        const int blocks = p.multiProcessorCount * wavesPerSimd;   // 256-thread blocks: 4 waves = 1 per SIMD each
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        const int iters = 20;
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        // per SIMD: wavesPerSimd waves x instr instructions; cycles per instr at clock f: us * f / (w * instr)
        printf("%-28s w/SIMD=%d  %8.2f us  -> %.2f ns per wave-instruction per SIMD (4 cycles @2.4GHz = 1.67 ns)\n", name, wavesPerSimd, us, us * 1e3 / (wavesPerSimd * instr));
    };
    for (int w : {1, 2, 4, 8}) {
        run("v_fma_f32 x4 chains", k_fma, w, N);
        run("v_pk_fma_f32 x4 chains", k_pk, w, N);
        run("v_fma_f32 dependent", k_dep, w, N);
        run("v_sqrt_f32(+add) x4", k_sqrt, w, N);
        run("div via f64 mul (+add) x4", k_divd, w, N);
        run("div 3-op f32 (+add) x4", k_divf, w, N);
    }
    return 0;
}
