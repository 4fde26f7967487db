// Microbenchmark: cost of dispatching a grid of (nearly) empty workgroups as a function of its geometry — how much of
// a ~16 us kernel is the dispatcher walking 1024 workgroups of 5 waves with 30 KB of LDS each?
// Usage: ./dispatchbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int VGPRS>
__global__ void empty_kernel(float* out, int n) {
    extern __shared__ float lds[];
    // touch LDS and keep VGPRS registers allocated so the workgroup needs its resources
    float acc = 0.f;
    if (n < 0) {
        float v[VGPRS];
        for (int i = 0; i < VGPRS; ++i) v[i] = out[i + threadIdx.x];
        for (int i = 0; i < VGPRS; ++i) acc += v[i] * v[(i * 7) % VGPRS];
        lds[threadIdx.x] = acc;
        out[threadIdx.x] = lds[(threadIdx.x + 1) % blockDim.x];
    }
    if (threadIdx.x == 0 && n == 12345) out[blockIdx.x] = acc;
}

template <int V>
static void run(const char* name, int grid, int block, int lds, float* out) {
    if (lds > 48 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(&empty_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(empty_kernel<V>, dim3(grid), dim3(block), lds, 0, out, 1);
    hipDeviceSynchronize();
    const int iters = 2000;
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(empty_kernel<V>, dim3(grid), dim3(block), lds, 0, out, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s grid %5d x %4d threads, %3d KB LDS: %.2f us per launch\n", name, grid, block, lds / 1024, ms * 1e3 / iters);
}

int main() {
    float* out; hipMalloc(&out, 1 << 20);
    run<16>("1 block", 1, 64, 0, out);
    run<16>("256 x 1 wave", 256, 64, 0, out);
    run<16>("1024 x 1 wave", 1024, 64, 0, out);
    run<16>("1024 x 5 waves", 1024, 320, 0, out);
    run<16>("1024 x 5 waves, 30 KB LDS", 1024, 320, 30 * 1024, out);
    run<80>("1024 x 5 waves, 30 KB LDS, 80 VGPRs", 1024, 320, 30 * 1024, out);
    run<80>("512 x 9 waves, 60 KB LDS, 80 VGPRs", 512, 576, 60 * 1024, out);
    run<80>("256 x 16 waves, 120 KB LDS, 80 VGPRs", 256, 1024, 120 * 1024, out);
    run<80>("2048 x 5 waves, 30 KB LDS, 80 VGPRs", 2048, 320, 30 * 1024, out);
    run<80>("4096 x 5 waves, 30 KB LDS, 80 VGPRs", 4096, 320, 30 * 1024, out);
    return 0;
}
