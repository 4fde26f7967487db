// clockbench.hip — what shader clock does a SHORT kernel see?  s_memtime (shader cycles) against s_memrealtime (100 MHz) around a
// dependent VALU chain of known length, for one block per CU: (a) launches separated by idle gaps, (b) back-to-back launches.
// Build: hipcc --offload-arch=gfx950 -O3 -o clockbench clockbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>
__global__ void chain(int n, float* out, long long* t) {
    const long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x * 1e-3f;
    for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;      // one dependent v_fma / v_mul+v_add per iteration (contract off: 2)
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0) { t[blockIdx.x * 2] = c1 - c0; t[blockIdx.x * 2 + 1] = w1 - w0; }
}
int main() {
    float* out; long long* t;
    hipMalloc(&out, 256 * 64 * 4); hipMalloc(&t, 256 * 2 * 8);
    std::vector<long long> h(512);
    for (int mode = 0; mode < 3; ++mode) {
        for (int n : {500, 2000, 8000, 64000}) {
            double cyc = 0, wall = 0;
            const int reps = 20;
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) usleep(2000);
                if (mode == 2) for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(chain, dim3(256), dim3(64), 0, 0, 64000, out, t);
                hipLaunchKernelGGL(chain, dim3(256), dim3(64), 0, 0, n, out, t);
                hipDeviceSynchronize();
                hipMemcpy(h.data(), t, 512 * 8, hipMemcpyDeviceToHost);
                for (int b = 0; b < 256; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
            }
            printf("%s n=%6d: %.0f shader cycles, %.2f us -> %.0f MHz, %.2f cycles per iteration\n",
                   mode == 0 ? "after 2 ms idle  " : mode == 1 ? "after a sync     " : "after 50 launches", n, cyc / reps / 256, wall / reps / 256 / 100.0,
                   cyc / wall * 100.0, cyc / reps / 256 / n);
        }
    }
    return 0;
}
