// Microbenchmark (profiling aid, not product): cost breakdown of streaming the obs rows with the
// real per-record arithmetic, to find which structure reaches the copy floor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "../../env_build_amd/csrc/eb_device.h"
#pragma clang fp contract(off)
using namespace eb;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int NV = 32, D = 9 + 4 * NV, HD = 9;

template <bool EXACT>
EB_DEV float4 predict_record(float x, float y, float v, float phi, const float4 tc, unsigned& tiny, float& sn, float& cs) {
    const float t1 = phi * PI_F;
    const float phi_rad = EXACT ? t1 / 180.0f : div_fast(t1, 180.0f, 1.0f / 180.0f);
    sincos_det(phi_rad, sn, cs);
    const bool middle = (x > -HALF_CROSS && x < HALF_CROSS) && (y > -HALF_CROSS && y < HALF_CROSS);
    const float v10 = EXACT ? v / 10.0f : div_fast(v, 10.0f, 1.0f / 10.0f);
    const float dx = v10 * cs, dy = v10 * sn;
    const float u = (EXACT ? v / tc.x : div_fast(v, tc.x, tc.y)) * tc.z;
    const float u10 = EXACT ? u / 10.0f : div_fast(u, 10.0f, 1.0f / 10.0f);
    const float dphi = (middle && tc.w != 0.0f) ? u10 : 0.0f;
    float nphi = phi_rad + dphi;
    if (nphi > PI_F) nphi = nphi - TWO_PI_F;
    if (nphi <= -PI_F) nphi = nphi + TWO_PI_F;
    const float t2 = nphi * 180.0f;
    const float nphi_deg = EXACT ? t2 / PI_F : div_fast(t2, PI_F, 1.0f / PI_F);
    if (!EXACT) {
        const unsigned a = (__builtin_bit_cast(unsigned, t1) << 1) - 1u;
        const unsigned b = (__builtin_bit_cast(unsigned, v) << 1) - 1u;
        const unsigned c = (__builtin_bit_cast(unsigned, t2) << 1) - 1u;
        const unsigned d = (__builtin_bit_cast(unsigned, u) << 1) - 1u;
        tiny = min(min(a, b), min(c, d)) < 2u * 0x0D000000u - 1u;
    }
    return make_float4(x + dx, y + dy, v, nphi_deg);
}

// MODE 0: copy record; 1: + predict; 2: + ego load + near test + inline near terms + LDS reduce (1 barrier)
// 3: like 2 but near records go through a block queue (3 barriers, old design)
template <int MODE>
__global__ __launch_bounds__(256, 8) void rec_kernel(int n_env, const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out5) {
    __shared__ float2 s_pen[256];
    __shared__ unsigned long long s_mask[8];
    __shared__ int s_cnt;
    __shared__ unsigned short s_q[256];
    const int tid = threadIdx.x;
    const int E = 256 / NV;
    const int e0 = blockIdx.x * E;
    const int e = tid / NV, j = tid - e * NV;
    const int ge = e0 + e;
    if (MODE >= 2) { if (tid < E) s_mask[tid] = 0ull; if (tid == 0) s_cnt = 0; }
    const size_t off = (size_t)ge * D + HD + 4 * j;
    f4u r = *reinterpret_cast<const f4u*>(in + off);
    float egx = 0, egy = 0, egphi = 0;
    if (MODE >= 2) { egx = in[(size_t)ge * D + 3]; egy = in[(size_t)ge * D + 4]; egphi = in[(size_t)ge * D + 5]; __syncthreads(); }
    f4u o = r;
    float sn = 0, cs = 0;
    if (MODE >= 1) {
        const int t = j & 3;
        const float4 tc = t == 1 ? make_float4(26.875f, 1.0f / 26.875f, 1.0f, 1.0f) : t == 2 ? make_float4(15.625f, 1.0f / 15.625f, -1.0f, 1.0f) : make_float4(1.0f, 1.0f, 0.0f, 0.0f);
        unsigned tiny = 0;
        float4 nv = predict_record<false>(r.x, r.y, r.z, r.w, tc, tiny, sn, cs);
        if (__builtin_expect(tiny != 0u, 0)) nv = predict_record<true>(r.x, r.y, r.z, r.w, tc, tiny, sn, cs);
        o.x = nv.x; o.y = nv.y; o.z = nv.z; o.w = nv.w;
    }
    *reinterpret_cast<f4u*>(out + off) = o;
    if (MODE == 2) {
        const float c2 = sq(r.x - egx) + sq(r.y - egy);
        if (c2 < 40.5f) {
            float es, ec, t35[4], t25[4];
            sincos_det(deg2rad(egphi), es, ec);
            const float4 pts = make_float4(egx + LWS * ec, egy + LWS * es, egx - LWS * ec, egy - LWS * es);
            veh2veh_terms(pts, r.x, r.y, sn, cs, t35, t25);
            const float p35 = ((t35[0] + t35[1]) + t35[2]) + t35[3], p25 = ((t25[0] + t25[1]) + t25[2]) + t25[3];
            if (p35 != 0.0f) { s_pen[tid] = make_float2(p35, p25); atomicOr(&s_mask[e], 1ull << j); }
        }
    }
    if (MODE == 3) {
        const float c2 = sq(r.x - egx) + sq(r.y - egy);
        if (c2 < 40.5f) s_q[atomicAdd(&s_cnt, 1)] = (unsigned short)tid;
        __syncthreads();
        const int n = s_cnt;
        if (tid < n) {
            const int it = s_q[tid];
            const int e2 = it / NV, j2 = it - e2 * NV;
            const float* row = in + (size_t)(e0 + e2) * D;
            const f4u v = *reinterpret_cast<const f4u*>(row + HD + 4 * j2);
            float es, ec, vs, vc, t35[4], t25[4];
            sincos_det(deg2rad(row[5]), es, ec);
            const float4 pts = make_float4(row[3] + LWS * ec, row[4] + LWS * es, row[3] - LWS * ec, row[4] - LWS * es);
            sincos_det(deg2rad(v.w), vs, vc);
            veh2veh_terms(pts, v.x, v.y, vs, vc, t35, t25);
            const float p35 = ((t35[0] + t35[1]) + t35[2]) + t35[3], p25 = ((t25[0] + t25[1]) + t25[2]) + t25[3];
            if (p35 != 0.0f) { s_pen[it] = make_float2(p35, p25); atomicOr(&s_mask[e2], 1ull << j2); }
        }
    }
    if (MODE >= 2) {
        __syncthreads();
        if (tid < E) {
            float a35 = 0, a25 = 0;
            unsigned long long m = s_mask[tid];
            while (m) { const int jj = __ffsll((long long)m) - 1; m &= m - 1; a35 += s_pen[tid * NV + jj].x; a25 += s_pen[tid * NV + jj].y; }
            const float* row = in + (size_t)(e0 + tid) * D;
            float es, ec;
            sincos_det(deg2rad(row[5]), es, ec);
            float rt = 0, rr = 0;
            road_terms<0>(row[3] + LWS * ec, row[4] + LWS * es, rt, rr);
            road_terms<0>(row[3] - LWS * ec, row[4] - LWS * es, rt, rr);
            const size_t n = n_env;
            out5[n + e0 + tid] = a35 + rt; out5[2 * n + e0 + tid] = a25 + rr; out5[3 * n + e0 + tid] = a25; out5[4 * n + e0 + tid] = rr;
        }
    }
}


// MODE 5: tile of 32 envs, 4 records per thread (loads up front), per-wave near queues (ballot, no
// atomics), 3 barriers; no env role.
template <int R, int VAR>
__global__ __launch_bounds__(256, 8) void tile_kernel(int n_env, const float* __restrict__ in, float* __restrict__ out, float* __restrict__ out5) {
    constexpr int E = R * 256 / NV;
    __shared__ float4 s_ego[E];
    __shared__ float2 s_pen[R * 256];
    __shared__ unsigned long long s_mask[E];
    __shared__ unsigned short s_q[4][R * 64];
    __shared__ f4u s_qr[VAR == 1 ? 4 : 1][VAR == 1 ? R * 64 : 1];
    __shared__ int s_wcnt[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int e0 = blockIdx.x * E;
    const float* tin = in + (size_t)e0 * D;
    float* tout = out + (size_t)e0 * D;
    f4u rec[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int item = k * 256 + tid, e = item / NV;
        rec[k] = *reinterpret_cast<const f4u*>(tin + 4 * item + (e + 1) * HD);
    }
    if (tid < E) { const float* h = tin + tid * D; s_ego[tid] = make_float4(h[3], h[4], h[5], 0.f); s_mask[tid] = 0ull; }
    __syncthreads();
    int wq = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int item = k * 256 + tid, e = item / NV, j = item - e * NV;
        const f4u r = rec[k];
        const float4 eg = s_ego[e];
        const bool near = sq(r.x - eg.x) + sq(r.y - eg.y) < 40.5f;
        const unsigned long long b = __ballot(near);
        if (VAR != 3 && near) { const int sl = wq + __popcll(b & ((1ull << lane) - 1ull)); s_q[wave][sl] = (unsigned short)item; if (VAR == 1) s_qr[wave][sl] = r; }
        wq += __popcll(b);
        const int t = j & 3;
        const float4 tc = t == 1 ? make_float4(26.875f, 1.0f / 26.875f, 1.0f, 1.0f) : t == 2 ? make_float4(15.625f, 1.0f / 15.625f, -1.0f, 1.0f) : make_float4(1.0f, 1.0f, 0.0f, 0.0f);
        unsigned tiny = 0; float sn, cs;
        float4 nv = predict_record<false>(r.x, r.y, r.z, r.w, tc, tiny, sn, cs);
        if (__builtin_expect(tiny != 0u, 0)) nv = predict_record<true>(r.x, r.y, r.z, r.w, tc, tiny, sn, cs);
        f4u o; o.x = nv.x; o.y = nv.y; o.z = nv.z; o.w = nv.w;
        *reinterpret_cast<f4u*>(tout + 4 * item + (e + 1) * HD) = o;
    }
    if (VAR == 3) return;
    if (lane == 0) s_wcnt[wave] = wq;
    __syncthreads();
    const int c0 = s_wcnt[0], c1 = c0 + s_wcnt[1], c2 = c1 + s_wcnt[2], n = c2 + s_wcnt[3];
    for (int s = tid; s < (VAR == 2 ? 0 : n); s += 256) {
        const int w = s < c0 ? 0 : s < c1 ? 1 : s < c2 ? 2 : 3;
        const int sl2 = s - (w == 0 ? 0 : w == 1 ? c0 : w == 2 ? c1 : c2);
        const int it = s_q[w][sl2];
        const int e2 = it / NV, j2 = it - e2 * NV;
        const f4u v = VAR == 1 ? s_qr[w][sl2] : *reinterpret_cast<const f4u*>(tin + 4 * it + (e2 + 1) * HD);
        const float4 eg = s_ego[e2];
        float es, ec, vs, vc, t35[4], t25[4];
        sincos_det(deg2rad(eg.z), es, ec);
        const float4 pts = make_float4(eg.x + LWS * ec, eg.y + LWS * es, eg.x - LWS * ec, eg.y - LWS * es);
        sincos_det(deg2rad(v.w), vs, vc);
        veh2veh_terms(pts, v.x, v.y, vs, vc, t35, t25);
        const float p35 = ((t35[0] + t35[1]) + t35[2]) + t35[3], p25 = ((t25[0] + t25[1]) + t25[2]) + t25[3];
        if (p35 != 0.0f) { s_pen[it] = make_float2(p35, p25); atomicOr(&s_mask[e2], 1ull << j2); }
    }
    __syncthreads();
    if (tid < E) {
        float a35 = 0, a25 = 0;
        unsigned long long m = s_mask[tid];
        while (m) { const int jj = __ffsll((long long)m) - 1; m &= m - 1; a35 += s_pen[tid * NV + jj].x; a25 += s_pen[tid * NV + jj].y; }
        const float4 eg = s_ego[tid];
        float es, ec;
        sincos_det(deg2rad(eg.z), es, ec);
        float rt = 0, rr = 0;
        road_terms<0>(eg.x + LWS * ec, eg.y + LWS * es, rt, rr);
        road_terms<0>(eg.x - LWS * ec, eg.y - LWS * es, rt, rr);
        const size_t nn = n_env;
        out5[nn + e0 + tid] = a35 + rt; out5[2 * nn + e0 + tid] = a25 + rr; out5[3 * nn + e0 + tid] = a25; out5[4 * nn + e0 + tid] = rr;
    }
}

// heads: one thread per env does a stand-in chain of CHAIN dependent ops, reads/writes 9 floats
template <int CHAIN>
__global__ void head_kernel(int n_env, const float* __restrict__ in, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env) return;
    float h[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) h[c] = in[(size_t)i * D + c];
    float a = h[0];
    for (int k = 0; k < CHAIN; ++k) a = a * 1.0001f + h[k % 9];
    h[0] = a;
#pragma unroll
    for (int c = 0; c < 9; ++c) out[(size_t)i * D + c] = h[c];
}

int main() {
    const int B = 65536;
    const size_t n = (size_t)B * D, bytes = n * 4;
    std::vector<float> h(n);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> U(-60, 60), V(0, 8), P(-180, 180), U01(0, 1);
    for (int e = 0; e < B; ++e) {
        float* row = h.data() + (size_t)e * D;
        row[0] = V(g); row[1] = 0; row[2] = 0; row[3] = U(g) * 0.3f; row[4] = U(g) * 0.3f; row[5] = P(g); row[6] = 0.1f; row[7] = 1.f; row[8] = -2.f;
        for (int j = 0; j < NV; ++j) {
            float* v = row + 9 + 4 * j;
            if (U01(g) < 0.25f) { float rad = 12 * sqrtf(U01(g)), a = P(g) * 0.0174f; v[0] = row[3] + rad * cosf(a); v[1] = row[4] + rad * sinf(a); }
            else { v[0] = U(g); v[1] = U(g); }
            v[2] = V(g); v[3] = P(g);
        }
    }
    float *buf[2], *o5;
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&buf[i], bytes)); CK(hipMemcpy(buf[i], h.data(), bytes, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&o5, 5 * B * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 300;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 10; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch(i);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s %8.2f us\n", name, ms * 1e3 / iters);
        // keep inputs bounded: reload
        for (int i = 0; i < 2; ++i) CK(hipMemcpy(buf[i], h.data(), bytes, hipMemcpyHostToDevice));
    };
    const int gb = B / (256 / NV);
#define REC(M) [&](int i) { hipLaunchKernelGGL(rec_kernel<M>, dim3(gb), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1], o5); }
    run("records copy", REC(0));
    run("records + predict", REC(1));
    run("records + predict + inline near + 1 barrier", REC(2));
    run("records + predict + queued near + 3 barriers", REC(3));
    run("tile R=4 (32 envs/block), queued near", [&](int i) { hipLaunchKernelGGL((tile_kernel<4, 0>), dim3(B / 32), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1], o5); });
    run("tile R=2 (16 envs/block), queued near", [&](int i) { hipLaunchKernelGGL((tile_kernel<2, 0>), dim3(B / 16), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1], o5); });
    run("tile R=1 (8 envs/block), queued near", [&](int i) { hipLaunchKernelGGL((tile_kernel<1, 0>), dim3(B / 8), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1], o5); });
    run("tile R=4 records kept in LDS queue", [&](int i) { hipLaunchKernelGGL((tile_kernel<4, 1>), dim3(B / 32), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1], o5); });
    run("tile R=4 skip queue processing", [&](int i) { hipLaunchKernelGGL((tile_kernel<4, 2>), dim3(B / 32), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1], o5); });
    run("tile R=4 no near, no post phases", [&](int i) { hipLaunchKernelGGL((tile_kernel<4, 3>), dim3(B / 32), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1], o5); });
    run("heads only chain=0 (256 thr)", [&](int i) { hipLaunchKernelGGL(head_kernel<0>, dim3(B / 256), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1]); });
    run("heads only chain=1000 (256 thr)", [&](int i) { hipLaunchKernelGGL(head_kernel<1000>, dim3(B / 256), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1]); });
    run("heads only chain=1000 (64 thr)", [&](int i) { hipLaunchKernelGGL(head_kernel<1000>, dim3(B / 64), dim3(64), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1]); });
    hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    run("records(3) then heads chain=1000 same stream", [&](int i) {
        hipLaunchKernelGGL(rec_kernel<3>, dim3(gb), dim3(256), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1], o5);
        hipLaunchKernelGGL(head_kernel<1000>, dim3(B / 64), dim3(64), 0, 0, B, (const float*)buf[i & 1], buf[(i + 1) & 1]); });
    return 0;
}
