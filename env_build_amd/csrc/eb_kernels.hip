// eb_kernels.hip — HIP kernels for gfx950 (MI355X) behind the C-ABI of include/envbuild.h.
//
// The hot path (EnvironmentModel.rollout_out, DAM:118-126, one fused launch per step) lives in
// eb_rollout.hip; the kernels here back the single-op entry points (f_xu, compute_rewards,
// tracking_error_vector, veh_predict, ss) and the episodic summary.
//
// Layout: obs rows are the reference's [ego 6 | tracking 3(n+1) | veh 4N] fp32, row-major.  HBM bound; no MFMA.
#include "eb_device.h"
#include "eb_kernels.h"

#pragma clang fp contract(off)

namespace eb {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access, 4-byte aligned

// ------------------------------------------------------------------------------------------------
// shared per-env pieces
// ------------------------------------------------------------------------------------------------
// closest point on path p by a full scan of the stride-10 table in global/L2 (single-op kernels)
EB_DEV int closest_index_scan(const PathTables& pt, int p, float x, float y) {
    const float2* red = pt.red[p];
    const int n = pt.red_len[p];
    float best = __builtin_inff();
    int bi = 0;
    for (int r = 0; r < n; ++r) {
        const float2 q = red[r];
        const float d = sq(x - q.x) + sq(y - q.y);  // DAM:712
        if (d < best) { best = d; bi = r; }          // first minimum, DAM:714
    }
    return bi * 10;
}
// the same over every `ratio`-th point of the full-resolution path (find_closest_point(ratio != 10), DAM:702-715)
EB_DEV int closest_index_scan_ratio(const PathTables& pt, int p, float x, float y, int ratio) {
    if (ratio == 10) return closest_index_scan(pt, p, x, y);
    const float* px = pt.x[p];
    const float* py = pt.y[p];
    const int len = pt.len[p];
    float best = __builtin_inff();
    int bi = 0;
    for (int i = 0; i < len; i += ratio) {
        const float d = sq(x - px[i]) + sq(y - py[i]);
        if (d < best) { best = d; bi = i; }
    }
    return bi;
}

template <int TASK>
EB_DEV void tracking_from_index(const PathTables& pt, int p, int idx, float ex, float ey, float ephi,
                                float ev, int n_future, float* out, int out_stride) {
    const int len = pt.len[p];
    const int ci = clamp_index(idx, len);
    const float rx = pt.x[p][ci], ry = pt.y[p][ci], rphi = pt.phi[p][ci];
    out[0] = two2one<TASK>(ex, ey, rx, ry);         // DAM:758
    out[out_stride] = deal_with_phi_diff(ephi - rphi);  // DAM:759
    out[2 * out_stride] = ev - EXP_V;               // DAM:760
    int cur = idx;
    for (int k = 0; k < n_future; ++k) {            // future_n_data, DAM:717-724
        cur += 80;
        if (cur >= len - 2) cur = len - 2;
        const int fi = clamp_index(cur, len);
        out[(3 + 3 * k) * out_stride] = pt.x[p][fi] - ex;                              // DAM:764
        out[(4 + 3 * k) * out_stride] = pt.y[p][fi] - ey;                              // DAM:765
        out[(5 + 3 * k) * out_stride] = deal_with_phi_diff(ephi - pt.phi[p][fi]);      // DAM:766
    }
}

// ------------------------------------------------------------------------------------------------
// single-op kernels (one thread per row unless noted)
// ------------------------------------------------------------------------------------------------
__global__ void f_xu_kernel(int n, const float* __restrict__ states, const float* __restrict__ actions,
                            float tau, float* __restrict__ next, float* __restrict__ params) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float st[6], nx[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) st[c] = states[6 * (size_t)i + c];
    const float steer = actions[2 * (size_t)i], a_x = actions[2 * (size_t)i + 1];
    const float phi_rad = deg2rad(st[5]);
    float sn, cs;
    sincos_det(phi_rad, sn, cs);
    f_xu_core(st, steer, a_x, tau, phi_rad, sn, cs, nx);
#pragma unroll
    for (int c = 0; c < 6; ++c) next[6 * (size_t)i + c] = nx[c];
    if (params) {
        float pr[4];
        f_xu_params(st, steer, a_x, pr);
#pragma unroll
        for (int c = 0; c < 4; ++c) params[4 * (size_t)i + c] = pr[c];
    }
}

hipError_t launch_f_xu(int n, const float* st, const float* ac, float tau, float* nx, float* pr, hipStream_t s) {
    hipLaunchKernelGGL(f_xu_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, st, ac, tau, nx, pr);
    return hipGetLastError();
}

__global__ void action_transform_kernel(int n, const float* __restrict__ in, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, a;
    action_transform(in[2 * (size_t)i], in[2 * (size_t)i + 1], s, a);
    out[2 * (size_t)i] = s;
    out[2 * (size_t)i + 1] = a;
}

hipError_t launch_action_transform(int n, const float* in, float* out, hipStream_t s) {
    hipLaunchKernelGGL(action_transform_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, in, out);
    return hipGetLastError();
}

// compute_rewards (DAM:186-320), one thread per env (eb_device.h:rewards_env)
template <int TASK>
__global__ void rewards_kernel(int n_env, int D, int n_future, int NV, const float* __restrict__ obs,
                               const float* __restrict__ act, float* __restrict__ out5,
                               float* __restrict__ d16) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env) return;
    rewards_env<TASK>(i, n_env, D, n_future, NV, obs, act[2 * (size_t)i], act[2 * (size_t)i + 1], out5, d16);
}

hipError_t launch_rewards(int task, int n_env, int D, int n_future, int NV, const float* obs, const float* act,
                          float* out5, float* d16, hipStream_t s) {
    const dim3 g((n_env + 127) / 128), b(128);
    switch (task) {
        case TASK_LEFT: hipLaunchKernelGGL(rewards_kernel<TASK_LEFT>, g, b, 0, s, n_env, D, n_future, NV, obs, act, out5, d16); break;
        case TASK_STRAIGHT: hipLaunchKernelGGL(rewards_kernel<TASK_STRAIGHT>, g, b, 0, s, n_env, D, n_future, NV, obs, act, out5, d16); break;
        default: hipLaunchKernelGGL(rewards_kernel<TASK_RIGHT>, g, b, 0, s, n_env, D, n_future, NV, obs, act, out5, d16); break;
    }
    return hipGetLastError();
}

// find_closest_point / tracking_error_vector (DAM:702-770), one thread per row
template <int TASK>
__global__ void tracking_kernel(int n, PathTables pt, const float* __restrict__ xs, const float* __restrict__ ys,
                                const float* __restrict__ phis, const float* __restrict__ vs,
                                const int* __restrict__ ref_idx, int path_id, int n_future, int ratio,
                                float* __restrict__ out, int* __restrict__ out_index, float* __restrict__ out_points) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = row_path(pt, ref_idx, path_id, i);
    const int T = 3 * (n_future + 1);
    if (p < 0) {
        if (out) for (int c = 0; c < T; ++c) out[(size_t)T * i + c] = 0.0f;
        if (out_index) out_index[i] = 0;
        if (out_points) { out_points[i] = 0.0f; out_points[(size_t)n + i] = 0.0f; out_points[2 * (size_t)n + i] = 0.0f; }
        return;
    }
    const int idx = closest_index_scan_ratio(pt, p, xs[i], ys[i], ratio);
    if (out_index) out_index[i] = idx;
    if (out_points) {
        const int ci = clamp_index(idx, pt.len[p]);
        out_points[i] = pt.x[p][ci];
        out_points[(size_t)n + i] = pt.y[p][ci];
        out_points[2 * (size_t)n + i] = pt.phi[p][ci];
    }
    if (out) tracking_from_index<TASK>(pt, p, idx, xs[i], ys[i], phis[i], vs[i], n_future, out + (size_t)T * i, 1);
}

hipError_t launch_tracking(int task, int n, const PathTables& pt, const float* xs, const float* ys, const float* phis,
                           const float* vs, const int* ref_idx, int path_id, int n_future, int ratio, float* out,
                           int* out_index, float* out_points, hipStream_t s) {
    const dim3 g((n + 127) / 128), b(128);
    switch (task) {
        case TASK_LEFT: hipLaunchKernelGGL(tracking_kernel<TASK_LEFT>, g, b, 0, s, n, pt, xs, ys, phis, vs, ref_idx, path_id, n_future, ratio, out, out_index, out_points); break;
        case TASK_STRAIGHT: hipLaunchKernelGGL(tracking_kernel<TASK_STRAIGHT>, g, b, 0, s, n, pt, xs, ys, phis, vs, ref_idx, path_id, n_future, ratio, out, out_index, out_points); break;
        default: hipLaunchKernelGGL(tracking_kernel<TASK_RIGHT>, g, b, 0, s, n, pt, xs, ys, phis, vs, ref_idx, path_id, n_future, ratio, out, out_index, out_points); break;
    }
    return hipGetLastError();
}

// indexs2points + future_n_data (DAM:717-733), one thread per row
__global__ void path_points_kernel(int n, PathTables pt, const int* __restrict__ index, const int* __restrict__ ref_idx,
                                   int path_id, int n_future, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = row_path(pt, ref_idx, path_id, i);
    int cur = index[i];
    for (int k = 0; k <= n_future; ++k) {
        float* o = out + (size_t)k * 3 * n;
        if (p < 0) { o[i] = 0.0f; o[(size_t)n + i] = 0.0f; o[2 * (size_t)n + i] = 0.0f; continue; }
        const int len = pt.len[p];
        if (k > 0) {                                   // DAM:719-722
            cur += 80;
            if (cur >= len - 2) cur = len - 2;
        }
        const int ci = clamp_index(cur, len);          // DAM:727-728
        o[i] = pt.x[p][ci]; o[(size_t)n + i] = pt.y[p][ci]; o[2 * (size_t)n + i] = pt.phi[p][ci];
    }
}
hipError_t launch_path_points(int n, const PathTables& pt, const int* index, const int* ref_idx, int path_id, int n_future,
                              float* out, hipStream_t s) {
    hipLaunchKernelGGL(path_points_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, pt, index, ref_idx, path_id, n_future, out);
    return hipGetLastError();
}

__global__ void phi_diff_kernel(int n, const float* __restrict__ in, float* __restrict__ out) {   // DAM:577-580
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = deal_with_phi_diff(in[i]);
}
hipError_t launch_phi_diff(int n, const float* in, float* out, hipStream_t s) {
    hipLaunchKernelGGL(phi_diff_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, in, out);
    return hipGetLastError();
}

// ego_predict (DAM:386-392): f_xu at 10 Hz, v_x clipped to [0, 35]
__global__ void ego_predict_kernel(int n, const float* __restrict__ ego, const float* __restrict__ actions,
                                   float* __restrict__ next) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float st[6], nx[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) st[c] = ego[6 * (size_t)i + c];
    const float phi_rad = deg2rad(st[5]);
    float sn, cs;
    sincos_det(phi_rad, sn, cs);
    f_xu_core(st, actions[2 * (size_t)i], actions[2 * (size_t)i + 1], TAU10, phi_rad, sn, cs, nx);   // DAM:387
    nx[0] = __builtin_fminf(__builtin_fmaxf(nx[0], 0.0f), 35.0f);                                    // DAM:390
#pragma unroll
    for (int c = 0; c < 6; ++c) next[6 * (size_t)i + c] = nx[c];
}
hipError_t launch_ego_predict(int n, const float* ego, const float* actions, float* next, hipStream_t s) {
    hipLaunchKernelGGL(ego_predict_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, ego, actions, next);
    return hipGetLastError();
}

// veh_predict (DAM:394-427), one thread per (env, vehicle) record
// (veh and out may be the same buffer: eb_env_step advances the traffic pool in place)
__global__ void veh_predict_kernel(int n_rec, int NV, VehModes modes, const float* veh, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const float4 v = reinterpret_cast<const float4*>(veh)[i];
    const float phi_rad = deg2rad(v.w);
    float sn, cs;
    sincos_det(phi_rad, sn, cs);
    reinterpret_cast<float4*>(out)[i] = veh_predict_one(v.x, v.y, v.z, phi_rad, sn, cs, modes.turn[i % NV]);
}

hipError_t launch_veh_predict(int n_env, int NV, const VehModes& modes, const float* veh, float* out, hipStream_t s) {
    const int n_rec = n_env * NV;
    hipLaunchKernelGGL(veh_predict_kernel, dim3((n_rec + 255) / 256), dim3(256), 0, s, n_rec, NV, modes, veh, out);
    return hipGetLastError();
}

// ss (DAM:134-184), one thread per env
template <int TASK>
__global__ void ss_kernel(int n_env, int D, int n_future, int NV, PathTables pt, VehModes modes,
                          const float* __restrict__ obs, const float* __restrict__ actions,
                          const int* __restrict__ ref_idx, int path_id, int training, float one_m_lam,
                          float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env) return;
    (void)pt; (void)ref_idx; (void)path_id; (void)training;  // the next tracking columns do not enter ss
    const float* o = obs + (size_t)D * i;
    const float* veh = o + 6 + 3 * (n_future + 1);
    float steer, a_x;
    action_transform(actions[2 * (size_t)i], actions[2 * (size_t)i + 1], steer, a_x);   // DAM:135
    float st[6], nx[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) st[c] = o[c];
    const float phi_rad = deg2rad(st[5]);
    float s0, c0, s1, c1;
    sincos_det(phi_rad, s0, c0);
    f_xu_core(st, steer, a_x, TAU10, phi_rad, s0, c0, nx);                               // DAM:136
    nx[0] = __builtin_fminf(__builtin_fmaxf(nx[0], 0.0f), 35.0f);
    sincos_det(deg2rad(nx[5]), s1, c1);
    const float ex[2] = {st[3] + LWS * c0, st[3] - LWS * c0}, ey[2] = {st[4] + LWS * s0, st[4] - LWS * s0};
    const float nex[2] = {nx[3] + LWS * c1, nx[3] - LWS * c1}, ney[2] = {nx[4] + LWS * s1, nx[4] - LWS * s1};
    float acc = 0.0f;
    for (int j = 0; j < NV; ++j) {
        const float* v = veh + 4 * j;
        const float e2v = __builtin_sqrtf(sq(st[3] - v[0]) + sq(st[4] - v[1]));          // DAM:159
        const float vphi_rad = deg2rad(v[3]);
        float vs, vc, ns, nc;
        sincos_det(vphi_rad, vs, vc);
        const float4 nv = veh_predict_one(v[0], v[1], v[2], vphi_rad, vs, vc, modes.turn[j]);
        sincos_det(deg2rad(nv.w), ns, nc);
        const float wx[2] = {v[0] + LWS * vc, v[0] - LWS * vc}, wy[2] = {v[1] + LWS * vs, v[1] - LWS * vs};
        const float nwx[2] = {nv.x + LWS * nc, nv.x - LWS * nc}, nwy[2] = {nv.y + LWS * ns, nv.y - LWS * ns};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float d = __builtin_sqrtf(sq(ex[a] - wx[b]) + sq(ey[a] - wy[b]));        // DAM:176-177
                const float nd = __builtin_sqrtf(sq(nex[a] - nwx[b]) + sq(ney[a] - nwy[b]));   // DAM:178-179
                const float next_g = nd - 2.5f, g = d - 2.5f;                                  // DAM:180-181
                const float t = next_g - one_m_lam * g;                                        // DAM:182
                acc += (t < 0.0f && e2v < 10.0f) ? sq(t) : 0.0f;
            }
    }
    out[i] = acc;
}

hipError_t launch_ss(int task, int n_env, int D, int n_future, int NV, const PathTables& pt, const VehModes& modes,
                     const float* obs, const float* actions, const int* ref_idx, int path_id, int training,
                     float one_m_lam, float* out, hipStream_t s) {
    const dim3 g((n_env + 127) / 128), b(128);
    switch (task) {
        case TASK_LEFT: hipLaunchKernelGGL(ss_kernel<TASK_LEFT>, g, b, 0, s, n_env, D, n_future, NV, pt, modes, obs, actions, ref_idx, path_id, training, one_m_lam, out); break;
        case TASK_STRAIGHT: hipLaunchKernelGGL(ss_kernel<TASK_STRAIGHT>, g, b, 0, s, n_env, D, n_future, NV, pt, modes, obs, actions, ref_idx, path_id, training, one_m_lam, out); break;
        default: hipLaunchKernelGGL(ss_kernel<TASK_RIGHT>, g, b, 0, s, n_env, D, n_future, NV, pt, modes, obs, actions, ref_idx, path_id, training, one_m_lam, out); break;
    }
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// episodic-return summary of a shard (eb_episode_summary): two launches, fixed reduction order
// ------------------------------------------------------------------------------------------------
// Stage 1: a block owns 64 envs at a time; its four waves split the `horizon` step records of
// out5_steps [H, 5, B] (wave w takes steps w, w+4, ...: every load is 64 consecutive floats, seven steps
// in flight per lane), accumulating in float64.  The per-env "punished at any step" flags meet in LDS,
// wave 0 adds the |delta_y| terms, and a shuffle + LDS tree leaves one 6-double partial per block.
// Stage 2: one block folds the partials in block order.  No atomics: the result depends on the grid
// size only, never on scheduling.  (A single launch with a last-block ticket was measured slower: the
// device-scope ticket atomics of ~1000 blocks serialise for longer than the second launch costs.)
constexpr int SUM_THREADS = 256;

struct Sum6 { double r, pt, pr, cnt, ady, mdy; };

EB_DEV Sum6 sum6_combine(const Sum6& a, const Sum6& b) {
    Sum6 o;
    o.r = a.r + b.r; o.pt = a.pt + b.pt; o.pr = a.pr + b.pr; o.cnt = a.cnt + b.cnt; o.ady = a.ady + b.ady;
    o.mdy = a.mdy > b.mdy ? a.mdy : b.mdy;
    return o;
}

EB_DEV Sum6 sum6_block_reduce(Sum6 v, Sum6* s_part) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Sum6 o;
        o.r = __shfl_down(v.r, off, 64); o.pt = __shfl_down(v.pt, off, 64); o.pr = __shfl_down(v.pr, off, 64);
        o.cnt = __shfl_down(v.cnt, off, 64); o.ady = __shfl_down(v.ady, off, 64); o.mdy = __shfl_down(v.mdy, off, 64);
        v = sum6_combine(v, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_part[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < SUM_THREADS / 64; ++w) v = sum6_combine(v, s_part[w]);
    }
    return v;   // valid in thread 0
}

__global__ __launch_bounds__(SUM_THREADS) void summary_partial_kernel(int n_env, int horizon, int D,
                                                                       const float* __restrict__ out5_steps,
                                                                       const float* __restrict__ obs_final,
                                                                       double* __restrict__ partials) {
    __shared__ Sum6 s_part[SUM_THREADS / 64];
    __shared__ unsigned char s_any[SUM_THREADS / 64][64];
    constexpr int NW = SUM_THREADS / 64;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Sum6 acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const size_t n = (size_t)n_env;
    for (int base = blockIdx.x * 64; base < n_env; base += gridDim.x * 64) {
        const int i = base + lane;
        const bool live = i < n_env;
        const int ic = live ? i : n_env - 1;
        float dyv = 0.0f;
        if (wave == 0) dyv = obs_final[(size_t)ic * D + 6];          // strided: issued first, used last
        bool any = false;
        double r = 0.0, pt = 0.0, prs = 0.0;
        constexpr int U = 7;                                          // steps in flight per lane (3 loads each): a 25-step horizon is ONE round of loads per wave
        for (int t0 = wave; t0 < horizon; t0 += U * NW) {
            float a[U], b[U], c[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * NW;
                const float* o5 = out5_steps + (size_t)(t < horizon ? t : horizon - 1) * 5 * n;
                a[u] = o5[ic]; b[u] = o5[n + ic]; c[u] = o5[2 * n + ic];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (t0 + u * NW < horizon) {
                    r += (double)a[u]; pt += (double)b[u]; prs += (double)c[u];
                    any = any || c[u] > 0.0f;
                }
        }
        if (live) { acc.r += r; acc.pt += pt; acc.pr += prs; }
        s_any[wave][lane] = any ? 1 : 0;
        __syncthreads();
        if (wave == 0 && live) {
            bool a = false;
            for (int w = 0; w < NW; ++w) a = a || s_any[w][lane] != 0;
            const double dy = (double)__builtin_fabsf(dyv);
            acc.cnt += a ? 1.0 : 0.0;
            acc.ady += dy;
            acc.mdy = dy > acc.mdy ? dy : acc.mdy;
        }
        __syncthreads();
    }
    const Sum6 b = sum6_block_reduce(acc, s_part);
    if (threadIdx.x == 0) {
        double* p = partials + 6 * (size_t)blockIdx.x;
        p[0] = b.r; p[1] = b.pt; p[2] = b.pr; p[3] = b.cnt; p[4] = b.ady; p[5] = b.mdy;
    }
}

__global__ __launch_bounds__(SUM_THREADS) void summary_final_kernel(int n_part, int n_env, int horizon,
                                                                     const double* __restrict__ partials,
                                                                     float* __restrict__ out8) {
    __shared__ Sum6 s_part[SUM_THREADS / 64];
    Sum6 tot = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int k0 = threadIdx.x; k0 < n_part; k0 += 4 * SUM_THREADS) {     // four records in flight per thread, folded in index order
        Sum6 x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * SUM_THREADS;
            const double* p = partials + 6 * (size_t)(k < n_part ? k : 0);
            x[u] = Sum6{p[0], p[1], p[2], p[3], p[4], p[5]};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (k0 + u * SUM_THREADS < n_part) tot = sum6_combine(tot, x[u]);
    }
    const Sum6 f = sum6_block_reduce(tot, s_part);
    if (threadIdx.x == 0) {
        out8[0] = (float)f.r; out8[1] = (float)f.pt; out8[2] = (float)f.pr; out8[3] = (float)f.cnt;
        out8[4] = (float)f.ady; out8[5] = (float)f.mdy; out8[6] = (float)n_env; out8[7] = (float)horizon;
    }
}

// The fold behind an ACCUMULATING rollout (eb_rollout_step_acc): the rollout launches have left one 32-byte record per block and
// step — the tile's float64 sums of the step and its "punished in this step" bits — and the last one a (sum, max) pair of the
// final rows' |delta_y| per block, so the summary is one pass over horizon x n_blocks x 32 bytes (0.8 MB at the headline size)
// instead of a second pass over out5_steps [H, 5, B] (20 MB).  A thread owns a block's column: its records in step order (four
// steps in flight), the bits OR-ed; then the shuffle + LDS tree over the threads.  Fixed order, no atomics.
constexpr int FOLD_THREADS = 1024;
__global__ __launch_bounds__(FOLD_THREADS) void acc_fold_kernel(int n_blocks, int n_env, int horizon, const double* __restrict__ records,
                                                                 const double* __restrict__ finals, float* __restrict__ out8) {
    __shared__ Sum6 s_part[FOLD_THREADS / 64];
    // (the record as 16-byte integer pairs, the sums re-typed one scalar at a time: __builtin_bit_cast applied to an ELEMENT of a
    // double vector read the vector's first element here — the count came out as the popcount of a sum)
    typedef unsigned long long q2v __attribute__((ext_vector_type(2)));
    auto f64 = [](unsigned long long b) { return __builtin_bit_cast(double, b); };
    Sum6 tot = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < n_blocks; b += FOLD_THREADS) {
        double r = 0.0, pt = 0.0, pr = 0.0;
        unsigned long long any = 0ull;
        constexpr int U = 4;
        for (int t0 = 0; t0 < horizon; t0 += U) {
            q2v x[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u < horizon ? t0 + u : horizon - 1;
                const q2v* p = reinterpret_cast<const q2v*>(records + ((size_t)t * n_blocks + b) * ACC_RECORD_DOUBLES);
                x[u][0] = p[0]; x[u][1] = p[1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (t0 + u < horizon) {
                    const unsigned long long a0 = x[u][0].x, a1 = x[u][0].y, a2 = x[u][1].x, a3 = x[u][1].y;
                    r += f64(a0); pt += f64(a1); pr += f64(a2); any |= a3;
                }
        }
        const q2v fin = reinterpret_cast<const q2v*>(finals)[b];
        const unsigned long long f0 = fin.x, f1 = fin.y;
        tot = sum6_combine(tot, Sum6{r, pt, pr, (double)__popcll(any), f64(f0), f64(f1)});
    }
    // sum6_block_reduce's tree for this block size
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        Sum6 o;
        o.r = __shfl_down(tot.r, off, 64); o.pt = __shfl_down(tot.pt, off, 64); o.pr = __shfl_down(tot.pr, off, 64);
        o.cnt = __shfl_down(tot.cnt, off, 64); o.ady = __shfl_down(tot.ady, off, 64); o.mdy = __shfl_down(tot.mdy, off, 64);
        tot = sum6_combine(tot, o);
    }
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        Sum6 f = tot;
        for (int w = 1; w < FOLD_THREADS / 64; ++w) f = sum6_combine(f, s_part[w]);
        out8[0] = (float)f.r; out8[1] = (float)f.pt; out8[2] = (float)f.pr; out8[3] = (float)f.cnt;
        out8[4] = (float)f.ady; out8[5] = (float)f.mdy; out8[6] = (float)n_env; out8[7] = (float)horizon;
    }
}

hipError_t launch_acc_fold(int n_blocks, int n_env, int horizon, const double* records, const double* finals, float* out8,
                           hipStream_t s) {
    hipLaunchKernelGGL(acc_fold_kernel, dim3(1), dim3(FOLD_THREADS), 0, s, n_blocks, n_env, horizon, records, finals, out8);
    return hipGetLastError();
}

hipError_t launch_summary(int n_env, int horizon, int D, const float* out5_steps, const float* obs_final,
                          double* partials, int max_parts, float* out8, hipStream_t s) {
    int g = (n_env + 63) / 64;
    g = g < 1 ? 1 : (g > max_parts ? max_parts : g);
    hipLaunchKernelGGL(summary_partial_kernel, dim3(g), dim3(SUM_THREADS), 0, s, n_env, horizon, D, out5_steps,
                       obs_final, partials);
    hipLaunchKernelGGL(summary_final_kernel, dim3(1), dim3(SUM_THREADS), 0, s, g, n_env, horizon, partials, out8);
    return hipGetLastError();
}

}  // namespace eb
