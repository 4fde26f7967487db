// eb_env_step.hip — CrossroadEnd2end.step (E2E:132-144) for a batch of envs as ONE launch, gfx950.
//
// eb_env_step used to be four launches (action scaling + reward + ego step | traffic step | observation + done code |
// pool re-entry): 62.5 MB of algorithmic traffic in 76-81 us at 65 536 envs x 16 candidates, i.e. ~10 % of the HBM peak —
// the candidates crossed HBM three times, the old observation was read by one thread per row (64 cache lines per load
// instruction), and every launch boundary cost its ~2 us.  Here a block owns a tile of 64 envs for the whole step:
//
//   wave 0 (one lane per env)                           waves 1-3 (192 lanes over the tile's records)
//   -------------------------------------------------   -----------------------------------------------------------
//   head loads: old obs columns 0..8, raw action, ego   traffic step: 16 B per candidate record, coalesced;
//   old ego circle centres -> LDS                       predict_for_a_mode (TRF:220-238's role) -> LDS
//   ------------------------------------------------- barrier 0 ----------------------------------------------------
//   action scaling (E2E:133), reward scalars, road      reward's per-vehicle terms (DAM:218-229), one lane per
//   walls, ego step (E2E:135) -> ego / params in        (env, slot) of the OLD observation -> LDS partials
//   place, new pose -> LDS
//   ------------------------------------------------- barrier 1 ----------------------------------------------------
//   penalty sums in vehicle order -> out5 / dict16;     observation slots (E2E:340-464): the distinct slot modes are
//   closest point (cell grid) + tracking error          dealt to the waves (wave 0 joins last); per mode one walk over
//   (E2E:293-297) -> LDS row; then its share of modes   the candidates -> in-range list -> repeated selection
//   ------------------------------------------------- barrier 2 ----------------------------------------------------
//   collision test shared by the four waves (TRF:263-295) -> LDS flags
//   ------------------------------------------------- barrier 3 ----------------------------------------------------
//   done code (E2E:200-221)                             all: observation rows -> HBM (coalesced); candidates -> HBM,
//                                                       with the pool's re-entry rule applied on the way out
//
// Every record crosses HBM once in each direction; nothing but the tile's LDS is shared between waves, so there is no
// cross-block traffic and no XCD consideration beyond "a tile's lines belong to one workgroup".  The arithmetic is the
// same device functions the single-entry kernels run (eb_env_device.h, eb_device.h): bit-identical to the six (seven)
// calls, which tests/test_gpu_parity.py::test_env_step_composite_equals_the_six_calls holds it to.
#include "eb_env_device.h"

#pragma clang fp contract(off)

namespace eb {

typedef float f4a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment (obs rows: D is odd)

EB_DEV int es_obs_stride(int D) { return D | 1; }                       // floats per LDS row, odd: no bank conflicts
EB_DEV int fast_div(int item, unsigned magic) { return magic ? (int)__umulhi((unsigned)item, magic) : item; }   // magic 0: / 1

size_t env_step_lds_bytes(int D, int NV, int m_cand) {
    const int rs4 = m_cand + ((m_cand & 1) ? 2 : 1), os = D | 1;
    size_t b = (size_t)64 * rs4 * 16;            // s_cand
    b += (size_t)64 * os * 4;                    // s_out
    b += (size_t)64 * NV * 8;                    // s_part
    b += (size_t)64 * 16;                        // s_pts
    b += (size_t)64 * 16;                        // s_ego
    b += (size_t)64 * (m_cand + 4);              // s_mode
    b += (size_t)256 * (m_cand + 1);             // s_list
    return (b + 15) & ~(size_t)15;
}

bool env_step_is_fused(int D, int NV, int m_cand, const float* cand) {
    return m_cand >= 1 && m_cand <= 64 && env_step_lds_bytes(D, NV, m_cand) <= 150 * 1024 &&
           (reinterpret_cast<uintptr_t>(cand) & 15) == 0;
}

template <int TASK>
__global__ __launch_bounds__(256) void env_step_kernel(const EnvStepArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint8_t smode[64], sturn[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e0 = blockIdx.x * 64;
    const int n_env = A.n_env, D = A.D, NV = A.NV, m_cand = A.m_cand, n_future = A.n_future;
    const int nE = n_env - e0 < 64 ? n_env - e0 : 64;
    const int i = e0 + lane;
    const bool live = i < n_env;
    const int RS4 = obs_cand_stride4(m_cand), OS = es_obs_stride(D), MS = m_cand + 4, T = 3 * (n_future + 1);
    float4* s_cand = reinterpret_cast<float4*>(smem);                            // [64][RS4] candidates after the traffic step
    float* s_out = reinterpret_cast<float*>(s_cand + (size_t)64 * RS4);          // [64][OS]  next observation rows
    float2* s_part = reinterpret_cast<float2*>(s_out + (size_t)64 * OS);         // [64][NV]  (veh2veh4training, veh2veh4real) per old slot
    float4* s_pts = reinterpret_cast<float4*>(s_part + (size_t)64 * NV);         // [64]      old ego circle centres (DAM:210-214)
    float4* s_ego = s_pts + 64;                                                  // [64]      new ego (x, y, phi, -)
    uint8_t* s_mode = reinterpret_cast<uint8_t*>(s_ego + 64);                    // [64][MS]
    uint8_t* s_list = s_mode + (size_t)64 * MS;                                  // [4][64][m_cand + 1]
    if (tid < 64) { smode[tid] = A.modes.mode[tid]; sturn[tid] = A.tturn.t[tid]; }
    __syncthreads();   // (kernel-argument tables only: nobody waits for memory here)

    // ---- wave 0: the env's head; waves 1-3: the traffic step ---------------------------------------------------
    float o9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st[6] = {0, 0, 0, 0, 0, 0}, raw0 = 0.0f, raw1 = 0.0f;
    float4 pts = make_float4(0, 0, 0, 0);
    if (wave == 0 && live) {
        const float* o = A.obs + (size_t)D * i;
        const f4a4 a = *reinterpret_cast<const f4a4*>(o), b = *reinterpret_cast<const f4a4*>(o + 4);
        o9[0] = a.x; o9[1] = a.y; o9[2] = a.z; o9[3] = a.w; o9[4] = b.x; o9[5] = b.y; o9[6] = b.z; o9[7] = b.w; o9[8] = o[8];
        const float2 r2 = reinterpret_cast<const float2*>(A.raw)[i];
        raw0 = r2.x; raw1 = r2.y;
        const float2* eg = reinterpret_cast<const float2*>(A.ego + 6 * (size_t)i);
        const float2 g0 = eg[0], g1 = eg[1], g2 = eg[2];
        st[0] = g0.x; st[1] = g0.y; st[2] = g1.x; st[3] = g1.y; st[4] = g2.x; st[5] = g2.y;
        float es, ec;
        sincos_det(deg2rad(o9[5]), es, ec);                                        // DAM:211
        pts = make_float4(o9[3] + LWS * ec, o9[4] + LWS * es, o9[3] - LWS * ec, o9[4] - LWS * es);
        s_pts[lane] = pts;
    } else if (wave != 0) {
        // the traffic step (TRF:220-238's role): the model's own prediction step per candidate, staged for the rest
        const int t = tid - 64;
        const float4* src = reinterpret_cast<const float4*>(A.cand) + (size_t)e0 * m_cand;
        const uint8_t* msrc = A.cand_mode + (size_t)e0 * m_cand;
        const int total = nE * m_cand;
        for (int idx = t; idx < total; idx += 192) {
            const int e = fast_div(idx, A.m_magic), c = idx - e * m_cand;
            const float4 v = src[idx];
            const float phi_rad = deg2rad(v.w);
            float sn, cs;
            sincos_det(phi_rad, sn, cs);
            s_cand[e * RS4 + c] = veh_predict_one(v.x, v.y, v.z, phi_rad, sn, cs, sturn[c]);
            s_mode[e * MS + c] = msrc[idx];
        }
    }
    __syncthreads();   // barrier 0: s_pts, s_cand, s_mode

    float steer = 0.0f, a_x = 0.0f, nx[6] = {0, 0, 0, 0, 0, 0}, pr[4] = {0, 0, 0, 0};
    float road_t = 0.0f, road_r = 0.0f;
    if (wave == 0) {
        if (live) {
            action_transform(raw0, raw1, steer, a_x);                              // E2E:133
            if (A.scaled) reinterpret_cast<float2*>(A.scaled)[i] = make_float2(steer, a_x);
            road_terms<TASK>(pts.x, pts.y, road_t, road_r);                        // DAM:231-295
            road_terms<TASK>(pts.z, pts.w, road_t, road_r);
            env_ego_step_row(st, steer, a_x, nx, pr);                              // E2E:135
            float2* eg = reinterpret_cast<float2*>(A.ego + 6 * (size_t)i);
            eg[0] = make_float2(nx[0], nx[1]); eg[1] = make_float2(nx[2], nx[3]); eg[2] = make_float2(nx[4], nx[5]);
            reinterpret_cast<float4*>(A.params)[i] = make_float4(pr[0], pr[1], pr[2], pr[3]);
            s_ego[lane] = make_float4(nx[3], nx[4], nx[5], nx[0]);
        }
    } else {
        const int t = tid - 64;
        // compute_rewards' vehicle loop on the CURRENT observation (DAM:218-229), one lane per (env, slot)
        const int pairs = nE * NV;
        for (int p = t; p < pairs; p += 192) {
            const int e = fast_div(p, A.nv_magic), j = p - e * NV;
            const f4a4 v = *reinterpret_cast<const f4a4*>(A.obs + (size_t)D * (e0 + e) + 6 + T + 4 * j);
            float vs, vc, t35[4], t25[4];
            sincos_det(deg2rad(v.w), vs, vc);
            veh2veh_terms(s_pts[e], v.x, v.y, vs, vc, t35, t25);
            s_part[e * NV + j] = make_float2(((t35[0] + t35[1]) + t35[2]) + t35[3], ((t25[0] + t25[1]) + t25[2]) + t25[3]);
        }
    }
    __syncthreads();   // barrier 1: s_cand, s_mode, s_part, s_ego

    if (live) {
        float* orow = s_out + lane * OS;
        const float4 eg = s_ego[lane];
        const float ex = eg.x, ey = eg.y;
        if (wave == 0) {
            // E2E:134: the reward of the step taken from the CURRENT observation
            float v2v_train = 0.0f, v2v_real = 0.0f;
            for (int j = 0; j < NV; ++j) {
                const float2 q = s_part[lane * NV + j];
                v2v_train += q.x;
                v2v_real += q.y;
            }
            const float punish_steer = -sq(steer), punish_a_x = -sq(a_x), punish_yaw_rate = -sq(o9[2]);
            const float devi_y = -sq(o9[6]), devi_phi = -sq(deg2rad(o9[7])), devi_v = -sq(o9[8]);
            const float rewards = 0.05f * devi_v + 0.8f * devi_y + 30.0f * devi_phi + 0.02f * punish_yaw_rate +
                                  5.0f * punish_steer + 0.05f * punish_a_x;
            const size_t n = (size_t)n_env;
            float* out5 = A.out5;
            out5[i] = rewards;
            out5[n + i] = v2v_train + road_t;
            out5[2 * n + i] = v2v_real + road_r;
            out5[3 * n + i] = v2v_real;
            out5[4 * n + i] = road_r;
            if (float* d16 = A.d16) {   // DAM:302-318
                d16[i] = punish_steer; d16[n + i] = punish_a_x; d16[2 * n + i] = punish_yaw_rate;
                d16[3 * n + i] = devi_v; d16[4 * n + i] = devi_y; d16[5 * n + i] = devi_phi;
                d16[6 * n + i] = 5.0f * punish_steer; d16[7 * n + i] = 0.05f * punish_a_x;
                d16[8 * n + i] = 0.02f * punish_yaw_rate; d16[9 * n + i] = 0.05f * devi_v;
                d16[10 * n + i] = 0.8f * devi_y; d16[11 * n + i] = 30.0f * devi_phi;
                d16[12 * n + i] = v2v_train; d16[13 * n + i] = road_t; d16[14 * n + i] = v2v_real; d16[15 * n + i] = road_r;
            }
            // E2E:329-338 ego vector, E2E:293-297 tracking error on the env's path
#pragma unroll
            for (int c = 0; c < 6; ++c) orow[c] = nx[c];
            const PathTables& pt = A.pt;
            const int p = row_path(pt, A.ref_idx, A.path_id, i);
            if (p < 0) { for (int c = 0; c < T; ++c) orow[6 + c] = 0.0f; }
            else {
                const float2* red = pt.red[p];
                const float fx = (ex - pt.gx0) * CELL_INV, fy = (ey - pt.gy0) * CELL_INV;
                int bi = 0;
                if (fx >= 0.0f && fx < (float)pt.gnx && fy >= 0.0f && fy < (float)pt.gny) {
                    const unsigned cw = pt.cells[(p * pt.gny + (int)fy) * pt.gnx + (int)fx];
                    const int lo = (int)(cw & 0xffffu), hi = (int)(cw >> 16);
                    float best = __builtin_inff();
                    for (int r = lo; r <= hi; r += 4) {        // same order, same strict '<' as the full scan: same index
                        typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
                        const f4a8 q01 = *reinterpret_cast<const f4a8*>(red + r), q23 = *reinterpret_cast<const f4a8*>(red + r + 2);
                        const float d0 = sq(ex - q01.x) + sq(ey - q01.y), d1 = sq(ex - q01.z) + sq(ey - q01.w);
                        const float d2 = sq(ex - q23.x) + sq(ey - q23.y), d3 = sq(ex - q23.z) + sq(ey - q23.w);
                        if (d0 < best) { best = d0; bi = r; }
                        if (r + 1 <= hi && d1 < best) { best = d1; bi = r + 1; }
                        if (r + 2 <= hi && d2 < best) { best = d2; bi = r + 2; }
                        if (r + 3 <= hi && d3 < best) { best = d3; bi = r + 3; }
                    }
                } else {
                    bi = closest_reduced_index(red, pt.rad + 32 * p, pt.red_len[p], ex, ey);
                }
                const int idx = bi * 10, len = pt.len[p];
                const int ci = clamp_index(idx, len);
                orow[6] = two2one<TASK>(ex, ey, pt.x[p][ci], pt.y[p][ci]);
                orow[7] = deal_with_phi_diff(eg.z - pt.phi[p][ci]);
                orow[8] = eg.w - EXP_V;
                int cur = idx;
                for (int k = 0; k < n_future; ++k) {
                    cur += 80;
                    if (cur >= len - 2) cur = len - 2;
                    const int fi = clamp_index(cur, len);
                    orow[9 + 3 * k] = pt.x[p][fi] - ex;
                    orow[10 + 3 * k] = pt.y[p][fi] - ey;
                    orow[11 + 3 * k] = deal_with_phi_diff(eg.z - pt.phi[p][fi]);
                }
            }
        }
        // E2E:340-464: the distinct modes of the slot list are dealt to the waves, wave 0 (which has the chain above) last
        const float4* crow = s_cand + lane * RS4;
        const uint8_t* mrow = s_mode + lane * MS;
        const bool light = (A.v_light && A.v_light[i] != 0) || (A.virtual_flag && A.virtual_flag[i] != 0);   // E2E:387-388
        const bool virt = TASK != TASK_RIGHT && light && ey < -HALF_CROSS;                                   // E2E:386-388
        float* ov = orow + 6 + T;
        uint8_t* list = s_list + (wave * 64 + lane) * (m_cand + 1);
        int distinct = 0;
        for (int s = 0; s < NV; ++s) {
            const int m = smode[s];
            bool first = true;
            for (int t2 = 0; t2 < s; ++t2) first = first && smode[t2] != m;
            if (!first) continue;
            if (((++distinct) & 3) != wave) continue;
            int L = 0;
            for (int c = 0; c <= m_cand; ++c) {
                V4 v;
                if (!fetch_candidate_lds(m, c, m_cand, crow, mrow, virt, v)) continue;
                if (!veh_in_range(TASK, m, v, ex, ey)) continue;
                list[L++] = (uint8_t)c;
            }
            V4 prev = {0, 0, 0, 0};
            int prev_i = -1;
            bool found = true;
            for (int s2 = s; s2 < NV; ++s2) {
                if (smode[s2] != m) continue;
                if (found) {
                    V4 best = {0, 0, 0, 0};
                    int best_i = -1;
                    for (int q = 0; q < L; ++q) {
                        const int c = list[q];
                        V4 v;
                        fetch_candidate_lds(m, c, m_cand, crow, mrow, virt, v);
                        if (prev_i >= 0) {
                            const int cp = veh_cmp(TASK, m, prev, v);
                            if (!(cp < 0 || (cp == 0 && prev_i < c))) continue;   // not after the previous pick
                        }
                        if (best_i < 0 || veh_cmp(TASK, m, v, best) < 0) { best = v; best_i = c; }
                    }
                    if (best_i < 0) found = false;
                    else { prev = best; prev_i = best_i; }
                }
                const V4 r = found ? prev : veh_fill_value(m);                 // slice_or_fill, E2E:431-437
                ov[4 * s2] = r.x; ov[4 * s2 + 1] = r.y; ov[4 * s2 + 2] = r.v; ov[4 * s2 + 3] = r.phi;
            }
        }
    }
    __syncthreads();   // barrier 2: s_out complete, index lists dead

    {   // E2E:141: the collision test shared by the four waves (candidates w, w + 4, ...)
        bool col = false;
        if (live) {
            const float4 eg = s_ego[lane];
            const EgoCircles E = ego_circles(eg.x, eg.y, eg.z);
            const float4* crow = s_cand + lane * RS4;
            const uint8_t* mrow = s_mode + lane * MS;
            for (int c = wave; c < m_cand; c += 4) {
                if (mrow[c] == EB_VMODE_EMPTY) continue;
                const size_t ck = (size_t)i * m_cand + c;
                col = col || collision_with(E, eg.x, eg.y, crow[c], A.cand_lw ? A.cand_lw[ck * 2] : 4.8f,
                                            A.cand_lw ? A.cand_lw[ck * 2 + 1] : 2.0f);
            }
        }
        s_list[wave * 64 + lane] = col ? 1 : 0;
    }
    __syncthreads();   // barrier 3
    if (wave == 0 && live) {
        const bool collision = (s_list[lane] | s_list[64 + lane] | s_list[128 + lane] | s_list[192 + lane]) != 0;
        A.done_code[i] = judge_code(TASK, collision, nx[0], nx[2], nx[3], nx[4], nx[5], pr[3], s_out[lane * OS + 6],
                                    A.v_light && A.v_light[i] != 0);
    }
    {   // observation rows out: the tile's rows are contiguous in memory
        float* dst = A.obs_out + (size_t)e0 * D;
        const int total = nE * D;
        for (int idx = tid; idx < total; idx += 256) {
            const int e = fast_div(idx, A.d_magic), c = idx - e * D;
            dst[idx] = s_out[e * OS + c];
        }
    }
    {   // candidates out, the pool's re-entry rule on the way (eb_traffic_respawn: after the observation saw this step's state)
        float4* dst = reinterpret_cast<float4*>(A.cand) + (size_t)e0 * m_cand;
        const int total = nE * m_cand;
        for (int idx = tid; idx < total; idx += 256) {
            const int e = fast_div(idx, A.m_magic), c = idx - e * m_cand;
            float4 v = s_cand[e * RS4 + c];
            if (A.respawn_entry && (__builtin_fabsf(v.x) > A.limit || __builtin_fabsf(v.y) > A.limit)) {
                const uint64_t base = (A.counter << 32) + (uint64_t)(e0 + e) * 128u + (uint64_t)c * 2u;
                const float u1 = u01(A.seed, base), u2 = u01(A.seed, base + 1);
                const float* en = A.respawn_entry + 5 * c;
                const float along = u1 * A.span;
                v = make_float4(en[0] + along * en[3], en[1] + along * en[4], u2 * A.v_max, en[2]);
            }
            dst[idx] = v;
        }
    }
}

hipError_t launch_env_step(int task, const EnvStepArgs& A, hipStream_t s) {
    const size_t lds = env_step_lds_bytes(A.D, A.NV, A.m_cand);
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
    hipError_t e = hipSuccess;
    const dim3 g((A.n_env + 63) / 64), b(256);
#define EB_ENV_STEP(T)                                                                                               \
    do {                                                                                                             \
        static size_t granted[64];   /* the > 48 KB opt-in is per kernel and device, and sticky */                   \
        if (lds > 48 * 1024 && lds > granted[dev]) {                                                                 \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_step_kernel<T>),                              \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
            if (e == hipSuccess) granted[dev] = lds;                                                                 \
        }                                                                                                            \
        if (e == hipSuccess) hipLaunchKernelGGL((env_step_kernel<T>), g, b, lds, s, A);                              \
    } while (0)
    switch (task) {
        case TASK_LEFT: EB_ENV_STEP(TASK_LEFT); break;
        case TASK_STRAIGHT: EB_ENV_STEP(TASK_STRAIGHT); break;
        default: EB_ENV_STEP(TASK_RIGHT); break;
    }
#undef EB_ENV_STEP
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace eb
