// eb_kernels.hip — HIP kernels for gfx950 (MI355X) behind the C-ABI of include/envbuild.h.
//
// K5 rollout_step_kernel is the hot path (EnvironmentModel.rollout_out, DAM:118-126) fused into one
// launch per step; the remaining kernels back the single-op entry points (f_xu, compute_rewards,
// tracking_error_vector, veh_predict, ss, and the real-env step pieces).
//
// Layout: obs rows are the reference's [ego 6 | tracking 3(n+1) | veh 4N] fp32, row-major; see the
// comment above rollout_step_kernel for how a block walks them.  HBM bound; no MFMA.
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
// K5: fused rollout step — one launch per step, two block roles
// ------------------------------------------------------------------------------------------------
// rollout_out (DAM:118-126) has two kinds of work with no data dependency between them inside a step:
//   * per-env (DAM:120, 198-207, 297-298, 386-392, 334-353): action transform, the six quadratic
//     reward terms, the ego's bicycle-model step, the closest-point search and tracking error of the
//     NEXT pose.  ~1.3k dependent VALU ops per env on 36 + 36 bytes of traffic: latency-bound.
//   * per-(env, vehicle) (DAM:218-229, 231-295, 394-427): predict every vehicle record, the 2x2
//     circle-pair penalties against the CURRENT ego pose, the road-wall penalties.  ~140 ops per
//     16-byte record in, 16-byte record out: HBM-bound.
// The launch carries two block roles: blocks [0, n_env_blocks) run env_role (one lane per env, 256 envs
// per block — one block per CU at the headline size, raised wave priority so that their long dependent
// chains are not starved), the remaining blocks run veh_role on tiles of whole envs.  The env blocks'
// chains execute underneath the vehicle blocks' streaming.
//
// veh_role tile = E whole envs (E * n_veh <= 1024 records, E <= 64), 256 threads, up to 4 records per
// thread (a 16-byte load/store each; consecutive lanes on consecutive records -> coalesced HBM
// streams):
//   1. all record loads are issued up front; lanes < E fetch the tile's ego poses and put
//      (x, y, sin phi, cos phi) into LDS;                                            -- barrier --
//   2. per record: test the centre distance to the ego (records beyond 6.364 m contribute exact zeros
//      to the penalty sums), queue the near ones in a per-wave LDS list (ballot prefix, no atomics),
//      predict, store;                                                               -- barrier --
//   3. the queued records, compacted across the block, one per lane: four circle-pair distances ->
//      per-record partial sums to LDS + a bit in the env's 64-bit mask;             -- barrier --
//   4. lanes < E: add the env's partial sums in vehicle order (DAM:218), road-wall terms, write the
//      four penalty outputs.
// HBM-bound; no MFMA (nothing here is a dense contraction).

// ---- closest point through the cell grid (PathTables::cells) --------------------------------------
// The cell of (px, py) names the index range [lo, hi] that holds the reference's argmin for every
// position in the cell (eb_capi.hip:build_cell_grid); scanning it in index order with the reference's
// fp32 expression and a strict '<' returns the index of the full scan (DAM:702-715) after ~6-10
// evaluations instead of ~370.  Positions outside the grid (or NaN) take closest_reduced_index.
// `red` must be readable up to 3 entries past hi (the staged table is padded).
EB_DEV int closest_cell_index(const PathTables& pt, int p, const float2* red, const float* rad, int n,
                              float px, float py) {
    const float fx = (px - pt.gx0) * CELL_INV, fy = (py - pt.gy0) * CELL_INV;
    const int nx = pt.gnx, ny = pt.gny;
    if (!(fx >= 0.0f && fx < (float)nx && fy >= 0.0f && fy < (float)ny))
        return closest_reduced_index(red, rad, n, px, py);
    const unsigned c = pt.cells[(p * ny + (int)fy) * nx + (int)fx];
    const int lo = (int)(c & 0xffffu), hi = (int)(c >> 16);
    float best = __builtin_inff();
    int bi = 0;   // all-NaN distances keep index 0, as the full scan does
    for (int r = lo; r <= hi; r += 4) {
        const float2 q0 = red[r], q1 = red[r + 1], q2 = red[r + 2], q3 = red[r + 3];
        const float d0 = sq(px - q0.x) + sq(py - q0.y), d1 = sq(px - q1.x) + sq(py - q1.y);   // DAM:712
        const float d2 = sq(px - q2.x) + sq(py - q2.y), d3 = sq(px - q3.x) + sq(py - q3.y);
        if (d0 < best) { best = d0; bi = r; }                                                   // first minimum, DAM:714
        if (r + 1 <= hi && d1 < best) { best = d1; bi = r + 1; }
        if (r + 2 <= hi && d2 < best) { best = d2; bi = r + 2; }
        if (r + 3 <= hi && d3 < best) { best = d3; bi = r + 3; }
    }
    return bi;
}


constexpr int RT = ROLLOUT_THREADS;
constexpr int ENVS_PER_EBLOCK = ROLLOUT_THREADS;   // env role: one thread per env
constexpr int RPT = ROLLOUT_TILE_RECS / RT;        // vehicle role: up to 4 records per thread
constexpr int TILE_E = ROLLOUT_TILE_ENVS;          // vehicle role: up to 64 envs per tile

template <int TASK>
EB_DEV void env_role(const RolloutArgs& A, unsigned char* smem) {
    const int D = A.obs_dim;
    const int T = 3 * (A.n_future + 1);
    float2* s_red = reinterpret_cast<float2*>(smem);                       // red_total_pad entries
    float* s_rad = reinterpret_cast<float*>(s_red + A.red_total_pad + 4);  // 3 x 32 block radii (4 pad entries before)
    float* s_head = s_rad + 96;                                            // 256 x 9 head words
    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * ENVS_PER_EBLOCK;
    const int nE = min(ENVS_PER_EBLOCK, A.n_env - e0);
    const float* tin = A.obs_in + (size_t)e0 * D;
    float* tout = A.obs_out + (size_t)e0 * D;

    // ---- stage: the block's 9-word heads (ego 6 + first tracking triple), cooperatively so that a
    //      wave touches ~8 cache lines per load instead of 64, and the stride-10 path tables ----
    {
        float hv[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int i = tid + u * RT;
            const int e = i / 9, c = i - 9 * e;
            hv[u] = (i < nE * 9) ? tin[e * D + c] : 0.0f;
        }
        const float4* red4 = reinterpret_cast<const float4*>(A.red_all + A.red_base);   // 2 points per load
        float4* s_red4 = reinterpret_cast<float4*>(s_red);
        const int n4 = A.red_total_pad >> 1;
        for (int base = 0; base < n4; base += 3 * RT) {
            float4 rv[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = base + tid + u * RT;
                rv[u] = (i < n4) ? red4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int i = base + tid + u * RT;
                if (i < n4) s_red4[i] = rv[u];
            }
        }
        if (tid < 96) s_rad[tid] = A.rad_all[tid];
        if (tid < 4) s_red[A.red_total_pad + tid] = make_float2(0.0f, 0.0f);   // read (masked) by the 4-wide range scan
#pragma unroll
        for (int u = 0; u < 9; ++u) s_head[tid + u * RT] = hv[u];
    }
    __syncthreads();

    if (tid < nE && !(A.ablate & 4)) {
        const int e = tid, ge = e0 + e;
        float st[6];
#pragma unroll
        for (int c = 0; c < 6; ++c) st[c] = s_head[9 * e + c];   // stride 9 words: conflict-free
        float steer, a_x;
        const float2 araw = reinterpret_cast<const float2*>(A.actions)[ge];
        if (A.actions_raw) action_transform(araw.x, araw.y, steer, a_x);   // DAM:120
        else { steer = araw.x; a_x = araw.y; }
        if (A.scaled_actions) reinterpret_cast<float2*>(A.scaled_actions)[ge] = make_float2(steer, a_x);
        if (A.do_rewards) {
            const float trk0 = s_head[9 * e + 6], trk1 = s_head[9 * e + 7], trk2 = s_head[9 * e + 8];
            const float punish_steer = -sq(steer), punish_a_x = -sq(a_x);   // DAM:198-199
            const float punish_yaw_rate = -sq(st[2]);                       // DAM:202
            const float devi_y = -sq(trk0);                                 // DAM:205
            const float devi_phi = -sq(deg2rad(trk1));                      // DAM:206
            const float devi_v = -sq(trk2);                                 // DAM:207
            A.out5[ge] = 0.05f * devi_v + 0.8f * devi_y + 30.0f * devi_phi + 0.02f * punish_yaw_rate +
                         5.0f * punish_steer + 0.05f * punish_a_x;          // DAM:297-298
        }
        const float phi_rad = deg2rad(st[5]);
        float es, ec;
        sincos_det(phi_rad, es, ec);
        float nx[6];
        f_xu_core(st, steer, a_x, TAU10, phi_rad, es, ec, nx);              // DAM:387
        nx[0] = __builtin_fminf(__builtin_fmaxf(nx[0], 0.0f), 35.0f);       // DAM:390
#pragma unroll
        for (int c = 0; c < 6; ++c) s_head[9 * e + c] = nx[c];
        // tracking error of the next pose on the env's path (DAM:334-353)
        const int p = A.training ? row_path(*A.dt, A.ref_idx, 0, ge) : A.path_id;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        if (p >= 0) {
            const float2* red = s_red + (A.training ? A.dt->red_off[p] : 0);
            const int bi = (A.ablate & 1) ? 0 : closest_cell_index(*A.dt, p, red, s_rad + 32 * p, A.dt->red_len[p], nx[3], nx[4]);
            const int idx = bi * 10;                                        // DAM:714
            const float2 r = red[bi];                                       // == path[idx]: idx < len always
            const float rphi = A.dt->phi[p][idx];
            t0 = two2one<TASK>(nx[3], nx[4], r.x, r.y);                     // DAM:758
            t1 = deal_with_phi_diff(nx[5] - rphi);                          // DAM:759
            t2 = nx[0] - EXP_V;                                             // DAM:760
            if (A.n_future > 0) {                                           // DAM:717-724, 763-768
                const int len = A.dt->len[p];
                float* otrk = tout + e * D + 9;
                int cur = idx;
                for (int k = 0; k < A.n_future; ++k) {
                    cur += 80;
                    if (cur >= len - 2) cur = len - 2;
                    const int fi = clamp_index(cur, len);
                    otrk[3 * k] = A.dt->x[p][fi] - nx[3];
                    otrk[3 * k + 1] = A.dt->y[p][fi] - nx[4];
                    otrk[3 * k + 2] = deal_with_phi_diff(nx[5] - A.dt->phi[p][fi]);
                }
            }
        } else if (A.n_future > 0) {
            float* otrk = tout + e * D + 9;
            for (int c = 0; c < T - 3; ++c) otrk[c] = 0.0f;                 // DAM:342, 352
        }
        s_head[9 * e + 6] = t0; s_head[9 * e + 7] = t1; s_head[9 * e + 8] = t2;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        const int i = tid + u * RT;
        const int e = i / 9, c = i - 9 * e;
        if (i < nE * 9) tout[e * D + c] = s_head[i];
    }
}

// LDS of a vehicle-role block (fixed offsets, all multiples of 16)
struct VehSmem {
    static constexpr int O_EGO = 0;                                  // float4[64]: (x, y, sin phi, cos phi) of the ego
    static constexpr int O_PEN = O_EGO + TILE_E * 16;                // float2[1024]: per record (3.5 m sum, 2.5 m sum)
    static constexpr int O_MASK = O_PEN + ROLLOUT_TILE_RECS * 8;     // u64[64]: per env, slots with a non-zero sum
    static constexpr int O_QID = O_MASK + TILE_E * 8;                // u16[4][256]: per-wave lists of near records
    static constexpr int O_WCNT = O_QID + ROLLOUT_TILE_RECS * 2;     // int[4]: their lengths
    static constexpr int BYTES = O_WCNT + 16;
};

template <int TASK>
EB_DEV void veh_role(const RolloutArgs& A, unsigned char* smem, int tile) {
    float4* s_ego = reinterpret_cast<float4*>(smem + VehSmem::O_EGO);
    float2* s_pen = reinterpret_cast<float2*>(smem + VehSmem::O_PEN);
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(smem + VehSmem::O_MASK);
    unsigned short* s_qid = reinterpret_cast<unsigned short*>(smem + VehSmem::O_QID);
    int* s_wcnt = reinterpret_cast<int*>(smem + VehSmem::O_WCNT);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int NV = A.n_veh, D = A.obs_dim, HD = D - 4 * NV;
    const int E = A.envs_per_tile, R = A.recs_per_thread;
    const int e0 = tile * E;
    const int nE = min(E, A.n_env - e0);
    const int items = nE * NV;
    const float* tin = A.obs_in + (size_t)e0 * D;
    float* tout = A.obs_out + (size_t)e0 * D;

    // ---- 1: issue every load of the tile ----
    f4u rec[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int item = k * RT + tid;
        const int e = (int)__umulhi((unsigned)item, A.nv_magic);
        if (k < R && item < items) rec[k] = *reinterpret_cast<const f4u*>(tin + 4 * item + (e + 1) * HD);   // == e*D + HD + 4*j
    }
    if (A.do_rewards) {
        if (tid < nE) {
            const float* h = tin + tid * D;
            float es, ec;
            sincos_det(deg2rad(h[5]), es, ec);                       // DAM:211
            s_ego[tid] = make_float4(h[3], h[4], es, ec);
            s_mask[tid] = 0ull;
        }
        __syncthreads();
    }

    // ---- 2: per record: near-ego test + queue, prediction, store ----
    int wq = 0;
    unsigned short* my_q = s_qid + wave * (RPT * 64);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int item = k * RT + tid;
        const bool valid = k < R && item < items;
        const int e = (int)__umulhi((unsigned)item, A.nv_magic), j = item - e * NV;
        if (A.do_rewards && k < R) {
            // A circle pair can only be closer than 3.5 m when the two vehicle centres are within
            // 3.5 + 2*1.4 = 6.3 m; records inside 6.364 m (slack >> fp32 rounding) are queued, every
            // other record contributes exact zeros to the penalty sums (DAM:228-229).
            bool near = false;
            if (valid) {
                const float4 eg = s_ego[e];
                near = sq(rec[k].x - eg.x) + sq(rec[k].y - eg.y) < 40.5f;
            }
            const unsigned long long b = __ballot(near);
            if (near) my_q[wq + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)item;
            wq += __popcll(b);
        }
        if (valid) {
            const int off = 4 * item + (e + 1) * HD;
            if (A.ablate & 2) {
                *reinterpret_cast<f4u*>(tout + off) = rec[k];
            } else {
                const int t = A.dt->turn[j];
                const float4 tc = t == TURN_LEFT ? make_float4(26.875f, 1.0f / 26.875f, 1.0f, 1.0f)
                                : t == TURN_RIGHT ? make_float4(15.625f, 1.0f / 15.625f, -1.0f, 1.0f)
                                                  : make_float4(1.0f, 1.0f, 0.0f, 0.0f);
                unsigned tiny = 0u;
                float psn, pcs;
                float4 nv = predict_record<false>(rec[k].x, rec[k].y, rec[k].z, rec[k].w, tc, tiny, psn, pcs);
                if (__builtin_expect(tiny != 0u, 0)) nv = predict_record<true>(rec[k].x, rec[k].y, rec[k].z, rec[k].w, tc, tiny, psn, pcs);
                f4u o;
                o.x = nv.x; o.y = nv.y; o.z = nv.z; o.w = nv.w;
                *reinterpret_cast<f4u*>(tout + off) = o;
            }
        }
    }
    if (!A.do_rewards) return;
    if (lane == 0) s_wcnt[wave] = wq;
    __syncthreads();

    // ---- 3: the queued near-ego records: four circle-pair distances, DAM:218-229 ----
    {
        const int c0 = s_wcnt[0], c1 = c0 + s_wcnt[1], c2 = c1 + s_wcnt[2], n_fl = c2 + s_wcnt[3];
        for (int s = tid; s < n_fl && !(A.ablate & 2); s += RT) {
            const int w = s < c0 ? 0 : s < c1 ? 1 : s < c2 ? 2 : 3;
            const int it2 = s_qid[w * (RPT * 64) + s - (w == 0 ? 0 : w == 1 ? c0 : w == 2 ? c1 : c2)];
            const int e2 = (int)__umulhi((unsigned)it2, A.nv_magic), j2 = it2 - e2 * NV;
            const f4u v = *reinterpret_cast<const f4u*>(tin + 4 * it2 + (e2 + 1) * HD);   // L1/L2 hit
            const float4 eg = s_ego[e2];
            float vs, vc, t35[4], t25[4];
            const float4 pts = make_float4(eg.x + LWS * eg.w, eg.y + LWS * eg.z, eg.x - LWS * eg.w, eg.y - LWS * eg.z);
            sincos_det(deg2rad(v.w), vs, vc);                                                   // DAM:221
            veh2veh_terms(pts, v.x, v.y, vs, vc, t35, t25);
            const float p35 = ((t35[0] + t35[1]) + t35[2]) + t35[3];
            const float p25 = ((t25[0] + t25[1]) + t25[2]) + t25[3];
            if (p35 != 0.0f) {   // p25 != 0 implies p35 != 0
                s_pen[it2] = make_float2(p35, p25);
                atomicOr(&s_mask[e2], 1ull << j2);
            }
        }
    }
    __syncthreads();

    // ---- 4: per env: penalty sums in vehicle order + road walls (DAM:231-295, 299-300) ----
    if (tid < nE) {
        const int ge = e0 + tid;
        float a35 = 0.0f, a25 = 0.0f;
        unsigned long long m = s_mask[tid];
        while (m) {
            const int jj = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float2 ps = s_pen[tid * NV + jj];
            a35 += ps.x;
            a25 += ps.y;
        }
        const float4 eg = s_ego[tid];
        float road_t = 0.0f, road_r = 0.0f;
        road_terms<TASK>(eg.x + LWS * eg.w, eg.y + LWS * eg.z, road_t, road_r);
        road_terms<TASK>(eg.x - LWS * eg.w, eg.y - LWS * eg.z, road_t, road_r);
        const size_t n = (size_t)A.n_env;
        A.out5[n + ge] = a35 + road_t;       // DAM:299
        A.out5[2 * n + ge] = a25 + road_r;   // DAM:300
        A.out5[3 * n + ge] = a25;
        A.out5[4 * n + ge] = road_r;
    }
}

template <int TASK>
__global__ __launch_bounds__(ROLLOUT_THREADS, 8) void rollout_step_kernel(const RolloutArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x < A.n_env_blocks) {
        if (A.ablate & 16) return;
        __builtin_amdgcn_s_setprio(3);
        env_role<TASK>(A, smem);
    } else {
        // XCD-aware tile order (speed only): workgroup b lands on XCD b % 8, each XCD has its own L2, and an
        // obs row's cache lines are touched by the env block of its 256-env group AND by the group's
        // vehicle tiles.  Put both on the same XCD: vehicle block i = x + 8*y takes tile
        // tpg * (x + 8 * (y / tpg)) + y % tpg, i.e. a tile of group G = x (mod 8) — the XCD of env block G.
        int i = (int)blockIdx.x - A.n_env_blocks;
        if (A.ablate & 8) return;
        if (A.xcd_remap) {
            const int tpg = A.tiles_per_group, x = i & 7, y = i >> 3;
            i = tpg * (x + 8 * (y / tpg)) + y % tpg;
        }
        veh_role<TASK>(A, smem, i);
    }
}

size_t rollout_lds_bytes(int red_total_pad) {
    const size_t env = (size_t)(red_total_pad + 4) * 8 + 96 * 4 + (size_t)RT * 9 * 4;
    const size_t veh = VehSmem::BYTES;
    return env > veh ? env : veh;
}

hipError_t launch_rollout(int task, const RolloutArgs& A, int grid, size_t lds, hipStream_t s) {
    switch (task) {
        case TASK_LEFT: hipLaunchKernelGGL(rollout_step_kernel<TASK_LEFT>, dim3(grid), dim3(RT), lds, s, A); break;
        case TASK_STRAIGHT: hipLaunchKernelGGL(rollout_step_kernel<TASK_STRAIGHT>, dim3(grid), dim3(RT), lds, s, A); break;
        default: hipLaunchKernelGGL(rollout_step_kernel<TASK_RIGHT>, dim3(grid), dim3(RT), lds, s, A); break;
    }
    return hipGetLastError();
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

// compute_rewards (DAM:186-320), one thread per env; same per-vehicle association as the fused kernel
template <int TASK>
__global__ void rewards_kernel(int n_env, int D, int n_future, int NV, const float* __restrict__ obs,
                               const float* __restrict__ act, float* __restrict__ out5,
                               float* __restrict__ d16) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env) return;
    const float* o = obs + (size_t)D * i;
    const float* veh = o + 6 + 3 * (n_future + 1);
    const float steer = act[2 * (size_t)i], a_x = act[2 * (size_t)i + 1];
    const float punish_steer = -sq(steer), punish_a_x = -sq(a_x), punish_yaw_rate = -sq(o[2]);
    const float devi_y = -sq(o[6]), devi_phi = -sq(deg2rad(o[7])), devi_v = -sq(o[8]);
    float es, ec;
    sincos_det(deg2rad(o[5]), es, ec);
    const float4 pts = make_float4(o[3] + LWS * ec, o[4] + LWS * es, o[3] - LWS * ec, o[4] - LWS * es);
    float v2v_train = 0.0f, v2v_real = 0.0f;
    for (int j = 0; j < NV; ++j) {
        const float* v = veh + 4 * j;
        float vs, vc, t35[4], t25[4];
        sincos_det(deg2rad(v[3]), vs, vc);
        veh2veh_terms(pts, v[0], v[1], vs, vc, t35, t25);
        v2v_train += ((t35[0] + t35[1]) + t35[2]) + t35[3];
        v2v_real += ((t25[0] + t25[1]) + t25[2]) + t25[3];
    }
    float road_t = 0.0f, road_r = 0.0f;
    road_terms<TASK>(pts.x, pts.y, road_t, road_r);
    road_terms<TASK>(pts.z, pts.w, road_t, road_r);
    const float rewards = 0.05f * devi_v + 0.8f * devi_y + 30.0f * devi_phi + 0.02f * punish_yaw_rate +
                          5.0f * punish_steer + 0.05f * punish_a_x;
    const size_t n = (size_t)n_env;
    out5[i] = rewards;
    out5[n + i] = v2v_train + road_t;
    out5[2 * n + i] = v2v_real + road_r;
    out5[3 * n + i] = v2v_real;
    out5[4 * n + i] = road_r;
    if (d16) {  // DAM:302-318
        d16[i] = punish_steer; d16[n + i] = punish_a_x; d16[2 * n + i] = punish_yaw_rate;
        d16[3 * n + i] = devi_v; d16[4 * n + i] = devi_y; d16[5 * n + i] = devi_phi;
        d16[6 * n + i] = 5.0f * punish_steer; d16[7 * n + i] = 0.05f * punish_a_x;
        d16[8 * n + i] = 0.02f * punish_yaw_rate; d16[9 * n + i] = 0.05f * devi_v;
        d16[10 * n + i] = 0.8f * devi_y; d16[11 * n + i] = 30.0f * devi_phi;
        d16[12 * n + i] = v2v_train; d16[13 * n + i] = road_t; d16[14 * n + i] = v2v_real; d16[15 * n + i] = road_r;
    }
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
                                const int* __restrict__ ref_idx, int path_id, int n_future,
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
    const int idx = closest_index_scan(pt, p, xs[i], ys[i]);
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
                           const float* vs, const int* ref_idx, int path_id, int n_future, float* out,
                           int* out_index, float* out_points, hipStream_t s) {
    const dim3 g((n + 127) / 128), b(128);
    switch (task) {
        case TASK_LEFT: hipLaunchKernelGGL(tracking_kernel<TASK_LEFT>, g, b, 0, s, n, pt, xs, ys, phis, vs, ref_idx, path_id, n_future, out, out_index, out_points); break;
        case TASK_STRAIGHT: hipLaunchKernelGGL(tracking_kernel<TASK_STRAIGHT>, g, b, 0, s, n, pt, xs, ys, phis, vs, ref_idx, path_id, n_future, out, out_index, out_points); break;
        default: hipLaunchKernelGGL(tracking_kernel<TASK_RIGHT>, g, b, 0, s, n, pt, xs, ys, phis, vs, ref_idx, path_id, n_future, out, out_index, out_points); break;
    }
    return hipGetLastError();
}

// veh_predict (DAM:394-427), one thread per (env, vehicle) record
__global__ void veh_predict_kernel(int n_rec, int NV, VehModes modes, const float* __restrict__ veh,
                                   float* __restrict__ out) {
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
// Stage 1: thread i walks env i, i + G*256, ... and, per env, the `horizon` step records of
// out5_steps [H, 5, B] (consecutive lanes read consecutive envs: coalesced), accumulating in float64;
// wave shuffle + LDS tree gives one 6-double partial per block.  Stage 2: one block folds the
// partials in block order.  No atomics, so the result does not depend on scheduling.
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
    Sum6 acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const size_t n = (size_t)n_env;
    for (int i = blockIdx.x * SUM_THREADS + threadIdx.x; i < n_env; i += gridDim.x * SUM_THREADS) {
        bool any = false;
        for (int t = 0; t < horizon; ++t) {
            const float* o5 = out5_steps + (size_t)t * 5 * n;
            const float pr = o5[2 * n + i];
            acc.r += (double)o5[i];
            acc.pt += (double)o5[n + i];
            acc.pr += (double)pr;
            any = any || pr > 0.0f;
        }
        const double dy = (double)__builtin_fabsf(obs_final[(size_t)i * D + 6]);
        acc.cnt += any ? 1.0 : 0.0;
        acc.ady += dy;
        acc.mdy = dy > acc.mdy ? dy : acc.mdy;
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
    Sum6 acc = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < n_part; b += SUM_THREADS) {
        const double* p = partials + 6 * (size_t)b;
        Sum6 v = {p[0], p[1], p[2], p[3], p[4], p[5]};
        acc = sum6_combine(acc, v);
    }
    const Sum6 r = sum6_block_reduce(acc, s_part);
    if (threadIdx.x == 0) {
        out8[0] = (float)r.r; out8[1] = (float)r.pt; out8[2] = (float)r.pr; out8[3] = (float)r.cnt;
        out8[4] = (float)r.ady; out8[5] = (float)r.mdy; out8[6] = (float)n_env; out8[7] = (float)horizon;
    }
}

hipError_t launch_summary(int n_env, int horizon, int D, const float* out5_steps, const float* obs_final,
                          double* partials, int max_parts, float* out8, hipStream_t s) {
    int g = (n_env + SUM_THREADS - 1) / SUM_THREADS;
    g = g < 1 ? 1 : (g > max_parts ? max_parts : g);
    hipLaunchKernelGGL(summary_partial_kernel, dim3(g), dim3(SUM_THREADS), 0, s, n_env, horizon, D, out5_steps,
                       obs_final, partials);
    hipLaunchKernelGGL(summary_final_kernel, dim3(1), dim3(SUM_THREADS), 0, s, g, n_env, horizon, partials, out8);
    return hipGetLastError();
}

}  // namespace eb
