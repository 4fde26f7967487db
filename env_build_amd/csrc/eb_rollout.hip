// eb_rollout.hip — K5, the fused rollout step (EnvironmentModel.rollout_out, DAM:118-126) for gfx950.
//
// One launch per step.  A block owns a tile of E whole envs (E <= 64, E * n_veh <= RW * 64 * RPT) and
// runs 1 + RW waves with two roles:
//
//   env wave (wave 0, one lane per env) — the per-env chain, ~450 VALU ops on 36 + 36 bytes:
//     load the 9-word head (ego 6 + tracking 3), the action and the path id; sin/cos of the ego heading;
//     put (x, y, sin, cos) of the CURRENT pose into LDS for the record waves            -- barrier 1 --
//     reward terms (DAM:198-207, 297-298), bicycle-model step (DAM:386-392), closest point of the NEXT
//     pose through the cell grid, tracking error (DAM:334-353, 735-770), head store      -- barrier 2 --
//     per-env penalty sums in vehicle order (DAM:218), road walls (DAM:231-295), the four penalty outputs.
//
//   record waves (waves 1..RW, one lane per (env, vehicle) record, RPT records per lane; every 16-byte
//   record load is issued before anything else, consecutive lanes on consecutive records -> coalesced
//   HBM streams):                                                                         -- barrier 1 --
//     per record: centre distance to the ego from LDS; records inside 6.364 m are pushed on the wave's
//     own LDS queue with their (x, y, sin, cos) (ballot prefix, no atomics) — every other record adds
//     exact zeros to the penalty sums (DAM:228-229); predict (DAM:405-427); store.
//     Then the queue, compacted one record per lane: four circle-pair distances (DAM:218-229) ->
//     per-record partial sums + a bit in the env's 64-bit slot mask in LDS               -- barrier 2 --
//
// The record waves never wait for the env wave's chain; the barriers only order LDS traffic
// (s_waitcnt lgkmcnt(0) + s_barrier: outstanding HBM loads/stores stay in flight across them).
// HBM-bound: 104 + 32 n_veh algorithmic bytes per env-step; no MFMA (nothing here is a dense contraction).
#include "eb_device.h"
#include "eb_kernels.h"

#pragma clang fp contract(off)

namespace eb {

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access, 4-byte aligned
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

// LDS-only workgroup barrier: orders this wave's LDS traffic, leaves global loads/stores in flight
EB_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int QCAP = 128;   // per-wave near-record queue (flushed whenever 64 entries are waiting)

// ---- closest point of (px, py) on path p (DAM:702-715), tables in global memory (L1/L2 resident) ----
// The cell of the position names the index range [lo, hi] that provably holds the reference's argmin for
// every position inside the cell (eb_capi.hip:build_cell_grid); scanning it in index order with the
// reference's fp32 expression and a strict '<' returns the index of the full scan after ~6-10 evaluations
// instead of ~370.  Positions outside the grid (or NaN) take the pruned full search.
EB_DEV int closest_cell_index(const FusedArgs& A, int p, int roff, float px, float py) {
    const float* xy = A.xy10 + 2 * roff;
    const float fx = (px - A.gx0) * CELL_INV, fy = (py - A.gy0) * CELL_INV;
    if (!(fx >= 0.0f && fx < (float)A.gnx && fy >= 0.0f && fy < (float)A.gny)) {
        const int n = p == 0 ? A.red_len[0] : p == 1 ? A.red_len[1] : A.red_len[2];
        return closest_reduced_index(reinterpret_cast<const float2*>(xy), A.rad_all + 32 * p, n, px, py);
    }
    const unsigned c = A.cells[(p * A.gny + (int)fy) * A.gnx + (int)fx];
    const int lo = (int)(c & 0xffffu), hi = (int)(c >> 16);
    float best = __builtin_inff();
    int bi = 0;
    for (int r = lo; r <= hi; r += 4) {
        const f4u q01 = *reinterpret_cast<const f4u*>(xy + 2 * r), q23 = *reinterpret_cast<const f4u*>(xy + 2 * r + 4);
        const float d0 = sq(px - q01.x) + sq(py - q01.y), d1 = sq(px - q01.z) + sq(py - q01.w);   // DAM:712
        const float d2 = sq(px - q23.x) + sq(py - q23.y), d3 = sq(px - q23.z) + sq(py - q23.w);
        if (d0 < best) { best = d0; bi = r; }                                                       // first minimum, DAM:714
        if (r + 1 <= hi && d1 < best) { best = d1; bi = r + 1; }
        if (r + 2 <= hi && d2 < best) { best = d2; bi = r + 2; }
        if (r + 3 <= hi && d3 < best) { best = d3; bi = r + 3; }
    }
    return bi;
}

template <int RW, int RPT>
struct FusedSmem {
    static constexpr int ITEMS = RW * 64 * RPT;
    float4 ego[64];                       // (x, y, sin phi, cos phi) of the current ego pose
    float4 tc[64];                        // per slot: (turn radius c, 1/c, sign, enabled), DAM:416-421
    unsigned long long mask[64];          // per env: slots with a non-zero penalty sum
    float2 pen[ITEMS];                    // per record: (3.5 m sum, 2.5 m sum), DAM:228-229
    float4 qd[RW][QCAP];                  // per record wave: queued near records (x, y, sin phi, cos phi)
    unsigned short qi[RW][QCAP];          // and their item ids
};

// ---- env wave -------------------------------------------------------------------------------------
template <int TASK, int RW, int RPT>
EB_DEV void env_wave(const FusedArgs& A, FusedSmem<RW, RPT>& S, int e0, int nE) {
    const int lane = threadIdx.x;   // wave 0
    const int D = A.obs_dim, NV = A.n_veh;
    const bool act = lane < nE;
    const int e = act ? lane : 0, ge = e0 + e;
    const float* hin = A.obs_in + (size_t)ge * D;
    float* hout = A.obs_out + (size_t)ge * D;

    // head (ego 6 | first tracking triple), action, path id
    const f4u h0 = *reinterpret_cast<const f4u*>(hin), h1 = *reinterpret_cast<const f4u*>(hin + 4);
    const float h8 = hin[8];
    const f2u araw = *reinterpret_cast<const f2u*>(A.actions + 2 * (size_t)ge);
    int p = A.path_id;
    if (A.training) {
        const int pr = A.ref_idx[ge];
        p = (pr >= 0 && pr < A.n_paths) ? pr : -1;                          // DAM:342, 352
    }
    if (lane < NV) {   // slot turn constants for the record waves (predict_for_a_mode, DAM:416-421)
        const int t = A.dt->turn[lane];
        S.tc[lane] = t == TURN_LEFT ? make_float4(26.875f, 1.0f / 26.875f, 1.0f, 1.0f)
                   : t == TURN_RIGHT ? make_float4(15.625f, 1.0f / 15.625f, -1.0f, 1.0f)
                                     : make_float4(1.0f, 1.0f, 0.0f, 0.0f);
    }
    const float st[6] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y};
    const float phi_rad = deg2rad(st[5]);
    float es, ec;
    sincos_det(phi_rad, es, ec);                                            // DAM:211 and DAM:79-80
    if (A.do_rewards) {
        S.ego[lane] = make_float4(st[3], st[4], es, ec);
        S.mask[lane] = 0ull;
    }
    lds_barrier();                                                          // ---- barrier 1 ----

    float steer, a_x;
    if (A.actions_raw) action_transform(araw.x, araw.y, steer, a_x);        // DAM:120
    else { steer = araw.x; a_x = araw.y; }
    if (act && A.scaled_actions) *reinterpret_cast<f2u*>(A.scaled_actions + 2 * (size_t)ge) = f2u{steer, a_x};
    if (act && A.do_rewards) {
        const float punish_steer = -sq(steer), punish_a_x = -sq(a_x);       // DAM:198-199
        const float punish_yaw_rate = -sq(st[2]);                           // DAM:202
        const float devi_y = -sq(h1.z);                                     // DAM:205
        const float devi_phi = -sq(deg2rad(h1.w));                          // DAM:206
        const float devi_v = -sq(h8);                                       // DAM:207
        A.out5[ge] = 0.05f * devi_v + 0.8f * devi_y + 30.0f * devi_phi + 0.02f * punish_yaw_rate +
                     5.0f * punish_steer + 0.05f * punish_a_x;              // DAM:297-298
    }
    float nx[6];
    f_xu_core(st, steer, a_x, TAU10, phi_rad, es, ec, nx);                  // DAM:387
    nx[0] = __builtin_fminf(__builtin_fmaxf(nx[0], 0.0f), 35.0f);           // DAM:390
    // tracking error of the next pose on the env's path (DAM:334-353)
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    if (p >= 0) {
        const int roff = p == 0 ? A.red_off[0] : p == 1 ? A.red_off[1] : A.red_off[2];
        const int bi = (A.ablate & 1) ? 0 : closest_cell_index(A, p, roff, nx[3], nx[4]);
        const f2u r = *reinterpret_cast<const f2u*>(A.xy10 + 2 * (roff + bi));   // == path[bi * 10]: bi * 10 < len always
        const float rphi = A.phi10[roff + bi];
        t0 = two2one<TASK>(nx[3], nx[4], r.x, r.y);                         // DAM:758
        t1 = deal_with_phi_diff(nx[5] - rphi);                              // DAM:759
        t2 = nx[0] - EXP_V;                                                 // DAM:760
        if (A.n_future > 0 && act) {                                        // DAM:717-724, 763-768
            const PathTables& pt = *A.dt;
            const int len = pt.len[p];
            float* otrk = hout + 9;
            int cur = bi * 10;                                              // DAM:714
            for (int k = 0; k < A.n_future; ++k) {
                cur += 80;
                if (cur >= len - 2) cur = len - 2;
                const int fi = clamp_index(cur, len);
                otrk[3 * k] = pt.x[p][fi] - nx[3];
                otrk[3 * k + 1] = pt.y[p][fi] - nx[4];
                otrk[3 * k + 2] = deal_with_phi_diff(nx[5] - pt.phi[p][fi]);
            }
        }
    } else if (A.n_future > 0 && act) {
        float* otrk = hout + 9;
        for (int c = 0; c < 3 * A.n_future; ++c) otrk[c] = 0.0f;            // DAM:342, 352
    }
    if (act) {
        *reinterpret_cast<f4u*>(hout) = f4u{nx[0], nx[1], nx[2], nx[3]};
        *reinterpret_cast<f4u*>(hout + 4) = f4u{nx[4], nx[5], t0, t1};
        hout[8] = t2;
    }
    if (!A.do_rewards) return;
    lds_barrier();                                                          // ---- barrier 2 ----

    // per env: penalty sums in vehicle order + road walls (DAM:231-295, 299-300)
    if (act) {
        float a35 = 0.0f, a25 = 0.0f;
        unsigned long long m = S.mask[lane];
        while (m) {
            const int jj = __ffsll((long long)m) - 1;
            m &= m - 1;
            const float2 ps = S.pen[lane * NV + jj];
            a35 += ps.x;
            a25 += ps.y;
        }
        float road_t = 0.0f, road_r = 0.0f;
        road_terms<TASK>(st[3] + LWS * ec, st[4] + LWS * es, road_t, road_r);
        road_terms<TASK>(st[3] - LWS * ec, st[4] - LWS * es, road_t, road_r);
        const size_t n = (size_t)A.n_env;
        A.out5[n + ge] = a35 + road_t;       // DAM:299
        A.out5[2 * n + ge] = a25 + road_r;   // DAM:300
        A.out5[3 * n + ge] = a25;
        A.out5[4 * n + ge] = road_r;
    }
}

// ---- record waves -----------------------------------------------------------------------------------
// one queue pass: entries [0, n) of this wave's queue, one per lane: DAM:218-229
template <int RW, int RPT>
EB_DEV void queue_pass(const FusedArgs& A, FusedSmem<RW, RPT>& S, int w, int lane, int n) {
    if (lane < n) {
        const float4 v = S.qd[w][lane];
        const int item = S.qi[w][lane];
        const int e2 = (int)__umulhi((unsigned)item, A.nv_magic), j2 = item - e2 * A.n_veh;
        const float4 eg = S.ego[e2];
        float t35[4], t25[4];
        const float4 pts = make_float4(eg.x + LWS * eg.w, eg.y + LWS * eg.z, eg.x - LWS * eg.w, eg.y - LWS * eg.z);
        veh2veh_terms(pts, v.x, v.y, v.z, v.w, t35, t25);
        const float p35 = ((t35[0] + t35[1]) + t35[2]) + t35[3];
        const float p25 = ((t25[0] + t25[1]) + t25[2]) + t25[3];
        if (p35 != 0.0f) {   // p25 != 0 implies p35 != 0
            S.pen[item] = make_float2(p35, p25);
            atomicOr(&S.mask[e2], 1ull << j2);
        }
    }
}

template <int TASK, int RW, int RPT>
EB_DEV void record_wave(const FusedArgs& A, FusedSmem<RW, RPT>& S, int e0, int nE) {
    constexpr int RL = RW * 64;                     // record lanes per block
    const int rtid = threadIdx.x - 64, w = rtid >> 6, lane = rtid & 63;
    const int NV = A.n_veh, D = A.obs_dim, HD = D - 4 * NV;
    const int items = nE * NV;
    const float* tin = A.obs_in + (size_t)e0 * D;
    float* tout = A.obs_out + (size_t)e0 * D;

    f4u rec[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        // lanes past the tile's last record re-read that record (branch-free loads; never stored)
        const int item = min(k * RL + rtid, items - 1);
        const int e = (int)__umulhi((unsigned)item, A.nv_magic);
        rec[k] = *reinterpret_cast<const f4u*>(tin + 4 * item + (e + 1) * HD);   // == e*D + HD + 4*j
    }
    lds_barrier();                                                          // ---- barrier 1 ----

    int qn = 0;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int item = k * RL + rtid;
        const bool valid = item < items;
        const int e = (int)__umulhi((unsigned)item, A.nv_magic), j = item - e * NV;
        float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
        float sn = 0.0f, cs = 1.0f;
        if (valid) {
            const float4 tc = S.tc[j];
            unsigned tiny = 0u;
            nv = predict_record<false>(rec[k].x, rec[k].y, rec[k].z, rec[k].w, tc, tiny, sn, cs);
            if (__builtin_expect(tiny != 0u, 0)) nv = predict_record<true>(rec[k].x, rec[k].y, rec[k].z, rec[k].w, tc, tiny, sn, cs);
            *reinterpret_cast<f4u*>(tout + 4 * item + (e + 1) * HD) = f4u{nv.x, nv.y, nv.z, nv.w};
        }
        if (A.do_rewards) {
            // A circle pair can only be closer than 3.5 m when the two vehicle centres are within
            // 3.5 + 2*1.4 = 6.3 m; records inside 6.364 m (slack >> fp32 rounding) are queued, every
            // other record contributes exact zeros to the penalty sums (DAM:228-229).
            bool near = false;
            if (valid) {
                const float4 eg = S.ego[e];
                near = sq(rec[k].x - eg.x) + sq(rec[k].y - eg.y) < 40.5f;
            }
            const unsigned long long b = __ballot(near);
            if (b) {
                if (near) {
                    const int pos = qn + __popcll(b & ((1ull << lane) - 1ull));
                    S.qd[w][pos] = make_float4(rec[k].x, rec[k].y, sn, cs);
                    S.qi[w][pos] = (unsigned short)item;
                }
                qn += __popcll(b);
                if (qn >= 64) {   // flush one full pass, move the remainder (< 64 entries) to the front
                    queue_pass<RW, RPT>(A, S, w, lane, 64);
                    const int rem = qn - 64;
                    float4 td = make_float4(0.f, 0.f, 0.f, 0.f);
                    unsigned short ti = 0;
                    if (lane < rem) { td = S.qd[w][64 + lane]; ti = S.qi[w][64 + lane]; }
                    if (lane < rem) { S.qd[w][lane] = td; S.qi[w][lane] = ti; }
                    qn = rem;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // one record at a time: keeps the live set at the loaded records + one record's temporaries
    }
    if (!A.do_rewards) return;
    if (qn > 0) queue_pass<RW, RPT>(A, S, w, lane, qn);
    lds_barrier();                                                          // ---- barrier 2 ----
}

// waves per SIMD the tile shape is sized for (bounds the VGPR budget): 2048-record tiles -> 4 blocks x 5
// waves per CU at the headline size; the smaller tiles fill all 8 wave slots of a SIMD
template <int RW, int RPT>
constexpr int fused_waves_per_simd() { return RW * RPT >= 32 ? 5 : 8; }

template <int TASK, int RW, int RPT>
__global__ __launch_bounds__((RW + 1) * 64, (fused_waves_per_simd<RW, RPT>())) void rollout_fused_kernel(const FusedArgs A) {
    __shared__ FusedSmem<RW, RPT> S;
    const int e0 = blockIdx.x * A.envs_per_tile;
    const int nE = min(A.envs_per_tile, A.n_env - e0);
    if (threadIdx.x < 64) {
        __builtin_amdgcn_s_setprio(2);
        env_wave<TASK, RW, RPT>(A, S, e0, nE);
    } else {
        record_wave<TASK, RW, RPT>(A, S, e0, nE);
    }
}

int fused_tile_records(int variant) { return variant == 0 ? 4 * 64 * 8 : variant == 1 ? 3 * 64 * 6 : 4 * 64 * 4; }

template <int RW, int RPT>
static hipError_t launch_v(int task, const FusedArgs& A, int grid, hipStream_t s) {
    const dim3 g(grid), b((RW + 1) * 64);
    switch (task) {
        case TASK_LEFT: hipLaunchKernelGGL((rollout_fused_kernel<TASK_LEFT, RW, RPT>), g, b, 0, s, A); break;
        case TASK_STRAIGHT: hipLaunchKernelGGL((rollout_fused_kernel<TASK_STRAIGHT, RW, RPT>), g, b, 0, s, A); break;
        default: hipLaunchKernelGGL((rollout_fused_kernel<TASK_RIGHT, RW, RPT>), g, b, 0, s, A); break;
    }
    return hipGetLastError();
}

hipError_t launch_rollout_fused(int task, int variant, const FusedArgs& A, int grid, hipStream_t s) {
    switch (variant) {
        case 0: return launch_v<4, 8>(task, A, grid, s);
        case 1: return launch_v<3, 6>(task, A, grid, s);
        default: return launch_v<4, 4>(task, A, grid, s);
    }
}

}  // namespace eb
