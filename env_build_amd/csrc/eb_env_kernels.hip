// eb_env_kernels.hip — batched real-env step pieces of CrossroadEnd2end (endtoend.py), gfx950.
//
// One thread per env: these are control-flow-heavy, tiny-data kernels (filter / select / pad of at
// most a few dozen candidate vehicles, a priority chain of predicates) written for bit-for-bit
// agreement with the oracle.  The observation kernel, the heaviest of them, stages a wave's 64 envs
// through LDS — candidates in, observation rows out, both as coalesced 16-byte accesses — because a
// thread walks its env's candidates a dozen times.  fp32 throughout (see the oracle's note on the
// reference's NumPy-version-dependent scalar promotion).
#include <cstring>

#include "eb_env_device.h"

#pragma clang fp contract(off)

namespace eb {


// (ego and next_ego may be the same buffer: eb_env_step updates the state in place)
__global__ void env_ego_step_kernel(int n, const float* ego, const float* __restrict__ actions,
                                    float* next_ego, float* __restrict__ params) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float st[6], nx[6], pr[4];
#pragma unroll
    for (int c = 0; c < 6; ++c) st[c] = ego[6 * (size_t)i + c];
    env_ego_step_row(st, actions[2 * (size_t)i], actions[2 * (size_t)i + 1], nx, pr);
#pragma unroll
    for (int c = 0; c < 6; ++c) next_ego[6 * (size_t)i + c] = nx[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) params[4 * (size_t)i + c] = pr[c];
}

// The first three calls of CrossroadEnd2end.step for one env in one launch (eb_env_step): action scaling (E2E:133),
// reward on the current obs (E2E:134) and the ego step (E2E:135), each the same device function its own kernel runs.
template <int TASK>
__global__ void env_pre_kernel(int n_env, int D, int n_future, int NV, const float* __restrict__ obs,
                               const float* __restrict__ raw, float* __restrict__ scaled, float* __restrict__ out5,
                               float* __restrict__ d16, float* __restrict__ ego, float* __restrict__ params) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env) return;
    float steer, a_x;
    action_transform(raw[2 * (size_t)i], raw[2 * (size_t)i + 1], steer, a_x);
    scaled[2 * (size_t)i] = steer;
    scaled[2 * (size_t)i + 1] = a_x;
    rewards_env<TASK>(i, n_env, D, n_future, NV, obs, steer, a_x, out5, d16);
    float st[6], nx[6], pr[4];
#pragma unroll
    for (int c = 0; c < 6; ++c) st[c] = ego[6 * (size_t)i + c];
    env_ego_step_row(st, steer, a_x, nx, pr);
#pragma unroll
    for (int c = 0; c < 6; ++c) ego[6 * (size_t)i + c] = nx[c];
#pragma unroll
    for (int c = 0; c < 4; ++c) params[4 * (size_t)i + c] = pr[c];
}

hipError_t launch_env_pre(int task, int n_env, int D, int n_future, int NV, const float* obs, const float* raw,
                          float* scaled, float* out5, float* d16, float* ego, float* params, hipStream_t s) {
    const dim3 g((n_env + 127) / 128), b(128);
    switch (task) {
        case TASK_LEFT: hipLaunchKernelGGL(env_pre_kernel<TASK_LEFT>, g, b, 0, s, n_env, D, n_future, NV, obs, raw, scaled, out5, d16, ego, params); break;
        case TASK_STRAIGHT: hipLaunchKernelGGL(env_pre_kernel<TASK_STRAIGHT>, g, b, 0, s, n_env, D, n_future, NV, obs, raw, scaled, out5, d16, ego, params); break;
        default: hipLaunchKernelGGL(env_pre_kernel<TASK_RIGHT>, g, b, 0, s, n_env, D, n_future, NV, obs, raw, scaled, out5, d16, ego, params); break;
    }
    return hipGetLastError();
}

hipError_t launch_env_ego_step(int n, const float* ego, const float* actions, float* next_ego, float* params,
                               hipStream_t s) {
    hipLaunchKernelGGL(env_ego_step_kernel, dim3((n + 127) / 128), dim3(128), 0, s, n, ego, actions, next_ego, params);
    return hipGetLastError();
}



__global__ void judge_done_kernel(int task, int n_env, int D, const float* __restrict__ ego,
                                  const float* __restrict__ params, const float* __restrict__ obs, int m_cand,
                                  const float* __restrict__ cand, const uint8_t* __restrict__ cand_mode,
                                  const float* __restrict__ cand_lw, const uint8_t* __restrict__ v_light,
                                  uint8_t* __restrict__ done_code) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env) return;
    const float* e = ego + 6 * (size_t)i;
    const float v_x = e[0], r = e[2], x = e[3], y = e[4], phi = e[5];
    const EgoCircles E = ego_circles(x, y, phi);
    bool collision = false;
    for (int k = 0; k < m_cand; ++k) {
        const size_t ck = (size_t)i * m_cand + k;
        if (cand_mode[ck] == EB_VMODE_EMPTY) continue;
        const float4 v = make_float4(cand[ck * 4], cand[ck * 4 + 1], cand[ck * 4 + 2], cand[ck * 4 + 3]);
        collision = collision || collision_with(E, x, y, v, cand_lw ? cand_lw[ck * 2] : 4.8f, cand_lw ? cand_lw[ck * 2 + 1] : 2.0f);
    }
    done_code[i] = judge_code(task, collision, v_x, r, x, y, phi, params[4 * (size_t)i + 3], obs[(size_t)D * i + 6],
                              v_light && v_light[i] != 0);
}

size_t get_obs_lds_bytes(int D, int m_cand) {
    const int rs4 = m_cand + ((m_cand & 1) ? 2 : 1);
    return (size_t)64 * rs4 * 16 + (size_t)64 * (D + 1) * 4 + (size_t)64 * (m_cand + 4) + (size_t)256 * (m_cand + 1);
}

// _judge_done appended to the observation kernel (eb_env_step): the tile's candidates and the new delta_y are in LDS
struct JudgeArgs {
    const float* params;      // [n_env, 4]
    const float* cand_lw;     // [n_env, m_cand, 2] or NULL
    uint8_t* done_code;       // NULL: observation only (eb_get_obs)
};

template <int TASK, bool STAGED>
__global__ __launch_bounds__(256) void get_obs_kernel(int n_env, int D, int n_future, int NV, PathTables pt, VehModes modes,
                               const float* __restrict__ ego, const int* __restrict__ ref_idx, int path_id,
                               int m_cand, const float* __restrict__ cand_all, const uint8_t* __restrict__ cmode_all,
                               const uint8_t* __restrict__ v_light, const uint8_t* __restrict__ virtual_flag,
                               float* __restrict__ obs_out, const JudgeArgs J, const uint8_t* __restrict__ row_mask) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // The slot modes come in the kernel-argument block; indexed with a loop variable they would be fetched from there by a
    // vector load each time — a memory round trip per access inside the slot loops below, which is what this kernel
    // used to spend most of its time on (43 us with 4 candidates per env).  One copy into LDS per block instead.
    __shared__ uint8_t smode[64];
    if (threadIdx.x < 64) smode[threadIdx.x] = modes.mode[threadIdx.x];
    if (!STAGED) __syncthreads();   // (the staged form has its own barrier below)
    // 64 envs per block, one per lane; the block's four waves share the slots of the observation (wave w builds
    // slots w, w + 4, ...), wave 0 also the ego and tracking columns
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, e0 = blockIdx.x * 64;
    const int nE = n_env - e0 < 64 ? n_env - e0 : 64;
    const int i = e0 + lane;
    const int RS4 = obs_cand_stride4(m_cand), OS = D + 1, MS = m_cand + 4;
    float4* s_cand = reinterpret_cast<float4*>(smem);                       // [64][RS4]
    float* s_out = reinterpret_cast<float*>(smem + (size_t)64 * RS4 * 16);  // [64][D + 1]
    uint8_t* s_mode = smem + (size_t)64 * RS4 * 16 + (size_t)64 * OS * 4;   // [64][m_cand + 4]
    uint8_t* s_list = s_mode + (size_t)64 * MS;                             // [4 waves][64][m_cand + 1] in-range index lists
    if (STAGED) {
        // the tile's candidates are contiguous in memory: coalesced 16-byte loads, one LDS row per env
        const float4* src = reinterpret_cast<const float4*>(cand_all) + (size_t)e0 * m_cand;
        const uint8_t* msrc = cmode_all + (size_t)e0 * m_cand;
        const int total = nE * m_cand;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int e = idx / m_cand, c = idx - e * m_cand;
            s_cand[e * RS4 + c] = src[idx];
            s_mode[e * MS + c] = msrc[idx];
        }
        __syncthreads();
    }
    if (i < n_env && (STAGED || !row_mask || row_mask[i])) {
    const float* e = ego + 6 * (size_t)i;
    float* o = STAGED ? s_out + lane * OS : obs_out + (size_t)D * i;
    const int T = 3 * (n_future + 1);
    const float ev = e[0], ex = e[3], ey = e[4], ephi = e[5];
    if (wave == 0) {
#pragma unroll
    for (int c = 0; c < 6; ++c) o[c] = e[c];                               // E2E:329-338
    const int p = row_path(pt, ref_idx, path_id, i);
    if (p < 0) { for (int c = 0; c < T; ++c) o[6 + c] = 0.0f; }
    else {
        // tracking_error_vector on the env's path, E2E:293-297
        // closest table point: the cell grid names the index range that holds it (eb_capi.hip:build_cell_grid);
        // outside the grid, the reference's full scan (DAM:702-715)
        const float2* red = pt.red[p];
        const float fx = (ex - pt.gx0) * CELL_INV, fy = (ey - pt.gy0) * CELL_INV;
        int bi = 0;
        unsigned c = 0xffffffffu;                                              // (also a corridor cell on the path's medial axis: eb_capi.hip)
        if (fx >= 0.0f && fx < (float)pt.gnx && fy >= 0.0f && fy < (float)pt.gny) c = pt.cells[(p * pt.gny + (int)fy) * pt.gnx + (int)fx];
        if (c != 0xffffffffu) {
            const int lo = (int)(c & 0xffffu), hi = (int)(c >> 16);
            float best = __builtin_inff();
            // four table points per trip (two 16-byte loads in flight instead of one dependent 8-byte load per point; the
            // table is readable four entries past its end, eb_set_paths); same order, same strict '<': same index
            for (int r = lo; r <= hi; r += 4) {
                typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));   // two table entries: 8-byte aligned
                const f4a8 q01 = *reinterpret_cast<const f4a8*>(red + r), q23 = *reinterpret_cast<const f4a8*>(red + r + 2);
                const float d0 = sq(ex - q01.x) + sq(ey - q01.y), d1 = sq(ex - q01.z) + sq(ey - q01.w);
                const float d2 = sq(ex - q23.x) + sq(ey - q23.y), d3 = sq(ex - q23.z) + sq(ey - q23.w);
                if (d0 < best) { best = d0; bi = r; }
                if (r + 1 <= hi && d1 < best) { best = d1; bi = r + 1; }
                if (r + 2 <= hi && d2 < best) { best = d2; bi = r + 2; }
                if (r + 3 <= hi && d3 < best) { best = d3; bi = r + 3; }
            }
        } else {
            // an ego that has left the map (or NaN): the exact pruned search over block centres and radii — the index of
            // the reference's full scan after ~2 x 32 + 16..48 evaluations with grouped loads, instead of ~380 in a chain
            int lo, hi, lo2, hi2;   // (the coarse level first: eb_device.h:coarse_cell_range)
            float rx, ry, rphi;
            const int how = coarse_cell_ranges(pt, p, ex, ey, lo, hi, lo2, hi2);
            if (how == 1) bi = closest_in_ranges(reinterpret_cast<const float*>(red), pt.phi10[p], lo, hi, lo2, hi2, ex, ey, rx, ry, rphi);
            else bi = closest_reduced_index<8>(red, pt.rad + 32 * p, pt.red_len[p], ex, ey, how == 2 ? lo : 0, how == 2 ? hi : 1 << 30);
        }
        const int idx = bi * 10, len = pt.len[p];
        const int ci = clamp_index(idx, len);
        o[6] = two2one<TASK>(ex, ey, pt.x[p][ci], pt.y[p][ci]);
        o[7] = deal_with_phi_diff(ephi - pt.phi[p][ci]);
        o[8] = ev - EXP_V;
        int cur = idx;
        for (int k = 0; k < n_future; ++k) {
            cur += 80;
            if (cur >= len - 2) cur = len - 2;
            const int fi = clamp_index(cur, len);
            o[9 + 3 * k] = pt.x[p][fi] - ex;
            o[10 + 3 * k] = pt.y[p][fi] - ey;
            o[11 + 3 * k] = deal_with_phi_diff(ephi - pt.phi[p][fi]);
        }
    }
    }   // wave 0
    const float* cand = cand_all + (size_t)i * m_cand * 4;
    const uint8_t* cmode = cmode_all + (size_t)i * m_cand;
    const float4* crow = s_cand + lane * RS4;
    const uint8_t* mrow = s_mode + lane * MS;
    const bool light = (v_light && v_light[i] != 0) || (virtual_flag && virtual_flag[i] != 0);        // E2E:387-388
    const bool virt = TASK != TASK_RIGHT && light && ey < -HALF_CROSS;                                // E2E:386-388
    float* ov = o + 6 + T;
    if (STAGED) {
        // The distinct modes of the slot list are dealt round-robin to the four waves.  For its mode a lane first
        // compacts the in-range candidates of that mode into a short index list (one walk over the candidates),
        // then fills the mode's slots in order, each the next candidate after the previous pick under (sort key,
        // insertion order) — the same sequence of picks as one selection pass per rank from scratch (E2E:414-437).
        uint8_t* list = s_list + (wave * 64 + lane) * (m_cand + 1);
        int distinct = 0;
        for (int s = 0; s < NV; ++s) {
            const int m = smode[s];
            bool first = true;
            for (int t = 0; t < s; ++t) first = first && smode[t] != m;
            if (!first) continue;
            if ((distinct++ & 3) != wave) continue;
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
    } else {
    for (int s = 0; s < NV; ++s) {
        const int m = smode[s];
        int rank = 0;
        for (int t = 0; t < s; ++t) rank += smode[t] == m;
        // select the rank-th candidate of mode m under (sort key, insertion order): E2E:414-437
        V4 prev = {0, 0, 0, 0};
        int prev_i = -1;
        bool found = true;
        for (int it = 0; it <= rank && found; ++it) {
            V4 best = {0, 0, 0, 0};
            int best_i = -1;
            for (int c = 0; c <= m_cand; ++c) {
                V4 v;
                if (!fetch_candidate(m, c, m_cand, cand, cmode, virt, v)) continue;
                if (!veh_in_range(TASK, m, v, ex, ey)) continue;
                if (prev_i >= 0) {
                    const int cp = veh_cmp(TASK, m, prev, v);
                    if (!(cp < 0 || (cp == 0 && prev_i < c))) continue;   // not after the previous pick
                }
                if (best_i < 0 || veh_cmp(TASK, m, v, best) < 0) { best = v; best_i = c; }
            }
            if (best_i < 0) found = false;
            else { prev = best; prev_i = best_i; }
        }
        const V4 r = found ? prev : veh_fill_value(m);                     // slice_or_fill, E2E:431-437
        ov[4 * s] = r.x; ov[4 * s + 1] = r.y; ov[4 * s + 2] = r.v; ov[4 * s + 3] = r.phi;
    }
    }   // !STAGED
    }   // i < n_env
    if (STAGED) {
        __syncthreads();
        if (J.done_code) {
            // E2E:141 on the state just built: the four waves share the collision test (candidates w, w + 4, ...),
            // wave 0 folds the flags and walks the priority chain
            bool col = false;
            if (i < n_env) {
                const float* e = ego + 6 * (size_t)i;
                const EgoCircles E = ego_circles(e[3], e[4], e[5]);
                const float4* crow = s_cand + lane * RS4;
                const uint8_t* mrow = s_mode + lane * MS;
                for (int c = wave; c < m_cand; c += 4) {
                    if (mrow[c] == EB_VMODE_EMPTY) continue;
                    const size_t ck = (size_t)i * m_cand + c;
                    col = col || collision_with(E, e[3], e[4], crow[c], J.cand_lw ? J.cand_lw[ck * 2] : 4.8f,
                                                J.cand_lw ? J.cand_lw[ck * 2 + 1] : 2.0f);
                }
            }
            s_list[wave * 64 + lane] = col ? 1 : 0;                         // the index lists are dead by now
            __syncthreads();
            if (wave == 0 && i < n_env) {
                const float* e = ego + 6 * (size_t)i;
                const bool collision = (s_list[lane] | s_list[64 + lane] | s_list[128 + lane] | s_list[192 + lane]) != 0;
                J.done_code[i] = judge_code(TASK, collision, e[0], e[2], e[3], e[4], e[5], J.params[4 * (size_t)i + 3],
                                            s_out[lane * OS + 6], v_light && v_light[i] != 0);
            }
        }
        float* dst = obs_out + (size_t)e0 * D;                              // the tile's rows are contiguous too
        const int total = nE * D;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int e = idx / D, c = idx - e * D;
            dst[idx] = s_out[e * OS + c];
        }
    }
}

bool get_obs_is_staged(int D, int m_cand, const float* cand) {
    return get_obs_lds_bytes(D, m_cand) <= 150 * 1024 && (reinterpret_cast<uintptr_t>(cand) & 15) == 0;
}

// ---- the 12-ego scene's exit-relative frames (multi_ego.py:33, 84-120; UTL:120-196; E2E:345-385) ----
// One thread per env, the oracle's algorithm line for line: a vehicle's (x, y, phi) are float64 after
// cal_info_in_transform_coordination, compared with ego-derived fp32 values after rounding to fp32 (NumPy >= 2 scalar
// rules) and with python constants / each other in float64; the observation holds them rounded to fp32.
struct V4d { double x, y, phi; float v; };
EB_DEV bool lt_ego(double a, float b) { return (float)a < b; }
EB_DEV bool gt_ego(double a, float b) { return (float)a > b; }

EB_DEV bool veh_in_range_d(int task, int m, const V4d& v, float ego_x, float ego_y) {   // E2E:393-411
    const double C2 = 25.0;
    switch (m) {
        case EB_VMODE_DL: return v.x > -C2 - 10 && gt_ego(v.y, ego_y - 2.0f);
        case EB_VMODE_DU: return gt_ego(v.y, ego_y - 2.0f) && v.y < C2 + 10 && lt_ego(v.x, ego_x + 5.0f);
        case EB_VMODE_DR: return v.x < C2 + 10 && gt_ego(v.y, ego_y);
        case EB_VMODE_RU: return v.x < C2 + 10 && v.y < C2 + 10;
        case EB_VMODE_UR:
            if (task == TASK_STRAIGHT) return lt_ego(v.x, ego_x + 7.0f) && gt_ego(v.y, ego_y) && v.y < C2 + 10;
            if (task == TASK_RIGHT) return v.x < C2 + 10 && v.y < C2;
            return true;
        case EB_VMODE_UD: {
            const float ey2 = ego_y - 2.0f;
            const bool lower = (-25.0f > ey2) ? (-C2 < v.y) : gt_ego(v.y, ey2);
            return lower && v.y < C2 && lt_ego(v.x, ego_x);
        }
        case EB_VMODE_UL: return -C2 - 10 < v.x && lt_ego(v.x, ego_x) && v.y < C2;
        case EB_VMODE_LR: return -C2 - 10 < v.x && v.x < C2 + 10;
        default: return true;
    }
}

EB_DEV int veh_cmp_d(int task, int m, const V4d& a, const V4d& b) {
#define EB_ASC(f) do { if (a.f < b.f) return -1; if (a.f > b.f) return 1; } while (0)
#define EB_DESC(f) do { if (a.f > b.f) return -1; if (a.f < b.f) return 1; } while (0)
    switch (m) {
        case EB_VMODE_DL: EB_ASC(y); EB_DESC(x); return 0;
        case EB_VMODE_DU: EB_ASC(y); return 0;
        case EB_VMODE_DR: EB_ASC(y); EB_ASC(x); return 0;
        case EB_VMODE_RU: EB_ASC(x); EB_DESC(y); return 0;
        case EB_VMODE_UR:
            if (task == TASK_STRAIGHT) { EB_ASC(y); return 0; }
            if (task == TASK_RIGHT) { EB_ASC(y); EB_DESC(x); return 0; }
            return 0;
        case EB_VMODE_UD: EB_ASC(y); return 0;
        case EB_VMODE_UL: EB_ASC(y); EB_ASC(x); return 0;
        case EB_VMODE_LR: EB_DESC(x); return 0;
        default: return 0;
    }
#undef EB_ASC
#undef EB_DESC
}

EB_DEV V4d veh_fill_value_d(int m) {   // mode2fillvalue, E2E:439-447 (python floats)
    const double C2 = 25.0, LW = 3.75;
    V4d f = {0.0, 0.0, 0.0, 0.0f};
    switch (m) {
        case EB_VMODE_DL: f.x = LW / 2; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_DU: f.x = LW * 1.5; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_DR: f.x = LW * 2.5; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_RU: f.x = C2 + 15; f.y = LW * 2.5; f.phi = 180; break;
        case EB_VMODE_UR: f.x = -LW / 2; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_UD: f.x = -LW * 1.5; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_UL: f.x = -LW * 2.5; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_LR: f.x = -(C2 + 20); f.y = -LW * 1.5; f.phi = 0; break;
        default: break;
    }
    return f;
}

// route of a WORLD mode seen from exit k: world direction d -> (d - k) mod 4 (E2E:345-385)
EB_DEV int exit_relative_mode(int world_mode, int k) {
    if (world_mode < 0 || world_mode >= EB_VMODE_COUNT) return EB_VMODE_EMPTY;
    const int st = world_mode / 3, en = (3 - world_mode % 3 + st) & 3;   // end directions of dl du dr | rd rl ru | ur ud ul | lu lr ld
    const int rs = (st - k + 4) & 3, re = (en - k + 4) & 3;
    return rs * 3 + ((rs - re + 4) & 3) - 1;                            // [start][end] -> EB_VMODE_*
}

// candidate c of the env under exit k and mode m, or the virtual red-light car at c == m_cand (E2E:386-390)
EB_DEV bool fetch_candidate_exit(int m, int c, int m_cand, const float* cand, const uint8_t* cmode, bool virt, int k,
                                 const ExitConsts& xc, V4d& v) {
    if (c < m_cand) {
        if (exit_relative_mode(cmode[c], k) != m) return false;
        const double x = (double)cand[4 * c] - 0, y = (double)cand[4 * c + 1] - 0;   // shift by (0, 0), UTL:116-117
        v.x = x * xc.c[k] + y * xc.s[k];                                             // UTL:131
        v.y = -x * xc.s[k] + y * xc.c[k];                                            // UTL:132
        const int ang = k == 0 ? 0 : k == 1 ? 90 : k == 2 ? 180 : -90;
        double t = (double)cand[4 * c + 3] - ang;                                    // UTL:133-139
        if (!wrap_bounded(t)) {}
        else if (t > 180) { while (t > 180) t = t - 360; }
        else if (t <= -180) { while (t <= -180) t = t + 360; }
        v.phi = t;
        v.v = cand[4 * c + 2];
        return true;
    }
    if (!virt || (m != EB_VMODE_DL && m != EB_VMODE_DU)) return false;
    v.x = m == EB_VMODE_DL ? 3.75 / 2 : 3.75 * 1.5;
    v.y = -25.0 + 2.5; v.v = 0.0f; v.phi = 90.0;
    return true;
}

template <int TASK>
__global__ __launch_bounds__(64) void get_obs_exit_kernel(int n_env, int D, int n_future, int NV, PathTables pt, VehModes modes,
                                    const float* __restrict__ ego, const int* __restrict__ ref_idx, int path_id,
                                    int m_cand, const float* __restrict__ cand_all, const uint8_t* __restrict__ cmode_all,
                                    const uint8_t* __restrict__ v_light, const uint8_t* __restrict__ virtual_flag,
                                    const uint8_t* __restrict__ exit_id, const ExitConsts xc, const uint8_t* __restrict__ row_mask,
                                    float* __restrict__ obs_out) {
    __shared__ uint8_t smode[64];                                           // see get_obs_kernel
    smode[threadIdx.x] = modes.mode[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env) return;
    if (row_mask && !row_mask[i]) return;
    const float* e = ego + 6 * (size_t)i;
    float* o = obs_out + (size_t)D * i;
    const int T = 3 * (n_future + 1);
    const float ev = e[0], ex = e[3], ey = e[4], ephi = e[5];
#pragma unroll
    for (int c = 0; c < 6; ++c) o[c] = e[c];                               // E2E:329-338
    const int p = row_path(pt, ref_idx, path_id, i);
    if (p < 0) { for (int c = 0; c < T; ++c) o[6 + c] = 0.0f; }
    else {
        const float2* red = pt.red[p];
        const int nr = pt.red_len[p];
        float best = __builtin_inff();
        int bi = 0;
        for (int r = 0; r < nr; ++r) {                                      // DAM:702-715 (full scan: this is not the hot path)
            const float2 q = red[r];
            const float d = sq(ex - q.x) + sq(ey - q.y);
            if (d < best) { best = d; bi = r; }
        }
        const int idx = bi * 10, len = pt.len[p];
        const int ci = clamp_index(idx, len);
        o[6] = two2one<TASK>(ex, ey, pt.x[p][ci], pt.y[p][ci]);
        o[7] = deal_with_phi_diff(ephi - pt.phi[p][ci]);
        o[8] = ev - EXP_V;
        int cur = idx;
        for (int k = 0; k < n_future; ++k) {
            cur += 80;
            if (cur >= len - 2) cur = len - 2;
            const int fi = clamp_index(cur, len);
            o[9 + 3 * k] = pt.x[p][fi] - ex;
            o[10 + 3 * k] = pt.y[p][fi] - ey;
            o[11 + 3 * k] = deal_with_phi_diff(ephi - pt.phi[p][fi]);
        }
    }
    if (exit_id[i] > 3) {   // not an EB_EXIT_* id: no plausible-looking frame — the whole row is NaN (the oracle, which can read
        for (int c = 0; c < D; ++c) o[c] = __builtin_nanf("");   // its host arguments, refuses the call with EB_EINVAL)
        return;
    }
    const int k = exit_id[i];
    int vl = v_light ? v_light[i] : 0;
    if (k == EB_EXIT_R || k == EB_EXIT_L) vl = vl != 2 ? 2 : 0;            // multi_ego.py:89-92
    const bool light = vl != 0 || (virtual_flag && virtual_flag[i] != 0);  // E2E:387-388
    const bool virt = TASK != TASK_RIGHT && light && ey < -HALF_CROSS;
    const float* cand = cand_all + (size_t)i * m_cand * 4;
    const uint8_t* cmode = cmode_all + (size_t)i * m_cand;
    float* ov = o + 6 + T;
    for (int s = 0; s < NV; ++s) {
        const int m = smode[s];
        int rank = 0;
        for (int t = 0; t < s; ++t) rank += smode[t] == m;
        // select the rank-th candidate of mode m under (sort key, insertion order): E2E:414-437
        V4d prev = {0, 0, 0, 0};
        int prev_i = -1;
        bool found = true;
        for (int it = 0; it <= rank && found; ++it) {
            V4d bestv = {0, 0, 0, 0};
            int best_i = -1;
            for (int c = 0; c <= m_cand; ++c) {
                V4d v;
                if (!fetch_candidate_exit(m, c, m_cand, cand, cmode, virt, k, xc, v)) continue;
                if (!veh_in_range_d(TASK, m, v, ex, ey)) continue;
                if (prev_i >= 0) {
                    const int cp = veh_cmp_d(TASK, m, prev, v);
                    if (!(cp < 0 || (cp == 0 && prev_i < c))) continue;   // not after the previous pick
                }
                if (best_i < 0 || veh_cmp_d(TASK, m, v, bestv) < 0) { bestv = v; best_i = c; }
            }
            if (best_i < 0) found = false;
            else { prev = bestv; prev_i = best_i; }
        }
        const V4d r = found ? prev : veh_fill_value_d(m);                  // slice_or_fill, E2E:431-437
        ov[4 * s] = (float)r.x; ov[4 * s + 1] = (float)r.y; ov[4 * s + 2] = r.v; ov[4 * s + 3] = (float)r.phi;
    }
}

// cal_ego_info_in_transform_coordination (UTL:184-196) on fp32 fields
__global__ void exit_frame_kernel(int n, const uint8_t* __restrict__ exit_id, int inverse, const ExitConsts xc,
                                  const float* __restrict__ ego, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* e = ego + 6 * (size_t)i;
    float* o = out + 6 * (size_t)i;
    if (exit_id[i] > 3) {   // not an EB_EXIT_* id: NaN pose (see get_obs_exit_kernel)
        o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = o[4] = o[5] = __builtin_nanf("");
        return;
    }
    const int k = exit_id[i], q = inverse ? 4 + k : k;
    const int a0 = k == 0 ? 0 : k == 1 ? 90 : k == 2 ? 180 : -90;
    const float a = (float)(inverse ? -a0 : a0);
    const float c = xc.cf[q], sn = xc.sf[q];
    const float x = e[3] - 0.0f, y = e[4] - 0.0f;                          // UTL:116-117
    const float tx = x * c + y * sn;                                        // UTL:131
    const float ty = -x * sn + y * c;                                       // UTL:132
    float d = e[5] - a;                                                     // UTL:133-139
    if (!wrap_bounded(d)) {}
    else if (d > 180.0f) { while (d > 180.0f) d = d - 360.0f; }
    else if (d <= -180.0f) { while (d <= -180.0f) d = d + 360.0f; }
    o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = tx; o[4] = ty; o[5] = d;
}

hipError_t launch_exit_frame(int n, const uint8_t* exit_id, int inverse, const ExitConsts& xc, const float* ego, float* out,
                             hipStream_t s) {
    hipLaunchKernelGGL(exit_frame_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, exit_id, inverse, xc, ego, out);
    return hipGetLastError();
}

// dst[e, :] = src[e, :] for the rows with mask[e] != 0 (eb_env_step(auto_reset) on the separate-launch path: the terminal
// observations of the finished envs -> final_obs)
__global__ void copy_rows_masked_kernel(int n_env, int D, const uint8_t* mask, const float* src, float* dst) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n_env * D) return;
    if (mask[idx / D]) dst[idx] = src[idx];
}
hipError_t launch_copy_rows_masked(int n_env, int D, const uint8_t* mask, const float* src, float* dst, hipStream_t s) {
    const size_t total = (size_t)n_env * D;
    hipLaunchKernelGGL(copy_rows_masked_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n_env, D, mask, src, dst);
    return hipGetLastError();
}

// done_code != NULL appends _judge_done (needs the staged form: check get_obs_is_staged first)
hipError_t launch_get_obs(int task, int n_env, int D, int n_future, int NV, const PathTables& pt,
                          const VehModes& modes, const float* ego, const int* ref_idx, int path_id, int m_cand,
                          const float* cand, const uint8_t* cand_mode, const uint8_t* v_light, const uint8_t* virtual_flag,
                          float* obs_out, hipStream_t s, const float* params, const float* cand_lw,
                          uint8_t* done_code, const uint8_t* exit_id, const ExitConsts* xc, const uint8_t* row_mask,
                          const EnvResetArgs* reset, int tile_envs, int env_waves, long long* trace, long long trace_words,
                          int scan_one_trip) {
    if (reset && (exit_id || done_code || !env_step_is_fused(D, NV, m_cand, cand, ego, nullptr, nullptr, reset->params))) return hipErrorInvalidValue;
    if (exit_id) {
        if (done_code || !xc) return hipErrorInvalidValue;
        const dim3 g((n_env + 63) / 64), b(64);
        switch (task) {
            case TASK_LEFT: hipLaunchKernelGGL(get_obs_exit_kernel<TASK_LEFT>, g, b, 0, s, n_env, D, n_future, NV, pt, modes, ego, ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, exit_id, *xc, row_mask, obs_out); break;
            case TASK_STRAIGHT: hipLaunchKernelGGL(get_obs_exit_kernel<TASK_STRAIGHT>, g, b, 0, s, n_env, D, n_future, NV, pt, modes, ego, ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, exit_id, *xc, row_mask, obs_out); break;
            default: hipLaunchKernelGGL(get_obs_exit_kernel<TASK_RIGHT>, g, b, 0, s, n_env, D, n_future, NV, pt, modes, ego, ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, exit_id, *xc, row_mask, obs_out); break;
        }
        return hipGetLastError();
    }
    if (!done_code && env_step_is_fused(D, NV, m_cand, cand, ego, nullptr, nullptr, reset ? reset->params : nullptr)) {
        // the observation alone through the one-launch step's machinery (eb_env_step.hip, OBS variant): pair-parallel
        // staging and the bit-set slot selection instead of one lane per env walking its candidates
        EnvStepArgs A;
        std::memset(&A, 0, sizeof A);
        auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
        A.n_env = n_env; A.D = D; A.n_future = n_future; A.NV = NV; A.m_cand = m_cand; A.path_id = path_id;
        A.d_magic = magic(D); A.m_magic = magic(m_cand); A.nv_magic = magic(NV);
        A.pt = pt; A.modes = modes;
        env_step_slot_plan(modes, NV, A);
        A.ref_idx = ref_idx; A.ego = const_cast<float*>(ego); A.cand = const_cast<float*>(cand); A.cand_mode = cand_mode;
        A.v_light = v_light; A.virtual_flag = virtual_flag; A.obs_out = obs_out; A.obs_only = 1; A.row_mask = row_mask;
        A.tile_envs = tile_envs; A.waves = env_waves; A.trace = trace; A.trace_words = trace_words; A.scan_one_trip = scan_one_trip;
        if (reset) {   // eb_env_reset_pool: the masked rows' state is drawn in the same launch
            A.reset = 1; A.training = reset->training; A.reset_seed = reset->seed; A.reset_counter = reset->counter;
            A.params = reset->params; A.ref_idx_out = reset->ref_idx; A.virtual_flag = reset->virtual_flag; A.virtual_out = reset->virtual_flag;
            A.v_light = nullptr; A.v_light_out = reset->v_light; A.done_code = reset->done_code;
            A.pool_entry = reset->entry; A.pool_span = reset->span; A.pool_v_max = reset->v_max; A.edge_span = reset->edge_span;
            A.pool_seed = reset->pool_seed; A.pool_counter = reset->pool_counter;
            A.obs = reset->obs_src; A.done_src = reset->done_src; A.episode_step = reset->episode_step;
        }
        return launch_env_step(task, A, s);
    }
    const size_t lds = get_obs_lds_bytes(D, m_cand);
    const bool staged = !row_mask && get_obs_is_staged(D, m_cand, cand);   // (a masked pass outside the one-launch machinery: one thread per env)
    if (done_code && !staged) return hipErrorInvalidValue;
    const JudgeArgs J{params, cand_lw, done_code};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
    hipError_t e = hipSuccess;
#define EB_GET_OBS(T)                                                                                                \
    do {                                                                                                             \
        const dim3 g((n_env + 63) / 64), b(staged ? 256 : 64);                                                       \
        if (staged) {                                                                                                \
            static size_t granted[64];   /* the > 48 KB opt-in is per kernel and device, and sticky */               \
            if (lds > 48 * 1024 && lds > granted[dev]) {                                                             \
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&get_obs_kernel<T, true>),                    \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
                if (e == hipSuccess) granted[dev] = lds;                                                             \
            }                                                                                                        \
            if (e == hipSuccess)                                                                                     \
                hipLaunchKernelGGL((get_obs_kernel<T, true>), g, b, lds, s, n_env, D, n_future, NV, pt, modes, ego,  \
                                   ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, obs_out, J, row_mask); \
        } else {                                                                                                     \
            hipLaunchKernelGGL((get_obs_kernel<T, false>), g, b, 0, s, n_env, D, n_future, NV, pt, modes, ego,       \
                               ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, obs_out, J, row_mask); \
        }                                                                                                            \
    } while (0)
    switch (task) {
        case TASK_LEFT: EB_GET_OBS(TASK_LEFT); break;
        case TASK_STRAIGHT: EB_GET_OBS(TASK_STRAIGHT); break;
        default: EB_GET_OBS(TASK_RIGHT); break;
    }
#undef EB_GET_OBS
    return e != hipSuccess ? e : hipGetLastError();
}


__global__ void traffic_respawn_kernel(int n_env, int m_cand, float* __restrict__ cand, const float* __restrict__ entry,
                                       float limit, float span, float v_max, uint64_t seed, uint64_t counter,
                                       const uint8_t* __restrict__ env_mask, uint8_t* __restrict__ respawned,
                                       const float* __restrict__ ego, float edge_span) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_env * m_cand) return;
    const int e = idx / m_cand, j = idx - e * m_cand;
    float4* c = reinterpret_cast<float4*>(cand) + idx;
    const float4 v = *c;
    const bool chosen = !env_mask || env_mask[e] != 0;
    const bool gone = chosen && (limit < 0.0f || __builtin_fabsf(v.x) > limit || __builtin_fabsf(v.y) > limit);
    if (gone) {
        const uint64_t base = (counter << 32) + (uint64_t)e * 128u + (uint64_t)j * 2u;
        const float u1 = u01(seed, base), u2 = u01(seed, base + 1);
        const float* en = entry + 5 * j;
        float along = u1 * span;
        float4 nv = make_float4(en[0] + along * en[3], en[1] + along * en[4], u2 * v_max, en[2]);
        if (ego && init_conflict(ego + 6 * (size_t)e, 4.8f, nv.x, nv.y, nv.w, nv.z, 4.8f)) {   // TRF:168-192: not on top of the ego
            along = u1 * edge_span;
            nv.x = en[0] + along * en[3];
            nv.y = en[1] + along * en[4];
        }
        *c = nv;
    }
    if (respawned) respawned[idx] = gone ? 1 : 0;
}

hipError_t launch_traffic_respawn(int n_env, int m_cand, float* cand, const float* entry, float limit, float span,
                                  float v_max, uint64_t seed, uint64_t counter, const uint8_t* env_mask, uint8_t* respawned,
                                  hipStream_t s, const float* ego, float edge_span) {
    const int n = n_env * m_cand;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(traffic_respawn_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n_env, m_cand, cand, entry, limit,
                       span, v_max, seed, counter, env_mask, respawned, ego, edge_span);
    return hipGetLastError();
}

// ---- a18: CrossroadEnd2end.reset + _reset_init_state for the masked envs (E2E:99-127, 472-499), one thread per env ----
__global__ void env_reset_kernel(int task, int n_env, PathTables pt, const uint8_t* __restrict__ mask, uint64_t seed,
                                 uint64_t counter, int training, float* __restrict__ ego, float* __restrict__ params,
                                 int* __restrict__ ref_idx, uint8_t* __restrict__ virtual_next, uint8_t* __restrict__ done_code,
                                 uint8_t* __restrict__ v_light, int* __restrict__ episode_step) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    if (mask && !mask[e]) return;
    const float span = task == TASK_LEFT ? 900 + 500 : task == TASK_STRAIGHT ? 1200 + 500 : 420 + 500;   // E2E:473-478
    const uint64_t base = (counter << 32) + (uint64_t)e * 128u;
    const float u0 = u01(seed, base), u1 = u01(seed, base + 1), u2 = u01(seed, base + 2), u3 = u01(seed, base + 3);
    int p = (int)(u0 * (float)pt.n_paths);                                 // DAM:591
    if (p > pt.n_paths - 1) p = pt.n_paths - 1;
    const int index = (int)(u1 * span) + 700;                              // E2E:474-478
    const int ci = clamp_index(index, pt.len[p]);                          // indexs2points, DAM:727-728
    float* g = ego + 6 * (size_t)e;
    g[0] = 8.0f * u2; g[1] = 0.0f; g[2] = 0.0f;                            // E2E:482-486
    g[3] = pt.x[p][ci]; g[4] = pt.y[p][ci]; g[5] = pt.phi[p][ci];
    float* pr = params + 4 * (size_t)e;
    pr[0] = 0.0f; pr[1] = 0.0f; pr[2] = VehParams::miu; pr[3] = VehParams::miu;   // E2E:110-113
    ref_idx[e] = p;
    if (virtual_next) virtual_next[e] = (training && u3 > 0.9f) ? 1 : 0;  // E2E:120-126
    if (done_code) done_code[e] = EB_DONE_NOT_YET;                         // E2E:119
    if (v_light) v_light[e] = 0;                                           // (eb_env_reset_pool: an episode starts at phase 0)
    if (episode_step) episode_step[e] = 0;                                 // eb_time_limit's count
}

// eb_env_reset_pool's last stage: the flags eb_env_reset drew replace the old ones AFTER the reset observation (E2E:120-126)
__global__ void flag_swap_kernel(int n_env, const uint8_t* __restrict__ mask, const uint8_t* __restrict__ next, uint8_t* __restrict__ flag) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_env && (!mask || mask[e])) flag[e] = next[e];
}
hipError_t launch_flag_swap(int n_env, const uint8_t* mask, const uint8_t* next, uint8_t* flag, hipStream_t s) {
    if (n_env <= 0) return hipSuccess;
    hipLaunchKernelGGL(flag_swap_kernel, dim3((n_env + 255) / 256), dim3(256), 0, s, n_env, mask, next, flag);
    return hipGetLastError();
}

hipError_t launch_env_reset(int task, int n_env, const PathTables& pt, const uint8_t* mask, uint64_t seed, uint64_t counter,
                            int training, float* ego, float* params, int* ref_idx, uint8_t* virtual_next, uint8_t* done_code,
                            hipStream_t s, uint8_t* v_light, int* episode_step) {
    if (n_env <= 0) return hipSuccess;
    hipLaunchKernelGGL(env_reset_kernel, dim3((n_env + 255) / 256), dim3(256), 0, s, task, n_env, pt, mask, seed, counter,
                       training, ego, params, ref_idx, virtual_next, done_code, v_light, episode_step);
    return hipGetLastError();
}

// eb_time_limit behind eb_judge_done (eb_env_step as separate launches): gym's TimeLimit — count the step, end the episode
// nothing else has ended when the count reaches the limit, restart the count of a finished env
__global__ void time_limit_kernel(int n_env, int* __restrict__ episode_step, int max_episode_steps, uint8_t* __restrict__ done_code) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_env) return;
    const int cnt = episode_step[e] + 1;
    uint8_t code = done_code[e];
    if (code == EB_DONE_NOT_YET && cnt >= max_episode_steps) { code = EB_DONE_TIME_LIMIT; done_code[e] = code; }
    episode_step[e] = code != EB_DONE_NOT_YET ? 0 : cnt;
}
hipError_t launch_time_limit(int n_env, int* episode_step, int max_episode_steps, uint8_t* done_code, hipStream_t s) {
    if (n_env <= 0) return hipSuccess;
    hipLaunchKernelGGL(time_limit_kernel, dim3((n_env + 255) / 256), dim3(256), 0, s, n_env, episode_step, max_episode_steps, done_code);
    return hipGetLastError();
}

// a15: CrossroadEnd2end._get_ego_dynamics (E2E:150-183) for a batch, one thread per env — the functions judge_bits evaluates
// (eb_env_device.h): what the done code is decided on is what this entry exports
__global__ void ego_dynamics_kernel(int n, const float* __restrict__ ego, const float* __restrict__ params, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* e = ego + 6 * (size_t)i;
    const float v_x = e[0], x = e[3], y = e[4], phi = e[5];
    const float miu_f = params[4 * (size_t)i + 2], miu_r = params[4 * (size_t)i + 3];
    float* o = out + 11 * (size_t)i;
    o[0] = ego_alpha_bound(miu_f, VehParams::F_zf, VehParams::C_f);          // E2E:164-165
    o[1] = ego_alpha_bound(miu_r, VehParams::F_zr, VehParams::C_r);          // E2E:166
    o[2] = ego_r_bound(miu_r, v_x);                                          // E2E:167
    float rs, rc;
    sincos_det(-phi * PI_F / 180.0f, rs, rc);
#pragma unroll
    for (int q = 0; q < 4; ++q) ego_corner(q, x, y, rs, rc, o[3 + 2 * q], o[4 + 2 * q]);   // E2E:171-176
}
hipError_t launch_ego_dynamics(int n, const float* ego, const float* params, float* out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(ego_dynamics_kernel, dim3((n + 127) / 128), dim3(128), 0, s, n, ego, params, out);
    return hipGetLastError();
}

// ---- Traffic.init_traffic's role for the flow source (TRF:151-195), one thread per (env, route) ----
__global__ void traffic_flow_reset_kernel(int n_env, int K, const uint8_t* __restrict__ mask, const float* __restrict__ ego,
                                          float* __restrict__ cand, uint8_t* __restrict__ active, float* __restrict__ timer,
                                          int* __restrict__ emitted, int* __restrict__ sim_step, uint8_t* __restrict__ phase0,
                                          const float* __restrict__ lane, const float* __restrict__ period,
                                          const float* __restrict__ v_max, const float* __restrict__ cand_len, float lane_len,
                                          int random_phase, int training, uint64_t seed, uint64_t counter,
                                          uint8_t* __restrict__ cand_mode, uint8_t* __restrict__ v_light) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_env * 12) return;
    const int e = idx / 12, r = idx - e * 12, M = 12 * K;
    if (mask && !mask[e]) return;
    const uint64_t env_base = (counter << 32) + (uint64_t)e * 256u;
    float expect = lane_len / 7.5f / period[r];
    if (expect > (float)K) expect = (float)K;
    const float p = expect / (float)K;
    for (int k = 0; k < K; ++k) {
        const int j = r * K + k;
        const size_t s = (size_t)e * M + j;
        const float u0 = u01(seed, env_base + 4u * j), u1 = u01(seed, env_base + 4u * j + 1), u2 = u01(seed, env_base + 4u * j + 2);
        bool on = u0 < p;
        if (on) {
            const float* ln = lane + 5 * j;
            const float along = u1 * lane_len;
            const float4 c = make_float4(ln[0] + along * ln[3], ln[1] + along * ln[4], u2 * v_max[j], ln[2]);
            reinterpret_cast<float4*>(cand)[s] = c;
            if (init_conflict(ego + 6 * (size_t)e, 4.8f, c.x, c.y, c.w, c.z, cand_len[j])) on = false;
        }
        active[s] = on ? 1 : 0;
        cand_mode[s] = on ? (uint8_t)r : (uint8_t)EB_VMODE_EMPTY;
    }
    timer[idx] = u01(seed, env_base + 4u * (r * K) + 3) * period[r];
    emitted[idx] = 0;
    if (r == 0) {
        sim_step[e] = 0;
        const uint8_t ph = (random_phase && u01(seed, env_base + 255u) > 0.5f) ? 2 : 0;   // TRF:158-161
        phase0[e] = ph;
        v_light[e] = training ? ph : 0;                                                    // TRF:222-223
    }
}

hipError_t launch_traffic_flow_reset(int n_env, int K, const uint8_t* mask, const float* ego, float* cand, uint8_t* active,
                                     float* timer, int* emitted, int* sim_step, uint8_t* phase0, const float* lane,
                                     const float* period, const float* v_max, const float* cand_len, float lane_len,
                                     int random_phase, int training, uint64_t seed, uint64_t counter, uint8_t* cand_mode,
                                     uint8_t* v_light, hipStream_t s) {
    const int n = n_env * 12;
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(traffic_flow_reset_kernel, dim3((n + 127) / 128), dim3(128), 0, s, n_env, K, mask, ego, cand, active,
                       timer, emitted, sim_step, phase0, lane, period, v_max, cand_len, lane_len, random_phase, training, seed,
                       counter, cand_mode, v_light);
    return hipGetLastError();
}

// ---- flow traffic source (eb_traffic_flow_step) ------------------------------------------------------
// A block owns a tile of FS_ENVS envs (FS_ENVS * 12 * K <= 960 slot records).  The first version ran one thread per
// (env, route) walking its K slots: 16-byte accesses 80 bytes apart across a wave, 47 us at 65 536 envs x 60 slots for
// 144 MB of traffic.  Now: (1) one lane per slot record, coalesced — exit test / acceleration, the `on` flag to LDS;
// (2) one lane per (env, route) — first vacant slot from the LDS flags, timer, emission; (3) one lane per slot — active
// and mode bytes out.  Same arithmetic and the same per-slot order of decisions as before (and as the oracle's twin).
constexpr int FS_ENVS = 16;
__global__ __launch_bounds__(256) void traffic_flow_step_kernel(int n_env, int K, float* __restrict__ cand, uint8_t* __restrict__ active,
                                         float* __restrict__ timer, int* __restrict__ emitted, int* __restrict__ sim_step,
                                         const float* __restrict__ lane, const float* __restrict__ period,
                                         const float* __restrict__ v_max, float dt, float exit_range, float accel,
                                         float lane_len, int light_cycle, uint64_t seed, uint64_t counter,
                                         uint8_t* __restrict__ cand_mode, uint8_t* __restrict__ v_light, unsigned m_magic) {
    __shared__ uint8_t s_on[FS_ENVS * 64];
    const int M = 12 * K, e0 = blockIdx.x * FS_ENVS;
    const int nE = n_env - e0 < FS_ENVS ? n_env - e0 : FS_ENVS;
    const int n_rec = nE * M;
    float4* ctile = reinterpret_cast<float4*>(cand) + (size_t)e0 * M;
    uint8_t* atile = active + (size_t)e0 * M;
    for (int idx = threadIdx.x; idx < n_rec; idx += 256) {                 // (1) per slot record
        bool on = atile[idx] != 0;
        if (on) {
            float4 v = ctile[idx];
            const int e = m_magic ? (int)__umulhi((unsigned)idx, m_magic) : idx, j = idx - e * M;
            float sn, cs;
            sincos_det(deg2rad(v.w), sn, cs);
            const bool outward = v.x * cs + v.y * sn > 0.0f;
            if (__builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y)) > exit_range && outward) {
                on = false;
            } else {
                const float vn = v.z + accel * dt, vm = v_max[j];
                v.z = vn < vm ? vn : vm;
                ctile[idx] = v;
            }
        }
        s_on[idx] = on ? 1 : 0;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nE * 12; q += 256) {                      // (2) per (env, route)
        const int e = q / 12, r = q - e * 12;
        const int ge = e0 + e;
        uint8_t* on = s_on + e * M + r * K;
        int vacant = -1;
        for (int k = K - 1; k >= 0; --k)
            if (!on[k]) vacant = k;
        const size_t ti = (size_t)ge * 12 + r;
        float t = timer[ti] + dt;
        const float per = period[r];
        if (t >= per && vacant >= 0) {
            const int j = r * K + vacant;
            const uint64_t base = (counter << 32) + (uint64_t)ge * 128u + (uint64_t)(r * K) * 2u;
            const float u1 = (float)(splitmix64(seed + 0x9E3779B97F4A7C15ull * base) >> 40) * 5.9604644775390625e-8f;
            const float u2 = (float)(splitmix64(seed + 0x9E3779B97F4A7C15ull * (base + 1)) >> 40) * 5.9604644775390625e-8f;
            const float* ln = lane + 5 * j;
            const float along = u1 * lane_len;
            ctile[e * M + j] = make_float4(ln[0] + along * ln[3], ln[1] + along * ln[4], u2 * v_max[j], ln[2]);
            on[vacant] = 1;
            t = t - per;
            emitted[ti] += 1;
        }
        timer[ti] = t;
        if (r == 0) {
            const int n = sim_step[ge] + 1;
            sim_step[ge] = n;
            if (light_cycle) {   // a.net.xml:145-150: 25 s phase 0, 5 s phase 1, 25 s phase 2, 5 s phase 3, in steps of dt
                const float tt = (float)(n % (int)(60.0f / dt + 0.5f)) * dt;
                v_light[ge] = tt < 25.0f ? 0 : (tt < 30.0f ? 1 : (tt < 55.0f ? 2 : 3));
            }
        }
    }
    __syncthreads();
    uint8_t* mtile = cand_mode + (size_t)e0 * M;
    for (int idx = threadIdx.x; idx < n_rec; idx += 256) {                 // (3) flags and mode bytes out
        const int e = m_magic ? (int)__umulhi((unsigned)idx, m_magic) : idx, j = idx - e * M;
        const bool on = s_on[idx] != 0;
        atile[idx] = on ? 1 : 0;
        mtile[idx] = on ? (uint8_t)(j / K) : (uint8_t)EB_VMODE_EMPTY;
    }
}

hipError_t launch_traffic_flow_step(int n_env, int K, float* cand, uint8_t* active, float* timer, int* emitted,
                                    int* sim_step, const float* lane, const float* period, const float* v_max, float dt,
                                    float exit_range, float accel, float lane_len, int light_cycle, uint64_t seed,
                                    uint64_t counter, uint8_t* cand_mode, uint8_t* v_light, hipStream_t s) {
    if (n_env <= 0) return hipSuccess;
    const int M = 12 * K;
    const unsigned m_magic = M <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)M - 1) / (unsigned)M);
    hipLaunchKernelGGL(traffic_flow_step_kernel, dim3((n_env + FS_ENVS - 1) / FS_ENVS), dim3(256), 0, s, n_env, K, cand, active,
                       timer, emitted, sim_step, lane, period, v_max, dt, exit_range, accel, lane_len, light_cycle, seed, counter,
                       cand_mode, v_light, m_magic);
    return hipGetLastError();
}

hipError_t launch_judge_done(int task, int n_env, int D, const float* ego, const float* params, const float* obs,
                             int m_cand, const float* cand, const uint8_t* cand_mode, const float* cand_lw,
                             const uint8_t* v_light, uint8_t* done_code, hipStream_t s) {
    hipLaunchKernelGGL(judge_done_kernel, dim3((n_env + 127) / 128), dim3(128), 0, s, task, n_env, D, ego, params,
                       obs, m_cand, cand, cand_mode, cand_lw, v_light, done_code);
    return hipGetLastError();
}

}  // namespace eb
