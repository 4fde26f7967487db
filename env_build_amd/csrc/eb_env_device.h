// eb_env_device.h — device functions shared by the env-side kernels (eb_env_kernels.hip: one kernel per C-ABI entry;
// eb_env_step.hip: CrossroadEnd2end.step as ONE launch).  Same arithmetic in both files by construction.
#pragma once
#include "eb_device.h"
#include "eb_kernels.h"
#include "../../include/envbuild.h"

#pragma clang fp contract(off)

namespace eb {

struct V4 { float x, y, v, phi; };

// a14: _get_next_ego_state, E2E:269-283
EB_DEV void env_ego_step_row(const float (&st)[6], float steer, float a_x, float (&nx)[6], float (&pr)[4]) {
    const float phi_rad = deg2rad(st[5]);
    float sn, cs;
    sincos_det(phi_rad, sn, cs);
    f_xu_core(st, steer, a_x, TAU10, phi_rad, sn, cs, nx);   // E2E:279
    f_xu_params(st, steer, a_x, pr);
    nx[0] = nx[0] >= 0.0f ? nx[0] : 0.0f;                     // E2E:281
    nx[5] = wrap_deal_with_phi(nx[5]);                        // E2E:282
}

// ---- a16: _construct_veh_vector_short, E2E:340-464 ------------------------------------------------
EB_DEV bool veh_in_range(int task, int m, const V4& v, float ego_x, float ego_y) {   // E2E:393-411
    const float C2 = HALF_CROSS;
    switch (m) {
        case EB_VMODE_DL: return v.x > -C2 - 10.0f && v.y > ego_y - 2.0f;
        case EB_VMODE_DU: return ego_y - 2.0f < v.y && v.y < C2 + 10.0f && v.x < ego_x + 5.0f;
        case EB_VMODE_DR: return v.x < C2 + 10.0f && v.y > ego_y;
        case EB_VMODE_RU: return v.x < C2 + 10.0f && v.y < C2 + 10.0f;
        case EB_VMODE_UR:
            if (task == TASK_STRAIGHT) return v.x < ego_x + 7.0f && ego_y < v.y && v.y < C2 + 10.0f;
            if (task == TASK_RIGHT) return v.x < C2 + 10.0f && v.y < C2;
            return true;
        case EB_VMODE_UD: return __builtin_fmaxf(ego_y - 2.0f, -C2) < v.y && v.y < C2 && ego_x > v.x;
        case EB_VMODE_UL: return -C2 - 10.0f < v.x && v.x < ego_x && v.y < C2;
        case EB_VMODE_LR: return -C2 - 10.0f < v.x && v.x < C2 + 10.0f;
        default: return true;
    }
}

// The same filter as data, for a wave-uniform mode: up to four strict bounds on (v.x, v.y) — which ones exist depends on the
// mode alone (uniform), their values on the ego (per lane).  box_in_range(range_box(task, m, ex, ey), x, y) ==
// veh_in_range(task, m, {x, y, ..}, ex, ey) for every input, NaN and infinities included: an absent bound is a flag, not an infinity.
struct RangeBox { bool has_lox, has_hix, has_loy, has_hiy; float lox, hix, loy, hiy; };
EB_DEV RangeBox range_box(int task, int m, float ego_x, float ego_y) {
    const float C2 = HALF_CROSS;
    RangeBox b = {false, false, false, false, 0.0f, 0.0f, 0.0f, 0.0f};
    switch (m) {
        case EB_VMODE_DL: b.has_lox = true; b.lox = -C2 - 10.0f; b.has_loy = true; b.loy = ego_y - 2.0f; break;
        case EB_VMODE_DU: b.has_loy = true; b.loy = ego_y - 2.0f; b.has_hiy = true; b.hiy = C2 + 10.0f; b.has_hix = true; b.hix = ego_x + 5.0f; break;
        case EB_VMODE_DR: b.has_hix = true; b.hix = C2 + 10.0f; b.has_loy = true; b.loy = ego_y; break;
        case EB_VMODE_RU: b.has_hix = true; b.hix = C2 + 10.0f; b.has_hiy = true; b.hiy = C2 + 10.0f; break;
        case EB_VMODE_UR:
            if (task == TASK_STRAIGHT) { b.has_hix = true; b.hix = ego_x + 7.0f; b.has_loy = true; b.loy = ego_y; b.has_hiy = true; b.hiy = C2 + 10.0f; }
            else if (task == TASK_RIGHT) { b.has_hix = true; b.hix = C2 + 10.0f; b.has_hiy = true; b.hiy = C2; }
            break;
        case EB_VMODE_UD: b.has_loy = true; b.loy = __builtin_fmaxf(ego_y - 2.0f, -C2); b.has_hiy = true; b.hiy = C2; b.has_hix = true; b.hix = ego_x; break;
        case EB_VMODE_UL: b.has_lox = true; b.lox = -C2 - 10.0f; b.has_hix = true; b.hix = ego_x; b.has_hiy = true; b.hiy = C2; break;
        case EB_VMODE_LR: b.has_lox = true; b.lox = -C2 - 10.0f; b.has_hix = true; b.hix = C2 + 10.0f; break;
        default: break;
    }
    return b;
}
EB_DEV bool box_in_range(const RangeBox& b, float x, float y) {   // (bitwise: no short-circuit branches)
    return ((!b.has_lox) | (x > b.lox)) & ((!b.has_hix) | (x < b.hix)) & ((!b.has_loy) | (y > b.loy)) & ((!b.has_hiy) | (y < b.hiy));
}

// sort key of each mode (E2E:414-428): <0 when a sorts before b, 0 when the keys tie
EB_DEV int veh_cmp(int task, int m, const V4& a, const V4& b) {
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

// The same sort keys as data, for a wave-uniform mode: key = (s1 * f1, s2 * f2) with f in {x, y, none}; DESC(f) is ASC(-f)
// (negation is exact and order-reversing, ties and NaN included), a missing key is the constant 0.  key_less == veh_cmp < 0,
// key_before == "sorts before under (key, insertion index)".
struct KeySpec { int f1, f2; float s1, s2; };   // f: 0 = none, 1 = x, 2 = y
EB_DEV KeySpec key_spec(int task, int m) {
    switch (m) {
        case EB_VMODE_DL: return KeySpec{2, 1, 1.0f, -1.0f};
        case EB_VMODE_DU: return KeySpec{2, 0, 1.0f, 1.0f};
        case EB_VMODE_DR: return KeySpec{2, 1, 1.0f, 1.0f};
        case EB_VMODE_RU: return KeySpec{1, 2, 1.0f, -1.0f};
        case EB_VMODE_UR:
            if (task == TASK_STRAIGHT) return KeySpec{2, 0, 1.0f, 1.0f};
            if (task == TASK_RIGHT) return KeySpec{2, 1, 1.0f, -1.0f};
            return KeySpec{0, 0, 1.0f, 1.0f};
        case EB_VMODE_UD: return KeySpec{2, 0, 1.0f, 1.0f};
        case EB_VMODE_UL: return KeySpec{2, 1, 1.0f, 1.0f};
        case EB_VMODE_LR: return KeySpec{1, 0, -1.0f, 1.0f};
        default: return KeySpec{0, 0, 1.0f, 1.0f};
    }
}
EB_DEV float2 key_of(const KeySpec& k, float x, float y) {
    const float a = k.f1 == 1 ? x : k.f1 == 2 ? y : 0.0f, b = k.f2 == 1 ? x : k.f2 == 2 ? y : 0.0f;
    return make_float2(k.s1 < 0.0f ? -a : a, k.s2 < 0.0f ? -b : b);
}
EB_DEV bool key_less(const float2 a, const float2 b) { return (a.x < b.x) | (!(a.x > b.x) & (a.y < b.y)); }
EB_DEV bool key_before(const float2 a, int ia, const float2 b, int ib) {
    const bool lt1 = a.x < b.x, gt1 = a.x > b.x, lt2 = a.y < b.y, gt2 = a.y > b.y;
    return lt1 || (!gt1 && (lt2 || (!gt2 && ia < ib)));
}

EB_DEV V4 veh_fill_value(int m) {   // mode2fillvalue, E2E:439-447
    const float C2 = HALF_CROSS, LW = LANE_W;
    V4 f = {0.0f, 0.0f, 0.0f, 0.0f};
    switch (m) {
        case EB_VMODE_DL: f.x = LW / 2; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_DU: f.x = LW * 1.5f; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_DR: f.x = LW * 2.5f; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_RU: f.x = C2 + 15; f.y = LW * 2.5f; f.phi = 180; break;
        case EB_VMODE_UR: f.x = -LW / 2; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_UD: f.x = -LW * 1.5f; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_UL: f.x = -LW * 2.5f; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_LR: f.x = -(C2 + 20); f.y = -LW * 1.5f; f.phi = 0; break;
        default: break;
    }
    return f;
}

// candidate i of mode m for this env; i == m_cand addresses the virtual red-light car (E2E:386-390)
EB_DEV bool fetch_candidate(int m, int i, int m_cand, const float* cand, const uint8_t* cmode, bool virt, V4& v) {
    if (i < m_cand) {
        if (cmode[i] != m) return false;
        v.x = cand[4 * i]; v.y = cand[4 * i + 1]; v.v = cand[4 * i + 2]; v.phi = cand[4 * i + 3];
        return true;
    }
    if (!virt || (m != EB_VMODE_DL && m != EB_VMODE_DU)) return false;
    v.x = m == EB_VMODE_DL ? LANE_W / 2 : LANE_W * 1.5f;
    v.y = -HALF_CROSS + 2.5f; v.v = 0.0f; v.phi = 90.0f;
    return true;
}

// ---- a17: _judge_done, E2E:200-256 ---------------------------------------------------------------
EB_DEV bool judge_feasible(float x, float y, int task) {   // UTL:73-104
    const float C2 = HALF_CROSS, LW = LANE_W;
    const bool middle = (-C2 < y && y < C2) && (-C2 < x && x < C2);
    if (task == TASK_LEFT)
        return (0.0f < x && x < LW && y <= -C2) || (0.0f < y && y < LW * 3.0f && x < -C2) || middle;
    if (task == TASK_STRAIGHT)
        return (LW < x && x < LW * 2.0f && y <= -C2) || (0.0f < x && x < LW * 3.0f && y >= C2) || middle;
    return (LW * 2.0f < x && x < LW * 3.0f && y <= -C2) || (-LW * 3.0f < y && y < 0.0f && x > C2) || middle;
}

// Traffic.collision_check (TRF:263-295) of one ego against the candidates c = first, first + stride, ...; `row` holds
// the candidates as (x, y, v, phi) float4s, `mrow` their mode ids, `lw` their (l, w) pairs or NULL for (4.8, 2.0)
struct EgoCircles { float x0, y0, x1, y1; };
EB_DEV EgoCircles ego_circles(float x, float y, float phi) {
    float es, ec;
    sincos_det(phi / 180.0f * PI_F, es, ec);
    const float ego_lw = (4.8f - 2.0f) / 2;
    return EgoCircles{x + ec * ego_lw, y + es * ego_lw, x - ec * ego_lw, y - es * ego_lw};
}
EB_DEV bool collision_with(const EgoCircles& E, float x, float y, const float4 v, float vl, float vw) {
    const float EGO_W = 2.0f;
    if (!(__builtin_fabsf(v.x - x) < 10.0f && __builtin_fabsf(v.y - y) < 10.0f)) return false;
    const float s_lw = (vl - vw) / 2;
    float ss, sc;
    sincos_det(v.w / 180.0f * PI_F, ss, sc);
    const float sx0 = v.x + sc * s_lw, sy0 = v.y + ss * s_lw, sx1 = v.x - sc * s_lw, sy1 = v.y - ss * s_lw;
    const float thr = sq((vw + EGO_W) / 2 + 0.5f);
    return sq(E.x0 - sx0) + sq(E.y0 - sy0) < thr || sq(E.x0 - sx1) + sq(E.y0 - sy1) < thr ||
           sq(E.x1 - sx1) + sq(E.y1 - sy1) < thr || sq(E.x1 - sx0) + sq(E.y1 - sy0) < thr;
}

// the rest of _judge_done (E2E:200-256) in two parts: the predicates that need only the ego state (a bit set) and the
// priority chain once the collision flag and delta_y are known — the one-launch step evaluates the parts on different waves
enum { JB_FEASIBLE = 1, JB_STABLE = 2, JB_RED = 4, JB_GOAL = 8, JB_TIMEOUT = 16 };
// _get_ego_dynamics' derived entries (E2E:150-183), one function each — judge_bits and eb_ego_dynamics evaluate the same code
EB_DEV float ego_r_bound(float miu_r, float v_x) { return miu_r * 9.81f / (__builtin_fabsf(v_x) + 1e-8f); }   // E2E:167
EB_DEV float ego_alpha_bound(float miu, float f_z, float c) { return 3.0f * miu * f_z / c; }                  // E2E:164-166
// corner q of the ego's box (E2E:171-176: (+l/2, +w/2), (+l/2, -w/2), (-l/2, +w/2), (-l/2, -w/2)) through
// rotate_and_shift_coordination(cx, cy, 0, -x, -y, -phi) (UTL:152-157); rs / rc = sin / cos of -phi in radians
EB_DEV void ego_corner(int q, float x, float y, float rs, float rc, float& X, float& Y) {
    const float EGO_L = 4.8f, EGO_W = 2.0f;
    const float cx = (q < 2 ? EGO_L : -EGO_L) / 2, cy = ((q & 1) ? -EGO_W : EGO_W) / 2;
    const float tx = cx * rc + cy * rs;
    const float ty = -cx * rs + cy * rc;
    X = tx - (-x); Y = ty - (-y);
}
EB_DEV unsigned judge_bits(int task, float v_x, float r, float x, float y, float phi, float miu_r, bool red_light) {
    // corner points (E2E:171-176, UTL:120-157) through judge_feasible
    float rs, rc;
    sincos_det(-phi * PI_F / 180.0f, rs, rc);
    bool feasible = true;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float X, Y;
        ego_corner(q, x, y, rs, rc, X, Y);
        feasible = feasible && judge_feasible(X, Y, task);
    }
    const float r_bound = ego_r_bound(miu_r, v_x);
    bool goal;
    if (task == TASK_LEFT) goal = x < -HALF_CROSS - 10.0f && 0.0f < y && y < 3.0f * LANE_W;
    else if (task == TASK_RIGHT) goal = x > HALF_CROSS + 10.0f && -3.0f * LANE_W < y && y < 0.0f;
    else goal = y > HALF_CROSS + 10.0f && 0.0f < x && x < 3.0f * LANE_W;
    return (feasible ? JB_FEASIBLE : 0u) | ((-r_bound < r && r < r_bound) ? JB_STABLE : 0u) |
           ((red_light && y > -HALF_CROSS && task != TASK_RIGHT) ? JB_RED : 0u) | (goal ? JB_GOAL : 0u);
}
// JB_TIMEOUT (eb_time_limit): the episode's step count has reached max_episode_steps — gym's TimeLimit wrapper around the
// registered env (README.md:55-59, max_episode_steps = 200): it ends an episode NO reference outcome has ended
EB_DEV uint8_t judge_merge(unsigned bits, bool collision, float delta_y) {
    if (collision) return EB_DONE_COLLISION;
    if (!(bits & JB_FEASIBLE)) return EB_DONE_BREAK_ROAD;
    if (__builtin_fabsf(delta_y) > 15.0f) return EB_DONE_DEVIATE;                 // E2E:224
    if (!(bits & JB_STABLE)) return EB_DONE_STABILITY;
    if (bits & JB_RED) return EB_DONE_RED_LIGHT;
    if (bits & JB_GOAL) return EB_DONE_GOOD;
    if (bits & JB_TIMEOUT) return EB_DONE_TIME_LIMIT;
    return EB_DONE_NOT_YET;
}
EB_DEV uint8_t judge_code(int task, bool collision, float v_x, float r, float x, float y, float phi, float miu_r,
                          float delta_y, bool red_light) {
    return judge_merge(judge_bits(task, v_x, r, x, y, phi, miu_r, red_light), collision, delta_y);
}

// the same for a candidate row staged in LDS as float4s
EB_DEV bool fetch_candidate_lds(int m, int i, int m_cand, const float4* row, const uint8_t* mrow, bool virt, V4& v) {
    if (i < m_cand) {
        if (mrow[i] != m) return false;
        const float4 q = row[i];
        v.x = q.x; v.y = q.y; v.v = q.z; v.phi = q.w;
        return true;
    }
    if (!virt || (m != EB_VMODE_DL && m != EB_VMODE_DU)) return false;
    v.x = m == EB_VMODE_DL ? LANE_W / 2 : LANE_W * 1.5f;
    v.y = -HALF_CROSS + 2.5f; v.v = 0.0f; v.phi = 90.0f;
    return true;
}

EB_DEV int obs_cand_stride4(int m_cand) { return m_cand + ((m_cand & 1) ? 2 : 1); }   // float4s per LDS row, odd: no bank conflicts

// ---- traffic pool re-entry (eb_traffic_respawn) ----------------------------------------------------
EB_DEV uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

EB_DEV float u01(uint64_t seed, uint64_t idx) {   // top 24 bits of splitmix64(seed + GOLDEN * idx) -> [0, 1)
    return (float)(splitmix64(seed + 0x9E3779B97F4A7C15ull * idx) >> 40) * 5.9604644775390625e-8f;
}

// ---- Traffic.init_traffic's conflict rule (TRF:168-192): is a vehicle at (x, y, a) on top of / right in front of the ego? ----
EB_DEV void shift_rotate(float x, float y, float d, float sx, float sy, float rd, float& ox, float& oy, float& od) {   // UTL:145-149
    const float hx = x - sx, hy = y - sy;
    float sn, cs;
    sincos_det(rd * PI_F / 180.0f, sn, cs);
    ox = hx * cs + hy * sn;
    oy = -hx * sn + hy * cs;
    float t = d - rd;
    if (!wrap_bounded(t)) {}
    else if (t > 180.0f) { while (t > 180.0f) t = t - 360.0f; }
    else if (t <= -180.0f) { while (t <= -180.0f) t = t + 360.0f; }
    od = t;
}
EB_DEV bool init_conflict(const float* ego6, float ego_l, float x, float y, float a, float veh_v, float veh_l) {   // TRF:168-192
    float xe, ye, ae, xv, yv, av;
    shift_rotate(x, y, a, ego6[3], ego6[4], ego6[5], xe, ye, ae);
    shift_rotate(0.0f, 0.0f, 0.0f, xe, ye, ae, xv, yv, av);
    return (-5.0f < xe && xe < 1.0f * ego6[0] + ego_l / 2.0f + veh_l / 2.0f + 2.0f && __builtin_fabsf(ye) < 3.0f) ||
           (-5.0f < xv && xv < 1.0f * veh_v + ego_l / 2.0f + veh_l / 2.0f + 2.0f && __builtin_fabsf(yv) < 3.0f);   // TRF:183-184
}


}  // namespace eb
