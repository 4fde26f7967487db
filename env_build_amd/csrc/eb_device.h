// eb_device.h — device-side arithmetic of the hot path, gfx950 only.
//
// Every function evaluates the reference's expression in the reference's op order, one IEEE fp32
// rounding per op (the translation unit is compiled with -ffp-contract=off, and the pragma below
// repeats that for the optimiser).  Citations: DAM = dynamics_and_models.py, E2E = endtoend.py,
// UTL = endtoend_env_utils.py, TRF = traffic.py of the reference.
//
// sin/cos/atan are branch-light Cephes-scheme fp32 kernels (3-term Cody-Waite + minimax
// polynomials, <= 2 ulp): cheaper than ocml's large-argument paths, and — because they use only
// IEEE add/mul/div/fma/rint — reproducible bit-for-bit by the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

#define EB_DEV __device__ __forceinline__

namespace eb {

constexpr float PI_F = 3.14159265358979323846f;       // np.pi -> fp32
constexpr float TWO_PI_F = 6.28318530717958647692f;   // 2*np.pi -> fp32 (DAM:424)
constexpr float LWS = (float)((4.8 - 2.0) / 2.);      // (L-W)/2. (DAM:210)
constexpr float HALF_CROSS = 25.0f;                   // CROSSROAD_SIZE/2
constexpr float LANE_W = 3.75f;
constexpr float EXP_V = 8.0f;
constexpr float TAU10 = (float)(1 / 10.);             // 1/base_frequency (DAM:85-87, 387)

enum { TASK_LEFT = 0, TASK_STRAIGHT = 1, TASK_RIGHT = 2 };
enum { TURN_NONE = 0, TURN_LEFT = 1, TURN_RIGHT = 2 };

EB_DEV float sq(float x) { return x * x; }

// IEEE-exact x / C for the five constant divisors of the path in 3 instructions and with no special cases:
//     x / C  ==  (float)((double)x * (1.0 / (double)C))          v_cvt_f64_f32, v_mul_f64, v_cvt_f32_f64
// The double product carries <= 2^-52 relative error, and a quotient of a 24-bit dividend by one of these
// divisors is never that close to a rounding boundary of the fp32 grid (normal or subnormal), so the single
// rounding of the conversion lands on the correctly rounded quotient; -0, +-inf, NaN and subnormals come out as
// IEEE division gives them.  Verified against x / C for ALL 2^32 dividends and each divisor
// (tests/test_exact_math.py; oracle/envbuild_oracle.c:eb_oracle_check_div_exact runs the same two operations on
// the CPU, where they are IEEE too).  Replaces the ~11-op v_div_scale / v_rcp / v_div_fmas / v_div_fixup
// sequence; an fp32-only 3-op form (q = x*rc, residual, correction) is a little cheaper but wrong for tiny
// non-zero dividends, -0 and +-inf, which then need guards and a second code path.
template <typename C>
EB_DEV float div_const(float x) {
    constexpr double rc = 1.0 / (double)C::value;
    return (float)((double)x * rc);
}
EB_DEV float div_by(float x, double rc) { return (float)((double)x * rc); }   // rc = 1.0 / (double)c for one of the five
struct C180 { static constexpr float value = 180.0f; };
struct CPi { static constexpr float value = 3.14159265358979323846f; };
struct C10 { static constexpr float value = 10.0f; };
struct C26875 { static constexpr float value = 26.875f; };
struct C15625 { static constexpr float value = 15.625f; };

EB_DEV float deg2rad(float d) { return div_const<C180>(d * PI_F); }  // x * np.pi / 180.  (DAM:54)
EB_DEV float rad2deg(float r) { return div_const<CPi>(r * 180.0f); } // x * 180 / np.pi   (DAM:81)

EB_DEV void sincos_det(float x, float& s_out, float& c_out) {
    // k = nearest integer to x / (pi/2); r = x - k*pi/2 by 3-term Cody-Waite with fused steps;
    // minimax polynomials on |r| <= pi/4 in Horner form with fused multiply-adds.  fmaf is correctly
    // rounded on both sides, so oracle/envbuild_oracle.c:eb_sincosf reproduces every bit.
    const float kf = __builtin_rintf(x * 0.636619747f);
    const int k = (int)kf;
    float r = __builtin_fmaf(-kf, 1.5703125f, x);
    r = __builtin_fmaf(-kf, 4.83751296997070312e-4f, r);
    r = __builtin_fmaf(-kf, 7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float s = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float c = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    const float a = (k & 1) ? c : s;
    const float b = (k & 1) ? -s : c;
    s_out = (k & 2) ? -a : a;
    c_out = (k & 2) ? -b : b;
}

EB_DEV float atan_det(float x) {
    float ax = __builtin_fabsf(x);
    float y, t;
    if (ax > 2.414213562373095f) {
        y = 1.5707963267948966f;
        t = -1.0f / ax;
    } else if (ax > 0.4142135623730950f) {
        y = 0.7853981633974483f;
        t = (ax - 1.0f) / (ax + 1.0f);
    } else {
        y = 0.0f;
        t = ax;
    }
    float z = t * t;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032e-1f;
    p = p * z + 1.99777106478e-1f;
    p = p * z - 3.33329491539e-1f;
    y = y + (p * z * t + t);
    return x < 0.0f ? -y : y;
}

// ---- a4: _action_transformation_for_end2end, DAM:128-132 -------------------------------------
EB_DEV void action_transform(float a0, float a1, float& steer, float& a_x) {
    a0 = __builtin_fminf(__builtin_fmaxf(a0, -1.05f), 1.05f);
    a1 = __builtin_fminf(__builtin_fmaxf(a1, -1.05f), 1.05f);
    steer = 0.4f * a0;
    a_x = 2.25f * a1 - 0.75f;
}

// ---- a2: VehicleDynamics.f_xu, DAM:52-83 -------------------------------------------------------
struct VehParams {  // DAM:37-45
    static constexpr float C_f = -155495.0f, C_r = -155495.0f, a = 1.19f, b = 1.46f, mass = 1520.0f,
                           I_z = 2642.0f, miu = 0.8f, g = 9.81f;
    // vehicle_params' F_zf / F_zr (DAM:48: float64, b * mass * g / (a + b) in that order) rounded to fp32 — what E2E:164-166 read
    static constexpr float F_zf = (float)(1.46 * 1520.0 * 9.81 / (1.19 + 1.46)), F_zr = (float)(1.19 * 1520.0 * 9.81 / (1.19 + 1.46));
};

// sn/cs = sin/cos of deg2rad(st[5]) (shared with the reward's ego circle centres, DAM:211)
EB_DEV void f_xu_core(const float (&st)[6], float steer, float a_x, float tau, float phi_rad, float sn,
                      float cs, float (&nx)[6]) {
    using P = VehParams;
    const float v_x = st[0], v_y = st[1], r = st[2], x = st[3], y = st[4];
    const float k1 = P::a * P::C_f - P::b * P::C_r;
    nx[0] = v_x + tau * (a_x + v_y * r);                                                       // DAM:73
    nx[1] = (P::mass * v_y * v_x + tau * k1 * r - tau * P::C_f * steer * v_x - tau * P::mass * sq(v_x) * r) /
            (P::mass * v_x - tau * (P::C_f + P::C_r));                                          // DAM:74-76
    nx[2] = (-P::I_z * r * v_x - tau * k1 * v_y + tau * P::a * P::C_f * steer * v_x) /
            (tau * (sq(P::a) * P::C_f + sq(P::b) * P::C_r) - P::I_z * v_x);                     // DAM:77-78
    nx[3] = x + tau * (v_x * cs - v_y * sn);                                                    // DAM:79
    nx[4] = y + tau * (v_x * sn + v_y * cs);                                                    // DAM:80
    nx[5] = rad2deg(phi_rad + tau * r);                                                         // DAM:81
}

EB_DEV void f_xu_params(const float (&st)[6], float steer, float a_x, float (&pr)[4]) {
    using P = VehParams;
    const float v_x = st[0], v_y = st[1], r = st[2];
    const float F_zf = P::b * P::mass * P::g / (P::a + P::b), F_zr = P::a * P::mass * P::g / (P::a + P::b); // DAM:65
    const float F_xf = a_x < 0 ? P::mass * a_x / 2 : 0.0f;                                      // DAM:66
    const float F_xr = a_x < 0 ? P::mass * a_x / 2 : P::mass * a_x;                             // DAM:67
    pr[2] = __builtin_sqrtf(sq(P::miu * F_zf) - sq(F_xf)) / F_zf;                               // DAM:68
    pr[3] = __builtin_sqrtf(sq(P::miu * F_zr) - sq(F_xr)) / F_zr;                               // DAM:69
    pr[0] = atan_det((v_y + P::a * r) / (v_x + 1e-8f)) - steer;                                 // DAM:70
    pr[1] = atan_det((v_y - P::b * r) / (v_x + 1e-8f));                                         // DAM:71
}

// ---- a5 road-wall terms, DAM:231-295: adds one ego point's 4 training + 4 real terms ------------
template <int TASK>
EB_DEV void road_terms(float px, float py, float& t, float& q) {
    constexpr float LWN = 11.25f;  // LANE_WIDTH*LANE_NUMBER
    constexpr float LW2 = 7.5f;    // 2*LANE_WIDTH
    if (TASK == TASK_LEFT) {       // DAM:233-251
        float c1 = (py < -HALF_CROSS && px < 1.0f) ? sq(px - 1.0f) : 0.0f;
        float c2 = (py < -HALF_CROSS && LANE_W - px < 1.0f) ? sq(LANE_W - px - 1.0f) : 0.0f;
        float c3t = (px < 0.0f && LWN - py < 1.0f) ? sq(LWN - py - 1.0f) : 0.0f;
        float c3r = (px < -HALF_CROSS && LWN - py < 1.0f) ? sq(LWN - py - 1.0f) : 0.0f;
        float c4 = (px < -HALF_CROSS && py - 0.0f < 1.0f) ? sq(py - 0.0f - 1.0f) : 0.0f;
        t += c1; t += c2; t += c3t; t += c4;
        q += c1; q += c2; q += c3r; q += c4;
    } else if (TASK == TASK_STRAIGHT) {  // DAM:252-272
        float c1 = (py < -HALF_CROSS && px - LANE_W < 1.0f) ? sq(px - LANE_W - 1.0f) : 0.0f;
        float c2 = (py < -HALF_CROSS && LW2 - px < 1.0f) ? sq(LW2 - px - 1.0f) : 0.0f;
        float c3 = (py > HALF_CROSS && LWN - px < 1.0f) ? sq(LWN - px - 1.0f) : 0.0f;
        float c4 = (py > HALF_CROSS && px - 0.0f < 1.0f) ? sq(px - 0.0f - 1.0f) : 0.0f;
        t += c1; t += c2; t += c3; t += c4;
        q += c1; q += c2; q += c3; q += c4;
    } else {                             // DAM:273-295
        float c1 = (py < -HALF_CROSS && px - LW2 < 1.0f) ? sq(px - LW2 - 1.0f) : 0.0f;
        float c2 = (py < -HALF_CROSS && LWN - px < 1.0f) ? sq(LWN - px - 1.0f) : 0.0f;
        float c3 = (px > HALF_CROSS && 0.0f - py < 1.0f) ? sq(0.0f - py - 1.0f) : 0.0f;
        float c4 = (px > HALF_CROSS && py - (-LWN) < 1.0f) ? sq(py - (-LWN) - 1.0f) : 0.0f;
        t += c1; t += c2; t += c3; t += c4;
        q += c1; q += c2; q += c3; q += c4;
    }
}

// ---- a5 per-vehicle collision terms, DAM:218-229 -------------------------------------------------
// e = (front x, front y, rear x, rear y) of the ego; v = vehicle (x, y, v, phi); vs/vc = sin/cos of
// deg2rad(v.phi).  t35/t25 get the four terms in the reference's order ff, fr, rf, rr.
EB_DEV void veh2veh_terms(const float4 e, float vx, float vy, float vs, float vc, float (&t35)[4],
                          float (&t25)[4]) {
    const float vfx = vx + LWS * vc, vfy = vy + LWS * vs;  // DAM:221-222
    const float vrx = vx - LWS * vc, vry = vy - LWS * vs;  // DAM:223-224
    const float ex[2] = {e.x, e.z}, ey[2] = {e.y, e.w}, wx[2] = {vfx, vrx}, wy[2] = {vfy, vry};
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float d = __builtin_sqrtf(sq(ex[p] - wx[q]) + sq(ey[p] - wy[q]));  // DAM:227
            const float a = d - 3.5f, b = d - 2.5f;
            t35[2 * p + q] = a < 0.0f ? sq(a) : 0.0f;                                // DAM:228
            t25[2 * p + q] = b < 0.0f ? sq(b) : 0.0f;                                // DAM:229
        }
}

// ---- a10: predict_for_a_mode, DAM:405-427 ---------------------------------------------------------
EB_DEV float4 veh_predict_one(float x, float y, float v, float phi_rad, float sn, float cs, int turn) {
    const bool middle = (x > -HALF_CROSS && x < HALF_CROSS) && (y > -HALF_CROSS && y < HALF_CROSS);  // DAM:409-410
    const float v10 = div_const<C10>(v);
    const float dx = v10 * cs;   // DAM:413
    const float dy = v10 * sn;   // DAM:414
    float dphi = 0.0f;
    if (turn == TURN_LEFT) dphi = middle ? div_const<C10>(div_const<C26875>(v)) : 0.0f;        // DAM:417
    else if (turn == TURN_RIGHT) dphi = middle ? div_const<C10>(-div_const<C15625>(v)) : 0.0f; // DAM:419
    float nphi = phi_rad + dphi;                                                 // DAM:423
    if (nphi > PI_F) nphi = nphi - TWO_PI_F;                                     // DAM:424
    if (nphi <= -PI_F) nphi = nphi + TWO_PI_F;                                   // DAM:425
    return make_float4(x + dx, y + dy, v, rad2deg(nphi));                        // DAM:422-427
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte access, 4-byte aligned

// predict_for_a_mode (DAM:405-427) on one record: the operations of eb_device.h:predict_record with the slot's turn
// constants passed in (so the same bits).  Scalar fp32 throughout: v_pk_*_f32 issues at half the rate of its scalar
// twins on this chip, so pairing the two polynomial chains saved nothing and cost four register moves per record
// plus two constant loads (the pair forms take no literals) — 9 full-rate instructions now instead of 4 paired + 5.
// slot turn constants (predict_for_a_mode, DAM:416-421): 1 / turn radius in double (for the exact division), the
// sign of the heading rate, and whether the slot turns at all
// (the sign rides on the reciprocal: rounding to nearest is symmetric, so fl(v * -rc) == -fl(v * rc) bit for bit)
struct TurnC { double rc; float enabled; };
EB_DEV TurnC turn_consts(int t) {
    return t == TURN_LEFT ? TurnC{1.0 / 26.875, 1.0f} : t == TURN_RIGHT ? TurnC{-1.0 / 15.625, 1.0f} : TurnC{1.0, 0.0f};
}
// The two polynomial constants of sincos_det that are added to a product of two registers: as literals each costs a
// v_mov per use (a VOP3 fma takes no literal on gfx950); a record wave keeps them in two VGPRs for its whole loop.
struct SinCosK { float s2, c2; };
EB_DEV SinCosK sincos_consts() {
    SinCosK k{8.3321608736e-3f, -1.388731625493765e-3f};
    asm volatile("" : "+v"(k.s2), "+v"(k.c2));        // opaque to the compiler: stays in registers instead of being rematerialised
    return k;
}
// sincos_det (eb_device.h) with those two constants from registers — same operations, same bits
EB_DEV void sincos_det_k(float x, const SinCosK K, float& s_out, float& c_out) {
    const float kf = __builtin_rintf(x * 0.636619747f);
    const int k = (int)kf;
    float r = __builtin_fmaf(-kf, 1.5703125f, x);
    r = __builtin_fmaf(-kf, 4.83751296997070312e-4f, r);
    r = __builtin_fmaf(-kf, 7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(-1.9515295891e-4f, z, K.s2);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float s = __builtin_fmaf(r * z, ps, r);
    float pc = __builtin_fmaf(2.443315711809948e-5f, z, K.c2);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float c = __builtin_fmaf(z * z, pc, __builtin_fmaf(-0.5f, z, 1.0f));
    const float a = (k & 1) ? c : s;
    const float b = (k & 1) ? -s : c;
    s_out = (k & 2) ? -a : a;
    c_out = (k & 2) ? -b : b;
}

// sn_out / cs_out: sin / cos of the record's CURRENT heading, deg2rad(rec.w) — exactly sincos_det(deg2rad(phi)), the
// pair the collision terms need too (DAM:221-224)
// kept_out: the heading rate was the literal zero of DAM:416-421 (a slot that does not turn, or a record outside the junction box) —
// the next heading is the current one up to the radian round trip's rounding (and a full turn of the wrap, DAM:424-425)
template <typename ST = float>
EB_DEV f4u predict_record_tc(const f4u rec, const TurnC tc, const SinCosK K, float& sn_out, float& cs_out, bool& kept_out) {
    const float v = rec.z;
    const float v10 = div_const<C10>(v);                                     // DAM:413
    const float phi_rad = div_const<C180>(rec.w * PI_F);                     // DAM:407
    float sn, cs;
    sincos_det_k(phi_rad, K, sn, cs);
    sn_out = sn; cs_out = cs;
    const float nx_ = rec.x + v10 * cs, ny_ = rec.y + v10 * sn;              // DAM:413-414, 422
    const bool middle = (rec.x > -HALF_CROSS && rec.x < HALF_CROSS) && (rec.y > -HALF_CROSS && rec.y < HALF_CROSS);   // DAM:409-410
    const float u = div_by(v, tc.rc);                                        // +-(v / radius), DAM:417, 419
    const float u10 = div_const<C10>(u);
    const bool turning = middle && tc.enabled != 0.0f;
    kept_out = !turning;
    const float dphi = turning ? u10 : 0.0f;                                 // DAM:416-421
    float nphi = phi_rad + dphi;                                             // DAM:423
    if (nphi > PI_F) nphi = nphi - TWO_PI_F;                                 // DAM:424
    if (nphi <= -PI_F) nphi = nphi + TWO_PI_F;                               // DAM:425
    const float nphi_deg = div_const<CPi>(nphi * 180.0f);                    // DAM:426
    return f4u{nx_, ny_, v, nphi_deg};                                       // DAM:422-427
}
template <typename ST = float>
EB_DEV f4u predict_record_tc(const f4u rec, const TurnC tc, const SinCosK K, float& sn_out, float& cs_out) {
    bool kept;
    return predict_record_tc<ST>(rec, tc, K, sn_out, cs_out, kept);
}

// ---- a8: tracking error pieces, DAM:577-580, 736-760 ----------------------------------------------
EB_DEV float deal_with_phi_diff(float d) {
    if (d > 180.0f) d = d - 360.0f;
    if (d < -180.0f) d = d + 360.0f;
    return d;
}

template <int TASK>
EB_DEV float two2one(float ex, float ey, float rx, float ry) {
    if (TASK == TASK_LEFT) {
        float delta = __builtin_sqrtf(sq(ex - (-HALF_CROSS)) + sq(ey - (-HALF_CROSS))) -
                      __builtin_sqrtf(sq(rx - (-HALF_CROSS)) + sq(ry - (-HALF_CROSS)));
        if (ey < -HALF_CROSS) delta = ex - rx;
        if (ex < -HALF_CROSS) delta = ey - ry;
        return -delta;
    } else if (TASK == TASK_STRAIGHT) {
        return -(ex - rx);
    } else {
        float delta = -(__builtin_sqrtf(sq(ex - HALF_CROSS) + sq(ey - (-HALF_CROSS))) -
                        __builtin_sqrtf(sq(rx - HALF_CROSS) + sq(ry - (-HALF_CROSS))));
        if (ey < -HALF_CROSS) delta = ex - rx;
        if (ex > HALF_CROSS) delta = -(ey - ry);
        return -delta;
    }
}

EB_DEV int clamp_index(int i, int len) {  // indexs2points, DAM:727-728
    i = i < 0 ? 0 : i;
    return i >= len ? len - 1 : i;
}

// device view of the path tables owned by a handle
struct PathTables {
    const float* x[3];      // full resolution, lens[k] floats
    const float* y[3];
    const float* phi[3];
    const float2* red[3];   // stride-10 (x, y) pairs, red_len[k] entries (DAM:704-706)
    const float* phi10[3];  // stride-10 headings, indexed like red: the heading of the point the search returns
    int len[3];
    int red_len[3];
    int n_paths;
    int red_off[3];         // first entry of path k in the concatenated stride-10 table
    uint8_t turn[64];       // TURN_* per vehicle slot (filled by eb_set_veh_modes)
    // closest-point cell grid (one geometry for all paths of the handle): cell (ix, iy) of path k at
    // cells[(k * gny + iy) * gnx + ix] = lo | hi << 16 — the stride-10 index range that holds the
    // closest table point of EVERY position inside the 0.5 m cell (built by eb_set_paths)
    const uint32_t* cells;
    float gx0, gy0;         // lower-left corner of the grid
    int gnx, gny;
    const float* rad;       // 3 x 32 block radii of the pruned full search (positions outside the grid), closest_reduced_index
    // the same for positions off that grid (an ego that has left the road, or finished and drives on): three coarser levels — 4 m cells
    // out to 250 m around the paths, 32 m cells out to 2 km, 256 m cells out to 16 km; every level's ranges are narrowed by witnesses
    // (eb_capi.hip:build_cell_grid).  A cell's word: one or two short ranges (lo | n - 1 << 9 | lo2 << 15 | n2 - 1 << 24 | has2 << 30),
    // or — a range that would still be long (> 16 entries: abreast of a long straight, far out) — bit 31 | lo | hi << 9: the pruned
    // search over the blocks of [lo, hi]; 0xffffffff = "not on this level" (the next level, then the pruned full search)
    struct Coarse {
        const uint32_t* cells;
        float x0, y0, inv;   // lower-left corner, 1 / cell size (a power of two: exact)
        int nx, ny;
    } coarse[3];
};
constexpr float CELL_INV = 2.0f;   // 1 / cell size (0.5 m): exact in fp32

// path used by row i: ref_idx[i] when given, else path_id; out of range -> -1 (zeros, DAM:342, 352)
EB_DEV int row_path(const PathTables& pt, const int* ref_idx, int path_id, int i) {
    const int p = ref_idx ? ref_idx[i] : path_id;
    return (p >= 0 && p < pt.n_paths) ? p : -1;
}

// The index ranges a coarse level names for (px, py) on path p: [lo, hi] and — a cell on the path's medial axis, as close to one stretch
// as to another — a second one [lo2, hi2] (lo2 > hi; hi2 < lo2: none).  -> 1: scan them; 2: one LONG range [lo, hi] (abreast of a long
// straight, far out: the pruned search over the blocks of that range); 0: off every level or NaN (the pruned search over the table).
EB_DEV int coarse_cell_ranges(const PathTables& pt, int p, float px, float py, int& lo, int& hi, int& lo2, int& hi2) {
#pragma unroll 1
    for (int l = 0; l < 3; ++l) {
        const PathTables::Coarse& g = pt.coarse[l];
        const float fx = (px - g.x0) * g.inv, fy = (py - g.y0) * g.inv;
        if (!(fx >= 0.0f && fx < (float)g.nx && fy >= 0.0f && fy < (float)g.ny)) continue;
        const unsigned c = g.cells[(p * g.ny + (int)fy) * g.nx + (int)fx];
        if (c == 0xffffffffu) continue;   // not on this level (it would otherwise decode as the long range [511, 511])
        lo = (int)(c & 0x1ffu);
        if (c >> 31) { hi = (int)((c >> 9) & 0x1ffu); return 2; }
        hi = lo + (int)((c >> 9) & 0x3fu);
        lo2 = (int)((c >> 15) & 0x1ffu); hi2 = (c >> 30) & 1u ? lo2 + (int)((c >> 24) & 0x3fu) : lo2 - 1;
        return 1;
    }
    return 0;
}

// ---- closest point inside the index range a grid cell names (eb_capi.hip:build_cell_grid), one lane per env ----
// The reference's argmin (DAM:712-714) restricted to entries [lo, hi] of a path's stride-10 table: index order, the reference's fp32
// expression, a strict '<' (first minimum) -> the index and the table point itself (x, y, heading).  xy: the path's (x, y) pairs,
// ph: its headings, both readable 4 entries past the path's end.
// The first PRE groups of four entries are fetched in ONE round trip, whatever the range's length: this chain of dependent reads —
// cell word, then the range — is the critical path of a wave that has one lane per env, and every round trip queues behind the
// record streams of the same CU.  (When this was written a corridor range held 6-10 entries, two or three groups, and the groups were
// fetched one per loop trip; since the ranges are narrowed by witnesses — eb_capi.hip:build_cell_grid — it holds 2-4: one group.)
// Same comparisons in the same order: same index, same bits.
template <int PRE = 3>
EB_DEV int closest_in_range(const float* xy, const float* ph, int lo, int hi, float px, float py, float& rx, float& ry, float& rphi) {
    typedef float f4x __attribute__((ext_vector_type(4), aligned(4)));
    float best = __builtin_inff();
    int bi = 0;
    rx = xy[0]; ry = xy[1]; rphi = ph[0];   // index 0 unless a distance compares below +inf, as in the full scan
    if constexpr (PRE == 0) {               // a group per loop trip (the tape / gated kernels: no registers to spare)
        for (int r = lo; r <= hi; r += 4) {
            const f4x a = *reinterpret_cast<const f4x*>(xy + 2 * r), b = *reinterpret_cast<const f4x*>(xy + 2 * r + 4);
            const f4x hh = *reinterpret_cast<const f4x*>(ph + r);
            const float d0 = sq(px - a.x) + sq(py - a.y), d1 = sq(px - a.z) + sq(py - a.w);   // DAM:712
            const float d2 = sq(px - b.x) + sq(py - b.y), d3 = sq(px - b.z) + sq(py - b.w);
            if (d0 < best) { best = d0; bi = r; rx = a.x; ry = a.y; rphi = hh.x; }                  // first minimum, DAM:714
            if (r + 1 <= hi && d1 < best) { best = d1; bi = r + 1; rx = a.z; ry = a.w; rphi = hh.y; }
            if (r + 2 <= hi && d2 < best) { best = d2; bi = r + 2; rx = b.x; ry = b.y; rphi = hh.z; }
            if (r + 3 <= hi && d3 < best) { best = d3; bi = r + 3; rx = b.z; ry = b.w; rphi = hh.w; }
        }
        return bi;
    }
    constexpr int NPRE = PRE > 0 ? PRE : 1;
    f4x q01[NPRE], q23[NPRE], h[NPRE];
#pragma unroll
    for (int g = 0; g < PRE; ++g) {
        const int r = min(lo + 4 * g, hi);   // a group past the range re-reads the range's last entries (in bounds); nothing of it is compared
        q01[g] = *reinterpret_cast<const f4x*>(xy + 2 * r);
        q23[g] = *reinterpret_cast<const f4x*>(xy + 2 * r + 4);
        h[g] = *reinterpret_cast<const f4x*>(ph + r);
    }
    auto group = [&](int r, const f4x a, const f4x b, const f4x hh) {
        const float d0 = sq(px - a.x) + sq(py - a.y), d1 = sq(px - a.z) + sq(py - a.w);   // DAM:712
        const float d2 = sq(px - b.x) + sq(py - b.y), d3 = sq(px - b.z) + sq(py - b.w);
        if (r <= hi && d0 < best) { best = d0; bi = r; rx = a.x; ry = a.y; rphi = hh.x; }   // first minimum, DAM:714
        if (r + 1 <= hi && d1 < best) { best = d1; bi = r + 1; rx = a.z; ry = a.w; rphi = hh.y; }
        if (r + 2 <= hi && d2 < best) { best = d2; bi = r + 2; rx = b.x; ry = b.y; rphi = hh.z; }
        if (r + 3 <= hi && d3 < best) { best = d3; bi = r + 3; rx = b.z; ry = b.w; rphi = hh.w; }
    };
#pragma unroll
    for (int g = 0; g < PRE; ++g) group(lo + 4 * g, q01[g], q23[g], h[g]);
    for (int r = lo + 4 * PRE; r <= hi; r += 4)   // a long range (off the corridor): one group per trip, as before
        group(r, *reinterpret_cast<const f4x*>(xy + 2 * r), *reinterpret_cast<const f4x*>(xy + 2 * r + 4), *reinterpret_cast<const f4x*>(ph + r));
    return bi;
}

// the same over one or two ranges in index order (the coarse levels, coarse_cell_ranges): a group of four entries per trip
EB_DEV int closest_in_ranges(const float* xy, const float* ph, int lo, int hi, int lo2, int hi2, float px, float py, float& rx, float& ry, float& rphi) {
    typedef float f4x __attribute__((ext_vector_type(4), aligned(4)));
    float best = __builtin_inff();
    int bi = 0;
    rx = xy[0]; ry = xy[1]; rphi = ph[0];
#pragma unroll 1
    for (int part = 0; part < 2; ++part) {
        const int a = part ? lo2 : lo, b = part ? hi2 : hi;
        for (int r = a; r <= b; r += 4) {
            const f4x q01 = *reinterpret_cast<const f4x*>(xy + 2 * r), q23 = *reinterpret_cast<const f4x*>(xy + 2 * r + 4);
            const f4x hh = *reinterpret_cast<const f4x*>(ph + r);
            const float d0 = sq(px - q01.x) + sq(py - q01.y), d1 = sq(px - q01.z) + sq(py - q01.w);   // DAM:712
            const float d2 = sq(px - q23.x) + sq(py - q23.y), d3 = sq(px - q23.z) + sq(py - q23.w);
            if (d0 < best) { best = d0; bi = r; rx = q01.x; ry = q01.y; rphi = hh.x; }                // first minimum, DAM:714
            if (r + 1 <= b && d1 < best) { best = d1; bi = r + 1; rx = q01.z; ry = q01.w; rphi = hh.y; }
            if (r + 2 <= b && d2 < best) { best = d2; bi = r + 2; rx = q23.x; ry = q23.y; rphi = hh.z; }
            if (r + 3 <= b && d3 < best) { best = d3; bi = r + 3; rx = q23.z; ry = q23.w; rphi = hh.w; }
        }
    }
    return bi;
}

// ---- closest point, one lane per env ---------------------------------------------------------------
// EXACTLY the index the reference's full scan + argmin returns (DAM:702-715) while visiting ~1/5 of
// the table.  The stride-10 table is cut into blocks of 16 consecutive points; for block b the host
// stored a radius R_b >= max_r |P_r - c_b| around the block's centre point c_b = P_min(16b+8, n-1).
//   1. M = min_b |p - c_b|  (every c_b is itself a table point, so the true minimum D* <= M);
//   2. block b can hold a point with |p - P_r| <= M only if |p - c_b| - R_b <= M; blocks failing
//      |p - c_b| <= M + R_b + 0.01 are skipped — the 0.01 m slack is ~100x the fp32 rounding of
//      these distances, so every skipped point's fp32 dist^2 is strictly above the winner's;
//   3. the surviving blocks are scanned in index order with the reference's fp32 expression and a
//      strict '<' (first minimum).
// NaN / inf coordinates end with index 0, as the full scan does.
// The table lives in global memory (L2): loads are issued in groups so that their latencies overlap.
// G loads in flight per round trip: a lane out here stalls its whole wave for the length of this chain — with groups of four 8 + 8 trips
// over the centres and 4 per surviving block, ~17 us at a loaded L2; eight per trip (the env step) halve that; the rollout kernels,
// whose register budget is the records', stay with four
template <int G = 4>
EB_DEV int closest_reduced_index(const float2* red, const float* rad, int n, float px, float py, int r_first = 0, int r_last = 1 << 30) {
    // [r_first, r_last]: a range known to hold the result (a coarse cell's long range) — only its blocks are looked at
    const int nb = (n + 15) >> 4;                 // <= 32 blocks (eb_set_paths limits a path to 512 table points)
    const int bf = min(max(r_first, 0) >> 4, nb - 1), bl = min(r_last >> 4, nb - 1);
    float m2 = __builtin_inff();
    for (int b0 = bf; b0 <= bl; b0 += G) {
        float2 q[G];
#pragma unroll
        for (int u = 0; u < G; ++u) q[u] = red[min(16 * min(b0 + u, bl) + 8, n - 1)];   // past the end: the last centre again
#pragma unroll
        for (int u = 0; u < G; ++u) m2 = __builtin_fminf(m2, sq(px - q[u].x) + sq(py - q[u].y));
    }
    const float m = __builtin_amdgcn_sqrtf(m2);   // approximate is enough: only feeds the slack test
    unsigned cand = 0u;
    for (int b0 = bf; b0 <= bl; b0 += G) {
        float2 q[G];
        float rr[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int b = min(b0 + u, bl);
            q[u] = red[min(16 * b + 8, n - 1)];
            rr[u] = rad[b];
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const float d2 = sq(px - q[u].x) + sq(py - q[u].y);
            const float thr = m + rr[u] + 0.01f;
            cand |= (b0 + u <= bl && d2 <= thr * thr) ? (1u << (b0 + u)) : 0u;
        }
    }
    float best = __builtin_inff();
    int bi = 0;
    while (cand) {
        const int b = __builtin_ctz(cand);
        cand &= cand - 1u;
        const int r1 = min(16 * b + 16, n);
        for (int r0 = 16 * b; r0 < r1; r0 += G) {
            float2 t[G];
#pragma unroll
            for (int u = 0; u < G; ++u) t[u] = red[min(r0 + u, n - 1)];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const float d = sq(px - t[u].x) + sq(py - t[u].y);     // DAM:712
                if (r0 + u < r1 && d < best) { best = d; bi = r0 + u; } // first minimum, DAM:714
            }
        }
    }
    return bi;
}

// The reference wraps angles with `while` loops (UTL:134-139, 232-237).  On a GPU a loop that never ends takes the device with
// it, and it never ends for +-inf (and takes > 10^4 turns beyond +-3.6e6 degrees, where a diverged state can land): such a
// value is returned as it is — both backends, same rule (oracle: EB_WRAP_MAX_DEG).  NaN fails every loop test by itself.
constexpr float WRAP_MAX_DEG = 3.6e6f;
EB_DEV bool wrap_bounded(float d) { return __builtin_fabsf(d) <= WRAP_MAX_DEG; }
EB_DEV bool wrap_bounded(double d) { return __builtin_fabs(d) <= (double)WRAP_MAX_DEG; }
EB_DEV float wrap_deal_with_phi(float phi) {  // UTL:232-237
    if (!wrap_bounded(phi)) return phi;
    while (phi > 180.0f) phi -= 360.0f;
    while (phi <= -180.0f) phi += 360.0f;
    return phi;
}

// compute_rewards (DAM:186-320) for env i with the scaled action (steer, a_x); same per-vehicle association as the
// fused rollout kernel
template <int TASK>
EB_DEV void rewards_env(int i, int n_env, int D, int n_future, int NV, const float* __restrict__ obs, float steer,
                        float a_x, float* __restrict__ out5, float* __restrict__ d16) {
    const float* o = obs + (size_t)D * i;
    const float* veh = o + 6 + 3 * (n_future + 1);
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

}  // namespace eb
