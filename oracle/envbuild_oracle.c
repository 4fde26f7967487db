/*
 * envbuild_oracle.c — TEST INFRASTRUCTURE.  CPU restatement (plain C, fp32, no FMA contraction)
 * of the reference's hot path, exporting the C-ABI of include/envbuild.h with HOST pointers.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (env_build_amd/) never does and fails loudly when its HIP library is missing.
 *
 * Every function cites the reference lines it follows (DAM = dynamics_and_models.py,
 * E2E = endtoend.py, UTL = endtoend_env_utils.py, TRF = traffic.py under /root/reference).
 *
 * Pinning status: the reference ships no golden vectors or asserting tests, and its
 * arithmetic executor (TensorFlow) plus `bezier` are absent from this image.  The oracle is pinned
 * against fixtures under tests/golden/ produced by running the reference's OWN Python files,
 * unmodified, over NumPy-fp32 stand-ins for the tf.* symbols (oracle/gen_golden.py).  At the
 * TF-kernel boundary itself (Eigen's sin/cos/atan) parity is UNPINNED: TF's, NumPy's and this
 * file's transcendental kernels each carry <= 2 ulp error, which is why the fixture comparison
 * uses rtol 1e-5 (+ a stated atol) rather than bit equality.
 *
 * Deliberate choices (all documented in DESIGN.md):
 *  - sin/cos/atan are the deterministic Cephes-style fp32 kernels below (only IEEE + - * / fma
 *    and rint), NOT libm's, so the HIP kernels can reproduce them bit-for-bit.
 *  - python-float constants are rounded to fp32 at the op where they meet a tensor, in the
 *    reference's evaluation order.
 */
#define _POSIX_C_SOURCE 200809L
#include "../include/envbuild.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(__FP_FAST_FMAF) && !defined(EB_ALLOW_FMA)
/* compiled with -ffp-contract=off in oracle/Makefile; this is only a reminder */
#endif

static __thread char g_err[256];
static int fail(int code, const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}
const char* eb_last_error(void) { return g_err; }
int eb_abi_version(void) { return EB_ABI_VERSION; }
const char* eb_backend(void) { return "oracle"; }

/* ------------------------------------------------------------------------------------------ */
/* constants: python doubles rounded to fp32 where they meet a tensor                          */
/* ------------------------------------------------------------------------------------------ */
#define PI_D 3.141592653589793
static const float PI_F = (float)PI_D;              /* np.pi -> fp32 */
static const float TWO_PI_F = (float)(2 * PI_D);    /* 2 * np.pi (python double) -> fp32, DAM:424 */
static const float LWS = (float)((4.8 - 2.0) / 2.); /* (L - W) / 2., DAM:210, UTL:14 */
static const float HALF_CROSS = 25.0f;              /* CROSSROAD_SIZE/2, UTL:17 */
static const float LANE_W = 3.75f;                  /* UTL:15 */
static const float EXP_V = 8.0f;                    /* EXPECTED_V, UTL:18 */

/* ------------------------------------------------------------------------------------------ */
/* deterministic fp32 transcendental kernels (Cephes single-precision scheme)                  */
/* ------------------------------------------------------------------------------------------ */
static void eb_sincosf(float x, float* s_out, float* c_out) {
    /* k = nearest integer to x / (pi/2); r = x - k*pi/2 by 3-term Cody-Waite with fused steps;
     * minimax polynomials on |r| <= pi/4, Horner with explicit fmaf (correctly rounded with or
     * without hardware FMA, so the HIP kernels' v_fma_f32 sequence gives the same bits). */
    float kf = nearbyintf(x * 0.636619747f);
    int k = (int)kf;
    float r = fmaf(-kf, 1.5703125f, x);
    r = fmaf(-kf, 4.83751296997070312e-4f, r);
    r = fmaf(-kf, 7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float s = fmaf(r * z, ps, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    switch (k & 3) {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}

static float eb_atanf(float x) {
    float ax = fabsf(x);
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

static inline float sq(float x) { return x * x; }
static inline float deg2rad(float d) { return d * PI_F / 180.0f; } /* x * np.pi / 180., DAM:54 */
static inline float rad2deg(float r) { return r * 180.0f / PI_F; } /* x * 180 / np.pi, DAM:81 */

/* ------------------------------------------------------------------------------------------ */
/* handle                                                                                      */
/* ------------------------------------------------------------------------------------------ */
struct eb_handle_s {
    eb_config cfg;
    int n_paths;
    int lens[EB_MAX_PATHS];
    float* px[EB_MAX_PATHS];
    float* py[EB_MAX_PATHS];
    float* pphi[EB_MAX_PATHS];
    uint8_t vmode[EB_MAX_VEH]; /* EB_VMODE_* per slot */
    uint8_t turn[EB_MAX_VEH];  /* 0 none, 1 left-turn, 2 right-turn: DAM:416-421 */
    int modes_set;
};

enum { TURN_NONE = 0, TURN_LEFT = 1, TURN_RIGHT = 2 };
static int turn_of_mode(int m) {
    switch (m) {
        case EB_VMODE_DL: case EB_VMODE_RD: case EB_VMODE_UR: case EB_VMODE_LU: return TURN_LEFT;  /* DAM:416 */
        case EB_VMODE_DR: case EB_VMODE_RU: case EB_VMODE_UL: case EB_VMODE_LD: return TURN_RIGHT; /* DAM:418 */
        default: return TURN_NONE;
    }
}

static int obs_dim(const eb_config* c) { return 6 + 3 * (c->n_future + 1) + 4 * c->n_veh; }

int eb_create(const eb_config* cfg, eb_handle* out) {
    if (!cfg || !out) return fail(EB_EINVAL, "eb_create: null argument");
    if (cfg->abi_version != EB_ABI_VERSION) return fail(EB_EINVAL, "eb_create: ABI version mismatch");
    if (cfg->task < 0 || cfg->task > 2) return fail(EB_EINVAL, "eb_create: task must be left/straight/right");
    if (cfg->n_veh < 1 || cfg->n_veh > EB_MAX_VEH) return fail(EB_EINVAL, "eb_create: n_veh out of range");
    if (cfg->n_future < 0 || cfg->n_future > 64) return fail(EB_EINVAL, "eb_create: n_future out of range");
    if (cfg->mode != EB_MODE_TRAINING && cfg->mode != EB_MODE_SELECTING)
        return fail(EB_EINVAL, "eb_create: bad mode");
    eb_handle h = (eb_handle)calloc(1, sizeof *h);
    if (!h) return fail(EB_ENOMEM, "eb_create: out of memory");
    h->cfg = *cfg;
    *out = h;
    return EB_OK;
}

int eb_destroy(eb_handle h) {
    if (!h) return EB_OK;
    for (int k = 0; k < EB_MAX_PATHS; ++k) { free(h->px[k]); free(h->py[k]); free(h->pphi[k]); }
    free(h);
    return EB_OK;
}

int eb_sync(eb_handle h) { (void)h; return EB_OK; }

int eb_set_paths(eb_handle h, const float* xs, const float* ys, const float* phis,
                 const int32_t* lens, int32_t n_paths) {
    if (!h || !xs || !ys || !phis || !lens) return fail(EB_EINVAL, "eb_set_paths: null argument");
    if (n_paths < 1 || n_paths > EB_MAX_PATHS) return fail(EB_EINVAL, "eb_set_paths: n_paths out of range");
    size_t off = 0;
    for (int k = 0; k < n_paths; ++k) {
        if (lens[k] < 3) return fail(EB_EINVAL, "eb_set_paths: path too short");
        size_t nb = (size_t)lens[k] * sizeof(float);
        free(h->px[k]); free(h->py[k]); free(h->pphi[k]);
        h->px[k] = (float*)malloc(nb); h->py[k] = (float*)malloc(nb); h->pphi[k] = (float*)malloc(nb);
        if (!h->px[k] || !h->py[k] || !h->pphi[k]) return fail(EB_ENOMEM, "eb_set_paths: out of memory");
        memcpy(h->px[k], xs + off, nb); memcpy(h->py[k], ys + off, nb); memcpy(h->pphi[k], phis + off, nb);
        h->lens[k] = lens[k];
        off += (size_t)lens[k];
    }
    h->n_paths = n_paths;
    return EB_OK;
}

int eb_set_veh_modes(eb_handle h, const uint8_t* mode_id, int32_t n) {
    if (!h || !mode_id) return fail(EB_EINVAL, "eb_set_veh_modes: null argument");
    if (n != h->cfg.n_veh) return fail(EB_EINVAL, "eb_set_veh_modes: n != n_veh");
    for (int j = 0; j < n; ++j)
        if (mode_id[j] >= EB_VMODE_COUNT) return fail(EB_EINVAL, "eb_set_veh_modes: bad mode id");
    for (int j = 0; j < n; ++j) {
        h->vmode[j] = mode_id[j];
        h->turn[j] = (uint8_t)turn_of_mode(mode_id[j]);
    }
    h->modes_set = 1;
    return EB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a2: VehicleDynamics.f_xu, DAM:52-83                                                         */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float C_f, C_r, a, b, mass, I_z, miu, g;
} veh_params_t;
static const veh_params_t VP = {-155495.0f, -155495.0f, 1.19f, 1.46f, 1520.0f, 2642.0f, 0.8f, 9.81f}; /* DAM:37-45 */

static void f_xu_row(const float* st, const float* ac, float tau, float* nx, float* pr) {
    float v_x = st[0], v_y = st[1], r = st[2], x = st[3], y = st[4], phi = st[5]; /* DAM:53 */
    phi = deg2rad(phi);                                                          /* DAM:54 */
    float steer = ac[0], a_x = ac[1];                                            /* DAM:55 */
    const float C_f = VP.C_f, C_r = VP.C_r, a = VP.a, b = VP.b, mass = VP.mass, I_z = VP.I_z,
                miu = VP.miu, g = VP.g;                                          /* DAM:56-63 */
    if (pr) {
        float F_zf = b * mass * g / (a + b), F_zr = a * mass * g / (a + b);      /* DAM:65 */
        float F_xf = a_x < 0 ? mass * a_x / 2 : 0.0f;                            /* DAM:66 */
        float F_xr = a_x < 0 ? mass * a_x / 2 : mass * a_x;                      /* DAM:67 */
        float miu_f = sqrtf(sq(miu * F_zf) - sq(F_xf)) / F_zf;                   /* DAM:68 */
        float miu_r = sqrtf(sq(miu * F_zr) - sq(F_xr)) / F_zr;                   /* DAM:69 */
        float alpha_f = eb_atanf((v_y + a * r) / (v_x + 1e-8f)) - steer;         /* DAM:70 */
        float alpha_r = eb_atanf((v_y - b * r) / (v_x + 1e-8f));                 /* DAM:71 */
        pr[0] = alpha_f; pr[1] = alpha_r; pr[2] = miu_f; pr[3] = miu_r;          /* DAM:83 */
    }
    float sn, cs;
    eb_sincosf(phi, &sn, &cs);
    float k1 = a * C_f - b * C_r;
    nx[0] = v_x + tau * (a_x + v_y * r);                                          /* DAM:73 */
    nx[1] = (mass * v_y * v_x + tau * k1 * r - tau * C_f * steer * v_x - tau * mass * sq(v_x) * r) /
            (mass * v_x - tau * (C_f + C_r));                                     /* DAM:74-76 */
    nx[2] = (-I_z * r * v_x - tau * k1 * v_y + tau * a * C_f * steer * v_x) /
            (tau * (sq(a) * C_f + sq(b) * C_r) - I_z * v_x);                      /* DAM:77-78 */
    nx[3] = x + tau * (v_x * cs - v_y * sn);                                      /* DAM:79 */
    nx[4] = y + tau * (v_x * sn + v_y * cs);                                      /* DAM:80 */
    nx[5] = rad2deg(phi + tau * r);                                               /* DAM:81 */
}

int eb_f_xu(eb_handle h, int32_t n, const float* states, const float* actions, float tau,
            float* next_states, float* params, void* stream) {
    (void)stream;
    if (!h || n < 0 || !states || !actions || !next_states) return fail(EB_EINVAL, "eb_f_xu: bad argument");
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        f_xu_row(states + 6 * (size_t)i, actions + 2 * (size_t)i, tau, next_states + 6 * (size_t)i,
                 params ? params + 4 * (size_t)i : NULL);
    return EB_OK;
}

/* a4: _action_transformation_for_end2end, DAM:128-132 */
static void action_transform_row(const float* in, float* out) {
    float a0 = fminf(fmaxf(in[0], -1.05f), 1.05f), a1 = fminf(fmaxf(in[1], -1.05f), 1.05f);
    out[0] = 0.4f * a0;
    out[1] = 2.25f * a1 - 0.75f;
}

int eb_action_transform(eb_handle h, int32_t n, const float* actions, float* scaled, void* stream) {
    (void)stream;
    if (!h || n < 0 || !actions || !scaled) return fail(EB_EINVAL, "eb_action_transform: bad argument");
    for (int i = 0; i < n; ++i) action_transform_row(actions + 2 * (size_t)i, scaled + 2 * (size_t)i);
    return EB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a5: EnvironmentModel.compute_rewards, DAM:186-320                                           */
/* ------------------------------------------------------------------------------------------ */
static void road_terms(int task, float px, float py, float* train, float* real) {
    const float LWN = LANE_W * 3.0f; /* LANE_WIDTH*LANE_NUMBER = 11.25 */
    float t = *train, q = *real;
    if (task == EB_TASK_LEFT) { /* DAM:233-251 */
        if (py < -HALF_CROSS && px < 1.0f) t += sq(px - 1.0f);
        if (py < -HALF_CROSS && LANE_W - px < 1.0f) t += sq(LANE_W - px - 1.0f);
        if (px < 0.0f && LWN - py < 1.0f) t += sq(LWN - py - 1.0f);
        if (px < -HALF_CROSS && py - 0.0f < 1.0f) t += sq(py - 0.0f - 1.0f);
        if (py < -HALF_CROSS && px < 1.0f) q += sq(px - 1.0f);
        if (py < -HALF_CROSS && LANE_W - px < 1.0f) q += sq(LANE_W - px - 1.0f);
        if (px < -HALF_CROSS && LWN - py < 1.0f) q += sq(LWN - py - 1.0f);
        if (px < -HALF_CROSS && py - 0.0f < 1.0f) q += sq(py - 0.0f - 1.0f);
    } else if (task == EB_TASK_STRAIGHT) { /* DAM:252-272 */
        const float LW2 = 2.0f * LANE_W;
        for (int pass = 0; pass < 2; ++pass) {
            float* acc = pass == 0 ? &t : &q;
            if (py < -HALF_CROSS && px - LANE_W < 1.0f) *acc += sq(px - LANE_W - 1.0f);
            if (py < -HALF_CROSS && LW2 - px < 1.0f) *acc += sq(LW2 - px - 1.0f);
            if (py > HALF_CROSS && LWN - px < 1.0f) *acc += sq(LWN - px - 1.0f);
            if (py > HALF_CROSS && px - 0.0f < 1.0f) *acc += sq(px - 0.0f - 1.0f);
        }
    } else { /* right, DAM:273-295 */
        const float LW2 = 2.0f * LANE_W;
        for (int pass = 0; pass < 2; ++pass) {
            float* acc = pass == 0 ? &t : &q;
            if (py < -HALF_CROSS && px - LW2 < 1.0f) *acc += sq(px - LW2 - 1.0f);
            if (py < -HALF_CROSS && LWN - px < 1.0f) *acc += sq(LWN - px - 1.0f);
            if (px > HALF_CROSS && 0.0f - py < 1.0f) *acc += sq(0.0f - py - 1.0f);
            if (px > HALF_CROSS && py - (-LWN) < 1.0f) *acc += sq(py - (-LWN) - 1.0f);
        }
    }
    *train = t; *real = q;
}

/* out5: rewards, punish_train, punish_real, veh2veh4real, veh2road4real; d16 nullable. */
static void rewards_row(const eb_config* c, const float* obs, const float* act, float* o5, float* d16) {
    const float* ego = obs;
    const float* trk = obs + 6;
    const float* veh = obs + 6 + 3 * (c->n_future + 1);
    float steers = act[0], a_xs = act[1];              /* DAM:196 */
    float punish_steer = -sq(steers);                  /* DAM:198 */
    float punish_a_x = -sq(a_xs);                      /* DAM:199 */
    float punish_yaw_rate = -sq(ego[2]);               /* DAM:202 */
    float devi_y = -sq(trk[0]);                        /* DAM:205 */
    float devi_phi = -sq(deg2rad(trk[1]));             /* DAM:206 */
    float devi_v = -sq(trk[2]);                        /* DAM:207 */
    float es, ec;
    eb_sincosf(deg2rad(ego[5]), &es, &ec);
    float efx = ego[3] + LWS * ec, efy = ego[4] + LWS * es; /* DAM:211-212 */
    float erx = ego[3] - LWS * ec, ery = ego[4] - LWS * es; /* DAM:213-214 */
    float v2v_real = 0.0f, v2v_train = 0.0f;                /* DAM:215-216 */
    for (int j = 0; j < c->n_veh; ++j) {                    /* DAM:218-229 */
        const float* v = veh + 4 * j;
        float vs, vc;
        eb_sincosf(deg2rad(v[3]), &vs, &vc);
        float vfx = v[0] + LWS * vc, vfy = v[1] + LWS * vs;
        float vrx = v[0] - LWS * vc, vry = v[1] - LWS * vs;
        const float ex[2] = {efx, erx}, ey[2] = {efy, ery}, wx[2] = {vfx, vrx}, wy[2] = {vfy, vry};
        for (int p = 0; p < 2; ++p)
            for (int q = 0; q < 2; ++q) {
                float d = sqrtf(sq(ex[p] - wx[q]) + sq(ey[p] - wy[q]));
                float t35 = d - 3.5f, t25 = d - 2.5f;
                v2v_train += t35 < 0.0f ? sq(t35) : 0.0f;
                v2v_real += t25 < 0.0f ? sq(t25) : 0.0f;
            }
    }
    float road_train = 0.0f, road_real = 0.0f;              /* DAM:231-232 */
    /* the reference interleaves, per ego point, 4 training terms then 4 real terms; each
     * accumulator only ever sees its own terms, in front-then-rear order */
    road_terms(c->task, efx, efy, &road_train, &road_real);
    road_terms(c->task, erx, ery, &road_train, &road_real);
    float rewards = 0.05f * devi_v + 0.8f * devi_y + 30.0f * devi_phi + 0.02f * punish_yaw_rate +
                    5.0f * punish_steer + 0.05f * punish_a_x; /* DAM:297-298 */
    o5[0] = rewards;
    o5[1] = v2v_train + road_train; /* DAM:299 */
    o5[2] = v2v_real + road_real;   /* DAM:300 */
    o5[3] = v2v_real;
    o5[4] = road_real;
    if (d16) { /* DAM:302-318 */
        d16[0] = punish_steer; d16[1] = punish_a_x; d16[2] = punish_yaw_rate;
        d16[3] = devi_v; d16[4] = devi_y; d16[5] = devi_phi;
        d16[6] = 5.0f * punish_steer; d16[7] = 0.05f * punish_a_x; d16[8] = 0.02f * punish_yaw_rate;
        d16[9] = 0.05f * devi_v; d16[10] = 0.8f * devi_y; d16[11] = 30.0f * devi_phi;
        d16[12] = v2v_train; d16[13] = road_train; d16[14] = v2v_real; d16[15] = road_real;
    }
}

int eb_compute_rewards(eb_handle h, int32_t n_env, const float* obs, const float* actions,
                       float* out5, float* out_dict16, void* stream) {
    (void)stream;
    if (!h || n_env < 0 || !obs || !actions || !out5) return fail(EB_EINVAL, "eb_compute_rewards: bad argument");
    const int D = obs_dim(&h->cfg);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_env; ++i) {
        float o5[5], d16[16];
        rewards_row(&h->cfg, obs + (size_t)D * i, actions + 2 * (size_t)i, o5, out_dict16 ? d16 : NULL);
        for (int k = 0; k < 5; ++k) out5[(size_t)k * n_env + i] = o5[k];
        if (out_dict16)
            for (int k = 0; k < 16; ++k) out_dict16[(size_t)k * n_env + i] = d16[k];
    }
    return EB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a8: ReferencePath closest point + tracking error, DAM:577-580, 702-770                      */
/* ------------------------------------------------------------------------------------------ */
static int closest_index_ratio(const eb_handle h, int p, float x, float y, int ratio) { /* DAM:702-715 */
    const int len = h->lens[p];
    const float* px = h->px[p];
    const float* py = h->py[p];
    float best = INFINITY;
    int best_i = 0;
    for (int i = 0; i < len; i += ratio) { /* np.arange(0, path_len, ratio), DAM:704 */
        float d = sq(x - px[i]) + sq(y - py[i]); /* DAM:712 */
        if (d < best) { best = d; best_i = i; }  /* tf.argmin: first minimum, DAM:714 */
    }
    /* NaN inputs: np/tf argmin returns the first NaN position; with every d NaN that is 0 */
    return best_i;
}

static int closest_index(const eb_handle h, int p, float x, float y) { return closest_index_ratio(h, p, x, y, 10); }

static inline int clamp_index(int i, int len) { /* indexs2points, DAM:727-728 */
    if (i < 0) i = 0;
    if (i >= len) i = len - 1;
    return i;
}

static inline float deal_with_phi_diff(float d) { /* DAM:577-580 */
    if (d > 180.0f) d = d - 360.0f;
    if (d < -180.0f) d = d + 360.0f;
    return d;
}

static float two2one(int task, float ex, float ey, float rx, float ry) { /* DAM:736-752 */
    float delta;
    if (task == EB_TASK_LEFT) {
        delta = sqrtf(sq(ex - (-HALF_CROSS)) + sq(ey - (-HALF_CROSS))) -
                sqrtf(sq(rx - (-HALF_CROSS)) + sq(ry - (-HALF_CROSS)));
        if (ey < -HALF_CROSS) delta = ex - rx;
        if (ex < -HALF_CROSS) delta = ey - ry;
        return -delta;
    } else if (task == EB_TASK_STRAIGHT) {
        delta = ex - rx;
        return -delta;
    } else {
        delta = -(sqrtf(sq(ex - HALF_CROSS) + sq(ey - (-HALF_CROSS))) -
                  sqrtf(sq(rx - HALF_CROSS) + sq(ry - (-HALF_CROSS))));
        if (ey < -HALF_CROSS) delta = ex - rx;
        if (ex > HALF_CROSS) delta = -(ey - ry);
        return -delta;
    }
}

/* out: 3*(n+1) floats */
static void tracking_row(const eb_handle h, int p, float ex, float ey, float ephi, float ev, int n,
                         float* out) {
    const int len = h->lens[p];
    int idx = closest_index(h, p, ex, ey);              /* DAM:754 */
    int ci = clamp_index(idx, len);
    float rx = h->px[p][ci], ry = h->py[p][ci], rphi = h->pphi[p][ci];
    out[0] = two2one(h->cfg.task, ex, ey, rx, ry);      /* DAM:758 */
    out[1] = deal_with_phi_diff(ephi - rphi);           /* DAM:759 */
    out[2] = ev - EXP_V;                                /* DAM:760 */
    int cur = idx;
    for (int k = 0; k < n; ++k) {                       /* future_n_data, DAM:717-724 */
        cur += 80;
        if (cur >= len - 2) cur = len - 2;
        int fi = clamp_index(cur, len);
        out[3 + 3 * k + 0] = h->px[p][fi] - ex;         /* DAM:764 */
        out[3 + 3 * k + 1] = h->py[p][fi] - ey;         /* DAM:765 */
        out[3 + 3 * k + 2] = deal_with_phi_diff(ephi - h->pphi[p][fi]); /* DAM:766 */
    }
}

static int check_paths(eb_handle h, const char* who) {
    if (!h) return fail(EB_EINVAL, who);
    if (h->n_paths < 1) return fail(EB_ESTATE, "paths not set (eb_set_paths)");
    return EB_OK;
}

/* path used by row i: ref_idx[i] when given, else path_id; <0 / >= n_paths -> -1 (zeros) */
static inline int row_path(const eb_handle h, const int32_t* ref_idx, int path_id, int i) {
    int p = ref_idx ? ref_idx[i] : path_id;
    return (p >= 0 && p < h->n_paths) ? p : -1;
}

int eb_find_closest_point(eb_handle h, int32_t n, const float* xs, const float* ys,
                          const int32_t* ref_idx, int32_t path_id, int32_t ratio, int32_t* out_index,
                          float* out_points, void* stream) {
    (void)stream;
    int rc = check_paths(h, "eb_find_closest_point: null handle");
    if (rc) return rc;
    if (n < 0 || !xs || !ys || !out_index || ratio < 1) return fail(EB_EINVAL, "eb_find_closest_point: bad argument");
    if (!ref_idx && (path_id < 0 || path_id >= h->n_paths)) return fail(EB_EINVAL, "eb_find_closest_point: bad path_id");
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int p = row_path(h, ref_idx, path_id, i);
        if (p < 0) {
            out_index[i] = 0;
            if (out_points) { out_points[i] = 0; out_points[(size_t)n + i] = 0; out_points[2 * (size_t)n + i] = 0; }
            continue;
        }
        int idx = closest_index_ratio(h, p, xs[i], ys[i], ratio);
        out_index[i] = idx;
        if (out_points) {
            int ci = clamp_index(idx, h->lens[p]);
            out_points[i] = h->px[p][ci];
            out_points[(size_t)n + i] = h->py[p][ci];
            out_points[2 * (size_t)n + i] = h->pphi[p][ci];
        }
    }
    return EB_OK;
}

int eb_tracking_error(eb_handle h, int32_t n, const float* xs, const float* ys, const float* phis,
                      const float* vs, const int32_t* ref_idx, int32_t path_id, int32_t n_future,
                      float* out, void* stream) {
    (void)stream;
    int rc = check_paths(h, "eb_tracking_error: null handle");
    if (rc) return rc;
    if (n < 0 || !xs || !ys || !phis || !vs || !out || n_future < 0) return fail(EB_EINVAL, "eb_tracking_error: bad argument");
    if (!ref_idx && (path_id < 0 || path_id >= h->n_paths)) return fail(EB_EINVAL, "eb_tracking_error: bad path_id");
    const int T = 3 * (n_future + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int p = row_path(h, ref_idx, path_id, i);
        float* o = out + (size_t)T * i;
        if (p < 0) { for (int k = 0; k < T; ++k) o[k] = 0.0f; continue; } /* DAM:342, 352 */
        tracking_row(h, p, xs[i], ys[i], phis[i], vs[i], n_future, o);
    }
    return EB_OK;
}

/* ReferencePath.indexs2points (DAM:726-733) + future_n_data (DAM:717-724) */
int eb_path_points(eb_handle h, int32_t n, const int32_t* index, const int32_t* ref_idx, int32_t path_id,
                   int32_t n_future, float* out_points, void* stream) {
    (void)stream;
    int rc = check_paths(h, "eb_path_points: null handle");
    if (rc) return rc;
    if (n < 0 || n_future < 0 || (n > 0 && (!index || !out_points))) return fail(EB_EINVAL, "eb_path_points: bad argument");
    if (!ref_idx && (path_id < 0 || path_id >= h->n_paths)) return fail(EB_EINVAL, "eb_path_points: bad path_id");
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const int p = row_path(h, ref_idx, path_id, i);
        int cur = index[i];
        for (int k = 0; k <= n_future; ++k) {
            float* o = out_points + (size_t)k * 3 * n;
            if (p < 0) { o[i] = 0; o[(size_t)n + i] = 0; o[2 * (size_t)n + i] = 0; continue; }
            const int len = h->lens[p];
            if (k > 0) {                               /* DAM:719-722 */
                cur += 80;
                if (cur >= len - 2) cur = len - 2;
            }
            const int ci = clamp_index(cur, len);      /* DAM:727-728 */
            o[i] = h->px[p][ci]; o[(size_t)n + i] = h->py[p][ci]; o[2 * (size_t)n + i] = h->pphi[p][ci];
        }
    }
    return EB_OK;
}

int eb_phi_diff(eb_handle h, int32_t n, const float* phi_diff, float* out, void* stream) { /* DAM:577-580 */
    (void)stream;
    if (!h || n < 0 || (n > 0 && (!phi_diff || !out))) return fail(EB_EINVAL, "eb_phi_diff: bad argument");
    for (int i = 0; i < n; ++i) out[i] = deal_with_phi_diff(phi_diff[i]);
    return EB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a10: veh_predict / predict_for_a_mode, DAM:394-427                                          */
/* ------------------------------------------------------------------------------------------ */
static void veh_predict_one(const float* v, int turn, float* o) {
    float x = v[0], y = v[1], vv = v[2], phi = v[3];   /* DAM:406 */
    float phi_rad = deg2rad(phi);                       /* DAM:407 */
    int middle = (x > -HALF_CROSS && x < HALF_CROSS) && (y > -HALF_CROSS && y < HALF_CROSS); /* DAM:409-410 */
    float sn, cs;
    eb_sincosf(phi_rad, &sn, &cs);
    float dx = vv / 10.0f * cs;                         /* DAM:413 */
    float dy = vv / 10.0f * sn;                         /* DAM:414 */
    float dphi = 0.0f;
    if (turn == TURN_LEFT) dphi = middle ? (vv / 26.875f) / 10.0f : 0.0f;          /* DAM:417 */
    else if (turn == TURN_RIGHT) dphi = middle ? -(vv / 15.625f) / 10.0f : 0.0f;   /* DAM:419 */
    float nphi = phi_rad + dphi;                        /* DAM:423 */
    if (nphi > PI_F) nphi = nphi - TWO_PI_F;            /* DAM:424 */
    if (nphi <= -PI_F) nphi = nphi + TWO_PI_F;          /* DAM:425 */
    o[0] = x + dx; o[1] = y + dy; o[2] = vv;            /* DAM:422-423 */
    o[3] = rad2deg(nphi);                               /* DAM:426 */
}

static int check_modes(eb_handle h) {
    if (!h->modes_set) return fail(EB_ESTATE, "vehicle modes not set (eb_set_veh_modes)");
    return EB_OK;
}

int eb_veh_predict(eb_handle h, int32_t n_env, const float* veh, float* veh_out, void* stream) {
    (void)stream;
    if (!h || n_env < 0 || !veh || !veh_out) return fail(EB_EINVAL, "eb_veh_predict: bad argument");
    int rc = check_modes(h);
    if (rc) return rc;
    const int N = h->cfg.n_veh;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_env; ++i)
        for (int j = 0; j < N; ++j)
            veh_predict_one(veh + ((size_t)i * N + j) * 4, h->turn[j], veh_out + ((size_t)i * N + j) * 4);
    return EB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a6/a7: compute_next_obses + ego_predict, DAM:322-358, 386-392                               */
/* ------------------------------------------------------------------------------------------ */
static void next_obs_row(const eb_handle h, const float* obs, const float* act, int p, float* out) {
    const eb_config* c = &h->cfg;
    const int T = 3 * (c->n_future + 1);
    float nx[6];
    f_xu_row(obs, act, (float)(1 / 10.), nx, NULL);      /* prediction at base_frequency 10, DAM:85-87, 387 */
    nx[0] = fminf(fmaxf(nx[0], 0.0f), 35.0f);            /* clip_by_value(v_xs, 0., 35.), DAM:390 */
    for (int k = 0; k < 6; ++k) out[k] = nx[k];
    if (p < 0) for (int k = 0; k < T; ++k) out[6 + k] = 0.0f;   /* DAM:342, 352 */
    else tracking_row(h, p, nx[3], nx[4], nx[5], nx[0], c->n_future, out + 6); /* DAM:335-339 / 347-351 */
    const float* veh = obs + 6 + T;
    for (int j = 0; j < c->n_veh; ++j) veh_predict_one(veh + 4 * j, h->turn[j], out + 6 + T + 4 * j); /* DAM:355 */
}

/* a7: EnvironmentModel.ego_predict, DAM:386-392 */
int eb_ego_predict(eb_handle h, int32_t n, const float* ego, const float* actions, float* next_ego, void* stream) {
    (void)stream;
    if (!h || n < 0 || (n > 0 && (!ego || !actions || !next_ego))) return fail(EB_EINVAL, "eb_ego_predict: bad argument");
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float nx[6];
        f_xu_row(ego + 6 * (size_t)i, actions + 2 * (size_t)i, (float)(1 / 10.), nx, NULL);   /* DAM:387 */
        nx[0] = fminf(fmaxf(nx[0], 0.0f), 35.0f);                                             /* DAM:390 */
        memcpy(next_ego + 6 * (size_t)i, nx, sizeof nx);
    }
    return EB_OK;
}

static int check_rollout(eb_handle h, int n_env, const int32_t* ref_idx, int path_id, const char* who) {
    (void)n_env;
    int rc = check_paths(h, who);
    if (rc) return rc;
    rc = check_modes(h);
    if (rc) return rc;
    if (h->cfg.mode == EB_MODE_TRAINING) {
        if (!ref_idx) return fail(EB_EINVAL, "training mode needs ref_idx (EnvironmentModel.reset(obses, ref_indexes))");
    } else if (path_id < 0 || path_id >= h->n_paths) return fail(EB_EINVAL, "bad path_id");
    return EB_OK;
}

int eb_compute_next_obses(eb_handle h, int32_t n_env, const float* obs, const float* actions,
                          const int32_t* ref_idx, int32_t path_id, float* obs_out, void* stream) {
    (void)stream;
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_compute_next_obses: null handle");
    if (rc) return rc;
    if (n_env < 0 || !obs || !actions || !obs_out) return fail(EB_EINVAL, "eb_compute_next_obses: bad argument");
    const int D = obs_dim(&h->cfg);
    const int training = h->cfg.mode == EB_MODE_TRAINING;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_env; ++i) {
        int p = training ? row_path(h, ref_idx, 0, i) : path_id;
        next_obs_row(h, obs + (size_t)D * i, actions + 2 * (size_t)i, p, obs_out + (size_t)D * i);
    }
    return EB_OK;
}

/* a11: rollout_out, DAM:118-126 */
int eb_rollout_step(eb_handle h, int32_t n_env, const float* obs_in, const float* actions,
                    const int32_t* ref_idx, int32_t path_id, float* obs_out, float* out5,
                    float* scaled_actions, void* stream) {
    (void)stream;
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_rollout_step: null handle");
    if (rc) return rc;
    if (n_env < 0 || !obs_in || !actions || !obs_out || !out5) return fail(EB_EINVAL, "eb_rollout_step: bad argument");
    const int D = obs_dim(&h->cfg);
    const int training = h->cfg.mode == EB_MODE_TRAINING;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_env; ++i) {
        float act[2], o5[5];
        action_transform_row(actions + 2 * (size_t)i, act);        /* DAM:120 */
        rewards_row(&h->cfg, obs_in + (size_t)D * i, act, o5, NULL); /* DAM:121-122 */
        int p = training ? row_path(h, ref_idx, 0, i) : path_id;
        float tmp[6 + 3 * 65 + 4 * EB_MAX_VEH];
        next_obs_row(h, obs_in + (size_t)D * i, act, p, tmp);       /* DAM:123 */
        memcpy(obs_out + (size_t)D * i, tmp, sizeof(float) * D);
        for (int k = 0; k < 5; ++k) out5[(size_t)k * n_env + i] = o5[k];
        if (scaled_actions) { scaled_actions[2 * (size_t)i] = act[0]; scaled_actions[2 * (size_t)i + 1] = act[1]; }
    }
    return EB_OK;
}

int eb_rollout_tape(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in,
                    const float* action_tape, const int32_t* ref_idx, int32_t path_id,
                    float* obs_work, float* obs_out, float* out5_steps, void* stream) {
    if (horizon < 1 || !obs_work || !obs_out || !action_tape || !out5_steps)
        return fail(EB_EINVAL, "eb_rollout_tape: bad argument");
    const float* cur = obs_in;
    /* ping-pong so that the last step lands in obs_out */
    for (int t = 0; t < horizon; ++t) {
        float* dst = ((horizon - 1 - t) % 2 == 0) ? obs_out : obs_work;
        int rc = eb_rollout_step(h, n_env, cur, action_tape + (size_t)t * n_env * 2, ref_idx, path_id, dst,
                                 out5_steps + (size_t)t * 5 * n_env, NULL, stream);
        if (rc) return rc;
        cur = dst;
    }
    return EB_OK;
}

/* gated rollout (include/envbuild.h): on the CPU there is nobody to wait for, so every gate must already be open; the
 * steps then run as in eb_rollout_tape, obs_steps[t] = the obs after step t, step_done[t][0][0..15] = 1 (one "block": one 64-byte record per step). */
int eb_rollout_gated_blocks(eb_handle h, int32_t n_env, int32_t* n_blocks) {
    if (!h || n_env < 0 || !n_blocks) return fail(EB_EINVAL, "eb_rollout_gated_blocks: bad argument");
    int rc = check_paths(h, "eb_rollout_gated_blocks: null handle");
    if (!rc) rc = check_modes(h);
    if (rc) return rc;
    *n_blocks = n_env > 0 ? 1 : 0;
    return EB_OK;
}

int eb_rollout_gated(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in, const float* action_tape,
                     const int32_t* ref_idx, int32_t path_id, float* obs_work, float* obs_out, float* out5_steps,
                     float* obs_steps, const uint32_t* step_ready, uint32_t* step_done, int32_t n_blocks,
                     uint32_t* status, int32_t spin_limit, void* stream) {
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_rollout_gated: null handle");
    if (rc) return rc;
    if (n_blocks != 1) return fail(EB_EINVAL, "eb_rollout_gated: n_blocks does not match the grid this handle launches now (call eb_rollout_gated_blocks again)");
    if (n_env < 0 || horizon < 1 || !obs_in || !action_tape || !obs_work || !obs_out || !out5_steps || !step_ready || !step_done ||
        !status || spin_limit < 1)
        return fail(EB_EINVAL, "eb_rollout_gated: bad argument");
    if (obs_work == obs_out || obs_in == obs_work || obs_in == obs_out)
        return fail(EB_EINVAL, "eb_rollout_gated: obs_in, obs_work and obs_out must be distinct buffers");
    const size_t row = (size_t)obs_dim(&h->cfg) * n_env;
    const float* cur = obs_in;
    for (int t = 0; t < horizon; ++t) {
        if (!step_ready[t]) { status[0] = 1; return EB_OK; }              /* the launch "gives up" at a shut gate */
        float* dst = ((horizon - 1 - t) % 2 == 0) ? obs_out : obs_work;
        rc = eb_rollout_step(h, n_env, cur, action_tape + (size_t)t * n_env * 2, ref_idx, path_id, dst,
                             out5_steps + (size_t)t * 5 * n_env, NULL, stream);
        if (rc) return rc;
        if (obs_steps) memcpy(obs_steps + (size_t)t * row, dst, row * sizeof(float));
        for (int w = 0; w < 16; ++w) step_done[(size_t)t * 16 + w] = 1;
        cur = dst;
    }
    return EB_OK;
}

int eb_gate_feed(eb_handle h, int32_t n_env, int32_t horizon, int32_t n_blocks, const float* staged_tape,
                 float* live_tape, uint32_t* step_ready, const uint32_t* step_done, uint32_t* status,
                 int32_t spin_limit, void* after_stream, int32_t wait_after, void* stream) {
    (void)stream; (void)step_done; (void)status; (void)after_stream; (void)wait_after;
    if (!h || n_env < 1 || (n_env & 1) || horizon < 1 || n_blocks < 1 || !staged_tape || !live_tape || !step_ready || !step_done ||
        !status || spin_limit < 1 || staged_tape == live_tape)
        return fail(EB_EINVAL, "eb_gate_feed: bad argument (n_env even: a step's actions are copied 16 bytes at a time)");
    memcpy(live_tape, staged_tape, (size_t)horizon * n_env * 2 * sizeof(float));   /* sequential host: the whole tape at once */
    for (int t = 0; t < horizon; ++t) step_ready[t] = 1;
    return EB_OK;
}

/* fp16 state storage (BASELINE.json configs[4]): rows are widened to fp32 (exact), stepped by the fp32 code above and
 * rounded to binary16, nearest-even, on the way out.  Software conversions: no F16C dependency. */
static float half_to_float(uint16_t hbits) {
    const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16;
    uint32_t e = (hbits >> 10) & 0x1Fu, m = hbits & 0x3FFu, bits;
    if (e == 0) {
        if (m == 0) bits = sign;
        else { /* subnormal: normalise */
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            bits = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 0x3FFu) << 13);
        }
    } else if (e == 31) bits = sign | 0x7F800000u | (m << 13);
    else bits = sign | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
static uint16_t float_to_half(float f) { /* round to nearest even */
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t a = x & 0x7FFFFFFFu;
    if (a >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (a > 0x7F800000u ? (0x200u | ((a >> 13) & 0x3FFu)) : 0u));
    if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);            /* >= 65520: rounds to inf */
    if (a < 0x33000001u) return sign;                                    /* <= 2^-25: rounds to zero */
    if (a < 0x38800000u) {                                               /* subnormal half */
        const int shift = 113 - (int)(a >> 23);                          /* 1..24 */
        const uint32_t mant = (a & 0x7FFFFFu) | 0x800000u;
        uint32_t h = mant >> (shift + 13);
        const uint32_t rem = mant & ((1u << (shift + 13)) - 1u), half = 1u << (shift + 12);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((a >> 23) - 112u) << 10 | ((a >> 13) & 0x3FFu);
    const uint32_t rem = a & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;              /* carries into the exponent correctly */
    return (uint16_t)(sign | h);
}

void eb_oracle_half_to_float(const uint16_t* in, float* out, int n) { for (int i = 0; i < n; ++i) out[i] = half_to_float(in[i]); }
void eb_oracle_float_to_half(const float* in, uint16_t* out, int n) { for (int i = 0; i < n; ++i) out[i] = float_to_half(in[i]); }

int eb_rollout_step_f16(eb_handle h, int32_t n_env, const uint16_t* obs_in, const float* actions,
                        const int32_t* ref_idx, int32_t path_id, uint16_t* obs_out, float* out5,
                        float* scaled_actions, void* stream) {
    if (h && n_env == 0) return EB_OK;
    if (!h) return fail(EB_EINVAL, "eb_rollout_step_f16: null handle");
    if (n_env < 0 || !obs_in || !obs_out) return fail(EB_EINVAL, "eb_rollout_step_f16: bad argument");
    const size_t n = (size_t)n_env * (size_t)obs_dim(&h->cfg);
    float* a = (float*)calloc(n ? n : 1, sizeof(float));
    float* b = (float*)calloc(n ? n : 1, sizeof(float));
    if (!a || !b) { free(a); free(b); return fail(EB_ENOMEM, "eb_rollout_step_f16: out of memory"); }
    for (size_t i = 0; i < n; ++i) a[i] = half_to_float(obs_in[i]);
    int rc = eb_rollout_step(h, n_env, a, actions, ref_idx, path_id, b, out5, scaled_actions, stream);
    if (rc == EB_OK)
        for (size_t i = 0; i < n; ++i) obs_out[i] = float_to_half(b[i]);
    free(a); free(b);
    return rc;
}

int eb_rollout_tape_f16(eb_handle h, int32_t n_env, int32_t horizon, const uint16_t* obs_in,
                        const float* action_tape, const int32_t* ref_idx, int32_t path_id,
                        uint16_t* obs_work, uint16_t* obs_out, float* out5_steps, void* stream) {
    if (horizon < 1 || !obs_work || !obs_out || !action_tape || !out5_steps)
        return fail(EB_EINVAL, "eb_rollout_tape_f16: bad argument");
    const uint16_t* cur = obs_in;
    for (int t = 0; t < horizon; ++t) {
        uint16_t* dst = ((horizon - 1 - t) % 2 == 0) ? obs_out : obs_work;
        int rc = eb_rollout_step_f16(h, n_env, cur, action_tape + (size_t)t * n_env * 2, ref_idx, path_id, dst,
                                     out5_steps + (size_t)t * 5 * n_env, NULL, stream);
        if (rc) return rc;
        cur = dst;
    }
    return EB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* a12: EnvironmentModel.ss, DAM:134-184                                                       */
/* ------------------------------------------------------------------------------------------ */
int eb_ss(eb_handle h, int32_t n_env, const float* obs, const float* actions, const int32_t* ref_idx,
          int32_t path_id, double lam, float* out, void* stream) {
    (void)stream;
    if (h && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_ss: null handle");
    if (rc) return rc;
    if (n_env < 0 || !obs || !actions || !out) return fail(EB_EINVAL, "eb_ss: bad argument");
    const eb_config* c = &h->cfg;
    const int D = obs_dim(c), T = 3 * (c->n_future + 1);
    const int training = c->mode == EB_MODE_TRAINING;
    /* (1-lam): python float minus python float in double, then cast where it meets the tensor */
    const float one_m_lam = (float)(1.0 - lam);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_env; ++i) {
        const float* o = obs + (size_t)D * i;
        float act[2];
        action_transform_row(actions + 2 * (size_t)i, act);          /* DAM:135 */
        float nobs[6 + 3 * 65 + 4 * EB_MAX_VEH];
        int p = training ? row_path(h, ref_idx, 0, i) : path_id;
        next_obs_row(h, o, act, p, nobs);                             /* DAM:136 */
        float s0, c0, s1, c1;
        eb_sincosf(deg2rad(o[5]), &s0, &c0);
        eb_sincosf(deg2rad(nobs[5]), &s1, &c1);
        float ex[2] = {o[3] + LWS * c0, o[3] - LWS * c0}, ey[2] = {o[4] + LWS * s0, o[4] - LWS * s0};            /* DAM:144-148 */
        float nex[2] = {nobs[3] + LWS * c1, nobs[3] - LWS * c1}, ney[2] = {nobs[4] + LWS * s1, nobs[4] - LWS * s1}; /* DAM:150-154 */
        float acc = 0.0f;                                             /* DAM:156 */
        for (int j = 0; j < c->n_veh; ++j) {                          /* DAM:157-183 */
            const float* v = o + 6 + T + 4 * j;
            const float* nv = nobs + 6 + T + 4 * j;
            float e2v = sqrtf(sq(o[3] - v[0]) + sq(o[4] - v[1]));     /* DAM:159 */
            float vs_, vc_, ns_, nc_;
            eb_sincosf(deg2rad(v[3]), &vs_, &vc_);
            eb_sincosf(deg2rad(nv[3]), &ns_, &nc_);
            float wx[2] = {v[0] + LWS * vc_, v[0] - LWS * vc_}, wy[2] = {v[1] + LWS * vs_, v[1] - LWS * vs_};
            float nwx[2] = {nv[0] + LWS * nc_, nv[0] - LWS * nc_}, nwy[2] = {nv[1] + LWS * ns_, nv[1] - LWS * ns_};
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    float d = sqrtf(sq(ex[a] - wx[b]) + sq(ey[a] - wy[b]));       /* DAM:176-177 */
                    float nd = sqrtf(sq(nex[a] - nwx[b]) + sq(ney[a] - nwy[b]));  /* DAM:178-179 */
                    float next_g = nd - 2.5f, g = d - 2.5f;                       /* DAM:180-181 */
                    float t = next_g - one_m_lam * g;                             /* DAM:182 */
                    if (t < 0.0f && e2v < 10.0f) acc += sq(t);
                }
        }
        out[i] = acc;
    }
    return EB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* real-env step pieces (endtoend.py), batched.  The reference evaluates these with a mix of    */
/* np.float32 scalars and python doubles whose promotion depends on the NumPy version           */
/* (SURVEY.md Appendix A); the restatement is fp32 throughout (NumPy >= 2 semantics), with the   */
/* deterministic sin/cos above.  Masks agree with the reference off-threshold (tests record     */
/* margins).                                                                                    */
/* ------------------------------------------------------------------------------------------ */
static inline float deal_with_phi(float phi) { /* UTL:232-237 */
    if (!(fabsf(phi) <= EB_WRAP_MAX_DEG)) return phi;   /* +-inf would never leave the loops (include/envbuild.h, "angle wrapping") */
    while (phi > 180.0f) phi -= 360.0f;
    while (phi <= -180.0f) phi += 360.0f;
    return phi;
}

/* a14: _get_next_ego_state, E2E:269-283 */
int eb_env_ego_step(eb_handle h, int32_t n, const float* ego, const float* actions, float* next_ego,
                    float* params, void* stream) {
    (void)stream;
    if (!h || n < 0 || !ego || !actions || !next_ego || !params) return fail(EB_EINVAL, "eb_env_ego_step: bad argument");
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float nx[6], pr[4];
        f_xu_row(ego + 6 * (size_t)i, actions + 2 * (size_t)i, (float)(1 / 10.), nx, pr); /* E2E:279 */
        nx[0] = nx[0] >= 0.0f ? nx[0] : 0.0f;   /* E2E:281 */
        nx[5] = deal_with_phi(nx[5]);           /* E2E:282 */
        memcpy(next_ego + 6 * (size_t)i, nx, sizeof nx);
        memcpy(params + 4 * (size_t)i, pr, sizeof pr);
    }
    return EB_OK;
}

/* a16: _construct_veh_vector_short, E2E:340-464.
 * Values flow as in the reference's Python: a vehicle's x, y, phi are python floats (float64) — fp32-valued in the
 * plain env, genuinely float64 after cal_info_in_transform_coordination in the 12-ego scene — while everything
 * derived from ego_dynamics is np.float32.  Comparing a python float with an np.float32 rounds the python float to
 * fp32 first (NumPy >= 2 scalar rules); comparing with a python constant, or two vehicles with each other in a sort
 * key, happens in float64.  For fp32-valued vehicles both kinds are plain fp32 comparisons. */
typedef struct { double x, y, phi; float v; } veh4;
static inline int lt_ego(double a, float b) { return (float)a < b; }
static inline int gt_ego(double a, float b) { return (float)a > b; }

static int veh_in_range(int task, int m, const veh4* v, float ego_x, float ego_y) { /* E2E:393-411 */
    const double C2 = 25.0;
    switch (m) {
        case EB_VMODE_DL: return v->x > -C2 - 10 && gt_ego(v->y, ego_y - 2.0f);
        case EB_VMODE_DU: return gt_ego(v->y, ego_y - 2.0f) && v->y < C2 + 10 && lt_ego(v->x, ego_x + 5.0f);
        case EB_VMODE_DR: return v->x < C2 + 10 && gt_ego(v->y, ego_y);
        case EB_VMODE_RU: return v->x < C2 + 10 && v->y < C2 + 10;
        case EB_VMODE_UR:
            if (task == EB_TASK_STRAIGHT) return lt_ego(v->x, ego_x + 7.0f) && gt_ego(v->y, ego_y) && v->y < C2 + 10;
            if (task == EB_TASK_RIGHT) return v->x < C2 + 10 && v->y < C2;
            return 1;
        case EB_VMODE_UD: {
            const float ey2 = ego_y - 2.0f;   /* max(ego_y - 2, -CROSSROAD_SIZE / 2): python's max keeps the first unless the second is larger */
            const int lower = (-25.0f > ey2) ? (-C2 < v->y) : gt_ego(v->y, ey2);
            return lower && v->y < C2 && lt_ego(v->x, ego_x);
        }
        case EB_VMODE_UL: return -C2 - 10 < v->x && lt_ego(v->x, ego_x) && v->y < C2;
        case EB_VMODE_LR: return -C2 - 10 < v->x && v->x < C2 + 10;
        default: return 1; /* rd rl lu ld: "not interest in case of traffic light", E2E:398-411 */
    }
}

/* <0 when a sorts before b under the mode's key (E2E:414-428); 0 = equal keys (stable order) */
static int veh_cmp(int task, int m, const veh4* a, const veh4* b) {
#define ASC(f) do { if (a->f < b->f) return -1; if (a->f > b->f) return 1; } while (0)
#define DESC(f) do { if (a->f > b->f) return -1; if (a->f < b->f) return 1; } while (0)
    switch (m) {
        case EB_VMODE_DL: ASC(y); DESC(x); return 0;           /* key (y, -x) */
        case EB_VMODE_DU: ASC(y); return 0;
        case EB_VMODE_DR: ASC(y); ASC(x); return 0;
        case EB_VMODE_RU: ASC(x); DESC(y); return 0;           /* key (-x, y), reverse=True */
        case EB_VMODE_UR:
            if (task == EB_TASK_STRAIGHT) { ASC(y); return 0; }
            if (task == EB_TASK_RIGHT) { ASC(y); DESC(x); return 0; } /* key (-y, x), reverse=True */
            return 0;
        case EB_VMODE_UD: ASC(y); return 0;
        case EB_VMODE_UL: ASC(y); ASC(x); return 0;            /* key (-y, -x), reverse=True */
        case EB_VMODE_LR: DESC(x); return 0;                   /* key -x */
        default: return 0;
    }
#undef ASC
#undef DESC
}

static veh4 veh_fill_value(int m) { /* mode2fillvalue, E2E:439-447 */
    const double C2 = 25.0, LW = 3.75;
    veh4 f = {0, 0, 0, 0};
    switch (m) {
        case EB_VMODE_DL: f.x = LW / 2; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_DU: f.x = LW * 1.5; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_DR: f.x = LW * 2.5; f.y = -(C2 + 30); f.phi = 90; break;
        case EB_VMODE_RU: f.x = C2 + 15; f.y = LW * 2.5; f.phi = 180; break;
        case EB_VMODE_UR: f.x = -LW / 2; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_UD: f.x = -LW * 1.5; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_UL: f.x = -LW * 2.5; f.y = C2 + 20; f.phi = -90; break;
        case EB_VMODE_LR: f.x = -(C2 + 20); f.y = -LW * 1.5; f.phi = 0; break;
        default: break; /* the reference defines no fill value for rd rl lu ld (never requested) */
    }
    return f;
}

/* ---- exit-relative frames of the 12-ego scene: multi_ego.py:33, 84-120; UTL:120-196; E2E:345-385 ---- */
static const int EXIT_ANGLE[4] = {0, 90, 180, -90};                  /* ROTATE_ANGLE, multi_ego.py:33 */
static const int MODE_END[12] = {3, 2, 1, 0, 3, 2, 1, 0, 3, 2, 1, 0}; /* end direction (d r u l = 0 1 2 3) of dl du dr rd rl ru ur ud ul lu lr ld */
static const int MODE_OF[4][4] = {{-1, 2, 1, 0}, {3, -1, 5, 4}, {7, 6, -1, 8}, {11, 10, 9, -1}};   /* [start][end] -> EB_VMODE_* */
/* route of a WORLD mode seen from exit k (E2E:345-385: exit R names world edge '2o' as 'do', ...) */
static int exit_relative_mode(int world_mode, int k) {
    if (world_mode < 0 || world_mode >= EB_VMODE_COUNT) return EB_VMODE_EMPTY;
    const int st = world_mode / 3, en = MODE_END[world_mode];
    return MODE_OF[(st - k + 4) & 3][(en - k + 4) & 3];
}
/* rotate_coordination (UTL:120-141) on python floats: float64 throughout */
static void rotate_f64(double x, double y, double d, int rotate_d, double* ox, double* oy, double* od) {
    const double r = rotate_d * PI_D / 180;
    *ox = x * cos(r) + y * sin(r);
    *oy = -x * sin(r) + y * cos(r);
    double t = d - rotate_d;
    if (!(fabs(t) <= (double)EB_WRAP_MAX_DEG)) {}
    else if (t > 180) { while (t > 180) t = t - 360; }
    else if (t <= -180) { while (t <= -180) t = t + 360; }
    *od = t;
}

static void build_veh_row(const eb_handle h, int m_cand, const float* cand, const uint8_t* cmode,
                          float ego_x, float ego_y, int light, int exit_k, float* out) {
    const int task = h->cfg.task, N = h->cfg.n_veh;
    int taken_rank[EB_MAX_VEH]; /* per slot: how many earlier slots share its mode */
    for (int s = 0; s < N; ++s) {
        int k = 0;
        for (int t = 0; t < s; ++t) k += h->vmode[t] == h->vmode[s];
        taken_rank[s] = k;
    }
    /* virtual red-light cars appended after the real ones, E2E:386-390 */
    const int virt = task != EB_TASK_RIGHT && light && ego_y < -HALF_CROSS;
    for (int s = 0; s < N; ++s) {
        const int m = h->vmode[s];
        /* gather this mode's candidates in insertion order */
        veh4 lst[256 + 1];
        int cnt = 0;
        for (int i = 0; i < m_cand && cnt < 256; ++i) {
            const int mi = exit_k < 0 ? cmode[i] : exit_relative_mode(cmode[i], exit_k);
            if (mi != m) continue;
            veh4 v = {cand[4 * i], cand[4 * i + 1], cand[4 * i + 3], cand[4 * i + 2]};
            if (exit_k >= 0)   /* cal_info_in_transform_coordination(vehicles, 0, 0, rotate_angle), multi_ego.py:87 */
                rotate_f64((double)cand[4 * i] - 0, (double)cand[4 * i + 1] - 0, (double)cand[4 * i + 3], EXIT_ANGLE[exit_k],
                           &v.x, &v.y, &v.phi);
            if (veh_in_range(task, m, &v, ego_x, ego_y)) lst[cnt++] = v;
        }
        if (virt && (m == EB_VMODE_DL || m == EB_VMODE_DU)) {
            veh4 v = {m == EB_VMODE_DL ? 3.75 / 2 : 3.75 * 1.5, -25.0 + 2.5, 90.0, 0.0f};
            if (veh_in_range(task, m, &v, ego_x, ego_y)) lst[cnt++] = v;
        }
        /* stable insertion sort under the mode's key */
        for (int i = 1; i < cnt; ++i) {
            veh4 key = lst[i];
            int j = i - 1;
            while (j >= 0 && veh_cmp(task, m, &key, &lst[j]) < 0) { lst[j + 1] = lst[j]; --j; }
            lst[j + 1] = key;
        }
        veh4 r = taken_rank[s] < cnt ? lst[taken_rank[s]] : veh_fill_value(m); /* slice_or_fill, E2E:431-437 */
        out[4 * s] = (float)r.x; out[4 * s + 1] = (float)r.y; out[4 * s + 2] = r.v; out[4 * s + 3] = (float)r.phi; /* E2E:460-463 */
    }
}

/* a16: _get_obs, E2E:285-303 */
int eb_get_obs(eb_handle h, int32_t n_env, const float* ego, const int32_t* ref_idx, int32_t path_id,
               int32_t m_cand, const float* cand, const uint8_t* cand_mode, const uint8_t* v_light,
               const uint8_t* virtual_flag, const uint8_t* exit_id, const uint8_t* row_mask, float* obs_out, void* stream) {
    (void)stream;
    int rc = check_paths(h, "eb_get_obs: null handle");
    if (rc) return rc;
    rc = check_modes(h);
    if (rc) return rc;
    if (n_env < 0 || !ego || m_cand < 0 || m_cand > 256 || (m_cand > 0 && (!cand || !cand_mode)) || !obs_out)
        return fail(EB_EINVAL, "eb_get_obs: bad argument (m_cand <= 256)");
    if (!ref_idx && (path_id < 0 || path_id >= h->n_paths)) return fail(EB_EINVAL, "eb_get_obs: bad path_id");
    if (exit_id)
        for (int i = 0; i < n_env; ++i)
            if (exit_id[i] > EB_EXIT_L) return fail(EB_EINVAL, "eb_get_obs: bad exit id");
    const eb_config* c = &h->cfg;
    const int D = obs_dim(c), T = 3 * (c->n_future + 1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_env; ++i) {
        if (row_mask && !row_mask[i]) continue;                        /* masked pass: the other rows keep their contents */
        const float* e = ego + 6 * (size_t)i;
        float* o = obs_out + (size_t)D * i;
        for (int k = 0; k < 6; ++k) o[k] = e[k];                       /* E2E:329-338 */
        int p = row_path(h, ref_idx, path_id, i);
        if (p < 0) for (int k = 0; k < T; ++k) o[6 + k] = 0.0f;
        else tracking_row(h, p, e[3], e[4], e[5], e[0], c->n_future, o + 6); /* E2E:293-297 */
        const int ek = exit_id ? exit_id[i] : -1;
        int vl = v_light ? v_light[i] : 0;
        if (ek == EB_EXIT_R || ek == EB_EXIT_L) vl = vl != 2 ? 2 : 0;   /* multi_ego.py:89-92 */
        const int light = vl != 0 || (virtual_flag && virtual_flag[i] != 0);   /* E2E:387-388 */
        build_veh_row(h, m_cand, cand + (size_t)i * m_cand * 4, cand_mode + (size_t)i * m_cand, e[3], e[4], light, ek,
                      o + 6 + T);
    }
    return EB_OK;
}

/* cal_ego_info_in_transform_coordination (UTL:184-196) on np.float32 fields: np.float32 * python float is an fp32
 * product with the python float rounded to fp32 (NumPy >= 2) */
int eb_exit_frame(eb_handle h, int32_t n, const uint8_t* exit_id, int32_t inverse, const float* ego, float* ego_out,
                  void* stream) {
    (void)stream;
    if (!h || n < 0 || (n > 0 && (!exit_id || !ego || !ego_out))) return fail(EB_EINVAL, "eb_exit_frame: bad argument");
    for (int i = 0; i < n; ++i)
        if (exit_id[i] > EB_EXIT_L) return fail(EB_EINVAL, "eb_exit_frame: bad exit id");
    for (int i = 0; i < n; ++i) {
        const float* e = ego + 6 * (size_t)i;
        float* o = ego_out + 6 * (size_t)i;
        const int a = inverse ? -EXIT_ANGLE[exit_id[i]] : EXIT_ANGLE[exit_id[i]];
        const double r = a * PI_D / 180;                               /* UTL:130 */
        const float c = (float)cos(r), sn = (float)sin(r);
        const float x = e[3] - 0.0f, y = e[4] - 0.0f;                  /* shift_coordination by (0, 0), UTL:116-117 */
        const float tx = x * c + y * sn;                               /* UTL:131 */
        const float ty = -x * sn + y * c;                              /* UTL:132 */
        float d = e[5] - (float)a;                                     /* UTL:133 */
        if (!(fabsf(d) <= EB_WRAP_MAX_DEG)) {}
        else if (d > 180.0f) { while (d > 180.0f) d = d - 360.0f; }    /* UTL:134-139 */
        else if (d <= -180.0f) { while (d <= -180.0f) d = d + 360.0f; }
        o[0] = e[0]; o[1] = e[1]; o[2] = e[2]; o[3] = tx; o[4] = ty; o[5] = d;
    }
    return EB_OK;
}

/* judge_feasible, UTL:73-104 */
static int judge_feasible(float x, float y, int task) {
    const float C2 = HALF_CROSS, LW = LANE_W;
    int middle = (-C2 < y && y < C2) && (-C2 < x && x < C2);
    if (task == EB_TASK_LEFT)
        return (0.0f < x && x < LW && y <= -C2) || (0.0f < y && y < LW * 3.0f && x < -C2) || middle;
    if (task == EB_TASK_STRAIGHT)
        return (LW < x && x < LW * 2.0f && y <= -C2) || (0.0f < x && x < LW * 3.0f && y >= C2) || middle;
    return (LW * 2.0f < x && x < LW * 3.0f && y <= -C2) || (-LW * 3.0f < y && y < 0.0f && x > C2) || middle;
}

/* a17: _judge_done, E2E:200-256 */
int eb_judge_done(eb_handle h, int32_t n_env, const float* ego, const float* params, const float* obs,
                  int32_t m_cand, const float* cand, const uint8_t* cand_mode, const float* cand_lw,
                  const uint8_t* v_light, uint8_t* done_code, void* stream) {
    (void)stream;
    if (!h || n_env < 0 || !ego || !params || !obs || m_cand < 0 || (m_cand > 0 && (!cand || !cand_mode)) || !done_code)
        return fail(EB_EINVAL, "eb_judge_done: bad argument");
    const int task = h->cfg.task, D = obs_dim(&h->cfg);
    const float EGO_L = 4.8f, EGO_W = 2.0f;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_env; ++i) {
        const float* e = ego + 6 * (size_t)i;
        const float v_x = e[0], r = e[2], x = e[3], y = e[4], phi = e[5];
        /* Traffic.collision_check, TRF:263-295 */
        float es, ec;
        eb_sincosf(phi / 180.0f * PI_F, &es, &ec);
        const float ego_lw = (EGO_L - EGO_W) / 2;
        float ex0 = x + ec * ego_lw, ey0 = y + es * ego_lw, ex1 = x - ec * ego_lw, ey1 = y - es * ego_lw;
        int collision = 0;
        for (int k = 0; k < m_cand; ++k) {
            if (cand_mode[(size_t)i * m_cand + k] == EB_VMODE_EMPTY) continue;
            const float* v = cand + ((size_t)i * m_cand + k) * 4;
            float vl = cand_lw ? cand_lw[((size_t)i * m_cand + k) * 2] : EGO_L;
            float vw = cand_lw ? cand_lw[((size_t)i * m_cand + k) * 2 + 1] : EGO_W;
            if (fabsf(v[0] - x) < 10.0f && fabsf(v[1] - y) < 10.0f) {
                float s_lw = (vl - vw) / 2;
                float ss_, sc_;
                eb_sincosf(v[3] / 180.0f * PI_F, &ss_, &sc_);
                float sx0 = v[0] + sc_ * s_lw, sy0 = v[1] + ss_ * s_lw, sx1 = v[0] - sc_ * s_lw, sy1 = v[1] - ss_ * s_lw;
                float thr = sq((vw + EGO_W) / 2 + 0.5f);
                if (sq(ex0 - sx0) + sq(ey0 - sy0) < thr) collision = 1;
                else if (sq(ex0 - sx1) + sq(ey0 - sy1) < thr) collision = 1;
                else if (sq(ex1 - sx1) + sq(ey1 - sy1) < thr) collision = 1;
                else if (sq(ex1 - sx0) + sq(ey1 - sy0) < thr) collision = 1;
            }
        }
        /* corner points: rotate_and_shift_coordination(+-l/2, +-w/2, 0, -x, -y, -phi), E2E:171-176, UTL:120-157 */
        float rs, rc_;
        eb_sincosf(-phi * PI_F / 180.0f, &rs, &rc_);
        int feasible = 1;
        for (int q = 0; q < 4; ++q) {
            float cx = (q < 2 ? EGO_L : -EGO_L) / 2, cy = ((q & 1) ? -EGO_W : EGO_W) / 2;
            float tx = cx * rc_ + cy * rs;
            float ty = -cx * rs + cy * rc_;
            float X = tx - (-x), Y = ty - (-y);
            feasible &= judge_feasible(X, Y, task);
        }
        float miu_r = params[4 * (size_t)i + 3];
        float r_bound = miu_r * 9.81f / (fabsf(v_x) + 1e-8f);          /* E2E:167 */
        float delta_y = obs[(size_t)D * i + 6];                         /* E2E:224 */
        int goal;
        if (task == EB_TASK_LEFT) goal = x < -HALF_CROSS - 10.0f && 0.0f < y && y < 3.0f * LANE_W;        /* E2E:251 */
        else if (task == EB_TASK_RIGHT) goal = x > HALF_CROSS + 10.0f && -3.0f * LANE_W < y && y < 0.0f;  /* E2E:253 */
        else goal = y > HALF_CROSS + 10.0f && 0.0f < x && x < 3.0f * LANE_W;                              /* E2E:256 */
        uint8_t code;
        if (collision) code = EB_DONE_COLLISION;                                     /* E2E:208 */
        else if (!feasible) code = EB_DONE_BREAK_ROAD;                               /* E2E:210, 227-229 */
        else if (fabsf(delta_y) > 15.0f) code = EB_DONE_DEVIATE;                     /* E2E:212, 223-225 */
        else if (!(-r_bound < r && r < r_bound)) code = EB_DONE_STABILITY;           /* E2E:214, 239 */
        else if (v_light && v_light[i] != 0 && y > -HALF_CROSS && task != EB_TASK_RIGHT) code = EB_DONE_RED_LIGHT; /* E2E:245 */
        else if (goal) code = EB_DONE_GOOD;
        else code = EB_DONE_NOT_YET;
        done_code[i] = code;
    }
    return EB_OK;
}

/* a15: CrossroadEnd2end._get_ego_dynamics (E2E:150-183) for a batch — the derived entries: the slip-angle bounds (E2E:164-166, with
 * vehicle_params' float64 F_zf / F_zr of DAM:48 rounded to fp32), r_bound (E2E:167) and the four corner points (E2E:171-176 through
 * rotate_and_shift_coordination, UTL:152-157), fp32 in the order eb_judge_done above evaluates them */
int eb_ego_dynamics(eb_handle h, int32_t n, const float* ego, const float* params, float* out, void* stream) {
    (void)stream;
    if (!h || n < 0 || (n > 0 && (!ego || !params || !out))) return fail(EB_EINVAL, "eb_ego_dynamics: bad argument");
    const float EGO_L = 4.8f, EGO_W = 2.0f;
    const float F_zf = (float)(1.46 * 1520.0 * 9.81 / (1.19 + 1.46)), F_zr = (float)(1.19 * 1520.0 * 9.81 / (1.19 + 1.46));   /* DAM:48 */
    for (int i = 0; i < n; ++i) {
        const float* e = ego + 6 * (size_t)i;
        const float v_x = e[0], x = e[3], y = e[4], phi = e[5];
        const float miu_f = params[4 * (size_t)i + 2], miu_r = params[4 * (size_t)i + 3];
        float* o = out + 11 * (size_t)i;
        o[0] = 3.0f * miu_f * F_zf / VP.C_f;                              /* E2E:164-165 */
        o[1] = 3.0f * miu_r * F_zr / VP.C_r;                              /* E2E:166 */
        o[2] = miu_r * 9.81f / (fabsf(v_x) + 1e-8f);                      /* E2E:167 */
        float rs, rc_;
        eb_sincosf(-phi * PI_F / 180.0f, &rs, &rc_);
        for (int q = 0; q < 4; ++q) {                                     /* E2E:171-176 */
            float cx = (q < 2 ? EGO_L : -EGO_L) / 2, cy = ((q & 1) ? -EGO_W : EGO_W) / 2;
            float tx = cx * rc_ + cy * rs;
            float ty = -cx * rs + cy * rc_;
            o[3 + 2 * q] = tx - (-x);
            o[4 + 2 * q] = ty - (-y);
        }
    }
    return EB_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* episodic summary, rollout plans, timing marks (no reference counterpart; include/envbuild.h) */
/* ------------------------------------------------------------------------------------------ */
int eb_episode_summary(eb_handle h, int32_t n_env, int32_t horizon, const float* out5_steps,
                       const float* obs_final, float* out8, void* stream) {
    (void)stream;
    if (!h || n_env < 0 || horizon < 0 || !out8 || (n_env > 0 && horizon > 0 && !out5_steps) || (n_env > 0 && !obs_final))
        return fail(EB_EINVAL, "eb_episode_summary: bad argument");
    const int D = obs_dim(&h->cfg);
    double r = 0, pt = 0, pr = 0, cnt = 0, ady = 0, mdy = 0;
    for (int i = 0; i < n_env; ++i) {
        int any = 0;
        for (int t = 0; t < horizon; ++t) {
            const float* o5 = out5_steps + (size_t)t * 5 * n_env;
            r += (double)o5[i];
            pt += (double)o5[(size_t)n_env + i];
            pr += (double)o5[2 * (size_t)n_env + i];
            any |= o5[2 * (size_t)n_env + i] > 0.0f;
        }
        double dy = (double)fabsf(obs_final[(size_t)i * D + 6]);
        cnt += any;
        ady += dy;
        if (dy > mdy) mdy = dy;
    }
    out8[0] = (float)r; out8[1] = (float)pt; out8[2] = (float)pr; out8[3] = (float)cnt;
    out8[4] = (float)ady; out8[5] = (float)mdy; out8[6] = (float)n_env; out8[7] = (float)horizon;
    return EB_OK;
}

/* The accumulating form (ABI 5; include/envbuild.h): the reference's callers add the returns of rollout_out up step by step
 * (hier_decision.py:96).  Workspace layout of THIS library (private): 8 doubles — sums of reward, punish_term_for_training,
 * real_punish_term, then sum and max of the final rows' |delta_y| — followed by one "punished at any step" byte per env. */
static size_t acc_bytes(const eb_config* c, int32_t n_env, int32_t horizon) {
    int e = 256 / c->n_veh;                    /* the HIP library's smallest tile (its formula: the two must agree) */
    if (e < 1) e = 1;
    if (e > 64) e = 64;
    return (size_t)((n_env + e - 1) / e) * ((size_t)horizon * 4 + 2) * sizeof(double) + (size_t)n_env + 64;
}
int eb_episode_acc_bytes(eb_handle h, int32_t n_env, int32_t horizon, int64_t* bytes) {
    if (!h || n_env < 0 || horizon < 0 || !bytes) return fail(EB_EINVAL, "eb_episode_acc_bytes: bad argument");
    *bytes = (int64_t)acc_bytes(&h->cfg, n_env, horizon);
    return EB_OK;
}
int eb_rollout_step_acc(eb_handle h, int32_t n_env, const float* obs_in, const float* actions,
                        const int32_t* ref_idx, int32_t path_id, float* obs_out, float* out5,
                        float* scaled_actions, void* acc, int32_t step, int32_t horizon, const float* prev_out5,
                        void* stream) {
    if (h && n_env == 0) return EB_OK;
    if (!acc || horizon < 1 || step < 0 || step >= horizon || (step > 0 && !prev_out5) || prev_out5 == out5)
        return fail(EB_EINVAL, "eb_rollout_step_acc: bad argument (0 <= step < horizon; prev_out5 = the previous step's out5 for step > 0)");
    if (((uintptr_t)acc & 15) != 0) return fail(EB_EINVAL, "eb_rollout_step_acc: acc must be 16-byte aligned");
    int rc = eb_rollout_step(h, n_env, obs_in, actions, ref_idx, path_id, obs_out, out5, scaled_actions, stream);
    if (rc) return rc;
    if (acc_bytes(&h->cfg, n_env, horizon) < 64 + (size_t)n_env) return fail(EB_EINVAL, "eb_rollout_step_acc: workspace too small");
    double* a = (double*)acc;
    unsigned char* any = (unsigned char*)acc + 64;
    if (step == 0) { memset(a, 0, 64); memset(any, 0, (size_t)n_env); }
    const int D = obs_dim(&h->cfg), last = step == horizon - 1;
    double r = 0, pt = 0, pr = 0, ady = 0, mdy = 0;
    for (int i = 0; i < n_env; ++i) {
        r += (double)out5[i];
        pt += (double)out5[(size_t)n_env + i];
        pr += (double)out5[2 * (size_t)n_env + i];
        if (out5[2 * (size_t)n_env + i] > 0.0f) any[i] = 1;
        if (last) {
            double dy = (double)fabsf(obs_out[(size_t)i * D + 6]);
            ady += dy;
            if (dy > mdy) mdy = dy;
        }
    }
    a[0] += r; a[1] += pt; a[2] += pr;
    if (last) { a[3] = ady; a[4] = mdy; }
    return EB_OK;
}
int eb_episode_acc_finish(eb_handle h, int32_t n_env, int32_t horizon, const void* acc, float* out8, void* stream) {
    (void)stream;
    if (!h || n_env < 0 || horizon < 1 || !out8 || (n_env > 0 && !acc)) return fail(EB_EINVAL, "eb_episode_acc_finish: bad argument");
    double z[8] = {0};
    const double* a = n_env > 0 ? (const double*)acc : z;
    double cnt = 0;
    for (int i = 0; i < n_env; ++i) cnt += ((const unsigned char*)acc + 64)[i] ? 1.0 : 0.0;
    out8[0] = (float)a[0]; out8[1] = (float)a[1]; out8[2] = (float)a[2]; out8[3] = (float)cnt;
    out8[4] = (float)a[3]; out8[5] = (float)a[4]; out8[6] = (float)n_env; out8[7] = (float)horizon;
    return EB_OK;
}

struct eb_plan_s {
    eb_handle h;
    int32_t n_env, horizon, path_id;
    const float *obs_in, *tape;
    const int32_t* ref_idx;
    float *obs_work, *obs_out, *out5_steps, *summary8;
    void *acc, *own_acc;
};

int eb_plan_create(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in,
                   const float* action_tape, const int32_t* ref_idx, int32_t path_id,
                   float* obs_work, float* obs_out, float* out5_steps, float* summary8, void* acc, eb_plan* out) {
    if (!out) return fail(EB_EINVAL, "eb_plan_create: null argument");
    *out = NULL;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_plan_create: null handle");
    if (rc) return rc;
    if (n_env < 1 || horizon < 1 || !obs_in || !action_tape || !obs_work || !obs_out || !out5_steps)
        return fail(EB_EINVAL, "eb_plan_create: bad argument (n_env >= 1, horizon >= 1, non-null buffers)");
    if (obs_work == obs_out || obs_in == obs_work || obs_in == obs_out)
        return fail(EB_EINVAL, "eb_plan_create: obs_in, obs_work and obs_out must be distinct buffers");
    if (acc && ((uintptr_t)acc & 15) != 0) return fail(EB_EINVAL, "eb_plan_create: acc must be 16-byte aligned");
    eb_plan p = (eb_plan)calloc(1, sizeof *p);
    if (!p) return fail(EB_ENOMEM, "eb_plan_create: out of memory");
    p->h = h; p->n_env = n_env; p->horizon = horizon; p->path_id = path_id; p->obs_in = obs_in;
    p->tape = action_tape; p->ref_idx = ref_idx; p->obs_work = obs_work; p->obs_out = obs_out;
    p->out5_steps = out5_steps; p->summary8 = summary8; p->acc = acc;
    *out = p;
    return EB_OK;
}

int eb_plan_launch(eb_plan p, void* stream) {
    if (!p) return fail(EB_EINVAL, "eb_plan_launch: null plan");
    if (!p->acc) {
        int rc0 = eb_rollout_tape(p->h, p->n_env, p->horizon, p->obs_in, p->tape, p->ref_idx, p->path_id,
                                  p->obs_work, p->obs_out, p->out5_steps, stream);
        if (rc0 == EB_OK && p->summary8)
            rc0 = eb_episode_summary(p->h, p->n_env, p->horizon, p->out5_steps, p->obs_out, p->summary8, stream);
        return rc0;
    }
    /* the accumulating launches, ping-ponging as eb_rollout_tape does */
    const float* cur = p->obs_in;
    for (int t = 0; t < p->horizon; ++t) {
        float* dst = ((p->horizon - 1 - t) % 2 == 0) ? p->obs_out : p->obs_work;
        int rc = eb_rollout_step_acc(p->h, p->n_env, cur, p->tape + (size_t)t * p->n_env * 2, p->ref_idx, p->path_id, dst,
                                     p->out5_steps + (size_t)t * 5 * p->n_env, NULL, p->acc, t, p->horizon,
                                     t > 0 ? p->out5_steps + (size_t)(t - 1) * 5 * p->n_env : NULL, stream);
        if (rc) return rc;
        cur = dst;
    }
    return p->summary8 ? eb_episode_acc_finish(p->h, p->n_env, p->horizon, p->acc, p->summary8, stream) : EB_OK;
}

int eb_plan_destroy(eb_plan p) {
    if (p) free(p->own_acc);
    free(p);
    return EB_OK;
}

struct eb_event_s { struct timespec ts; };

int eb_event_create(eb_handle h, eb_event* out) {
    if (!h || !out) return fail(EB_EINVAL, "eb_event_create: null argument");
    *out = (eb_event)calloc(1, sizeof **out);
    return *out ? EB_OK : fail(EB_ENOMEM, "eb_event_create: out of memory");
}
int eb_event_record(eb_event e, void* stream) {
    (void)stream;
    if (!e) return fail(EB_EINVAL, "eb_event_record: null event");
    clock_gettime(CLOCK_MONOTONIC, &e->ts);
    return EB_OK;
}
int eb_event_elapsed_ms(eb_event start, eb_event stop, float* ms) {
    if (!start || !stop || !ms) return fail(EB_EINVAL, "eb_event_elapsed_ms: null argument");
    *ms = (float)((stop->ts.tv_sec - start->ts.tv_sec) * 1e3 + (stop->ts.tv_nsec - start->ts.tv_nsec) * 1e-6);
    return EB_OK;
}
int eb_event_destroy(eb_event e) {
    free(e);
    return EB_OK;
}

/* a13: CrossroadEnd2end.step as one call, E2E:132-144 */
int eb_env_step(eb_handle h, eb_handle traffic, int32_t n_env, const float* obs, const float* actions,
                const int32_t* ref_idx, int32_t path_id, float* ego, float* params, int32_t m_cand, float* cand,
                const uint8_t* cand_mode, const float* cand_lw, const uint8_t* v_light, const uint8_t* virtual_flag,
                float* scaled_actions, float* out5, float* out_dict16, float* obs_out, uint8_t* done_code,
                const eb_respawn* respawn, const eb_auto_reset* auto_reset, const eb_flow_rule* flow,
                const eb_time_limit* time_limit, void* stream) {
    if (!h || !traffic) return fail(EB_EINVAL, "eb_env_step: null handle");
    if (n_env < 0 || !obs || !actions || !ego || !params || !out5 || !obs_out || !done_code || obs == obs_out ||
        m_cand < 0 || m_cand > 256 || (m_cand > 0 && (!cand || !cand_mode)))
        return fail(EB_EINVAL, "eb_env_step: bad argument");
    if (respawn && (!respawn->entry || !(respawn->limit >= 0.0f) || m_cand < 1 || m_cand > 64))
        return fail(EB_EINVAL, "eb_env_step: bad respawn rule");
    if (traffic->cfg.n_veh != m_cand) return fail(EB_EINVAL, "eb_env_step: the traffic handle must have n_veh == m_cand");
    int rc = check_paths(h, "eb_env_step: null handle");
    if (!rc) rc = check_modes(h);
    if (!rc) rc = check_modes(traffic);
    if (rc) return rc;
    if (!ref_idx && (path_id < 0 || path_id >= h->n_paths)) return fail(EB_EINVAL, "eb_env_step: bad path_id");
    if (auto_reset) {   /* the same checks as the HIP library, before anything is written */
        const eb_auto_reset* ar = auto_reset;
        if (!flow && !ar->pool.entry)
            return fail(EB_EINVAL, "eb_env_step: auto_reset needs a traffic source to reset — the pool rule (auto_reset->pool.entry) or the flow rule of the call");
        if (m_cand < 1 || m_cand > 64) return fail(EB_EINVAL, "eb_env_step: auto_reset needs 1..64 candidates");
        if (!ref_idx || !virtual_flag || ar->ref_idx != ref_idx || ar->virtual_flag != virtual_flag || ar->v_light != v_light)
            return fail(EB_EINVAL, "eb_env_step: auto_reset rewrites the ref_idx / virtual_flag / v_light arrays of the call: they must be given and be the call's own");
        if (flow && (!ar->flow_cand_len || !ar->flow_phase0))
            return fail(EB_EINVAL, "eb_env_step: auto_reset over the flow source needs flow_cand_len and flow_phase0");
        if (ar->final_obs && (ar->final_obs == obs_out || ar->final_obs == obs))
            return fail(EB_EINVAL, "eb_env_step: final_obs must be an array of its own");
    }
    if (flow) {   /* the same checks as the HIP library, before anything is written */
        if (respawn) return fail(EB_EINVAL, "eb_env_step: the flow rule excludes respawn (the pool's rule)");
        if (flow->per_route < 1 || 12 * flow->per_route != m_cand || m_cand > 64 || !flow->active || !flow->timer || !flow->emitted ||
            !flow->sim_step || !flow->lane || !flow->period || !flow->v_max || !v_light || flow->v_light != v_light || flow->cand_mode != cand_mode)
            return fail(EB_EINVAL, "eb_env_step: bad flow rule (m_cand == 12 * per_route <= 64, every array given, cand_mode / v_light the call's own)");
    }
    if (time_limit && (!time_limit->episode_step || time_limit->max_episode_steps < 1))
        return fail(EB_EINVAL, "eb_env_step: bad time limit (episode_step given, max_episode_steps >= 1)");
    if (n_env == 0) return EB_OK;
    float* own_scaled = NULL;
    if (!scaled_actions) {                                                                       /* nullable output */
        own_scaled = (float*)malloc((size_t)n_env * 2 * sizeof(float));
        if (!own_scaled) return fail(EB_ENOMEM, "eb_env_step: out of memory");
        scaled_actions = own_scaled;
    }
    rc = eb_action_transform(h, n_env, actions, scaled_actions, stream);                         /* E2E:133 */
    if (!rc) rc = eb_compute_rewards(h, n_env, obs, scaled_actions, out5, out_dict16, stream);      /* E2E:134 */
    if (!rc) rc = eb_env_ego_step(h, n_env, ego, scaled_actions, ego, params, stream);           /* E2E:135 */
    free(own_scaled);
    if (!rc) rc = eb_veh_predict(traffic, n_env, cand, cand, stream);                            /* TRF:220-238's role */
    if (!rc) rc = eb_get_obs(h, n_env, ego, ref_idx, path_id, m_cand, cand, cand_mode, v_light, virtual_flag, NULL, NULL, obs_out, stream);   /* E2E:140 */
    if (!rc) rc = eb_judge_done(h, n_env, ego, params, obs_out, m_cand, cand, cand_mode, cand_lw, v_light, done_code, stream);   /* E2E:141 */
    if (!rc && time_limit)   /* gym's TimeLimit around the registered env (README.md:55-59): elapsed += 1; elapsed >= max ends an episode
                              * nothing else has ended ('truncated'); the count of a finished env restarts */
        for (int e = 0; e < n_env; ++e) {
            const int cnt = time_limit->episode_step[e] + 1;
            if (done_code[e] == EB_DONE_NOT_YET && cnt >= time_limit->max_episode_steps) done_code[e] = EB_DONE_TIME_LIMIT;
            time_limit->episode_step[e] = done_code[e] != EB_DONE_NOT_YET ? 0 : cnt;
        }
    if (!rc && respawn)   /* the pool's re-entry, after the observation saw this step's state */
        rc = eb_traffic_respawn(traffic, n_env, m_cand, cand, respawn->entry, respawn->limit, respawn->span, respawn->v_max,
                                respawn->seed, respawn->counter, NULL, NULL, NULL, 0.0f, stream);
    if (!rc && flow && auto_reset) {   /* ABI 5, the flow source: its step, then E2E:99-127 for the finished envs with the source's own reset */
        const eb_auto_reset* ar = auto_reset;
        const size_t D = (size_t)obs_dim(&h->cfg);
        rc = eb_traffic_flow_step(traffic, n_env, flow->per_route, cand, flow->active, flow->timer, flow->emitted, flow->sim_step, flow->lane,
                                  flow->period, flow->v_max, flow->dt, flow->exit_range, flow->accel, flow->lane_len, flow->light_cycle,
                                  flow->seed, flow->counter, flow->cand_mode, flow->v_light, stream);
        uint8_t* mask = (uint8_t*)malloc((size_t)n_env * 2);
        if (!mask) return fail(EB_ENOMEM, "eb_env_step: out of memory");
        uint8_t* vnext = mask + n_env;
        for (int e = 0; e < n_env; ++e) {
            mask[e] = done_code[e] != 0;
            if (mask[e] && ar->final_obs) memcpy(ar->final_obs + D * e, obs_out + D * e, D * sizeof(float));
        }
        if (!rc) rc = eb_env_reset(h, n_env, mask, ar->seed, ar->counter, ar->training, ego, params, ar->ref_idx, vnext, NULL, NULL, stream);   /* E2E:100-101 */
        if (!rc) rc = eb_traffic_flow_reset(traffic, n_env, flow->per_route, mask, ego, cand, flow->active, flow->timer, flow->emitted,
                                            flow->sim_step, ar->flow_phase0, flow->lane, flow->period, flow->v_max, ar->flow_cand_len,
                                            flow->lane_len, ar->flow_random_phase, ar->training, ar->flow_seed, ar->flow_counter,
                                            flow->cand_mode, flow->v_light, stream);                                                          /* E2E:102-103 */
        if (!rc) rc = eb_get_obs(h, n_env, ego, ar->ref_idx, 0, m_cand, cand, flow->cand_mode, flow->v_light, ar->virtual_flag, NULL, mask,
                                 obs_out, stream);                                                                                             /* E2E:116 */
        if (!rc)
            for (int e = 0; e < n_env; ++e)
                if (mask[e]) ar->virtual_flag[e] = vnext[e];                                                                                   /* E2E:120-126 */
        free(mask);
        return rc;
    }
    if (!rc && auto_reset) {   /* hier_decision.py:109-135 / E2E:99-127: the envs this step finished start their next episode */
        const eb_auto_reset* ar = auto_reset;
        const size_t D = (size_t)obs_dim(&h->cfg);
        uint8_t* mask = (uint8_t*)malloc((size_t)n_env);
        if (!mask) return fail(EB_ENOMEM, "eb_env_step: out of memory");
        for (int e = 0; e < n_env; ++e) {
            mask[e] = done_code[e] != 0;
            if (mask[e] && ar->final_obs) memcpy(ar->final_obs + D * e, obs_out + D * e, D * sizeof(float));   /* the terminal observation */
        }
        rc = eb_env_reset_pool(h, traffic, n_env, mask, ar->seed, ar->counter, ar->training, ego, params, ar->ref_idx, ar->virtual_flag,
                               ar->v_light, NULL, NULL, m_cand, cand, cand_mode, &ar->pool, obs_out, NULL, NULL, stream);
        free(mask);
    }
    if (!rc && flow)   /* TRF:220-238's role for the flow source: exits, accelerations, emissions, clock and light, after the observation */
        rc = eb_traffic_flow_step(traffic, n_env, flow->per_route, cand, flow->active, flow->timer, flow->emitted, flow->sim_step, flow->lane,
                                  flow->period, flow->v_max, flow->dt, flow->exit_range, flow->accel, flow->lane_len, flow->light_cycle,
                                  flow->seed, flow->counter, flow->cand_mode, flow->v_light, stream);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* helpers for the CPU-baseline leg of bench.py                                                */
/* ------------------------------------------------------------------------------------------ */
int eb_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* exposes the deterministic kernels so tests can bound their error against libm */
void eb_oracle_sincosf(const float* x, float* s, float* c, int n) {
    for (int i = 0; i < n; ++i) eb_sincosf(x[i], s + i, c + i);
}
void eb_oracle_atanf(const float* x, float* y, int n) {
    for (int i = 0; i < n; ++i) y[i] = eb_atanf(x[i]);
}

/* Checks the kernels' constant division (eb_device.h:div_const: (float)((double)x * (1.0 / (double)c)), i.e.
 * v_cvt_f64_f32, v_mul_f64, v_cvt_f32_f64) against IEEE x / c for every float bit pattern in [first, last] with
 * stride `step`.  The same two IEEE operations run here on the CPU.  Returns the number of patterns whose result
 * differs in any bit (NaN dividends are skipped: payloads are the FPU's business); *first_bad gets the first one.
 * form == 1 checks the fp32-only 3-op form (q = x*rc, residual, correction) instead — kept to document why it
 * is not used: it differs for -0.0, +-inf and non-zero magnitudes below 2^-101. */
long long eb_oracle_check_div_exact(float c, uint32_t first, uint32_t last, uint32_t step, int form, uint32_t* first_bad) {
    const float rc = 1.0f / c;
    const double rcd = 1.0 / (double)c;
    long long bad = 0;
    uint32_t fb = 0xffffffffu;
    const uint64_t n = ((uint64_t)last - first) / step + 1;
#ifdef _OPENMP
#pragma omp parallel for reduction(+ : bad) reduction(min : fb) schedule(static)
#endif
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t bits = first + (uint32_t)(i * step);
        float x;
        memcpy(&x, &bits, 4);
        if (x != x) continue;
        float fast;
        if (form == 1) {
            const float q = x * rc;
            fast = fmaf(fmaf(-q, c, x), rc, q);
        } else {
            fast = (float)((double)x * rcd);
        }
        const float exact = x / c;
        uint32_t a, b;
        memcpy(&a, &fast, 4);
        memcpy(&b, &exact, 4);
        if (a != b && !(fast != fast && exact != exact)) {
            ++bad;
            if (bits < fb) fb = bits;
        }
    }
    if (first_bad) *first_bad = fb;
    return bad;
}

/* traffic pool re-entry: the rule stated in include/envbuild.h (not a reference function — the reference's
 * traffic is SUMO); restated here so that the batched env's traffic is reproducible on the CPU too */
static uint64_t eb_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline float eb_u01(uint64_t seed, uint64_t idx) {   /* top 24 bits of splitmix64(seed + GOLDEN * idx) -> [0, 1) */
    return (float)(eb_splitmix64(seed + 0x9E3779B97F4A7C15ull * idx) >> 40) * 5.9604644775390625e-8f;
}

static int init_conflict(const float* ego6, float ego_l, float x, float y, float a, float veh_v, float veh_l);
int eb_traffic_respawn(eb_handle h, int32_t n_env, int32_t m_cand, float* cand, const float* entry, float limit,
                       float span, float v_max, uint64_t seed, uint64_t counter, const uint8_t* env_mask,
                       uint8_t* respawned, const float* ego, float edge_span, void* stream) {
    (void)stream;
    if (!h || n_env < 0 || m_cand < 1 || m_cand > 64 || (n_env > 0 && (!cand || !entry)))
        return fail(EB_EINVAL, "eb_traffic_respawn: bad argument");
    for (int e = 0; e < n_env; ++e)
        for (int j = 0; j < m_cand; ++j) {
            float* c = cand + ((size_t)e * m_cand + j) * 4;
            const int chosen = !env_mask || env_mask[e] != 0;
            const int gone = chosen && (limit < 0.0f || fabsf(c[0]) > limit || fabsf(c[1]) > limit);
            if (gone) {
                const uint64_t base = (counter << 32) + (uint64_t)e * 128u + (uint64_t)j * 2u;
                const float u1 = eb_u01(seed, base), u2 = eb_u01(seed, base + 1);
                const float* en = entry + 5 * j;
                float along = u1 * span;
                c[2] = u2 * v_max;
                c[3] = en[2];
                c[0] = en[0] + along * en[3];
                c[1] = en[1] + along * en[4];
                if (ego && init_conflict(ego + 6 * (size_t)e, 4.8f, c[0], c[1], c[3], c[2], 4.8f)) {   /* TRF:168-192 */
                    along = u1 * edge_span;
                    c[0] = en[0] + along * en[3];
                    c[1] = en[1] + along * en[4];
                }
            }
            if (respawned) respawned[(size_t)e * m_cand + j] = gone ? 1 : 0;
        }
    return EB_OK;
}

/* a18: CrossroadEnd2end.reset + _reset_init_state for the masked envs of a batch (E2E:99-127, 472-499) */
int eb_env_reset(eb_handle h, int32_t n_env, const uint8_t* mask, uint64_t seed, uint64_t counter, int32_t training,
                 float* ego, float* params, int32_t* ref_idx, uint8_t* virtual_next, uint8_t* done_code,
                 int32_t* episode_step, void* stream) {
    (void)stream;
    int rc = check_paths(h, "eb_env_reset: null handle");
    if (rc) return rc;
    if (n_env < 0 || (n_env > 0 && (!ego || !params || !ref_idx))) return fail(EB_EINVAL, "eb_env_reset: bad argument");
    const float span = h->cfg.task == EB_TASK_LEFT ? 900 + 500 : h->cfg.task == EB_TASK_STRAIGHT ? 1200 + 500 : 420 + 500; /* E2E:473-478 */
    for (int e = 0; e < n_env; ++e) {
        if (mask && !mask[e]) continue;
        const uint64_t base = (counter << 32) + (uint64_t)e * 128u;
        const float u0 = eb_u01(seed, base), u1 = eb_u01(seed, base + 1), u2 = eb_u01(seed, base + 2), u3 = eb_u01(seed, base + 3);
        int p = (int)(u0 * (float)h->n_paths);                          /* DAM:591 */
        if (p > h->n_paths - 1) p = h->n_paths - 1;
        const int index = (int)(u1 * span) + 700;                       /* E2E:474-478 */
        const int ci = clamp_index(index, h->lens[p]);                  /* indexs2points, DAM:727-728 */
        float* g = ego + 6 * (size_t)e;
        g[0] = 8.0f * u2; g[1] = 0.0f; g[2] = 0.0f;                     /* E2E:482-486: EXPECTED_V * random */
        g[3] = h->px[p][ci]; g[4] = h->py[p][ci]; g[5] = h->pphi[p][ci];
        float* pr = params + 4 * (size_t)e;
        pr[0] = 0.0f; pr[1] = 0.0f; pr[2] = VP.miu; pr[3] = VP.miu;    /* E2E:110-113 */
        ref_idx[e] = p;
        if (virtual_next) virtual_next[e] = (training && u3 > 0.9f) ? 1 : 0;   /* E2E:120-126 */
        if (done_code) done_code[e] = EB_DONE_NOT_YET;                  /* E2E:119 */
        if (episode_step) episode_step[e] = 0;                          /* eb_time_limit's count (TimeLimit.reset) */
    }
    return EB_OK;
}

int eb_traffic_respawn(eb_handle h, int32_t n_env, int32_t m_cand, float* cand, const float* entry, float limit,
                       float span, float v_max, uint64_t seed, uint64_t counter, const uint8_t* env_mask,
                       uint8_t* respawned, const float* ego, float edge_span, void* stream);
/* the masked reset over the pool as one call: the composition its header comment spells out */
int eb_env_reset_pool(eb_handle h, eb_handle traffic, int32_t n_env, const uint8_t* mask, uint64_t seed, uint64_t counter,
                      int32_t training, float* ego, float* params, int32_t* ref_idx, uint8_t* virtual_flag, uint8_t* v_light,
                      uint8_t* done_code, int32_t* episode_step, int32_t m_cand, float* cand, const uint8_t* cand_mode,
                      const eb_respawn* pool, float* obs, const float* obs_src, const uint8_t* done_src, void* stream) {
    if (!h || !traffic || !pool || !pool->entry) return fail(EB_EINVAL, "eb_env_reset_pool: null argument");
    if (n_env < 0 || m_cand < 1 || m_cand > 64 || (n_env > 0 && (!ego || !params || !ref_idx || !virtual_flag || !cand || !cand_mode || !obs)))
        return fail(EB_EINVAL, "eb_env_reset_pool: bad argument");
    if (mask && (mask == done_code || mask == virtual_flag || mask == v_light))
        return fail(EB_EINVAL, "eb_env_reset_pool: mask must not be one of the arrays the call writes (pass the done codes as mask and done_src, a fresh array as done_code)");
    if (traffic->cfg.n_veh != m_cand) return fail(EB_EINVAL, "eb_env_reset_pool: the traffic handle must have n_veh == m_cand");
    int rc = check_paths(h, "eb_env_reset_pool: null handle");
    if (!rc) rc = check_modes(h);
    if (rc) return rc;
    if (n_env == 0) return EB_OK;
    uint8_t* vnext = (uint8_t*)malloc((size_t)n_env);
    if (!vnext) return fail(EB_ENOMEM, "eb_env_reset_pool: out of memory");
    /* the rows outside the mask: carried over from the caller's previous arrays */
    if (mask && obs_src && obs_src != obs) memcpy(obs, obs_src, (size_t)n_env * (size_t)obs_dim(&h->cfg) * sizeof(float));
    if (mask && done_src && done_code && done_src != done_code) memcpy(done_code, done_src, (size_t)n_env);
    rc = eb_env_reset(h, n_env, mask, seed, counter, training, ego, params, ref_idx, vnext, done_code, episode_step, stream);
    if (!rc) rc = eb_traffic_respawn(traffic, n_env, m_cand, cand, pool->entry, -1.0f, pool->span, pool->v_max, pool->seed,
                                     pool->counter, mask, NULL, ego, pool->edge_span, stream);
    if (!rc && v_light)
        for (int e = 0; e < n_env; ++e)
            if (!mask || mask[e]) v_light[e] = 0;
    if (!rc) rc = eb_get_obs(h, n_env, ego, ref_idx, 0, m_cand, cand, cand_mode, v_light, virtual_flag, NULL, mask, obs, stream);
    if (!rc)
        for (int e = 0; e < n_env; ++e)
            if (!mask || mask[e]) virtual_flag[e] = vnext[e];
    free(vnext);
    return rc;
}

/* Traffic.init_traffic's conflict test, TRF:168-192: the vehicle (x, y, a, speed v, length l) against the ego pose,
 * through shift_and_rotate_coordination (UTL:145-149), fp32 with the deterministic sin / cos */
static void shift_rotate_f32(float x, float y, float d, float sx, float sy, float rd, float* ox, float* oy, float* od) {
    const float hx = x - sx, hy = y - sy;                               /* UTL:116-117 */
    float sn, cs;
    eb_sincosf(rd * PI_F / 180.0f, &sn, &cs);                           /* UTL:130 */
    *ox = hx * cs + hy * sn;                                            /* UTL:131 */
    *oy = -hx * sn + hy * cs;                                           /* UTL:132 */
    float t = d - rd;                                                   /* UTL:133-139 */
    if (!(fabsf(t) <= EB_WRAP_MAX_DEG)) {}
    else if (t > 180.0f) { while (t > 180.0f) t = t - 360.0f; }
    else if (t <= -180.0f) { while (t <= -180.0f) t = t + 360.0f; }
    *od = t;
}
static int init_conflict(const float* ego6, float ego_l, float x, float y, float a, float veh_v, float veh_l) {
    const float ego_v_x = ego6[0], ego_x = ego6[3], ego_y = ego6[4], ego_phi = ego6[5];
    float xe, ye, ae, xv, yv, av;
    shift_rotate_f32(x, y, a, ego_x, ego_y, ego_phi, &xe, &ye, &ae);    /* TRF:177-178 */
    shift_rotate_f32(0.0f, 0.0f, 0.0f, xe, ye, ae, &xv, &yv, &av);      /* TRF:179-182 */
    return (-5.0f < xe && xe < 1.0f * ego_v_x + ego_l / 2.0f + veh_l / 2.0f + 2.0f && fabsf(ye) < 3.0f) ||
           (-5.0f < xv && xv < 1.0f * veh_v + ego_l / 2.0f + veh_l / 2.0f + 2.0f && fabsf(yv) < 3.0f);   /* TRF:183-184 */
}
/* test hook: the predicate alone on n (ego [5] = x, y, phi, v_x, l; veh [5] = x, y, a, v, l) pairs */
void eb_oracle_init_conflict(int n, const float* ego5, const float* veh5, uint8_t* hit) {
    for (int i = 0; i < n; ++i) {
        const float* e = ego5 + 5 * (size_t)i;
        const float* v = veh5 + 5 * (size_t)i;
        const float ego6[6] = {e[3], 0, 0, e[0], e[1], e[2]};
        hit[i] = (uint8_t)init_conflict(ego6, e[4], v[0], v[1], v[2], v[3], v[4]);
    }
}

int eb_traffic_flow_reset(eb_handle h, int32_t n_env, int32_t per_route, const uint8_t* mask, const float* ego,
                          float* cand, uint8_t* active, float* timer, int32_t* emitted, int32_t* sim_step,
                          uint8_t* phase0, const float* lane, const float* period, const float* v_max,
                          const float* cand_len, float lane_len, int32_t random_phase, int32_t training,
                          uint64_t seed, uint64_t counter, uint8_t* cand_mode, uint8_t* v_light, void* stream) {
    (void)stream;
    if (!h || n_env < 0 || per_route < 1 || per_route * 12 > 64 ||
        (n_env > 0 && (!ego || !cand || !active || !timer || !emitted || !sim_step || !phase0 || !lane || !period || !v_max ||
                       !cand_len || !cand_mode || !v_light)))
        return fail(EB_EINVAL, "eb_traffic_flow_reset: bad argument");
    const int K = per_route, M = 12 * K;
    for (int e = 0; e < n_env; ++e) {
        if (mask && !mask[e]) continue;
        const uint64_t env_base = (counter << 32) + (uint64_t)e * 256u;
        for (int r = 0; r < 12; ++r) {
            float expect = lane_len / 7.5f / period[r];
            if (expect > (float)K) expect = (float)K;
            const float p = expect / (float)K;
            for (int k = 0; k < K; ++k) {
                const int j = r * K + k;
                const size_t s = (size_t)e * M + j;
                const float u0 = eb_u01(seed, env_base + 4u * j), u1 = eb_u01(seed, env_base + 4u * j + 1),
                            u2 = eb_u01(seed, env_base + 4u * j + 2);
                int on = u0 < p;
                if (on) {
                    const float* ln = lane + 5 * j;
                    const float along = u1 * lane_len;
                    float* c = cand + s * 4;
                    c[0] = ln[0] + along * ln[3];
                    c[1] = ln[1] + along * ln[4];
                    c[2] = u2 * v_max[j];
                    c[3] = ln[2];
                    if (init_conflict(ego + 6 * (size_t)e, 4.8f, c[0], c[1], c[3], c[2], cand_len[j])) on = 0;
                }
                active[s] = (uint8_t)on;
                cand_mode[s] = on ? (uint8_t)r : (uint8_t)EB_VMODE_EMPTY;
            }
            timer[(size_t)e * 12 + r] = eb_u01(seed, env_base + 4u * (r * K) + 3) * period[r];
            emitted[(size_t)e * 12 + r] = 0;
        }
        sim_step[e] = 0;
        const uint8_t ph = (random_phase && eb_u01(seed, env_base + 255u) > 0.5f) ? 2 : 0;   /* TRF:158-161 */
        phase0[e] = ph;
        v_light[e] = training ? ph : 0;                                                       /* TRF:222-223 */
    }
    return EB_OK;
}

/* diagnostics of the HIP library: accepted and ignored here */
int eb_debug_set_tile(eb_handle h, int32_t variant) { (void)variant; return h ? EB_OK : fail(EB_EINVAL, "eb_debug_set_tile: null handle"); }
int eb_debug_set_tape_stepwise(eb_handle h, int32_t on) { (void)on; return h ? EB_OK : fail(EB_EINVAL, "eb_debug_set_tape_stepwise: null handle"); }
int eb_debug_set_trace(eb_handle h, long long* device_buf, int64_t capacity_words) { (void)device_buf; (void)capacity_words; return h ? EB_OK : fail(EB_EINVAL, "eb_debug_set_trace: null handle"); }
int eb_debug_set_stage_paths(eb_handle h, int32_t mode) { (void)mode; return h ? EB_OK : fail(EB_EINVAL, "eb_debug_set_stage_paths: null handle"); }
int eb_debug_set_scan_prefetch(eb_handle h, int32_t on) { (void)on; return h ? EB_OK : fail(EB_EINVAL, "eb_debug_set_scan_prefetch: null handle"); }
int eb_debug_rollout_plan(eb_handle h, int32_t n_env, int32_t* out4) {
    (void)h; (void)n_env; (void)out4;
    return fail(EB_EINVAL, "eb_debug_rollout_plan: the CPU library launches nothing");
}
int eb_debug_set_rollout_sched(eb_handle h, int32_t rolling, int32_t by_progress) {
    if (!h || rolling < -1 || rolling > 1 || by_progress < -1 || by_progress > 1)
        return fail(EB_EINVAL, "eb_debug_set_rollout_sched: bad argument (-1 = by grid size, 0, 1)");
    return EB_OK;
}
int eb_debug_check_grids(eb_handle h, int32_t samples_per_cell, uint64_t seed, int64_t* n_checked, int64_t* n_bad) {   /* the oracle's closest point IS the full scan: no levels to check */
    (void)samples_per_cell; (void)seed;
    if (!h || !n_checked || !n_bad) return fail(EB_EINVAL, "eb_debug_check_grids: bad argument");
    *n_checked = 0; *n_bad = 0;
    return EB_OK;
}
int eb_debug_set_env_waves(eb_handle h, int32_t waves) { (void)waves; return h ? EB_OK : fail(EB_EINVAL, "eb_debug_set_env_waves: null handle"); }

int eb_traffic_flow_step(eb_handle h, int32_t n_env, int32_t per_route, float* cand, uint8_t* active, float* timer,
                         int32_t* emitted, int32_t* sim_step, const float* lane, const float* period,
                         const float* v_max, float dt, float exit_range, float accel, float lane_len,
                         int32_t light_cycle, uint64_t seed, uint64_t counter, uint8_t* cand_mode, uint8_t* v_light,
                         void* stream) {
    (void)stream;
    if (!h || n_env < 0 || per_route < 1 || per_route * 12 > 64 || !(dt > 0.0f) ||
        (n_env > 0 && (!cand || !active || !timer || !emitted || !sim_step || !lane || !period || !v_max || !cand_mode || !v_light)))
        return fail(EB_EINVAL, "eb_traffic_flow_step: bad argument");
    const int K = per_route, M = 12 * K;
    for (int e = 0; e < n_env; ++e) {
        for (int r = 0; r < 12; ++r) {
            const size_t s0 = (size_t)e * M + (size_t)r * K;
            int vacant = -1;
            for (int k = 0; k < K; ++k) {
                float* c = cand + (s0 + k) * 4;
                int on = active[s0 + k] != 0;
                if (on) {
                    float sn, cs;
                    eb_sincosf(deg2rad(c[3]), &sn, &cs);
                    const int outward = c[0] * cs + c[1] * sn > 0.0f;
                    if (fmaxf(fabsf(c[0]), fabsf(c[1])) > exit_range && outward) {
                        on = 0;
                    } else {
                        const float vn = c[2] + accel * dt, vm = v_max[r * K + k];
                        c[2] = vn < vm ? vn : vm;
                    }
                }
                active[s0 + k] = (uint8_t)on;
                if (!on && vacant < 0) vacant = k;
            }
            const size_t idx = (size_t)e * 12 + r;
            float t = timer[idx] + dt;
            const float per = period[r];
            if (t >= per && vacant >= 0) {
                const int j = r * K + vacant;
                const uint64_t base = (counter << 32) + (uint64_t)e * 128u + (uint64_t)(r * K) * 2u;
                const float u1 = (float)(eb_splitmix64(seed + 0x9E3779B97F4A7C15ull * base) >> 40) * 5.9604644775390625e-8f;
                const float u2 = (float)(eb_splitmix64(seed + 0x9E3779B97F4A7C15ull * (base + 1)) >> 40) * 5.9604644775390625e-8f;
                const float* ln = lane + 5 * j;
                const float along = u1 * lane_len;
                float* c = cand + (s0 + vacant) * 4;
                c[0] = ln[0] + along * ln[3];
                c[1] = ln[1] + along * ln[4];
                c[2] = u2 * v_max[j];
                c[3] = ln[2];
                active[s0 + vacant] = 1;
                t = t - per;
                emitted[idx] += 1;
            }
            timer[idx] = t;
            for (int k = 0; k < K; ++k) cand_mode[s0 + k] = active[s0 + k] ? (uint8_t)r : (uint8_t)EB_VMODE_EMPTY;
        }
        const int n = sim_step[e] + 1;
        sim_step[e] = n;
        if (light_cycle) { /* a.net.xml:145-150 */
            const float tt = (float)(n % (int)(60.0f / dt + 0.5f)) * dt;
            v_light[e] = tt < 25.0f ? 0 : (tt < 30.0f ? 1 : (tt < 55.0f ? 2 : 3));
        }
    }
    return EB_OK;
}

/* ================================================================================================
 * Policy network in the loop — TEST INFRASTRUCTURE like the rest of this file.
 * MLPNet (utils/model.py:18-43), the 'scale' preprocessor (utils/preprocessor.py:116-123), the deterministic
 * action of Policy4Toyota.compute_action (utils/policy.py:85-92) and the shield loop `is_safe`
 * (hierarchical_decision/hier_decision.py:89-97, multi_env/multi_ego.py:187-197).
 * TensorFlow's matmul / exp / tanh kernels are third-party code absent from /root/reference (unpinned,
 * README.md:37) and leave the summation order unspecified; the contract stated in include/envbuild.h fixes it:
 * every output is a chain of fused multiply-adds over k = 0, 1, ... starting from the bias.  Pinning: the reference
 * holds no weights or test vectors for the network; fixture G13 (oracle/gen_golden.py:g13_policy) runs the reference's
 * own MLPNet / Policy4Toyota / Preprocessor / LoadPolicy.run_batch classes over a NumPy stand-in of the tf.keras slice
 * they use, on seeded weights, and tests/test_policy_oracle.py holds this file to those outputs (rtol 1e-5, atol 5e-6;
 * observed excess 2.5e-7); at TensorFlow's own matmul / exp / tanh kernels parity stays UNPINNED, as for the path.
 * tests/ also check it against torch fp32 within 1e-5 and the HIP kernel against this file bit for bit.
 * ================================================================================================ */
struct eb_mlp_s {
    eb_mlp_config cfg;
    float* w[EB_MLP_MAX_HIDDEN + 1]; /* [in, out] row-major, as given */
    float* b[EB_MLP_MAX_HIDDEN + 1];
    float* scale;
    int has_scale;
    unsigned layers_set;
};

static float eb_expf(float x) { /* Cephes expf scheme with explicit fma; device twin: eb_policy.hip:exp_det */
    if (!(x == x)) return x;
    x = x > 88.0f ? 88.0f : (x < -87.0f ? -87.0f : x);
    const float fx = rintf(x * 1.44269504088896341f);
    float r = fmaf(-fx, 0.693359375f, x);
    r = fmaf(-fx, -2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float y = fmaf(p, z, r) + 1.0f;
    const int n = (int)fx;
    const uint32_t bits = (uint32_t)(n + 127) << 23;
    float s;
    memcpy(&s, &bits, 4);
    return y * s;
}

static float eb_tanhf(float x) { /* Cephes tanhf scheme; device twin: tanh_det */
    const float ax = fabsf(x);
    if (ax > 44.0f) return x > 0.0f ? 1.0f : -1.0f;
    if (ax >= 0.625f) {
        const float s = eb_expf(ax + ax);
        const float t = 1.0f - 2.0f / (s + 1.0f);
        return x < 0.0f ? -t : t;
    }
    const float z = x * x;
    float p = -5.70498872745e-3f;
    p = fmaf(p, z, 2.06390887954e-2f);
    p = fmaf(p, z, -5.37397155531e-2f);
    p = fmaf(p, z, 1.33314422036e-1f);
    p = fmaf(p, z, -3.33332819422e-1f);
    return fmaf(p * z, x, x);
}

static float mlp_activate(int act, float x) {
    switch (act) {
        case EB_ACT_RELU: return x > 0.0f ? x : 0.0f;
        case EB_ACT_ELU: return x > 0.0f ? x : eb_expf(x) - 1.0f;
        case EB_ACT_TANH: return eb_tanhf(x);
        default: return x;
    }
}

static void mlp_layer_dims(const struct eb_mlp_s* m, int layer, int* k, int* cols) {
    *k = layer == 0 ? m->cfg.obs_dim : m->cfg.n_units;
    *cols = layer == m->cfg.n_hidden ? m->cfg.out_dim : m->cfg.n_units;
}

int eb_mlp_create(const eb_mlp_config* cfg, eb_mlp* out) {
    if (!cfg || !out) return fail(EB_EINVAL, "eb_mlp_create: null argument");
    if (cfg->abi_version != EB_ABI_VERSION) return fail(EB_EINVAL, "eb_mlp_create: ABI version mismatch");
    if (cfg->obs_dim < 1 || cfg->n_hidden < 1 || cfg->n_hidden > EB_MLP_MAX_HIDDEN || cfg->n_units < 1 ||
        cfg->n_units > EB_MLP_MAX_UNITS || cfg->out_dim < 1 || cfg->out_dim > 32)
        return fail(EB_EINVAL, "eb_mlp_create: bad dimensions");
    if (cfg->hidden_act < EB_ACT_LINEAR || cfg->hidden_act > EB_ACT_TANH || cfg->out_act < EB_ACT_LINEAR ||
        cfg->out_act > EB_ACT_TANH)
        return fail(EB_EINVAL, "eb_mlp_create: unknown activation");
    struct eb_mlp_s* m = (struct eb_mlp_s*)calloc(1, sizeof *m);
    if (!m) return fail(EB_ENOMEM, "eb_mlp_create: out of memory");
    m->cfg = *cfg;
    for (int L = 0; L <= cfg->n_hidden; ++L) {
        int k, cols;
        mlp_layer_dims(m, L, &k, &cols);
        m->w[L] = (float*)calloc((size_t)k * cols, sizeof(float));
        m->b[L] = (float*)calloc((size_t)cols, sizeof(float));
        if (!m->w[L] || !m->b[L]) { eb_mlp_destroy(m); return fail(EB_ENOMEM, "eb_mlp_create: out of memory"); }
    }
    m->scale = (float*)calloc((size_t)cfg->obs_dim, sizeof(float));
    if (!m->scale) { eb_mlp_destroy(m); return fail(EB_ENOMEM, "eb_mlp_create: out of memory"); }
    *out = m;
    return EB_OK;
}

int eb_mlp_destroy(eb_mlp m) {
    if (!m) return EB_OK;
    for (int L = 0; L <= EB_MLP_MAX_HIDDEN; ++L) { free(m->w[L]); free(m->b[L]); }
    free(m->scale);
    free(m);
    return EB_OK;
}

int eb_mlp_set_layer(eb_mlp m, int32_t layer, const float* kernel, const float* bias) {
    if (!m || !kernel || !bias || layer < 0 || layer > m->cfg.n_hidden) return fail(EB_EINVAL, "eb_mlp_set_layer: bad argument");
    int k, cols;
    mlp_layer_dims(m, layer, &k, &cols);
    memcpy(m->w[layer], kernel, sizeof(float) * (size_t)k * cols);
    memcpy(m->b[layer], bias, sizeof(float) * (size_t)cols);
    m->layers_set |= 1u << layer;
    return EB_OK;
}

int eb_mlp_set_obs_scale(eb_mlp m, const float* scale) {
    if (!m) return fail(EB_EINVAL, "eb_mlp_set_obs_scale: null handle");
    m->has_scale = scale != NULL;
    if (scale) memcpy(m->scale, scale, sizeof(float) * (size_t)m->cfg.obs_dim);
    return EB_OK;
}

/* one observation through the network; `a` and `b` are scratch rows of max(obs_dim, n_units, out_dim) floats */
static void mlp_row(const struct eb_mlp_s* m, const float* obs, float* a, float* b, float* logits) {
    for (int k = 0; k < m->cfg.obs_dim; ++k) a[k] = m->has_scale ? obs[k] * m->scale[k] : obs[k]; /* preprocessor.py:121 */
    for (int L = 0; L <= m->cfg.n_hidden; ++L) {
        int K, cols;
        mlp_layer_dims(m, L, &K, &cols);
        const float* w = m->w[L];
        for (int j = 0; j < cols; ++j) b[j] = m->b[L][j];
        for (int k = 0; k < K; ++k) { /* the chain: k in order, every output j takes one fused multiply-add per k */
            const float x = a[k];
            const float* wk = w + (size_t)k * cols;
#pragma omp simd
            for (int j = 0; j < cols; ++j) b[j] = fmaf(x, wk[j], b[j]);
        }
        const int act = L == m->cfg.n_hidden ? m->cfg.out_act : m->cfg.hidden_act;
        for (int j = 0; j < cols; ++j) b[j] = mlp_activate(act, b[j]);
        float* t = a; a = b; b = t;
    }
    for (int j = 0; j < m->cfg.out_dim; ++j) logits[j] = a[j];
}

static int mlp_run(eb_mlp m, int32_t n, const float* obs, float* out, int action_head, float action_range, const char* who) {
    if (!m) return fail(EB_EINVAL, who);
    if (n < 0 || (n > 0 && (!obs || !out))) return fail(EB_EINVAL, "eb_mlp: bad argument");
    if (m->layers_set != (1u << (m->cfg.n_hidden + 1)) - 1u)
        return fail(EB_ESTATE, "eb_mlp: eb_mlp_set_layer has not been called for every layer");
    if (action_head && (m->cfg.out_dim < 2 || (m->cfg.out_dim & 1))) return fail(EB_EINVAL, "eb_policy_run_batch: out_dim must be 2 * act_dim");
    int width = m->cfg.obs_dim > m->cfg.n_units ? m->cfg.obs_dim : m->cfg.n_units;
    if (width < m->cfg.out_dim) width = m->cfg.out_dim;
    const int od = m->cfg.out_dim, ad = od / 2;
    int oom = 0;
#ifdef _OPENMP
#pragma omp parallel
#endif
    {
        float* a = (float*)malloc(sizeof(float) * (size_t)width * 2 + sizeof(float) * 32);
        if (!a) {
#ifdef _OPENMP
#pragma omp atomic write
#endif
            oom = 1;
        }
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int i = 0; i < n; ++i) {
            if (!a) continue;
            float* logits = a + 2 * (size_t)width;
            mlp_row(m, obs + (size_t)i * m->cfg.obs_dim, a, a + width, logits);
            if (!action_head) {
                for (int j = 0; j < od; ++j) out[(size_t)i * od + j] = logits[j];
            } else { /* mean, log_std = split(logits); action_range * tanh(mean), policy.py:89-92 */
                for (int j = 0; j < ad; ++j)
                    out[(size_t)i * ad + j] = action_range > 0.0f ? action_range * eb_tanhf(logits[j]) : logits[j];
            }
        }
        free(a);
    }
    return oom ? fail(EB_ENOMEM, "eb_mlp: out of memory") : EB_OK;
}

int eb_mlp_forward(eb_mlp m, int32_t n, const float* obs, float* out, void* stream) {
    (void)stream;
    return mlp_run(m, n, obs, out, 0, 0.0f, "eb_mlp_forward: null handle");
}

int eb_policy_run_batch(eb_mlp m, int32_t n, const float* obs, float action_range, float* actions, void* stream) {
    (void)stream;
    return mlp_run(m, n, obs, actions, 1, action_range, "eb_policy_run_batch: null handle");
}

int eb_shield_is_safe(eb_handle h, eb_mlp policy, int32_t n_env, const float* obs_in, const int32_t* ref_idx,
                      int32_t path_id, int32_t steps, int32_t penalty, float action_range, float* obs_a,
                      float* obs_b, float* actions, float* out5, float* punish, uint8_t* safe, void* stream) {
    (void)stream;
    if (h && policy && n_env == 0) return EB_OK;
    int rc = check_rollout(h, n_env, ref_idx, path_id, "eb_shield_is_safe: null handle");
    if (rc) return rc;
    if (!policy) return fail(EB_EINVAL, "eb_shield_is_safe: null policy");
    if (n_env < 0 || steps < 1 || !obs_in || !obs_a || !obs_b || !actions || !out5 || !punish || !safe ||
        obs_a == obs_b || obs_in == obs_a || obs_in == obs_b)
        return fail(EB_EINVAL, "eb_shield_is_safe: bad argument");
    if (penalty != EB_PENALTY_VEH2VEH4REAL && penalty != EB_PENALTY_REAL_PUNISH_TERM)
        return fail(EB_EINVAL, "eb_shield_is_safe: unknown penalty");
    if (policy->cfg.obs_dim != obs_dim(&h->cfg) || policy->cfg.out_dim != 4)
        return fail(EB_EINVAL, "eb_shield_is_safe: the policy does not fit the model (obs_dim, out_dim = 4)");
    const float* pen = out5 + (size_t)(penalty == EB_PENALTY_VEH2VEH4REAL ? 3 : 2) * n_env;
    const float* cur = obs_in;
    for (int t = 0; t < steps; ++t) { /* hier_decision.py:92-96 */
        float* dst = (t & 1) ? obs_b : obs_a;
        rc = eb_policy_run_batch(policy, n_env, cur, action_range, actions, NULL);   /* action = self.policy.run_batch(obs) */
        if (rc) return rc;
        rc = eb_rollout_step(h, n_env, cur, actions, ref_idx, path_id, dst, out5, NULL, NULL); /* obs, ... = rollout_out(action) */
        if (rc) return rc;
        for (int i = 0; i < n_env; ++i) punish[i] = (t == 0 ? 0.0f : punish[i]) + pen[i];      /* punish += veh2veh4real */
        cur = dst;
    }
    for (int i = 0; i < n_env; ++i) safe[i] = punish[i] > 0.0f ? 0 : 1;                          /* False if punish > 0 */
    return EB_OK;
}
