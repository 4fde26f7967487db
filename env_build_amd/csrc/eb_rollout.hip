// eb_rollout.hip — K5, the fused rollout step (EnvironmentModel.rollout_out, DAM:118-126) for gfx950.
//
// One launch per step.  A block owns a tile of E whole envs (E <= 64, E * n_veh <= RW * 64 * RPT) and
// runs 1 + RW waves with two roles:
//
//   env wave (wave 0, one lane per env) — the per-env chain, ~450 VALU ops on 36 + 36 bytes:
//     load the 9-word head (ego 6 + tracking 3), the action and the path id; sin/cos of the ego heading;
//     put (x, y, sin, cos) of the CURRENT pose into LDS for the record waves             -- hand-off 1 --
//     reward terms (DAM:198-207, 297-298), bicycle-model step (DAM:386-392), closest point of the NEXT
//     pose through the cell grid (tables in L2), tracking error (DAM:334-353, 735-770), head store
//                                                                                        -- hand-off 2 --
//     per-env penalty sums in vehicle order (DAM:218), road walls (DAM:231-295), the four penalty outputs.
//
//   record waves (waves 1..RW, one lane per (env, vehicle) record, RPT records per lane; every 16-byte
//   record load is issued up front, consecutive lanes on consecutive records -> coalesced HBM streams):
//     per record: predict (DAM:405-427), store — nothing here depends on the env wave    -- hand-off 1 --
//     per record: centre distance to the ego from LDS; records inside 6.364 m are pushed on the wave's
//     own LDS queue (ballot prefix, no atomics) — every other record adds exact zeros to the penalty
//     sums (DAM:228-229).  Then the queue, compacted one record per lane: four circle-pair distances
//     (DAM:218-229) -> per-record partial sums + a bit in the env's 64-bit slot mask     -- hand-off 2 --
//
// The hand-offs are LDS flags (see lds_publish / lds_wait_until), not barriers: neither role ever waits for
// the other's HBM traffic.  HBM-bound: 104 + 32 n_veh algorithmic bytes per env-step; no MFMA (nothing here
// is a dense contraction).
//
// What a launch's first microseconds are (round 6, per-wave entry marks — scripts/trace_rollout.py, profiles/r6_ab2.txt): every
// one of the 5 120 waves of the headline grid is INSIDE the kernel within 0.5 us; what earlier rounds read as a 3.7 us dispatch ramp
// is the read phase itself — a CU's miss path moves ~11 bytes per cycle (MI355X_MICROARCH.md), its 144 KB of rows take 5.5 us, and a
// wave's load instructions issue as the queue ahead of them drains.  Hence: nothing may stand between a wave's entry and its first
// load (the slot-turn table's address arrives as a preloaded argument, the entry barrier comes AFTER the loads are out); the record
// waves that are BEHIND issue first (s_setprio by progress: the last wave, not the average one, ends a launch: 14.85 -> 14.0 us at
// 65 536 x 32); a grid of at most three tiles per CU — and any grid of 32-env tiles — keeps three record loads in flight per lane
// and requests record k + 3 when record k is done (every wave's first records come back sooner, the block's life shrinks: 9.3 ->
// 8.5 us per step at 32 768 x 32, 25.7 -> 24.5 at 65 536 x 64).  Which grid gets what: eb_capi.hip:rollout_fused, from the sweeps of
// profiles/r6_sched_sweep1-2.txt.
#include <type_traits>

#include "eb_device.h"
#include "eb_kernels.h"

#pragma clang fp contract(off)

namespace eb {

typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float v2f __attribute__((ext_vector_type(2)));                 // register pair for v_pk_*_f32
EB_DEV v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }

// Storage of the obs rows: fp32 (the reference's layout) or IEEE binary16 (BASELINE configs[4]: state stored in
// fp16, every operation and the reward accumulation in fp32, results rounded to nearest-even on the store).
template <typename ST> struct Stored;
template <> struct Stored<float> {
    static EB_DEV f4u load4(const float* p) { return *reinterpret_cast<const f4u*>(p); }
    static EB_DEV float load1(const float* p) { return *p; }
    static EB_DEV void store4(float* p, f4u v) { *reinterpret_cast<f4u*>(p) = v; }
    static EB_DEV void store1(float* p, float v) { *p = v; }
    static EB_DEV float round(float v) { return v; }                    // what a store + load does to a value
    // write-through forms (sc1: the bytes leave this XCD's L2 at once, so another kernel can read them while this one
    // runs; the issuing wave drains them with s_waitcnt vmcnt(0) before it raises a flag)
    // (the s_nop: a store of more than 64 bits reads its data registers a little after it issues, and the compiler's hazard
    // check — which would keep the next instruction from overwriting them — does not look inside an asm statement)
    static EB_DEV void store4_wt(float* p, f4u v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
    static EB_DEV void store1_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
};
typedef _Float16 h4u __attribute__((ext_vector_type(4), aligned(2)));   // 8-byte access, 2-byte aligned
template <> struct Stored<_Float16> {
    static EB_DEV f4u load4(const _Float16* p) {
        const h4u h = *reinterpret_cast<const h4u*>(p);
        return f4u{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    }
    static EB_DEV float load1(const _Float16* p) { return (float)*p; }
    static EB_DEV void store4(_Float16* p, f4u v) {
        *reinterpret_cast<h4u*>(p) = h4u{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    }
    static EB_DEV void store1(_Float16* p, float v) { *p = (_Float16)v; }
    static EB_DEV float round(float v) { return (float)(_Float16)v; }
    static EB_DEV void store4_wt(_Float16* p, f4u v) {
        typedef unsigned u2v __attribute__((ext_vector_type(2)));
        const u2v bits = __builtin_bit_cast(u2v, h4u{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w});
        asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(p), "v"(bits) : "memory");
    }
    static EB_DEV void store1_wt(_Float16* p, float v) {
        const unsigned bits = __builtin_bit_cast(unsigned short, (_Float16)v);
        asm volatile("global_store_short %0, %1, off sc1" :: "v"(p), "v"(bits) : "memory");
    }
};

// LDS-only workgroup barrier: orders this wave's LDS traffic, leaves global loads/stores in flight
EB_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// The two hand-offs between the roles are LDS flags, not barriers: a barrier would hold the env wave until
// every record wave has its HBM data, and the record waves until the env wave's chain is through.
// (All waves of a block are resident together, so polling cannot deadlock.)
// (The flags are accessed through LDS-typed pointers: a volatile access through a generic pointer stays a FLAT
// instruction — the address-space inference pass leaves volatile operations alone — which is slower to poll.)
typedef __attribute__((address_space(3))) int lds_int;
EB_DEV void lds_publish(int* flag, int value) {   // everything this wave wrote to LDS before is visible first
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *(volatile lds_int*)flag = value;
}
EB_DEV void lds_wait_until(int* flag, int value) {
    while (*(volatile lds_int*)flag != value) __builtin_amdgcn_s_sleep(2);
    asm volatile("" ::: "memory");
}

// Wave-wide float64 sum on the DPP network (no LDS traffic), complete in lane 63: an inclusive scan inside each row of 16
// lanes (row_shr 1, 2, 4, 8; lanes without a source add 0), then the row totals forwarded (row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3).  Every lane must be active.  A fixed order: the episodic accumulator stays deterministic.
template <int CTRL, int ROW_MASK>
EB_DEV double dpp_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
EB_DEV double wave_sum_f64(double v) {
    v += dpp_f64<0x111, 0xf>(v);
    v += dpp_f64<0x112, 0xf>(v);
    v += dpp_f64<0x114, 0xf>(v);
    v += dpp_f64<0x118, 0xf>(v);
    v += dpp_f64<0x142, 0xa>(v);
    v += dpp_f64<0x143, 0xc>(v);
    return v;
}
// the same network for the maximum of non-negative, non-NaN floats (0 is the identity the missing lanes supply)
template <int CTRL, int ROW_MASK>
EB_DEV float dpp_max_f32(float v) {
    const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
    return o > v ? o : v;
}
EB_DEV float wave_max_f32(float v) {
    v = dpp_max_f32<0x111, 0xf>(v);
    v = dpp_max_f32<0x112, 0xf>(v);
    v = dpp_max_f32<0x114, 0xf>(v);
    v = dpp_max_f32<0x118, 0xf>(v);
    v = dpp_max_f32<0x142, 0xa>(v);
    v = dpp_max_f32<0x143, 0xc>(v);
    return v;
}

// One step's record of the episodic accumulator (eb_rollout_step_acc): the tile's three sums (float64 on the DPP network, a fixed
// order) and its "punished in this step" bits — 32 bytes from lane 63, no read-modify-write.  lane = env of the tile; every lane active.
EB_DEV void acc_record(double* recs, bool act, int lane, float v_r, float v_t, float v_p) {
    const double s_r = wave_sum_f64(act ? (double)v_r : 0.0);
    const double s_t = wave_sum_f64(act ? (double)v_t : 0.0);
    const double s_p = wave_sum_f64(act ? (double)v_p : 0.0);
    const unsigned long long any = __builtin_amdgcn_ballot_w64(act && v_p > 0.0f);
    if (lane == 63) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v* rec = reinterpret_cast<d2v*>(recs + (size_t)ACC_RECORD_DOUBLES * blockIdx.x);
        rec[0] = d2v{s_r, s_t};
        rec[1] = d2v{s_p, __builtin_bit_cast(double, any)};
    }
}

// Device-scope accesses of the gated rollout's flags and action words, spelled as global instructions with the sc1 bit
// (served by L2 / memory, never by this CU's L1 — MI355X_MICROARCH.md, "inter-workgroup visibility").  Inline assembly:
// the flat-pointer forms of the __hip_atomic builtins do not survive instruction selection here.
EB_DEV unsigned agent_load_u32(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
EB_DEV unsigned long long agent_load_u64(const void* p) {
    unsigned long long v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
EB_DEV void agent_store_u32(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
EB_DEV void agent_add_u32(unsigned* p, unsigned v) { asm volatile("global_atomic_add %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }

// The arguments every wave needs before it can issue its first HBM load travel as separate kernel parameters:
// with -mllvm -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave instead of behind an s_load.
// (The same-named members of FusedArgs are only read on the host side.)
template <typename ST>
struct FusedHot {
    const ST* obs_in;
    ST* obs_out;
    int n_env, obs_dim, n_veh, envs_per_tile;
    unsigned nv_magic;
    int do_rewards;                       // FusedArgs::do_rewards | HOT_PRIO (host: launch_rollout_fused)
    double* acc_rec;                      // episodic accumulator: this step's records, or NULL (FusedArgs::acc_rec)
    const unsigned char* turn;            // PathTables::turn of the handle's tables: the record waves' first load needs no s_load
    long long t_entry;                    // wall clock at the wave's first instruction (trace only)
};
constexpr int HOT_REWARDS = 1, HOT_PRIO = 2;   // bits of FusedHot::do_rewards

// item / n_veh by the multiplicative inverse nv_magic = ceil(2^32 / n_veh) (exact for item < 65 536, checked; items stay
// below 2048); n_veh == 1 has no 32-bit inverse and is flagged by nv_magic == 0
template <typename ST>
EB_DEV int env_of_item(const FusedHot<ST>& H, int item) {
    return H.nv_magic ? (int)__umulhi((unsigned)item, H.nv_magic) : item;
}

// profiling aid: mark slot `i` of this wave's trace row with the 100 MHz wall clock (lane 0 only)
#define EB_MARK(A, row, i) do { if ((A).trace && (threadIdx.x & 63) == 0) (A).trace[(size_t)(row) * 8 + (i)] = wall_clock64(); } while (0)
// slot 7 of a wave's trace row: where it ran — HW_REG_XCC_ID << 32 | HW_REG_HW_ID (simd [5:4], cu [11:8], sh [12], se [15:13])
#define EB_MARK_PLACE(A, row) do { if ((A).trace && (threadIdx.x & 63) == 0) (A).trace[(size_t)(row) * 8 + 7] = \
    ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); } while (0)

// per-wave near-record queue: normally one drain at the end; in a crowded tile the in-loop tests stop while 64
// slots are still free and the rest is tested record by record with a drain before each (see record_wave).  4 blocks of 2048 records must fit a CU's LDS with
// room to spare: blocks above ~32 KB were seen to run 3 per CU, i.e. a second round of blocks.
constexpr int QCAP = 160;
constexpr int TAPE_QCAP = 192;   // the tape kernel drains when more than 64 entries wait and adds at most 2 x 64 before the next check

// ---- closest point of (px, py) on path p (DAM:702-715), tables in global memory (L1/L2 resident) ----
// The cell of the position names the index range [lo, hi] that provably holds the reference's argmin for
// every position inside the cell (eb_capi.hip:build_cell_grid); scanning it in index order with the
// reference's fp32 expression and a strict '<' returns the index of the full scan after 2-4 evaluations (ranges narrowed by witnesses, round 5)
// instead of ~370.  Positions outside the grid (or NaN) take the pruned full search.
// Returns the table index and the table point itself (x, y, heading).
// xy10 / phi10: the stride-10 tables — in global memory (the per-step kernel: L1/L2 resident), or the copy a block of the
// tape / gated kernels keeps in LDS for its whole launch (north_star: "LDS staging of the reference path per block")
// PRE: groups of four table entries fetched in one round trip (eb_device.h:closest_in_range) — 3 in the per-step kernel, whose tables
// sit in L2; 0 — a group per loop trip, as before — in the tape / gated kernels (their register budget is the records', and small grids read the tables from LDS)
// COARSE: off the corridor's grid, try the coarse level before the pruned full search (the per-step kernel; the tape / gated kernels'
// 2048-record tile has no register for a second scan loop: 12-20 bytes of scratch with it — same index either way)
template <int PRE = 3, bool COARSE = true>
EB_DEV int closest_cell_index(const FusedArgs& A, const float* xy10, const float* phi10, int p, int roff, float px, float py,
                              float& rx, float& ry, float& rphi) {
    const float* xy = xy10 + 2 * roff;
    const float* ph = phi10 + roff;
    const float fx = (px - A.gx0) * CELL_INV, fy = (py - A.gy0) * CELL_INV;
    unsigned c = 0xffffffffu;                                                  // (also a corridor cell on the path's medial axis: eb_capi.hip)
    if (fx >= 0.0f && fx < (float)A.gnx && fy >= 0.0f && fy < (float)A.gny) c = A.cells[(p * A.gny + (int)fy) * A.gnx + (int)fx];
    if (c == 0xffffffffu) {
        // off the corridor's grid: the coarse levels (eb_device.h:coarse_cell_ranges; described in the handle's table block — a rare
        // path, the descriptors are read from memory rather than carried in the kernel arguments), then the pruned full search
        int r_first = 0, r_last = 1 << 30;
        if constexpr (COARSE) {
            const PathTables& pt = *A.dt;
            int lo, hi, lo2, hi2;
            const int how = coarse_cell_ranges(pt, p, px, py, lo, hi, lo2, hi2);
            if (how == 1) return closest_in_ranges(xy, ph, lo, hi, lo2, hi2, px, py, rx, ry, rphi);
            if (how == 2) { r_first = lo; r_last = hi; }
        }
        const int n = p == 0 ? A.red_len[0] : p == 1 ? A.red_len[1] : A.red_len[2];
        const int bi = closest_reduced_index(reinterpret_cast<const float2*>(xy), A.rad_all + 32 * p, n, px, py, r_first, r_last);
        rx = xy[2 * bi]; ry = xy[2 * bi + 1]; rphi = ph[bi];
        return bi;
    }
    if (PRE > 0 && A.scan_one_trip) return closest_in_range<0>(xy, ph, (int)(c & 0xffffu), (int)(c >> 16), px, py, rx, ry, rphi);
    return closest_in_range<PRE>(xy, ph, (int)(c & 0xffffu), (int)(c >> 16), px, py, rx, ry, rphi);
}

template <int RW, int RPT>
struct FusedSmem {
    static constexpr int ITEMS = RW * 64 * RPT;
    float4 ego[64];                       // (x, y, sin phi, cos phi) of the current ego pose
    unsigned char turn[64];               // per slot: TURN_* (every record wave writes the same codes, reads its own)
    unsigned long long mask[64];          // per env: slots with a non-zero penalty sum
    float2 pen[ITEMS];                    // per record: (3.5 m sum, 2.5 m sum), DAM:228-229
    v2f qxy[RW][QCAP];                    // per record wave: queued near records: (x, y),
    v2f qsc[RW][QCAP];                    //   (sin, cos) of the heading — the prediction has just computed them (DAM:221 = DAM:413-414's angle),
    int qitem[RW][QCAP];                  //   item id
    int ego_ready;                        // set by the env wave once ego[] and mask[] are written
    int waves_done;                       // record waves that have published their partial sums
};

// ---- env wave -------------------------------------------------------------------------------------
template <int TASK, int RW, int RPT, typename ST>
EB_DEV void env_wave(const FusedHot<ST>& H, const FusedArgs& A, FusedSmem<RW, RPT>& S, int e0, int nE) {
    const int lane = threadIdx.x;   // wave 0
    const int D = H.obs_dim, NV = H.n_veh;
    const bool act = lane < nE;
    const int e = act ? lane : 0, ge = e0 + e;
    const ST* hin = H.obs_in + (size_t)ge * D;
    ST* hout = H.obs_out + (size_t)ge * D;

    // head (ego 6 | first tracking triple), action, path id
    const f4u h0 = Stored<ST>::load4(hin), h1 = Stored<ST>::load4(hin + 4);
    const float h8 = Stored<ST>::load1(hin + 8);
    const f2u araw = *reinterpret_cast<const f2u*>(A.actions + 2 * (size_t)ge);
    int p = A.path_id;
    if (A.training) {
        const int pr = A.ref_idx[ge];
        p = (pr >= 0 && pr < A.n_paths) ? pr : -1;                          // DAM:342, 352
    }
    // the shield's running penalty of this env (eb_shield_is_safe's look-aheads after the first): fetched with the head, used at the tail
    float sh_prev = 0.0f;
    if (A.shield_punish && !A.shield_first && act) sh_prev = A.shield_punish[ge];
    const int trow = blockIdx.x * (RW + 1);
    // the block's only barrier, behind the loads: it orders the flags' initial zeros before any wave polls them
    if (lane == 0) { S.ego_ready = 0; S.waves_done = 0; }
    lds_barrier();
    EB_MARK(A, trow, 0);                                                    // loads issued
    EB_MARK_PLACE(A, trow);
    const bool do_rewards = (H.do_rewards & HOT_REWARDS) != 0;
    const float st[6] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y};
    const float phi_rad = deg2rad(st[5]);
    float es, ec;
    sincos_det(phi_rad, es, ec);                                            // DAM:211 and DAM:79-80
    if (do_rewards) {
        S.ego[lane] = make_float4(st[3], st[4], es, ec);
        S.mask[lane] = 0ull;
        lds_publish(&S.ego_ready, 1);                                       // ---- hand-off 1 ----
        EB_MARK(A, trow, 1);                                                // head arrived, ego published
    }

    float steer, a_x;
    if (A.actions_raw) action_transform(araw.x, araw.y, steer, a_x);        // DAM:120
    else { steer = araw.x; a_x = araw.y; }
    if (act && A.scaled_actions) *reinterpret_cast<f2u*>(A.scaled_actions + 2 * (size_t)ge) = f2u{steer, a_x};
    float rew = 0.0f;
    if (act && do_rewards) {
        const float punish_steer = -sq(steer), punish_a_x = -sq(a_x);       // DAM:198-199
        const float punish_yaw_rate = -sq(st[2]);                           // DAM:202
        const float devi_y = -sq(h1.z);                                     // DAM:205
        const float devi_phi = -sq(deg2rad(h1.w));                          // DAM:206
        const float devi_v = -sq(h8);                                       // DAM:207
        rew = 0.05f * devi_v + 0.8f * devi_y + 30.0f * devi_phi + 0.02f * punish_yaw_rate +
              5.0f * punish_steer + 0.05f * punish_a_x;                     // DAM:297-298
        A.out5[ge] = rew;
    }
    float nx[6];
    f_xu_core(st, steer, a_x, TAU10, phi_rad, es, ec, nx);                  // DAM:387
    nx[0] = __builtin_fminf(__builtin_fmaxf(nx[0], 0.0f), 35.0f);           // DAM:390
    // tracking error of the next pose on the env's path (DAM:334-353)
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    if (p >= 0) {
        const int roff = p == 0 ? A.red_off[0] : p == 1 ? A.red_off[1] : A.red_off[2];
        EB_MARK(A, trow, 2);                                                // bicycle step done
        float rx = 0.0f, ry = 0.0f, rphi = 0.0f;                            // == path[bi * 10]: bi * 10 < len always
        // (the range's first groups of table entries in one round trip on the small tiles — same-box A/B, round 5: 16 384 x 32 on
        // 4 x 4 tiles 6.61 -> 6.43 us; nothing at 4 096 x 16 and 32 768 x 32; the 2048-record tile, whose env wave shares its 80
        // VGPRs with eight records per lane, is 0.1-0.2 us FASTER with a group per trip: profiles/r5_ab_scan.txt)
        constexpr int SCAN_PRE = RW * RPT >= 32 ? 0 : 3;
        const int bi = closest_cell_index<SCAN_PRE>(A, A.xy10, A.phi10, p, roff, nx[3], nx[4], rx, ry, rphi);
        t0 = two2one<TASK>(nx[3], nx[4], rx, ry);                           // DAM:758
        if (A.trace) { asm volatile("" :: "v"(t0)); EB_MARK(A, trow, 3); }  // closest point found
        t1 = deal_with_phi_diff(nx[5] - rphi);                              // DAM:759
        t2 = nx[0] - EXP_V;                                                 // DAM:760
        if (A.n_future > 0 && act) {                                        // DAM:717-724, 763-768
            const PathTables& pt = *A.dt;
            const int len = pt.len[p];
            ST* otrk = hout + 9;
            int cur = bi * 10;                                              // DAM:714
            for (int k = 0; k < A.n_future; ++k) {
                cur += 80;
                if (cur >= len - 2) cur = len - 2;
                const int fi = clamp_index(cur, len);
                Stored<ST>::store1(otrk + 3 * k, pt.x[p][fi] - nx[3]);
                Stored<ST>::store1(otrk + 3 * k + 1, pt.y[p][fi] - nx[4]);
                Stored<ST>::store1(otrk + 3 * k + 2, deal_with_phi_diff(nx[5] - pt.phi[p][fi]));
            }
        }
    } else if (A.n_future > 0 && act) {
        ST* otrk = hout + 9;
        for (int c = 0; c < 3 * A.n_future; ++c) Stored<ST>::store1(otrk + c, 0.0f);   // DAM:342, 352
    }
    if (act) {
        Stored<ST>::store4(hout, f4u{nx[0], nx[1], nx[2], nx[3]});
        Stored<ST>::store4(hout + 4, f4u{nx[4], nx[5], t0, t1});
        Stored<ST>::store1(hout + 8, t2);
    }
    EB_MARK(A, trow, 4);                                                    // head stored
    if (!do_rewards) return;
    // the road walls (DAM:231-295) need nothing from the record waves: done while those are still at work
    float road_t = 0.0f, road_r = 0.0f;
    road_terms<TASK>(st[3] + LWS * ec, st[4] + LWS * es, road_t, road_r);
    road_terms<TASK>(st[3] - LWS * ec, st[4] - LWS * es, road_t, road_r);
    lds_wait_until(&S.waves_done, RW);                                      // ---- hand-off 2 ----
    EB_MARK(A, trow, 5);                                                    // record waves done

    // per env: penalty sums in vehicle order (DAM:218-229, 299-300)
    float pun_t = 0.0f, pun_r = 0.0f;
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
        const size_t n = (size_t)H.n_env;
        pun_t = a35 + road_t;                // DAM:299
        pun_r = a25 + road_r;                // DAM:300
        A.out5[n + ge] = pun_t;
        A.out5[2 * n + ge] = pun_r;
        A.out5[3 * n + ge] = a25;
        A.out5[4 * n + ge] = road_r;
        if (A.shield_punish) {   // hier_decision.py:93-97: punish += penalty — the accumulation a launch of its own used to do
            const float pacc = sh_prev + (A.shield_row == 3 ? a25 : pun_r);
            A.shield_punish[ge] = pacc;
            if (A.shield_last) A.shield_safe[ge] = pacc > 0.0f ? 0 : 1;
        }
    }
    // The launch that ends a rollout has no successor to make its record: it makes it here, and adds the |delta_y| statistics of
    // the rows it has just written (t0 IS the final obs' column 6) — once per horizon.
    if (H.acc_rec && A.acc_final) {
        acc_record(H.acc_rec, act, lane, rew, pun_t, pun_r);
        const float dy = __builtin_fabsf(Stored<ST>::round(t0));
        const double s_dy = wave_sum_f64(act ? (double)dy : 0.0);
        const float m_dy = wave_max_f32(act && dy > 0.0f ? dy : 0.0f);       // (a NaN never becomes the maximum, as in the two-pass summary)
        if (lane == 63) {
            typedef double d2v __attribute__((ext_vector_type(2)));
            reinterpret_cast<d2v*>(A.acc_final)[blockIdx.x] = d2v{s_dy, (double)m_dy};
        }
    }
    EB_MARK(A, trow, 6);                                                    // end
}

// ---- record waves -----------------------------------------------------------------------------------
// one queue pass: entries [base, base + n) of this wave's queue, one per lane: DAM:218-229
// the four circle-pair terms of one queued record (DAM:218-229) -> its partial sums + a bit in its env's slot mask
template <typename SM, typename ST>
EB_DEV void queue_terms(const FusedHot<ST>& H, SM& S, const float4* ego, int item, float x, float y, float vs, float vc) {
    const int e2 = env_of_item(H, item), j2 = item - e2 * H.n_veh;
    const float4 eg = ego[e2];
    float t35[4], t25[4];
    const float4 pts = make_float4(eg.x + LWS * eg.w, eg.y + LWS * eg.z, eg.x - LWS * eg.w, eg.y - LWS * eg.z);
    veh2veh_terms(pts, x, y, vs, vc, t35, t25);
    const float p35 = ((t35[0] + t35[1]) + t35[2]) + t35[3];
    const float p25 = ((t25[0] + t25[1]) + t25[2]) + t25[3];
    if (p35 != 0.0f) {   // p25 != 0 implies p35 != 0
        S.pen[item] = make_float2(p35, p25);
        atomicOr(&S.mask[e2], 1ull << j2);
    }
}
// one queue pass: entries [base, base + n) of this wave's queue, one per lane.  The tape kernel queues a record's heading
// (its near tests run before the prediction), the per-step kernel the heading's sin / cos (computed by the prediction).
template <typename SM, typename ST>
EB_DEV void queue_pass(const FusedHot<ST>& H, SM& S, const float4* ego, int w, int lane, int base, int n) {
    if (lane < n) {
        const v2f vxy = S.qxy[w][base + lane];
        float vs, vc;
        sincos_det(deg2rad(S.qphi[w][base + lane]), vs, vc);                // DAM:221
        queue_terms(H, S, ego, S.qitem[w][base + lane], vxy.x, vxy.y, vs, vc);
    }
}
template <typename SM, typename ST>
EB_DEV void queue_pass_sc(const FusedHot<ST>& H, SM& S, const float4* ego, int w, int lane, int base, int n) {
    if (lane < n) {
        const v2f vxy = S.qxy[w][base + lane], sc = S.qsc[w][base + lane];
        queue_terms(H, S, ego, S.qitem[w][base + lane], vxy.x, vxy.y, sc.x, sc.y);
    }
}

// (predict_record_tc, TurnC / turn_consts, SinCosK / sincos_consts / sincos_det_k: eb_device.h — shared with eb_env_step.hip)

// FAST: RW * 64 % n_veh == 0 — a lane keeps its vehicle slot over all its records and its env advances by a
// fixed step, so slot constants are fetched once and addresses advance by a uniform stride.
// PF: record loads in flight per lane.  RPT — every record requested up front; less — record k + PF is requested when record k is
// done (FusedArgs::rolling: the queue ahead of every wave's first records is shorter, and a record's registers are reused).
template <int TASK, int RW, int RPT, bool FAST, int PF, typename ST>
EB_DEV void record_wave(const FusedHot<ST>& H, const FusedArgs& A, FusedSmem<RW, RPT>& S, int e0, int nE) {
    constexpr int RL = RW * 64;                     // record lanes per block
    const int rtid = threadIdx.x - 64, w = rtid >> 6, lane = rtid & 63;
    const int NV = H.n_veh, D = H.obs_dim, HD = D - 4 * NV;
    const int items = nE * NV;
    const ST* tin = H.obs_in + (size_t)e0 * D;
    ST* tout = H.obs_out + (size_t)e0 * D;
    const int e_first = env_of_item(H, rtid), j_first = rtid - e_first * NV;
    const int epk = RL / NV;                                  // FAST: envs per k step
    const int off_first = 4 * rtid + (e_first + 1) * HD, off_step = 4 * RL + epk * HD;
    // record k of this lane: item id, env (tile-local), float offset of the record in the tile (== e*D + HD + 4*j)
    auto item_of = [&](int k) { return k * RL + rtid; };
    auto env_of = [&](int k) { return FAST ? e_first + k * epk : env_of_item(H, item_of(k)); };
    auto off_of = [&](int k) { return FAST ? off_first + k * off_step : 4 * item_of(k) + (env_of(k) + 1) * HD; };

    // slot turn codes first (in-order return: the record loads behind it do not hold the table up), then the records.  Nothing
    // here waits for a scalar load: the table's address is a preloaded argument (FusedHot::turn), as are the rows' (obs_in)
    static_assert(PF >= 1 && PF <= RPT, "record loads in flight per lane");
    const int turn_code = H.turn[lane];
    f4u rec[PF];
    // A tile that holds its full RL * RPT records (every tile but a batch's last one) needs no per-record bounds
    // checks: the two forms of each loop below differ only in that (wave-uniform choice, same results).
    const bool full_tile = items == RL * RPT;
    auto load_records = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            // lanes past the tile's last record re-read the tile's last record (branch-free loads; never stored)
            const bool valid = FULL || item_of(k) < items;
            rec[k] = Stored<ST>::load4(tin + (valid ? off_of(k) : 4 * (items - 1) + nE * HD));
            if (!valid) rec[k].x = 1e30f;            // never near an ego (and never stored)
        }
    };
    if (full_tile) load_records(std::true_type{}); else load_records(std::false_type{});
    // Episodic accumulator: the PREVIOUS step's record is made by this launch's first record wave (lane = env of the tile) — three
    // coalesced loads behind its records now, the sums at its very end, when it would otherwise just leave: the env wave, whose
    // chain IS the block's duration, does not see any of it (in the env wave the same ~110 instructions cost 0.4 us per launch)
    const bool acc_prev = w == 0 && H.acc_rec && A.prev_out5;
    float pv_r = 0.0f, pv_t = 0.0f, pv_p = 0.0f;
    if (acc_prev && lane < nE) {
        const size_t n = (size_t)H.n_env;
        const float* po = A.prev_out5 + e0 + lane;
        pv_r = po[0]; pv_t = po[n]; pv_p = po[2 * n];
    }
    const int trow = blockIdx.x * (RW + 1) + 1 + w;
    lds_barrier();   // the block's only barrier (the env wave zeroes the hand-off flags in front of it), behind this wave's loads
    EB_MARK(A, trow, 0);                                                    // loads issued
    EB_MARK_PLACE(A, trow);
    if (A.trace && lane == 0) A.trace[(size_t)trow * 8 + 6] = H.t_entry;    // when this wave reached its first instruction
    const bool do_rewards = (H.do_rewards & HOT_REWARDS) != 0;
    // Issue priority by progress (HOT_PRIO): a wave on its first records outranks one on its last — the launch ends with its LAST
    // wave, and the hardware's oldest-first issue otherwise lets the first-placed waves of a SIMD run ahead and leave the youngest to
    // finish alone (record waves end 4.8-10.9 us after the first entry without it, 6.0-10.7 with).
    // The env wave runs at 2 throughout: below the record waves' first quarter, above their second half.
    const bool by_progress = (H.do_rewards & HOT_PRIO) != 0;
    S.turn[lane] = (unsigned char)turn_code;   // same bytes from every record wave; a wave reads back its own write
    const TurnC tc_lane = turn_consts(S.turn[FAST ? j_first : 0]);
    const SinCosK SK = sincos_consts();

    // ---- near-ego records -> this wave's queue ----
    // A circle pair can only be closer than 3.5 m when the two vehicle centres are within 3.5 + 2*1.4 = 6.3 m;
    // records inside 6.364 m (slack >> fp32 rounding) are queued, every other record contributes exact zeros
    // to the penalty sums (DAM:228-229).
    int qn = 0;
    auto drain = [&]() {
        for (int base = 0; base < qn; base += 64) queue_pass_sc(H, S, S.ego, w, lane, base, min(64, qn - base));
        qn = 0;
    };
    // (lanes past the tile's last record carry x = 1e30 in `r`, see the loads: never near)
    auto near_test = [&](int item, const v2f eg, const f4u r, const v2f sc) {
        const v2f d = v2f{r.x, r.y} - eg;
        const v2f d2 = d * d;
        const bool near = d2.x + d2.y < 40.5f;
        const unsigned long long b = __builtin_amdgcn_ballot_w64(near);
        if (b) {
            if (near) {
                // queue slot = entries so far + near lanes below this one (v_mbcnt with the running count as its addend)
                const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, (unsigned)qn));
                S.qxy[w][pos] = v2f{r.x, r.y};          // three plain LDS writes straight from the record's registers
                S.qsc[w][pos] = sc;
                S.qitem[w][pos] = item;
            }
            qn += __popcll(b);
        }
    };
    // The env wave's head is the first load of the block, so the ego poses are normally in LDS by the time this
    // wave's records arrive: wait for them here (hand-off 1), then test each record as it is predicted — a record's
    // registers are free again after its iteration and only the queue pass is left once the stores are out.
    // Crowded tiles: once fewer than 64 queue slots are free the in-loop tests stop (k_late); the remaining records
    // are re-read (L2) and tested after the loop, with the queue drained in between.
    if (do_rewards) {
        lds_wait_until(&S.ego_ready, 1);                                    // ---- hand-off 1: ego poses are in LDS ----
        EB_MARK(A, trow, 3);                                                // ego seen
    }
    const bool test_near = do_rewards;
    int k_late = test_near ? RPT : 0;
    auto main_loop = [&](auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        // the ego positions of this lane's records, all LDS reads in flight at once (one wait instead of one per record)
        v2f egoxy[RPT];
        if (test_near) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int item = k * RL + rtid;
                const int env = FAST ? e_first + k * epk : env_of_item(H, item);
                egoxy[k] = *reinterpret_cast<const v2f*>(&S.ego[(FULL || item < items) ? env : 0]);
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            // Recompute this record's item id / env / offset from the lane's bases (an add each) instead of keeping the
            // eight copies made for the loads alive through the whole loop: the empty asm hides the per-step constants.
            int k_item = k * RL, k_env = k * epk, k_off = k * off_step;
            asm volatile("" : "+s"(k_item), "+s"(k_env), "+s"(k_off));
            const int item = k_item + rtid;
            const int env = FAST ? e_first + k_env : env_of_item(H, item);
            const int off = FAST ? off_first + k_off : 4 * item + (env + 1) * HD;
            const bool valid = FULL || item < items;
            // the prediction first: it yields sin / cos of the record's heading, which a near record takes to the queue
            const TurnC tc = FAST ? tc_lane : turn_consts(S.turn[valid ? item - env * NV : 0]);
            float sn, cs;
            if ((k * 4) % RPT == 0 && by_progress)                              // a quarter of the records further: one priority level down
                switch (3 - (k * 4) / RPT) {
                    case 3: __builtin_amdgcn_s_setprio(3); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    default: __builtin_amdgcn_s_setprio(0); break;
                }
            const f4u nv = predict_record_tc<ST>(rec[k % PF], tc, SK, sn, cs);
            if (k < k_late) {
                near_test(item, egoxy[k], rec[k % PF], v2f{sn, cs});
                if (qn > QCAP - 64) k_late = k + 1;
            }
            if (k + PF < RPT) {                      // rolling loads: this record's registers take record k + PF
                const bool v2 = FULL || item_of(k + PF) < items;
                rec[k % PF] = Stored<ST>::load4(tin + (v2 ? off_of(k + PF) : 4 * (items - 1) + nE * HD));
                if (!v2) rec[k % PF].x = 1e30f;
            }
            // a 32-bit byte offset from the tile's (wave-uniform) base: the store then takes the base from SGPRs and the
            // offset from one VGPR (an element offset would be widened to a 64-bit address in three VALU instructions)
            if (valid) Stored<ST>::store4(reinterpret_cast<ST*>(reinterpret_cast<char*>(tout) + (unsigned)off * (unsigned)sizeof(ST)), nv);
            if (k == 0) EB_MARK(A, trow, 1);                                    // first record stored
            if (k == RPT - 1) EB_MARK(A, trow, 2);                              // last record stored
            if (k & 1) __builtin_amdgcn_sched_barrier(0);   // two records at a time: bounds the live set, leaves some ILP
        }
    };
    if (full_tile) main_loop(std::true_type{}); else main_loop(std::false_type{});
    if (!do_rewards) return;   // (eb_compute_next_obses: never an accumulating launch)
    if (by_progress) __builtin_amdgcn_s_setprio(0);
    if (test_near && k_late < RPT) {
        for (int k = k_late; k < RPT; ++k) {       // not unrolled: rare path
            drain();
            const bool valid = item_of(k) < items;
            f4u r = Stored<ST>::load4(tin + (valid ? off_of(k) : 4 * (items - 1) + nE * HD));
            if (!valid) r.x = 1e30f;
            float sn, cs;
            sincos_det(deg2rad(r.w), sn, cs);
            near_test(item_of(k), *reinterpret_cast<const v2f*>(&S.ego[valid ? env_of(k) : 0]), r, v2f{sn, cs});
        }
    }
    EB_MARK(A, trow, 4);                                                    // near tests done
    drain();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // ---- hand-off 2: partial sums are in LDS ----
    if (lane == 0) atomicAdd(&S.waves_done, 1);
    if (acc_prev) {
        // (the values are first TOUCHED here: otherwise the float -> double conversions — and the wait for the three loads — are
        // hoisted to the top of the wave, in front of the records' own arrival)
        asm volatile("" : "+v"(pv_r), "+v"(pv_t), "+v"(pv_p));
        acc_record(A.prev_rec, lane < nE, lane, pv_r, pv_t, pv_p);
    }
    EB_MARK(A, trow, 5);                                                    // end
}

// ====================================================================================================
// Open-loop rollout over an action tape in ONE launch (eb_rollout_tape; the MPC callers' cost_function,
// mpc/main.py:470-479): the same two roles, with a tile's records and ego states kept in registers across the
// H steps.  HBM is touched for the initial obs, the actions and out5 of every step, the table gathers and the
// final obs: per env-step that is 28 + (72 + 32 N) / H bytes — the kernel is VALU-bound, not HBM-bound, and is
// reported separately from the per-step kernel.  Arithmetic and order of operations are those of
// the per-step kernel, so the results equal H calls of eb_rollout_step bit for bit (fp16 storage: every step's
// state passes through Stored<ST>::round, i.e. the binary16 rounding a store + load would apply).
// Hand-offs per step t: ego_ready = t + 1 (env wave -> record waves, ego poses double-buffered),
// waves_done = RW * (t + 1) (record waves -> env wave).
template <int RW, int RPT>
struct TapeSmem {
    static constexpr int ITEMS = RW * 64 * RPT;
    float4 ego[2][64];
    unsigned char turn[64];
    unsigned long long mask[64];
    float2 pen[ITEMS];
    v2f qxy[RW][TAPE_QCAP];
    float qphi[RW][TAPE_QCAP];
    int qitem[RW][TAPE_QCAP];
    int ego_ready;
    int waves_done;
    int pub_done;                         // gated rollout: record waves whose step outputs have left for memory
    int abort;                            // gated rollout: a gate timed out, every wave leaves
};
// wait for *flag >= value; false when the block is giving up (gated rollout only)
EB_DEV bool lds_wait_or_abort(int* flag, int value, int* abort) {
    while (*(volatile lds_int*)flag < value) {
        if (*(volatile lds_int*)abort) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    asm volatile("" ::: "memory");
    return true;
}

// Gated rollout: the env wave never waits for memory inside the step loop.  Two COURIER waves of the block do that: the
// in-courier polls gate_ready[t], fetches the tile's actions[t] past the L1 and hands them over in LDS; the out-courier
// takes a step's outputs from LDS, writes them through to memory, waits for them (and the record waves' rows) to have
// arrived and raises the block's word of gate_done[t].  Both run a step or two apart from the env wave (double buffers), so
// with open gates a step costs what it costs the open-loop tape kernel, and with a producer in the loop the memory
// round trips of the hand-off are the only thing on the critical path.
struct GateSmem {
    f2u act[2][64];                       // actions[t] of the tile's envs, by t & 1
    float out5[2][5][64];                 // the five outputs of step t
    float head[2][9][64];                 // the next obs head (6 ego + 3 tracking) of step t — only used with obs_steps
    int bi[2][64];                        //   and the closest-point index its look-ahead columns start from
    int act_ready;                        // in-courier -> env wave: actions of step (value - 1) are in LDS
    int out_ready;                        // env wave -> couriers: outputs of step (value - 1) are in LDS
    int sent;                             // out-courier -> env wave: outputs of step (value - 1) have been shipped
};

// obs head row: 6 ego + 3 tracking columns, then 3 * n_future look-ahead columns from table index bi * 10 (DAM:717-724,
// 763-768); WT: written through (another kernel reads the row while this one runs)
template <typename ST, bool WT>
EB_DEV void store_head_row(const FusedArgs& A, ST* row, const float (&hv)[9], int bi, int p) {
    auto s4 = [&](ST* q, f4u v) { if (WT) Stored<ST>::store4_wt(q, v); else Stored<ST>::store4(q, v); };
    auto s1 = [&](ST* q, float v) { if (WT) Stored<ST>::store1_wt(q, v); else Stored<ST>::store1(q, v); };
    s4(row, f4u{hv[0], hv[1], hv[2], hv[3]});
    s4(row + 4, f4u{hv[4], hv[5], hv[6], hv[7]});
    s1(row + 8, hv[8]);
    ST* otrk = row + 9;
    if (p >= 0) {
        const PathTables& pt = *A.dt;
        const int len = pt.len[p];
        int cur = bi * 10;                                                  // DAM:714
        for (int k = 0; k < A.n_future; ++k) {
            cur += 80;
            if (cur >= len - 2) cur = len - 2;
            const int fi = clamp_index(cur, len);
            s1(otrk + 3 * k, pt.x[p][fi] - hv[3]);
            s1(otrk + 3 * k + 1, pt.y[p][fi] - hv[4]);
            s1(otrk + 3 * k + 2, deal_with_phi_diff(hv[5] - pt.phi[p][fi]));
        }
    } else {
        for (int c = 0; c < 3 * A.n_future; ++c) s1(otrk + c, 0.0f);       // DAM:342, 352
    }
}

// the path an env follows: its own ref_idx in training mode (out of range: none, DAM:342, 352), else the handle's
EB_DEV int env_path(const FusedArgs& A, int ge) {
    if (!A.training) return A.path_id;
    const int pr = A.ref_idx[ge];
    return (pr >= 0 && pr < A.n_paths) ? pr : -1;
}

template <int RW, int RPT>
EB_DEV void in_courier(const FusedArgs& A, TapeSmem<RW, RPT>& S, GateSmem& G, int e0, int nE, int n_env, int horizon) {
    const int lane = threadIdx.x & 63;
    const int ge = e0 + (lane < nE ? lane : 0);
    for (int t = 0; t < horizon; ++t) {
        // act[t & 1] is free once the env wave has taken step t - 2's actions, which it has when that step's outputs are out
        if (t >= 2 && !lds_wait_or_abort(&G.out_ready, t - 1, &S.abort)) return;
        // the step gate: the producer of actions[t] raises gate_ready[t] after its stores have left for memory.  One look
        // right away (a producer that runs ahead has opened it already); if it is shut, the producer is presumably waiting
        // for step t - 1's results, so polling starts once those are on their way out — a few hundred blocks polling one
        // word through a whole step would only stand in the way of the stores that open it.
        unsigned open = agent_load_u32(A.gate_ready + t);
        if (!open && t >= 1 && !lds_wait_or_abort(&G.out_ready, t, &S.abort)) return;
        for (int spins = 0; !open && spins <= A.gate_spin; ++spins) {
            open = agent_load_u32(A.gate_ready + t);
            if (open || *(volatile lds_int*)&S.abort) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (!open) {                                                        // give up: tell the host and the other waves
            if (lane == 0) agent_store_u32(A.gate_status, 1u);
            lds_publish(&S.abort, 1);
            return;
        }
        const unsigned long long bits = agent_load_u64(A.actions + 2 * ((size_t)t * n_env + ge));
        G.act[t & 1][lane] = __builtin_bit_cast(f2u, bits);
        lds_publish(&G.act_ready, t + 1);
    }
}

template <int RW, int RPT, typename ST>
EB_DEV void out_courier(const FusedHot<ST>& H, const FusedArgs& A, TapeSmem<RW, RPT>& S, GateSmem& G, int e0, int nE, int horizon) {
    const int lane = threadIdx.x & 63;
    const bool act = lane < nE;
    const int ge = e0 + (act ? lane : 0);
    const size_t n = (size_t)H.n_env;
    const int p = env_path(A, ge);
    for (int t = 0; t < horizon; ++t) {
        if (!lds_wait_or_abort(&G.out_ready, t + 1, &S.abort)) return;
        if (act) {
            float* out5 = A.out5 + (size_t)t * 5 * n + ge;
#pragma unroll
            for (int c = 0; c < 5; ++c) Stored<float>::store1_wt(out5 + c * n, G.out5[t & 1][c][lane]);
            if (A.gate_obs) {
                float hv[9];
#pragma unroll
                for (int c = 0; c < 9; ++c) hv[c] = G.head[t & 1][c][lane];
                store_head_row<ST, true>(A, reinterpret_cast<ST*>(A.gate_obs) + ((size_t)t * n + ge) * H.obs_dim, hv, G.bi[t & 1][lane], p);
            }
        }
        // step t is out once these stores have arrived (vmcnt) and the record waves' rows too (pub_done): the block raises
        // its own 64-byte record gate_done[t][block][0..15] — four lanes write the whole record in one instruction, so the
        // memory side sees one full write per block; one shared counter would serialise the device's few hundred atomics
        // in one memory channel (13 ns each, measured: more than the step itself), and neighbouring 4-byte flags would do
        // the same to their partial writes.  A consumer that finds word 0 of all gridDim.x records of step t set may read
        // out5[t] (and obs_steps[t]).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (A.gate_obs && !lds_wait_or_abort(&S.pub_done, RW * (t + 1), &S.abort)) return;
        if (lane < 4) {
            typedef unsigned u4w __attribute__((ext_vector_type(4)));
            const u4w ones = {1u, 1u, 1u, 1u};
            unsigned* rec = A.gate_done + ((size_t)t * gridDim.x + blockIdx.x) * 16 + 4 * lane;
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(rec), "v"(ones) : "memory");
        }
        lds_publish(&G.sent, t + 1);
    }
}

template <int TASK, int RW, int RPT, bool GATED, typename ST>
EB_DEV void env_wave_tape(const FusedHot<ST>& H, const FusedArgs& A, TapeSmem<RW, RPT>& S, GateSmem* Gp, int e0, int nE, int horizon,
                          const float* xy10, const float* phi10) {
    const int lane = threadIdx.x;   // wave 0
    const int D = H.obs_dim, NV = H.n_veh;
    const bool act = lane < nE;
    const int e = act ? lane : 0, ge = e0 + e;
    const ST* hin = H.obs_in + (size_t)ge * D;
    ST* hout = H.obs_out + (size_t)ge * D;
    const f4u h0 = Stored<ST>::load4(hin), h1 = Stored<ST>::load4(hin + 4);
    float st[6] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y};
    float trk[3] = {h1.z, h1.w, Stored<ST>::load1(hin + 8)};
    const int p = env_path(A, ge);
    const int roff = p == 1 ? A.red_off[1] : p == 2 ? A.red_off[2] : A.red_off[0];
    const size_t n = (size_t)H.n_env;
    constexpr bool gated = GATED;
    f2u araw = f2u{0.0f, 0.0f};
    if (!gated) araw = *reinterpret_cast<const f2u*>(A.actions + 2 * (size_t)ge);
    const int trow = blockIdx.x * (RW + 1);
    long long waited = 0;
    EB_MARK(A, trow, 0);
    for (int t = 0; t < horizon; ++t) {
        float* out5 = A.out5 + (size_t)t * 5 * n;
        f2u araw_next = araw;
        if (gated) {
            // actions[t] come from the in-courier; out5 / head of step t go to buffer t & 1, free once step t - 2 is shipped
            if (!lds_wait_or_abort(&Gp->act_ready, t + 1, &S.abort)) return;
            araw = Gp->act[t & 1][lane];
            if (t >= 2 && !lds_wait_or_abort(&Gp->sent, t - 1, &S.abort)) return;
        } else if (t + 1 < horizon) {
            araw_next = *reinterpret_cast<const f2u*>(A.actions + 2 * ((size_t)(t + 1) * n + ge));   // prefetch
        }
        const float phi_rad = deg2rad(st[5]);
        float es, ec;
        sincos_det(phi_rad, es, ec);                                        // DAM:211 and DAM:79-80
        S.ego[t & 1][lane] = make_float4(st[3], st[4], es, ec);
        S.mask[lane] = 0ull;                       // the record waves are through with step t - 1 (waited for below)
        lds_publish(&S.ego_ready, t + 1);                                   // ---- hand-off 1 ----
        float steer, a_x;
        action_transform(araw.x, araw.y, steer, a_x);                       // DAM:120
        // step outputs: plain stores, or handed to the out-courier through LDS (gated)
        auto put = [&](int c, float v) { if (gated) Gp->out5[t & 1][c][lane] = v; else out5[(size_t)c * n + ge] = v; };
        if (act) {
            const float punish_steer = -sq(steer), punish_a_x = -sq(a_x);   // DAM:198-199
            const float punish_yaw_rate = -sq(st[2]);                       // DAM:202
            const float devi_y = -sq(trk[0]);                               // DAM:205
            const float devi_phi = -sq(deg2rad(trk[1]));                    // DAM:206
            const float devi_v = -sq(trk[2]);                               // DAM:207
            put(0, 0.05f * devi_v + 0.8f * devi_y + 30.0f * devi_phi + 0.02f * punish_yaw_rate +
                   5.0f * punish_steer + 0.05f * punish_a_x);               // DAM:297-298
        }
        float nx[6];
        f_xu_core(st, steer, a_x, TAU10, phi_rad, es, ec, nx);              // DAM:387
        nx[0] = __builtin_fminf(__builtin_fmaxf(nx[0], 0.0f), 35.0f);       // DAM:390
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        int bi = 0;
        if (p >= 0) {                                                       // DAM:334-353
            float rx = 0.0f, ry = 0.0f, rphi = 0.0f;
            bi = closest_cell_index<0, false>(A, xy10, phi10, p, roff, nx[3], nx[4], rx, ry, rphi);
            t0 = two2one<TASK>(nx[3], nx[4], rx, ry);                       // DAM:758
            t1 = deal_with_phi_diff(nx[5] - rphi);                          // DAM:759
            t2 = nx[0] - EXP_V;                                             // DAM:760
        }
        // the head of the next obs (+ look-ahead columns, which feed nothing on the way): to the final obs after the last
        // step; a gated rollout that publishes its states hands it to the out-courier after every step
        const float hv[9] = {nx[0], nx[1], nx[2], nx[3], nx[4], nx[5], t0, t1, t2};
        if (t == horizon - 1 && act) store_head_row<ST, false>(A, hout, hv, bi, p);
        if (GATED && A.gate_obs) {
#pragma unroll
            for (int c = 0; c < 9; ++c) Gp->head[t & 1][c][lane] = hv[c];
            Gp->bi[t & 1][lane] = bi;
        }
        float road_t = 0.0f, road_r = 0.0f;                                 // the road walls, DAM:231-295: while the record waves work
        road_terms<TASK>(st[3] + LWS * ec, st[4] + LWS * es, road_t, road_r);
        road_terms<TASK>(st[3] - LWS * ec, st[4] - LWS * es, road_t, road_r);
        const long long w0 = A.trace ? wall_clock64() : 0;
        if (GATED) { if (!lds_wait_or_abort(&S.waves_done, RW * (t + 1), &S.abort)) return; }
        else lds_wait_until(&S.waves_done, RW * (t + 1));                    // ---- hand-off 2 ----
        if (A.trace) waited += wall_clock64() - w0;
        if (act) {                                                          // DAM:218-229, 299-300
            float a35 = 0.0f, a25 = 0.0f;
            unsigned long long m = S.mask[lane];
            while (m) {
                const int jj = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float2 ps = S.pen[lane * NV + jj];
                a35 += ps.x;
                a25 += ps.y;
            }
            put(1, a35 + road_t);                   // DAM:299
            put(2, a25 + road_r);                   // DAM:300
            put(3, a25);
            put(4, road_r);
        }
        if (gated) lds_publish(&Gp->out_ready, t + 1);                      // ---- step t is in LDS: over to the out-courier
#pragma unroll
        for (int c = 0; c < 6; ++c) st[c] = Stored<ST>::round(nx[c]);
        trk[0] = Stored<ST>::round(t0); trk[1] = Stored<ST>::round(t1); trk[2] = Stored<ST>::round(t2);
        araw = araw_next;
    }
    EB_MARK(A, trow, 1);
    if (A.trace && lane == 0) A.trace[(size_t)trow * 8 + 2] = waited;
}

template <int TASK, int RW, int RPT, bool FAST, bool GATED, typename ST>
EB_DEV void record_wave_tape(const FusedHot<ST>& H, const FusedArgs& A, TapeSmem<RW, RPT>& S, int e0, int nE, int horizon) {
    constexpr int RL = RW * 64;
    const int rtid = threadIdx.x - 64, w = rtid >> 6, lane = rtid & 63;
    const int NV = H.n_veh, D = H.obs_dim, HD = D - 4 * NV;
    const int items = nE * NV;
    const ST* tin = H.obs_in + (size_t)e0 * D;
    ST* tout = H.obs_out + (size_t)e0 * D;
    const int e_first = env_of_item(H, rtid), j_first = rtid - e_first * NV;
    const int epk = RL / NV;
    const int off_first = 4 * rtid + (e_first + 1) * HD, off_step = 4 * RL + epk * HD;
    auto item_of = [&](int k) { return k * RL + rtid; };
    auto env_of = [&](int k) { return FAST ? e_first + k * epk : env_of_item(H, item_of(k)); };
    auto off_of = [&](int k) { return FAST ? off_first + k * off_step : 4 * item_of(k) + (env_of(k) + 1) * HD; };
    const int turn_code = A.dt->turn[lane];
    f4u rec[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const bool valid = item_of(k) < items;
        rec[k] = Stored<ST>::load4(tin + (valid ? off_of(k) : 4 * (items - 1) + nE * HD));
        if (!valid) rec[k].x = 1e30f;            // never near an ego, never stored (stays that way: x + dx of 1e30)
    }
    S.turn[lane] = (unsigned char)turn_code;
    const TurnC tc_lane = turn_consts(S.turn[FAST ? j_first : 0]);
    const SinCosK SK = sincos_consts();
    const int trow = blockIdx.x * (RW + 1) + 1 + w;
    long long waited = 0;
    EB_MARK(A, trow, 0);
    for (int t = 0; t < horizon; ++t) {
        const float4* ego = S.ego[t & 1];
        int qn = 0;
        auto drain = [&]() {
            for (int base = 0; base < qn; base += 64) queue_pass(H, S, ego, w, lane, base, min(64, qn - base));
            qn = 0;
        };
        const long long w0 = A.trace ? wall_clock64() : 0;
        if (GATED) { if (!lds_wait_or_abort(&S.ego_ready, t + 1, &S.abort)) return; }
        else lds_wait_until(&S.ego_ready, t + 1);                            // ---- hand-off 1 ----
        if (A.trace) waited += wall_clock64() - w0;
        // near tests of step t on the records as they stand, then the queue, then hand-off 2 ...
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            int k_item = k * RL, k_env = k * epk;
            asm volatile("" : "+s"(k_item), "+s"(k_env));
            const int item = k_item + rtid;
            const int env = FAST ? e_first + k_env : env_of_item(H, item);
            if (k > 0 && (k & 1) == 0 && qn > 64) drain();                  // at most 64 + 2 * 64 = TAPE_QCAP entries ever wait
            const float4 eg = ego[item < items ? env : 0];
            const v2f d = v2f{rec[k].x, rec[k].y} - v2f{eg.x, eg.y};
            const v2f d2 = d * d;
            const bool near = d2.x + d2.y < 40.5f;                          // DAM:228-229, see record_wave
            const unsigned long long b = __builtin_amdgcn_ballot_w64(near);
            if (b) {
                if (near) {
                    const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, (unsigned)qn));
                    S.qxy[w][pos] = v2f{rec[k].x, rec[k].y};
                    S.qphi[w][pos] = rec[k].w;
                    S.qitem[w][pos] = item;
                }
                qn += __popcll(b);
            }
        }
        drain();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  // ---- hand-off 2 ----
        if (lane == 0) atomicAdd(&S.waves_done, 1);
        // ... and the prediction, which needs nothing from the env wave, while that adds up step t and prepares t + 1
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            int k_item = k * RL, k_env = k * epk;
            asm volatile("" : "+s"(k_item), "+s"(k_env));
            const int item = k_item + rtid;
            if (item < items) {
                const int env = FAST ? e_first + k_env : env_of_item(H, item);
                const TurnC tc = FAST ? tc_lane : turn_consts(S.turn[item - env * NV]);
                float sn_, cs_;
                const f4u nv = predict_record_tc<ST>(rec[k], tc, SK, sn_, cs_);
                rec[k] = f4u{Stored<ST>::round(nv.x), Stored<ST>::round(nv.y), Stored<ST>::round(nv.z), Stored<ST>::round(nv.w)};
                // a gated rollout that publishes its states: the record of obs_steps[t], written through
                if (GATED && A.gate_obs) Stored<ST>::store4_wt(reinterpret_cast<ST*>(A.gate_obs) + ((size_t)t * H.n_env + e0) * D + off_of(k), rec[k]);
            }
            if (k & 1) __builtin_amdgcn_sched_barrier(0);
        }
        if (GATED && A.gate_obs) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this step's records have left for memory
            if (lane == 0) atomicAdd(&S.pub_done, 1);
        }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k)
        if (item_of(k) < items) Stored<ST>::store4(tout + off_of(k), rec[k]);
    EB_MARK(A, trow, 1);
    if (A.trace && lane == 0) A.trace[(size_t)trow * 8 + 2] = waited;
}

template <int TASK, int RW, int RPT, bool FAST, bool GATED, typename ST>
EB_DEV void tape_body(const FusedHot<ST>& H, const FusedArgs& A, int horizon) {
    __shared__ TapeSmem<RW, RPT> S;
    extern __shared__ __attribute__((aligned(16))) float staged[];   // A.stage_entries > 0: [2 * entries] (x, y) pairs, then [entries] headings
    const int e0 = blockIdx.x * H.envs_per_tile;
    const int nE = min(H.envs_per_tile, H.n_env - e0);
    __shared__ std::conditional_t<GATED, GateSmem, int> Gs;   // the couriers' hand-off buffers (gated rollout only)
    GateSmem* Gp = GATED ? reinterpret_cast<GateSmem*>(&Gs) : nullptr;
    if (threadIdx.x == 0) {
        S.ego_ready = 0; S.waves_done = 0; S.pub_done = 0; S.abort = 0;
        if (GATED) { Gp->act_ready = 0; Gp->out_ready = 0; Gp->sent = 0; }
    }
    // The stride-10 path tables (DAM:704-706: <= 3 x 512 points, 12 bytes each) into LDS once per launch: the closest-point
    // scan of every step then reads LDS instead of L2 — the env wave's per-step chain is the serial part of a small batch.
    const float* xy10 = A.xy10;
    const float* phi10 = A.phi10;
    if (A.stage_entries > 0) {
        const int n_xy = 2 * A.stage_entries, n_all = 3 * A.stage_entries;
        for (int i = threadIdx.x; i < n_all; i += (RW + (GATED ? 3 : 1)) * 64) staged[i] = i < n_xy ? A.xy10[i] : A.phi10[i - n_xy];
        xy10 = staged;
        phi10 = staged + n_xy;
    }
    lds_barrier();
    if (threadIdx.x < 64) {
        __builtin_amdgcn_s_setprio(3);   // the env wave's per-step chain is the serial part of a step: it goes first
        env_wave_tape<TASK, RW, RPT, GATED, ST>(H, A, S, Gp, e0, nE, horizon, xy10, phi10);
    }
    else if (threadIdx.x < (RW + 1) * 64) record_wave_tape<TASK, RW, RPT, FAST, GATED, ST>(H, A, S, e0, nE, horizon);
    else if (GATED && threadIdx.x < (RW + 2) * 64) in_courier<RW, RPT>(A, S, *Gp, e0, nE, H.n_env, horizon);
    else if (GATED) out_courier<RW, RPT, ST>(H, A, S, *Gp, e0, nE, horizon);
}

template <int TASK, int RW, int RPT, bool FAST, int PF, typename ST>
EB_DEV void fused_body(const FusedHot<ST>& H, const FusedArgs& A) {
    __shared__ FusedSmem<RW, RPT> S;
    const int e0 = blockIdx.x * H.envs_per_tile;
    const int nE = min(H.envs_per_tile, H.n_env - e0);
    // (no barrier here: each role issues its first loads, THEN meets the others at the block's only barrier — a wave that waited
    // for its block's last wave, and then for the scalar load of a table's address, issued its loads ~0.3 us later)
    // Occupancy pad for the 2048-record tile: 4 blocks x 5 waves per CU are 5 waves per SIMD when spread evenly.
    // Holding 73-80 VGPRs caps a SIMD at 6 waves, which keeps the dispatcher from stacking 7 or 8 on one SIMD and 3
    // on another (measured: 17.2 us with 58 VGPRs, 16.2 us with 77; a cap of exactly 5 makes some blocks wait a round).
    if (RW * RPT >= 32) asm volatile("; keep v79 allocated" ::: "v79");
    if (threadIdx.x < 64) {
        __builtin_amdgcn_s_setprio(2);
        env_wave<TASK, RW, RPT, ST>(H, A, S, e0, nE);
    } else {
        record_wave<TASK, RW, RPT, FAST, PF, ST>(H, A, S, e0, nE);
    }
}

// One kernel per tile shape, with the VGPR budget spelled out.  2048-record tiles run 4 blocks x 5 waves per
// CU at the headline size = 5 waves per SIMD on average, but a block's 5 waves land 2-1-1-1 on the SIMDs from
// a varying start, so one SIMD can be asked for a 6th: budget for 6 (80 VGPRs) or that block waits a whole
// round.  (The smaller tiles had 64 VGPRs for 8 waves per SIMD until round 5: see below.)
// (the leading 14 dwords of the arguments — everything up to and including turn_tab — arrive in SGPRs: build.py's
//  -amdgpu-kernarg-preload-count; PF: record loads in flight per lane, see record_wave)
#define EB_FUSED_KERNEL(NAME, RW, RPT, WAVES, VGPRS)                                                     \
    template <int TASK, bool FAST, int PF, typename ST>                                                  \
    __global__ __launch_bounds__((RW + 1) * 64, WAVES) __attribute__((amdgpu_num_vgpr(VGPRS))) void NAME( \
        const ST* obs_in, ST* obs_out, int n_env, int obs_dim, int n_veh, int envs_per_tile,             \
        unsigned nv_magic, int do_rewards, double* acc_rec, const unsigned char* turn_tab, const FusedArgs A) { \
        const FusedHot<ST> H{obs_in, obs_out, n_env, obs_dim, n_veh, envs_per_tile, nv_magic, do_rewards, acc_rec, turn_tab, \
                             wall_clock64()};                                                            \
        fused_body<TASK, RW, RPT, FAST, PF, ST>(H, A);                                                   \
    }
EB_FUSED_KERNEL(rollout_fused_4x8, 4, 8, 6, 80)
EB_FUSED_KERNEL(rollout_fused_4x4, 4, 4, 6, 80)   // (round 5: 64 -> 80 VGPRs — the env wave keeps three groups of table entries in flight;
EB_FUSED_KERNEL(rollout_fused_1x4, 1, 4, 6, 80)   //  these tiles run on grids of a few blocks per CU: latency, not occupancy)
// (round 5, measured and dropped: the 2048-record tile on THREE waves — two record waves x 16 records per lane, 117 VGPRs, the same bits —
//  so that a step launches 3 072 waves instead of 5 120: 16.8 vs 15.3 us at 65 536 x 32, 11.7 vs 9.5 at 32 768; profiles/r5_ab_tile3.txt)

#define EB_TAPE_KERNEL(NAME, RW, RPT, GATED, WAVES)                                                      \
    template <int TASK, bool FAST, typename ST>                                                          \
    __global__ __launch_bounds__((RW + (GATED ? 3 : 1)) * 64, WAVES) void NAME(                          \
        const ST* obs_in, ST* obs_out, int n_env, int obs_dim, int n_veh, int envs_per_tile,             \
        unsigned nv_magic, int horizon, const FusedArgs A) {                                             \
        const FusedHot<ST> H{obs_in, obs_out, n_env, obs_dim, n_veh, envs_per_tile, nv_magic, HOT_REWARDS, nullptr, nullptr, 0ll}; \
        tape_body<TASK, RW, RPT, FAST, GATED, ST>(H, A, horizon);                                        \
    }
EB_TAPE_KERNEL(rollout_tape_4x8, 4, 8, false, 6)
EB_TAPE_KERNEL(rollout_tape_4x4, 4, 4, false, 1)
EB_TAPE_KERNEL(rollout_tape_1x4, 1, 4, false, 1)
// the same with step gates (eb_rollout_gated): its whole grid has to be resident, so it may as well take the registers
EB_TAPE_KERNEL(rollout_gated_4x8, 4, 8, true, 4)
EB_TAPE_KERNEL(rollout_gated_4x4, 4, 4, true, 1)
EB_TAPE_KERNEL(rollout_gated_1x4, 1, 4, true, 1)

// The open-loop tape kernel's tile for a slot count that does not divide the 256 record lanes (the native 9 and 5): the 4 x 8 tile's per-record item / env / offset / turn state for eight records per lane does not fit six waves per SIMD
// (it spilled 1 008 bytes of scratch per lane, and unbounded it takes 256 VGPRs) — such tapes run on the 4 x 4 tile (93 VGPRs, no
// scratch; every tile shape computes the same bits).  That instantiation of rollout_tape_4x8 does not exist.
int tape_tile_variant(int variant, int n_veh, int storage_f16) {
    (void)storage_f16;   // (the binary16 instantiation spilled too — 8 bytes, task `straight` — and goes the same way)
    return (variant == 0 && (4 * 64) % n_veh != 0) ? 1 : variant;
}

int fused_tile_records(int variant) {
    switch (variant) {
        case 0: return 4 * 64 * 8;
        case 1: return 4 * 64 * 4;
        default: return 1 * 64 * 4;
    }
}

#define EB_HOT_ARGS(ST) reinterpret_cast<const ST*>(A.obs_in), reinterpret_cast<ST*>(A.obs_out), A.n_env, A.obs_dim, A.n_veh, \
                        A.envs_per_tile, A.nv_magic, (A.do_rewards ? HOT_REWARDS : 0) | (A.by_progress ? HOT_PRIO : 0), A.acc_rec, \
                        reinterpret_cast<const unsigned char*>(A.dt) + offsetof(PathTables, turn)
#define EB_LAUNCH_TASK(KERNEL, FAST_, PF_, ST)                                                                  \
    switch (task) {                                                                                             \
        case TASK_LEFT: hipLaunchKernelGGL((KERNEL<TASK_LEFT, FAST_, PF_, ST>), g, b, 0, s, EB_HOT_ARGS(ST), A); break; \
        case TASK_STRAIGHT: hipLaunchKernelGGL((KERNEL<TASK_STRAIGHT, FAST_, PF_, ST>), g, b, 0, s, EB_HOT_ARGS(ST), A); break; \
        default: hipLaunchKernelGGL((KERNEL<TASK_RIGHT, FAST_, PF_, ST>), g, b, 0, s, EB_HOT_ARGS(ST), A); break; \
    }
#define EB_LAUNCH_FAST(KERNEL, RW, PF_, ST)                                                                     \
    if ((RW * 64) % A.n_veh == 0) { EB_LAUNCH_TASK(KERNEL, true, PF_, ST) } else { EB_LAUNCH_TASK(KERNEL, false, PF_, ST) }
#define EB_LAUNCH(KERNEL, RW, PF_)                                                                              \
    {                                                                                                           \
        const dim3 g(grid), b((RW + 1) * 64);                                                                   \
        if (A.storage_f16) { EB_LAUNCH_FAST(KERNEL, RW, PF_, _Float16) } else { EB_LAUNCH_FAST(KERNEL, RW, PF_, float) } \
    }

#define EB_TAPE_ARGS(ST) reinterpret_cast<const ST*>(A.obs_in), reinterpret_cast<ST*>(A.obs_out), A.n_env, A.obs_dim, A.n_veh, \
                         A.envs_per_tile, A.nv_magic, horizon
#define EB_TAPE_TASK(KERNEL, FAST_, ST)                                                                         \
    switch (task) {                                                                                             \
        case TASK_LEFT: hipLaunchKernelGGL((KERNEL<TASK_LEFT, FAST_, ST>), g, b, dyn, s, EB_TAPE_ARGS(ST), A); break; \
        case TASK_STRAIGHT: hipLaunchKernelGGL((KERNEL<TASK_STRAIGHT, FAST_, ST>), g, b, dyn, s, EB_TAPE_ARGS(ST), A); break; \
        default: hipLaunchKernelGGL((KERNEL<TASK_RIGHT, FAST_, ST>), g, b, dyn, s, EB_TAPE_ARGS(ST), A); break;    \
    }
#define EB_TAPE_FAST(KERNEL, RW, ST)                                                                            \
    if ((RW * 64) % A.n_veh == 0) { EB_TAPE_TASK(KERNEL, true, ST) } else { EB_TAPE_TASK(KERNEL, false, ST) }
#define EB_TAPE_LAUNCH(KERNEL, RW, EXTRA_WAVES)                                                                 \
    {                                                                                                           \
        const dim3 g(grid), b((RW + 1 + EXTRA_WAVES) * 64);                                                               \
        const size_t dyn = (size_t)A.stage_entries * 12;                                                        \
        if (A.storage_f16) { EB_TAPE_FAST(KERNEL, RW, _Float16) } else { EB_TAPE_FAST(KERNEL, RW, float) }       \
    }

#define EB_TAPE_OCC(KERNEL, RW)                                                                                  \
    {                                                                                                           \
        int nb = 0;                                                                                             \
        const bool fast = (RW * 64) % n_veh == 0;                                                               \
        hipError_t e;                                                                                           \
        if (storage_f16) e = fast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, KERNEL<TASK_LEFT, true, _Float16>, (RW + 3) * 64, dyn_bytes) \
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, KERNEL<TASK_LEFT, false, _Float16>, (RW + 3) * 64, dyn_bytes); \
        else e = fast ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, KERNEL<TASK_LEFT, true, float>, (RW + 3) * 64, dyn_bytes)       \
                      : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, KERNEL<TASK_LEFT, false, float>, (RW + 3) * 64, dyn_bytes);     \
        return e == hipSuccess ? nb : 0;                                                                        \
    }
// resident blocks per CU of the tape kernel (the three tasks compile to the same resources; `left` stands for all)
int tape_blocks_per_cu(int task, int variant, int n_veh, int storage_f16, size_t dyn_bytes) {
    (void)task;
    switch (variant) {
        case 0: EB_TAPE_OCC(rollout_gated_4x8, 4)
        case 1: EB_TAPE_OCC(rollout_gated_4x4, 4)
        default: EB_TAPE_OCC(rollout_gated_1x4, 1)
    }
}

// eb_gate_feed: the reference producer of a gated rollout — one block that hands the staged action tape over step by
// step: waits until every block of the rollout has reported step t - 1, copies actions[t] (written through), raises
// gate_ready[t].  What a policy kernel in the loop does, minus the policy.
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gate_feed_kernel(int horizon, unsigned n_blocks, size_t step_words, const u4v* __restrict__ staged,
                                                        u4v* live, unsigned* gate_ready, const unsigned* gate_done,
                                                        unsigned* status, int spin) {
    constexpr int PRE = 8;   // 16-byte words per thread fetched from the staged tape BEFORE the wait (the "policy" is instantaneous)
    for (int t = 0; t < horizon; ++t) {
        const u4v* src = staged + (size_t)t * step_words;
        u4v* dst = live + (size_t)t * step_words;
        u4v pre[PRE];
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const size_t i = threadIdx.x + (size_t)k * blockDim.x;
            pre[k] = src[i < step_words ? i : 0];
        }
        int good = 1;
        if (t > 0) {                                     // word 0 of every block's 64-byte record of step t - 1, 256 at a time
            for (unsigned b = threadIdx.x; b < n_blocks && good; b += blockDim.x) {
                const unsigned* w = gate_done + ((size_t)(t - 1) * n_blocks + b) * 16;
                for (int spins = 0; !agent_load_u32(w); ++spins) {
                    if (spins >= spin || ((spins & 31) == 31 && agent_load_u32(status))) { good = 0; break; }   // or the rollout gave up
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        if (!__syncthreads_and(good)) {
            if (threadIdx.x == 0) agent_store_u32(status + 1, 1u);
            return;
        }
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const size_t i = threadIdx.x + (size_t)k * blockDim.x;
            if (i < step_words) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst + i), "v"(pre[k]) : "memory");
        }
        for (size_t i = threadIdx.x + (size_t)PRE * blockDim.x; i < step_words; i += blockDim.x) {   // long steps: the rest
            const u4v v = src[i];
            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst + i), "v"(v) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) agent_store_u32(gate_ready + t, 1u);
    }
}

hipError_t launch_gate_feed(int horizon, int n_blocks, size_t step_bytes, const void* staged, void* live,
                            unsigned* gate_ready, const unsigned* gate_done, unsigned* status, int spin, hipStream_t s) {
    hipLaunchKernelGGL(gate_feed_kernel, dim3(1), dim3(256), 0, s, horizon, (unsigned)n_blocks, step_bytes / 16,
                       reinterpret_cast<const u4v*>(staged), reinterpret_cast<u4v*>(live), gate_ready, gate_done, status, spin);
    return hipGetLastError();
}

// A.actions = the tape [horizon, n_env, 2], A.out5 = [horizon, 5, n_env]
// a mark buffer (eb_debug_set_trace) that does not hold a row of 8 words for every wave of the launch is dropped: the kernels do not test
static FusedArgs trace_checked(const FusedArgs& A_in, int grid, int waves_per_block) {
    FusedArgs A = A_in;
    if (A.trace && A.trace_words < (long long)grid * waves_per_block * 8) A.trace = nullptr;
    return A;
}

hipError_t launch_rollout_tape_fused(int task, int variant, const FusedArgs& A_in, int horizon, int grid, hipStream_t s) {
    const FusedArgs A = trace_checked(A_in, grid, (variant == 2 ? 1 : 4) + 1);   // (rows are indexed by blockIdx.x * (RW + 1) + wave: the couriers do not mark)
    if (A.gate_ready) {
        switch (variant) {
            case 0: EB_TAPE_LAUNCH(rollout_gated_4x8, 4, 2) break;
            case 1: EB_TAPE_LAUNCH(rollout_gated_4x4, 4, 2) break;
            default: EB_TAPE_LAUNCH(rollout_gated_1x4, 1, 2) break;
        }
        return hipGetLastError();
    }
    switch (variant) {
        case 0: {
            const dim3 g(grid), b((4 + 1) * 64);
            const size_t dyn = (size_t)A.stage_entries * 12;
            if ((4 * 64) % A.n_veh != 0) return hipErrorInvalidValue;            // the host routes these to the 4 x 4 tile (tape_tile_variant)
            if (A.storage_f16) { EB_TAPE_TASK(rollout_tape_4x8, true, _Float16) } else { EB_TAPE_TASK(rollout_tape_4x8, true, float) }
        } break;
        case 1: EB_TAPE_LAUNCH(rollout_tape_4x4, 4, 0) break;
        default: EB_TAPE_LAUNCH(rollout_tape_1x4, 1, 0) break;
    }
    return hipGetLastError();
}

hipError_t launch_rollout_fused(int task, int variant, const FusedArgs& A_in, int grid, hipStream_t s) {
    const FusedArgs A = trace_checked(A_in, grid, (variant == 2 ? 1 : 4) + 1);
    // (A.rolling / A.by_progress: decided by the grid and the tile's envs — eb_capi.hip:rollout_fused — or forced,
    // eb_debug_set_rollout_sched; every combination computes the same bits.  Rolling loads exist for the 2048-record tile only:
    // four records per lane leave little to roll, and the 256-record tile runs on grids that are launch-bound either way)
    switch (variant) {
        case 0: if (A.rolling) EB_LAUNCH(rollout_fused_4x8, 4, 3) else EB_LAUNCH(rollout_fused_4x8, 4, 8) break;
        case 1: EB_LAUNCH(rollout_fused_4x4, 4, 4) break;
        default: EB_LAUNCH(rollout_fused_1x4, 1, 4) break;
    }
    return hipGetLastError();
}

}  // namespace eb
