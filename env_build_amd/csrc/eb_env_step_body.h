// eb_env_step_body.h — the kernels of eb_env_step.hip (one translation unit per task instantiates them: eb_env_step.hip, _t1, _t2).
// CrossroadEnd2end.step (E2E:132-144) for a batch of envs as ONE launch, gfx950.
//
// eb_env_step used to be four launches (action scaling + reward + ego step | traffic step | observation + done code |
// pool re-entry): 62.5 MB of algorithmic traffic in 76-81 us at 65 536 envs x 16 candidates, i.e. ~10 % of the HBM peak —
// the candidates crossed HBM three times, the old observation was read by one thread per row (64 cache lines per load
// instruction), and every launch boundary cost its ~2 us.  Here a block owns a tile of 64 envs for the whole step, every
// record crosses HBM once in each direction, and the work is laid out so that no phase is one lane per env walking its
// candidates (the first one-launch version was: its slot-building phase alone took 8-10 us per block, a chain of
// dependent LDS reads and divergent branches):
//
//   phase 1   wave 0, lane = env: ego state + raw action -> action scaling (E2E:133), ego step (E2E:135: f_xu core, floor,
//             wrap) -> new pose to LDS (to HBM after the barrier: wave 1 reads the old row).   wave 1, lane = env: old observation head -> ego circle centres
//             (DAM:210-214) to LDS; tyre parameters (DAM:65-71, two atan) -> HBM; reward scalars and road walls (E2E:134).
//             Whoever is free: the traffic step (TRF:220-238's role) — 16-byte candidate records in coalesced chunks,
//             predict_for_a_mode, staged in LDS with their mode byte.                                       barrier
//   phase 2   wave 0: closest point through the cell grid + tracking error (E2E:293-297) -> observation row head.
//             waves 1-3, one lane per (env, old slot): compute_rewards' vehicle terms (DAM:218-229) — a centre-distance
//             test first, the few pairs inside 6.364 m are compacted (ballot / mbcnt) into a per-wave queue and evaluated
//             densely, everything else contributes exact zeros.  One lane per (env, candidate): the range filter of
//             E2E:393-411 as a table-driven, branch-free test -> a tag byte (mode or 0xFF); the 10 m box test of
//             TRF:263-295 -> per-wave queue -> two-circle test -> per-env collision flag.                   barrier
//   phase 3   (no barrier in front of it) the distinct slot modes come off a counter in LDS: a wave takes the next one when its
//             phase-2 work is done; per mode (wave-uniform) and env (lane): the tag row
//             becomes a 64-bit candidate set (4 tags per LDS dword, zero-byte trick), the mode's slots are filled by
//             repeated selection over that set (E2E:414-437; typically 0-3 members).  wave 1 first adds up the penalty
//             partials in vehicle order -> out5 / dict16, then the done predicates that need only the ego (E2E:223-256).
//                                                                                                           barrier
//   phase 4   wave 0: priority chain -> done code (E2E:200-221).  all: observation rows -> HBM (coalesced); candidates ->
//             HBM with the pool's re-entry rule applied on the way out (eb_traffic_respawn).
//
// Nothing but the tile's LDS is shared between waves: no cross-block traffic, no XCD consideration beyond "a tile's lines
// belong to one workgroup".  The arithmetic is the same device functions the single-entry kernels run (eb_env_device.h,
// eb_device.h): bit-identical to the six (seven) calls — tests/_env_step_check.py holds it to that.
#pragma once
#include "eb_env_device.h"

#pragma clang fp contract(off)

namespace eb {

typedef float f4a4 __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment (obs rows: D is odd)
typedef __attribute__((address_space(3))) int lds_int;
typedef float f2a4 __attribute__((ext_vector_type(2), aligned(4)));    // 8-byte access at 4-byte alignment

EB_DEV int es_obs_stride(int D) { return D | 1; }                       // floats per LDS row, odd: no bank conflicts
EB_DEV int es_tag_stride4(int m_cand) { return ((m_cand + 3) >> 2) | 1; }   // dwords per tag row, odd
EB_DEV int fast_div(int item, unsigned magic) { return magic ? (int)__umulhi((unsigned)item, magic) : item; }   // magic 0: / 1
constexpr int ES_QCAP = 128;   // per-wave queue: flushed whenever 64 entries are waiting, so 64 + 64 suffice

// profiling aid (eb_debug_set_trace on the model handle): slot k of this wave's row [16] <- the 100 MHz wall clock, lane 0 only
// (rows of 16 words, one per wave: [n_blocks * NW][16] — NW is 4 or 8 by grid size, so a caller sizes the buffer for 8 and passes its
// capacity; launch_env_step drops a buffer that is too small for the launch at hand.  No bounds test in here: one — even on 32-bit
// indices — cost the step kernel two VGPRs, 95 -> 97, i.e. a wave of occupancy: 18.5 -> 26.7 us at 65 536 envs)
#define ES_MARK(k) do { if (A.trace && (threadIdx.x & 63) == 0) A.trace[((size_t)blockIdx.x * NW + (threadIdx.x >> 6)) * 16 + (k)] = wall_clock64(); } while (0)

// A per-wave queue of 16-bit item ids: `hit` lanes append (ballot / mbcnt), and whenever 64 are waiting the wave runs
// `body(item)` on a full set of lanes; flush() runs the rest.
template <class Body>
struct WaveQueue {
    unsigned short* q;
    int n;
    Body body;
    EB_DEV void push(bool hit, int item) {
        const unsigned long long b = __builtin_amdgcn_ballot_w64(hit);
        if (b) {
            if (hit) {
                const int pos = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, (unsigned)n));
                q[pos] = (unsigned short)item;
            }
            n += __popcll(b);
            if (n >= 64) { run(64); }
        }
    }
    EB_DEV void run(int count) {
        const int lane = threadIdx.x & 63;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int item = lane < count ? q[lane] : -1;
        const int rest = lane < n - count ? q[count + lane] : 0;    // (n - count <= 64: push flushes at >= 64)
        if (item >= 0) body(item);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane < n - count) q[lane] = (unsigned short)rest;
        n -= count;
    }
    EB_DEV void flush() { if (n > 0) run(n); }
};

// The first two of a candidate set under (key, insertion index) for ONE mode known at compile time (E2E:393-437): the mode's range
// filter and sort key fold to the two or three comparisons they are (veh_in_range / key_of with constant arguments), the running
// pair is kept as (key, index) and updated by selects — straight-line code, one round per set bit of the wave's largest set; only
// (x, y) of a candidate is read in the loop, the winners' records at the end.  The first version re-ran the mode switch and a nest
// of divergent branches per candidate (~850 cycles each); a data-driven straight-line version still spent ~75 instructions on
// flags and selects per candidate.  `crow`: this lane's candidate row in LDS; virt: the stop-line car exists for this env.
template <int TASK, int MODE>
EB_DEV void slot_pair_walk(const float4* crow, unsigned long long elig, float ex, float ey, bool virt, int m_cand, float* ov, int sa, int sb) {
    const KeySpec ks = key_spec(TASK, MODE);
    const V4 fill = veh_fill_value(MODE);
    float2 k1 = make_float2(0.0f, 0.0f), k2 = k1;
    int i1 = -1, i2 = -1;
    // candidates arrive in ascending index: c sorts before an earlier one only with a strictly smaller key
    auto offer = [&](const bool valid, const float2 kk, const int c) {
        const bool first = valid & ((i1 < 0) | key_less(kk, k1));
        const bool second = valid & !first & ((i2 < 0) | key_less(kk, k2));
        k2.x = first ? k1.x : (second ? kk.x : k2.x); k2.y = first ? k1.y : (second ? kk.y : k2.y);
        i2 = first ? i1 : (second ? c : i2);
        k1.x = first ? kk.x : k1.x; k1.y = first ? kk.y : k1.y;
        i1 = first ? c : i1;
    };
    const float2* cxy = reinterpret_cast<const float2*>(crow);           // (x, y) of candidate c at cxy[2 * c]
    {   // the first four members of the set without a loop: their indices first, then their (x, y) all in flight together, then the
        // four offers — no branch, no LDS round trip per candidate (a mode rarely has more: the loop below takes the rest)
        constexpr int UNR = 4;
        int cs[UNR];
        bool hs[UNR];
        float2 q[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            hs[u] = elig != 0ull;
            cs[u] = hs[u] ? __builtin_ctzll(elig) : 0;
            elig &= elig - 1ull;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) q[u] = cxy[2 * cs[u]];
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            offer(hs[u] & veh_in_range(TASK, MODE, V4{q[u].x, q[u].y, 0.0f, 0.0f}, ex, ey), key_of(ks, q[u].x, q[u].y), cs[u]);
    }
    bool has = elig != 0ull;
    int c = has ? __builtin_ctzll(elig) : 0;
    elig &= elig - 1ull;
    while (__builtin_amdgcn_ballot_w64(has)) {
        const float2 q = cxy[2 * c];
        const bool hq = has;
        const int cq = c;
        has = elig != 0ull;
        c = has ? __builtin_ctzll(elig) : 0;
        elig &= elig - 1ull;
        offer(hq & veh_in_range(TASK, MODE, V4{q.x, q.y, 0.0f, 0.0f}, ex, ey), key_of(ks, q.x, q.y), cq);
    }
    const float4 vv4 = make_float4(MODE == EB_VMODE_DL ? LANE_W / 2 : LANE_W * 1.5f, -HALF_CROSS + 2.5f, 0.0f, 90.0f);
    if (MODE == EB_VMODE_DL || MODE == EB_VMODE_DU)                      // the stop-line car (E2E:386-390), index m_cand: after every real one
        offer(virt & veh_in_range(TASK, MODE, V4{vv4.x, vv4.y, 0.0f, 90.0f}, ex, ey), key_of(ks, vv4.x, vv4.y), m_cand);
    const float4 fill4 = make_float4(fill.x, fill.y, fill.v, fill.phi);   // slice_or_fill, E2E:431-437
    const float4 q1 = crow[i1 < 0 || i1 >= m_cand ? 0 : i1], q2 = crow[i2 < 0 || i2 >= m_cand ? 0 : i2];
    const float4 r1 = i1 < 0 ? fill4 : (i1 >= m_cand ? vv4 : q1), r2 = i2 < 0 ? fill4 : (i2 >= m_cand ? vv4 : q2);
    *reinterpret_cast<f4a4*>(ov + 4 * sa) = f4a4{r1.x, r1.y, r1.z, r1.w};
    if (sb >= 0) *reinterpret_cast<f4a4*>(ov + 4 * sb) = f4a4{r2.x, r2.y, r2.z, r2.w};
}

// ET: envs per tile (64 or 16); lanes >= ET of the per-env roles idle.  OBS: the observation alone (eb_get_obs on an ego and
// candidates given as they are: no action, reward, ego step, traffic step, collision test or done code — phases 1-4 shrink to
// staging, tracking, slots and the row store; the arithmetic of what remains is the same code).
// RESET (with OBS): eb_env_reset_pool in one launch — the masked rows get eb_env_reset's draws (wave 0, lane = env), then a fresh
// pool clear of that ego (the staging lanes, eb_traffic_respawn's arithmetic with init_traffic's conflict rule), then their
// observation from the state just made; the drawn virtual-red-light flag replaces the old one at the end (E2E:116-126).
// AUTO (step only): eb_env_step(auto_reset) — the rows whose done code came out non-zero take RESET's path in the same block after
// the step's own phases: terminal observation -> final_obs, draws, pool re-entry clear of the new ego, reset observation (OLD
// flag) -> obs_out, flag swap.  A tile without a finished row leaves after phase 4 as before.
// NW: waves per block.  4: the roles share four waves (a wave walks a slot mode after its phase-2 work).  8 (small and medium
// batches, where a block has its CU nearly to itself and a wave's instruction stream IS the step's duration): waves 4-7 own the
// slot modes and walk them right after barrier 1, beside the tracking (wave 0), the reward pairs (wave 1) and the collision pass
// (waves 2, 3); the staging is spread over six waves.
template <int TASK, int ET, bool OBS, bool RESET, bool AUTO = false, int NW = 4>
EB_DEV void env_step_body(const EnvStepArgs A) {
    constexpr int NT = NW * 64;                                                  // threads per block
    constexpr int KS = NW == 4 ? 3 : 2;                                          // chunks of a group per staging wave
    constexpr int GCH = (NW - 2) * KS + 2, GREC = GCH * 64;                      // chunks / records per staging group (8 / 512, 14 / 896)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint8_t smode[64], sturn[64], s_col[ET], s_jb[ET];
    __shared__ float s_vmax[(ET == 16 && NW == 4 && !OBS) ? 64 : 1];                    // eb_flow_rule's v_max per slot (the fused flow rule of the 16-env tiles)
    __shared__ int s_modeq;                                                      // the next distinct mode to be walked (fill_slots)
    // tiles of up to 32 envs (LDS to spare): the candidates of every (env, mode) as a 64-bit set, OR-ed together by the staging lanes —
    // the slot pass reads ONE word pair per (env, mode) instead of scanning the env's mode bytes (15 dwords at 60 candidates)
    constexpr bool ELIG = ET <= 32;
    __shared__ unsigned s_elig32[ELIG ? ET * EB_VMODE_COUNT * 2 : 2];
    __shared__ float s_miu[ET], s_r[ET];                                         // miu_r / yaw rate of the step (the stability predicate's inputs)
    // AUTO: the start state a reset would give every env of the tile (drawn at kernel start, under the latency of the first loads),
    // the tile's finished envs as a list, the slot plan as a table
    __shared__ float4 s_rst[AUTO ? ET : 1];                                      // (x, y, phi, v_x)
    __shared__ unsigned s_rflag[AUTO ? ET : 1];                                  // path | drawn virtual-red-light flag << 2
    __shared__ uint8_t s_finlist[AUTO ? ET : 1];
    __shared__ unsigned s_dm[(AUTO || (ET == 16 && !OBS)) ? EB_VMODE_COUNT : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, e0 = blockIdx.x * ET;
    const int n_env = A.n_env, D = A.D, NV = A.NV, m_cand = A.m_cand, n_future = A.n_future;
    const int nE = n_env - e0 < ET ? n_env - e0 : ET;
    const int i = e0 + lane;
    const bool live = lane < nE && !(OBS && A.row_mask && A.row_mask[i] == 0);   // (a masked observation pass: the other rows are left alone)
    if (OBS && A.row_mask && __builtin_amdgcn_ballot_w64(live) == 0ull) {         // same lanes -> envs in every wave: the whole block leaves
        if (RESET) {                                                              // (its rows are carried over from the previous arrays)
            if (A.obs)
                for (int idx = tid; idx < nE * A.D; idx += NT) A.obs_out[(size_t)e0 * A.D + idx] = A.obs[(size_t)e0 * A.D + idx];
            if (A.done_src && A.done_code && tid < nE) A.done_code[e0 + tid] = A.done_src[e0 + tid];
        }
        return;
    }
    const int RS4 = obs_cand_stride4(m_cand), OS = es_obs_stride(D), TS4 = es_tag_stride4(m_cand), T = 3 * (n_future + 1);
    float4* s_cand = reinterpret_cast<float4*>(smem);                            // [64][RS4] candidates after the traffic step
    float* s_out = reinterpret_cast<float*>(s_cand + (size_t)ET * RS4);          // [64][OS]  next observation rows
    float2* s_part = reinterpret_cast<float2*>(s_out + (size_t)ET * OS);         // [64][NV]  (veh2veh4training, veh2veh4real) per old slot
    float4* s_pts = reinterpret_cast<float4*>(s_part + (size_t)ET * NV);         // [64]      old ego circle centres (DAM:210-214)
    float4* s_ego = s_pts + ET;                                                  // [64]      new ego (x, y, phi, v_x)
    float2* s_oldc = reinterpret_cast<float2*>(s_ego + ET);                      // [64]      old ego centre (obs columns 3, 4)
    unsigned* s_tag32 = reinterpret_cast<unsigned*>(s_oldc + ET);                // [64][TS4] mode bytes, then range tags
    uint8_t* s_tag = reinterpret_cast<uint8_t*>(s_tag32);
    unsigned short* s_queue = reinterpret_cast<unsigned short*>(s_tag32 + (size_t)ET * TS4);   // [4][ES_QCAP]
    // eb_flow_rule (flow_on): per (env, route) the vehicle an emission puts into the route's first vacant slot and that slot (or
    // -1), per slot "a vehicle is here after the exit test"
    float4* s_new = reinterpret_cast<float4*>(s_queue + 4 * ES_QCAP);           // [64][12]   (not on 16-env tiles)
    int* s_emit = reinterpret_cast<int*>(s_new + (size_t)((ET == 16 && NW == 4) ? 0 : ET) * 12);               // [64][12]   (not on 16-env tiles x four waves)
    uint8_t* s_on = reinterpret_cast<uint8_t*>(s_emit + (size_t)((ET == 16 && NW == 4) ? 0 : ET) * 12);        // [64][m_cand]
    constexpr bool EVEN = ET == 16 && NW == 4;
    constexpr bool PAIR_STEP = ET == 16 && !OBS && !RESET;   // the step's slot phase as (env, mode) pairs per lane (pair_walk below)
    constexpr bool FUSED_FLOW = EVEN && !OBS;   // the flow rule's per-slot part inside the staging of a record (below) instead of a pass of its own
    ES_MARK(0);
    // Loads first, all of them — the bytes of the small tables the block keeps in LDS (they come from the kernel-argument segment: a
    // memory round trip, and loads return in order, so they go out AHEAD of the records), then the candidate records of this thread
    // (16 bytes each, consecutive threads on consecutive records), further down the per-env flags and — lane = slot — the slot modes.
    // The table set-up below and its barrier wait for the table bytes and LDS only: the records stay in flight across it (they used to
    // be issued behind that barrier — a round trip of every block spent with nothing else outstanding), and the role work of waves
    // 0 / 1 runs under their latency.
    unsigned tb_mode = 0, tb_turn = 0, tb_dm = 0;
    if (tid < 64) { tb_mode = A.modes.mode[tid]; tb_turn = A.tturn.t[tid]; }
    if ((AUTO || (ET == 16 && !OBS)) && tid < EB_VMODE_COUNT) tb_dm = A.dm[tid];
    float tb_vmax = 0.0f;                                                             // the flow rule's per-slot speed limit: read per record
    if (FUSED_FLOW && A.flow_on && tid < m_cand) tb_vmax = A.flow_v_max[tid];          // (from global memory it was a round trip — and a wait for the previous record's stores — in every staged chunk)
    const bool tb_skip = OBS && A.row_mask && tid < ET && !(tid < nE && A.row_mask[e0 + tid] != 0);   // OBS: a row not to be written
    asm volatile("" ::: "memory");   // (program order of the loads = issue order)
    const float4* csrc = reinterpret_cast<const float4*>(A.cand) + (size_t)e0 * m_cand;
    const uint8_t* msrc = A.cand_mode + (size_t)e0 * m_cand;
    const int n_rec = nE * m_cand;
    // chunks of 64 records in groups of eight: waves 2 and 3 take three chunks of a group each, waves 0 and 1 — which
    // have the ego step and the tyre parameters to do — one each.  16-env tiles on four waves (many candidates per env: the flow
    // source's 60 are 15 chunks a tile): waves 0 / 1 stood at barrier 1 for 3 us of the tile's 15 while 2 / 3 staged six chunks each.
    // They share every group evenly, two chunks per wave (4 / 4 / 4 / 3): two chunk registers less per lane are what lets the kernel
    // fit 80 VGPRs = SIX tiles per CU with the 25 KB of LDS a tile is down to (53 against 56-58 us at 65 536 x 60: the kernel waits
    // 58 % of its wave cycles, a sixth tile is a sixth more in flight).  (Sharing only the first group evenly — 3 / 2 / 5 / 5, waves
    // 0 / 1 have their per-env chains too — is 1 us less of phase 1 per tile, but costs those two registers.)
    auto rec_index = [&](int group, int k) -> int {
        if (EVEN) return k < 2 ? (group * GCH + wave + NW * k) * 64 + lane : -1;
        const int chunk = wave >= 2 ? (wave - 2) + (NW - 2) * k : (k == 0 ? (NW - 2) * KS + wave : -1);   // (NW = 4: 0 2 4 / 1 3 5 / 6 / 7)
        return chunk < 0 || k >= KS ? -1 : (group * GCH + chunk) * 64 + lane;
    };
    float4 cv[2][3];
    unsigned cm[2][3];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int idx = rec_index(g, k);
            cv[g][k] = make_float4(0, 0, 0, 0); cm[g][k] = EB_VMODE_EMPTY;
            if (idx >= 0 && idx < n_rec) {
                cv[g][k] = csrc[idx]; cm[g][k] = msrc[idx];
                if (!OBS && A.flow_on) cm[g][k] |= (unsigned)A.flow_active[(size_t)e0 * m_cand + idx] << 8;   // (the flow rule's flag: bits 8..)
            }
        }
    if (tid == 0) s_modeq = 0;
    if (ELIG)
        for (int w = tid; w < ET * EB_VMODE_COUNT * 2; w += NT) s_elig32[w] = 0u;
    for (int w = tid; w < ET * TS4; w += NT) s_tag32[w] = 0xffffffffu;                // padding bytes never match a mode
    if (tid < 64) { smode[tid] = (uint8_t)tb_mode; sturn[tid] = (uint8_t)tb_turn; }
    if ((AUTO || (ET == 16 && !OBS)) && tid < EB_VMODE_COUNT) s_dm[tid] = tb_dm;
    if (tid < ET) s_col[tid] = tb_skip ? 1 : 0;                                       // OBS: 1 = row not to be written
    if (FUSED_FLOW && tid < 64) s_vmax[tid] = tb_vmax;
    // Issue priority by phase (A.by_progress): a block that is still in phase 1 outranks one that is past barrier 1, and that one a
    // block that is storing its rows — the launch ends with its LAST block (as the rollout kernel's priority by progress, round 6)
    const bool by_phase = A.by_progress != 0;
    if (by_phase) __builtin_amdgcn_s_setprio(2);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: the record loads stay in flight

    // ---- phase 1 ---------------------------------------------------------------------------------------------
    // (wave 1) the first batch of compute_rewards' (env, old slot) pairs: their (x, y) are needed in phase 2 only.  Phase 2's two
    // pair-parallel passes have one owner each — wave 1 the reward pairs, waves 2 and 3 the collision test — so that a wave pays
    // for ONE queue flush (a serial chain of ~250 instructions behind a sin / cos), not two
    const int n_pairs = nE * NV, pt_ = lane, ct_ = tid - 128;
    float2 pxy[4];
    auto load_pairs = [&](int base) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = base + pt_ + 64 * k;
            pxy[k] = make_float2(1e30f, 1e30f);                                  // (past the end: never near)
            if (!OBS && wave == 1 && p < n_pairs) {
                const int e = fast_div(p, A.nv_magic), j = p - e * NV;
                const f2a4 q = *reinterpret_cast<const f2a4*>(A.obs + (size_t)D * (e0 + e) + 6 + T + 4 * j);
                pxy[k] = make_float2(q.x, q.y);
            }
        }
    };
    load_pairs(0);
    const int slot_mode = lane < NV ? A.modes.mode[lane] : 0xff;                        // lane = slot
    const bool red_light = !RESET && live && A.v_light && A.v_light[i] != 0;   // (a reset clears v_light before its observation)
    // (wave 0) the env's path id now: the tracking chain of phase 2 starts with it
    const int path_pre = (!RESET && wave == 0 && live) ? row_path(A.pt, A.ref_idx, A.path_id, i) : -1;
    const bool vflag = live && A.virtual_flag && A.virtual_flag[i] != 0;
    const bool light = red_light || vflag;                                      // E2E:387-388
    // eb_time_limit (wave 3, which owns the ego-only done predicates): the episode's step count with this step in it
    int ep_cnt = 0;
    if (!OBS && wave == 3 && live && A.episode_step) ep_cnt = A.episode_step[i] + 1;
    float nx[6] = {0, 0, 0, 0, 0, 0};                                        // wave 0
    float steer = 0.0f, a_x = 0.0f, o9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};     // wave 1
    float road_t = 0.0f, road_r = 0.0f;
    int reset_path = 0;
    bool virtual_next = false;
    // eb_env_reset's draws for this lane's env (env_reset_kernel: same keys, same arithmetic) -> nx, reset_path, virtual_next, and
    // the new state to HBM and s_ego
    auto draw_values = [&](float (&st)[6], int& path, bool& vnext) {
        const float span = TASK == TASK_LEFT ? 900 + 500 : TASK == TASK_STRAIGHT ? 1200 + 500 : 420 + 500;   // E2E:473-478
        const uint64_t base = (A.reset_counter << 32) + (uint64_t)i * 128u;
        const float u0 = u01(A.reset_seed, base), u1 = u01(A.reset_seed, base + 1), u2 = u01(A.reset_seed, base + 2),
                    u3 = u01(A.reset_seed, base + 3);
        int p = (int)(u0 * (float)A.pt.n_paths);                        // DAM:591
        if (p > A.pt.n_paths - 1) p = A.pt.n_paths - 1;
        const int ci = clamp_index((int)(u1 * span) + 700, A.pt.len[p]);   // E2E:474-478; indexs2points, DAM:727-728
        st[0] = 8.0f * u2; st[1] = 0.0f; st[2] = 0.0f;                  // E2E:482-486
        st[3] = A.pt.x[p][ci]; st[4] = A.pt.y[p][ci]; st[5] = A.pt.phi[p][ci];
        path = p;
        vnext = A.training && u3 > 0.9f;                                // E2E:120-126
    };
    // what a reset writes besides the observation and the candidates (this lane's env; nx holds the drawn state), and the new pose
    auto store_reset_state = [&]() {
        float2* ego_out = reinterpret_cast<float2*>(A.ego + 6 * (size_t)i);
        ego_out[0] = make_float2(nx[0], nx[1]); ego_out[1] = make_float2(nx[2], nx[3]); ego_out[2] = make_float2(nx[4], nx[5]);
        reinterpret_cast<float4*>(A.params)[i] = make_float4(0.0f, 0.0f, VehParams::miu, VehParams::miu);   // E2E:110-113
        A.ref_idx_out[i] = reset_path;
        if (RESET && A.done_code) A.done_code[i] = EB_DONE_NOT_YET;     // E2E:119 (AUTO: done_code keeps the step's codes)
        if (RESET && A.episode_step) A.episode_step[i] = 0;             // (AUTO: the step itself has restarted the count)
        if (A.v_light_out && !(AUTO && A.flow_on)) A.v_light_out[i] = 0;   // (the flow source's reset sets its own light, below)
        s_ego[lane] = make_float4(nx[3], nx[4], nx[5], nx[0]);
    };
    auto draw_reset = [&]() {
        draw_values(nx, reset_path, virtual_next);
        store_reset_state();
    };
    // eb_traffic_respawn's unconditional re-entry of candidate c of tile row e, clear of the NEW ego in s_ego (init_traffic's
    // conflict rule, TRF:168-192) -> HBM and s_cand
    auto respawn_fresh = [&](int e, int c, const float4* pose) {
        const uint64_t ub = (A.pool_counter << 32) + (uint64_t)(e0 + e) * 128u + (uint64_t)c * 2u;
        const float u1 = u01(A.pool_seed, ub), u2 = u01(A.pool_seed, ub + 1);
        const float* en = A.pool_entry + 5 * c;
        float along = u1 * A.pool_span;
        float4 nv = make_float4(en[0] + along * en[3], en[1] + along * en[4], u2 * A.pool_v_max, en[2]);
        const float4 eg = pose[e];
        const float ego6[6] = {eg.w, 0.0f, 0.0f, eg.x, eg.y, eg.z};
        if (init_conflict(ego6, 4.8f, nv.x, nv.y, nv.w, nv.z, 4.8f)) {   // TRF:168-192: not on top of the ego
            along = u1 * A.edge_span;
            nv.x = en[0] + along * en[3];
            nv.y = en[1] + along * en[4];
        }
        reinterpret_cast<float4*>(A.cand)[((size_t)e0 + e) * m_cand + c] = nv;
        s_cand[e * RS4 + c] = nv;
    };
    // (waves 0, 1) the step's per-env inputs: issued here, ahead of what follows
    float2 in_r2 = make_float2(0.0f, 0.0f), in_g0 = in_r2, in_g1 = in_r2, in_g2 = in_r2;
    if (!OBS && wave < 2 && live) {
        in_r2 = reinterpret_cast<const float2*>(A.raw)[i];
        const float2* eg = reinterpret_cast<const float2*>(A.ego + 6 * (size_t)i);
        in_g0 = eg[0]; in_g1 = eg[1]; in_g2 = eg[2];
    }
    // (16-env tiles x four waves — the flow source's shape — draw in the tail instead, for the finished envs only: the speculation's
    // registers are live together with the chunk registers, and without it the auto-reset variant fits the 80 VGPRs of SIX tiles per
    // CU with no scratch; step + flow rule + reset 66.6 -> 61.7 us at 65 536 x 60)
    constexpr bool DRAW_LATE = ET == 16 && NW == 4;
    if (AUTO && !DRAW_LATE && wave == (NW == 8 ? NW - 1 : 0) && live) {   // (eight waves: the last one stages the fewest records)
        // Ahead of time, under the latency of the loads just issued: the start state a reset would give this env — the draws depend
        // on (seed, counter, env) alone.  If the step finishes the env, the tail finds its pose here instead of running eight 64-bit
        // multiplies and a dependent table read per draw behind the step.
        float rs[6];
        int rpath;
        bool rvn;
        draw_values(rs, rpath, rvn);
        s_rst[lane] = make_float4(rs[3], rs[4], rs[5], rs[0]);
        s_rflag[lane] = (unsigned)rpath | (rvn ? 4u : 0u);
    }
    if (OBS) {
        if (wave == 0 && live) {
            if (RESET) draw_reset();
            else {                                                              // the ego as given (eb_get_obs)
                const float2* eg = reinterpret_cast<const float2*>(A.ego + 6 * (size_t)i);
                const float2 g0 = eg[0], g1 = eg[1], g2 = eg[2];
                nx[0] = g0.x; nx[1] = g0.y; nx[2] = g1.x; nx[3] = g1.y; nx[4] = g2.x; nx[5] = g2.y;
                s_ego[lane] = make_float4(nx[3], nx[4], nx[5], nx[0]);
            }
        }
        if (RESET) __syncthreads();   // the pool's re-entry below stays clear of the NEW ego
    } else if (wave < 2 && live) {
        const float2 r2 = in_r2, g0 = in_g0, g1 = in_g1, g2 = in_g2;
        const float st[6] = {g0.x, g0.y, g1.x, g1.y, g2.x, g2.y};
        if (wave == 1) {
            const float* o = A.obs + (size_t)D * i;
            const f4a4 a = *reinterpret_cast<const f4a4*>(o), b = *reinterpret_cast<const f4a4*>(o + 4);
            o9[0] = a.x; o9[1] = a.y; o9[2] = a.z; o9[3] = a.w; o9[4] = b.x; o9[5] = b.y; o9[6] = b.z; o9[7] = b.w; o9[8] = o[8];
        }
        action_transform(r2.x, r2.y, steer, a_x);                              // E2E:133
        if (wave == 0) {
            // E2E:135 = env_ego_step_row without the tyre parameters (wave 1 has those)
            const float phi_rad = deg2rad(st[5]);
            float sn, cs;
            sincos_det(phi_rad, sn, cs);
            f_xu_core(st, steer, a_x, TAU10, phi_rad, sn, cs, nx);             // E2E:279
            nx[0] = nx[0] >= 0.0f ? nx[0] : 0.0f;                              // E2E:281
            nx[5] = wrap_deal_with_phi(nx[5]);                                 // E2E:282
            s_ego[lane] = make_float4(nx[3], nx[4], nx[5], nx[0]);
            s_r[lane] = nx[2];
            // (the new state goes to HBM after barrier 1: wave 1 computes the tyre parameters from the OLD row of the same array,
            // and nothing orders its load before a store issued here — seen once in a while as a `params` row of the new state;
            // the scaled action waits there too, so that a caller may scale its action array in place)
        } else {
            float es, ec;
            sincos_det(deg2rad(o9[5]), es, ec);                                // DAM:211
            const float4 pts = make_float4(o9[3] + LWS * ec, o9[4] + LWS * es, o9[3] - LWS * ec, o9[4] - LWS * es);
            s_pts[lane] = pts;
            s_oldc[lane] = make_float2(o9[3], o9[4]);
            float pr[4];
            f_xu_params(st, steer, a_x, pr);                                   // E2E:279 (the parameters of the same f_xu call)
            reinterpret_cast<float4*>(A.params)[i] = make_float4(pr[0], pr[1], pr[2], pr[3]);
            s_miu[lane] = pr[3];
            road_terms<TASK>(pts.x, pts.y, road_t, road_r);                    // DAM:231-295
            road_terms<TASK>(pts.z, pts.w, road_t, road_r);
        }
    }
    {   // the traffic step (TRF:220-238's role): the model's own prediction step per candidate, staged for the rest
        const SinCosK SK = sincos_consts();
        auto stage = [&](int idx, const float4 v, unsigned mode) {
            const int e = fast_div(idx, A.m_magic), c = idx - e * m_cand;
            float sn, cs;     // (the record loop of the rollout kernel: one code path for every turn class, eb_device.h)
            if (RESET) {
                if (!s_col[e]) respawn_fresh(e, c, s_ego);                            // a row of the mask: eb_traffic_respawn, unconditional
                else s_cand[e * RS4 + c] = v;
            } else if (OBS) s_cand[e * RS4 + c] = v;
            else {
                bool kept;
                const f4u r = predict_record_tc(f4u{v.x, v.y, v.z, v.w}, turn_consts(sturn[c]), SK, sn, cs, kept);
                float4 o = make_float4(r.x, r.y, r.z, r.w);
                s_cand[e * RS4 + c] = o;
                if (FUSED_FLOW && A.flow_on) {
                    // eb_traffic_flow_step's per-slot part right here (16-env tiles: the flow source's shape): a vehicle far out and heading
                    // away leaves (its record stays where the prediction put it), the others accelerate towards their vType's speed — what
                    // the NEXT step sees goes to HBM, the LDS copy above stays this step's state.  The rule asks for the SIGN of
                    // x cos + y sin at the NEW heading; a far record is outside the junction box, so its heading rate was the literal
                    // zero (`kept`): the new heading is the old one up to the rounding of the radian round trip (2e-7 relative) and the
                    // sin / cos the prediction has just made answer the question whenever the sum is not within 1 % of zero — the exact
                    // expression (a sin / cos of its own: what the separate pass paid for every record) only for what is left.
                    bool on = (mode >> 8) != 0;
                    const float ax = __builtin_fabsf(o.x), ay = __builtin_fabsf(o.y);
                    const bool far = __builtin_fmaxf(ax, ay) > A.flow_exit_range;
                    const float sa = o.x * cs + o.y * sn;
                    const bool sure = kept && __builtin_fabsf(v.w) <= 1000.0f && __builtin_fabsf(sa) > 0.01f * (ax + ay);
                    bool outward = sa > 0.0f;
                    if (__builtin_amdgcn_ballot_w64(on && far && !sure)) {
                        float fs, fc;
                        sincos_det(deg2rad(o.w), fs, fc);
                        if (!sure) outward = o.x * fc + o.y * fs > 0.0f;
                    }
                    if (on) {
                        if (far && outward) on = false;
                        else {
                            const float vn = o.z + A.flow_accel * A.flow_dt, vm = s_vmax[c];
                            o.z = vn < vm ? vn : vm;
                        }
                    }
                    s_on[idx] = on ? 1 : 0;
                    reinterpret_cast<float4*>(A.cand)[(size_t)e0 * m_cand + idx] = o;
                    // the slot's flag and mode byte for the next step (an emission into the slot overwrites all three behind barrier 3)
                    A.flow_active[(size_t)e0 * m_cand + idx] = on ? 1 : 0;
                    A.flow_mode_out[(size_t)e0 * m_cand + idx] = on ? (uint8_t)fast_div(c, A.k_magic) : (uint8_t)EB_VMODE_EMPTY;
                    s_tag[e * TS4 * 4 + c] = (uint8_t)mode;
                    if (ELIG && (mode & 0xffu) < (unsigned)EB_VMODE_COUNT) atomicOr(&s_elig32[(e * EB_VMODE_COUNT + (int)(mode & 0xffu)) * 2 + (c >> 5)], 1u << (c & 31));
                    return;
                }
                // the record goes back to HBM right here, from the lane that loaded it, with the pool's re-entry rule on the way
                // (eb_traffic_respawn: it applies AFTER the observation and the done code saw this step's state — both read the LDS
                // copy above).  It used to leave from LDS in phase 2, 16 KB through one wave that had the tracking chain to do.
                if (A.respawn_entry && (__builtin_fabsf(o.x) > A.limit || __builtin_fabsf(o.y) > A.limit)) {
                    const uint64_t ub = (A.counter << 32) + (uint64_t)(e0 + e) * 128u + (uint64_t)c * 2u;
                    const float u1 = u01(A.seed, ub), u2 = u01(A.seed, ub + 1);
                    const float* en = A.respawn_entry + 5 * c;
                    const float along = u1 * A.span;
                    o = make_float4(en[0] + along * en[3], en[1] + along * en[4], u2 * A.v_max, en[2]);
                }
                if (A.flow_on) s_on[idx] = (uint8_t)(mode >> 8);            // (the flow rule stores the record in its own pass below)
                else reinterpret_cast<float4*>(A.cand)[(size_t)e0 * m_cand + idx] = o;
            }
            s_tag[e * TS4 * 4 + c] = (uint8_t)mode;
            if (ELIG && (mode & 0xffu) < (unsigned)EB_VMODE_COUNT) atomicOr(&s_elig32[(e * EB_VMODE_COUNT + (int)(mode & 0xffu)) * 2 + (c >> 5)], 1u << (c & 31));
        };
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = rec_index(g, k);
                if (idx >= 0 && idx < n_rec) stage(idx, cv[g][k], cm[g][k]);
                if (PAIR_STEP && g == 0 && k == 0) ES_MARK(7);                   // (16-env tiles: the slot marks 7 / 11 are free) first chunk staged
                if (PAIR_STEP && g == 0 && k == 2) ES_MARK(11);                  // first group staged
            }
        for (int g = 2; g * GREC < n_rec; ++g) {                                // more than 16 candidates per env: one group at a time
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = rec_index(g, k);
                if (idx >= 0 && idx < n_rec) {
                    cv[0][k] = csrc[idx]; cm[0][k] = msrc[idx];
                    if (!OBS && A.flow_on) cm[0][k] |= (unsigned)A.flow_active[(size_t)e0 * m_cand + idx] << 8;
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = rec_index(g, k);
                if (idx >= 0 && idx < n_rec) stage(idx, cv[0][k], cm[0][k]);
            }
        }
    }
    if (!OBS && !FUSED_FLOW && A.flow_on) {
        // eb_traffic_flow_step, per slot (one copy of the code, a pass of its own over this lane's records): a vehicle far out and
        // heading away leaves (its record stays where the prediction put it), the others accelerate towards their vType's speed —
        // what the NEXT step sees goes to HBM; the LDS copy stays this step's state
        for (int g = 0; g * GREC < n_rec; ++g)
#pragma unroll 1
            for (int k = 0; k < 3; ++k) {
                const int idx = rec_index(g, k);
                if (idx < 0 || idx >= n_rec) continue;
                const int e = fast_div(idx, A.m_magic), c = idx - e * m_cand;
                float4 o = s_cand[e * RS4 + c];
                bool on = s_on[idx] != 0;
                if (on) {
                    float fs, fc;
                    sincos_det(deg2rad(o.w), fs, fc);
                    const bool outward = o.x * fc + o.y * fs > 0.0f;
                    if (__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)) > A.flow_exit_range && outward) on = false;
                    else {
                        const float vn = o.z + A.flow_accel * A.flow_dt, vm = A.flow_v_max[c];
                        o.z = vn < vm ? vn : vm;
                    }
                }
                s_on[idx] = on ? 1 : 0;
                reinterpret_cast<float4*>(A.cand)[(size_t)e0 * m_cand + idx] = o;
            }
    }
    float delta_y = 0.0f;
    // E2E:329-338 ego vector, E2E:293-297 tracking error on path p: the head of this lane's observation row (from nx) -> s_out.
    // Called by EVERY lane of a wave (`on`: this lane has a row to make): the few lanes whose position no grid level answers with a
    // short range — beyond every level, or abreast of a long straight far out — are resolved by the whole wave together, one after the
    // other: each lane takes eight consecutive table entries (one round trip for the whole table), then a wave-wide first-minimum.
    // A lane doing that search alone (the pruned search: ~8 dependent round trips at ~1 us each in a loaded step kernel) held its 63
    // neighbours for as long; and a launch that is one generation of blocks lasts as long as its slowest tile.
    auto track_row = [&](const bool on, int p) {
        float* orow = s_out + lane * OS;
        const float ex = nx[3], ey = nx[4];
        const PathTables& pt = A.pt;
        int bi = 0;
        // the table point itself comes out of the scan — stride-10 entry bi IS path point 10 * bi (x, y, heading): no
        // third dependent round trip to the full-resolution tables (the rollout kernel's closest_cell_index does the same)
        float rx = 0.0f, ry = 0.0f, rphi = 0.0f;
        bool whole = false;                                                   // this lane needs the search over the table
        if (on) {
#pragma unroll
            for (int c = 0; c < 6; ++c) orow[c] = nx[c];
        }
        if (on && p >= 0) {
            const float2* red = pt.red[p];
            const float* ph10 = pt.phi10[p];
            const float fx = (ex - pt.gx0) * CELL_INV, fy = (ey - pt.gy0) * CELL_INV;
            unsigned cw = 0xffffffffu;                                          // (also what a corridor cell on the path's medial axis holds: eb_capi.hip)
            if (fx >= 0.0f && fx < (float)pt.gnx && fy >= 0.0f && fy < (float)pt.gny) cw = pt.cells[(p * pt.gny + (int)fy) * pt.gnx + (int)fx];
            if (cw != 0xffffffffu) {
                // same order, same strict '<' as the full scan: same index (eb_device.h; a group of entries per loop trip: prefetched groups cost this kernel two VGPRs, i.e. a wave of occupancy, and bought nothing in the rollout kernel's A/B)
                bi = closest_in_range<0>(reinterpret_cast<const float*>(red), ph10, (int)(cw & 0xffffu), (int)(cw >> 16), ex, ey, rx, ry, rphi);
            } else {
                int lo, hi, lo2, hi2;   // off the corridor's grid: the coarse levels (an ego that has left the road, or finished and drives on)
                if (coarse_cell_ranges(pt, p, ex, ey, lo, hi, lo2, hi2) == 1)
                    bi = closest_in_ranges(reinterpret_cast<const float*>(red), ph10, lo, hi, lo2, hi2, ex, ey, rx, ry, rphi);
                else whole = true;
            }
        }
        unsigned long long pend = __builtin_amdgcn_ballot_w64(whole);
        if (pend != 0ull) {
            if (__popcll(pend) <= 8) {
                while (pend) {
                    const int src = __builtin_ctzll(pend);
                    pend &= pend - 1ull;
                    const float qx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ex), src));
                    const float qy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ey), src));
                    const int qp = __builtin_amdgcn_readlane(p, src);
                    const float* xy = reinterpret_cast<const float*>(pt.red[qp]);
                    const int n = pt.red_len[qp];                             // <= 512 = 64 lanes x 8 entries
                    typedef float f4x __attribute__((ext_vector_type(4), aligned(4)));
                    f4x q[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f4x*>(xy + 2 * min(8 * lane + 2 * u, n - 1));   // (readable 4 entries past the end)
                    float best = __builtin_inff();
                    int cb = 1 << 30;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int r = 8 * lane + 2 * u;
                        const float d0 = sq(qx - q[u].x) + sq(qy - q[u].y), d1 = sq(qx - q[u].z) + sq(qy - q[u].w);   // DAM:712
                        if (r < n && d0 < best) { best = d0; cb = r; }                                          // first minimum, DAM:714
                        if (r + 1 < n && d1 < best) { best = d1; cb = r + 1; }
                    }
#pragma unroll
                    for (int m = 1; m < 64; m <<= 1) {                        // (distance, index) minimum over the wave: the FIRST minimum
                        const float ob = __shfl_xor(best, m, 64);
                        const int oi = __shfl_xor(cb, m, 64);
                        if (ob < best || (ob == best && oi < cb)) { best = ob; cb = oi; }
                    }
                    if (lane == src) bi = cb == (1 << 30) ? 0 : cb;           // (nothing compared below +inf — NaN / inf coordinates: index 0, as the full scan)
                }
            } else if (whole) {
                bi = closest_reduced_index<8>(pt.red[p], pt.rad + 32 * p, pt.red_len[p], ex, ey);   // (a wave of far egos: every lane for itself)
            }
            if (whole) { rx = pt.red[p][bi].x; ry = pt.red[p][bi].y; rphi = pt.phi10[p][bi]; }
        }
        if (on) {
            if (p < 0) { for (int c = 0; c < T; ++c) orow[6 + c] = 0.0f; }
            else {
                const int idx = bi * 10, len = pt.len[p];
                delta_y = two2one<TASK>(ex, ey, rx, ry);
                orow[6] = delta_y;
                orow[7] = deal_with_phi_diff(nx[5] - rphi);
                orow[8] = nx[0] - EXP_V;
                int cur = idx;
                for (int k = 0; k < n_future; ++k) {
                    cur += 80;
                    if (cur >= len - 2) cur = len - 2;
                    const int fi = clamp_index(cur, len);
                    orow[9 + 3 * k] = pt.x[p][fi] - ex;
                    orow[10 + 3 * k] = pt.y[p][fi] - ey;
                    orow[11 + 3 * k] = deal_with_phi_diff(nx[5] - pt.phi[p][fi]);
                }
            }
        }
    };
    // (the step proper, wave 0) the tracking of the new pose right here, in front of barrier 1: it needs the ego step's result and the
    // path tables, nothing of the other waves — and wave 0, which stages the fewest records, used to stand at that barrier for ~1.5 us
    // and then run these two or three dependent table reads while waves 1-3 were at their pairs
    constexpr bool TRACK_EARLY = false;   // (measured: the tracking in front of barrier 1 moved that barrier by the tracking's own 1-2 us — the loads of a wave return in order, its table reads queue behind its records — and phase 2 got no shorter; r5l)
    if (TRACK_EARLY && wave == 0) track_row(live, path_pre);
    // eb_traffic_flow_step, per (env, route): the route's timer, and — when it is due and a slot of the route is vacant — the vehicle
    // that enters: into LDS; stored behind the observation (phase 4).  Wave 1's job, behind its reward pairs
    // (16-env tiles: 192 (env, route) pairs = three rounds of 64 — rounds 0 and 1 on wave 1, whose reward pairs are few there, round
    // 2 on wave 3 behind its share of the collision pass; a round is two or three dependent round trips, ~0.8 us, and one wave doing
    // all three was the tile's longest chain between barriers 1 and 3)
    float ft[3] = {0.0f, 0.0f, 0.0f};
    const int em_first = !FUSED_FLOW ? 0 : wave == 1 ? 0 : 2, em_last = !FUSED_FLOW ? (1 << 30) : wave == 1 ? 1 : wave == 3 ? 2 : -1;
    if (FUSED_FLOW && A.flow_on) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k >= em_first && k <= em_last && lane + 64 * k < nE * 12) ft[k] = A.flow_timer[(size_t)e0 * 12 + lane + 64 * k];
    }
    // (16-env tiles: phase 2 only DECIDES — the route's timer and its first vacant slot; the slot index stays in a register of the lane,
    // emr[round], and the vehicle itself — two draws, the lane table — is made in phase 4 by the same lane, right where it is stored:
    // no LDS for it (3.8 KB a tile: the sixth tile of a CU), and the draws leave the chain between barriers 1 and 3)
    int emr[2] = {-1, -1};
    auto flow_vehicle = [&](int q, int vacant) -> float4 {
        const int K = A.flow_K, e = q / 12, r = q - e * 12, j = r * K + vacant;
        const uint64_t base = (A.counter << 32) + (uint64_t)(e0 + e) * 128u + (uint64_t)(r * K) * 2u;
        const float u1 = u01(A.seed, base), u2 = u01(A.seed, base + 1);
        const float* ln = A.flow_lane + 5 * j;
        const float along = u1 * A.flow_lane_len;
        return make_float4(ln[0] + along * ln[3], ln[1] + along * ln[4], u2 * A.flow_v_max[j], ln[2]);
    };
    auto flow_emission = [&]() {
        const int K = A.flow_K;
        for (int kq = em_first, q = lane + 64 * em_first; q < nE * 12 && kq <= em_last; q += 64, ++kq) {
            const int e = q / 12, r = q - e * 12, ge = e0 + e;
            const uint8_t* on = s_on + e * m_cand + r * K;
            int vacant = -1;
            for (int k = K - 1; k >= 0; --k)
                if (!on[k]) vacant = k;
            const size_t ti = (size_t)ge * 12 + r;
            float t = ((FUSED_FLOW && kq < 3) ? (kq == 0 ? ft[0] : kq == 1 ? ft[1] : ft[2]) : A.flow_timer[ti]) + A.flow_dt;
            const float per = A.flow_period[r];
            int em = -1;
            if (t >= per && vacant >= 0) {
                if (!FUSED_FLOW) s_new[q] = flow_vehicle(q, vacant);
                em = vacant;
                t = t - per;
                atomicAdd(&A.flow_emitted[ti], 1);   // (no value returned: nothing waits for it)
            }
            if (FUSED_FLOW) { if (kq == em_first) emr[0] = em; else emr[1] = em; }
            else s_emit[q] = em;
            A.flow_timer[ti] = t;
        }
    };
    ES_MARK(1);
    __syncthreads();   // barrier: s_ego, s_pts, s_oldc, s_cand, mode bytes
    if (by_phase) __builtin_amdgcn_s_setprio(1);
    ES_MARK(8);

    // ---- phase 2 ---------------------------------------------------------------------------------------------
    if (wave == 0) {
        if (!TRACK_EARLY) track_row(live, RESET ? reset_path : path_pre);
        ES_MARK(9);
        if (!OBS && live) {   // E2E:135: the ego state in place — only now, when every wave has read the old one (barrier 1), and
            // behind the tracking's dependent table reads rather than in front of them
            float2* ego_out = reinterpret_cast<float2*>(A.ego + 6 * (size_t)i);
            ego_out[0] = make_float2(nx[0], nx[1]); ego_out[1] = make_float2(nx[2], nx[3]); ego_out[2] = make_float2(nx[4], nx[5]);
            if (A.scaled) reinterpret_cast<float2*>(A.scaled)[i] = make_float2(steer, a_x);
        }
    } else if (!OBS) {
        unsigned short* myq = s_queue + wave * ES_QCAP;
        if (wave == 1) {   // compute_rewards' vehicle loop on the CURRENT observation (DAM:218-229), one lane per (env, slot).  A circle pair
            // can only be closer than 3.5 m when the two centres are within 3.5 + 2 * 1.4 = 6.3 m: pairs inside 6.364 m
            // (slack >> fp32 rounding; the rollout kernel's test) are queued, every other pair contributes exact zeros.
            auto body = [&](int item) {
                const int e = item >> 6, j = item & 63;
                const f4a4 v = *reinterpret_cast<const f4a4*>(A.obs + (size_t)D * (e0 + e) + 6 + T + 4 * j);
                float vs, vc, t35[4], t25[4];
                sincos_det(deg2rad(v.w), vs, vc);
                veh2veh_terms(s_pts[e], v.x, v.y, vs, vc, t35, t25);
                s_part[e * NV + j] = make_float2(((t35[0] + t35[1]) + t35[2]) + t35[3], ((t25[0] + t25[1]) + t25[2]) + t25[3]);
            };
            WaveQueue<decltype(body)> Q{myq, 0, body};
            for (int base = 0; base < n_pairs; base += 64 * 4) {               // four pairs per lane per batch
                if (base > 0) load_pairs(base);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int p = base + pt_ + 64 * k;
                    const bool valid = p < n_pairs;
                    const int e = valid ? fast_div(p, A.nv_magic) : 0, j = valid ? p - e * NV : 0;
                    const float2 c = s_oldc[e];
                    const float dx = pxy[k].x - c.x, dy = pxy[k].y - c.y;
                    const bool near = valid && dx * dx + dy * dy < 40.5f;
                    if (valid && !near) s_part[e * NV + j] = make_float2(0.0f, 0.0f);
                    Q.push(near, e * 64 + j);
                }
            }
            Q.flush();
            if (A.flow_on) flow_emission();
        }
        ES_MARK(5);
        if (wave == 2 || wave == 3) {   // one lane per (env, candidate): the collision test (TRF:263-295), its 10 m box first
            auto body = [&](int item) {
                const int e = item >> 6, c = item & 63;
                const float4 eg = s_ego[e];
                const EgoCircles E = ego_circles(eg.x, eg.y, eg.z);
                const size_t ck = (size_t)(e0 + e) * m_cand + c;
                if (collision_with(E, eg.x, eg.y, s_cand[e * RS4 + c], A.cand_lw ? A.cand_lw[ck * 2] : 4.8f,
                                   A.cand_lw ? A.cand_lw[ck * 2 + 1] : 2.0f))
                    s_col[e] = 1;
            };
            WaveQueue<decltype(body)> Q{myq, 0, body};
            for (int base = 0; base < n_rec; base += 128 * 4) {                // four records per lane in flight
                float2 v3[4], eg3[4];
                int m3[4], e3[4], c3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = base + ct_ + 128 * k;
                    e3[k] = 0; c3[k] = 0;
                    if (idx < n_rec) { e3[k] = fast_div(idx, A.m_magic); c3[k] = idx - e3[k] * m_cand; }
                    v3[k] = *reinterpret_cast<const float2*>(s_cand + e3[k] * RS4 + c3[k]);
                    eg3[k] = *reinterpret_cast<const float2*>(s_ego + e3[k]);
                    m3[k] = s_tag[e3[k] * TS4 * 4 + c3[k]];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool box = base + ct_ + 128 * k < n_rec && m3[k] != EB_VMODE_EMPTY &&
                                     __builtin_fabsf(v3[k].x - eg3[k].x) < 10.0f && __builtin_fabsf(v3[k].y - eg3[k].y) < 10.0f;
                    Q.push(box, e3[k] * 64 + c3[k]);
                }
            }
            Q.flush();
            if (FUSED_FLOW && A.flow_on && wave == 3) flow_emission();
        }
    }
    ES_MARK(2);
    // (no barrier here any more: the slots below need the ego, the candidates and the mode bytes — final since barrier 1 — so a wave
    // walks its modes as soon as its own pair / collision / tracking work is done; what DOES depend on the other waves' phase-2
    // results — the penalty sums, the done code — runs behind the one barrier that also completes the rows)
    // (the penalty sums are wave 1's own business since every reward pair of the tile is: no barrier between its pair pass and them)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!OBS && wave == 1 && live) {
        // E2E:134: the reward of the step taken from the CURRENT observation; penalty partials in vehicle order
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
    }
    if (!OBS && wave == 3 && lane < ET) {   // the done predicates that need only the new ego state (E2E:223-256): a byte for the merge
        const float4 eg = s_ego[lane];      // behind barrier 3 (wave 3, after its share of the collision pass: wave 0 has the tracking chain)
        s_jb[lane] = live ? (uint8_t)(judge_bits(TASK, eg.w, s_r[lane], eg.x, eg.y, eg.z, s_miu[lane], red_light) |
                                      (A.episode_step && ep_cnt >= A.max_episode_steps ? JB_TIMEOUT : 0u)) : (uint8_t)0xff;
    }
    ES_MARK(6);
    // E2E:340-464 for the lanes with `on` (lane = env): the vehicle slots of this wave's modes -> s_out
    auto fill_slots = [&](const bool on, const bool light_on, const bool dyn) {
        // E2E:340-464.  The slot plan is scalar: lane s of `slot_mode` holds the mode of slot s, so the distinct modes (first
        // occurrences: A.first_mask), who walks which (dyn: the next mode off s_modeq; else fixed owner waves) and a mode's slots (a ballot) cost no memory
        // access; per mode (wave-uniform) and env (lane) the tag row becomes a candidate set and the slots are filled by
        // repeated selection over it with branch-free key comparisons.
        const float4 eg = s_ego[lane];
        const float ex = eg.x, ey = eg.y;
        const float4* crow = s_cand + lane * RS4;
        const unsigned* trow = s_tag32 + lane * TS4;
        const bool virt = TASK != TASK_RIGHT && light_on && ey < -HALF_CROSS;                                // E2E:386-388
        float* ov = s_out + lane * OS + 6 + T;
        const int nw = (m_cand + 3) >> 2;
        // one distinct mode m (wave-uniform) for the lanes that call it: its slots (ascending, as a bit set) <- the env's candidates of the mode
        auto one_mode = [&](const int m, unsigned long long slots) {
            // the env's candidates of this mode as a bit set (mode byte == m), 4 bytes per dword; the range filter of
            // E2E:393-411 is applied in the walk below, where the mode is wave-uniform
            unsigned long long elig = 0ull;
            if (ELIG) {
                const uint2 w2 = *reinterpret_cast<const uint2*>(&s_elig32[(lane * EB_VMODE_COUNT + (m < EB_VMODE_COUNT ? m : 0)) * 2]);
                elig = m < EB_VMODE_COUNT ? (unsigned long long)w2.x | (unsigned long long)w2.y << 32 : 0ull;
            } else {
                const unsigned mm = (unsigned)m * 0x01010101u;
                for (int w = 0; w < nw; ++w) {
                    const unsigned x = trow[w] ^ mm;
                    const unsigned z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);   // 0x80 in every zero byte of x
                    const unsigned nib = (((z >> 7) * 0x01020408u) >> 24) & 0xfu;
                    elig |= (unsigned long long)nib << (4 * w);
                }
            }
            ES_MARK(7);
            if (__popcll(slots) <= 2) {
                // the usual case (the native lists have at most two slots per mode, VEHICLE_MODE_DICT UTL:21-23)
                const int sa = __builtin_ctzll(slots);
                slots &= slots - 1ull;
                const int sb = slots ? __builtin_ctzll(slots) : -1;
                switch (m) {   // (wave-uniform: one scalar jump per mode, not per candidate)
#define EB_SLOT_CASE(M) case M: slot_pair_walk<TASK, M>(crow, elig, ex, ey, virt, m_cand, ov, sa, sb); break
                    EB_SLOT_CASE(EB_VMODE_DL); EB_SLOT_CASE(EB_VMODE_DU); EB_SLOT_CASE(EB_VMODE_DR); EB_SLOT_CASE(EB_VMODE_RU);
                    EB_SLOT_CASE(EB_VMODE_UR); EB_SLOT_CASE(EB_VMODE_UD); EB_SLOT_CASE(EB_VMODE_UL); EB_SLOT_CASE(EB_VMODE_LR);
#undef EB_SLOT_CASE
                    default: slot_pair_walk<TASK, EB_VMODE_RD>(crow, elig, ex, ey, virt, m_cand, ov, sa, sb); break;   // rd rl lu ld: no filter, no key, zero fill
                }
                ES_MARK(12);
                return;
            }
            const KeySpec ks = key_spec(TASK, m);
            // the virtual red-light car of the mode (E2E:386-390), candidate index m_cand
            const V4 vv = {m == EB_VMODE_DL ? LANE_W / 2 : LANE_W * 1.5f, -HALF_CROSS + 2.5f, 0.0f, 90.0f};
            const bool has_virt = virt && (m == EB_VMODE_DL || m == EB_VMODE_DU) && veh_in_range(TASK, m, vv, ex, ey);
            const float2 vk = key_of(ks, vv.x, vv.y);
            const V4 fill = veh_fill_value(m);
            const float4 fill4 = make_float4(fill.x, fill.y, fill.v, fill.phi);   // slice_or_fill, E2E:431-437
            const float4 vv4 = make_float4(vv.x, vv.y, vv.v, vv.phi);
            {   // more than two slots of one mode: the range filter first, then one selection pass per slot
                unsigned long long in = 0ull, rest = elig;
                while (rest) {
                    const int c = __builtin_ctzll(rest);
                    rest &= rest - 1ull;
                    const float4 q = crow[c];
                    if (veh_in_range(TASK, m, V4{q.x, q.y, q.z, q.w}, ex, ey)) in |= 1ull << c;
                }
                elig = in;
            }
            float2 prev_k = make_float2(0.0f, 0.0f);
            int prev_i = -1;
            bool found = true;
            while (slots) {
                const int s2 = __builtin_ctzll(slots);
                slots &= slots - 1ull;
                float4 r = fill4;
                if (found) {
                    float2 best_k = make_float2(0.0f, 0.0f);
                    int best_i = -1;
                    unsigned long long rest = elig;
                    while (rest) {
                        const int c = __builtin_ctzll(rest);
                        rest &= rest - 1ull;
                        const float2 xy = *reinterpret_cast<const float2*>(crow + c);
                        const float2 kk = key_of(ks, xy.x, xy.y);
                        const bool after = prev_i < 0 || key_before(prev_k, prev_i, kk, c);      // not yet picked
                        const bool better = best_i < 0 || key_less(kk, best_k);
                        if (after && better) { best_k = kk; best_i = c; }
                    }
                    if (has_virt) {
                        const bool after = prev_i < 0 || key_before(prev_k, prev_i, vk, m_cand);
                        const bool better = best_i < 0 || key_less(vk, best_k);
                        if (after && better) { best_k = vk; best_i = m_cand; }
                    }
                    if (best_i < 0) found = false;
                    else {
                        prev_k = best_k; prev_i = best_i;
                        r = best_i < m_cand ? crow[best_i] : vv4;
                    }
                }
                *reinterpret_cast<f4a4*>(ov + 4 * s2) = f4a4{r.x, r.y, r.z, r.w};
            }
        };
        unsigned long long firsts = A.first_mask;
        const int n_first = __popcll(A.first_mask);
        for (int k = 0;; ++k) {
            int s;
            if (dyn) {
                // the step proper: a wave takes the next distinct mode off a counter in LDS when its own pair / collision / tracking
                // work is done — the waves reach this point up to 2 us apart (the reward pairs are one wave's), and a mode is a mode
                int g = 0;
                if (lane == 0) g = atomicAdd(&s_modeq, 1);
                g = __builtin_amdgcn_readfirstlane(g);
                if (g >= n_first) break;
                unsigned long long f = A.first_mask;
                for (int j = 0; j < g; ++j) f &= f - 1ull;
                s = __builtin_ctzll(f);
            } else {
                if (!firsts) break;
                s = __builtin_ctzll(firsts);
                firsts &= firsts - 1ull;
                if ((NW == 4 ? ((0x1e >> (2 * (k & 3))) & 3) : 4 + (k & 3)) != wave) continue;   // owners in turn: waves 2, 3, 1, 0 (NW = 8: 4, 5, 6, 7)
            }
            const int m = __builtin_amdgcn_readlane(slot_mode, s);
            unsigned long long slots = __builtin_amdgcn_ballot_w64(slot_mode == m);   // the mode's slots, ascending
            if (on) one_mode(m, slots);      // (no lane leaves the loop early: the counter is read by the whole wave)
        }
    };
    // One (env, distinct mode) pair per LANE — the mode per lane, its range box, key and fill value by data: slot_pair_walk's selection
    // without a wave-uniform mode.  The auto-reset tail's slot pass (a handful of finished envs per tile), and the step's own on 16-env
    // tiles, where a lane per env would leave three quarters of every wave idle in each of the ~8 per-mode passes.
    auto pair_walk = [&](const bool act, const int e, const int j, const float ex, const float ey, const bool lit) {
        const unsigned dm = s_dm[j];
        const int m = (int)(dm & 0xffu), sa = (int)((dm >> 8) & 0xffu), sb = (int)((dm >> 16) & 0xffu);
        const bool virt = TASK != TASK_RIGHT && lit && ey < -HALF_CROSS;                        // E2E:386-388
        const float4* crow = s_cand + e * RS4;
        const unsigned* trow = s_tag32 + e * TS4;
        float* ov = s_out + e * OS + 6 + T;
        const int nw = (m_cand + 3) >> 2;
        unsigned long long elig = 0ull;
        if (ELIG) {
            const uint2 w2 = *reinterpret_cast<const uint2*>(&s_elig32[(e * EB_VMODE_COUNT + (m < EB_VMODE_COUNT ? m : 0)) * 2]);
            elig = m < EB_VMODE_COUNT ? (unsigned long long)w2.x | (unsigned long long)w2.y << 32 : 0ull;
        } else {
            const unsigned mm = (unsigned)m * 0x01010101u;
            for (int w = 0; w < nw; ++w) {
                const unsigned x = trow[w] ^ mm;
                const unsigned z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);   // 0x80 in every zero byte of x
                elig |= (unsigned long long)((((z >> 7) * 0x01020408u) >> 24) & 0xfu) << (4 * w);
            }
        }
        if (!act) elig = 0ull;
        const RangeBox rb = range_box(TASK, m, ex, ey);
        const KeySpec ks = key_spec(TASK, m);
        const V4 fill = veh_fill_value(m);
        float2 k1 = make_float2(0.0f, 0.0f), k2 = k1;
        int i1 = -1, i2 = -1;
        auto offer = [&](const bool valid, const float2 kk, const int c) {        // slot_pair_walk's, the mode per lane
            const bool first = valid & ((i1 < 0) | key_less(kk, k1));
            const bool second = valid & !first & ((i2 < 0) | key_less(kk, k2));
            k2.x = first ? k1.x : (second ? kk.x : k2.x); k2.y = first ? k1.y : (second ? kk.y : k2.y);
            i2 = first ? i1 : (second ? c : i2);
            k1.x = first ? kk.x : k1.x; k1.y = first ? kk.y : k1.y;
            i1 = first ? c : i1;
        };
        const float2* cxy = reinterpret_cast<const float2*>(crow);
        {   // the first four members of the set without a loop, their (x, y) reads in flight together (slot_pair_walk's opening: a mode rarely
            // has more — the flow source's routes have at most K = 5 slots —, and a loop trip is a dependent LDS round trip)
            constexpr int UNR = 4;
            int cs[UNR];
            bool hs[UNR];
            float2 q[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                hs[u] = elig != 0ull;
                cs[u] = hs[u] ? __builtin_ctzll(elig) : 0;
                elig &= elig - 1ull;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) q[u] = cxy[2 * cs[u]];
#pragma unroll
            for (int u = 0; u < UNR; ++u) offer(hs[u] & box_in_range(rb, q[u].x, q[u].y), key_of(ks, q[u].x, q[u].y), cs[u]);
        }
        bool has = elig != 0ull;
        int c = has ? __builtin_ctzll(elig) : 0;
        elig &= elig - 1ull;
        while (__builtin_amdgcn_ballot_w64(has)) {
            const float2 xy = cxy[2 * c];
            const bool hq = has;
            const int cq = c;
            has = elig != 0ull;
            c = has ? __builtin_ctzll(elig) : 0;
            elig &= elig - 1ull;
            offer(hq & box_in_range(rb, xy.x, xy.y), key_of(ks, xy.x, xy.y), cq);
        }
        const float4 vv4 = make_float4(m == EB_VMODE_DL ? LANE_W / 2 : LANE_W * 1.5f, -HALF_CROSS + 2.5f, 0.0f, 90.0f);
        offer(act & virt & ((m == EB_VMODE_DL) | (m == EB_VMODE_DU)) & box_in_range(rb, vv4.x, vv4.y), key_of(ks, vv4.x, vv4.y), m_cand);
        const float4 fill4 = make_float4(fill.x, fill.y, fill.v, fill.phi);   // slice_or_fill, E2E:431-437
        const float4 q1 = crow[i1 < 0 || i1 >= m_cand ? 0 : i1], q2 = crow[i2 < 0 || i2 >= m_cand ? 0 : i2];
        const float4 r1 = i1 < 0 ? fill4 : (i1 >= m_cand ? vv4 : q1), r2 = i2 < 0 ? fill4 : (i2 >= m_cand ? vv4 : q2);
        if (act) {
            *reinterpret_cast<f4a4*>(ov + 4 * sa) = f4a4{r1.x, r1.y, r1.z, r1.w};
            if (sb != 0xff) *reinterpret_cast<f4a4*>(ov + 4 * sb) = f4a4{r2.x, r2.y, r2.z, r2.w};
        }
    };
    if (PAIR_STEP && A.dm_ok) {
        // the step's slots on a 16-env tile: 64 (env, mode) pairs at a time, off the counter like the modes of the larger tiles
        const unsigned long long lmask = __builtin_amdgcn_ballot_w64(light);     // lane = env: the same in every wave
        const int n_pairs_s = nE * A.n_dm;
        for (;;) {
            int g = 0;
            if (lane == 0) g = atomicAdd(&s_modeq, 1);
            g = __builtin_amdgcn_readfirstlane(g);
            if (g * 64 >= n_pairs_s) break;
            const int q = g * 64 + lane;
            const bool act = q < n_pairs_s;
            const int e = act ? fast_div(q, A.dm_magic) : 0, j = act ? q - e * A.n_dm : 0;
            const float4 eg = s_ego[e];
            pair_walk(act, e, j, eg.x, eg.y, (lmask >> e) & 1ull);
        }
    } else
    fill_slots(live, light, true);
    ES_MARK(3);
    // (16-env tiles with the flow rule: an entering vehicle is stored behind this barrier by ANOTHER lane than the one that stored the
    // slot's record, flag and mode byte in phase 1 — those stores, some 4 us old, are complete before anybody passes the barrier)
    if (FUSED_FLOW && A.flow_on) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // barrier: s_out complete; s_part, s_col, s_jb
    if (by_phase) __builtin_amdgcn_s_setprio(0);
    ES_MARK(10);
    // E2E:200-221, the priority chain: every wave merges the tile's done codes for itself (lane = env: a byte, a flag and delta_y
    // from LDS, a dozen instructions) — the finished rows as a wave-uniform bit mask, no further barrier before the rows leave
    unsigned long long finmask = 0ull;
    if (!OBS) {
        uint8_t code = EB_DONE_NOT_YET;
        if (lane < ET && s_jb[lane] != 0xff) code = judge_merge(s_jb[lane], s_col[lane] != 0, s_out[lane * OS + 6]);
        if (wave == 0 && live) A.done_code[i] = code;
        if (wave == 3 && live && A.episode_step) A.episode_step[i] = code != EB_DONE_NOT_YET ? 0 : ep_cnt;   // a finished env's next step is step 1 of its next episode
        finmask = __builtin_amdgcn_ballot_w64(code != EB_DONE_NOT_YET);
    }

    // ---- phase 4 ---------------------------------------------------------------------------------------------
    if (RESET && wave == 0 && live) A.virtual_out[i] = virtual_next ? 1 : 0;   // every wave read the old flag before the barriers above
    if (RESET && wave == 0 && !live && lane < nE && A.done_src && A.done_code) A.done_code[i] = A.done_src[i];
    // observation rows out: the tile's rows are contiguous in memory (four LDS reads in flight per lane) — every row (OBS with a row
    // mask: the masked rows; RESET: the others carried over).  AUTO: the finished envs' rows go out here too, as everybody's; their
    // copy for final_obs and, later, the reset rows that replace them are a wave per row (below)
    auto store_rows = [&](const int) {
        float* dst = A.obs_out + (size_t)e0 * D;
        const int total = nE * D;
        for (int base = tid; base < total; base += 4 * NT) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + NT * k < total ? base + NT * k : 0;
                const int e = fast_div(idx, A.d_magic), c = idx - e * D;
                v[k] = s_out[e * OS + c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = base + NT * k;
                if (idx >= total) continue;
                if (!(OBS && A.row_mask && s_col[fast_div(idx, A.d_magic)])) dst[idx] = v[k];
                else if (RESET && A.obs) dst[idx] = A.obs[(size_t)e0 * D + idx];            // a row outside the mask: carried over
            }
        }
    };
    // AUTO, a tile with a finished env: the tail below lets OTHER threads overwrite what this thread stored during the step (candidate
    // records, params) — those stores left in phases 1-3 and are waited for here, where they have long been acknowledged, instead of
    // behind the row store, whose acknowledgements would be waited for with them
    if (AUTO && finmask != 0ull) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_rows(0);
    if (!OBS && A.flow_on) {
        // eb_traffic_flow_step, the rest: a slot's flag and mode byte for the next step, the entering vehicle into its slot — by the
        // lane that staged the slot (it stored the slot's record in phase 1 and read its mode byte: program order settles both) —
        // and the env's clock and light (every wave has used the old light: barrier 3)
        const int K = A.flow_K;
        if (FUSED_FLOW) {
            // 16-env tiles: the staging lanes have stored every slot's record, flag and mode byte (phase 1; complete: the wait in front
            // of barrier 3) — what is left is the entering vehicles, made and stored by the lanes that decided them (emr)
            if (wave == 1 || wave == 3)
                for (int kq = em_first, q = lane + 64 * em_first; q < nE * 12 && kq <= em_last; q += 64, ++kq) {
                    const int em = kq == em_first ? emr[0] : emr[1];
                    if (em < 0) continue;
                    const int e = q / 12, r = q - e * 12;
                    const size_t sidx = (size_t)(e0 + e) * m_cand + r * K + em;
                    reinterpret_cast<float4*>(A.cand)[sidx] = flow_vehicle(q, em);
                    A.flow_active[sidx] = 1;
                    A.flow_mode_out[sidx] = (uint8_t)r;
                }
        } else
        for (int g = 0; g * GREC < n_rec; ++g)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int idx = rec_index(g, k);
                if (idx < 0 || idx >= n_rec) continue;
                const int e = fast_div(idx, A.m_magic), c = idx - e * m_cand;
                const int r = fast_div(c, A.k_magic), kk = c - r * K, q = e * 12 + r;
                bool on = s_on[idx] != 0;
                if (s_emit[q] == kk) {
                    reinterpret_cast<float4*>(A.cand)[(size_t)e0 * m_cand + idx] = s_new[q];
                    on = true;
                }
                A.flow_active[(size_t)e0 * m_cand + idx] = on ? 1 : 0;
                A.flow_mode_out[(size_t)e0 * m_cand + idx] = on ? (uint8_t)r : (uint8_t)EB_VMODE_EMPTY;
            }
        if (wave == 0 && live) {
            const int n = A.flow_sim_step[i] + 1;
            A.flow_sim_step[i] = n;
            if (A.flow_light_cycle) {   // a.net.xml:145-150: 25 s phase 0, 5 s phase 1, 25 s phase 2, 5 s phase 3, in steps of dt
                const float tt = (float)(n % (int)(60.0f / A.flow_dt + 0.5f)) * A.flow_dt;
                A.v_light_out[i] = tt < 25.0f ? 0 : (tt < 30.0f ? 1 : (tt < 55.0f ? 2 : 3));
            }
        }
    }
    if (AUTO && A.flow_on && finmask != 0ull) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the flow rule's stores, before the tail rewrites the finished envs' slots
    ES_MARK(4);
    if (AUTO) {
        // ---- the envs this step finished start their next episode (E2E:99-127) — eb_env_reset_pool's arithmetic on those rows ----
        const bool fin = (finmask >> lane) & 1ull;                               // the same in every wave
        if (finmask == 0ull) return;                                             // nobody in this tile: the usual case per row, not per tile
        const int n_fin = __popcll(finmask);
        auto nth_fin = [&](int k) -> int {                                       // the k-th finished env of the tile (k is small)
            unsigned long long mbits = finmask;
            for (int t = 0; t < k; ++t) mbits &= mbits - 1ull;
            return __builtin_ctzll(mbits);
        };
        if (DRAW_LATE && wave == 0 && fin) {
            float rs[6];
            int rpath;
            bool rvn;
            draw_values(rs, rpath, rvn);
            s_rst[lane] = make_float4(rs[3], rs[4], rs[5], rs[0]);
            s_rflag[lane] = (unsigned)rpath | (rvn ? 4u : 0u);
        }
        // (the flow source's reset below ORs a bit per re-entered slot into the finished envs' route sets — spent since barrier 3: cleared here)
        if (ELIG && A.flow_on)
            for (int q = tid; q < n_fin * 12; q += NT) {
                const int k = q / 12, w2 = (nth_fin(k) * EB_VMODE_COUNT + (q - 12 * k)) * 2;
                s_elig32[w2] = 0u; s_elig32[w2 + 1] = 0u;
            }
        // everybody's reads of s_ego / s_cand are over, and (the wait in front of the row store) everybody's candidate / params stores of
        // the step are complete before ANOTHER thread overwrites them below
        __syncthreads();
        ES_MARK(11);
        if (A.final_obs && wave >= 2)                                            // the terminal observations: a wave per finished env, next to
            for (int k = wave - 2; k < n_fin; k += NW - 2) {                     // the new state (waves 0, 1) — s_out is not written before the
                const int e = nth_fin(k);                                        // next barrier
                for (int c = lane; c < D; c += 64) A.final_obs[((size_t)e0 + e) * D + c] = s_out[e * OS + c];
            }
        if (wave == 0 && fin) {                                                  // E2E:100-101, 110-113: the state drawn at kernel start
            const float4 q = s_rst[lane];
            const unsigned fl = s_rflag[lane];
            nx[0] = q.w; nx[1] = 0.0f; nx[2] = 0.0f; nx[3] = q.x; nx[4] = q.y; nx[5] = q.z;
            reset_path = (int)(fl & 3u);
            virtual_next = (fl & 4u) != 0u;
            store_reset_state();
        }
        if (wave == 1 && fin)                                                    // the finished envs as a list, for the compact slot pass
            s_finlist[__popcll(finmask & ((1ull << lane) - 1ull))] = (uint8_t)lane;
        // E2E:102-103 (init_traffic, TRF:151-195): the pool of the finished envs re-enters clear of the NEW ego (s_rst: written before
        // barrier 1) — one lane per (finished env, candidate), not a sweep over the tile's records
        if (!A.flow_on) {
            for (int q = tid; q < n_fin * m_cand; q += NT) {
                const int k = fast_div(q, A.m_magic);
                respawn_fresh(nth_fin(k), q - k * m_cand, s_rst);
            }
        } else {
            // the flow source (ABI 5): Traffic.init_traffic's role for the finished envs = eb_traffic_flow_reset's arithmetic, one lane
            // per (finished env, SLOT): presence draw, depart position / speed, the conflict test against the NEW ego (TRF:168-192) —
            // to HBM and to the tile's LDS copy (record, mode byte, a bit in the route's candidate set: zeroed in front of the barrier
            // above), from which the reset observation is built below; a route's first slot also does the route's timer and count,
            // route 0's the env's clock and light.  (Round 5 began with one lane per (finished env, route) walking the route's K slots:
            // three 64-bit draws, a table read and the conflict test K times in a row on a dozen lanes — 7 us of the tail's 10.)
            const int K = A.flow_K;
            for (int q = tid; q < n_fin * m_cand; q += NT) {
                const int k = fast_div(q, A.m_magic), j = q - k * m_cand, e = nth_fin(k), ge = e0 + e;
                const int r = fast_div(j, A.k_magic), kk = j - r * K;
                const uint64_t env_base = (A.flow_reset_counter << 32) + (uint64_t)ge * 256u;
                const float per = A.flow_period[r];
                float expect = A.flow_lane_len / 7.5f / per;
                if (expect > (float)K) expect = (float)K;
                const float pp = expect / (float)K;
                const size_t sidx = (size_t)ge * m_cand + j;
                const float u0 = u01(A.flow_reset_seed, env_base + 4u * j), u1 = u01(A.flow_reset_seed, env_base + 4u * j + 1),
                            u2 = u01(A.flow_reset_seed, env_base + 4u * j + 2);
                bool on = u0 < pp;
                if (on) {
                    const float4 eg = s_rst[e];
                    const float ego6[6] = {eg.w, 0.0f, 0.0f, eg.x, eg.y, eg.z};
                    const float* ln = A.flow_lane + 5 * j;
                    const float along = u1 * A.flow_lane_len;
                    const float4 c = make_float4(ln[0] + along * ln[3], ln[1] + along * ln[4], u2 * A.flow_v_max[j], ln[2]);
                    reinterpret_cast<float4*>(A.cand)[sidx] = c;
                    s_cand[e * RS4 + j] = c;
                    if (init_conflict(ego6, 4.8f, c.x, c.y, c.w, c.z, A.flow_cand_len[j])) on = false;
                }
                A.flow_active[sidx] = on ? 1 : 0;
                A.flow_mode_out[sidx] = on ? (uint8_t)r : (uint8_t)EB_VMODE_EMPTY;
                s_tag[e * TS4 * 4 + j] = on ? (uint8_t)r : (uint8_t)EB_VMODE_EMPTY;
                if (ELIG && on) atomicOr(&s_elig32[(e * EB_VMODE_COUNT + r) * 2 + (j >> 5)], 1u << (j & 31));
                if (kk == 0) {
                    A.flow_timer[(size_t)ge * 12 + r] = u01(A.flow_reset_seed, env_base + 4u * (r * K) + 3) * per;
                    A.flow_emitted[(size_t)ge * 12 + r] = 0;
                    if (r == 0) {
                        A.flow_sim_step[ge] = 0;
                        const uint8_t ph = (A.flow_random_phase && u01(A.flow_reset_seed, env_base + 255u) > 0.5f) ? 2 : 0;   // TRF:158-161
                        A.flow_phase0[ge] = ph;
                        const uint8_t nl = A.training ? ph : 0;                                                           // TRF:222-223
                        A.v_light_out[ge] = nl;
                        s_col[e] = nl;            // (the step's collision flags are spent: the merge above read them) — the reset observation's light
                    }
                }
            }
        }
        ES_MARK(13);
        __syncthreads();   // barrier: the re-entered candidates, s_ego, the list
        ES_MARK(12);
        // E2E:116: the reset observation, built with the OLD virtual flag (v_light already cleared).  The step's own slot phase spends
        // four instruction streams (one mode per wave) on 64 lanes; here one or two lanes of a tile would be alive in each of them — at
        // 65 536 envs more than half of the tiles have a finished env, and the tail was issue-bound on masked-off lanes (8.5 us).  So:
        // wave 0 does the tracking of the new poses, wave 1 ALL (finished env, mode) pairs of the tile as lanes — one stream, the mode
        // per lane (range box, key spec and fill value by data) — and waves 2, 3 wait at the barrier.
        if (A.dm_ok) {
            if (wave == 0) {
                track_row(fin, reset_path);
            } else if (NW == 8 || wave == 1) {
                // (eight waves: the block has its CU nearly to itself and the stream's LENGTH is what counts — waves 1-7 take one distinct
                // mode each, lanes = the finished envs, the mode wave-uniform again: no divergence in the per-mode switches)
                const unsigned long long vmask = __builtin_amdgcn_ballot_w64(vflag);      // lane = env: the OLD flags of the tile
                const int n_pairs_f = n_fin * A.n_dm;
                const int rounds = NW == 8 ? (A.n_dm - (wave - 1) + (NW - 2)) / (NW - 1) : (n_pairs_f + 63) >> 6;
                for (int rd = 0; rd < rounds; ++rd) {
                    int ford, j;
                    bool act;
                    if (NW == 8) { j = wave - 1 + rd * (NW - 1); ford = lane; act = lane < n_fin; if (!act) ford = 0; }
                    else {
                        const int q = rd * 64 + lane;
                        act = q < n_pairs_f;
                        ford = act ? fast_div(q, A.dm_magic) : 0; j = act ? q - ford * A.n_dm : 0;
                    }
                    const int e = s_finlist[ford];
                    const float4 eg = s_rst[e];
                    const bool lit = ((vmask >> e) & 1ull) || (A.flow_on && s_col[e] != 0);                 // E2E:387-388: the OLD flag, or the light the flow source's reset set
                    pair_walk(act, e, j, eg.x, eg.y, lit);
                }
            }
        } else {                                                                 // a mode with more than two slots: the step's own slot code
            if (wave == 0) track_row(fin, reset_path);
            fill_slots(fin, vflag || (A.flow_on && lane < ET && s_col[lane < ET ? lane : 0] != 0), false);
        }
        ES_MARK(14);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the step's rows have left (phase 4, some 3 us ago) before
        __syncthreads();                                                         // another thread writes the finished envs' rows again
        for (int k = wave; k < n_fin; k += NW) {                                 // the reset rows: a wave per finished env, a lane per column
            const int e = nth_fin(k);
            for (int c = lane; c < D; c += 64) A.obs_out[((size_t)e0 + e) * D + c] = s_out[e * OS + c];
        }
        if (wave == 0 && fin) A.virtual_out[i] = virtual_next ? 1 : 0;           // E2E:120-126
        ES_MARK(15);
    }
}

template <int TASK, int ET, bool OBS, bool AUTO = false, int NW = 4>
__global__ __launch_bounds__(NW * 64, (ET == 16 && NW == 4 && !OBS && TASK != TASK_RIGHT) ? 6 : 1)   /* (six tiles per CU for the flow source's step, plain and with auto reset; the right-turn instantiations need 81-89 VGPRs — scratch under the bound — and stay at five) */ void env_step_kernel(const EnvStepArgs A) { env_step_body<TASK, ET, OBS, false, AUTO, NW>(A); }
template <int TASK, int ET, int NW = 4>
__global__ __launch_bounds__(NW * 64) void env_reset_pool_kernel(const EnvStepArgs A) { env_step_body<TASK, ET, true, true, false, NW>(A); }

// The launches of ONE task's instantiations (step / observation / auto reset / masked reset x tile shapes x waves per block): a template,
// explicitly instantiated once per task in a translation unit of its own — the three tasks' ~20 kernels each compile side by side (the
// file took 95 s of a 115 s build as one unit).  Shape decisions (tile, waves, LDS bytes) are the caller's: launch_env_step.
template <int TASK>
hipError_t launch_env_step_task(const EnvStepArgs& A, int ET, bool w8, int n_blocks, size_t lds, int dev, hipStream_t s) {
    hipError_t e = hipSuccess;
    const dim3 g(n_blocks), b(w8 ? 512 : 256);
#define EB_ENV_STEP_W(T, E, O, AU, W)                                                                                 \
    do {                                                                                                             \
        static size_t granted[64];   /* the > 48 KB opt-in is per kernel and device, and sticky */                   \
        if (lds > 48 * 1024 && lds > granted[dev]) {                                                                 \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_step_kernel<T, E, O, AU, W>),                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
            if (e == hipSuccess) granted[dev] = lds;                                                                 \
        }                                                                                                            \
        if (e == hipSuccess) hipLaunchKernelGGL((env_step_kernel<T, E, O, AU, W>), g, b, lds, s, A);                 \
    } while (0)
#define EB_ENV_STEP(T, E, O, AU) do { if ((E) <= 32 && w8) EB_ENV_STEP_W(T, (E) <= 32 ? (E) : 32, O, AU, 8); else EB_ENV_STEP_W(T, E, O, AU, 4); } while (0)
#define EB_ENV_RESET_W(T, E, W)                                                                                       \
    do {                                                                                                             \
        static size_t granted[64];                                                                                   \
        if (lds > 48 * 1024 && lds > granted[dev]) {                                                                 \
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&env_reset_pool_kernel<T, E, W>),                  \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
            if (e == hipSuccess) granted[dev] = lds;                                                                 \
        }                                                                                                            \
        if (e == hipSuccess) hipLaunchKernelGGL((env_reset_pool_kernel<T, E, W>), g, b, lds, s, A);                  \
    } while (0)
#define EB_ENV_RESET(T, E) do { if ((E) <= 32 && w8) EB_ENV_RESET_W(T, (E) <= 32 ? (E) : 32, 8); else EB_ENV_RESET_W(T, E, 4); } while (0)
#define EB_ENV_STEP_T(T)                                                                                             \
    do {                                                                                                             \
        if (A.reset) { if (ET == 16) EB_ENV_RESET(T, 16); else if (ET == 32) EB_ENV_RESET(T, 32); else EB_ENV_RESET(T, 64); } \
        else if (A.obs_only) { if (ET == 16) EB_ENV_STEP(T, 16, true, false); else if (ET == 32) EB_ENV_STEP(T, 32, true, false); else EB_ENV_STEP(T, 64, true, false); } \
        else if (A.auto_reset) { if (ET == 16) EB_ENV_STEP(T, 16, false, true); else if (ET == 32) EB_ENV_STEP(T, 32, false, true); else EB_ENV_STEP(T, 64, false, true); } \
        else { if (ET == 16) EB_ENV_STEP(T, 16, false, false); else if (ET == 32) EB_ENV_STEP(T, 32, false, false); else EB_ENV_STEP(T, 64, false, false); } \
    } while (0)
    EB_ENV_STEP_T(TASK);
#undef EB_ENV_STEP_T
#undef EB_ENV_RESET
#undef EB_ENV_RESET_W
#undef EB_ENV_STEP
#undef EB_ENV_STEP_W
    return e != hipSuccess ? e : hipGetLastError();
}

}  // namespace eb
