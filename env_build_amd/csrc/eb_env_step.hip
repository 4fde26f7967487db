// eb_env_step.hip — CrossroadEnd2end.step (E2E:132-144) for a batch of envs as ONE launch: the host side (tile shape, LDS sizing, the
// slot plan, the dispatcher) and the TASK_LEFT instantiations of the kernels in eb_env_step_body.h (TASK_STRAIGHT / TASK_RIGHT:
// eb_env_step_t1.hip / _t2.hip — the same header, a translation unit each, so that the three compile side by side).
#include "eb_env_step_body.h"

namespace eb {

size_t env_step_lds_bytes(int D, int NV, int m_cand, int tile_envs, bool flow, bool four_waves) {
    const int rs4 = m_cand + ((m_cand & 1) ? 2 : 1), os = D | 1, ts4 = ((m_cand + 3) >> 2) | 1;
    const size_t E = (size_t)tile_envs;
    size_t b = E * rs4 * 16;                     // s_cand
    b += E * os * 4;                             // s_out
    b += E * NV * 8;                             // s_part
    b += E * 16 * 2;                             // s_pts, s_ego
    b += E * 8;                                  // s_oldc
    b += E * ts4 * 4;                            // s_tag
    b += (size_t)4 * ES_QCAP * 2;                // s_queue
    if (flow) b += ((tile_envs == 16 && four_waves) ? 0 : E * 12 * 16 + E * 12 * 4) + E * m_cand;   // s_new, s_emit (not on 16-env tiles x four waves: registers, FUSED_FLOW), s_on (eb_flow_rule)
    return (b + 15) & ~(size_t)15;
}
// envs per block: 64 for throughput; small and medium batches take 32-env tiles (16 below 1 024 envs) with EIGHT waves per block
// (launch_env_step) — a step of a few thousand envs is latency, i.e. the length of a wave's instruction stream, not bandwidth.
// Many candidates per env (the flow source: 60) make a 64-env tile too big for four blocks per CU (> 40 KB of LDS): 16-env
// tiles then, at any batch size (measured at 65 536 envs x 60 candidates: 119 us with one 85 KB block per CU).
// Measured (16 candidates, us per step at tiles of 16 / 32 / 64 envs, round 4 — profiles/r4_env_tile_sweep.txt): 2 048 envs
// 7.3 / 7.2 / 11.0; 4 096: 7.5 / 7.3 / 11.1; 8 192: 8.5 / 7.6 / 11.4; 16 384: 14.9 / 9.2 / 11.9; 24 576: 21.4 / 14.3 / 13.2;
// 32 768: 27.4 / 16.4 / 13.6; 65 536: 51.7 / 29.4 / 18.2.  (Round 3, four waves everywhere: 4 096: 10.8 / 11.4 / 13.2.)
int env_step_tile_envs(int n_env, int D, int NV, int m_cand, bool flow) {
    if (env_step_lds_bytes(D, NV, m_cand, 64, flow) > 40 * 1024) return 16;
    return n_env <= 1024 ? 16 : n_env <= 20480 ? 32 : 64;
}

void env_step_slot_plan(const VehModes& modes, int NV, EnvStepArgs& A) {
    A.first_mask = 0; A.n_dm = 0; A.dm_ok = 1;
    for (int sl = 0; sl < NV; ++sl) {
        int first = -1, rank = 0;
        for (int t = 0; t < sl; ++t)
            if (modes.mode[t] == modes.mode[sl]) { if (first < 0) first = t; ++rank; }
        if (first < 0) {
            A.first_mask |= 1ull << sl;
            if (A.n_dm < EB_VMODE_COUNT) A.dm[A.n_dm] = (unsigned)modes.mode[sl] | (unsigned)sl << 8 | 0xffu << 16;
            ++A.n_dm;
        } else if (rank == 1) {
            for (int j = 0; j < A.n_dm && j < EB_VMODE_COUNT; ++j)
                if ((A.dm[j] & 0xffu) == modes.mode[sl]) A.dm[j] = (A.dm[j] & 0xffffu) | (unsigned)sl << 16;
        } else A.dm_ok = 0;              // a third slot of one mode
    }
    if (A.n_dm > EB_VMODE_COUNT) A.dm_ok = 0;   // (cannot happen: twelve mode ids)
    A.dm_magic = A.n_dm <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)A.n_dm - 1) / (unsigned)A.n_dm);
}

// cand and params are accessed as float4, ego / actions / scaled actions as float2: a buffer that is not aligned to its vector
// access (an offset view handed in through the C-ABI) takes the separate launches instead
// (the gate counts what the launch will ask for: the flow rule's arrays when the call carries one, and — ES_STATIC_LDS — the
// kernel's statically allocated tables, so that a shape just under the limit takes the separate launches instead of failing)
constexpr size_t ES_STATIC_LDS = 6 * 1024;
bool env_step_is_fused(int D, int NV, int m_cand, const float* cand, const float* ego, const float* actions,
                       const float* scaled, const float* params, bool flow) {
    auto al = [](const void* p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    return m_cand >= 1 && m_cand <= 64 && env_step_lds_bytes(D, NV, m_cand, 16, flow) + ES_STATIC_LDS <= 156 * 1024 &&
           al(cand, 16) && al(ego, 8) && al(actions, 8) && al(scaled, 8) && al(params, 16);
}

template hipError_t launch_env_step_task<TASK_LEFT>(const EnvStepArgs&, int, bool, int, size_t, int, hipStream_t);
extern template hipError_t launch_env_step_task<TASK_STRAIGHT>(const EnvStepArgs&, int, bool, int, size_t, int, hipStream_t);
extern template hipError_t launch_env_step_task<TASK_RIGHT>(const EnvStepArgs&, int, bool, int, size_t, int, hipStream_t);

hipError_t launch_env_step(int task, const EnvStepArgs& A_in, hipStream_t s) {
    EnvStepArgs A = A_in;
    int ET = A.tile_envs == 16 || A.tile_envs == 32 || A.tile_envs == 64 ? A.tile_envs
                                                                           : env_step_tile_envs(A.n_env, A.D, A.NV, A.m_cand, A.flow_on != 0);
    if (ET != 16 && env_step_lds_bytes(A.D, A.NV, A.m_cand, ET, A.flow_on != 0) + ES_STATIC_LDS > 156 * 1024) ET = 16;   // a forced shape that does not fit
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev = dev < 0 || dev >= 64 ? 0 : dev;
    // eight waves per block at small and medium batches (16- / 32-env tiles: few blocks per CU, the launch is a wave's instruction
    // stream) — the step, the observation and the masked reset alike —, four otherwise; A.waves = 4 / 8 forces it (eb_debug_set_env_waves)
    const int wforce = A.waves;
    static int n_cu[64];
    if (!n_cu[dev]) {
        int cu = 0;
        n_cu[dev] = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0 ? cu : 256;
    }
    const int n_blocks = (A.n_env + ET - 1) / ET;
    // issue priority by phase (eb_env_step_body.h; profiles/r6_ab_envprio.txt): the 64-env tiles' step 18.2 -> 17.6 us at 65 536 x 16, with
    // auto reset 23.8 -> 22.8; the flow source's 16-env tiles: with auto reset 62.0 -> 60.3, the plain step 53.2 -> 54.6 (off there);
    // nothing either way at 4 096 envs.  eb_debug_set_rollout_sched forces it
    if (A.by_progress < 0) A.by_progress = (ET == 64 || (A.flow_on && A.auto_reset)) ? 1 : 0;
    // (a grid of many small tiles — the flow source's 60 candidates force 16-env tiles at any batch size — is throughput again: with
    // eight waves per block only two blocks fit a CU's registers; measured at 65 536 x 60: 133 us against 104)
    const bool w8 = wforce != 4 && ET <= 32 && (n_blocks <= 3 * n_cu[dev] || wforce == 8);
    const size_t lds = env_step_lds_bytes(A.D, A.NV, A.m_cand, ET, A.flow_on != 0, !w8);
    if (A.trace && A.trace_words < (long long)n_blocks * (w8 ? 8 : 4) * 16) A.trace = nullptr;   // a mark buffer too small for this launch: no marks
    switch (task) {
        case TASK_LEFT: return launch_env_step_task<TASK_LEFT>(A, ET, w8, n_blocks, lds, dev, s);
        case TASK_STRAIGHT: return launch_env_step_task<TASK_STRAIGHT>(A, ET, w8, n_blocks, lds, dev, s);
        default: return launch_env_step_task<TASK_RIGHT>(A, ET, w8, n_blocks, lds, dev, s);
    }
}

}  // namespace eb
