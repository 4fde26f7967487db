// eb_kernels.h — host-visible launch interface between eb_capi.hip and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eb_device.h"

namespace eb {

constexpr int EB_MAX_VEH_SLOTS = 64;

struct VehModes {
    uint8_t turn[64];   // TURN_* per slot (predict_for_a_mode, DAM:416-421)
    uint8_t mode[64];   // EB_VMODE_* per slot
};

// ---- fused rollout step (eb_rollout.hip): one block = one env wave + RW record waves ----
struct FusedArgs {
    int storage_f16;           // 0: obs rows are fp32; 1: IEEE binary16 (obs_in / obs_out then point at uint16 rows)
    const float* obs_in;
    const float* actions;
    const int* ref_idx;
    float* obs_out;
    float* out5;
    float* scaled_actions;
    const PathTables* dt;      // full tables (look-ahead points when n_future > 0) and the slot turns
    const float* xy10;         // all paths' stride-10 (x, y) pairs back to back, readable 4 entries past the end
    const float* phi10;        // their headings, same indexing
    const float* rad_all;      // 3 x 32 block radii of the pruned search (positions outside the cell grid)
    const uint32_t* cells;     // closest-point cell grid, PathTables::cells
    float gx0, gy0;
    int gnx, gny;
    int red_off[3], red_len[3], n_paths;
    int n_env, obs_dim, n_veh, n_future;
    int envs_per_tile;         // whole envs per block: <= 64 and envs_per_tile * n_veh <= RW * 64 * RPT
    unsigned nv_magic;         // ceil(2^32 / n_veh): item / n_veh == umulhi(item, nv_magic)
    int path_id, training;
    int actions_raw;           // 1: raw [-1,1] actions (rollout_out), 0: already scaled
    int do_rewards;            // 0: compute_next_obses only
    int rolling;               // per-step kernel, 2048-record tile: 1 = three record loads in flight per lane, record k + 3 requested when
                               // record k is done; 0 = every record requested up front (eb_capi.hip:rollout_fused decides)
    int by_progress;           // per-step kernel: 1 = a record wave's issue priority falls as it advances
    long long* trace;          // profiling aid (eb_debug_set_trace): [n_waves][8] s_memrealtime marks, or NULL
    long long trace_words;     //   its capacity in 64-bit words: a mark past it is dropped
    int scan_one_trip;         // A/B aid (eb_debug_set_scan_prefetch 0): the closest-point range one group of entries per loop trip, as rounds 1-4
    // step gates of eb_rollout_gated (tape kernel only; all NULL / 0 otherwise)
    const unsigned* gate_ready;   // [horizon]: step t may start once gate_ready[t] != 0 (written by the action producer)
    unsigned* gate_done;          // [horizon]: += 1 per block once step t's outputs are visible device-wide
    void* gate_obs;               // [horizon, n_env, D] or NULL: the obs after every step, written through to memory
    unsigned* gate_status;        // [0] = 1 when a gate was not opened within gate_spin polls (the launch gives up)
    int gate_spin;
    int stage_entries;            // tape / gated kernels: > 0 = copy this many stride-10 table entries (+ 4 readable past the end) into LDS
    // episodic accumulator (eb_rollout_step_acc; per-step kernel only; all NULL / 0 otherwise).  Store-only: a launch leaves one
    // ACC_RECORD_DOUBLES record per block and step — no read-modify-write, nothing to fetch before the kernel's first load.
    double* acc_rec;           // this step's records [grid][ACC_RECORD_DOUBLES] (travels as a kernel parameter of its own: FusedHot)
    const float* prev_out5;    // the out5 array of the rollout's PREVIOUS step (NULL for its first): that step's record is made by
                               // THIS launch, under the wait for the record waves — the values come back as three coalesced loads
    double* prev_rec;          // where the previous step's records go
    double* acc_final;         // non-NULL in the rollout's last launch: [grid][2] = (sum, max) of |delta_y| of the rows it writes;
                               // that launch also makes its own step's record, at its tail
    // the safety shield's accumulation inside the step's launch (eb_shield_is_safe; per-step kernel only; NULL otherwise):
    // punish[i] = (first ? 0 : punish[i]) + out5[row][i]  (hier_decision.py:93-97), safe[i] = !(punish[i] > 0) in the last look-ahead
    float* shield_punish;
    uint8_t* shield_safe;
    int shield_row;            // 2: real_punish_term, 3: veh2veh4real (rows of out5)
    int shield_first, shield_last;
};
// a block's record of one step: [0..2] = float64 sums of reward, punish_term_for_training, real_punish_term over its envs,
// [3] = bit e set when env e of the tile had real_punish_term > 0 in that step (a 64-bit mask in the double's bytes)
constexpr int ACC_RECORD_DOUBLES = 4;
// variant: 0 = 4 record waves x 8 records per lane (2048-record tiles), 1 = 4 x 4 (1024), 2 = 1 x 4 (256)
int fused_tile_records(int variant);
// the tile shape the open-loop tape kernel takes when the per-step kernel would take `variant` (eb_rollout.hip)
int tape_tile_variant(int variant, int n_veh, int storage_f16);
hipError_t launch_rollout_fused(int task, int variant, const FusedArgs& A, int grid, hipStream_t s);
// open-loop rollout of `horizon` steps in one launch: A.actions = tape [horizon, n_env, 2], A.out5 = [horizon, 5, n_env]
hipError_t launch_rollout_tape_fused(int task, int variant, const FusedArgs& A, int horizon, int grid, hipStream_t s);
// blocks of the tape kernel of `variant` that can be resident on one CU at once (a gated rollout needs its whole grid resident)
int tape_blocks_per_cu(int task, int variant, int n_veh, int storage_f16, size_t dyn_bytes);
// eb_gate_feed: the reference action producer of a gated rollout (one block)
hipError_t launch_gate_feed(int horizon, int n_blocks, size_t step_bytes, const void* staged, void* live,
                            unsigned* gate_ready, const unsigned* gate_done, unsigned* status, int spin, hipStream_t s);

hipError_t launch_f_xu(int n, const float* st, const float* ac, float tau, float* nx, float* pr, hipStream_t s);
hipError_t launch_action_transform(int n, const float* in, float* out, hipStream_t s);
hipError_t launch_rewards(int task, int n_env, int D, int n_future, int NV, const float* obs, const float* act,
                          float* out5, float* d16, hipStream_t s);
hipError_t launch_tracking(int task, int n, const PathTables& pt, const float* xs, const float* ys, const float* phis,
                           const float* vs, const int* ref_idx, int path_id, int n_future, int ratio, float* out,
                           int* out_index, float* out_points, hipStream_t s);
hipError_t launch_path_points(int n, const PathTables& pt, const int* index, const int* ref_idx, int path_id, int n_future,
                              float* out, hipStream_t s);
hipError_t launch_phi_diff(int n, const float* in, float* out, hipStream_t s);
hipError_t launch_ego_predict(int n, const float* ego, const float* actions, float* next, hipStream_t s);
hipError_t launch_veh_predict(int n_env, int NV, const VehModes& modes, const float* veh, float* out, hipStream_t s);
hipError_t launch_ss(int task, int n_env, int D, int n_future, int NV, const PathTables& pt, const VehModes& modes,
                     const float* obs, const float* actions, const int* ref_idx, int path_id, int training,
                     float one_m_lam, float* out, hipStream_t s);

constexpr int SUMMARY_MAX_PARTS = 1024;   // blocks of the stage-1 summary reduction (handle scratch: 6 doubles each)
hipError_t launch_summary(int n_env, int horizon, int D, const float* out5_steps, const float* obs_final,
                          double* partials, int max_parts, float* out8, hipStream_t s);
// fold of the records an accumulating rollout left — records [horizon][n_blocks][ACC_RECORD_DOUBLES], finals [n_blocks][2] —
// -> the same 8 floats, one launch
hipError_t launch_acc_fold(int n_blocks, int n_env, int horizon, const double* records, const double* finals, float* out8,
                           hipStream_t s);

// real-env step pieces (eb_env_kernels.hip)
hipError_t launch_env_ego_step(int n, const float* ego, const float* actions, float* next_ego, float* params,
                               hipStream_t s);
hipError_t launch_env_pre(int task, int n_env, int D, int n_future, int NV, const float* obs, const float* raw,
                          float* scaled, float* out5, float* d16, float* ego, float* params, hipStream_t s);
bool get_obs_is_staged(int D, int m_cand, const float* cand);
// cos / sin of the four exit angles (multi_ego.py:33), evaluated by the host's libm in float64 exactly as the reference's
// math.cos / math.sin do; *_f: the same rounded to fp32 for the forward (index k) and inverse (index 4 + k) ego transform
struct ExitConsts {
    double c[4], s[4];
    float cf[8], sf[8];
};
// done_code != NULL appends _judge_done to the observation kernel (only in its LDS-staged form: get_obs_is_staged).
// exit_id != NULL: the 12-ego scene's frames (one thread per env, float64 vehicle coordinates); needs `xc`.
struct EnvResetArgs;
hipError_t launch_get_obs(int task, int n_env, int D, int n_future, int NV, const PathTables& pt,
                          const VehModes& modes, const float* ego, const int* ref_idx, int path_id, int m_cand,
                          const float* cand, const uint8_t* cand_mode, const uint8_t* v_light, const uint8_t* virtual_flag,
                          float* obs_out, hipStream_t s, const float* params = nullptr, const float* cand_lw = nullptr,
                          uint8_t* done_code = nullptr, const uint8_t* exit_id = nullptr, const ExitConsts* xc = nullptr,
                          const uint8_t* row_mask = nullptr, const EnvResetArgs* reset = nullptr,
                          int tile_envs = 0, int env_waves = 0, long long* trace = nullptr, long long trace_words = 0,
                          int scan_one_trip = 0);
                          // tile_envs / env_waves / trace: EnvStepArgs::tile_envs / waves / trace for the one-launch machinery
hipError_t launch_exit_frame(int n, const uint8_t* exit_id, int inverse, const ExitConsts& xc, const float* ego, float* out,
                             hipStream_t s);
hipError_t launch_env_reset(int task, int n_env, const PathTables& pt, const uint8_t* mask, uint64_t seed, uint64_t counter,
                            int training, float* ego, float* params, int* ref_idx, uint8_t* virtual_next, uint8_t* done_code,
                            hipStream_t s, uint8_t* v_light = nullptr, int* episode_step = nullptr);
// eb_env_step's separate-launch path: the step counts and the time-limit code behind eb_judge_done (eb_time_limit)
hipError_t launch_time_limit(int n_env, int* episode_step, int max_episode_steps, uint8_t* done_code, hipStream_t s);
// _get_ego_dynamics (E2E:150-183) for a batch: out [n, 11] = alpha_f_bound, alpha_r_bound, r_bound, 4 corner points (x, y)
hipError_t launch_ego_dynamics(int n, const float* ego, const float* params, float* out, hipStream_t s);
hipError_t launch_copy_rows_masked(int n_env, int D, const uint8_t* mask, const float* src, float* dst, hipStream_t s);
hipError_t launch_flag_swap(int n_env, const uint8_t* mask, const uint8_t* next, uint8_t* flag, hipStream_t s);
hipError_t launch_traffic_flow_reset(int n_env, int K, const uint8_t* mask, const float* ego, float* cand, uint8_t* active,
                                     float* timer, int* emitted, int* sim_step, uint8_t* phase0, const float* lane,
                                     const float* period, const float* v_max, const float* cand_len, float lane_len,
                                     int random_phase, int training, uint64_t seed, uint64_t counter, uint8_t* cand_mode,
                                     uint8_t* v_light, hipStream_t s);
hipError_t launch_judge_done(int task, int n_env, int D, const float* ego, const float* params, const float* obs,
                             int m_cand, const float* cand, const uint8_t* cand_mode, const float* cand_lw,
                             const uint8_t* v_light, uint8_t* done_code, hipStream_t s);

// CrossroadEnd2end.step as one launch (eb_env_step.hip)
struct SlotTurns { uint8_t t[64]; };
struct EnvStepArgs {
    int n_env, D, n_future, NV, m_cand, path_id;
    unsigned d_magic, m_magic, nv_magic;   // ceil(2^32 / d) for d = D, m_cand, NV (0 when d == 1): item / d == umulhi(item, magic)
    PathTables pt;
    VehModes modes;                        // the observation's slot modes (handle h)
    SlotTurns tturn;                       // turn class of every candidate slot (the traffic handle's modes)
    unsigned long long first_mask;         // bit s: slot s is the first slot of its mode
    // the slot plan as a table (env_step_slot_plan): entry j = mode | first slot << 8 | second slot << 16 (0xff: none) of the j-th
    // distinct slot mode; dm_ok: no mode has more than two slots (the native lists, UTL:21-23) — what the compact reset tail needs
    unsigned dm[12];                       // (EB_VMODE_COUNT entries)
    int n_dm, dm_ok;
    unsigned dm_magic;                     // item / n_dm == umulhi(item, dm_magic) (0: n_dm == 1)
    const float* obs;                      // [n_env, D] current observation
    const float* raw;                      // [n_env, 2] raw actions
    const int* ref_idx;
    float* ego;                            // [n_env, 6] in place
    float* params;                         // [n_env, 4] out
    float* cand;                           // [n_env, m_cand, 4] in place, 16-byte aligned
    const uint8_t* cand_mode;
    const float* cand_lw;                  // nullable
    const uint8_t* v_light;                // nullable
    const uint8_t* virtual_flag;           // nullable
    float* scaled;                         // nullable
    float* out5;
    float* d16;                            // nullable
    float* obs_out;
    uint8_t* done_code;
    const float* respawn_entry;            // NULL: no re-entry stage
    float limit, span, v_max;
    uint64_t seed, counter;
    long long* trace;                      // profiling aid (eb_debug_set_trace): [n_blocks * waves per block (4 or 8)][16] wall-clock marks, or NULL
    long long trace_words;                 //   its capacity in 64-bit words: a mark past it is dropped
    int waves;                             // 0: by grid size (launch_env_step); 4 / 8: forced (eb_debug_set_env_waves)
    int scan_one_trip;                     // A/B aid (eb_debug_set_scan_prefetch 0), as FusedArgs::scan_one_trip
    int by_progress;                       // 1: a block's issue priority falls from phase to phase (eb_debug_set_rollout_sched's by_progress; launch_env_step decides by the grid)
    int obs_only;                          // 1: eb_get_obs — ego / cand are inputs, only obs_out is written
    const uint8_t* row_mask;               // obs_only: nullable [n_env]; rows with a zero byte are left alone
    int tile_envs;                         // 0: by batch size (env_step_tile_envs); 16 / 32 / 64: forced (eb_debug_set_tile 2 / 1 / 0)
    // reset (eb_env_reset_pool as one launch; obs_only = 1 too): the rows of row_mask get a fresh ego (eb_env_reset's draws), a
    // fresh pool clear of that ego (respawn_entry / span / v_max / seed / counter / edge_span), their reset observation
    int reset;
    int training;
    uint64_t reset_seed, reset_counter;
    float edge_span;
    int* ref_idx_out;                      // [n_env] the drawn path
    uint8_t* virtual_out;                  // [n_env] == virtual_flag: the OLD flag feeds the observation, the drawn one replaces it after
    uint8_t* v_light_out;                  // nullable: cleared
    const uint8_t* done_src;               // nullable (obs: the observation source of the rows outside the mask, nullable)
    // the pool's part of a reset (reset = 1, or auto_reset = 1 next to the step's own re-entry rule above)
    const float* pool_entry;
    float pool_span, pool_v_max;
    uint64_t pool_seed, pool_counter;
    // auto_reset (eb_env_step, ABI 4): the rows whose done code is non-zero are reset in the same launch — reset_seed / reset_counter /
    // training / edge_span / ref_idx_out / virtual_out / v_light_out as for reset; final_obs (nullable) takes their terminal rows
    int auto_reset;
    float* final_obs;
    // flow (eb_env_step, ABI 4): eb_traffic_flow_step's rule applied on the way out (seed / counter above are its draws' keys;
    // v_light_out = the light it writes; flow_mode_out = cand_mode, rewritten)
    int flow_on, flow_K, flow_light_cycle;
    unsigned k_magic;                      // item / flow_K
    uint8_t* flow_active;
    float* flow_timer;
    int* flow_emitted;
    int* flow_sim_step;
    const float* flow_lane;
    const float* flow_period;
    const float* flow_v_max;
    float flow_dt, flow_exit_range, flow_accel, flow_lane_len;
    uint8_t* flow_mode_out;
    // flow_on && auto_reset (ABI 5): the flow source's part of reset — eb_traffic_flow_reset's arithmetic for the finished envs —
    // instead of the pool's re-entry (pool_* unused)
    const float* flow_cand_len;
    uint8_t* flow_phase0;
    int flow_random_phase;
    uint64_t flow_reset_seed, flow_reset_counter;
    // eb_time_limit (eb_env_step, ABI 5): per-env steps of the running episode, + 1 per step; an env NO reference outcome has
    // finished takes EB_DONE_TIME_LIMIT when the count reaches max_episode_steps; the count of a finished env restarts at 0.
    // reset = 1 (eb_env_reset_pool): the masked rows' counts are cleared (max_episode_steps unused)
    int* episode_step;
    int max_episode_steps;
};
struct EnvResetArgs {                      // launch_get_obs(..., reset): what eb_env_reset_pool adds to a masked observation pass
    uint64_t seed, counter;                // eb_env_reset's
    int training;
    float* params;
    int* ref_idx;
    uint8_t* virtual_flag;
    uint8_t* v_light;                      // nullable
    uint8_t* done_code;                    // nullable
    const float* entry;                    // the pool rule
    float span, v_max, edge_span;
    uint64_t pool_seed, pool_counter;
    const float* obs_src;                  // nullable: the observation rows of the envs outside the mask
    const uint8_t* done_src;               // nullable: their done codes
    int* episode_step;                     // nullable: the masked rows' episode step counts are cleared
};
size_t env_step_lds_bytes(int D, int NV, int m_cand, int tile_envs, bool flow = false, bool four_waves = false);
int env_step_tile_envs(int n_env, int D, int NV, int m_cand, bool flow = false);
bool env_step_is_fused(int D, int NV, int m_cand, const float* cand, const float* ego = nullptr, const float* actions = nullptr,
                       const float* scaled = nullptr, const float* params = nullptr,   // NULL: not an argument of the call at hand
                       bool flow = false);                                             // the call carries an eb_flow_rule
void env_step_slot_plan(const VehModes& modes, int NV, EnvStepArgs& A);   // first_mask, dm, n_dm, dm_ok, dm_magic
hipError_t launch_env_step(int task, const EnvStepArgs& A, hipStream_t s);

hipError_t launch_traffic_respawn(int n_env, int m_cand, float* cand, const float* entry, float limit, float span,
                                  float v_max, uint64_t seed, uint64_t counter, const uint8_t* env_mask, uint8_t* respawned,
                                  hipStream_t s, const float* ego = nullptr, float edge_span = 0.0f);

hipError_t launch_traffic_flow_step(int n_env, int K, float* cand, uint8_t* active, float* timer, int* emitted,
                                    int* sim_step, const float* lane, const float* period, const float* v_max, float dt,
                                    float exit_range, float accel, float lane_len, int light_cycle, uint64_t seed,
                                    uint64_t counter, uint8_t* cand_mode, uint8_t* v_light, hipStream_t s);

// ---- policy network in the loop (eb_policy.hip): fused MLP on the f32 matrix cores ----
constexpr int MLP_ROWS = 64;       // observations per block
constexpr int MLP_THREADS = 256;   // 4 waves
constexpr int MLP_MAX_HIDDEN = 8;
enum { MLP_ACT_LINEAR = 0, MLP_ACT_RELU = 1, MLP_ACT_ELU = 2, MLP_ACT_TANH = 3 };
enum { MLP_HEAD_LOGITS = 0, MLP_HEAD_ACTION = 1 };
struct MlpLayer {
    const float* w;   // packed weights (pack_weights)
    const float* b;   // bias, zero-padded to the tile width
    int k_pad;        // inputs, padded to a multiple of 8
    int pad_;
};
struct MlpArgs {
    const float* obs;
    const float* scale;   // obs_scale or NULL
    float* out;           // logits [n, out_dim] or actions [n, out_dim / 2]
    int n, obs_dim, n_hidden;
    int units;            // padded hidden width: 64 / 128 / 256 / 512
    int out_dim, hidden_act, out_act, head;
    float action_range;
    int row_stride;       // LDS floats per row: max(k_pad of layer 0, units) + 4
    MlpLayer hid[MLP_MAX_HIDDEN];
    MlpLayer outl;
};
int mlp_padded_units(int n_units);
size_t mlp_lds_bytes(const MlpArgs& A);
void pack_weights(const float* kernel, int k_real, int cols_real, int k_pad, int col_tiles, float* out);
void pack_weights16(const float* kernel, int k_real, int cols_real, int k_pad, float* out);   // the output layer's 16-column tiles
hipError_t launch_mlp(const MlpArgs& A, hipStream_t s);
}  // namespace eb
