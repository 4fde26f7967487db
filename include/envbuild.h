/*
 * envbuild.h — C-ABI of the MI355X-native hot path of idthanm/env_build.
 *
 * The reference has no FFI layer: its boundary is the Python class surface of
 * dynamics_and_models.py / endtoend.py.  Each entry point below replaces
 * the arithmetic of one reference method; the Python façade in env_build_amd/ keeps the
 * reference's class/method names and calls these through ctypes (INTEGRATION.md shows the
 * binding a maintainer would add to the reference itself).
 *
 * Two shared libraries export this identical symbol set:
 *   - env_build_amd/lib/libenvbuild_hip.so   (product; HIP kernels for gfx950; all data
 *     pointers are DEVICE pointers unless marked HOST)
 *   - oracle/_build/libenvbuild_oracle.so    (test infrastructure only; plain C; all data
 *     pointers are host pointers; `stream` ignored)
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (EB_E*); eb_last_error() returns a
 *     thread-local message.  No exceptions / aborts cross the ABI.
 *   - the caller owns every data buffer; a handle owns only its path tables / mode table.
 *   - a handle is bound to one device and is not thread-safe; distinct handles are independent.
 *   - launches are asynchronous on `stream` (a hipStream_t; NULL = the HIP null stream);
 *     eb_sync() blocks until all work on the handle's device is done.
 *   - all floating point is IEEE fp32, evaluated op-for-op in the reference's order with no FMA
 *     contraction; sin/cos/atan use the deterministic kernels documented in DESIGN.md so that
 *     the HIP path and the oracle agree bit-for-bit.
 *   - row layouts are the reference's: obs row = [ego 6 | tracking 3*(n_future+1) | veh 4*n_veh]
 *     (DAM:104-106, 189-194, 356), fp32, row-major [n_env, obs_dim].
 */
#ifndef ENVBUILD_H
#define ENVBUILD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EB_ABI_VERSION 5

/* error codes */
#define EB_OK 0
#define EB_EINVAL (-1)   /* bad argument (task, sizes, null pointer) */
#define EB_ESTATE (-2)   /* handle not fully configured (paths / modes missing) */
#define EB_EDEVICE (-3)  /* HIP runtime error, no device, launch failure */
#define EB_ENOMEM (-4)

/* task ids: training_task strings of the reference (DAM:91, E2E:46) */
#define EB_TASK_LEFT 0
#define EB_TASK_STRAIGHT 1
#define EB_TASK_RIGHT 2

/* mode: DAM:93 / DAM:334 — 'training' selects the path per env by ref_indexes (DAM:342-353);
 * anything else tracks the single current path of ref_path (DAM:334-339). */
#define EB_MODE_TRAINING 0
#define EB_MODE_SELECTING 1

/* vehicle mode ids, in the order of the twelve lists of E2E:354 */
#define EB_VMODE_DL 0
#define EB_VMODE_DU 1
#define EB_VMODE_DR 2
#define EB_VMODE_RD 3
#define EB_VMODE_RL 4
#define EB_VMODE_RU 5
#define EB_VMODE_UR 6
#define EB_VMODE_UD 7
#define EB_VMODE_UL 8
#define EB_VMODE_LU 9
#define EB_VMODE_LR 10
#define EB_VMODE_LD 11
#define EB_VMODE_COUNT 12
#define EB_VMODE_EMPTY 255 /* unused candidate slot */

/* done codes of eb_judge_done, in the priority order of E2E:200-221 */
#define EB_DONE_NOT_YET 0
#define EB_DONE_COLLISION 1
#define EB_DONE_BREAK_ROAD 2
#define EB_DONE_DEVIATE 3
#define EB_DONE_STABILITY 4
#define EB_DONE_RED_LIGHT 5
#define EB_DONE_GOOD 6
/* not a reference outcome: the episode's step count reached eb_time_limit.max_episode_steps (gym's TimeLimit wrapper around the
 * registered env, README.md:55-59: max_episode_steps = 200) while every predicate above said "not done yet" */
#define EB_DONE_TIME_LIMIT 7

/* exit of the crossroad an ego enters from — the 12-ego scene's frames (multi_env/multi_ego.py:33 ROTATE_ANGLE =
 * D 0, R 90, U 180, L -90 degrees; E2E:345-348 name_settings) */
#define EB_EXIT_D 0
#define EB_EXIT_R 1
#define EB_EXIT_U 2
#define EB_EXIT_L 3

#define EB_MAX_PATHS 3
#define EB_MAX_VEH 64

/* Angle wrapping.  The reference wraps headings with `while` loops (UTL:134-139, 232-237: eb_env_ego_step / eb_env_step,
 * eb_exit_frame, eb_get_obs(exit_id), the traffic reset's conflict test).  Such a loop never ends for +-inf and spins for more
 * than 10^4 turns beyond +-3.6e6 degrees — on a GPU that is a hung device, not a slow env.  Both backends therefore return a
 * value whose magnitude exceeds EB_WRAP_MAX_DEG (or is not a number) UNWRAPPED; every value the reference wraps in reasonable
 * time is wrapped by the same subtractions, bit for bit. */
#define EB_WRAP_MAX_DEG 3.6e6f

typedef struct eb_handle_s* eb_handle;

typedef struct eb_config {
    int32_t abi_version; /* EB_ABI_VERSION */
    int32_t task;        /* EB_TASK_* */
    int32_t n_veh;       /* vehicle slots per env, 1..EB_MAX_VEH (native: 8 / 9 / 5, UTL:40-42) */
    int32_t n_future;    /* num_future_data (DAM:91), >= 0 */
    int32_t mode;        /* EB_MODE_* */
    int32_t device;      /* HIP device ordinal; ignored by the oracle */
} eb_config;

const char* eb_last_error(void);
int eb_abi_version(void);
/* "hip" or "oracle" */
const char* eb_backend(void);

int eb_create(const eb_config* cfg, eb_handle* out);
int eb_destroy(eb_handle h);
int eb_sync(eb_handle h);

/* ReferencePath tables (DAM:598-700 builds them; host side stays Python).  HOST pointers.
 * xs/ys/phis: n_paths concatenated arrays, path k occupying lens[k] floats each. */
int eb_set_paths(eb_handle h, const float* xs, const float* ys, const float* phis,
                 const int32_t* lens, int32_t n_paths);
/* VEHICLE_MODE_LIST[task] (UTL:44-46): the mode of every vehicle slot as EB_VMODE_* ids.
 * HOST pointer, n == n_veh.  The turn class of predict_for_a_mode (DAM:416-421) and the
 * per-mode slot counts of _construct_veh_vector_short (E2E:449-451) derive from it. */
int eb_set_veh_modes(eb_handle h, const uint8_t* mode_id, int32_t n);

/* VehicleDynamics.f_xu (DAM:52-83).  states [n,6], actions [n,2] -> next [n,6], params [n,4]. */
int eb_f_xu(eb_handle h, int32_t n, const float* states, const float* actions, float tau,
            float* next_states, float* params, void* stream);

/* EnvironmentModel._action_transformation_for_end2end (DAM:128-132). [n,2] -> [n,2]. */
int eb_action_transform(eb_handle h, int32_t n, const float* actions, float* scaled,
                        void* stream);

/* EnvironmentModel.compute_rewards (DAM:186-320).  obs [n_env,D], actions [n_env,2] ALREADY
 * scaled.  out5 = 5 contiguous arrays of n_env floats: rewards, punish_term_for_training,
 * real_punish_term, veh2veh4real, veh2road4real.  out_dict16 (nullable) = 16 arrays of n_env in
 * the key order of DAM:302-318. */
int eb_compute_rewards(eb_handle h, int32_t n_env, const float* obs, const float* actions,
                       float* out5, float* out_dict16, void* stream);

/* EnvironmentModel.compute_next_obses (DAM:322-358): ego_predict + tracking + veh_predict.
 * ref_idx: [n_env] int32, used in training mode (may be NULL otherwise); path_id: the current
 * path of ref_path in the other modes. */
int eb_compute_next_obses(eb_handle h, int32_t n_env, const float* obs, const float* actions,
                          const int32_t* ref_idx, int32_t path_id, float* obs_out, void* stream);

/* EnvironmentModel.rollout_out (DAM:118-126): action transform -> rewards on the current obs ->
 * next obs, fused in one launch.  actions are the raw [-1,1] policy outputs.
 * scaled_actions (nullable) receives self.actions (DAM:120). */
int eb_rollout_step(eb_handle h, int32_t n_env, const float* obs_in, const float* actions,
                    const int32_t* ref_idx, int32_t path_id, float* obs_out, float* out5,
                    float* scaled_actions, void* stream);

/* Open-loop rollout of `horizon` steps over an action tape [horizon, n_env, 2] (the MPC callers'
 * cost_function, mpc/main.py:470-479).  out5_steps: [horizon, 5, n_env]; obs_work is a scratch
 * obs buffer [n_env, D]; obs_out receives the final obs.  Equivalent to `horizon` calls of
 * eb_rollout_step. */
int eb_rollout_tape(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in,
                    const float* action_tape, const int32_t* ref_idx, int32_t path_id,
                    float* obs_work, float* obs_out, float* out5_steps, void* stream);

/* fp16 state storage (BASELINE.json configs[4]: N_veh = 64, "fp16 state with fp32 reward accumulate").
 * Same as eb_rollout_step / eb_rollout_tape with the obs rows stored as IEEE binary16 (uint16_t bit patterns,
 * same column layout [ego 6 | tracking 3*(n_future+1) | veh 4*n_veh], row-major): every row is widened to
 * fp32 on load, ALL arithmetic of DAM:118-427 runs in fp32 exactly as in the fp32 entry points, rewards and
 * penalties are accumulated and returned in fp32, and the next obs is rounded to binary16 (round to nearest
 * even) on the store.  actions, ref_idx, out5 and scaled_actions keep their 32-bit types.
 * Algorithmic bytes per env-step: 68 + 16 * n_veh. */
int eb_rollout_step_f16(eb_handle h, int32_t n_env, const uint16_t* obs_in, const float* actions,
                        const int32_t* ref_idx, int32_t path_id, uint16_t* obs_out, float* out5,
                        float* scaled_actions, void* stream);
int eb_rollout_tape_f16(eb_handle h, int32_t n_env, int32_t horizon, const uint16_t* obs_in,
                        const float* action_tape, const int32_t* ref_idx, int32_t path_id,
                        uint16_t* obs_work, uint16_t* obs_out, float* out5_steps, void* stream);

/* The same H-step rollout in ONE launch with a GATE in front of every step — the closed-loop form without a host round
 * trip or a kernel boundary per step: step t starts once step_ready[t] != 0, which the producer of actions[t] (a policy
 * kernel on another stream; eb_gate_feed below is the reference producer) sets after its action stores have left for
 * memory; after step t every block b sets its own 64-byte record step_done[(t * n_blocks + b) * 16 + 0..15] = 1 (n_blocks =
 * eb_rollout_gated_blocks()) once its part of out5_steps[t] — and obs_steps[t], when given — is visible device-wide
 * (written through, drained, then flagged; one full 64-byte write per block: a shared counter would serialise a few
 * hundred atomics in one memory channel).  A consumer that finds word 0 of all n_blocks records of step t set may read
 * them and produce actions[t + 1].  Both sides of the hand-off go past the per-XCD L2: a consumer KERNEL polls the records
 * and reads out5_steps[t] / obs_steps[t] with device-scope loads (sc1 on gfx950; __hip_atomic_load at agent scope) — a plain
 * load may be served from a stale line of its own XCD's L2 — and a producer writes actions[t] with device-scope stores (or
 * writes back its L2) and waits for them (s_waitcnt vmcnt(0)) before it raises step_ready[t].
 *   action_tape [horizon, n_env, 2] is read step by step, each step after its gate (device-scope loads);
 *   obs_steps (nullable) [horizon, n_env, D]: the obs after every step; obs_work / obs_out / out5_steps as in
 *   eb_rollout_tape; step_ready: uint32 [horizon], step_done: uint32 [horizon, n_blocks, 16], device memory, step_done zeroed by the caller;
 *   status: uint32 [2] zeroed by the caller — [0] becomes 1 when a gate stayed shut for spin_limit polls (the launch
 *   then ends early and the results are void), [1] when eb_gate_feed gave up.
 * The whole grid has to be resident at once (otherwise a block that has not started would hold every gate shut), and
 * the producer beside it: n_env beyond HALF of the device's block slots is refused with EB_EINVAL.  Results equal `horizon` calls of eb_rollout_step bit for bit. */
int eb_rollout_gated(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in, const float* action_tape,
                     const int32_t* ref_idx, int32_t path_id, float* obs_work, float* obs_out, float* out5_steps,
                     float* obs_steps, const uint32_t* step_ready, uint32_t* step_done, int32_t n_blocks,
                     uint32_t* status, int32_t spin_limit, void* stream);
/* number of blocks eb_rollout_gated launches for n_env envs = the 64-byte records per step of step_done (0 if n_env is too
 * large).  The handle must be configured (paths and slot modes set) as for the launch: the answer depends on both.
 * eb_rollout_gated takes the number back as `n_blocks` and refuses (EB_EINVAL, nothing launched) when the grid it would
 * launch NOW differs — the tile shape (eb_debug_set_tile), the staging of the path tables and the occupancy are
 * re-derived from the handle's state at every call, and step_done was sized from the earlier answer. */
int eb_rollout_gated_blocks(eb_handle h, int32_t n_env, int32_t* n_blocks);
/* The reference action producer for eb_rollout_gated, to be enqueued on ANOTHER stream before it: for t = 0 .. horizon - 1
 * wait until all n_blocks records of step_done[t - 1] are set (t > 0), copy staged_tape[t] -> live_tape[t] ([n_env, 2] floats, n_env even),
 * raise step_ready[t].  What a policy kernel in the loop does, minus the policy.  stream = NULL: the handle's own
 * producer stream — a HIGH-PRIORITY stream, i.e. a hardware queue of its own: two streams of equal priority may share
 * one, and a producer queued behind the rollout it feeds (or the other way round) would never meet it.
 * Ordering: the feed reads staged_tape and the flags the caller zeroed; that stream has no order against the stream the
 * caller produced them on.  wait_after != 0: the feed is ordered (event record + stream wait, no host wait) behind
 * everything enqueued so far on `after_stream` (a hipStream_t; NULL = the null stream).  wait_after == 0: the caller
 * guarantees that staged_tape and the zero fills of step_ready / step_done / status are COMPLETE before the call. */
int eb_gate_feed(eb_handle h, int32_t n_env, int32_t horizon, int32_t n_blocks, const float* staged_tape,
                 float* live_tape, uint32_t* step_ready, const uint32_t* step_done, uint32_t* status,
                 int32_t spin_limit, void* after_stream, int32_t wait_after, void* stream);

/* Episodic-return summary of one shard of envs after a rollout of `horizon` steps — the only
 * quantity north_star exchanges between GPUs (one all-gather of this vector per rollout; the
 * reference's callers accumulate the same sums step by step, hier_decision.py:96).
 *   out5_steps [horizon, 5, n_env] as written by eb_rollout_tape / eb_plan_launch;
 *   obs_final  [n_env, D] the obs after the last step.
 *   out8[0] = sum rewards, [1] = sum punish_term_for_training, [2] = sum real_punish_term (all over
 *   steps and envs, accumulated in float64, rounded once), [3] = number of envs with
 *   real_punish_term > 0 at any step, [4] = sum |delta_y| of obs_final, [5] = max |delta_y|,
 *   [6] = n_env, [7] = horizon.  Deterministic (fixed reduction order). */
#define EB_SUMMARY_LEN 8
int eb_episode_summary(eb_handle h, int32_t n_env, int32_t horizon, const float* out5_steps,
                       const float* obs_final, float* out8, void* stream);

/* The same summary collected BY the rollout launches (ABI 5) — what the reference's callers do when they add up the
 * returns of rollout_out step by step (hier_decision.py:96, multi_ego.py:195) instead of re-reading them afterwards:
 *   eb_rollout_step_acc = eb_rollout_step (DAM:118-126; same outputs, same bits) as step `step` of a rollout of `horizon`
 *       steps that also leaves, in the workspace `acc`, the per-tile float64 sums of a step's rewards, punish_term_for_training
 *       and real_punish_term and the "real_punish_term > 0" flags of its envs.  prev_out5 = the out5 array the rollout's
 *       PREVIOUS step wrote (NULL for step 0): the HIP library sums a step's outputs in the launch of the NEXT step, where they
 *       cost three coalesced loads and a wait the kernel has anyway, and the rollout's last launch (step == horizon - 1) adds its
 *       own and the |delta_y| statistics of the final obs it writes.  Nothing is read from acc: no zero fill, no step order
 *       beyond the data dependence of the rollout itself.
 *   eb_episode_acc_finish: acc -> out8, the 8 floats of eb_episode_summary(out5_steps of those steps, the last obs_out)
 *       (sums within rtol 1e-6 of it — another fixed float64 order —, count and maximum equal), one small launch.
 * acc: device memory of eb_episode_acc_bytes(n_env, horizon) bytes, 16-byte aligned, private layout, one per rollout in flight;
 * every step of one rollout must go to the same handle with the same n_env and horizon (and the same eb_debug_set_tile
 * setting: a block writes its own records).  The summary is then one pass over horizon x n_blocks x 32 bytes instead of a second
 * pass over out5_steps [horizon, 5, n_env] — for a caller that does not keep out5_steps around; where it is kept anyway (the plans,
 * bench.py) the second pass is the cheaper form on an MI355X: the accumulating launch costs 0.2 us more than the plain one. */
int eb_episode_acc_bytes(eb_handle h, int32_t n_env, int32_t horizon, int64_t* bytes);
int eb_rollout_step_acc(eb_handle h, int32_t n_env, const float* obs_in, const float* actions,
                        const int32_t* ref_idx, int32_t path_id, float* obs_out, float* out5,
                        float* scaled_actions, void* acc, int32_t step, int32_t horizon, const float* prev_out5,
                        void* stream);
int eb_episode_acc_finish(eb_handle h, int32_t n_env, int32_t horizon, const void* acc, float* out8, void* stream);

/* A rollout plan = eb_rollout_tape over FIXED buffers, recorded once and replayed: the HIP library
 * captures the `horizon` launches into a hipGraph, so that a replay costs one host call instead of `horizon`.
 * acc != NULL (ABI 5): the launches are the accumulating ones into the caller's workspace; summary8 != NULL then adds
 * eb_episode_acc_finish -> summary8 to the plan, summary8 == NULL leaves the fold to the caller.  acc == NULL, summary8 != NULL:
 * eb_episode_summary(out5_steps, obs_out) -> summary8 behind the launches.  Buffer contents may change between launches,
 * addresses and sizes may not.  The plan borrows every buffer; destroy it before the handle. */
typedef struct eb_plan_s* eb_plan;
int eb_plan_create(eb_handle h, int32_t n_env, int32_t horizon, const float* obs_in,
                   const float* action_tape, const int32_t* ref_idx, int32_t path_id,
                   float* obs_work, float* obs_out, float* out5_steps, float* summary8, void* acc,
                   eb_plan* out);
int eb_plan_launch(eb_plan p, void* stream);
int eb_plan_destroy(eb_plan p);

/* Stream-ordered timing marks for benchmarks (hipEvent pairs on the launch stream).
 * eb_event_elapsed_ms blocks until `stop` has completed.  The oracle uses the host clock. */
typedef struct eb_event_s* eb_event;
int eb_event_create(eb_handle h, eb_event* out);
int eb_event_record(eb_event e, void* stream);
int eb_event_elapsed_ms(eb_event start, eb_event stop, float* ms);
int eb_event_destroy(eb_event e);

/* ReferencePath.find_closest_point (DAM:702-715): argmin over every `ratio`-th path point (the reference's default
 * and every caller on the hot path use ratio = 10; any ratio >= 1 is accepted).  out_index: [n] int32 (already
 * multiplied by ratio); out_points: 3 arrays of n floats (x, y, phi).  ref_idx nullable -> path_id for every row. */
int eb_find_closest_point(eb_handle h, int32_t n, const float* xs, const float* ys,
                          const int32_t* ref_idx, int32_t path_id, int32_t ratio, int32_t* out_index,
                          float* out_points, void* stream);

/* ReferencePath.indexs2points (DAM:726-733) and future_n_data (DAM:717-724): for k = 0 .. n_future the point of the
 * row's path at index idx_k, idx_0 = clamp(index, 0, len - 1) and idx_k = min(idx_{k-1} + 80, len - 2) (the look-ahead
 * rule, applied to the UNclamped running index as DAM:719-722 does).  out_points [n_future + 1, 3, n] (x, y, phi). */
int eb_path_points(eb_handle h, int32_t n, const int32_t* index, const int32_t* ref_idx, int32_t path_id,
                   int32_t n_future, float* out_points, void* stream);

/* deal_with_phi_diff (DAM:577-580): one wrap of a heading difference into [-180, 180]. */
int eb_phi_diff(eb_handle h, int32_t n, const float* phi_diff, float* out, void* stream);

/* EnvironmentModel.ego_predict (DAM:386-392): f_xu at 10 Hz on the ego columns, v_x clipped to [0, 35].
 * ego [n,6], actions [n,2] SCALED -> next [n,6]. */
int eb_ego_predict(eb_handle h, int32_t n, const float* ego, const float* actions, float* next_ego, void* stream);

/* ReferencePath.tracking_error_vector (DAM:735-770). out [n, 3*(n_future+1)]. */
int eb_tracking_error(eb_handle h, int32_t n, const float* xs, const float* ys,
                      const float* phis, const float* vs, const int32_t* ref_idx,
                      int32_t path_id, int32_t n_future, float* out, void* stream);

/* EnvironmentModel.veh_predict (DAM:394-427). veh [n_env, 4*n_veh] -> same shape. */
int eb_veh_predict(eb_handle h, int32_t n_env, const float* veh, float* veh_out, void* stream);

/* EnvironmentModel.ss (DAM:134-184). actions raw [-1,1]; lam is the python float (double). out [n_env]. */
int eb_ss(eb_handle h, int32_t n_env, const float* obs, const float* actions,
          const int32_t* ref_idx, int32_t path_id, double lam, float* out, void* stream);

/* ---- real-env step pieces (endtoend.py), batched over n_env independent single-ego envs ---- */

/* CrossroadEnd2end._get_next_ego_state (E2E:269-283): f_xu at 10 Hz, v_x floored at 0 (NOT
 * clipped at 35), phi wrapped by deal_with_phi (UTL:232-237).  actions are SCALED (E2E:133).
 * ego [n,6] -> next_ego [n,6], params [n,4] (alpha_f, alpha_r, miu_f, miu_r). */
int eb_env_ego_step(eb_handle h, int32_t n, const float* ego, const float* actions,
                    float* next_ego, float* params, void* stream);

/* CrossroadEnd2end._get_obs (E2E:285-303) = ego vector (E2E:329-338) | tracking_error_vector on
 * the env's path | _construct_veh_vector_short (E2E:340-464).
 *   ego [n_env,6]; ref_idx nullable [n_env] (else path_id);
 *   cand [n_env, m_cand, 4] = (x, y, v, phi) of every vehicle in all_vehicles;
 *   cand_mode [n_env, m_cand] = EB_VMODE_* of each (route classified by the caller, E2E:355-385)
 *   or EB_VMODE_EMPTY;  v_light [n_env] (nullable = 0) the light phase and virtual_flag [n_env] (nullable = 0)
 *   virtual_red_light_vehicle: the stop-line cars of E2E:386-390 appear when v_light != 0 or the flag is set.
 *   exit_id NULL: the plain env (exit_ = 'D', candidates already in the ego's frame).
 *   exit_id [n_env] (EB_EXIT_*): the 12-ego scene, multi_ego.py:84-104 — cand / cand_mode / v_light are WORLD
 *   values shared by the egos; per env the candidates go through cal_info_in_transform_coordination (UTL:160-181:
 *   x' = x cos a + y sin a, y' = -x sin a + y cos a, phi' = wrap(phi - a) in float64, a = ROTATE_ANGLE of the exit), their
 *   routes are renamed relative to the exit (E2E:345-385: world direction d -> (d - exit) mod 4), the light follows
 *   multi_ego.py:89-92 (exits R / L see phase 2 as 0 and every other phase as 2); `ego` must already be in the
 *   exit's frame (eb_exit_frame).  Filters and sort keys use the float64 transformed values the way the reference's
 *   Python does (against ego-derived fp32 values after rounding to fp32, against constants in float64); the
 *   observation holds them rounded to fp32.  obs_out [n_env, D].
 *   An exit id above 3 is an error: the oracle (host arguments) returns EB_EINVAL; the HIP library cannot read device
 *   memory on the host and writes NaN into every column of that env's row instead (eb_exit_frame: NaN x, y, phi).
 *   row_mask (nullable, uint8 [n_env]): only the rows with a non-zero byte are computed and written, the others keep
 *   what obs_out holds — the observation pass of a masked reset (CrossroadEnd2end.reset(mask=...)) costs what the
 *   reset envs cost, not the batch. */
int eb_get_obs(eb_handle h, int32_t n_env, const float* ego, const int32_t* ref_idx,
               int32_t path_id, int32_t m_cand, const float* cand, const uint8_t* cand_mode,
               const uint8_t* v_light, const uint8_t* virtual_flag, const uint8_t* exit_id,
               const uint8_t* row_mask, float* obs_out, void* stream);

/* cal_ego_info_in_transform_coordination (UTL:184-196) for a batch: (x, y, phi) of ego [n,6] rotated into
 * (inverse = 0) or out of (inverse = 1: angle -a, multi_ego.py:118) the frame of each env's exit; the other
 * columns are copied.  fp32: x' = x * c + y * s, y' = -x * s + y * c with c, s = cos / sin of a * pi / 180 evaluated
 * in float64 and rounded to fp32 (NumPy >= 2 scalar semantics), phi' = wrap(phi - a) into (-180, 180]. */
int eb_exit_frame(eb_handle h, int32_t n, const uint8_t* exit_id, int32_t inverse, const float* ego,
                  float* ego_out, void* stream);

/* CrossroadEnd2end._judge_done (E2E:200-256) with Traffic.collision_check (TRF:263-295),
 * _get_ego_dynamics' r_bound and corner points (E2E:163-177) and judge_feasible (UTL:73-104).
 *   ego [n_env,6]; params [n_env,4]; obs [n_env,D] (delta_y = obs[:,6], E2E:224);
 *   cand / cand_mode as in eb_get_obs (every non-empty candidate takes part in the collision
 *   test); cand_lw [n_env, m_cand, 2] = (l, w) per candidate, NULL -> (4.8, 2.0);
 *   v_light [n_env] uint8.  done_code [n_env] uint8 = EB_DONE_*. */
int eb_judge_done(eb_handle h, int32_t n_env, const float* ego, const float* params,
                  const float* obs, int32_t m_cand, const float* cand, const uint8_t* cand_mode,
                  const float* cand_lw, const uint8_t* v_light, uint8_t* done_code, void* stream);


/* CrossroadEnd2end.step (E2E:132-144) for a batch of envs, as one call: action scaling (E2E:133) -> reward on the
 * CURRENT obs (E2E:134; out5 / out_dict16 (nullable) as in eb_compute_rewards) -> ego step (E2E:135, eb_env_ego_step) -> traffic step ->
 * observation (E2E:140, eb_get_obs with exit_id = NULL) -> done code (E2E:141, eb_judge_done).  The reference advances the traffic with
 * SUMO (TRF:220-238); here `traffic` is a second handle whose n_veh slots are the m_cand candidates of every env
 * (its slot modes = their modes) and the candidates move by the model's own prediction step (eb_veh_predict).
 * In-place state: ego [n_env,6], cand [n_env, m_cand, 4]; params [n_env,4] is written.  obs [n_env,D] is the
 * current observation (input), obs_out the next one; they must differ.  cand_lw (nullable) as in eb_judge_done,
 * v_light / virtual_flag (nullable) as in eb_get_obs.  scaled_actions and out_dict16 are nullable; scaled_actions may be the
 * actions array itself (scaling in place).
 * respawn (nullable): the traffic pool's re-entry rule applied AFTER the observation and the done code were taken (the
 * observation sees the pool as this step left it, the way the reference sees SUMO's state of the step) =
 * eb_traffic_respawn(traffic, n_env, m_cand, cand, entry, limit, span, v_max, seed, counter, NULL, NULL) as a seventh call.
 * auto_reset (nullable, ABI 4): "step, then reset the envs this step finished" — the loop of a vectorised driver
 * (hier_decision.py:109-135 steps, tests done and calls reset, E2E:99-127) as ONE call.  After the steps above, with
 * mask[e] = (done_code[e] != 0):
 *   final_obs[e, :] = obs_out[e, :] for the masked envs (nullable; the other rows of final_obs are not touched) — the
 *       terminal observation, Gym-vector's `final_observation`;
 *   eb_env_reset_pool(h, traffic, n_env, mask, seed, counter, training, ego, params, ref_idx, virtual_flag, v_light,
 *       NULL, NULL, m_cand, cand, cand_mode, &pool, obs_out, NULL, NULL): fresh state and flags, the pool re-entered clear of the
 *       new ego, v_light cleared, the reset observation (built with the OLD flag, E2E:116) in place of the terminal one,
 *       the drawn flag swapped in (E2E:120-126) — for the masked envs only.
 * done_code keeps the step's codes (what the driver reads to know who finished and why).  ref_idx / virtual_flag /
 * v_light inside the struct are the writable views of the arrays passed as the const arguments of the same name and must
 * BE those arrays (EB_EINVAL otherwise; ref_idx and virtual_flag must not be NULL, v_light may be).
 * Every argument is validated before the first launch: an error return leaves the state untouched.  Equivalent to the
 * six (seven) calls in that order (+ the reset above); the HIP library runs them as ONE launch (csrc/eb_env_step.hip)
 * when the candidate tile fits the LDS, m_cand <= 64 and cand / ego / actions / scaled_actions / params are aligned to
 * their vector accesses (16 / 8 / 8 / 8 / 16 bytes), and as separate launches otherwise. */
typedef struct eb_respawn {
    const float* entry; /* [m_cand, 5] = (x, y, phi, dx, dy) per slot, as eb_traffic_respawn */
    float limit;        /* a candidate with |x| or |y| beyond it re-enters (>= 0) */
    float span;
    float v_max;
    uint64_t seed;
    uint64_t counter;
    float edge_span;    /* eb_env_reset_pool only: where a candidate goes that would start on top of the ego (eb_traffic_respawn) */
} eb_respawn;
/* flow (nullable, ABI 4; not together with respawn; together with auto_reset since ABI 5 — "step, then reset the envs it finished"
 * over the flow source: after the flow step below, with mask[e] = (done_code[e] != 0): final_obs rows; eb_env_reset(h, n_env, mask, seed,
 * counter, training, ego, params, ref_idx, <next flags>, NULL, NULL); eb_traffic_flow_reset as spelled out at eb_auto_reset; eb_get_obs(h,
 * ..., v_light, virtual_flag, NULL, mask, obs_out) with the OLD flags (E2E:116) and the light the reset has just set; the drawn flags
 * swapped in — ONE launch in the HIP library under the conditions of eb_env_step): the step of the SUMO-free FLOW traffic source as the last stage
 * of the call — eb_traffic_flow_step(traffic, n_env, per_route, cand, active, timer, emitted, sim_step, lane, period, v_max, dt,
 * exit_range, accel, lane_len, light_cycle, seed, counter, cand_mode, v_light) AFTER the observation and the done code were taken
 * (they see the slots as this step's prediction left them and the modes / light the call was given; the exits, accelerations,
 * emissions, the new mode bytes and the new light are what the NEXT step sees).  m_cand must be 12 * per_route; cand_mode / v_light
 * inside the struct are the writable views of the call's arguments of the same name and must BE those arrays (v_light not NULL). */
typedef struct eb_flow_rule {
    int32_t per_route;
    uint8_t* active;         /* [n_env, m_cand] */
    float* timer;            /* [n_env, 12] */
    int32_t* emitted;        /* [n_env, 12] */
    int32_t* sim_step;       /* [n_env] */
    const float* lane;       /* [m_cand, 5] */
    const float* period;     /* [12] */
    const float* v_max;      /* [m_cand] */
    float dt, exit_range, accel, lane_len;
    int32_t light_cycle;
    uint64_t seed, counter;
    uint8_t* cand_mode;      /* == the cand_mode argument */
    uint8_t* v_light;        /* == the v_light argument */
} eb_flow_rule;
typedef struct eb_auto_reset {
    uint64_t seed, counter;  /* eb_env_reset's draws for the finished envs */
    int32_t training;        /* E2E:120-126: the virtual red-light flag is drawn in training mode only */
    int32_t* ref_idx;        /* == the ref_idx argument: the finished envs get their drawn path */
    uint8_t* virtual_flag;   /* == the virtual_flag argument */
    uint8_t* v_light;        /* == the v_light argument (nullable): cleared for the finished envs */
    eb_respawn pool;         /* the pool's part of reset: entry, span, v_max, seed, counter, edge_span (limit unused) */
    float* final_obs;        /* nullable [n_env, D]: the terminal observation rows of the finished envs */
    /* together with a flow rule (ABI 5; pool is then unused): the FLOW source's part of reset for the finished envs —
     * eb_traffic_flow_reset(traffic, n_env, flow->per_route, mask, ego, cand, flow->active, flow->timer, flow->emitted, flow->sim_step,
     * flow_phase0, flow->lane, flow->period, flow->v_max, flow_cand_len, flow->lane_len, flow_random_phase, training, flow_seed,
     * flow_counter, cand_mode, v_light) in the place of the pool's re-entry: Traffic.init_traffic (TRF:151-195) */
    const float* flow_cand_len; /* [m_cand] vehicle length per slot */
    uint8_t* flow_phase0;       /* [n_env] */
    int32_t flow_random_phase;
    uint64_t flow_seed, flow_counter;
} eb_auto_reset;
/* time_limit (nullable, ABI 5): the episode step limit of the REGISTERED env — callers reach CrossroadEnd2end through
 * gym.make('CrossroadEnd2end-v0') with max_episode_steps = 200 (README.md:55-59, mpc/main.py:542-576), i.e. inside gym's TimeLimit
 * wrapper: elapsed += 1 per step; elapsed >= max_episode_steps ends the episode, info['TimeLimit.truncated'] = not done.  Here:
 * episode_step [n_env] int32 (the caller's, zero at the start) is incremented by every step; an env whose done code came out
 * EB_DONE_NOT_YET takes EB_DONE_TIME_LIMIT once its count has reached max_episode_steps — below every reference outcome in
 * priority, so 'truncated' == (done_code == EB_DONE_TIME_LIMIT); a finished env (any code) is finished for auto_reset too, and its
 * count restarts at 0.  eb_env_reset / eb_env_reset_pool clear the counts of the rows they reset (their episode_step argument). */
typedef struct eb_time_limit {
    int32_t* episode_step;
    int32_t max_episode_steps; /* >= 1; the reference's registration: 200 */
} eb_time_limit;
int eb_env_step(eb_handle h, eb_handle traffic, int32_t n_env, const float* obs, const float* actions,
                const int32_t* ref_idx, int32_t path_id, float* ego, float* params, int32_t m_cand, float* cand,
                const uint8_t* cand_mode, const float* cand_lw, const uint8_t* v_light, const uint8_t* virtual_flag,
                float* scaled_actions, float* out5, float* out_dict16, float* obs_out, uint8_t* done_code,
                const eb_respawn* respawn, const eb_auto_reset* auto_reset, const eb_flow_rule* flow,
                const eb_time_limit* time_limit, void* stream);

/* CrossroadEnd2end._get_ego_dynamics (E2E:150-183) for a batch: the derived entries of the reference's ego dict from the ego state
 * [n, 6] and the tyre parameters [n, 4] = (alpha_f, alpha_r, miu_f, miu_r) as eb_env_ego_step / eb_env_step / eb_env_reset write them:
 *   out [n, 11] = alpha_f_bound, alpha_r_bound = 3 miu F_z / C (E2E:164-166; F_zf, F_zr the float64 values of vehicle_params,
 *   DAM:48, rounded to fp32), r_bound = miu_r g / (|v_x| + 1e-8) (E2E:167), then the four corner points (x, y) in the order of
 *   E2E:171-176 — (+l/2, +w/2), (+l/2, -w/2), (-l/2, +w/2), (-l/2, -w/2) through rotate_and_shift_coordination (UTL:152-157).
 * fp32 with the deterministic sin / cos; r_bound and the corners are the very values eb_judge_done / eb_env_step decide
 * 'break_stability' and 'break_road_constrain' on (the same device functions).  The copied entries of the dict (v_x ... miu_r,
 * l = 4.8, w = 2.0) are the inputs themselves. */
int eb_ego_dynamics(eb_handle h, int32_t n, const float* ego, const float* params, float* out, void* stream);

/* CrossroadEnd2end.reset (E2E:99-127) with _reset_init_state (E2E:472-499) for the envs of a batch whose mask byte is
 * non-zero (mask NULL = every env); the other envs keep their state.  Per env, with u_k in [0, 1) the counter-based
 * draws of eb_traffic_respawn keyed by (seed, counter, env, k):
 *   ref_idx = min(int(u_0 * n_paths), n_paths - 1)            a fresh ReferencePath's random choice (E2E:100, DAM:591)
 *   index   = int(u_1 * span) + 700, span = 1400 / 1700 / 920 for left / straight / right (E2E:473-478),
 *             clamped to the path like indexs2points (DAM:727-728)
 *   ego     = (8 * u_2, 0, 0, x, y, phi of that path point)   (E2E:480-499)
 *   params  = (0, 0, 0.8, 0.8)                                 (E2E:110-113: alpha_f, alpha_r, miu, miu)
 *   virtual_next = training ? (u_3 > 0.9) : 0                  (E2E:120-126 — the reference redraws the flag AFTER the
 *             reset observation; the caller swaps virtual_next in after its eb_get_obs)
 *   done_code = EB_DONE_NOT_YET;
 *   episode_step (nullable, ABI 5) = 0: eb_time_limit's count. */
int eb_env_reset(eb_handle h, int32_t n_env, const uint8_t* mask, uint64_t seed, uint64_t counter, int32_t training,
                 float* ego, float* params, int32_t* ref_idx, uint8_t* virtual_next, uint8_t* done_code,
                 int32_t* episode_step, void* stream);

/* CrossroadEnd2end.reset (E2E:99-127) over the traffic POOL for the envs of a batch whose mask byte is non-zero (NULL = all),
 * as ONE call — what a vectorised driver issues after every step for the envs that finished:
 *   eb_env_reset(h, n_env, mask, seed, counter, training, ego, params, ref_idx, <next flags>, done_code, episode_step)   E2E:100-101, 119
 *   eb_traffic_respawn(traffic, n_env, m_cand, cand, pool->entry, -1 (unconditional), pool->span, pool->v_max,
 *                      pool->seed, pool->counter, mask, NULL, ego, pool->edge_span)      E2E:102-103 (init_traffic, TRF:151-195)
 *   v_light[e] = 0 (nullable): the pool has no light programme, an episode starts at phase 0
 *   eb_get_obs(h, ..., v_light, virtual_flag, NULL, mask, obs): the reset observation, built with the OLD flags    E2E:116
 *   virtual_flag[e] = the flag eb_env_reset drew                                                       E2E:120-126
 * each for the masked envs only; the rows of the other envs (state, candidates, flags) are not touched.
 * obs_src / done_src (nullable): where the observation / done-code rows of the envs OUTSIDE the mask come from — a driver that
 * must leave the arrays of the last step as they were passes them here and fresh arrays as obs / done_code, and gets the whole
 * batch's current rows without a copy of its own.  NULL (or the same array): those rows of obs / done_code are left alone.
 * mask may be the previous done-code array itself (non-zero = reset) but must not be an array this call writes (EB_EINVAL).
 * The HIP library runs all of it as ONE launch (csrc/eb_env_step.hip, env_reset_pool_kernel) under the conditions of eb_env_step. */
int eb_env_reset_pool(eb_handle h, eb_handle traffic, int32_t n_env, const uint8_t* mask, uint64_t seed, uint64_t counter,
                      int32_t training, float* ego, float* params, int32_t* ref_idx, uint8_t* virtual_flag, uint8_t* v_light,
                      uint8_t* done_code, int32_t* episode_step, int32_t m_cand, float* cand, const uint8_t* cand_mode,
                      const eb_respawn* pool, float* obs, const float* obs_src, const uint8_t* done_src, void* stream);

/* The traffic pool's re-entry rule (the SUMO flows' role for the batched env, TRF:37-238 is out of scope): every
 * candidate of cand [n_env, m_cand, 4] that has left the square |x|, |y| <= limit is put back on its entry lane,
 *   (x, y, v, phi) = (entry.x + u1 * span * entry.dx, entry.y + u1 * span * entry.dy, u2 * v_max, entry.phi),
 * entry [m_cand, 5] = (x, y, phi, dx, dy) per slot.  u1, u2 in [0, 1) come from a counter-based generator —
 * the top 24 bits of splitmix64(seed + 0x9E3779B97F4A7C15 * (counter * 2^32 + env * 128 + slot * 2 + k)), k = 0, 1 —
 * so the result depends on (seed, counter, env, slot) only and the two libraries agree bit for bit.
 * env_mask (nullable, uint8 [n_env]): only the envs with a non-zero byte are touched; limit < 0 re-enters every
 * candidate of those envs (the pool's part of reset).  `respawned` (nullable, uint8 [n_env, m_cand]) marks the
 * slots that were re-entered.
 * ego (nullable, [n_env, 6]): Traffic.init_traffic's conflict rule for the pool (TRF:168-192 pushes conflicting cars away
 * when an episode starts): a re-entered candidate whose new pose conflicts with its env's ego — the box test of
 * TRF:183-184 in the ego's frame or in the vehicle's, both lengths 4.8 m — is put within `edge_span` metres of its lane's
 * start instead, (entry.x + u1 * edge_span * entry.dx, ...), so that no episode starts inside a collision. */
int eb_traffic_respawn(eb_handle h, int32_t n_env, int32_t m_cand, float* cand, const float* entry, float limit,
                       float span, float v_max, uint64_t seed, uint64_t counter, const uint8_t* env_mask,
                       uint8_t* respawned, const float* ego, float edge_span, void* stream);

/* One step of the SUMO-free FLOW traffic source (env_build_amd/traffic.py states the rules and where each number
 * comes from in sumo_files/cross.rou.xml and a.net.xml) AFTER the slots have been advanced by eb_veh_predict:
 * slots are route-major, `per_route` per route, 12 routes in EB_VMODE_* order (m = 12 * per_route <= 64).
 * Per env and route, in slot order: an active vehicle that is farther than exit_range from the centre and heading
 * away from it is dropped; the others accelerate, v = min(v + accel * dt, v_max[slot]).  Then timer += dt and, when
 * timer >= period[route] and the route has a vacant slot, the first vacant slot receives a vehicle at
 * lane[slot] + u1 * lane_len along the lane with speed u2 * v_max[slot] (u1, u2 as in eb_traffic_respawn, keyed by
 * the route's first slot), timer -= period, emitted += 1.  sim_step[env] += 1 and, when light_cycle != 0,
 * v_light[env] = phase of the 25 / 5 / 25 / 5 s programme at sim_step * dt.  cand_mode = route id or EB_VMODE_EMPTY.
 *   cand [n_env, m, 4], active uint8 [n_env, m], timer [n_env, 12], emitted int32 [n_env, 12], sim_step int32 [n_env],
 *   lane [m, 5] = (x, y, phi, dx, dy), period [12], v_max [m]; outputs cand_mode uint8 [n_env, m], v_light uint8 [n_env]. */
int eb_traffic_flow_step(eb_handle h, int32_t n_env, int32_t per_route, float* cand, uint8_t* active, float* timer,
                         int32_t* emitted, int32_t* sim_step, const float* lane, const float* period,
                         const float* v_max, float dt, float exit_range, float accel, float lane_len,
                         int32_t light_cycle, uint64_t seed, uint64_t counter, uint8_t* cand_mode, uint8_t* v_light,
                         void* stream);

/* Traffic.init_traffic's role (TRF:151-195) for the flow source, for the envs whose mask byte is non-zero (NULL = all):
 * per route r and slot j of the route (u_k keyed by (seed, counter, env, slot, k), the route / env draws by the route's
 * first slot with k = 3 and by index 255):
 *   a vehicle is present with probability p_r = min(per_route, lane_len / 7.5 / period[r]) / per_route (as many as
 *   the flow keeps on its approach lane at ~7.5 m/s), at lane[j] + u_1 * lane_len along the lane with speed
 *   u_2 * v_max[j] (departPos / departSpeed "random", cross.rou.xml:18-44);
 *   it is removed again when it conflicts with the env's ego pose — TRF:168-192: the vehicle inside the box
 *   -5 < x < v_ego + l_ego / 2 + l_veh / 2 + 2, |y| < 3 of the ego's frame, or the ego inside the same box of the
 *   vehicle's frame (shift_and_rotate_coordination, UTL:145-149; fp32 with the deterministic sin / cos);
 *   timer[r] = u_3 * period[r], emitted[r] = 0, sim_step = 0; phase0 = (random_phase && u > 0.5) ? 2 : 0 (TRF:158-161,
 *   task 'right'); v_light = training ? phase0 : 0 (TRF:222-223); cand_mode = route id or EB_VMODE_EMPTY.
 *   ego [n_env, 6]; cand_len [m] vehicle length per slot; phase0 uint8 [n_env]; the rest as in eb_traffic_flow_step. */
int eb_traffic_flow_reset(eb_handle h, int32_t n_env, int32_t per_route, const uint8_t* mask, const float* ego,
                          float* cand, uint8_t* active, float* timer, int32_t* emitted, int32_t* sim_step,
                          uint8_t* phase0, const float* lane, const float* period, const float* v_max,
                          const float* cand_len, float lane_len, int32_t random_phase, int32_t training,
                          uint64_t seed, uint64_t counter, uint8_t* cand_mode, uint8_t* v_light, void* stream);

/* ---- diagnostics (tests and profiling scripts; no reference counterpart) ----
 * Every tuning knob of the HIP library is a setting of the HANDLE (ABI 5: no environment variable is read by the library).
 * eb_debug_set_tile: force the rollout kernel's tile shape — 0: 2048-record tiles, 1: 1024, 2: 256; -1: by batch size
 * (every shape computes the same bits); the one-launch eb_env_step / eb_get_obs / eb_env_reset_pool take 64- / 32- / 16-env
 * tiles for 0 / 1 / 2 and pick by batch size otherwise.  eb_debug_set_env_waves: 4 / 8 waves per block of those kernels
 * (0: eight on grids of at most three blocks per CU with tiles of at most 32 envs, four otherwise).  eb_debug_set_tape_stepwise:
 * 1 = eb_rollout_tape[_f16] as `horizon` per-step launches, 0 = the one-launch tape kernel.  eb_debug_set_stage_paths: the tape /
 * gated kernels' LDS copy of the stride-10 path tables on (1) / off (0) / by grid size (-1).  eb_debug_set_scan_prefetch: 0 = the
 * closest-point search reads its index range one group of four table entries per loop trip (a memory round trip each, rounds 1-4's
 * form: A/B aid), 1 (default) = the first groups in one round trip; same comparisons, same bits.
 * eb_debug_set_rollout_sched (round 6; no prototype of ABI 5 changed): how the per-step rollout kernel's record waves use the memory
 * queue and the issue slots — rolling: 1 = three record loads in flight per lane, the next one requested when a record is done,
 * 0 = every record requested up front; by_progress: 1 = a wave's issue priority falls as it advances through its records, 0 = the
 * hardware's oldest-first; -1 (default) each = by the launch's grid against the device's CUs.  Same bits every way.
 * eb_debug_rollout_plan: what eb_rollout_step would launch for n_env envs on this handle — out4 = {tile shape (0: 2048 records,
 * 1: 1024, 2: 256), workgroups, rolling loads 0 / 1, priority by progress 0 / 1} (the oracle has no launch: EB_EINVAL).
 * eb_debug_set_trace: a device buffer of capacity_words int64 the kernels fill with wall-clock marks (NULL = off): the rollout
 * kernel writes rows of 8 words, one per wave — [n_blocks * waves per block][8] —, the one-launch env step rows of 16 —
 * [n_blocks * W][16] with W = 4 or 8 as above: size it for 8.  A launch whose marks would not fit capacity_words writes none.
 * The oracle accepts and ignores all of them. */
int eb_debug_set_tile(eb_handle h, int32_t variant);
int eb_debug_set_env_waves(eb_handle h, int32_t waves);
int eb_debug_set_tape_stepwise(eb_handle h, int32_t on);
int eb_debug_set_stage_paths(eb_handle h, int32_t mode);
int eb_debug_set_scan_prefetch(eb_handle h, int32_t on);
int eb_debug_set_rollout_sched(eb_handle h, int32_t rolling, int32_t by_progress);
int eb_debug_rollout_plan(eb_handle h, int32_t n_env, int32_t* out4);
/* eb_debug_check_grids (host-side self-check of the closest-point search's precomputed levels, DAM:702-715): the kernels do not scan the
 * stride-10 table; eb_set_paths precomputes, per cell of four nested grids (0.5 m / 4 m / 32 m / 256 m cells) and per path, the index
 * range(s) that hold the reference's first-minimum argmin for EVERY position of the cell.  This samples samples_per_cell positions in
 * every cell of every level (uniformly, against the edges, into the corners), takes the reference's argmin over the whole table with
 * the kernels' fp32 expression and counts the positions whose argmin the cell's ranges miss: *n_bad must come back 0.  The oracle has
 * no such levels (it IS the full scan): it reports 0 positions checked. */
int eb_debug_check_grids(eb_handle h, int32_t samples_per_cell, uint64_t seed, int64_t* n_checked, int64_t* n_bad);
int eb_debug_set_trace(eb_handle h, long long* device_buf, int64_t capacity_words);

/* ---- the policy in the loop: MLPNet + LoadPolicy.run_batch + the safety shield ----
 *
 * utils/model.py:18-43 `MLPNet`: Dense(obs_dim -> n_units, act) , (n_hidden - 1) x Dense(n_units -> n_units, act),
 * Dense(n_units -> out_dim, out_act), all fp32.  Kernels are Keras Dense kernels: HOST float [in, out] row-major,
 * biases HOST float [out]; layer 0 .. n_hidden - 1 are the hidden layers, layer n_hidden is the output layer.
 * Arithmetic contract (both libraries, bit for bit): y_j = act(chain_j) where chain_j starts at bias_j and takes
 * chain = fmaf(x_k, W[k][j], chain) for k = 0, 1, ... in order — the order v_mfma_f32_32x32x2_f32 accumulates in —
 * exp / tanh are the deterministic fp32 kernels of DESIGN.md.  The preprocessor of utils/preprocessor.py:116-123
 * ('scale': obs * obs_scale, one fp32 multiply) is applied to the input when a scale vector is set. */
#define EB_ACT_LINEAR 0
#define EB_ACT_RELU 1
#define EB_ACT_ELU 2
#define EB_ACT_TANH 3
#define EB_MLP_MAX_HIDDEN 8
#define EB_MLP_MAX_UNITS 512
typedef struct eb_mlp_config {
    int32_t abi_version; /* EB_ABI_VERSION */
    int32_t obs_dim;     /* input width (args.obs_dim, utils/policy.py:28) */
    int32_t n_hidden;    /* num_hidden_layers, 1..EB_MLP_MAX_HIDDEN */
    int32_t n_units;     /* num_hidden_units, 1..EB_MLP_MAX_UNITS */
    int32_t out_dim;     /* 2*act_dim for the policy net (mean | log_std, utils/policy.py:32), 1 for obj_v; <= 32 */
    int32_t hidden_act;  /* EB_ACT_* (args.hidden_activation) */
    int32_t out_act;     /* EB_ACT_* (policy_out_activation / 'relu' for obj_v, utils/policy.py:33-38) */
    int32_t device;      /* HIP device ordinal (ignored by the oracle) */
} eb_mlp_config;
typedef struct eb_mlp_s* eb_mlp;
int eb_mlp_create(const eb_mlp_config* cfg, eb_mlp* out);
int eb_mlp_destroy(eb_mlp m);
/* MLPNet weights of one Dense layer (Model.set_weights order: kernel, bias).  HOST pointers. */
int eb_mlp_set_layer(eb_mlp m, int32_t layer, const float* kernel, const float* bias);
/* Preprocessor obs_scale (HOST [obs_dim]); NULL = no preprocessing. */
int eb_mlp_set_obs_scale(eb_mlp m, const float* scale);
/* MLPNet.call on the preprocessed obs (utils/model.py:39-43): obs [n, obs_dim] -> out [n, out_dim]. */
int eb_mlp_forward(eb_mlp m, int32_t n, const float* obs, float* out, void* stream);
/* LoadPolicy.run_batch with a deterministic policy (utils/load_policy.py:53-57, utils/policy.py:85-92):
 * actions [n, out_dim/2] = action_range * tanh(mean), mean = the first half of the logits; action_range <= 0
 * stands for `action_range is None` (the mean itself). */
int eb_policy_run_batch(eb_mlp m, int32_t n, const float* obs, float action_range, float* actions, void* stream);

/* The model-predictive safety shield `is_safe` for a batch of start states (hier_decision.py:89-97: 5 steps,
 * veh2veh4real; multi_ego.py:187-197: 20 steps, real_punish_term): `steps` times action = run_batch(obs);
 * obs, penalties = rollout_out(action); punish += penalty.  obs_in [n_env, D] is not modified.
 * Workspace owned by the caller: obs_a, obs_b [n_env, D], actions [n_env, 2], out5 [5, n_env].
 * Outputs: punish [n_env] (0 + p_1 + ... + p_steps in fp32), safe [n_env] uint8 = !(punish > 0). */
#define EB_PENALTY_VEH2VEH4REAL 0
#define EB_PENALTY_REAL_PUNISH_TERM 1
int eb_shield_is_safe(eb_handle h, eb_mlp policy, int32_t n_env, const float* obs_in, const int32_t* ref_idx,
                      int32_t path_id, int32_t steps, int32_t penalty, float action_range, float* obs_a,
                      float* obs_b, float* actions, float* out5, float* punish, uint8_t* safe, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ENVBUILD_H */
