/* uavx.h — C ABI of libuavx.so: batched UAV collision-avoidance environments on MI355X (gfx950).
 *
 * The reference (dazchi/gym-uav-collision-avoidance) is pure Python and has NO FFI boundary; its
 * boundary is the gym.Env API of two classes.  Each entry point below therefore cites the Python
 * method it replaces (MUW = gym_uav_collision_avoidance/envs/multi_uav_world_2d.py,
 * AG = .../uav_agent.py, UW = .../uav_world_2d.py) and INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add to route those methods here.
 *
 * Conventions
 *   - every function returns 0 (UAVX_OK) or a negative uavx_status; nothing throws across the ABI;
 *     uavx_last_error(h) returns a message for the last failure on that handle.
 *   - all buffer arguments are DEVICE pointers owned by the caller (e.g. torch tensors' data_ptr())
 *     on the device the handle was created on; the library never allocates on the step path and
 *     never synchronises: work is enqueued on `stream` (a hipStream_t passed as void*, NULL = the
 *     null stream) and is complete when that stream reaches it.  The command and output buffers of a step may
 *     also be PINNED HOST memory (hipHostMalloc, a pinned torch tensor: mapped into the device's address space at
 *     the same address): the kernels then read / write it over the host link and the outputs are visible to the
 *     host once the stream has been synchronised -- what the single-env facades do for their ~100-byte steps
 *     (one launch, no copy calls); a batch belongs in HBM.
 *   - a handle is not thread-safe; use one handle per device / per caller thread.
 *   - E = number of envs, N = agents per env, agent slot a = e*N + i (agent index fastest).
 */
#ifndef UAVX_H
#define UAVX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVX_VERSION 3
#define UAVX_OBS_DIM 10     /* MUW:98-109 */
#define UAVX_UW_OBS_DIM 4   /* UW:107-112 */
#define UAVX_MAX_AGENTS 64  /* one wavefront holds a whole env: learners + scripted bodies <= 64 */
#define UAVX_MAX_LEVELS 16  /* curriculum levels per handle */
#define UAVX_BODY_DIM 6     /* floats of a scripted body's record: x, y, dx, dy, heading, legs (uavx_set_body_rule) */

#define UAVX_FLAG_DONE 1u      /* AG:19 */
#define UAVX_FLAG_COLLIDED 2u  /* AG:20 */
#define UAVX_FLAG_VEL_F32 4u   /* UW only: velocity is still reset()'s float32 draw (UW:122) */
#define UAVX_FLAG_INACTIVE 32u /* extension: learner parked by its curriculum level (n_active), see uavx_set_curriculum */

typedef enum {
    UAVX_OK = 0,
    UAVX_ERR_INVALID_ARG = -1,
    UAVX_ERR_HIP = -2,          /* a HIP runtime call failed; message has hipGetErrorString */
    UAVX_ERR_NO_DEVICE = -3,
    UAVX_ERR_UNSUPPORTED = -4,
    UAVX_ERR_ALLOC = -5
} uavx_status;

typedef enum { UAVX_F32 = 0, UAVX_F64 = 1 } uavx_dtype;

/* MultiUAVWorld2D.__init__ keyword arguments (MUW:13) + tau (MUW:26).
 * num_bodies (0 for the reference's world; the field was `reserved` = 0 in ABI version 1): scripted dynamic obstacles
 * per env, the "16 dynamic obstacles" of BASELINE.json configs[4].  The reference has no such entity (its only
 * obstacles are the other UAVs), so this is an EXTENSION with no reference semantics -- parity unpinned; defined below
 * at uavx_set_body_rule and restated by the test oracle.  num_agents stays the number of LEARNERS L: every caller
 * buffer (actions, obs, rew, done, state views) is [E*L...]; num_agents + num_bodies <= UAVX_MAX_AGENTS. */
typedef struct {
    double x_size, y_size;
    double max_speed, max_acceleration;
    double collider_radius, d_sense;
    double tau;
    int32_t num_agents;
    int32_t num_bodies;
} uavx_config;

/* UAVWorld2D.__init__ keyword arguments (UW:14) + tau (UW:26). */
typedef struct {
    double x_size, y_size, max_speed, max_acceleration, tau;
} uavx_uw_config;

/* State exchange view (uavx_get_state / uavx_set_state): plain SoA device arrays, any pointer may
 * be NULL to skip that component.  Mirrors the UAVAgent fields (AG:13-20) and the env counters
 * (MUW:166-168). */
typedef struct {
    float *loc;         /* [E*N*2] AG:13  location (float32, MUW:126) */
    double *vel;        /* [E*N*2] AG:14  velocity (float64) */
    float *tgt;         /* [E*N*2] AG:16  target_location */
    float *init_d;      /* [E*N]   AG:17  init_distance */
    float *prev_d;      /* [E*N]   AG:18  prev_distance */
    uint8_t *flags;     /* [E*N]   UAVX_FLAG_* */
    uint32_t *counters; /* [E*4]   steps, target_reach_count, collision_count, episode index */
} uavx_state_view;

/* float64-position episodes (see uavx_set_position_mode): the same fields as Python floats / float64 arrays. */
typedef struct {
    double *loc;     /* [E*N*2] */
    double *tgt;     /* [E*N*2] */
    double *init_d;  /* [E*N]   */
    double *prev_d;  /* [E*N]   */
} uavx_state_view_f64;

typedef enum { UAVX_POS_F32 = 0, UAVX_POS_F64 = 1 } uavx_position_mode;

typedef struct {
    float *loc;        /* [E*2] UW:121 */
    double *vel;       /* [E*2] UW:122 */
    float *tgt;        /* [E*2] UW:126 */
    float *init_d;     /* [E]   UW:129 */
    float *prev_d;     /* [E]   UW:130 */
    uint8_t *flags;    /* [E]   UAVX_FLAG_VEL_F32 */
    uint32_t *counters; /* [E*2] steps (UW:131), episode index */
} uavx_uw_state_view;

typedef struct uavx_handle uavx_handle;       /* E x MultiUAVWorld2D */
typedef struct uavx_uw_handle uavx_uw_handle; /* E x UAVWorld2D */

int uavx_version(void);
/* Identity of the sources the library was built from (hash over csrc/ and this header, set by the package's build; "" for
 * a hand-run make).  The Python loader rebuilds a library whose hash is not the one of the sources next to it. */
const char *uavx_build_info(void);
/* Hardware self test of the hand-written arithmetic the kernels rely on for bit-exactness: the 9-instruction
 * correctly rounded square root (csrc/uavx_device.hpp sqrt_rn) against the compiler's IEEE sqrtf on every float32
 * bit pattern.  *mismatches = number of differing results (0 expected).  Synchronous; about 10 ms. */
int uavx_selftest(int device, uint64_t *mismatches);
const char *uavx_strerror(int status);

/* ---------------------------------------------------------------------------------------------
 * MultiUAVWorld2D
 * ------------------------------------------------------------------------------------------- */

/* Replaces MultiUAVWorld2D.__init__ (MUW:13-58) for num_envs independent worlds on HIP device
 * `device`.  env_offset = global index of this handle's env 0 (multi-GPU sharding by env index:
 * Philox reset streams are keyed by global env id so results do not depend on the shard count). */
int uavx_create(const uavx_config *cfg, int64_t num_envs, int64_t env_offset, int device,
                uavx_handle **out);
int uavx_destroy(uavx_handle *h);
const char *uavx_last_error(const uavx_handle *h);
int64_t uavx_num_envs(const uavx_handle *h);
int uavx_num_agents(const uavx_handle *h);
/* Curriculum hook (no reference counterpart: the reference rebuilds the env object to change the world, e.g.
 * test_sac_multi_score.py:37 per agent count): replace the scalar world parameters of MUW:13-26 -- box size,
 * speed / acceleration limits, collider radius, d_sense, tau -- for every LATER launch on this handle.
 * num_agents must equal the handle's.  Agent state is untouched: agents outside a shrunken box terminate on
 * their next step exactly as MUW:224-229 would; the observation normalisers follow the new parameters.
 * Host-only call (no launch, no synchronisation); a hipGraph captured earlier keeps its old parameters. */
int uavx_set_config(uavx_handle *h, const uavx_config *cfg);

/* ---- extension of BASELINE.json configs[4]: scripted bodies + randomized-reset curriculum (no reference counterpart) ----
 * Bodies.  The B bodies of an env are slots L .. L+B-1 of its neighbour model: they are what the learners'
 * uavs_in_range (AG:44-64), collision tests (MUW:197-210) and neighbour observation features (MUW:75-95) see, exactly
 * like further UAVs, and they are stepped BEFORE the learners of the env's sequential loop (the world's scripted traffic
 * advances, then the UAVs move in MUW:181 order): a learner's collision test and its observation see the same, new, body
 * positions.  They read no action and produce no observation / reward / done.
 * A body travels in LEGS.  Its record is six float32 {x, y, dx, dy, heading, legs}: position, displacement per env step,
 * direction of travel (what a learner's MUW:82-85 feature shows) and the number of steps of the leg during which it moves.
 * A leg starts at reset (leg 0) and at every env step s > 0 with s % period == 0 (leg s / period): the body takes waypoint
 * W = Philox4x32-10, key = seed, counter (global env[31:0], env[47:32] | slot << 16, 0x80000000 | leg, episode), words 0,1
 * uniform over the env's box (float32), and from its position P (float32 arithmetic, no FMA):
 *     d = ||W - P||,  (dx, dy) = (W - P) * (speed*tau / d),  legs = floor(d / (speed*tau)),  heading = atan2(W - P)
 * (d == 0: no displacement, no legs).  In env step s it moves (x += dx, y += dy) iff s % period < legs: straight towards the
 * waypoint at `speed` m/s, stopping less than one step short of it until the next leg starts.  Between two waypoints the
 * kernel therefore reads 24 B and writes 8 B per body-step and spends two additions on it (round 2 kept {x, y, wx, wy} and
 * paid a norm, a division and an atan2 per body-step: 130 of the 1100 vector instructions of a wavefront).
 * reset draws the bodies' start points after the learners' in slot order under the same > 2R rejection rule (MUW:127-137).
 * period must be a power of two.  Defaults: speed 5 m/s, period 128, seed 0.  Host-only call: later launches see the new
 * rule (a running leg keeps the displacement it was started with). */
typedef struct {
    double speed;
    int32_t period;
    int32_t reserved;
    uint64_t seed;
} uavx_body_rule;
int uavx_set_body_rule(uavx_handle *h, const uavx_body_rule *rule);
int uavx_num_bodies(const uavx_handle *h);
/* Body records [E*B*UAVX_BODY_DIM] float32 {x, y, dx, dy, heading, legs}, device pointers. */
int uavx_get_bodies(uavx_handle *h, float *records, void *stream);
int uavx_set_bodies(uavx_handle *h, const float *records, void *stream);

/* Curriculum.  A level replaces x_size, y_size, collider_radius and d_sense of uavx_config for ONE env and says how many
 * of its learners (n_active in 1..L; learners >= n_active are parked: UAVX_FLAG_INACTIVE, never a neighbour, obs 0,
 * reward 0, done 1) and bodies (b_active in 0..B) take part.  An env takes its level when it is (re-)initialised -- by
 * uavx_reset or by the auto-reset of uavx_step_ex -- and keeps it for the whole episode:
 *   level_lo >= 0: drawn uniformly in [level_lo, level_hi] by Philox (key = the reset's seed, counter (global env[31:0],
 *                  env[47:32] | 0xFFFF << 16, 0, episode), word 0): the randomized-reset curriculum; the caller moves the
 *                  window as training progresses;
 *   level_lo <  0: the level assigned to the env with uavx_set_env_levels (default 0).
 * The reference's analogue is building a new env object per world (test_sac_multi_score.py:31-37).  n_levels = 0 removes
 * the table (every env back on uavx_config; follow it with uavx_reset: learners a level had parked stay parked until their
 * env is re-initialised).  Enqueues one tiny launch on `stream`.
 * hipGraphs: the level table lives in device memory and is rewritten in place, while the window [level_lo, level_hi], the
 * number of levels and the version that invalidates parked auto-reset layouts travel in the kernel arguments of each launch.
 * A graph captured BEFORE this call therefore replays with the old window and version against the NEW table: its envs draw
 * old-window levels and may still consume layouts parked for the old table (their positions lie in the old levels' boxes).
 * Nothing faults, but it is not the curriculum that was asked for: re-capture graphs after uavx_set_curriculum (as after
 * uavx_set_config and uavx_set_body_rule). */
typedef struct {
    double x_size, y_size, collider_radius, d_sense;
    int32_t n_active, b_active;
} uavx_level;
int uavx_set_curriculum(uavx_handle *h, const uavx_level *levels, int32_t n_levels, int32_t level_lo, int32_t level_hi,
                        void *stream);
int uavx_set_env_levels(uavx_handle *h, const uint8_t *levels, void *stream); /* [E] device: level at the env's NEXT reset */
int uavx_get_env_levels(uavx_handle *h, uint8_t *levels, void *stream);       /* [E] device: level in force */

/* Replaces MultiUAVWorld2D.reset (MUW:116-175) for the envs with mask[e] != 0 (mask == NULL: all).
 * Start/target points are drawn by Philox4x32-10 keyed with `seed`, counter (global env, draw,
 * episode) under the reference's rejection rules (MUW:127-153); velocities, flags and the three
 * counters are cleared (MUW:118-123,166-168) and the env's episode index is incremented.
 * obs (may be NULL) receives the observations of ALL envs, [E*N*10] float32 (MUW:170-172). */
int uavx_reset(uavx_handle *h, const uint8_t *mask, uint64_t seed, float *obs, void *stream);

/* Replaces MultiUAVWorld2D.step (MUW:177-241) incl. UAVAgent.step / finish / uavs_in_range
 * (AG:23-64) and _get_obs (MUW:60-109) for all E envs in one launch.
 *   actions  [E*N*2] float32 or float64 (action_dtype), the velocity commands n_action[i]
 *   evaluate MUW:177 `evaluate` flag (out-of-box no longer terminates, MUW:225)
 *   obs      [E*N*10] float32    rew [E*N] float32    done [E*N] uint8 (0/1) */
int uavx_step(uavx_handle *h, const void *actions, int action_dtype, int evaluate, float *obs,
              float *rew, uint8_t *done, void *stream);

/* K consecutive steps from an action tape [K][E*N*2] in ONE launch (state stays in registers):
 * obs/rew/done receive tapes [K][...] when tape_out != 0, else only the last step's values.
 * Same semantics as K calls of uavx_step. */
int uavx_step_k(uavx_handle *h, int k, const void *actions, int action_dtype, int evaluate,
                int tape_out, float *obs, float *rew, uint8_t *done, void *stream);

/* ---- the trainer loop's work around env.step, fused into the step launch ----
 * The reference's callers (a) convert a policy output a in [-1,1]^2 to a velocity command
 * (test_sac_multi.py:77-80), (b) add up rewards and read the counters at episode end
 * (test_sac_multi.py:106,157,164-165) and (c) reset the env on dones[0] (training, :112), all(dones)
 * (evaluation, :116,161) or a step cap (:17,67).  uavx_step_ex does all three on the device. */
typedef enum {
    UAVX_ACTION_CARTESIAN = 0, /* actions are velocity commands (what env.step takes) */
    UAVX_ACTION_POLAR = 1      /* actions are a in [-1,1]^2: v = (a0/2+0.5)*||action_space.high||, theta = a1*pi,
                                  command = (v cos theta, v sin theta), float32 arithmetic */
} uavx_action_mode;
typedef enum {
    UAVX_RESET_NEVER = 0,       /* like the reference: the caller resets */
    UAVX_RESET_AGENT0_DONE = 1, /* test_sac_multi.py:112 */
    UAVX_RESET_ALL_DONE = 2     /* test_sac_multi.py:116,161 */
} uavx_reset_policy;

typedef enum { UAVX_FLAGS_ARRAYS = 0, UAVX_FLAGS_IN_DONE = 1 } uavx_flags_mode;

typedef struct {
    const void *actions;   /* [E*N*2] */
    int32_t action_dtype;  /* uavx_dtype */
    int32_t action_mode;   /* uavx_action_mode */
    int32_t evaluate;      /* MUW:177 */
    int32_t reset_policy;  /* uavx_reset_policy */
    uint32_t step_cap;     /* 0 = none; an env whose `steps` reaches the cap ends its episode too */
    int32_t track_returns; /* accumulate per-env episode return / evaluation score */
    uint64_t seed;         /* Philox key for the auto-resets */
    float *obs;            /* [E*N*10] */
    float *rew;            /* [E*N] */
    uint8_t *done;         /* [E*N] */
    uint8_t *reset_mask;   /* [E] or NULL: 1 where this call re-initialised the env instead of stepping it */
    /* ABI version 2 (appended): the episode-end signal AT the transition a learner stores.  ended[e] = 1 where this
     * call's step ended the env's episode (the env is re-initialised by the next call); truncated[e] = 1 where that end
     * came from the step cap alone (test_sac_multi.py:17,67) and no terminal condition of the reset policy held
     * (:112,:116) -- a time-limit truncation, to be bootstrapped through, as opposed to a terminal state.  Either may
     * be NULL.  A call that re-initialises an env reports 0 / 0 for it. */
    uint8_t *ended;        /* [E] or NULL */
    uint8_t *truncated;    /* [E] or NULL */
    /* ABI version 3 (appended).  UAVX_FLAGS_IN_DONE: the three per-env flags are NOT written to reset_mask / ended /
     * truncated (which are ignored and may be NULL) but ride in the done byte of the env's agent 0:
     *     done[e*N + 0] = done | reset_mask << 1 | ended << 2 | truncated << 3        (the other agents' bytes stay 0 / 1)
     * so that a launch has no one-byte-per-env partial line writes (0.3 us of a 7 us launch at 65 536 x 4); readers take
     * done & 1.  UAVX_FLAGS_ARRAYS (0, the default: a zero-filled struct behaves like version 2) keeps the three arrays. */
    int32_t flags_mode;    /* uavx_flags_mode */
    int32_t reserved;
} uavx_step_args;

/* Auto-reset is "next-step": an env whose episode ended at call t keeps its terminal observation in
 * that call's outputs (so s' of the last transition is the true successor) and is re-initialised by
 * call t+1 INSTEAD of being stepped: that call returns the fresh observation with reward 0, done 0
 * and reset_mask 1 for the env, and its action is ignored.  Ended episodes are folded into the
 * per-env episode statistics (also by uavx_reset). */
int uavx_step_ex(uavx_handle *h, const uavx_step_args *args, void *stream);

/* Layouts drawn ahead of time (default: every = 64 * min(floor(64 / slots), 8) / envs-per-workgroup -- 32 at 4 UAVs, 64 at 8,
 * 16 for 8 learners + 16 bodies: staging workgroups for num_envs / 128 layouts per launch, what episodes of 128 steps consume).  The start / target layout of an env's next episode depends only on (seed,
 * global env id, episode index, level rule), so every uavx_step_ex launch with an auto-reset policy or a step cap carries
 * ceil(G / every) extra workgroups beside its G env-workgroups (in front of them while those leave wavefront slots free,
 * behind them when they fill the device on their own).  They step nothing.  Each alternates between two short jobs: it looks
 * at one window of envs (a lane per env; every window comes round every few launches), finds the envs whose parked layouts
 * -- an env keeps TWO, for its next episode and the one after -- are missing, consumed or drawn for another seed / world /
 * level, and notes the first few down; in the next launch it draws those into a staging area, one lane per slot of the
 * neighbour model (after checking against the env's record that they are still wanted).  A step workgroup then
 * re-initialises an env with 16-byte copies instead of running the serial accept / reject chain of MUW:127-153 on one
 * wavefront while the rest of the chip waits for it (with 16 scripted bodies that chain is 8-19 us on its own;
 * profiles/r03_ab_notes.md).  A layout that is not there when it is needed -- first use, a new seed / world / curriculum
 * window (all parked layouts are then worked off again over the following launches), two episode ends of one env within a
 * few launches -- is drawn in the step workgroup as before: results are identical either way.  every = 0 switches staging
 * off.  No second kernel, stream or event is involved: one launch per call, on `stream`; a captured graph behaves like eager
 * calls (nothing on the host counts launches). */
int uavx_set_prefetch(uavx_handle *h, int every);

/* Per-env statistics over the episodes ended so far (by auto-reset or uavx_reset):
 * counts  [E*4] uint32 = episodes, sum of steps, sum of target_reach_count, sum of collision_count
 * returns [E*2] float32 = sum of agent-0 returns (test_sac_multi.py:106 `score`),
 *                         sum of sum_i r_i*(1-done_i) (test_sac_multi.py:157 `total_score`)
 * SR/CR of test_sac_multi.py:174-175 = sum(reach or coll) / (N * sum(episodes)).  Either may be NULL. */
int uavx_get_episode_stats(uavx_handle *h, uint32_t *counts, float *returns, void *stream);
int uavx_clear_episode_stats(uavx_handle *h, void *stream);

/* Replaces MultiUAVWorld2D._get_obs for every agent (MUW:60-109): obs [E*N*10] float32. */
int uavx_observe(uavx_handle *h, float *obs, void *stream);

/* Read / overwrite the agents' fields and env counters (the reference's callers poke
 * env.agent_list[i].location etc. directly, test_sac_multi_plot_trajectory.py:43-49). */
int uavx_get_state(uavx_handle *h, const uavx_state_view *dst, void *stream);
int uavx_set_state(uavx_handle *h, const uavx_state_view *src, void *stream);

/* Position dtype of the running episodes.  After reset() the reference's agent.location / target_location are
 * float32 arrays (MUW:126,131,144) -- UAVX_POS_F32, the default and the fast path.  reset(circular=True)
 * (MUW:157-163) and callers that assign their own arrays (test_sac_multi_plot_trajectory.py:43-49) install
 * float64 arrays instead, and every position expression of that episode is then float64 (AG:29,33,51; MUW:190-
 * 194,203,207,218).  UAVX_POS_F64 restates that episode type for ALL envs of the handle: positions, targets,
 * init / prev distances are held as doubles and stepped by a one-thread-per-env kernel (evaluation / plotting
 * scenario, not tuned); masks, positions, velocities and counters stay bit-exact against the reference.
 *   uavx_set_position_mode  converts the stored state (F32->F64 exact; F64->F32 rounds) and selects the
 *                           arithmetic of later uavx_step / uavx_step_ex / uavx_observe calls.  The first switch
 *                           to F64 allocates 48 B per agent (the only allocation after uavx_create).
 *   uavx_set_state_f64      overwrites any subset of the float64 fields; enters F64 mode if needed.
 *   uavx_get_state_f64      F64 mode only (UAVX_ERR_UNSUPPORTED otherwise).
 * In F64 mode uavx_get_state / uavx_set_state exchange rounded / widened float32 views; uavx_reset(mask=NULL)
 * returns the handle to F32 like the reference's reset(); uavx_reset with a mask, uavx_step_k with k > 1 and
 * uavx_step_ex with an auto-reset policy or step cap return UAVX_ERR_UNSUPPORTED. */
int uavx_set_position_mode(uavx_handle *h, int mode, void *stream);
int uavx_get_position_mode(const uavx_handle *h);
int uavx_set_state_f64(uavx_handle *h, const uavx_state_view_f64 *src, void *stream);
int uavx_get_state_f64(uavx_handle *h, const uavx_state_view_f64 *dst, void *stream);

/* NaN / Inf tripwire.  np.clip lets a NaN command through (AG:26-27), and an agent whose velocity is NaN stays poisoned
 * for the rest of its episode; the reference's trainers notice by printing the agents' state and quitting
 * (test_ddpg_multi.py:114-130).  counts [E] uint32 = agent-steps of the env's RUNNING episode whose reward came out
 * non-finite (cleared with the MUW:166-168 counters at every reset): `counts.any()` is the batched form of that tripwire,
 * and the env index says whose command to look at.  One compare per agent-step on the step path, an atomic on the event. */
int uavx_get_nonfinite(uavx_handle *h, uint32_t *counts, void *stream);

/* Exact snapshot / restore of everything a handle holds (the reference never checkpoints its env -- only the agents,
 * sac.py:101-139 -- so a 65 536-env training run could not be resumed where it stopped): agent state, counters, running and
 * ended-episode statistics, body records, levels, the parked layouts of the auto-reset and the host-side parameters later
 * launches take by value (world, body rule, curriculum, staging cadence).  After uavx_load the handle continues bit for bit
 * as the saved one would have (same outputs, same auto-reset draws, same uavx_get_episode_stats).
 *   uavx_snapshot_bytes  size of the DEVICE buffer a snapshot needs (fixed for a handle).
 *   uavx_save            enqueues the copy on `stream` (device to device; 256-byte aligned buffer); no synchronisation.
 *   uavx_load            the handle must have the shape (envs, agents, bodies) the snapshot was taken from; reads the
 *                        snapshot's header on the host first, so it WAITS for `stream` once, then enqueues the copy.
 *                        Every header field that becomes a copy length, a kernel argument or a table index is checked
 *                        before anything is copied (a truncated or damaged snapshot: UAVX_ERR_INVALID_ARG, handle untouched);
 *                        the buffer must hold uavx_snapshot_bytes() bytes. */
int64_t uavx_snapshot_bytes(const uavx_handle *h);
int uavx_save(uavx_handle *h, void *snapshot, void *stream);
int uavx_load(uavx_handle *h, const void *snapshot, void *stream);

/* Episode metrics of MUW:166-168,209,221,238 as the trainers read them before reset
 * (test_sac_multi.py:164-165): counters [E*4] uint32 = steps, target_reach_count,
 * collision_count, episode index. */
int uavx_get_metrics(uavx_handle *h, uint32_t *counters, void *stream);

/* ---------------------------------------------------------------------------------------------
 * UAVWorld2D
 * ------------------------------------------------------------------------------------------- */
int uavx_uw_create(const uavx_uw_config *cfg, int64_t num_envs, int64_t env_offset, int device,
                   uavx_uw_handle **out);                                       /* UW:14-75 */
int uavx_uw_destroy(uavx_uw_handle *h);
const char *uavx_uw_last_error(const uavx_uw_handle *h);
/* UW:119-135; obs [E*4] float32 of all envs (may be NULL). */
int uavx_uw_reset(uavx_uw_handle *h, const uint8_t *mask, uint64_t seed, float *obs, void *stream);
/* UW:137-173; actions [E*2]; obs [E*4] f32, rew [E] f32, done [E] u8, info_distance [E] f32
 * (UW:114-117, may be NULL). */
int uavx_uw_step(uavx_uw_handle *h, const void *actions, int action_dtype, float *obs, float *rew,
                 uint8_t *done, float *info_distance, void *stream);
int uavx_uw_observe(uavx_uw_handle *h, float *obs, void *stream);               /* UW:77-112 */

/* UAVWorld2D counterpart of uavx_step_ex: what test_sac.py:62-109 does around env.step — action
 * conversion v = (a0/2+0.5)*action_space.high[0], theta = a1*pi (:77-80, float32 on the device), score
 * accumulation (:98) and reset when done (:106-109) or at a step cap (:17).  Same next-step auto-reset
 * contract as uavx_step_ex (terminal observation stays in the ending call's outputs). */
typedef struct {
    const void *actions;   /* [E*2] */
    int32_t action_dtype;  /* uavx_dtype */
    int32_t action_mode;   /* uavx_action_mode */
    int32_t auto_reset;    /* 0: the caller resets; 1: reset an env after a step that returned done */
    uint32_t step_cap;     /* 0 = none */
    int32_t track_returns;
    int32_t reserved;
    uint64_t seed;
    float *obs;            /* [E*4] */
    float *rew;            /* [E] */
    uint8_t *done;         /* [E] */
    float *info_distance;  /* [E] or NULL */
    uint8_t *reset_mask;   /* [E] or NULL */
    /* ABI version 2 (appended), as in uavx_step_args: ended[e] = 1 where this call's step ended the episode (done with
     * auto_reset, or the step cap), truncated[e] = 1 where the step cap alone did (done was 0).  Either may be NULL. */
    uint8_t *ended;        /* [E] or NULL */
    uint8_t *truncated;    /* [E] or NULL */
} uavx_uw_step_args;
int uavx_uw_step_ex(uavx_uw_handle *h, const uavx_uw_step_args *args, void *stream);
/* counts [E*4] uint32 = episodes, sum of steps, episodes that ended at the target (UW:159), reserved;
 * returns [E] float32 = sum of episode returns (test_sac.py:98 `score`).  Either may be NULL. */
int uavx_uw_get_episode_stats(uavx_uw_handle *h, uint32_t *counts, float *returns, void *stream);
int uavx_uw_clear_episode_stats(uavx_uw_handle *h, void *stream);
int uavx_uw_get_state(uavx_uw_handle *h, const uavx_uw_state_view *dst, void *stream);
int uavx_uw_set_state(uavx_uw_handle *h, const uavx_uw_state_view *src, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UAVX_H */
