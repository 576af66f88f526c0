/* uavx_oracle.h — CPU restatement (TEST INFRASTRUCTURE ONLY) of the reference env step/reset path.
 *
 * This is the parity oracle: a scalar C restatement of
 *   MUW = /root/reference/gym_uav_collision_avoidance/envs/multi_uav_world_2d.py
 *   AG  = /root/reference/gym_uav_collision_avoidance/envs/uav_agent.py
 *   UW  = /root/reference/gym_uav_collision_avoidance/envs/uav_world_2d.py
 * following the reference's op order and dtypes as evaluated by numpy 2.2.6 (NEP 50) + glibc libm.
 * PARITY PINNED: bit-exact against fixtures generated from the reference itself
 * (tests/golden/make_golden.py writes tests/golden/ npz files; tests/test_oracle_golden.py) and against the live
 * reference on fresh seeds wherever it is mounted (tests/test_reference_live.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (gym_uav_collision_avoidance_amd) never links, imports or falls back to it.
 */
#ifndef UAVX_ORACLE_H
#define UAVX_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVO_OBS_DIM 10
#define UAVO_UW_OBS_DIM 4
#define UAVO_FLAG_DONE 1u
#define UAVO_FLAG_COLLIDED 2u

/* MultiUAVWorld2D ctor arguments (MUW:13) + tau (MUW:26). */
typedef struct {
    double x_size, y_size;
    double max_speed, max_acceleration;
    double collider_radius, d_sense;
    double tau;
    int32_t num_agents;
    int32_t _pad;
} uavo_config;

/* State of E independent worlds, N agents each.  Positions are held as doubles; when f64pos[e]==0
 * (the normal case, MUW:126,131,144 astype(float32)) every value is float32-representable and all
 * position arithmetic is done in float32 exactly like numpy does; when f64pos[e]==1 (after
 * reset(circular=True), MUW:157-163) arithmetic is float64.  Index = (e*N + i)*2 + axis. */
typedef struct {
    int64_t num_envs;
    int32_t num_agents;
    int32_t _pad;
    double *loc;      /* [E*N*2] */
    double *vel;      /* [E*N*2]  AG:14 velocity (float64) */
    double *tgt;      /* [E*N*2] */
    double *init_d;   /* [E*N] */
    double *prev_d;   /* [E*N] */
    uint8_t *flags;   /* [E*N]  bit0 done (AG:19), bit1 collided (AG:20) */
    uint32_t *counters; /* [E*4] steps, target_reach_count, collision_count, episode index */
    uint8_t *f64pos;  /* [E] */
} uavo_state;

/* MT19937 with numpy's legacy seeding, so that uavo_reset_mt() consumes the same stream as the
 * reference's np.random.uniform calls after np.random.seed(seed). */
typedef struct { uint32_t mt[624]; int32_t idx; } uavo_mt;
void uavo_mt_seed(uavo_mt *g, uint32_t seed);
double uavo_mt_double(uavo_mt *g);

/* Philox4x32-10 (the generator the device reset kernel uses). out[4]. */
void uavo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* MUW:116-175.  One env, draws from the MT stream.  circular!=0 applies MUW:157-163. */
void uavo_reset_mt(const uavo_config *cfg, uavo_state *st, int64_t env, uavo_mt *g, int circular);
/* Same rejection rules, draws from Philox keyed by (seed) with counter (global_env, episode, draw):
 * restatement of the device reset kernel's generator for bit-exact comparison with it.
 * mask==NULL resets all envs; else only envs with mask[e]!=0.  env_offset = first global env id. */
void uavo_reset_philox(const uavo_config *cfg, uavo_state *st, const uint8_t *mask, uint64_t seed,
                       int64_t env_offset, int nthreads);

/* MUW:60-109 for every agent of every env: obs[(e*N+i)*10 + k] (float64 like the reference). */
void uavo_observe(const uavo_config *cfg, const uavo_state *st, double *obs, int nthreads);

/* MUW:177-241 on every env.  actions [(e*N+i)*2] float64 (float32 callers widen exactly).
 * Outputs: obs [E*N*10] f64, reward [E*N] f64, done [E*N] u8. */
void uavo_step(const uavo_config *cfg, uavo_state *st, const double *actions, int evaluate,
               double *obs, double *reward, uint8_t *done, int nthreads);

/* ---- trainer-loop fusions of uavx_step_ex (NEW semantics of the build, not reference behaviour: the
 * env step inside is the pinned uavo_step; conversion / auto-reset / statistics restate what the
 * reference's trainer scripts do around env.step, test_sac_multi.py:67-119,132-183) ---- */
typedef struct {
    int32_t action_mode;   /* 0 cartesian, 1 polar (test_sac_multi.py:77-80 in float32) */
    int32_t reset_policy;  /* 0 never, 1 agent-0 done (:112), 2 all done (:116,161) */
    int32_t track_returns;
    uint32_t step_cap;     /* :17,67 */
    uint64_t seed;
    int64_t env_offset;
} uavo_step_opts;

typedef struct {
    uint8_t *pending;      /* [E] */
    float *ep_run;         /* [E*2] running {agent-0 return, sum_i r_i (1-done_i)} */
    uint32_t *fin_counts;  /* [E*4] episodes, steps, reach, coll over ended episodes */
    float *fin_returns;    /* [E*2] */
} uavo_episode_state;

/* float32 restatement of the device's polar -> velocity-command conversion (same fmaf sequence). */
void uavo_polar_to_command(float a0, float a1, float vmax_norm, double out[2]);
/* next-step auto-reset + statistics; reset_mask may be NULL. */
void uavo_step_ex(const uavo_config *cfg, uavo_state *st, uavo_episode_state *ep, const uavo_step_opts *opt,
                  const double *actions, int evaluate, double *obs, double *reward, uint8_t *done,
                  uint8_t *reset_mask, int nthreads);
/* the statistics side of an explicit reset (uavx_reset folds the running episode the same way) */
void uavo_fold_episode(uavo_state *st, uavo_episode_state *ep, int64_t env);

/* ---- world extension of BASELINE.json configs[4] (NO reference counterpart: "parity unpinned" by the reference;
 * these functions restate the build's own definition in include/uavx.h so the HIP path can be checked bit for bit) ----
 *  - scripted bodies: B non-learning records per env that sit in the neighbour model as agents L .. L+B-1 (they are
 *    what uavs_in_range / the collision tests of the L learners see, AG:44-64, MUW:197-210), stepped BEFORE the learners
 *    of the env's sequential loop (MUW:181) by a waypoint rule keyed by Philox: legs of constant displacement per step;
 *  - curriculum levels: per-env box size / d_sense / collider radius / number of active learners and bodies, chosen
 *    when the env is reset (randomized-reset curriculum);
 *  - step_ex reports which envs ended at this call and which of those were cut by the step cap (truncated). */
#define UAVO_FLAG_INACTIVE 32u /* learner parked by its level's n_active: not stepped, never a neighbour */
#define UAVO_MAX_LEVELS 16
#define UAVO_BODY_DIM 6
typedef struct {
    double x_size, y_size, collider_radius, d_sense;
    int32_t n_active;   /* learners 0 .. n_active-1 take part (1 .. L) */
    int32_t b_active;   /* bodies 0 .. b_active-1 take part (0 .. B) */
} uavo_level;

typedef struct {
    int32_t num_bodies;     /* B */
    int32_t body_period;    /* a body draws a new waypoint every body_period env steps */
    double body_speed;      /* cruise speed, m/s */
    uint64_t body_seed;     /* Philox key of the waypoint streams */
    int32_t n_levels;       /* 0: one level made of cfg itself, all learners / bodies active */
    int32_t level_lo, level_hi; /* a reset draws the env's level uniformly in [lo, hi]; lo < 0: take next_level[e] */
    int32_t _pad;
    const uavo_level *levels;
} uavo_ext;

typedef struct {
    float *body;         /* [E*B*6] x, y, displacement per step x, y, heading, steps of the leg that move (float32) */
    uint8_t *level;      /* [E] level in force since the env's last reset */
    uint8_t *next_level; /* [E] level an explicit assignment asked for (used when level_lo < 0) */
} uavo_ext_state;

/* ext == NULL or xs == NULL gives exactly uavo_reset_philox / uavo_observe / uavo_step / uavo_step_ex.
 * ended / truncated ([E], may be NULL): 1 where this call ended the env's episode (it will be re-initialised by the next
 * call) / where that end came from the step cap alone (no terminal condition of the reset policy held). */
void uavo_reset_philox_x(const uavo_config *cfg, const uavo_ext *ext, uavo_state *st, uavo_ext_state *xs,
                         const uint8_t *mask, uint64_t seed, int64_t env_offset, int nthreads);
void uavo_observe_x(const uavo_config *cfg, const uavo_ext *ext, const uavo_state *st, const uavo_ext_state *xs,
                    double *obs, int nthreads);
void uavo_step_x(const uavo_config *cfg, const uavo_ext *ext, uavo_state *st, uavo_ext_state *xs, const double *actions,
                 int evaluate, int64_t env_offset, double *obs, double *reward, uint8_t *done, int nthreads);
void uavo_step_ex_x(const uavo_config *cfg, const uavo_ext *ext, uavo_state *st, uavo_ext_state *xs,
                    uavo_episode_state *ep, const uavo_step_opts *opt, const double *actions, int evaluate, double *obs,
                    double *reward, uint8_t *done, uint8_t *reset_mask, uint8_t *ended, uint8_t *truncated, int nthreads);

/* ---- UAVWorld2D (UW) ---- */
typedef struct {
    double x_size, y_size, max_speed, max_acceleration, tau;
} uavo_uw_config;

typedef struct {
    int64_t num_envs;
    double *loc;     /* [E*2] float32-representable (UW:121) */
    double *vel;     /* [E*2] */
    double *tgt;     /* [E*2] */
    double *init_d;  /* [E] */
    double *prev_d;  /* [E] */
    uint32_t *steps; /* [E] */
    uint32_t *episode; /* [E] reset count, Philox counter word 3 */
    uint8_t *vel_f32; /* [E] 1 while velocity is still the float32 array drawn by reset (UW:122) */
} uavo_uw_state;

/* uavx_uw_step_ex restatement (test_sac.py:62-109 around env.step): NEW semantics, see uavo_step_ex. */
typedef struct {
    uint8_t *pending;      /* [E] */
    uint8_t *reached;      /* [E] the last step ended at the target (UW:159) */
    float *ep_return;      /* [E] */
    uint32_t *fin_counts;  /* [E*4] episodes, steps, episodes ended at the target, 0 */
    float *fin_return;     /* [E] */
} uavo_uw_episode_state;
void uavo_uw_fold_episode(uavo_uw_state *st, uavo_uw_episode_state *ep, int64_t env);
void uavo_uw_step_ex(const uavo_uw_config *cfg, uavo_uw_state *st, uavo_uw_episode_state *ep, int action_mode,
                     int auto_reset, uint32_t step_cap, int track_returns, uint64_t seed, int64_t env_offset,
                     const double *actions, int action_is_f32, double *obs, double *reward, uint8_t *done,
                     double *info_distance, uint8_t *reset_mask, int nthreads);

void uavo_uw_reset_mt(const uavo_uw_config *cfg, uavo_uw_state *st, int64_t env, uavo_mt *g);
void uavo_uw_reset_philox(const uavo_uw_config *cfg, uavo_uw_state *st, const uint8_t *mask,
                          uint64_t seed, int64_t env_offset, int nthreads);
void uavo_uw_observe(const uavo_uw_config *cfg, const uavo_uw_state *st, double *obs, int nthreads);
/* actions [E*2] f64; action_is_f32: the caller's array dtype was float32 (matters only on the first
 * step after reset, see UW:142 under NEP 50).  reward f64 (value of the np.float32 the reference
 * returns), done u8, info_distance [E] (UW:114-117). */
void uavo_uw_step(const uavo_uw_config *cfg, uavo_uw_state *st, const double *actions,
                  int action_is_f32, double *obs, double *reward, uint8_t *done,
                  double *info_distance, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
