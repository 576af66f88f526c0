// uavx_uw.hip — E x UAVWorld2D (single UAV, 4-dim observation) on one MI355X.
// Reference: UW = gym_uav_collision_avoidance/envs/uav_world_2d.py of dazchi/gym-uav-collision-avoidance.
//
// One lane per env (there is no cross-agent work): every load/store is lane-contiguous.
// HBM layout (lean: 97 B of real traffic per env-step against the 93 B algorithmic figure of SURVEY.md 8d):
//   pos  float2[E]  {x, y}                     read+write    8 B
//   vel  double2[E] {vx, vy}                   read+write   16 B
//   goal {tx, ty, init_d, flags}[E]            read         16 B   (one dwordx4; the flags word is stored only when it
//                                                                   changes: first step of an episode, arrival)
//   prev_distance (UW:130,172) is NOT stored: it always equals ||target - location|| (reset sets it to init_distance,
//   every step to the new distance), so it is recomputed from the loaded position; a value a caller pokes that breaks
//   the identity lives in prev_ovr[E] behind a flag bit until the next step.
//   steps (UW:131,170) = wave_steps[wavefront] - rec.x: a step launch bumps ONE counter per wavefront;
//   rec uint4[E] {steps base, episode | pending << 31, running return, -}: reset / step_ex / state exchange only.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <string>

#include "../../include/uavx.h"
#include "uavx_device.hpp"

namespace uavx {

struct UwGoal { float tx, ty, init_d; uint32_t flags; };

struct UwParams {
    double tau, rtau, amax, vmax;  // rtau = RN(1/tau) for div_tau()
    double lox, loy, hix, hiy;
    float lo_x, lo_y, hi_x, hi_y;  // float32 forms of the box test (exact: smallest float32 >= lox, ...)
    int recip_ok;
    float tau_f;       // float32(tau): UW:142 divides a float32 array by the python float
    float high0;       // float32(action_space.high[0]) = max_speed (test_sac.py:77)
    float inv_vmax;    // 1/max_speed[0]            UW:88
    float inv_diag;    // 1/‖(x_size,y_size)‖       UW:17,97
    int64_t E, env_offset;
    float2 *pos;
    double2 *vel;
    UwGoal *goal;
    float *prev_ovr;
    uint32_t *wave_steps;
    uint4 *rec;           // [E] {steps base, episode index | pending bit 31, running episode return (float bits), unused}
    // episode bookkeeping (uavx_uw_step_ex / uavx_uw_reset)
    uint4 *fin_counts;    // [E] {episodes, steps, episodes ended at the target, -}
    float *fin_return;    // [E] sum of ended episodes' returns
};

struct UwExtra {
    int action_mode, auto_reset, track_returns;
    uint32_t step_cap, seed_lo, seed_hi;
    uint8_t *reset_mask, *ended, *truncated;
};

// UW:88-97 in float32
__device__ __forceinline__ float4 uw_obs(const UwParams &p, float speed, float theta, float dist_t, float dth) {
    return make_float4(speed * p.inv_vmax, theta * kInvPi, dist_t * p.inv_diag, dth * kInvPi);
}

constexpr uint32_t kUwReached = 8u;   // internal flag bit: the last step ended at the target (d < 0.5, UW:159)
constexpr uint32_t kUwPrevOvr = 16u;  // prev_distance is prev_ovr[e], not ||target - location||
constexpr uint32_t kUwPending = 0x80000000u;  // rec.y bit 31: episode ended, re-initialise at the next step_ex

struct UwRegs {
    float x, y, prev_d, tx, ty, init_d;
    uint32_t flags;
    double vx, vy;
};

// UW:137-173 for one env held in registers.  act_f32: the command came as float32 (first-step quirk of UW:142).
__device__ __forceinline__ void uw_step_env(const UwParams &p, UwRegs &s, double ax, double ay, bool act_f32,
                                            float4 &obs, float &rew, uint32_t &done, float &dist) {
    const bool f32_first = act_f32 && (s.flags & UAVX_FLAG_VEL_F32) != 0;  // float32 action - float32 velocity, / float32(tau)
    double qx, qy;
    if (f32_first) {
        qx = (double)(((float)ax - (float)s.vx) / p.tau_f);
        qy = (double)(((float)ay - (float)s.vy) / p.tau_f);
    } else {
        qx = div_tau(ax - s.vx, p.tau, p.rtau, p.recip_ok != 0);
        qy = div_tau(ay - s.vy, p.tau, p.rtau, p.recip_ok != 0);
    }
    s.vx = clip64(s.vx + clip64(qx, -p.amax, p.amax) * p.tau, -p.vmax, p.vmax);  // UW:142-144
    s.vy = clip64(s.vy + clip64(qy, -p.amax, p.amax) * p.tau, -p.vmax, p.vmax);
    s.x = (float)((double)s.x + s.vx * p.tau);                                   // UW:145-146
    s.y = (float)((double)s.y + s.vy * p.tau);
    const bool oob = !(s.x >= p.lo_x && s.x <= p.hi_x && s.y >= p.lo_y && s.y <= p.hi_y);  // UW:149,162 (exact float32 form)
    const float tdx = s.tx - s.x, tdy = s.ty - s.y;
    const float d = norm32(tdx, tdy);                        // UW:150
    const float theta = atan2_fast((float)s.vy, (float)s.vx);    // UW:89
    const float dth = wrap_pi(atan2_fast(tdy, tdx) - theta);     // UW:155-156
    float r = 0.0f - 1.0f / s.init_d;                        // UW:152-153 (float32 under NEP 50)
    r = r + 10.0f * (s.prev_d - d);                          // UW:154
    r = r - 0.1f * fabsf(dth);                               // UW:157
    done = 0;
    if (d < 0.5f) { done = 1; r = r + 1000.0f; }             // UW:159-161
    else if (oob) done = 1;                                  // UW:162-163
    const float speed = sqrtf((float)fma(s.vy, s.vy, s.vx * s.vx));
    obs = uw_obs(p, speed, theta, d, dth);                   // UW:168
    rew = r;
    dist = d;
    s.prev_d = d;                                            // UW:172
    // UW:147: velocity is float64 now; prev_distance is the natural one again
    s.flags = (s.flags & ~(UAVX_FLAG_VEL_F32 | kUwReached | kUwPrevOvr)) | (d < 0.5f ? kUwReached : 0u);
}

__device__ __forceinline__ void uw_load(const UwParams &p, int64_t e, UwRegs &s) {
    const float2 d2 = p.pos[e];
    const double2 v = p.vel[e];
    const UwGoal g = p.goal[e];
    s.x = d2.x; s.y = d2.y;
    s.vx = v.x; s.vy = v.y;
    s.tx = g.tx; s.ty = g.ty; s.init_d = g.init_d; s.flags = g.flags;
    s.prev_d = norm32(s.tx - s.x, s.ty - s.y);
    if (s.flags & kUwPrevOvr) s.prev_d = p.prev_ovr[e];    // rare: only after a caller poked the state
}
// flags_in: the flags word as loaded (stored only if the step changed it)
__device__ __forceinline__ void uw_store(const UwParams &p, int64_t e, const UwRegs &s, uint32_t flags_in) {
    p.pos[e] = make_float2(s.x, s.y);
    p.vel[e] = make_double2(s.vx, s.vy);
    if (s.flags != flags_in) p.goal[e].flags = s.flags;
}
template <bool ACT64>
__device__ __forceinline__ void uw_load_action(const void *__restrict__ actions, int64_t e, double &ax, double &ay) {
    if (ACT64) {
        const double2 a = reinterpret_cast<const double2 *>(actions)[e];
        ax = a.x; ay = a.y;
    } else {
        const float2 a = reinterpret_cast<const float2 *>(actions)[e];
        ax = (double)a.x; ay = (double)a.y;
    }
}

// The pointers the FIRST instructions need and the env count are leading scalar kernel arguments: gfx950 preloads them into SGPRs
// (Makefile: -mllvm -amdgpu-kernarg-preload-count), so the four loads leave before any scalar load of the argument struct (see
// step_kernel in uavx_multi.hip).  A/B (profiles/r04_ab_notes.md section 10): 4 096 envs 2.58 -> 2.48 us, 65 536 3.14 -> 3.00,
// 1 Mi 17.7 -> 16.7 (0.69 -> 0.73 of the HBM figure).
// (pos_in / vel_in / goal_in are p.pos / p.vel / p.goal, which the kernel also stores through: not `__restrict__`.  The same
//  treatment of uw_step_ex_kernel -- record, command and state requested together -- measured no gain: 5.27 vs 5.23 us at 65 536
//  envs, 24.0 vs 23.3 at 1 Mi; not kept.)
template <bool ACT64>
__global__ __launch_bounds__(kBlock) void uw_step_kernel(const void *__restrict__ actions, const float2 *pos_in,
                                                         const double2 *vel_in, const UwGoal *goal_in,
                                                         int64_t num_envs, UwParams p, float4 *__restrict__ obs_out,
                                                         float *__restrict__ rew_out, uint8_t *__restrict__ done_out,
                                                         float *__restrict__ info_out) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= num_envs) return;
    UwRegs s;
    double ax, ay;
    uw_load_action<ACT64>(actions, e, ax, ay);   // requested before the state: prev_distance is arithmetic on what was loaded
    const float2 d2 = pos_in[e];
    const double2 v = vel_in[e];
    const UwGoal g = goal_in[e];
    __builtin_amdgcn_sched_barrier(0);
    s.x = d2.x; s.y = d2.y;
    s.vx = v.x; s.vy = v.y;
    s.tx = g.tx; s.ty = g.ty; s.init_d = g.init_d; s.flags = g.flags;
    s.prev_d = norm32(s.tx - s.x, s.ty - s.y);
    if (s.flags & kUwPrevOvr) s.prev_d = p.prev_ovr[e];    // rare: only after a caller poked the state
    const uint32_t flags_in = s.flags;
    float4 obs; float rew, dist; uint32_t dn;
    uw_step_env(p, s, ax, ay, !ACT64, obs, rew, dn, dist);
    obs_out[e] = obs;
    rew_out[e] = rew;
    done_out[e] = (uint8_t)dn;
    if (info_out) info_out[e] = dist;                        // UW:114-117
    uw_store(p, e, s, flags_in);
    if (threadIdx.x == 0) atomicAdd(&p.wave_steps[blockIdx.x], 1u);   // UW:170 for every env of this wavefront (no-return)
}

// UW:119-131 for one env into registers; stream = the one uw_reset_kernel uses (counter: env, draw, episode).
__device__ __forceinline__ void uw_draw_episode(const UwParams &p, int64_t e, uint32_t episode, uint32_t k0, uint32_t k1,
                                                UwRegs &s) {
    const uint64_t ge = (uint64_t)(p.env_offset + e);
    PhiloxDraws rng{(uint32_t)ge, (uint32_t)(ge >> 32), episode, k0, k1, 0u};
    float vx, vy;
    rng.point32(p.lox, p.loy, p.hix, p.hiy, s.x, s.y);                   // UW:121
    rng.point32(-p.vmax, -p.vmax, p.vmax, p.vmax, vx, vy);               // UW:122
    rng.point32(p.lox, p.loy, p.hix, p.hiy, s.tx, s.ty);                 // UW:126
    s.vx = (double)vx; s.vy = (double)vy;
    s.init_d = s.prev_d = norm32(s.tx - s.x, s.ty - s.y);                // UW:129-130
    s.flags = UAVX_FLAG_VEL_F32;
}
__device__ __forceinline__ void uw_store_fresh(const UwParams &p, int64_t e, const UwRegs &s) {
    p.pos[e] = make_float2(s.x, s.y);
    p.vel[e] = make_double2(s.vx, s.vy);
    p.goal[e] = UwGoal{s.tx, s.ty, s.init_d, s.flags};
}

// An episode of env e ends: fold it into the statistics (test_sac.py:98,106-109).
__device__ __forceinline__ void uw_fold(const UwParams &p, int64_t e, uint32_t steps, bool reached, float ep_return) {
    if (steps != 0) {
        uint4 c = p.fin_counts[e];
        c.x += 1; c.y += steps; c.z += reached ? 1u : 0u;
        p.fin_counts[e] = c;
        p.fin_return[e] += ep_return;
    }
}

// uavx_uw_step_ex: step + polar conversion + next-step auto-reset + episode statistics.
template <bool ACT64>
__global__ __launch_bounds__(kBlock) void uw_step_ex_kernel(UwParams p, UwExtra x, const void *__restrict__ actions,
                                                            float4 *__restrict__ obs_out, float *__restrict__ rew_out,
                                                            uint8_t *__restrict__ done_out, float *__restrict__ info_out) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = e < p.E;
    const uint32_t wave_count = p.wave_steps[blockIdx.x];
    if (live) {
        const uint4 rec = p.rec[e];
        uint32_t steps = wave_count - rec.x;
        UwRegs s;
        if (rec.y & kUwPending) {  // the env starts a new episode instead of stepping
            const uint32_t episode = rec.y & ~kUwPending;
            uw_fold(p, e, steps, (p.goal[e].flags & kUwReached) != 0, __uint_as_float(rec.z));
            uw_draw_episode(p, e, episode, x.seed_lo, x.seed_hi, s);
            uw_store_fresh(p, e, s);
            p.rec[e] = make_uint4(wave_count + 1u, episode + 1u, 0u, 0u);   // UW:131 steps = 0 after this launch
            const float tdx = s.tx - s.x, tdy = s.ty - s.y;
            const float theta = atan2_fast((float)s.vy, (float)s.vx);
            obs_out[e] = uw_obs(p, norm32((float)s.vx, (float)s.vy), theta, s.init_d, wrap_pi(atan2_fast(tdy, tdx) - theta));
            rew_out[e] = 0.f;
            done_out[e] = 0;
            if (info_out) info_out[e] = s.init_d;
            if (x.reset_mask) x.reset_mask[e] = 1;
            if (x.ended) x.ended[e] = 0;
            if (x.truncated) x.truncated[e] = 0;
        } else {
            double ax, ay;
            uw_load_action<ACT64>(actions, e, ax, ay);
            uw_load(p, e, s);
            const uint32_t flags_in = s.flags;
            bool act_f32 = !ACT64;
            if (x.action_mode == UAVX_ACTION_POLAR) {  // test_sac.py:77-80 in float32
                const float v = fmaf((float)ax, 0.5f, 0.5f) * p.high0;
                float sn, cs;
                sincospi32((float)ay, sn, cs);
                ax = (double)(v * cs); ay = (double)(v * sn);
                act_f32 = true;
            }
            float4 obs; float rew, dist; uint32_t dn;
            uw_step_env(p, s, ax, ay, act_f32, obs, rew, dn, dist);
            obs_out[e] = obs;
            rew_out[e] = rew;
            done_out[e] = (uint8_t)dn;
            if (info_out) info_out[e] = dist;
            uw_store(p, e, s, flags_in);
            steps += 1;                                                      // UW:170
            const bool terminal = x.auto_reset && dn;                                 // test_sac.py:106-109
            const bool ended = terminal || (x.step_cap != 0 && steps >= x.step_cap);   // :17
            if (x.ended) x.ended[e] = ended ? 1 : 0;
            if (x.truncated) x.truncated[e] = (ended && !terminal) ? 1 : 0;
            uint4 out = rec;
            out.y = (rec.y & ~kUwPending) | (ended ? kUwPending : 0u);
            if (x.track_returns) out.z = __float_as_uint(__uint_as_float(rec.z) + rew);   // test_sac.py:98
            if (out.y != rec.y || out.z != rec.z) p.rec[e] = out;
            if (x.reset_mask) x.reset_mask[e] = 0;
        }
    }
    if (threadIdx.x == 0) p.wave_steps[blockIdx.x] = wave_count + 1u;   // single writer: this wavefront
}

__global__ __launch_bounds__(kBlock) void uw_episode_stats_kernel(UwParams p, uint32_t *counts, float *returns, int clear) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    if (clear) { p.fin_counts[e] = make_uint4(0, 0, 0, 0); p.fin_return[e] = 0.f; return; }
    if (counts) {
        const uint4 c = p.fin_counts[e];
        counts[4 * e] = c.x; counts[4 * e + 1] = c.y; counts[4 * e + 2] = c.z; counts[4 * e + 3] = 0;
    }
    if (returns) returns[e] = p.fin_return[e];
}

__global__ __launch_bounds__(kBlock) void uw_observe_kernel(UwParams p, float4 *__restrict__ obs_out) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    const float2 d2 = p.pos[e];
    const double2 v = p.vel[e];
    const UwGoal g = p.goal[e];
    const float tdx = g.tx - d2.x, tdy = g.ty - d2.y;
    const float theta = atan2_fast((float)v.y, (float)v.x);
    const float dth = wrap_pi(atan2_fast(tdy, tdx) - theta);
    // UW:88: while the velocity is still reset()'s float32 draw the norm is a float32 one
    const float speed = (g.flags & UAVX_FLAG_VEL_F32) ? norm32((float)v.x, (float)v.y) : sqrtf((float)fma(v.y, v.y, v.x * v.x));
    obs_out[e] = uw_obs(p, speed, theta, norm32(tdx, tdy), dth);
}

// UW:119-131
__global__ __launch_bounds__(kBlock) void uw_reset_kernel(UwParams p, const uint8_t *__restrict__ mask, uint64_t seed) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    if (mask && !mask[e]) return;
    const uint4 rec = p.rec[e];
    const uint32_t wc = p.wave_steps[blockIdx.x];
    const uint32_t episode = rec.y & ~kUwPending;
    UwRegs s;
    uw_draw_episode(p, e, episode, (uint32_t)seed, (uint32_t)(seed >> 32), s);
    uw_fold(p, e, wc - rec.x, (p.goal[e].flags & kUwReached) != 0, __uint_as_float(rec.z));
    uw_store_fresh(p, e, s);
    p.rec[e] = make_uint4(wc, episode + 1u, 0u, 0u);                     // UW:131 steps = 0, new episode, nothing pending
}

__global__ __launch_bounds__(kBlock) void uw_get_state_kernel(UwParams p, uavx_uw_state_view v) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    const float2 d = p.pos[e];
    const UwGoal g = p.goal[e];
    if (v.loc) { v.loc[2 * e] = d.x; v.loc[2 * e + 1] = d.y; }
    if (v.prev_d) v.prev_d[e] = (g.flags & kUwPrevOvr) ? p.prev_ovr[e] : norm32(g.tx - d.x, g.ty - d.y);
    if (v.flags) v.flags[e] = (uint8_t)(g.flags & UAVX_FLAG_VEL_F32);
    if (v.vel) { const double2 w = p.vel[e]; v.vel[2 * e] = w.x; v.vel[2 * e + 1] = w.y; }
    if (v.tgt) { v.tgt[2 * e] = g.tx; v.tgt[2 * e + 1] = g.ty; }
    if (v.init_d) v.init_d[e] = g.init_d;
    if (v.counters) {
        const uint4 rec = p.rec[e];
        v.counters[2 * e] = p.wave_steps[e / kBlock] - rec.x; v.counters[2 * e + 1] = rec.y & ~kUwPending;
    }
}

// Any subset of the fields.  prev_distance keeps the VALUE the reference would hold: what the caller does not pass
// stays what it was, and whenever that value is not the one derived from the new (location, target) it is parked in
// prev_ovr[] behind the PREV_OVR bit until the next step.
__global__ __launch_bounds__(kBlock) void uw_set_state_kernel(UwParams p, uavx_uw_state_view v) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    float2 d = p.pos[e];
    UwGoal g = p.goal[e];
    const float old_prev = (g.flags & kUwPrevOvr) ? p.prev_ovr[e] : norm32(g.tx - d.x, g.ty - d.y);
    if (v.loc) { d.x = v.loc[2 * e]; d.y = v.loc[2 * e + 1]; }
    if (v.tgt) { g.tx = v.tgt[2 * e]; g.ty = v.tgt[2 * e + 1]; }
    if (v.init_d) g.init_d = v.init_d[e];
    uint32_t flags = g.flags & ~kUwPrevOvr;
    if (v.flags) flags = (flags & ~UAVX_FLAG_VEL_F32) | ((uint32_t)v.flags[e] & UAVX_FLAG_VEL_F32);
    const float want = v.prev_d ? v.prev_d[e] : old_prev;
    const float nat = norm32(g.tx - d.x, g.ty - d.y);
    if (__float_as_uint(want) != __float_as_uint(nat)) {
        flags |= kUwPrevOvr;
        p.prev_ovr[e] = want;
    }
    g.flags = flags;
    p.pos[e] = d;
    p.goal[e] = g;
    if (v.vel) p.vel[e] = make_double2(v.vel[2 * e], v.vel[2 * e + 1]);
    if (v.counters) {
        uint4 rec = p.rec[e];
        rec.x = p.wave_steps[e / kBlock] - v.counters[2 * e];
        rec.y = (rec.y & kUwPending) | (v.counters[2 * e + 1] & ~kUwPending);
        p.rec[e] = rec;
    }
}

}  // namespace uavx

using namespace uavx;

struct uavx_uw_handle {
    uavx_uw_config cfg;
    UwParams p;
    int device;
    void *slab;
    std::string err;
};

bool uavx_recip_division_exact(double tau);  // uavx_multi.hip
float uavx_f32_at_or_above(double b);
float uavx_f32_at_or_below(double b);

namespace {

int uw_fail(uavx_uw_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    return code;
}
#define UW_HIP(h, call)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return uw_fail(h, UAVX_ERR_HIP, std::string(#call ": ") + hipGetErrorString(e_)); \
    } while (0)

struct UwDeviceGuard {
    int prev = -1, want;
    hipError_t err = hipSuccess;
    explicit UwDeviceGuard(int device) : want(device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != want) err = hipSetDevice(want);
    }
    ~UwDeviceGuard() {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
};
#define UW_ENTER(h)                  \
    UwDeviceGuard guard_((h)->device); \
    if (guard_.err != hipSuccess) return uw_fail((h), UAVX_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err))

inline size_t uw_align(size_t x) { return (x + 255) / 256 * 256; }
inline dim3 env_grid(const uavx_uw_handle *h) { return dim3((unsigned)((h->p.E + kBlock - 1) / kBlock)); }

}  // namespace

extern "C" {

int uavx_uw_create(const uavx_uw_config *cfg, int64_t num_envs, int64_t env_offset, int device, uavx_uw_handle **out) {
    if (!cfg || !out || num_envs <= 0 || env_offset < 0) return UAVX_ERR_INVALID_ARG;
    if (!(cfg->tau > 0) || !(cfg->max_speed > 0) || !(cfg->max_acceleration > 0) || !(cfg->x_size > 0) || !(cfg->y_size > 0))
        return UAVX_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return UAVX_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return UAVX_ERR_INVALID_ARG;
    uavx_uw_handle *h = new (std::nothrow) uavx_uw_handle();
    if (!h) return UAVX_ERR_ALLOC;
    h->cfg = *cfg;
    h->device = device;
    h->slab = nullptr;
    UwParams &p = h->p;
    std::memset(&p, 0, sizeof p);
    p.tau = cfg->tau; p.amax = cfg->max_acceleration; p.vmax = cfg->max_speed;
    p.rtau = 1.0 / cfg->tau;
    p.recip_ok = uavx_recip_division_exact(cfg->tau) ? 1 : 0;
    p.lox = -cfg->x_size / 2.0; p.loy = -cfg->y_size / 2.0; p.hix = cfg->x_size / 2.0; p.hiy = cfg->y_size / 2.0;
    p.lo_x = uavx_f32_at_or_above(p.lox); p.lo_y = uavx_f32_at_or_above(p.loy);
    p.hi_x = uavx_f32_at_or_below(p.hix); p.hi_y = uavx_f32_at_or_below(p.hiy);
    p.tau_f = (float)cfg->tau;
    p.high0 = (float)cfg->max_speed;
    p.inv_vmax = (float)(1.0 / cfg->max_speed);
    p.inv_diag = (float)(1.0 / std::sqrt(std::fma(cfg->y_size, cfg->y_size, cfg->x_size * cfg->x_size)));
    p.E = num_envs;
    p.env_offset = env_offset;
    UwDeviceGuard guard(device);
    if (guard.err != hipSuccess) { delete h; return UAVX_ERR_HIP; }
    const size_t E = (size_t)num_envs;
    size_t off = 0;
    const size_t o_pos = off;  off = uw_align(off + E * sizeof(float2));
    const size_t o_vel = off;  off = uw_align(off + E * sizeof(double2));
    const size_t o_goal = off; off = uw_align(off + E * sizeof(UwGoal));
    const size_t o_ovr = off;  off = uw_align(off + E * 4);
    const size_t o_wsteps = off; off = uw_align(off + ((E + kBlock - 1) / kBlock) * 4);
    const size_t o_rec = off;   off = uw_align(off + E * sizeof(uint4));
    const size_t o_finc = off;  off = uw_align(off + E * sizeof(uint4));
    const size_t o_finr = off;  off = uw_align(off + E * 4);
    if (hipMalloc(&h->slab, off) != hipSuccess) { delete h; return UAVX_ERR_ALLOC; }
    if (hipMemset(h->slab, 0, off) != hipSuccess) { (void)hipFree(h->slab); delete h; return UAVX_ERR_HIP; }
    char *b = static_cast<char *>(h->slab);
    p.pos = reinterpret_cast<float2 *>(b + o_pos);
    p.vel = reinterpret_cast<double2 *>(b + o_vel);
    p.goal = reinterpret_cast<UwGoal *>(b + o_goal);
    p.prev_ovr = reinterpret_cast<float *>(b + o_ovr);
    p.wave_steps = reinterpret_cast<uint32_t *>(b + o_wsteps);
    p.rec = reinterpret_cast<uint4 *>(b + o_rec);
    p.fin_counts = reinterpret_cast<uint4 *>(b + o_finc);
    p.fin_return = reinterpret_cast<float *>(b + o_finr);
    *out = h;
    return UAVX_OK;
}

int uavx_uw_destroy(uavx_uw_handle *h) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (h->slab) {
        UwDeviceGuard guard(h->device);
        (void)hipFree(h->slab);
    }
    delete h;
    return UAVX_OK;
}

const char *uavx_uw_last_error(const uavx_uw_handle *h) { return h ? h->err.c_str() : "null handle"; }

int uavx_uw_observe(uavx_uw_handle *h, float *obs, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (!obs || (reinterpret_cast<uintptr_t>(obs) & 15u))
        return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_observe: obs is NULL or not 16-byte aligned");
    UW_ENTER(h);
    hipLaunchKernelGGL(uw_observe_kernel, env_grid(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->p,
                       reinterpret_cast<float4 *>(obs));
    UW_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_uw_reset(uavx_uw_handle *h, const uint8_t *mask, uint64_t seed, float *obs, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (obs && (reinterpret_cast<uintptr_t>(obs) & 15u))
        return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_reset: obs not 16-byte aligned");
    UW_ENTER(h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(uw_reset_kernel, env_grid(h), dim3(kBlock), 0, st, h->p, mask, seed);
    UW_HIP(h, hipGetLastError());
    if (obs) {
        hipLaunchKernelGGL(uw_observe_kernel, env_grid(h), dim3(kBlock), 0, st, h->p, reinterpret_cast<float4 *>(obs));
        UW_HIP(h, hipGetLastError());
    }
    return UAVX_OK;
}

int uavx_uw_step(uavx_uw_handle *h, const void *actions, int action_dtype, float *obs, float *rew, uint8_t *done,
                 float *info_distance, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (!actions || !obs || !rew || !done) return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_step: NULL buffer");
    if (action_dtype != UAVX_F32 && action_dtype != UAVX_F64)
        return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_step: action_dtype must be UAVX_F32 or UAVX_F64");
    if ((reinterpret_cast<uintptr_t>(obs) & 15u) || (reinterpret_cast<uintptr_t>(actions) & (action_dtype == UAVX_F64 ? 15u : 7u)) ||
        (reinterpret_cast<uintptr_t>(rew) & 3u) || (reinterpret_cast<uintptr_t>(info_distance) & 3u))
        return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_step: obs must be 16-byte aligned, actions 8 (float32) / 16 (float64), rew and info 4");
    UW_ENTER(h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (action_dtype == UAVX_F64)
        hipLaunchKernelGGL((uw_step_kernel<true>), env_grid(h), dim3(kBlock), 0, st, actions, h->p.pos, h->p.vel, h->p.goal, (int64_t)h->p.E,
                           h->p, reinterpret_cast<float4 *>(obs), rew, done, info_distance);
    else
        hipLaunchKernelGGL((uw_step_kernel<false>), env_grid(h), dim3(kBlock), 0, st, actions, h->p.pos, h->p.vel, h->p.goal, (int64_t)h->p.E,
                           h->p, reinterpret_cast<float4 *>(obs), rew, done, info_distance);
    UW_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_uw_step_ex(uavx_uw_handle *h, const uavx_uw_step_args *a, void *stream) {
    if (!h || !a) return UAVX_ERR_INVALID_ARG;
    if (!a->actions || !a->obs || !a->rew || !a->done) return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_step_ex: NULL buffer");
    if (a->action_dtype != UAVX_F32 && a->action_dtype != UAVX_F64)
        return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_step_ex: action_dtype must be UAVX_F32 or UAVX_F64");
    if (a->action_mode != UAVX_ACTION_CARTESIAN && a->action_mode != UAVX_ACTION_POLAR)
        return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_step_ex: unknown action_mode");
    if ((reinterpret_cast<uintptr_t>(a->obs) & 15u) || (reinterpret_cast<uintptr_t>(a->actions) & (a->action_dtype == UAVX_F64 ? 15u : 7u)) ||
        (reinterpret_cast<uintptr_t>(a->rew) & 3u) || (reinterpret_cast<uintptr_t>(a->info_distance) & 3u))
        return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_step_ex: obs must be 16-byte aligned, actions 8 (float32) / 16 (float64), rew and info 4");
    UW_ENTER(h);
    UwExtra x;
    x.action_mode = a->action_mode; x.auto_reset = a->auto_reset; x.track_returns = a->track_returns;
    x.step_cap = a->step_cap; x.seed_lo = (uint32_t)a->seed; x.seed_hi = (uint32_t)(a->seed >> 32);
    x.reset_mask = a->reset_mask; x.ended = a->ended; x.truncated = a->truncated;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (a->action_dtype == UAVX_F64)
        hipLaunchKernelGGL((uw_step_ex_kernel<true>), env_grid(h), dim3(kBlock), 0, st, h->p, x, a->actions,
                           reinterpret_cast<float4 *>(a->obs), a->rew, a->done, a->info_distance);
    else
        hipLaunchKernelGGL((uw_step_ex_kernel<false>), env_grid(h), dim3(kBlock), 0, st, h->p, x, a->actions,
                           reinterpret_cast<float4 *>(a->obs), a->rew, a->done, a->info_distance);
    UW_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_uw_get_episode_stats(uavx_uw_handle *h, uint32_t *counts, float *returns, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    UW_ENTER(h);
    hipLaunchKernelGGL(uw_episode_stats_kernel, env_grid(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->p,
                       counts, returns, 0);
    UW_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_uw_clear_episode_stats(uavx_uw_handle *h, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    UW_ENTER(h);
    hipLaunchKernelGGL(uw_episode_stats_kernel, env_grid(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->p,
                       (uint32_t *)nullptr, (float *)nullptr, 1);
    UW_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_uw_get_state(uavx_uw_handle *h, const uavx_uw_state_view *dst, void *stream) {
    if (!h || !dst) return UAVX_ERR_INVALID_ARG;
    UW_ENTER(h);
    hipLaunchKernelGGL(uw_get_state_kernel, env_grid(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->p, *dst);
    UW_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_uw_set_state(uavx_uw_handle *h, const uavx_uw_state_view *src, void *stream) {
    if (!h || !src) return UAVX_ERR_INVALID_ARG;
    UW_ENTER(h);
    hipLaunchKernelGGL(uw_set_state_kernel, env_grid(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->p, *src);
    UW_HIP(h, hipGetLastError());
    return UAVX_OK;
}

}  // extern "C"
