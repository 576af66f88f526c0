// uavx_multi.hip — E x MultiUAVWorld2D on one MI355X: fused step / reset / observe kernels and the
// C ABI of include/uavx.h.  Hand-written for gfx950 (wave64, LDS-staged neighbour exchange).
//
// Reference semantics (cited per block): MUW = gym_uav_collision_avoidance/envs/multi_uav_world_2d.py,
// AG = .../uav_agent.py of dazchi/gym-uav-collision-avoidance.
//
// Work mapping.  One LANE per agent, floor(64/N) whole envs per WAVEFRONT, one wavefront per workgroup, so that all
// cross-agent traffic of an env stays inside its wave (agent counts that would leave many lanes of one wavefront idle use
// workgroups of 2..4 wavefronts instead, pick_group_waves(): an env may then span two waves of ONE workgroup, and the
// wave-local ordering points below become workgroup barriers):
//   1. every lane integrates its own agent (float64 velocity, float32 position, AG:23-36) — the
//      reference's sequential loop over agents (MUW:181) has no real dependency here because an
//      agent's motion only reads its own state;
//   2. old + new position and heading of every agent are staged in LDS (20 B per lane);
//   3. every lane scans the N-1 other agents of its env from LDS: the Gauss-Seidel rule "agents
//      j<i already moved, j>i not yet" (MUW:181-231) becomes a per-pair select between the staged
//      new/old position; min-distance feeds the collision tests, the two nearest at final positions
//      feed the observation (AG:44-64, MUW:75-95);
//   4. reward / collision / termination / finish() per lane (MUW:188-231, AG:38-42);
//   5. the 10 observation floats per agent are transposed through LDS so the wave writes its
//      2560-byte obs block with 16-byte-per-lane contiguous stores.
// HBM layout (agent slot a = e*N + i, agent fastest => lane-contiguous):
//   pos  float2[A]  {x, y}                     read+write    8 B
//   vel  double2[A] {vx, vy}                   read+write   16 B
//   goal float4[A]  {tx, ty, init_d, flags}    read         16 B   (flags word rewritten only when it changes)
//   prev_distance (AG:18) is NOT stored: for an agent that is not done it always equals ||target - location||
//   (AG:33-34, MUW:155,229), so it is recomputed from the loaded position; values a caller pokes in that
//   break this identity live in prev_ovr[A] behind the PREVD_OVR flag bit (read only when the bit is set).
//   per workgroup: wave_steps[G] (one no-return atomic per launch; env.steps = wave_steps - env_rec.x)
//   per env:   reach[E] / coll[E] (atomics on the rare events), env_rec[E] (16 B: steps base, episode index,
//              running returns; touched by reset / step_ex only), episode statistics (fin_*).
// Kernel arguments: the step kernels take what their FIRST instructions need as leading scalars (command pointer, base of the
// state allocation + 32-bit array offsets, the numbers of the lane mapping), which gfx950 preloads into SGPRs with the wavefront
// (Makefile: -mllvm -amdgpu-kernarg-preload-count), and everything else in the by-value MultiParams behind them: the state loads
// are the first thing a wavefront does (step_kernel, step_ex_kernel; profiles/r04_ab_notes.md section 10).
// BASELINE configs[4] extension (scripted bodies, per-env curriculum levels; EXT kernel variants): see MultiParams and
// include/uavx.h; lanes stay one per LEARNER there and the bodies are extra rows of the env's LDS neighbour tile.
// A body is a position (float2, read + written while it moves) and a leg record {dx, dy, heading, legs} (float4, read
// only between two waypoint changes): 24 B read + 8 B written per body-step, two float32 additions of arithmetic.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <type_traits>

#include "../../include/uavx.h"
#include "uavx_device.hpp"

namespace uavx {

// LATE(on, p, member): p.member -- with `on` (a compile-time flag of the kernel variant) fetched where it is used instead of
// at the top of the kernel (late_karg(), uavx_device.hpp): for members that only a rare branch or the last instructions of a
// wavefront need, in the variants whose scalar registers are tight (8 wavefronts per SIMD = 80 SGPRs).  `p` must be the kernel's
// FIRST argument -- or, in step_ex_kernel, sit kExLead bytes into the segment (the macros add that).  NOT a free lunch, hence per variant: the headline kernel (4 UAVs, 66
// SGPRs, nothing to gain) lost 0.34 us of 5.85 with its `prev_ovr` / counter pointers fetched late -- the scalar loads at the top
// regroup (one dwordx8 became a dwordx2 + a dwordx4) and that launch is latency-shaped (profiles/r04_ab_notes.md).
// -DUAVX_LATE=0 turns every site off (A/B).
#ifndef UAVX_LATE
#define UAVX_LATE 1
#endif
template <bool ON>
__device__ __forceinline__ karg_ptr kargs_if() {
    if constexpr (ON && UAVX_LATE) return late_kargs();
    else return nullptr;
}
template <bool ON, class T>
__device__ __forceinline__ T karg_if(T plain, uint32_t byte_off) {   // (by value: an unused read of a kernel argument folds away)
    if constexpr (ON && UAVX_LATE) return late_karg<T>(byte_off);
    else return plain;
}
template <bool ON, class T>
__device__ __forceinline__ T karg_if(T plain, uint32_t byte_off, karg_ptr ka) {
    if constexpr (ON && UAVX_LATE) return late_karg<T>(byte_off, ka);
    else return plain;
}
// (every site that is switched on lives in step_ex_kernel, whose MultiParams sits behind kExLead bytes of leading scalar arguments)
constexpr uint32_t kExLead = 56;
#define LATE(on, p, member) karg_if<(on)>((p).member, kExLead + (uint32_t)offsetof(MultiParams, member))
// several members at one place: `LATE_BASE(on, ka);` once, then LATE_AT(on, ka, p, member) (one laundering point for all of them)
#define LATE_BASE(on, ka) const karg_ptr ka = kargs_if<(on)>()
#define LATE_AT(on, ka, p, member) karg_if<(on)>((p).member, kExLead + (uint32_t)offsetof(MultiParams, member), ka)
// members of uavx_step_ex's options block: the SECOND argument of step_ex_kernel, directly behind the first (static_assert below)
#define LATE_X(on, x, member) karg_if<(on)>((x).member, kExLead + (uint32_t)(sizeof(MultiParams) + offsetof(StepExtra, member)))
#define LATE_X_AT(on, ka, x, member) karg_if<(on)>((x).member, kExLead + (uint32_t)(sizeof(MultiParams) + offsetof(StepExtra, member)), ka)
// which sites a variant of step_ex_kernel switches on (bit mask, -DUAVX_LATE_EX=... for A/B): 1 the counter atomics at the end of a step,
// 2 a body's new waypoint (stage_bodies), 4 the episode fold, 8 step_ex's re-initialisation block, its tail pointers and flag arrays
#ifndef UAVX_LATE_EX
#define UAVX_LATE_EX 15       // step_ex_kernel with bodies / levels, and its 8-UAV specialisation
#endif

struct Goal { float tx, ty, init_d; uint32_t flags; };  // 16 B, one dwordx4 load; flags word stored only on change

// flag bits kept in Goal::flags (bits 0,1 are the public UAVX_FLAG_DONE / UAVX_FLAG_COLLIDED)
constexpr uint32_t kFlagPublic = UAVX_FLAG_DONE | UAVX_FLAG_COLLIDED | UAVX_FLAG_INACTIVE;
constexpr uint32_t kFlagPrevOvr = 8u;    // prev_distance is the value in prev_ovr[a], not ||target - location||
constexpr uint32_t kFlagJustDone = 16u;  // finished during the last step: prev_distance is still the distance then
                                         // (MUW:229 stores it once more; from the next step on it is 0, AG:24-25)

// One curriculum level as the kernels use it (uavx_level with the exact comparison limits precomputed on the host).
struct alignas(16) LevelParams {
    float lo_x, lo_y, hi_x, hi_y;                    // x inside [lox, hix] (MUW:213,224) <=> lo_x <= x <= hi_x in float32
    float sq_sense, sq_two_r, inv_sense, inv_diag;   // as the MultiParams fields of the same names
    double lox, loy, hix, hiy;                       // reset box (MUW:19-20); reset path only
    int32_t n_active, b_active, pad0, pad1;
};
// The world limits one lane works with: the handle's (kernel arguments, scalar registers) or its env's level's.
struct WorldLims {
    float lo_x, lo_y, hi_x, hi_y;
    float sq_sense, sq_two_r, inv_sense, inv_diag;
};

// The kernels take this struct BY VALUE: it is (most of) their kernel-argument segment, fetched by scalar loads, and a
// 65 536 x 4 step launch is latency-shaped -- so the ORDER of the members is a tuning parameter, and not an
// intuitive one.  (Since round 4 the arguments the first instructions need travel in front of it as preloaded leading scalars.)  Measured on one box (profiles/r03_ab_notes.md): a new pointer inserted after `coll`
// cost the headline launch 0.35 us of 5.72 with NO other change to the kernel (the members behind it moved across the
// scalar-load groups the compiler forms, nine loads instead of six sat in front of the first wait); the same pointer appended
// at the end costs nothing beyond its own use (5.78 with the tripwire it serves); a deliberate "hot members first, one
// 64-byte line per phase" order was WORSE at 4 UAVs (5.93) and better with scripted bodies (16.9 vs 17.4); parameters read
// from a device-resident block instead (one pointer in the kernel arguments) gave 5.78 / 17.7.  New members go at the END.
struct MultiParams {
    double tau, rtau, amax, vmax;  // rtau = RN(1/tau), see div_tau()
    double lox, loy, hix, hiy;
    float lo_x, lo_y, hi_x, hi_y;  // float32 forms of the box test: (double)x >= lox  <=>  x >= lo_x  (smallest float32 >= lox) etc.
    double speed_sq_lim;  // ‖v‖ < 0.2 (MUW:218)  <=>  fma(vy,vy,vx*vx) < speed_sq_lim
    // exact float32 limits on the SQUARED distance s = fl(dx*dx)+fl(dy*dy) (sqrtf is monotone):
    float sq_sense;       // sqrtf(s) <  float32(d_sense)   <=>  s <  sq_sense   (AG:52)
    float sq_two_r;       // sqrtf(s) <= float32(2R)        <=>  s <= sq_two_r   (MUW:203)
    float sq_hard;        // sqrtf(s) <= 1.0                <=>  s <= sq_hard    (MUW:207)
    float inv_sense;      // 1/float32(d_sense)             MUW:77
    float vmax_norm;      // ‖(max_speed,max_speed)‖        MUW:62,183
    float inv_vmax_norm;
    float inv_diag;       // 1/‖(x_size,y_size)‖            MUW:17,68
    float two_r_reset;    // float32(2R), reset rejection (MUW:135,146,151)
    int recip_ok;         // div_tau() may use the reciprocal form for this tau
    int N, epw, magic;    // agents per env, envs per wave, 65536/N + 1 (lane / N == (lane * magic) >> 16 for lane < 64)
    int64_t E, env_offset;
    float2 *pos;
    float *prev_ovr;
    double2 *vel;
    Goal *goal;
    // env.steps (MUW:238) = wave_steps[wave of the env] - steps_base[env]: a step launch bumps ONE counter per
    // wavefront (every env of a wave is stepped by the same launches) instead of one word per env; reset and
    // set_state move the env's base (A/B at 65536x4: per-env counter updates cost 3 % of the launch).
    uint32_t *wave_steps;
    // per-env record, ONE 16-byte word (one load / one store per env in step_ex instead of four / three):
    //   .x steps_base   .y episode index (bits 0..30) | episode-ended flag (bit 31)   .z,.w running episode
    //   return of agent 0 and evaluation score sum_i r_i*(1-done_i) (float bits)
    uint4 *env_rec;
    uint32_t *reach, *coll;
    // episode bookkeeping (uavx_step_ex / uavx_reset)
    uint4 *fin_counts;  // [E] over ended episodes: {episodes, steps, target_reach_count, collision_count}
    float2 *fin_returns;  // [E] over ended episodes: {agent-0 return sum, evaluation score sum}
    // ---- configs[4] extension (scripted bodies + curriculum levels; EXT kernel variants only, see include/uavx.h) ----
    // Lanes stay one per LEARNER (N = L above); the B bodies of an env are slots L..L+B-1 of its LDS neighbour rows and
    // are moved by the env's learner lanes, body b by lane b % L in trip b / L.
    int B, nslots, kb;            // bodies per env, L + B, ceil(B / L)
    int body_pmask, body_pshift;  // period - 1, log2(period) (period is a power of two)
    float body_step;              // float32(speed * tau): metres per env step
    uint32_t body_k0, body_k1;    // Philox key of the waypoint streams
    int n_levels, level_lo, level_hi;
    float2 *body_pos;             // [E*B] {x, y}; +inf for a body its env's level switches off
    float4 *body_leg;             // [E*B] {dx, dy, heading, legs}: displacement per env step, direction of travel, steps of the leg that move
    uint8_t *lvl_cur, *lvl_next;  // [E] level in force / level assigned for the next reset
    const LevelParams *levels;    // [UAVX_MAX_LEVELS] device table, read only while a curriculum is installed (n_levels > 0)
    // ---- pre-drawn layouts (uavx_step_ex auto-reset; see stage_ahead) ----
    // The layout of an env's NEXT episode is a pure function of (seed, global env, episode index, level rule), so it is
    // drawn ahead of time by staging workgroups at the front of an earlier step launch and parked here; the step launch that
    // re-initialises the env then copies 16 B per slot instead of running the serial accept / reject chain on one wavefront while the
    // rest of the chip waits for it.  stage_tag says exactly what a parked layout was drawn for; anything else is a miss
    // and falls back to drawing in the step launch.
    // TWO parked layouts per env, for its next episode and the one after (slot = episode index & 1, slot-major arrays): the
    // layout of episode y + 1 is already there when episode y's is consumed, so an env only ever draws in place when two of its
    // episodes end within the two or three launches it takes to park a layout again
    float4 *stage_agent;          // [2][E*L] {sx, sy, tx, ty}
    float2 *stage_bpos;           // [2][E*B] as body_pos
    float4 *stage_bleg;           // [2][E*B] as body_leg
    uint4 *stage_tag;             // [2][E] {episode index, seed lo, seed hi, level | world version << 8 | valid << 31}
    int magic_s;                  // 65536 / (L + B) + 1: thread / (L + B) of a staging workgroup by multiply-shift
    uint32_t world_version;       // bumped by every call that changes what a layout depends on (config, curriculum, body rule)
    // agent-steps of the running episode whose reward came out non-finite (uavx_get_nonfinite): a NaN command or state
    // poisons an agent for good (AG:26-27 lets it through), and at 65 536 envs nobody scans the observations for it.
    // (LAST on purpose, see the note above the struct.)
    uint32_t *nonfin;
};
constexpr uint32_t kStageValid = 0x80000000u;
constexpr uint32_t kRecEnded = 0x80000000u;  // env_rec.y bit 31: episode ended, re-initialise at the next step_ex
constexpr uint32_t kFlagInactive = UAVX_FLAG_INACTIVE;
constexpr int kLevelShift = 8;       // Goal::flags bits 8..11: the env's curriculum level (same value in every agent of the env)
constexpr uint32_t kLevelMask = 0xFu << kLevelShift;
constexpr int kExtSlots = 192;       // LDS neighbour rows per wave of an EXT kernel: epw * (L + B) <= 192
constexpr int kHintJobs = 8;         // layouts a staging workgroup takes on per draw: its 8 hint slots are ONE 64-byte scalar load

// options of uavx_step_ex that the kernel needs (uavx_step_args minus the buffers)
struct StepExtra {
    int action_mode, reset_policy, track_returns;
    uint32_t step_cap;
    uint32_t seed_lo, seed_hi;
    uint8_t *reset_mask;
    uint8_t *ended, *truncated;
    int flags_in_done;   // UAVX_FLAGS_IN_DONE: the three per-env flags travel in bits 1..3 of the env's first done byte
    int use_stage;   // consult the pre-drawn layouts
    // layouts drawn ahead: pf_blocks workgroups of the launch do not step anything -- they look for, and draw, the layouts of
    // the NEXT episodes (see stage_ahead); env-workgroup w is workgroup step_first + w
    uint32_t pf_blocks, pf_groups;   // staging workgroups, env-workgroups of the launch
    uint32_t stage_first, step_first;   // block id of the first staging / first env-workgroup: (0, pf_blocks) or (pf_groups, 0)
    uint2 *hints;                    // [pf_blocks][kHintJobs] {env + 1 (0: none), episode}: what a staging workgroup's last scan found
};

static_assert(sizeof(MultiParams) % alignof(StepExtra) == 0, "LATE_X: StepExtra must follow MultiParams without padding in the kernel-argument segment");

// The caller's buffers follow StepExtra in step_ex_kernel's argument list as plain parameters (`__restrict__`: they do not
// alias, and the compiler orders loads against stores on that knowledge -- handing them over in a struct cost the 4-UAV fused
// launch 0.2 us of 7.0).  Their places in the kernel-argument segment, for the variants that fetch the two output pointers
// again at their end (LATE_IO): each parameter sits at the next multiple of its alignment.
constexpr uint32_t kIoBase = kExLead + (uint32_t)(sizeof(MultiParams) + sizeof(StepExtra));   // int evaluate
constexpr uint32_t kIoRew = kIoBase + 16, kIoDone = kIoBase + 24;                    // (float *obs_out +8,) rew_out, done_out
static_assert(sizeof(StepExtra) % 8 == 0, "the pointer parameters behind StepExtra start on its end");

struct LaneMap {
    int lane;        // thread in its workgroup (the lane when the workgroup is one wavefront)
    int i, base;     // agent index in its env, first lane of the env's group
    int rbase, nslots;  // first LDS neighbour row of the env, rows per env (N unless the env has scripted bodies)
    int nlearn;         // the first nlearn rows of an env are agents with a lane each (= N)
    int g;              // env index within the workgroup
    bool active;
    uint32_t e, a;   // env, agent slot (E*N < 2^26, checked by uavx_create)
    uint32_t a0;     // first agent slot of this workgroup
    uint32_t wave;   // workgroup index (= index into wave_steps)
    int obs0;        // first float of this wavefront's obs staging tile in lds.obs (0 unless the workgroup holds several tiles)
    int cnt;         // active agent slots in this workgroup: threads [0, cnt), slots [a0, a0 + cnt)
};

// Work mapping of one launch: a workgroup of W wavefronts holds epw = floor(64 W / N) whole envs, one thread per agent,
// packed from thread 0 (so agent slot = a0 + thread id).  W = 1 everywhere except for agent counts that would leave
// many lanes of a single wavefront idle (N = 24: 48 of 64; three wavefronts hold 8 envs with none idle).
template <int NT, bool EXT = false, int W = 1>
__device__ __forceinline__ LaneMap lane_map_from(uint32_t E, int n_agents, int envs_per_group, int magic, int nslots, uint32_t wave,
                                                 uint32_t lane = threadIdx.x, uint32_t tile = 0u) {
    LaneMap m;
    const int N = NT ? NT : n_agents;
    const int epw = NT ? (kWave / (NT ? NT : 1)) : envs_per_group;
    m.lane = lane;                   // thread in its workgroup (= lane for W == 1; tiled workgroups pass their lane)
    int g;
    if (NT) {
        g = m.lane / (NT ? NT : 1);
        m.i = m.lane % (NT ? NT : 1);
    } else {
        g = (m.lane * magic) >> 16;  // floor(thread / N) for thread < 256
        m.i = m.lane - g * N;
    }
    m.wave = wave;
    const uint32_t e0 = wave * epw;
    const uint32_t envs_here = e0 < E ? min(E - e0, (uint32_t)epw) : 0u;
    m.e = e0 + g;
    m.active = (uint32_t)g < envs_here;
    m.base = m.active ? (g * N) & (kWave - 1) : 0;  // first lane of the env's group in its wavefront (W == 1: ballot shifts)
    m.g = m.active ? g : 0;
    m.nslots = EXT ? nslots : N;
    m.nlearn = N;
    m.rbase = (m.active ? g * m.nslots : 0) + (int)tile * kWave;  // idle lanes still execute the LDS scan: keep it in bounds
    m.obs0 = (int)tile * (kWave * UAVX_OBS_DIM);
    m.a0 = e0 * N;
    m.a = m.a0 + m.lane;            // whole envs are packed from thread 0: slot = a0 + thread
    m.cnt = (int)envs_here * N;
    return m;
}
template <int NT, bool EXT = false, int W = 1>
__device__ __forceinline__ LaneMap lane_map(const MultiParams &p, uint32_t wave = blockIdx.x) {   // wave: env-workgroup index
    return lane_map_from<NT, EXT, W>((uint32_t)p.E, p.N, p.epw, p.magic, p.nslots, wave);
}

struct AgentRegs {
    float x, y, prev_d;
    uint32_t flags;
    float tx, ty, init_d;
    double vx, vy;
};

// LDS of one workgroup (W wavefronts; W > 1 only on the runtime-N path, see pick_group_waves()).
template <bool EXT, int W = 1>
struct LdsT {
    static constexpr int kW = W;
    static constexpr int kRows = (EXT && kExtSlots > kWave * W) ? kExtSlots : kWave * W;
    float4 pos[kRows];            // {old.x, old.y, new.x, new.y} per neighbour slot
    float theta[kRows];           // heading atan2(vy, vx)
    float obs[kWave * W * UAVX_OBS_DIM];
};
// With scripted bodies the neighbour rows alone are 3.8 KB per wavefront; the obs staging tile shares their bytes: it is
// written after the last read of the rows (one wavefront per workgroup: DS operations execute in issue order), so the
// workgroup needs 3 840 B instead of 6 400 B and a CU holds 28 wavefronts (the register limit) instead of 25.
template <>
struct LdsT<true, 1> {
    static constexpr int kW = 1;
    static constexpr int kRows = kExtSlots;
    union {
        struct {
            float4 pos[kRows];
            float theta[kRows];
        };
        float obs[kWave * UAVX_OBS_DIM];
    };
};
// All cross-agent traffic of an env stays inside its workgroup.  With one wavefront per workgroup a compiler-level
// ordering point is enough (wave_lds_sync); an env that spans two wavefronts needs the workgroup barrier.
template <int W>
__device__ __forceinline__ void group_sync() {
    if (W == 1) wave_lds_sync();
    else __syncthreads();
}
template <int W>
__device__ __forceinline__ bool group_any(bool v) {   // same answer in every thread of the workgroup
    if (W == 1) return __ballot(v) != 0ull;
    return __syncthreads_or(v ? 1 : 0) != 0;
}
using Lds = LdsT<false>;
// T one-wavefront TILES side by side in one workgroup (step_kernel / step_ex_kernel, T > 1): tile t owns rows [64 t, 64 t + 64)
// of each array -- the tile offset rides in the lane map's row base and obs0, so no LDS address needs a register of its own --
// and orders its traffic at wavefront level like a one-wavefront workgroup (kW = 1).
template <int T>
struct LdsTiles {
    static constexpr int kW = 1;
    static constexpr int kRows = kWave * T;
    float4 pos[kRows];
    float theta[kRows];
    float obs[kWave * T * UAVX_OBS_DIM];
};

// World limits of this lane's env: kernel arguments, or (EXT) the level its flags word names -- two 16-byte loads from a
// table every lane of the chip shares, i.e. an L1/L2 hit whose latency hides under the kinematics.
// d_sense alone (all the neighbour scan needs): the other seven limits are fetched AFTER the scan, where they are used --
// per-lane values loaded at the top of the step stayed in eight registers across the scan, the most register-hungry stretch
template <bool EXT>
__device__ __forceinline__ float sense_limit(const MultiParams &p, uint32_t flags) {
    if (EXT && p.n_levels > 0) return p.levels[(flags & kLevelMask) >> kLevelShift].sq_sense;
    return p.sq_sense;
}
template <bool EXT>
__device__ __forceinline__ WorldLims world_lims(const MultiParams &p, uint32_t flags) {
    WorldLims w;
    if (EXT && p.n_levels > 0) {   // uniform: no curriculum installed -> the handle's own world, as in the plain kernels
        const float4 *t = reinterpret_cast<const float4 *>(&p.levels[(flags & kLevelMask) >> kLevelShift]);
        const float4 a = t[0], b = t[1];
        w.lo_x = a.x; w.lo_y = a.y; w.hi_x = a.z; w.hi_y = a.w;
        w.sq_sense = b.x; w.sq_two_r = b.y; w.inv_sense = b.z; w.inv_diag = b.w;
    } else {
        w.lo_x = p.lo_x; w.lo_y = p.lo_y; w.hi_x = p.hi_x; w.hi_y = p.hi_y;
        w.sq_sense = p.sq_sense; w.sq_two_r = p.sq_two_r; w.inv_sense = p.inv_sense; w.inv_diag = p.inv_diag;
    }
    return w;
}

// Agent slots are addressed with 32-bit lane offsets from scalar base pointers (saddr + voffset
// addressing; uavx_create rejects E*N >= 2^26).
// prev_distance as the reference would hold it for this (flags, position, target)
__device__ __forceinline__ float natural_prev_d(uint32_t flags, float x, float y, float tx, float ty) {
    const float d = norm32(tx - x, ty - y);
    return ((flags & UAVX_FLAG_DONE) && !(flags & kFlagJustDone)) ? 0.f : d;
}

__device__ __forceinline__ void load_agent(const MultiParams &p, uint32_t a, AgentRegs &s) {
    const float2 d = p.pos[a];
    const double2 v = p.vel[a];
    const Goal g = p.goal[a];
    s.x = d.x; s.y = d.y;
    s.vx = v.x; s.vy = v.y;
    s.tx = g.tx; s.ty = g.ty; s.init_d = g.init_d; s.flags = g.flags;
    s.prev_d = natural_prev_d(s.flags, s.x, s.y, s.tx, s.ty);
    if (s.flags & kFlagPrevOvr) s.prev_d = p.prev_ovr[a];  // rare: only after a caller poked the state
}
// flags_in: the flags word as loaded (the word is stored only if the step changed it)
// (pos / vel / goal: p's arrays, handed over separately because the register-tight kernels fetch those pointers again at
//  their end instead of holding them in scalar registers from the loads at the top on, see LATE())
__device__ __forceinline__ void store_agent(const MultiParams &p, float2 *pos, double2 *vel, Goal *goal, uint32_t a,
                                            const AgentRegs &s, uint32_t flags_in) {
    // velocity: 16 B per lane, written through (sc1) like the obs tile so that it drains during the launch instead
    // of at the kernel boundary (A/B at 65536x4: 6.71 -> 6.21 us); the 8-byte position store stays plain
    // (narrow sc1 stores are slow: 6.29 us with both)
    store16_wt(make_rsrc(vel, (uint32_t)p.E * (uint32_t)p.N * 16u), a * 16u, make_double2(s.vx, s.vy));
    pos[a] = make_float2(s.x, s.y);
    if (s.flags != flags_in) goal[a].flags = s.flags;
}
__device__ __forceinline__ void store_agent(const MultiParams &p, uint32_t a, const AgentRegs &s, uint32_t flags_in) {
    store_agent(p, p.pos, p.vel, p.goal, a, s, flags_in);
}

// Neighbour scan of one agent over the other N-1 agents of its env (positions staged in LDS).
//  * obs part (AG:44-64 as used by MUW:75-95): the two nearest strictly within d_sense at FINAL
//    positions, ascending by the float32 distance, ties -> lower index (agents are visited in
//    ascending index order and only a strictly smaller distance displaces an entry);
//  * STEP part (MUW:198-210): the reference tests the nearest in-range agents against the thresholds
//    2R and 1.0, so only the MINIMUM in-range distance under the Gauss-Seidel rule (j<i moved, j>i
//    not yet) matters.  sqrtf is monotone, so every threshold test is done on the squared distance
//    fl(dx*dx)+fl(dy*dy) against a host-computed exact float32 limit (no sqrt on this part).
struct Neigh {
    float d1, d2;
    int j1, j2;
    float step_sq_min;
};

template <int NT, bool STEP, class LDS>
__device__ __forceinline__ Neigh scan_neighbours_exact(float sq_sense, const LaneMap &m, const LDS &lds, float nx,
                                                       float ny) {
    const int N = NT ? NT : m.nslots;
    Neigh r;
    r.d1 = r.d2 = INFINITY;
    r.j1 = r.j2 = -1;
    r.step_sq_min = INFINITY;
    const float4 *row = &lds.pos[m.rbase];
    // Branch-free: out-of-range agents enter the insertion with distance +inf, which never displaces.
    auto visit = [&](int j, float4 q) {
        const float dxn = q.z - nx, dyn = q.w - ny;  // target_agent.location - self.location (AG:51)
        const float ax = dxn * dxn, ay = dyn * dyn;
        const float sn = ax + ay;
        if (STEP) {
            const float dxo = q.x - nx, dyo = q.y - ny;
            const float bx = dxo * dxo, by = dyo * dyo;
            const float so = bx + by;
            const float ss = (j < m.i) ? sn : so;      // j<i already moved this step, j>i not yet
            r.step_sq_min = fminf(r.step_sq_min, (ss < sq_sense) ? ss : INFINITY);
        }
        const float dn = (sn < sq_sense) ? sqrt_rn(sn) : INFINITY;  // AG:51-52 (IEEE-rounded sqrt)
        const bool lt1 = dn < r.d1, lt2 = dn < r.d2;
        r.d2 = lt1 ? r.d1 : (lt2 ? dn : r.d2);
        r.j2 = lt1 ? r.j1 : (lt2 ? j : r.j2);
        r.d1 = lt1 ? dn : r.d1;
        r.j1 = lt1 ? j : r.j1;
    };
    if (NT) {
        // all N-1 LDS reads are issued before the first use (one lgkmcnt wait instead of N-1)
        constexpr int M = NT > 1 ? NT - 1 : 1;
        float4 q[M];
        int js[M];
#pragma unroll
        for (int k = 0; k < NT - 1; k++) {
            js[k] = k + (k >= m.i ? 1 : 0);  // ascending over the other agents, self skipped
            q[k] = row[js[k]];
        }
#pragma unroll
        for (int k = 0; k < NT - 1; k++) visit(js[k], q[k]);
    } else {
        for (int k = 0; k < N - 1; k++) {
            const int j = k + (k >= m.i ? 1 : 0);
            visit(j, row[j]);
        }
    }
    return r;
}

// Same result for N > 5 (and the N = 8 specialisation) at about half the per-neighbour cost: the scan keeps the
// three smallest KEYS, key = (bits of the squared distance with the low 6 bits replaced by a name of the neighbour), with
// one v_min_u32 + two v_med3_u32 per neighbour instead of a compare/select insertion of (distance, index)
// pairs, and takes square roots only of the two winners (their exact squared distances are recomputed from LDS).
// Non-negative floats order like their bit patterns and NaN / +inf patterns sort above every finite value, so
// keys order by (squared distance truncated to 2^-17 relative, name).  sqrtf is monotone and two squared
// distances whose truncations differ by two or more steps have different float32 roots, so the order by key
// equals the reference's order by (float32 distance, index) (AG:52-62) unless two of the kept keys are within
// one truncation step of each other at or below the sensing limit -- then (about 1 wave in 1000 on random
// layouts; always on symmetric ones like reset(circular=True)) the wave falls back to the exact scan.  An agent
// outside the kept three can only belong in the top two if the third key is within a step of the second, which
// is one of the fallback conditions.  Bit-identical to scan_neighbours_exact.  A slot that does not take part
// (parked learner, inactive body: extension) is staged at +inf: its key sorts above every real one and its
// squared distance fails every threshold.
#ifdef UAVX_STAMPS
__shared__ int g_dbg_fallback;   // diagnostic build: this workgroup took the exact scan / the finish() branch (bits 0 / 1)
#endif
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int NT, bool STEP, class LDS>
__device__ __forceinline__ Neigh scan_neighbours(float sq_sense, const LaneMap &m, const LDS &lds, float nx,
                                                 float ny) {
    if (NT != 0 && NT <= 5) return scan_neighbours_exact<NT, STEP>(sq_sense, m, lds, nx, ny);
    const int N = NT ? NT : m.nslots;
    if (NT == 0 && N <= 5) return scan_neighbours_exact<NT, STEP>(sq_sense, m, lds, nx, ny);  // <= 4 others: nothing to save
    const float4 *row = &lds.pos[m.rbase];
    uint32_t k1 = 0xffffffffu, k2 = 0xffffffffu, k3 = 0xffffffffu;
    float step_min = INFINITY;
    // The low 6 bits of a key only have to NAME the neighbour (two keys that agree above them are a near tie and take the
    // exact scan): they hold c = the neighbour's rank among the OTHER slots of the env (slot j = c + (c >= i)), which is
    // the same number in every lane -- the row address is then the lane's base + 16 c (+ 16 from agent i upwards) and the
    // key needs no per-lane index register.
    // (the mask sits in a VGPR so that the wave-uniform name can be the one scalar operand of a single v_and_or_b32)
    uint32_t keep;
    asm("v_mov_b32 %0, 0xffffffc0" : "=v"(keep));
    auto visit = [&](uint32_t c, bool below, float4 q) {   // below: slot < m.i (already moved in this step)
        const float dxn = q.z - nx, dyn = q.w - ny;
        const float ax = dxn * dxn, ay = dyn * dyn;
        const float sn = ax + ay;
        if (STEP) {
            const float dxo = q.x - nx, dyo = q.y - ny;
            const float bx = dxo * dxo, by = dyo * dyo;
            const float so = bx + by;
            step_min = fminf(step_min, below ? sn : so);  // fminf drops NaN; the d_sense test follows the loop
        }
        uint32_t key;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(sn), "v"(keep), "s"(c));
        k3 = med3_u32(k2, k3, key);
        k2 = med3_u32(k1, k2, key);
        k1 = min(k1, key);
    };
    const char *rowb = reinterpret_cast<const char *>(row);
    auto other = [&](int c, bool below) {   // the c-th other slot of the env as seen from agent i
        return *reinterpret_cast<const float4 *>(rowb + c * 16 + (below ? 0 : 16));
    };
    if (NT) {  // compile-time N: the LDS reads of a batch are issued before its first use
#ifndef UAVX_SCANBATCH
#define UAVX_SCANBATCH 7
#endif
        constexpr int M = NT > 1 ? NT - 1 : 1;
        constexpr int BATCH = M < UAVX_SCANBATCH ? M : UAVX_SCANBATCH;
#pragma unroll
        for (int c0 = 0; c0 < NT - 1; c0 += BATCH) {
            float4 q[BATCH];
#pragma unroll
            for (int c = 0; c < BATCH; c++)
                if (c0 + c < NT - 1) q[c] = other(c0 + c, c0 + c < m.i);
#pragma unroll
            for (int c = 0; c < BATCH; c++)
                if (c0 + c < NT - 1) visit((uint32_t)(c0 + c), c0 + c < m.i, q[c]);
        }
    } else {
        // two neighbours per trip, written out (inline asm is convergent in HIP, which rules out the unroll pragma)
        const int NL = m.nlearn;   // slots [0, NL) are agents with a lane each; [NL, N) scripted bodies (extension)
        int c = 0;
        for (; c + 1 < NL - 1; c += 2) {
            const bool la = c < m.i, lb = c + 1 < m.i;
            const float4 qa = other(c, la), qb = other(c + 1, lb);
            visit((uint32_t)c, la, qa);
            visit((uint32_t)c + 1u, lb, qb);
        }
        if (c < NL - 1) {
            const bool la = c < m.i;
            visit((uint32_t)c, la, other(c, la));
            c++;
        }
        // bodies sit above every learner and have moved before any of them: ONE squared distance serves the collision test
        // and the observation, and the Gauss-Seidel select folds away (rows {x, y, x, y}: 8-byte reads of the upper half)
        auto body = [&](uint32_t c, int r) {
            const float2 q = *reinterpret_cast<const float2 *>(&row[r].z);
            const float dxn = q.x - nx, dyn = q.y - ny;
            const float ax = dxn * dxn, ay = dyn * dyn;
            const float sn = ax + ay;
            if (STEP) step_min = fminf(step_min, sn);
            uint32_t key;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(sn), "v"(keep), "s"(c));
            k3 = med3_u32(k2, k3, key);
            k2 = med3_u32(k1, k2, key);
            k1 = min(k1, key);
        };
        for (; c + 1 < N - 1; c += 2) {
            body((uint32_t)c, c + 1);
            body((uint32_t)c + 1u, c + 2);
        }
        if (c < N - 1) body((uint32_t)c, c + 1);
    }
    // N > 5: at least five neighbours were visited, so k1..k3 are real keys
    const uint32_t t1 = k1 >> 6, t2 = k2 >> 6, t3 = k3 >> 6, ts = __float_as_uint(sq_sense) >> 6;
    const bool near_tie = m.active && ((t2 - t1 <= 1u && t1 <= ts) || (t3 - t2 <= 1u && t2 <= ts));
#ifdef UAVX_STAMPS
    if (__any(near_tie)) g_dbg_fallback = 1;
#endif
    const int c1 = (int)(k1 & 63u), c2 = (int)(k2 & 63u);
    const int j1 = c1 + (c1 >= m.i ? 1 : 0), j2 = c2 + (c2 >= m.i ? 1 : 0);
    const float4 q1 = row[j1], q2 = row[j2];
    const float ex1 = q1.z - nx, ey1 = q1.w - ny, ex2 = q2.z - nx, ey2 = q2.w - ny;
    const float mx1 = ex1 * ex1, my1 = ey1 * ey1, mx2 = ex2 * ex2, my2 = ey2 * ey2;
    const float s1 = mx1 + my1, s2 = mx2 + my2;
    const bool in1 = s1 < sq_sense, in2 = s2 < sq_sense;  // AG:52
    Neigh r;
    r.step_sq_min = (step_min < sq_sense) ? step_min : INFINITY;
    r.d1 = in1 ? sqrt_rn(s1) : INFINITY;
    r.d2 = in2 ? sqrt_rn(s2) : INFINITY;
    r.j1 = in1 ? j1 : -1;
    r.j2 = in2 ? j2 : -1;
    // Near ties (about one wavefront in 800 on random layouts; every wavefront of a symmetric one): the two nearest of a TIED
    // LANE are found again, exactly, by the whole wavefront -- lane t takes the tied agent's t-th neighbour, float32 distance
    // with the IEEE root, and two 64-bit minimum reductions over (distance bits, slot) give the reference's order (AG:52-62:
    // ascending distance, ties -> lower index).  About 150 instructions per tied lane.  Round 2 sent the whole wavefront
    // through the compare / select scan instead (+700 instructions, +43 % on the wavefront's life): with ~10 such wavefronts
    // in every 65 536-env launch those were the ones each launch ended with (tools/exp_stamps.py).  The minimum for the
    // collision tests (step_sq_min) is exact on the key path as it is.
    unsigned long long tied = __ballot(near_tie);
    if (__popcll(tied) > 6) return scan_neighbours_exact<NT, STEP>(sq_sense, m, lds, nx, ny);   // a symmetric layout: everybody ties
    while (tied != 0ull) {   // wave-uniform
        const int tl = (int)__builtin_ctzll(tied);
        tied &= tied - 1ull;
        const float ax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nx), tl));
        const float ay = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ny), tl));
        const float lim = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sq_sense), tl));
        const int ti = __builtin_amdgcn_readlane(m.i, tl), trb = __builtin_amdgcn_readlane(m.rbase, tl);
        const int t = (int)(threadIdx.x & (kWave - 1));
        const bool mine = t < N - 1;
        const int j = mine ? t + (t >= ti ? 1 : 0) : ti;                     // idle lanes look at the agent itself (distance 0, masked out)
        const float4 q = lds.pos[trb + j];
        const float dx = q.z - ax, dy = q.w - ay;
        const float xx = dx * dx, yy = dy * dy;
        const float sn = xx + yy;
        const float dn = sqrt_rn(sn);                                       // (wave-uniform inside: every lane calls it)
        unsigned long long key = (mine && sn < lim) ? ((unsigned long long)__float_as_uint(dn) << 32) | (uint32_t)j : ~0ull;   // AG:52
        auto wave_min = [](unsigned long long k) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t hi = __shfl_xor((uint32_t)(k >> 32), off), lo = __shfl_xor((uint32_t)k, off);
                const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                k = o < k ? o : k;
            }
            return k;
        };
        const unsigned long long m1 = wave_min(key);
        const unsigned long long m2 = wave_min(key == m1 ? ~0ull : key);
        if (t == tl) {
            r.d1 = m1 == ~0ull ? INFINITY : __uint_as_float((uint32_t)(m1 >> 32));
            r.j1 = m1 == ~0ull ? -1 : (int)(uint32_t)m1;
            r.d2 = m2 == ~0ull ? INFINITY : __uint_as_float((uint32_t)(m2 >> 32));
            r.j2 = m2 == ~0ull ? -1 : (int)(uint32_t)m2;
        }
    }
    return r;
}

// MUW:60-109 in float32 (angles compared on the circle; see DESIGN.md numerics).
template <class LDS>
__device__ __forceinline__ void assemble_obs(const MultiParams &p, const WorldLims &w, const LaneMap &m, const LDS &lds,
                                             const Neigh &nb, float nx, float ny, float speed, float theta, float dist_t,
                                             float dth, float o[10]) {
    o[0] = speed * p.inv_vmax_norm;  // MUW:62
    o[1] = theta * kInvPi;           // MUW:64
    o[2] = dist_t * w.inv_diag;      // MUW:68
    o[3] = dth * kInvPi;             // MUW:72
    // absent neighbour: d=1, bearing (pi + theta) - theta wraps to +-pi -> +-1 (one point on the circle), heading 0
    const bool has1 = nb.j1 >= 0, has2 = nb.j2 >= 0;
    const int i1 = m.rbase + (has1 ? nb.j1 : 0), i2 = m.rbase + (has2 ? nb.j2 : 0);
    const float4 q1 = lds.pos[i1], q2 = lds.pos[i2];
    const float t1 = lds.theta[i1], t2 = lds.theta[i2];
    const float b1 = wrap_pi(atan2_fast(q1.w - ny, q1.z - nx) - theta) * kInvPi;  // MUW:78-81
    const float b2 = wrap_pi(atan2_fast(q2.w - ny, q2.z - nx) - theta) * kInvPi;  // MUW:88-91
    const float h1 = wrap_pi(t1 - theta) * kInvPi;                                // MUW:82-85
    const float h2 = wrap_pi(t2 - theta) * kInvPi;                                // MUW:92-95
    o[4] = has1 ? nb.d1 * w.inv_sense : 1.f;                                      // MUW:77
    o[5] = has1 ? b1 : 1.f;
    o[6] = has1 ? h1 : 0.f;
    o[7] = has2 ? nb.d2 * w.inv_sense : 1.f;                                      // MUW:87
    o[8] = has2 ? b2 : 1.f;
    o[9] = has2 ? h2 : 0.f;
}

// Wave-cooperative store of the wave's contiguous obs block (cnt*40 B starting at slot a0): the
// lane-major [64][10] tile is staged in LDS and written back with lane-contiguous vector stores.
// Even N: a0 and cnt are even, so the block is 16-byte aligned and a whole number of float4
// (uavx_create/step check the 16-byte alignment of the caller's obs pointer); otherwise float2.
template <int NT, class LDS>
__device__ __forceinline__ void store_obs_block(const MultiParams &p, const LaneMap &m, LDS &lds, const float o[10],
                                                float *obs_out) {
    constexpr int T = kWave * LDS::kW;
    float *stage = lds.obs + m.obs0;
    if (m.active) {
        float2 *dst = reinterpret_cast<float2 *>(stage + m.lane * UAVX_OBS_DIM);
#pragma unroll
        for (int k = 0; k < 5; k++) dst[k] = make_float2(o[2 * k], o[2 * k + 1]);
    }
    group_sync<LDS::kW>();
    const int nfloat = m.cnt * UAVX_OBS_DIM;
    const rsrc_t r = make_rsrc(obs_out, (uint32_t)p.E * (uint32_t)p.N * (UAVX_OBS_DIM * 4u));
    const uint32_t gbase = m.a0 * (UAVX_OBS_DIM * 4u);  // byte offset of the workgroup's block
    if (NT ? (NT % 2 == 0) : ((p.N & 1) == 0)) {   // uniform: even N => 16-byte aligned block of whole float4
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int f = (k * T + m.lane) * 4;
            if (f < nfloat) store16_wt(r, gbase + f * 4u, *reinterpret_cast<const float4 *>(stage + f));
        }
    } else {
        // odd N: the block starts 8 bytes off a 16-byte boundary in every second workgroup and ends likewise.  16-byte
        // stores for the aligned middle, one 8-byte store for a misaligned head / tail (8-byte write-through stores run at
        // 0.54-0.70x the 16-byte rate: round 1 wrote the whole block that way)
        const int head = (gbase & 8u) ? 2 : 0;          // uniform over the workgroup
        const int mid = (nfloat - head) / 4;              // float4 count
        const int tail = head + mid * 4;                  // first float after the middle (nfloat - tail is 0 or 2)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int q = k * T + m.lane;
            if (q < mid) {
                const float *src = stage + head + q * 4;   // only 8-byte aligned in LDS: two ds_read_b64
                const float2 lo = *reinterpret_cast<const float2 *>(src), hi = *reinterpret_cast<const float2 *>(src + 2);
                store16_wt(r, gbase + (uint32_t)(head + q * 4) * 4u, make_float4(lo.x, lo.y, hi.x, hi.y));
            }
        }
        if (m.lane == 0 && head) store8_wt(r, gbase, *reinterpret_cast<const float2 *>(stage));
        if (m.lane == 1 && tail < nfloat) store8_wt(r, gbase + (uint32_t)tail * 4u, *reinterpret_cast<const float2 *>(stage + tail));
    }
    group_sync<LDS::kW>();
}

// configs[4] extension: a body starts a leg at (x, y) towards waypoint (wx, wy) -- include/uavx.h, uavx_set_body_rule; float32,
// no FMA, IEEE division and square root, restated bit for bit by the oracle (body_leg).  Off the per-step path (reset, and one
// env step in `period`).
__device__ __forceinline__ float4 make_leg(float body_step, float x, float y, float wx, float wy) {
    const float dx = wx - x, dy = wy - y;
    const float d = norm32(dx, dy);
    float4 leg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d > 0.f) {
        const float sc = body_step / d;
        leg.x = dx * sc; leg.y = dy * sc;
        leg.w = floorf(d / body_step);   // body_step == 0 (static obstacle): +inf legs of zero displacement
    }
    leg.z = atan2_exact(dy, dx);
    return leg;
}

// configs[4] extension: the scripted bodies of this lane's env.  Body b is handled by the env's learner lane b % L in trip
// b / L: position (8 B) and leg record (16 B) loaded, moved when MOVE -- the bodies move BEFORE the learners of the env's
// sequential loop, so the collision tests and the observations of this step both see the new positions --, staged into its
// neighbour row {x, y, x, y} + heading, position stored back (8 B) if it moved.  A body that does not take part (b >= the
// level's b_active) is staged at +inf.
//   ready   the env was re-initialised by this call: its bodies' rows were staged by the reset path, they do not move.
template <bool MOVE, class LDS, bool LATEW = false>
__device__ __forceinline__ void stage_bodies(const MultiParams &p, const LaneMap &m, LDS &lds, uint32_t flags, bool ready,
                                             uint32_t steps, uint32_t ep_draw) {
    const int L = p.N;
    const bool leveled = p.n_levels > 0;
    const LevelParams *lv = &p.levels[(flags & kLevelMask) >> kLevelShift];   // read only when a curriculum is installed
    const int b_active = m.active ? (leveled ? lv->b_active : p.B) : 0;
    const uint32_t kk = steps & (uint32_t)p.body_pmask;     // steps of the current leg that lie behind the body
    const bool retarget = MOVE && steps != 0u && kk == 0u;  // a new waypoint every `period` steps
    const float kf = (float)kk;
#pragma unroll 1
    for (int k = 0; k < p.kb; k++) {
        const int b = k * L + m.i;
        const bool valid = m.active && b < p.B && !ready;
        const bool on = valid && b < b_active;
        const uint32_t gi = m.e * (uint32_t)p.B + (uint32_t)b;
        float2 q = make_float2(INFINITY, INFINITY);
        float4 leg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) { q = p.body_pos[gi]; leg = p.body_leg[gi]; }
        if (on && retarget) {
            // (once per `period` steps: LATEW fetches what this needs from the kernel arguments here, not at the top)
            const ResetCandidates c = reset_candidates((uint64_t)LATE(LATEW, p, env_offset) + m.e, (uint32_t)(L + b),
                                                       0x80000000u | (steps >> p.body_pshift), ep_draw, LATE(LATEW, p, body_k0),
                                                       LATE(LATEW, p, body_k1), leveled ? lv->lox : LATE(LATEW, p, lox),
                                                       leveled ? lv->loy : LATE(LATEW, p, loy), leveled ? lv->hix : LATE(LATEW, p, hix),
                                                       leveled ? lv->hiy : LATE(LATEW, p, hiy));
            leg = make_leg(p.body_step, q.x, q.y, c.sx, c.sy);
            p.body_leg[gi] = leg;
        }
        if (MOVE && on && kf < leg.w) {
            q.x = q.x + leg.x; q.y = q.y + leg.y;
            p.body_pos[gi] = q;
        }
        if (valid) {
            lds.pos[m.rbase + L + b] = make_float4(q.x, q.y, q.x, q.y);
            lds.theta[m.rbase + L + b] = leg.z;
        }
    }
}

// One env step for this lane's agent (state in registers).  MUW:177-241.
//   frozen: the env was re-initialised by this call (auto-reset); the agent only observes.
//   EXT: env_steps / ep_draw = the env's step count before this step and the episode index its reset drew with
//        (scripted bodies); frozen envs had their bodies' rows staged by the reset path.
template <int NT, bool EXT, class LDS, bool LATEW = false>
__device__ __forceinline__ void step_agent(const MultiParams &p, const LaneMap &m, LDS &lds, AgentRegs &s, double ax,
                                           double ay, int evaluate, float o[10], float &rew, uint32_t &done_out,
                                           uint32_t &reach_ev, uint32_t &coll_ev, bool frozen = false,
                                           uint32_t env_steps = 0, uint32_t ep_draw = 0) {
    const float sq_sense = sense_limit<EXT>(p, s.flags);
    const bool was_done = (s.flags & UAVX_FLAG_DONE) != 0;
    const bool parked = EXT && (s.flags & kFlagInactive) != 0;  // extension: learner switched off by its env's level
    if (!frozen) s.flags &= ~(kFlagPrevOvr | kFlagJustDone);  // from here on prev_distance is the natural one again
    const float ox = s.x, oy = s.y;
    float pd = 0.f, d = 0.f;  // AG:24-25: a done agent returns (0, 0) and does not move
    if (!was_done && !frozen && !parked) {
        axis_update(ax, p.tau, p.rtau, p.recip_ok != 0, p.amax, p.vmax, s.vx, s.x);  // AG:26-29
        axis_update(ay, p.tau, p.rtau, p.recip_ok != 0, p.amax, p.vmax, s.vy, s.y);
        pd = s.prev_d;                                       // AG:32
    }
    const float tdx = s.tx - s.x, tdy = s.ty - s.y;
    const float dist_t = norm32(tdx, tdy);                   // AG:33 / MUW:67
    if (!was_done) d = dist_t;
    // heading; finish() only rescales the velocity (AG:40), so one atan2 serves reward and obs
    const float theta = atan2_fast((float)s.vy, (float)s.vx);   // MUW:63,185
    const float dth = wrap_pi(atan2_fast(tdy, tdx) - theta);    // MUW:184-186 == MUW:69-71

    if (m.active) {
        lds.pos[m.rbase + m.i] = make_float4(ox, oy, s.x, s.y);   // a parked learner sits at +inf
        lds.theta[m.rbase + m.i] = theta;
    }
    if (EXT && p.B > 0) stage_bodies<true, LDS, LATEW>(p, m, lds, s.flags, frozen, env_steps, ep_draw);
    group_sync<LDS::kW>();
    const Neigh nb = scan_neighbours<NT, true>(sq_sense, m, lds, s.x, s.y);
    const WorldLims w = world_lims<EXT>(p, s.flags);

    // reward shaping, MUW:188-195 (float32, reciprocals instead of divisions; |error| << 1e-5)
    const float inv_init = __builtin_amdgcn_rcpf(s.init_d);
    float r = -0.01f * fminf(p.vmax_norm * inv_init, 1.0f);  // MUW:189
    r += (50.0f * p.inv_vmax_norm) * (pd - d);               // MUW:190 (pd - d is a float32 subtraction there too)
    const float frac = d * inv_init * (1.0f / 1.5f);         // MUW:192,194
    r *= (r > 0.f) ? (1.0f - frac) : (1.0f + frac);
    r -= 0.01f * fabsf(dth);                                 // MUW:195

    // collisions, MUW:197-210 (exact threshold tests on the squared distance)
    const bool collision = nb.step_sq_min <= w.sq_two_r;     // MUW:203  dist <= 2R
    if (collision) r = -2.0f;                                // MUW:204
    coll_ev = 0;
    if (nb.step_sq_min <= p.sq_hard && !(s.flags & (UAVX_FLAG_DONE | UAVX_FLAG_COLLIDED)) && !frozen) {  // MUW:207-208
        coll_ev = 1;                                         // MUW:209
        s.flags |= UAVX_FLAG_COLLIDED;                       // MUW:210
    }
    // termination, MUW:213-227
    const double sq = fma(s.vy, s.vy, s.vx * s.vx);          // MUW:214 (np.linalg.norm's float64 dot)
    const bool oob = !(s.x >= w.lo_x && s.x <= w.hi_x && s.y >= w.lo_y && s.y <= w.hi_y);  // MUW:213,224 (exact float32 form)
    float speed = __builtin_amdgcn_sqrtf((float)sq);         // obs feature only
    reach_ev = 0;
    if (frozen) {
        done_out = 0;
        r = 0.f;
    } else if (d < 0.5f && !collision && sq < p.speed_sq_lim) {     // MUW:218
        done_out = 1;
        reach_ev = was_done ? 0u : 1u;                       // MUW:220-221
        s.flags |= UAVX_FLAG_DONE | (was_done ? 0u : kFlagJustDone);  // AG:39
        const double nv = sqrt(sq);                          // AG:40
        double fx = s.vx / nv * 0.001, fy = s.vy / nv * 0.001;
        if (fx != fx || fy != fy) { fx = 0.0; fy = 0.0; }    // AG:41-42
        s.vx = fx; s.vy = fy;
        speed = __builtin_amdgcn_sqrtf((float)fma(fy, fy, fx * fx));
        r += 10.0f;                                          // MUW:223
    } else if (oob) {
        done_out = evaluate ? 0u : 1u;                       // MUW:224-225
    } else {
        done_out = 0;
    }
    if (!frozen) s.prev_d = d;                               // MUW:229
    rew = r;
    assemble_obs(p, w, m, lds, nb, s.x, s.y, speed, theta, dist_t, dth, o);  // MUW:233-235
    if (parked) {  // extension: a parked learner reports an all-zero observation, no reward, done
#pragma unroll
        for (int k = 0; k < UAVX_OBS_DIM; k++) o[k] = 0.f;
        rew = 0.f;
        done_out = frozen ? 0u : 1u;
    }
}

template <bool ACT64>
__device__ __forceinline__ void load_action(const void *__restrict__ actions, uint32_t a, double &ax, double &ay) {
    if (ACT64) {
        const double2 v = reinterpret_cast<const double2 *>(actions)[a];
        ax = v.x; ay = v.y;
    } else {
        const float2 v = reinterpret_cast<const float2 *>(actions)[a];
        ax = (double)v.x; ay = (double)v.y;
    }
}

// One env step per launch (the RL loop's shape: the policy runs between two launches).
// The arguments the FIRST instructions need -- command pointer, the base of the state allocation and the 32-bit offsets of its
// arrays, the numbers the lane mapping is made of -- are LEADING SCALAR kernel arguments: gfx950 preloads those into SGPRs before
// the wavefront starts (Makefile: -mllvm -amdgpu-kernarg-preload-count), so the state loads are the first thing a wavefront does
// instead of waiting for a scalar load of the argument segment; everything else of the argument struct is fetched behind them
// (scheduling barrier).  A/B, same library, three runs each (profiles/r04_ab_notes.md section 10): 65 536 x 4 5.76 -> 5.56 us,
// x 8 10.4 -> 10.0, x 2 4.19 -> 4.09, 32 768 x 4 4.53 -> 4.35.
// (with bodies the allocator lands on 65 VGPRs = 7 wavefronts per SIMD; asking for 8 gives 62 without a spill)
#ifndef UAVX_STEPB
#define UAVX_STEPB 8
#endif
// T > 1: T independent one-wavefront tiles per workgroup (own LDS slice, wavefront-level ordering only) -- fewer workgroups for
// the dispatcher to place.  Pays only where one-wavefront workgroups fill every slot exactly once and live short (65 536 x 8:
// the 2.3 us over which 8 192 workgroups are placed is a large share of a 6 us wavefront); see tiles_for().
template <int NT, bool ACT64, bool EXT, int W, int T = 1>
__global__ __launch_bounds__(kWave * W * T, (EXT && W == 1) ? UAVX_STEPB : 1) void step_kernel(
    const void *__restrict__ actions, char *slab, uint32_t off_vel, uint32_t off_goal, uint32_t off_rec, uint32_t off_wsteps,
    uint32_t num_envs, uint32_t n_agents, uint32_t envs_per_group, uint32_t magic, uint32_t nslots, MultiParams p, int evaluate,
    float *__restrict__ obs_out, float *__restrict__ rew_out, uint8_t *__restrict__ done_out) {
    static_assert(T == 1 || W == 1, "tiles are one-wavefront workgroups side by side");
    using LDS = std::conditional_t<(T > 1), LdsTiles<T>, LdsT<EXT, W>>;
    static_assert(T == 1 || !EXT, "tiles: the plain variants only");
    __shared__ LDS lds;
    const uint32_t tile = T > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave)) : 0u;   // (a scalar)
    float2 *const pos_b = reinterpret_cast<float2 *>(slab);
    double2 *const vel_b = reinterpret_cast<double2 *>(slab + off_vel);
    Goal *const goal_b = reinterpret_cast<Goal *>(slab + off_goal);
    const uint32_t wave_id = blockIdx.x * T + tile;
    const LaneMap m = lane_map_from<NT, EXT, W>(num_envs, (int)n_agents, (int)envs_per_group, (int)magic, (int)nslots, wave_id,
                                                T > 1 ? threadIdx.x % kWave : threadIdx.x, tile);
    AgentRegs s = {};
    double ax = 0.0, ay = 0.0;
    uint4 rec = make_uint4(0, 0, 0, 0);
    uint32_t wave_count = 0;
    {
        // Unconditional (idle lanes of the last workgroup read slot 0 and drop what they compute): the requests leave in front
        // of every scalar load of the argument struct.  The command goes first: prev_distance is arithmetic on the state, and
        // a load placed behind that would start a second memory round trip after the first one has come back.
        const uint32_t el = m.active ? m.e : 0u, al = m.active ? m.a : 0u;
        if (EXT) {  // the bodies' waypoint schedule runs on the env's step count and episode index
            rec = reinterpret_cast<const uint4 *>(slab + off_rec)[el];
            wave_count = reinterpret_cast<const uint32_t *>(slab + off_wsteps)[wave_id];
        }
        load_action<ACT64>(actions, al, ax, ay);
        const float2 d = pos_b[al];
        const double2 v = vel_b[al];
        const Goal g = goal_b[al];
        __builtin_amdgcn_sched_barrier(0);
        s.x = d.x; s.y = d.y; s.vx = v.x; s.vy = v.y;
        s.tx = g.tx; s.ty = g.ty; s.init_d = g.init_d; s.flags = g.flags;
        s.prev_d = natural_prev_d(s.flags, s.x, s.y, s.tx, s.ty);
        if (m.active && (s.flags & kFlagPrevOvr)) s.prev_d = p.prev_ovr[m.a];  // rare: only after a caller poked the state
    }
    const uint32_t flags_in = s.flags;
    float o[10], rew;
    uint32_t dn, re, ce;
    step_agent<NT, EXT, LDS, false>(p, m, lds, s, ax, ay, evaluate, o, rew, dn, re, ce, false, wave_count - rec.x,
                                    ((rec.y & ~kRecEnded) - 1u) & ~kRecEnded);
    if (m.active) {
        if (!(EXT && (flags_in & kFlagInactive))) store_agent(p, pos_b, vel_b, goal_b, m.a, s, flags_in);
        rew_out[m.a] = rew;
        done_out[m.a] = (uint8_t)dn;
        if (re) atomicAdd(&p.reach[m.e], 1u);                // MUW:221
        if (ce) atomicAdd(&p.coll[m.e], 1u);                 // MUW:209
        if (!(fabsf(rew) < INFINITY)) atomicAdd(&p.nonfin[m.e], 1u);   // the tripwire of test_ddpg_multi.py:114-130, per env
        if (m.lane == 0) {
            uint32_t *const ws = reinterpret_cast<uint32_t *>(slab + off_wsteps);
            if (EXT) ws[m.wave] = wave_count + 1u;             // single writer: this wave (MUW:238)
            else atomicAdd(&ws[m.wave], 1u);                   // MUW:238 for every env of this wave (no-return)
        }
    }
    store_obs_block<NT>(p, m, lds, o, obs_out);
}

// One round of the accept/reject chain: does any agent of the workgroup clash, and which is the lowest-indexed clashing
// agent of MY env?  One wavefront per workgroup: a ballot.  Several: an LDS min per env (scratch in the obs tile, which is
// only used at the very end of a launch) and a workgroup-wide OR.  Returns false when nobody clashes (uniform).
template <class LDS, class MAP>
__device__ __forceinline__ bool lowest_clash(const MAP &m, LDS &lds, unsigned long long group, bool clash, int &low) {
    if (LDS::kW == 1) {
        const unsigned long long bits = __ballot(clash);
        const unsigned long long mine = (bits >> m.base) & group;     // clashing agents of my env
        low = mine ? (int)__builtin_ctzll(mine) : 64;
        return bits != 0ull;
    } else {
        int *slot = reinterpret_cast<int *>(lds.obs) + m.g;
        if (m.i == 0) *slot = 64;
        __syncthreads();
        if (clash) atomicMin(slot, m.i);
        const bool any = __syncthreads_or(clash ? 1 : 0) != 0;
        low = *slot;
        return any;
    }
}

// ||a - b|| <= float32(2R) on the squared distance (exact: sqrtf is monotone, limit from sq_limit_le)
__device__ __forceinline__ bool too_close(float sq_two_r, float ax, float ay, float bx, float by) {
    const float dx = ax - bx, dy = ay - by;
    const float xx = dx * dx, yy = dy * dy;
    return xx + yy <= sq_two_r;
}

// MUW:116-155 for the envs of this wave flagged `go` (all lanes of an env agree), wave-cooperative.
// Every lane draws its agent's first start/target candidates with ONE Philox call.  The reference's
// sequential accept/reject chain (agent i keeps the first candidate that is clear of the ACCEPTED points of
// agents j < i, MUW:127-153) is replayed without a per-agent turn loop: all lanes test their current
// candidate against the lower-indexed ones at once; if any clash, only the LOWEST-indexed clashing agent of
// each env redraws (everyone below it is already final, everyone above it still holds its first candidate),
// and the test repeats.  With no clash — the common case, probability ~N^2*pi*R^2/area — that is one pass for
// the start points and one for the targets; each clash costs one more pass.  (A per-agent turn loop here
// made 6 % of the resets take thousands of cycles, and with ~100 resets per launch that long tail was in
// EVERY launch: +3 us at 65 536 x 4.)  Same distribution as the reference; the stream layout is
// reset_candidates(), restated by the CPU test oracle.
// EXT (include/uavx.h, curriculum + bodies): the env first takes its level; learners >= the level's n_active are parked;
// the level's bodies then draw their start points in slot order by the same chain, trip by trip (body b belongs to
// lane b % L, trip b / L), against the learners' accepted starts and the lower-indexed bodies, and take waypoint 0.
// Body records (position + leg 0) are stored by this function, and the bodies' neighbour rows of a re-initialised env are
// left in LDS in their staged form ({x, y, x, y} + heading; +inf for a body that does not take part): stage_bodies skips them.
// bpos_out / bleg_out / lvl_out: where the body records and the env's level go -- the live arrays (p.body_pos, p.body_leg,
// p.lvl_cur), or the staging area of a pre-drawn layout (p.stage_bpos, p.stage_bleg, nullptr: the level then only travels
// in s.flags).
#ifndef UAVX_CHAINROWS
#define UAVX_CHAINROWS 4
#endif
constexpr int kChainRows = UAVX_CHAINROWS;   // rows of accepted points a clash test reads per trip (8-byte halves of the rows: the points only)
template <int NT, bool EXT, class LDS>
__device__ __forceinline__ void reset_envs_wave(const MultiParams &p, const LaneMap &m, LDS &lds, bool go,
                                                uint32_t episode, uint32_t k0, uint32_t k1, AgentRegs &s,
                                                float2 *bpos_out, float4 *bleg_out, uint8_t *lvl_out) {
    const int N = NT ? NT : p.N;
    float4 *row = &lds.pos[m.rbase];
    const uint64_t ge = (uint64_t)p.env_offset + m.e;
    const unsigned long long group = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
    double lox = p.lox, loy = p.loy, hix = p.hix, hiy = p.hiy;
    float sq2r = p.sq_two_r;
    uint32_t lvl = 0;
    int nl = N, nb = 0;
    if (EXT) {
        if (go && p.n_levels > 0) {
            if (p.level_lo >= 0) {  // randomized-reset curriculum: uniform in [lo, hi] from the env's pseudo-slot 0xFFFF
                uint32_t o[4];
                reset_words(ge, 0xFFFFu, 0u, episode, k0, k1, o);
                lvl = (uint32_t)p.level_lo + __umulhi(o[0], (uint32_t)(p.level_hi - p.level_lo + 1));
            } else {
                lvl = p.lvl_next[m.e];
            }
            lvl = min(lvl, (uint32_t)(p.n_levels - 1));
        }
        nb = p.B;
        if (p.n_levels > 0) {
            const LevelParams *lv = &p.levels[lvl];
            lox = lv->lox; loy = lv->loy; hix = lv->hix; hiy = lv->hiy;
            sq2r = lv->sq_two_r;
            nl = lv->n_active; nb = lv->b_active;
        }
    }
    const bool gl = go && m.i < nl;  // this lane's learner takes part
    ResetCandidates c = {0.f, 0.f, 0.f, 0.f};
    if (gl) {
        c = reset_candidates(ge, m.i, 0u, episode, k0, k1, lox, loy, hix, hiy);
        row[m.i] = make_float4(c.sx, c.sy, c.tx, c.ty);
    }
    group_sync<LDS::kW>();
#pragma unroll 1
    for (int phase = 0; phase < 2; phase++) {  // 0: start points MUW:126-137, 1: targets MUW:140-153
        uint32_t attempt = 0;
#pragma unroll 1
        for (;;) {
            bool clash = false;
            if (gl) {
                const float qx = phase ? c.tx : c.sx, qy = phase ? c.ty : c.sy;
                clash = phase ? too_close(sq2r, qx, qy, c.sx, c.sy) : false;                       // MUW:146
#pragma unroll 1
                for (int j0 = 0; j0 < m.i; j0 += kChainRows) {  // several rows per trip (LDS round trips bound this loop)
                    float2 o[kChainRows];
#pragma unroll
                    for (int u = 0; u < kChainRows; u++) {
                        const float4 *r4 = &row[min(j0 + u, m.i - 1)];
                        o[u] = *reinterpret_cast<const float2 *>(phase ? &r4->z : &r4->x);
                    }
#pragma unroll
                    for (int u = 0; u < kChainRows; u++) clash = clash || too_close(sq2r, o[u].x, o[u].y, qx, qy);  // MUW:135,151
                }
            }
            int low;   // lowest-indexed clashing agent of my env (>= N: none)
            if (!lowest_clash<LDS>(m, lds, group, clash, low)) break;
            const bool redraw = gl && m.i == low;
            group_sync<LDS::kW>();
            if (redraw) {  // the lowest-indexed clashing agent takes its next candidate
                const ResetCandidates r = reset_candidates(ge, m.i, ++attempt, episode, k0, k1, lox, loy, hix, hiy);
                if (phase) { c.tx = r.tx; c.ty = r.ty; row[m.i].z = c.tx; row[m.i].w = c.ty; }
                else { c.sx = r.sx; c.sy = r.sy; row[m.i].x = c.sx; row[m.i].y = c.sy; }
            }
            group_sync<LDS::kW>();
        }
    }
    if (EXT) {
#pragma unroll 1
        for (int k = 0; k < p.kb; k++) {
            const int b = k * N + m.i;
            const bool on = go && b < nb;
            const int slot = N + b;
            uint32_t attempt = 0;
            float qx = 0.f, qy = 0.f;
            if (on) {
                const ResetCandidates r = reset_candidates(ge, (uint32_t)slot, 0u, episode, k0, k1, lox, loy, hix, hiy);
                qx = r.sx; qy = r.sy;
                row[slot].x = qx; row[slot].y = qy;
            }
            group_sync<LDS::kW>();
#pragma unroll 1
            for (;;) {
                bool clash = false;
                if (on) {
                    // against the learners' accepted start points (rows 0..nl-1) and the lower-indexed bodies (rows
                    // N..N+b-1), four rows per trip: the loop is bound by LDS round trips, not by arithmetic, and the
                    // slowest resetting wave of a launch is what the whole launch waits for
                    const int cnt = nl + b;
#pragma unroll 1
                    for (int j0 = 0; j0 < cnt; j0 += kChainRows) {
                        float2 o[kChainRows];
#pragma unroll
                        for (int u = 0; u < kChainRows; u++) {
                            const int j = min(j0 + u, cnt - 1);
                            o[u] = *reinterpret_cast<const float2 *>(&row[j < nl ? j : N + (j - nl)].x);
                        }
#pragma unroll
                        for (int u = 0; u < kChainRows; u++) clash = clash || too_close(sq2r, o[u].x, o[u].y, qx, qy);
                    }
                }
                int low;
                if (!lowest_clash<LDS>(m, lds, group, clash, low)) break;
                const bool redraw = on && m.i == low;
                group_sync<LDS::kW>();
                if (redraw) {
                    const ResetCandidates r = reset_candidates(ge, (uint32_t)slot, ++attempt, episode, k0, k1, lox, loy, hix, hiy);
                    qx = r.sx; qy = r.sy;
                    row[slot].x = qx; row[slot].y = qy;
                }
                group_sync<LDS::kW>();
            }
            if (go && b < p.B) {
                float2 q = make_float2(INFINITY, INFINITY);    // a body that does not take part
                float4 leg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on) {
                    const ResetCandidates w0 = reset_candidates(ge, (uint32_t)slot, 0x80000000u, episode & ~kRecEnded, p.body_k0,
                                                                p.body_k1, lox, loy, hix, hiy);
                    q = make_float2(qx, qy);
                    leg = make_leg(p.body_step, qx, qy, w0.sx, w0.sy);   // leg 0: towards waypoint 0
                }
                row[slot] = make_float4(q.x, q.y, q.x, q.y);
                lds.theta[m.rbase + slot] = leg.z;
                bpos_out[m.e * (uint32_t)p.B + (uint32_t)b] = q;
                bleg_out[m.e * (uint32_t)p.B + (uint32_t)b] = leg;
            }
        }
    }
    group_sync<LDS::kW>();
    if (go) {
        s.x = c.sx; s.y = c.sy; s.tx = c.tx; s.ty = c.ty;
        s.init_d = s.prev_d = norm32(c.tx - c.sx, c.ty - c.sy);  // MUW:154-155
        s.vx = 0.0; s.vy = 0.0; s.flags = 0;                      // MUW:120-123
        if (EXT) {
            s.flags = lvl << kLevelShift;
            if (!gl) {  // parked learner: never a neighbour (+inf), reports obs 0 / reward 0 / done 1
                s.x = s.y = INFINITY; s.tx = s.ty = 0.f;
                s.init_d = s.prev_d = INFINITY;
                s.flags |= kFlagInactive;
            }
            if (m.i == 0 && lvl_out) lvl_out[m.e] = (uint8_t)lvl;
        }
    }
}

// What a parked layout must have been drawn for to serve env e's next reset: {episode index, seed, level rule | world
// version | valid}.  With an installed curriculum and the random window off the level is the one assigned to the env.
template <bool EXT, class P>   // P: MultiParams, or the same struct seen through the laundered kernel-argument pointer
__device__ __forceinline__ uint4 stage_want(const P &p, uint32_t e, uint32_t episode, uint32_t k0, uint32_t k1) {
    uint32_t lvl = 0xFFu;   // "drawn by the layout itself" (random window) or no curriculum
    if (EXT && p.n_levels > 0 && p.level_lo < 0) lvl = min((uint32_t)p.lvl_next[e], (uint32_t)(p.n_levels - 1));
    return make_uint4(episode, k0, k1, kStageValid | ((p.world_version & 0x7FFFFFu) << 8) | lvl);
}
__device__ __forceinline__ bool stage_hit(uint4 have, uint4 want) {   // the level byte of `have` is the level it drew
    const bool lvl_ok = (want.w & 0xFFu) == 0xFFu || (want.w & 0xFFu) == (have.w & 0xFFu);
    return have.x == want.x && have.y == want.y && have.z == want.z && (have.w >> 8) == (want.w >> 8) && lvl_ok;
}

// An episode of env e ends (reset): fold its counters into the per-env statistics the evaluation
// loop reads (test_sac_multi.py:157,164-165) and clear the running values.  One lane per env.
struct EpisodeFold {
    uint4 c; float2 f; uint32_t reach, coll;
};
template <bool LATEF = false>
__device__ __forceinline__ EpisodeFold fold_load(const MultiParams &p, uint32_t e) {  // all loads up front: one latency
    EpisodeFold v;
    LATE_BASE(LATEF, ka);
    v.c = LATE_AT(LATEF, ka, p, fin_counts)[e]; v.f = LATE_AT(LATEF, ka, p, fin_returns)[e];
    v.reach = LATE_AT(LATEF, ka, p, reach)[e]; v.coll = LATE_AT(LATEF, ka, p, coll)[e];
    return v;
}
// An episode of env e ends (reset): fold its counters into the per-env statistics the evaluation loop reads
// (test_sac_multi.py:157,164-165) and clear them (MUW:167-168).  One lane per env; the caller rewrites env_rec.
template <bool LATEF = false>
__device__ __forceinline__ void fold_store(const MultiParams &p, uint32_t e, uint32_t steps, float2 run, EpisodeFold v) {
    LATE_BASE(LATEF, ka);
    if (steps != 0) {
        v.c.x += 1; v.c.y += steps; v.c.z += v.reach; v.c.w += v.coll;
        v.f.x += run.x; v.f.y += run.y;
        LATE_AT(LATEF, ka, p, fin_counts)[e] = v.c;
        LATE_AT(LATEF, ka, p, fin_returns)[e] = v.f;
    }
    LATE_AT(LATEF, ka, p, reach)[e] = 0; LATE_AT(LATEF, ka, p, coll)[e] = 0;  // MUW:167-168
    LATE_AT(LATEF, ka, p, nonfin)[e] = 0;
}

// test_sac_multi.py:77-80 in float32: a in [-1,1]^2 -> velocity command.
__device__ __forceinline__ void polar_to_command(const MultiParams &p, float a0, float a1, double &ax, double &ay) {
    const float v = fmaf(a0, 0.5f, 0.5f) * p.vmax_norm;
    float sn, cs;
    sincospi32(a1, sn, cs);
    ax = (double)(v * cs);
    ay = (double)(v * sn);
}

#ifdef UAVX_STAMPS
// diagnostic build (tools/exp_stamps.py): every wavefront of a uavx_step_ex launch logs {start, mid, end, kind | xcc << 8 | block << 16}
__device__ unsigned long long g_stamps[8 * 16384];
__device__ unsigned int g_stamp_n;
// (s_memtime counts per compute unit: differences inside one wavefront only; slot 6 carries the 100 MHz s_memrealtime of the
//  wavefront's first and last stamp, low words, which IS one clock for the whole device: the launch's dispatch timeline)
#define STAMP(k) do { stamps[k] = __builtin_amdgcn_s_memtime(); if ((k) == 0) stamps[6] = __builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull; } while (0)
__device__ __forceinline__ void stamp_log(unsigned long long *st, unsigned int kind) {
    if ((threadIdx.x & 63) != 0) return;
    st[6] |= (unsigned long long)__builtin_amdgcn_s_memrealtime() << 32;
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned int k = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;   // a slot per wavefront: no shared counter to queue on
    if (k < 16384) {
        for (int t = 0; t < 7; t++) g_stamps[8 * k + t] = st[t];
        g_stamps[8 * k + 7] = 0x80000000ull << 32 | kind | ((xcc & 15u) << 8) | ((unsigned long long)blockIdx.x << 16);
    }
    if (k == 0) g_stamp_n = gridDim.x * (blockDim.x / 64);
}
#else
#define STAMP(k)
#endif

// Layouts drawn ahead of time, inside the step launch.  The layout of an env's NEXT episode is a pure function of (seed,
// global env, episode index, level rule); re-initialising an env from a parked layout costs 16-byte copies, drawing it in
// place costs a serial accept / reject chain on ONE wavefront that the whole launch then waits for (with 16 scripted bodies:
// 8 us for a lucky env running alone on its SIMD, 19 us for the unluckiest of the ~100 envs that reset in a launch).  So
// pf_blocks extra workgroups of every auto-resetting uavx_step_ex launch draw instead of stepping -- in front of the
// env-workgroups or behind them (uavx_step_ex decides by the shape of the launch).  A launch is as long as its slowest
// wavefront, so what matters is how long ONE staging workgroup lives and whom it keeps waiting, not how many there are
// (per-wavefront timelines on the device-wide clock: tools/exp_stamps.py, profiles/r03_ab_notes.md):
//   * a staging workgroup alternates between two short jobs.  SCAN (no hints left from its last launch): one window of 64 W
//     envs, a lane each -- record and the tags of the env's two parked layouts, one memory round trip -- and the first few
//     envs that miss a layout are written down as HINTS {env, episode} in the workgroup's own eight slots (no atomics, nobody
//     else writes them); 2.5 us.  DRAW (the next launch finds the hints with one scalar load, a few hundred cycles): the
//     hinted layouts are drawn at once.  Until this round one workgroup scanned AND drew in the same launch: 9 us with the
//     chain waiting behind the scan's round trip at the most congested moment of the launch;
//   * a hint is one launch old, so whether the layout is still wanted and whether its slot may be written NOW is decided from
//     the env's record and the slot's tag as THIS launch finds them (the rule below) -- those two loads are requested before
//     the Philox rounds and waited for in front of the stores, the whole chain runs under them on the hinted (env, episode),
//     and a layout that fails the test is not stored.  Hints can be stale, lost or doubled: results never depend on them;
//   * an env keeps TWO parked layouts, for its next episode (index y, slot y & 1) and the one after: consuming one leaves
//     the other in place, so how soon a layout is parked again (a few launches: window rotation + one for the hint) is not
//     critical and an env draws in place only at first use, after a changed seed / world, or when two of its episodes end
//     within those few launches;
//   * DRAW maps ONE LANE PER SLOT of the neighbour model (S = L + B lanes per layout, learners and bodies alike; up to
//     min(floor(64 W / S), 8) layouts per workgroup): every slot draws its candidates in the same fused Philox loop and the
//     chain is one fixed-point iteration over all start points followed by one over the learners' targets (round 2 mapped a
//     lane per learner and walked the bodies in ceil(B / L) sequential trips, each with its own Philox calls and clash loops).
//     A full invalidation -- creation, a new seed or world -- is worked off at about pf_blocks / 2 workgroups' worth of
//     layouts per launch while the envs that need one meanwhile draw in place as before (same result either way).
// Safe against the step workgroups of the SAME launch: those read the staging arrays only of an env they re-initialise, i.e.
// one whose record carried the "ended" mark when the launch began, and only slot y & 1 of it -- and exactly that slot of
// exactly those envs is left alone here (nothing orders our stores against another workgroup's loads inside a launch); their
// other slot (episode y + 1) may be drawn at any time.  INVARIANT this rests on: a step workgroup clears the mark (its
// env_rec store, the last thing it does) only after every load it made from the staging arrays has returned -- the record's
// new "ended" bit is computed from the step's done flags, which are computed from the loaded layout, so the store cannot be
// issued earlier; a staging workgroup that sees the mark cleared (and the episode index moved on) may therefore overwrite
// the consumed slot at once.  The rule does not care WHEN in the launch the record is read, which is what lets the staging
// workgroups run behind the env-workgroups as well as in front of them.  (tests/test_gpu_ext.py steps with caps of 1 and 2
// and staging in every launch for that overlap, on both positions.)
struct StageMap {   // lane-per-slot mapping of a staging workgroup (the fields lowest_clash() reads are named as in LaneMap)
    int i, base, g, rbase;
    bool active;
    uint32_t e;
};
template <int NT, bool EXT, int W, class LDS, class P, class X>   // P / X: MultiParams / StepExtra, plain or in the kernel-argument address space
__device__ __forceinline__ void stage_ahead(const P &p, const X &x, LDS &lds, uint32_t sb) {   // sb: which staging workgroup
#ifdef UAVX_STAMPS
    unsigned long long stamps[7] = {};
    STAMP(0);
#endif
    const int L = NT ? NT : p.N;
    const int S = EXT ? p.nslots : L;                 // lanes per layout
    const int epg = min((kWave * W) / S, kHintJobs);    // layouts a staging workgroup draws at once
    uint32_t *cnt = reinterpret_cast<uint32_t *>(lds.obs);        // job list: word 0 = count, job g = {env, episode} in words 1 + 2 g, 2 + 2 g
    // ---- what did this workgroup's last scan find?  Its own hint slots, ONE scalar load (wave-uniform, through the scalar
    // cache: a few hundred cycles at a moment when a vector load queues behind the first loads of every wavefront of the launch)
    typedef uint32_t HintWords __attribute__((ext_vector_type(2 * kHintJobs)));
    HintWords hw;
    uint2 *myhints = x.hints + (size_t)sb * kHintJobs;
    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(hw) : "s"(myhints) : "memory");
    int n = 0;
#pragma unroll
    for (int k = 0; k < kHintJobs; k++) {      // uniform: scalar code, the list itself goes to LDS through lane 0
        const uint32_t he = hw[2 * k], hp = hw[2 * k + 1];
        if (k < epg && he != 0u && he <= (uint32_t)p.E) {
            if (threadIdx.x == 0) { cnt[1 + 2 * n] = he - 1u; cnt[2 + 2 * n] = hp; }
            n++;
        }
    }
    if (n == 0) {
        // ---- scan: one window of 64 W envs, a lane each; what it finds is drawn by THIS workgroup in the NEXT launch ----
        const uint32_t span = kWave * W;
        const uint32_t windows = ((uint32_t)p.E + span - 1u) / span;
        // Which window?  Staging workgroup b owns the windows b, b + pf_blocks, b + 2 pf_blocks ... and looks at one of them per
        // launch, picked by the low bits of the clock: any window is as good as any other (results never depend on what is
        // parked; nothing on the host or in device memory counts launches, so a captured graph behaves like eager calls).
        const uint32_t turns = (windows + x.pf_blocks - 1u) / x.pf_blocks;
        const uint32_t turn = __builtin_amdgcn_readfirstlane((uint32_t)(__builtin_amdgcn_s_memtime() >> 7)) % turns;
        const uint32_t se = ((sb + turn * x.pf_blocks) % windows) * span + threadIdx.x;
        uint32_t want_ep = 0;
        bool need = false;
        if (se < (uint32_t)p.E) {
            const uint32_t y = p.env_rec[se].y;
            const uint32_t ep = y & ~kRecEnded;
            const uint4 t0 = p.stage_tag[se], t1 = p.stage_tag[(uint32_t)p.E + se];
            const uint4 ta = (ep & 1u) ? t1 : t0, tb = (ep & 1u) ? t0 : t1;      // tags of the slots of episodes ep / ep + 1
            const bool miss_a = !(y & kRecEnded) && !stage_hit(ta, stage_want<EXT>(p, se, ep, x.seed_lo, x.seed_hi));
            const bool miss_b = !stage_hit(tb, stage_want<EXT>(p, se, (ep + 1u) & ~kRecEnded, x.seed_lo, x.seed_hi));
            need = miss_a || miss_b;
            want_ep = miss_a ? ep : ((ep + 1u) & ~kRecEnded);
        }
        if (threadIdx.x == 0) cnt[0] = 0u;
        group_sync<W>();
        if (need) {   // the first epg of them (which ones does not matter; the rest is found again when the window comes round)
            const uint32_t k = atomicAdd(cnt, 1u);
            if ((int)k < epg) myhints[k] = make_uint2(se + 1u, want_ep);
        }
#ifdef UAVX_STAMPS
        STAMP(1); STAMP(2);
        stamp_log(stamps, __any(need) ? 11u : 10u);   // scanned: left hints / nothing to draw
#endif
        return;
    }
    if (threadIdx.x == 0) cnt[0] = (uint32_t)n;
    // the level table (at most 16 x 80 B) rides along into LDS: the chain then reads its env's box from there instead of from
    // memory (a dependent load behind the level draw)
    // (16 levels x 5 float4 = 80 rows behind the 64 rows the chain of a one-wavefront workgroup works on; EXT kernels have 192)
    constexpr int kLvlF4 = (int)(sizeof(LevelParams) / 16);
    constexpr bool kLvlLds = EXT && W == 1 && LDS::kRows >= kWave + UAVX_MAX_LEVELS * kLvlF4;
    static_assert(UAVX_MAX_LEVELS * kLvlF4 <= 2 * kWave, "the level table is two rows per lane");
    // requested here, put into LDS behind the Philox rounds (the loads' latency rides under those)
    float4 lvl_row0 = make_float4(0.f, 0.f, 0.f, 0.f), lvl_row1 = lvl_row0;
    if (kLvlLds && p.n_levels > 0) {
        const int last = p.n_levels * kLvlF4 - 1;
        lvl_row0 = reinterpret_cast<const float4 *>(p.levels)[min((int)threadIdx.x, last)];
        lvl_row1 = reinterpret_cast<const float4 *>(p.levels)[min((int)threadIdx.x + kWave, last)];
    }
    group_sync<W>();
    // The hints are marked "taken" only BEHIND this barrier: `n` must be the same in every wavefront of the workgroup (W > 1:
    // each wavefront reads the slots with its own scalar load above), and a clear in front of the barrier could reach memory
    // before a late sibling's load -- that wavefront would see n == 0, take the scan path and leave the others alone at the
    // barriers of the chain.  NO store to the hint slots may be placed in front of this barrier.
    if ((int)threadIdx.x < kHintJobs) myhints[threadIdx.x] = make_uint2(0u, 0u);   // taken
    // ---- the chain, one lane per slot ----
    StageMap m;
    uint32_t episode = 0;
    {
        const int lane = threadIdx.x;
        const int g = (lane * p.magic_s) >> 16;       // floor(lane / S) for lane < 256 (host test)
        m.i = lane - g * S;
        m.g = g < n ? g : 0;
        m.active = g < n;
        m.base = m.active ? (g * S) & (kWave - 1) : 0;
        m.rbase = m.active ? g * S : 0;
        m.e = m.active ? cnt[1 + 2 * g] : 0u;
        episode = m.active ? cnt[2 + 2 * g] : 0u;
    }
    const bool go = m.active;
    // A hint is one launch old: is the layout still wanted, and may its slot be written NOW?  Same rule as the scan applies --
    // the env's record and the slot's tag as THIS launch finds them (see the invariant above) -- but the two loads are only
    // waited for in front of the stores: the Philox rounds, the level and the whole chain run meanwhile on the hinted
    // (env, episode), and a layout that fails the test is simply not stored.
    const uint32_t rec_y = p.env_rec[m.e].y;
    const uint4 tag_now = p.stage_tag[(episode & 1u) * (uint32_t)p.E + m.e];
    group_sync<W>();   // (the job list lives in words the chain's scratch reuses)
    STAMP(1);
    __builtin_amdgcn_s_setprio(3);   // a serial chain the launch must not end up waiting for: issue ahead of the SIMD mates
    const uint32_t k0 = x.seed_lo, k1 = x.seed_hi;
    const uint64_t ge = (uint64_t)p.env_offset + m.e;
    const unsigned long long group = (S >= 64) ? ~0ull : ((1ull << S) - 1ull);
    const bool learner = m.i < L;
    // Every Philox stream whose counter is known up front runs in ONE rolled loop (four independent multiply chains fill each
    // other's latency; called one after the other they were four loops of dependent multiplies): the slot's first and second
    // candidates (attempts 0 and 1: most layouts need a redraw somewhere, few slots need two), the env's level, and -- bodies --
    // waypoint 0.  Further attempts of a slot are drawn on demand.
    uint32_t cw[4][4];
    {
        const uint32_t e_lo = (uint32_t)ge, e_hi = (uint32_t)(ge >> 32) & 0xFFFFu;
        uint32_t st[4][4] = {{e_lo, e_hi | ((uint32_t)m.i << 16), 0u, episode},             // candidates, attempt 0
                             {e_lo, e_hi | ((uint32_t)m.i << 16), 1u, episode},             // candidates, attempt 1
                             {e_lo, e_hi | (0xFFFFu << 16), 0u, episode},                   // level of the episode (pseudo-slot 0xFFFF)
                             {e_lo, e_hi | ((uint32_t)m.i << 16), 0x80000000u, episode}};   // waypoint 0 (bodies; key = the body seed)
        uint32_t ka0 = k0, ka1 = k1, kb0 = p.body_k0, kb1 = p.body_k1;
#pragma unroll 1
        for (int r = 0; r < 10; r++) {   // Philox4x32-10, the rounds of reset_words()
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint64_t p0 = (uint64_t)0xD2511F53u * st[q][0];
                const uint64_t p1 = (uint64_t)0xCD9E8D57u * st[q][2];
                const uint32_t n0 = (uint32_t)(p1 >> 32) ^ st[q][1] ^ (q == 3 ? kb0 : ka0);
                const uint32_t n2 = (uint32_t)(p0 >> 32) ^ st[q][3] ^ (q == 3 ? kb1 : ka1);
                st[q][1] = (uint32_t)p1; st[q][3] = (uint32_t)p0; st[q][0] = n0; st[q][2] = n2;
            }
            ka0 += 0x9E3779B9u; ka1 += 0xBB67AE85u; kb0 += 0x9E3779B9u; kb1 += 0xBB67AE85u;
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int k = 0; k < 4; k++) cw[q][k] = st[q][k];
    }
    STAMP(3);
    if (kLvlLds && p.n_levels > 0) {
        const int t = (int)threadIdx.x;
        if (t < p.n_levels * kLvlF4) lds.pos[kWave + t] = lvl_row0;
        if (t + kWave < p.n_levels * kLvlF4) lds.pos[2 * kWave + t] = lvl_row1;
        group_sync<W>();
    }
    double lox = p.lox, loy = p.loy, hix = p.hix, hiy = p.hiy;
    float sq2r = p.sq_two_r;
    uint32_t lvl = 0;
    int nl = L, nb = EXT ? p.B : 0;
    if (EXT && p.n_levels > 0) {   // MUW:116 extended: the env takes its level first (every lane of the env computes the same one)
        if (go) {
            lvl = (p.level_lo >= 0) ? (uint32_t)p.level_lo + __umulhi(cw[2][0], (uint32_t)(p.level_hi - p.level_lo + 1)) : (uint32_t)p.lvl_next[m.e];
            lvl = min(lvl, (uint32_t)(p.n_levels - 1));
        }
        const LevelParams *lv = kLvlLds ? reinterpret_cast<const LevelParams *>(&lds.pos[kWave]) + lvl : &p.levels[lvl];
        lox = lv->lox; loy = lv->loy; hix = lv->hix; hiy = lv->hiy;
        sq2r = lv->sq_two_r;
        nl = lv->n_active; nb = lv->b_active;
    }
    const double bx = hix - lox, by = hiy - loy, inv32 = 1.0 / 4294967296.0;
    auto point = [&](uint32_t wx, uint32_t wy, float &px, float &py) {   // lo + (hi - lo) * U cast to float32, as reset_candidates()
        px = (float)(lox + bx * ((double)wx * inv32));
        py = (float)(loy + by * ((double)wy * inv32));
    };
    const bool part = go && (learner ? m.i < nl : m.i - L < nb);   // this lane's slot takes part in the episode
    float4 *row = &lds.pos[m.rbase];
    ResetCandidates c = {INFINITY, INFINITY, INFINITY, INFINITY};    // a slot that does not take part never clashes with anyone
    if (part) { point(cw[0][0], cw[0][1], c.sx, c.sy); point(cw[0][2], cw[0][3], c.tx, c.ty); }
    if (go) row[m.i] = make_float4(c.sx, c.sy, c.tx, c.ty);
    group_sync<LDS::kW>();
    STAMP(4);
#pragma unroll 1
    for (int phase = 0; phase < 2; phase++) {  // 0: all start points in slot order (MUW:126-137, bodies after learners), 1: targets MUW:140-153
        uint32_t attempt = 0;
        const bool mine = part && (phase == 0 || learner);
        const int below = phase ? min(m.i, L) : m.i;     // lower-indexed slots whose accepted point mine must keep clear of
        float qx = phase ? c.tx : c.sx, qy = phase ? c.ty : c.sy;
        // which lower-indexed slots this one is too close to: a bit each.  The whole row of tests is made ONCE; a redraw
        // changes one point of the env, so afterwards every lane re-tests against that point only.
        unsigned long long cm = 0ull;
        bool self = mine && phase && too_close(sq2r, qx, qy, c.sx, c.sy);                          // MUW:146
        if (mine) {
#pragma unroll 1
            for (int j0 = 0; j0 < below; j0 += kChainRows) {
                float2 o[kChainRows];
#pragma unroll
                for (int u = 0; u < kChainRows; u++) {
                    const float4 *r4 = &row[min(j0 + u, below - 1)];
                    o[u] = *reinterpret_cast<const float2 *>(phase ? &r4->z : &r4->x);
                }
#pragma unroll
                for (int u = 0; u < kChainRows; u++)
                    if (j0 + u < below && too_close(sq2r, o[u].x, o[u].y, qx, qy)) cm |= 1ull << (j0 + u);  // MUW:135,151
            }
        }
#pragma unroll 1
        for (;;) {
            int low;   // lowest-indexed clashing slot of my env (everything below it is final, everything above keeps its candidate)
            if (!lowest_clash<LDS>(m, lds, group, mine && (self || cm != 0ull), low)) break;
            const bool redraw = mine && m.i == low;
            group_sync<LDS::kW>();
            if (redraw) {
                float rx, ry;
                if (++attempt == 1u) {
                    point(phase ? cw[1][2] : cw[1][0], phase ? cw[1][3] : cw[1][1], rx, ry);
                } else {
                    const ResetCandidates r = reset_candidates(ge, (uint32_t)m.i, attempt, episode, k0, k1, lox, loy, hix, hiy);
                    rx = phase ? r.tx : r.sx; ry = phase ? r.ty : r.sy;
                }
                qx = rx; qy = ry;
                if (phase) { c.tx = rx; c.ty = ry; row[m.i].z = rx; row[m.i].w = ry; self = too_close(sq2r, rx, ry, c.sx, c.sy); }
                else { c.sx = rx; c.sy = ry; row[m.i].x = rx; row[m.i].y = ry; }
            }
            group_sync<LDS::kW>();
            if (W == 1) {
                // everybody re-tests against the ONE point of its env that moved: slots above it update that bit of theirs, the
                // slots below it answer for the redrawn slot's own row of tests (the test is symmetric) through a ballot
                const bool any_low = low < S;
                const float4 *r4 = &row[any_low ? low : 0];
                const float2 np = *reinterpret_cast<const float2 *>(phase ? &r4->z : &r4->x);
                const bool t = mine && any_low && m.i != low && too_close(sq2r, np.x, np.y, qx, qy);
                const unsigned long long bits = __ballot(t && m.i < low);
                if (any_low && m.i > low) cm = (cm & ~(1ull << low)) | ((unsigned long long)t << low);
                if (redraw) cm = (bits >> m.base) & ((1ull << low) - 1ull);
            } else if (mine) {   // an env may span two wavefronts: the full row of tests again
                cm = 0ull;
#pragma unroll 1
                for (int j0 = 0; j0 < below; j0++) {
                    const float4 *r4 = &row[j0];
                    const float2 o = *reinterpret_cast<const float2 *>(phase ? &r4->z : &r4->x);
                    if (too_close(sq2r, o.x, o.y, qx, qy)) cm |= 1ull << j0;
                }
            }
        }
    }
    STAMP(5);
    const uint32_t ep_now = rec_y & ~kRecEnded;
    const bool wanted = go && ((episode == ep_now && !(rec_y & kRecEnded)) || episode == ((ep_now + 1u) & ~kRecEnded)) &&
                        !stage_hit(tag_now, stage_want<EXT>(p, m.e, episode, k0, k1));
    if (wanted) {
        const uint32_t sl = episode & 1u;   // the slot of this episode's layout
        if (learner) {   // a parked learner sits at +inf with target 0 (what reset_envs_wave leaves in its record)
            p.stage_agent[(sl * (uint32_t)p.E + m.e) * (uint32_t)L + (uint32_t)m.i] =
                part ? make_float4(c.sx, c.sy, c.tx, c.ty) : make_float4(INFINITY, INFINITY, 0.f, 0.f);
        } else if (EXT) {
            float2 q = make_float2(INFINITY, INFINITY);    // a body that does not take part
            float4 leg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (part) {
                float wx, wy;
                point(cw[3][0], cw[3][1], wx, wy);
                q = make_float2(c.sx, c.sy);
                leg = make_leg(p.body_step, c.sx, c.sy, wx, wy);   // leg 0: towards waypoint 0
            }
            const uint32_t gi = (sl * (uint32_t)p.E + m.e) * (uint32_t)p.B + (uint32_t)(m.i - L);
            p.stage_bpos[gi] = q;
            p.stage_bleg[gi] = leg;
        }
    }
    // the tag goes last, behind every store of the layout it vouches for (it is read by a LATER launch, across a kernel
    // boundary; the order only matters for whoever inspects the arrays while this launch runs: nobody does)
    group_sync<W>();
    if (wanted && m.i == 0) {
        uint4 tag = stage_want<EXT>(p, m.e, episode, k0, k1);
        tag.w = (tag.w & ~0xFFu) | (EXT ? lvl : 0u);   // the level it drew
        p.stage_tag[(episode & 1u) * (uint32_t)p.E + m.e] = tag;
    }
#ifdef UAVX_STAMPS
    STAMP(2);
    stamp_log(stamps, 13u);   // drew layouts
#endif
}

// uavx_step_ex: the step launch plus the trainer loop's bookkeeping (polar action conversion,
// episode returns, next-step auto-reset).  Same step_agent body as step_kernel.
// Register budget (profiles/r02_ab_notes.md): one agent record instead of three and the statistics fold read back at the end
// took the variant with bodies from 83 to 72 VGPRs and the N = 8 one from 89 to 79; with the staging path (stage_ahead) in the
// same kernel the variant with bodies is bounded at 6 wavefronts per SIMD (74 VGPRs, no spill; 7 = 72 VGPRs with scratch
// reloads in the hot path: 22.3 -> 23.8 us).  The same kind of bound on the N = 8 variant spills in its hot path, not applied.
#ifndef UAVX_EXB
#define UAVX_EXB 8
#endif
#ifndef UAVX_EX8B
#define UAVX_EX8B 8    // the 8-UAV specialisation: 65 536 x 8 is exactly 8 wavefronts per SIMD
#endif
template <int NT, bool ACT64, bool EXT, int W, int T = 1>   // T: one-wavefront tiles per workgroup (see step_kernel)
__global__ __launch_bounds__(kWave * W * T, (W != 1) ? 1 : (EXT ? UAVX_EXB : (NT == 8 ? UAVX_EX8B : 1))) void step_ex_kernel(const void *__restrict__ actions, char *slab, uint32_t off_vel, uint32_t off_goal,
                                                            uint32_t off_rec, uint32_t off_wsteps, uint32_t num_envs, uint32_t stage_first,
                                                            uint32_t pf_blocks, uint32_t step_first, uint32_t shape_packed, uint32_t magic,
                                                            MultiParams p, StepExtra x, int evaluate,
                                                            float *__restrict__ obs_out, float *__restrict__ rew_out_arg,
                                                            uint8_t *__restrict__ done_out_arg) {
    // The first twelve parameters (kExLead = 56 bytes: all 14 dwords the preload takes) are LEADING SCALARS -- shape_packed =
    // agents | envs per workgroup << 8 | neighbour slots per env << 16 --: gfx950 preloads them into SGPRs before the wavefront
    // starts (see step_kernel), so the staging / step decision and the first loads -- env record, step counter, command,
    // state -- need no scalar load of the argument segment.  The state arrays are one allocation: its base + 32-bit offsets
    // (uavx_create checks they fit) instead of five pointers; the END of the kernel stores through the same registers, so the
    // register-tight variants no longer fetch those pointers a second time.
    static_assert(kExLead == 2 * sizeof(void *) + 10 * sizeof(uint32_t), "leading scalar arguments of step_ex_kernel");
    static_assert(T == 1 || W == 1, "tiles are one-wavefront workgroups side by side");
    using LDS = std::conditional_t<(T > 1), LdsTiles<T>, LdsT<EXT, W>>;
    static_assert(T == 1 || !EXT, "tiles: the plain variants only");
    __shared__ LDS lds;
    const uint32_t tile = T > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave)) : 0u;   // (a scalar)
    const int N = NT ? NT : (int)(shape_packed & 0xFFu);
    float2 *const pos_b = reinterpret_cast<float2 *>(slab);
    double2 *const vel_b = reinterpret_cast<double2 *>(slab + off_vel);
    Goal *const goal_b = reinterpret_cast<Goal *>(slab + off_goal);
    uint4 *const rec_b = reinterpret_cast<uint4 *>(slab + off_rec);
    uint32_t *const wsteps_b = reinterpret_cast<uint32_t *>(slab + off_wsteps);
    // Register-tight variants (one resident round of 8 192 wavefronts needs 8 per SIMD: 64 VGPRs and 80 SGPRs): arguments that
    // only a rare branch or the END of the kernel needs are fetched there (LATE) instead of living in scalar registers across
    // the step -- bounded to 8 wavefronts per SIMD the compiler otherwise parks them in VGPR lanes (v_writelane / v_readlane
    // in the hot path: 65 536 x 8 fused 13.9 -> 15.5 us in round 3).
    constexpr bool kTight = (EXT || NT == 8) && W == 1;
    constexpr int kLate = kTight ? UAVX_LATE_EX : 0;   // LATE() sites of this variant
    constexpr bool kLateR = (kLate & 8) != 0;
    {
        // the staging workgroups of the launch: [stage_first, stage_first + pf_blocks) -- in front of the env-workgroups or
        // behind them (uavx_step_ex picks; one unsigned compare serves both)
        const uint32_t sb = blockIdx.x - stage_first;
        if (sb < pf_blocks) {   // uniform per workgroup
            if (T > 1 && tile != 0u) return;   // a staging workgroup is ONE wavefront of work: the other tiles leave
            // The staging path reads its arguments through the laundered segment pointer: left to itself the compiler hoists
            // THOSE scalar loads in front of this branch, into the prologue of every step wavefront, and with 80 SGPRs parks
            // them in VGPR lanes there (20 v_writelane at the top of step_ex_kernel<8>).
#ifndef UAVX_STAGE_LAUNDER
#define UAVX_STAGE_LAUNDER 3     // 0: off, 1: the 8-UAV specialisation, 2: also the variants with bodies / levels, 3: every variant
#endif
            // (A/B, profiles/r04_ab_notes.md: 65 536 x 8 fused 13.95 -> 13.55 us.  Every variant since the leading arguments
            //  are preloaded: the branch above is decided from registers, and hoisted staging loads + their wait in front of
            //  it would hold up the first state loads of every step wavefront again)
            if constexpr (UAVX_LATE && ((UAVX_STAGE_LAUNDER >= 1 && W == 1 && NT == 8 && !EXT) || (UAVX_STAGE_LAUNDER >= 2 && W == 1 && EXT) || UAVX_STAGE_LAUNDER >= 3)) {
                typedef const __attribute__((address_space(4))) MultiParams KP;
                typedef const __attribute__((address_space(4))) StepExtra KX;
                const karg_ptr ka = late_kargs();
                stage_ahead<NT, EXT, W>(*(KP *)(ka + kExLead), *(KX *)(ka + kExLead + sizeof(MultiParams)), lds, sb);
            } else {
                stage_ahead<NT, EXT, W>(p, x, lds, sb);
            }
            return;
        }
    }
    // (everything the mapping needs arrived in registers with the wavefront)
    const LaneMap m = lane_map_from<NT, EXT, W>(num_envs, N, (int)((shape_packed >> 8) & 0xFFu), (int)magic,
                                                (int)(shape_packed >> 16),
                                                (blockIdx.x - step_first) * T + tile,
                                                T > 1 ? threadIdx.x % kWave : threadIdx.x, tile);
#ifdef UAVX_STAMPS
    unsigned long long stamps[7] = {};
    STAMP(0);
    g_dbg_fallback = 0;
#endif
    AgentRegs s = {};
    double ax = 0.0, ay = 0.0;
    // The env record is requested FIRST and the 48 B of agent state after it: vmcnt retires in issue order, so the
    // (rare) re-initialisation below can start as soon as the small load is back and runs underneath the
    // state loads of the launch-wide read burst.  The wave's step counter comes through the scalar cache.
    // Unconditional (idle lanes of the last workgroup read slot 0 and drop what they compute): the requests leave in front of
    // every scalar load of the argument structs (scheduling barrier below).
    uint4 rec0 = make_uint4(0, 0, 0, 0);
    uint32_t wave_count;
    if constexpr (kTight) {
        // (at the 64-VGPR edge the unconditional form below costs a spill: these variants keep the loads under `active`)
        if (m.active) rec0 = rec_b[m.e];
        wave_count = wsteps_b[m.wave];
        __builtin_amdgcn_sched_barrier(0);
        if (m.active) {
            load_action<ACT64>(actions, m.a, ax, ay);
            const float2 d = pos_b[m.a];
            const double2 v = vel_b[m.a];
            const Goal g = goal_b[m.a];
            s.x = d.x; s.y = d.y; s.vx = v.x; s.vy = v.y;
            s.tx = g.tx; s.ty = g.ty; s.init_d = g.init_d; s.flags = g.flags;
            s.prev_d = natural_prev_d(s.flags, s.x, s.y, s.tx, s.ty);
            if (s.flags & kFlagPrevOvr) s.prev_d = p.prev_ovr[m.a];
        }
    } else {
        const uint32_t el = m.active ? m.e : 0u, al = m.active ? m.a : 0u;
        rec0 = rec_b[el];
        wave_count = wsteps_b[m.wave];
        // the command is requested BEFORE the state: prev_distance is arithmetic on what was loaded, and a load placed behind
        // that would start a second memory round trip after the first one has come back
        load_action<ACT64>(actions, al, ax, ay);
        const float2 d = pos_b[al];
        const double2 v = vel_b[al];
        const Goal g = goal_b[al];
        __builtin_amdgcn_sched_barrier(0);
        s.x = d.x; s.y = d.y; s.vx = v.x; s.vy = v.y;
        s.tx = g.tx; s.ty = g.ty; s.init_d = g.init_d; s.flags = g.flags;
        s.prev_d = natural_prev_d(s.flags, s.x, s.y, s.tx, s.ty);
        if (m.active && (s.flags & kFlagPrevOvr)) s.prev_d = p.prev_ovr[m.a];  // rare: only after a caller poked the state
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!m.active) rec0.y = 0u;   // an idle lane holds env 0's record: it must not take part in a re-initialisation
    const bool do_reset = (rec0.y & kRecEnded) != 0;
    const uint32_t episode = rec0.y & ~kRecEnded;
    uint32_t steps_v = wave_count - rec0.x;
    // Register budget (the variants with 8 agents / bodies sit at the 64-VGPR edge of 8 wavefronts per SIMD): a
    // re-initialised env's record is written straight over the loaded one (`s`), and what the statistics fold needs is
    // read back from `rec` and memory at the END of the launch, by the (rare) lanes that need it.
    const bool wave_resets = group_any<W>(do_reset);   // uniform over the workgroup
    if (wave_resets) {  // wave-uniform: at least one env of this wave starts a new episode
        // A wave that re-initialises an env has a few hundred more instructions to issue than its three SIMD
        // mates and would finish last (launch time = slowest wave): let it issue ahead of them for the rest
        // of its life; the mates lose only issue slots they had to spare.
        __builtin_amdgcn_s_setprio(3);
        // The layout was normally drawn ahead of time by a staging workgroup (16-byte copies); only a miss -- first use, a
        // changed seed / world, an episode that ended within two or three launches of its start -- draws here.
        // Everything the parked layout consists of is requested together with its tag, before the tag is looked at: the
        // wavefront is one memory round trip behind its mates instead of five (record -> tag -> agents -> bodies, trip by
        // trip), and a launch is as long as its slowest wavefront.
        // (what only this branch needs from the kernel arguments -- seed, staging arrays -- is fetched here, LATE())
        LATE_BASE(kLateR, ka);
        const uint32_t seed_lo = LATE_X_AT(kLateR, ka, x, seed_lo), seed_hi = LATE_X_AT(kLateR, ka, x, seed_hi);
        const uint4 *const stage_tag = LATE_AT(kLateR, ka, p, stage_tag);
        const float4 *const stage_agent = LATE_AT(kLateR, ka, p, stage_agent);
        const float2 *const stage_bpos = EXT ? LATE_AT(kLateR, ka, p, stage_bpos) : nullptr;
        const float4 *const stage_bleg = EXT ? LATE_AT(kLateR, ka, p, stage_bleg) : nullptr;
        bool hit = false;
        uint4 tag = make_uint4(0, 0, 0, 0);
        float4 st = make_float4(0.f, 0.f, 0.f, 0.f);
        float2 bq0 = make_float2(0.f, 0.f), bq1 = bq0;
        float4 bl0 = make_float4(0.f, 0.f, 0.f, 0.f), bl1 = bl0;
        // the parked layout of episode `episode` lives in slot episode & 1 of the (slot-major) staging arrays
        const uint32_t sl = episode & 1u;
        const uint32_t sbase = sl * (uint32_t)p.E * (uint32_t)p.B + m.e * (uint32_t)p.B;   // first staged body of this env
        if (do_reset && LATE_X_AT(kLateR, ka, x, use_stage)) {
            tag = stage_tag[sl * (uint32_t)p.E + m.e];
            st = stage_agent[sl * (uint32_t)p.E * (uint32_t)N + m.a];
            if (EXT) {   // the first two body trips (all of them up to B = 2 L); further ones below
                if (m.i < p.B) { bq0 = stage_bpos[sbase + (uint32_t)m.i]; bl0 = stage_bleg[sbase + (uint32_t)m.i]; }
                if (N + m.i < p.B) { bq1 = stage_bpos[sbase + (uint32_t)(N + m.i)]; bl1 = stage_bleg[sbase + (uint32_t)(N + m.i)]; }
            }
            hit = stage_hit(tag, stage_want<EXT>(p, m.e, episode, seed_lo, seed_hi));
        }
        const uint32_t hit_lvl = tag.w & 0xFFu;
        if (EXT && hit) {   // the bodies of a parked layout: staged -> live, and into the env's LDS rows (frees their registers first)
            if (m.i == 0) LATE_AT(kLateR, ka, p, lvl_cur)[m.e] = (uint8_t)hit_lvl;
            auto place = [&](int k, float2 q, float4 leg) {
                const int b = k * N + m.i;
                if (b < p.B) {
                    const uint32_t gi = m.e * (uint32_t)p.B + (uint32_t)b;
                    p.body_pos[gi] = q;
                    p.body_leg[gi] = leg;
                    lds.pos[m.rbase + N + b] = make_float4(q.x, q.y, q.x, q.y);
                    lds.theta[m.rbase + N + b] = leg.z;
                }
            };
            place(0, bq0, bl0);
            place(1, bq1, bl1);
#pragma unroll 1
            for (int k = 2; k < p.kb; k++) {
                const int b = k * N + m.i;
                const uint32_t gi = sbase + (uint32_t)min(b, p.B - 1);
                place(k, stage_bpos[gi], stage_bleg[gi]);
            }
        }
#ifndef UAVX_X_NOMISS
        if (group_any<W>(do_reset && !hit)) {
            // The accept / reject chain needs some forty registers of its own.  Inlined into the step with the loaded state
            // and command alive across it, it set the register count of the WHOLE kernel (74 with scripted bodies: 6
            // wavefronts per SIMD instead of 8).  So nothing loaded at the top of the launch is carried across it: a wavefront
            // that draws a layout in place reads the command and the state of its other envs AGAIN afterwards -- one more
            // memory round trip on that (rare) wavefront instead of 10 registers on every wavefront of every launch.
            const bool draw = do_reset && !hit;
            AgentRegs t = {};
            reset_envs_wave<NT, EXT>(p, m, lds, draw, episode, seed_lo, seed_hi, t, p.body_pos, p.body_leg, LATE_AT(kLateR, ka, p, lvl_cur));
            asm volatile("" ::: "memory");   // (the loads below must not be folded into the ones at the top)
            AgentRegs r = {};
            ax = 0.0; ay = 0.0;
            if (m.active) {   // (rare path: the state arrays through the argument struct, not the preloaded registers)
                load_action<ACT64>(actions, m.a, ax, ay);
                if (!do_reset) load_agent(p, m.a, r);
            }
            s = draw ? t : r;
        }
#endif
        if (hit) {
            s.x = st.x; s.y = st.y; s.tx = st.z; s.ty = st.w;
            s.vx = 0.0; s.vy = 0.0; s.flags = 0;                 // MUW:120-123
            // whether the learner is parked is read off the layout itself (a parked learner is staged at +inf), not off the
            // level table: a table rewritten since the layout was drawn (uavx_set_curriculum under a graph captured before
            // it, whose launches still carry the old world version and so still accept the old layouts) then cannot produce a
            // learner that takes part without a position
            bool parked = false;
            if (EXT) {
                parked = st.x == INFINITY;
                s.flags = (hit_lvl << kLevelShift) | (parked ? kFlagInactive : 0u);
            }
            s.init_d = s.prev_d = parked ? INFINITY : norm32(s.tx - s.x, s.ty - s.y);  // MUW:154-155
        }
        if (do_reset) {
            Goal *goal_w = goal_b;
            if constexpr (kLateR && UAVX_LATE) goal_w = LATE_AT(true, ka, p, goal);
            goal_w[m.a] = Goal{s.tx, s.ty, s.init_d, s.flags};
            steps_v = 0;                                           // MUW:166
        }
    }
    STAMP(1);
    const uint32_t flags_in = s.flags;
    if (x.action_mode == UAVX_ACTION_POLAR) polar_to_command(p, (float)ax, (float)ay, ax, ay);
    float o[10], rew;
    uint32_t dn, re, ce;
    // a freshly re-initialised env draws its bodies' waypoints with the episode index `episode`, a running one with the
    // index its own reset used (one less than the stored one)
    step_agent<NT, EXT, LDS, (kLate & 2) != 0>(p, m, lds, s, ax, ay, evaluate, o, rew, dn, re, ce, do_reset, steps_v,
                                               (episode - (do_reset ? 0u : 1u)) & ~kRecEnded);
    // episode end test for the NEXT call (test_sac_multi.py:67,112,116)
    bool all_done;
    if (W == 1) {
        const unsigned long long done_bits = __ballot(dn != 0);
        const unsigned long long group = (N >= 64) ? ~0ull : ((1ull << N) - 1ull);
        all_done = ((done_bits >> m.base) & group) == group;
    } else {  // the env may span two wavefronts: count its done agents in LDS (obs tile scratch, free until the final store)
        int *cnt = reinterpret_cast<int *>(lds.obs) + m.g;
        if (m.i == 0) *cnt = 0;
        __syncthreads();
        if (m.active && dn != 0) atomicAdd(cnt, 1);
        __syncthreads();
        all_done = *cnt == N;
        __syncthreads();
    }
    if (x.track_returns) {
        if (m.active) lds.theta[m.rbase + m.i] = do_reset ? 0.f : rew * (1.0f - (float)dn);  // test_sac_multi.py:157
        group_sync<LDS::kW>();
    }
    // The observation tile leaves FIRST: its ten registers per lane are free for the bookkeeping below, and the 40 B per
    // agent of write-through stores drain underneath it.  (The tile never overlaps the theta rows the score sum below reads:
    // separate arrays, or -- with scripted bodies -- the first 2 560 B of a union whose theta rows start at byte 3 072.)
    store_obs_block<NT>(p, m, lds, o, obs_out);
    float2 *pos_p = pos_b;
    double2 *vel_p = vel_b;
    Goal *goal_p = goal_b;
    uint4 *rec_p = rec_b;
    uint32_t *wsteps_p = wsteps_b;
    float *rew_out = rew_out_arg;
    uint8_t *done_out = done_out_arg;
    if constexpr (kLateR && UAVX_LATE) {
        // (register-tight variants: neither the preloaded base + offsets nor the output pointers stay alive across the step --
        //  the pointers the tail stores through are fetched from the argument struct here; A/B at 65 536 x 8 fused: holding
        //  the six preloaded registers instead cost 12.7 -> 13.2 us)
        LATE_BASE(true, kt);
        pos_p = LATE_AT(true, kt, p, pos); vel_p = LATE_AT(true, kt, p, vel); goal_p = LATE_AT(true, kt, p, goal);
        rec_p = LATE_AT(true, kt, p, env_rec); wsteps_p = LATE_AT(true, kt, p, wave_steps);
        rew_out = late_karg<float *>(kIoRew, kt); done_out = late_karg<uint8_t *>(kIoDone, kt);
    }
    if (m.active) {
        if (!(EXT && (s.flags & kFlagInactive))) store_agent(p, pos_p, vel_p, goal_p, m.a, s, flags_in);
        else if (do_reset) { pos_p[m.a] = make_float2(s.x, s.y); vel_p[m.a] = make_double2(0.0, 0.0); }  // parked at +inf
        rew_out[m.a] = rew;
        // episode end test for the NEXT call (test_sac_multi.py:67,112,116); meaningful in the env's first lane
        const uint32_t steps_next = do_reset ? 0u : steps_v + 1u;
        const bool terminal = (x.reset_policy == UAVX_RESET_AGENT0_DONE && dn != 0) ||
                              (x.reset_policy == UAVX_RESET_ALL_DONE && all_done);          // test_sac_multi.py:112,116
        const bool capped = x.step_cap != 0 && steps_next >= x.step_cap;                   // :17,67
        const bool ended = (terminal || capped) && !do_reset;
        // UAVX_FLAGS_IN_DONE: reset_mask / ended / truncated ride in bits 1..3 of the done byte of the env's agent 0 -- a
        // byte of a line this launch writes in full anyway -- instead of three more one-byte-per-env arrays, each a partial
        // line write per env (A/B at 65 536 x 4: the three byte stores are 0.28 us of a 7 us launch)
        uint32_t dbyte = dn;
        if (x.flags_in_done && m.i == 0) dbyte |= (do_reset ? 2u : 0u) | (ended ? 4u : 0u) | ((ended && !terminal) ? 8u : 0u);
        done_out[m.a] = (uint8_t)dbyte;
        if (re) atomicAdd(&LATE(kLate & 1, p, reach)[m.e], 1u);                // MUW:221
        if (ce) atomicAdd(&LATE(kLate & 1, p, coll)[m.e], 1u);                 // MUW:209
        if (!(fabsf(rew) < INFINITY)) atomicAdd(&LATE(kLate & 1, p, nonfin)[m.e], 1u);
        if (m.lane == 0) wsteps_p[m.wave] = wave_count + 1u;  // single writer: this wave (MUW:238)
        if (m.i == 0) {
            // With scripted bodies the env record is read AGAIN here by the one lane that rewrites it, instead of being
            // carried through the step in four registers of every lane (nothing has written it since the load at the top of
            // the launch): that kernel fits 64 VGPRs that way, i.e. 8 wavefronts per SIMD and ONE resident round for the
            // 8 192 wavefronts of a 65 536-env launch.  The other variants have registers to spare and keep it.
            const uint4 rec = (EXT || NT == 8) ? rec_p[m.e] : rec0;
            float2 run = do_reset ? make_float2(0.f, 0.f) : make_float2(__uint_as_float(rec.z), __uint_as_float(rec.w));
            if (!x.flags_in_done) {   // (three pointers nobody needs before this line: fetched here)
                uint8_t *const rm = LATE_X(kLateR, x, reset_mask), *const en = LATE_X(kLateR, x, ended), *const tr = LATE_X(kLateR, x, truncated);
                if (rm) rm[m.e] = do_reset ? 1 : 0;
                if (en) en[m.e] = ended ? 1 : 0;
                if (tr) tr[m.e] = (ended && !terminal) ? 1 : 0;
            }
            uint4 out = rec;
            if (do_reset) {  // fold the ended episode, start the new one: steps == 0 after this launch (MUW:166)
                fold_store<(kLate & 4) != 0>(p, m.e, wave_count - rec.x, make_float2(__uint_as_float(rec.z), __uint_as_float(rec.w)),
                                             fold_load<(kLate & 4) != 0>(p, m.e));
                out.x = wave_count + 1u;
                out.y = episode + 1u;
            }
            out.y = (out.y & ~kRecEnded) | (ended ? kRecEnded : 0u);
            if (x.track_returns) {
                float score = 0.f;
                for (int j = 0; j < N; j++) score += lds.theta[m.rbase + j];
                run.x += do_reset ? 0.f : rew;               // test_sac_multi.py:106 score += rewards[0]
                run.y += score;
            }
            out.z = __float_as_uint(run.x); out.w = __float_as_uint(run.y);
            if (out.x != rec.x || out.y != rec.y || out.z != rec.z || out.w != rec.w) rec_p[m.e] = out;
        }
    }
#ifdef UAVX_STAMPS
    STAMP(2);
    stamp_log(stamps, (wave_resets ? 1u : 0u) | (g_dbg_fallback ? 4u : 0u));
#endif
}

// K consecutive steps per launch from an action tape (open-loop rollouts): agent state stays in
// registers, only actions are read and obs/rew/done written per step.
template <int NT, bool ACT64, int W>
__global__ __launch_bounds__(kWave * W) void step_k_kernel(MultiParams p, const void *__restrict__ actions, int evaluate,
                                                           int K, int tape_out, float *__restrict__ obs_out,
                                                           float *__restrict__ rew_out, uint8_t *__restrict__ done_out) {
    using LDS = LdsT<false, W>;
    __shared__ LDS lds;
    const int N = NT ? NT : p.N;
    const LaneMap m = lane_map<NT, false, W>(p);
    AgentRegs s = {};
    if (m.active) load_agent(p, m.a, s);
    const uint32_t flags_in = s.flags;
    const size_t A = (size_t)p.E * N;
    uint32_t reach_acc = 0, coll_acc = 0, nonfin_acc = 0;
    for (int k = 0; k < K; k++) {
        double ax = 0.0, ay = 0.0;
        const size_t abytes = (ACT64 ? 16 : 8) * A * k;
        if (m.active) load_action<ACT64>(reinterpret_cast<const char *>(actions) + abytes, m.a, ax, ay);
        float o[10], rew;
        uint32_t dn, re, ce;
        step_agent<NT, false>(p, m, lds, s, ax, ay, evaluate, o, rew, dn, re, ce);
        reach_acc += re;
        coll_acc += ce;
        nonfin_acc += !(fabsf(rew) < INFINITY) ? 1u : 0u;
        if (tape_out || k == K - 1) {
            const size_t off = tape_out ? (size_t)k * A : 0;
            if (m.active) {
                (rew_out + off)[m.a] = rew;
                (done_out + off)[m.a] = (uint8_t)dn;
            }
            store_obs_block<NT>(p, m, lds, o, obs_out + off * UAVX_OBS_DIM);
        } else {
            group_sync<LDS::kW>();
        }
    }
    if (m.active) {
        store_agent(p, m.a, s, flags_in);
        if (reach_acc) atomicAdd(&p.reach[m.e], reach_acc);  // MUW:221
        if (coll_acc) atomicAdd(&p.coll[m.e], coll_acc);     // MUW:209
        if (nonfin_acc) atomicAdd(&p.nonfin[m.e], nonfin_acc);
        if (m.lane == 0) atomicAdd(&p.wave_steps[m.wave], (uint32_t)K);  // MUW:238
    }
}

template <int NT, bool EXT, int W>
__global__ __launch_bounds__(kWave * W) void observe_kernel(MultiParams p, float *__restrict__ obs_out) {
    using LDS = LdsT<EXT, W>;
    __shared__ LDS lds;
    const LaneMap m = lane_map<NT, EXT, W>(p);
    AgentRegs s = {};
    if (m.active) load_agent(p, m.a, s);
    const WorldLims w = world_lims<EXT>(p, s.flags);
    const float tdx = s.tx - s.x, tdy = s.ty - s.y;
    const float dist_t = norm32(tdx, tdy);
    const float theta = atan2_fast((float)s.vy, (float)s.vx);
    const float dth = wrap_pi(atan2_fast(tdy, tdx) - theta);
    if (EXT && p.B > 0) stage_bodies<false>(p, m, lds, s.flags, false, 0u, 0u);
    if (m.active) {
        lds.pos[m.rbase + m.i] = make_float4(s.x, s.y, s.x, s.y);
        lds.theta[m.rbase + m.i] = theta;
    }
    group_sync<LDS::kW>();
    const Neigh nb = scan_neighbours<NT, false>(w.sq_sense, m, lds, s.x, s.y);
    const float speed = __builtin_amdgcn_sqrtf((float)fma(s.vy, s.vy, s.vx * s.vx));
    float o[10];
    assemble_obs(p, w, m, lds, nb, s.x, s.y, speed, theta, dist_t, dth, o);
    if (EXT && (s.flags & kFlagInactive)) {
#pragma unroll
        for (int k = 0; k < UAVX_OBS_DIM; k++) o[k] = 0.f;
    }
    store_obs_block<NT>(p, m, lds, o, obs_out);
}

// MUW:116-168 for the masked envs, same lane-per-agent mapping and sampler as the in-step auto-reset.
template <int NT, bool EXT, int W>
__global__ __launch_bounds__(kWave * W) void reset_kernel(MultiParams p, const uint8_t *__restrict__ mask, uint64_t seed) {
    using LDS = LdsT<EXT, W>;
    __shared__ LDS lds;
    const LaneMap m = lane_map<NT, EXT, W>(p);
    const bool go = m.active && (!mask || mask[m.e] != 0);
    if (!group_any<W>(go)) return;
    AgentRegs s = {};
    uint4 rec = make_uint4(0, 0, 0, 0);
    if (go) rec = p.env_rec[m.e];
    const uint32_t episode = rec.y & ~kRecEnded;
    EpisodeFold fold = {};
    uint32_t wc = 0;
    if (go && m.i == 0) {  // statistics words requested before the draw, consumed after it
        fold = fold_load(p, m.e);
        wc = p.wave_steps[m.wave];
    }
    reset_envs_wave<NT, EXT>(p, m, lds, go, episode, (uint32_t)seed, (uint32_t)(seed >> 32), s, p.body_pos, p.body_leg, p.lvl_cur);
    if (go) {
        p.pos[m.a] = make_float2(s.x, s.y);
        p.vel[m.a] = make_double2(0.0, 0.0);
        p.goal[m.a] = Goal{s.tx, s.ty, s.init_d, s.flags};
        if (m.i == 0) {
            fold_store(p, m.e, wc - rec.x, make_float2(__uint_as_float(rec.z), __uint_as_float(rec.w)), fold);
            p.env_rec[m.e] = make_uint4(wc, episode + 1u, 0u, 0u);  // MUW:166 steps = 0, new episode, no running return
        }
    }
}

// extension plumbing: level table upload, per-env level arrays, body records
struct LevelTable { LevelParams l[UAVX_MAX_LEVELS]; };
__global__ __launch_bounds__(64) void upload_levels_kernel(LevelParams *dst, LevelTable t) {
    if (threadIdx.x < UAVX_MAX_LEVELS) dst[threadIdx.x] = t.l[threadIdx.x];
}
__global__ __launch_bounds__(kBlock) void env_levels_kernel(MultiParams p, const uint8_t *set_next, uint8_t *get_cur) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    if (set_next) p.lvl_next[e] = set_next[e];
    if (get_cur) get_cur[e] = p.lvl_cur[e];
}
// public body record (include/uavx.h): UAVX_BODY_DIM = 6 floats {x, y, dx, dy, heading, legs}
__global__ __launch_bounds__(kBlock) void bodies_kernel(MultiParams p, const float *set, float *get) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= p.E * p.B) return;
    if (set) {
        const float *r = set + i * UAVX_BODY_DIM;
        p.body_pos[i] = make_float2(r[0], r[1]);
        p.body_leg[i] = make_float4(r[2], r[3], r[4], r[5]);
    }
    if (get) {
        const float2 q = p.body_pos[i];
        const float4 leg = p.body_leg[i];
        float *r = get + i * UAVX_BODY_DIM;
        r[0] = q.x; r[1] = q.y; r[2] = leg.x; r[3] = leg.y; r[4] = leg.z; r[5] = leg.w;
    }
}

__global__ __launch_bounds__(kBlock) void episode_stats_kernel(MultiParams p, uint32_t *counts, float *returns, int clear) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    if (clear) {
        p.fin_counts[e] = make_uint4(0, 0, 0, 0);
        p.fin_returns[e] = make_float2(0.f, 0.f);
        return;
    }
    if (counts) {
        const uint4 c = p.fin_counts[e];
        counts[4 * e] = c.x; counts[4 * e + 1] = c.y; counts[4 * e + 2] = c.z; counts[4 * e + 3] = c.w;
    }
    if (returns) {
        const float2 f = p.fin_returns[e];
        returns[2 * e] = f.x; returns[2 * e + 1] = f.y;
    }
}

__global__ __launch_bounds__(kBlock) void get_state_kernel(MultiParams p, uavx_state_view v) {
    const int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t A = p.E * p.N;
    if (a < A) {
        const float2 d = p.pos[a];
        const Goal g = p.goal[a];
        if (v.loc) { v.loc[2 * a] = d.x; v.loc[2 * a + 1] = d.y; }
        if (v.prev_d) v.prev_d[a] = (g.flags & kFlagPrevOvr) ? p.prev_ovr[a] : natural_prev_d(g.flags, d.x, d.y, g.tx, g.ty);
        if (v.flags) v.flags[a] = (uint8_t)(g.flags & kFlagPublic);
        if (v.vel) { const double2 w = p.vel[a]; v.vel[2 * a] = w.x; v.vel[2 * a + 1] = w.y; }
        if (v.tgt) { v.tgt[2 * a] = g.tx; v.tgt[2 * a + 1] = g.ty; }
        if (v.init_d) v.init_d[a] = g.init_d;
    }
    if (a < p.E && v.counters) {
        const uint4 rec = p.env_rec[a];
        v.counters[4 * a + 0] = p.wave_steps[a / p.epw] - rec.x; v.counters[4 * a + 1] = p.reach[a];
        v.counters[4 * a + 2] = p.coll[a];  v.counters[4 * a + 3] = rec.y & ~kRecEnded;
    }
}

// Overwrites any subset of the UAVAgent fields.  prev_distance keeps the VALUE the reference would hold: a
// field the caller does not pass stays what it was (e.g. poking only .location leaves prev_distance stale,
// test_sac_multi_plot_trajectory.py:43-49), and whenever that value is not the one derived from the new
// (flags, location, target) it is parked in prev_ovr[] behind the PREVD_OVR bit until the next step.
__global__ __launch_bounds__(kBlock) void set_state_kernel(MultiParams p, uavx_state_view v) {
    const int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t A = p.E * p.N;
    const bool agent_fields = v.loc || v.vel || v.tgt || v.init_d || v.prev_d || v.flags;
    if (a < A && agent_fields) {
        float2 d = p.pos[a];
        Goal g = p.goal[a];
        const float old_prev = (g.flags & kFlagPrevOvr) ? p.prev_ovr[a] : natural_prev_d(g.flags, d.x, d.y, g.tx, g.ty);
        if (v.loc) { d.x = v.loc[2 * a]; d.y = v.loc[2 * a + 1]; }
        if (v.tgt) { g.tx = v.tgt[2 * a]; g.ty = v.tgt[2 * a + 1]; }
        if (v.init_d) g.init_d = v.init_d[a];
        uint32_t flags = g.flags & ~kFlagPrevOvr;
        if (v.flags) flags = ((uint32_t)v.flags[a] & kFlagPublic) | (g.flags & kLevelMask);  // a caller-set done flag is not "just finished"
        const float want = v.prev_d ? v.prev_d[a] : old_prev;
        const float nat = natural_prev_d(flags, d.x, d.y, g.tx, g.ty);
        if (__float_as_uint(want) != __float_as_uint(nat)) {
            flags |= kFlagPrevOvr;
            p.prev_ovr[a] = want;
        }
        g.flags = flags;
        p.pos[a] = d;
        p.goal[a] = g;
        if (v.vel) p.vel[a] = make_double2(v.vel[2 * a], v.vel[2 * a + 1]);
    }
    if (a < p.E && v.counters) {
        uint4 rec = p.env_rec[a];
        rec.x = p.wave_steps[a / p.epw] - v.counters[4 * a + 0];
        rec.y = (rec.y & kRecEnded) | (v.counters[4 * a + 3] & ~kRecEnded);
        p.env_rec[a] = rec;
        p.reach[a] = v.counters[4 * a + 1]; p.coll[a] = v.counters[4 * a + 2];
    }
}

#include "uavx_multi_f64.hpp"

// uavx_selftest(): sqrt_rn() against the compiler's IEEE sqrtf on every float32 bit pattern 0 ... 0x7f800000 (all
// non-negative values and +inf) plus the NaN / negative patterns of one exponent; counts differing results.
__global__ __launch_bounds__(256) void sqrt_selftest_kernel(unsigned long long *mismatches) {
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned int bad = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= 0x7f800000ull + 0x00800000ull; b += stride) {
        const float s = __uint_as_float((uint32_t)b);     // the last 2^23 patterns are NaNs
        const uint32_t got = __float_as_uint(sqrt_rn(s)), want = __float_as_uint(sqrtf(s));
        const bool both_nan = (got & 0x7fffffffu) > 0x7f800000u && (want & 0x7fffffffu) > 0x7f800000u;
        bad += (got != want && !both_nan) ? 1u : 0u;
    }
    if (bad) atomicAdd(mismatches, (unsigned long long)bad);
}

}  // namespace uavx

// ------------------------------------------------------------------------------------------------
// host side: handle + C ABI
// ------------------------------------------------------------------------------------------------
using namespace uavx;

struct uavx_handle {
    uavx_config cfg;
    MultiParams p;
    int device;
    void *slab;  // one allocation holding every state array
    size_t slab_bytes = 0;
    uint32_t off_vel = 0, off_goal = 0, off_rec = 0, off_wsteps = 0;   // byte offsets of vel / goal / env_rec / wave_steps in it (pos: 0)
    // float64-position mode (uavx_set_position_mode): arrays allocated on first use
    bool wide = false;
    WideState w = {};
    WideLimits wl = {};
    void *wide_slab = nullptr;
    // configs[4] extension: scripted bodies and / or an installed curriculum select the EXT kernel variants
    int gw = 1;  // wavefronts per workgroup of the step / reset / observe launches (pick_group_waves)
    int tiles = 1;  // one-wavefront tiles per workgroup of the step launches (tiles_for)
    // layouts drawn ahead (stage_ahead): every auto-resetting uavx_step_ex launch carries ceil(G / prefetch_every) staging
    // workgroups beside its G env-workgroups
    int prefetch_every = 16;   // 0: off
    uint2 *hints = nullptr;    // [env-workgroups + 1][kHintJobs] what each staging workgroup's last scan found (in the slab)
    int wave_slots = 8192;     // wavefronts the device holds at once (compute units x 32)
    int stage_behind = -1;     // staging workgroups behind (1) / in front of (0) the env-workgroups; -1: by launch shape.  A/B
                               // knob, read once from UAVX_STAGE_BEHIND when the handle is made; results do not depend on it
    bool ext = false;
    uavx_body_rule rule = {5.0, 128, 0, 0};
    LevelTable levels = {};
    LevelParams *levels_dev = nullptr;
    std::string err;
};

namespace {

int fail(uavx_handle *h, int code, const std::string &msg) {
    if (h) h->err = msg;
    return code;
}
int hip_fail(uavx_handle *h, hipError_t e, const char *what) {
    return fail(h, UAVX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define UAVX_HIP(h, call)                                   \
    do {                                                    \
        hipError_t e_ = (call);                             \
        if (e_ != hipSuccess) return hip_fail(h, e_, #call); \
    } while (0)

// smallest double s with sqrt(s) >= lim, so that  sqrt(s) < lim  <=>  s < result  (sqrt is
// correctly rounded and monotone): lets the device test MUW:218's speed without a float64 sqrt.
double sq_threshold(double lim) {
    double s = lim * lim;
    while (std::sqrt(s) >= lim) s = std::nextafter(s, 0.0);
    while (std::sqrt(s) < lim) s = std::nextafter(s, INFINITY);
    return s;
}

// div_tau() on the device replaces x/tau by a reciprocal + two fma; confirm on this tau that the
// form returns the IEEE quotient (differences a - v of the magnitudes the kinematics produce, plus
// the band |x| < amax*tau where the quotient is not clipped away).
}  // namespace
float uavx_f32_at_or_above(double b);
float uavx_f32_at_or_below(double b);
bool uavx_recip_division_exact(double tau) {
    const double r = 1.0 / tau;
    uint64_t s0 = 0x9E3779B97F4A7C15ull, s1 = 0xD1B54A32D192ED03ull;
    for (int i = 0; i < 100000; i++) {
        uint64_t a = s0, b = s1;
        s0 = b; a ^= a << 23; s1 = a ^ b ^ (a >> 17) ^ (b >> 26);
        const uint64_t u = s1 + b;
        double x = ((double)(u >> 11) / 9007199254740992.0) * 2.0 - 1.0;  // (-1, 1)
        x = std::ldexp(x, (i % 3 == 0) ? 5 : ((i % 3 == 1) ? -3 : -(int)(u % 60)));
        const double q0 = x * r;
        const double q = std::fma(std::fma(-q0, tau, x), r, q0);
        if (q != x / tau) return false;
    }
    return true;
}
namespace {

// float32 limits for threshold tests on squared distances (host sqrtf is correctly rounded):
// smallest s with sqrtf(s) >= lim   ->   sqrtf(s) <  lim  <=>  s <  result
float sq_limit_lt(float lim) {
    float s = lim * lim;
    while (s > 0.f && std::sqrt(s) >= lim) s = std::nextafterf(s, 0.f);
    while (std::sqrt(s) < lim) s = std::nextafterf(s, INFINITY);
    return s;
}
// largest s with sqrtf(s) <= lim    ->   sqrtf(s) <= lim  <=>  s <= result
float sq_limit_le(float lim) {
    float s = lim * lim;
    while (std::sqrt(s) <= lim) s = std::nextafterf(s, INFINITY);
    while (s > 0.f && std::sqrt(s) > lim) s = std::nextafterf(s, 0.f);
    return s;
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Kernel variant of a handle: compile-time agent count NT (1, 2, 4, 5, 8; 0 = runtime N), EXT (scripted bodies / curriculum),
// W wavefronts per workgroup (runtime-N path only).  dispatch() calls l.run<NT, EXT, W>() for the handle's variant.
template <class L>
void dispatch(const uavx_handle *h, const L &l) {
    if (h->ext) {   // bodies pin W = 1 (uavx_create); a curriculum alone keeps the mapping the agent count selected
        switch (h->gw) {
            case 2: return l.template run<0, true, 2>();
            case 3: return l.template run<0, true, 3>();
            case 4: return l.template run<0, true, 4>();
            default: return l.template run<0, true, 1>();
        }
    }
    switch (h->p.N) {
        case 1: return l.template run<1, false, 1>();
        case 2: return l.template run<2, false, 1>();
        case 4: return l.template run<4, false, 1>();
        case 5: return l.template run<5, false, 1>();   // (run_multi.py:5, test_pytorch_multi.py:27)
        case 8: return l.template run<8, false, 1>();
        default: break;
    }
    switch (h->gw) {
        case 2: return l.template run<0, false, 2>();
        case 3: return l.template run<0, false, 3>();
        case 4: return l.template run<0, false, 4>();
        default: return l.template run<0, false, 1>();
    }
}

// Tiles per workgroup of THIS launch: the handle's choice (uavx_create) as long as it runs the plain 8-UAV kernels.
inline int launch_tiles(const uavx_handle *h) { return (h->tiles == 2 && !h->ext && h->p.N == 8 && h->gw == 1) ? 2 : 1; }

struct StepLaunch {
    uavx_handle *h; dim3 grid; hipStream_t st;
    const void *actions; int action_dtype, evaluate, K, tape_out;
    float *obs, *rew; uint8_t *done;
    template <int NT, bool EXT, int W> void run() const {
        const dim3 blk(kWave * W);
        char *slab = static_cast<char *>(h->slab);
        const MultiParams &q = h->p;
        if constexpr (NT == 8 && !EXT && W == 1) {
            if (K == 1 && launch_tiles(h) == 2) {   // pairs of one-wavefront tiles (uavx_create)
                const dim3 g2(grid.x / 2), b2(kWave * 2);
                if (action_dtype == UAVX_F64)
                    hipLaunchKernelGGL((step_kernel<NT, true, EXT, W, 2>), g2, b2, 0, st, actions, slab, h->off_vel, h->off_goal, h->off_rec, h->off_wsteps,
                                       (uint32_t)q.E, (uint32_t)q.N, (uint32_t)q.epw, (uint32_t)q.magic, (uint32_t)q.nslots, q, evaluate, obs, rew, done);
                else
                    hipLaunchKernelGGL((step_kernel<NT, false, EXT, W, 2>), g2, b2, 0, st, actions, slab, h->off_vel, h->off_goal, h->off_rec, h->off_wsteps,
                                       (uint32_t)q.E, (uint32_t)q.N, (uint32_t)q.epw, (uint32_t)q.magic, (uint32_t)q.nslots, q, evaluate, obs, rew, done);
                return;
            }
        }
        if (K == 1) {
            if (action_dtype == UAVX_F64)
                hipLaunchKernelGGL((step_kernel<NT, true, EXT, W>), grid, blk, 0, st, actions, slab, h->off_vel, h->off_goal, h->off_rec, h->off_wsteps,
                                   (uint32_t)q.E, (uint32_t)q.N, (uint32_t)q.epw, (uint32_t)q.magic, (uint32_t)q.nslots, q, evaluate, obs, rew, done);
            else
                hipLaunchKernelGGL((step_kernel<NT, false, EXT, W>), grid, blk, 0, st, actions, slab, h->off_vel, h->off_goal, h->off_rec, h->off_wsteps,
                                   (uint32_t)q.E, (uint32_t)q.N, (uint32_t)q.epw, (uint32_t)q.magic, (uint32_t)q.nslots, q, evaluate, obs, rew, done);
        } else if constexpr (!EXT) {
            if (action_dtype == UAVX_F64)
                hipLaunchKernelGGL((step_k_kernel<NT, true, W>), grid, blk, 0, st, h->p, actions, evaluate, K, tape_out, obs, rew, done);
            else
                hipLaunchKernelGGL((step_k_kernel<NT, false, W>), grid, blk, 0, st, h->p, actions, evaluate, K, tape_out, obs, rew, done);
        }
    }
};

struct StepExLaunch {
    uavx_handle *h; dim3 grid; hipStream_t st; StepExtra x; const uavx_step_args *a;
    template <int NT, bool EXT, int W> void run() const {
        const dim3 blk(kWave * W);
        char *slab = static_cast<char *>(h->slab);
        const uint32_t ov = h->off_vel, og = h->off_goal, orc = h->off_rec, ow = h->off_wsteps, ne = (uint32_t)h->p.E;
        const uint32_t shape = (uint32_t)h->p.N | ((uint32_t)h->p.epw << 8) | ((uint32_t)h->p.nslots << 16);   // each <= 192
        if constexpr (NT == 8 && !EXT && W == 1) {
            if (launch_tiles(h) == 2) {   // grid / stage_first / step_first were laid out in 128-thread workgroups by uavx_step_ex
                const dim3 b2(kWave * 2);
                if (a->action_dtype == UAVX_F64)
                    hipLaunchKernelGGL((step_ex_kernel<NT, true, EXT, W, 2>), grid, b2, 0, st, a->actions, slab, ov, og, orc, ow, ne, x.stage_first,
                                       x.pf_blocks, x.step_first, shape, (uint32_t)h->p.magic, h->p, x, a->evaluate, a->obs, a->rew, a->done);
                else
                    hipLaunchKernelGGL((step_ex_kernel<NT, false, EXT, W, 2>), grid, b2, 0, st, a->actions, slab, ov, og, orc, ow, ne, x.stage_first,
                                       x.pf_blocks, x.step_first, shape, (uint32_t)h->p.magic, h->p, x, a->evaluate, a->obs, a->rew, a->done);
                return;
            }
        }
        if (a->action_dtype == UAVX_F64)
            hipLaunchKernelGGL((step_ex_kernel<NT, true, EXT, W>), grid, blk, 0, st, a->actions, slab, ov, og, orc, ow, ne, x.stage_first, x.pf_blocks,
                               x.step_first, shape, (uint32_t)h->p.magic, h->p, x, a->evaluate, a->obs, a->rew, a->done);
        else
            hipLaunchKernelGGL((step_ex_kernel<NT, false, EXT, W>), grid, blk, 0, st, a->actions, slab, ov, og, orc, ow, ne, x.stage_first, x.pf_blocks,
                               x.step_first, shape, (uint32_t)h->p.magic, h->p, x, a->evaluate, a->obs, a->rew, a->done);
    }
};

struct ObserveLaunch {
    uavx_handle *h; dim3 grid; hipStream_t st; float *obs;
    template <int NT, bool EXT, int W> void run() const {
        hipLaunchKernelGGL((observe_kernel<NT, EXT, W>), grid, dim3(kWave * W), 0, st, h->p, obs);
    }
};

struct ResetLaunch {
    uavx_handle *h; dim3 grid; hipStream_t st; const uint8_t *mask; uint64_t seed;
    template <int NT, bool EXT, int W> void run() const {
        hipLaunchKernelGGL((reset_kernel<NT, EXT, W>), grid, dim3(kWave * W), 0, st, h->p, mask, seed);
    }
};

// Wavefronts per workgroup for the runtime-N path.
// Measured (one box, bare step at 1.57 M agent slots, W = 1 / 2 / 3 / 4, profiles/r04_ab_notes.md section 12): a workgroup
// whose envs fill 64 W lanes EXACTLY (N = 3, 6, 12, 24, 48 with W = 3: every array of the workgroup's block starts and ends
// on a 64-byte sector and the obs / velocity tiles leave as whole 16-byte rows) gains 9-19 %; nearly-full pairs gain 7-10 % at
// N = 9, 10, 15, 20 and 37 % at N = 40; three wavefronts also at N = 7 and 11 (+9-10 %).  Five- and seven-wavefront
// workgroups (exact for N = 5, 10 / 7, 14) LOSE 8-30 % (not through their barriers: removing two of the three changed nothing).  Agent counts outside the table: the smallest W in 1..4 with the fewest idle lanes, if that beats one wavefront
// by more than 10 % (wider workgroups cost 0-2 % at W = 2 / 3 and 7-13 % at W = 4 where one wavefront is already aligned).
int pick_group_waves(int N) {
    if (N == 1 || N == 2 || N == 4 || N == 5 || N == 8) return 1;   // compile-time specialisations: one wavefront
    switch (N) {
        case 3: case 6: case 7: case 11: case 12: case 24: case 48: return 3;
        case 9: case 10: case 15: case 20: case 40: return 2;
        case 13: case 14: case 16: case 28: case 32: case 64: return 1;
        default: break;
    }
    const double u1 = (double)((kWave / N) * N) / kWave;
    double best = u1;
    int w = 1;
    for (int c = 2; c <= 4; c++) {
        const double u = (double)((kWave * c / N) * N) / (kWave * c);
        if (u > best + 1e-9) { best = u; w = c; }
    }
    return best > 1.10 * u1 ? w : 1;
}

// float32 forms of a float64 bound b, exact for every float32 x:  (double)x >= b <=> x >= f32_at_or_above(b),
// (double)x <= b <=> x <= f32_at_or_below(b)
float f32_at_or_above(double b) { return uavx_f32_at_or_above(b); }
float f32_at_or_below(double b) { return uavx_f32_at_or_below(b); }

// Everything of MultiParams that follows from the world's scalar parameters (MUW:13-58), shared by uavx_create and
// uavx_set_config.
void derive_world_params(const uavx_config &c, MultiParams &p) {
    const uavx_config *cfg = &c;
    p.tau = cfg->tau; p.amax = cfg->max_acceleration; p.vmax = cfg->max_speed;
    p.rtau = 1.0 / cfg->tau;
    p.recip_ok = uavx_recip_division_exact(cfg->tau) ? 1 : 0;
    p.lox = -cfg->x_size / 2.0; p.loy = -cfg->y_size / 2.0;  // MUW:19
    p.hix = cfg->x_size / 2.0; p.hiy = cfg->y_size / 2.0;    // MUW:20
    p.lo_x = f32_at_or_above(p.lox); p.lo_y = f32_at_or_above(p.loy);
    p.hi_x = f32_at_or_below(p.hix); p.hi_y = f32_at_or_below(p.hiy);
    p.speed_sq_lim = sq_threshold(0.2);
    p.two_r_reset = (float)(2 * cfg->collider_radius);
    p.sq_sense = sq_limit_lt((float)cfg->d_sense);
    p.sq_two_r = sq_limit_le(p.two_r_reset);
    p.sq_hard = sq_limit_le(1.0f);  // 2 * HARD_COLLISION_RADIUS, MUW:8,207
    p.inv_sense = 1.0f / (float)cfg->d_sense;
    p.vmax_norm = (float)std::sqrt(std::fma(cfg->max_speed, cfg->max_speed, cfg->max_speed * cfg->max_speed));
    p.inv_vmax_norm = 1.0f / p.vmax_norm;
    p.inv_diag = (float)(1.0 / std::sqrt(std::fma(cfg->y_size, cfg->y_size, cfg->x_size * cfg->x_size)));
}

// One curriculum level in kernel form: the handle's config with the level's four world parameters swapped in.
LevelParams make_level(const uavx_config &base, const uavx_level *lv, int L, int B) {
    uavx_config c = base;
    int nl = L, nb = B;
    if (lv) {
        c.x_size = lv->x_size; c.y_size = lv->y_size; c.collider_radius = lv->collider_radius; c.d_sense = lv->d_sense;
        nl = lv->n_active; nb = lv->b_active;
    }
    MultiParams t;
    std::memset(&t, 0, sizeof t);
    derive_world_params(c, t);
    LevelParams o;
    std::memset(&o, 0, sizeof o);
    o.lo_x = t.lo_x; o.lo_y = t.lo_y; o.hi_x = t.hi_x; o.hi_y = t.hi_y;
    o.sq_sense = t.sq_sense; o.sq_two_r = t.sq_two_r; o.inv_sense = t.inv_sense; o.inv_diag = t.inv_diag;
    o.lox = t.lox; o.loy = t.loy; o.hix = t.hix; o.hiy = t.hiy;
    o.n_active = nl; o.b_active = nb;
    return o;
}

void apply_body_rule(uavx_handle *h) {
    MultiParams &p = h->p;
    p.body_step = (float)(h->rule.speed * h->cfg.tau);
    p.body_pmask = h->rule.period - 1;
    p.body_pshift = 0;
    while ((1 << p.body_pshift) < h->rule.period) p.body_pshift++;
    p.body_k0 = (uint32_t)h->rule.seed; p.body_k1 = (uint32_t)(h->rule.seed >> 32);
}

WideLimits derive_wide_limits(const uavx_config &c) {  // the python-float comparands of the float64 episodes
    WideLimits l;
    l.d_sense = c.d_sense;                       // AG:52
    l.two_r = 2 * c.collider_radius;             // MUW:203
    l.two_hard = 2 * 0.5;                        // MUW:8,207
    l.vmax_norm = std::sqrt(std::fma(c.max_speed, c.max_speed, c.max_speed * c.max_speed));  // MUW:62,183
    l.diag = std::sqrt(std::fma(c.y_size, c.y_size, c.x_size * c.x_size));                   // MUW:17
    return l;
}

bool config_valid(const uavx_config *cfg) {
    return cfg->tau > 0 && cfg->max_speed > 0 && cfg->max_acceleration > 0 && cfg->x_size > 0 && cfg->y_size > 0 &&
           cfg->d_sense > 0 && cfg->collider_radius >= 0;
}

dim3 wave_grid(const uavx_handle *h) {
    return dim3((unsigned)((h->p.E + h->p.epw - 1) / h->p.epw));  // one workgroup per epw envs
}

// Launches go to the handle's device; the caller's current device is restored afterwards.
struct DeviceGuard {
    int prev = -1, want;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) : want(device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != want) err = hipSetDevice(want);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
};
#define UAVX_ENTER(h)                                                     \
    DeviceGuard guard_((h)->device);                                      \
    if (guard_.err != hipSuccess) return hip_fail((h), guard_.err, "hipSetDevice")

}  // namespace

float uavx_f32_at_or_above(double b) {   // shared with uavx_uw.hip
    float f = (float)b;
    if ((double)f < b) f = std::nextafterf(f, INFINITY);
    return f;
}
float uavx_f32_at_or_below(double b) {
    float f = (float)b;
    if ((double)f > b) f = std::nextafterf(f, -INFINITY);
    return f;
}

extern "C" {

int uavx_version(void) { return UAVX_VERSION; }

#ifndef UAVX_SRC_HASH
#define UAVX_SRC_HASH ""
#endif
static const char kBuildInfo[] = "UAVX_SRC_HASH=" UAVX_SRC_HASH;   // the loader finds this marker in the file without mapping it
const char *uavx_build_info(void) { return kBuildInfo + 14; }

int uavx_selftest(int device, uint64_t *mismatches) {
    if (!mismatches) return UAVX_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return UAVX_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return UAVX_ERR_INVALID_ARG;
    DeviceGuard guard(device);
    if (guard.err != hipSuccess) return UAVX_ERR_HIP;
    unsigned long long *d = nullptr, h = 0;
    if (hipMalloc(&d, sizeof h) != hipSuccess) return UAVX_ERR_ALLOC;
    hipError_t e = hipMemset(d, 0, sizeof h);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sqrt_selftest_kernel, dim3(256 * 32), dim3(256), 0, 0, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return UAVX_ERR_HIP;
    *mismatches = h;
    return UAVX_OK;
}

const char *uavx_strerror(int status) {
    switch (status) {
        case UAVX_OK: return "ok";
        case UAVX_ERR_INVALID_ARG: return "invalid argument";
        case UAVX_ERR_HIP: return "HIP runtime error";
        case UAVX_ERR_NO_DEVICE: return "no HIP device";
        case UAVX_ERR_UNSUPPORTED: return "unsupported";
        case UAVX_ERR_ALLOC: return "allocation failed";
        default: return "unknown status";
    }
}

int uavx_create(const uavx_config *cfg, int64_t num_envs, int64_t env_offset, int device, uavx_handle **out) {
    if (!cfg || !out || num_envs <= 0 || env_offset < 0) return UAVX_ERR_INVALID_ARG;
    if (cfg->num_agents < 1 || cfg->num_agents > UAVX_MAX_AGENTS) return UAVX_ERR_INVALID_ARG;
    if (cfg->num_bodies < 0 || cfg->num_agents + cfg->num_bodies > UAVX_MAX_AGENTS) return UAVX_ERR_INVALID_ARG;
    if (!config_valid(cfg)) return UAVX_ERR_INVALID_ARG;
    if (num_envs * (int64_t)cfg->num_agents >= (int64_t(1) << 26)) return UAVX_ERR_UNSUPPORTED;  // 32-bit byte offsets (obs: 40 B/agent)
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return UAVX_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return UAVX_ERR_INVALID_ARG;
    uavx_handle *h = new (std::nothrow) uavx_handle();
    if (!h) return UAVX_ERR_ALLOC;
    h->cfg = *cfg;
    h->device = device;
    h->slab = nullptr;
    MultiParams &p = h->p;
    std::memset(&p, 0, sizeof p);
    const int N = cfg->num_agents;
    derive_world_params(*cfg, p);
    h->wl = derive_wide_limits(*cfg);
    const int B = cfg->num_bodies;
    p.N = N;
    p.B = B; p.nslots = N + B; p.kb = (B + N - 1) / N;
    h->gw = B > 0 ? 1 : pick_group_waves(N);
    if (const char *gv = getenv("UAVX_GW")) {   // A/B switch: wavefronts per workgroup of the runtime-N kernels
        const int w = atoi(gv);
        if (w >= 1 && w <= 4 && B == 0 && !(N == 1 || N == 2 || N == 4 || N == 5 || N == 8)) h->gw = w;
    }
    p.epw = std::min(kWave * h->gw / N, kExtSlots / (N + B));  // an EXT wave keeps epw * (L + B) neighbour rows in LDS
    p.magic = 65536 / N + 1;
    p.E = num_envs;
    p.env_offset = env_offset;
    h->ext = B > 0;
    // layouts drawn ahead: a staging workgroup draws up to min(floor(64 W / S), 8) layouts every other launch (it scans in
    // between), and there are enough of them for E / 128 layouts per launch -- what a batch whose episodes last 128 steps on
    // average consumes (a random-initialised actor: 170-190 steps at 4 UAVs, tools/closed_loop.py; shorter episodes draw the
    // excess in place): one per 32 env-workgroups at 4 UAVs, per 64 at 8, per 16 with 8 learners + 16 bodies.  At ~70 episode
    // ends per launch (bench.py --fused) the launch time is flat from 16 to 256 (4 UAVs) / 8 to 64 (8 + 16); with ~350 per launch
    // 64 at 4 UAVs already falls behind (env launch of the closed loop 8.0 -> 9.4 us).
    {
        const int lps = std::min(kHintJobs, std::max(1, kWave * h->gw / (N + B)));
        h->prefetch_every = std::max(1, 64 * lps / std::max(1, p.epw));
    }
    apply_body_rule(h);
    h->levels.l[0] = make_level(*cfg, nullptr, N, B);

    DeviceGuard guard(device);
    hipError_t e = guard.err;
    if (e != hipSuccess) { delete h; return UAVX_ERR_HIP; }
    const size_t A = (size_t)num_envs * N, E = (size_t)num_envs;
    size_t off = 0;
    const size_t o_pos = off;  off = align_up(off + A * sizeof(float2), 256);
    const size_t o_ovr = off;  off = align_up(off + A * sizeof(float), 256);
    const size_t o_vel = off;  off = align_up(off + A * sizeof(double2), 256);
    const size_t o_goal = off; off = align_up(off + A * sizeof(Goal), 256);
    const size_t o_steps = off; off = align_up(off + E * sizeof(uint4), 256);
    const size_t o_wsteps = off; off = align_up(off + ((E + p.epw - 1) / p.epw) * 4, 256);
    const size_t o_reach = off; off = align_up(off + E * 4, 256);
    const size_t o_coll = off;  off = align_up(off + E * 4, 256);
    const size_t o_nonfin = off; off = align_up(off + E * 4, 256);
    const size_t o_finc = off;  off = align_up(off + E * sizeof(uint4), 256);
    const size_t o_finr = off;  off = align_up(off + E * sizeof(float2), 256);
    const size_t o_bpos = off;  off = align_up(off + E * (size_t)B * sizeof(float2), 256);
    const size_t o_bleg = off;  off = align_up(off + E * (size_t)B * sizeof(float4), 256);
    const size_t o_lcur = off;  off = align_up(off + E, 256);
    const size_t o_lnext = off; off = align_up(off + E, 256);
    const size_t o_levels = off; off = align_up(off + sizeof(LevelTable), 256);
    const size_t o_sagent = off; off = align_up(off + 2 * A * sizeof(float4), 256);
    const size_t o_sbpos = off;  off = align_up(off + 2 * E * (size_t)B * sizeof(float2), 256);
    const size_t o_sbleg = off;  off = align_up(off + 2 * E * (size_t)B * sizeof(float4), 256);
    const size_t o_stag = off;   off = align_up(off + 2 * E * sizeof(uint4), 256);
    const size_t o_hint = off;   off = align_up(off + ((E + p.epw - 1) / p.epw + 1) * kHintJobs * sizeof(uint2), 256);   // (staging workgroups <= env-workgroups)
    // step_ex_kernel addresses the state arrays as slab base + 32-bit offsets (leading scalar kernel arguments)
    static_assert(sizeof(size_t) >= 8, "64-bit host");
    if (o_pos != 0 || o_wsteps >= (size_t(1) << 32)) { delete h; return UAVX_ERR_UNSUPPORTED; }
    h->off_vel = (uint32_t)o_vel; h->off_goal = (uint32_t)o_goal; h->off_rec = (uint32_t)o_steps; h->off_wsteps = (uint32_t)o_wsteps;
    if (p.epw > 255 || p.nslots > 255) { delete h; return UAVX_ERR_UNSUPPORTED; }   // (packed into one leading argument; <= 64 today)
    e = hipMalloc(&h->slab, off);
    if (e != hipSuccess) { delete h; return UAVX_ERR_ALLOC; }
    h->slab_bytes = off;
    e = hipMemset(h->slab, 0, off);
    if (e != hipSuccess) { (void)hipFree(h->slab); delete h; return UAVX_ERR_HIP; }
    char *b = static_cast<char *>(h->slab);
    p.pos = reinterpret_cast<float2 *>(b + o_pos);
    p.prev_ovr = reinterpret_cast<float *>(b + o_ovr);
    p.vel = reinterpret_cast<double2 *>(b + o_vel);
    p.goal = reinterpret_cast<Goal *>(b + o_goal);
    p.env_rec = reinterpret_cast<uint4 *>(b + o_steps);
    p.wave_steps = reinterpret_cast<uint32_t *>(b + o_wsteps);
    p.reach = reinterpret_cast<uint32_t *>(b + o_reach);
    p.coll = reinterpret_cast<uint32_t *>(b + o_coll);
    p.nonfin = reinterpret_cast<uint32_t *>(b + o_nonfin);
    p.fin_counts = reinterpret_cast<uint4 *>(b + o_finc);
    p.fin_returns = reinterpret_cast<float2 *>(b + o_finr);
    p.body_pos = reinterpret_cast<float2 *>(b + o_bpos);
    p.body_leg = reinterpret_cast<float4 *>(b + o_bleg);
    p.lvl_cur = reinterpret_cast<uint8_t *>(b + o_lcur);
    p.lvl_next = reinterpret_cast<uint8_t *>(b + o_lnext);
    h->levels_dev = reinterpret_cast<LevelParams *>(b + o_levels);
    p.levels = h->levels_dev;
    p.n_levels = 0; p.level_lo = -1; p.level_hi = -1;
    p.stage_agent = reinterpret_cast<float4 *>(b + o_sagent);
    p.stage_bpos = reinterpret_cast<float2 *>(b + o_sbpos);
    p.stage_bleg = reinterpret_cast<float4 *>(b + o_sbleg);
    p.stage_tag = reinterpret_cast<uint4 *>(b + o_stag);   // zero-filled: no layout is valid yet
    h->hints = reinterpret_cast<uint2 *>(b + o_hint);      // zero-filled: no hints
    if (const char *sb = getenv("UAVX_STAGE_BEHIND")) h->stage_behind = atoi(sb);
    {
        int cus = 0, tpc = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
            hipDeviceGetAttribute(&tpc, hipDeviceAttributeMaxThreadsPerMultiProcessor, device) == hipSuccess && cus > 0 && tpc >= kWave)
            h->wave_slots = cus * (tpc / kWave);
    }
    {
        // Pairs of one-wavefront tiles per workgroup (the 8-UAV specialisation only): measured in one session (r03_ab_notes.md,
        // r04_ab_notes.md section 11), they pay where the launch fills the wavefront slots ONCE -- 65 536 x 8: 8 192 workgroups
        // take the dispatcher 2.3 us to place, half as many 1.2 -- and cost a few percent where it runs in two rounds or leaves
        // half the slots free (the pairs then land unevenly on the SIMDs of an issue-bound launch).
        const long waves = (long)wave_grid(h).x;
        h->tiles = (N == 8 && !h->ext && h->gw == 1 && waves % 2 == 0 && 2 * waves > (long)h->wave_slots && waves <= (long)h->wave_slots) ? 2 : 1;
        if (const char *tv = getenv("UAVX_TILES")) {   // A/B switch
            const int t = atoi(tv);
            if (t == 1 || (t == 2 && N == 8 && !h->ext && h->gw == 1 && waves % 2 == 0)) h->tiles = t;
        }
    }
    p.magic_s = 65536 / (N + B) + 1;
    p.world_version = 1;
    hipLaunchKernelGGL(upload_levels_kernel, dim3(1), dim3(64), 0, 0, h->levels_dev, h->levels);  // level 0 = the config
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(0);
    if (e != hipSuccess) { (void)hipFree(h->slab); delete h; return UAVX_ERR_HIP; }
    *out = h;
    return UAVX_OK;
}

int uavx_destroy(uavx_handle *h) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (h->slab) {
        DeviceGuard guard(h->device);
        (void)hipFree(h->slab);
        if (h->wide_slab) (void)hipFree(h->wide_slab);
    }
    delete h;
    return UAVX_OK;
}

int uavx_set_config(uavx_handle *h, const uavx_config *cfg) {
    if (!h || !cfg) return UAVX_ERR_INVALID_ARG;
    if (cfg->num_agents != h->p.N) return fail(h, UAVX_ERR_INVALID_ARG, "uavx_set_config: num_agents is fixed at creation");
    if (!config_valid(cfg)) return fail(h, UAVX_ERR_INVALID_ARG, "uavx_set_config: parameter out of range");
    if (cfg->num_bodies != h->p.B) return fail(h, UAVX_ERR_INVALID_ARG, "uavx_set_config: num_bodies is fixed at creation");
    if (h->ext && h->p.n_levels > 0)
        return fail(h, UAVX_ERR_UNSUPPORTED, "uavx_set_config: a curriculum is installed; change the world through uavx_set_curriculum");
    h->cfg = *cfg;
    derive_world_params(*cfg, h->p);  // kernel arguments are taken by value at launch: later launches see the new world
    h->wl = derive_wide_limits(*cfg);
    apply_body_rule(h);
    h->p.world_version++;             // pre-drawn layouts of the old world are stale
    return UAVX_OK;   // host-only: without a curriculum the EXT kernels take the world from their arguments too
}

int uavx_num_bodies(const uavx_handle *h) { return h ? h->p.B : -1; }

int uavx_set_prefetch(uavx_handle *h, int every) {
    if (!h || every < 0) return UAVX_ERR_INVALID_ARG;
    h->prefetch_every = every;
    return UAVX_OK;
}

int uavx_set_body_rule(uavx_handle *h, const uavx_body_rule *rule) {
    if (!h || !rule) return UAVX_ERR_INVALID_ARG;
    if (!(rule->speed >= 0) || rule->period < 1 || (rule->period & (rule->period - 1)) != 0)
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_set_body_rule: speed must be >= 0 and period a power of two");
    h->rule = *rule;
    apply_body_rule(h);
    h->p.world_version++;
    return UAVX_OK;
}

int uavx_set_curriculum(uavx_handle *h, const uavx_level *levels, int32_t n_levels, int32_t level_lo, int32_t level_hi,
                        void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (n_levels < 0 || n_levels > UAVX_MAX_LEVELS || (n_levels > 0 && !levels))
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_set_curriculum: 0 <= n_levels <= UAVX_MAX_LEVELS");
    if (h->wide) return fail(h, UAVX_ERR_UNSUPPORTED, "uavx_set_curriculum: not available for float64-position episodes");
    if (n_levels > 0 && level_lo >= 0 && (level_hi < level_lo || level_hi >= n_levels))
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_set_curriculum: need 0 <= level_lo <= level_hi < n_levels (or level_lo < 0)");
    for (int i = 0; i < n_levels; i++) {
        const uavx_level &l = levels[i];
        if (!(l.x_size > 0 && l.y_size > 0 && l.d_sense > 0 && l.collider_radius >= 0) || l.n_active < 1 ||
            l.n_active > h->p.N || l.b_active < 0 || l.b_active > h->p.B)
            return fail(h, UAVX_ERR_INVALID_ARG, "uavx_set_curriculum: level parameter out of range");
    }
    UAVX_ENTER(h);
    if (n_levels == 0) {
        h->levels.l[0] = make_level(h->cfg, nullptr, h->p.N, h->p.B);
    } else {
        for (int i = 0; i < n_levels; i++) h->levels.l[i] = make_level(h->cfg, &levels[i], h->p.N, h->p.B);
    }
    hipLaunchKernelGGL(upload_levels_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), h->levels_dev, h->levels);
    UAVX_HIP(h, hipGetLastError());
    h->p.world_version++;
    h->p.n_levels = n_levels;
    h->p.level_lo = n_levels > 0 ? level_lo : -1;
    h->p.level_hi = n_levels > 0 ? level_hi : -1;
    h->ext = h->p.B > 0 || n_levels > 0;
    return UAVX_OK;
}

int uavx_set_env_levels(uavx_handle *h, const uint8_t *levels, void *stream) {
    if (!h || !levels) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    hipLaunchKernelGGL(env_levels_kernel, dim3((unsigned)((h->p.E + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), h->p, levels, (uint8_t *)nullptr);
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_get_env_levels(uavx_handle *h, uint8_t *levels, void *stream) {
    if (!h || !levels) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    hipLaunchKernelGGL(env_levels_kernel, dim3((unsigned)((h->p.E + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), h->p, (const uint8_t *)nullptr, levels);
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

static int bodies_exchange(uavx_handle *h, const float *set, float *get, void *stream) {
    if (h->p.B == 0) return fail(h, UAVX_ERR_UNSUPPORTED, "the handle has no scripted bodies");
    if ((reinterpret_cast<uintptr_t>(set) | reinterpret_cast<uintptr_t>(get)) & 3u)
        return fail(h, UAVX_ERR_INVALID_ARG, "body records must be 4-byte aligned");
    UAVX_ENTER(h);
    const int64_t n = h->p.E * h->p.B;
    hipLaunchKernelGGL(bodies_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), h->p, set, get);
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}
int uavx_get_bodies(uavx_handle *h, float *records, void *stream) {
    if (!h || !records) return UAVX_ERR_INVALID_ARG;
    return bodies_exchange(h, nullptr, records, stream);
}
int uavx_set_bodies(uavx_handle *h, const float *records, void *stream) {
    if (!h || !records) return UAVX_ERR_INVALID_ARG;
    return bodies_exchange(h, records, nullptr, stream);
}

const char *uavx_last_error(const uavx_handle *h) { return h ? h->err.c_str() : "null handle"; }
int64_t uavx_num_envs(const uavx_handle *h) { return h ? h->p.E : -1; }
int uavx_num_agents(const uavx_handle *h) { return h ? h->p.N : -1; }

static dim3 env_grid(const uavx_handle *h) { return dim3((unsigned)((h->p.E + kBlock - 1) / kBlock)); }
static dim3 agent_grid(const uavx_handle *h) { return dim3((unsigned)((h->p.E * h->p.N + kBlock - 1) / kBlock)); }

static int launch_observe(uavx_handle *h, float *obs, hipStream_t st) {
    if (h->wide) {
        hipLaunchKernelGGL(observe64_kernel, env_grid(h), dim3(kBlock), 0, st, h->p, h->w, h->wl, obs);
        UAVX_HIP(h, hipGetLastError());
        return UAVX_OK;
    }
    dispatch(h, ObserveLaunch{h, wave_grid(h), st, obs});
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_observe(uavx_handle *h, float *obs, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (!obs || (reinterpret_cast<uintptr_t>(obs) & 15u))
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_observe: obs is NULL or not 16-byte aligned");
    UAVX_ENTER(h);
    return launch_observe(h, obs, static_cast<hipStream_t>(stream));
}

int uavx_reset(uavx_handle *h, const uint8_t *mask, uint64_t seed, float *obs, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    if (obs && (reinterpret_cast<uintptr_t>(obs) & 15u))
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_reset: obs not 16-byte aligned");
    if (h->wide && mask)
        return fail(h, UAVX_ERR_UNSUPPORTED, "uavx_reset: a masked reset would mix float32 and float64 episodes in one handle");
    h->wide = false;  // MUW:126,131,144: reset() installs float32 arrays again
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid = wave_grid(h);
    dispatch(h, ResetLaunch{h, grid, st, mask, seed});
    UAVX_HIP(h, hipGetLastError());
    if (obs) return launch_observe(h, obs, st);
    return UAVX_OK;
}

static int launch_step64(uavx_handle *h, const void *actions, int action_dtype, int action_mode, int track_returns,
                         int evaluate, float *obs, float *rew, uint8_t *done, hipStream_t st) {
    if (action_dtype == UAVX_F64)
        hipLaunchKernelGGL((step64_kernel<true>), env_grid(h), dim3(kBlock), 0, st, h->p, h->w, h->wl, actions, action_mode,
                           track_returns, evaluate, obs, rew, done);
    else
        hipLaunchKernelGGL((step64_kernel<false>), env_grid(h), dim3(kBlock), 0, st, h->p, h->w, h->wl, actions, action_mode,
                           track_returns, evaluate, obs, rew, done);
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_step_k(uavx_handle *h, int k, const void *actions, int action_dtype, int evaluate, int tape_out, float *obs,
                float *rew, uint8_t *done, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (!actions || !obs || !rew || !done) return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step: NULL buffer");
    if (k < 1) return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step_k: k < 1");
    if (action_dtype != UAVX_F32 && action_dtype != UAVX_F64)
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step: action_dtype must be UAVX_F32 or UAVX_F64");
    if ((reinterpret_cast<uintptr_t>(obs) & 15u) || (reinterpret_cast<uintptr_t>(actions) & (action_dtype == UAVX_F64 ? 15u : 7u)) ||
        (reinterpret_cast<uintptr_t>(rew) & 3u))
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step: obs must be 16-byte aligned, actions 8 (float32) / 16 (float64), rew 4");
    UAVX_ENTER(h);
    if (h->wide) {
        if (k != 1) return fail(h, UAVX_ERR_UNSUPPORTED, "uavx_step_k: k > 1 is not available for float64-position episodes");
        return launch_step64(h, actions, action_dtype, UAVX_ACTION_CARTESIAN, 0, evaluate, obs, rew, done,
                             static_cast<hipStream_t>(stream));
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid = wave_grid(h);
    if (h->ext && k != 1)
        return fail(h, UAVX_ERR_UNSUPPORTED, "uavx_step_k: k > 1 is not available with scripted bodies / a curriculum");
    dispatch(h, StepLaunch{h, grid, st, actions, action_dtype, evaluate, k, tape_out, obs, rew, done});
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_step(uavx_handle *h, const void *actions, int action_dtype, int evaluate, float *obs, float *rew,
              uint8_t *done, void *stream) {
    return uavx_step_k(h, 1, actions, action_dtype, evaluate, 0, obs, rew, done, stream);
}

int uavx_step_ex(uavx_handle *h, const uavx_step_args *a, void *stream) {
    if (!h || !a) return UAVX_ERR_INVALID_ARG;
    if (!a->actions || !a->obs || !a->rew || !a->done) return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step_ex: NULL buffer");
    if (a->action_dtype != UAVX_F32 && a->action_dtype != UAVX_F64)
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step_ex: action_dtype must be UAVX_F32 or UAVX_F64");
    if (a->action_mode != UAVX_ACTION_CARTESIAN && a->action_mode != UAVX_ACTION_POLAR)
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step_ex: unknown action_mode");
    if (a->reset_policy < UAVX_RESET_NEVER || a->reset_policy > UAVX_RESET_ALL_DONE)
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step_ex: unknown reset_policy");
    if (a->flags_mode != UAVX_FLAGS_ARRAYS && a->flags_mode != UAVX_FLAGS_IN_DONE)
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step_ex: unknown flags_mode");
    if ((reinterpret_cast<uintptr_t>(a->obs) & 15u) || (reinterpret_cast<uintptr_t>(a->actions) & (a->action_dtype == UAVX_F64 ? 15u : 7u)) ||
        (reinterpret_cast<uintptr_t>(a->rew) & 3u))
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_step_ex: obs must be 16-byte aligned, actions 8 (float32) / 16 (float64), rew 4");
    UAVX_ENTER(h);
    if (h->wide) {
        if (a->reset_policy != UAVX_RESET_NEVER || a->step_cap != 0)
            return fail(h, UAVX_ERR_UNSUPPORTED, "uavx_step_ex: no auto-reset / step cap for float64-position episodes");
        if (a->reset_mask) UAVX_HIP(h, hipMemsetAsync(a->reset_mask, 0, (size_t)h->p.E, static_cast<hipStream_t>(stream)));
        if (a->ended) UAVX_HIP(h, hipMemsetAsync(a->ended, 0, (size_t)h->p.E, static_cast<hipStream_t>(stream)));
        if (a->truncated) UAVX_HIP(h, hipMemsetAsync(a->truncated, 0, (size_t)h->p.E, static_cast<hipStream_t>(stream)));
        return launch_step64(h, a->actions, a->action_dtype, a->action_mode, a->track_returns, a->evaluate, a->obs, a->rew,
                             a->done, static_cast<hipStream_t>(stream));
    }
    StepExtra x;
    x.action_mode = a->action_mode; x.reset_policy = a->reset_policy; x.track_returns = a->track_returns;
    x.step_cap = a->step_cap; x.seed_lo = (uint32_t)a->seed; x.seed_hi = (uint32_t)(a->seed >> 32);
    x.reset_mask = a->reset_mask;
    x.ended = a->ended; x.truncated = a->truncated;
    x.flags_in_done = (a->flags_mode == UAVX_FLAGS_IN_DONE) ? 1 : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid = wave_grid(h);
    // auto-resetting call: the launch carries staging workgroups that draw the layouts of the next episodes (stage_ahead)
    const bool resets = a->reset_policy != UAVX_RESET_NEVER || a->step_cap != 0;
    x.use_stage = (h->prefetch_every > 0 && resets) ? 1 : 0;
    x.pf_blocks = 0; x.pf_groups = grid.x;
    x.stage_first = 0; x.step_first = 0;
    x.hints = h->hints;
    // env-workgroups of the launch: pairs of tiles divide evenly (uavx_create); a handle that has since been given a curriculum
    // runs the kernels with levels, which keep one tile per workgroup
    const unsigned step_blocks = grid.x / (unsigned)launch_tiles(h);
    dim3 launch(step_blocks);
    if (x.use_stage) {
        x.pf_blocks = (grid.x + (unsigned)h->prefetch_every - 1u) / (unsigned)h->prefetch_every;
        launch.x = step_blocks + x.pf_blocks;
        // Where in the launch?  Workgroups are dispatched in block order.  While the env-workgroups leave wavefront slots free
        // (65 536 x 4: 4 096 of 8 192) the staging workgroups go IN FRONT and run beside them.  When the env-workgroups alone fill
        // every slot (65 536 x 8, with or without bodies: exactly 8 192 one-wavefront workgroups), whatever comes on top waits
        // for a slot: in front, 512 step wavefronts start 4-12 us late -- the ones behind a drawing workgroup last, and the
        // launch ends with them (per-wavefront timelines, tools/exp_stamps.py: 19.5 us from first start to last end against
        // 17.1 without staging).  BEHIND the env-workgroups the staging workgroups start when the first step wavefronts retire
        // (11 us) and work in the shadow of the ones still running (their ends spread over 10-18 us): nothing that steps is
        // displaced.  Measured in one session, in front / behind: 8 learners + 16 bodies with levels 22.0 / 21.5 us, without
        // levels 22.3 / 21.8, 8 UAVs 13.8 / 13.5; 4 UAVs 7.03 / 7.02, half-full and multi-round launches within 1 %.  Workgroups
        // of several wavefronts (24 UAVs: 42 / 53 us) stay in front: their chain runs on __syncthreads and is long.
        const bool behind = h->stage_behind < 0 ? (h->gw == 1 && (long)grid.x + (long)x.pf_blocks > (long)h->wave_slots) : h->stage_behind != 0;
        if (behind) x.stage_first = step_blocks; else x.step_first = x.pf_blocks;
    }
    dispatch(h, StepExLaunch{h, launch, st, x, a});
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

#ifdef UAVX_STAMPS
extern "C" int uavx_debug_stamps(unsigned long long *host_out, unsigned int *n) {  // debug builds only
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(n, HIP_SYMBOL(g_stamp_n), sizeof(unsigned int));
    hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * 16384);
    unsigned int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_n), &zero, sizeof zero);
    void *dst = nullptr;
    hipGetSymbolAddress(&dst, HIP_SYMBOL(g_stamps));
    hipMemset(dst, 0, sizeof(unsigned long long) * 8 * 16384);
    return 0;
}
#endif

int uavx_get_nonfinite(uavx_handle *h, uint32_t *counts, void *stream) {
    if (!h || !counts) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    UAVX_HIP(h, hipMemcpyAsync(counts, h->p.nonfin, (size_t)h->p.E * 4, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return UAVX_OK;
}

int uavx_get_episode_stats(uavx_handle *h, uint32_t *counts, float *returns, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    hipLaunchKernelGGL(episode_stats_kernel, dim3((unsigned)((h->p.E + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), h->p, counts, returns, 0);
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_clear_episode_stats(uavx_handle *h, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    hipLaunchKernelGGL(episode_stats_kernel, dim3((unsigned)((h->p.E + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), h->p, (uint32_t *)nullptr, (float *)nullptr, 1);
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

static int wide_exchange(uavx_handle *h, const uavx_state_view &v, const uavx_state_view_f64 &v64, int set, hipStream_t st) {
    hipLaunchKernelGGL(wide_exchange_kernel, agent_grid(h), dim3(kBlock), 0, st, h->p, h->w, v, v64, set);
    UAVX_HIP(h, hipGetLastError());
    if (v.counters) {  // env counters live in the shared arrays: the float32-mode kernels handle them
        uavx_state_view c;
        std::memset(&c, 0, sizeof c);
        c.counters = v.counters;
        if (set) hipLaunchKernelGGL(set_state_kernel, agent_grid(h), dim3(kBlock), 0, st, h->p, c);
        else hipLaunchKernelGGL(get_state_kernel, agent_grid(h), dim3(kBlock), 0, st, h->p, c);
        UAVX_HIP(h, hipGetLastError());
    }
    return UAVX_OK;
}

int uavx_get_state(uavx_handle *h, const uavx_state_view *dst, void *stream) {
    if (!h || !dst) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    if (h->wide) return wide_exchange(h, *dst, uavx_state_view_f64{}, 0, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(get_state_kernel, agent_grid(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->p, *dst);
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_set_state(uavx_handle *h, const uavx_state_view *src, void *stream) {
    if (!h || !src) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    if (h->wide) return wide_exchange(h, *src, uavx_state_view_f64{}, 1, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(set_state_kernel, agent_grid(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->p, *src);
    UAVX_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_get_position_mode(const uavx_handle *h) { return h ? (h->wide ? UAVX_POS_F64 : UAVX_POS_F32) : UAVX_ERR_INVALID_ARG; }

int uavx_set_position_mode(uavx_handle *h, int mode, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
    if (mode != UAVX_POS_F32 && mode != UAVX_POS_F64) return fail(h, UAVX_ERR_INVALID_ARG, "uavx_set_position_mode: unknown mode");
    UAVX_ENTER(h);
    if ((mode == UAVX_POS_F64) == h->wide) return UAVX_OK;
    if (mode == UAVX_POS_F64 && h->ext)
        return fail(h, UAVX_ERR_UNSUPPORTED, "uavx_set_position_mode: float64-position episodes are not available with scripted bodies / a curriculum");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (mode == UAVX_POS_F64) {
        if (!h->wide_slab) {  // 48 B per agent, once per handle
            const size_t A = (size_t)h->p.E * h->p.N;
            const size_t o_tgt = align_up(A * sizeof(double2), 256), o_init = o_tgt + align_up(A * sizeof(double2), 256);
            const size_t o_prev = o_init + align_up(A * sizeof(double), 256), total = o_prev + align_up(A * sizeof(double), 256);
            if (hipMalloc(&h->wide_slab, total) != hipSuccess) {
                h->wide_slab = nullptr;
                return fail(h, UAVX_ERR_ALLOC, "uavx_set_position_mode: hipMalloc of the float64 arrays failed");
            }
            char *b = static_cast<char *>(h->wide_slab);
            h->w.pos = reinterpret_cast<double2 *>(b);
            h->w.tgt = reinterpret_cast<double2 *>(b + o_tgt);
            h->w.init_d = reinterpret_cast<double *>(b + o_init);
            h->w.prev_d = reinterpret_cast<double *>(b + o_prev);
        }
        hipLaunchKernelGGL(widen_state_kernel, agent_grid(h), dim3(kBlock), 0, st, h->p, h->w);
    } else {
        hipLaunchKernelGGL(narrow_state_kernel, agent_grid(h), dim3(kBlock), 0, st, h->p, h->w);
    }
    UAVX_HIP(h, hipGetLastError());
    h->wide = (mode == UAVX_POS_F64);
    return UAVX_OK;
}

int uavx_set_state_f64(uavx_handle *h, const uavx_state_view_f64 *src, void *stream) {
    if (!h || !src) return UAVX_ERR_INVALID_ARG;
    const int rc = uavx_set_position_mode(h, UAVX_POS_F64, stream);  // assigning float64 arrays makes the episode float64
    if (rc != UAVX_OK) return rc;
    UAVX_ENTER(h);
    uavx_state_view none;
    std::memset(&none, 0, sizeof none);
    return wide_exchange(h, none, *src, 1, static_cast<hipStream_t>(stream));
}

int uavx_get_state_f64(uavx_handle *h, const uavx_state_view_f64 *dst, void *stream) {
    if (!h || !dst) return UAVX_ERR_INVALID_ARG;
    if (!h->wide) return fail(h, UAVX_ERR_UNSUPPORTED, "uavx_get_state_f64: the handle is in float32-position mode");
    UAVX_ENTER(h);
    uavx_state_view none;
    std::memset(&none, 0, sizeof none);
    return wide_exchange(h, none, *dst, 0, static_cast<hipStream_t>(stream));
}

// ---- exact snapshot / restore of a handle (checkpoint row) ----
namespace {
constexpr uint64_t kSnapMagic = 0x3358564155ull;   // "UAVX3"
struct SnapHeader {
    uint64_t magic;
    uint32_t version, header_bytes;
    uint64_t slab_bytes, wide_bytes;
    int64_t E, env_offset;
    int32_t N, B, wide, ext, n_levels, level_lo, level_hi, prefetch_every;
    uint32_t world_version, epw;   // epw: envs per workgroup -- what the per-workgroup step counters of the slab are indexed by
    uavx_config cfg;
    uavx_body_rule rule;
    LevelTable levels;
};
size_t wide_slab_bytes(const uavx_handle *h) {   // the float64-position arrays (uavx_set_position_mode): 48 B per agent
    const size_t A = (size_t)h->p.E * h->p.N;
    return 2 * align_up(A * sizeof(double2), 256) + 2 * align_up(A * sizeof(double), 256);
}
size_t snap_header_bytes() { return align_up(sizeof(SnapHeader), 256); }
__global__ void snap_header_kernel(SnapHeader *dst, SnapHeader hd) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = hd;
}
}  // namespace

int64_t uavx_snapshot_bytes(const uavx_handle *h) {
    if (!h) return -1;
    return (int64_t)(snap_header_bytes() + h->slab_bytes + wide_slab_bytes(h));
}

int uavx_save(uavx_handle *h, void *dst, void *stream) {
    if (!h || !dst) return UAVX_ERR_INVALID_ARG;
    if (reinterpret_cast<uintptr_t>(dst) & 255u) return fail(h, UAVX_ERR_INVALID_ARG, "uavx_save: the snapshot buffer must be 256-byte aligned");
    UAVX_ENTER(h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    SnapHeader hd;
    std::memset(&hd, 0, sizeof hd);
    hd.magic = kSnapMagic; hd.version = UAVX_VERSION; hd.header_bytes = (uint32_t)snap_header_bytes();
    hd.slab_bytes = h->slab_bytes; hd.wide_bytes = h->wide ? wide_slab_bytes(h) : 0;
    hd.E = h->p.E; hd.env_offset = h->p.env_offset; hd.N = h->p.N; hd.B = h->p.B; hd.wide = h->wide ? 1 : 0; hd.ext = h->ext ? 1 : 0;
    hd.n_levels = h->p.n_levels; hd.level_lo = h->p.level_lo; hd.level_hi = h->p.level_hi; hd.prefetch_every = h->prefetch_every;
    hd.world_version = h->p.world_version; hd.epw = (uint32_t)h->p.epw;
    hd.cfg = h->cfg; hd.rule = h->rule; hd.levels = h->levels;
    char *b = static_cast<char *>(dst);
    hipLaunchKernelGGL(snap_header_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<SnapHeader *>(b), hd);   // (by value: no host buffer to keep alive)
    UAVX_HIP(h, hipGetLastError());
    UAVX_HIP(h, hipMemcpyAsync(b + hd.header_bytes, h->slab, h->slab_bytes, hipMemcpyDeviceToDevice, st));
    if (h->wide) UAVX_HIP(h, hipMemcpyAsync(b + hd.header_bytes + h->slab_bytes, h->wide_slab, hd.wide_bytes, hipMemcpyDeviceToDevice, st));
    return UAVX_OK;
}

int uavx_load(uavx_handle *h, const void *src, void *stream) {
    if (!h || !src) return UAVX_ERR_INVALID_ARG;
    UAVX_ENTER(h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    SnapHeader hd;
    UAVX_HIP(h, hipMemcpyAsync(&hd, src, sizeof hd, hipMemcpyDeviceToHost, st));
    UAVX_HIP(h, hipStreamSynchronize(st));   // the header decides what follows: this call waits for `stream`
    if (hd.magic != kSnapMagic || hd.version != UAVX_VERSION || hd.header_bytes != snap_header_bytes())
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_load: not a snapshot of this library version");
    if (hd.E != h->p.E || hd.N != h->p.N || hd.B != h->p.B || hd.slab_bytes != h->slab_bytes || hd.epw != (uint32_t)h->p.epw)
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_load: the snapshot was taken from a handle of another shape (envs / agents / bodies / envs per workgroup)");
    // everything else the header carries goes into copy lengths, kernel arguments and table indices: a truncated or damaged
    // snapshot is refused here, not found out by a kernel
    const bool lvl_ok = hd.n_levels >= 0 && hd.n_levels <= UAVX_MAX_LEVELS &&
                        (hd.n_levels == 0 ? (hd.level_lo == -1 && hd.level_hi == -1)
                                          : (hd.level_lo < 0 || (hd.level_lo <= hd.level_hi && hd.level_hi < hd.n_levels)));
    if ((hd.wide != 0 && hd.wide != 1) || (hd.ext != 0 && hd.ext != 1) || hd.wide_bytes != (hd.wide ? wide_slab_bytes(h) : 0) ||
        !lvl_ok || hd.prefetch_every < 0 || hd.env_offset < 0 || !config_valid(&hd.cfg) || hd.cfg.num_agents != h->p.N ||
        hd.cfg.num_bodies != h->p.B || !(hd.rule.speed >= 0) || hd.rule.period < 1 || (hd.rule.period & (hd.rule.period - 1)) != 0 ||
        (hd.ext == 0 && (hd.B > 0 || hd.n_levels > 0)) || (hd.wide && hd.ext))
        return fail(h, UAVX_ERR_INVALID_ARG, "uavx_load: inconsistent snapshot header (truncated or corrupted snapshot)");
    const char *b = static_cast<const char *>(src);
    if (hd.wide) {   // the float64-position arrays exist from the first switch to that mode on
        const int rc = uavx_set_position_mode(h, UAVX_POS_F64, stream);
        if (rc != UAVX_OK) return rc;
    }
    UAVX_HIP(h, hipMemcpyAsync(h->slab, b + hd.header_bytes, h->slab_bytes, hipMemcpyDeviceToDevice, st));
    if (hd.wide) UAVX_HIP(h, hipMemcpyAsync(h->wide_slab, b + hd.header_bytes + h->slab_bytes, hd.wide_bytes, hipMemcpyDeviceToDevice, st));
    // host side of the handle: world, body rule, curriculum, staging cadence -- everything later launches take by value
    h->cfg = hd.cfg;
    derive_world_params(h->cfg, h->p);
    h->wl = derive_wide_limits(h->cfg);
    h->rule = hd.rule;
    apply_body_rule(h);
    h->levels = hd.levels;
    h->p.n_levels = hd.n_levels; h->p.level_lo = hd.level_lo; h->p.level_hi = hd.level_hi;
    h->p.world_version = hd.world_version;
    h->p.env_offset = hd.env_offset;      // the snapshot brings its own global env ids (the Philox streams are keyed by them)
    h->prefetch_every = hd.prefetch_every;
    h->ext = hd.ext != 0;
    h->wide = hd.wide != 0;
    return UAVX_OK;
}

int uavx_get_metrics(uavx_handle *h, uint32_t *counters, void *stream) {
    if (!h || !counters) return UAVX_ERR_INVALID_ARG;
    uavx_state_view v;
    std::memset(&v, 0, sizeof v);
    v.counters = counters;
    return uavx_get_state(h, &v, stream);
}

}  // extern "C"
