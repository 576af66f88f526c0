// uavx_uw.hip — E x UAVWorld2D (single UAV, 4-dim observation) on one MI355X.
// Reference: UW = gym_uav_collision_avoidance/envs/uav_world_2d.py of dazchi/gym-uav-collision-avoidance.
//
// One lane per env (there is no cross-agent work): every load/store is lane-contiguous.
// HBM layout: dyn float4[E] {x, y, prev_d, flags} r/w 16 B; vel double2[E] r/w 16 B;
// goal float[3E] {tx, ty, init_d} read 12 B; steps u32[E] r/w; episode u32[E] (reset only).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <string>

#include "../../include/uavx.h"
#include "uavx_device.hpp"

namespace uavx {

struct UwParams {
    double tau, amax, vmax;
    double lox, loy, hix, hiy;
    float tau_f;       // float32(tau): UW:142 divides a float32 array by the python float
    float inv_vmax;    // 1/max_speed[0]            UW:88
    float inv_diag;    // 1/‖(x_size,y_size)‖       UW:17,97
    int64_t E, env_offset;
    float4 *dyn;
    double2 *vel;
    float *goal;
    uint32_t *steps, *episode;
};

// UW:88-97 in float32
__device__ __forceinline__ float4 uw_obs(const UwParams &p, float speed, float theta, float dist_t, float dth) {
    return make_float4(speed * p.inv_vmax, theta * kInvPi, dist_t * p.inv_diag, dth * kInvPi);
}

template <bool ACT64>
__global__ __launch_bounds__(kBlock) void uw_step_kernel(UwParams p, const void *__restrict__ actions,
                                                         float4 *__restrict__ obs_out, float *__restrict__ rew_out,
                                                         uint8_t *__restrict__ done_out, float *__restrict__ info_out) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    const float4 d4 = p.dyn[e];
    double2 v = p.vel[e];
    float x = d4.x, y = d4.y;
    const float prev_d = d4.z;
    const uint32_t flags = __float_as_uint(d4.w);
    const float tx = p.goal[3 * e], ty = p.goal[3 * e + 1], init_d = p.goal[3 * e + 2];
    double ax, ay;
    bool f32_first = false;
    if (ACT64) {
        const double2 a = reinterpret_cast<const double2 *>(actions)[e];
        ax = a.x; ay = a.y;
    } else {
        const float2 a = reinterpret_cast<const float2 *>(actions)[e];
        ax = (double)a.x; ay = (double)a.y;
        f32_first = (flags & UAVX_FLAG_VEL_F32) != 0;  // float32 action - float32 velocity, / float32(tau)
    }
    {   // UW:142-147
        double qx, qy;
        if (f32_first) {
            qx = (double)(((float)ax - (float)v.x) / p.tau_f);
            qy = (double)(((float)ay - (float)v.y) / p.tau_f);
        } else {
            qx = (ax - v.x) / p.tau;
            qy = (ay - v.y) / p.tau;
        }
        v.x = clip64(v.x + clip64(qx, -p.amax, p.amax) * p.tau, -p.vmax, p.vmax);
        v.y = clip64(v.y + clip64(qy, -p.amax, p.amax) * p.tau, -p.vmax, p.vmax);
        x = (float)((double)x + v.x * p.tau);
        y = (float)((double)y + v.y * p.tau);
    }
    const bool oob = !((double)x >= p.lox && (double)x <= p.hix && (double)y >= p.loy && (double)y <= p.hiy);  // UW:149,162
    const float tdx = tx - x, tdy = ty - y;
    const float d = norm32(tdx, tdy);                        // UW:150
    const float theta = atan2f((float)v.y, (float)v.x);      // UW:89
    const float dth = wrap_pi(atan2f(tdy, tdx) - theta);     // UW:155-156
    float r = 0.0f - 1.0f / init_d;                          // UW:152-153 (float32 under NEP 50)
    r = r + 10.0f * (prev_d - d);                            // UW:154
    r = r - 0.1f * fabsf(dth);                               // UW:157
    uint32_t dn = 0;
    if (d < 0.5f) { dn = 1; r = r + 1000.0f; }               // UW:159-161
    else if (oob) dn = 1;                                    // UW:162-163
    const float speed = sqrtf((float)fma(v.y, v.y, v.x * v.x));
    obs_out[e] = uw_obs(p, speed, theta, d, dth);            // UW:168
    rew_out[e] = r;
    done_out[e] = (uint8_t)dn;
    if (info_out) info_out[e] = d;                           // UW:114-117
    p.dyn[e] = make_float4(x, y, d, __uint_as_float(flags & ~UAVX_FLAG_VEL_F32));  // UW:172
    p.vel[e] = v;
    p.steps[e] += 1;                                         // UW:170
}

__global__ __launch_bounds__(kBlock) void uw_observe_kernel(UwParams p, float4 *__restrict__ obs_out) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    const float4 d4 = p.dyn[e];
    const double2 v = p.vel[e];
    const float tdx = p.goal[3 * e] - d4.x, tdy = p.goal[3 * e + 1] - d4.y;
    const float theta = atan2f((float)v.y, (float)v.x);
    const float dth = wrap_pi(atan2f(tdy, tdx) - theta);
    const float speed = sqrtf((float)fma(v.y, v.y, v.x * v.x));
    obs_out[e] = uw_obs(p, speed, theta, norm32(tdx, tdy), dth);
}

// UW:119-131
__global__ __launch_bounds__(kBlock) void uw_reset_kernel(UwParams p, const uint8_t *__restrict__ mask, uint64_t seed) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    if (mask && !mask[e]) return;
    const uint64_t ge = (uint64_t)(p.env_offset + e);
    PhiloxDraws rng{(uint32_t)ge, (uint32_t)(ge >> 32), p.episode[e], (uint32_t)seed, (uint32_t)(seed >> 32), 0u};
    float x, y, vx, vy, tx, ty;
    rng.point32(p.lox, p.loy, p.hix, p.hiy, x, y);                       // UW:121
    rng.point32(-p.vmax, -p.vmax, p.vmax, p.vmax, vx, vy);               // UW:122
    rng.point32(p.lox, p.loy, p.hix, p.hiy, tx, ty);                     // UW:126
    const float d0 = norm32(tx - x, ty - y);                             // UW:129
    p.dyn[e] = make_float4(x, y, d0, __uint_as_float(UAVX_FLAG_VEL_F32)); // UW:130
    p.vel[e] = make_double2((double)vx, (double)vy);
    p.goal[3 * e] = tx; p.goal[3 * e + 1] = ty; p.goal[3 * e + 2] = d0;
    p.steps[e] = 0;                                                      // UW:131
    p.episode[e] += 1;
}

__global__ __launch_bounds__(kBlock) void uw_get_state_kernel(UwParams p, uavx_uw_state_view v) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    const float4 d = p.dyn[e];
    if (v.loc) { v.loc[2 * e] = d.x; v.loc[2 * e + 1] = d.y; }
    if (v.prev_d) v.prev_d[e] = d.z;
    if (v.flags) v.flags[e] = (uint8_t)__float_as_uint(d.w);
    if (v.vel) { const double2 w = p.vel[e]; v.vel[2 * e] = w.x; v.vel[2 * e + 1] = w.y; }
    if (v.tgt) { v.tgt[2 * e] = p.goal[3 * e]; v.tgt[2 * e + 1] = p.goal[3 * e + 1]; }
    if (v.init_d) v.init_d[e] = p.goal[3 * e + 2];
    if (v.counters) { v.counters[2 * e] = p.steps[e]; v.counters[2 * e + 1] = p.episode[e]; }
}

__global__ __launch_bounds__(kBlock) void uw_set_state_kernel(UwParams p, uavx_uw_state_view v) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    float4 d = p.dyn[e];
    if (v.loc) { d.x = v.loc[2 * e]; d.y = v.loc[2 * e + 1]; }
    if (v.prev_d) d.z = v.prev_d[e];
    if (v.flags) d.w = __uint_as_float((uint32_t)v.flags[e]);
    p.dyn[e] = d;
    if (v.vel) p.vel[e] = make_double2(v.vel[2 * e], v.vel[2 * e + 1]);
    if (v.tgt) { p.goal[3 * e] = v.tgt[2 * e]; p.goal[3 * e + 1] = v.tgt[2 * e + 1]; }
    if (v.init_d) p.goal[3 * e + 2] = v.init_d[e];
    if (v.counters) { p.steps[e] = v.counters[2 * e]; p.episode[e] = v.counters[2 * e + 1]; }
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
    p.lox = -cfg->x_size / 2.0; p.loy = -cfg->y_size / 2.0; p.hix = cfg->x_size / 2.0; p.hiy = cfg->y_size / 2.0;
    p.tau_f = (float)cfg->tau;
    p.inv_vmax = (float)(1.0 / cfg->max_speed);
    p.inv_diag = (float)(1.0 / std::sqrt(std::fma(cfg->y_size, cfg->y_size, cfg->x_size * cfg->x_size)));
    p.E = num_envs;
    p.env_offset = env_offset;
    UwDeviceGuard guard(device);
    if (guard.err != hipSuccess) { delete h; return UAVX_ERR_HIP; }
    const size_t E = (size_t)num_envs;
    size_t off = 0;
    const size_t o_dyn = off;  off = uw_align(off + E * sizeof(float4));
    const size_t o_vel = off;  off = uw_align(off + E * sizeof(double2));
    const size_t o_goal = off; off = uw_align(off + E * 3 * sizeof(float));
    const size_t o_steps = off; off = uw_align(off + E * 4);
    const size_t o_epi = off;   off = uw_align(off + E * 4);
    if (hipMalloc(&h->slab, off) != hipSuccess) { delete h; return UAVX_ERR_ALLOC; }
    if (hipMemset(h->slab, 0, off) != hipSuccess) { (void)hipFree(h->slab); delete h; return UAVX_ERR_HIP; }
    char *b = static_cast<char *>(h->slab);
    p.dyn = reinterpret_cast<float4 *>(b + o_dyn);
    p.vel = reinterpret_cast<double2 *>(b + o_vel);
    p.goal = reinterpret_cast<float *>(b + o_goal);
    p.steps = reinterpret_cast<uint32_t *>(b + o_steps);
    p.episode = reinterpret_cast<uint32_t *>(b + o_epi);
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
    if (!obs) return uw_fail(h, UAVX_ERR_INVALID_ARG, "uavx_uw_observe: obs is NULL");
    UW_ENTER(h);
    hipLaunchKernelGGL(uw_observe_kernel, env_grid(h), dim3(kBlock), 0, static_cast<hipStream_t>(stream), h->p,
                       reinterpret_cast<float4 *>(obs));
    UW_HIP(h, hipGetLastError());
    return UAVX_OK;
}

int uavx_uw_reset(uavx_uw_handle *h, const uint8_t *mask, uint64_t seed, float *obs, void *stream) {
    if (!h) return UAVX_ERR_INVALID_ARG;
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
    UW_ENTER(h);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (action_dtype == UAVX_F64)
        hipLaunchKernelGGL((uw_step_kernel<true>), env_grid(h), dim3(kBlock), 0, st, h->p, actions,
                           reinterpret_cast<float4 *>(obs), rew, done, info_distance);
    else
        hipLaunchKernelGGL((uw_step_kernel<false>), env_grid(h), dim3(kBlock), 0, st, h->p, actions,
                           reinterpret_cast<float4 *>(obs), rew, done, info_distance);
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
