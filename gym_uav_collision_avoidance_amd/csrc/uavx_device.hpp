// uavx_device.hpp — device-side building blocks shared by the MultiUAVWorld2D and UAVWorld2D
// kernels (gfx950 / CDNA4, wave64).  Compiled with -ffp-contract=off: the reference's float32
// norms and float64 kinematics are plain IEEE mul/add/div/sqrt chains with no FMA, and positions,
// distances and every mask must come out bit-identical to them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uavx {

constexpr int kWave = 64;
constexpr int kBlock = 256;               // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kInvPi = 0.318309886183790671538f;
constexpr float kTwoPi = 6.28318530717958647692f;
constexpr float kInvTwoPi = 0.159154943091895335769f;

// LDS traffic below is wave-private (an env never spans two waves), so a compiler-level ordering
// point is all that is needed: DS instructions of one wave execute in issue order.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// np.linalg.norm on a float32 pair: fl(fl(x*x) + fl(y*y)) then correctly rounded sqrt (AG:33,51).
__device__ __forceinline__ float norm32(float x, float y) {
    float a = x * x;
    float b = y * y;
    return sqrtf(a + b);  // IEEE-rounded: built with -fhip-fp32-correctly-rounded-divide-sqrt
}

// np.clip on float64 scalars (AG:26-27): minimum(maximum(x, lo), hi)
__device__ __forceinline__ double clip64(double x, double lo, double hi) {
    double m = (x < lo) ? lo : x;
    return (m > hi) ? hi : m;
}

// atan2(sin x, cos x) for x in [-2pi, 2pi] (difference of two atan2 results), float32.  Equal to
// the reference's wrap on the circle; at the +-pi seam the sign may differ.
__device__ __forceinline__ float wrap_pi(float x) {
    float k = rintf(x * kInvTwoPi);
    return fmaf(-k, kTwoPi, x);
}

// One double-integrator axis update, AG:26-29, float64 with a float32 position accumulate.
__device__ __forceinline__ void axis_update(double a, double tau, double amax, double vmax, double &v,
                                            float &x) {
    double dv = clip64((a - v) / tau, -amax, amax);   // AG:26
    v = clip64(v + dv * tau, -vmax, vmax);            // AG:27
    x = (float)((double)x + v * tau);                 // AG:28-29 (float32 array += float64 array)
}

// Philox4x32-10, counter-based: one call yields the two 53-bit uniforms of one
// np.random.uniform(lo, hi, (2,)) draw (MUW:126,131,144; UW:121-126).
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double bits53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
struct PhiloxDraws {
    uint32_t env_lo, env_hi, episode, k0, k1, draw;
    // lo + (hi-lo)*U cast to float32, per axis, like np.random.uniform(...).astype(np.float32)
    __device__ __forceinline__ void point32(double lox, double loy, double hix, double hiy, float &px, float &py) {
        uint32_t o[4];
        philox4x32(env_lo, env_hi, draw++, episode, k0, k1, o);
        px = (float)(lox + (hix - lox) * bits53(o[0], o[1]));
        py = (float)(loy + (hiy - loy) * bits53(o[2], o[3]));
    }
};

}  // namespace uavx
