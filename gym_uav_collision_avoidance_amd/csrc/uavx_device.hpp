// uavx_device.hpp — device-side building blocks shared by the MultiUAVWorld2D and UAVWorld2D
// kernels (gfx950 / CDNA4, wave64).  Compiled with -ffp-contract=off: the reference's float32
// norms and float64 kinematics are plain IEEE mul/add/div/sqrt chains with no FMA, and positions,
// distances and every mask must come out bit-identical to them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uavx {

constexpr int kWave = 64;
#ifndef UAVX_BLOCK
#define UAVX_BLOCK 64  // A/B 64..1024 on MI355X: within 2 %, 64 marginally best (single-wave workgroups)
#endif
constexpr int kBlock = UAVX_BLOCK;        // waves per workgroup = kBlock / 64
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

// Write-through (sc1) vector stores.  A kernel boundary writes back whatever the launch left dirty
// in the XCD L2s (MI355X_MICROARCH.md price list, row "boundary": + B / 6 TB/s); the step kernel
// dirties ~20 MB per launch, so its bulk outputs are stored write-through and drain while the other
// waves still compute instead of serialising behind the last wave.  Out-of-range offsets are dropped
// by the buffer bounds check.
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int kAuxSc1 = 16;  // gfx940+ cache-policy bit SC1 (write-through to memory)

__device__ __forceinline__ rsrc_t make_rsrc(const void *base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), /*stride*/ 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void store16_wt(rsrc_t r, uint32_t byte_off, float4 v) {
    u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, 0, kAuxSc1);
}
__device__ __forceinline__ void store16_wt(rsrc_t r, uint32_t byte_off, double2 v) {
    const unsigned long long a = __double_as_longlong(v.x), b = __double_as_longlong(v.y);
    u32x4 d = {(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)byte_off, 0, kAuxSc1);
}
__device__ __forceinline__ void store8_wt(rsrc_t r, uint32_t byte_off, float2 v) {
    u32x2 d = {__float_as_uint(v.x), __float_as_uint(v.y)};
    __builtin_amdgcn_raw_buffer_store_b64(d, r, (int)byte_off, 0, kAuxSc1);
}

// A kernel argument fetched WHERE IT IS USED.  The compiler treats the kernel-argument segment as invariant, dereferenceable
// memory and hoists every scalar load from it to the top of the kernel, where the value then occupies scalar registers for the
// whole life of the wavefront -- also the pointers that only a rare branch (an atomic on a goal arrival, a body's new waypoint)
// or the last few instructions need.  At 8 wavefronts per SIMD a wavefront has 80 scalar registers (800 per SIMD, 16 of each
// wavefront's share set aside for the trap handler, blocks of 16): the step kernels with scripted bodies went over and spilled
// such pointers into VGPR lanes (v_writelane / v_readlane).  Here the segment's address goes through an empty asm, which hides
// what it points at, so the load stays where it is written: one s_load through the scalar cache at the use (every wavefront of
// the launch reads the same line) instead of registers held from the first instruction on.
// `byte_off` is the argument's offset in the segment: offsetof() into the FIRST by-value argument of the kernel.  (Taking the
// address of a member of the argument itself will not do: an escaping address pins the kernel's private copy of the whole
// struct into scratch memory.)
typedef const __attribute__((address_space(4))) char *karg_ptr;
__device__ __forceinline__ karg_ptr late_kargs() {   // the segment, laundered: loads through it stay behind this point
    karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return ka;
}
template <class T>
__device__ __forceinline__ T late_karg(uint32_t byte_off, karg_ptr ka = late_kargs()) {
    return *(const __attribute__((address_space(4))) T *)(ka + byte_off);
}

// Correctly rounded sqrtf in 9 instructions instead of the compiler's 17: v_sqrt_f32 (<= 1 ulp) followed by the
// same +-1 ulp residual test the compiler's IEEE expansion uses (e = s - c*r for the two neighbours c of r), but
// WITHOUT its input scaling by 2^32 / output scaling by 2^-16, which only serves s < 2^-96 (v_sqrt_f32 flushes
// denormal inputs and the residuals would underflow).  Squared distances of float32 world coordinates are either
// 0 or far above that (two distinct points closer than 1.1e-14 m would need coordinates of that size), so the
// scaled sequence sits behind a wave-uniform branch that is never taken in practice.  s = 0, +inf and NaN fall
// through the residual tests unchanged (every compare is false).  The step kernel spends a fifth of its VALU
// issue slots on square roots (prev / new target distance, N-1 neighbour distances); uavx_selftest() compares
// this function with sqrtf on every float32 bit pattern.
__device__ __forceinline__ float sqrt_rn(float s) {
    if (__builtin_expect(__any((__float_as_uint(s) - 1u) < 0x0F7FFFFFu), 0)) return sqrtf(s);  // 0 < s < 2^-96
    const float r = __builtin_amdgcn_sqrtf(s);
    const float lo = __uint_as_float(__float_as_uint(r) - 1u), hi = __uint_as_float(__float_as_uint(r) + 1u);
    const float e_lo = fmaf(-lo, r, s), e_hi = fmaf(-hi, r, s);
    float out = (e_lo <= 0.f) ? lo : r;
    out = (e_hi > 0.f) ? hi : out;
    return out;
}

// np.linalg.norm on a float32 pair: fl(fl(x*x) + fl(y*y)) then correctly rounded sqrt (AG:33,51).
__device__ __forceinline__ float norm32(float x, float y) {
    float a = x * x;
    float b = y * y;
    return sqrt_rn(a + b);
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

// atan2f for finite inputs, ~2 ulp: octant reduction + Cephes atanf polynomial on |t| <= tan(pi/8).
// (Results only feed float32 observation/reward features that are compared at 1e-5.)
__device__ __forceinline__ float atan2_fast(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const bool big = mn > 0.41421356237f * mx;
    const float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    den = (mx == 0.f) ? 1.f : den;              // atan2(0, 0) = 0
    const float t = num * __builtin_amdgcn_rcpf(den);
    const float z = t * t;
    float pl = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    pl = fmaf(pl, z, 1.99777106478e-1f);
    pl = fmaf(pl, z, -3.33329491539e-1f);
    float r = fmaf(pl * z, t, t);
    r = big ? r + 0.78539816339744830962f : r;
    r = (ay > ax) ? 1.57079632679489661923f - r : r;
    r = (x < 0.f) ? kPi - r : r;
    return copysignf(r, y);
}

// The same reduction and polynomial with an IEEE division instead of v_rcp_f32: every operation is correctly rounded, so
// the CPU test oracle restates it bit for bit (its atan2_leg).  Used where an angle becomes STATE (the heading
// a scripted body keeps for a whole leg); off the per-step path, so its cost does not matter.
__device__ __forceinline__ float atan2_exact(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const bool big = mn > 0.41421356237f * mx;
    const float num = big ? mn - mx : mn;
    float den = big ? mn + mx : mx;
    den = (mx == 0.f) ? 1.f : den;
    const float t = num / den;
    const float z = t * t;
    float pl = fmaf(8.05374449538e-2f, z, -1.38776856032e-1f);
    pl = fmaf(pl, z, 1.99777106478e-1f);
    pl = fmaf(pl, z, -3.33329491539e-1f);
    float r = fmaf(pl * z, t, t);
    r = big ? r + 0.78539816339744830962f : r;
    r = (ay > ax) ? 1.57079632679489661923f - r : r;
    r = (x < 0.f) ? kPi - r : r;
    return copysignf(r, y);
}

// v_max_f64 / v_min_f64 against a wave-uniform limit, as ONE instruction each.  fmax() / fmin() make hipcc canonicalise
// every operand first (v_max_f64 x, x, x -- also the limits held in scalar registers, again at every use): 20 float64
// instructions for the 12 clamps of one agent step, and a float64 instruction costs two float32 issue slots.  For
// non-NaN operands the result is the same; a NaN operand gives the other one back (what the callers count on: they
// re-poison NaN inputs afterwards).  `lim` must be uniform (kernel argument / literal): it is passed in scalar registers.
__device__ __forceinline__ double max64u(double x, double lim) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(lim));
    return r;
}
__device__ __forceinline__ double min64u(double x, double lim) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(lim));
    return r;
}

// x / tau, correctly rounded, without the hardware division sequence: with r = RN(1/tau) the
// Markstein form q0 = RN(x*r), rem = fma(-q0, tau, x) (exact), q = fma(rem, r, q0) returns the IEEE
// quotient (checked against x/tau on 2.1e9 samples for tau = 0.02 and six other steps; uavx_create
// re-checks the handle's tau on 1e5 samples and falls back to '/' if a single one differs).
// |x| is capped at 1e300 first so an infinite action still ends at the clip limit like np.clip does
// (a NaN turns into a finite value here; axis_update() poisons the result afterwards).
__device__ __forceinline__ double div_tau(double x, double tau, double rtau, bool recip_ok) {
    if (!recip_ok) return x / tau;
    x = min64u(max64u(x, -1e300), 1e300);
    const double q0 = x * rtau;
    const double rem = fma(-q0, tau, x);
    return fma(rem, rtau, q0);
}

// One double-integrator axis update, AG:26-29, float64 with a float32 position accumulate.  np.clip is
// done with v_max_f64 / v_min_f64 (one instruction each instead of compare + two selects); those return the
// non-NaN operand, so a NaN action or velocity is detected once (unordered compare) and put back, which
// is what np.clip / the reference would propagate.
__device__ __forceinline__ void axis_update(double a, double tau, double rtau, bool recip_ok, double amax, double vmax,
                                            double &v, float &x) {
    const bool poisoned = __builtin_isunordered(a, v);
    const double dv = min64u(max64u(div_tau(a - v, tau, rtau, recip_ok), -amax), amax);  // AG:26
    double nv = min64u(max64u(v + dv * tau, -vmax), vmax);                               // AG:27
    nv = poisoned ? __builtin_nan("") : nv;
    v = nv;
    x = (float)((double)x + v * tau);                                                // AG:28-29 (float32 array += float64 array)
}

// sin(pi*t), cos(pi*t) in float32: quarter-turn reduction + Taylor polynomials on |r| <= 1/4
// (|error| < 1e-7).  Fixed fmaf sequence so the CPU oracle can restate it bit for bit.
__device__ __forceinline__ void sincospi32(float t, float &sn, float &cs) {
    const float k = rintf(2.0f * t);
    const float r = fmaf(-0.5f, k, t);
    const float z = r * r;
    float ps = fmaf(z, 0.0821458866f, -0.599264529f);    //  pi^9/9!, -pi^7/7!
    ps = fmaf(ps, z, 2.55016404f);                       //  pi^5/5!
    ps = fmaf(ps, z, -5.16771278f);                      // -pi^3/3!
    ps = fmaf(ps, z, 3.14159265f);
    ps = ps * r;
    float pc = fmaf(z, -0.0258068913f, 0.235330630f);    // -pi^10/10!, pi^8/8!
    pc = fmaf(pc, z, -1.33526277f);                      // -pi^6/6!
    pc = fmaf(pc, z, 4.05871213f);                       //  pi^4/4!
    pc = fmaf(pc, z, -4.93480220f);                      // -pi^2/2!
    pc = fmaf(pc, z, 1.0f);
    const int q = (int)k & 3;
    sn = (q == 0) ? ps : (q == 1) ? pc : (q == 2) ? -ps : -pc;
    cs = (q == 0) ? pc : (q == 1) ? -ps : (q == 2) ? -pc : ps;
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
// Candidates of the multi-agent reset.  Agent i of an env owns the Philox sequence with counter
// (global env [31:0], global env [47:32] | i << 16, attempt k, episode); call k yields the k-th START
// candidate from words 0,1 and the k-th TARGET candidate from words 2,3 (32-bit uniforms: 1e-8 m
// granularity over a 50 m box, far below float32 resolution).  Independent per-agent sequences let
// the lanes of an env draw in parallel while the accept/reject chain (MUW:127-153) keeps agent order.
// The reset path runs in a few waves per launch only, so its instructions are never warm in the
// instruction cache: code SIZE is what it costs (measured: an unrolled, 3x-inlined version added 3.6 us
// of tail to a 7 us step launch with 0.2 % of the envs resetting).  Hence rolled rounds; and inlined, because a
// real call makes the callee wait for vmcnt(0), i.e. for every load the wave still has in flight.
struct ResetCandidates {
    float sx, sy, tx, ty;
};
// the four Philox4x32-10 output words of counter (env[31:0], env[47:32] | agent << 16, attempt, episode); rolled rounds
__device__ __forceinline__ void reset_words(uint64_t global_env, uint32_t agent, uint32_t attempt, uint32_t episode,
                                            uint32_t k0, uint32_t k1, uint32_t o[4]) {
    uint32_t c0 = (uint32_t)global_env, c1 = ((uint32_t)(global_env >> 32) & 0xFFFFu) | (agent << 16);
    uint32_t c2 = attempt, c3 = episode;
#pragma unroll 1
    for (int r = 0; r < 10; r++) {  // Philox4x32-10, same rounds as philox4x32() above
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
__device__ __forceinline__ ResetCandidates reset_candidates(uint64_t global_env, uint32_t agent, uint32_t attempt,
                                                            uint32_t episode, uint32_t k0, uint32_t k1, double lox,
                                                            double loy, double hix, double hiy) {
    uint32_t o[4];
    reset_words(global_env, agent, attempt, episode, k0, k1, o);
    const double sx = hix - lox, sy = hiy - loy, inv = 1.0 / 4294967296.0;
    ResetCandidates c;  // lo + (hi-lo)*U cast to float32 like np.random.uniform(lo, hi).astype(np.float32)
    c.sx = (float)(lox + sx * ((double)o[0] * inv));
    c.sy = (float)(loy + sy * ((double)o[1] * inv));
    c.tx = (float)(lox + sx * ((double)o[2] * inv));
    c.ty = (float)(loy + sy * ((double)o[3] * inv));
    return c;
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
