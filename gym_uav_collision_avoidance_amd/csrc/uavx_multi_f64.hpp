// uavx_multi_f64.hpp — float64-POSITION mode of MultiUAVWorld2D (included by uavx_multi.hip inside namespace uavx).
//
// The reference keeps agent.location / target_location as float32 arrays after reset() (MUW:126,131,144), but
// reset(circular=True) (MUW:157-163) and callers that assign their own arrays (test_sac_multi_plot_trajectory.py:
// 43-49) replace them by float64 arrays, and from then on every position expression of that episode is float64:
// `location += velocity*tau` accumulates in double (AG:29), the norms are float64 (AG:33,51; numpy's 2-element
// float64 dot is sqrt(fma(y,y,x*x))), the d_sense / 2R / 1.0 / 0.5 tests compare doubles, prev_distance -
// distance and distance/(1.5*init_distance) are float64 (MUW:190-194).  This file restates THAT episode type.
//
// It is the evaluation / trajectory-plot scenario, not the training hot path (a deterministic layout makes every
// env of a batch identical unless the policy is stochastic), so the mapping is the plain one: ONE THREAD PER ENV,
// the reference's sequential agent loop as written (Gauss-Seidel falls out of updating the position array in
// place), state in separate float64 arrays.  Masks, positions, velocities, prev/init distances and counters are
// bit-exact against the reference (sqrt, fma, +,-,*,/ are IEEE on both sides); observation / reward VALUES go
// through the device's float64 atan2/sin/cos (ocml, <= 2 ulp from glibc's) and are returned as float32.
#pragma once

struct WideState {          // [A] arrays, allocated when the mode is first entered (uavx_set_position_mode)
    double2 *pos, *tgt;
    double *init_d, *prev_d;
};

struct WideLimits {         // the float64 comparands of AG:52, MUW:203,207 (python floats in the reference)
    double d_sense, two_r, two_hard, vmax_norm, diag;
};

__device__ __forceinline__ double nrm64(double x, double y) { return sqrt(fma(y, y, x * x)); }
__device__ __forceinline__ double clip64_np(double x, double lo, double hi) {  // np.clip: NaN propagates
    double m = (x < lo) ? lo : x;
    m = (x != x) ? x : m;
    double r = (m > hi) ? hi : m;
    return (m != m) ? m : r;
}
__device__ __forceinline__ double wrap64(double d) { return atan2(sin(d), cos(d)); }  // MUW:71,80,84,186

// AG:44-64 as its callers use it: the (up to) two nearest other agents strictly within d_sense at the positions
// currently in pos[], ascending, ties -> lower index.
__device__ inline int nearest_two64(const double2 *pos, int n, int self, double d_sense, int idx[2], double dist[2]) {
    int cnt = 0;
    const double2 me = pos[self];
    for (int j = 0; j < n; j++) {
        if (j == self) continue;
        const double2 q = pos[j];
        const double d = nrm64(q.x - me.x, q.y - me.y);            // AG:51
        if (!(d < d_sense)) continue;                              // AG:52
        if (cnt == 0) { idx[0] = j; dist[0] = d; cnt = 1; }
        else if (d < dist[0]) { idx[1] = idx[0]; dist[1] = dist[0]; idx[0] = j; dist[0] = d; cnt = 2; }
        else if (cnt == 1 || d < dist[1]) { idx[1] = j; dist[1] = d; cnt = 2; }
    }
    return cnt;
}

// MUW:60-109 for agent i (float64 throughout, stored as float32)
__device__ inline void observe_agent64(const MultiParams &p, const WideLimits &L, const double2 *pos, const double2 *tgt,
                                       const double2 *vel, int n, int i, float *o) {
    const double kPi64 = 3.14159265358979323846;
    const double2 v = vel[i], me = pos[i], t = tgt[i];
    const double theta = atan2(v.y, v.x);                                           // MUW:63
    o[0] = (float)(nrm64(v.x, v.y) / L.vmax_norm);                                  // MUW:62
    o[1] = (float)(theta / kPi64);                                                  // MUW:64
    o[2] = (float)(nrm64(t.x - me.x, t.y - me.y) / L.diag);                         // MUW:67-68
    o[3] = (float)(wrap64(atan2(t.y - me.y, t.x - me.x) - theta) / kPi64);          // MUW:69-72
    int idx[2]; double dist[2];
    const int cnt = nearest_two64(pos, n, i, L.d_sense, idx, dist);                 // MUW:75
    for (int k = 0; k < 2; k++) {
        double nd = 1.0, rel = kPi64 + theta, dir = theta;                          // MUW:96-108 defaults
        if (cnt > k) {
            const double2 q = pos[idx[k]], w = vel[idx[k]];
            nd = dist[k] / L.d_sense;                                               // MUW:77,87
            rel = atan2(q.y - me.y, q.x - me.x);                                    // MUW:78,88
            dir = atan2(w.y, w.x);                                                  // MUW:82,92
        }
        o[4 + 3 * k] = (float)nd;
        o[5 + 3 * k] = (float)(wrap64(rel - theta) / kPi64);                        // MUW:79-81
        o[6 + 3 * k] = (float)(wrap64(dir - theta) / kPi64);                        // MUW:83-85
    }
}

__global__ __launch_bounds__(kBlock) void observe64_kernel(MultiParams p, WideState w, WideLimits L, float *obs) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    const int n = p.N;
    const int64_t a0 = e * n;
    for (int i = 0; i < n; i++) observe_agent64(p, L, w.pos + a0, w.tgt + a0, p.vel + a0, n, i, obs + (a0 + i) * 10);
}

// MUW:177-241 for one env per thread.  action_mode / track_returns as in uavx_step_ex (no auto-reset in this mode).
template <bool ACT64>
__global__ __launch_bounds__(kBlock) void step64_kernel(MultiParams p, WideState w, WideLimits L, const void *actions,
                                                        int action_mode, int track_returns, int evaluate, float *obs,
                                                        float *rew_out, uint8_t *done_out) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= p.E) return;
    const int n = p.N;
    const int64_t a0 = e * n;
    double2 *pos = w.pos + a0, *vel = p.vel + a0;
    const double2 *tgt = w.tgt + a0;
    uint32_t reach = 0, coll = 0;
    float run0 = 0.f, score = 0.f;
    for (int i = 0; i < n; i++) {                                                    // MUW:181
        double ax, ay;
        load_action<ACT64>(actions, (uint32_t)(a0 + i), ax, ay);
        if (action_mode == UAVX_ACTION_POLAR) polar_to_command(p, (float)ax, (float)ay, ax, ay);
        uint32_t flags = p.goal[a0 + i].flags & kFlagPublic;
        const uint32_t flags_in = flags;
        const bool was_done = (flags & UAVX_FLAG_DONE) != 0;
        double2 v = vel[i], x = pos[i];
        const double2 t = tgt[i];
        const double init_d = w.init_d[a0 + i];
        double pd = 0.0, d = 0.0;                                                    // AG:24-25
        if (!was_done) {                                                             // AG:26-36
            const double dvx = clip64_np((ax - v.x) / p.tau, -p.amax, p.amax), dvy = clip64_np((ay - v.y) / p.tau, -p.amax, p.amax);
            v.x = clip64_np(v.x + dvx * p.tau, -p.vmax, p.vmax);
            v.y = clip64_np(v.y + dvy * p.tau, -p.vmax, p.vmax);
            x.x = x.x + v.x * p.tau;                                                 // AG:28-29, float64 array
            x.y = x.y + v.y * p.tau;
            pos[i] = x;
            pd = w.prev_d[a0 + i];                                                   // AG:32
            d = nrm64(t.x - x.x, t.y - x.y);                                         // AG:33
        }
        // reward shaping, MUW:183-195
        const double dth = wrap64(atan2(t.y - x.y, t.x - x.x) - atan2(v.y, v.x));    // MUW:184-186
        const double q = L.vmax_norm / init_d;
        double r = 0.0 - 0.01 * ((1.0 < q) ? 1.0 : q);                               // MUW:188-189
        r += 50.0 * ((pd - d) / L.vmax_norm);                                        // MUW:190
        const double frac = d / (1.5 * init_d);
        r *= (r > 0) ? (1 - frac) : (1 + frac);                                      // MUW:191-194
        r -= 0.01 * fabs(dth);                                                       // MUW:195
        // collisions with the <= 2 nearest in-range agents at the CURRENT array (j<i moved, j>i not), MUW:197-210
        bool collision = false;
        int idx[2]; double dist[2];
        const int nn = nearest_two64(pos, n, i, L.d_sense, idx, dist);
        for (int k = 0; k < nn; k++) {
            if (dist[k] <= L.two_r) { r = -2.0; collision = true; }                  // MUW:203-205
            if (dist[k] <= L.two_hard && !(flags & (UAVX_FLAG_DONE | UAVX_FLAG_COLLIDED))) {
                coll += 1; flags |= UAVX_FLAG_COLLIDED;                              // MUW:207-210
            }
        }
        // termination, MUW:213-227
        const bool oob = !(x.x >= p.lox && x.x <= p.hix && x.y >= p.loy && x.y <= p.hiy);
        uint32_t dn = 0;
        if (d < 0.5 && !collision && nrm64(v.x, v.y) < 0.2) {                        // MUW:218
            dn = 1;
            if (!(flags & UAVX_FLAG_DONE)) reach += 1;                               // MUW:220-221
            flags |= UAVX_FLAG_DONE;                                                 // AG:38-42
            const double nv = nrm64(v.x, v.y);
            double fx = v.x / nv * 0.001, fy = v.y / nv * 0.001;
            if (fx != fx || fy != fy) { fx = 0.0; fy = 0.0; }
            v = make_double2(fx, fy);
            r += 10;                                                                 // MUW:223
        } else if (oob) {
            dn = evaluate ? 0u : 1u;                                                 // MUW:224-225
        }
        vel[i] = v;
        w.prev_d[a0 + i] = d;                                                        // MUW:229
        if (flags != flags_in) p.goal[a0 + i].flags = flags;
        const float rf = (float)r;
        rew_out[a0 + i] = rf;
        done_out[a0 + i] = (uint8_t)dn;
        if (i == 0) run0 = rf;
        score += rf * (1.0f - (float)dn);                                            // test_sac_multi.py:157
    }
    for (int i = 0; i < n; i++) observe_agent64(p, L, pos, tgt, vel, n, i, obs + (a0 + i) * 10);  // MUW:233-235
    if (reach) p.reach[e] += reach;
    if (coll) p.coll[e] += coll;
    uint4 rec = p.env_rec[e];
    rec.x -= 1u;                                                                     // MUW:238: steps = wave_steps - rec.x
    if (track_returns) {
        rec.z = __float_as_uint(__uint_as_float(rec.z) + run0);                      // test_sac_multi.py:106
        rec.w = __float_as_uint(__uint_as_float(rec.w) + score);
    }
    p.env_rec[e] = rec;
}

// Representation change of every agent: float32 record -> float64 arrays (exact widening; prev_distance is the
// value the float32 mode would use next), or back (round to nearest float32; prev_distance parked behind the
// override bit when it is not the derived one, exactly like uavx_set_state does).
__global__ __launch_bounds__(kBlock) void widen_state_kernel(MultiParams p, WideState w) {
    const int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (a >= p.E * p.N) return;
    const float2 d = p.pos[a];
    const Goal g = p.goal[a];
    w.pos[a] = make_double2((double)d.x, (double)d.y);
    w.tgt[a] = make_double2((double)g.tx, (double)g.ty);
    w.init_d[a] = (double)g.init_d;
    w.prev_d[a] = (double)((g.flags & kFlagPrevOvr) ? p.prev_ovr[a] : natural_prev_d(g.flags, d.x, d.y, g.tx, g.ty));
    p.goal[a].flags = g.flags & kFlagPublic;
}
__global__ __launch_bounds__(kBlock) void narrow_state_kernel(MultiParams p, WideState w) {
    const int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (a >= p.E * p.N) return;
    const double2 x = w.pos[a], t = w.tgt[a];
    Goal g;
    g.tx = (float)t.x; g.ty = (float)t.y; g.init_d = (float)w.init_d[a];
    uint32_t flags = p.goal[a].flags & kFlagPublic;
    const float2 d = make_float2((float)x.x, (float)x.y);
    const float want = (float)w.prev_d[a];
    if (__float_as_uint(want) != __float_as_uint(natural_prev_d(flags, d.x, d.y, g.tx, g.ty))) {
        flags |= kFlagPrevOvr;
        p.prev_ovr[a] = want;
    }
    g.flags = flags;
    p.pos[a] = d;
    p.goal[a] = g;
}

// uavx_get_state / uavx_set_state (float32 views) and their _f64 siblings while the handle is in float64 mode.
// which: 0 get, 1 set.  Velocity / flags / counters live in the shared arrays and go through the float32-mode
// kernels' code (flags without the private bits).
__global__ __launch_bounds__(kBlock) void wide_exchange_kernel(MultiParams p, WideState w, uavx_state_view v,
                                                               uavx_state_view_f64 v64, int set) {
    const int64_t a = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (a >= p.E * p.N) return;
    if (set) {
        if (v.loc) w.pos[a] = make_double2((double)v.loc[2 * a], (double)v.loc[2 * a + 1]);
        if (v.tgt) w.tgt[a] = make_double2((double)v.tgt[2 * a], (double)v.tgt[2 * a + 1]);
        if (v.init_d) w.init_d[a] = (double)v.init_d[a];
        if (v.prev_d) w.prev_d[a] = (double)v.prev_d[a];
        if (v64.loc) w.pos[a] = make_double2(v64.loc[2 * a], v64.loc[2 * a + 1]);
        if (v64.tgt) w.tgt[a] = make_double2(v64.tgt[2 * a], v64.tgt[2 * a + 1]);
        if (v64.init_d) w.init_d[a] = v64.init_d[a];
        if (v64.prev_d) w.prev_d[a] = v64.prev_d[a];
        if (v.flags) p.goal[a].flags = (uint32_t)v.flags[a] & kFlagPublic;
        if (v.vel) p.vel[a] = make_double2(v.vel[2 * a], v.vel[2 * a + 1]);
    } else {
        const double2 x = w.pos[a], t = w.tgt[a];
        if (v.loc) { v.loc[2 * a] = (float)x.x; v.loc[2 * a + 1] = (float)x.y; }
        if (v.tgt) { v.tgt[2 * a] = (float)t.x; v.tgt[2 * a + 1] = (float)t.y; }
        if (v.init_d) v.init_d[a] = (float)w.init_d[a];
        if (v.prev_d) v.prev_d[a] = (float)w.prev_d[a];
        if (v64.loc) { v64.loc[2 * a] = x.x; v64.loc[2 * a + 1] = x.y; }
        if (v64.tgt) { v64.tgt[2 * a] = t.x; v64.tgt[2 * a + 1] = t.y; }
        if (v64.init_d) v64.init_d[a] = w.init_d[a];
        if (v64.prev_d) v64.prev_d[a] = w.prev_d[a];
        if (v.flags) v.flags[a] = (uint8_t)(p.goal[a].flags & kFlagPublic);
        if (v.vel) { const double2 q = p.vel[a]; v.vel[2 * a] = q.x; v.vel[2 * a + 1] = q.y; }
    }
}
