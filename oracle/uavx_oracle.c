/* uavx_oracle.c — CPU restatement of the reference step/reset path.  TEST INFRASTRUCTURE ONLY.
 *
 * Every function cites the reference lines it follows (MUW / AG / UW, see uavx_oracle.h).  The
 * arithmetic reproduces what numpy 2.2.6 (NEP 50 promotion) + CPython's math module (glibc libm)
 * evaluate for those lines; build with -ffp-contract=off so no FMA is formed except where numpy
 * itself uses one (the float64 2-element dot inside np.linalg.norm, see nrm64()).
 * Pinned by tests/test_oracle_golden.py against fixtures generated from the reference.
 */
#include "uavx_oracle.h"

#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define HARD_COLLISION_RADIUS 0.5 /* MUW:8 */
#define MAXN 64

/* ------------------------------------------------------------------------------------------------
 * numpy building blocks
 * ---------------------------------------------------------------------------------------------- */

/* np.linalg.norm of a 2-element float64 vector = sqrt(dot(x,x)); the OpenBLAS ddot numpy 2.2.6
 * ships evaluates the 2-term dot as fma(y,y,x*x) (checked against numpy on 20 000 samples, 0
 * mismatches; the plain x*x+y*y form differs in ~8 % of samples by 1 ulp). */
static inline double nrm64(double x, double y) { return sqrt(fma(y, y, x * x)); }

/* np.linalg.norm of a 2-element float32 vector: float32 mul, mul, add, sqrtf, no FMA
 * (200 000-sample check in SURVEY.md §0.5). */
static inline float nrm32(float x, float y) {
    float a = x * x;
    float b = y * y;
    float s = a + b;
    return sqrtf(s);
}

/* np.clip(x, lo, hi) == minimum(maximum(x, lo), hi) */
static inline double clipd(double x, double lo, double hi) {
    double m = (x < lo) ? lo : x; /* NaN stays NaN like np.maximum */
    if (x != x) m = x;
    double r = (m > hi) ? hi : m;
    if (m != m) r = m;
    return r;
}

/* atan2(sin d, cos d): the reference's angle wrap (MUW:71,80,84,90,94,186; UW:93,156) */
static inline double wrap_angle(double d) { return atan2(sin(d), cos(d)); }

/* ‖a − b‖ as the reference's position dtype evaluates it (float32 arrays unless circular). */
static inline double pos_dist(int f64pos, double ax, double ay, double bx, double by) {
    if (f64pos) return nrm64(ax - bx, ay - by);
    float dx = (float)ax - (float)bx;
    float dy = (float)ay - (float)by;
    return (double)nrm32(dx, dy);
}
/* component of (a − b) as a Python float handed to math.atan2 */
static inline double pos_sub(int f64pos, double a, double b) {
    if (f64pos) return a - b;
    return (double)((float)a - (float)b);
}

/* ------------------------------------------------------------------------------------------------
 * RNGs
 * ---------------------------------------------------------------------------------------------- */
void uavo_mt_seed(uavo_mt *g, uint32_t seed) { /* numpy legacy np.random.seed(int) == init_genrand */
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}
static uint32_t mt_next(uavo_mt *g) {
    if (g->idx >= 624) {
        uint32_t *mt = g->mt;
        for (int k = 0; k < 624; k++) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            mt[k] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
static inline double bits53(uint32_t a, uint32_t b) { /* numpy random_sample: 53-bit double */
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
double uavo_mt_double(uavo_mt *g) {
    uint32_t a = mt_next(g), b = mt_next(g);
    return bits53(a, b);
}

void uavo_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
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

/* A draw source yields pairs of U[0,1) doubles.  MT path: the process-global stream in the order
 * np.random.uniform(lo,hi,(2,)) consumes it (the reference).  Philox path (the device resets):
 *  - `addressed` == 0 (UAVWorld2D): counter (env_lo, env_hi, running draw index, episode);
 *  - `addressed` == 1 (MultiUAVWorld2D): every agent owns a candidate sequence, counter
 *    (env_lo, env_hi[15:0] | agent << 16, attempt, episode); call k gives the k-th START candidate
 *    from words 0,1 and the k-th TARGET candidate from words 2,3 as 32-bit uniforms, so the device
 *    can draw the agents of an env in parallel (uavx_device.hpp reset_candidates()). */
typedef struct {
    uavo_mt *mt;          /* MT path */
    uint32_t key[2];      /* Philox path */
    uint32_t ctr_env[2], episode, draw;
    int addressed;
    uint32_t agent, kind, attempt[2];
} draw_src;

static void draw_pair(draw_src *s, double u[2]) {
    if (s->mt) {
        u[0] = uavo_mt_double(s->mt);
        u[1] = uavo_mt_double(s->mt);
    } else {
        uint32_t ctr[4] = {s->ctr_env[0], s->ctr_env[1], s->draw++, s->episode}, o[4];
        if (s->addressed) {
            ctr[1] = (s->ctr_env[1] & 0xFFFFu) | (s->agent << 16);
            ctr[2] = s->attempt[s->kind]++;
            uavo_philox4x32(ctr, s->key, o);
            u[0] = (double)o[2 * s->kind] * (1.0 / 4294967296.0);
            u[1] = (double)o[2 * s->kind + 1] * (1.0 / 4294967296.0);
            return;
        }
        uavo_philox4x32(ctr, s->key, o);
        u[0] = bits53(o[0], o[1]);
        u[1] = bits53(o[2], o[3]);
    }
}
/* positions the addressed Philox source at the first candidate of (agent, kind); no-op for MT */
static void draw_begin(draw_src *s, uint32_t agent, uint32_t kind) {
    s->agent = agent; s->kind = kind; s->attempt[kind] = 0;
}
/* np.random.uniform(low, high, (2,)).astype(np.float32): low + (high-low)*random_sample, then cast */
static void draw_point32(draw_src *s, double lox, double loy, double hix, double hiy, double p[2]) {
    double u[2];
    draw_pair(s, u);
    p[0] = (double)(float)(lox + (hix - lox) * u[0]);
    p[1] = (double)(float)(loy + (hiy - loy) * u[1]);
}

/* ------------------------------------------------------------------------------------------------
 * MultiUAVWorld2D
 * ---------------------------------------------------------------------------------------------- */

/* Per-env view used by every MultiUAVWorld2D routine below: the env's EFFECTIVE config (cfg with its curriculum level
 * applied) and, for the configs[4] extension, its scripted bodies.  Without an extension this is just cfg. */
typedef struct {
    uavo_config cfg;
    int L;               /* learner slots (st->num_agents) */
    int nl;              /* learners taking part (level.n_active) */
    int B, nb;           /* body slots / bodies taking part */
    float *body;         /* [B*6] x, y, dx, dy, heading, legs of this env (uavx.h, uavx_set_body_rule), or NULL */
    float body_step;     /* float32(body_speed * tau): distance a body covers per env step */
    int period;
    uint32_t key[2], env_ctr[2];
} envx;

static void make_envx(const uavo_config *cfg, const uavo_ext *ext, const uavo_ext_state *xs, const uavo_state *st,
                      int64_t e, int64_t env_offset, envx *x) {
    memset(x, 0, sizeof *x);
    x->cfg = *cfg;
    x->L = x->nl = st->num_agents;
    if (!ext || !xs) return;
    x->B = x->nb = ext->num_bodies;
    x->body = (ext->num_bodies > 0) ? xs->body + (size_t)e * ext->num_bodies * UAVO_BODY_DIM : NULL;
    x->body_step = (float)(ext->body_speed * cfg->tau);
    x->period = ext->body_period > 0 ? ext->body_period : 1;
    x->key[0] = (uint32_t)ext->body_seed; x->key[1] = (uint32_t)(ext->body_seed >> 32);
    const uint64_t ge = (uint64_t)(env_offset + e);
    x->env_ctr[0] = (uint32_t)ge; x->env_ctr[1] = (uint32_t)(ge >> 32);
    if (ext->n_levels > 0) {
        const uavo_level *lv = &ext->levels[xs->level[e] < ext->n_levels ? xs->level[e] : ext->n_levels - 1];
        x->cfg.x_size = lv->x_size; x->cfg.y_size = lv->y_size;
        x->cfg.collider_radius = lv->collider_radius; x->cfg.d_sense = lv->d_sense;
        x->nl = lv->n_active < 1 ? 1 : (lv->n_active > x->L ? x->L : lv->n_active);
        x->nb = lv->b_active < 0 ? 0 : (lv->b_active > x->B ? x->B : lv->b_active);
    }
}

/* Waypoint `leg` of body b in the episode whose reset drew with episode index `ep_draw`: Philox counter
 * (env[31:0], env[47:32] | slot << 16, 0x80000000 | leg, ep_draw) -- the reset candidates of the same slot use
 * counter word 2 = attempt < 2^31 -- words 0,1 as 32-bit uniforms over the env's box, cast to float32. */
static void body_waypoint(const envx *x, int b, uint32_t leg, uint32_t ep_draw, float wp[2]) {
    const uint32_t ctr[4] = {x->env_ctr[0], (x->env_ctr[1] & 0xFFFFu) | ((uint32_t)(x->L + b) << 16), 0x80000000u | leg, ep_draw};
    uint32_t o[4];
    uavo_philox4x32(ctr, x->key, o);
    const double lox = -x->cfg.x_size / 2.0, loy = -x->cfg.y_size / 2.0, sx = x->cfg.x_size / 2.0 - lox, sy = x->cfg.y_size / 2.0 - loy;
    wp[0] = (float)(lox + sx * ((double)o[0] * (1.0 / 4294967296.0)));
    wp[1] = (float)(loy + sy * ((double)o[1] * (1.0 / 4294967296.0)));
}

/* atan2f for the heading of a leg: the device's octant reduction + Cephes polynomial (csrc/uavx_device.hpp,
 * atan2_exact) with an IEEE division, every operation correctly rounded and in the same order, so that the stored
 * heading is the same float32 on both sides. */
static float atan2_leg(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const int big = mn > 0.41421356237f * mx;
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
    r = (x < 0.f) ? 3.14159265358979323846f - r : r;
    return copysignf(r, y);
}

/* A body starts a leg at its position (r[0], r[1]) towards waypoint wp (all float32, no FMA): displacement per env
 * step = (wp - P) * (body_step / ||wp - P||), heading = direction of travel, legs = number of whole steps that fit
 * into the distance (the body stops less than one step short of the waypoint and waits there for the next leg). */
static void body_leg(const envx *x, float *r, const float wp[2]) {
    const float dx = wp[0] - r[0], dy = wp[1] - r[1];
    const float d = nrm32(dx, dy);
    if (d > 0.f) {
        const float sc = x->body_step / d;
        r[2] = dx * sc; r[3] = dy * sc;
        r[5] = floorf(d / x->body_step);       /* body_step == 0 (static obstacle): +inf legs of zero displacement */
    } else {
        r[2] = r[3] = 0.f; r[5] = 0.f;
    }
    r[4] = atan2_leg(dy, dx);
}

/* One env step of body b: at the start of every `period`-th step it takes a new waypoint and starts a leg from
 * where it is; it then moves by its displacement while the leg still has whole steps left. */
static void body_move(const envx *x, int b, uint32_t steps_before, uint32_t ep_draw) {
    float *r = x->body + UAVO_BODY_DIM * b;
    const uint32_t k = steps_before % (uint32_t)x->period;
    if (steps_before != 0 && k == 0) {
        float wp[2];
        body_waypoint(x, b, steps_before / (uint32_t)x->period, ep_draw, wp);
        body_leg(x, r, wp);
    }
    if ((float)k < r[5]) { r[0] = r[0] + r[2]; r[1] = r[1] + r[3]; }
}

/* positions (and, for the observation, velocities) of every slot of the neighbour model: learners 0..L-1 then
 * bodies L..L+B-1; a slot that does not take part sits at +inf and is never within d_sense. */
static int gather_slots(const envx *x, const double *loc, const double *vel, double *px, double *py, double *vx, double *vy) {
    for (int j = 0; j < x->L; j++) {
        const int on = j < x->nl;
        px[j] = on ? loc[2 * j] : INFINITY; py[j] = on ? loc[2 * j + 1] : INFINITY;
        if (vx) { vx[j] = vel[2 * j]; vy[j] = vel[2 * j + 1]; }
    }
    for (int b = 0; b < x->B; b++) {
        const int j = x->L + b, on = b < x->nb;
        const float *r = x->body + UAVO_BODY_DIM * b;
        px[j] = on ? (double)r[0] : INFINITY; py[j] = on ? (double)r[1] : INFINITY;
        if (vx) { vx[j] = (double)r[4]; vy[j] = 0.0; }  /* a body's heading is the stored float32 angle (slots >= L: see observe_agent) */
    }
    return x->L + x->B;
}

/* AG:44-64 restricted to what the callers use (MUW:75-95,198-199): the (up to) two nearest other
 * agents strictly within d_sense, ascending by distance, ties -> lower index (argsort on <16
 * elements is a stable insertion sort).  px/py are the positions to use for each agent. */
static int nearest_two(int n, int self, int f64pos, const double *px, const double *py,
                       double d_sense, int idx[2], double dist[2]) {
    int cnt = 0;
    double lim = f64pos ? d_sense : (double)(float)d_sense; /* f32 < python scalar -> f32 compare */
    for (int j = 0; j < n; j++) {
        if (j == self) continue;
        double d = pos_dist(f64pos, px[j], py[j], px[self], py[self]); /* AG:51 */
        if (!(d < lim)) continue;                                        /* AG:52 */
        if (cnt == 0) {
            idx[0] = j; dist[0] = d; cnt = 1;
        } else if (d < dist[0]) {
            idx[1] = idx[0]; dist[1] = dist[0];
            idx[0] = j; dist[0] = d; cnt = 2;
        } else if (cnt == 1 || d < dist[1]) {
            idx[1] = j; dist[1] = d; cnt = 2;
        }
    }
    return cnt;
}

/* MUW:60-109 for agent i of env e. */
static void observe_agent(const envx *x, const uavo_state *st, int64_t e, int i, double *o) {
    const uavo_config *cfg = &x->cfg;
    const int n = st->num_agents;
    const int f64pos = st->f64pos[e];
    const double *loc = st->loc + e * n * 2, *vel = st->vel + e * n * 2, *tgt = st->tgt + e * n * 2;
    double px[MAXN], py[MAXN], nvx[MAXN], nvy[MAXN];
    const int ntot = gather_slots(x, loc, vel, px, py, nvx, nvy);
    if (i >= x->nl) { /* parked learner (extension): all-zero observation */
        for (int k = 0; k < UAVO_OBS_DIM; k++) o[k] = 0.0;
        return;
    }

    const double vx = vel[2 * i], vy = vel[2 * i + 1];
    o[0] = nrm64(vx, vy) / nrm64(cfg->max_speed, cfg->max_speed);                 /* MUW:62 */
    const double theta = atan2(vy, vx);                                             /* MUW:63 */
    o[1] = theta / M_PI;                                                            /* MUW:64 */
    const double rtd = pos_dist(f64pos, tgt[2 * i], tgt[2 * i + 1], px[i], py[i]);  /* MUW:67 */
    o[2] = rtd / nrm64(cfg->x_size, cfg->y_size);                                   /* MUW:68,17 */
    const double rtt = atan2(pos_sub(f64pos, tgt[2 * i + 1], py[i]),
                             pos_sub(f64pos, tgt[2 * i], px[i]));                   /* MUW:69 */
    o[3] = wrap_angle(rtt - theta) / M_PI;                                          /* MUW:70-72 */

    int idx[2]; double dist[2];
    const int cnt = nearest_two(ntot, i, f64pos, px, py, cfg->d_sense, idx, dist);  /* MUW:75 */
    for (int k = 0; k < 2; k++) {
        double nd, rel_theta, dir;
        if (cnt > k) {
            const int j = idx[k];
            /* MUW:77/87: float32 norm / python scalar d_sense -> float32 division */
            nd = f64pos ? dist[k] / cfg->d_sense : (double)((float)dist[k] / (float)cfg->d_sense);
            rel_theta = atan2(pos_sub(f64pos, py[j], py[i]), pos_sub(f64pos, px[j], px[i])); /* MUW:78/88 */
            dir = (j >= x->L) ? nvx[j] : atan2(nvy[j], nvx[j]);                     /* MUW:82/92; a body (slot >= L): its stored heading */
        } else {
            nd = 1.0;
            rel_theta = M_PI + theta;
            dir = theta;
        }
        o[4 + 3 * k] = nd;
        o[5 + 3 * k] = wrap_angle(rel_theta - theta) / M_PI;                        /* MUW:79-81 */
        o[6 + 3 * k] = wrap_angle(dir - theta) / M_PI;                              /* MUW:83-85 */
    }
}

static void observe_env(const envx *x, const uavo_state *st, int64_t e, double *obs) {
    for (int i = 0; i < st->num_agents; i++)
        observe_agent(x, st, e, i, obs + (e * st->num_agents + i) * UAVO_OBS_DIM);
}

void uavo_observe_x(const uavo_config *cfg, const uavo_ext *ext, const uavo_state *st, const uavo_ext_state *xs,
                    double *obs, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < st->num_envs; e++) {
        envx x;
        make_envx(cfg, ext, xs, st, e, 0, &x);
        observe_env(&x, st, e, obs);
    }
}
void uavo_observe(const uavo_config *cfg, const uavo_state *st, double *obs, int nthreads) {
    uavo_observe_x(cfg, NULL, st, NULL, obs, nthreads);
}

/* MUW:116-168 (state part of reset; observations are produced by uavo_observe).  Extension: learners >= x->nl are
 * parked (flag INACTIVE, position +inf) and take no part in the draws; the bodies taking part draw their start points
 * after the learners' (slot order) under the same > 2R rule and take waypoint leg 0 as their first goal. */
static void reset_env(const envx *x, uavo_state *st, int64_t e, draw_src *src, int circular) {
    const uavo_config *cfg = &x->cfg;
    const int n = st->num_agents, nl = x->nl;
    double *loc = st->loc + e * n * 2, *vel = st->vel + e * n * 2, *tgt = st->tgt + e * n * 2;
    double *init_d = st->init_d + e * n, *prev_d = st->prev_d + e * n;
    uint8_t *flags = st->flags + e * n;
    const double lox = -cfg->x_size / 2.0, loy = -cfg->y_size / 2.0;   /* MUW:19 */
    const double hix = cfg->x_size / 2.0, hiy = cfg->y_size / 2.0;     /* MUW:20 */
    const float two_r = (float)(2 * cfg->collider_radius);             /* f32 <= python float */

    for (int i = 0; i < n; i++) { vel[2 * i] = vel[2 * i + 1] = 0.0; flags[i] = 0; } /* MUW:118-123 */
    st->f64pos[e] = 0;

    draw_begin(src, 0, 0);
    draw_point32(src, lox, loy, hix, hiy, loc);                        /* MUW:126 */
    for (int i = 1; i < nl; i++) {                                     /* MUW:127-137 */
        int replicated = 1;
        draw_begin(src, (uint32_t)i, 0);
        while (replicated) {
            draw_point32(src, lox, loy, hix, hiy, loc + 2 * i);
            replicated = 0;
            for (int j = 0; j < i; j++) {
                if ((float)pos_dist(0, loc[2 * j], loc[2 * j + 1], loc[2 * i], loc[2 * i + 1]) <= two_r) {
                    replicated = 1;
                    break;
                }
            }
        }
    }
    for (int b = 0; b < x->B; b++) {                                   /* extension: body start points */
        float *r = x->body + UAVO_BODY_DIM * b;
        if (b >= x->nb) { r[0] = r[1] = INFINITY; r[2] = r[3] = r[4] = r[5] = 0.f; continue; }
        int replicated = 1;
        double q[2];
        draw_begin(src, (uint32_t)(x->L + b), 0);
        while (replicated) {
            draw_point32(src, lox, loy, hix, hiy, q);
            replicated = 0;
            for (int j = 0; j < nl && !replicated; j++)
                if ((float)pos_dist(0, loc[2 * j], loc[2 * j + 1], q[0], q[1]) <= two_r) replicated = 1;
            for (int j = 0; j < b && !replicated; j++)
                if ((float)pos_dist(0, (double)x->body[UAVO_BODY_DIM * j], (double)x->body[UAVO_BODY_DIM * j + 1], q[0], q[1]) <= two_r) replicated = 1;
        }
        r[0] = (float)q[0]; r[1] = (float)q[1];
        float wp[2];
        body_waypoint(x, b, 0u, src->episode & 0x7FFFFFFFu, wp);
        body_leg(x, r, wp);
    }
    for (int i = 0; i < nl; i++) {                                     /* MUW:140-155 */
        int replicated = 1;
        draw_begin(src, (uint32_t)i, 1);
        while (replicated) {
            draw_point32(src, lox, loy, hix, hiy, tgt + 2 * i);
            replicated = 0;
            if ((float)pos_dist(0, tgt[2 * i], tgt[2 * i + 1], loc[2 * i], loc[2 * i + 1]) <= two_r)
                replicated = 1;
            for (int j = 0; j < i; j++) {
                if ((float)pos_dist(0, tgt[2 * j], tgt[2 * j + 1], tgt[2 * i], tgt[2 * i + 1]) <= two_r) {
                    replicated = 1;
                    break;
                }
            }
        }
        init_d[i] = pos_dist(0, tgt[2 * i], tgt[2 * i + 1], loc[2 * i], loc[2 * i + 1]);
        prev_d[i] = init_d[i];
    }
    for (int i = nl; i < n; i++) {                                     /* extension: parked learners */
        loc[2 * i] = loc[2 * i + 1] = INFINITY;
        tgt[2 * i] = tgt[2 * i + 1] = 0.0;
        init_d[i] = prev_d[i] = INFINITY;
        flags[i] = UAVO_FLAG_INACTIVE;
    }
    if (circular) {                                                    /* MUW:157-163 */
        st->f64pos[e] = 1;
        for (int i = 0; i < n; i++) {
            double theta = 2 * i * M_PI / n;
            loc[2 * i] = 20.0 * 1.0 * cos(theta);
            loc[2 * i + 1] = 20.0 * 1.0 * sin(theta);
            tgt[2 * i] = 23.0 * 1.0 * cos(theta + M_PI);
            tgt[2 * i + 1] = 23.0 * 1.0 * sin(theta + M_PI);
            init_d[i] = pos_dist(1, tgt[2 * i], tgt[2 * i + 1], loc[2 * i], loc[2 * i + 1]);
            prev_d[i] = init_d[i];
        }
    }
    uint32_t *c = st->counters + e * 4;
    c[0] = 0; c[1] = 0; c[2] = 0;                                      /* MUW:166-168 */
}

void uavo_reset_mt(const uavo_config *cfg, uavo_state *st, int64_t env, uavo_mt *g, int circular) {
    draw_src s;
    envx x;
    memset(&s, 0, sizeof s);
    s.mt = g;
    make_envx(cfg, NULL, NULL, st, env, 0, &x);
    reset_env(&x, st, env, &s, circular);
}

/* Level of the episode that starts now (extension): drawn uniformly in [level_lo, level_hi] from the Philox stream of
 * the pseudo-slot 0xFFFF of this env (counter word 2 = 0, word 3 = episode), or the explicitly assigned next_level. */
static void pick_level(const uavo_ext *ext, uavo_ext_state *xs, int64_t e, const draw_src *s) {
    if (!ext || !xs || ext->n_levels <= 0) return;
    int lvl;
    if (ext->level_lo >= 0) {
        const uint32_t ctr[4] = {s->ctr_env[0], (s->ctr_env[1] & 0xFFFFu) | (0xFFFFu << 16), 0u, s->episode};
        uint32_t o[4];
        uavo_philox4x32(ctr, s->key, o);
        const uint32_t span = (uint32_t)(ext->level_hi - ext->level_lo + 1);
        lvl = ext->level_lo + (int)(((uint64_t)o[0] * span) >> 32);
    } else {
        lvl = xs->next_level[e];
    }
    if (lvl >= ext->n_levels) lvl = ext->n_levels - 1;
    xs->level[e] = (uint8_t)lvl;
}

static void reset_env_philox(const uavo_config *cfg, const uavo_ext *ext, uavo_state *st, uavo_ext_state *xs, int64_t e,
                             uint64_t seed, int64_t env_offset) {
    draw_src s;
    envx x;
    memset(&s, 0, sizeof s);
    uint64_t ge = (uint64_t)(env_offset + e);
    s.key[0] = (uint32_t)seed; s.key[1] = (uint32_t)(seed >> 32);
    s.ctr_env[0] = (uint32_t)ge; s.ctr_env[1] = (uint32_t)(ge >> 32);
    s.episode = st->counters[e * 4 + 3];
    s.addressed = 1;
    pick_level(ext, xs, e, &s);
    make_envx(cfg, ext, xs, st, e, env_offset, &x);
    reset_env(&x, st, e, &s, 0);
    st->counters[e * 4 + 3] += 1; /* next reset of this env draws a fresh layout */
}

void uavo_reset_philox_x(const uavo_config *cfg, const uavo_ext *ext, uavo_state *st, uavo_ext_state *xs,
                         const uint8_t *mask, uint64_t seed, int64_t env_offset, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < st->num_envs; e++) {
        if (mask && !mask[e]) continue;
        reset_env_philox(cfg, ext, st, xs, e, seed, env_offset);
    }
}
void uavo_reset_philox(const uavo_config *cfg, uavo_state *st, const uint8_t *mask, uint64_t seed,
                       int64_t env_offset, int nthreads) {
    uavo_reset_philox_x(cfg, NULL, st, NULL, mask, seed, env_offset, nthreads);
}

/* MUW:177-241 for one env. */
static void step_env(const envx *x, uavo_state *st, int64_t e, const double *actions,
                     int evaluate, double *obs, double *reward, uint8_t *done_out) {
    const uavo_config *cfg = &x->cfg;
    const int n = st->num_agents;
    const int f64pos = st->f64pos[e];
    double *loc = st->loc + e * n * 2, *vel = st->vel + e * n * 2, *tgt = st->tgt + e * n * 2;
    double *init_d = st->init_d + e * n, *prev_d = st->prev_d + e * n;
    uint8_t *flags = st->flags + e * n;
    uint32_t *cnt = st->counters + e * 4;
    const double tau = cfg->tau, amax = cfg->max_acceleration, vmax = cfg->max_speed;
    const double two_r = f64pos ? 2 * cfg->collider_radius : (double)(float)(2 * cfg->collider_radius);
    const double two_hard = 2 * HARD_COLLISION_RADIUS;
    const double lox = -cfg->x_size / 2.0, loy = -cfg->y_size / 2.0;
    const double hix = cfg->x_size / 2.0, hiy = cfg->y_size / 2.0;
    double px[MAXN], py[MAXN];
    /* extension: the scripted bodies move FIRST (the world's traffic advances, then the UAVs move in MUW:181 order), so a
     * learner's collision test and its observation see the same -- new -- body positions */
    for (int b = 0; b < x->nb; b++) body_move(x, b, cnt[0], (cnt[3] - 1u) & 0x7FFFFFFFu);
    const int ntot = gather_slots(x, loc, vel, px, py, NULL, NULL);

    for (int i = 0; i < n; i++) {                                       /* MUW:181 */
        const double *a = actions + (e * n + i) * 2;
        if (i >= x->nl) { /* parked learner (extension) */
            reward[e * n + i] = 0.0; done_out[e * n + i] = 1;
            continue;
        }
        double pd, d;
        const int was_done = (flags[i] & UAVO_FLAG_DONE) != 0;
        /* ---- UAVAgent.step, AG:23-36 ---- */
        if (was_done) {
            pd = 0.0; d = 0.0;                                          /* AG:24-25 */
        } else {
            for (int k = 0; k < 2; k++) {
                double dv = clipd((a[k] - vel[2 * i + k]) / tau, -amax, amax);       /* AG:26 */
                vel[2 * i + k] = clipd(vel[2 * i + k] + dv * tau, -vmax, vmax);     /* AG:27 */
                double dx = vel[2 * i + k] * tau;                                   /* AG:28 */
                double nl = loc[2 * i + k] + dx;       /* AG:29: float32 array += float64 array */
                loc[2 * i + k] = f64pos ? nl : (double)(float)nl;
            }
            px[i] = loc[2 * i]; py[i] = loc[2 * i + 1];
            pd = prev_d[i];                                                          /* AG:32 */
            d = pos_dist(f64pos, tgt[2 * i], tgt[2 * i + 1], px[i], py[i]);          /* AG:33 */
            prev_d[i] = d;                                                           /* AG:34 */
        }
        /* ---- reward shaping, MUW:183-195 ---- */
        const double max_speed = nrm64(vmax, vmax);                                  /* MUW:183 */
        double dth = atan2(pos_sub(f64pos, tgt[2 * i + 1], py[i]), pos_sub(f64pos, tgt[2 * i], px[i]))
                     - atan2(vel[2 * i + 1], vel[2 * i]);                            /* MUW:184-185 */
        dth = wrap_angle(dth);                                                       /* MUW:186 */
        const double q = max_speed / init_d[i];
        double r = 0.0 - 0.01 * ((1.0 < q) ? 1.0 : q);                               /* MUW:188-189 */
        double prog; /* prev_distance - distance: float32 - float32 unless circular / done */
        if (was_done) prog = 0.0;
        else if (f64pos) prog = pd - d;
        else prog = (double)((float)pd - (float)d);
        r += 50.0 * (prog / max_speed);                                              /* MUW:190 */
        double frac; /* distance/(1.5*init_distance): float32 chain under NEP 50 unless circular */
        if (f64pos) frac = d / (1.5 * init_d[i]);
        else frac = (double)((float)d / (1.5f * (float)init_d[i]));
        if (r > 0) r *= f64pos ? (1 - frac) : (double)(1.0f - (float)frac);          /* MUW:191-192 */
        else r *= f64pos ? (1 + frac) : (double)(1.0f + (float)frac);                /* MUW:193-194 */
        r -= 0.01 * fabs(dth);                                                       /* MUW:195 */

        /* ---- collisions with the <=2 nearest in-range agents, MUW:197-210 ---- */
        int collision = 0;
        int idx[2]; double dist[2];
        const int nn = nearest_two(ntot, i, f64pos, px, py, cfg->d_sense, idx, dist);   /* MUW:198 */
        for (int k = 0; k < nn; k++) {                                               /* MUW:199 */
            if (dist[k] <= two_r) { r = -2.0; collision = 1; }                       /* MUW:203-205 */
            if (dist[k] <= two_hard) {                                               /* MUW:207 */
                if (!(flags[i] & UAVO_FLAG_DONE) && !(flags[i] & UAVO_FLAG_COLLIDED)) {
                    cnt[2] += 1;                                                     /* MUW:209 */
                    flags[i] |= UAVO_FLAG_COLLIDED;                                  /* MUW:210 */
                }
            }
        }
        /* ---- termination, MUW:213-227 ---- */
        const int oob = !(px[i] >= lox && px[i] <= hix && py[i] >= loy && py[i] <= hiy); /* MUW:213,224 */
        const double speed = nrm64(vel[2 * i], vel[2 * i + 1]);                      /* MUW:214 */
        int dn;
        if (d < 0.5 && !collision && speed < 0.2) {                                  /* MUW:218 */
            dn = 1;
            if (!(flags[i] & UAVO_FLAG_DONE)) cnt[1] += 1;                           /* MUW:220-221 */
            /* UAVAgent.finish, AG:38-42 */
            flags[i] |= UAVO_FLAG_DONE;
            double nv = nrm64(vel[2 * i], vel[2 * i + 1]);
            double fx = vel[2 * i] / nv * 0.001, fy = vel[2 * i + 1] / nv * 0.001;
            if (fx != fx || fy != fy) { fx = 0.0; fy = 0.0; }
            vel[2 * i] = fx; vel[2 * i + 1] = fy;
            r += 10;                                                                 /* MUW:223 */
        } else if (oob) {
            dn = evaluate ? 0 : 1;                                                   /* MUW:224-225 */
        } else {
            dn = 0;
        }
        prev_d[i] = d;                                                               /* MUW:229 */
        reward[e * n + i] = r;
        done_out[e * n + i] = (uint8_t)dn;
    }
    observe_env(x, st, e, obs);                                                      /* MUW:233-235 */
    cnt[0] += 1;                                                                     /* MUW:238 */
}

void uavo_step_x(const uavo_config *cfg, const uavo_ext *ext, uavo_state *st, uavo_ext_state *xs, const double *actions,
                 int evaluate, int64_t env_offset, double *obs, double *reward, uint8_t *done, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < st->num_envs; e++) {
        envx x;
        make_envx(cfg, ext, xs, st, e, env_offset, &x);
        step_env(&x, st, e, actions, evaluate, obs, reward, done);
    }
}
void uavo_step(const uavo_config *cfg, uavo_state *st, const double *actions, int evaluate,
               double *obs, double *reward, uint8_t *done, int nthreads) {
    uavo_step_x(cfg, NULL, st, NULL, actions, evaluate, 0, obs, reward, done, nthreads);
}

/* ------------------------------------------------------------------------------------------------
 * uavx_step_ex restatement (new semantics; see header)
 * ---------------------------------------------------------------------------------------------- */
static void sincospi32(float t, float *sn, float *cs) {
    const float k = rintf(2.0f * t);
    const float r = fmaf(-0.5f, k, t);
    const float z = r * r;
    float ps = fmaf(z, 0.0821458866f, -0.599264529f);
    ps = fmaf(ps, z, 2.55016404f);
    ps = fmaf(ps, z, -5.16771278f);
    ps = fmaf(ps, z, 3.14159265f);
    ps = ps * r;
    float pc = fmaf(z, -0.0258068913f, 0.235330630f);
    pc = fmaf(pc, z, -1.33526277f);
    pc = fmaf(pc, z, 4.05871213f);
    pc = fmaf(pc, z, -4.93480220f);
    pc = fmaf(pc, z, 1.0f);
    const int q = (int)k & 3;
    *sn = (q == 0) ? ps : (q == 1) ? pc : (q == 2) ? -ps : -pc;
    *cs = (q == 0) ? pc : (q == 1) ? -ps : (q == 2) ? -pc : ps;
}

void uavo_polar_to_command(float a0, float a1, float vmax_norm, double out[2]) {
    const float v = fmaf(a0, 0.5f, 0.5f) * vmax_norm; /* test_sac_multi.py:77 */
    float sn, cs;
    sincospi32(a1, &sn, &cs);                         /* :78 theta = a1*pi */
    out[0] = (double)(v * cs);                        /* :80 */
    out[1] = (double)(v * sn);
}

void uavo_fold_episode(uavo_state *st, uavo_episode_state *ep, int64_t e) {
    uint32_t *c = st->counters + e * 4;
    if (c[0] != 0) {
        ep->fin_counts[4 * e + 0] += 1;
        ep->fin_counts[4 * e + 1] += c[0];
        ep->fin_counts[4 * e + 2] += c[1];
        ep->fin_counts[4 * e + 3] += c[2];
        ep->fin_returns[2 * e + 0] += ep->ep_run[2 * e + 0];
        ep->fin_returns[2 * e + 1] += ep->ep_run[2 * e + 1];
    }
    ep->ep_run[2 * e] = 0.f; ep->ep_run[2 * e + 1] = 0.f;
    ep->pending[e] = 0;
}

void uavo_step_ex_x(const uavo_config *cfg, const uavo_ext *ext, uavo_state *st, uavo_ext_state *xs,
                    uavo_episode_state *ep, const uavo_step_opts *opt, const double *actions, int evaluate, double *obs,
                    double *reward, uint8_t *done, uint8_t *reset_mask, uint8_t *ended_out, uint8_t *truncated_out,
                    int nthreads) {
    const int n = st->num_agents;
    const float vmax_norm = (float)nrm64(cfg->max_speed, cfg->max_speed);
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < st->num_envs; e++) {
        uint32_t *c = st->counters + e * 4;
        envx x;
        if (ep->pending[e]) { /* the env starts a new episode instead of stepping */
            uavo_fold_episode(st, ep, e);
            reset_env_philox(cfg, ext, st, xs, e, opt->seed, opt->env_offset);
            make_envx(cfg, ext, xs, st, e, opt->env_offset, &x);
            observe_env(&x, st, e, obs);
            for (int i = 0; i < n; i++) { reward[e * n + i] = 0.0; done[e * n + i] = 0; }
            if (reset_mask) reset_mask[e] = 1;
            if (ended_out) ended_out[e] = 0;
            if (truncated_out) truncated_out[e] = 0;
            continue;
        }
        make_envx(cfg, ext, xs, st, e, opt->env_offset, &x);
        double act[2 * MAXN];
        for (int i = 0; i < n; i++) {
            const double *a = actions + (e * n + i) * 2;
            if (opt->action_mode == 1) uavo_polar_to_command((float)a[0], (float)a[1], vmax_norm, act + 2 * i);
            else { act[2 * i] = a[0]; act[2 * i + 1] = a[1]; }
        }
        step_env(&x, st, e, act - (e * n) * 2, evaluate, obs, reward, done);
        int all_done = 1;
        for (int i = 0; i < n; i++) all_done &= done[e * n + i] != 0;
        const int terminal = (opt->reset_policy == 1 && done[e * n]) || (opt->reset_policy == 2 && all_done);
        const int capped = opt->step_cap != 0 && c[0] >= opt->step_cap;
        const int ended = terminal || capped;
        ep->pending[e] = (uint8_t)(ended ? 1 : 0);
        if (reset_mask) reset_mask[e] = 0;
        if (ended_out) ended_out[e] = (uint8_t)ended;
        if (truncated_out) truncated_out[e] = (uint8_t)(capped && !terminal);
        if (opt->track_returns) {
            float score = 0.f;
            for (int i = 0; i < n; i++) score += (float)reward[e * n + i] * (1.0f - (float)done[e * n + i]);
            ep->ep_run[2 * e] += (float)reward[e * n];
            ep->ep_run[2 * e + 1] += score;
        }
    }
}
void uavo_step_ex(const uavo_config *cfg, uavo_state *st, uavo_episode_state *ep, const uavo_step_opts *opt,
                  const double *actions, int evaluate, double *obs, double *reward, uint8_t *done,
                  uint8_t *reset_mask, int nthreads) {
    uavo_step_ex_x(cfg, NULL, st, NULL, ep, opt, actions, evaluate, obs, reward, done, reset_mask, NULL, NULL, nthreads);
}

/* ------------------------------------------------------------------------------------------------
 * UAVWorld2D
 * ---------------------------------------------------------------------------------------------- */
static void uw_observe_env(const uavo_uw_config *cfg, const uavo_uw_state *st, int64_t e, double *o) {
    const double vx = st->vel[2 * e], vy = st->vel[2 * e + 1];
    /* UW:88: after reset the velocity is a float32 array (UW:122) -> float32 norm */
    const double sp = st->vel_f32[e] ? (double)nrm32((float)vx, (float)vy) : nrm64(vx, vy);
    o[0] = sp / cfg->max_speed;                                                      /* UW:88 */
    const double theta = atan2(vy, vx);                                              /* UW:89 */
    o[1] = theta / M_PI;                                                             /* UW:90 */
    const double rtt = atan2(pos_sub(0, st->tgt[2 * e + 1], st->loc[2 * e + 1]),
                             pos_sub(0, st->tgt[2 * e], st->loc[2 * e]));            /* UW:91 */
    o[3] = wrap_angle(rtt - theta) / M_PI;                                           /* UW:92-94 */
    const double rtd = pos_dist(0, st->tgt[2 * e], st->tgt[2 * e + 1], st->loc[2 * e], st->loc[2 * e + 1]);
    o[2] = rtd / nrm64(cfg->x_size, cfg->y_size);                                    /* UW:96-97,17 */
}

void uavo_uw_observe(const uavo_uw_config *cfg, const uavo_uw_state *st, double *obs, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < st->num_envs; e++) uw_observe_env(cfg, st, e, obs + e * UAVO_UW_OBS_DIM);
}

static void uw_reset_env(const uavo_uw_config *cfg, uavo_uw_state *st, int64_t e, draw_src *s) {
    const double lox = -cfg->x_size / 2.0, loy = -cfg->y_size / 2.0, hix = cfg->x_size / 2.0, hiy = cfg->y_size / 2.0;
    draw_point32(s, lox, loy, hix, hiy, st->loc + 2 * e);                            /* UW:121 */
    draw_point32(s, -cfg->max_speed, -cfg->max_speed, cfg->max_speed, cfg->max_speed, st->vel + 2 * e); /* UW:122 */
    st->vel_f32[e] = 1;
    draw_point32(s, lox, loy, hix, hiy, st->tgt + 2 * e);                            /* UW:126 */
    st->init_d[e] = pos_dist(0, st->tgt[2 * e], st->tgt[2 * e + 1], st->loc[2 * e], st->loc[2 * e + 1]); /* UW:129 */
    st->prev_d[e] = st->init_d[e];                                                   /* UW:130 */
    st->steps[e] = 0;                                                                /* UW:131 */
}

void uavo_uw_reset_mt(const uavo_uw_config *cfg, uavo_uw_state *st, int64_t env, uavo_mt *g) {
    draw_src s;
    memset(&s, 0, sizeof s);
    s.mt = g;
    uw_reset_env(cfg, st, env, &s);
}

void uavo_uw_reset_philox(const uavo_uw_config *cfg, uavo_uw_state *st, const uint8_t *mask,
                          uint64_t seed, int64_t env_offset, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < st->num_envs; e++) {
        if (mask && !mask[e]) continue;
        draw_src s;
        memset(&s, 0, sizeof s);
        uint64_t ge = (uint64_t)(env_offset + e);
        s.key[0] = (uint32_t)seed; s.key[1] = (uint32_t)(seed >> 32);
        s.ctr_env[0] = (uint32_t)ge; s.ctr_env[1] = (uint32_t)(ge >> 32);
        s.episode = st->episode[e];
        uw_reset_env(cfg, st, e, &s);
        st->episode[e] += 1;
    }
}

static void uw_step_env(const uavo_uw_config *cfg, uavo_uw_state *st, int64_t e, const double *actions,
                        int action_is_f32, double *obs, double *reward, uint8_t *done, double *info) {
    const double tau = cfg->tau, amax = cfg->max_acceleration, vmax = cfg->max_speed;
    double *loc = st->loc + 2 * e, *vel = st->vel + 2 * e, *tgt = st->tgt + 2 * e;
    const double *a = actions + 2 * e;
    for (int k = 0; k < 2; k++) {
        double qv;
        if (st->vel_f32[e] && action_is_f32)         /* UW:142: f32 array / python float -> f32 */
            qv = (double)(((float)a[k] - (float)vel[k]) / (float)tau);
        else
            qv = (a[k] - vel[k]) / tau;
        double dv = clipd(qv, -amax, amax);                                          /* UW:142 */
        vel[k] = clipd(vel[k] + dv * tau, -vmax, vmax);                              /* UW:144 */
        loc[k] = (double)(float)(loc[k] + vel[k] * tau);                             /* UW:145-146 */
    }
    st->vel_f32[e] = 0;                                                              /* UW:147 */
    const double lox = -cfg->x_size / 2.0, loy = -cfg->y_size / 2.0, hix = cfg->x_size / 2.0, hiy = cfg->y_size / 2.0;
    const int oob = !(loc[0] >= lox && loc[0] <= hix && loc[1] >= loy && loc[1] <= hiy); /* UW:149,162 */
    const float d = (float)pos_dist(0, tgt[0], tgt[1], loc[0], loc[1]);              /* UW:150 */
    float r = 0.0f - 1.0f / (float)st->init_d[e];                                    /* UW:152-153 */
    r = r + 10.0f * ((float)st->prev_d[e] - d);                                      /* UW:154 */
    double dth = atan2(pos_sub(0, tgt[1], loc[1]), pos_sub(0, tgt[0], loc[0])) - atan2(vel[1], vel[0]); /* UW:155 */
    dth = wrap_angle(dth);                                                           /* UW:156 */
    r = r - (float)(0.1 * fabs(dth));                                                /* UW:157 */
    int dn;
    if (d < 0.5f) { dn = 1; r = r + 1000.0f; }                                       /* UW:159-161 */
    else if (oob) dn = 1;                                                            /* UW:162-163 */
    else dn = 0;
    uw_observe_env(cfg, st, e, obs + e * UAVO_UW_OBS_DIM);                           /* UW:168 */
    info[e] = (double)d;                                                             /* UW:169,114-117 */
    st->steps[e] += 1;                                                               /* UW:170 */
    st->prev_d[e] = (double)d;                                                       /* UW:172 */
    reward[e] = (double)r;
    done[e] = (uint8_t)dn;
}

void uavo_uw_step(const uavo_uw_config *cfg, uavo_uw_state *st, const double *actions,
                  int action_is_f32, double *obs, double *reward, uint8_t *done,
                  double *info_distance, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < st->num_envs; e++)
        uw_step_env(cfg, st, e, actions, action_is_f32, obs, reward, done, info_distance);
}

/* ---- uavx_uw_step_ex restatement ---- */
void uavo_uw_fold_episode(uavo_uw_state *st, uavo_uw_episode_state *ep, int64_t e) {
    if (st->steps[e] != 0) {
        ep->fin_counts[4 * e + 0] += 1;
        ep->fin_counts[4 * e + 1] += st->steps[e];
        ep->fin_counts[4 * e + 2] += ep->reached[e] ? 1u : 0u;
        ep->fin_return[e] += ep->ep_return[e];
    }
    ep->ep_return[e] = 0.f;
    ep->pending[e] = 0;
}

void uavo_uw_step_ex(const uavo_uw_config *cfg, uavo_uw_state *st, uavo_uw_episode_state *ep, int action_mode,
                     int auto_reset, uint32_t step_cap, int track_returns, uint64_t seed, int64_t env_offset,
                     const double *actions, int action_is_f32, double *obs, double *reward, uint8_t *done,
                     double *info_distance, uint8_t *reset_mask, int nthreads) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int64_t e = 0; e < st->num_envs; e++) {
        if (ep->pending[e]) {
            draw_src s;
            memset(&s, 0, sizeof s);
            uint64_t ge = (uint64_t)(env_offset + e);
            s.key[0] = (uint32_t)seed; s.key[1] = (uint32_t)(seed >> 32);
            s.ctr_env[0] = (uint32_t)ge; s.ctr_env[1] = (uint32_t)(ge >> 32);
            s.episode = st->episode[e];
            uavo_uw_fold_episode(st, ep, e);
            uw_reset_env(cfg, st, e, &s);
            st->episode[e] += 1;
            uw_observe_env(cfg, st, e, obs + e * UAVO_UW_OBS_DIM);
            reward[e] = 0.0; done[e] = 0; info_distance[e] = st->init_d[e];
            if (reset_mask) reset_mask[e] = 1;
            continue;
        }
        double act[2] = {actions[2 * e], actions[2 * e + 1]};
        int is_f32 = action_is_f32;
        if (action_mode == 1) { /* test_sac.py:77-80 in float32: v = (a0/2+0.5)*high[0], theta = a1*pi */
            uavo_polar_to_command((float)act[0], (float)act[1], (float)cfg->max_speed, act);
            is_f32 = 1;
        }
        uw_step_env(cfg, st, e, act - 2 * e, is_f32, obs, reward, done, info_distance);
        ep->reached[e] = (uint8_t)(info_distance[e] < 0.5);
        int ended = (auto_reset && done[e]) || (step_cap != 0 && st->steps[e] >= step_cap);
        ep->pending[e] = (uint8_t)(ended ? 1 : 0);
        if (track_returns) ep->ep_return[e] += (float)reward[e];
        if (reset_mask) reset_mask[e] = 0;
    }
}
