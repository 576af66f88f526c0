/* Native client of the ABI entry points added in versions 2 and 3 (include/uavx.h): scripted bodies, curriculum levels, the fused
 * uavx_step_ex with its ended / truncated outputs -- plain C, no Python, no torch; the checker is the oracle's C library
 * (uavo_*_x).  The extension has no reference counterpart (see uavx.h); what is compared is the HIP path against the
 * oracle's restatement of the same definition: reset / done / ended / truncated masks, levels, body records, learner
 * positions and counters bit for bit, observations / rewards within 1e-5 (angles on the circle).
 *
 * Built and run by tests/test_abi_native.py like abi_client.c.
 * usage: abi_client_ext [num_envs] [learners] [bodies] [steps]        exit code 0 = parity
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "uavx.h"
#include "uavx_oracle.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_UAVX(h, x) do { int rc_ = (x); if (rc_ != UAVX_OK) { \
    fprintf(stderr, "%s:%d %s -> %d (%s: %s)\n", __FILE__, __LINE__, #x, rc_, uavx_strerror(rc_), uavx_last_error(h)); \
    return 2; } } while (0)

static uint64_t lcg_state = 0xD1B54A32D192ED03ull;
static double lcg_uniform(double lo, double hi) {
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return lo + (hi - lo) * (double)(lcg_state >> 11) * (1.0 / 9007199254740992.0);
}
static double circ_diff(double a, double b) {
    double d = fabs(a - b);
    return d > 1.0 ? fabs(2.0 - d) : d;
}

int main(int argc, char **argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 256;
    const int L = argc > 2 ? atoi(argv[2]) : 8;
    const int B = argc > 3 ? atoi(argv[3]) : 16;
    const int steps = argc > 4 ? atoi(argv[4]) : 160;
    const uint64_t seed = 77;
    const uint32_t cap = 37;
    const int64_t A = E * L;
    static const int angle_col[UAVO_OBS_DIM] = {0, 1, 0, 1, 0, 1, 1, 0, 1, 1};

    uavx_config cfg = {30.0, 30.0, 10.0, 5.0, 1.0, 12.0, 0.02, L, B};
    uavo_config ocfg = {30.0, 30.0, 10.0, 5.0, 1.0, 12.0, 0.02, L, 0};
    const uavx_body_rule rule = {3.0, 16, 0, 5};
    const uavx_level levels[3] = {{16.0, 14.0, 0.4, 6.0, L > 2 ? 2 : 1, B / 4},
                                  {22.0, 22.0, 0.7, 9.0, L > 4 ? 4 : L, B / 2},
                                  {30.0, 30.0, 1.0, 12.0, L, B}};
    uavo_level olevels[3];
    for (int i = 0; i < 3; i++) {
        olevels[i].x_size = levels[i].x_size; olevels[i].y_size = levels[i].y_size;
        olevels[i].collider_radius = levels[i].collider_radius; olevels[i].d_sense = levels[i].d_sense;
        olevels[i].n_active = levels[i].n_active; olevels[i].b_active = levels[i].b_active;
    }
    uavo_ext ext = {B, rule.period, rule.speed, rule.seed, 3, 0, 2, 0, olevels};

    uavx_handle *h = NULL;
    int rc = uavx_create(&cfg, E, 0, 0, &h);
    if (rc != UAVX_OK) { fprintf(stderr, "uavx_create -> %d (%s)\n", rc, uavx_strerror(rc)); return 2; }
    if (uavx_num_bodies(h) != B) { fprintf(stderr, "uavx_num_bodies\n"); return 2; }
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    CHECK_UAVX(h, uavx_set_body_rule(h, &rule));
    CHECK_UAVX(h, uavx_set_curriculum(h, levels, 3, 0, 2, stream));

    float *d_act, *d_obs, *d_rew, *d_loc, *d_body; uint8_t *d_done, *d_flags3, *d_lvl, *d_aflags; uint32_t *d_cnt;
    CHECK_HIP(hipMalloc((void **)&d_act, A * 2 * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_obs, A * UAVO_OBS_DIM * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_rew, A * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_done, A));
    CHECK_HIP(hipMalloc((void **)&d_loc, A * 2 * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_aflags, A));
    CHECK_HIP(hipMalloc((void **)&d_body, (size_t)E * (B ? B : 1) * UAVX_BODY_DIM * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_flags3, 3 * E));
    CHECK_HIP(hipMalloc((void **)&d_lvl, E));
    CHECK_HIP(hipMalloc((void **)&d_cnt, E * 4 * sizeof(uint32_t)));

    uavo_state st;
    memset(&st, 0, sizeof st);
    st.num_envs = E; st.num_agents = L;
    st.loc = calloc(A * 2, sizeof(double)); st.vel = calloc(A * 2, sizeof(double)); st.tgt = calloc(A * 2, sizeof(double));
    st.init_d = calloc(A, sizeof(double)); st.prev_d = calloc(A, sizeof(double)); st.flags = calloc(A, 1);
    st.counters = calloc(E * 4, sizeof(uint32_t)); st.f64pos = calloc(E, 1);
    uavo_ext_state xs = {calloc((size_t)E * (B ? B : 1) * UAVO_BODY_DIM, sizeof(float)), calloc(E, 1), calloc(E, 1)};
    uavo_episode_state ep = {calloc(E, 1), calloc(E * 2, sizeof(float)), calloc(E * 4, sizeof(uint32_t)), calloc(E * 2, sizeof(float))};
    double *o_act = malloc(A * 2 * sizeof(double)), *o_obs = malloc(A * UAVO_OBS_DIM * sizeof(double)), *o_rew = malloc(A * sizeof(double));
    uint8_t *o_done = malloc(A), *o_rm = malloc(E), *o_en = malloc(E), *o_tr = malloc(E);
    float *h_act = malloc(A * 2 * sizeof(float)), *g_obs = malloc(A * UAVO_OBS_DIM * sizeof(float)), *g_rew = malloc(A * sizeof(float));
    float *g_loc = malloc(A * 2 * sizeof(float)), *g_body = malloc((size_t)E * (B ? B : 1) * UAVX_BODY_DIM * sizeof(float));
    uint8_t *g_done = malloc(A), *g_flags3 = malloc(3 * E), *g_lvl = malloc(E), *g_aflags = malloc(A);
    uint32_t *g_cnt = malloc(E * 4 * sizeof(uint32_t));

    CHECK_UAVX(h, uavx_reset(h, NULL, seed, d_obs, stream));
    uavo_reset_philox_x(&ocfg, &ext, &st, &xs, NULL, seed, 0, 4);

    uavx_step_args args;
    memset(&args, 0, sizeof args);
    args.actions = d_act; args.action_dtype = UAVX_F32; args.action_mode = UAVX_ACTION_POLAR; args.evaluate = 1;
    args.reset_policy = UAVX_RESET_ALL_DONE; args.step_cap = cap; args.track_returns = 1; args.seed = seed;
    args.obs = d_obs; args.rew = d_rew; args.done = d_done;
    args.reset_mask = d_flags3; args.ended = d_flags3 + E; args.truncated = d_flags3 + 2 * E;
    const uavo_step_opts opt = {1, 2, 1, cap, seed, 0};

    double worst_obs = 0, worst_rew = 0;
    long bad = 0, resets = 0, truncs = 0, levels_seen[3] = {0, 0, 0};
    for (int t = 0; t < steps; t++) {
        for (int64_t a = 0; a < A * 2; a++) { h_act[a] = (float)lcg_uniform(-1, 1); o_act[a] = h_act[a]; }
        CHECK_HIP(hipMemcpyAsync(d_act, h_act, A * 2 * sizeof(float), hipMemcpyHostToDevice, stream));
        CHECK_UAVX(h, uavx_step_ex(h, &args, stream));
        uavx_state_view view = {d_loc, NULL, NULL, NULL, NULL, d_aflags, d_cnt};
        CHECK_UAVX(h, uavx_get_state(h, &view, stream));
        if (B) CHECK_UAVX(h, uavx_get_bodies(h, d_body, stream));
        CHECK_UAVX(h, uavx_get_env_levels(h, d_lvl, stream));
        CHECK_HIP(hipMemcpyAsync(g_obs, d_obs, A * UAVO_OBS_DIM * sizeof(float), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_rew, d_rew, A * sizeof(float), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_done, d_done, A, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_flags3, d_flags3, 3 * E, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_loc, d_loc, A * 2 * sizeof(float), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_aflags, d_aflags, A, hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_cnt, d_cnt, E * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_lvl, d_lvl, E, hipMemcpyDeviceToHost, stream));
        if (B) CHECK_HIP(hipMemcpyAsync(g_body, d_body, (size_t)E * B * UAVX_BODY_DIM * sizeof(float), hipMemcpyDeviceToHost, stream));
        uavo_step_ex_x(&ocfg, &ext, &st, &xs, &ep, &opt, o_act, 1, o_obs, o_rew, o_done, o_rm, o_en, o_tr, 4);
        CHECK_HIP(hipStreamSynchronize(stream));
        for (int64_t e = 0; e < E; e++) {
            bad += (g_flags3[e] != o_rm[e]) + (g_flags3[E + e] != o_en[e]) + (g_flags3[2 * E + e] != o_tr[e]) + (g_lvl[e] != xs.level[e]);
            for (int c = 0; c < 4; c++) bad += g_cnt[4 * e + c] != st.counters[4 * e + c];
            resets += o_rm[e]; truncs += o_tr[e]; levels_seen[xs.level[e] % 3]++;
        }
        for (int64_t a = 0; a < A; a++) {
            bad += (g_done[a] != o_done[a]) + (g_aflags[a] != (st.flags[a] & (3u | UAVO_FLAG_INACTIVE)));
            if (!(st.flags[a] & UAVO_FLAG_INACTIVE))
                bad += ((double)g_loc[2 * a] != st.loc[2 * a]) + ((double)g_loc[2 * a + 1] != st.loc[2 * a + 1]);
            const double dr = fabs((double)g_rew[a] - o_rew[a]) / fmax(1.0, fabs(o_rew[a]));
            if (dr > worst_rew) worst_rew = dr;
        }
        for (int64_t k = 0; k < (int64_t)E * B * UAVX_BODY_DIM; k++) bad += memcmp(&g_body[k], &xs.body[k], 4) != 0;
        for (int64_t k = 0; k < A * UAVO_OBS_DIM; k++) {
            const double d = angle_col[k % UAVO_OBS_DIM] ? circ_diff(g_obs[k], o_obs[k]) : fabs(g_obs[k] - o_obs[k]);
            if (d > worst_obs) worst_obs = d;
        }
    }
    /* error behaviour of the new entry points */
    const uavx_body_rule bad_rule = {3.0, 48, 0, 0};
    if (uavx_set_body_rule(h, &bad_rule) != UAVX_ERR_INVALID_ARG) bad++;
    if (uavx_set_curriculum(h, levels, UAVX_MAX_LEVELS + 1, 0, 0, stream) != UAVX_ERR_INVALID_ARG) bad++;
    if (uavx_set_curriculum(h, levels, 3, 2, 1, stream) != UAVX_ERR_INVALID_ARG) bad++;
    if (uavx_set_position_mode(h, UAVX_POS_F64, stream) != UAVX_ERR_UNSUPPORTED) bad++;
    if (uavx_set_prefetch(h, -1) != UAVX_ERR_INVALID_ARG) bad++;
    CHECK_UAVX(h, uavx_destroy(h));

    printf("abi_client_ext: %lld envs x (%d UAVs + %d bodies), %d steps: mismatches %ld, worst obs err %.3g, worst reward err "
           "%.3g, re-initialisations %ld, truncations %ld, env-steps per level %ld / %ld / %ld\n", (long long)E, L, B, steps, bad,
           worst_obs, worst_rew, resets, truncs, levels_seen[0], levels_seen[1], levels_seen[2]);
    if (bad || worst_obs > 1e-5 || worst_rew > 1e-5) return 1;
    if (resets == 0 || truncs == 0 || !levels_seen[0] || !levels_seen[1] || !levels_seen[2]) { fprintf(stderr, "scenario too tame\n"); return 3; }
    return 0;
}
