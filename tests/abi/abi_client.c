/* Native client of the C ABI (include/uavx.h): plain C, no Python, no torch -- device buffers come from
 * hipMalloc, the checker is the oracle's C library (oracle/uavx_oracle.h, test infrastructure).
 * What a non-Python host (the reference has none; SURVEY.md 8b "C-ABI (new)") would do:
 *   create -> reset(seed) -> K x step -> get_state / get_metrics -> destroy,
 * compared with uavo_reset_philox / uavo_step on the same seed and actions: done masks, positions,
 * velocities, flags and counters bit for bit, observations / rewards within 1e-5 (angles on the circle).
 *
 * Built and run by tests/test_abi_native.py with the plain C compiler:
 *   gcc -D__HIP_PLATFORM_AMD__ abi_client.c -Iinclude -Ioracle -I/opt/rocm/include \
 *       -L<csrc> -luavx -L<oracle/_build> -luavx_oracle -L/opt/rocm/lib -lamdhip64 -lm
 * usage: abi_client [num_envs] [num_agents] [steps]        exit code 0 = parity
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

static uint64_t lcg_state = 0x9E3779B97F4A7C15ull;
static double lcg_uniform(double lo, double hi) {   /* any deterministic action source will do */
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return lo + (hi - lo) * (double)(lcg_state >> 11) * (1.0 / 9007199254740992.0);
}

static double circ_diff(double a, double b) {       /* normalised angles live on a circle of length 2 */
    double d = fabs(a - b);
    return d > 1.0 ? fabs(2.0 - d) : d;
}

int main(int argc, char **argv) {
    const int64_t E = argc > 1 ? atoll(argv[1]) : 512;
    const int N = argc > 2 ? atoi(argv[2]) : 4;
    const int steps = argc > 3 ? atoi(argv[3]) : 400;
    const uint64_t seed = 2024;
    const int64_t A = E * N;
    static const int angle_col[UAVO_OBS_DIM] = {0, 1, 0, 1, 0, 1, 1, 0, 1, 1};

    uavx_config cfg = {26.0, 22.0, 10.0, 5.0, 1.0, 9.0, 0.02, N, 0};
    uavo_config ocfg = {26.0, 22.0, 10.0, 5.0, 1.0, 9.0, 0.02, N, 0};

    /* device side */
    uavx_handle *h = NULL;
    int rc = uavx_create(&cfg, E, 0, 0, &h);
    if (rc != UAVX_OK) { fprintf(stderr, "uavx_create -> %d (%s)\n", rc, uavx_strerror(rc)); return 2; }
    float *d_act, *d_obs, *d_rew, *d_loc; double *d_vel; uint8_t *d_done, *d_flags; uint32_t *d_cnt;
    CHECK_HIP(hipMalloc((void **)&d_act, A * 2 * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_obs, A * UAVO_OBS_DIM * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_rew, A * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_done, A));
    CHECK_HIP(hipMalloc((void **)&d_loc, A * 2 * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_vel, A * 2 * sizeof(double)));
    CHECK_HIP(hipMalloc((void **)&d_flags, A));
    CHECK_HIP(hipMalloc((void **)&d_cnt, E * 4 * sizeof(uint32_t)));
    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));

    /* checker side */
    uavo_state st;
    memset(&st, 0, sizeof st);
    st.num_envs = E; st.num_agents = N;
    st.loc = calloc(A * 2, sizeof(double)); st.vel = calloc(A * 2, sizeof(double)); st.tgt = calloc(A * 2, sizeof(double));
    st.init_d = calloc(A, sizeof(double)); st.prev_d = calloc(A, sizeof(double)); st.flags = calloc(A, 1);
    st.counters = calloc(E * 4, sizeof(uint32_t)); st.f64pos = calloc(E, 1);
    double *o_act = malloc(A * 2 * sizeof(double)), *o_obs = malloc(A * UAVO_OBS_DIM * sizeof(double));
    double *o_rew = malloc(A * sizeof(double));
    uint8_t *o_done = malloc(A), *g_done = malloc(A), *g_flags = malloc(A);
    float *h_act = malloc(A * 2 * sizeof(float)), *g_obs = malloc(A * UAVO_OBS_DIM * sizeof(float));
    float *g_rew = malloc(A * sizeof(float)), *g_loc = malloc(A * 2 * sizeof(float));
    double *g_vel = malloc(A * 2 * sizeof(double));
    uint32_t *g_cnt = malloc(E * 4 * sizeof(uint32_t));

    CHECK_UAVX(h, uavx_reset(h, NULL, seed, d_obs, stream));
    uavo_reset_philox(&ocfg, &st, NULL, seed, 0, 4);
    uavo_observe(&ocfg, &st, o_obs, 4);
    CHECK_HIP(hipMemcpyAsync(g_obs, d_obs, A * UAVO_OBS_DIM * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    double worst_obs = 0, worst_rew = 0;
    long bad = 0, dones = 0;
    for (int64_t k = 0; k < A * UAVO_OBS_DIM; k++) {
        const double d = angle_col[k % UAVO_OBS_DIM] ? circ_diff(g_obs[k], o_obs[k]) : fabs(g_obs[k] - o_obs[k]);
        if (d > worst_obs) worst_obs = d;
    }
    for (int t = 0; t < steps; t++) {
        for (int64_t a = 0; a < A; a++) {   /* seek the target with noise: finishes, collisions and OOB all occur */
            const double dx = st.tgt[2 * a] - st.loc[2 * a], dy = st.tgt[2 * a + 1] - st.loc[2 * a + 1];
            const int wild = lcg_uniform(0, 1) < 0.03;
            h_act[2 * a] = (float)(wild ? lcg_uniform(-10, 10) : 1.5 * dx + lcg_uniform(-0.05, 0.05));
            h_act[2 * a + 1] = (float)(wild ? lcg_uniform(-10, 10) : 1.5 * dy + lcg_uniform(-0.05, 0.05));
            o_act[2 * a] = h_act[2 * a]; o_act[2 * a + 1] = h_act[2 * a + 1];
        }
        const int evaluate = (t % 7) == 6;
        CHECK_HIP(hipMemcpyAsync(d_act, h_act, A * 2 * sizeof(float), hipMemcpyHostToDevice, stream));
        CHECK_UAVX(h, uavx_step(h, d_act, UAVX_F32, evaluate, d_obs, d_rew, d_done, stream));
        CHECK_HIP(hipMemcpyAsync(g_obs, d_obs, A * UAVO_OBS_DIM * sizeof(float), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_rew, d_rew, A * sizeof(float), hipMemcpyDeviceToHost, stream));
        CHECK_HIP(hipMemcpyAsync(g_done, d_done, A, hipMemcpyDeviceToHost, stream));
        uavo_step(&ocfg, &st, o_act, evaluate, o_obs, o_rew, o_done, 4);   /* overlaps with the launch */
        CHECK_HIP(hipStreamSynchronize(stream));
        for (int64_t a = 0; a < A; a++) {
            if (g_done[a] != o_done[a]) bad++;
            dones += o_done[a];
            const double dr = fabs((double)g_rew[a] - o_rew[a]) / fmax(1.0, fabs(o_rew[a]));
            if (dr > worst_rew) worst_rew = dr;
        }
        for (int64_t k = 0; k < A * UAVO_OBS_DIM; k++) {
            const double d = angle_col[k % UAVO_OBS_DIM] ? circ_diff(g_obs[k], o_obs[k]) : fabs(g_obs[k] - o_obs[k]);
            if (d > worst_obs) worst_obs = d;
        }
    }
    uavx_state_view view = {d_loc, d_vel, NULL, NULL, NULL, d_flags, NULL};
    CHECK_UAVX(h, uavx_get_state(h, &view, stream));
    CHECK_UAVX(h, uavx_get_metrics(h, d_cnt, stream));
    CHECK_HIP(hipMemcpyAsync(g_loc, d_loc, A * 2 * sizeof(float), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(g_vel, d_vel, A * 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(g_flags, d_flags, A, hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipMemcpyAsync(g_cnt, d_cnt, E * 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    long reach = 0, coll = 0;
    for (int64_t a = 0; a < A * 2; a++) bad += ((double)g_loc[a] != st.loc[a]) + (g_vel[a] != st.vel[a]);
    for (int64_t a = 0; a < A; a++) bad += (g_flags[a] & 3u) != (st.flags[a] & 3u);
    for (int64_t e = 0; e < E; e++) {
        for (int c = 0; c < 4; c++) bad += g_cnt[4 * e + c] != st.counters[4 * e + c];
        reach += st.counters[4 * e + 1]; coll += st.counters[4 * e + 2];
    }
    /* error behaviour of the boundary: status codes and messages, never a crash */
    if (uavx_step(h, NULL, UAVX_F32, 0, d_obs, d_rew, d_done, stream) != UAVX_ERR_INVALID_ARG) bad++;
    if (uavx_step(h, d_act, 7, 0, d_obs, d_rew, d_done, stream) != UAVX_ERR_INVALID_ARG) bad++;
    if (strlen(uavx_last_error(h)) == 0) bad++;
    CHECK_UAVX(h, uavx_destroy(h));

    printf("abi_client: %lld envs x %d UAVs, %d steps: mismatches %ld, worst obs err %.3g, worst reward err %.3g, "
           "done flags seen %ld, reach %ld, collisions %ld\n", (long long)E, N, steps, bad, worst_obs, worst_rew,
           dones, reach, coll);
    if (bad || worst_obs > 1e-5 || worst_rew > 1e-5) return 1;
    if (dones == 0 || reach == 0 || coll == 0) { fprintf(stderr, "scenario too tame\n"); return 3; }
    return 0;
}
