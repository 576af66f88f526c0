// Experiment: the batch as S handles stepped from S host threads on S streams (eager launches, no graph): do two hardware queues
// overlap the launch boundary of one chain with the kernels of the other?
//   hipcc -O2 -o two_queues two_queues.cpp -I../../include -L../../gym_uav_collision_avoidance_amd/csrc -luavx -Wl,-rpath,...
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "uavx.h"

int main(int argc, char **argv) {
    const long E = argc > 1 ? atol(argv[1]) : 65536;
    const int N = 4, K = 4000;
    for (int S : {1, 2, 4, 1, 2}) {
        const long e = E / S;
        std::vector<uavx_handle *> h(S);
        std::vector<hipStream_t> st(S);
        std::vector<float *> act(S), obs(S), rew(S);
        std::vector<uint8_t *> done(S);
        uavx_config cfg = {50, 50, 10, 5, 1.0, 15, 0.02, N, 0};
        for (int k = 0; k < S; k++) {
            if (uavx_create(&cfg, e, (uint64_t)k * e, 0, &h[k]) != 0) { printf("create failed\n"); return 1; }
            hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
            hipMalloc(&act[k], e * N * 8); hipMemset(act[k], 0, e * N * 8);
            hipMalloc(&obs[k], e * N * 40); hipMalloc(&rew[k], e * N * 4); hipMalloc(&done[k], e * N);
            uavx_reset(h[k], nullptr, 0, obs[k], st[k]);
        }
        hipDeviceSynchronize();
        auto run = [&](int k, int steps) {
            for (int i = 0; i < steps; i++) uavx_step(h[k], act[k], UAVX_F32, 0, obs[k], rew[k], done[k], st[k]);
            hipStreamSynchronize(st[k]);
        };
        { std::vector<std::thread> t; for (int k = 0; k < S; k++) t.emplace_back(run, k, 200); for (auto &x : t) x.join(); }
        auto t0 = std::chrono::steady_clock::now();
        { std::vector<std::thread> t; for (int k = 0; k < S; k++) t.emplace_back(run, k, K); for (auto &x : t) x.join(); }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / K;
        printf("E=%ld handles/threads/streams=%d: %.3f us per step of the whole batch (%.2f G env-steps/s)\n", E, S, dt * 1e6, E / dt / 1e9);
        for (int k = 0; k < S; k++) { uavx_destroy(h[k]); hipFree(act[k]); hipFree(obs[k]); hipFree(rew[k]); hipFree(done[k]); hipStreamDestroy(st[k]); }
    }
    return 0;
}
