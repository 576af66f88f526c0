// What does this part stream at when a kernel READS 48 B and WRITES 64 B per thread (the step kernel's mix: 41 % reads, 59 %
// writes) over footprints from 0.25 to 4 GB per launch?  The step kernel's algorithmic rate falls from 0.78 of 8 TB/s at 1 Mi envs
// (0.49 GB per launch) to 0.65-0.69 at 2-4 Mi envs with the SAME bytes per env (profiles/r04_ab_notes.md): is that the kernel or
// the memory system?  Three 16-byte loads + four 16-byte stores per thread, lane-contiguous, separate arrays (as the SoA state),
// back-to-back launches, HIP events.   hipcc --offload-arch=gfx950 -O3 -o stream_mix stream_mix.hip && ./stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <bool WT>
__global__ __launch_bounds__(64) void mix(const float4 *__restrict__ a, const float4 *__restrict__ b, const float4 *__restrict__ c,
                                          float4 *__restrict__ w, float4 *__restrict__ x, float4 *__restrict__ y, float4 *__restrict__ z,
                                          size_t n) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float4 p = a[i], q = b[i], r = c[i];
    const float4 s = make_float4(p.x + q.x, p.y + q.y, p.z + r.z, p.w + r.w);
    if (WT) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        auto nt = [](float4 v, float4 *dst) { __builtin_nontemporal_store((f4){v.x, v.y, v.z, v.w}, reinterpret_cast<f4 *>(dst)); };
        nt(s, &w[i]); nt(p, &x[i]); nt(q, &y[i]); nt(r, &z[i]);
    } else {
        w[i] = s; x[i] = p; y[i] = q; z[i] = r;
    }
}
__global__ __launch_bounds__(64) void rd(const float4 *__restrict__ a, float4 *__restrict__ sink, size_t n) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float4 p = a[i];
    if (p.x == 12345.678f) sink[0] = p;
}
__global__ __launch_bounds__(64) void wr(float4 *__restrict__ w, size_t n) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < n) w[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

template <typename F> static float us_per_launch(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    const size_t nmax = (size_t)4 << 30;   // bytes per launch at the largest point
    float4 *buf[7];
    const size_t per_thread = 7 * 16, nthreads_max = nmax / per_thread;
    for (auto &p : buf) hipMalloc(&p, nthreads_max * 16);
    for (auto &p : buf) hipMemset(p, 0, nthreads_max * 16);
    printf("bytes per launch (GB) : mixed 48 B read + 64 B written per thread, plain stores / nontemporal stores; read-only; write-only  [TB/s]\n");
    for (double gb : {0.25, 0.5, 1.0, 2.0, 4.0}) {
        const size_t n = (size_t)(gb * (1 << 30)) / per_thread;
        const unsigned g = (unsigned)((n + 63) / 64);
        const int reps = gb < 1 ? 60 : 20;
        const float t0 = us_per_launch([&] { hipLaunchKernelGGL(mix<false>, dim3(g), dim3(64), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], n); }, reps);
        const float t1 = us_per_launch([&] { hipLaunchKernelGGL(mix<true>, dim3(g), dim3(64), 0, 0, buf[0], buf[1], buf[2], buf[3], buf[4], buf[5], buf[6], n); }, reps);
        const size_t n1 = (size_t)(gb * (1 << 30)) / 16 / 7;   // the single-array kernels sweep one of the seven arrays: compare per-byte rates
        const unsigned g1 = (unsigned)((n1 + 63) / 64);
        const float t2 = us_per_launch([&] { hipLaunchKernelGGL(rd, dim3(g1), dim3(64), 0, 0, buf[0], buf[6], n1); }, reps);
        const float t3 = us_per_launch([&] { hipLaunchKernelGGL(wr, dim3(g1), dim3(64), 0, 0, buf[3], n1); }, reps);
        const double bytes = (double)n * per_thread, b1 = (double)n1 * 16;
        printf("%5.2f : %6.2f %6.2f   read-only (%.2f GB) %6.2f   write-only %6.2f\n", gb, bytes / t0 / 1e6, bytes / t1 / 1e6, b1 / 1e9, b1 / t2 / 1e6, b1 / t3 / 1e6);
    }
    return 0;
}
