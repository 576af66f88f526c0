// Does a workgroup's LDS allocation (or its kernel-argument size) slow the dispatcher down?  G one-wavefront workgroups that do
// nothing, with 0 / 4 KB / 16 KB of static LDS and with a 16 B / 512 B kernel-argument segment, back to back (eager launches, HIP
// events).  hipcc --offload-arch=gfx950 -O3 -o launchrate_lds launchrate_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>

struct Big { unsigned long long w[62]; };   // 496 B: with the pointer a 504-byte kernel-argument segment (the step kernels: ~450 B)

template <int LDSB>
__global__ void k_lds(unsigned *sink) {
    __shared__ unsigned buf[LDSB / 4 > 0 ? LDSB / 4 : 1];
    if (LDSB) buf[threadIdx.x] = threadIdx.x;
    if (blockIdx.x == 0xffffffffu) *sink = LDSB ? buf[(threadIdx.x + 1) & 63] : 1u;
}
__global__ void k_args(unsigned *sink, Big b) {
    if (blockIdx.x == 0xffffffffu) *sink = (unsigned)b.w[threadIdx.x & 31];
}

template <typename F> static float time_us(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}

int main() {
    unsigned *sink; hipMalloc(&sink, 4);
    Big big = {};
    for (int g : {1024, 4096, 8192, 16384}) {
        const float a = time_us([&] { hipLaunchKernelGGL(k_lds<0>, dim3(g), dim3(64), 0, 0, sink); }, 500);
        const float b = time_us([&] { hipLaunchKernelGGL(k_lds<4096>, dim3(g), dim3(64), 0, 0, sink); }, 500);
        const float c = time_us([&] { hipLaunchKernelGGL(k_lds<16384>, dim3(g), dim3(64), 0, 0, sink); }, 500);
        const float d = time_us([&] { hipLaunchKernelGGL(k_args, dim3(g), dim3(64), 0, 0, sink, big); }, 500);
        const float e = time_us([&] { hipLaunchKernelGGL(k_lds<4096>, dim3(g / 2), dim3(128), 0, 0, sink); }, 500);
        printf("%6d one-wavefront workgroups: no LDS %6.2f us   4 KB LDS %6.2f   16 KB LDS %6.2f   504 B of arguments %6.2f   (as %d x 128 threads, 4 KB: %6.2f)\n",
               g, a, b, c, d, g / 2, e);
    }
    return 0;
}
