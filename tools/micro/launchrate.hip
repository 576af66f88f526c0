// Wave launch rate micro-benchmark: how long does the GPU take to start (and retire) G workgroups of B threads
// that do almost nothing, back to back?  hipcc --offload-arch=gfx950 -O3 -o launchrate launchrate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_empty(unsigned *sink) {
    if (threadIdx.x == 0 && blockIdx.x == 0xffffffffu) *sink = 1;
}
// each lane loads 16 B, adds, stores 16 B (a minimal streaming body)
__global__ void k_stream(const float4 *in, float4 *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = in[i];
    v.x += 1.f;
    out[i] = v;
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
    const size_t n = 4096 * 64 * 16;
    float4 *in, *out; hipMalloc(&in, n * 16); hipMalloc(&out, n * 16); hipMemset(in, 0, n * 16);
    struct { int g, b; } shapes[] = {{256, 64}, {1024, 64}, {4096, 64}, {16384, 64}, {65536, 64}, {1024, 256}, {4096, 256},
                                     {256, 1024}, {1024, 1024}, {4096, 1024}};
    for (auto s : shapes) {
        float e = time_us([&] { hipLaunchKernelGGL(k_empty, dim3(s.g), dim3(s.b), 0, 0, sink); }, 500);
        float t = time_us([&] { hipLaunchKernelGGL(k_stream, dim3(s.g), dim3(s.b), 0, 0, in, out); }, 500);
        printf("grid %6d x %4d threads (%6d waves): empty %6.2f us   16B-stream %6.2f us\n", s.g, s.b, s.g * s.b / 64, e, t);
    }
    return 0;
}
