// How many wavefronts does one SIMD of this part hold at a given SGPR / VGPR count?  (The compiler's own limit for
// __launch_bounds__(64, 8) is 80 SGPRs: 800 per SIMD / 8, minus 16 per wavefront for the trap handler, in blocks of 16 -- is that
// what the hardware does?)  Every wavefront spins on the 100 MHz real-time clock for a fixed time T; a launch of CUs x 4 SIMDs x k
// one-wavefront workgroups takes about T while k wavefronts fit a SIMD and 2 T as soon as they do not.
//   hipcc --offload-arch=gfx950 -O3 -o occupancy occupancy.hip && ./occupancy
// (the loop leaves on elapsed time alone, so every wavefront finishes whatever the placement)
#include <hip/hip_runtime.h>
#include <cstdio>

#define SPIN_KERNEL(NAME, CLOBBER_ASM, CLOBBER_REG)                                             \
    __global__ __launch_bounds__(64) void NAME(unsigned long long ticks, unsigned *sink) {      \
        asm volatile(CLOBBER_ASM ::: CLOBBER_REG);                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                             \
        unsigned n = 0;                                                                         \
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) n++;                                  \
        if (n == 0xffffffffu) *sink = n;                                                        \
    }
SPIN_KERNEL(spin_s40, "s_mov_b32 s40, 0", "s40")
SPIN_KERNEL(spin_s72, "s_mov_b32 s72, 0", "s72")
SPIN_KERNEL(spin_s74, "s_mov_b32 s74, 0", "s74")
SPIN_KERNEL(spin_s88, "s_mov_b32 s88, 0", "s88")
SPIN_KERNEL(spin_s90, "s_mov_b32 s90, 0", "s90")
SPIN_KERNEL(spin_s100, "s_mov_b32 s100, 0", "s100")
SPIN_KERNEL(spin_v63, "v_mov_b32 v63, 0", "v63")
SPIN_KERNEL(spin_v64, "v_mov_b32 v64, 0", "v64")
SPIN_KERNEL(spin_v72, "v_mov_b32 v72, 0", "v72")

template <typename K> static float run_us(K kern, int grid, unsigned long long ticks, unsigned *sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, ticks, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, ticks, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 100.f;   // us per launch
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int simds = pr.multiProcessorCount * 4;
    unsigned *sink; hipMalloc(&sink, 4);
    const unsigned long long T = 5000;   // s_memrealtime ticks (100 MHz: 50 us)
    printf("%s: %d CUs; launch of CUs x 4 x k one-wavefront workgroups spinning T each: us per launch\n", pr.name, pr.multiProcessorCount);
    printf("%-10s", "kernel");
    for (int k = 4; k <= 10; k++) printf("   k=%-2d", k);
    printf("\n");
#define ROW(NAME)                                                                   \
    do {                                                                            \
        printf("%-10s", #NAME);                                                     \
        for (int k = 4; k <= 10; k++) printf(" %6.1f", run_us(NAME, simds * k, T, sink)); \
        printf("\n");                                                               \
    } while (0)
    ROW(spin_s40); ROW(spin_s72); ROW(spin_s74); ROW(spin_s88); ROW(spin_s90); ROW(spin_s100);
    ROW(spin_v63); ROW(spin_v64); ROW(spin_v72);
    return 0;
}
