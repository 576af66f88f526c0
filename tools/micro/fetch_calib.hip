// fetch_calib.hip — calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE (and the raw TCC counters behind them) on
// KNOWN byte counts in the access shapes the uavx step kernels use: one lane per agent, 8-byte (float2 position /
// float2 action) and 16-byte (double2 velocity, goal record) loads, 1/4/8/16-byte plain and 16-byte write-through
// (sc1) stores.  MI355X_MICROARCH.md §HBM says FETCH_SIZE reads exactly half of a wide (16 B/lane) coalesced
// stream on gfx950 and that other widths are uncalibrated; this program is the "calibrate on a known byte count
// in your own access pattern" step.  Run under rocprofv3 --pmc ...; every kernel name carries its byte count.
//
//   hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip && ./fetch_calib
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t kBytes = 256ull << 20;  // bytes moved by every launch

template <typename T>
__global__ __launch_bounds__(64) void read_kernel(const T *__restrict__ src, float *sink) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    const T v = src[i];
    float acc;
    if constexpr (sizeof(T) == 16) acc = (float)v.x + (float)v.y;
    else if constexpr (sizeof(T) == 8) acc = v.x + v.y;
    else acc = (float)v;
    if (acc == 1234.5678f) sink[0] = acc;  // never true on the zero-filled buffer: keeps the load alive
}

template <typename T, bool WT>
__global__ __launch_bounds__(64) void write_kernel(T *__restrict__ dst, float seed) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if constexpr (sizeof(T) == 16) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        if constexpr (WT) {
            const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, (int)(kBytes), 0x00020000);
            u32x4 d = {(unsigned)i, 1u, 2u, 3u};
            __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)(i * 16), 0, 16 /* sc1 */);
        } else {
            dst[i] = make_float4(seed, 1.f, 2.f, (float)i);
        }
    } else if constexpr (sizeof(T) == 8) {
        dst[i] = make_float2(seed, (float)i);
    } else {
        dst[i] = (T)i;
    }
}

template <typename T>
void run_read(const char *name, const char *buf, float *sink) {
    const size_t n = kBytes / sizeof(T);
    for (int rep = 0; rep < 4; rep++) {  // four disjoint 256 MiB regions: nothing a launch reads was touched by the one before
        hipLaunchKernelGGL((read_kernel<T>), dim3((unsigned)(n / 64)), dim3(64), 0, 0, reinterpret_cast<const T *>(buf + rep * kBytes), sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("%s: 4 launches x %zu bytes\n", name, kBytes);
}

template <typename T, bool WT>
void run_write(const char *name, char *buf) {
    const size_t n = kBytes / sizeof(T);
    for (int rep = 0; rep < 4; rep++) {
        hipLaunchKernelGGL((write_kernel<T, WT>), dim3((unsigned)(n / 64)), dim3(64), 0, 0, reinterpret_cast<T *>(buf + rep * kBytes), 1.0f);
    }
    CHECK(hipDeviceSynchronize());
    printf("%s: 4 launches x %zu bytes\n", name, kBytes);
}

int main() {
    char *buf = nullptr;
    float *sink = nullptr;
    CHECK(hipMalloc(&buf, 4 * kBytes));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(buf, 0, 4 * kBytes));
    CHECK(hipDeviceSynchronize());
    run_read<float>("read4", buf, sink);
    run_read<float2>("read8", buf, sink);
    run_read<float4>("read16", buf, sink);
    run_read<double2>("read16d", buf, sink);
    run_write<unsigned char, false>("write1", buf);
    run_write<float, false>("write4", buf);
    run_write<float2, false>("write8", buf);
    run_write<float4, false>("write16", buf);
    run_write<float4, true>("write16_sc1", buf);
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
