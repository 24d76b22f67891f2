// bw_probe.hip -- diagnostics: what this box sustains for the access mixes the clip kernels produce.
// hipcc --offload-arch=gfx950 -O3 -o tools/bw_probe tools/bw_probe.hip && tools/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode bits: 1 = non-temporal stores, 2 = non-temporal loads
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16, int wfrac256) {
    // every block moves UNROLL * 256 groups; writes only the first wfrac256/256 of what it reads... (wfrac256 > 256: writes more)
    const size_t base = (size_t)blockIdx.x * (UNROLL * 256) + threadIdx.x;
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) v[u] = (MODE & 2) ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) {
            if (MODE & 1) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u];
        }
    }
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const u32x4 *__restrict__ src, uint32_t *sink, size_t n16) {
    const size_t base = (size_t)blockIdx.x * (UNROLL * 256) + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) { u32x4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_write(u32x4 *__restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * (UNROLL * 256) + threadIdx.x;
    const u32x4 v = {1u, 2u, 3u, (uint32_t)blockIdx.x};
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const size_t i = base + (size_t)u * 256;
        if (i < n16) { if (MODE & 1) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
    }
}
// read A bytes, write B bytes (B = 1.25 A): the clip kernels' mix; every block reads 4 KiB x UNROLL and writes 5/4 of it
template <int MODE, int UNROLL>
__global__ __launch_bounds__(256) void k_mix(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16) {
    const size_t base = (size_t)blockIdx.x * (UNROLL * 256) + threadIdx.x;
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const size_t i = base + (size_t)u * 256;
        v[u] = i < n16 ? src[i] : u32x4{0, 0, 0, 0};
    }
    const size_t obase = (size_t)blockIdx.x * (UNROLL * 320) + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
        const size_t o = obase + (size_t)u * 320;
        if (MODE & 1) __builtin_nontemporal_store(v[u], dst + o); else dst[o] = v[u];
        if (threadIdx.x < 64) { if (MODE & 1) __builtin_nontemporal_store(v[u], dst + o + 256); else dst[o + 256] = v[u]; }
    }
}
template <typename F>
static float timeit(F f, int reps = 5) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const size_t bytes = 20ull << 30; // 20 GiB source
    const size_t n16 = bytes / 16;
    u32x4 *src, *dst;
    uint32_t *sink;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes / 4 * 5 + (1 << 20)));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, bytes));
    CK(hipMemset(dst, 0, bytes / 4 * 5));
    const double gb = bytes / 1e9;
#define RUN(name, call, moved) { float ms = timeit([&] { call; }); printf("%-44s %8.3f ms  %7.2f TB/s\n", name, ms, (moved) / ms / 1e9 * 1e3 / 1e3); }
    { const unsigned g = (unsigned)((n16 + 4 * 256 - 1) / (4 * 256));
      RUN("read  x4", (k_read<4><<<g, 256>>>(src, sink, n16)), gb);
      RUN("write x4", (k_write<0, 4><<<g, 256>>>(dst, n16)), gb);
      RUN("write x4 nt", (k_write<1, 4><<<g, 256>>>(dst, n16)), gb);
      RUN("copy  x4", (k_copy<0, 4><<<g, 256>>>(src, dst, n16, 256)), 2 * gb);
      RUN("copy  x4 nt-store", (k_copy<1, 4><<<g, 256>>>(src, dst, n16, 256)), 2 * gb);
      RUN("copy  x4 nt-load nt-store", (k_copy<3, 4><<<g, 256>>>(src, dst, n16, 256)), 2 * gb);
      RUN("mix r20+w25 x4", (k_mix<0, 4><<<g, 256>>>(src, dst, n16)), 2.25 * gb);
      RUN("mix r20+w25 x4 nt-store", (k_mix<1, 4><<<g, 256>>>(src, dst, n16)), 2.25 * gb); }
    { const unsigned g = (unsigned)((n16 + 8 * 256 - 1) / (8 * 256));
      RUN("read  x8", (k_read<8><<<g, 256>>>(src, sink, n16)), gb);
      RUN("copy  x8", (k_copy<0, 8><<<g, 256>>>(src, dst, n16, 256)), 2 * gb);
      RUN("copy  x8 nt-store", (k_copy<1, 8><<<g, 256>>>(src, dst, n16, 256)), 2 * gb);
      RUN("mix r20+w25 x8 nt-store", (k_mix<1, 8><<<g, 256>>>(src, dst, n16)), 2.25 * gb); }
    { const unsigned g = (unsigned)((n16 + 1 * 256 - 1) / (1 * 256));
      RUN("read  x1", (k_read<1><<<g, 256>>>(src, sink, n16)), gb);
      RUN("copy  x1", (k_copy<0, 1><<<g, 256>>>(src, dst, n16, 256)), 2 * gb);
      RUN("copy  x1 nt-store", (k_copy<1, 1><<<g, 256>>>(src, dst, n16, 256)), 2 * gb); }
    return 0;
}
