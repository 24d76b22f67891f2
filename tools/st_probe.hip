// st_probe.hip -- diagnostics: does the store SHAPE of the clip kernel matter?  Every wave owns a 20 KiB stretch (a record) and walks it
// in 2 KiB steps, like rb_k_liftover_stream.  Shapes of one step (2 x global_*_dwordx4 per lane):
//   pair : lane l touches bytes [32 l, 32 l + 16) and [32 l + 16, 32 l + 32)   (8 consecutive ops per lane: what the kernel does)
//   flat : lane l touches bytes [16 l, 16 l + 16) and [1024 + 16 l, ...)        (each instruction covers 1 KiB of whole lines)
// hipcc --offload-arch=gfx950 -O3 -o /tmp/st_probe tools/st_probe.hip && /tmp/st_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define STEPS 10
// MODE bit0: flat shape; bit1: read; bit2: write slot 0; bit3: write slot 1 (a second copy 1/5 of the time ~ 1.2x output);
// bit4: the two slots of a record lie side by side (slot k of record w at (2 w + k) * 20 KiB) instead of 20 GB apart; bit5: nt stores
template <int MODE>
__global__ __launch_bounds__(256) void k(const char *__restrict__ src, char *__restrict__ d0, char *__restrict__ d1, size_t n_rec, uint32_t *sink) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_rec) return;
    const int lane = threadIdx.x & 63;
    const size_t base = w * (size_t)(STEPS * 2048) + ((w * 7) & 7) * 16; // records start at any 16-byte phase
    if (MODE & 16) {
        d1 = d0 + (w + 1) * (size_t)(STEPS * 2048);
        d0 = d0 + w * (size_t)(STEPS * 2048);
    }
    const size_t oa = (MODE & 1) ? (size_t)lane * 16 : (size_t)lane * 32, ob = (MODE & 1) ? 1024 + (size_t)lane * 16 : (size_t)lane * 32 + 16;
    uint32_t acc = 0;
    u32x4 a = {1, 2, 3, 4}, b = {5, 6, 7, 8};
#pragma unroll 2
    for (int s = 0; s < STEPS; s++) {
        const size_t o = base + (size_t)s * 2048;
        if (MODE & 2) {
            a = *(const u32x4 *)(src + o + oa);
            b = *(const u32x4 *)(src + o + ob);
            acc += a.x ^ b.y;
        }
        if (MODE & 4) {
            if (MODE & 32) { __builtin_nontemporal_store(a, (u32x4 *)(d0 + o + oa)); __builtin_nontemporal_store(b, (u32x4 *)(d0 + o + ob)); }
            else { *(u32x4 *)(d0 + o + oa) = a; *(u32x4 *)(d0 + o + ob) = b; }
        }
        if ((MODE & 8) && (s % 5) == 0) {
            if (MODE & 32) { __builtin_nontemporal_store(a, (u32x4 *)(d1 + o + oa)); __builtin_nontemporal_store(b, (u32x4 *)(d1 + o + ob)); }
            else { *(u32x4 *)(d1 + o + oa) = a; *(u32x4 *)(d1 + o + ob) = b; }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
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
// `st_probe chunks`: the buffers from 2 MB physical chunks mapped side by side (what rb_dev_alloc does for a resident batch) instead of hipMalloc
static bool g_chunks = false;
static hipError_t alloc(char **p, size_t bytes) {
    if (!g_chunks) return hipMalloc((void **)p, bytes);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t chunk = (size_t)2 << 20, n = (bytes + chunk - 1) / chunk;
    void *va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0);
    if (e != hipSuccess) return e;
    for (size_t k = 0; k < n; k++) {
        hipMemGenericAllocationHandle_t h;
        if ((e = hipMemCreate(&h, chunk, &prop, 0)) != hipSuccess) return e;
        if ((e = hipMemMap((char *)va + k * chunk, chunk, 0, h, 0)) != hipSuccess) return e;
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(va, n * chunk, &acc, 1)) != hipSuccess) return e;
    *p = (char *)va;
    return hipSuccess;
}
int main(int argc, char **argv) {
    g_chunks = argc > 1 && !strcmp(argv[1], "chunks");
    printf("buffers: %s\n", g_chunks ? "2 MB physical chunks" : "hipMalloc");
    const size_t n_rec = 1000000, bytes = n_rec * (size_t)(STEPS * 2048) + 4096;
    char *src, *d0, *d1;
    uint32_t *sink;
    CK(alloc(&src, bytes)); CK(alloc(&d0, 2 * bytes)); CK(alloc(&d1, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(d0, 0, bytes)); CK(hipMemset(d1, 0, bytes));
    const unsigned g = (unsigned)((n_rec + 3) / 4);
    const double gb = (double)n_rec * STEPS * 2048 / 1e9;
#define RUN(name, M, moved) { float ms = timeit([&] { k<M><<<g, 256>>>(src, d0, d1, n_rec, sink); }); printf("%-40s %8.3f ms  %6.2f TB/s\n", name, ms, (moved) / ms); }
    RUN("read  pair", 2, gb);
    RUN("read  flat", 3, gb);
    RUN("write pair (1 slot)", 4, gb);
    RUN("write flat (1 slot)", 5, gb);
    RUN("write pair (1.2x, 2 slots)", 12, 1.2 * gb);
    RUN("write flat (1.2x, 2 slots)", 13, 1.2 * gb);
    RUN("read + write pair (1.2x, 2 slots)", 14, 2.2 * gb);
    RUN("read + write flat (1.2x, 2 slots)", 15, 2.2 * gb);
    RUN("write pair (1.2x, slots side by side)", 12 + 16, 1.2 * gb);
    RUN("write flat (1.2x, slots side by side)", 13 + 16, 1.2 * gb);
    RUN("read + write pair (side by side)", 14 + 16, 2.2 * gb);
    RUN("read + write flat (side by side)", 15 + 16, 2.2 * gb);
    RUN("read + write pair nt (2 slots)", 14 + 32, 2.2 * gb);
    RUN("read + write flat nt (side by side)", 15 + 16 + 32, 2.2 * gb);
    RUN("read + write pair (1 slot, 1.0x)", 6, 2.0 * gb);
    RUN("read + write flat (1 slot, 1.0x)", 7, 2.0 * gb);
    return 0;
}
