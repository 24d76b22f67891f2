// vmcnt_probe.hip -- does a VMEM instruction issued with EXEC == 0 count in vmcnt on gfx950?
// L (real load into a sentinel-filled register), then a store / load issued with an empty exec mask, then
// s_waitcnt vmcnt(1) and an immediate read of L's destination.  If the masked instruction counts (and retires in
// order), vmcnt(1) implies L has landed; if it does not count, vmcnt(1) lets L stay in flight and the sentinel shows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint32_t *src, uint32_t *dst, uint32_t *bad, int mode, size_t stride) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * stride; // cold, far-apart addresses
    uint32_t v = 0xDEADBEEFu, w = 0x12345678u;
    unsigned long long sv;
    const uint32_t *p = src + i;
    uint32_t *q = dst + i;
    if (mode == 0) { // masked STORE between
        asm volatile("global_load_dword %[v], %[p], off\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, 0\n\t"
                     "global_store_dword %[q], %[w], off\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %[w], %[v]\n\t"
                     "s_waitcnt vmcnt(0)"
                     : [v] "+v"(v), [w] "+v"(w), [sv] "=&s"(sv) : [p] "v"(p), [q] "v"(q) : "memory");
    } else if (mode == 1) { // masked LOAD between
        uint32_t x = 0;
        asm volatile("global_load_dword %[v], %[p], off\n\t"
                     "s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, 0\n\t"
                     "global_load_dword %[x], %[q], off\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %[w], %[v]\n\t"
                     "s_waitcnt vmcnt(0)"
                     : [v] "+v"(v), [w] "+v"(w), [x] "+v"(x), [sv] "=&s"(sv) : [p] "v"(p), [q] "v"(q) : "memory");
    } else { // control: nothing between; vmcnt(1) does not cover L, the sentinel MUST show
        asm volatile("global_load_dword %[v], %[p], off\n\t"
                     "s_waitcnt vmcnt(1)\n\t"
                     "v_mov_b32 %[w], %[v]\n\t"
                     "s_waitcnt vmcnt(0)"
                     : [v] "+v"(v), [w] "+v"(w) : [p] "v"(p) : "memory");
    }
    if (w == 0xDEADBEEFu) atomicAdd(bad, 1u);
}
int main() {
    const size_t stride = 4096, n = 256 * 1024;
    uint32_t *src, *dst, *bad;
    hipMalloc(&src, n * stride * 4);
    hipMalloc(&dst, n * stride * 4);
    hipMalloc(&bad, 4);
    hipMemset(src, 1, n * stride * 4);
    for (int mode = 0; mode < 3; mode++) {
        hipMemset(bad, 0, 4);
        probe<<<n / 256, 256>>>(src, dst, bad, mode, stride);
        uint32_t h = 0;
        hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
        printf("%s: sentinel seen in %u of %zu lanes -> %s\n", mode == 0 ? "masked store" : mode == 1 ? "masked load " : "control     ", h, n,
               mode == 2 ? (h ? "control works (an uncovered load is caught)" : "control failed: probe cannot tell")
                         : (h ? "does NOT count in vmcnt" : "counts in vmcnt (or retired early)"));
    }
    return 0;
}
