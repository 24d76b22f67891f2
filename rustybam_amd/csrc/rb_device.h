// rb_device.h -- device-side helpers shared by the gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/rustybam_amd.h"

#define RB_WAVE 64
#define RB_NULL_OP 15u /* padding op code: consumes nothing, never equals a real op */

// op-class bitmasks indexed by BAM op code (paf.rs:946-975)
#define RB_REF_MASK 0x18Du   /* M D N = X   */
#define RB_QRY_MASK 0x193u   /* M I S = X   */
#define RB_MATCH_MASK 0x181u /* M = X       */
#define RB_INDEL_MASK 0x006u /* I D         */
#define RB_REGULAR_MASK 0x18Fu /* M I D N = X: what the streaming clip kernel handles (N behaves like D, but is never stripped) */

__device__ __forceinline__ uint32_t rb_opc(uint32_t v) { return v & 15u; }
__device__ __forceinline__ uint32_t rb_len(uint32_t v) { return v >> 4; }
__device__ __forceinline__ bool rb_in(uint32_t mask, uint32_t opc) { return (mask >> opc) & 1u; }

__device__ __forceinline__ int rb_lane() { return (int)(threadIdx.x & 63u); }

template <typename T>
__device__ __forceinline__ T rb_readlane(T v, int lane);
template <>
__device__ __forceinline__ uint32_t rb_readlane<uint32_t>(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
template <>
__device__ __forceinline__ int rb_readlane<int>(int v, int lane) {
    return __builtin_amdgcn_readlane(v, lane);
}
template <>
__device__ __forceinline__ uint64_t rb_readlane<uint64_t>(uint64_t v, int lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t rb_first(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t rb_first64(uint64_t v) {
    uint32_t lo = rb_first((uint32_t)v), hi = rb_first((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in the body
template <int N, int I = 0, typename F>
__device__ __forceinline__ void rb_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rb_static_for<N, I + 1>(f);
    }
}

// DPP controls (GFX9 encoding)
#define RB_DPP_ROW_SHR(n) (0x110 + (n))
#define RB_DPP_WAVE_SHR1 0x138
#define RB_DPP_ROW_BCAST15 0x142
#define RB_DPP_ROW_BCAST31 0x143

// wave64 inclusive prefix sum of one u32 per lane: 4 row_shr steps inside each row of 16 lanes,
// then row_bcast:15 / row_bcast:31 to carry row totals (6 DPP adds, no LDS).
__device__ __forceinline__ uint32_t rb_wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(8), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_BCAST15, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_BCAST31, 0xc, 0xf, false);
    return v;
}

// wave64 inclusive prefix maximum of one u32 per lane (same network as the sum)
__device__ __forceinline__ uint32_t rb_wave_scan_incl_max_u32(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(1), 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(2), 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(4), 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(8), 0xf, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_BCAST15, 0xa, 0xf, false));
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_BCAST31, 0xc, 0xf, false));
    return v;
}

// value of lane-1 (lane 0 receives `carry`)
__device__ __forceinline__ uint32_t rb_prev_lane(uint32_t v, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, RB_DPP_WAVE_SHR1, 0xf, 0xf, false);
}

__device__ __forceinline__ uint32_t rb_wave_sum_u32(uint32_t v) {
    return rb_readlane<uint32_t>(rb_wave_scan_incl(v), 63);
}
__device__ __forceinline__ uint64_t rb_wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ uint32_t rb_wave_or_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v |= __shfl_xor(v, off, 64);
    return v;
}

// 16-byte load of 4 packed ops from a 4-byte-aligned address (global memory tolerates it)
struct __attribute__((packed, aligned(4))) rb_u4_unaligned {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ uint4 rb_load4_unaligned(const uint32_t *p) {
    rb_u4_unaligned t = *reinterpret_cast<const rb_u4_unaligned *>(p);
    return make_uint4(t.x, t.y, t.z, t.w);
}
