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

// Lengths of 2^28 and more (a rust-htslib Cigar holds a u32) take TWO words: (len & (2^28 - 1)) << 4 | op, then the continuation
// word (len >> 28) << 4 | RB_OP_CONT (include/rustybam_amd.h, "packed ops").  A record that holds one is not "regular": only the
// general kernels ever see the word, and they WALK it as one more op of its owner's type with the length payload << 28 -- to the
// reference's per-base arrays an op and the same op cut in two are the same thing, and every general kernel merges neighbours of
// one type on the way out anyway (paf.rs:602-620).  Where the op itself counts and not its bases (the end-indel strip and its
// quirks, the indel events of stats, the long indels of break-paf, invert) the owner and its continuation are taken together.
#define RB_LEN_BITS_WORD 28
#define RB_LEN_MASK_WORD 0x0FFFFFFFu
// walk form of word i of a record's op array (i counts from the array's first word; a continuation word has an owner in front)
__device__ __forceinline__ uint32_t rb_wopc(const uint32_t *ops, uint32_t i) {
    const uint32_t c = ops[i] & 15u;
    return c == RB_OP_CONT ? (i ? ops[i - 1u] & 15u : RB_NULL_OP) : c;
}
__device__ __forceinline__ uint32_t rb_wlen(const uint32_t *ops, uint32_t i) {
    const uint32_t w = ops[i];
    return (w & 15u) == RB_OP_CONT ? ((w >> 4) & 15u) << RB_LEN_BITS_WORD : w >> 4;
}
// the op that STARTS at word i of ops[0 .. n) / that ENDS at word i: code, whole length; returns the words it takes (1 or 2)
__device__ __forceinline__ uint32_t rb_op_fwd(const uint32_t *ops, uint64_t n, uint64_t i, uint32_t *opc, uint32_t *len) {
    const uint32_t w = ops[i];
    *opc = w & 15u, *len = w >> 4;
    if (*opc != RB_OP_CONT && i + 1 < n && (ops[i + 1] & 15u) == RB_OP_CONT) {
        *len += ((ops[i + 1] >> 4) & 15u) << RB_LEN_BITS_WORD;
        return 2u;
    }
    return 1u;
}
__device__ __forceinline__ uint32_t rb_op_bwd(const uint32_t *ops, uint64_t i, uint32_t *opc, uint32_t *len) {
    const uint32_t w = ops[i];
    if ((w & 15u) == RB_OP_CONT && i > 0 && (ops[i - 1] & 15u) != RB_OP_CONT) {
        *opc = ops[i - 1] & 15u, *len = (ops[i - 1] >> 4) + (((w >> 4) & 15u) << RB_LEN_BITS_WORD);
        return 2u;
    }
    *opc = w & 15u, *len = w >> 4;
    return 1u;
}
// a run of `len` bases of type `opc` as words at out[0..]: returns how many (1 or 2)
__device__ __forceinline__ uint32_t rb_emit_run(uint32_t *out, uint32_t len, uint32_t opc) {
    out[0] = ((len & RB_LEN_MASK_WORD) << 4) | opc;
    if (len >> RB_LEN_BITS_WORD) {
        out[1] = ((len >> RB_LEN_BITS_WORD) << 4) | RB_OP_CONT;
        return 2u;
    }
    return 1u;
}

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
// v_writelane_b32: lane `lane` of `old` becomes the wave-uniform `val` (this clang has no builtin for it; the intrinsic is there)
extern "C" __device__ int rb_llvm_writelane(int, int, int) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t rb_writelane(uint32_t val, uint32_t lane, uint32_t old) {
    return (uint32_t)rb_llvm_writelane((int)val, (int)lane, (int)old);
}
// wave64 ballot straight from the comparison (HIP's __ballot goes through an i32: v_cndmask + v_cmp_ne on top of the v_cmp)
#define rb_ballot(pred) ((unsigned long long)__builtin_amdgcn_ballot_w64(pred))
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

// ---- a ROW of 16 lanes (the unit the DPP row operations work on): k_trim4.hip gives a row to a pair, k_records.hip to a short record ----
#define RB_DPP_ROW_SHL(n) (0x100 + (n))
#define RB_DPP_ROW_ROR(n) (0x120 + (n))

__device__ __forceinline__ uint32_t rb_row_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(1), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(2), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(4), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(8), 0xf, 0xf, false);
    return v;
}
__device__ __forceinline__ uint32_t rb_row_sum(uint32_t v) { // every lane of the row gets the row's sum
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_ROR(8), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_ROR(4), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_ROR(2), 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_ROR(1), 0xf, 0xf, false);
    return v;
}
// lane 15 of the row, to every lane of it.  As inline assembly: through the builtin the compiler folds the move into the instruction
// that uses it (v_subrev_u32_dpp ... row_newbcast:15 bound_ctrl:1), and that form returned the lane's OWN value on the MI355X boxes of
// round 6 (the plain v_mov_b32_dpp is right).  The s_nop covers the two wait states between a VALU write and a DPP read of a register,
// which the compiler's hazard pass does not see inside an asm.
__device__ __forceinline__ uint32_t rb_row_last(uint32_t v) {
    uint32_t r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t rb_row_next(uint32_t v) { // lane + 1 of the row; lane 15 gets 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHL(1), 0xf, 0xf, true);
}
__device__ __forceinline__ uint32_t rb_row_ror(uint32_t v, int n) {
    switch (n) {
    case 8: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_ROR(8), 0xf, 0xf, false);
    case 4: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_ROR(4), 0xf, 0xf, false);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_ROR(2), 0xf, 0xf, false);
    default: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_ROR(1), 0xf, 0xf, false);
    }
}
// the row's 16 bits of a wave ballot (lanes of other rows that sit in other branches do not matter: their bits are cut off)
#define rb_row_ballot(pred, gbase) ((uint32_t)(rb_ballot(pred) >> (gbase)) & 0xFFFFu)
__device__ __forceinline__ uint32_t rb_row_read(uint32_t v, uint32_t gbase, uint32_t l) { // lane l (row-uniform, 0 .. 15) of the row
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)((gbase + l) << 2), (int)v);
}

// 16-byte load of 4 packed ops from a 4-byte-aligned address (global memory tolerates it)
struct __attribute__((packed, aligned(4))) rb_u4_unaligned {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ uint4 rb_load4_unaligned(const uint32_t *p) {
    rb_u4_unaligned t = *reinterpret_cast<const rb_u4_unaligned *>(p);
    return make_uint4(t.x, t.y, t.z, t.w);
}
