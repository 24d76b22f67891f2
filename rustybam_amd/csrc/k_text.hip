// k_text.hip -- CIGAR text <-> packed ops on the device, gfx950 (wave64, CDNA4).
//
// The data format either side of the CIGAR walk: a PAF line carries its CIGAR as text
// (`cg:Z:12=1X3I...`, 95 % of the bytes of an assembly PAF), and every output line prints one again.
//
//   rb_k_parse_cigars   replaces CigarString::try_from(value.as_bytes()) at paf.rs:398-399 (rust-htslib 0.44.1:
//                       decimal u32 length, then one op character of MIDNSHP=X; anything else is the
//                       `.expect("Unable to parse cigar string.")` panic -> per-record status here).
//                       One wavefront per record, 16 text bytes per lane and step; a lane walks its bytes as a
//                       tiny state machine, a number that straddles two lanes is completed with the previous lane's
//                       unfinished digits (DPP wave_shr:1); a wave scan of the per-lane op counts places the ops.
//                       Run twice: count (ops per record -> exclusive scan -> op_off), then fill.
//   rb_k_format_cigars  replaces `impl Display for CigarString` as used by `impl Display for PafRecord`
//                       (paf.rs:923-944): items = runs of ops with an optional clipped first / last length
//                       (exactly what the clip descriptors of rb_dev_liftover describe), so clipped CIGARs are
//                       printed straight from the record's original ops.  One wavefront per item, 4 ops per lane
//                       and step; count (bytes per item -> scan -> text_off), then fill.
//
// Roofline: HBM (byte work, no MFMA).  Algorithmic bytes: parse = text bytes read twice + 4 B per op written;
// format = 4 B per op read twice + text bytes written.
#include "rb_device.h"

struct rb_parse_params {
    uint64_t n_rec;
    const uint8_t *text;      // all CIGAR strings, any layout
    const uint64_t *text_off; // [n_rec + 1] record r's string is text[text_off[r] .. text_end[r])
    const uint64_t *text_end; // [n_rec] (NULL: strings are back to back, end = text_off[r + 1])
    uint64_t *op_off;         // [n_rec + 1] counts (count pass) / offsets (fill pass)
    uint32_t *ops;
    uint64_t ops_cap;
    uint8_t *status;          // [n_rec] RB_TEXT_*
};
struct rb_format_params {
    uint64_t n_items;
    const uint32_t *ops;
    const uint32_t *ops_alt;   // second source: items whose first[] has bit 63 set index this array (NULL if unused)
    const uint64_t *first;     // [n_items] index of the item's first op in ops[] (bit 63: in ops_alt[])
    const uint32_t *count;     // [n_items] ops in the item (0 = empty text)
    const uint32_t *first_len; // [n_items] or NULL: != 0 replaces the length of the first op
    const uint32_t *last_len;  // [n_items] or NULL: != 0 replaces the length of the last op; a one-op item with both keeps first + last - len
    uint64_t *text_off;        // [n_items + 1] counts / offsets
    uint8_t *text;
    uint64_t text_cap;
};

// op character -> code (MIDNSHP=X -> 0..8), 255 = not an op.  Branch-free: the nine characters lie in '=' (61) .. 'X' (88),
// so the code is a nibble of a packed table indexed by c - 61 (15 = not an op).
__device__ __forceinline__ uint32_t rb_op_code_of(uint32_t c) {
    const unsigned long long lo = 0xfff15fff2ffffff7ull; // offsets 0..15 from '=': '=' 0 -> 7, 'D' 7 -> 2, 'H' 11 -> 5, 'I' 12 -> 1
    const unsigned long long hi = 0xffff8ffff4ff6f30ull; // offsets 16..31: 'M' 16 -> 0, 'N' 17 -> 3, 'P' 19 -> 6, 'S' 22 -> 4, 'X' 27 -> 8
    const uint32_t o = c - 61u;
    const unsigned long long t = (o & 16u) ? hi : lo;
    const uint32_t nib = (uint32_t)(t >> ((o & 15u) * 4u)) & 15u;
    return (o < 32u && nib != 15u) ? nib : 255u;
}

__device__ __constant__ uint64_t rb_pow10[10] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull};

template <bool FILL>
__global__ __launch_bounds__(256) void rb_k_parse_cigars(rb_parse_params p) {
    __shared__ uint32_t stage_all[FILL ? 4 : 1][FILL ? 512 + 64 : 1]; // fill pass: the ops of one step per wave (+ a scrap word per lane)
    const uint32_t wib = rb_first(threadIdx.x >> 6);
    const uint64_t r = (uint64_t)blockIdx.x * 4u + wib;
    if (r >= p.n_rec) return;
    const int lane = rb_lane();
    const uint64_t b0 = rb_first64(p.text_off[r]);
    const uint64_t b1 = rb_first64(p.text_end ? p.text_end[r] : p.text_off[r + 1]);
    const uint64_t a0 = b0 & ~15ull;                  // aligned start: bytes before b0 are masked
    const uint64_t n_steps = (b1 - a0 + 1023u) >> 10; // 1 KiB of text per step
    uint64_t out_base = FILL ? rb_first64(p.op_off[r]) : 0; // next op slot of this record
    uint32_t total = 0;                               // ops found
    uint32_t err = 0;                                 // RB_TEXT_* of this lane
    // unfinished number at the end of lane 63 of the previous step
    uint32_t carry_val = 0, carry_nd = 0;
    for (uint64_t st = 0; st < n_steps; st++) {
        const uint64_t la = a0 + (st << 10) + (uint64_t)lane * 16u;
        uint4 q = make_uint4(0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u);
        if (la < b1) q = *reinterpret_cast<const uint4 *>(p.text + la); // (the buffer is padded to 16 bytes)
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
        // lane-local walk: ops that end in this chunk.  An op takes at least two bytes, so the op whose character sits
        // in byte pair k / 2 gets slot k / 2 (statically indexed registers); the first one may have begun in the
        // previous lane.
        uint32_t slen[8], sinfo[8]; // length so far; code | digits << 8 | overflow << 16 | present << 24
#pragma unroll
        for (int j = 0; j < 8; j++) slen[j] = 0u, sinfo[j] = 0u;
        uint32_t cnt = 0;
        uint32_t acc = 0;     // value of the number being read (32-bit; `ovf` says it went past u32::MAX)
        bool ovf = false;
        uint32_t nd = 0;      // digits of the number being read
        uint32_t lead_nd = 0; // digits before the first op character of the chunk (to combine with the carry)
        bool seen = false, any_valid = false;
        // (written with selects, not branches: sixteen divergent ifs per lane cost ~1600 scalar instructions of exec-mask
        //  handling per step and made the kernel scalar-issue-bound)
        const uint32_t first_valid = b0 > la ? (uint32_t)(b0 - la) : 0u;                     // bytes [first_valid, end_valid) of the
        const uint32_t end_valid = b1 > la ? (b1 - la < 16u ? (uint32_t)(b1 - la) : 16u) : 0u; // chunk belong to the string
        uint32_t errb = 0; // bit 0: malformed, bit 1: a length >= 2^28
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const bool valid = (uint32_t)k >= first_valid && (uint32_t)k < end_valid;
            const uint32_t c = (w[k >> 2] >> ((k & 3) * 8)) & 255u;
            const uint32_t d = c - 48u;
            const bool isdig = valid & (d < 10u), islet = valid & !(d < 10u);
            const uint32_t code = rb_op_code_of(c);
            // digit: the value leaves u32 (leading zeros may run on), acc * 10 + d without the slow 32-bit multiply
            ovf |= isdig & ((acc > 429496729u) | ((acc == 429496729u) & (d > 5u)));
            const uint32_t acc10 = ((acc << 3) + (acc << 1)) + d;
            // op character
            const bool later = islet & seen; // (the first op of the chunk is completed below, with the incoming digits)
            errb |= (islet & (code == 255u)) ? 1u : 0u;
            errb |= (later & ((nd == 0u) | ovf)) ? 1u : 0u;                      // no length / overflow of u32
            errb |= (later & !ovf & (acc >= (1u << 28))) ? 2u : 0u;              // not representable in the packed form
            errb |= (islet & (sinfo[k >> 1] != 0u)) ? 1u : 0u;                   // two op characters in one byte pair
            lead_nd = (islet & !seen) ? nd : lead_nd;
            slen[k >> 1] = islet ? acc : slen[k >> 1];                           // (completed below for the first op)
            sinfo[k >> 1] = islet ? ((code & 15u) | (ovf ? 0x10000u : 0u) | (seen ? 0u : 0x20000u) | 0x1000000u) : sinfo[k >> 1];
            cnt += islet ? 1u : 0u;
            seen |= islet;
            any_valid |= valid;
            acc = isdig ? acc10 : (islet ? 0u : acc);
            nd = isdig ? nd + 1u : (islet ? 0u : nd);
            ovf = ovf & !islet;
        }
        if (errb & 1u) err = RB_TEXT_BAD;
        else if ((errb & 2u) && err == 0) err = RB_TEXT_TOO_LONG;
        // the string must end with an op character
        if (la + 16u >= b1 && la < b1 && nd != 0u) err = RB_TEXT_BAD;
        // my unfinished tail -> the next lane; lane 0 takes the previous step's lane 63
        const uint32_t tail_val = acc, tail_nd = nd | (ovf ? 0x100u : 0u);
        const uint32_t in_val = rb_prev_lane(tail_val, carry_val), in_nd = rb_prev_lane(tail_nd, carry_nd);
        // a chunk of digits only is the (short) head of the string, or part of a number padded with zeros to more than a
        // lane holds: legal for the reference (u32::from_str), not handled here -> the host decides (RB_TEXT_UNUSUAL)
        if (!seen && any_valid && ((in_nd & 0xFFu) != 0u || nd >= 16u)) err = err ? err : RB_TEXT_UNUSUAL;
        carry_val = rb_readlane<uint32_t>(tail_val, 63);
        carry_nd = rb_readlane<uint32_t>(tail_nd, 63);
        const uint32_t incl = rb_wave_scan_incl(cnt);
        const uint32_t step_total = rb_readlane<uint32_t>(incl, 63);
        if (FILL) {
            // the ops of this step go through LDS: every lane drops its (at most 8) ops at their rank -- absent slots go to a
            // scrap word, so there is no branch per slot -- and the wave then writes the step's ops out side by side
            uint32_t *stg = stage_all[wib];
            uint32_t rank = incl - cnt;
            // the first op of the chunk (at most one per lane) takes the digits that came in from the lanes before: done once, outside
            // the slot loop, so that the loop has no branch
            uint32_t l_first = 0, first_ovf = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const bool f = (sinfo[j] & 0x20000u) != 0u;
                l_first = f ? slen[j] : l_first;
                first_ovf = f ? (sinfo[j] & 0x10000u) : first_ovf;
            }
            uint32_t fixed_first = l_first;
            if (seen) {
                const uint32_t ind = in_nd & 0xFFu;
                const uint32_t tnd = ind + lead_nd;
                uint64_t full = l_first;
                if (ind && in_val != 0u) // (incoming zeros change nothing; a non-zero value followed by ten more digits is past u32::MAX)
                    full = lead_nd >= 10u ? ~0ull : (uint64_t)in_val * rb_pow10[lead_nd] + l_first; // < 2^32 * 10^9
                if ((in_nd & 0x100u) || first_ovf) full = ~0ull;
                if (tnd == 0u || full > 0xFFFFFFFFull) err = (err == 0 || err == RB_TEXT_TOO_LONG) ? RB_TEXT_BAD : err;
                else if (full >= (1ull << 28) && err == 0) err = RB_TEXT_TOO_LONG;
                fixed_first = (uint32_t)full;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const bool present = (sinfo[j] & 0x1000000u) != 0u;
                const uint32_t l = (sinfo[j] & 0x20000u) ? fixed_first : slen[j];
                stg[present ? rank : 512u + (uint32_t)lane] = (l << 4) | (sinfo[j] & 15u);
                rank += present ? 1u : 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (uint32_t k = (uint32_t)lane; k < step_total; k += 64u)
                if (out_base + k < p.ops_cap) p.ops[out_base + k] = stg[k];
            __builtin_amdgcn_wave_barrier();
            out_base += step_total;
        }
        total += step_total;
    }
    // (an empty string is an empty CIGAR)
    const uint32_t any_err = rb_wave_or_u32(err == RB_TEXT_BAD ? 1u : (err == RB_TEXT_TOO_LONG ? 2u : (err == RB_TEXT_UNUSUAL ? 4u : 0u)));
    if (lane == 0) {
        if (!FILL) p.op_off[r] = total;
        else // (an unusual string may also be a bad one: the host's parser has the last word on it)
            p.status[r] = (uint8_t)((any_err & 4u) ? RB_TEXT_UNUSUAL : ((any_err & 1u) ? RB_TEXT_BAD : ((any_err & 2u) ? RB_TEXT_TOO_LONG : RB_TEXT_OK)));
    }
}

// decimal digits of v (v < 2^28: at most 9)
__device__ __forceinline__ uint32_t rb_ndigits(uint32_t v) {
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u);
}

#define RB_FMT_STAGE (256 * 10 + 32) // bytes of text one step of 256 ops can make (9 digits + the op character each) + the 16-byte phase
template <bool FILL>
__global__ __launch_bounds__(256) void rb_k_format_cigars(rb_format_params p) {
    // fill pass: the text of a step is put together in LDS (one byte per digit, at the 16-byte phase it has in memory) and leaves with
    // aligned 16-byte stores; only the ragged head and tail of a step -- bytes of 16-byte groups it shares with its neighbours -- go
    // out byte by byte.  (One global byte store per character made this pass 1.8e9 scattered stores per 1.8 GB.)
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[FILL ? 4 : 1][FILL ? RB_FMT_STAGE : 16];
    const uint32_t wib = rb_first(threadIdx.x >> 6);
    const uint64_t it = (uint64_t)blockIdx.x * 4u + wib;
    if (it >= p.n_items) return;
    const int lane = rb_lane();
    const uint64_t f_raw = rb_first64(p.first[it]);
    const uint32_t *__restrict__ src = (f_raw >> 63) ? p.ops_alt : p.ops;
    const uint64_t f0 = f_raw & ~(1ull << 63);
    const uint32_t n = rb_first(p.count[it]);
    const uint32_t fl = p.first_len ? rb_first(p.first_len[it]) : 0u;
    const uint32_t ll = p.last_len ? rb_first(p.last_len[it]) : 0u;
    uint64_t out = FILL ? rb_first64(p.text_off[it]) : 0;
    uint64_t bytes = 0;
    uint8_t *stg = stage_all[wib];
    for (uint32_t i0 = 0; i0 < n; i0 += 256u) {
        uint32_t len[4], opc[4], nb[4];
        uint32_t mine = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t i = i0 + (uint32_t)lane * 4u + (uint32_t)q;
            uint32_t v = i < n ? src[f0 + i] : 0u;
            uint32_t l = v >> 4;
            if (n == 1u && fl && ll) l = fl + ll - l; // the middle of one op (liftover.rs via subset_cigar, paf.rs:593-620)
            else if (i == 0u && fl) l = fl;             // first op keeps its tail
            else if (i + 1u == n && ll) l = ll;         // last op keeps its head
            len[q] = l;
            opc[q] = v & 15u;
            nb[q] = i < n ? rb_ndigits(l) + 1u : 0u;
            mine += nb[q];
        }
        const uint32_t incl = rb_wave_scan_incl(mine);
        const uint32_t step_bytes = rb_readlane<uint32_t>(incl, 63);
        if (FILL) {
            const uint32_t phase = (uint32_t)(out & 15u);
            uint32_t o = phase + (incl - mine); // place in the stage buffer: byte k of the buffer is byte (out - phase + k) of the text
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (nb[q]) {
                    const uint32_t nd = nb[q] - 1u;
                    uint32_t l = len[q];
                    stg[o + nd] = (uint8_t)("MIDNSHP=X??????"[opc[q] < 9u ? opc[q] : 9u]);
                    for (uint32_t k = nd; k-- > 0u;) {
                        const uint32_t t = l / 10u;
                        stg[o + k] = (uint8_t)(48u + (l - t * 10u));
                        l = t;
                    }
                    o += nb[q];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t cap_end = p.text_cap;
            const uint32_t end = phase + step_bytes;                 // buffer bytes [phase, end) are this step's
            const uint32_t a16 = (phase + 15u) & ~15u, b16 = end & ~15u; // whole 16-byte groups [a16, b16)
            uint8_t *__restrict__ dst = p.text + (out - phase);      // 16-byte aligned (text is, by contract)
            if (out + step_bytes <= cap_end) {
                if (a16 < b16) {
                    for (uint32_t k = a16 + 16u * (uint32_t)lane; k < b16; k += 1024u)
                        *reinterpret_cast<uint4 *>(dst + k) = *reinterpret_cast<const uint4 *>(stg + k);
                    if ((uint32_t)lane < a16 - phase) dst[phase + (uint32_t)lane] = stg[phase + (uint32_t)lane];         // head (< 16 bytes)
                    if ((uint32_t)lane < end - b16) dst[b16 + (uint32_t)lane] = stg[b16 + (uint32_t)lane];               // tail (< 16 bytes)
                } else { // a step of a few bytes inside one or two groups
                    for (uint32_t k = phase + (uint32_t)lane; k < end; k += 64u) dst[k] = stg[k];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            out += step_bytes;
        }
        bytes += step_bytes;
    }
    if (!FILL && lane == 0) p.text_off[it] = bytes;
}

extern "C" hipError_t rb_launch_parse_cigars(const rb_parse_params *p, bool fill, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((p->n_rec + 3) / 4);
    if (fill) hipLaunchKernelGGL(rb_k_parse_cigars<true>, dim3(blocks), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(rb_k_parse_cigars<false>, dim3(blocks), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t rb_launch_format_cigars(const rb_format_params *p, bool fill, hipStream_t stream) {
    if (p->n_items == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((p->n_items + 3) / 4);
    if (fill) hipLaunchKernelGGL(rb_k_format_cigars<true>, dim3(blocks), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(rb_k_format_cigars<false>, dim3(blocks), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
