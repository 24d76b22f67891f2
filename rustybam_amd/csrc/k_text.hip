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

#ifndef RB_PARSE_STOP
#define RB_PARSE_STOP 0 // diagnostics (timing only): 1 = the fill pass of the parser stops behind the values of a step, 2 = behind the ops in LDS
#endif

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
    int plain_ops;             // != 0: the items of ops[] hold no continuation words (a batch the device parsed: rb_k_parse_cigars makes none)
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

// One step = 1 KiB of text, 16 bytes per lane.  A lane classifies its bytes four at a time (SWAR on the dwords: letters are >= ':',
// anything below '0' or above 0x7f is malformed), which gives a 16-bit map L of where op characters sit; everything about LENGTHS OF
// DIGIT RUNS -- a letter with no digits, ten or more digits, a string that does not end with a letter -- is bit arithmetic on L.
// Only the values need the bytes one after the other: acc = acc * 10 + digit, reset behind a letter (five instructions a byte, no
// branch).  An op takes at least two bytes, so the op whose character sits in byte pair j gets slot j; its code comes from a
// 256-byte table in LDS.  A number that straddles two lanes is completed with the previous lane's unfinished digits (DPP
// wave_shr:1); a number of ten or more digits (zero-padded, or past u32) is left to the host (RB_TEXT_UNUSUAL).
template <bool FILL>
__global__ __launch_bounds__(256) void rb_k_parse_cigars(rb_parse_params p) {
    __shared__ uint32_t stage_all[FILL ? 4 : 1][FILL ? 512 + 64 : 1]; // fill pass: the ops of one step per wave (+ a scrap word per lane)
    __shared__ uint8_t code_lut[FILL ? 256 : 4];                       // op character -> code 0..8, 0x80: not an op character
    if (FILL) {
        code_lut[threadIdx.x] = (uint8_t)(rb_op_code_of(threadIdx.x) == 255u ? 0x80u : rb_op_code_of(threadIdx.x));
        __syncthreads();
    }
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
    uint32_t err = 0;                                 // bit 0: malformed, bit 1: a length >= 2^28, bit 2: left to the host
    uint32_t carry_val = 0, carry_nd = 0;             // unfinished number at the end of lane 63 of the previous step
    for (uint64_t st = 0; st < n_steps; st++) {
        const uint64_t la = a0 + (st << 10) + (uint64_t)lane * 16u;
        uint4 q = make_uint4(0x30303030u, 0x30303030u, 0x30303030u, 0x30303030u);
        if (la < b1) q = *reinterpret_cast<const uint4 *>(p.text + la); // (the buffer is padded to 16 bytes)
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
        const uint32_t first_valid = b0 > la ? (uint32_t)(b0 - la) : 0u;                       // bytes [first_valid, end_valid) of the
        const uint32_t end_valid = b1 > la ? (b1 - la < 16u ? (uint32_t)(b1 - la) : 16u) : 0u; // chunk belong to the string
        const uint32_t fv = first_valid < 16u ? first_valid : 16u;
        if (__ballot(fv != 0u || end_valid != 16u) != 0ull) { // first / last step only: bytes outside the string read as '0'
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int32_t lo = (int32_t)fv - 4 * j, hi = (int32_t)end_valid - 4 * j;
                const uint32_t l2 = lo < 0 ? 0u : (lo > 4 ? 4u : (uint32_t)lo), h2 = hi < 0 ? 0u : (hi > 4 ? 4u : (uint32_t)hi);
                const uint32_t m = h2 > l2 ? ((0xFFFFFFFFu << (8u * l2)) & (0xFFFFFFFFu >> (32u - 8u * h2))) : 0u; // (l2 <= 3, h2 >= 1 here)
                w[j] = (w[j] & m) | (0x30303030u & ~m);
            }
        }
        const bool any_valid = end_valid > fv;
        // where the op characters are
        uint32_t L = 0, junk = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t x = w[j];
            junk |= (x & 0x80808080u) | ((((x | 0x80808080u) - 0x30303030u) ^ 0x80808080u) & 0x80808080u); // > 0x7f, < '0'
            uint32_t t = (((x & 0x7F7F7F7Fu) + 0x46464646u) & 0x80808080u) >> 7;                          // >= ':' -> bit 0 of its byte
            if (!FILL) {
                L += (uint32_t)__builtin_popcount(t); // (the count pass wants nothing else)
            } else {
                t |= t >> 7;
                t |= t >> 14;
                L |= (t & 15u) << (4 * j);
            }
        }
        uint32_t cnt;
        if (!FILL) {
            cnt = L;
        } else {
            cnt = (uint32_t)__builtin_popcount((L | (L >> 1)) & 0x5555u); // byte pairs with an op character = slots filled
            if (junk) err |= 1u;
            // runs of digits, from L alone.  V: the bytes of the string in this chunk
            const uint32_t V = ((1u << end_valid) - 1u) & ~((1u << fv) - 1u);
            if (L & (L << 1)) err |= 1u;                                  // two op characters in a row: no length
            {
                const uint32_t Z = ~L & V;                                  // digits
                uint32_t rr = Z & (Z >> 1);
                rr &= rr >> 2;
                rr &= rr >> 4;                                              // eight in a row
                if (rr & (Z >> 8) & (Z >> 9)) err |= 4u;                    // ten: zero-padded or past u32 -- the host decides
            }
            const bool seen = L != 0u;
            const uint32_t f_pos = seen ? (uint32_t)__builtin_ctz(L) : 16u;           // first op character of the chunk
            const uint32_t lead_nd = seen ? f_pos - fv : 0u;                          // digits in front of it
            const uint32_t l_pos = seen ? 31u - (uint32_t)__builtin_clz(L) : 0u;      // last op character
            const uint32_t trail_nd = any_valid ? (seen ? end_valid - 1u - l_pos : end_valid - fv) : 0u;
            if (la + 16u >= b1 && la < b1 && trail_nd != 0u) err |= 1u;  // the string must end with an op character
            // values: one pass over the bytes
            uint32_t slen[8];
            uint32_t acc = 0;
            const uint32_t Ln = ~L;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t x = w[j >> 1] >> (16 * (j & 1));
                const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)L, 2 * j, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)L, 2 * j + 1, 1);
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_sbfe((int)Ln, 2 * j, 1), k1 = (uint32_t)__builtin_amdgcn_sbfe((int)Ln, 2 * j + 1, 1);
                const uint32_t a_before0 = acc;
                acc = ((acc << 3) + (acc << 1) + (x & 15u)) & k0;
                const uint32_t a_before1 = acc;
                acc = ((acc << 3) + (acc << 1) + ((x >> 8) & 15u)) & k1;
                slen[j] = (a_before0 & m0) | (a_before1 & m1); // (both set: flagged above)
            }
            // my unfinished tail -> the next lane; lane 0 takes the previous step's lane 63
            const uint32_t tail_val = acc, tail_nd = trail_nd;
            const uint32_t in_val = rb_prev_lane(tail_val, carry_val), in_nd = rb_prev_lane(tail_nd, carry_nd);
            // a chunk of digits only is the (short) head of the string, or part of a number of more digits than a lane holds: the
            // second is left to the host
            if (!seen && any_valid && (in_nd != 0u || trail_nd >= 16u)) err |= 4u;
            carry_val = rb_readlane<uint32_t>(tail_val, 63);
            carry_nd = rb_readlane<uint32_t>(tail_nd, 63);
            // the first op of the chunk takes the digits that came in from the lane before
            uint32_t fixed_first = 0;
            if (seen) {
                const uint32_t tnd = in_nd + lead_nd;
                if (tnd == 0u) err |= 1u;     // an op character with no length
                if (tnd >= 10u) err |= 4u;
                uint32_t own = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) own = (f_pos >> 1) == (uint32_t)j ? slen[j] : own;
                fixed_first = in_nd ? in_val * (uint32_t)rb_pow10[lead_nd < 10u ? lead_nd : 9u] + own : own;
            }
            uint32_t mx = fixed_first;
#pragma unroll
            for (int j = 0; j < 8; j++) mx = mx > slen[j] ? mx : slen[j];
            if (mx >= (1u << 28)) err |= 2u; // not representable in the packed form
#if RB_PARSE_STOP == 1
            { // (diagnostics: the step ends behind the values; a dependency on every one of them keeps them computed)
                uint32_t chk = fixed_first;
#pragma unroll
                for (int j = 0; j < 8; j++) chk ^= slen[j];
                if (chk == 0x9E3779B9u) err |= 8u;
                const uint32_t tot_ = rb_wave_sum_u32(cnt);
                out_base += tot_, total += tot_;
                continue;
            }
#endif
            // the ops of this step go through LDS: every lane drops its (at most 8) ops at their rank -- absent slots go to a scrap
            // word -- and the wave then writes the step's ops out side by side
            const uint32_t incl0 = rb_wave_scan_incl(cnt);
            uint32_t *stg = stage_all[wib];
            const uint32_t rank0 = incl0 - cnt;
            const uint32_t P = (L | (L >> 1)) & 0x5555u; // bit 2 j: slot j holds an op
            uint32_t badc = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t x = (w[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                const uint32_t c = ((L >> (2 * j)) & 1u) ? (x & 255u) : (x >> 8);
                const uint32_t code = code_lut[c];
                const bool present = (P >> (2 * j)) & 1u;
                badc |= present ? code : 0u;
                const uint32_t rank = rank0 + (uint32_t)__builtin_popcount(P & ((1u << (2 * j)) - 1u));
                stg[present ? rank : 512u + (uint32_t)lane] = (slen[j] << 4) | (code & 15u);
            }
            if (badc & 0x80u) err |= 1u; // not one of MIDNSHP=X
            if (seen) stg[rank0] = (fixed_first << 4) | (stg[rank0] & 15u);
            const uint32_t step_total_f = rb_readlane<uint32_t>(incl0, 63);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#if RB_PARSE_STOP != 2
            for (uint32_t k = (uint32_t)lane; k < step_total_f; k += 64u)
                if (out_base + k < p.ops_cap) p.ops[out_base + k] = stg[k];
#else
            if (step_total_f == 0x7FFFFFFFu) p.ops[out_base] = stg[lane]; // (diagnostics: nothing leaves LDS)
#endif
            __builtin_amdgcn_wave_barrier();
            out_base += step_total_f;
            total += step_total_f;
            continue;
        }
        total += rb_wave_sum_u32(cnt);
    }
    // (an empty string is an empty CIGAR)
    if (FILL) {
        const uint32_t any_err = rb_wave_or_u32(err);
        if (lane == 0) // (an unusual string may also be a bad one: the host's parser has the last word on it)
            p.status[r] = (uint8_t)((any_err & 4u) ? RB_TEXT_UNUSUAL : ((any_err & 1u) ? RB_TEXT_BAD : ((any_err & 2u) ? RB_TEXT_TOO_LONG : RB_TEXT_OK)));
    } else if (lane == 0) {
        p.op_off[r] = total;
    }
}

// decimal digits of v, at most 9 (an op with a continuation word may have ten: its caller adds the tenth)
__device__ __forceinline__ uint32_t rb_ndigits(uint32_t v) {
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) + (v >= 100000000u);
}

#ifndef RB_FMT_WORDS
#define RB_FMT_WORDS 0 // 1: an op's text as one 8-byte LDS store instead of a byte store per digit -- bit-exact (133 tests), and 2 % SLOWER
                       // (1.85 against 1.81 ms, profiles/r04_text_summary.md): kept as a switch, not the product
#endif
#ifndef RB_FMT_STOP
#define RB_FMT_STOP 0 // diagnostics (timing only, wrong text): 1 = the fill pass stops behind the byte counts of a step, 2 = behind the digits in LDS
#endif
#define RB_FMT_STAGE (256 * 10 + 48) // bytes of text one step of 256 words can make (9 digits + the op character each; an op with a continuation word: 11 bytes for its two words) + the 16-byte phase + 16 bytes of slack in front
// the four low decimal digits of x < 10000, least significant first, as one byte each of a dword: multiplications by constants that
// fit 24 bits (v_mul_u32_u24: full rate; the per-digit x / 10 of round 2 was a v_mul_hi_u32 -- quarter rate -- per digit)
__device__ __forceinline__ uint32_t rb_digits4(uint32_t x) {
    const uint32_t hi = (x * 5243u) >> 19, lo = x - hi * 100u; // x / 100, x % 100 (exact below 43699)
    const uint32_t a = (hi * 103u) >> 10, b = hi - a * 10u;     // y / 10, y % 10 (exact below 179)
    const uint32_t c = (lo * 103u) >> 10, d = lo - c * 10u;
    return d | (c << 8) | (b << 16) | (a << 24);
}
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
    // (the item's five scalars: all loads issued before the first one is waited for -- one trip to memory, not five)
    const uint64_t f_v = p.first[it];
    const uint32_t n_v = p.count[it];
    const uint32_t fl_v = p.first_len ? p.first_len[it] : 0u;
    const uint32_t ll_v = p.last_len ? p.last_len[it] : 0u;
    const uint64_t out_v = FILL ? p.text_off[it] : 0;
    const uint64_t f_raw = rb_first64(f_v);
    const uint32_t *__restrict__ src = (f_raw >> 63) ? p.ops_alt : p.ops;
    const uint64_t f0 = f_raw & ~(1ull << 63);
    const bool look_for_cont = !p.plain_ops || (f_raw >> 63) != 0;
    const uint32_t n = rb_first(n_v);
    const uint32_t fl = rb_first(fl_v);
    const uint32_t ll = rb_first(ll_v);
    uint64_t out = rb_first64(out_v);
    uint64_t bytes = 0;
    uint8_t *stg = stage_all[wib] + (FILL ? 16 : 0); // (16 bytes in front: the digit stores below address from nine bytes before an op's text)
    // A step loads 256 words and prints 255 of them: the last one is only looked at (is it a continuation word? then the word in
    // front of it prints the whole length) and printed by the next step -- unless the item ends inside the step.
    for (uint32_t i0 = 0, lim = 0; i0 < n; i0 = lim) {
        uint32_t len[4], opc[4], nb[4];
        uint32_t mine = 0;
        lim = n - i0 <= 256u ? n : i0 + 255u; // words [i0, lim) are this step's
        // my four ops: one 16-byte load where all four exist (the array may end with the item: no reading past it)
        const uint32_t ib = i0 + (uint32_t)lane * 4u;
        uint32_t vv[4] = {0u, 0u, 0u, 0u};
        if (ib + 3u < n) {
            const uint4 t4 = rb_load4_unaligned(src + f0 + ib);
            vv[0] = t4.x, vv[1] = t4.y, vv[2] = t4.z, vv[3] = t4.w;
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (ib + (uint32_t)q < n) vv[q] = src[f0 + ib + (uint32_t)q];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t i = i0 + (uint32_t)lane * 4u + (uint32_t)q;
            uint32_t v = vv[q];
            uint32_t l = v >> 4;
            if (n == 1u && fl && ll) l = fl + ll - l; // the middle of one op (liftover.rs via subset_cigar, paf.rs:593-620)
            else if (i == 0u && fl) l = fl;             // first op keeps its tail
            else if (i + 1u == n && ll) l = ll;         // last op keeps its head
            len[q] = l;
            opc[q] = v & 15u;
            nb[q] = i < lim ? rb_ndigits(l) + 1u : 0u;
        }
        // continuation words (rb_device.h; only in what the general kernels wrote, never together with first_len / last_len): the
        // word prints nothing, its owner prints the whole length.  Rare: everything about them sits behind one ballot.
        auto b13 = [](uint32_t w) { return __builtin_amdgcn_ubfe(w, 1u, 3u); }; // bits 1..3 of the code: 7 only for 14 (and 15, which no op has)
        bool cont_here = false;
        if (look_for_cont) { // (wave-uniform: not for the clips of a device-parsed batch, which are runs of the records' own ops)
            const uint32_t m01 = b13(vv[0]) > b13(vv[1]) ? b13(vv[0]) : b13(vv[1]), m23 = b13(vv[2]) > b13(vv[3]) ? b13(vv[2]) : b13(vv[3]);
            cont_here = (m01 > m23 ? m01 : m23) == 7u;
        }
        if (look_for_cont && __ballot(cont_here) != 0ull) {
            const uint32_t nxt = (uint32_t)__shfl_down((int)vv[0], 1, 64); // (lane 63's last word is never printed by this step unless the item ends with it)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t nw = q < 3 ? vv[q < 3 ? q + 1 : 3] : nxt;
                if (opc[q] == RB_OP_CONT) {
                    nb[q] = 0u;
                } else if (ib + (uint32_t)q < lim && ib + (uint32_t)q + 1u < n && (nw & 15u) == RB_OP_CONT) {
                    len[q] += ((nw >> 4) & 15u) << RB_LEN_BITS_WORD;
                    nb[q] = rb_ndigits(len[q]) + (len[q] >= 1000000000u ? 1u : 0u) + 1u;
                }
            }
        }
        mine = nb[0] + nb[1] + nb[2] + nb[3];
        const uint32_t incl = rb_wave_scan_incl(mine);
        const uint32_t step_bytes = rb_readlane<uint32_t>(incl, 63);
        if (FILL && RB_FMT_STOP != 1) {
            const uint32_t phase = (uint32_t)(out & 15u);
            uint32_t o = phase + (incl - mine); // place in the stage buffer: byte k of the buffer is byte (out - phase + k) of the text
            // Round 4: where no op of the step has more than four digits (nearly every step), an op's text leaves for LDS as ONE
            // 8-byte store that ENDS at the op's last byte: its digits and character in the top bytes, the text in front of it -- the
            // lane's earlier ops, or the tail of the lane before -- in the bytes below.  Every byte a store carries is the byte that
            // belongs there, so stores may overlap in any order.  (Rounds 2 - 3: one predicated byte store per digit.)
            const bool small_ops = nb[0] <= 5u && nb[1] <= 5u && nb[2] <= 5u && nb[3] <= 5u;
            if (RB_FMT_WORDS && __ballot(!small_ops) == 0ull) {
                unsigned long long S = 0ull, W[4];
                uint32_t e[4], cum[4], c_ = 0u;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t n_ = nb[q];
                    const uint32_t d4 = rb_digits4(len[q]) | 0x30303030u; // digit j of the length in byte j
                    const uint32_t ch = (uint32_t)(uint8_t)("MIDNSHP=X??????"[opc[q] < 9u ? opc[q] : 9u]);
                    const uint32_t hi = __builtin_amdgcn_perm(ch, d4, 0x04000102u);   // bytes: digit 2, digit 1, digit 0, the character
                    const unsigned long long T = ((unsigned long long)hi << 32) | (unsigned long long)(d4 & 0xFF000000u); // ... digit 3 below them
                    const uint32_t sh = 8u * n_;
                    const unsigned long long keep = ~0ull << ((64u - sh) & 63u);       // the top n_ bytes (n_ = 0: not used)
                    S = n_ ? ((S >> sh) | (T & keep)) : S;
                    W[q] = S;
                    c_ += n_, o += n_;
                    cum[q] = c_, e[q] = o;
                }
                // the tail of the lane in front (a lane that has an op is behind a lane with four: at least eight bytes, all its own)
                const uint32_t p_lo = rb_prev_lane((uint32_t)S, 0u), p_hi = rb_prev_lane((uint32_t)(S >> 32), 0u);
                const unsigned long long prevS = ((unsigned long long)p_hi << 32) | p_lo;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    if (nb[q]) {
                        const unsigned long long w = W[q] | (cum[q] < 8u ? (prevS >> (8u * cum[q])) : 0ull);
                        __builtin_memcpy(stg + e[q] - 8u, &w, 8); // (one ds_write_b64 at a byte address: gfx950 takes it)
                    }
                }
            } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (nb[q]) {
                    const uint32_t nd = nb[q] - 1u;
                    uint32_t l = len[q];
                    stg[o + nd] = (uint8_t)("MIDNSHP=X??????"[opc[q] < 9u ? opc[q] : 9u]);
                    // digits, least significant first: digit j goes to byte o + nd - 1 - j = tail[8 - j].  Lengths below 10000 (all but
                    // a handful) take four predicated stores and no division; the rest one quarter-rate multiply per four more digits
                    uint8_t *tail = stg + o + nd - 9u;
                    const bool big = l >= 10000u;
                    uint32_t rest = 0;
                    if (big) {
                        rest = (uint32_t)(((uint64_t)l * 3518437209ull) >> 45); // l / 10000 (exact for 32-bit l)
                        l -= rest * 10000u;
                    }
                    const uint32_t d4 = rb_digits4(l) | 0x30303030u;
                    tail[8] = (uint8_t)d4;
                    if (nd > 1u) tail[7] = (uint8_t)(d4 >> 8);
                    if (nd > 2u) tail[6] = (uint8_t)(d4 >> 16);
                    if (nd > 3u) tail[5] = (uint8_t)(d4 >> 24);
                    if (big) { // digits 4..9 (rest < 429497)
                        const uint32_t top = (uint32_t)(((uint64_t)rest * 3518437209ull) >> 45); // rest / 10000: the ninth and tenth digit (0..42)
                        const uint32_t e4 = rb_digits4(rest - top * 10000u) | 0x30303030u;
                        tail[4] = (uint8_t)e4;
                        if (nd > 5u) tail[3] = (uint8_t)(e4 >> 8);
                        if (nd > 6u) tail[2] = (uint8_t)(e4 >> 16);
                        if (nd > 7u) tail[1] = (uint8_t)(e4 >> 24);
                        const uint32_t t10 = (top * 103u) >> 10; // top / 10 (exact below 179)
                        if (nd > 8u) tail[0] = (uint8_t)(48u + top - t10 * 10u);
                        if (nd > 9u) stg[o] = (uint8_t)(48u + t10); // (the tenth digit: lengths of 10^9 and more)
                    }
                    o += nb[q];
                }
            }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint64_t cap_end = p.text_cap;
            const uint32_t end = phase + step_bytes;                 // buffer bytes [phase, end) are this step's
            const uint32_t a16 = (phase + 15u) & ~15u, b16 = end & ~15u; // whole 16-byte groups [a16, b16)
            uint8_t *__restrict__ dst = p.text + (out - phase);      // 16-byte aligned (text is, by contract)
            if (RB_FMT_STOP != 2 && out + step_bytes <= cap_end) {
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
