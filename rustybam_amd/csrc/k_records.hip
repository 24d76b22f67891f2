// k_records.hip -- K1: one streaming pass over every record's CIGAR (gfx950, wave64).
//
// One wavefront per record.  Each lane loads 4 packed ops (16 B, coalesced 1 KiB per wave
// instruction) per step, accumulates per-class length sums, the bamstats counters and the
// "regular" flags; a single cross-lane reduction at the end of the record produces
//   rb_reduce_row  = infer_n_bases + check_integrity (paf.rs:631-654, :825-857) on the record as
//                    loaded (Paf::from_file, paf.rs:70) + add_stats_from_cigar (bamstats.rs:107-142)
//   rb_norm_row    = remove_trailing_indels (paf.rs:656-783): kept op range, shifted coordinates
//                    (including the quirks at :668-701 and the strand swap :764-769) and the
//                    check_integrity().unwrap() outcome at :782.
// Roofline: HBM read, 4 B per op + 48 B header per record; 128 B of rows written per record.
#include "rb_device.h"
#include <algorithm>
#include <type_traits>

struct rb_scan_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint64_t *t_st, *t_en, *q_st, *q_en;
    const uint8_t *strand;
    rb_reduce_row *reduce_rows;
    rb_norm_row *norm_rows;
    // list mode (fused liftover): only the records list[0 .. *n_list) are scanned
    const uint32_t *list;
    const uint64_t *n_list;
};

// remove_trailing_indels (paf.rs:656-783) needs only the runs of I/D ops at the two ends of a record: how many ops go,
// how many reference / query bases they hold, and the shifted coordinates (with the quirks at :668-701 and the strand
// swap :764-769).  Serial, almost always zero iterations.
struct rb_strip {
    uint32_t lead, trail;         // ops removed in front / behind
    uint64_t dR, dQ;              // reference / query bases of the removed ops
    uint64_t t_st, t_en, q_st, q_en; // coordinates after the shift
};
__device__ __forceinline__ rb_strip rb_strip_ends(const uint32_t *ops, uint64_t n, uint64_t t_st, uint64_t t_en, uint64_t q_st, uint64_t q_en,
                                                  bool minus) {
    rb_strip o;
    uint64_t lead = 0, rm_st_t = 0, rm_st_q = 0;
    uint32_t prev = RB_NULL_OP;
    o.dR = o.dQ = 0;
    while (lead < n) { // paf.rs:663-701 leading run (an op with its continuation word is ONE op here: the quirks count ops)
        uint32_t opc, len;
        const uint32_t words = rb_op_fwd(ops, n, lead, &opc, &len);
        if (!rb_in(RB_INDEL_MASK, opc)) break;
        if (opc == RB_OP_D) {
            rm_st_t += len;
            rm_st_q += 1; // :673
            o.dR += len;
        } else {
            rm_st_q += len;
            o.dQ += len;
        }
        if (prev != RB_NULL_OP && prev != opc) { // :690-701 D/I or I/D neighbours
            rm_st_t += 1;
            rm_st_q -= 1;
        }
        prev = opc;
        lead += words;
    }
    uint64_t trail = 0, rm_en_t = 0, rm_en_q = 0;
    while (trail < n) { // :704-723
        uint32_t opc, len;
        const uint32_t words = rb_op_bwd(ops, n - 1 - trail, &opc, &len);
        if (!rb_in(RB_INDEL_MASK, opc)) break;
        if (opc == RB_OP_D) {
            rm_en_t += len;
            if (lead + trail < n) o.dR += len; // (an all-indel record is reported, its sums are not used)
        } else {
            rm_en_q += len;
            if (lead + trail < n) o.dQ += len;
        }
        trail += words;
    }
    o.lead = (uint32_t)lead;
    o.trail = (uint32_t)trail;
    o.t_st = t_st + rm_st_t, o.t_en = t_en - rm_en_t; // :760-761
    if (minus) {                                      // :764-766
        const uint64_t t = rm_st_q;
        rm_st_q = rm_en_q;
        rm_en_q = t;
    }
    o.q_st = q_st + rm_st_q, o.q_en = q_en - rm_en_q; // :768-769
    return o;
}

// Fused liftover (RB_LIFT_FUSED_SCAN): the clip kernel itself verifies a record while it streams it, so up front only
// the ends are looked at.  One thread per record writes a PROVISIONAL row: kept op range and shifted coordinates are
// final; integrity, regularity, nmatch and aln_len are filled in by the clip kernel (or by the full scan in list mode
// for the records it hands back).
__global__ __launch_bounds__(256) void rb_k_peek_norm(rb_scan_params p) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n_rec) return;
    const uint64_t o0 = p.op_off[r], n = p.op_off[r + 1] - o0;
    rb_norm_row w;
    w.t_st = w.t_en = w.q_st = w.q_en = 0;
    w.first_op = w.n_ops = w.lead_ops = w.trail_ops = w.nmatch = w.aln_len = 0;
    w.flags = RB_F_PROVISIONAL;
    uint32_t st = RB_ST_OK;
    if (n == 0) {
        st = RB_ST_PANIC_EMPTY_CIGAR; // paf.rs:663
    } else {
        const bool minus = p.strand && p.strand[r] == (uint8_t)'-';
        const rb_strip sp = rb_strip_ends(p.ops + o0, n, p.t_st[r], p.t_en[r], p.q_st[r], p.q_en[r], minus);
        w.lead_ops = sp.lead;
        w.trail_ops = sp.trail;
        if (sp.lead || sp.trail) w.flags |= RB_F_STRIPPED;
        if ((uint64_t)sp.lead + sp.trail > n) {
            st = RB_ST_PANIC_ALL_INDEL; // :757
        } else {
            w.t_st = sp.t_st, w.t_en = sp.t_en, w.q_st = sp.q_st, w.q_en = sp.q_en;
            w.first_op = sp.lead;
            w.n_ops = (uint32_t)(n - sp.lead - sp.trail);
            // the streaming kernel wants a match-type op at both ends of the kept range; otherwise the record goes straight to
            // the full scan and the generic kernel
            const uint32_t fo = rb_opc(p.ops[o0 + sp.lead]), lo_ = rb_opc(p.ops[o0 + n - 1 - sp.trail]);
            if (fo > 8u || lo_ > 8u || !rb_in(RB_MATCH_MASK, fo) || !rb_in(RB_MATCH_MASK, lo_)) w.flags |= RB_F_ENDS_NOT_MATCH;
        }
    }
    if (st != RB_ST_OK) w.flags &= ~(uint32_t)RB_F_PROVISIONAL; // nothing left to verify: the reference panics on this record
    w.status = st;
    p.norm_rows[r] = w;
}

// The rows of one record from its sums (one lane): rb_reduce_row = infer_n_bases + check_integrity + the stats counters of the record as
// loaded, rb_norm_row = remove_trailing_indels.  L[t]: bases of op code t (H and P together in L[RB_OP_H]); bad: bit0 an op outside
// M I D N = X, bit1 a zero length, bit2 adjacent ops of one type, bit3 a code above 8.
__device__ __forceinline__ void rb_scan_finish(const rb_scan_params &p, const uint64_t r, const uint64_t (&L)[9], const uint32_t ins_events,
                                               const uint32_t del_events, const uint32_t bad, const uint64_t o0, const uint64_t o1, const uint64_t t_st,
                                               const uint64_t t_en, const uint64_t q_st, const uint64_t q_en, const bool minus) {
    const uint64_t n = o1 - o0;
    const uint64_t R = L[RB_OP_M] + L[RB_OP_D] + L[RB_OP_N] + L[RB_OP_EQ] + L[RB_OP_X];
    const uint64_t Q = L[RB_OP_M] + L[RB_OP_I] + L[RB_OP_S] + L[RB_OP_EQ] + L[RB_OP_X];
    const uint64_t M = L[RB_OP_M] + L[RB_OP_EQ] + L[RB_OP_X];
    uint64_t U = 0;
#pragma unroll
    for (int t = 0; t < 9; t++) U += L[t];
    uint32_t flags = 0;
    if (bad == 0 && U <= 0xFFFFFFFFull) flags |= RB_F_REGULAR;
    if (L[RB_OP_M] != 0) flags |= RB_F_HAS_M;

    if (p.reduce_rows) {
        rb_reduce_row w;
        w.t_bases = R;
        w.q_bases = Q;
        w.nmatch = (uint32_t)M;
        w.aln_len = (uint32_t)U;
        // bamstats.rs:107-127, u32 counters
        w.equal = (uint32_t)L[RB_OP_EQ];
        w.diff = (uint32_t)(L[RB_OP_X] + L[RB_OP_M]);
        w.ins = (uint32_t)L[RB_OP_I];
        w.del = (uint32_t)L[RB_OP_D];
        w.matches = (uint32_t)L[RB_OP_M];
        w.ins_events = ins_events;
        w.del_events = del_events;
        // bamstats.rs:138-142: (100.0 * equal as f32) / (u32 sum) as f32
        const float num = 100.0f * (float)w.equal;
        w.id_by_all = num / (float)(uint32_t)(w.equal + w.diff + w.del + w.ins);
        w.id_by_events = num / (float)(uint32_t)(w.equal + w.diff + w.del_events + w.ins_events);
        w.id_by_matches = num / (float)(uint32_t)(w.equal + w.diff);
        uint32_t st = RB_ST_OK;
        if (U > 0xFFFFFFFFull)
            st = RB_ST_PANIC_OVERFLOW;
        else if (t_en < t_st || t_en - t_st != R)
            st = RB_ST_PANIC_INTEGRITY_T;
        else if (q_en < q_st || q_en - q_st != Q)
            st = RB_ST_PANIC_INTEGRITY_Q;
        w.status = st;
        w.flags = flags;
        p.reduce_rows[r] = w;
    }

    if (p.norm_rows) {
        rb_norm_row w;
        w.t_st = w.t_en = w.q_st = w.q_en = 0;
        w.first_op = w.n_ops = w.lead_ops = w.trail_ops = w.nmatch = w.aln_len = 0;
        w.flags = flags;
        uint32_t st = RB_ST_OK;
        if (n == 0) {
            st = RB_ST_PANIC_EMPTY_CIGAR; // paf.rs:663
        } else {
            const rb_strip sp = rb_strip_ends(p.ops + o0, n, t_st, t_en, q_st, q_en, minus);
            w.lead_ops = sp.lead;
            w.trail_ops = sp.trail;
            if (sp.lead || sp.trail) w.flags |= RB_F_STRIPPED;
            if ((uint64_t)sp.lead + sp.trail > n) {
                st = RB_ST_PANIC_ALL_INDEL; // :757
            } else {
                const uint64_t Rn = R - sp.dR, Qn = Q - sp.dQ, Un = U - sp.dR - sp.dQ; // stripped ops are I/D
                if (Un > 0xFFFFFFFFull)
                    st = RB_ST_PANIC_OVERFLOW;
                else if (sp.t_en < sp.t_st || sp.t_en - sp.t_st != Rn)
                    st = RB_ST_PANIC_INTEGRITY_T; // :782
                else if (sp.q_en < sp.q_st || sp.q_en - sp.q_st != Qn)
                    st = RB_ST_PANIC_INTEGRITY_Q;
                if (st == RB_ST_OK) {
                    w.t_st = sp.t_st;
                    w.t_en = sp.t_en;
                    w.q_st = sp.q_st;
                    w.q_en = sp.q_en;
                    w.first_op = sp.lead;
                    w.n_ops = (uint32_t)(n - sp.lead - sp.trail);
                    w.nmatch = (uint32_t)M;
                    w.aln_len = (uint32_t)Un;
                    // the streaming kernel wants a match-type op at both ends of the kept range (N is never stripped)
                    const uint32_t fo = rb_opc(p.ops[o0 + sp.lead]), lo_ = rb_opc(p.ops[o1 - 1 - sp.trail]);
                    if (!rb_in(RB_MATCH_MASK, fo) || !rb_in(RB_MATCH_MASK, lo_)) w.flags &= ~(uint32_t)RB_F_REGULAR;
                }
            }
        }
        w.status = st;
        p.norm_rows[r] = w;
    }
}

// Per-class accumulation goes through LDS: every lane owns one 64-bit counter per op code
// (hist[code][lane], conflict-free: consecutive lanes hit consecutive banks) and adds
// len | 1 << 40 to it with one ds_add_u64 per op, so the low 40 bits sum the lengths and the high
// 24 bits count the ops (the ins/del "events" of bamstats.rs:112-117).  That replaces nine
// compare/select/64-bit-add chains per op in registers and keeps the kernel HBM-bound.
#define RB_LEN_BITS 40

__global__ __launch_bounds__(256) void rb_k_scan_records(rb_scan_params p) {
    __shared__ unsigned long long hist_all[4][9][64];
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = rb_lane();
    unsigned long long(*hist)[64] = hist_all[threadIdx.x >> 6];
    // list mode: a small fixed grid strides over the (usually empty) list; otherwise one wave per record
    const uint64_t n_work = p.list ? rb_first64(*p.n_list) : p.n_rec;
    const uint64_t stride = p.list ? (uint64_t)gridDim.x * 4u : n_work + 1;
  for (uint64_t wave = wave0; wave < n_work; wave += stride) {
    const uint64_t r = p.list ? (uint64_t)rb_first(p.list[wave]) : rb_first64(wave);
    const uint64_t o0 = p.op_off[r], o1 = p.op_off[r + 1];
    const uint64_t n = o1 - o0;
#pragma unroll
    for (int t = 0; t < 9; t++) hist[t][lane] = 0ull;
    uint32_t bad = 0; // bit0: op outside M I D = X, bit1: zero length, bit2: adjacent ops of one type, bit3: code > 8

    // ---- streaming pass: 8 ops (32 contiguous bytes) per lane and step, two steps in flight in a statically indexed
    //      ring (rotating it with moves, or predicating the loads, makes the compiler wait for every load in flight);
    //      loads past the record's end re-read its last group and are masked on the last step ----
    const uint64_t g0 = o0 & ~3ull;
    const int32_t head = (int32_t)(o0 - g0);
    const uint32_t n32 = (uint32_t)n; // (a record has fewer than 2^32 ops)
    const uint32_t n_steps = n ? (uint32_t)((o1 - g0 + 511u) >> 9) : 0u;
    const uint32_t *__restrict__ gbase = p.ops + g0;
    const uint32_t last_off = n ? (uint32_t)(((o1 - 1u) & ~3ull) - g0) : 0u;
    auto load_half = [&](uint32_t stp, uint32_t half) -> uint4 {
        uint32_t off = (stp << 9) + half * 4u + (uint32_t)lane * 8u;
        off = off < last_off ? off : last_off;
        return *reinterpret_cast<const uint4 *>(gbase + off);
    };
    uint32_t carry_w = 0xFu; // last op word of the previous step in adjacency form (code 15 equals nothing)
    uint32_t v_reg = 0xFFFFFFFFu, v_minlen = 0xFFFFFFFFu, v_adj = 0xFFFFFFFFu, v_big = 0u;
    uint4 pf[2][2];
    if (n_steps) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            pf[q][0] = load_half((uint32_t)q, 0u);
            pf[q][1] = load_half((uint32_t)q, 1u);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const uint32_t my_hist = (uint32_t)lane; // hist[code][lane]
    for (uint32_t st0 = 0; st0 < n_steps; st0 += 2) {
#pragma unroll
        for (int ring = 0; ring < 2; ring++) {
            const uint32_t st = st0 + (uint32_t)ring;
            if (st < n_steps) {
                const uint32_t raw[8] = {pf[ring][0].x, pf[ring][0].y, pf[ring][0].z, pf[ring][0].w,
                                         pf[ring][1].x, pf[ring][1].y, pf[ring][1].z, pf[ring][1].w};
                const int32_t idx0 = (int32_t)(st << 9) + lane * 8 - head;
                auto step = [&](auto edge_c) {
                    constexpr bool edge = decltype(edge_c)::value; // first / last step: ops of the neighbours are skipped
                    uint32_t mylast = raw[7];
                    if (edge && (uint32_t)(idx0 + 7) >= n32) mylast = 0xFu;
                    uint32_t prevw = rb_prev_lane(mylast, carry_w);
                    carry_w = rb_readlane<uint32_t>(mylast, 63);
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const bool ok = !edge || (uint32_t)(idx0 + q) < n32;
                        const uint32_t w = raw[q], opc = w & 15u, len = w >> 4;
                        if (ok) {
                            if (opc <= 8u) {
                                __hip_atomic_fetch_add(&hist[opc][my_hist], (unsigned long long)len | (1ull << RB_LEN_BITS), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
                            } else {
                                v_big |= 8u;
                                // a continuation word: bits 28.. of the length of the op in front of it (no event of its own)
                                if (opc == RB_OP_CONT && (prevw & 15u) <= 8u)
                                    __hip_atomic_fetch_add(&hist[prevw & 15u][my_hist], (unsigned long long)(len & 15u) << RB_LEN_BITS_WORD, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            v_reg &= (uint32_t)__builtin_amdgcn_sbfe((int)0x018F018Fu, w, 1u); // M I D N = X
                            v_minlen = v_minlen < len ? v_minlen : len;
                            const uint32_t x = (w ^ prevw) & 15u;
                            v_adj = v_adj < x ? v_adj : x;
                        }
                        prevw = ok ? w : 0xFu;
                    }
                };
                if (st == 0 || st + 1 == n_steps) step(std::true_type{});
                else step(std::false_type{});
            }
            pf[ring][0] = load_half(st + 2u, 0u);
            pf[ring][1] = load_half(st + 2u, 1u);
        }
    }
    // bit0: op outside M I D = X, bit1: zero length, bit2: adjacent ops of one type, bit3: code > 8
    if (n_steps) bad = (v_reg != 0xFFFFFFFFu ? 1u : 0u) | (v_minlen == 0u ? 2u : 0u) | (v_adj == 0u ? 4u : 0u) | v_big;

    // ---- cross-lane reduction ----
    // Round 5: DPP sums only -- a code's length in two 20-bit pieces (a lane's length sum stays below 2^40, as the LDS counters assume)
    // instead of a 64-bit sum through twelve ds_bpermute; the op counts of I and D alone (the events of bamstats.rs:112-117: nothing asks
    // for the others); H and P, which only the unit total sees, as one sum; the flags by ballot.  18 wave sums where there were 9 through
    // the LDS crossbar and 9 by DPP: 14.4 -> 10.0 ms per 1e7 records of 500 ops with the first of these changes alone
    // (tools/r05_scan_ab.sh, same box, rows equal); -DRB_SCAN_SHFL_SUMS is the old form.
    uint64_t L[9];
#ifndef RB_SCAN_SHFL_SUMS
    auto sum40 = [&](unsigned long long v) -> uint64_t {
        const uint32_t s_lo = rb_wave_sum_u32((uint32_t)v & 0xFFFFFu), s_hi = rb_wave_sum_u32((uint32_t)(v >> 20) & 0xFFFFFu);
        return (uint64_t)s_lo + ((uint64_t)s_hi << 20);
    };
#pragma unroll
    for (int t = 0; t < 9; t++)
        if (t != RB_OP_H && t != RB_OP_P) L[t] = sum40(hist[t][lane]);
    {
        const unsigned long long h = hist[RB_OP_H][lane], q = hist[RB_OP_P][lane]; // piece by piece (a piece of the two together takes 21 bits)
        const uint32_t s_lo = rb_wave_sum_u32(((uint32_t)h & 0xFFFFFu) + ((uint32_t)q & 0xFFFFFu));
        const uint32_t s_hi = rb_wave_sum_u32(((uint32_t)(h >> 20) & 0xFFFFFu) + ((uint32_t)(q >> 20) & 0xFFFFFu));
        L[RB_OP_H] = (uint64_t)s_lo + ((uint64_t)s_hi << 20);
        L[RB_OP_P] = 0;
    }
    const uint32_t ins_events = rb_wave_sum_u32((uint32_t)(hist[RB_OP_I][lane] >> RB_LEN_BITS));
    const uint32_t del_events = rb_wave_sum_u32((uint32_t)(hist[RB_OP_D][lane] >> RB_LEN_BITS));
    bad = (__ballot(bad & 1u) ? 1u : 0u) | (__ballot(bad & 2u) ? 2u : 0u) | (__ballot(bad & 4u) ? 4u : 0u) | (__ballot(bad & 8u) ? 8u : 0u);
#else
    uint32_t C[9];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const unsigned long long v = hist[t][lane];
        L[t] = rb_wave_sum_u64(v & ((1ull << RB_LEN_BITS) - 1ull));
        C[t] = rb_wave_sum_u32((uint32_t)(v >> RB_LEN_BITS));
    }
    const uint32_t ins_events = C[RB_OP_I];
    const uint32_t del_events = C[RB_OP_D];
    bad = rb_wave_or_u32(bad);
#endif
    if (lane != 0) continue;

    rb_scan_finish(p, r, L, ins_events, del_events, bad, o0, o1, p.t_st[r], p.t_en[r], p.q_st[r], p.q_en[r], p.strand && p.strand[r] == (uint8_t)'-');
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Short records, FOUR per wavefront (round 6).  A record of a few hundred ops is one step of the kernel above: a wavefront that waits for
// its op offsets, then for one load, then sums eighteen quantities over 64 lanes and lets lane 0 write two rows -- 1e7 records of 500 ops
// took 9.7 ms, a quarter of what their bytes take (profiles/r05_reclen_summary.md).  Here a record is a ROW of 16 lanes (rb_device.h): a
// lane takes the 16-byte groups gl, gl + 16, ... of its record (a row's load is 256 contiguous bytes), the loads of four steps are in
// flight together, the per-class sums go through the same LDS counters, and the reductions are DPP sums inside the row, so one
// instruction stream finishes four records and four lanes write their rows side by side.  Records of more than RB_SQ_MAX ops, and of
// fewer than 4, are listed for the kernel above (its list mode).  Same sums, same flags, same rows: rb_scan_finish is shared.
// ------------------------------------------------------------------------------------------------------------------------------------
#define RB_SQ_MAX 2048u
#ifndef RB_SQ_DIAG
#define RB_SQ_DIAG 0
#endif
#ifndef RB_SQ_BATCH
#define RB_SQ_BATCH 4 // steps of a row whose loads are in flight together (same box, 1e7 records of 300 - 700 ops: 1 -> 4.96 ms, 2 -> 4.47, 3 - 5 -> 4.25 - 4.34,
                      // 8 -> 4.7 - 4.9, 12 -> 4.95, 16 -> 5.6: registers, i.e. wavefronts per SIMD, are worth more than loads per wavefront)
#endif
__global__ __launch_bounds__(256) void rb_k_scan_rows(rb_scan_params p, unsigned long long *n_long, uint32_t *long_list) {
    __shared__ unsigned long long hist_all[4][9][64];
    const int lane = rb_lane();
    const uint32_t gbase = (uint32_t)lane & 48u, gl = (uint32_t)lane & 15u;
    unsigned long long(*hist)[64] = hist_all[threadIdx.x >> 6];
    const uint64_t r = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u + ((uint32_t)lane >> 4);
    const bool live = r < p.n_rec;
    const uint64_t o0 = live ? p.op_off[r] : 0ull, o1 = live ? p.op_off[r + 1] : 0ull;
    const uint64_t n = o1 - o0;
    const bool take = live && n >= 4u && n <= RB_SQ_MAX;
    if (live && !take && gl == 0u) long_list[atomicAdd(n_long, 1ull)] = (uint32_t)r;
    // (the coordinates: asked for now, used by the row's first lane at the very end)
    uint64_t t_st = 0, t_en = 0, q_st = 0, q_en = 0;
    bool minus = false;
    if (take) t_st = p.t_st[r], t_en = p.t_en[r], q_st = p.q_st[r], q_en = p.q_en[r], minus = p.strand && p.strand[r] == (uint8_t)'-';
#pragma unroll
    for (int t = 0; t < 9; t++) hist[t][lane] = 0ull;
    // aligned 16-byte groups from the group that holds the record's first op; a row's step is 64 ops
    const uint64_t g0 = o0 & ~3ull;
    const int32_t head = (int32_t)(o0 - g0);
    const uint32_t n32 = (uint32_t)n;
    const uint32_t n_steps = take ? (uint32_t)((o1 - g0 + 63u) >> 6) : 0u;
    const uint32_t *__restrict__ gsrc = p.ops + g0;
    const uint32_t last_off = take ? (uint32_t)(((o1 - 1u) & ~3ull) - g0) : 0u;
    uint32_t w_steps = n_steps; // the wavefront's steps: the longest of its four records
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)w_steps, off, 64);
        w_steps = w_steps > o ? w_steps : o;
    }
    w_steps = rb_first(w_steps);
    uint32_t carry_w = 0xFu; // last op word of the row's previous step in adjacency form (code 15 equals nothing)
    uint32_t v_reg = 0xFFFFFFFFu, v_minlen = 0xFFFFFFFFu, v_adj = 0xFFFFFFFFu, v_big = 0u;
    for (uint32_t b0 = 0; b0 < w_steps; b0 += RB_SQ_BATCH) {
        uint4 pf[RB_SQ_BATCH];
#pragma unroll
        for (int s = 0; s < RB_SQ_BATCH; s++) {
            uint32_t off = ((b0 + (uint32_t)s) << 6) + gl * 4u;
            off = off < last_off ? off : last_off; // (past the record's end: its last group again; masked below)
            pf[s] = take ? *reinterpret_cast<const uint4 *>(gsrc + off) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int s = 0; s < RB_SQ_BATCH; s++) {
            const uint32_t st = b0 + (uint32_t)s;
            if (st < n_steps) { // (row-uniform: a row whose record has ended sits the step out)
                const uint32_t raw[4] = {pf[s].x, pf[s].y, pf[s].z, pf[s].w};
                const int32_t idx0 = (int32_t)(st << 6) + (int32_t)gl * 4 - head;
                // a step that is some row's first or last one masks the ops of the neighbouring records; the others are all inside
                const bool any_edge = rb_ballot(st == 0u || st + 1u == n_steps) != 0ull;
                auto step = [&](auto edge_c) {
                    constexpr bool edge = decltype(edge_c)::value;
                    uint32_t mylast = raw[3];
                    if (edge && (uint32_t)(idx0 + 3) >= n32) mylast = 0xFu;
                    uint32_t prevw = (uint32_t)__builtin_amdgcn_update_dpp((int)carry_w, (int)mylast, RB_DPP_ROW_SHR(1), 0xf, 0xf, false);
                    carry_w = rb_row_last(mylast);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const bool ok = !edge || (uint32_t)(idx0 + q) < n32;
                        const uint32_t w = raw[q], opc = w & 15u, len = w >> 4;
                        if (ok) {
                            if (opc <= 8u) {
#if RB_SQ_DIAG == 1 // (diagnostics, timing only: no LDS add -- what the counters cost)
                                v_big += len;
#else
                                __hip_atomic_fetch_add(&hist[opc][lane], (unsigned long long)len | (1ull << RB_LEN_BITS), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
                            } else {
                                v_big |= 8u;
                                // a continuation word: bits 28.. of the length of the op in front of it (no event of its own)
                                if (opc == RB_OP_CONT && (prevw & 15u) <= 8u)
                                    __hip_atomic_fetch_add(&hist[prevw & 15u][lane], (unsigned long long)(len & 15u) << RB_LEN_BITS_WORD, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                            }
                            v_reg &= (uint32_t)__builtin_amdgcn_sbfe((int)0x018F018Fu, w, 1u); // M I D N = X
                            v_minlen = v_minlen < len ? v_minlen : len;
                            const uint32_t x = (w ^ prevw) & 15u;
                            v_adj = v_adj < x ? v_adj : x;
                        }
                        prevw = ok ? w : 0xFu;
                    }
                };
                if (any_edge) step(std::true_type{});
                else step(std::false_type{});
            }
        }
    }
    // ---- the row's sums: a code's length in two 20-bit pieces (a lane's sum stays below 2^40), the events of I and D, the flags ----
    uint64_t L[9];
    auto sum40 = [&](unsigned long long v) -> uint64_t {
        const uint32_t s_lo = rb_row_sum((uint32_t)v & 0xFFFFFu), s_hi = rb_row_sum((uint32_t)(v >> 20) & 0xFFFFFu);
        return (uint64_t)s_lo + ((uint64_t)s_hi << 20);
    };
#pragma unroll
    for (int t = 0; t < 9; t++)
        if (t != RB_OP_H && t != RB_OP_P) L[t] = sum40(hist[t][lane]);
    {
        const unsigned long long h = hist[RB_OP_H][lane], q = hist[RB_OP_P][lane]; // (a piece of the two together takes 21 bits)
        const uint32_t s_lo = rb_row_sum(((uint32_t)h & 0xFFFFFu) + ((uint32_t)q & 0xFFFFFu));
        const uint32_t s_hi = rb_row_sum(((uint32_t)(h >> 20) & 0xFFFFFu) + ((uint32_t)(q >> 20) & 0xFFFFFu));
        L[RB_OP_H] = (uint64_t)s_lo + ((uint64_t)s_hi << 20);
        L[RB_OP_P] = 0;
    }
    const uint32_t ins_events = rb_row_sum((uint32_t)(hist[RB_OP_I][lane] >> RB_LEN_BITS));
    const uint32_t del_events = rb_row_sum((uint32_t)(hist[RB_OP_D][lane] >> RB_LEN_BITS));
    uint32_t bad = (v_reg != 0xFFFFFFFFu ? 1u : 0u) | (v_minlen == 0u ? 2u : 0u) | (v_adj == 0u ? 4u : 0u) | v_big;
    bad = (rb_row_ballot((bad & 1u) != 0u, gbase) ? 1u : 0u) | (rb_row_ballot((bad & 2u) != 0u, gbase) ? 2u : 0u) |
          (rb_row_ballot((bad & 4u) != 0u, gbase) ? 4u : 0u) | (rb_row_ballot((bad & 8u) != 0u, gbase) ? 8u : 0u);
    if (take && gl == 0u) rb_scan_finish(p, r, L, ins_events, del_events, bad, o0, o1, t_st, t_en, q_st, q_en, minus);
}

extern "C" hipError_t rb_launch_peek_norm(const rb_scan_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_peek_norm, dim3((unsigned)((p->n_rec + 255) / 256)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t rb_launch_scan_records(const rb_scan_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    const uint64_t blocks = p->list ? std::min<uint64_t>((p->n_rec + 3) / 4, 1024) : (p->n_rec + 3) / 4;
    hipLaunchKernelGGL(rb_k_scan_records, dim3((unsigned)blocks), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
// a batch of short records (the caller decides: mean ops per record): the row form, then the kernel above over what it listed.
// long_buf: [count (256 B) | n_rec indices], device memory, count zeroed by the caller.
extern "C" hipError_t rb_launch_scan_rows(const rb_scan_params *p, void *long_buf, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    unsigned long long *n_long = (unsigned long long *)long_buf;
    uint32_t *long_list = (uint32_t *)((char *)long_buf + 256);
    hipLaunchKernelGGL(rb_k_scan_rows, dim3((unsigned)((p->n_rec + 15) / 16)), dim3(256), 0, stream, *p, n_long, long_list);
    rb_scan_params q = *p;
    q.list = long_list, q.n_list = (const uint64_t *)n_long;
    const uint64_t blocks = std::min<uint64_t>((p->n_rec + 3) / 4, 8192);
    hipLaunchKernelGGL(rb_k_scan_records, dim3((unsigned)blocks), dim3(256), 0, stream, q);
    return hipGetLastError();
}
