// k_tile.hip -- the clip kernel for SHORT records: one wavefront per TILE of consecutive records (gfx950, wave64).
//
// rb_k_liftover_stream (k_liftover.hip) gives every record a wavefront of its own.  A record of 500 ops is one or two steps of
// that kernel's stream, and everything around the stream -- the job, the windows, the resolution of the boundaries, the verdict
// of the fused scan, the rows -- is paid per record: 1082 vector + 779 scalar instructions and five dependent memory trips for 2 KB
// of ops (profiles/r05_reclen_summary.md: 0.21 - 0.25 of the HBM roofline on BASELINE config 4's record shape).  Here a wave takes a
// TILE: up to RBT_REC records that lie one behind the other in the ops array, 8 .. short_max ops each, RBT_OPS - 32 ops together
// (rb_plan_create cuts the tiles; records longer than short_max keep the per-record kernel).  The tile is streamed like ONE long
// record -- the same load ring, the same per-lane sums and wave scans, the same speculative stores into the positional output slots --
// with running totals that simply run on across the records; what belongs to a record is done afterwards, one LANE per record or per
// hit:
//
//   set-up      lane j = record j of the tile: its job (rb_k_make_jobs), eligibility, the reference offset P_j at which it starts in
//               the tile's running totals (a scan of the header spans t_en - t_st: the CIGAR sums equal them, or the tile is handed
//               back).  lane h = hit h of the tile (the records' hits one behind the other, at most RBT_HITS): window, the two
//               boundary offsets D = P_j + offset, kept in the lane.
//   stream      as in the per-record kernel; a record boundary inside a lane's 8 ops only matters to the check "no two adjacent ops of
//               one type" (the neighbours belong to different records) and to break-paf's cut state.  Checkpoints (R, Q, U before
//               every 8 ops) for the whole tile stay in LDS: there is one resolution, behind the stream.
//   verdict     P_j measured (checkpoint + the ops in front of the record in its chunk), record totals against the headers, the
//               fused scan's conditions over the whole tile.  ANY failure hands the whole tile to the per-record kernel (fb_list):
//               nothing has been written by then but speculative stores into slot lines that only this tile's records own.
//   resolve     lane h: start boundary, then end boundary (rb_resolve, rb_lift.h -- the reference's tpos_to_idx + walk rules,
//               paf.rs:541-561), both against the LDS checkpoints made relative to the hit's record.
//   rows        lane h: the row (liftover.rs:57-104), the two end words of its clip, list entries for what the general kernels take.
//
// Results are those of the per-record kernel (same rows, same clips; only out_off differs: a tile's clips lie in the slot lines of
// its FIRST record, 32 ra + position).  Roofline: HBM; algorithmic bytes as for rb_k_liftover_stream.
#include "rb_lift.h"
#include <type_traits>

#ifndef RBT_OPS
#define RBT_OPS 4096 // ops a tile's stream covers at most (its records + up to 31 ops in front of the first one)
#endif
#define RBT_STEPS (RBT_OPS / 512)
#define RBT_CP (RBT_OPS / 8) // one checkpoint per lane and step
#define RBT_REC 32           // records per tile at most
#define RBT_HITS 64          // hits (liftover) / pieces (break-paf) per tile at most: one lane each
#define RBT_GAP_MAX 1024u    // RB_LIFT_OP_STARTS: ops between two records of a tile at most (more: the per-record kernel takes the tile)
#ifndef RBT_RING_BASE
#define RBT_RING_BASE 80
#endif
#ifndef RBT_WPE
#define RBT_WPE 5, 6
#endif
#define RBT_STR2(x) #x
#define RBT_STR(x) RBT_STR2(x)
#ifndef RBT_PF
#define RBT_PF 2 // steps of stream loads in flight per wave (2 KiB each): the ring is 8 RBT_PF registers
#endif
#ifndef RBT_SPILL_ROOM
#define RBT_SPILL_ROOM 0 // registers between the compiler's allocation and the ring (it parks spilled scalar registers right behind its own)
#endif
#define RBT_RING_TOP_N (RBT_RING_BASE + 8 * RBT_PF - 1)
#if RBT_RING_TOP_N == 79
#define RBT_RING_TOP "v79"
#elif RBT_RING_TOP_N == 87
#define RBT_RING_TOP "v87"
#elif RBT_RING_TOP_N == 95
#define RBT_RING_TOP "v95"
#elif RBT_RING_TOP_N == 103
#define RBT_RING_TOP "v103"
#elif RBT_RING_TOP_N == 111
#define RBT_RING_TOP "v111"
#elif RBT_RING_TOP_N == 119
#define RBT_RING_TOP "v119"
#elif RBT_RING_TOP_N == 127
#define RBT_RING_TOP "v127"
#else
#error "RBT_RING_BASE + 8 RBT_PF - 1: 79, 87, 95, 103, 111, 119 or 127"
#endif
// the load ring: RBT_PF steps of 8 registers, outside the compiler's allocation (amdgpu_num_vgpr), named literally -- k_liftover.hip says why
#define RBT_RREG(OFF, W) "v[" RBT_STR(RBT_RING_BASE) "+" #OFF ":" RBT_STR(RBT_RING_BASE) "+" #OFF "+" #W "]"
#define RBT_RING_CASE(RING, M)                                                                                                  \
    if constexpr ((RING) == 0) { M(0, 2, 4, 6) } else if constexpr ((RING) == 1) { M(8, 10, 12, 14) } else if constexpr ((RING) == 2) { M(16, 18, 20, 22) } else { M(24, 26, 28, 30) }
static_assert(RBT_PF >= 2 && RBT_PF <= 4, "two to four steps in flight");
#define RBT_STEP_VMEM (2 * RB_MS + 2) // vector-memory instructions a step issues, always
#define RBT_RING_WAIT ((RBT_PF - 1) * RBT_STEP_VMEM)
#define RBT_GRAN 16 // speculative stores are widened to whole 64-byte granules (liftover form)
#ifndef RBT_STOP
#define RBT_STOP 0 // diagnostics (tools/mkvariant.sh --src k_tile.hip -DRBT_STOP=n; wrong results, only the time is of interest): a tile ends
                   // 1 behind its set-up, 2 behind its stream, 3 behind the resolution of its boundaries
#endif
#ifndef RBT_NOSTORE
#define RBT_NOSTORE 0 // diagnostics: no speculative store moves anything (the read side alone)
#endif

typedef uint32_t rbt_u32x4 __attribute__((ext_vector_type(4)));

template <bool BRK>
__device__ __forceinline__ void rb_tile_body() {
    const rb_kparams kp = (rb_kparams)__builtin_amdgcn_kernarg_segment_ptr();
    rb_kparams kq = rb_kp_here(kp); // (fields are read where they are used, k_liftover.hip)
#define p (*kq)
    __shared__ uint32_t cp_all[4][3][RBT_CP];
    __shared__ uint32_t bq_all[4][RBT_CP / 4];  // per chunk of 8 ops: where in it a record starts (8: nowhere), one byte each
    __shared__ uint32_t hr_all[4][RBT_HITS];    // set-up scratch: the record of every hit
    __shared__ uint32_t gq_all[4][RBT_CP / 4];  // RB_LIFT_OP_STARTS: per chunk of 8 ops, the ops that lie in a gap between two records, one bit each
    const uint32_t wib = rb_first(threadIdx.x >> 6);
    const uint32_t tile = blockIdx.x * 4u + wib;
    if (tile >= p.n_tiles) return;
    const int lane = rb_lane();
    uint32_t *cpR = cp_all[wib][0], *cpQ = cp_all[wib][1], *cpU = cp_all[wib][2];
    uint8_t *bq_s = reinterpret_cast<uint8_t *>(bq_all[wib]);
    uint32_t *hr_s = hr_all[wib];
    uint8_t *gq_s = reinterpret_cast<uint8_t *>(gq_all[wib]);
    uint32_t ra = rb_first(p.tile_first[3u * tile]);
    uint32_t nrec = rb_first(p.tile_first[3u * tile + 1u]);
    uint32_t slot0 = rb_first(p.tile_first[3u * tile + 2u]); // (the jobs of a tile's records lie side by side)
    const bool passthrough = (ra >> 31) != 0u;
    ra &= 0x7FFFFFFFu;
    // the tile goes to the per-record kernel as it is
    auto fallback = [&]() {
        unsigned long long b0 = 0;
        if (lane == 0) b0 = atomicAdd(p.fb_count, (unsigned long long)nrec);
        b0 = rb_first64(b0);
        if ((uint32_t)lane < nrec) p.fb_list[b0 + (uint32_t)lane] = ra + (uint32_t)lane;
    };
    if (passthrough || nrec == 0u || nrec > RBT_REC) {
        if (nrec) fallback();
        return;
    }
    const bool fused = p.fused != 0;
    const bool explicit_w = !BRK && p.x_st != nullptr;
    const bool desc_mode = p.desc_mode != 0;
    const uint32_t n_slots = (uint32_t)p.n_slots;
    const uint32_t ns1 = n_slots ? n_slots : 1u;

    // ---- set-up, lane j = record ra + j ----
    // RB_LIFT_OP_STARTS (a batch trim-paf has cut in place, include/rustybam_amd.h): op_off is a table of starts, a record's extent is its
    // row's (the job's), and the records of a tile lie one behind the other with GAPS between them -- what the clips left of their ends
    const bool starts = p.op_starts != 0;
    bool isrec, active = false, gone = false;
    uint32_t r, slot, jflags = 0, jn = 0, jnh = 0, jlo = 0, jh0 = 0, spanR = 0, spanQ = 0;
    uint64_t oo0 = 0, oo1 = 0, jrec0 = 0, t_st = 0, t_en = 0, tot = 0;
    // the records' jobs into the lanes; the cut where the tile's hits fill the lanes; which records the tile kernel can take.  Returns the
    // lanes whose record it cannot (gone: the whole tile went to the per-record kernel).  Run once -- or twice: below.
    auto setup = [&]() -> uint64_t {
        isrec = (uint32_t)lane < nrec;
        r = ra + (isrec ? (uint32_t)lane : 0u);
        slot = slot0 + (isrec ? (uint32_t)lane : 0u);
        oo0 = p.op_off[r], oo1 = starts ? 0ull : p.op_off[r + 1];
        const rb_job *jp = &p.jobs[slot];
        jrec0 = jp->rec0, jn = jp->n, jflags = jp->flags, jnh = jp->nh, jlo = jp->lo, jh0 = jp->h0;
        t_st = jp->t_st, t_en = jp->t_en;
        if constexpr (!BRK) {
            // more hits than lanes: the tile is cut behind the last record whose hits still fit -- the records behind it go to the per-record
            // kernel, the tile kernel keeps the front (dense windows over short records: a tile of 31 records with three hits each)
            const uint32_t nh0 = (isrec && (jflags & RB_JOB_VALID) != 0u) ? jnh : 0u;
            const uint32_t inc0 = rb_wave_scan_incl(nh0);
            if (rb_readlane<uint32_t>(inc0, 63) > RBT_HITS) {
                const uint32_t m = (uint32_t)__builtin_popcountll(rb_ballot(isrec && inc0 <= RBT_HITS)); // (the counts do not decrease: a run of low lanes)
                if (m == 0u) {
                    fallback();
                    gone = true;
                    return 0ull;
                }
                unsigned long long b0 = 0;
                if (lane == 0) b0 = atomicAdd(p.fb_count, (unsigned long long)(nrec - m));
                b0 = rb_first64(b0);
                if ((uint32_t)lane >= m && (uint32_t)lane < nrec) p.fb_list[b0 + ((uint32_t)lane - m)] = ra + (uint32_t)lane;
                nrec = m;
                isrec = (uint32_t)lane < nrec;
            }
        }
        const uint64_t q_st = jp->q_st, q_en = jp->q_en;
        active = (jflags & RB_JOB_VALID) != 0u;
        const bool passive = !BRK && (jflags & (RB_JOB_VALID | RB_JOB_ROWS_OVERFLOW)) == 0u && jnh == 0u; // (no window overlaps it and nothing is to be verified: its ops only run past)
        const uint64_t sR = t_en - t_st, sQ = q_en - q_st;
        bool ok = (active || passive) && (jflags & RB_JOB_REGULAR) != 0u && (BRK || explicit_w || (jflags & RB_JOB_MONO) != 0u) &&
                  (starts || (jrec0 == oo0 && (uint64_t)jn == oo1 - oo0)) && jn >= 8u && t_en >= t_st && q_en >= q_st && sR < (1ull << 31) && sQ < (1ull << 31);
        spanR = isrec ? (uint32_t)sR : 0u, spanQ = isrec ? (uint32_t)sQ : 0u;
        if (!isrec) ok = true, active = false;
        tot = rb_wave_sum_u64(isrec && ok ? sR + sQ : 0ull); // (U <= R + Q: below 2^32 every running total of the tile is exact)
        return rb_ballot(!ok);
    };
    uint64_t badm = setup();
    if (gone) return;
    if (badm != 0ull) {
        // A record the tile kernel cannot take (irregular CIGAR, stripped end indels, an extent that is not its offsets', ...): the tile is
        // cut to its longest run of records it can take, everything else goes to the per-record kernel -- the record itself, which that
        // kernel hands to the general ones, and the shorter side of the tile -- and the set-up runs once more over the run (straight
        // line, twice: a loop around the set-up cost the scalar bases of the ring their wave-uniformity in the compiler's eyes).
        // (Round 5 handed the whole tile back: with 1 % of such records among records of 300 - 700 ops that was 7.7 % of all records
        // and a quarter to a third more time for the step.)
        uint32_t best_s = 0, best_n = 0, prev = 0; // longest run [best_s, best_s + best_n) of lanes without a bad record
        uint64_t m = badm;
        for (;;) {
            const uint32_t bb = m ? (uint32_t)__builtin_ctzll(m) : nrec; // (bad records are records: bb < nrec)
            if (bb - prev > best_n) best_s = prev, best_n = bb - prev;
            if (!m) break;
            m &= m - 1;
            prev = bb + 1u;
        }
        if (best_n == 0u) {
            fallback();
            return;
        }
        const uint64_t out_m = ((nrec < 64u ? (1ull << nrec) : 0ull) - 1ull) & ~(((best_n < 64u ? (1ull << best_n) : 0ull) - 1ull) << best_s); // the records outside the run
        unsigned long long b0 = 0;
        if (lane == 0) b0 = atomicAdd(p.fb_count, (unsigned long long)__builtin_popcountll(out_m));
        b0 = rb_first64(b0);
        if ((out_m >> lane) & 1ull) p.fb_list[b0 + (uint32_t)__builtin_popcountll(out_m & ((1ull << lane) - 1ull))] = ra + (uint32_t)lane;
        ra = rb_first(ra + best_s), slot0 = rb_first(slot0 + best_s), nrec = rb_first(best_n);
        badm = setup();
        if (gone) return;
        if (badm != 0ull) { // (cannot be: the run holds none)
            fallback();
            return;
        }
    }
    if (tot >= (1ull << 32)) {
        fallback();
        return;
    }
    const uint32_t PassR = rb_wave_scan_incl(spanR) - spanR; // where record j starts in the tile's running reference total, if its CIGAR sums to its header
    const uint64_t g_first = rb_first64(jrec0);              // (lane 0: the tile's first op)
    const uint64_t g0 = g_first & ~31ull;
    const uint64_t gend = rb_first64(rb_readlane<uint64_t>(jrec0 + jn, (int)nrec - 1));
    // the gap in front of record j (lane j): its ops are streamed like everybody else's and count for nothing
    uint32_t gap = 0;
    bool has_gaps = false;
    if (starts) {
        const uint64_t end_prev = __shfl_up(jrec0 + jn, 1, 64);
        const bool mine = isrec && lane != 0;
        const bool bad_order = mine && (jrec0 < end_prev || jrec0 - end_prev > RBT_GAP_MAX);
        if (rb_ballot(bad_order) != 0ull || gend < g_first) { // (a record that was moved elsewhere, or an extent nobody should stream)
            fallback();
            return;
        }
        gap = mine ? (uint32_t)(jrec0 - end_prev) : 0u;
        has_gaps = rb_ballot(gap != 0u) != 0ull;
    }
    const uint32_t n_tile = (uint32_t)(gend - g_first);
    const uint32_t n_steps = (uint32_t)((gend - g0 + 511u) >> 9);
    if (n_steps > RBT_STEPS) { // (rb_plan_create does not make such a tile)
        fallback();
        return;
    }
    const uint32_t head = (uint32_t)(g_first - g0);
    const uint32_t rel0 = isrec ? (uint32_t)(jrec0 - g0) : 0u; // the record's first op, counted from g0
    const uint32_t *__restrict__ gbase0 = p.ops + g0;
    const uint32_t first_boff = head * 4u;
    const uint32_t last_boff = (uint32_t)(((gend - 1u) & ~3ull) - g0) * 4u;
    const uint32_t last_cboff = last_boff & ~31u;
    unsigned long long sv_exec;
    asm volatile("s_mov_b64 %0, exec" : "=s"(sv_exec));
    const uint32_t lane_boff = (uint32_t)lane * 32u;
#define RBT_RING_LOAD_ASM(A, B_, C_, D_)                                                                                        \
    asm volatile("s_mov_b64 exec, %[lm]\n\t"                                                                                    \
                 "global_load_dwordx4 " RBT_RREG(A, 3) ", %[o], %[sb]\n\t"                                                      \
                 "global_load_dwordx4 " RBT_RREG(C_, 3) ", %[o], %[sb] offset:16\n\t"                                           \
                 "s_mov_b64 exec, %[sv]"                                                                                        \
                 :                                                                                                              \
                 : [o] "v"(lo_), [sb] "s"(gb_), [lm] "s"(lm_), [sv] "s"(sv_)                                                    \
                 : "memory", RBT_RING_TOP);
#define RBT_RING_LOAD(RING, STP)                                                                                                \
    {                                                                                                                           \
        const uint32_t stp_ = (STP);                                                                                            \
        uint32_t lo_ = (stp_ << 11) + lane_boff;                                                                                \
        lo_ = lo_ < last_cboff ? lo_ : last_cboff;                                                                              \
        const uint32_t *const gb_ = gbase0;                                                                                     \
        const unsigned long long sv_ = sv_exec;                                                                                 \
        const unsigned long long lm_ = stp_ < n_steps ? sv_ : 0ull;                                                             \
        RBT_RING_CASE(RING, RBT_RING_LOAD_ASM)                                                                                  \
    }
#define RBT_RING_NOSTORES                                                                                                       \
    _Pragma("unroll") for (int q_ = 0; q_ < 2 * RB_MS; q_++)                                                                    \
        asm volatile("s_mov_b64 exec, 0\n\tglobal_store_dword %0, %0, %1\n\ts_mov_b64 exec, %2" ::"v"(0u), "s"(gbase0), "s"(sv_exec) : "memory");

    // hits of the tile, one lane each: the records' hits one behind the other
    const uint32_t nhj = (!BRK && active) ? jnh : 0u;
    const uint32_t hb_incl = rb_wave_scan_incl(nhj);
    const uint32_t hb = hb_incl - nhj;
    uint32_t H = rb_readlane<uint32_t>(hb_incl, 63);
    if (!BRK && H > RBT_HITS) {
        fallback();
        return;
    }
    if (RBT_STOP == 1) return;
    // the ring's first loads go out HERE, in front of the window loads of the hits (a dependent trip of their own), not behind them
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): the ring's wait counts count from here
    RBT_RING_LOAD(0, 0u)
    { RBT_RING_NOSTORES }
    RBT_RING_LOAD(1, 1u)
#if RBT_PF >= 3
    { RBT_RING_NOSTORES }
    RBT_RING_LOAD(2, 2u)
#endif
#if RBT_PF >= 4
    { RBT_RING_NOSTORES }
    RBT_RING_LOAD(3, 3u)
#endif
    // Dst / Den: the hit's boundaries as offsets in the tile's running reference total, D = (offset of the boundary base) + 1 as in
    // rb_k_liftover_stream; hmeta = record (tile-local) | ordinal of the hit in its record << 8
    uint32_t Dst = 0xFFFFFFFFu, Den = 0xFFFFFFFFu, hmeta = 0u;
    bool inside = false;
    if constexpr (!BRK) {
        for (uint32_t k = 0; rb_ballot(k < nhj) != 0ull; k++)
            if (k < nhj) hr_s[hb + k] = (uint32_t)lane;
        const bool ish = (uint32_t)lane < H;
        const uint32_t hj = ish ? hr_s[lane] : 0u;
        const uint64_t ht_st = __shfl(t_st, (int)hj, 64), ht_en = __shfl(t_en, (int)hj, 64);
        const uint32_t hlo = (uint32_t)__shfl((int)jlo, (int)hj, 64), hhb = (uint32_t)__shfl((int)hb, (int)hj, 64);
        const uint32_t hP = (uint32_t)__shfl((int)PassR, (int)hj, 64), hh0 = (uint32_t)__shfl((int)jh0, (int)hj, 64);
        const uint32_t jl = (uint32_t)lane - hhb;
        if (ish) {
            uint64_t wst, wen;
            if (explicit_w) wst = p.x_st[(uint64_t)hh0 + jl], wen = p.x_en[(uint64_t)hh0 + jl];
            else wst = p.w_st[(uint64_t)hlo + jl], wen = p.w_en[(uint64_t)hlo + jl];
            inside = ht_st > wst && ht_en < wen;                                        // liftover.rs:23-25
            Dst = hP + (uint32_t)((wst > ht_st ? wst : ht_st) - ht_st) + 1u;            // liftover.rs:28
            Den = hP + (uint32_t)((wen < ht_en ? wen : ht_en) - ht_st);                 // :38-40
            hmeta = hj | (jl << 8);
        }
    }
    // per class (output slot) the hits that belong to it, and those the stream is done with
    unsigned long long cmask[RB_MS], cfin[RB_MS];
#pragma unroll
    for (int q = 0; q < RB_MS; q++) {
        cmask[q] = BRK ? 0ull : rb_ballot((uint32_t)lane < H && ((hmeta >> 8) % ns1) == (uint32_t)q && (uint32_t)q < ns1);
        cfin[q] = 0ull;
    }
    // where records start inside a chunk (the first record's start is the tile's: nothing to tell)
    for (uint32_t k = (uint32_t)lane; k < RBT_CP / 4; k += 64u) bq_all[wib][k] = 0x08080808u;
    if (isrec && lane != 0) bq_s[rel0 >> 3] = (uint8_t)(rel0 & 7u);
    // which ops of a chunk are gap ops, one bit each (a chunk holds ops of one gap at most: records are 8 ops and more)
    if (has_gaps) {
        for (uint32_t k = (uint32_t)lane; k < RBT_CP / 4; k += 64u) gq_all[wib][k] = 0u;
        for (uint32_t a = rel0 - gap; a < rel0;) { // (lanes without a gap: no trip)
            const uint32_t c = a >> 3, e = (c + 1u) << 3 < rel0 ? (c + 1u) << 3 : rel0;
            gq_s[c] = (uint8_t)(((1u << (e - (c << 3))) - 1u) & ~((1u << (a & 7u)) - 1u));
            a = e;
        }
    }

    uint32_t *const out_ops_ = p.out_ops;
    const uint64_t slot_stride_ = p.slot_stride;
    const uint32_t brk_max_ = BRK ? p.brk_max : 0u;
    const uint64_t slot_row0 = 32ull * ra + g0; // out_ops index of coordinate 0 (counted from g0) in slot 0: the lines of the tile's first record
    const bool spec = n_slots != 0u && !desc_mode && (BRK || H != 0u);
    // ---- the stream ----
    uint32_t Rb = 0, Qb = 0, Ub = 0; // running totals of the tile
    uint32_t v_reg = 0u, v_minw = 0xFFFFFFFFu, v_adj = 0xFFFFFFFFu, v_maxsu = 0u, v_carry = 0xFu;
    unsigned long long v_utot = 0;
    // break-paf: the cut state runs along the tile.  cur = the record being streamed, brk_pre = where its open piece starts (an offset
    // in the tile's running reference total), brk_cnt = pieces closed so far = the lane of the open one, brk_p0 = brk_cnt when the
    // record began.  Per record (its lane): rp0 = its first piece, rcnt = its pieces.
    uint32_t cur = 0, brk_pre = 0, brk_cnt = 0, brk_p0 = 0;
    uint32_t rp0 = 0, rcnt = 0;
    bool brk_over = false;
    auto brk_open = [&]() { // piece brk_cnt opens at brk_pre (its lanes are rewritten if it turns out to hold no reference base)
        if (brk_cnt < RBT_HITS) {
            const uint32_t jl = brk_cnt - brk_p0;
            Dst = rb_writelane((brk_pre + 1u), brk_cnt, Dst);
            Den = rb_writelane(0xFFFFFFFFu, brk_cnt, Den);
            hmeta = rb_writelane((cur | (jl << 8)), brk_cnt, hmeta);
            const unsigned long long bit = 1ull << brk_cnt;
#pragma unroll
            for (int q = 0; q < RB_MS; q++) cmask[q] = (cmask[q] & ~bit) | ((jl % ns1) == (uint32_t)q ? bit : 0ull);
        } // (no lane left: only a piece that CLOSES there is one too many -- brk_close; one that turns out to hold no reference base is not a piece)
    };
    auto brk_close = [&](const uint32_t rx) { // liftover.rs:191 / :213-224: the open piece ends in front of offset rx, if it holds reference bases
        if (rx > brk_pre) {
            if (brk_cnt < RBT_HITS) Den = rb_writelane(rx, brk_cnt, Den);
            else brk_over = true; // a 65th piece: the tile goes to the per-record kernel
            brk_cnt++;
        }
    };
    if constexpr (BRK) brk_open();

    uint32_t c_h[RB_MS], c_ds[RB_MS], c_de[RB_MS];
    auto clip_fetch = [&](const int q) { // the current clip of class q: the first one the stream is not done with
        const unsigned long long m = cmask[q] & ~cfin[q];
        const uint32_t h = m ? (uint32_t)__builtin_ctzll(m) : 64u;
        const uint32_t hh = h < 64u ? h : 0u;
        const uint32_t ds_ = rb_readlane<uint32_t>(Dst, (int)hh), de_ = rb_readlane<uint32_t>(Den, (int)hh);
        c_h[q] = h;
        c_ds[q] = h < 64u ? ds_ : 0xFFFFFFFFu;
        c_de[q] = h < 64u ? de_ : 0xFFFFFFFFu;
    };
#pragma unroll
    for (int q = 0; q < RB_MS; q++) clip_fetch(q);

    auto step = [&](auto ring_c, auto edge_c, const uint32_t st) {
        constexpr int ring = decltype(ring_c)::value;
        constexpr bool edge = decltype(edge_c)::value;
        unsigned long long a0, a1, a2, a3;
#define RBT_RING_TAKE(A, B_, C_, D_)                                                                                            \
    asm volatile("s_waitcnt vmcnt(%4)\n\t"                                                                                      \
                 "v_mov_b64 %0, " RBT_RREG(A, 1) "\n\tv_mov_b64 %1, " RBT_RREG(B_, 1) "\n\tv_mov_b64 %2, " RBT_RREG(C_, 1) "\n\tv_mov_b64 %3, " RBT_RREG(D_, 1) \
                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3)                                                                       \
                 : "n"(RBT_RING_WAIT));
        RBT_RING_CASE(ring, RBT_RING_TAKE)
#undef RBT_RING_TAKE
        uint32_t w[8] = {(uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32), (uint32_t)a2, (uint32_t)(a2 >> 32), (uint32_t)a3, (uint32_t)(a3 >> 32)};
        uint32_t c[8];
#pragma unroll
        for (int q = 0; q < 8; q++) c[q] = w[q];
        if (edge) { // ops of the records in front of and behind the tile: zero for the sums, an alternating I / D of length 1 for the verification
            const int32_t idx0 = (int32_t)(st << 9) + lane * 8 - (int32_t)head;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const bool ok = (uint32_t)(idx0 + q) < n_tile;
                c[q] = ok ? w[q] : ((q & 1) ? 0x12u : 0x11u);
                w[q] = ok ? w[q] : 0u;
            }
        }
        if (has_gaps) { // gap ops: zero for the sums, an alternating I / D of length 1 for the verification, like the neighbours' ops at the edges
            const uint32_t gm = gq_s[(st << 6) + (uint32_t)lane];
            if (rb_ballot(gm != 0u) != 0ull) {
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const bool ok = !((gm >> q) & 1u);
                    c[q] = ok ? c[q] : ((q & 1) ? 0x12u : 0x11u);
                    w[q] = ok ? w[q] : 0u;
                }
            }
        }
        const uint32_t bqv = bq_s[(st << 6) + (uint32_t)lane]; // the op of my 8 at which a record starts (8: none)
        if (fused) {
            const uint32_t prevw = rb_prev_lane(c[7], v_carry);
            v_carry = rb_readlane<uint32_t>(c[7], 63);
            // v_reg collects the op codes seen as bits (1 << (word & 31): the code, and bit 4 = the length's lowest bit -- either half of
            // the word means the same code), checked against M I D N = X at the end; the op at which a record starts has nothing in front
            // of it to be equal to: its nibble of `poison` is ORed into its difference
            const uint32_t poison = bqv < 8u ? 15u << (4u * bqv) : 0u;
            uint32_t x[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                v_reg |= 1u << (c[q] & 31u);
                x[q] = ((c[q] ^ (q ? c[q - 1] : prevw)) & 15u) | __builtin_amdgcn_ubfe(poison, 4u * q, 4u);
            }
            auto min3 = [](uint32_t a, uint32_t b, uint32_t d) { const uint32_t t = a < b ? a : b; return t < d ? t : d; };
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                v_minw = min3(v_minw, c[q], c[q + 1]);
                v_adj = min3(v_adj, x[q], x[q + 1]);
            }
        }
        uint32_t sr = 0, sq = 0, su = 0;
        [[maybe_unused]] uint32_t rpre[8]; // break-paf: the reference bases of my ops in front of op q (the partial sums, kept)
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const uint32_t len = rb_len(w[q]);
            rpre[q] = sr;
            sr += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFFDFFFDu, w[q], 1u); // ref = not I
            sq += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFF3FFF3u, w[q], 1u); // query = not D, not N
            su += len;
        }
        const uint32_t ir = rb_wave_scan_incl(sr), iq = rb_wave_scan_incl(sq), iu = rb_wave_scan_incl(su);
        {
            const uint32_t t = (st << 6) + (uint32_t)lane;
            cpR[t] = Rb + ir - sr;
            cpQ[t] = Qb + iq - sq;
            cpU[t] = Ub + iu - su;
        }
        const uint32_t R0 = Rb;
        Rb += rb_readlane<uint32_t>(ir, 63);
        Qb += rb_readlane<uint32_t>(iq, 63);
        Ub += rb_readlane<uint32_t>(iu, 63);
        v_maxsu = v_maxsu > su ? v_maxsu : su;
        v_utot += rb_readlane<uint32_t>(iu, 63);
        if constexpr (BRK) {
            // events of this step in op order, wave-uniform: a record starts (the open piece of the record in front of it closes where
            // that record ends, the cut state begins anew), or an indel longer than brk_max cuts (liftover.rs:187-206)
            // (per lane a mask of the ops that are events; the wave then visits the event ops only: a step has one or two)
            uint32_t evm = bqv < 8u ? 1u << bqv : 0u;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t cq = w[q] & 15u;
                evm |= ((cq == RB_OP_I || cq == RB_OP_D) && rb_len(w[q]) > brk_max_) ? 1u << q : 0u;
            }
            unsigned long long cm = rb_ballot(evm != 0u);
            const bool had = cm != 0ull;
            const uint32_t lane_r0 = R0 + ir - sr;
            while (cm) {
                const int l = __builtin_ctzll(cm);
                cm &= cm - 1ull;
                const uint32_t r0l = rb_readlane<uint32_t>(lane_r0, l), ml = rb_readlane<uint32_t>(evm, l);
                const uint32_t bql = rb_readlane<uint32_t>(bqv, l);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (!(ml & (1u << q))) continue;
                    const uint32_t rx = r0l + rb_readlane<uint32_t>(rpre[q], l);
                    if (bql == (uint32_t)q) { // the next record starts here
                        brk_close(rx);
                        rcnt = rb_writelane((brk_cnt - brk_p0), cur, rcnt);
                        cur++;
                        brk_p0 = brk_cnt, brk_pre = rx;
                        rp0 = rb_writelane(brk_cnt, cur, rp0);
                        brk_open();
                    }
                    const uint32_t wq = rb_readlane<uint32_t>(w[q], l);
                    const uint32_t cq = wq & 15u, lq = rb_len(wq);
                    if ((cq == RB_OP_I || cq == RB_OP_D) && lq > brk_max_) {
                        brk_close(rx);
                        brk_pre = rx + (cq == RB_OP_I ? 0u : lq); // :203-206
                        brk_open();
                    }
                }
            }
            if (had) {
#pragma unroll
                for (int q = 0; q < RB_MS; q++) clip_fetch(q);
            }
        }
        unsigned long long msk[RB_MS];
#pragma unroll
        for (int q = 0; q < RB_MS; q++) msk[q] = 0ull;
        if (spec) {
            // chunks whose reference span reaches from a clip's first base to its last one are stored as they are (k_liftover.hip)
            const uint32_t cR = R0 + ir - sr, cE = R0 + ir;
#pragma unroll
            for (int q = 0; q < RB_MS; q++) {
                unsigned long long mk = 0ull;
                for (;;) {
                    mk |= rb_ballot(cR < c_de[q] && cE >= c_ds[q]);
                    if (c_de[q] > Rb || c_h[q] >= 64u) break; // the clip reaches past this step (or there is none)
                    cfin[q] |= 1ull << c_h[q];                 // it ends in this step: the class's next clip may begin in it
                    clip_fetch(q);
                }
                msk[q] = mk;
            }
        }
        {
            const uint32_t so = (st << 11) + lane_boff;
            const unsigned long long sv_st = sv_exec;
            unsigned long long v0 = sv_st, v1 = sv_st;
            if (edge) { // groups in front of the tile's first op and behind its last one are not this tile's to write
                v0 = rb_ballot(so + 16u > first_boff && so <= last_boff);
                v1 = rb_ballot(so + 32u > first_boff && so + 16u <= last_boff);
            }
#pragma unroll
            for (int q = 0; q < RB_MS; q++) {
                unsigned long long m0 = msk[q] & v0, m1 = msk[q] & v1;
                { // whole 64-byte granules (two lanes): fewer lines written in part
                    unsigned long long q4 = (m0 | m1);
                    q4 = (q4 | (q4 >> 1)) & 0x5555555555555555ull;
                    q4 |= q4 << 1;
                    const unsigned long long keep0 = m0 | ~msk[q], keep1 = m1 | ~msk[q]; // (what the edge filter took away stays away)
                    m0 = q4 & keep0;
                    m1 = q4 & keep1;
                }
                if (RBT_NOSTORE) m0 = m1 = 0ull;
                const uint32_t *sb = out_ops_ + slot_row0 + (uint64_t)q * slot_stride_;
#define RBT_RING_STORE(A, B_, C_, D_)                                                                                           \
    asm volatile("s_mov_b64 exec, %[m0]\n\t"                                                                                    \
                 "global_store_dwordx4 %[o], " RBT_RREG(A, 3) ", %[sb]\n\t"                                                     \
                 "s_mov_b64 exec, %[m1]\n\t"                                                                                    \
                 "global_store_dwordx4 %[o], " RBT_RREG(C_, 3) ", %[sb] offset:16\n\t"                                          \
                 "s_mov_b64 exec, %[sv]"                                                                                        \
                 :                                                                                                              \
                 : [m0] "s"(m0), [m1] "s"(m1), [o] "v"(so), [sb] "s"(sb), [sv] "s"(sv_st)                                       \
                 : "memory");
                RBT_RING_CASE(ring, RBT_RING_STORE)
#undef RBT_RING_STORE
            }
        }
        RBT_RING_LOAD(ring, st + RBT_PF)
    };
    for (uint32_t st0 = 0; st0 < n_steps; st0 += RBT_PF) {
        rb_static_for<RBT_PF>([&](auto ring_c) {
            const uint32_t st = st0 + (uint32_t)decltype(ring_c)::value;
            if (st == 0 || st + 1 >= n_steps) step(ring_c, std::true_type{}, st);
            else step(ring_c, std::false_type{}, st);
        });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef RBT_RING_LOAD
#undef RBT_RING_LOAD_ASM
#undef RBT_RING_NOSTORES
    if (RBT_STOP == 2) return;
    kq = rb_kp_here(kp);
    if constexpr (BRK) { // what lies behind the last long indel of the last record (liftover.rs:213-224)
        brk_close(Rb);
        rcnt = rb_writelane((brk_cnt - brk_p0), cur, rcnt);
        H = brk_cnt;
        if (brk_over || H > RBT_HITS || cur + 1u != nrec) {
            fallback();
            return;
        }
    }

    // ---- the verdict.  P_j as measured: the running totals in front of record j = the checkpoint of the chunk its first op lies in
    //      + the ops in front of it in that chunk (they belong to record j - 1: records are 8 ops and more) ----
    uint32_t PR = 0, PQ = 0, PU = 0;
    if (isrec && lane != 0) {
        const uint32_t cj = rel0 >> 3, bqj = rel0 & 7u;
        const uint4 *gq = reinterpret_cast<const uint4 *>(gbase0 + 8u * cj);
        const uint4 g0_ = gq[0], g1_ = gq[1];
        const uint32_t g[8] = {g0_.x, g0_.y, g0_.z, g0_.w, g1_.x, g1_.y, g1_.z, g1_.w};
        PR = cpR[cj], PQ = cpQ[cj], PU = cpU[cj];
        const uint32_t gmj = has_gaps ? (uint32_t)gq_s[cj] : 0u; // (the ops in front of it that lie in the gap count for nothing, as in the stream)
#pragma unroll
        for (int q = 0; q < 7; q++) {
            const uint32_t len = ((uint32_t)q < bqj && !((gmj >> q) & 1u)) ? rb_len(g[q]) : 0u;
            PR += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFFDFFFDu, g[q], 1u);
            PQ += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFF3FFF3u, g[q], 1u);
            PU += len;
        }
    }
    uint32_t totR, totQ, totU; // of record j
    {
        const bool last = (uint32_t)lane + 1u >= nrec;
        const uint32_t nR = (uint32_t)__shfl_down((int)PR, 1, 64), nQ = (uint32_t)__shfl_down((int)PQ, 1, 64), nU = (uint32_t)__shfl_down((int)PU, 1, 64);
        totR = (last ? Rb : nR) - PR, totQ = (last ? Qb : nQ) - PQ, totU = (last ? Ub : nU) - PU;
    }
    {
        // check_integrity (paf.rs:825-857) on every record of the tile and, with the fused scan, the conditions of the fast path over all of
        // its ops.  (v_maxsu: per-lane sums below 2^25 keep the 64-lane scans inside 32 bits; v_utot: the exact total of all lengths.)
        bool lane_bad = (v_maxsu >> 25) != 0u || (isrec && (totR != spanR || totQ != spanQ || PR != PassR));
        if (fused) lane_bad |= ((v_reg | (v_reg >> 16)) & 0xFFFFu & ~(uint32_t)RB_REGULAR_MASK) != 0u || v_minw < 16u || v_adj == 0u;
        if (rb_ballot(lane_bad) != 0ull || rb_first64(v_utot) > 0xFFFFFFFFull) {
            fallback();
            return;
        }
    }
    if (H == 0u && !fused) return;

    // ---- resolution, lane h = hit h: first the start boundary of every hit, then the end boundary ----
    const bool ish = (uint32_t)lane < H;
    const uint32_t hj = hmeta & 255u, jl = hmeta >> 8;
    const uint32_t h_rel0n = (uint32_t)__shfl((int)(rel0 | (jn << 16)), (int)hj, 64);
    const uint32_t h_rel0 = h_rel0n & 0xFFFFu, h_n = h_rel0n >> 16;
    const uint32_t hPR = (uint32_t)__shfl((int)PR, (int)hj, 64), hPQ = (uint32_t)__shfl((int)PQ, (int)hj, 64), hPU = (uint32_t)__shfl((int)PU, (int)hj, 64);
    const uint32_t hTR = (uint32_t)__shfl((int)totR, (int)hj, 64), hTQ = (uint32_t)__shfl((int)totQ, (int)hj, 64), hTU = (uint32_t)__shfl((int)totU, (int)hj, 64);
    const uint32_t *h_ops = gbase0 + h_rel0; // the hit's record: its first op
    const int policy_ = p.policy;
    auto resolve1 = [&](const uint32_t Dg, auto is_start_c) -> rb_bres {
        constexpr bool is_start = decltype(is_start_c)::value;
        rb_bres O;
        O.st = RB_S_UNRES;
        O.op = O.part = O.R = O.Q = O.U = 0;
        if (ish && !inside) {
            const uint32_t D = Dg - hPR; // relative to the record
            if (D == hTR) { // on the record's last base; the last op is match-type
                const uint32_t lv = h_ops[h_n - 1u];
                O.st = RB_S_OK, O.op = h_n - 1u;
                if (is_start) O.part = rb_part_pack(1u, lv), O.R = hTR - 1u, O.Q = hTQ - 1u, O.U = hTU - 1u;
                else O.part = rb_part_pack(rb_len(lv), lv), O.R = hTR, O.Q = hTQ, O.U = hTU;
            } else if (D < hTR) {
                const uint32_t cj = h_rel0 >> 3, ce = (h_rel0 + h_n - 1u) >> 3;
                uint32_t lo_t = cj, hi_t = ce + 1u; // last chunk of the record whose prefix (relative: zero at the record's first chunk) is <= D
                while (hi_t - lo_t > 1u) {
                    const uint32_t mid = (lo_t + hi_t) >> 1;
                    if (cpR[mid] - hPR <= D) lo_t = mid; else hi_t = mid;
                }
                const bool first = lo_t == cj;
                O = rb_resolve(h_ops, h_n, (int32_t)(8u * lo_t) - (int32_t)h_rel0, first ? 0u : cpR[lo_t] - hPR, first ? 0u : cpQ[lo_t] - hPQ,
                               first ? 0u : cpU[lo_t] - hPU, D, is_start, policy_);
            } else {
                O.st = RB_S_DEFER; // (cannot happen: D <= the record's reference span)
            }
        }
        return O;
    };
    const rb_bres A = resolve1(Dst, std::true_type{});
    const rb_bres B = resolve1(Den, std::false_type{});

    if (RBT_STOP == 3) {
        if (ish && (A.st | B.st) == 0x7777u) p.counters->overflow = 1; // (keeps the resolution alive)
        return;
    }
    // ---- rows, lane h ----
    const uint32_t hr = ra + hj; // the record
    uint32_t status = RB_ST_OK, out_n = 0, a_op = 0;
    uint64_t o_tst = 0, o_ten = 0, o_qst = 0, o_qen = 0;
    uint32_t o_nm = 0, o_al = 0;
    bool defer = false;
    uint64_t h0r = 0;
    uint32_t win = jl;
    const uint32_t rec_nm = fused ? hTR + hTQ - hTU : 0u, rec_al = fused ? hTU : 0u;
    if (ish) {
        const rb_job *jp = &p.jobs[slot0 + hj];
        const uint64_t jt_st = jp->t_st, jt_en = jp->t_en, jq_st = jp->q_st, jq_en = jp->q_en;
        const bool minus = (jp->flags & RB_JOB_MINUS) != 0u;
        h0r = jp->h0;
        if (!BRK && !explicit_w) win = p.w_orig[(uint64_t)jp->lo + jl];
        if (inside) {
            out_n = h_n;
            o_tst = jt_st, o_ten = jt_en, o_qst = jq_st, o_qen = jq_en;
            o_nm = rec_nm, o_al = rec_al;
            if (!fused) o_nm = p.norm[hr].nmatch, o_al = p.norm[hr].aln_len;
        } else if (A.st == RB_S_DEFER || B.st == RB_S_DEFER || A.st == RB_S_UNRES || B.st == RB_S_UNRES) {
            defer = true;
        } else if (A.st == RB_S_NONE || B.st == RB_S_NONE || A.U >= B.U) {
            status = RB_ST_NONE_INDEL; // liftover.rs:52-54
        } else {
            a_op = A.op;
            o_tst = jt_st + A.R; // liftover.rs:57-60, :77-82
            o_ten = jt_st + B.R;
            if (!minus) o_qst = jq_st + A.Q, o_qen = jq_st + B.Q;
            else o_qst = jq_en - B.Q, o_qen = jq_en - A.Q;
            o_al = B.U - A.U;
            o_nm = (B.R + B.Q - B.U) - (A.R + A.Q - A.U);
            out_n = B.op - A.op + 1u;
        }
    }
    if constexpr (BRK) { // a boundary only the general kernel resolves: the per-record kernel declines such a record as a whole -- it gets the tile
        if (rb_ballot(ish && defer) != 0ull) {
            fallback();
            return;
        }
    }
    // break-paf: rows go to scratch, a place for all pieces of the tile from one of the bump cursors; counts and places per record
    uint64_t brk_row0 = 0;
    if constexpr (BRK) {
        unsigned long long b0 = 0;
        const uint32_t ar = tile % p.brk_n_arena;
        if (lane == 0 && H) b0 = atomicAdd(&p.brk_cursor[(size_t)ar * 16u], (unsigned long long)H);
        b0 = rb_first64(b0);
        const bool shortfall = b0 + H > p.brk_arena_cap;
        if (isrec) p.hit_off[r] = rcnt, p.brk_off[r] = shortfall ? ~0ull : (uint64_t)ar * p.brk_arena_cap + b0 + rp0;
        if (shortfall) {
            if (lane == 0) p.counters->brk_scratch_short = 1;
            return;
        }
        brk_row0 = (uint64_t)ar * p.brk_arena_cap + b0;
    }
    if (fused && isrec && active) { // the record's row, completed (RB_LIFT_FUSED_SCAN)
        rb_norm_row *wn = &p.norm_w[r];
        const uint32_t fl = p.norm[r].flags;
        wn->nmatch = totR + totQ - totU;
        wn->aln_len = totU;
        wn->flags = (fl & RB_F_STRIPPED) | RB_F_REGULAR;
    }
    if (H == 0u) return;
    const bool emits = ish && !defer && status == RB_ST_OK && !desc_mode;
    const uint32_t e_first = h_rel0 + a_op; // coordinate (counted from g0) of the clip's first op
    const uint32_t e_cnt = emits ? out_n : 0u;
    const uint32_t eg_last = e_first + e_cnt - 1u;
    // a clip keeps its place in its slot when it begins behind the last op of every earlier clip of its record and class: the
    // speculative stores all carry the record's ops as they are, only the end words differ from clip to clip -- and every speculative
    // store of the tile is out before the first end word is
    const uint32_t cls = jl % ns1;
    const uint32_t lgp = e_cnt ? eg_last + 1u : 0u;
    uint32_t pm = lgp;
    for (uint32_t d = ns1; d < 64u; d <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)pm, d, 64);
        if (jl >= d) pm = pm > t ? pm : t;
    }
    uint32_t before = (uint32_t)__shfl_up((int)pm, ns1, 64);
    if (jl < ns1) before = 0u;
    const bool in_slot = spec && emits && e_cnt != 0u && e_first >= before;
    const bool copied = emits && !in_slot;
    const uint64_t my_off = (uint64_t)cls * slot_stride_ + slot_row0 + e_first;
    const uint64_t my_row = (BRK ? brk_row0 + (uint32_t)lane : h0r + jl);
    if (ish) {
        rb_hit_row *row = &p.rows[my_row];
        if (defer) {
            row->rec = hr;
            row->win = win;
            row->flags = RB_HIT_GENERIC;
            const unsigned long long g = atomicAdd((unsigned long long *)&p.counters->n_generic, 1ull);
            p.gen_list[g] = (uint32_t)my_row;
        } else {
            rb_hit_row wv;
            wv.rec = hr;
            wv.win = win;
            wv.status = (uint16_t)status;
            wv.flags = (inside ? RB_HIT_INSIDE : 0) | ((desc_mode && status == RB_ST_OK) ? RB_HIT_DESCRIPTOR : 0);
            wv.out_n = status == RB_ST_OK ? out_n : 0;
            wv.t_st = o_tst, wv.t_en = o_ten, wv.q_st = o_qst, wv.q_en = o_qen;
            wv.nmatch = o_nm, wv.aln_len = o_al;
            wv.out_off = status == RB_ST_OK ? (desc_mode ? 4ull * my_row : (in_slot ? my_off : 0ull)) : 0;
            *row = wv;
            if (desc_mode && status == RB_ST_OK)
                *reinterpret_cast<uint4 *>(out_ops_ + 4ull * my_row) = make_uint4(a_op, out_n, inside ? 0u : rb_part(A.part), inside ? 0u : rb_part(B.part));
        }
    }
    // the clip's first op keeps its tail, its last op its head: two words made from what the resolution left in registers
    if (in_slot && !inside) {
        uint32_t *__restrict__ dst = out_ops_ + (uint64_t)cls * slot_stride_ + slot_row0;
        if (e_cnt == 1u) {
            __builtin_nontemporal_store(((B.U - A.U) << 4) | (A.part >> 28), dst + e_first);
        } else {
            __builtin_nontemporal_store(rb_part_word(A.part), dst + e_first);
            __builtin_nontemporal_store(rb_part_word(B.part), dst + eg_last);
        }
    }
    { // clips without a place of their own: rb_k_copy_clips copies them
        const unsigned long long cm = rb_ballot(copied);
        if (cm) {
            unsigned long long c0 = 0;
            if (lane == 0) c0 = atomicAdd(p.copy_count, (unsigned long long)__popcll(cm));
            c0 = rb_first64(c0);
            if (copied) {
                const uint64_t at = c0 + (uint64_t)__popcll(cm & ((1ull << lane) - 1ull));
                p.copy_list[at] = make_uint4((uint32_t)my_row, a_op, inside ? 0u : rb_part(A.part), inside ? 0u : rb_part(B.part));
            }
        }
    }
#undef p
}

#define RBT_KERNEL(NAME, BRK)                                                                                                     \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RBT_WPE), amdgpu_num_vgpr(RBT_RING_BASE - RBT_SPILL_ROOM))) void NAME(rb_lift_params p_) { \
        (void)p_;                                                                                                                 \
        rb_tile_body<BRK>();                                                                                                      \
    }
RBT_KERNEL(rb_k_liftover_tile, false)
RBT_KERNEL(rb_k_liftover_tile_brk, true)

extern "C" hipError_t rb_launch_liftover_tiles(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0 || p->n_tiles == 0) return hipSuccess;
    const unsigned blocks = (unsigned)(((uint64_t)p->n_tiles + 3) / 4);
    if (p->brk_mode) hipLaunchKernelGGL(rb_k_liftover_tile_brk, dim3(blocks), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(rb_k_liftover_tile, dim3(blocks), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" uint32_t rb_tile_max_ops(void) { return RBT_OPS - 32u; }
extern "C" uint32_t rb_tile_max_records(void) { return RBT_REC; }
