// rb_lift.h -- parameters and device helpers of the liftover / break-paf clip kernels (k_liftover.hip; also included by
// capi.hip for the launch parameters): window search, per-boundary resolution, the per-pass window setup.
#pragma once
#include "rb_device.h"
#include <cstdlib>


#define RB_HMAX 32            // hits resolved per streaming pass of one record (lanes 0-31 starts, 32-63 ends)
#define RB_LDS_PER_HIT 6      // dwords of per-hit (start) state in LDS
#define RB_ARENA_STRIDE 16    // u64 words between arena cursors (128 B)
#ifndef RB_MS
#define RB_MS 2               // positional copies ("slots") of the batch the streaming kernel can emit clips into
#endif

// One clip job per schedule slot, 64 bytes, written by rb_k_make_jobs after the hit scan: everything a wave needs to
// start streaming its record arrives with one load instead of a chain of dependent ones (schedule -> record row ->
// hit offsets -> window bounds).
struct __attribute__((aligned(64))) rb_job {
    uint64_t rec0;  // global index of the record's first kept op
    uint32_t n;     // kept ops
    uint32_t r;     // record
    uint32_t h0;    // first hit row of the record (rows_cap < 2^32)
    uint32_t flags; // RB_JOB_*
    uint32_t nh;    // hits
    uint32_t lo;    // first overlapping window (grouped index) of a monotone window list
    uint64_t t_st, t_en, q_st, q_en; // normalised coordinates
};
enum { RB_JOB_VALID = 1, RB_JOB_REGULAR = 2, RB_JOB_MINUS = 4, RB_JOB_MONO = 8, RB_JOB_ROWS_OVERFLOW = 16 };

struct rb_lift_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint32_t *contig;
    const uint8_t *strand;
    const rb_norm_row *norm;
    // schedule
    const uint32_t *sched;     // [n_rec] record handled by wave w (longest first)
    const uint32_t *slot_of;   // [n_rec] inverse of sched: the wave that handles record r
    const uint32_t *canon_pos; // [n_rec] position of record r in canonical order
    // windows grouped by contig (BED order kept inside a contig) + original order
    const uint64_t *w_st, *w_en; // grouped
    const uint32_t *w_orig;      // grouped -> BED index
    const uint64_t *wo_st, *wo_en; // BED order
    const uint64_t *cw_off;    // [n_contig + 1]
    const uint8_t *cw_mono;    // [n_contig]
    uint32_t n_contig;
    // explicit per-hit windows (break-paf); NULL for BED windows
    const uint64_t *x_st, *x_en;
    // rows
    uint64_t *hit_off; // [n_rec + 1], canonical order; holds counts before the scan
    uint32_t *win_lo;  // [n_rec] first overlapping window of a monotone slice (grouped index), by record
    rb_hit_row *rows;
    uint64_t rows_cap;
    uint32_t *out_ops;
    uint64_t out_cap;
    // output arenas
    unsigned long long *arena_cur; // [n_arena * RB_ARENA_STRIDE]
    uint64_t arena_size;           // ops per arena (multiple of 4)
    uint32_t n_arena;
    // generic list
    uint32_t *gen_list; // [rows_cap]
    rb_counters *counters;
    int policy;
    int early_exit; // stop streaming a record once every boundary of the pass is resolved
    int desc_mode;  // RB_LIFT_DESCRIPTORS: 4-word clip descriptors at out_ops[4 * row] instead of copied ops
    uint64_t arena_origin; // first op of the arena area inside out_ops (descriptor mode: after the descriptors)
    int debug_skip; // diagnostics only (wrong results): 1 = no emission, 2 = no resolution, 4 = no streaming
    uint32_t wave0, wave_end; // slice of the schedule this launch covers (records are classed by length)
    rb_job *jobs;             // [n_rec], schedule order
    // RB_LIFT_FUSED_SCAN: norm[] holds provisional rows (rb_k_peek_norm); the clip kernel verifies every record while it
    // streams it, completes the row, and hands records it cannot vouch for back through pend_list for the full scan
    int fused;
    rb_norm_row *norm_w;           // == norm, writable
    uint32_t *pend_list;           // [n_rec]
    unsigned long long *pend_count;
    // positional output: slot k of out_ops mirrors the input (op at batch index g of record r -> out_ops[k * slot_stride + 4 r + g]);
    // the arena area behind the slots takes what the generic kernel and rb_k_copy_clips emit
    uint32_t n_slots;              // 0 .. RB_MS
    uint64_t slot_stride;          // ops per slot (a multiple of 4): pad4(n_ops) + 4 n_rec + 8
    uint64_t needed_base;          // ops the slots the plan asks for would take (what out_ops_needed reports in front of the arenas)
    uint4 *copy_list;              // [rows_cap] {row, first kept op, clipped first / last length}: clips that found no place in a slot
    unsigned long long *copy_count;
    // break-paf in ONE walk (RB_BREAK_ONE_WALK): the clip kernel finds the long indels itself while it streams a record, so the
    // pieces of a record are known only when it ends: rows go to a place taken from one of brk_n_arena bump cursors in `rows`
    // (a scratch array then), the counts to hit_off, the place to brk_off; after the scan of the counts rb_k_break_gather moves
    // the rows to rows_final in record order.  What this path does not take (irregular records, a boundary it cannot resolve) sets counters->redo_two_walk: the caller runs the call again without the flag.
    int brk_mode;                  // != 0: this is that call
    uint32_t brk_max;              // indels longer than this cut (0: every indel)
    uint32_t brk_n_arena;
    uint64_t brk_arena_cap;        // rows per arena
    unsigned long long *brk_cursor; // [brk_n_arena * 16]
    uint64_t *brk_off;             // [n_rec] first scratch row of the record; ~0: none (no room in the scratch rows, or declined)
    rb_hit_row *rows_final;
    // records the one-walk kernel declines (irregular CIGARs, a record its verification hands back, a boundary only the generic
    // kernel resolves): listed here, ONE BY ONE -- their pieces are then found by rb_k_break_pieces in list mode and clipped by the
    // generic kernel, everybody else's rows stay (round 2 redid the whole batch with two walks when one record declined)
    uint32_t *brk_decl_list;       // [n_rec]
    unsigned long long *brk_decl_count;
    // checkpoints of the records the generic kernel works on: before kept op RB_GCP * k of record r, at gen_cp[(op_off[r] + first_op)
    // / RB_GCP + r + k], the units / reference / query / match bases of the record so far (rb_k_generic_checkpoints): a hit starts its
    // walks at the checkpoint in front of its window instead of at the record's first op.  NULL: every walk starts at the first op.
    uint4 *gen_cp;
    // diagnostics build of the clip kernel only (debug_skip & 256): [n_rec] the 100 MHz clock (low 32 bits) at which schedule slot w's
    // wave was done with its record -- how long a launch runs on after most of its waves have retired
    uint32_t *diag_stamps;
    // short records (k_tile.hip): tile t = records [tile_first[t], tile_first[t + 1]) -- consecutive in memory, each of 8 .. short_max
    // ops, at most RBT_REC of them and RBT_OPS - 32 ops together -- streamed by ONE wave as if they were one record (bit 31 of
    // tile_first[t]: a run of records too small for a tile, handed to the per-record kernel as they are).  A tile the tile kernel does
    // not take (a record that is not regular or was stripped, too many hits, a record its verification does not pass, ...) lists its
    // records in fb_list; the per-record kernel then runs over that list (rb_k_liftover_stream_list).
    // the generic wave kernel's per-hit descriptors (round 5, rb_k_generic_jobs): what a wave needs about hit g in ONE trip instead of seven
    // (list entry -> row -> record's row and offsets -> windows -> checkpoint searches).  They live in workspace arrays that are free by the
    // time the generic kernel runs: gj_a = the scratch rows of break-paf's one walk (64 B per row), gj_b = the copy list (16 B), gj_c = the
    // piece windows' temporary (8 B).
    struct rb_gja *gj_a;
    uint4 *gj_b;                   // {row, record, window, status << 16 | row flags << 8 | has checkpoints << 1 | minus strand}
    uint2 *gj_c;                   // {units of the record (aln_len), checkpoint the walks start at}
    const uint32_t *tile_first;    // [3 n_tiles]: {first record | pass-through << 31, records, schedule slot of the first record (the others follow)}
    uint32_t n_tiles;
    uint32_t *fb_list;             // [n_rec]
    unsigned long long *fb_count;
    // RB_LIFT_OP_STARTS: op_off[r] is where record r starts, nothing more (op_off[r + 1] says nothing about its end): a batch that
    // trim-paf has cut in place.  Extents come from norm[] alone; the tile kernel streams over the gaps between a tile's records.
    int op_starts;
};
#define RB_GCP 64u
struct __attribute__((aligned(64))) rb_gja { // half A of a generic hit's descriptor
    uint64_t ops_off;                // the record's first kept op in ops[]
    uint64_t t_st, t_en, q_st, q_en; // normalised coordinates
    uint64_t wst, wen;               // the window
    uint32_t n;                      // kept ops
    uint32_t k2;                     // checkpoint in front of the units at the window's end (0: no jump)
};

// the kernel-argument segment of a kernel whose one argument is an rb_lift_params, read with scalar loads where a field is used
// (rb_k_liftover_stream); rb_kp_here makes a copy of the pointer the compiler cannot see through, so that a load through it is
// neither merged with the others nor moved to the kernel's entry
typedef const __attribute__((address_space(4))) rb_lift_params *rb_kparams;
__device__ __forceinline__ rb_kparams rb_kp_here(rb_kparams q) {
    asm volatile("" : "+s"(q));
    return q;
}

// ------------------------------------------------------------------------------------------------
// hit counting: paf_overlaps_rgn (paf.rs:622-627) on the NORMALISED record (trim_helper runs
// aligned_pairs, hence remove_trailing_indels, before the filter: liftover.rs:119-127)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t rb_lower_en_gt(const uint64_t *w_en, uint64_t lo, uint64_t hi, uint64_t t_st) {
    while (lo < hi) { // first idx with en > t_st (en non-decreasing)
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (w_en[mid] > t_st) hi = mid; else lo = mid + 1;
    }
    return lo;
}
__device__ __forceinline__ uint64_t rb_lower_st_ge(const uint64_t *w_st, uint64_t lo, uint64_t hi, uint64_t t_en) {
    while (lo < hi) { // first idx with st >= t_en (st non-decreasing)
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (w_st[mid] >= t_en) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------------
// streaming kernel
// ------------------------------------------------------------------------------------------------
enum { RB_S_UNRES = 0, RB_S_OK = 1, RB_S_NONE = 2, RB_S_DEFER = 3 };

// first/last set helpers on 64-bit masks
__device__ __forceinline__ int rb_ffs64(unsigned long long m) { return __ffsll((long long)m) - 1; }

// append every hit of a record to the generic list (record not eligible for the streaming path)
__device__ inline void rb_defer_record(rb_kparams kp, uint32_t r, const rb_norm_row *nr, uint64_t h0, uint64_t nh,
                                bool explicit_w, bool mono, uint64_t ws, uint64_t we, int lane) {
    // (an opaque copy of the lane id: this rare path is inlined into the clip kernel's pass loop, and its row addresses
    //  would otherwise be hoisted out of that loop and carried -- spilled -- through every record)
    asm volatile("" : "+v"(lane));
    const rb_kparams q_ = rb_kp_here(kp);
    struct { const uint64_t *w_st, *w_en; const uint32_t *w_orig; rb_hit_row *rows; uint64_t rows_cap; uint32_t *gen_list; rb_counters *counters; } p =
        {q_->w_st, q_->w_en, q_->w_orig, q_->rows, q_->rows_cap, q_->gen_list, q_->counters};
    if (explicit_w || mono) {
        uint64_t lo = 0;
        if (!explicit_w) lo = rb_lower_en_gt(p.w_en, ws, we, nr->t_st);
        for (uint64_t j = lane; j < nh; j += 64) {
            const uint64_t h = h0 + j;
            if (h < p.rows_cap) {
                rb_hit_row *row = &p.rows[h];
                row->rec = r;
                row->win = explicit_w ? (uint32_t)j : p.w_orig[lo + j];
                row->flags = RB_HIT_GENERIC;
                const unsigned long long g = atomicAdd((unsigned long long *)&p.counters->n_generic, 1ull);
                p.gen_list[g] = (uint32_t)h;
            }
        }
    } else {
        // non-monotone window list: enumerate in BED order, 64 windows per step
        uint64_t done = 0;
        for (uint64_t b = ws; b < we; b += 64) {
            const uint64_t i = b + lane;
            const bool hit = i < we && nr->t_en > p.w_st[i] && nr->t_st < p.w_en[i];
            const unsigned long long ball = __ballot(hit);
            if (hit) {
                const uint64_t j = done + __popcll(ball & ((1ull << lane) - 1ull));
                const uint64_t h = h0 + j;
                if (h < p.rows_cap) {
                    rb_hit_row *row = &p.rows[h];
                    row->rec = r;
                    row->win = p.w_orig[i];
                    row->flags = RB_HIT_GENERIC;
                    const unsigned long long g = atomicAdd((unsigned long long *)&p.counters->n_generic, 1ull);
                    p.gen_list[g] = (uint32_t)h;
                }
            }
            done += __popcll(ball);
        }
    }
}

// ---- lane-local boundary resolution --------------------------------------------------------------
// One lane resolves one window boundary.  It starts from a checkpoint (exclusive prefixes R,Q,U at an
// op index that is a multiple of 16, written to LDS by the streaming pass), walks at most 16 ops held
// in registers to the reference-consuming op that contains offset D, then applies the reference's
// tpos_to_idx + walk-to-match rules (paf.rs:541-561) with short look-ahead / look-back loads.
struct rb_bres {
    uint32_t st;            // RB_S_OK / NONE / DEFER
    uint32_t op, part;      // op index; start: ops' remaining length (len - off), end: used length (off + 1) -- with the op's code in
                            // the top four bits (a length has 28): rb_part() / rb_part_word() below
    uint32_t R, Q, U;       // start: exclusive counts at the unit; end: inclusive counts
};

// rb_bres.part: the clipped length of the boundary op and the op's code; rb_part_word = the op as the clip holds it
__device__ __forceinline__ uint32_t rb_part_pack(uint32_t part, uint32_t op_word) { return part | (op_word << 28); }
__device__ __forceinline__ uint32_t rb_part(uint32_t packed) { return packed & 0x0FFFFFFFu; }
__device__ __forceinline__ uint32_t rb_part_word(uint32_t packed) { return (packed << 4) | (packed >> 28); }

#define RB_WALK_MAX 24

// regular records only (M I D N = X): ref = not I, query = not D and not N
__device__ __forceinline__ uint32_t rb_rl(uint32_t v) { return rb_opc(v) == RB_OP_I ? 0u : rb_len(v); }
__device__ __forceinline__ uint32_t rb_ql(uint32_t v) { return (rb_opc(v) == RB_OP_D || rb_opc(v) == RB_OP_N) ? 0u : rb_len(v); }
__device__ __forceinline__ bool rb_ism(uint32_t v) { return rb_in(RB_MATCH_MASK, rb_opc(v)); }

// ops[] = the record's kept ops, n of them.  (cR,cQ,cU) = prefixes at op index cidx (checkpoint).
// D in [cR, next checkpoint's R) and D < Rtot.  is_start selects search-right (true) / search-left.
#ifndef RB_CP_OPS
#define RB_CP_OPS 8 // ops between two checkpoints of the streaming kernel (16: every second lane leaves one; 8: every lane)
#endif
// Round 3: the search inside the checkpoint group is branch-free.  With the exclusive prefixes R_k of the reference lengths,
// "R_{k+1} <= D" is a monotone predicate p_k over the group's ops (ops in front of or behind the record count as M of length 0), so
// the op f that holds offset D is the first one with p_k false, its index the number of true ones, and the prefixes at f are the
// inclusive prefixes of the last op with p_k true: one conditional move each.  No index comparisons, no six-way moves under an
// exec mask per op (the round-2 form: 27 vector instructions per op, 430 per group).  The op in front of
// the group is loaded with the group, not after the search has shown that it is needed.
__device__ __forceinline__ rb_bres rb_resolve(const uint32_t *__restrict__ ops, uint32_t n, int32_t cidx, uint32_t cR, uint32_t cQ,
                                              uint32_t cU, uint32_t D, bool is_start, int policy) {
    rb_bres o;
    o.st = RB_S_DEFER;
    o.op = o.part = o.R = o.Q = o.U = 0;
    uint32_t g[RB_CP_OPS];
    uint32_t pv;
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(ops + cidx); // 16-byte aligned by construction
        const uint4 a0 = q[0], a1 = q[1];
        g[0] = a0.x; g[1] = a0.y; g[2] = a0.z; g[3] = a0.w; g[4] = a1.x; g[5] = a1.y; g[6] = a1.z; g[7] = a1.w;
#if RB_CP_OPS == 16
        const uint4 a2 = q[2], a3 = q[3];
        g[8] = a2.x; g[9] = a2.y; g[10] = a2.z; g[11] = a2.w; g[12] = a3.x; g[13] = a3.y; g[14] = a3.z; g[15] = a3.w;
#endif
        pv = ops[cidx > 0 ? cidx - 1 : 0]; // the op in front of the group (used only where the group's first op is f and f > 0)
    }
    uint32_t np = 0, fR = cR, fQ = cQ, fU = cU, fv = 0;
    {
        uint32_t R = cR, Q = cQ, U = cU;
        bool before = true; // p_{k-1}: every op so far ends at or in front of D
#pragma unroll
        for (int k = 0; k < RB_CP_OPS; k++) {
            const uint32_t v = (uint32_t)(cidx + k) < n ? g[k] : 0u; // (also the negative indices of the aligned head)
            const uint32_t len = rb_len(v);
            R += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFFDFFFDu, v, 1u); // regular records: ref = not I
            Q += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFF3FFF3u, v, 1u); //                  query = not D, not N
            U += len;
            const bool pk = R <= D;
            fv = (before && !pk) ? v : fv; // the first op that reaches past D
            np += pk ? 1u : 0u;
            fR = pk ? R : fR; // (R, Q, U do not decrease and p is monotone: the last assignment is the prefix at f)
            fQ = pk ? Q : fQ;
            fU = pk ? U : fU;
            pv = pk ? v : pv;
            before = pk;
        }
    }
    if (np >= (uint32_t)RB_CP_OPS) return o; // should not happen; the generic kernel sorts it out
    const int32_t fi = cidx + (int32_t)np;
    const uint32_t off = D - fR;
    if (is_start) {
        int32_t X; // first match-type op with index >= X
        if (off > 0) { // the boundary base and the next base share op f
            if (rb_ism(fv)) {
                o.st = RB_S_OK, o.op = (uint32_t)fi, o.part = rb_part_pack(rb_len(fv) - (off - 1), fv), o.R = fR + off - 1, o.Q = fQ + off - 1, o.U = fU + off - 1;
                return o;
            }
            X = fi + 1;
        } else { // boundary base is the last unit before op f: the last equal element is the unit before f
            if (fi > 0 && rb_ism(pv)) {
                o.st = RB_S_OK, o.op = (uint32_t)(fi - 1), o.part = rb_part_pack(1u, pv), o.R = fR - 1, o.Q = fQ - 1, o.U = fU - 1;
                return o;
            }
            // duplicates in tpos_aln (units of an insertion share the boundary's tpos): which one
            // binary_search returns depends on the Rust std generation -> generic kernel decides
            if (policy == RB_BSEARCH_LEGACY && fi > 0 && rb_opc(pv) == RB_OP_I) return o;
            X = fi;
        }
        // walk right (paf.rs:551-553) from op fi
        uint32_t R = fR, Q = fQ, U = fU;
        uint32_t v = fv;
        int32_t i = fi;
#pragma nounroll // (unrolled 24-fold, each level of the nest parks an exec mask in scalar registers: 100 spills)
        for (int t = 0; t < RB_WALK_MAX; t++) {
            if (i >= X && rb_ism(v)) {
                o.st = RB_S_OK, o.op = (uint32_t)i, o.part = rb_part_pack(rb_len(v), v), o.R = R, o.Q = Q, o.U = U;
                return o;
            }
            R += rb_rl(v);
            Q += rb_ql(v);
            U += rb_len(v);
            i++;
            if ((uint32_t)i >= n) {
                o.st = RB_S_NONE; // ran off the end: start_idx == N (liftover.rs:52)
                return o;
            }
            v = ops[i];
        }
        return o; // too far: generic
    } else {
        int32_t Y; // last match-type op with index <= Y
        if (off > 0) {
            if (rb_ism(fv)) {
                o.st = RB_S_OK, o.op = (uint32_t)fi, o.part = rb_part_pack(off, fv), o.R = D, o.Q = fQ + off, o.U = fU + off;
                return o;
            }
            Y = fi - 1;
        } else {
            if (fi > 0 && rb_ism(pv)) {
                o.st = RB_S_OK, o.op = (uint32_t)(fi - 1), o.part = rb_part_pack(rb_len(pv), pv), o.R = fR, o.Q = fQ, o.U = fU;
                return o;
            }
            Y = fi - 2;
        }
        // walk left (paf.rs:555-557): (R,Q,U) are the prefixes at the END of op i
        uint32_t R = fR, Q = fQ, U = fU;
        int32_t i = fi - 1;
#pragma nounroll // (unrolled 24-fold, each level of the nest parks an exec mask in scalar registers: 100 spills)
        for (int t = 0; t < RB_WALK_MAX; t++) {
            if (i < 0) {
                o.st = RB_S_NONE; // stops at unit 0, which lies before any start
                return o;
            }
            const uint32_t v = ops[i];
            if (i <= Y && rb_ism(v)) {
                o.st = RB_S_OK, o.op = (uint32_t)i, o.part = rb_part_pack(rb_len(v), v), o.R = R, o.Q = Q, o.U = U;
                return o;
            }
            R -= rb_rl(v);
            Q -= rb_ql(v);
            U -= rb_len(v);
            i--;
        }
        return o;
    }
}


// ---- windows of one pass ---------------------------------------------------------------------------
// Lanes j and j + 32 both receive window jb + j of the record (lane j resolves its start boundary, lane
// j + 32 its end boundary).  Monotone window lists and explicit (break-paf) windows are indexed directly;
// windows of a contig that are not sorted are collected in BED order, 64 candidates per ballot, with the
// same test as rb_k_count_hits so the counts agree.  widx = RB_HMAX words of LDS scratch of this wave.
struct rb_pass_win {
    uint64_t wst, wen;
    uint32_t win;
};
__device__ __forceinline__ rb_pass_win rb_pass_windows(rb_kparams kp, uint32_t *widx, bool explicit_w, bool mono, uint64_t ws,
                                                       uint64_t we, uint64_t lo, uint64_t h0, uint64_t jb, uint32_t nb, uint64_t t_st,
                                                       uint64_t t_en, uint64_t &scan_pos, int lane) {
    rb_pass_win o;
    o.wst = o.wen = 0;
    o.win = 0;
    asm volatile("" : "+v"(lane)); // (opaque: keeps lo + lane from being hoisted out of the pass loop and spilled)
    const rb_kparams q_ = rb_kp_here(kp);
    struct { const uint64_t *w_st, *w_en, *x_st, *x_en; const uint32_t *w_orig; } p = {q_->w_st, q_->w_en, q_->x_st, q_->x_en, q_->w_orig};
    const uint32_t hl = (uint32_t)lane & 31u;
    const bool own = hl < nb;
    if (!explicit_w && !mono) {
        uint32_t filled = 0;
        while (filled < nb && scan_pos < we) {
            const uint64_t i = scan_pos + (uint64_t)lane;
            const bool hit = i < we && t_en > p.w_st[i] && t_st < p.w_en[i];
            const unsigned long long ball = __ballot(hit);
            const uint32_t room = nb - filled, cnt = (uint32_t)__popcll(ball);
            const uint32_t rank = (uint32_t)__popcll(ball & ((1ull << lane) - 1ull));
            if (hit && rank < room) widx[filled + rank] = (uint32_t)(i - ws);
            if (cnt <= room) {
                filled += cnt;
                scan_pos += 64;
            } else { // the pass is full: resume after the room-th hit next time
                unsigned long long m = ball;
                for (uint32_t q = 1; q < room; q++) m &= m - 1;
                scan_pos += (uint64_t)rb_ffs64(m) + 1u;
                filled += room;
            }
        }
        if (own) {
            const uint64_t idx = ws + widx[hl];
            o.wst = p.w_st[idx];
            o.wen = p.w_en[idx];
            o.win = p.w_orig[idx];
        }
    } else if (own) {
        if (explicit_w) {
            o.wst = p.x_st[h0 + jb + hl];
            o.wen = p.x_en[h0 + jb + hl];
            o.win = (uint32_t)(jb + hl);
        } else {
            o.wst = p.w_st[lo + jb + hl];
            o.wen = p.w_en[lo + jb + hl];
            o.win = p.w_orig[lo + jb + hl];
        }
    }
    return o;
}
