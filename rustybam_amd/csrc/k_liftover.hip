// k_liftover.hip -- liftover / break-paf clip kernels for gfx950 (wave64, CDNA4).
//
// Replaces liftover::trim_helper + trim_paf_rec_to_rgn (liftover.rs:17-132) and everything they
// call (aligned_pairs paf.rs:501-538, tpos_to_idx_match :541-561, subset_cigar /
// collapse_long_cigar :593-620).  The reference expands every CIGAR to per-base arrays (24 B per
// aligned base) and binary-searches them; here the walk stays in op space:
//
//   rb_k_count_hits     one thread per record: number of overlapping windows (paf.rs:622-627)
//   rb_k_scan_*         exclusive scan of the counts -> first row of every record (canonical order)
//   rb_k_liftover_stream  ONE WAVEFRONT PER RECORD.  The record's packed ops stream from HBM once
//                       (16 B per lane, 1 KiB per wave instruction); a 6-step DPP prefix scan
//                       gives the running (ref, query, unit) offsets of every op in registers; window
//                       boundaries are resolved on the fly with a ballot + readlane, up to 64 windows
//                       per pass, their state kept in LDS.  The clipped CIGARs are then copied out
//                       of L2 (the record has just been streamed) into space the wave reserves with
//                       one atomic per record.
//   rb_k_liftover_generic  one thread per hit, serial walk: every case the streaming kernel declines
//                       (irregular CIGARs: N/S/H/P, zero lengths, adjacent ops of one type that must
//                       merge (paf.rs:602-620); non-monotone window lists; the legacy binary-search
//                       policy when the duplicate choice matters; lookbacks across a 256-op step).
//
// Roofline: HBM.  Algorithmic bytes: 4 B per input op + 48 B per record + 88 B per hit + 4 B per
// emitted op.  No MFMA: integer / index work only.
#include "rb_device.h"

#define RB_HMAX 64            // hits resolved per streaming pass of one record
#define RB_LDS_PER_HIT 12     // dwords of per-hit state in LDS
#define RB_ARENA_STRIDE 16    // u64 words between arena cursors (128 B)

struct rb_lift_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint32_t *contig;
    const uint8_t *strand;
    const rb_norm_row *norm;
    // schedule
    const uint32_t *sched;     // [n_rec] record handled by wave w (longest first)
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
};

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

__global__ __launch_bounds__(256) void rb_k_count_hits(rb_lift_params p) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n_rec) return;
    uint64_t cnt = 0;
    const rb_norm_row *nr = &p.norm[r];
    const uint32_t c = p.contig[r];
    if (nr->status == RB_ST_OK && c < p.n_contig) {
        const uint64_t ws = p.cw_off[c], we = p.cw_off[c + 1];
        const uint64_t t_st = nr->t_st, t_en = nr->t_en;
        if (p.cw_mono[c]) {
            const uint64_t lo = rb_lower_en_gt(p.w_en, ws, we, t_st);
            const uint64_t hi = rb_lower_st_ge(p.w_st, ws, we, t_en);
            cnt = hi > lo ? hi - lo : 0;
            p.win_lo[r] = (uint32_t)lo;
        } else {
            for (uint64_t i = ws; i < we; i++) cnt += (t_en > p.w_st[i] && t_st < p.w_en[i]) ? 1 : 0;
        }
    }
    p.hit_off[p.canon_pos[r]] = cnt;
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of u64 counts, in place, n + 1 outputs (3 small launches)
// ------------------------------------------------------------------------------------------------
#define RB_SCAN_PER_BLOCK 2048
__global__ __launch_bounds__(256) void rb_k_scan_partial(const uint64_t *v, uint64_t n, uint64_t *block_sums) {
    __shared__ uint64_t sh[4];
    const uint64_t base = (uint64_t)blockIdx.x * RB_SCAN_PER_BLOCK;
    uint64_t s = 0;
    for (int k = 0; k < RB_SCAN_PER_BLOCK / 256; k++) {
        uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
        if (i < n) s += v[i];
    }
    s = rb_wave_sum_u64(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void rb_k_scan_top(uint64_t *block_sums, uint64_t n_blocks) {
    // single block: serial over chunks of 256 (n_blocks is a few thousand at most)
    __shared__ uint64_t sh[256];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t b = 0; b < n_blocks; b += 256) {
        uint64_t i = b + threadIdx.x;
        uint64_t x = i < n_blocks ? block_sums[i] : 0;
        sh[threadIdx.x] = x;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            uint64_t y = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
            __syncthreads();
            sh[threadIdx.x] += y;
            __syncthreads();
        }
        uint64_t incl = sh[threadIdx.x];
        if (i < n_blocks) block_sums[i] = carry + incl - x;
        __syncthreads();
        if (threadIdx.x == 255) carry += incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[n_blocks] = carry;
}
__global__ __launch_bounds__(256) void rb_k_scan_apply(uint64_t *v, uint64_t n, const uint64_t *block_sums, rb_counters *counters) {
    __shared__ uint64_t sh[256];
    const uint64_t base = (uint64_t)blockIdx.x * RB_SCAN_PER_BLOCK;
    // each thread owns 8 consecutive elements
    uint64_t x[RB_SCAN_PER_BLOCK / 256];
    uint64_t s = 0;
    const uint64_t i0 = base + (uint64_t)threadIdx.x * (RB_SCAN_PER_BLOCK / 256);
#pragma unroll
    for (int k = 0; k < RB_SCAN_PER_BLOCK / 256; k++) {
        x[k] = (i0 + k < n) ? v[i0 + k] : 0;
        s += x[k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        uint64_t y = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += y;
        __syncthreads();
    }
    uint64_t run = block_sums[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int k = 0; k < RB_SCAN_PER_BLOCK / 256; k++) {
        if (i0 + k < n) v[i0 + k] = run;
        run += x[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) {
        const uint64_t total = block_sums[gridDim.x];
        v[n] = total;
        counters->n_hits = total;
    }
}

// ------------------------------------------------------------------------------------------------
// streaming kernel
// ------------------------------------------------------------------------------------------------
enum { RB_S_UNRES = 0, RB_S_OK = 1, RB_S_NONE = 2, RB_S_DEFER = 3 };
// LDS slots per hit
enum { H_AOP = 0, H_AFIRST = 1, H_RA = 2, H_QA = 3, H_UA = 4, H_BOP = 5, H_BLAST = 6, H_NRB = 7, H_NQB = 8, H_NUB = 9, H_AST = 10, H_BST = 11 };

struct rb_step { // one 256-op step, per lane: 4 ops with exclusive prefixes (record relative)
    uint32_t opc[4], len[4];
    uint32_t Rx[4], Qx[4], Ux[4];
    int32_t idx0; // record-relative op index of slot 0 (may be < 0 in the aligned head)
};

// first/last set helpers on 64-bit masks
__device__ __forceinline__ int rb_ffs64(unsigned long long m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ int rb_fls64(unsigned long long m) { return 63 - __clzll((long long)m); }

// select slot value by (per-lane) slot index
__device__ __forceinline__ uint32_t rb_sel(const uint32_t v[4], int s) {
    return s == 0 ? v[0] : (s == 1 ? v[1] : (s == 2 ? v[2] : v[3]));
}

struct rb_found { // wave-uniform description of one op
    int32_t idx;  // record-relative op index
    uint32_t opc, len, Rx, Qx, Ux;
    uint32_t prev_opc, prev_len; // op idx-1 (RB_NULL_OP if none)
    bool ok;
};

// op that contains reference offset D (the ref-consuming op with Rx <= D < Rx + len)
__device__ __forceinline__ rb_found rb_find_ref(const rb_step &s, uint32_t D, uint32_t carry_tail) {
    bool m[4];
#pragma unroll
    for (int r = 0; r < 4; r++) m[r] = rb_in(RB_REF_MASK, s.opc[r]) && (uint32_t)(D - s.Rx[r]) < s.len[r];
    const bool any = m[0] || m[1] || m[2] || m[3];
    const unsigned long long ball = __ballot(any);
    rb_found f;
    f.ok = ball != 0;
    if (!f.ok) return f;
    const int L = rb_ffs64(ball);
    const int sl = m[0] ? 0 : (m[1] ? 1 : (m[2] ? 2 : 3));
    // previous op: slot sl-1 of this lane, or slot 3 of the previous lane (lane 0: carry)
    const uint32_t pl3 = rb_prev_lane((s.len[3] << 4) | s.opc[3], carry_tail);
    const uint32_t prevp = sl == 0 ? pl3 : ((rb_sel(s.len, sl - 1) << 4) | rb_sel(s.opc, sl - 1));
    f.idx = rb_readlane<int>(s.idx0 + sl, L);
    f.opc = rb_readlane<uint32_t>(rb_sel(s.opc, sl), L);
    f.len = rb_readlane<uint32_t>(rb_sel(s.len, sl), L);
    f.Rx = rb_readlane<uint32_t>(rb_sel(s.Rx, sl), L);
    f.Qx = rb_readlane<uint32_t>(rb_sel(s.Qx, sl), L);
    f.Ux = rb_readlane<uint32_t>(rb_sel(s.Ux, sl), L);
    const uint32_t pp = rb_readlane<uint32_t>(prevp, L);
    f.prev_opc = f.idx > 0 ? rb_opc(pp) : RB_NULL_OP;
    f.prev_len = rb_len(pp);
    return f;
}

// first match-type op with record index >= X inside this step
__device__ __forceinline__ rb_found rb_find_match_fwd(const rb_step &s, int32_t X) {
    bool m[4];
#pragma unroll
    for (int r = 0; r < 4; r++) m[r] = rb_in(RB_MATCH_MASK, s.opc[r]) && (s.idx0 + r) >= X;
    const bool any = m[0] || m[1] || m[2] || m[3];
    const unsigned long long ball = __ballot(any);
    rb_found f;
    f.ok = ball != 0;
    if (!f.ok) return f;
    const int L = rb_ffs64(ball);
    const int sl = m[0] ? 0 : (m[1] ? 1 : (m[2] ? 2 : 3));
    f.idx = rb_readlane<int>(s.idx0 + sl, L);
    f.opc = rb_readlane<uint32_t>(rb_sel(s.opc, sl), L);
    f.len = rb_readlane<uint32_t>(rb_sel(s.len, sl), L);
    f.Rx = rb_readlane<uint32_t>(rb_sel(s.Rx, sl), L);
    f.Qx = rb_readlane<uint32_t>(rb_sel(s.Qx, sl), L);
    f.Ux = rb_readlane<uint32_t>(rb_sel(s.Ux, sl), L);
    f.prev_opc = RB_NULL_OP;
    f.prev_len = 0;
    return f;
}

// last match-type op with record index <= Y inside this step
__device__ __forceinline__ rb_found rb_find_match_bwd(const rb_step &s, int32_t Y) {
    bool m[4];
#pragma unroll
    for (int r = 0; r < 4; r++) m[r] = rb_in(RB_MATCH_MASK, s.opc[r]) && (s.idx0 + r) <= Y;
    const bool any = m[0] || m[1] || m[2] || m[3];
    const unsigned long long ball = __ballot(any);
    rb_found f;
    f.ok = ball != 0;
    if (!f.ok) return f;
    const int L = rb_fls64(ball);
    const int sl = m[3] ? 3 : (m[2] ? 2 : (m[1] ? 1 : 0));
    f.idx = rb_readlane<int>(s.idx0 + sl, L);
    f.opc = rb_readlane<uint32_t>(rb_sel(s.opc, sl), L);
    f.len = rb_readlane<uint32_t>(rb_sel(s.len, sl), L);
    f.Rx = rb_readlane<uint32_t>(rb_sel(s.Rx, sl), L);
    f.Qx = rb_readlane<uint32_t>(rb_sel(s.Qx, sl), L);
    f.Ux = rb_readlane<uint32_t>(rb_sel(s.Ux, sl), L);
    f.prev_opc = RB_NULL_OP;
    f.prev_len = 0;
    return f;
}

__device__ __forceinline__ void rb_set_start(uint32_t *h, int lane, uint32_t aop, uint32_t afirst, uint32_t Ra, uint32_t Qa, uint32_t Ua, uint32_t st) {
    if (lane == 0) {
        h[H_AOP] = aop;
        h[H_AFIRST] = afirst;
        h[H_RA] = Ra;
        h[H_QA] = Qa;
        h[H_UA] = Ua;
        h[H_AST] = st;
    }
}
__device__ __forceinline__ void rb_set_end(uint32_t *h, int lane, uint32_t bop, uint32_t blast, uint32_t nR, uint32_t nQ, uint32_t nU, uint32_t st) {
    if (lane == 0) {
        h[H_BOP] = bop;
        h[H_BLAST] = blast;
        h[H_NRB] = nR;
        h[H_NQB] = nQ;
        h[H_NUB] = nU;
        h[H_BST] = st;
    }
}

// append every hit of a record to the generic list (record not eligible for the streaming path)
__device__ void rb_defer_record(const rb_lift_params &p, uint32_t r, const rb_norm_row *nr, uint64_t h0, uint64_t nh,
                                bool explicit_w, bool mono, uint64_t ws, uint64_t we, int lane) {
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

__global__ __launch_bounds__(256) void rb_k_liftover_stream(rb_lift_params p) {
    __shared__ uint32_t lds_all[4][RB_HMAX * RB_LDS_PER_HIT];
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wave >= p.n_rec) return;
    const int lane = rb_lane();
    uint32_t *lds = lds_all[threadIdx.x >> 6];
    const uint32_t r = rb_first(p.sched[wave]);
    const rb_norm_row *nr = &p.norm[r];
    if (nr->status != RB_ST_OK) return;
    const uint64_t k = p.canon_pos[r];
    const uint64_t h0 = rb_first64(p.hit_off[k]);
    const uint64_t nh = rb_first64(p.hit_off[k + 1]) - h0;
    if (nh == 0) return;
    if (h0 + nh > p.rows_cap) { // rows do not fit: flag and leave (host retries with more room)
        if (lane == 0) p.counters->overflow = 1;
        return;
    }
    const bool explicit_w = p.x_st != nullptr;
    const uint32_t c = p.contig[r];
    uint64_t ws = 0, we = 0;
    bool mono = true;
    if (!explicit_w) {
        ws = p.cw_off[c];
        we = p.cw_off[c + 1];
        mono = p.cw_mono[c] != 0;
    }
    const bool fast = (nr->flags & RB_F_REGULAR) && (explicit_w || mono);
    if (!fast) {
        rb_defer_record(p, r, nr, h0, nh, explicit_w, mono, ws, we, lane);
        return;
    }
    const uint64_t t_st = nr->t_st, t_en = nr->t_en, q_st = nr->q_st, q_en = nr->q_en;
    const uint32_t n = nr->n_ops;
    const bool minus = p.strand[r] == (uint8_t)'-';
    const uint64_t rec0 = p.op_off[r] + nr->first_op; // global index of the record's first kept op
    const uint32_t *rec_ops = p.ops + rec0;
    const uint64_t lo = explicit_w ? 0 : p.win_lo[r];
    const uint32_t arena = (uint32_t)(wave % p.n_arena);

    for (uint64_t jb = 0; jb < nh; jb += RB_HMAX) {
        const uint32_t nb = (uint32_t)((nh - jb) < RB_HMAX ? (nh - jb) : RB_HMAX);
        // ---- per-hit setup: lane j owns hit jb + j ----
        uint64_t wst = 0, wen = 0;
        uint32_t win = 0;
        bool mine = (uint32_t)lane < nb;
        if (mine) {
            if (explicit_w) {
                wst = p.x_st[h0 + jb + lane];
                wen = p.x_en[h0 + jb + lane];
                win = (uint32_t)(jb + lane);
            } else {
                wst = p.w_st[lo + jb + lane];
                wen = p.w_en[lo + jb + lane];
                win = p.w_orig[lo + jb + lane];
            }
        }
        const bool inside = mine && (t_st > wst && t_en < wen); // liftover.rs:23-25
        // D = (relative ref offset of the boundary base) + 1
        const uint32_t Ds = (uint32_t)((wst > t_st ? wst : t_st) - t_st) + 1u; // liftover.rs:28
        const uint32_t De = (uint32_t)((wen < t_en ? wen : t_en) - t_st);       // (min(en,t_en) - 1 - t_st) + 1, :38-40
        for (int i = lane; i < (int)(nb * RB_LDS_PER_HIT); i += 64) lds[i] = 0;
        unsigned long long need_s = __ballot(mine && !inside);
        unsigned long long need_e = need_s;
        uint32_t nextDs = need_s ? rb_readlane<uint32_t>(Ds, rb_ffs64(need_s)) : 0xFFFFFFFFu;
        uint32_t nextDe = need_e ? rb_readlane<uint32_t>(De, rb_ffs64(need_e)) : 0xFFFFFFFFu;
        unsigned long long pend = 0; // starts waiting for the next match-type op

        // ---- stream the record ----
        uint32_t Rb = 0, Qb = 0, Ub = 0; // running totals before this step
        uint32_t carry_tail = (RB_NULL_OP); // packed last op of the previous step
        uint32_t tail_len = 0;
        if (need_s | need_e) {
            const uint64_t g0 = rec0 & ~3ull;
            const uint64_t gend = rec0 + n;
            const uint64_t n_steps = (gend - g0 + 255u) >> 8;
            // 4 steps (4 KiB per wave) of loads stay in flight ahead of the step being processed
            const uint64_t glane = g0 + (uint64_t)lane * 4u;
            auto load_step = [&](uint64_t stp) -> uint4 {
                const uint64_t gi = glane + (stp << 8);
                return gi < gend ? *reinterpret_cast<const uint4 *>(p.ops + gi) : make_uint4(0, 0, 0, 0);
            };
            uint4 pf0 = load_step(0), pf1 = load_step(1), pf2 = load_step(2), pf3 = load_step(3);
            for (uint64_t st = 0; st < n_steps; st++) {
                const uint64_t gi = glane + (st << 8);
                const uint4 cur = pf0;
                pf0 = pf1;
                pf1 = pf2;
                pf2 = pf3;
                pf3 = load_step(st + 4);
                rb_step s;
                s.idx0 = (int32_t)((int64_t)gi - (int64_t)rec0);
                const uint32_t raw[4] = {cur.x, cur.y, cur.z, cur.w};
                uint32_t rl[4], ql[4];
                uint32_t sr = 0, sq = 0, su = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool valid = (uint32_t)(s.idx0 + q) < n; // also rejects negative indices
                    s.opc[q] = valid ? rb_opc(raw[q]) : RB_NULL_OP;
                    s.len[q] = valid ? rb_len(raw[q]) : 0u;
                    // regular records: only M I D = X, so ref = not I, query = not D
                    rl[q] = (s.opc[q] == RB_OP_I) ? 0u : s.len[q];
                    ql[q] = (s.opc[q] == RB_OP_D) ? 0u : s.len[q];
                    s.Rx[q] = sr;
                    s.Qx[q] = sq;
                    s.Ux[q] = su;
                    sr += rl[q];
                    sq += ql[q];
                    su += s.len[q];
                }
                const uint32_t ir = rb_wave_scan_incl(sr), iq = rb_wave_scan_incl(sq), iu = rb_wave_scan_incl(su);
                const uint32_t er = Rb + ir - sr, eq = Qb + iq - sq, eu = Ub + iu - su;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    s.Rx[q] += er;
                    s.Qx[q] += eq;
                    s.Ux[q] += eu;
                }
                const uint32_t Rend = Rb + rb_readlane<uint32_t>(ir, 63);
                const uint32_t Qend = Qb + rb_readlane<uint32_t>(iq, 63);
                const uint32_t Uend = Ub + rb_readlane<uint32_t>(iu, 63);

                // (1) starts waiting for the first match-type op at or after this step
                if (pend) {
                    rb_found f = rb_find_match_fwd(s, 0); // every op of this step lies after the boundary
                    if (f.ok) {
                        while (pend) {
                            const int j = rb_ffs64(pend);
                            pend &= pend - 1;
                            rb_set_start(lds + j * RB_LDS_PER_HIT, lane, (uint32_t)f.idx, f.len, f.Rx, f.Qx, f.Ux, RB_S_OK);
                        }
                    }
                }
                // (2) window starts whose boundary op lies in this step (liftover.rs:29, search right)
                while (nextDs < Rend) {
                    const int j = rb_ffs64(need_s);
                    need_s &= need_s - 1;
                    uint32_t *h = lds + j * RB_LDS_PER_HIT;
                    rb_found f = rb_find_ref(s, nextDs, carry_tail);
                    const uint32_t off = nextDs - f.Rx;
                    bool want_fwd = false;
                    int32_t X = 0;
                    if (off > 0) { // the boundary base and the next base share op f
                        if (rb_in(RB_MATCH_MASK, f.opc)) {
                            rb_set_start(h, lane, (uint32_t)f.idx, f.len - (off - 1), f.Rx + off - 1, f.Qx + off - 1, f.Ux + off - 1, RB_S_OK);
                        } else {
                            want_fwd = true;
                            X = f.idx + 1;
                        }
                    } else { // boundary base is the last unit before op f: last equal element is the unit before f
                        if (rb_in(RB_MATCH_MASK, f.prev_opc)) {
                            rb_set_start(h, lane, (uint32_t)(f.idx - 1), 1u, f.Rx - 1, f.Qx - 1, f.Ux - 1, RB_S_OK);
                        } else if (p.policy == RB_BSEARCH_LEGACY && !rb_in(RB_REF_MASK, f.prev_opc)) {
                            // duplicates in tpos_aln: which one binary_search returns is version dependent
                            rb_set_start(h, lane, 0, 0, 0, 0, 0, RB_S_DEFER);
                        } else {
                            want_fwd = true;
                            X = f.idx;
                        }
                    }
                    if (want_fwd) {
                        rb_found g = rb_find_match_fwd(s, X);
                        if (g.ok)
                            rb_set_start(h, lane, (uint32_t)g.idx, g.len, g.Rx, g.Qx, g.Ux, RB_S_OK);
                        else
                            pend |= 1ull << j;
                    }
                    nextDs = need_s ? rb_readlane<uint32_t>(Ds, rb_ffs64(need_s)) : 0xFFFFFFFFu;
                }
                // (3) window ends (liftover.rs:40, search left)
                while (nextDe < Rend) {
                    const int j = rb_ffs64(need_e);
                    need_e &= need_e - 1;
                    uint32_t *h = lds + j * RB_LDS_PER_HIT;
                    rb_found f = rb_find_ref(s, nextDe, carry_tail);
                    const uint32_t off = nextDe - f.Rx;
                    bool want_bwd = false;
                    int32_t Y = 0;
                    if (off > 0) {
                        if (rb_in(RB_MATCH_MASK, f.opc))
                            rb_set_end(h, lane, (uint32_t)f.idx, off, nextDe, f.Qx + off, f.Ux + off, RB_S_OK);
                        else {
                            want_bwd = true;
                            Y = f.idx - 1;
                        }
                    } else {
                        if (rb_in(RB_MATCH_MASK, f.prev_opc))
                            rb_set_end(h, lane, (uint32_t)(f.idx - 1), f.prev_len, f.Rx, f.Qx, f.Ux, RB_S_OK);
                        else {
                            want_bwd = true;
                            Y = f.idx - 2;
                        }
                    }
                    if (want_bwd) {
                        rb_found g = rb_find_match_bwd(s, Y);
                        if (g.ok) {
                            rb_set_end(h, lane, (uint32_t)g.idx, g.len, g.Rx + g.len, g.Qx + g.len, g.Ux + g.len, RB_S_OK);
                        } else {
                            const int32_t first_idx = rb_readlane<int>(s.idx0, 0);
                            if (first_idx <= 0) {
                                // nothing before: walk-left stops at unit 0 (paf.rs:556), which is before any start
                                rb_set_end(h, lane, 0, 0, 0, 0, 0, RB_S_NONE);
                            } else {
                                // the op we want sits in an earlier step: walk back over the (L2-resident)
                                // ops just streamed; Rb/Qb/Ub are the prefixes at the end of op first_idx-1
                                int32_t jb2 = first_idx - 1;
                                uint32_t wr = Rb, wq = Qb, wu = Ub, st_end = RB_S_DEFER;
                                for (int t = 0; t < 32 && jb2 >= 0; t++, jb2--) {
                                    const uint32_t v = rec_ops[jb2];
                                    const uint32_t vo = rb_opc(v), vl = rb_len(v);
                                    if (jb2 <= Y && rb_in(RB_MATCH_MASK, vo)) {
                                        rb_set_end(h, lane, (uint32_t)jb2, vl, wr, wq, wu, RB_S_OK);
                                        st_end = RB_S_OK;
                                        break;
                                    }
                                    wr -= (vo == RB_OP_I) ? 0u : vl;
                                    wq -= (vo == RB_OP_D) ? 0u : vl;
                                    wu -= vl;
                                }
                                if (st_end != RB_S_OK) rb_set_end(h, lane, 0, 0, 0, 0, 0, jb2 < 0 ? RB_S_NONE : RB_S_DEFER);
                            }
                        }
                    }
                    nextDe = need_e ? rb_readlane<uint32_t>(De, rb_ffs64(need_e)) : 0xFFFFFFFFu;
                }
                // carry to the next step
                {
                    const uint32_t packed3 = (s.len[3] << 4) | s.opc[3];
                    // last VALID op of the step: lane 63 slot 3 unless the record ends inside the step
                    const int32_t last_idx = (int32_t)n - 1;
                    const int32_t rel = last_idx - rb_readlane<int>(s.idx0, 0);
                    if (rel >= 255) {
                        carry_tail = rb_readlane<uint32_t>(packed3, 63);
                    } else {
                        const int L = rel >> 2, sl = rel & 3;
                        const uint32_t pk[4] = {(s.len[0] << 4) | s.opc[0], (s.len[1] << 4) | s.opc[1], (s.len[2] << 4) | s.opc[2], packed3};
                        carry_tail = rb_readlane<uint32_t>(rb_sel(pk, sl), L);
                    }
                    tail_len = rb_len(carry_tail);
                }
                Rb = Rend;
                Qb = Qend;
                Ub = Uend;
                // optional: nothing downstream depends on the rest of the record once every boundary
                // of the pass is resolved (boundaries on the last base are by construction pending)
                if (p.early_exit && !(need_s | need_e | pend)) break;
            }
            // ---- boundaries on the record's last base (D == total ref bases); last op is match-type ----
            while (need_s) {
                const int j = rb_ffs64(need_s);
                need_s &= need_s - 1;
                rb_set_start(lds + j * RB_LDS_PER_HIT, lane, n - 1, 1u, Rb - 1, Qb - 1, Ub - 1, RB_S_OK);
            }
            while (need_e) {
                const int j = rb_ffs64(need_e);
                need_e &= need_e - 1;
                rb_set_end(lds + j * RB_LDS_PER_HIT, lane, n - 1, tail_len, Rb, Qb, Ub, RB_S_OK);
            }
            while (pend) { // no match-type op after the boundary: start_idx == N > end_idx
                const int j = rb_ffs64(pend);
                pend &= pend - 1;
                rb_set_start(lds + j * RB_LDS_PER_HIT, lane, 0, 0, 0, 0, 0, RB_S_NONE);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): LDS writes of lane 0 visible to the wave
        __builtin_amdgcn_wave_barrier();

        // ---- finalize: lane j computes the row of hit jb + j ----
        uint32_t status = RB_ST_OK, out_n = 0, a_op = 0, a_first = 0, b_op = 0, b_last = 0;
        uint64_t o_tst = 0, o_ten = 0, o_qst = 0, o_qen = 0;
        uint32_t o_nm = 0, o_al = 0;
        bool defer = false;
        if (mine) {
            if (inside) {
                out_n = n;
                o_tst = t_st;
                o_ten = t_en;
                o_qst = q_st;
                o_qen = q_en;
                o_nm = nr->nmatch;
                o_al = nr->aln_len;
                a_op = 0;
                b_op = n - 1;
            } else {
                const uint32_t *h = lds + lane * RB_LDS_PER_HIT;
                const uint32_t as = h[H_AST], bs = h[H_BST];
                if (as == RB_S_DEFER || bs == RB_S_DEFER || as == RB_S_UNRES || bs == RB_S_UNRES) {
                    defer = true;
                } else if (as == RB_S_NONE || bs == RB_S_NONE || h[H_UA] >= h[H_NUB]) {
                    status = RB_ST_NONE_INDEL; // liftover.rs:52-54
                } else {
                    a_op = h[H_AOP];
                    a_first = h[H_AFIRST];
                    b_op = h[H_BOP];
                    b_last = h[H_BLAST];
                    const uint32_t Ra = h[H_RA], Qa = h[H_QA], Ua = h[H_UA];
                    const uint32_t nR = h[H_NRB], nQ = h[H_NQB], nU = h[H_NUB];
                    o_tst = t_st + Ra; // liftover.rs:57-60, :77-82
                    o_ten = t_st + nR;
                    if (!minus) {
                        o_qst = q_st + Qa;
                        o_qen = q_st + nQ;
                    } else {
                        o_qst = q_en - nQ;
                        o_qen = q_en - Qa;
                    }
                    o_al = nU - Ua;
                    o_nm = (nR + nQ - nU) - (Ra + Qa - Ua); // match units = ref + query - all (M I D = X only)
                    out_n = b_op - a_op + 1;
                }
            }
        }
        // space for the clipped cigars: one atomic per pass, each hit padded to 4 ops
        const uint32_t padded = (mine && !defer && status == RB_ST_OK) ? ((out_n + 3u) & ~3u) : 0u;
        const uint32_t incl = rb_wave_scan_incl(padded);
        const uint32_t total = rb_readlane<uint32_t>(incl, 63);
        uint64_t base = 0;
        if (total) {
            unsigned long long b0 = 0;
            if (lane == 0) b0 = atomicAdd(&p.arena_cur[(uint64_t)arena * RB_ARENA_STRIDE], (unsigned long long)total);
            base = rb_first64(b0);
        }
        const bool fits = base + total <= p.arena_size;
        if (!fits && lane == 0) p.counters->overflow = 1;
        const uint64_t my_off = (uint64_t)arena * p.arena_size + base + (incl - padded);
        if (mine) {
            rb_hit_row *row = &p.rows[h0 + jb + lane];
            if (defer) {
                row->rec = r;
                row->win = win;
                row->flags = RB_HIT_GENERIC;
                const unsigned long long g = atomicAdd((unsigned long long *)&p.counters->n_generic, 1ull);
                p.gen_list[g] = (uint32_t)(h0 + jb + lane);
            } else {
                rb_hit_row w;
                w.rec = r;
                w.win = win;
                w.status = (uint16_t)status;
                w.flags = inside ? RB_HIT_INSIDE : 0;
                w.out_n = status == RB_ST_OK ? out_n : 0;
                w.t_st = o_tst;
                w.t_en = o_ten;
                w.q_st = o_qst;
                w.q_en = o_qen;
                w.nmatch = o_nm;
                w.aln_len = o_al;
                w.out_off = status == RB_ST_OK ? my_off : 0;
                *row = w;
            }
        }
        // ---- emit: copy ops[a_op .. b_op] (L2-resident), patch the two clipped ends ----
        if (fits) {
            unsigned long long todo = __ballot(padded != 0);
            const unsigned long long inside_mask = __ballot(inside);
            while (todo) {
                const int j = rb_ffs64(todo);
                todo &= todo - 1;
                const uint32_t e_aop = rb_readlane<uint32_t>(a_op, j);
                const uint32_t e_n = rb_readlane<uint32_t>(out_n, j);
                const uint32_t e_afirst = rb_readlane<uint32_t>(a_first, j);
                const uint32_t e_blast = rb_readlane<uint32_t>(b_last, j);
                const bool e_inside = (inside_mask >> j) & 1ull;
                const uint64_t e_off = rb_readlane<uint64_t>(my_off, j);
                const uint32_t *__restrict__ src = rec_ops + e_aop;
                uint32_t *__restrict__ dst = p.out_ops + e_off;
                // 1024 ops per round: the 4 loads of a lane are issued back to back (the ops array is
                // padded, so reading up to 3 ops past the clip is safe; they are zeroed below)
                for (uint32_t i0 = 0; i0 < e_n; i0 += 1024u) {
                    uint4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t i = i0 + (uint32_t)u * 256u + (uint32_t)lane * 4u;
                        v[u] = i < e_n ? rb_load4_unaligned(src + i) : make_uint4(0, 0, 0, 0);
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const uint32_t i = i0 + (uint32_t)u * 256u + (uint32_t)lane * 4u;
                        if (i < e_n) {
                            if (i + 1 >= e_n) v[u].y = 0u;
                            if (i + 2 >= e_n) v[u].z = 0u;
                            if (i + 3 >= e_n) v[u].w = 0u;
                            if (!e_inside) {
                                if (i == 0) { // first op keeps its tail, or the middle if the clip is a single op
                                    const uint32_t l0 = e_n == 1 ? (e_afirst + e_blast - rb_len(v[u].x)) : e_afirst;
                                    v[u].x = (l0 << 4) | rb_opc(v[u].x);
                                }
                                if (e_n > 1 && e_n - 1 - i < 4u) { // last op keeps its head
                                    const uint32_t q = e_n - 1 - i;
                                    const uint32_t lastv = q == 0 ? v[u].x : (q == 1 ? v[u].y : (q == 2 ? v[u].z : v[u].w));
                                    const uint32_t nv = (e_blast << 4) | rb_opc(lastv);
                                    if (q == 0) v[u].x = nv; else if (q == 1) v[u].y = nv; else if (q == 2) v[u].z = nv; else v[u].w = nv;
                                }
                            }
                            *reinterpret_cast<uint4 *>(dst + i) = v[u];
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// generic kernel: one thread per hit, serial, fully general (unit semantics evaluated in op space)
// ------------------------------------------------------------------------------------------------
struct rb_gwalk {
    const uint32_t *ops;
    uint32_t n;
};

// legacy Rust binary_search (1.52..1.81) on a virtual array whose equal range is [klo, khi]
__device__ uint64_t rb_legacy_probe(uint64_t N, uint64_t klo, uint64_t khi) {
    uint64_t size = N, left = 0, right = N;
    while (left < right) {
        const uint64_t mid = left + size / 2;
        if (mid < klo)
            left = mid + 1;
        else if (mid > khi)
            right = mid;
        else
            return mid;
        size = right - left;
    }
    return klo;
}

__global__ __launch_bounds__(256) void rb_k_liftover_generic(rb_lift_params p) {
    const uint64_t n_gen = p.counters->n_generic;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_gen; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t hrow = p.gen_list[g];
        rb_hit_row *row = &p.rows[hrow];
        const uint32_t r = row->rec, win = row->win;
        const rb_norm_row *nr = &p.norm[r];
        const uint64_t t_st = nr->t_st, t_en = nr->t_en, q_st = nr->q_st, q_en = nr->q_en;
        const bool minus = p.strand[r] == (uint8_t)'-';
        const uint32_t n = nr->n_ops;
        const uint32_t *ops = p.ops + p.op_off[r] + nr->first_op;
        const uint64_t wst = p.x_st ? p.x_st[hrow] : p.wo_st[win];
        const uint64_t wen = p.x_en ? p.x_en[hrow] : p.wo_en[win];
        rb_hit_row w;
        w.rec = r;
        w.win = win;
        w.flags = RB_HIT_GENERIC;
        w.status = RB_ST_OK;
        w.out_n = 0;
        w.out_off = 0;
        w.t_st = w.t_en = w.q_st = w.q_en = 0;
        w.nmatch = w.aln_len = 0;
        const uint32_t arena = (uint32_t)(g % p.n_arena);

        if (t_st > wst && t_en < wen) { // liftover.rs:23-25: verbatim clone, own id
            w.flags |= RB_HIT_INSIDE;
            w.t_st = t_st;
            w.t_en = t_en;
            w.q_st = q_st;
            w.q_en = q_en;
            w.nmatch = nr->nmatch;
            w.aln_len = nr->aln_len;
            w.out_n = n;
            const uint32_t padded = (n + 3u) & ~3u;
            const unsigned long long b0 = atomicAdd(&p.arena_cur[(uint64_t)arena * RB_ARENA_STRIDE], (unsigned long long)padded);
            if (b0 + padded <= p.arena_size) {
                w.out_off = (uint64_t)arena * p.arena_size + b0;
                for (uint32_t i = 0; i < n; i++) p.out_ops[w.out_off + i] = ops[i];
            } else {
                p.counters->overflow = 1;
            }
            *row = w;
            continue;
        }
        // positions to look up (liftover.rs:28, :38-40)
        const int64_t ps = (int64_t)(wst > t_st ? wst : t_st);
        const int64_t pe = (int64_t)(wen < t_en ? wen : t_en) - 1;
        // pass 1: equal ranges of ps and pe in the virtual tpos_aln, total units (paf.rs:505-534)
        uint64_t N = 0;
        uint64_t s_lo = 0, s_hi = 0, e_lo = 0, e_hi = 0;
        bool s_found = false, e_found = false;
        {
            int64_t tpos = (int64_t)t_st - 1;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t opc = rb_opc(ops[i]), len = rb_len(ops[i]);
                if (len == 0) continue;
                if (opc <= 8 && rb_in(RB_REF_MASK, opc)) {
                    // units N..N+len-1 hold tpos+1 .. tpos+len
                    if (ps > tpos && ps <= tpos + (int64_t)len) {
                        const uint64_t u = N + (uint64_t)(ps - tpos - 1);
                        if (!s_found) { s_found = true; s_lo = u; }
                        s_hi = u;
                    }
                    if (pe > tpos && pe <= tpos + (int64_t)len) {
                        const uint64_t u = N + (uint64_t)(pe - tpos - 1);
                        if (!e_found) { e_found = true; e_lo = u; }
                        e_hi = u;
                    }
                    tpos += len;
                } else {
                    if (ps == tpos && tpos >= 0) {
                        if (!s_found) { s_found = true; s_lo = N; }
                        s_hi = N + len - 1;
                    }
                    if (pe == tpos && tpos >= 0) {
                        if (!e_found) { e_found = true; e_lo = N; }
                        e_hi = N + len - 1;
                    }
                }
                N += len;
            }
        }
        if (!s_found || !e_found) { // binary_search Err -> panic (liftover.rs:31, :42)
            w.status = RB_ST_PANIC_NOTFOUND;
            *row = w;
            continue;
        }
        const uint64_t ks = p.policy == RB_BSEARCH_LEGACY ? rb_legacy_probe(N, s_lo, s_hi) : s_hi;
        const uint64_t ke = p.policy == RB_BSEARCH_LEGACY ? rb_legacy_probe(N, e_lo, e_hi) : e_hi;
        // pass 2: a = first match-type unit >= ks (else N); b = last match-type unit <= ke (else 0)
        uint64_t a = N, b = 0;
        uint64_t Ra = 0, Qa = 0, Ma = 0, nRb = 0, nQb = 0, nMb = 0;
        {
            uint64_t U = 0, R = 0, Q = 0, M = 0;
            bool a_set = false;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t opc = rb_opc(ops[i]), len = rb_len(ops[i]);
                if (len == 0) continue;
                const bool isref = opc <= 8 && rb_in(RB_REF_MASK, opc), isq = opc <= 8 && rb_in(RB_QRY_MASK, opc);
                const bool ism = opc <= 8 && rb_in(RB_MATCH_MASK, opc);
                if (ism) {
                    if (!a_set && U + len > ks) {
                        a = ks > U ? ks : U;
                        const uint64_t off = a - U;
                        Ra = R + off;
                        Qa = Q + off;
                        Ma = M + off;
                        a_set = true;
                    }
                    if (U <= ke) {
                        b = (U + len - 1) < ke ? (U + len - 1) : ke;
                        const uint64_t off = b - U;
                        nRb = R + off + 1;
                        nQb = Q + off + 1;
                        nMb = M + off + 1;
                    }
                }
                U += len;
                if (isref) R += len;
                if (isq) Q += len;
                if (ism) M += len;
            }
        }
        if (a > b || a >= N) { // liftover.rs:52-54
            w.status = RB_ST_NONE_INDEL;
            *row = w;
            continue;
        }
        // pass 3: count run-length-merged ops of units [a, b] (paf.rs:602-620)
        uint32_t out_n = 0;
        {
            uint64_t U = 0;
            uint32_t prev = RB_NULL_OP;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t opc = rb_opc(ops[i]), len = rb_len(ops[i]);
                if (len == 0) continue;
                const uint64_t u0 = U, u1 = U + len - 1;
                U += len;
                if (u1 < a) continue;
                if (u0 > b) break;
                if (opc != prev) out_n++;
                prev = opc;
            }
        }
        w.t_st = t_st + Ra; // liftover.rs:57-60, :77-82 (a and b are match-type units)
        w.t_en = t_st + nRb;
        if (!minus) {
            w.q_st = q_st + Qa;
            w.q_en = q_st + nQb;
        } else {
            w.q_st = q_en - nQb;
            w.q_en = q_en - Qa;
        }
        w.nmatch = (uint32_t)(nMb - Ma);
        w.aln_len = (uint32_t)(b - a + 1);
        w.out_n = out_n;
        const uint32_t padded = (out_n + 3u) & ~3u;
        const unsigned long long b0 = atomicAdd(&p.arena_cur[(uint64_t)arena * RB_ARENA_STRIDE], (unsigned long long)padded);
        if (b0 + padded > p.arena_size) {
            p.counters->overflow = 1;
            *row = w;
            continue;
        }
        w.out_off = (uint64_t)arena * p.arena_size + b0;
        { // pass 4: emit
            uint64_t U = 0;
            uint32_t prev = RB_NULL_OP, run = 0;
            uint64_t o = w.out_off;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t opc = rb_opc(ops[i]), len = rb_len(ops[i]);
                if (len == 0) continue;
                const uint64_t u0 = U, u1 = U + len - 1;
                U += len;
                if (u1 < a) continue;
                if (u0 > b) break;
                const uint64_t c0 = u0 > a ? u0 : a, c1 = u1 < b ? u1 : b;
                const uint32_t piece = (uint32_t)(c1 - c0 + 1);
                if (opc != prev) {
                    if (prev != RB_NULL_OP) p.out_ops[o++] = (run << 4) | prev;
                    prev = opc;
                    run = piece;
                } else {
                    run += piece;
                }
            }
            if (prev != RB_NULL_OP) p.out_ops[o++] = (run << 4) | prev;
        }
        *row = w;
    }
}

// out_ops_used / out_ops_needed from the arena cursors
__global__ __launch_bounds__(64) void rb_k_finish(rb_lift_params p) {
    unsigned long long mx = 0, sum = 0;
    for (uint32_t a = threadIdx.x; a < p.n_arena; a += 64) {
        const unsigned long long c = p.arena_cur[(uint64_t)a * RB_ARENA_STRIDE];
        mx = c > mx ? c : mx;
        sum += c;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(mx, off, 64);
        mx = o > mx ? o : mx;
        sum += __shfl_xor(sum, off, 64);
    }
    if (threadIdx.x != 0) return;
    p.counters->out_ops_used = sum;
    p.counters->out_ops_needed = (mx + 3ull) / 4ull * 4ull * p.n_arena;
    if (p.counters->n_hits > p.rows_cap) p.counters->overflow = 1;
}

extern "C" hipError_t rb_launch_count_and_scan(const rb_lift_params *p, uint64_t *block_sums, bool do_count, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    if (do_count) {
        const unsigned blocks = (unsigned)((p->n_rec + 255) / 256);
        hipLaunchKernelGGL(rb_k_count_hits, dim3(blocks), dim3(256), 0, stream, *p);
    }
    const uint64_t nb = (p->n_rec + RB_SCAN_PER_BLOCK - 1) / RB_SCAN_PER_BLOCK;
    hipLaunchKernelGGL(rb_k_scan_partial, dim3((unsigned)nb), dim3(256), 0, stream, (const uint64_t *)p->hit_off, p->n_rec, block_sums);
    hipLaunchKernelGGL(rb_k_scan_top, dim3(1), dim3(256), 0, stream, block_sums, nb);
    hipLaunchKernelGGL(rb_k_scan_apply, dim3((unsigned)nb), dim3(256), 0, stream, p->hit_off, p->n_rec, (const uint64_t *)block_sums, p->counters);
    return hipGetLastError();
}

extern "C" hipError_t rb_launch_liftover_stream(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((p->n_rec + 3) / 4);
    hipLaunchKernelGGL(rb_k_liftover_stream, dim3(blocks), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t rb_launch_liftover_tail(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_liftover_generic, dim3(1024), dim3(256), 0, stream, *p);
    hipLaunchKernelGGL(rb_k_finish, dim3(1), dim3(64), 0, stream, *p);
    return hipGetLastError();
}

extern "C" size_t rb_scan_block_sums_count(uint64_t n_rec) { return (size_t)((n_rec + RB_SCAN_PER_BLOCK - 1) / RB_SCAN_PER_BLOCK + 2); }
