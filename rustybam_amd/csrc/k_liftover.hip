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
#include <cstdlib>

#define RB_HMAX 32            // hits resolved per streaming pass of one record (lanes 0-31 starts, 32-63 ends)
#define RB_LDS_PER_HIT 6      // dwords of per-hit (start) state in LDS
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
    int desc_mode;  // RB_LIFT_DESCRIPTORS: 4-word clip descriptors at out_ops[4 * row] instead of copied ops
    uint64_t arena_origin; // first op of the arena area inside out_ops (descriptor mode: after the descriptors)
    int debug_skip; // diagnostics only (wrong results): 1 = no emission, 2 = no resolution, 4 = no streaming
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

// first/last set helpers on 64-bit masks
__device__ __forceinline__ int rb_ffs64(unsigned long long m) { return __ffsll((long long)m) - 1; }

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

// ---- emission helpers: one lane moves 4 ops (16 B) of its team's clip --------------------------------
// i = op position inside the clip (multiple of 4), e_n = ops in the clip.  The ops array is padded, so
// the load may read up to 3 ops past the clip; they are zeroed before the store.
__device__ __forceinline__ uint4 rb_emit_load(const uint32_t *src, uint32_t i, uint32_t e_n, int dbg = 0) {
    if (dbg & 16) return make_uint4(i, 0, 0, 0);
    return i < e_n ? rb_load4_unaligned(src + i) : make_uint4(0, 0, 0, 0);
}
__device__ __forceinline__ void rb_emit_store(uint32_t *dst, uint32_t i, uint32_t e_n, uint32_t afirst, uint32_t blast, bool verbatim, uint4 v, int dbg = 0) {
    if (i >= e_n) return;
    if (dbg & 8) { if (v.x == 0xFFFFFFF1u) dst[0] = v.y; return; }
    if (i + 1 >= e_n) v.y = 0u;
    if (i + 2 >= e_n) v.z = 0u;
    if (i + 3 >= e_n) v.w = 0u;
    if (!verbatim) {
        if (i == 0) { // first op keeps its tail, or the middle if the clip is a single op
            const uint32_t l0 = e_n == 1 ? (afirst + blast - rb_len(v.x)) : afirst;
            v.x = (l0 << 4) | rb_opc(v.x);
        }
        if (e_n > 1 && e_n - 1 - i < 4u) { // last op keeps its head
            const uint32_t q = e_n - 1 - i;
            const uint32_t lastv = q == 0 ? v.x : (q == 1 ? v.y : (q == 2 ? v.z : v.w));
            const uint32_t nv = (blast << 4) | rb_opc(lastv);
            if (q == 0) v.x = nv; else if (q == 1) v.y = nv; else if (q == 2) v.z = nv; else v.w = nv;
        }
    }
    // streaming (non-temporal) store: the clipped cigars are written once and never re-read here, so they
    // should not displace the record's ops from L2
    typedef uint32_t rb_u32x4 __attribute__((ext_vector_type(4)));
    rb_u32x4 nv4 = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(nv4, reinterpret_cast<rb_u32x4 *>(dst + i));
}

// ---- lane-local boundary resolution --------------------------------------------------------------
// One lane resolves one window boundary.  It starts from a checkpoint (exclusive prefixes R,Q,U at an
// op index that is a multiple of 16, written to LDS by the streaming pass), walks at most 16 ops held
// in registers to the reference-consuming op that contains offset D, then applies the reference's
// tpos_to_idx + walk-to-match rules (paf.rs:541-561) with short look-ahead / look-back loads.
struct rb_bres {
    uint32_t st;            // RB_S_OK / NONE / DEFER
    uint32_t op, part;      // op index; start: ops' remaining length (len - off), end: used length (off + 1)
    uint32_t R, Q, U;       // start: exclusive counts at the unit; end: inclusive counts
};

#define RB_WALK_MAX 24

// regular records only (M I D = X): ref = not I, query = not D
__device__ __forceinline__ uint32_t rb_rl(uint32_t v) { return rb_opc(v) == RB_OP_I ? 0u : rb_len(v); }
__device__ __forceinline__ uint32_t rb_ql(uint32_t v) { return rb_opc(v) == RB_OP_D ? 0u : rb_len(v); }
__device__ __forceinline__ bool rb_ism(uint32_t v) { return rb_in(RB_MATCH_MASK, rb_opc(v)); }

// ops[] = the record's kept ops, n of them.  (cR,cQ,cU) = prefixes at op index cidx (checkpoint).
// D in [cR, next checkpoint's R) and D < Rtot.  is_start selects search-right (true) / search-left.
__device__ __forceinline__ rb_bres rb_resolve(const uint32_t *__restrict__ ops, uint32_t n, int32_t cidx, uint32_t cR, uint32_t cQ,
                                              uint32_t cU, uint32_t D, bool is_start, int policy) {
    rb_bres o;
    o.st = RB_S_DEFER;
    o.op = o.part = o.R = o.Q = o.U = 0;
    // 16 ops of the checkpoint group (cidx may be negative by up to 3 in the aligned head: masked)
    uint32_t g[16];
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(ops + cidx); // 16-byte aligned by construction
        const uint4 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
        g[0] = a0.x; g[1] = a0.y; g[2] = a0.z; g[3] = a0.w; g[4] = a1.x; g[5] = a1.y; g[6] = a1.z; g[7] = a1.w;
        g[8] = a2.x; g[9] = a2.y; g[10] = a2.z; g[11] = a2.w; g[12] = a3.x; g[13] = a3.y; g[14] = a3.z; g[15] = a3.w;
    }
    // find the ref-consuming op f with Rx <= D < Rx + len
    int32_t fi = -1;
    uint32_t fv = 0, fR = 0, fQ = 0, fU = 0, pv = (RB_NULL_OP);
    {
        uint32_t R = cR, Q = cQ, U = cU, prev = RB_NULL_OP;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int32_t idx = cidx + k;
            const bool valid = (uint32_t)idx < n;
            const uint32_t v = valid ? g[k] : RB_NULL_OP;
            const uint32_t rl = valid ? rb_rl(v) : 0u;
            if (fi < 0 && rl != 0 && (uint32_t)(D - R) < rl) {
                fi = idx;
                fv = v;
                fR = R;
                fQ = Q;
                fU = U;
                pv = prev;
            }
            R += rl;
            Q += valid ? rb_ql(v) : 0u;
            U += valid ? rb_len(v) : 0u;
            if (valid) prev = v;
        }
    }
    if (fi < 0) return o; // should not happen; the generic kernel sorts it out
    if (fi > 0 && pv == RB_NULL_OP) pv = ops[fi - 1]; // previous op lives in the group before
    const uint32_t off = D - fR;
    if (is_start) {
        int32_t X; // first match-type op with index >= X
        if (off > 0) { // the boundary base and the next base share op f
            if (rb_ism(fv)) {
                o.st = RB_S_OK, o.op = (uint32_t)fi, o.part = rb_len(fv) - (off - 1), o.R = fR + off - 1, o.Q = fQ + off - 1, o.U = fU + off - 1;
                return o;
            }
            X = fi + 1;
        } else { // boundary base is the last unit before op f: the last equal element is the unit before f
            if (fi > 0 && rb_ism(pv)) {
                o.st = RB_S_OK, o.op = (uint32_t)(fi - 1), o.part = 1u, o.R = fR - 1, o.Q = fQ - 1, o.U = fU - 1;
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
        for (int t = 0; t < RB_WALK_MAX; t++) {
            if (i >= X && rb_ism(v)) {
                o.st = RB_S_OK, o.op = (uint32_t)i, o.part = rb_len(v), o.R = R, o.Q = Q, o.U = U;
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
                o.st = RB_S_OK, o.op = (uint32_t)fi, o.part = off, o.R = D, o.Q = fQ + off, o.U = fU + off;
                return o;
            }
            Y = fi - 1;
        } else {
            if (fi > 0 && rb_ism(pv)) {
                o.st = RB_S_OK, o.op = (uint32_t)(fi - 1), o.part = rb_len(pv), o.R = fR, o.Q = fQ, o.U = fU;
                return o;
            }
            Y = fi - 2;
        }
        // walk left (paf.rs:555-557): (R,Q,U) are the prefixes at the END of op i
        uint32_t R = fR, Q = fQ, U = fU;
        int32_t i = fi - 1;
        for (int t = 0; t < RB_WALK_MAX; t++) {
            if (i < 0) {
                o.st = RB_S_NONE; // stops at unit 0, which lies before any start
                return o;
            }
            const uint32_t v = ops[i];
            if (i <= Y && rb_ism(v)) {
                o.st = RB_S_OK, o.op = (uint32_t)i, o.part = rb_len(v), o.R = R, o.Q = Q, o.U = U;
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

#define RB_SMAX 20 // steps (of 256 ops) whose checkpoints fit in LDS at once
#define RB_CP_PER_STEP 16
#ifndef RB_PF
#define RB_PF 4 // steps (1 KiB each) of stream loads in flight per wave
#endif
#ifndef RB_EB
#define RB_EB 4 // output groups (16 B) per lane and emission batch
#endif

__global__ __launch_bounds__(256) void rb_k_liftover_stream(rb_lift_params p) {
    // checkpoints: exclusive (R,Q,U) prefixes every 16 ops, SoA so that R can be binary-searched
    __shared__ uint32_t cp_all[4][3][RB_SMAX * RB_CP_PER_STEP];
    __shared__ uint32_t et_all[4][5][RB_HMAX]; // per clip: first output op, a_op, out_n, first len, last len | verbatim
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wave >= p.n_rec) return;
    const int lane = rb_lane();
    uint32_t *cpR = cp_all[threadIdx.x >> 6][0], *cpQ = cp_all[threadIdx.x >> 6][1], *cpU = cp_all[threadIdx.x >> 6][2];
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
    const uint32_t cg = p.contig[r];
    uint64_t ws = 0, we = 0;
    bool mono = true;
    if (!explicit_w) {
        ws = p.cw_off[cg];
        we = p.cw_off[cg + 1];
        mono = p.cw_mono[cg] != 0;
    }
    const bool fast = (nr->flags & RB_F_REGULAR) != 0; // window order does not matter: resolution is per lane
    if (!fast) {
        rb_defer_record(p, r, nr, h0, nh, explicit_w, mono, ws, we, lane);
        return;
    }
    const uint64_t t_st = nr->t_st, t_en = nr->t_en, q_st = nr->q_st, q_en = nr->q_en;
    const uint32_t n = nr->n_ops;
    const bool minus = p.strand[r] == (uint8_t)'-';
    const uint64_t rec0 = p.op_off[r] + nr->first_op; // global index of the record's first kept op
    const uint32_t *rec_ops = p.ops + rec0;
    const uint64_t lo = (explicit_w || !mono) ? 0 : p.win_lo[r];
    uint64_t scan_pos = ws; // non-monotone window lists: next window of the slice to test
    const uint32_t arena = (uint32_t)(wave % p.n_arena);
    const uint64_t g0 = rec0 & ~3ull, gend = rec0 + n;
    const uint32_t n_steps = (uint32_t)((gend - g0 + 255u) >> 8);
    const int32_t head = (int32_t)(rec0 - g0); // 0..3 padding ops in front of the record in step 0
    const uint64_t glane = g0 + (uint64_t)lane * 4u;

    for (uint64_t jb = 0; jb < nh; jb += RB_HMAX) {
        const uint32_t nb = (uint32_t)((nh - jb) < RB_HMAX ? (nh - jb) : RB_HMAX);
        // ---- per-hit setup: lanes j and j + 32 both look at window jb + j; lane j resolves its start
        //      boundary, lane j + 32 its end boundary; lane j then owns the row ----
        uint64_t wst = 0, wen = 0;
        uint32_t win = 0;
        const uint32_t hl = (uint32_t)lane & 31u;
        const bool own = hl < nb;
        const bool mine = own && lane < 32;
        const bool is_start = lane < 32;
        if (!explicit_w && !mono) {
            // windows of this contig are not sorted: collect the next nb overlapping ones in BED order, 64
            // candidates per ballot (same test as rb_k_count_hits, so the counts agree)
            uint32_t *widx = &et_all[threadIdx.x >> 6][0][0];
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
                wst = p.w_st[idx];
                wen = p.w_en[idx];
                win = p.w_orig[idx];
            }
        } else if (own) {
            if (explicit_w) {
                wst = p.x_st[h0 + jb + hl];
                wen = p.x_en[h0 + jb + hl];
                win = (uint32_t)(jb + hl);
            } else {
                wst = p.w_st[lo + jb + hl];
                wen = p.w_en[lo + jb + hl];
                win = p.w_orig[lo + jb + hl];
            }
        }
        const bool inside = own && (t_st > wst && t_en < wen); // liftover.rs:23-25
        // D = (relative ref offset of the boundary base) + 1
        const uint32_t D = is_start ? (uint32_t)((wst > t_st ? wst : t_st) - t_st) + 1u // liftover.rs:28
                                    : (uint32_t)((wen < t_en ? wen : t_en) - t_st);     // (min(en,t_en) - 1 - t_st) + 1, :38-40
        bool need = own && !inside;
        rb_bres O;
        O.st = RB_S_UNRES;
        O.op = O.part = O.R = O.Q = O.U = 0;

        // ---- stream the record, RB_SMAX steps per segment; resolve after each segment ----
        uint32_t Rb = 0, Qb = 0, Ub = 0; // running totals
        if (__ballot(need) != 0 && !(p.debug_skip & 4)) {
            auto load_step = [&](uint32_t stp) -> uint4 {
                const uint64_t gi = glane + ((uint64_t)stp << 8);
                return gi < gend ? *reinterpret_cast<const uint4 *>(p.ops + gi) : make_uint4(0, 0, 0, 0);
            };
            uint4 pf[RB_PF];
#pragma unroll
            for (int q = 0; q < RB_PF; q++) pf[q] = load_step((uint32_t)q);
            for (uint32_t seg0 = 0; seg0 < n_steps; seg0 += RB_SMAX) {
                const uint32_t seg1 = (seg0 + RB_SMAX < n_steps) ? seg0 + RB_SMAX : n_steps;
                const uint32_t Rseg = Rb;
                for (uint32_t st = seg0; st < seg1; st++) {
                    const uint4 cur = pf[0];
#pragma unroll
                    for (int q = 0; q + 1 < RB_PF; q++) pf[q] = pf[q + 1];
                    pf[RB_PF - 1] = load_step(st + RB_PF);
                    const int32_t idx0 = (int32_t)(st << 8) + lane * 4 - head;
                    const uint32_t raw[4] = {cur.x, cur.y, cur.z, cur.w};
                    uint32_t sr = 0, sq = 0, su = 0;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const bool valid = (uint32_t)(idx0 + q) < n; // also rejects the negative head indices
                        const uint32_t len = valid ? rb_len(raw[q]) : 0u;
                        const uint32_t opc = rb_opc(raw[q]);
                        sr += (opc == RB_OP_I) ? 0u : len;
                        sq += (opc == RB_OP_D) ? 0u : len;
                        su += len;
                    }
                    const uint32_t ir = rb_wave_scan_incl(sr), iq = rb_wave_scan_incl(sq), iu = rb_wave_scan_incl(su);
                    if ((lane & 3) == 0) { // checkpoint every 16 ops
                        const uint32_t t = (st - seg0) * RB_CP_PER_STEP + ((uint32_t)lane >> 2);
                        cpR[t] = Rb + ir - sr;
                        cpQ[t] = Qb + iq - sq;
                        cpU[t] = Ub + iu - su;
                    }
                    Rb += rb_readlane<uint32_t>(ir, 63);
                    Qb += rb_readlane<uint32_t>(iq, 63);
                    Ub += rb_readlane<uint32_t>(iu, 63);
                }
                // ---- lane-parallel resolution of the boundaries that fall in this segment ----
                const bool last_seg = seg1 == n_steps;
                const uint32_t n_cp = (seg1 - seg0) * RB_CP_PER_STEP;
                const int32_t cp_idx0 = (int32_t)(seg0 << 8) - head; // op index of checkpoint 0
                {
                    const bool todo = need && D >= Rseg && (D < Rb || (last_seg && D == Rb));
                    if (todo && !(p.debug_skip & 2)) {
                        if (D == Rb) { // boundary on the record's last base; the last op is match-type
                            const uint32_t lv = rec_ops[n - 1];
                            O.st = RB_S_OK, O.op = n - 1;
                            if (is_start) O.part = 1u, O.R = Rb - 1, O.Q = Qb - 1, O.U = Ub - 1;
                            else O.part = rb_len(lv), O.R = Rb, O.Q = Qb, O.U = Ub;
                        } else {
                            // last checkpoint with R <= D (R is non-decreasing)
                            uint32_t lo_t = 0, hi_t = n_cp;
                            while (hi_t - lo_t > 1) {
                                const uint32_t mid = (lo_t + hi_t) >> 1;
                                if (cpR[mid] <= D) lo_t = mid; else hi_t = mid;
                            }
                            O = rb_resolve(rec_ops, n, cp_idx0 + (int32_t)lo_t * 16, cpR[lo_t], cpQ[lo_t], cpU[lo_t], D, is_start, p.policy);
                        }
                        need = false;
                    }
                }
                if (p.early_exit && __ballot(need) == 0) break;
            }
        }

        // ---- finalize: lane j (< 32) computes the row of hit jb + j; the end comes from lane j + 32 ----
        const rb_bres A = O;
        rb_bres B;
        B.st = (uint32_t)__shfl((int)O.st, lane + 32, 64);
        B.op = (uint32_t)__shfl((int)O.op, lane + 32, 64);
        B.part = (uint32_t)__shfl((int)O.part, lane + 32, 64);
        B.R = (uint32_t)__shfl((int)O.R, lane + 32, 64);
        B.Q = (uint32_t)__shfl((int)O.Q, lane + 32, 64);
        B.U = (uint32_t)__shfl((int)O.U, lane + 32, 64);
        uint32_t status = RB_ST_OK, out_n = 0, a_op = 0;
        uint64_t o_tst = 0, o_ten = 0, o_qst = 0, o_qen = 0;
        uint32_t o_nm = 0, o_al = 0;
        bool defer = false;
        if (mine) {
            if (inside) {
                out_n = n;
                o_tst = t_st, o_ten = t_en, o_qst = q_st, o_qen = q_en;
                o_nm = nr->nmatch, o_al = nr->aln_len;
            } else if (A.st == RB_S_DEFER || B.st == RB_S_DEFER || A.st == RB_S_UNRES || B.st == RB_S_UNRES) {
                defer = true;
            } else if (A.st == RB_S_NONE || B.st == RB_S_NONE || A.U >= B.U) {
                status = RB_ST_NONE_INDEL; // liftover.rs:52-54
            } else {
                a_op = A.op;
                o_tst = t_st + A.R; // liftover.rs:57-60, :77-82
                o_ten = t_st + B.R;
                if (!minus) {
                    o_qst = q_st + A.Q;
                    o_qen = q_st + B.Q;
                } else {
                    o_qst = q_en - B.Q;
                    o_qen = q_en - A.Q;
                }
                o_al = B.U - A.U;
                o_nm = (B.R + B.Q - B.U) - (A.R + A.Q - A.U); // match units = ref + query - all (M I D = X only)
                out_n = B.op - A.op + 1;
            }
        }
        // space for the clipped cigars: one atomic per pass, each hit padded to 4 ops
        const uint32_t padded = (mine && !defer && status == RB_ST_OK && !p.desc_mode) ? ((out_n + 3u) & ~3u) : 0u;
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
        const uint64_t my_off = p.arena_origin + (uint64_t)arena * p.arena_size + base + (incl - padded);
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
                w.flags = (inside ? RB_HIT_INSIDE : 0) | ((p.desc_mode && status == RB_ST_OK) ? RB_HIT_DESCRIPTOR : 0);
                w.out_n = status == RB_ST_OK ? out_n : 0;
                w.t_st = o_tst;
                w.t_en = o_ten;
                w.q_st = o_qst;
                w.q_en = o_qen;
                w.nmatch = o_nm;
                w.aln_len = o_al;
                w.out_off = status == RB_ST_OK ? (p.desc_mode ? 4ull * (h0 + jb + lane) : my_off) : 0;
                *row = w;
                if (p.desc_mode && status == RB_ST_OK) // which ops of the ORIGINAL cigar the clip keeps
                    *reinterpret_cast<uint4 *>(p.out_ops + 4ull * (h0 + jb + lane)) =
                        make_uint4(nr->first_op + a_op, out_n, inside ? 0u : A.part, inside ? 0u : B.part);
            }
        }
        // ---- emit: the clips of this pass occupy one contiguous output region [base, base + total).
        //      Lane l moves output groups (4 ops, 16 B aligned) l, l + 64, ...; the clip a group belongs to is
        //      found by a 5-step search of the clips' output offsets in LDS (offsets are non-decreasing; the
        //      last clip starting at or before the group is the one that contains it).  8 independent 16 B
        //      loads per lane are in flight per batch. ----
        if (fits && total && !(p.debug_skip & 1)) {
            uint32_t *et = &et_all[threadIdx.x >> 6][0][0];
            if (mine) {
                et[0 * RB_HMAX + lane] = incl - padded; // first output op of the clip
                et[1 * RB_HMAX + lane] = a_op;
                et[2 * RB_HMAX + lane] = padded ? out_n : 0u;
                et[3 * RB_HMAX + lane] = A.part;
                et[4 * RB_HMAX + lane] = B.part | (inside ? 0x80000000u : 0u);
            }
            uint32_t *__restrict__ dst = p.out_ops + (p.arena_origin + (uint64_t)arena * p.arena_size + base);
            for (uint32_t gb = 0; gb * 4u < total; gb += RB_EB * 64u) {
                uint4 v[RB_EB];
                uint32_t cj[RB_EB], cpos[RB_EB];
#pragma unroll
                for (int u = 0; u < RB_EB; u++) {
                    const uint32_t o = (gb + (uint32_t)u * 64u + (uint32_t)lane) * 4u; // output op of this group
                    uint32_t lo_j = 0, hi_j = nb;
                    while (hi_j - lo_j > 1) { // last clip with first output op <= o
                        const uint32_t mid = (lo_j + hi_j) >> 1;
                        if (et[mid] <= o) lo_j = mid; else hi_j = mid;
                    }
                    cj[u] = lo_j;
                    const uint32_t pos = o - et[lo_j];
                    const uint32_t e_n = et[2 * RB_HMAX + lo_j];
                    const bool live = o < total && pos < e_n;
                    cpos[u] = live ? pos : 0xFFFFFFFFu;
                    v[u] = live ? rb_emit_load(rec_ops + et[1 * RB_HMAX + lo_j], pos, e_n, p.debug_skip) : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < RB_EB; u++) {
                    if (cpos[u] != 0xFFFFFFFFu) {
                        const uint32_t o = (gb + (uint32_t)u * 64u + (uint32_t)lane) * 4u;
                        const uint32_t j = cj[u];
                        const uint32_t bl = et[4 * RB_HMAX + j];
                        rb_emit_store(dst + (o - cpos[u]), cpos[u], et[2 * RB_HMAX + j], et[3 * RB_HMAX + j], bl & 0x7FFFFFFFu, (bl >> 31) != 0, v[u], p.debug_skip);
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
                w.out_off = p.arena_origin + (uint64_t)arena * p.arena_size + b0;
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
        w.out_off = p.arena_origin + (uint64_t)arena * p.arena_size + b0;
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
    // diagnostics: RB_DEBUG_DYN_LDS=<bytes> adds unused dynamic LDS to lower the occupancy
    static const unsigned dyn = getenv("RB_DEBUG_DYN_LDS") ? (unsigned)atoi(getenv("RB_DEBUG_DYN_LDS")) : 0u;
    hipLaunchKernelGGL(rb_k_liftover_stream, dim3(blocks), dim3(256), dyn, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t rb_launch_liftover_tail(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_liftover_generic, dim3(1024), dim3(256), 0, stream, *p);
    hipLaunchKernelGGL(rb_k_finish, dim3(1), dim3(64), 0, stream, *p);
    return hipGetLastError();
}

extern "C" size_t rb_scan_block_sums_count(uint64_t n_rec) { return (size_t)((n_rec + RB_SCAN_PER_BLOCK - 1) / RB_SCAN_PER_BLOCK + 2); }
