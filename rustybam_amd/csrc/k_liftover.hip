// k_liftover.hip -- liftover / break-paf clip kernels for gfx950 (wave64, CDNA4).
//
// Replaces liftover::trim_helper + trim_paf_rec_to_rgn (liftover.rs:17-132) and everything they
// call (aligned_pairs paf.rs:501-538, tpos_to_idx_match :541-561, subset_cigar /
// collapse_long_cigar :593-620).  The reference expands every CIGAR to per-base arrays (24 B per
// aligned base) and binary-searches them; here the walk stays in op space:
//
//   rb_k_count_hits     one thread per record: number of overlapping windows (paf.rs:622-627)
//   rb_k_scan_*         exclusive scan of the counts -> first row of every record (canonical order)
//   rb_k_make_jobs      one thread per schedule slot: a 64-byte job descriptor, so that a wave starts on its record
//                       after ONE load instead of a chain of dependent ones
//   rb_k_liftover_stream  ONE WAVEFRONT PER RECORD.  The record's packed ops stream from HBM once (32 contiguous
//                       bytes per lane, 2 KiB per step, two steps in flight); per lane the reference / query /
//                       unit lengths of 8 ops are summed (op class -> mask with one v_bfe_i32), three 6-step DPP
//                       prefix scans give the running offsets, every second lane leaves a 16-op checkpoint in
//                       LDS; window boundaries are resolved lane-parallel against the checkpoints (lane j: start
//                       of window j, lane j + 32: its end).  One atomic per pass reserves the output; every clip
//                       is placed so that it keeps the 16-byte phase of its ops in the input, hence its interior
//                       is copied (out of L2 / Infinity Cache: the record has just been streamed) with aligned
//                       16-byte loads and stores through a hand-pipelined 4-buffer ring, and only the two end
//                       groups of a clip are patched.
//   rb_k_liftover_generic  one thread per hit, serial walk: every case the streaming kernel declines
//                       (irregular CIGARs: N/S/H/P, zero lengths, adjacent ops of one type that must
//                       merge (paf.rs:602-620); the legacy binary-search policy when the duplicate
//                       choice matters; look-aheads / look-backs longer than RB_WALK_MAX ops).
//
// Roofline: HBM.  Algorithmic bytes: 4 B per input op + 48 B per record + 88 B per hit + 4 B per
// emitted op.  No MFMA: integer / index work only.
#include "rb_lift.h"
#include <type_traits>

// diagnostics (debug_skip & 32): shader-clock time of each phase of a record, every 16th record, summed in units of 16
// cycles into counters->_pad[0..6]: job + windows, stream + resolve, verdict + finalize, reservation, rows + end groups,
// interior copy, (unused)
#define RB_PHASE(i)                                                                                                  \
    if (p.debug_skip & 32) {                                                                                         \
        const long long t_now = clock64();                                                                           \
        if (lane == 0 && (wave & 15) == 0) atomicAdd(&p.counters->_pad[i], (uint32_t)((t_now - t_prev) >> 4));       \
        t_prev = t_now;                                                                                              \
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
__global__ __launch_bounds__(256) void rb_k_scan_apply(uint64_t *v, uint64_t n, const uint64_t *block_sums, uint64_t *total_out) {
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
        if (total_out) *total_out = total;
    }
}

// ------------------------------------------------------------------------------------------------
// clip jobs: what a wave needs about its record, gathered into the record's schedule slot
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rb_k_make_jobs(rb_lift_params p) {
    // one thread per RECORD (rows, offsets and hit counts are read in memory order), the job goes to the record's slot
    const uint64_t r64 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r64 >= p.n_rec) return;
    const uint32_t r = (uint32_t)r64;
    const uint64_t w = p.slot_of[r];
    const rb_norm_row *nr = &p.norm[r];
    rb_job j;
    j.r = r;
    j.n = nr->n_ops;
    j.rec0 = p.op_off[r] + nr->first_op;
    j.t_st = nr->t_st, j.t_en = nr->t_en, j.q_st = nr->q_st, j.q_en = nr->q_en;
    const uint64_t k = p.canon_pos[r];
    const uint64_t h0 = p.hit_off[k], nh = p.hit_off[k + 1] - h0;
    const bool explicit_w = p.x_st != nullptr;
    const uint32_t cg = p.contig[r];
    const bool mono = explicit_w || (cg < p.n_contig && p.cw_mono[cg] != 0);
    uint32_t f = 0;
    const bool provisional = (nr->flags & RB_F_PROVISIONAL) != 0; // fused scan: the clip kernel verifies the record itself,
    if (nr->status == RB_ST_OK && (nh != 0 || provisional)) {     // also when no window overlaps it (liftover.rs:119-121)
        if (h0 + nh > p.rows_cap) f |= RB_JOB_ROWS_OVERFLOW;
        else f |= RB_JOB_VALID;
    }
    if ((nr->flags & RB_F_REGULAR) || (provisional && !(nr->flags & RB_F_ENDS_NOT_MATCH))) f |= RB_JOB_REGULAR;
    if (p.strand[r] == (uint8_t)'-') f |= RB_JOB_MINUS;
    if (mono) f |= RB_JOB_MONO;
    j.flags = f;
    j.h0 = (uint32_t)h0;
    j.nh = (uint32_t)nh;
    j.lo = (explicit_w || !mono || !(f & RB_JOB_VALID) || nh == 0) ? 0u : p.win_lo[r];
    p.jobs[w] = j;
}

// ------------------------------------------------------------------------------------------------
// streaming kernel
// ------------------------------------------------------------------------------------------------
#define RB_STEP_SHIFT 9  // a step is 512 ops: 8 ops (two 16-byte loads, 32 contiguous bytes) per lane
#define RB_CP_PER_STEP 32 // one checkpoint per 16 ops (every second lane)
#ifndef RB_PF
#define RB_PF 2 // steps (2 KiB each) of stream loads in flight per wave
#endif
#define RB_ET (RB_HMAX + 1) // slots per column of the clip table (one sentinel)
#ifndef RB_SEG
#define RB_SEG 10
#endif
#define RB_SMAX ((RB_SEG / RB_PF) * RB_PF) // steps whose checkpoints fit in LDS at once; whole turns of the load ring
#ifndef RB_WPE
#define RB_WPE 5, 6 // waves per SIMD the register budget is cut for
#endif
#ifdef RB_NO_NT
#define RB_ST_NT ""
#endif
#ifndef RB_ST_NT
#define RB_ST_NT " nt" // clip stores are non-temporal: written once, never re-read by this kernel
#endif
#ifndef RB_EB
#define RB_EB 4 // emission buffers (one 16-byte group per lane each) in rotation
#endif

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RB_WPE))) void rb_k_liftover_stream(rb_lift_params p) {
    // checkpoints: exclusive (R,Q,U) prefixes every 16 ops, SoA so that R can be binary-searched
    __shared__ uint32_t cp_all[4][3][RB_SMAX * RB_CP_PER_STEP];
    __shared__ uint32_t et_all[4][3][RB_HMAX + 1]; // per clip of a pass: region offset, first group, last group (+ sentinel)
    const uint32_t wib = rb_first(threadIdx.x >> 6); // wave in block (told to the compiler as the wave-uniform value it is)
    const uint64_t wave = (uint64_t)p.wave0 + (uint64_t)blockIdx.x * 4u + wib;
    if (wave >= p.wave_end) return;
    const int lane = rb_lane();
    long long t_prev = (p.debug_skip & 32) ? clock64() : 0;
    uint32_t *cpR = cp_all[wib][0], *cpQ = cp_all[wib][1], *cpU = cp_all[wib][2];
    const rb_job jb_ = p.jobs[wave]; // (uniform address: one 64-byte request)
    const uint32_t jflags = rb_first(jb_.flags);
    if (jflags & RB_JOB_ROWS_OVERFLOW) { // rows do not fit: flag and leave (host retries with more room)
        if (lane == 0) p.counters->overflow = 1;
        return;
    }
    if (!(jflags & RB_JOB_VALID)) return;
    const uint32_t r = rb_first(jb_.r);
    const rb_norm_row *nr = &p.norm[r];
    const uint64_t h0 = rb_first(jb_.h0);
    const uint64_t nh = rb_first(jb_.nh);
    const bool explicit_w = p.x_st != nullptr;
    const bool mono = (jflags & RB_JOB_MONO) != 0;
    uint64_t ws = 0, we = 0;
    if (!explicit_w && (!mono || !(jflags & RB_JOB_REGULAR))) { // the contig's window slice: only the rare paths need it
        const uint32_t cg = p.contig[r];
        ws = p.cw_off[cg];
        we = p.cw_off[cg + 1];
    }
    if (!(jflags & RB_JOB_REGULAR)) { // window order does not matter on the fast path: resolution is per lane
        if (p.fused && lane == 0) { // (a provisional row that cannot take the fast path: the full scan completes it)
            const unsigned long long i = atomicAdd(p.pend_count, 1ull);
            p.pend_list[i] = r;
        }
        rb_defer_record(p, r, nr, h0, nh, explicit_w, mono, ws, we, lane);
        return;
    }
    // wave-uniform coordinates: pinned to scalar registers (left alone, the compiler keeps vector copies alive through the
    // whole record and spills them)
    auto sgpr64 = [](uint64_t v) -> uint64_t {
        uint32_t lo = rb_first((uint32_t)v), hi = rb_first((uint32_t)(v >> 32));
        asm volatile("" : "+s"(lo), "+s"(hi));
        return ((uint64_t)hi << 32) | lo;
    };
    const uint64_t t_st = sgpr64(jb_.t_st), t_en = sgpr64(jb_.t_en), q_st = sgpr64(jb_.q_st), q_en = sgpr64(jb_.q_en);
    const uint32_t n = rb_first(jb_.n);
    const bool minus = (jflags & RB_JOB_MINUS) != 0;
    const uint64_t rec0 = rb_first64(jb_.rec0); // global index of the record's first kept op
    const uint32_t *rec_ops = p.ops + rec0;
    const uint64_t lo = rb_first(jb_.lo);
    uint64_t scan_pos = ws; // non-monotone window lists: next window of the slice to test
    const uint32_t arena = (uint32_t)(wave % p.n_arena);
    const uint64_t g0 = rec0 & ~3ull, gend = rec0 + n;
    const uint32_t n_steps = (uint32_t)((gend - g0 + (1u << RB_STEP_SHIFT) - 1u) >> RB_STEP_SHIFT);
    const int32_t head = (int32_t)(rec0 - g0); // 0..3 padding ops in front of the record in step 0
    const uint32_t *__restrict__ gbase0 = p.ops + g0;                 // the record's first (aligned) 16-byte group
    const uint32_t last_off = (uint32_t)(((gend - 1u) & ~3ull) - g0); // last 16-byte group that holds an op of this record

    const bool fused = p.fused != 0;
    uint32_t rec_nmatch = 0, rec_aln_len = 0; // of the whole (normalised) record: taken from its row, or from the fused verification
    if (!fused) rec_nmatch = nr->nmatch, rec_aln_len = nr->aln_len;
    const uint64_t n_items = (nh == 0 && fused) ? 1 : nh; // (a record no window overlaps is still streamed once, to verify it)
    for (uint64_t jb = 0; jb < n_items; jb += RB_HMAX) {
        const uint32_t nb = (uint32_t)((nh - jb) < RB_HMAX ? (nh - jb) : RB_HMAX);
        const bool validate = fused && jb == 0;
        // ---- per-hit setup: lanes j and j + 32 both look at window jb + j; lane j resolves its start
        //      boundary, lane j + 32 its end boundary; lane j then owns the row ----
        const uint32_t hl = (uint32_t)lane & 31u;
        const bool own = hl < nb;
        const bool mine = own && lane < 32;
        const bool is_start = lane < 32;
        const rb_pass_win pw = rb_pass_windows(p, &et_all[wib][0][0], explicit_w, mono, ws, we, lo, h0, jb, nb, t_st, t_en, scan_pos, lane);
        const uint64_t wst = pw.wst, wen = pw.wen;
        const uint32_t win = pw.win;
        const bool inside = own && (t_st > wst && t_en < wen); // liftover.rs:23-25
        // D = (relative ref offset of the boundary base) + 1
        const uint32_t D = is_start ? (uint32_t)((wst > t_st ? wst : t_st) - t_st) + 1u // liftover.rs:28
                                    : (uint32_t)((wen < t_en ? wen : t_en) - t_st);     // (min(en,t_en) - 1 - t_st) + 1, :38-40
        bool need = own && !inside;
        rb_bres O;
        O.st = RB_S_UNRES;
        O.op = O.part = O.R = O.Q = O.U = 0;
        if (p.debug_skip & 32) { // (the window values must have arrived for the phase boundary to mean anything)
            asm volatile("s_waitcnt vmcnt(0)");
        }
        RB_PHASE(0)

        // ---- stream the record, RB_SMAX steps per segment; resolve after each segment ----
        uint32_t Rb = 0, Qb = 0, Ub = 0; // running totals
        // fused verification (first pass): AND of the "regular op" masks, minimum length, minimum of code XOR previous code
        // (0 = two adjacent ops of one type), minimum op code (0 = an M); sums too large for 32-bit scans zero the minimum length
        uint32_t v_reg = 0xFFFFFFFFu, v_minlen = 0xFFFFFFFFu, v_adj = 0xFFFFFFFFu, v_mincode = 15u;
        uint32_t v_carry = 0xFu;       // last op word of the previous step (code 15: equals nothing)
        unsigned long long v_utot = 0; // 64-bit sum of all lengths
        if ((__ballot(need) != 0 || validate) && !(p.debug_skip & 4)) {
            auto load_half = [&](uint32_t stp, uint32_t half) -> uint4 {
                // unconditional (an exec-masked load makes the compiler drain the whole ring at the loop edge):
                // groups past the record's end re-read its last group; the tail step masks them out
                uint32_t off = (stp << RB_STEP_SHIFT) + half * 4u + (uint32_t)lane * 8u; // (uniform base + 32-bit lane offset)
                off = off < last_off ? off : last_off;
                return *reinterpret_cast<const uint4 *>(gbase0 + off);
            };
            uint4 pf[RB_PF][2];
#pragma unroll
            for (int q = 0; q < RB_PF; q++) {
                pf[q][0] = load_half((uint32_t)q, 0u);
                pf[q][1] = load_half((uint32_t)q, 1u);
                __builtin_amdgcn_sched_barrier(0); // keep issue order = consumption order, so that vmcnt counts stay exact
            }
            for (uint32_t seg0 = 0; seg0 < n_steps; seg0 += RB_SMAX) {
                const uint32_t seg1 = (seg0 + RB_SMAX < n_steps) ? seg0 + RB_SMAX : n_steps;
                const uint32_t Rseg = Rb;
                // the ring is indexed statically (unrolled by RB_PF): rotating it with register moves would make
                // every step wait for ALL loads in flight (the moves read their destination registers)
                for (uint32_t st0 = seg0; st0 < seg1; st0 += RB_PF) {
#pragma unroll
                    for (int ring = 0; ring < RB_PF; ring++) {
                        const uint32_t st = st0 + (uint32_t)ring;
                        if (st < seg1) { // (no break: the ring must be in the same state on every path)
                            uint32_t raw[8] = {pf[ring][0].x, pf[ring][0].y, pf[ring][0].z, pf[ring][0].w,
                                               pf[ring][1].x, pf[ring][1].y, pf[ring][1].z, pf[ring][1].w};
                            if (st == 0 || st + 1 == n_steps) { // only the first / last step can hold ops of the neighbours
                                const int32_t idx0 = (int32_t)(st << RB_STEP_SHIFT) + lane * 8 - head;
#pragma unroll
                                for (int q = 0; q < 8; q++)
                                    if ((uint32_t)(idx0 + q) >= n) raw[q] = 0u; // (also the negative head indices); 0 = a 0-length M
                            }
                            if (validate) {
                                const int32_t idx0v = (int32_t)(st << RB_STEP_SHIFT) + lane * 8 - head;
                                auto verify = [&](auto edge_c) {
                                    constexpr bool edge = decltype(edge_c)::value; // first / last step: ops of the neighbours are skipped
                                    uint32_t mylast = raw[7];                      // adjacency form: 0xF where there is no op
                                    if (edge && (uint32_t)(idx0v + 7) >= n) mylast = 0xFu;
                                    uint32_t prevw = rb_prev_lane(mylast, v_carry); // previous lane's last op; lane 0: previous step's
                                    v_carry = rb_readlane<uint32_t>(mylast, 63);
#pragma unroll
                                    for (int q = 0; q < 8; q++) {
                                        const bool ok = !edge || (uint32_t)(idx0v + q) < n;
                                        const uint32_t w = raw[q];
                                        if (ok) {
                                            v_reg &= (uint32_t)__builtin_amdgcn_sbfe((int)0x018F018Fu, w, 1u); // M I D N = X
                                            v_minlen = v_minlen < rb_len(w) ? v_minlen : rb_len(w);
                                            const uint32_t x = (w ^ prevw) & 15u;
                                            v_adj = v_adj < x ? v_adj : x;
                                            v_mincode = v_mincode < (w & 15u) ? v_mincode : (w & 15u);
                                        }
                                        prevw = ok ? w : 0xFu;
                                    }
                                };
                                if (st == 0 || st + 1 == n_steps) verify(std::true_type{});
                                else verify(std::false_type{});
                            }
                            // per-lane sums of the reference / query / unit lengths of 8 ops; regular records hold only
                            // M I D N = X, so "consumes the reference" = not I and "consumes the query" = not D / N: one
                            // v_bfe_i32 per class turns the op code (low bits of the word) into an all-ones / zero mask
                            uint32_t sr = 0, sq = 0, su = 0;
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const uint32_t len = rb_len(raw[q]);
                                sr += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFFDFFFDu, raw[q], 1u);
                                sq += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFF3FFF3u, raw[q], 1u); // not D, not N
                                su += len;
                            }
                            const uint32_t ir = rb_wave_scan_incl(sr), iq = rb_wave_scan_incl(sq), iu = rb_wave_scan_incl(su);
                            if ((lane & 1) == 0) { // checkpoint every 16 ops
                                const uint32_t t = (st - seg0) * RB_CP_PER_STEP + ((uint32_t)lane >> 1);
                                cpR[t] = Rb + ir - sr;
                                cpQ[t] = Qb + iq - sq;
                                cpU[t] = Ub + iu - su;
                            }
                            Rb += rb_readlane<uint32_t>(ir, 63);
                            Qb += rb_readlane<uint32_t>(iq, 63);
                            Ub += rb_readlane<uint32_t>(iu, 63);
                            if (validate) {
                                v_minlen = (su >> 25) ? 0u : v_minlen; // per-lane sums below 2^25 keep the 64-lane scans inside 32 bits (else: handed back like a zero length)
                                v_utot += rb_readlane<uint32_t>(iu, 63);
                            }
                        }
                        // reload the slot only after its ops are consumed: the load then targets the same registers
                        // and the compiler needs no copy (which would wait for every load in flight)
                        pf[ring][0] = load_half(st + RB_PF, 0u);
                        pf[ring][1] = load_half(st + RB_PF, 1u);
                    }
                }
                // ---- lane-parallel resolution of the boundaries that fall in this segment ----
                const bool last_seg = seg1 == n_steps;
                const uint32_t n_cp = (seg1 - seg0) * RB_CP_PER_STEP;
                const int32_t cp_idx0 = (int32_t)(seg0 << RB_STEP_SHIFT) - head; // op index of checkpoint 0
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
                if (p.early_exit && !validate && __ballot(need) == 0) break;
            }
        }

        RB_PHASE(1)
        if (validate) {
            // ---- the verdict of the fused scan: check_integrity (paf.rs:825-857) on the normalised record and the
            //      conditions of the fast path.  A record that fails any of them is handed back: the full record scan
            //      (list mode) decides its status, the generic kernel clips it if it is merely irregular ----
            const bool lane_bad = v_reg != 0xFFFFFFFFu || v_minlen == 0u || v_adj == 0u;
            const bool bad = __ballot(lane_bad) != 0 || rb_first64(v_utot) > 0xFFFFFFFFull || t_en < t_st || q_en < q_st ||
                             (uint64_t)Rb != t_en - t_st || (uint64_t)Qb != q_en - q_st;
            if (bad) {
                if (lane == 0) {
                    const unsigned long long i = atomicAdd(p.pend_count, 1ull);
                    p.pend_list[i] = r;
                }
                if (nh) {
                    if (!explicit_w) {
                        const uint32_t cg = p.contig[r];
                        ws = p.cw_off[cg];
                        we = p.cw_off[cg + 1];
                    }
                    rb_defer_record(p, r, nr, h0, nh, explicit_w, mono, ws, we, lane);
                }
                return;
            }
            rec_nmatch = Rb + Qb - Ub; // match units = ref + query - all (M I D = X only)
            rec_aln_len = Ub;
            const bool has_m = __ballot(v_mincode == 0u) != 0;
            if (lane == 0) {
                rb_norm_row *w = &p.norm_w[r];
                w->nmatch = rec_nmatch;
                w->aln_len = rec_aln_len;
                w->flags = (nr->flags & RB_F_STRIPPED) | RB_F_REGULAR | (has_m ? RB_F_HAS_M : 0u);
            }
            if (nh == 0) return;
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
                o_nm = rec_nmatch, o_al = rec_aln_len;
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
        // space for the clipped cigars: one atomic per pass.  A clip starts `lead` ops into its region so that it keeps the
        // 16-byte phase its ops have in the input: interior 16-byte groups are then copied with aligned loads AND aligned
        // stores and need no patching; only the group(s) holding a clip's first and last op are rewritten op by op.
        const bool emits = mine && !defer && status == RB_ST_OK && !p.desc_mode;
        const uint32_t e_first = (uint32_t)head + a_op; // coordinate (op index + head, counted from the aligned g0) of the first op
        const uint32_t e_cnt = emits ? out_n : 0u;
        const uint32_t eg_last = e_first + e_cnt - 1u;
        const uint32_t eg_f = e_first & ~3u, eg_l = e_cnt ? (eg_last & ~3u) : eg_f;
        // the two end groups are re-read here, BEFORE any store of this pass: vmcnt retires in order on gfx9, a load issued
        // after stores would wait for every one of them; this way the latency hides behind the reservation atomic.
        // (unconditional: a conditional load is sunk to its use; the ops array is padded, a quad may reach past the record)
        const uint32_t *__restrict__ gsrc = rec_ops - head; // coordinate c -> gsrc[c]; 16-byte aligned at c % 4 == 0
        uint4 eg_q0 = *reinterpret_cast<const uint4 *>(gsrc + eg_f);
        uint4 eg_q1 = *reinterpret_cast<const uint4 *>(gsrc + eg_l);
        __builtin_amdgcn_sched_barrier(0);
        RB_PHASE(2)
        const uint32_t lead = e_first & 3u;
        const uint32_t padded = emits ? ((lead + out_n + 3u) & ~3u) : 0u;
        const uint32_t incl = rb_wave_scan_incl(padded);
        const uint32_t total = rb_readlane<uint32_t>(incl, 63);
        uint64_t base = 0;
        if (total) {
            unsigned long long b0 = 0;
            if (lane == 0) b0 = atomicAdd(&p.arena_cur[(uint64_t)arena * RB_ARENA_STRIDE], (unsigned long long)total);
            base = rb_first64(b0);
        }
        RB_PHASE(3)
        const bool fits = base + total <= p.arena_size;
        if (!fits && lane == 0) p.counters->overflow = 1;
        const uint64_t region0 = p.arena_origin + (uint64_t)arena * p.arena_size + base; // multiple of 4 ops
        const uint32_t my_rel = (incl - padded) + lead;                                   // first op of my clip inside the region
        const uint64_t my_off = region0 + my_rel;
        // (the row index is formed from an opaque copy of the lane id: otherwise the compiler hoists the row addresses above
        //  the streaming loop and carries -- or spills -- them through it)
        uint32_t lane_late = (uint32_t)lane;
        asm volatile("" : "+v"(lane_late));
        const uint64_t my_row = h0 + jb + lane_late;
        if (mine) {
            rb_hit_row *row = &p.rows[my_row];
            if (defer) {
                row->rec = r;
                row->win = win;
                row->flags = RB_HIT_GENERIC;
                const unsigned long long g = atomicAdd((unsigned long long *)&p.counters->n_generic, 1ull);
                p.gen_list[g] = (uint32_t)my_row;
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
                w.out_off = status == RB_ST_OK ? (p.desc_mode ? 4ull * my_row : my_off) : 0;
                *row = w;
                if (p.desc_mode && status == RB_ST_OK) // which ops of the ORIGINAL cigar the clip keeps
                    *reinterpret_cast<uint4 *>(p.out_ops + 4ull * my_row) =
                        make_uint4(nr->first_op + a_op, out_n, inside ? 0u : A.part, inside ? 0u : B.part);
            }
        }
        // ---- emit.  (1) end groups: the lane that owns a clip writes the group(s) holding its first and last op, op by op,
        //      with the clipped lengths patched in.  (2) interior groups: the pass's clips occupy one contiguous region
        //      [0, total); lane l copies its 16-byte groups l, l + 64, ... (aligned load from the input, aligned store).
        //      The clip a group belongs to only moves forward as the group index grows, so each lane keeps a cursor into the
        //      clip table in LDS instead of searching it per group. ----
        if (fits && total && !(p.debug_skip & 1)) {
            uint32_t *region = p.out_ops + region0;
            if (e_cnt) { // slots of my region before the first / behind the last op are padding: written as zeros
                uint32_t *__restrict__ dst = region + (incl - padded); // my region; its first group holds coordinate eg_f
                uint32_t q0[4] = {eg_q0.x, eg_q0.y, eg_q0.z, eg_q0.w}, q1[4] = {eg_q1.x, eg_q1.y, eg_q1.z, eg_q1.w};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const uint32_t c0 = eg_f + (uint32_t)q, c1 = eg_l + (uint32_t)q;
                    uint32_t w0 = q0[q], w1 = q1[q];
                    if (!inside) {
                        if (e_cnt == 1u) { // the middle of one op
                            if (c0 == e_first) w0 = ((A.part + B.part - rb_len(w0)) << 4) | rb_opc(w0);
                        } else {
                            if (c0 == e_first) w0 = (A.part << 4) | rb_opc(w0); // first op keeps its tail
                            if (c0 == eg_last) w0 = (B.part << 4) | rb_opc(w0); // (clips of 2..4 ops inside one group)
                            if (c1 == eg_last) w1 = (B.part << 4) | rb_opc(w1); // last op keeps its head
                        }
                    }
                    q0[q] = (c0 < e_first || c0 > eg_last) ? 0u : w0;
                    q1[q] = (c1 > eg_last) ? 0u : w1;
                }
                typedef uint32_t rb_u32x4 __attribute__((ext_vector_type(4)));
                const rb_u32x4 s0 = {q0[0], q0[1], q0[2], q0[3]}, s1 = {q1[0], q1[1], q1[2], q1[3]};
                __builtin_nontemporal_store(s0, reinterpret_cast<rb_u32x4 *>(dst));
                if (eg_l != eg_f) __builtin_nontemporal_store(s1, reinterpret_cast<rb_u32x4 *>(dst + (eg_l - eg_f)));
            }
            RB_PHASE(4)
            // clip table: region offset of the clip's first group, coordinates of its first and last group
            uint32_t *et = &et_all[wib][0][0];
            if (mine) {
                et[0 * RB_ET + lane] = incl - padded;
                et[1 * RB_ET + lane] = eg_f;
                et[2 * RB_ET + lane] = e_cnt ? eg_l : eg_f; // (no ops: no interior group either)
            }
            if (lane == 0) et[0 * RB_ET + nb] = 0xFFFFFFFFu; // sentinel (the table has RB_HMAX + 1 slots)
            uint32_t cj = 0, c_start = et[0], c_fg = et[1 * RB_ET], c_lg = et[2 * RB_ET], n_start = et[1];
            // RB_EB buffers of one group per lane rotate: turn t loads the group of turn t into buffer t % RB_EB and stores
            // the group loaded RB_EB - 1 turns earlier.  The load / wait / store triple is written by hand: vmcnt retires in
            // order on gfx9, and left to the compiler every load of this loop waits for every store before it (it falls back
            // to vmcnt(0) around the predicated accesses).  Between the load of turn t - K (K = RB_EB - 1) and the wait of
            // turn t exactly 2K vector-memory instructions are issued (K loads, K stores; instructions issued with an empty
            // exec mask count too, tools/vmcnt_probe.hip), so s_waitcnt vmcnt(2K) is satisfied exactly when that load has
            // landed and leaves K loads and K stores in flight.  Store data is read when the store issues, so a buffer may be
            // reloaded right after.
            typedef uint32_t rb_u32x4 __attribute__((ext_vector_type(4)));
            const rb_u32x4 z4 = {0, 0, 0, 0};
            rb_u32x4 b0 = z4, b1 = z4, b2 = z4, b3 = z4;
            uint32_t d0 = 0xFFFFFFFFu, d1 = d0, d2 = d0, d3 = d0; // byte offset of the buffer's group in the region, or dead
#if RB_EB == 6
            rb_u32x4 b4 = z4, b5 = z4;
            uint32_t d4 = d0, d5 = d0;
#define RB_PIN_BUFS [p0] "+v"(b0), [p1] "+v"(b1), [p2] "+v"(b2), [p3] "+v"(b3), [p4] "+v"(b4), [p5] "+v"(b5)
#define RB_EMIT_WAIT "s_waitcnt vmcnt(10)\n\t"
#elif RB_EB == 4
#define RB_PIN_BUFS [p0] "+v"(b0), [p1] "+v"(b1), [p2] "+v"(b2), [p3] "+v"(b3)
#define RB_EMIT_WAIT "s_waitcnt vmcnt(6)\n\t"
#else
#error "RB_EB must be 4 or 6"
#endif
            const uint32_t n_turns = ((total >> 2) + 63u) >> 6; // 64 groups per turn
            // address work of turn t: the group this lane loads, and where it goes
            auto plan = [&](uint32_t t, uint32_t &dl) -> uint32_t {
                const uint32_t o = (t * 64u + (uint32_t)lane) * 4u; // region offset of this lane's group in turn t
                while (o >= n_start) {                             // next clip (offsets are increasing)
                    cj++;
                    c_start = n_start;
                    c_fg = et[1 * RB_ET + cj];
                    c_lg = et[2 * RB_ET + cj];
                    n_start = et[cj + 1];
                }
                const uint32_t c = c_fg + (o - c_start); // coordinate of the group's first op
                const bool live = t < n_turns && o < total && c > c_fg && c < c_lg;
                dl = live ? o * 4u : 0xFFFFFFFFu;
                return (live ? c : c_fg) * 4u; // (dead lanes re-read their clip's first group, which is in L2, and store nothing)
            };
            // (one asm body for every turn, the drain turns load with an empty exec mask: two variants would make the
            //  compiler copy buffers between them while their loads are still in flight)
#define RB_EMIT_TURN(T, L, DL, S, DS)                                                                                          \
    {                                                                                                                          \
        const uint32_t t_ = (T);                                                                                               \
        const uint32_t src_ = plan(t_, DL);                                                                                    \
        const unsigned long long lmask_ = rb_first64(t_ < n_turns ? ~0ull : 0ull); /* (scalar register operand) */             \
        unsigned long long sv_;                                                                                                \
        asm volatile("s_mov_b64 %[sv], exec\n\t"                                                                               \
                     "s_and_b64 exec, %[sv], %[lmask]\n\t"                                                                     \
                     "global_load_dwordx4 %[p" #L "], %[src], %[sbase]\n\t"                                                     \
                     "s_mov_b64 exec, %[sv]\n\t" RB_EMIT_WAIT "v_cmpx_ne_u32_e32 vcc, -1, %[ds]\n\t"                           \
                     "global_store_dwordx4 %[ds], %[p" #S "], %[dbase]" RB_ST_NT "\n\t"                                          \
                     "s_mov_b64 exec, %[sv]"                                                                                   \
                     : [sv] "=&s"(sv_), RB_PIN_BUFS                                                                            \
                     : [src] "v"(src_), [sbase] "s"(gsrc), [ds] "v"(DS), [dbase] "s"(region), [lmask] "s"(lmask_)              \
                     : "vcc", "scc", "memory");                                                                                \
    }
            for (uint32_t t0 = 0; t0 < n_turns + (RB_EB - 1); t0 += RB_EB) { // RB_EB - 1 extra turns drain the pipeline
#if RB_EB == 6
                RB_EMIT_TURN(t0 + 0u, 0, d0, 1, d1)
                RB_EMIT_TURN(t0 + 1u, 1, d1, 2, d2)
                RB_EMIT_TURN(t0 + 2u, 2, d2, 3, d3)
                RB_EMIT_TURN(t0 + 3u, 3, d3, 4, d4)
                RB_EMIT_TURN(t0 + 4u, 4, d4, 5, d5)
                RB_EMIT_TURN(t0 + 5u, 5, d5, 0, d0)
#else
                RB_EMIT_TURN(t0 + 0u, 0, d0, 1, d1)
                RB_EMIT_TURN(t0 + 1u, 1, d1, 2, d2)
                RB_EMIT_TURN(t0 + 2u, 2, d2, 3, d3)
                RB_EMIT_TURN(t0 + 3u, 3, d3, 0, d0)
#endif
            }
#undef RB_EMIT_TURN
#undef RB_PIN_BUFS
#undef RB_EMIT_WAIT
            RB_PHASE(5)
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

// tpos_aln of a record whose target start is 0 and whose first ops consume no reference begins with units at
// t_pos = -1, i.e. u64::MAX (paf.rs:505, :531): the array is then NOT sorted and slice::binary_search returns whatever
// its probe sequence leads to.  This reproduces that probe sequence on the virtual array (value of a unit = walk of the
// ops), for both generations of the Rust standard library.  Returns true and the index on Ok, false on Err.
__device__ uint64_t rb_unit_tpos(const uint32_t *ops, uint32_t n, uint64_t t_st, uint64_t unit) {
    int64_t tpos = (int64_t)t_st - 1;
    uint64_t U = 0;
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t opc = rb_opc(ops[i]), len = rb_len(ops[i]);
        const bool isref = opc <= 8 && rb_in(RB_REF_MASK, opc);
        if (unit < U + len) return (uint64_t)(isref ? tpos + (int64_t)(unit - U) + 1 : tpos); // (-1 wraps to u64::MAX)
        U += len;
        if (isref) tpos += len;
    }
    return ~0ull;
}
__device__ bool rb_bsearch_units(const uint32_t *ops, uint32_t n, uint64_t t_st, uint64_t N, uint64_t key, int policy, uint64_t *idx) {
    auto cmp = [&](uint64_t mid) -> int {
        const uint64_t v = rb_unit_tpos(ops, n, t_st, mid);
        return v < key ? -1 : (v > key ? 1 : 0);
    };
    if (policy != RB_BSEARCH_LEGACY) { // rustc >= 1.82
        uint64_t size = N;
        if (size == 0) return false;
        uint64_t base = 0;
        while (size > 1) {
            const uint64_t half = size / 2, mid = base + half;
            base = cmp(mid) > 0 ? base : mid;
            size -= half;
        }
        *idx = base;
        return cmp(base) == 0;
    }
    uint64_t size = N, left = 0, right = N; // 1.52 .. 1.81
    while (left < right) {
        const uint64_t mid = left + size / 2;
        const int c = cmp(mid);
        if (c < 0) left = mid + 1;
        else if (c > 0) right = mid;
        else {
            *idx = mid;
            return true;
        }
        size = right - left;
    }
    return false;
}

__global__ __launch_bounds__(256) void rb_k_liftover_generic(rb_lift_params p) {
    const uint64_t n_gen = p.counters->n_generic;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_gen; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t hrow = p.gen_list[g];
        rb_hit_row *row = &p.rows[hrow];
        const uint32_t r = row->rec, win = row->win;
        const rb_norm_row *nr = &p.norm[r];
        if (nr->status != RB_ST_OK) { // fused scan: the record was handed back and the full scan found the reference would panic on it
            row->status = (uint16_t)nr->status;
            row->out_n = 0;
            row->out_off = 0;
            row->t_st = row->t_en = row->q_st = row->q_en = 0;
            row->nmatch = row->aln_len = 0;
            continue;
        }
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
        // units at t_pos = -1 in front of the first reference-consuming op (only possible with t_st == 0): the array is not
        // sorted, so the equal ranges do not tell what binary_search returns; its probe sequence is replayed instead
        bool wrapped = false;
        if (t_st == 0)
            for (uint32_t i = 0; i < n; i++) {
                if (rb_len(ops[i]) == 0) continue; // (a zero-length op adds no unit)
                const uint32_t opc = rb_opc(ops[i]);
                wrapped = !(opc <= 8 && rb_in(RB_REF_MASK, opc));
                break;
            }
        uint64_t ks, ke;
        if (wrapped) {
            if (!rb_bsearch_units(ops, n, t_st, N, (uint64_t)ps, p.policy, &ks) || !rb_bsearch_units(ops, n, t_st, N, (uint64_t)pe, p.policy, &ke)) {
                w.status = RB_ST_PANIC_NOTFOUND;
                *row = w;
                continue;
            }
        } else {
            if (!s_found || !e_found) { // binary_search Err -> panic (liftover.rs:31, :42)
                w.status = RB_ST_PANIC_NOTFOUND;
                *row = w;
                continue;
            }
            ks = p.policy == RB_BSEARCH_LEGACY ? rb_legacy_probe(N, s_lo, s_hi) : s_hi;
            ke = p.policy == RB_BSEARCH_LEGACY ? rb_legacy_probe(N, e_lo, e_hi) : e_hi;
        }
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

extern "C" hipError_t rb_launch_exclusive_scan(uint64_t *v, uint64_t n, uint64_t *block_sums, uint64_t *total_out, hipStream_t stream);
extern "C" hipError_t rb_launch_count_and_scan(const rb_lift_params *p, uint64_t *block_sums, bool do_count, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    if (do_count) {
        const unsigned blocks = (unsigned)((p->n_rec + 255) / 256);
        hipLaunchKernelGGL(rb_k_count_hits, dim3(blocks), dim3(256), 0, stream, *p);
    }
    return rb_launch_exclusive_scan(p->hit_off, p->n_rec, block_sums, &p->counters->n_hits, stream);
}
// in-place exclusive scan of n u64 counts (n + 1 outputs); block_sums: rb_scan_block_sums_count(n) words of scratch
extern "C" hipError_t rb_launch_exclusive_scan(uint64_t *v, uint64_t n, uint64_t *block_sums, uint64_t *total_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const uint64_t nb = (n + RB_SCAN_PER_BLOCK - 1) / RB_SCAN_PER_BLOCK;
    hipLaunchKernelGGL(rb_k_scan_partial, dim3((unsigned)nb), dim3(256), 0, stream, (const uint64_t *)v, n, block_sums);
    hipLaunchKernelGGL(rb_k_scan_top, dim3(1), dim3(256), 0, stream, block_sums, nb);
    hipLaunchKernelGGL(rb_k_scan_apply, dim3((unsigned)nb), dim3(256), 0, stream, v, n, (const uint64_t *)block_sums, total_out);
    return hipGetLastError();
}
extern "C" hipError_t rb_launch_make_jobs(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_make_jobs, dim3((unsigned)((p->n_rec + 255) / 256)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

extern "C" hipError_t rb_launch_liftover_stream(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0 || p->wave_end <= p->wave0) return hipSuccess;
    const unsigned blocks = (unsigned)(((uint64_t)(p->wave_end - p->wave0) + 3) / 4);
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
