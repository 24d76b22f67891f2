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
//                       bytes per lane, 2 KiB per step, two steps in flight in a register ring the compiler cannot
//                       see); per lane the reference / query / unit lengths of 8 ops are summed (op class -> mask
//                       with one v_bfe_i32), three 6-step DPP prefix scans give the running offsets, every second
//                       lane leaves a 16-op checkpoint in LDS; window boundaries are resolved lane-parallel against
//                       the checkpoints (lane j: start of window j, lane j + 32: its end).  Clips are emitted FROM
//                       THE LOAD RING while the record streams: output slot k mirrors the input's op positions
//                       (out_ops[k * slot_stride + 32 r + position]), clip j of a record goes to slot j mod n_slots,
//                       and a lane stores the 8 ops it has just loaded into every slot whose current clip its
//                       reference span touches -- no size is needed to place a clip, so there is no reservation,
//                       no atomic and no second read.  After the segment's resolution only the two end groups of a
//                       clip are patched (clipped first / last length).  Clips that would land within four lines of
//                       the previous clip of their slot, and clips of windows that are not sorted, are listed and
//                       copied by rb_k_copy_clips into the arenas behind the slots.
//   rb_k_liftover_generic_wave  one wavefront per hit the streaming kernel declines: three passes over the record's
//                       ops (boundaries, merge of adjacent runs through LDS, emission); rb_k_liftover_generic, the
//                       serial one-thread-per-hit form, stays as the diagnostic reference (RB_DEBUG_GENERIC_SERIAL).
//                       They take every case the streaming kernel declines
//                       (irregular CIGARs: N/S/H/P, zero lengths, adjacent ops of one type that must
//                       merge (paf.rs:602-620); the legacy binary-search policy when the duplicate
//                       choice matters; look-aheads / look-backs longer than RB_WALK_MAX ops).
//
// Roofline: HBM.  Algorithmic bytes: 4 B per input op + 48 B per record + 88 B per hit + 4 B per
// emitted op.  No MFMA: integer / index work only.
#include "rb_lift.h"
#include <type_traits>
#include <algorithm>

// diagnostics (debug_skip & 32): shader-clock time of each phase of a record, every 16th record, summed in units of 16
// cycles into counters->phase[0..4]: job + windows, stream + resolve, verdict + finalize, reservation, rows + end groups
#define RB_PHASE(i)                                                                                                  \
    if (dbg & 32) {                                                                                                  \
        const long long t_now = clock64();                                                                           \
        if (lane == 0 && (wave & 15) == 0) atomicAdd(&p.counters->phase[i], (uint32_t)((t_now - t_prev) >> 4));       \
        t_prev = t_now;                                                                                              \
    }

#ifndef RB_LIST_TU // (k_liftover_list.hip compiles only the per-record kernel's list form out of this file)
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
    const bool brk = p.brk_mode != 0; // one-walk break-paf: the pieces are not known yet (one pass per record, rows placed afterwards)
    const uint64_t h0 = brk ? 0ull : p.hit_off[k], nh = brk ? 1ull : p.hit_off[k + 1] - h0;
    const bool explicit_w = brk || p.x_st != nullptr;
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

#endif // !RB_LIST_TU
// ------------------------------------------------------------------------------------------------
// streaming kernel
// ------------------------------------------------------------------------------------------------
// RB_OPL: ops per lane and step.  8 ("pair", the product): a step is 512 ops, two 16-byte loads of 32 contiguous bytes per lane.
// 4 ("flat", a build switch): a step is 256 ops, ONE 16-byte load per lane -- every load and store instruction covers 1 KiB of whole
// 128-byte lines, which the memory-mix probe prices 6 % under the pair shape (bench.py box.probe_flat_ms) -- at twice the steps,
// i.e. twice the wave scans and scalar bookkeeping per op.
#ifndef RB_OPL
#define RB_OPL 8
#endif
static_assert(RB_OPL == 8 || RB_OPL == 4, "ops per lane and step: 8 or 4");
#define RB_GRP (RB_OPL / 4)                    // 16-byte groups per lane and step
#define RB_STEP_SHIFT (RB_OPL == 8 ? 9 : 8)    // log2 of the ops of a step
#define RB_CP_PER_STEP ((64 * RB_OPL) / RB_CP_OPS) // one checkpoint per RB_CP_OPS ops
#define RB_CP_LANES (RB_CP_OPS / RB_OPL)       // lanes that share a checkpoint (the first of them leaves it)
#ifndef RB_PF
#define RB_PF (RB_OPL == 8 ? 2 : 4) // steps of stream loads in flight per wave (4 KiB either way)
#endif
#ifndef RB_SEG
#define RB_SEG (RB_OPL == 8 ? 10 : 20)
#endif
#define RB_SMAX ((RB_SEG / RB_PF) * RB_PF) // steps whose checkpoints fit in LDS at once; whole turns of the load ring
#ifndef RB_WPE
#define RB_WPE 5, 6 // waves per SIMD the register budget is cut for
#endif
#ifdef RB_USE_NT
#define RB_ST_NT " nt"
#endif
#ifndef RB_ST_NT
#define RB_ST_NT "" // plain stores: nothing in this kernel is read twice any more, and non-temporal stores take 2 ms longer here
#endif
// vector-memory instructions a step of the streaming loop issues, always (stores whose mask is empty are issued with an empty
// exec mask: they move nothing but they count, tools/vmcnt_probe.hip, which keeps every s_waitcnt immediate exact)
#define RB_STEP_VMEM (RB_GRP * RB_MS + RB_GRP)
#ifndef RB_RING_WAIT
#define RB_RING_WAIT ((RB_PF - 1) * RB_STEP_VMEM)
#endif

typedef uint32_t rb_u32x4 __attribute__((ext_vector_type(4)));

// The load ring lives in VGPRs the compiler does not know: it allocates v0 .. v(RB_RING_BASE - 1) (amdgpu_num_vgpr), the ring is
// v[RB_RING_BASE .. RB_RING_BASE + 8 RB_PF), named literally in the asm statements that load, store and copy it out.  A ring the
// compiler can see gets copied between registers where two code paths meet (phi copies) -- harmless for ordinary values, fatal
// for registers with a load in flight, which no s_waitcnt of the compiler's covers.
#define RB_STR2(x) #x
#define RB_STR(x) RB_STR2(x)
#ifndef RB_RING_BASE
#define RB_RING_BASE 80
#endif
#if RB_RING_BASE == 80 && RB_PF == 2 && RB_OPL == 8
#define RB_RING_TOP_N 95
#elif RB_RING_BASE == 88 && RB_PF == 2
#define RB_RING_TOP_N 103
#elif RB_RING_BASE == 96 && RB_PF == 2
#define RB_RING_TOP_N 111
#elif RB_RING_BASE == 104 && RB_PF == 2
#define RB_RING_TOP_N 119
#elif RB_RING_BASE == 80 && RB_PF == 4 && RB_OPL == 4
#define RB_RING_TOP_N 95
#elif RB_RING_BASE == 96 && RB_PF == 4 && RB_OPL == 4
#define RB_RING_TOP_N 111
#elif RB_RING_BASE == 96 && RB_PF == 3
#define RB_RING_TOP_N 119
#elif RB_RING_BASE == 104 && RB_PF == 3
#define RB_RING_TOP_N 127
#else
#error "RB_RING_BASE / RB_PF: 80, 88, 96 or 104 with two slots; 96 or 104 with three"
#endif
#ifndef RB_RING_TOP
#define RB_RING_TOP "v" RB_STR(RB_RING_TOP_N) // the last register of the ring (RB_RING_BASE + 8 RB_PF - 1): named as a clobber so that the kernel's register count covers it
// (NOT all sixteen: a register named as clobbered is one the compiler may use for its own temporaries between two asm statements --
//  tried, it did.  What keeps the compiler out of the ring is amdgpu_num_vgpr, with one gap: the VGPRs it spills scalar registers into
//  are placed behind its own allocation, and one build of this kernel had them at v78 v79 v80.  tests/test_ring_registers.py
//  disassembles both builds and fails if anything outside the asm statements names v80..v95.)
#endif
#ifndef RB_SPILL_ROOM
#define RB_SPILL_ROOM 0 // registers between the compiler's allocation and the ring, for the VGPRs it parks spilled scalar registers in
#endif
// registers OFF .. OFF + W of the ring, as the assembler reads them (it evaluates the sums)
#define RB_RREG(OFF, W) "v[" RB_STR(RB_RING_BASE) "+" #OFF ":" RB_STR(RB_RING_BASE) "+" #OFF "+" #W "]"
// RB_RING_CASE(ring, M): M(<the slot's registers>) for the ring slot `ring` (a compile-time constant)
#if RB_OPL == 8
#define RB_RING_CASE(RING, M)                                                                                                   \
    if constexpr ((RING) == 0) { M(0, 2, 4, 6) } else if constexpr ((RING) == 1) { M(8, 10, 12, 14) } else { M(16, 18, 20, 22) }
#else
#define RB_RING_CASE(RING, M)                                                                                                   \
    if constexpr ((RING) == 0) { M(0, 2, 0, 0) } else if constexpr ((RING) == 1) { M(4, 6, 0, 0) } else if constexpr ((RING) == 2) { M(8, 10, 0, 0) } else { M(12, 14, 0, 0) }
#endif
static_assert((RB_OPL == 8 && (RB_PF == 2 || RB_PF == 3)) || (RB_OPL == 4 && RB_PF == 4), "the ring's asm statements are written out for two or three slots of eight registers, or four of four");
// BRK: break-paf in one walk (rb_lift.h, brk_max): the windows of a record are not given, they are the stretches between the indels
// longer than brk_max, found while the record streams; 32 pieces a pass.  The liftover build has none of that code.
// DIAG: the diagnostics build of the same kernel (bench.py --debug-skip: phases switched off, phase timers, clock stamps); the product
// launches DIAG = false, in which no stamp executes and no debug bit is looked at.
// LIST_WAVE: the schedule slot comes from the caller (rb_k_liftover_stream_list: the records the tile kernel handed back, k_tile.hip)
template <bool BRK, bool DIAG, bool LIST = false>
__device__ __forceinline__ void rb_stream_record(const uint64_t list_wave = 0) {
    // The 408 bytes of parameters are NOT read through `p_`: the compiler loads every by-value kernel argument a kernel uses in
    // its entry block and then carries -- spills -- those hundred scalar registers through the whole record (round 2: 233 SGPR
    // spills, parked in VGPRs right under the load ring).  `p.field` below reads the field from the kernel-argument segment where
    // it is used, through a pointer the compiler cannot see through (one s_load at that place); what the streaming loop needs is
    // copied into locals in front of it.
    const rb_kparams kp = (rb_kparams)__builtin_amdgcn_kernarg_segment_ptr();
    rb_kparams kq = rb_kp_here(kp); // the pointer of the current phase (set-up / after the stream of a pass): loads through it stay inside the phase
#define p (*kq)
    // diagnostics (bench.py --debug-skip: phases of the kernel switched off, phase timers, clock stamps) only in the DIAG
    // instantiation: every tested bit is a wave-uniform boolean, i.e. two scalar registers held through the whole record
    const int dbg = DIAG ? p.debug_skip : 0;
    // checkpoints: exclusive (R,Q,U) prefixes every 16 ops, SoA so that R can be binary-searched
    __shared__ uint32_t cp_all[4][3][RB_SMAX * RB_CP_PER_STEP];
    __shared__ uint32_t wx_all[4][RB_HMAX + 1]; // window indices of one pass over a window list that is not sorted
    const uint32_t wib = rb_first(threadIdx.x >> 6); // wave in block (told to the compiler as the wave-uniform value it is)
    uint64_t wave;
    if constexpr (LIST) {
        wave = list_wave;
    } else {
        wave = (uint64_t)p.wave0 + (uint64_t)blockIdx.x * 4u + wib;
        if (wave >= p.wave_end) return;
    }
    const int lane = rb_lane();
    long long t_prev = (dbg & 32) ? clock64() : 0;
    uint32_t *cpR = cp_all[wib][0], *cpQ = cp_all[wib][1], *cpU = cp_all[wib][2];
#ifndef RB_JOB_AHEAD
#define RB_JOB_AHEAD 0 // (build switch: touch the job of the wave that starts RB_JOB_AHEAD waves from now -- a multiple of 32, i.e. a workgroup of the same XCD)
#endif
#if RB_JOB_AHEAD
    uint32_t job_touch = 0; // (a load whose value nobody wants: the register stays reserved until the job's own load, issued behind it, has landed)
    if (wave + RB_JOB_AHEAD < p.wave_end) asm volatile("global_load_dword %0, %1, off" : "=v"(job_touch) : "v"(&p.jobs[wave + RB_JOB_AHEAD]) : "memory");
#endif
    const rb_job jb_ = p.jobs[wave]; // (uniform address: one 64-byte request)
    const uint32_t jflags = rb_first(jb_.flags);
#if RB_JOB_AHEAD
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(job_touch) : : "memory");
#endif
    if (jflags & RB_JOB_ROWS_OVERFLOW) { // rows do not fit: flag and leave (host retries with more room)
        if (lane == 0) p.counters->overflow = 1;
        return;
    }
    if (!(jflags & RB_JOB_VALID)) return;
    const uint32_t r = rb_first(jb_.r);
    const rb_norm_row *nr = &p.norm[r];
    const uint64_t h0 = rb_first(jb_.h0);
    const uint64_t nh = rb_first(jb_.nh);
    const bool explicit_w = BRK || p.x_st != nullptr;
    const bool mono = BRK || (jflags & RB_JOB_MONO) != 0;
    uint64_t ws = 0, we = 0;
    if (!explicit_w && (!mono || !(jflags & RB_JOB_REGULAR))) { // the contig's window slice: only the rare paths need it
        const uint32_t cg = p.contig[r];
        ws = p.cw_off[cg];
        we = p.cw_off[cg + 1];
    }
    // BRK: this record is not the one-walk path's: on the list it goes (rb_k_break_pieces finds its pieces, the generic kernel clips them)
    // (one store here: rb_k_break_list_declined collects the marked records afterwards -- the list and its counter would be two more
    //  pointers for this kernel to carry)
    auto brk_decline = [&]() {
        if (lane == 0) p.brk_off[r] = ~1ull;
    };
    if (!(jflags & RB_JOB_REGULAR)) { // window order does not matter on the fast path: resolution is per lane
        if (p.fused && lane == 0) { // (a provisional row that cannot take the fast path: the full scan completes it)
            const unsigned long long i = atomicAdd(p.pend_count, 1ull);
            p.pend_list[i] = r;
        }
        if constexpr (BRK) {
            brk_decline();
            return;
        }
        rb_defer_record(kp, r, nr, h0, nh, explicit_w, mono, ws, we, lane);
        return;
    }
    // the record's coordinates are needed in front of the stream of a pass (the boundaries' offsets) and behind it (the rows), not in
    // between: they are read again from the job behind the stream instead of being carried -- in eight scalar registers, round 2 --
    // through it (the job is 64 bytes at an address the whole wave shares)
    // (RB_COORD_RELOAD: the liftover build is better off with the round-2 form -- pinned to scalar registers --, the break build with
    //  the reload; what decides is where the compiler then parks its spilled scalar registers, tools/check_ring.py)
    auto sgpr64 = [](uint64_t v) -> uint64_t {
        uint32_t lo = rb_first((uint32_t)v), hi = rb_first((uint32_t)(v >> 32));
        asm volatile("" : "+s"(lo), "+s"(hi));
        return ((uint64_t)hi << 32) | lo;
    };
    constexpr bool coord_reload = BRK;
    uint64_t t_st = jb_.t_st, t_en = jb_.t_en, q_st = jb_.q_st, q_en = jb_.q_en;
    if constexpr (!coord_reload) t_st = sgpr64(t_st), t_en = sgpr64(t_en), q_st = sgpr64(q_st), q_en = sgpr64(q_en);
    const uint32_t n = rb_first(jb_.n);
    const uint64_t rec0 = rb_first64(jb_.rec0); // global index of the record's first kept op
    const uint32_t *rec_ops = p.ops + rec0;
    const uint64_t lo = rb_first(jb_.lo);
    uint64_t scan_pos = ws; // non-monotone window lists: next window of the slice to test
    // the stream starts on the 128-byte line that holds the record's first op: loads and stores of a step then cover whole lines
#ifdef RB_DBG_ALIGN4
    const uint64_t g0 = rec0 & ~3ull, gend = rec0 + n;
#else
    const uint64_t g0 = rec0 & ~31ull, gend = rec0 + n;
#endif
    const uint32_t n_steps = (uint32_t)((gend - g0 + (1u << RB_STEP_SHIFT) - 1u) >> RB_STEP_SHIFT);
    const int32_t head = (int32_t)(rec0 - g0); // 0..31 ops of the previous record (or nothing) in front of the record in step 0
    const uint32_t *__restrict__ gbase0 = p.ops + g0;
    const uint32_t first_boff = (uint32_t)head * 4u;                          // byte offset (from g0) of the record's first op
    const uint32_t last_boff = (uint32_t)(((gend - 1u) & ~3ull) - g0) * 4u;   // ... of the last 16-byte group that holds an op of it
    const uint32_t last_cboff = last_boff & ~(4u * RB_OPL - 1u);              // ... of the chunk (RB_OPL ops, one lane) with that group

    // ---- where clips go.  Output copy ("slot") k of the batch mirrors the input positions: the op at coordinate c of this
    //      record (c counted from g0, the first op of the record's first 128-byte line) lives at
    //      out_ops[k * slot_stride + 32 r + g0 + c].  The 32 r keeps the lines of neighbouring records apart (they share one in
    //      the input when a record does not start on a multiple of 32).  Clip j of the record goes to slot j mod n_slots, so a step of the stream can be stored
    //      from the registers it was loaded into, with the aligned address it was loaded from: no size is needed to place a
    //      clip, hence no reservation, no atomic and no second read of the ops. ----
    const uint32_t n_slots = (uint32_t)p.n_slots;
    uint32_t *const out_ops_ = p.out_ops;        // (locals of the streaming loop, see the top of the kernel)
    const uint64_t slot_stride_ = p.slot_stride;
    const uint32_t brk_max_ = BRK ? p.brk_max : 0u;
    const int policy_ = p.policy, early_exit_ = p.early_exit, desc_mode_ = p.desc_mode;
    const uint64_t slot_row0 = 32ull * r + g0; // out_ops index of coordinate 0 in slot 0
    // first coordinate a clip of class k may start at in a later pass: behind every clip the earlier passes put there
    uint32_t carry[RB_MS];
#pragma unroll
    for (int q = 0; q < RB_MS; q++) carry[q] = 0u;

    const bool fused = p.fused != 0;
    uint32_t rec_nmatch = 0, rec_aln_len = 0; // of the whole (normalised) record: taken from its row, or from the fused verification
    if (!fused) rec_nmatch = nr->nmatch, rec_aln_len = nr->aln_len;
    uint64_t n_items = BRK ? 1 : ((nh == 0 && fused) ? 1 : nh); // (a record no window overlaps is still streamed once, to verify it; BRK: set by the first pass)
    // BRK, across passes: where the scratch rows of the record begin, and the cut state (pieces closed, end of the last long indel)
    // at the start of the segment in which the next pass's first piece opens
    uint64_t brk_row0 = 0;
    uint32_t brk_res_cnt = 0, brk_res_pre = 0;
    // where a later pass may start streaming: the segment in which the previous pass resolved the start of its last window
    // (windows are sorted, so nothing of the next pass lies before it), with the running totals at that point
    uint32_t resume_seg = 0, resume_R = 0, resume_Q = 0, resume_U = 0;
    unsigned long long sv_exec;
    asm volatile("s_mov_b64 %0, exec" : "=s"(sv_exec));
    const uint32_t lane_boff = (uint32_t)lane * (4u * RB_OPL);
    // (chunks past the record's end are not loaded: lanes behind the last chunk re-read it, and the loads of steps
    //  behind the last one run with an empty exec mask)
// RB_LINE_ROUND (lanes of 32 bytes per granule: 2 = 64 bytes, 4 = 128 bytes, 0 = off): the speculative stores of a step are widened to
// whole granules -- a lane in front of a clip's first chunk or behind its last one, in the same granule, stores what it loaded too
// (the record's own neighbouring ops; a slot's lines are this record's alone and nobody reads a slot outside a clip) --, and no two
// clips of a slot may share a granule (RB_GRAN ops) instead of a 16-byte group.  Fewer lines written in part: a launch that writes
// none at all (diagnostics: this + no end ops) is 5 % shorter on a fast box and 11 % on a slow one, profiles/r04_stream_summary.md.
#ifndef RB_LINE_ROUND
#define RB_LINE_ROUND (RB_OPL == 8 ? 2 : 0) // 64-byte granules: -0.6 % on a fast box, -1.7 % on a slow one (r04_ab7, r04_hs8); 128 bytes: nothing
#endif
#if RB_LINE_ROUND
#define RB_GRAN (8 * RB_LINE_ROUND)
#else
#define RB_GRAN 4
#endif
#ifndef RB_LD_NT
#define RB_LD_NT "" // (build switch: " nt" marks the stream's loads non-temporal -- what is read once should not push the jobs, windows and rows out of L2)
#endif
#if RB_OPL == 8
#define RB_RING_LOAD_ASM(A, B_, C_, D_)                                                                                         \
    asm volatile("s_mov_b64 exec, %[lm]\n\t"                                                                                    \
         "global_load_dwordx4 " RB_RREG(A, 3) ", %[o], %[sb]" RB_LD_NT "\n\t"                                             \
         "global_load_dwordx4 " RB_RREG(C_, 3) ", %[o], %[sb] offset:16" RB_LD_NT "\n\t"                                  \
         "s_mov_b64 exec, %[sv]"                                                                                        \
         :                                                                                                              \
         : [o] "v"(lo_), [sb] "s"(gb_), [lm] "s"(lm_), [sv] "s"(sv_)                                                    \
         : "memory", RB_RING_TOP);
#else
#define RB_RING_LOAD_ASM(A, B_, C_, D_)                                                                                         \
    asm volatile("s_mov_b64 exec, %[lm]\n\t"                                                                                    \
         "global_load_dwordx4 " RB_RREG(A, 3) ", %[o], %[sb]" RB_LD_NT "\n\t"                                             \
         "s_mov_b64 exec, %[sv]"                                                                                        \
         :                                                                                                              \
         : [o] "v"(lo_), [sb] "s"(gb_), [lm] "s"(lm_), [sv] "s"(sv_)                                                    \
         : "memory", RB_RING_TOP);
#endif
#define RB_RING_LOAD(RING, STP)                                                                                                 \
    {                                                                                                                           \
        const uint32_t stp_ = (STP);                                                                                            \
        uint32_t lo_ = (stp_ << (RB_STEP_SHIFT + 2)) + lane_boff;                                                               \
        lo_ = lo_ < last_cboff ? lo_ : last_cboff;                                                                              \
        const uint32_t *const gb_ = gbase0; /* (named copies: a generic lambda does not capture what only an asm operand uses) */ \
        const unsigned long long sv_ = sv_exec;                                                                                 \
        const unsigned long long lm_ = stp_ < n_steps ? sv_ : 0ull;                                                             \
        RB_RING_CASE(RING, RB_RING_LOAD_ASM)                                                                                    \
    }
#define RB_RING_NOSTORES                                                                                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < RB_GRP * RB_MS; q_++)                                                               \
        asm volatile("s_mov_b64 exec, 0\n\tglobal_store_dword %0, %0, %1\n\ts_mov_b64 exec, %2" ::"v"(0u), "s"(gbase0), "s"(sv_exec) : "memory");
    // The first pass of a record streams it from its first step: the ring's first loads go out HERE, in front of the pass's window loads
    // (a chain of dependent loads of its own), not behind them -- one memory latency per record instead of two in front of the first step.
#ifndef RB_PRELOAD
#define RB_PRELOAD 1
#endif
    const bool preloaded = RB_PRELOAD && !(dbg & 4);
    if (preloaded) {
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) (see the stream)
        RB_RING_LOAD(0, 0u)
        { RB_RING_NOSTORES }
        RB_RING_LOAD(1, 1u)
#if RB_PF >= 3
        { RB_RING_NOSTORES }
        RB_RING_LOAD(2, 2u)
#endif
#if RB_PF >= 4
        { RB_RING_NOSTORES }
        RB_RING_LOAD(3, 3u)
#endif
    }
    for (uint64_t jb = 0; jb < n_items; jb += RB_HMAX) {
        uint32_t nb = BRK ? 0u : (uint32_t)((nh - jb) < RB_HMAX ? (nh - jb) : RB_HMAX); // (BRK: pieces of this pass known so far, the open one included)
        const bool validate = fused && jb == 0;
        // ---- per-hit setup: lanes j and j + 32 both look at window jb + j; lane j resolves its start
        //      boundary, lane j + 32 its end boundary; lane j then owns the row ----
        const uint32_t hl = (uint32_t)lane & 31u;
        bool own = hl < nb;
        bool mine = own && lane < 32;
        const bool is_start = lane < 32;
        uint64_t wst = 0, wen = 0;
        uint32_t win = (uint32_t)jb + hl; // (BRK: the piece's ordinal)
        if constexpr (!BRK) {
            const rb_pass_win pw = rb_pass_windows(kp, &wx_all[wib][0], explicit_w, mono, ws, we, lo, h0, jb, nb, t_st, t_en, scan_pos, lane);
            wst = pw.wst, wen = pw.wen, win = pw.win;
        }
        const bool inside = !BRK && own && (t_st > wst && t_en < wen); // liftover.rs:23-25 (a piece never contains its record)
        // D = (relative ref offset of the boundary base) + 1
        uint32_t D = is_start ? (uint32_t)((wst > t_st ? wst : t_st) - t_st) + 1u // liftover.rs:28
                              : (uint32_t)((wen < t_en ? wen : t_en) - t_st);     // (min(en,t_en) - 1 - t_st) + 1, :38-40
        bool need = own && !inside;
        // BRK: piece j is the stretch from the end of a long indel (liftover.rs:203-206: pre_tpos) to the start of the next one
        // (cur_tpos, :190-201), kept if it holds reference bases; lane j carries its start, lane j + 32 its end, both as D above.
        // Piece brk_cnt is OPEN: its start is known (brk_pre), its end not yet (D = ~0 keeps it in every test below).
        // A pass takes pieces jb .. jb + 31; the first pass streams the whole record and counts all of them, a later one starts at
        // the segment in which its first piece opened (the cut state of that segment's start comes with it) and stops when its
        // pieces are closed and resolved.
        uint32_t brk_cnt = 0, brk_pre = 0;
        uint32_t brk_seg_cnt = 0, brk_seg_pre = 0; // ... at the start of the current segment
        uint32_t brk_nx_cnt = 0, brk_nx_pre = 0;   // ... of the segment the next pass resumes at
        unsigned long long brk_def = 0ull;         // lanes whose D is final
        const uint32_t brk_j0 = (uint32_t)jb;
        if constexpr (BRK) {
            if (jb != 0) brk_cnt = rb_first(brk_res_cnt), brk_pre = rb_first(brk_res_pre);
            D = 0xFFFFFFFFu;
            if (brk_cnt >= brk_j0 && brk_cnt - brk_j0 < 32u) { // the open piece is one of this pass's
                D = (uint32_t)lane == brk_cnt - brk_j0 ? brk_pre + 1u : D;
                brk_def = 1ull << (brk_cnt - brk_j0);
                nb = brk_cnt - brk_j0 + 1u;
            }
            need = false;
        }
        rb_bres O;
        O.st = RB_S_UNRES;
        O.op = O.part = O.R = O.Q = O.U = 0;
        if (dbg & 32) { // (the window values must have arrived for the phase boundary to mean anything)
            asm volatile("s_waitcnt vmcnt(0)");
        }
        RB_PHASE(0)
        // speculative emission: sorted windows only (the clips of a class then follow one another along the record)
        const bool spec = n_slots != 0u && mono && nb != 0u && !desc_mode_ && !(dbg & 1);
        const bool any_inside = __ballot(inside && mine) != 0; // (a clip that is the whole record needs the whole stream)
        const bool resumable = mono && jb != 0 && !any_inside; // later passes start where the previous one found its last start, and stop when done

        // ---- stream the record, RB_SMAX steps per segment; resolve after each segment ----
        uint32_t Rb = 0, Qb = 0, Ub = 0; // running totals
        uint32_t seg_first = 0;
        if (resumable) seg_first = resume_seg, Rb = resume_R, Qb = resume_Q, Ub = resume_U;
        if constexpr (BRK) // (set inside the step lambda: told to the compiler as the wave-uniform values they are)
            seg_first = rb_first(seg_first), Rb = rb_first(Rb), Qb = rb_first(Qb), Ub = rb_first(Ub);
        uint32_t next_seg = seg_first, next_R = Rb, next_Q = Qb, next_U = Ub; // resume point for the pass after this one
        // fused verification (first pass), per lane: AND of the "regular op" masks, minimum op word (below 16: a zero length),
        // minimum of code XOR previous code (0: two adjacent ops of one type), maximum of the per-lane length sums (2^25 and
        // more: the 64-lane scans could leave 32 bits; handed back like a zero length)
        uint32_t v_reg = 0xFFFFFFFFu, v_minw = 0xFFFFFFFFu, v_adj = 0xFFFFFFFFu, v_maxsu = 0u;
        uint32_t v_carry = 0xFu;       // last op word of the previous step (code 15: equals nothing)
        unsigned long long v_utot = 0; // 64-bit sum of all lengths
        const bool streams = BRK || ((__ballot(need) != 0 || validate || (spec && any_inside)) && !(dbg & 4));
        // diagnostics (dbg & 128): the clock this kernel holds while it streams -- s_memtime counts shader cycles, s_memrealtime a
        // constant 100 MHz -- stamped around the streaming loop of every 16th record, summed into counters->phase[0] (64 cycles) and
        // phase[1] (10 ns); phase[2] counts the stamped records.  Nothing reads these words on the device.
        unsigned long long ck_c0 = 0, ck_r0 = 0;
        if (dbg & 128) ck_c0 = __builtin_amdgcn_s_memtime(), ck_r0 = __builtin_amdgcn_s_memrealtime();
        if (streams) {
            // The load ring and the speculative stores are written by hand.  vmcnt retires in issue order on gfx9 and counts
            // loads and stores together; left to the compiler, the wait for a step's loads would also wait for the stores of
            // the step before (it sees only its own loads between a load and its use).  Every step issues exactly
            // RB_STEP_VMEM vector-memory instructions (stores of an empty mask and loads past the record's end are issued
            // with an empty exec mask), so "all but the youngest (RB_PF - 1) * RB_STEP_VMEM" is exactly "this step's loads
            // have landed".  Store data is read when the store issues: a ring slot is reloaded right behind its stores.
            // (a wait the compiler knows about: whatever load it still tracks as pending on a register the ring is about to
            //  take would otherwise cost an s_waitcnt vmcnt(0) inside the loop, on every step)
            __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
            if (!(preloaded && jb == 0)) {
            RB_RING_LOAD(0, seg_first * RB_SMAX)
            { RB_RING_NOSTORES }
            RB_RING_LOAD(1, seg_first * RB_SMAX + 1u)
#if RB_PF >= 3
            { RB_RING_NOSTORES }
            RB_RING_LOAD(2, seg_first * RB_SMAX + 2u)
#endif
#if RB_PF >= 4
            { RB_RING_NOSTORES }
            RB_RING_LOAD(3, seg_first * RB_SMAX + 3u)
#endif
            }
            // Speculative emission, per slot ("class") q: the CURRENT clip of the class -- its index among the pass's clips and its span in
            // reference offsets [c_ds, c_de) --, wave-uniform.  The clips of a class follow one another along the record (sorted windows),
            // so a step looks at the current clip and moves on only when that clip ends inside the step (round 3 kept a window [j_lo, j_hi)
            // over all classes and ran a scalar loop over it in every step: 60 of the step's 140 scalar instructions).
            uint32_t c_j[RB_MS], c_ds[RB_MS], c_de[RB_MS];
            auto clip_fetch = [&](const int q) { // (D lives in lanes: lane j the start of clip j, lane 32 + j its end)
                const bool ok = c_j[q] < nb;
                const uint32_t jj = ok ? c_j[q] : 0u;
                const uint32_t ds_ = rb_readlane<uint32_t>(D, (int)jj), de_ = rb_readlane<uint32_t>(D, (int)(32u + jj));
                c_ds[q] = ok ? ds_ : 0xFFFFFFFFu; // (no clip: a span no chunk reaches)
                c_de[q] = ok ? de_ : 0xFFFFFFFFu;
            };
#pragma unroll
            for (int q = 0; q < RB_MS; q++) { // clip j of the pass is of class (jb + j) mod n_slots
                const uint32_t ns_ = n_slots ? n_slots : 1u;
                c_j[q] = (uint32_t)q < n_slots ? ((uint32_t)q + ns_ - (uint32_t)(jb % ns_)) % ns_ : 0x7FFFFFFFu;
                clip_fetch(q);
            }
            uint32_t Rseg = 0, Qseg = 0, Useg = 0, seg0 = 0;
            // one step of the stream; EDGE = the record's first or last step (ops of the neighbours around it)
            // FAST = what nearly every step is, told to the compiler as constants: an interior step (no op of a neighbouring record in
            // it, inside its segment) of a record's first pass with the fused verification and the speculative stores on.  The
            // general form tests each of these per step; where the two forms met in one loop body the compiler moved two dozen
            // values from one set of registers to another between steps.
            auto step = [&](auto ring_c, auto edge_c, auto fast_c, const uint32_t st, const uint32_t seg1) {
                constexpr int ring = decltype(ring_c)::value;
                constexpr bool edge = decltype(edge_c)::value;
                constexpr bool fast = decltype(fast_c)::value;
                static_assert(!(fast && edge), "a fast step is an interior step");
                const bool validate_s = fast ? true : validate, spec_s = fast ? true : spec, later_s = fast ? false : (jb != 0);
                // the step's 8 ops leave the ring for registers of the compiler's choosing once they have landed
                // (no memory clobber: the compiler takes an asm that may load for a load still in flight and waits vmcnt(0) at the first use of its outputs)
                unsigned long long a0, a1, a2 = 0, a3 = 0;
#if RB_OPL == 8
#define RB_RING_TAKE(A, B_, C_, D_)                                                                                             \
    asm volatile("s_waitcnt vmcnt(%4)\n\t"                                                                                      \
                 "v_mov_b64 %0, " RB_RREG(A, 1) "\n\tv_mov_b64 %1, " RB_RREG(B_, 1) "\n\tv_mov_b64 %2, " RB_RREG(C_, 1) "\n\tv_mov_b64 %3, " RB_RREG(D_, 1) \
                 : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3)                                                                       \
                 : "n"(RB_RING_WAIT));
#else
#define RB_RING_TAKE(A, B_, C_, D_)                                                                                             \
    asm volatile("s_waitcnt vmcnt(%2)\n\t"                                                                                      \
                 "v_mov_b64 %0, " RB_RREG(A, 1) "\n\tv_mov_b64 %1, " RB_RREG(B_, 1)                                             \
                 : "=v"(a0), "=v"(a1)                                                                                           \
                 : "n"(RB_RING_WAIT));
#endif
                RB_RING_CASE(ring, RB_RING_TAKE)
#undef RB_RING_TAKE
                unsigned long long msk[RB_MS];
#pragma unroll
                for (int q = 0; q < RB_MS; q++) msk[q] = 0ull;
                if (fast || st < seg1) { // (no break: the ring must be in the same state on every path)
                    const uint32_t w_all[8] = {(uint32_t)a0, (uint32_t)(a0 >> 32), (uint32_t)a1, (uint32_t)(a1 >> 32),
                                               (uint32_t)a2, (uint32_t)(a2 >> 32), (uint32_t)a3, (uint32_t)(a3 >> 32)};
                    uint32_t w[RB_OPL], c[RB_OPL]; // the lane's ops; what the verification looks at
#pragma unroll
                    for (int q = 0; q < RB_OPL; q++) w[q] = c[q] = w_all[q];
                    if (edge) {
                        // ops of the neighbouring records: zero for the sums (a 0-length M); for the verification an I / D of length
                        // 1 by position parity -- regular, alternating, and never equal to the match op a normalised record ends on
                        const int32_t idx0 = (int32_t)(st << RB_STEP_SHIFT) + lane * RB_OPL - head;
#pragma unroll
                        for (int q = 0; q < RB_OPL; q++) {
                            const bool ok = (uint32_t)(idx0 + q) < n; // (also the negative head indices)
                            c[q] = ok ? w[q] : ((q & 1) ? 0x12u : 0x11u);
                            w[q] = ok ? w[q] : 0u;
                        }
                    }
                    if (validate_s) {
                        const uint32_t prevw = rb_prev_lane(c[RB_OPL - 1], v_carry); // previous lane's last op; lane 0: the previous step's
                        v_carry = rb_readlane<uint32_t>(c[RB_OPL - 1], 63);
                        uint32_t rg[RB_OPL], x[RB_OPL];
#pragma unroll
                        for (int q = 0; q < RB_OPL; q++) {
                            rg[q] = (uint32_t)__builtin_amdgcn_sbfe((int)0x018F018Fu, c[q], 1u); // M I D N = X
                            x[q] = (c[q] ^ (q ? c[q - 1] : prevw)) & 15u;
                        }
                        auto min3 = [](uint32_t a, uint32_t b, uint32_t d) { const uint32_t t = a < b ? a : b; return t < d ? t : d; };
#pragma unroll
                        for (int q = 0; q < RB_OPL; q += 2) {
                            v_reg &= rg[q] & rg[q + 1];
                            v_minw = min3(v_minw, c[q], c[q + 1]);
                            v_adj = min3(v_adj, x[q], x[q + 1]);
                        }
                    }
                    // per-lane sums of the reference / query / unit lengths of 8 ops; regular records hold only
                    // M I D N = X, so "consumes the reference" = not I and "consumes the query" = not D / N: one
                    // v_bfe_i32 per class turns the op code (low bits of the word) into an all-ones / zero mask
                    uint32_t sr = 0, sq = 0, su = 0;
#pragma unroll
                    for (int q = 0; q < RB_OPL; q++) {
                        const uint32_t len = rb_len(w[q]);
                        sr += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFFDFFFDu, w[q], 1u);
                        sq += len & (uint32_t)__builtin_amdgcn_sbfe((int)0xFFF3FFF3u, w[q], 1u); // not D, not N
                        su += len;
                    }
                    const uint32_t ir = rb_wave_scan_incl(sr), iq = rb_wave_scan_incl(sq), iu = rb_wave_scan_incl(su);
                    if (RB_CP_LANES == 1 || ((uint32_t)lane & (RB_CP_LANES - 1u)) == 0u) { // checkpoint every RB_CP_OPS ops: the first of the lanes that share it
                        const uint32_t t = (st - seg0) * RB_CP_PER_STEP + (uint32_t)lane / RB_CP_LANES;
                        cpR[t] = Rb + ir - sr;
                        cpQ[t] = Qb + iq - sq;
                        cpU[t] = Ub + iu - su;
                    }
                    const uint32_t R0 = Rb;
                    Rb += rb_readlane<uint32_t>(ir, 63);
                    Qb += rb_readlane<uint32_t>(iq, 63);
                    Ub += rb_readlane<uint32_t>(iu, 63);
                    if (validate_s) {
                        v_maxsu = v_maxsu > su ? v_maxsu : su;
                        v_utot += rb_readlane<uint32_t>(iu, 63);
                    }
                    if constexpr (BRK) {
                        // long indels among my 8 ops (ops of the neighbouring records are zero words here: an M of length 0).  They are
                        // rare -- one step in six has one -- and are taken one by one, in op order, with wave-uniform arithmetic
                        bool lane_big = false;
#pragma unroll
                        for (int q = 0; q < RB_OPL; q++) {
                            const uint32_t cq = w[q] & 15u;
                            lane_big |= (cq == RB_OP_I || cq == RB_OP_D) && rb_len(w[q]) > brk_max_;
                        }
                        unsigned long long cm = __ballot(lane_big);
                        const bool had_big = cm != 0ull;
                        const uint32_t lane_r0 = R0 + ir - sr; // reference offset of my first op
                        while (cm) {
                            const int l = __builtin_ctzll(cm);
                            cm &= cm - 1ull;
                            uint32_t rx = rb_readlane<uint32_t>(lane_r0, l);
#pragma unroll
                            for (int q = 0; q < RB_OPL; q++) {
                                const uint32_t wq = rb_readlane<uint32_t>(w[q], l);
                                const uint32_t cq = wq & 15u, lq = rb_len(wq);
                                const uint32_t rlq = cq == RB_OP_I ? 0u : lq; // (regular records: M I D N = X)
                                if ((cq == RB_OP_I || cq == RB_OP_D) && lq > brk_max_) {
                                    if (rx > brk_pre) { // liftover.rs:191: the piece in front of the indel, if it holds reference bases
                                        const uint32_t li = brk_cnt - brk_j0; // (its lane in this pass; wraps far above 32 for earlier pieces)
                                        if (li < 32u) {
                                            D = (uint32_t)lane == 32u + li ? rx : D;
                                            brk_def |= 1ull << (32u + li);
                                        }
                                        brk_cnt++;
                                        if (brk_cnt == brk_j0 + 32u) // the next pass's first piece opens in this segment: it resumes here
                                            next_seg = seg0 / RB_SMAX, next_R = Rseg, next_Q = Qseg, next_U = Useg, brk_nx_cnt = brk_seg_cnt, brk_nx_pre = brk_seg_pre;
                                    }
                                    brk_pre = rx + rlq; // :203-206
                                    {
                                        const uint32_t li = brk_cnt - brk_j0;
                                        if (li < 32u) { // the next piece opens here (its lanes are rewritten if it turns out empty)
                                            D = (uint32_t)lane == li ? brk_pre + 1u : ((uint32_t)lane == 32u + li ? 0xFFFFFFFFu : D);
                                            brk_def |= 1ull << li;
                                            brk_def &= ~(1ull << (32u + li));
                                        }
                                    }
                                }
                                rx += rlq;
                            }
                        }
                        nb = brk_cnt < brk_j0 ? 0u : (brk_cnt - brk_j0 + 1u < 32u ? brk_cnt - brk_j0 + 1u : 32u);
                        if (had_big) { // pieces were closed / opened: the classes' current clips are read again
#pragma unroll
                            for (int q = 0; q < RB_MS; q++) clip_fetch(q);
                        }
                    }
                    if (spec_s) {
                        // Which of this lane's 8 ops a clip keeps is not known yet (boundaries are resolved per segment),
                        // but which 8-op chunks can hold ops of clip j is: those whose reference span [cR, cE) reaches
                        // from the clip's first base (D_start - 1) to its last one (D_end - 1); walking to the next /
                        // previous match op only shrinks a clip.  Those chunks are stored as they are -- what lies
                        // outside the clip is never read, and the two end groups are rewritten with the clipped
                        // lengths once the boundaries are resolved.
                        const uint32_t cR = R0 + ir - sr, cE = R0 + ir;
#pragma unroll
                        for (int q = 0; q < RB_MS; q++) {
                            unsigned long long mk = 0ull;
                            for (;;) {
                                mk |= rb_ballot(cR < c_de[q] && cE >= c_ds[q]);
                                if (c_de[q] > Rb) break; // the clip reaches past this step (or there is none): it stays the current one
                                c_j[q] += n_slots;       // it ends in this step: the class's next clip may begin in it
                                clip_fetch(q);
                                if (c_j[q] >= nb) break;
                            }
                            msk[q] = mk;
                        }
                    }
                }
                // ---- stores of this step (always 2 * RB_MS instructions), then the slot's next loads ----
                {
                    const uint32_t so = (st << (RB_STEP_SHIFT + 2)) + lane_boff;
                    const unsigned long long sv_st = sv_exec;
                    unsigned long long v0 = sv_st, v1 = sv_st;
                    if (edge) { // groups in front of the record's first and behind its last one are not this record's to write
                        v0 = rb_ballot(so + 16u > first_boff && so <= last_boff);
                        if (RB_GRP == 2) v1 = rb_ballot(so + 32u > first_boff && so + 16u <= last_boff);
                    }
#pragma unroll
                    for (int q = 0; q < RB_MS; q++) {
                        unsigned long long m0 = msk[q] & v0, m1 = msk[q] & v1;
                        if (later_s && msk[q] != 0ull) { // later passes stay clear of the end groups earlier passes have patched
                            const uint32_t c0 = (st << RB_STEP_SHIFT) + (uint32_t)lane * (uint32_t)RB_OPL;
                            m0 &= rb_ballot(c0 >= carry[q]);
                            m1 &= rb_ballot(c0 + 4u >= carry[q]);
                        }
#if RB_LINE_ROUND
                        static_assert(RB_OPL == 8 && (RB_LINE_ROUND == 2 || RB_LINE_ROUND == 4), "RB_LINE_ROUND: lanes (of 32 bytes) per granule, 64 or 128 bytes");
                        if constexpr (!BRK) { // whole granules of 64 / 128 bytes (2 / 4 lanes); break-paf's pieces lie op to op: its groups stay 16 bytes
                            unsigned long long q4 = (m0 | m1);
                            if (RB_LINE_ROUND >= 2) q4 = (q4 | (q4 >> 1)), q4 &= RB_LINE_ROUND == 2 ? 0x5555555555555555ull : ~0ull;
                            if (RB_LINE_ROUND == 4) q4 = (q4 | (q4 >> 2)) & 0x1111111111111111ull;
                            if (RB_LINE_ROUND >= 2) q4 |= q4 << 1;
                            if (RB_LINE_ROUND == 4) q4 |= q4 << 2;
                            const unsigned long long keep0 = m0 | ~msk[q], keep1 = m1 | ~msk[q]; // (what the edge / carry filters took away stays away)
                            // (a line of a slot belongs to one record -- slot_row0 --: lanes in front of the record's first op or behind its last
                            //  one, in a granule that holds an op of a clip, write what they loaded: a neighbour's ops, nobody's to read)
                            m0 = q4 & keep0;
                            m1 = q4 & keep1;
                        }
#endif
                        if (dbg & 64) m0 = m1 = 0ull; // diagnostics: everything but the stores themselves
                        const uint32_t *sb = out_ops_ + slot_row0 + (uint64_t)q * slot_stride_;
#if RB_OPL == 8
#define RB_RING_STORE(A, B_, C_, D_)                                                                                            \
    asm volatile("s_mov_b64 exec, %[m0]\n\t"                                                                                    \
                 "global_store_dwordx4 %[o], " RB_RREG(A, 3) ", %[sb]" RB_ST_NT "\n\t"                                          \
                 "s_mov_b64 exec, %[m1]\n\t"                                                                                    \
                 "global_store_dwordx4 %[o], " RB_RREG(C_, 3) ", %[sb] offset:16" RB_ST_NT "\n\t"                               \
                 "s_mov_b64 exec, %[sv]"                                                                                        \
                 :                                                                                                              \
                 : [m0] "s"(m0), [m1] "s"(m1), [o] "v"(so), [sb] "s"(sb), [sv] "s"(sv_st)                                       \
                 : "memory");
#else
#define RB_RING_STORE(A, B_, C_, D_)                                                                                            \
    asm volatile("s_mov_b64 exec, %[m0]\n\t"                                                                                    \
                 "global_store_dwordx4 %[o], " RB_RREG(A, 3) ", %[sb]" RB_ST_NT "\n\t"                                          \
                 "s_mov_b64 exec, %[sv]"                                                                                        \
                 :                                                                                                              \
                 : [m0] "s"(m0), [o] "v"(so), [sb] "s"(sb), [sv] "s"(sv_st)                                                     \
                 : "memory");
#endif
                        RB_RING_CASE(ring, RB_RING_STORE)
#undef RB_RING_STORE
                    }
                }
                RB_RING_LOAD(ring, st + RB_PF)
            };
#ifndef RB_FAST
#define RB_FAST 0 // (1 takes 84 vector registers: the ring then has to sit at v96 -- 4 waves per SIMD instead of 5 --, and on one box the two
                  //  builds ran 9.81 and 9.74 ms: profiles/r04_stream_summary.md.  Kept as a build switch, not the product)
#endif
            const bool fast_ok = RB_FAST && validate && spec && jb == 0 && !DIAG;
            for (seg0 = seg_first * RB_SMAX; seg0 < n_steps; seg0 += RB_SMAX) {
                const uint32_t seg1 = (seg0 + RB_SMAX < n_steps) ? seg0 + RB_SMAX : n_steps;
                Rseg = Rb, Qseg = Qb, Useg = Ub;
                if constexpr (BRK) brk_seg_cnt = brk_cnt, brk_seg_pre = brk_pre;
                // the ring is indexed statically (unrolled by RB_PF): rotating it with register moves would make
                // every step wait for ALL loads in flight (the moves read their destination registers)
                for (uint32_t st0 = seg0; st0 < seg1; st0 += RB_PF) {
#ifndef RB_DBG_ALL_EDGE
                    if (fast_ok && st0 != 0u && st0 + RB_PF < n_steps) { // every step of the turn is interior (and inside the segment: RB_SMAX is whole turns)
                        rb_static_for<RB_PF>([&](auto ring_c) { step(ring_c, std::false_type{}, std::true_type{}, st0 + (uint32_t)decltype(ring_c)::value, seg1); });
                        continue;
                    }
#endif
                    rb_static_for<RB_PF>([&](auto ring_c) {
                        const uint32_t st = st0 + (uint32_t)decltype(ring_c)::value;
#ifdef RB_DBG_ALL_EDGE
                        if (true) step(ring_c, std::true_type{}, std::false_type{}, st, seg1);
#else
                        if (st == 0 || st + 1 >= n_steps) step(ring_c, std::true_type{}, std::false_type{}, st, seg1);
#endif
                        else step(ring_c, std::false_type{}, std::false_type{}, st, seg1);
                    });
                }
                // ---- lane-parallel resolution of the boundaries that fall in this segment ----
                const bool last_seg = seg1 == n_steps;
                const uint32_t n_cp = (seg1 - seg0) * RB_CP_PER_STEP;
                const int32_t cp_idx0 = (int32_t)(seg0 << RB_STEP_SHIFT) - head; // op index of checkpoint 0
                if constexpr (BRK) {
                    if (last_seg) { // liftover.rs:213-224: what lies behind the last long indel
                        const uint32_t li = brk_cnt - brk_j0;
                        if (Rb > brk_pre) {
                            if (li < 32u) {
                                D = (uint32_t)lane == 32u + li ? Rb : D;
                                brk_def |= 1ull << (32u + li);
                            }
                            brk_cnt++;
                        } else if (li < 32u) {
                            brk_def &= ~(1ull << li); // the record ends with a long indel: no piece was open after all
                        }
                    }
                    need = ((brk_def >> lane) & 1ull) != 0ull && O.st == RB_S_UNRES;
                }
                {
                    const bool todo = need && D >= Rseg && (D < Rb || (last_seg && D == Rb));
                    if (todo && !(dbg & 2)) {
                        if (D == Rb) { // boundary on the record's last base; the last op is match-type
                            const uint32_t lv = rec_ops[n - 1];
                            O.st = RB_S_OK, O.op = n - 1;
                            if (is_start) O.part = rb_part_pack(1u, lv), O.R = Rb - 1, O.Q = Qb - 1, O.U = Ub - 1;
                            else O.part = rb_part_pack(rb_len(lv), lv), O.R = Rb, O.Q = Qb, O.U = Ub;
                        } else {
                            // last checkpoint with R <= D (R is non-decreasing)
                            uint32_t lo_t = 0, hi_t = n_cp;
                            while (hi_t - lo_t > 1) {
                                const uint32_t mid = (lo_t + hi_t) >> 1;
                                if (cpR[mid] <= D) lo_t = mid; else hi_t = mid;
                            }
                            O = rb_resolve(rec_ops, n, cp_idx0 + (int32_t)lo_t * RB_CP_OPS, cpR[lo_t], cpQ[lo_t], cpU[lo_t], D, is_start, policy_);
                        }
                        need = false;
                    }
                    // (the group loads above may still be tracked as pending where a lane left rb_resolve early; the compiler
                    //  would then wait vmcnt(0) at the first write to one of their registers, which is inside the step loop.
                    //  The resolving lanes have waited for those loads -- the youngest in the queue -- anyway.)
                    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0)
                    // the start of the pass's last window was found in this segment: the next pass begins here
                    if (!BRK && nb && ((__ballot(todo) >> (nb - 1u)) & 1ull)) next_seg = seg0 / RB_SMAX, next_R = Rseg, next_Q = Qseg, next_U = Useg;
                }
                if (!BRK && (early_exit_ || resumable) && !validate && !(spec && any_inside) && __ballot(need) == 0) break;
                if constexpr (BRK) { // a later pass is done when its 32 pieces are closed and every boundary of theirs is resolved
                    if (jb != 0 && brk_cnt >= brk_j0 + 32u && __ballot(((brk_def >> lane) & 1ull) != 0ull && O.st == RB_S_UNRES) == 0ull) break;
                }
            }
            // nothing of the ring may still be in flight when its registers go back to the compiler (a pass that leaves early
            // has loads out), and the end groups below must land after the speculative stores to the same addresses
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef RB_RING_LOAD
#undef RB_RING_LOAD_ASM
#undef RB_RING_NOSTORES
        }
        if (!streams && preloaded && jb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the ring's loads are out: nothing of them may be in flight when the registers are the compiler's again)
        if (dbg & 128) {
            const unsigned long long ck_c1 = __builtin_amdgcn_s_memtime(), ck_r1 = __builtin_amdgcn_s_memrealtime();
            if (lane == 0 && (wave & 15) == 0) {
                atomicAdd(&p.counters->phase[0], (uint32_t)((ck_c1 - ck_c0) >> 6));
                atomicAdd(&p.counters->phase[1], (uint32_t)(ck_r1 - ck_r0));
                atomicAdd(&p.counters->phase[2], 1u);
            }
        }
        resume_seg = next_seg, resume_R = next_R, resume_Q = next_Q, resume_U = next_U;
        kq = rb_kp_here(kp); // (what the rows, patches and lists below need is loaded from here on, not carried through the stream)
        // ---- behind the stream.  What this part needs to know about the record it reads again from the job (64 bytes at an address
        //      the whole wave shares, L2-resident) and derives afresh: computed in front of the stream these values -- row and slot
        //      addresses, flags, counts -- were carried through it in scalar registers the streaming loop has no room for, i.e.
        //      spilled (round 2: 233 spills) ----
        const rb_job jp = p.jobs[wave];
        if constexpr (coord_reload) t_st = jp.t_st, t_en = jp.t_en, q_st = jp.q_st, q_en = jp.q_en;
        {
        const uint32_t r = rb_first(jp.r), n = rb_first(jp.n);
        const rb_norm_row *nr = &p.norm[r];
        const uint64_t rec0 = rb_first64(jp.rec0);
        const uint32_t *rec_ops = p.ops + rec0;
        const int32_t head = (int32_t)(rec0 & 31ull);
        const uint64_t slot_row0 = 32ull * r + (rec0 & ~31ull);
        const uint64_t h0 = rb_first(jp.h0), nh = rb_first(jp.nh);
        const bool minus = (rb_first(jp.flags) & RB_JOB_MINUS) != 0;

        RB_PHASE(1)
        if (validate) {
            // ---- the verdict of the fused scan: check_integrity (paf.rs:825-857) on the normalised record and the
            //      conditions of the fast path.  A record that fails any of them is handed back: the full record scan
            //      (list mode) decides its status, the generic kernel clips it if it is merely irregular ----
            const bool lane_bad = v_reg != 0xFFFFFFFFu || v_minw < 16u || v_adj == 0u || (v_maxsu >> 25) != 0u;
            const bool bad = __ballot(lane_bad) != 0 || rb_first64(v_utot) > 0xFFFFFFFFull || t_en < t_st || q_en < q_st ||
                             (uint64_t)Rb != t_en - t_st || (uint64_t)Qb != q_en - q_st;
            if (bad) {
                if (lane == 0) {
                    const unsigned long long i = atomicAdd(p.pend_count, 1ull);
                    p.pend_list[i] = r;
                }
                if constexpr (BRK) {
                    brk_decline();
                    return;
                }
                if (!BRK && nh) {
                    if (!explicit_w) {
                        const uint32_t cg = p.contig[r];
                        ws = p.cw_off[cg];
                        we = p.cw_off[cg + 1];
                    }
                    rb_defer_record(kp, r, nr, h0, nh, explicit_w, mono, ws, we, lane);
                }
                return;
            }
            rec_nmatch = Rb + Qb - Ub; // match units = ref + query - all (M I D = X only)
            rec_aln_len = Ub;
            if (lane == 0) { // (RB_F_HAS_M is reported by rb_dev_scan_records only: nothing on this path reads it)
                rb_norm_row *w = &p.norm_w[r];
                w->nmatch = rec_nmatch;
                w->aln_len = rec_aln_len;
                w->flags = (nr->flags & RB_F_STRIPPED) | RB_F_REGULAR;
            }
            if (!BRK && nh == 0) { // (no window overlaps the record: it has been walked and verified, liftover.rs:119-121, and that is all)
                if ((dbg & 256) && lane == 0) p.diag_stamps[wave] = (uint32_t)__builtin_amdgcn_s_memrealtime();
                return;
            }
        }
        if constexpr (BRK) {
            brk_cnt = rb_first(brk_cnt), brk_nx_cnt = rb_first(brk_nx_cnt), brk_nx_pre = rb_first(brk_nx_pre);
            if (jb == 0) {
                // the first pass has seen every piece.  Rows: a place for all of them from one of the bump cursors (one atomic per
                // record; a single cursor would serialise the records at one L2 line), the count for the scan that orders the rows
                n_items = brk_cnt;
                unsigned long long b0 = 0;
                const uint32_t ar = (uint32_t)(wave % p.brk_n_arena);
                if (lane == 0 && brk_cnt) b0 = atomicAdd(&p.brk_cursor[(size_t)ar * 16u], (unsigned long long)brk_cnt);
                b0 = rb_first64(b0);
                if (b0 + brk_cnt > p.brk_arena_cap) { // (this cursor's share of the scratch rows is used up: rb_k_finish asks for more rows)
                    if (lane == 0) p.counters->brk_scratch_short = 1, p.hit_off[r] = brk_cnt, p.brk_off[r] = ~0ull;
                    return;
                }
                brk_row0 = (uint64_t)ar * p.brk_arena_cap + b0;
                if (lane == 0) p.hit_off[r] = brk_cnt, p.brk_off[r] = brk_row0;
                if (brk_cnt == 0) return;
            }
            nb = (uint32_t)(n_items - jb < 32u ? n_items - jb : 32u);
            own = hl < nb, mine = own && lane < 32;
            brk_res_cnt = brk_nx_cnt, brk_res_pre = brk_nx_pre;
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
        if constexpr (BRK) { // a boundary only the generic kernel resolves (it wants the piece's window in its row's place): the whole record goes
            if (__ballot(mine && defer) != 0ull) {
                brk_decline();
                return;
            }
        }
        const bool emits = mine && !defer && status == RB_ST_OK && !desc_mode_;
        const uint32_t e_first = (uint32_t)head + a_op; // coordinate (op index + head, counted from the aligned g0) of the first op
        const uint32_t e_cnt = emits ? out_n : 0u;
        const uint32_t eg_last = e_first + e_cnt - 1u;
        constexpr uint32_t gran = BRK ? 4u : (uint32_t)RB_GRAN; // ops per group no two clips of a slot may share
        const uint32_t eg_f = e_first & ~(gran - 1u), eg_l = e_cnt ? (eg_last & ~(gran - 1u)) : eg_f;
        // ---- which clips own their place in a slot.  Clip j (class j mod n_slots) does when it starts behind the last group
        //      of every earlier clip of its class: then no two clips of a slot share a 16-byte group, and the groups a clip
        //      rewrites (its first and last) are nobody else's.  With windows that overlap at most n_slots deep that is every
        //      clip; the others are copied to the arena area by rb_k_copy_clips. ----
        const uint32_t ns1 = n_slots ? n_slots : 1u;
        const uint32_t cls = (uint32_t)((jb + hl) % ns1);
        const uint32_t lgp = (emits && e_cnt) ? eg_l + gran : 0u; // first coordinate behind my clip's last group (0: no clip)
        uint32_t pm = lane < 32 ? lgp : 0u;                      // inclusive prefix maximum over the lanes of my class
        for (uint32_t d = ns1; d < 32u; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)pm, d, 64);
            if (hl >= d && lane < 32) pm = pm > t ? pm : t;
        }
        uint32_t before = (uint32_t)__shfl_up((int)pm, ns1, 64); // ... of the earlier clips of my class in this pass
        if (hl < ns1) before = 0u;
        uint32_t cprev = 0u;                                     // ... and of the passes before
#pragma unroll
        for (int q = 0; q < RB_MS; q++) cprev = (cls == (uint32_t)q) ? carry[q] : cprev;
        before = before > cprev ? before : cprev;
        const bool in_slot = spec && streams && emits && e_cnt != 0u && eg_f >= before;
        const bool copied = emits && !in_slot; // (an empty clip cannot happen: out_n >= 1 for an OK row)
        if (jb + RB_HMAX < (BRK ? n_items : nh)) { // another pass follows: what it must stay behind
#pragma unroll
            for (int q = 0; q < RB_MS; q++) {
                uint32_t v = (lane < 32 && cls == (uint32_t)q) ? pm : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
                    v = v > o ? v : o;
                }
                carry[q] = carry[q] > v ? carry[q] : rb_first(v);
            }
        }
        // the two end groups are re-read (the record has just been streamed: L2 or Infinity Cache)
        // (unconditional: a conditional load is sunk to its use; the ops array is padded, a quad may reach past the record)
        [[maybe_unused]] const uint32_t *__restrict__ gsrc = rec_ops - head; // coordinate c -> gsrc[c]; 16-byte aligned at c % 4 == 0
        // (diagnostics, dbg & 512: what the end groups cost -- their loads go to the record's first line, their stores are left out)
#ifndef RB_PATCH_WORDS
#define RB_PATCH_WORDS 1 // the clip's first and last op are written as two words made from what the resolution left in registers (0: the
                         // round-2 form -- their 16-byte groups read again from the record, patched, written back)
#endif
#if !RB_PATCH_WORDS && !(RB_GRAN > 4 && RB_GRAN_PATCH)
        static_assert(RB_GRAN == 4, "the 16-byte end groups are the geometry of RB_GRAN = 4");
        uint4 eg_q0 = *reinterpret_cast<const uint4 *>(gsrc + ((dbg & 512) ? 0u : eg_f));
        uint4 eg_q1 = *reinterpret_cast<const uint4 *>(gsrc + ((dbg & 512) ? 0u : eg_l));
        __builtin_amdgcn_sched_barrier(0);
#endif
        RB_PHASE(2)
        const uint64_t my_off = (uint64_t)cls * slot_stride_ + slot_row0 + e_first; // out_ops index of my clip's first op
        // (the row index is formed from an opaque copy of the lane id: otherwise the compiler hoists the row addresses above
        //  the streaming loop and carries -- or spills -- them through it)
        uint32_t lane_late = (uint32_t)lane;
        asm volatile("" : "+v"(lane_late));
        const uint64_t my_row = (BRK ? brk_row0 + jb : h0 + jb) + lane_late;
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
                w.flags = (inside ? RB_HIT_INSIDE : 0) | ((desc_mode_ && status == RB_ST_OK) ? RB_HIT_DESCRIPTOR : 0);
                w.out_n = status == RB_ST_OK ? out_n : 0;
                w.t_st = o_tst;
                w.t_en = o_ten;
                w.q_st = o_qst;
                w.q_en = o_qen;
                w.nmatch = o_nm;
                w.aln_len = o_al;
                w.out_off = status == RB_ST_OK ? (desc_mode_ ? 4ull * my_row : (in_slot ? my_off : 0ull)) : 0; // (copied clips: rb_k_copy_clips fills it in)
                *row = w;
                if (desc_mode_ && status == RB_ST_OK) // which ops of the ORIGINAL cigar the clip keeps
                    *reinterpret_cast<uint4 *>(out_ops_ + 4ull * my_row) =
                        make_uint4(nr->first_op + a_op, out_n, inside ? 0u : rb_part(A.part), inside ? 0u : rb_part(B.part));
            }
        }
        RB_PHASE(3)
        // ---- the end groups: the lane that owns a clip writes the group(s) holding its first and last op with the clipped
        //      lengths patched in; slots of those groups outside the clip are written as zeros ----
#ifndef RB_GRAN_PATCH
#define RB_GRAN_PATCH 0 // (1, with RB_LINE_ROUND: the end ops are patched by rewriting their whole granules -- measured: the bigger the
                        //  patch, the slower, 9.5 / 10.3 / 11.6 ms for 16 / 64 / 128 bytes on one box; kept for the record)
#endif
#if RB_GRAN > 4 && RB_GRAN_PATCH
        // whole granules: the granule that holds the clip's first op and the one that holds its last op are read again in full and
        // written in full, the two ops patched; what lies in them outside the clip is written as it was read (the slot's lines are
        // this record's alone, and nobody reads a slot outside a clip)
        if (in_slot && !(dbg & (1 | 512))) {
            uint32_t *__restrict__ dst = out_ops_ + (uint64_t)cls * slot_stride_ + slot_row0; // coordinate 0
            const uint32_t c_lastq = ((uint32_t)head + n - 1u) & ~3u; // the last group that holds an op of the record (nothing behind it is read)
            auto granule = [&](const uint32_t gc0) {
#pragma nounroll // (64 bytes at a time: a 128-byte granule in one go takes 32 registers, and the compiler then reaches into the ring)
                for (uint32_t gc = gc0; gc < gc0 + (uint32_t)RB_GRAN; gc += 16u) {
                    rb_u32x4 v[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const uint32_t c = gc + 4u * (uint32_t)k;
                        v[k] = *reinterpret_cast<const rb_u32x4 *>(gsrc + (c < c_lastq ? c : c_lastq));
                    }
                    if (!inside) {
#pragma unroll
                        for (int k = 0; k < 4; k++) {
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                const uint32_t c = gc + 4u * (uint32_t)k + (uint32_t)q;
                                uint32_t w = v[k][q];
                                if (e_cnt == 1u) { // the middle of one op
                                    if (c == e_first) w = ((rb_part(A.part) + rb_part(B.part) - rb_len(w)) << 4) | rb_opc(w);
                                } else {
                                    if (c == e_first) w = (rb_part(A.part) << 4) | rb_opc(w); // first op keeps its tail
                                    if (c == eg_last) w = (rb_part(B.part) << 4) | rb_opc(w); // last op keeps its head
                                }
                                v[k][q] = w;
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; k++) __builtin_nontemporal_store(v[k], reinterpret_cast<rb_u32x4 *>(dst + gc + 4u * (uint32_t)k));
                }
            };
            granule(eg_f);
            if (eg_l != eg_f) granule(eg_l);
        }
#elif RB_PATCH_WORDS
        // ---- the end ops: the speculative stores put the record's ops there as they are; the clip's first op keeps its tail, its last op
        //      its head.  Both words are made from what the resolution of the two boundaries left in registers (clipped length and op code:
        //      rb_bres.part) -- nothing is read again.  A clip that is the middle of ONE op holds B.U - A.U units of it. ----
        if (in_slot && !inside && !(dbg & (1 | 512))) {
            uint32_t *__restrict__ dst = out_ops_ + (uint64_t)cls * slot_stride_ + slot_row0; // coordinate 0
            if (e_cnt == 1u) {
                __builtin_nontemporal_store(((B.U - A.U) << 4) | (A.part >> 28), dst + e_first);
            } else {
                __builtin_nontemporal_store(rb_part_word(A.part), dst + e_first);
                __builtin_nontemporal_store(rb_part_word(B.part), dst + eg_last);
            }
        }
#else
        if (in_slot && !(dbg & (1 | 512))) {
            uint32_t *__restrict__ dst = out_ops_ + (uint64_t)cls * slot_stride_ + slot_row0 + eg_f; // the group holding coordinate eg_f
            uint32_t q0[4] = {eg_q0.x, eg_q0.y, eg_q0.z, eg_q0.w}, q1[4] = {eg_q1.x, eg_q1.y, eg_q1.z, eg_q1.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t c0 = eg_f + (uint32_t)q, c1 = eg_l + (uint32_t)q;
                uint32_t w0 = q0[q], w1 = q1[q];
                if (!inside) {
                    if (e_cnt == 1u) { // the middle of one op
                        if (c0 == e_first) w0 = ((rb_part(A.part) + rb_part(B.part) - rb_len(w0)) << 4) | rb_opc(w0);
                    } else {
                        if (c0 == e_first) w0 = (rb_part(A.part) << 4) | rb_opc(w0); // first op keeps its tail
                        if (c0 == eg_last) w0 = (rb_part(B.part) << 4) | rb_opc(w0); // (clips of 2..4 ops inside one group)
                        if (c1 == eg_last) w1 = (rb_part(B.part) << 4) | rb_opc(w1); // last op keeps its head
                    }
                }
                q0[q] = (c0 < e_first || c0 > eg_last) ? 0u : w0;
                q1[q] = (c1 > eg_last) ? 0u : w1;
            }
            const rb_u32x4 s0 = {q0[0], q0[1], q0[2], q0[3]}, s1 = {q1[0], q1[1], q1[2], q1[3]};
#ifdef RB_PATCH_PLAIN
            *reinterpret_cast<rb_u32x4 *>(dst) = s0;
            if (eg_l != eg_f) *reinterpret_cast<rb_u32x4 *>(dst + (eg_l - eg_f)) = s1;
#else
            __builtin_nontemporal_store(s0, reinterpret_cast<rb_u32x4 *>(dst));
            if (eg_l != eg_f) __builtin_nontemporal_store(s1, reinterpret_cast<rb_u32x4 *>(dst + (eg_l - eg_f)));
#endif
        }
#endif
        // ---- clips without a place of their own: one list entry each, copied by rb_k_copy_clips ----
        {
            const unsigned long long cm = __ballot(copied);
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
        RB_PHASE(4)
        } // (behind the stream)
    }
    if ((dbg & 256) && lane == 0) p.diag_stamps[wave] = (uint32_t)__builtin_amdgcn_s_memrealtime();
}
#undef p
// the builds of the kernel (an attribute cannot depend on a template parameter): liftover, break-paf in one walk, and the
// diagnostics build of the liftover form
#define RB_STREAM_KERNEL(NAME, BRK, DIAG, ROOM)                                                                                   \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RB_WPE), amdgpu_num_vgpr(RB_RING_BASE - (ROOM)))) void NAME(rb_lift_params p_) { \
        (void)p_; /* (read through the kernel-argument segment, see the top of rb_stream_record) */                                \
        rb_stream_record<BRK, DIAG>();                                                                                            \
    }
#ifndef RB_LIST_TU
RB_STREAM_KERNEL(rb_k_liftover_stream, false, false, RB_SPILL_ROOM)
RB_STREAM_KERNEL(rb_k_liftover_stream_brk, true, false, RB_SPILL_ROOM)
RB_STREAM_KERNEL(rb_k_liftover_stream_diag, false, true, 4)
#endif
#ifdef RB_LIST_TU
// ... and over a LIST of records (fb_list: the records of the tiles k_tile.hip did not take), workgroups that stay and take entry after entry
#define RB_LIST_BLOCKS 2560u // workgroups of the list form (twice what the chip holds at five per CU: entries differ in length)
// (the loop's state is ONE vector register -- the entry index, kept opaque --: everything else is read again from the kernel-argument
//  segment per entry.  Scalar registers carried around the record's body are spilled, and this build's spills reached into the ring)
#define RB_STREAM_LIST_KERNEL(NAME, BRK)                                                                                          \
    __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RB_WPE), amdgpu_num_vgpr(RB_RING_BASE - RB_SPILL_ROOM))) void NAME(rb_lift_params p_) { \
        (void)p_;                                                                                                                 \
        const rb_kparams kp_ = (rb_kparams)__builtin_amdgcn_kernarg_segment_ptr();                                                \
        uint32_t i_v = blockIdx.x * 4u + (threadIdx.x >> 6);                                                                      \
        for (;;) {                                                                                                                \
            asm volatile("" : "+v"(i_v));                                                                                         \
            const rb_kparams kl_ = rb_kp_here(kp_);                                                                               \
            const uint32_t i_ = rb_first(i_v);                                                                                    \
            if ((unsigned long long)i_ >= *kl_->fb_count) break;                                                                  \
            const uint64_t w_ = rb_first(kl_->slot_of[rb_first(kl_->fb_list[i_])]);                                               \
            rb_stream_record<BRK, false, true>(w_);                                                                               \
            i_v += RB_LIST_BLOCKS * 4u;                                                                                           \
        }                                                                                                                         \
    }
RB_STREAM_LIST_KERNEL(rb_k_liftover_stream_list, false)
RB_STREAM_LIST_KERNEL(rb_k_liftover_stream_brk_list, true)
extern "C" hipError_t rb_launch_liftover_stream_list(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0 || p->n_tiles == 0) return hipSuccess;
    if (p->brk_mode) hipLaunchKernelGGL(rb_k_liftover_stream_brk_list, dim3(RB_LIST_BLOCKS), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(rb_k_liftover_stream_list, dim3(RB_LIST_BLOCKS), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
#endif // RB_LIST_TU
#ifndef RB_LIST_TU
// (no diagnostics build of the break form: its spilled scalar registers land in the ring -- tools/check_ring.py --, and nothing asks for it)

// ------------------------------------------------------------------------------------------------
// clips that found no place of their own in a slot (windows overlapping deeper than the slots, window lists that are
// not sorted, no room for slots in out_ops): one wavefront per clip copies it into the arena area
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rb_k_copy_clips(rb_lift_params p) {
    const uint64_t n_copy = *p.copy_count;
    const int lane = rb_lane();
    for (uint64_t e = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); e < n_copy; e += (uint64_t)gridDim.x * 4u) {
        const uint4 d = p.copy_list[e]; // row, first kept op (of the normalised record), clipped first / last length (0: keep)
        rb_hit_row *row = &p.rows[d.x];
        const uint32_t rec = row->rec, n = row->out_n;
        const uint32_t *src = p.ops + p.op_off[rec] + p.norm[rec].first_op + d.y;
        const uint32_t padded = (n + 3u) & ~3u;
        const uint32_t arena = (uint32_t)(e % p.n_arena);
        unsigned long long b0 = 0;
        if (lane == 0) b0 = atomicAdd(&p.arena_cur[(uint64_t)arena * RB_ARENA_STRIDE], (unsigned long long)padded);
        b0 = rb_first64(b0);
        if (b0 + padded > p.arena_size) {
            if (lane == 0) p.counters->overflow = 1;
            continue;
        }
        const uint64_t off = p.arena_origin + (uint64_t)arena * p.arena_size + b0;
        uint32_t *dst = p.out_ops + off;
        for (uint32_t k = (uint32_t)lane; k < n; k += 64u) {
            uint32_t w = src[k];
            const uint32_t len0 = rb_len(w);
            uint32_t len = len0;
            if (n == 1u && d.z && d.w) len = d.z + d.w - len0;
            else {
                if (k == 0 && d.z) len = d.z;
                if (k == n - 1u && d.w) len = d.w;
            }
            dst[k] = (len << 4) | rb_opc(w);
        }
        if (lane == 0) row->out_off = off;
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
        const uint32_t opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
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

// one hit, one thread, serial (the general case of the general case: also what the wave kernel below hands unsorted arrays to)
__device__ void rb_generic_serial_hit(const rb_lift_params &p, const uint64_t g) {
    for (int once = 0; once < 1; once++) {
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
                const uint32_t opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
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
                if (rb_wlen(ops, i) == 0) continue; // (a zero-length op adds no unit)
                const uint32_t opc = rb_wopc(ops, i);
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
                const uint32_t opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
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
        uint32_t out_n = 0; // (in words: a merged run of 2^28 bases and more takes two)
        {
            uint64_t U = 0;
            uint32_t prev = RB_NULL_OP, run = 0;
            for (uint32_t i = 0; i < n; i++) {
                const uint32_t opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
                if (len == 0) continue;
                const uint64_t u0 = U, u1 = U + len - 1;
                U += len;
                if (u1 < a) continue;
                if (u0 > b) break;
                const uint64_t c0 = u0 > a ? u0 : a, c1 = u1 < b ? u1 : b;
                if (opc != prev) {
                    if (prev != RB_NULL_OP) out_n += 1u + ((run >> RB_LEN_BITS_WORD) ? 1u : 0u);
                    run = 0;
                }
                run += (uint32_t)(c1 - c0 + 1);
                prev = opc;
            }
            if (prev != RB_NULL_OP) out_n += 1u + ((run >> RB_LEN_BITS_WORD) ? 1u : 0u);
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
                const uint32_t opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
                if (len == 0) continue;
                const uint64_t u0 = U, u1 = U + len - 1;
                U += len;
                if (u1 < a) continue;
                if (u0 > b) break;
                const uint64_t c0 = u0 > a ? u0 : a, c1 = u1 < b ? u1 : b;
                const uint32_t piece = (uint32_t)(c1 - c0 + 1);
                if (opc != prev) {
                    if (prev != RB_NULL_OP) o += rb_emit_run(p.out_ops + o, run, prev);
                    prev = opc;
                    run = piece;
                } else {
                    run += piece;
                }
            }
            if (prev != RB_NULL_OP) o += rb_emit_run(p.out_ops + o, run, prev);
        }
        *row = w;
    }
}
__global__ __launch_bounds__(256) void rb_k_liftover_generic(rb_lift_params p) { // (diagnostics: RB_DEBUG_GENERIC_SERIAL)
    const uint64_t n_gen = p.counters->n_generic;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_gen; g += (uint64_t)gridDim.x * blockDim.x) rb_generic_serial_hit(p, g);
}

// ------------------------------------------------------------------------------------------------
// generic kernel, one WAVEFRONT per hit: the same unit semantics, every walk of the ops a pass of 64 ops per step with wave scans.
//   pass 1  equal ranges of the window's first / last base in the virtual tpos_aln (paf.rs:505-534) and the number of units
//   pass 2  first match-type unit >= the start index, last one <= the end index (paf.rs:551-558), with the prefixes there
//   pass 3  the ops between them, first / last length cut, adjacent ops of one type merged (paf.rs:602-620), zero lengths dropped
// A tpos_aln that is not sorted (units at position -1, see rb_bsearch_units) goes to the serial code above.
// ------------------------------------------------------------------------------------------------
// Checkpoints for the generic kernel (rb_lift_params::gen_cp): one wavefront per entry of the generic list walks the entry's record
// once and leaves, before every RB_GCP-th kept op, the units / reference / query / match bases so far.  The hits of a record follow
// each other in the list: only the first of a run builds (a record that appears in two runs is built twice, with the same values).
__device__ __forceinline__ uint4 *rb_gen_cp_of(const rb_lift_params &p, uint32_t r, uint32_t first_op) {
    return p.gen_cp + ((p.op_off[r] + first_op) / RB_GCP + r);
}
__global__ __launch_bounds__(256) void rb_k_generic_checkpoints(rb_lift_params p) {
    static_assert(RB_GCP == 64u, "one 16-byte load per lane covers four checkpoint intervals: lanes 0, 16, 32, 48 stand at their starts");
    const uint32_t wib = threadIdx.x >> 6;
    const int lane = rb_lane();
    const uint64_t n_gen = p.counters->n_generic;
    for (uint64_t g = (uint64_t)blockIdx.x * 4u + wib; g < n_gen; g += (uint64_t)gridDim.x * 4u) {
        const uint32_t r = p.rows[p.gen_list[g]].rec;
        if (g > 0 && p.rows[p.gen_list[g - 1]].rec == r) continue;
        const rb_norm_row *nr = &p.norm[r];
        if (nr->status != RB_ST_OK) continue;
        const uint32_t n = nr->n_ops;
        if (n <= RB_GCP) continue; // (one checkpoint, the record's start: nothing to look up)
        const uint32_t *ops = p.ops + p.op_off[r] + nr->first_op;
        uint4 *cp = rb_gen_cp_of(p, r, nr->first_op);
        auto load = [&](uint32_t c0) -> uint4 { // my four ops of the 256 that start at c0 (past the record: zero-length M ops)
            const uint32_t i = c0 + 4u * (uint32_t)lane;
            if (i + 3u < n) return rb_load4_unaligned(ops + i);
            return make_uint4(i < n ? ops[i] : 0u, i + 1u < n ? ops[i + 1u] : 0u, i + 2u < n ? ops[i + 2u] : 0u, 0u);
        };
        // The fields are 32 bits wide.  A record whose units in front of a checkpoint reach 2^32 (continuation words: up to 15 * 2^28
        // bases a word) gets none: checkpoint 0 -- zeros otherwise -- says so, and rb_k_liftover_generic_wave walks such a record
        // from its first op with its own 64-bit sums, as it does without checkpoints.
        // Round 5: a checkpoint every 64 ops (256 before): a walk of the wave kernel starts in the step that holds what it looks for
        // instead of up to three steps in front of it -- the kernel is bound by the instructions of its steps.  A lane sums its four
        // ops, wave scans give every lane the sums in front of it, and the lanes that stand at a multiple of 64 ops write.
        uint64_t U = 0;
        uint32_t R = 0, Q = 0, M = 0;
        uint4 nxt = load(0u);
        for (uint32_t c0 = 0; c0 < n; c0 += 256u) {
            const uint4 cur = nxt;
            if (c0 + 256u < n) nxt = load(c0 + 256u); // (in flight while these ops are summed)
            const uint32_t w4[4] = {cur.x, cur.y, cur.z, cur.w};
            uint64_t u = 0; // (the reference / query / match sums are parts of it: they stay below 2^32 wherever a checkpoint is written)
            uint32_t rr = 0, q = 0, m = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i = c0 + 4u * (uint32_t)lane + (uint32_t)k;
                uint32_t opc = rb_opc(w4[k]), len = rb_len(w4[k]);
                if (opc == RB_OP_CONT && i < n) opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
                const bool okc = opc <= 8u;
                u += len;
                rr += okc && rb_in(RB_REF_MASK, opc) ? len : 0u;
                q += okc && rb_in(RB_QRY_MASK, opc) ? len : 0u;
                m += okc && rb_in(RB_MATCH_MASK, opc) ? len : 0u;
            }
            // (u < 2^34 a lane: scanned as its low 24 bits and what is above them, neither of which can leave 32 bits over 64 lanes)
            const uint64_t iu = (uint64_t)rb_wave_scan_incl((uint32_t)u & 0xFFFFFFu) + ((uint64_t)rb_wave_scan_incl((uint32_t)(u >> 24)) << 24);
            const uint32_t ir = rb_wave_scan_incl(rr), iq = rb_wave_scan_incl(q), im = rb_wave_scan_incl(m);
            const bool mine = (lane & 15) == 0 && c0 + 4u * (uint32_t)lane < n; // a checkpoint stands in front of my first op
            const uint64_t Uv = U + iu - u;
            if (__ballot(mine && (Uv >> 32)) != 0ull) {
                if (lane == 0) cp[0] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                break;
            }
            if (mine) cp[c0 / RB_GCP + ((uint32_t)lane >> 4)] = make_uint4((uint32_t)Uv, R + ir - rr, Q + iq - q, M + im - m);
            U += rb_readlane<uint64_t>(iu, 63), R += rb_readlane<uint32_t>(ir, 63), Q += rb_readlane<uint32_t>(iq, 63), M += rb_readlane<uint32_t>(im, 63);
        }
    }
}
__device__ __forceinline__ uint64_t rb_wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
__device__ __forceinline__ uint64_t rb_wave_max_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint64_t o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}
// One thread per generic hit: everything the wave kernel needs to start on it, gathered -- the dependent chain list entry -> row -> record's
// row, offsets, strand -> windows -> checkpoint searches is walked here by as many threads as there are hits, not by a wave per hit in front
// of its first op (round 5; the stream kernel got its rb_job the same way in round 1).
__global__ __launch_bounds__(256) void rb_k_generic_jobs(rb_lift_params p) {
    const uint64_t n_gen = p.counters->n_generic;
    for (uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x; g < n_gen; g += (uint64_t)gridDim.x * 256u) {
        const uint32_t hrow = p.gen_list[g];
        const rb_hit_row *row = &p.rows[hrow];
        const uint32_t r = row->rec, win = row->win;
        const rb_norm_row *nr = &p.norm[r];
        rb_gja A;
        A.t_st = nr->t_st, A.t_en = nr->t_en, A.q_st = nr->q_st, A.q_en = nr->q_en;
        A.n = nr->n_ops;
        A.ops_off = p.op_off[r] + nr->first_op;
        A.wst = p.x_st ? p.x_st[hrow] : p.wo_st[win];
        A.wen = p.x_en ? p.x_en[hrow] : p.wo_en[win];
        A.k2 = 0;
        uint32_t k1 = 0, has_cp = 0;
        const uint32_t status = nr->status;
        if (status == RB_ST_OK && !(A.t_st > A.wst && A.t_en < A.wen)) {
            const uint4 *gcp = (p.gen_cp && A.n > RB_GCP) ? rb_gen_cp_of(p, r, nr->first_op) : nullptr;
            if (gcp && gcp[0].x == 0xFFFFFFFFu) gcp = nullptr; // (a record of 2^32 units and more has no checkpoints)
            if (gcp) {
                has_cp = 1;
                const uint32_t ncp = (A.n + RB_GCP - 1u) / RB_GCP;
                auto last_le = [&](uint64_t target) -> uint32_t { // the last checkpoint with at most `target` reference bases in front of it (checkpoint 0 holds zeros)
                    uint32_t lo = 0, hi = ncp;
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((uint64_t)gcp[mid].y <= target) lo = mid; else hi = mid;
                    }
                    return lo;
                };
                const int64_t ps = (int64_t)(A.wst > A.t_st ? A.wst : A.t_st), pe = (int64_t)(A.wen < A.t_en ? A.wen : A.t_en) - 1;
                k1 = last_le((uint64_t)ps - A.t_st);
                const uint32_t k2 = pe >= ps ? last_le((uint64_t)pe - A.t_st) : 0u;
                if (k2 > k1 + 1u) A.k2 = k2;
            }
        }
        p.gj_a[g] = A;
        p.gj_b[g] = make_uint4(hrow, r, win, (status << 16) | (((uint32_t)row->flags & 0xFFu) << 8) | (has_cp << 1) | (p.strand[r] == (uint8_t)'-' ? 1u : 0u));
        p.gj_c[g] = make_uint2(nr->aln_len, k1);
    }
}
#ifndef RB_GW_WPE
#define RB_GW_WPE 4 // (round 5: the kernel wants 127 registers; at five waves per SIMD (96) it spilled 74 of them to scratch, at four it spills none:
                    //  11.58 -> 10.48 ms on the irregular workload, and two, three or four waves take the same time -- the kernel is bound by the
                    //  instructions it issues, not by what it waits for.  tools/r05_gw_ab.sh, same box.  Earlier in the round, with groups of
                    //  four steps: 5 waves 12.5 ms, 6 waves 16.8, 8 waves 17.0 -- against 15.0 for the round-3 form at 8 waves)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(RB_GW_WPE))) void rb_k_liftover_generic_wave(rb_lift_params p) {
    __shared__ uint32_t run_tot_all[4][64], run_opc_all[4][64];
    const uint32_t wib = threadIdx.x >> 6;
    uint32_t *run_tot = run_tot_all[wib], *run_opc = run_opc_all[wib];
    const int lane = rb_lane();
    const uint64_t n_gen = p.counters->n_generic;
    for (uint64_t g = (uint64_t)blockIdx.x * 4u + wib; g < n_gen; g += (uint64_t)gridDim.x * 4u) {
        // the hit's descriptor (rb_k_generic_jobs): three loads at addresses the whole wave shares, one trip
        const rb_gja A_ = p.gj_a[g];
        const uint4 B_ = p.gj_b[g];
        const uint2 C_ = p.gj_c[g];
        const uint64_t hrow = rb_first(B_.x);
        const uint32_t r = rb_first(B_.y), win = rb_first(B_.z), gfl = rb_first(B_.w);
        rb_hit_row *row = &p.rows[hrow];
        rb_hit_row w;
        w.rec = r;
        w.win = win;
        w.flags = RB_HIT_GENERIC;
        w.status = RB_ST_OK;
        w.out_n = 0;
        w.out_off = 0;
        w.t_st = w.t_en = w.q_st = w.q_en = 0;
        w.nmatch = w.aln_len = 0;
        if ((gfl >> 16) != RB_ST_OK) { // fused scan: the record was handed back and the full scan found the reference would panic on it
            w.status = (uint16_t)(gfl >> 16);
            w.flags = (uint16_t)((gfl >> 8) & 0xFFu);
            if (lane == 0) *row = w;
            continue;
        }
        const uint64_t t_st = rb_first64(A_.t_st), t_en = rb_first64(A_.t_en), q_st = rb_first64(A_.q_st), q_en = rb_first64(A_.q_en);
        const bool minus = (gfl & 1u) != 0u;
        const uint32_t n = rb_first(A_.n);
        const uint64_t ops_off = rb_first64(A_.ops_off);
        const uint32_t *ops = p.ops + ops_off;
        const uint64_t wst = rb_first64(A_.wst), wen = rb_first64(A_.wen);
        const uint32_t rec_units = rb_first(C_.x);
        const uint32_t arena = (uint32_t)(g % p.n_arena);
        auto reserve = [&](uint32_t padded, uint64_t *off) -> bool { // room in an arena, for the whole wave
            unsigned long long b0 = 0;
            if (lane == 0) b0 = atomicAdd(&p.arena_cur[(uint64_t)arena * RB_ARENA_STRIDE], (unsigned long long)padded);
            b0 = rb_first64(b0);
            if (b0 + padded > p.arena_size) {
                if (lane == 0) p.counters->overflow = 1;
                return false;
            }
            *off = p.arena_origin + (uint64_t)arena * p.arena_size + b0;
            return true;
        };
        if (t_st > wst && t_en < wen) { // liftover.rs:23-25: verbatim clone, own id
            w.flags |= RB_HIT_INSIDE;
            w.t_st = t_st, w.t_en = t_en, w.q_st = q_st, w.q_en = q_en;
            w.nmatch = p.norm[r].nmatch, w.aln_len = rec_units;
            w.out_n = n;
            uint64_t off;
            if (reserve((n + 3u) & ~3u, &off)) {
                w.out_off = off;
                for (uint32_t i = (uint32_t)lane; i < n; i += 64u) p.out_ops[off + i] = ops[i];
            }
            if (lane == 0) *row = w;
            continue;
        }
        const int64_t ps = (int64_t)(wst > t_st ? wst : t_st); // positions to look up (liftover.rs:28, :38-40)
        const int64_t pe = (int64_t)(wen < t_en ? wen : t_en) - 1;
        // ---- where the walks start: the last checkpoint with at most ps - t_st reference bases in front of it (every unit at
        //      position ps, and everything behind it, lies at or behind that op).  The question pass 1 answers with the record's
        //      first ops -- do units at position -1 come first? -- is asked of them directly then. ----
        uint32_t c_start = 0, c_end1 = 0;               // c_end1: where the walk for the window's END may resume (pass 1)
        uint4 cp0 = make_uint4(0u, 0u, 0u, 0u), cp_e1 = cp0;
        bool first_seen = false, wrapped = false;
        const uint4 *gcp = (gfl & 2u) ? p.gen_cp + (ops_off / RB_GCP + r) : nullptr; // (rb_gen_cp_of; rb_k_generic_jobs has looked: there are checkpoints)
        const uint32_t ncp = (n + RB_GCP - 1u) / RB_GCP;
        // the last checkpoint whose field (1: reference bases, 0: units) is <= target (checkpoint 0 holds zeros); the fields never decrease
        auto cp_search = [&](int field, uint64_t target) -> uint32_t {
            uint32_t k1 = 0;
            for (uint32_t kb = 0; kb < ncp; kb += 64u) {
                const uint32_t k = kb + (uint32_t)lane;
                const uint4 c = gcp[k < ncp ? k : 0u];
                const uint64_t m = __ballot(k < ncp && (uint64_t)(field ? c.y : c.x) <= target);
                if (!m) break;
                k1 = kb + (uint32_t)__builtin_popcountll(m) - 1u; // (m is a run of low bits)
                if (m != ~0ull) break;
            }
            return k1;
        };
        if (gcp) {
            const uint32_t k1 = rb_first(C_.y), k2 = rb_first(A_.k2); // (searched by rb_k_generic_jobs: k2 != 0 means k2 > k1 + 1)
            if (k2) c_end1 = k2 * RB_GCP, cp_e1 = gcp[k2];
            if (k1) {
                c_start = k1 * RB_GCP;
                cp0 = gcp[k1];
                if (t_st == 0) { // (only then can a unit sit at position -1)
                    for (uint32_t c0 = 0; c0 < n && !first_seen; c0 += 64u) {
                        const uint32_t i = c0 + (uint32_t)lane;
                        uint32_t opc = RB_NULL_OP, len = 0u;
                        if (i < n) opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
                        const uint64_t m = __ballot(len != 0u);
                        if (m) {
                            first_seen = true;
                            wrapped = !((__ballot(opc <= 8u && rb_in(RB_REF_MASK, opc)) >> __builtin_ctzll(m)) & 1ull);
                        }
                    }
                }
                first_seen = true;
            }
        }
        // ---- pass 1 ----
        uint64_t s_lo = ~0ull, s_hi = 0, e_lo = ~0ull, e_hi = 0;
        uint64_t Ub = cp0.x, Rb = cp0.y;
        // (the next step's ops are asked for before this step's are looked at: a hit is a chain of dependent steps, and the load is the
        //  longest link of each)
        auto ld = [&](uint32_t c0_) -> uint32_t { return c0_ + (uint32_t)lane < n ? ops[c0_ + (uint32_t)lane] : 0u; };
        // Round 5: the ops come in GROUPS of steps (G loads out at once), and the group behind the one being walked is asked for when its
        // predecessor is entered: a hit was a chain of some 25 dependent steps of 3.8 us each (one 256-byte load a step, one step
        // ahead); it is a chain of groups now.  win_take(G, c0): the 64 ops of the step at c0 (a multiple of 64 from the pass's start;
        // behind a jump the window is refilled).
#ifndef RB_GW_GROUP
#define RB_GW_GROUP 1  // passes 1 and 2: with a checkpoint in front of every step they are walks of one or two steps, and every step asked for
                       // beyond them is a load for nothing (groups of 2 / 3 / 4 in all passes: 7.68 / 8.49 / 8.85 ms on the irregular workload)
#endif
#ifndef RB_GW_GROUP3
#define RB_GW_GROUP3 2 // pass 3 walks ia .. ib, known in advance (eight steps on the bench's windows; past ib no load is issued) -- and still
                       // does not gain from loads further ahead: groups of 2 / 4 / 8 with passes 1 and 2 at one: 7.64 / 7.82 / 8.28 ms.  What a
                       // larger group adds is the selects of win_take and the copies between the two groups; the kernel is bound by what it
                       // issues (its steps are full of wave-uniform decisions: 286 scalar instructions beside 387 vector ones in the
                       // general step of pass 3), not by its loads -- two, three or four waves per SIMD take the same time
#endif
        constexpr int RB_GW_GMAX = RB_GW_GROUP > RB_GW_GROUP3 ? RB_GW_GROUP : RB_GW_GROUP3;
        using rb_g12 = std::integral_constant<int, RB_GW_GROUP>;
        using rb_g3 = std::integral_constant<int, RB_GW_GROUP3>;
        uint32_t wc[RB_GW_GMAX], wx[RB_GW_GMAX];
        uint32_t wcb = 0xFFFFFFFFu, wxb = 0xFFFFFFFFu; // first op of the current group / of the group ahead (none)
        auto win_fill = [&](auto G_, uint32_t base, uint32_t (&dst)[RB_GW_GMAX], auto &&ldf) {
            constexpr int G = decltype(G_)::value;
#pragma unroll
            for (int k = 0; k < G; k++) dst[k] = ldf(base + 64u * (uint32_t)k);
        };
        auto win_take = [&](auto G_, uint32_t c0_, auto &&ldf) -> uint32_t {
            constexpr int G = decltype(G_)::value;
            const uint32_t d = c0_ - wcb;
            if (wcb == 0xFFFFFFFFu || d >= 64u * G || (d & 63u)) { // not in the current group
                if (c0_ == wxb) {
#pragma unroll
                    for (int k = 0; k < G; k++) wc[k] = wx[k];
                } else {
                    win_fill(G_, c0_, wc, ldf);
                }
                wcb = c0_, wxb = c0_ + 64u * G;
                win_fill(G_, wxb, wx, ldf); // (the group behind it: out now, wanted G steps from now)
            }
            const uint32_t k = (c0_ - wcb) >> 6;
            uint32_t v = wc[0];
#pragma unroll
            for (int q = 1; q < G; q++) v = k == (uint32_t)q ? wc[q] : v;
            return v;
        };
        auto win_reset = [&]() { wcb = wxb = 0xFFFFFFFFu; };
        for (uint32_t c0 = c_start; c0 < n; c0 += 64u) {
            const uint32_t i = c0 + (uint32_t)lane;
            const uint32_t wv = win_take(rb_g12{}, c0, ld);
            uint32_t opc = rb_opc(wv), len = i < n ? rb_len(wv) : 0u;
            if (opc == RB_OP_CONT) opc = rb_wopc(ops, i), len = rb_wlen(ops, i); // (walk form: one more op of its owner's type)
            const bool isref = opc <= 8u && rb_in(RB_REF_MASK, opc);
            const uint32_t iu = rb_wave_scan_incl(len), ir = rb_wave_scan_incl(isref ? len : 0u);
            if (!first_seen) {
                const uint64_t m = __ballot(len != 0u);
                if (m) {
                    first_seen = true;
                    wrapped = t_st == 0 && !((__ballot(isref) >> __builtin_ctzll(m)) & 1ull);
                }
            }
            if (len) {
                const uint64_t U = Ub + iu - len;
                const int64_t tpos = (int64_t)t_st - 1 + (int64_t)(Rb + ir - (isref ? len : 0u));
                if (isref) { // units U .. U + len - 1 hold tpos + 1 .. tpos + len
                    if (ps > tpos && ps <= tpos + (int64_t)len) {
                        const uint64_t u = U + (uint64_t)(ps - tpos - 1);
                        s_lo = s_lo < u ? s_lo : u, s_hi = s_hi > u ? s_hi : u;
                    }
                    if (pe > tpos && pe <= tpos + (int64_t)len) {
                        const uint64_t u = U + (uint64_t)(pe - tpos - 1);
                        e_lo = e_lo < u ? e_lo : u, e_hi = e_hi > u ? e_hi : u;
                    }
                } else {
                    if (ps == tpos && tpos >= 0) s_lo = s_lo < U ? s_lo : U, s_hi = s_hi > U + len - 1 ? s_hi : U + len - 1;
                    if (pe == tpos && tpos >= 0) e_lo = e_lo < U ? e_lo : U, e_hi = e_hi > U + len - 1 ? e_hi : U + len - 1;
                }
            }
            Ub += rb_readlane<uint32_t>(iu, 63);
            Rb += rb_readlane<uint32_t>(ir, 63);
            if ((int64_t)t_st - 1 + (int64_t)Rb > pe) break; // every unit behind this step lies behind the window's last base
            // every unit at position ps is behind us: on to the checkpoint in front of the units at pe (a long window is not walked)
            if (c_end1 > c0 + 64u && (int64_t)t_st - 1 + (int64_t)Rb > ps) {
                c0 = c_end1 - 64u;
                Ub = cp_e1.x, Rb = cp_e1.y;
                c_end1 = 0;
            }
        }
        if (wrapped) { // an unsorted tpos_aln: binary_search returns what its probe sequence leads to (serial replay)
            if (lane == 0) rb_generic_serial_hit(p, g);
            continue;
        }
        const uint64_t N = rec_units; // all units of the (normalised) record
#if defined(RB_GW_STOP) && RB_GW_STOP == 1 // (diagnostics, timing only: the hit ends behind pass 1)
        if (s_lo != 0x12345ull) { if (lane == 0) row->status = (uint16_t)(s_lo & 1u); continue; }
#endif
        s_lo = rb_wave_min_u64(s_lo), s_hi = rb_wave_max_u64(s_hi), e_lo = rb_wave_min_u64(e_lo), e_hi = rb_wave_max_u64(e_hi);
        if (s_lo == ~0ull || e_lo == ~0ull) { // binary_search Err -> panic (liftover.rs:31, :42)
            w.status = RB_ST_PANIC_NOTFOUND;
            if (lane == 0) *row = w;
            continue;
        }
        const uint64_t ks = p.policy == RB_BSEARCH_LEGACY ? rb_legacy_probe(N, s_lo, s_hi) : s_hi;
        const uint64_t ke = p.policy == RB_BSEARCH_LEGACY ? rb_legacy_probe(N, e_lo, e_hi) : e_hi;
        // ---- pass 2: a = first match-type unit >= ks (else N); b = last match-type unit <= ke (else 0) ----
        uint64_t a = N, b = 0, Ra = 0, Qa = 0, Ma = 0, nRb = 0, nQb = 0, nMb = 0, Ua_op = 0, Ub_op = 0;
        uint32_t ia = 0, ib = 0, len_a = 0, len_b = 0;
        bool a_set = false, b_set = false;
        {
            uint64_t U0 = cp0.x, R0 = cp0.y, Q0 = cp0.z, M0 = cp0.w; // (ks, the unit the start resolves to, lies behind the checkpoint too)
            // b is looked for from the checkpoint in front of unit ke; if the stretch from there to ke holds no match-type unit, the
            // walk is done again without the jump (jump2 = 0)
            uint32_t c_end2 = 0;
            uint4 cp_e2 = cp0;
            if (gcp && ke >= ks) {
                const uint32_t k3 = cp_search(0, ke);
                if (k3 * RB_GCP > c_start + RB_GCP) c_end2 = k3 * RB_GCP, cp_e2 = gcp[k3];
            }
          for (int attempt = 0; attempt < 2; attempt++) {
            bool jumped = false;
            if (attempt) U0 = cp0.x, R0 = cp0.y, Q0 = cp0.z, M0 = cp0.w, a_set = false, b_set = false, a = N, b = 0, c_end2 = 0;
            win_reset();
            for (uint32_t c0 = c_start; c0 < n; c0 += 64u) {
                if (a_set && U0 > ke) break; // (nothing behind this can be <= ke)
                const uint32_t i = c0 + (uint32_t)lane;
                const uint32_t wv = win_take(rb_g12{}, c0, ld);
                uint32_t opc = rb_opc(wv), len = i < n ? rb_len(wv) : 0u;
                if (opc == RB_OP_CONT) opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
                const bool okc = opc <= 8u;
                const bool isref = okc && rb_in(RB_REF_MASK, opc), isq = okc && rb_in(RB_QRY_MASK, opc), ism = okc && rb_in(RB_MATCH_MASK, opc);
                const uint32_t iu = rb_wave_scan_incl(len), ir = rb_wave_scan_incl(isref ? len : 0u), iq = rb_wave_scan_incl(isq ? len : 0u),
                               im = rb_wave_scan_incl(ism ? len : 0u);
                const uint64_t U = U0 + iu - len, R = R0 + ir - (isref ? len : 0u), Q = Q0 + iq - (isq ? len : 0u), M = M0 + im - (ism ? len : 0u);
                if (!a_set) {
                    const uint64_t m = __ballot(ism && len != 0u && U + len > ks);
                    if (m) {
                        const int l = __builtin_ctzll(m);
                        const uint64_t Ul = rb_readlane<uint64_t>(U, l);
                        a = ks > Ul ? ks : Ul;
                        const uint64_t off = a - Ul;
                        Ra = rb_readlane<uint64_t>(R, l) + off, Qa = rb_readlane<uint64_t>(Q, l) + off, Ma = rb_readlane<uint64_t>(M, l) + off;
                        Ua_op = Ul, ia = c0 + (uint32_t)l, len_a = rb_readlane<uint32_t>(len, l);
                        a_set = true;
                    }
                }
                {
                    const uint64_t m = __ballot(ism && len != 0u && U <= ke);
                    if (m) {
                        const int l = 63 - __builtin_clzll(m);
                        const uint64_t Ul = rb_readlane<uint64_t>(U, l);
                        const uint32_t ll = rb_readlane<uint32_t>(len, l);
                        b = Ul + ll - 1 < ke ? Ul + ll - 1 : ke;
                        const uint64_t off = b - Ul;
                        nRb = rb_readlane<uint64_t>(R, l) + off + 1, nQb = rb_readlane<uint64_t>(Q, l) + off + 1, nMb = rb_readlane<uint64_t>(M, l) + off + 1;
                        Ub_op = Ul, ib = c0 + (uint32_t)l, len_b = ll;
                        b_set = true;
                    }
                }
                U0 += rb_readlane<uint32_t>(iu, 63), R0 += rb_readlane<uint32_t>(ir, 63), Q0 += rb_readlane<uint32_t>(iq, 63), M0 += rb_readlane<uint32_t>(im, 63);
                if (a_set && c_end2 > c0 + 64u) { // a is known: straight to the checkpoint in front of unit ke
                    c0 = c_end2 - 64u;
                    U0 = cp_e2.x, R0 = cp_e2.y, Q0 = cp_e2.z, M0 = cp_e2.w;
                    c_end2 = 0, jumped = true, b_set = false;
                }
            }
            if (!jumped || b_set) break; // (jumped and found nothing behind the jump: the last match-type unit <= ke lies in what was skipped)
          }
        }
        if (a > b || a >= N || !a_set || !b_set) { // liftover.rs:52-54
            w.status = RB_ST_NONE_INDEL;
            if (lane == 0) *row = w;
            continue;
        }
#if defined(RB_GW_STOP) && RB_GW_STOP == 2 // (diagnostics, timing only: the hit ends behind pass 2)
        if (a != 0x12345ull) { if (lane == 0) row->status = (uint16_t)((a + b + Ra + Qa + Ma + nRb + nQb + nMb + ia + ib) & 1u); continue; }
#endif
        w.t_st = t_st + Ra; // liftover.rs:57-60, :77-82 (a and b are match-type units)
        w.t_en = t_st + nRb;
        if (!minus) w.q_st = q_st + Qa, w.q_en = q_st + nQb;
        else w.q_st = q_en - nQb, w.q_en = q_en - Qa;
        w.nmatch = (uint32_t)(nMb - Ma);
        w.aln_len = (uint32_t)(b - a + 1);
        // ---- pass 3: ops ia .. ib, zero lengths dropped, first / last cut, runs of one type merged ----
        uint64_t off;
        if (!reserve((ib - ia + 2u + 3u) & ~3u, &off)) { // (at least as many slots as the merge leaves; + 1: a clip that begins in the bases of a continuation word)
            if (lane == 0) *row = w;
            continue;
        }
        uint32_t out_pos = 0, c_tot = 0, c_opc = 0;
        bool has_carry = false;
        auto ld3 = [&](uint32_t c0_) -> uint32_t {
            const uint32_t i_ = c0_ + (uint32_t)lane;
            return (i_ >= ia && i_ <= ib) ? ops[i_] : 0u;
        };
        win_reset();
        for (uint32_t c0 = ia & ~63u; c0 <= ib; c0 += 64u) {
            const uint32_t i = c0 + (uint32_t)lane;
            const uint32_t wv = win_take(rb_g3{}, c0, ld3);
            uint32_t opc = rb_opc(wv), len = rb_len(wv);
            if (opc == RB_OP_CONT) opc = rb_wopc(ops, i), len = rb_wlen(ops, i);
            const bool in = i >= ia && i <= ib && len != 0u;
            uint32_t piece = len;
            if (i == ia) piece = ia == ib ? (uint32_t)(b - a + 1) : (uint32_t)(Ua_op + len_a - a);
            else if (i == ib) piece = (uint32_t)(b - Ub_op + 1);
#ifndef RB_GW_NO_PLAIN_STEP
            // Round 5: the step that needs none of the machinery below -- every op of the range kept (no zero length), a plain word, no two
            // neighbours of one type (the run carried over is lane 0's neighbour): the ops leave as they are, cut at the ends, and the
            // last one is carried.  An irregular record is irregular in a few places; this is the step of all its other ops (the
            // general step is some 390 vector instructions for 64 ops, and the kernel is bound by them).
            {
                const bool inr = i >= ia && i <= ib;
                const uint32_t ptype = rb_prev_lane(opc, has_carry ? c_opc : 0xFFu);
                const bool odd = inr && (len == 0u || rb_opc(wv) == RB_OP_CONT || (piece >> RB_LEN_BITS_WORD) != 0u || (ptype == opc && (lane == 0 || i > ia)));
                if (!__ballot(odd)) {
                    if (has_carry) {
                        if (lane == 0) rb_emit_run(p.out_ops + off + out_pos, c_tot, c_opc);
                        out_pos += 1u + ((c_tot >> RB_LEN_BITS_WORD) ? 1u : 0u);
                    }
                    const uint64_t im = __ballot(inr); // (not empty: the loop runs over the steps of ia .. ib)
                    const uint32_t cnt = (uint32_t)__builtin_popcountll(im), rank = (uint32_t)__builtin_popcountll(im & ((1ull << lane) - 1ull));
                    if (inr && rank + 1u < cnt) p.out_ops[off + out_pos + rank] = (piece << 4) | opc;
                    out_pos += cnt - 1u;
                    const int last = 63 - __builtin_clzll(im);
                    c_tot = rb_readlane<uint32_t>(piece, last), c_opc = rb_readlane<uint32_t>(opc, last);
                    has_carry = true;
                    continue;
                }
            }
#endif
            // code of the last kept op in front of this lane (bit 4: there is one); lane 0 takes the run carried over
            const uint32_t key = in ? ((uint32_t)lane << 5) | 16u | opc : 0u;
            const uint32_t incl = rb_wave_scan_incl_max_u32(key);
            const uint32_t ckey = has_carry ? 16u | c_opc : 0u;
            uint32_t prev = rb_prev_lane(incl, ckey);
            prev = (prev & 16u) ? prev : ckey; // (no kept op in front of this lane in this step: the run carried over)
            const bool start = in && !((prev & 16u) && (prev & 15u) == opc);
            const uint64_t sm = __ballot(start);
            const uint32_t nstarts = (uint32_t)__builtin_popcountll(sm);
            const int32_t ridx = (int32_t)__builtin_popcountll(sm & ((2ull << lane) - 1ull)) - 1; // run of this lane inside the step (-1: the carried one)
            run_tot[lane] = 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (in && ridx >= 0) atomicAdd(&run_tot[ridx], piece);
            if (start) run_opc[ridx] = opc;
            c_tot += rb_wave_sum_u32((in && ridx < 0) ? piece : 0u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (nstarts) { // (runs of 2^28 bases and more leave as two words: rb_emit_run)
                if (has_carry) {
                    const uint32_t cw = 1u + ((c_tot >> RB_LEN_BITS_WORD) ? 1u : 0u);
                    if (lane == 0) rb_emit_run(p.out_ops + off + out_pos, c_tot, c_opc);
                    out_pos += cw;
                }
                const bool mine_out = (uint32_t)lane + 1u < nstarts;
                const uint32_t my_tot = mine_out ? run_tot[lane] : 0u;
                const uint32_t my_words = mine_out ? 1u + ((my_tot >> RB_LEN_BITS_WORD) ? 1u : 0u) : 0u;
                const uint32_t wincl = rb_wave_scan_incl(my_words);
                if (mine_out) rb_emit_run(p.out_ops + off + out_pos + (wincl - my_words), my_tot, run_opc[lane]);
                out_pos += rb_readlane<uint32_t>(wincl, 63);
                c_tot = run_tot[nstarts - 1u], c_opc = run_opc[nstarts - 1u];
                has_carry = true;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (has_carry) {
            if (lane == 0) rb_emit_run(p.out_ops + off + out_pos, c_tot, c_opc);
            out_pos += 1u + ((c_tot >> RB_LEN_BITS_WORD) ? 1u : 0u);
        }
        w.out_n = out_pos;
        w.out_off = off;
        if (lane == 0) *row = w;
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
    p.counters->out_ops_used = p.arena_origin + sum;
    // what makes the job fit: the slots the window lists ask for, plus every arena as large as the fullest one got
    p.counters->out_ops_needed = p.needed_base + (mx + 3ull) / 4ull * 4ull * p.n_arena + 1024ull * p.n_arena;
    if (p.counters->n_hits > p.rows_cap) p.counters->overflow = 1;
    if (p.n_tiles && !p.debug_skip) { // (the diagnostics words, outside the diagnostics builds: how the short records went)
        p.counters->phase[3] = p.n_tiles;
        p.counters->phase[4] = (uint32_t)*p.fb_count; // records the tile kernel handed to the per-record kernel
    }
    if (p.brk_mode && p.counters->brk_scratch_short) { // one of the scratch-row cursors ran out before the rows did: ask for a quarter more
        p.counters->overflow = 1;
        const uint64_t have = p.counters->n_hits > p.rows_cap ? p.counters->n_hits : p.rows_cap;
        p.counters->n_hits = have + have / 4u + 1024u;
    }
}

// one-walk break-paf: the rows of a record leave their scratch place for rows_final[hit_off[r] ..] (hit_off scanned by now)
__global__ __launch_bounds__(256) void rb_k_break_gather(rb_lift_params p) {
    // four lanes per record, 16 bytes of a 64-byte row each (a record has a handful of pieces: a wavefront per record idles)
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    const uint64_t r = t >> 2;
    if (r >= p.n_rec) return;
    const uint64_t h0 = p.hit_off[r], n = p.hit_off[r + 1] - h0, src = p.brk_off[r];
    if (src == ~0ull) return;
    const uint32_t part = (uint32_t)(t & 3u);
    const uint4 *from = reinterpret_cast<const uint4 *>(p.rows + src);
    uint4 *to = reinterpret_cast<uint4 *>(p.rows_final + h0);
    for (uint64_t j = 0; j < n; j++)
        if (h0 + j < p.rows_cap) to[4 * j + part] = from[4 * j + part];
}
extern "C" hipError_t rb_launch_break_gather(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_copy_clips, dim3(2048), dim3(256), 0, stream, *p); // (clips without a slot: their rows are still where the list says)
    hipLaunchKernelGGL(rb_k_break_gather, dim3((unsigned)((p->n_rec * 4 + 255) / 256)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
// one-walk break-paf: the pieces of the records the clip kernel declined get their rows (in the final array, hit_off scanned by
// now) and their entries in the generic list; then the generic kernel clips them (p.rows = the final rows, p.x_st / x_en = the
// windows rb_k_break_pieces wrote in list mode) and rb_k_finish sums up
__global__ __launch_bounds__(256) void rb_k_break_list_declined(rb_lift_params p) {
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= p.n_rec || p.brk_off[r] != ~1ull) return;
    p.brk_off[r] = ~0ull; // (the gather leaves the record alone)
    p.hit_off[r] = 0;     // (rb_k_break_pieces counts its pieces next)
    p.brk_decl_list[atomicAdd(p.brk_decl_count, 1ull)] = (uint32_t)r;
}
extern "C" hipError_t rb_launch_break_list_declined(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_break_list_declined, dim3((unsigned)((p->n_rec + 255) / 256)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void rb_k_break_declined_rows(rb_lift_params p) {
    const uint64_t n_list = *p.brk_decl_count;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n_list; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t r = p.brk_decl_list[i];
        const uint64_t h0 = p.hit_off[r], n = p.hit_off[r + 1] - h0;
        for (uint64_t j = 0; j < n; j++) {
            const uint64_t h = h0 + j;
            if (h >= p.rows_cap) break;
            rb_hit_row *row = &p.rows[h];
            row->rec = r;
            row->win = (uint32_t)j;
            row->flags = RB_HIT_GENERIC;
            const unsigned long long g = atomicAdd((unsigned long long *)&p.counters->n_generic, 1ull);
            p.gen_list[g] = (uint32_t)h;
        }
    }
}
extern "C" hipError_t rb_launch_break_declined(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_break_declined_rows, dim3((unsigned)std::min<uint64_t>((p->n_rec + 255) / 256, 256)), dim3(256), 0, stream, *p);
    if (p->gen_cp) hipLaunchKernelGGL(rb_k_generic_checkpoints, dim3(2048), dim3(256), 0, stream, *p);
    hipLaunchKernelGGL(rb_k_generic_jobs, dim3(1024), dim3(256), 0, stream, *p);
    hipLaunchKernelGGL(rb_k_liftover_generic_wave, dim3(2048), dim3(256), 0, stream, *p);
    hipLaunchKernelGGL(rb_k_finish, dim3(1), dim3(64), 0, stream, *p);
    return hipGetLastError();
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
    if (p->debug_skip) { // diagnostics only (bench.py --debug-skip, the box block's clock stamps)
        if (!p->brk_mode) {
            hipLaunchKernelGGL(rb_k_liftover_stream_diag, dim3(blocks), dim3(256), dyn, stream, *p);
            return hipGetLastError();
        }
    }
    if (p->brk_mode) hipLaunchKernelGGL(rb_k_liftover_stream_brk, dim3(blocks), dim3(256), dyn, stream, *p);
    else hipLaunchKernelGGL(rb_k_liftover_stream, dim3(blocks), dim3(256), dyn, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t rb_launch_liftover_tail(const rb_lift_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_copy_clips, dim3(2048), dim3(256), 0, stream, *p);
    static const bool serial_generic = getenv("RB_DEBUG_GENERIC_SERIAL") != nullptr; // diagnostics: one thread per hit, as in round 1
    if (serial_generic) {
        hipLaunchKernelGGL(rb_k_liftover_generic, dim3(1024), dim3(256), 0, stream, *p);
    } else {
        if (p->gen_cp) hipLaunchKernelGGL(rb_k_generic_checkpoints, dim3(2048), dim3(256), 0, stream, *p);
        hipLaunchKernelGGL(rb_k_generic_jobs, dim3(1024), dim3(256), 0, stream, *p);
        hipLaunchKernelGGL(rb_k_liftover_generic_wave, dim3(2048), dim3(256), 0, stream, *p);
    }
    hipLaunchKernelGGL(rb_k_finish, dim3(1), dim3(64), 0, stream, *p);
    return hipGetLastError();
}

extern "C" size_t rb_scan_block_sums_count(uint64_t n_rec) { return (size_t)((n_rec + RB_SCAN_PER_BLOCK - 1) / RB_SCAN_PER_BLOCK + 2); }
#endif // !RB_LIST_TU
