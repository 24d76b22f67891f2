// k_misc.hip -- break-paf piece enumeration, invert (swap), synthetic workload fill (gfx950).
#include "rb_device.h"
#include <algorithm>
#include "synth.h"

// ------------------------------------------------------------------------------------------------
// break-paf (liftover.rs:182-226): every I/D longer than max_size closes the window [pre, cur) on
// the target and opens the next one after it.  In op space, with Rx the reference offset of op i:
//   big(i)   = indel(i) && len(i) > max_size
//   pre(i)   = max over big j < i of (Rx(j) + reflen(j))          (0 if none; Rx is monotone)
//   piece(i) = big(i) && Rx(i) > pre(i)  ->  window [t_st + pre(i), t_st + Rx(i))
//   and a final window [t_st + pre(n), t_st + Rtot) if Rtot > pre(n).
// One wavefront per record, streaming; a prefix-sum (ref offsets, piece ordinals) and a prefix-max
// (pre) per 256-op step.  Run twice: count, then fill after the exclusive scan of the counts.
// ------------------------------------------------------------------------------------------------
struct rb_break_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const rb_norm_row *norm;
    const uint32_t *sched;
    uint64_t *hit_off;
    uint64_t *x_st, *x_en;
    uint64_t rows_cap;
    uint32_t max_size;
    int fill; // 0: count pieces; 1: write the windows of every record (or, with redo_only, of the records the collect pass gave up on);
              // 2 (collect): count AND keep the windows, in LDS while the record streams, then in tmp[] at a slot from tmp_cursor
    int redo_only;
    uint2 *tmp;                    // [rows_cap] (start, end) of a piece relative to the record's t_st
    uint64_t *tmp_off;             // [n_rec] where the record's pieces sit in tmp[]; ~0 = not kept (more than RB_BP_CAP pieces, or no room)
    unsigned long long *tmp_cursor; // one bump cursor per arena, 128 bytes apart (a single cursor would serialise every record at one L2 line)
    uint32_t n_arena;
    uint64_t arena_cap;             // slots of tmp[] per arena
    // list mode (break-paf in one walk: the records its clip kernel declined): wave w takes record list[w], w < *n_list
    const uint32_t *list;
    const unsigned long long *n_list;
};
#define RB_BP_CAP 512 // pieces of one record kept in LDS by the collect pass

__device__ __forceinline__ uint32_t rb_umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t rb_wave_scan_incl_max(uint32_t v) {
    v = rb_umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(1), 0xf, 0xf, false));
    v = rb_umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(2), 0xf, 0xf, false));
    v = rb_umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(4), 0xf, 0xf, false));
    v = rb_umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_SHR(8), 0xf, 0xf, false));
    v = rb_umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_BCAST15, 0xa, 0xf, false));
    v = rb_umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, RB_DPP_ROW_BCAST31, 0xc, 0xf, false));
    return v;
}

__device__ void rb_break_pieces_record(const rb_break_params &p, uint64_t wave, uint32_t r, uint2 *keep) {
    const int lane = rb_lane();
    const rb_norm_row *nr = &p.norm[r];
    const bool collect = p.fill == 2;
    if (p.redo_only && p.tmp_off[r] != ~0ull) return;
    if (nr->status != RB_ST_OK) {
        if (p.fill != 1 && lane == 0) {
            p.hit_off[r] = 0;
            if (collect) p.tmp_off[r] = 0;
        }
        return;
    }
    const uint32_t n = nr->n_ops;
    const uint64_t t_st = nr->t_st;
    const uint64_t rec0 = p.op_off[r] + nr->first_op;
    const uint64_t h0 = p.fill == 1 ? rb_first64(p.hit_off[r]) : 0;
    // 8 ops (32 contiguous bytes) per lane and step, two steps in flight in a statically indexed ring; loads past the
    // record's end re-read its last group and are masked on the last step (see k_liftover.hip)
    const uint64_t g0 = rec0 & ~3ull, gend = rec0 + n;
    const int32_t head = (int32_t)(rec0 - g0);
    const uint32_t n_steps = (uint32_t)((gend - g0 + 511u) >> 9);
    const uint32_t *__restrict__ gbase = p.ops + g0;
    const uint32_t last_off = (uint32_t)(((gend - 1u) & ~3ull) - g0);
    auto load_half = [&](uint32_t stp, uint32_t half) -> uint4 {
        uint32_t off = (stp << 9) + half * 4u + (uint32_t)lane * 8u;
        off = off < last_off ? off : last_off;
        return *reinterpret_cast<const uint4 *>(gbase + off);
    };
    uint32_t Rb = 0, pre = 0, cnt = 0; // ref bases so far, end of the last big indel, pieces so far
    uint32_t carry_last = RB_NULL_OP;  // last word of the step before
    uint4 pf[2][2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        pf[q][0] = load_half((uint32_t)q, 0u);
        pf[q][1] = load_half((uint32_t)q, 1u);
        __builtin_amdgcn_sched_barrier(0);
    }
    for (uint32_t st0 = 0; st0 < n_steps; st0 += 2) {
#pragma unroll
        for (int ring = 0; ring < 2; ring++) {
            const uint32_t st = st0 + (uint32_t)ring;
            if (st < n_steps) {
                const uint32_t raw[8] = {pf[ring][0].x, pf[ring][0].y, pf[ring][0].z, pf[ring][0].w,
                                         pf[ring][1].x, pf[ring][1].y, pf[ring][1].z, pf[ring][1].w};
                const int32_t idx0 = (int32_t)(st << 9) + lane * 8 - head;
                const bool edge = st == 0 || st + 1 == n_steps;
                uint32_t rl[8], Rx[8];
                bool big[8];
                uint32_t sr = 0;
                bool cont_here = false;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const bool valid = !edge || (uint32_t)(idx0 + q) < n;
                    const uint32_t opc = valid ? rb_opc(raw[q]) : RB_NULL_OP, len = valid ? rb_len(raw[q]) : 0u;
                    rl[q] = (opc <= 8u && rb_in(RB_REF_MASK, opc)) ? len : 0u;
                    big[q] = valid && (opc == RB_OP_I || opc == RB_OP_D) && len > p.max_size;
                    cont_here |= opc == RB_OP_CONT;
                }
                // continuation words (lengths of 2^28 and more, rb_device.h): the word walks as `hi << 28` more bases of its owner's
                // type, and an indel is long by its WHOLE length (liftover.rs:187-188) -- owner and continuation are marked together.
                // The word behind this step's last one is the first of the step already waiting in the ring.
                const uint32_t next_step_first = rb_readlane<uint32_t>(pf[ring ^ 1][0].x, 0);
                const uint32_t step_last = rb_readlane<uint32_t>(raw[7], 63);
                if (__ballot(cont_here) != 0ull || rb_opc(next_step_first) == RB_OP_CONT) {
                    const uint32_t prev_last = rb_prev_lane(raw[7], carry_last);
                    uint32_t next_first = (uint32_t)__shfl_down((int)raw[0], 1, 64);
                    if (lane == 63) next_first = next_step_first;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        const uint32_t i = (uint32_t)(idx0 + q);
                        const bool valid = i < n;
                        const uint32_t w = raw[q], c = rb_opc(w);
                        const uint32_t pw = q ? raw[q > 0 ? q - 1 : 0] : prev_last, nw = q < 7 ? raw[q < 7 ? q + 1 : 7] : next_first;
                        if (valid && c == RB_OP_CONT && i >= 1u && rb_opc(pw) <= 8u) {
                            const uint32_t type = rb_opc(pw), hi = (rb_len(w) & 15u) << RB_LEN_BITS_WORD;
                            rl[q] = rb_in(RB_REF_MASK, type) ? hi : 0u;
                            big[q] = (type == RB_OP_I || type == RB_OP_D) && rb_len(pw) + hi > p.max_size;
                        } else if (valid && (c == RB_OP_I || c == RB_OP_D) && i + 1u < n && rb_opc(nw) == RB_OP_CONT) {
                            big[q] = rb_len(w) + ((rb_len(nw) & 15u) << RB_LEN_BITS_WORD) > p.max_size;
                        }
                    }
                }
                carry_last = step_last;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    Rx[q] = sr;
                    sr += rl[q];
                }
                const uint32_t ir = rb_wave_scan_incl(sr);
                const uint32_t er = Rb + ir - sr;
                // lane-local: end of the last big indel inside the lane (absolute offsets are monotone)
                uint32_t lane_end = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    Rx[q] += er;
                    if (big[q]) lane_end = Rx[q] + rl[q];
                }
                const uint32_t im = rb_wave_scan_incl_max(lane_end);
                uint32_t run_pre = rb_umax(pre, rb_prev_lane(im, 0u)); // end of the last big indel before this lane
                uint32_t pc[8], pst[8];
                uint32_t lane_cnt = 0;
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const bool piece = big[q] && Rx[q] > run_pre;
                    pc[q] = piece ? 1u : 0u;
                    pst[q] = run_pre;
                    lane_cnt += pc[q];
                    if (big[q]) run_pre = Rx[q] + rl[q];
                }
                const uint32_t ic = rb_wave_scan_incl(lane_cnt);
                if (p.fill == 1) {
                    uint32_t ord = cnt + ic - lane_cnt;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        if (pc[q]) {
                            const uint64_t h = h0 + ord;
                            if (h < p.rows_cap) {
                                p.x_st[h] = t_st + pst[q];
                                p.x_en[h] = t_st + Rx[q];
                            }
                            ord++;
                        }
                    }
                } else if (collect && __ballot(lane_cnt != 0) != 0) {
                    uint32_t ord = cnt + ic - lane_cnt;
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        if (pc[q]) {
                            if (ord < RB_BP_CAP) keep[ord] = make_uint2(pst[q], Rx[q]);
                            ord++;
                        }
                    }
                }
                cnt += rb_readlane<uint32_t>(ic, 63);
                pre = rb_umax(pre, rb_readlane<uint32_t>(im, 63));
                Rb += rb_readlane<uint32_t>(ir, 63);
            }
            pf[ring][0] = load_half(st + 2u, 0u);
            pf[ring][1] = load_half(st + 2u, 1u);
        }
    }
    const bool last = Rb > pre; // liftover.rs:213-224
    if (p.fill == 1) {
        if (lane == 0 && last && h0 + cnt < p.rows_cap) {
            p.x_st[h0 + cnt] = t_st + pre;
            p.x_en[h0 + cnt] = t_st + Rb;
        }
        return;
    }
    const uint32_t total = cnt + (last ? 1u : 0u);
    if (lane == 0) p.hit_off[r] = (uint64_t)total;
    if (!collect) return;
    // the record's windows leave LDS for one slot of tmp[]; rb_k_break_place moves them to their rows once the row offsets exist
    uint64_t base = ~0ull;
    if (total <= RB_BP_CAP) {
        const uint32_t a = (uint32_t)(wave % p.n_arena);
        unsigned long long b0 = 0;
        if (lane == 0) b0 = atomicAdd(&p.tmp_cursor[(size_t)a * 16u], (unsigned long long)total);
        base = rb_first64(b0);
        base = base + total <= p.arena_cap ? (uint64_t)a * p.arena_cap + base : ~0ull; // (arena full: the second walk does this record)
    }
    if (base != ~0ull) {
        if (last && lane == 0) keep[cnt] = make_uint2(pre, Rb);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t j = (uint32_t)lane; j < total; j += 64) p.tmp[base + j] = keep[j];
    }
    if (lane == 0) p.tmp_off[r] = base;
}

// windows of the collect pass -> x_st / x_en at the record's row offset (thread per record: a record has a handful of pieces)
__global__ __launch_bounds__(256) void rb_k_break_place(rb_break_params p) {
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= p.n_rec) return;
    const uint64_t base = p.tmp_off[r];
    if (base == ~0ull) return;
    const rb_norm_row *nr = &p.norm[r];
    if (nr->status != RB_ST_OK) return;
    const uint64_t h0 = p.hit_off[r], n = p.hit_off[r + 1] - h0, t_st = nr->t_st;
    for (uint64_t j = 0; j < n; j++) {
        const uint2 w = p.tmp[base + j];
        if (h0 + j < p.rows_cap) {
            p.x_st[h0 + j] = t_st + w.x;
            p.x_en[h0 + j] = t_st + w.y;
        }
    }
}

__global__ __launch_bounds__(256) void rb_k_break_pieces(rb_break_params p) {
    __shared__ uint2 keep_all[4][RB_BP_CAP];
    uint2 *keep = keep_all[threadIdx.x >> 6];
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (p.list) { // a fixed grid strides over the list (its length is known on the device only)
        const uint64_t n_list = *p.n_list;
        for (uint64_t w = wave; w < n_list; w += (uint64_t)gridDim.x * 4u) rb_break_pieces_record(p, w, rb_first(p.list[w]), keep);
        return;
    }
    if (wave >= p.n_rec) return;
    rb_break_pieces_record(p, wave, rb_first(p.sched[wave]), keep);
}

extern "C" hipError_t rb_launch_break_pieces(const rb_break_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    const unsigned blocks = p->list ? (unsigned)std::min<uint64_t>((p->n_rec + 3) / 4, 1024) : (unsigned)((p->n_rec + 3) / 4);
    hipLaunchKernelGGL(rb_k_break_pieces, dim3(blocks), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t rb_launch_break_place(const rb_break_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_break_place, dim3((unsigned)((p->n_rec + 255) / 256)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// invert: cigar_swap_target_query (paf.rs:1050-1065): I <-> D, reversed when the strand is '-'
// ------------------------------------------------------------------------------------------------
struct rb_swap_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint8_t *strand;
    uint32_t *out_ops;
};
__global__ __launch_bounds__(256) void rb_k_swap(rb_swap_params p) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wave >= p.n_rec) return;
    const uint64_t o0 = p.op_off[wave], n = p.op_off[wave + 1] - o0;
    const bool minus = p.strand[wave] == (uint8_t)'-';
    for (uint64_t j = rb_lane(); j < n; j += 64) {
        const uint64_t src = minus ? n - 1 - j : j;
        uint32_t v = p.ops[o0 + src];
        if (minus) { // an op and its continuation word (rb_device.h) keep their order when the ops are reversed
            if (rb_opc(v) == RB_OP_CONT) {
                if (src > 0 && rb_opc(p.ops[o0 + src - 1]) != RB_OP_CONT) v = p.ops[o0 + src - 1];
            } else if (src + 1 < n && rb_opc(p.ops[o0 + src + 1]) == RB_OP_CONT) {
                v = p.ops[o0 + src + 1];
            }
        }
        const uint32_t opc = rb_opc(v);
        if (opc == RB_OP_I) v = (v & ~15u) | RB_OP_D;
        else if (opc == RB_OP_D) v = (v & ~15u) | RB_OP_I;
        p.out_ops[o0 + j] = v;
    }
}
extern "C" hipError_t rb_launch_swap(const rb_swap_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_swap, dim3((unsigned)((p->n_rec + 3) / 4)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// synthetic workload
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rb_k_synth(uint64_t seed, uint64_t first_record, uint64_t n_rec, const uint64_t *op_off, uint32_t *ops) {
    const uint64_t wave = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wave >= n_rec) return;
    const uint64_t o0 = op_off[wave], n = op_off[wave + 1] - o0;
    for (uint64_t j = rb_lane(); j < n; j += 64) ops[o0 + j] = rb_synth_op(seed, first_record + wave, j);
}
extern "C" hipError_t rb_launch_synth(uint64_t seed, uint64_t first_record, uint64_t n_rec, const uint64_t *op_off, uint32_t *ops, hipStream_t stream) {
    if (n_rec == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_synth, dim3((unsigned)((n_rec + 3) / 4)), dim3(256), 0, stream, seed, first_record, n_rec, op_off, ops);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// verification aid: order-sensitive digest of hit rows and the clipped CIGARs they point at
// (include/rustybam_amd.h, rb_dev_digest_rows).  One wavefront per row.
// ------------------------------------------------------------------------------------------------
struct rb_digest_params {
    const uint32_t *ops;     // the batch's packed ops (descriptor rows are expanded through them)
    const uint64_t *op_off;
    const rb_hit_row *rows;
    uint64_t n_rows;
    const uint32_t *out_ops;
    uint64_t row_base, rec_base;
    unsigned long long *digest;
};
__global__ __launch_bounds__(256) void rb_k_digest_rows(rb_digest_params p) {
    __shared__ unsigned long long part[4];
    const uint32_t wib = threadIdx.x >> 6;
    const uint64_t i = (uint64_t)blockIdx.x * 4u + wib;
    const int lane = rb_lane();
    unsigned long long contrib = 0;
    if (i < p.n_rows) {
        const rb_hit_row h = p.rows[i];
        const bool ok = h.status == RB_ST_OK;
        uint64_t hops = 0;
        if (ok) {
            const bool desc = (h.flags & RB_HIT_DESCRIPTOR) != 0;
            const uint32_t n = h.out_n;
            const uint32_t *src = p.out_ops + h.out_off;
            uint32_t fl = 0, ll = 0;
            if (desc) { // {first kept op in the record's original cigar, op count, first length, last length}
                const uint32_t first = src[0];
                fl = src[2], ll = src[3];
                src = p.ops + p.op_off[h.rec] + first;
            }
            for (uint32_t k = (uint32_t)lane; k < n; k += 64u) {
                uint32_t w = src[k];
                if (desc) {
                    const uint32_t len0 = rb_len(w);
                    uint32_t len = len0;
                    if (n == 1u && fl && ll) len = fl + ll - len0;
                    else {
                        if (k == 0 && fl) len = fl;
                        if (k == n - 1u && ll) len = ll;
                    }
                    w = (len << 4) | rb_opc(w);
                }
                hops += rb_splitmix64(((uint64_t)k << 32) | w);
            }
            hops = rb_wave_sum_u64(hops);
        }
        if (lane == 0) {
            uint64_t x = rb_splitmix64((uint64_t)h.rec + p.rec_base);
            x = rb_splitmix64(x ^ h.win);
            x = rb_splitmix64(x ^ (uint64_t)h.status);
            if (ok) {
                x = rb_splitmix64(x ^ ((h.flags & RB_HIT_INSIDE) ? 1ull : 0ull));
                x = rb_splitmix64(x ^ h.out_n);
                x = rb_splitmix64(x ^ h.t_st);
                x = rb_splitmix64(x ^ h.t_en);
                x = rb_splitmix64(x ^ h.q_st);
                x = rb_splitmix64(x ^ h.q_en);
                x = rb_splitmix64(x ^ h.nmatch);
                x = rb_splitmix64(x ^ h.aln_len);
                x = rb_splitmix64(x ^ hops);
            }
            contrib = x * (2ull * (p.row_base + i) + 1ull);
        }
    }
    if (lane == 0) part[wib] = contrib;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long s = part[0] + part[1] + part[2] + part[3];
        if (s) atomicAdd(p.digest, s);
    }
}
extern "C" hipError_t rb_launch_digest_rows(const rb_digest_params *p, hipStream_t stream) {
    if (p->n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_digest_rows, dim3((unsigned)((p->n_rows + 3) / 4)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

// ---- clips of a finished clip call, packed side by side in row order (host-buffer callers: the slots of out_ops mirror the input's op
//      positions and are as large as the batch; what goes over PCIe is the clips alone) ----------------------------------------------
struct rb_compact_params {
    uint64_t n_rows;
    rb_hit_row *rows;
    const uint32_t *src; // out_ops of the clip call
    uint64_t *off;       // [n_rows + 1] words per row, then their exclusive prefix
    uint32_t *dst;
    int fill;
};
__device__ __forceinline__ uint64_t rb_row_words(const rb_hit_row &h) {
    return h.status != RB_ST_OK ? 0ull : ((h.flags & RB_HIT_DESCRIPTOR) ? 4ull : (uint64_t)h.out_n);
}
__global__ __launch_bounds__(256) void rb_k_compact_clips(rb_compact_params p) {
    if (!p.fill) {
        const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < p.n_rows) p.off[i] = rb_row_words(p.rows[i]);
        return;
    }
    const uint64_t i = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (i >= p.n_rows) return;
    const uint64_t at = p.off[i], n = p.off[i + 1] - at;
    const uint32_t *src = p.src + p.rows[i].out_off;
    for (uint64_t j = rb_lane(); j < n; j += 64) p.dst[at + j] = src[j];
    __builtin_amdgcn_wave_barrier();
    if (rb_lane() == 0) { // (every lane has read out_off by now: the loads above were issued before this store in program order)
        p.rows[i].out_off = n ? at : 0ull;
        if (!n) p.rows[i].out_n = 0;
    }
}
extern "C" hipError_t rb_launch_compact_clips(const rb_compact_params *p, hipStream_t stream) {
    if (p->n_rows == 0) return hipSuccess;
    if (!p->fill) hipLaunchKernelGGL(rb_k_compact_clips, dim3((unsigned)((p->n_rows + 255) / 256)), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(rb_k_compact_clips, dim3((unsigned)((p->n_rows + 3) / 4)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// rb_dev_box_probe: what THIS box moves at the clip kernel's memory mix, without the clip kernel's instructions.  Every wave owns a
// 20 KiB stretch of `src` (a 5120-op record) and walks it in ten 2 KiB steps in the clip kernel's shape -- 32 contiguous bytes per lane as
// two 16-byte loads --, storing every step into slot 0 and every fifth one into slot 1 as well (about 1.2 bytes written per byte read:
// config 3's ratio).  Around the loop the wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz): the clock the chip holds
// under this memory load comes out with the time (bench.py's `box` block; tools/st_probe.hip is the stand-alone form of it).
// ------------------------------------------------------------------------------------------------
typedef uint32_t rb_bp_u32x4 __attribute__((ext_vector_type(4)));
// SCATTER: wave w takes stretch (w mod 4096) * (n / 4096) + w / 4096 -- the waves that run at the same time are then spread over the
// whole array, 5 MB apart, as the clip kernel's are (its records run longest first, i.e. in no memory order), instead of side by side.
// FLAT (scatter code 2 / 3): lane l takes bytes [16 l, 16 l + 16) and [1024 + 16 l, ...) of a 2 KiB step -- every instruction covers 1 KiB of
// whole lines -- instead of the clip kernel's 32 contiguous bytes per lane: what the other load shape would be worth on these buffers.
template <bool SCATTER, bool FLAT = false>
__global__ __launch_bounds__(256) void rb_k_box_probe(const char *__restrict__ src, char *__restrict__ d0, char *__restrict__ d1, uint64_t n_stretch,
                                                      uint32_t *stamps /* [3]: sum of cycles >> 6, sum of 10 ns ticks, stamped waves */,
                                                      int no_stores, int no_loads /* the read side / the write side of the mix alone */) {
    uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_stretch) return;
    if (SCATTER) {
        const uint64_t per = n_stretch / 4096u;
        if (per && w < per * 4096u) w = (w % 4096u) * per + w / 4096u;
    }
    const int lane = (int)(threadIdx.x & 63);
    const uint64_t base = w * (uint64_t)(10 * 2048) + (uint64_t)lane * (FLAT ? 16 : 32);
    constexpr uint64_t second = FLAT ? 1024 : 16;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll 2
    for (int s = 0; s < 10; s++) {
        const uint64_t o = base + (uint64_t)s * 2048;
        rb_bp_u32x4 a = {(uint32_t)s, (uint32_t)lane, 0u, 0u}, b = a;
        if (!no_loads) a = *(const rb_bp_u32x4 *)(src + o), b = *(const rb_bp_u32x4 *)(src + o + second);
        acc += a.x ^ b.y;
        if (no_stores) continue;
        *(rb_bp_u32x4 *)(d0 + o) = a;
        *(rb_bp_u32x4 *)(d0 + o + second) = b;
        if (s % 5 == 0) {
            *(rb_bp_u32x4 *)(d1 + o) = a;
            *(rb_bp_u32x4 *)(d1 + o + second) = b;
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0 && (w & 15) == 0) {
        atomicAdd(&stamps[0], (uint32_t)((c1 - c0) >> 6));
        atomicAdd(&stamps[1], (uint32_t)(r1 - r0));
        atomicAdd(&stamps[2], 1u);
    }
    if (acc == 0x12345678u) stamps[3] = acc; // (keeps the loads)
}
extern "C" hipError_t rb_launch_box_probe(const void *src, void *d0, void *d1, uint64_t n_stretch, uint32_t *stamps, int scatter, hipStream_t stream) {
    if (n_stretch == 0) return hipSuccess;
    const dim3 g((unsigned)((n_stretch + 3) / 4)), b(256);
    const int no_stores = (scatter & 4) != 0, no_loads = (scatter & 8) != 0; // (codes 4 .. 15: the read side / the write side alone)
    scatter &= 3;
    if (scatter == 3) hipLaunchKernelGGL((rb_k_box_probe<true, true>), g, b, 0, stream, (const char *)src, (char *)d0, (char *)d1, n_stretch, stamps, no_stores, no_loads);
    else if (scatter == 2) hipLaunchKernelGGL((rb_k_box_probe<false, true>), g, b, 0, stream, (const char *)src, (char *)d0, (char *)d1, n_stretch, stamps, no_stores, no_loads);
    else if (scatter) hipLaunchKernelGGL((rb_k_box_probe<true, false>), g, b, 0, stream, (const char *)src, (char *)d0, (char *)d1, n_stretch, stamps, no_stores, no_loads);
    else hipLaunchKernelGGL((rb_k_box_probe<false, false>), g, b, 0, stream, (const char *)src, (char *)d0, (char *)d1, n_stretch, stamps, no_stores, no_loads);
    return hipGetLastError();
}
