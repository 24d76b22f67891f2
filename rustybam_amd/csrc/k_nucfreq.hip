// k_nucfreq.hip -- A/C/G/T counts at every covered reference position (`rb nucfreq`), gfx950 / wave64.
//
// Replaces nucfreq::nucfreq (nucfreq.rs:61-95) over the reads region_nucfreq fetches (:111-125) for every 10 kb piece
// main.rs:100-110 cuts a region into.  The reference runs htslib's pileup: position by position, every read that covers
// the position resolves the position inside its CIGAR (sam.c resolve_cigar2) and contributes the base it aligns there
// unless the position falls in a D / N op.  Positions no read covers are not reported.
//
// Here the loop is turned inside out (partition by position range, not by read):
//   rb_k_nf_read_spans  wave per read: reference span of the CIGAR, the pileup's flag filter, the cases where htslib
//                       asserts, sortedness of (tid, pos); writes each read's end and a key (tid << 32 | end)
//   rb_k_nf_pmax_*      inclusive prefix MAXIMUM of the end keys in file order: the first read that can reach a position
//                       is then one binary search away (reads are sorted by start, not by end)
//   rb_k_nf_plan_tiles  thread per tile of NF_TILE positions: its region, and the range of reads that can overlap it
//   rb_k_nf_tiles       workgroup per tile: the tile's counters live in LDS (4 x u16 per position + a coverage
//                       difference array); each wave takes reads of the range in turn, walks the CIGAR 64 ops at a time
//                       (wave scans give every op its reference / query start), and for every match-type op that
//                       overlaps the tile the lanes stride over its bases: one byte load, one LDS atomic.  At the end
//                       a block scan of the difference array gives the depth (coverage + the htslib depth-cap check),
//                       and the tile is written out with 16-byte stores.
// HBM traffic: each read's packed bases once per tile it overlaps (4 bits / base), its CIGAR likewise, 16 B written per
// position.  Bound: HBM (the counters never leave LDS).
#include "rb_device.h"

#define NF_TILE 4096
#define NF_THREADS 512
#define NF_WAVES (NF_THREADS / 64)
#define NF_PLP_MASK (0x4u | 0x100u | 0x200u | 0x400u) // htslib BAM_DEF_MASK: UNMAP | SECONDARY | QCFAIL | DUP

struct rb_nf_params {
    uint64_t n_reads;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint8_t *seq;
    const uint64_t *seq_off;
    const uint32_t *l_seq;
    const int32_t *tid;
    const int64_t *pos;
    const uint32_t *flag;
    uint64_t n_regions;
    const int32_t *rg_tid;
    const uint64_t *rg_st, *rg_en, *out_off;
    uint32_t *counts;
    uint32_t *read_status;
    rb_nucfreq_counters *counters;
    // workspace
    uint64_t *end_key;  // [n_reads] tid << 32 | end, then its inclusive prefix maximum
    uint32_t *rd_end;   // [n_reads] end of each read (exclusive), 0 for reads that take no part
    uint64_t *tile_off; // [n_regions + 1] exclusive prefix of tiles per region
    uint64_t *blk;      // block partials of the scans
    uint64_t *tile_lo, *tile_hi; // [max_tiles] reads that can overlap the tile
    uint64_t max_tiles;
};

__device__ __forceinline__ uint64_t nf_key(int32_t tid, uint64_t pos32) { return ((uint64_t)(uint32_t)tid << 32) | (pos32 & 0xFFFFFFFFull); }

// ---- per-read spans ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rb_k_nf_read_spans(rb_nf_params p) {
    const uint64_t i = (uint64_t)blockIdx.x * 4u + rb_first(threadIdx.x >> 6);
    if (i >= p.n_reads) return;
    const int lane = rb_lane();
    const int32_t tid = p.tid[i];
    const int64_t pos = p.pos[i];
    const uint32_t flag = p.flag[i];
    const uint64_t o0 = p.op_off[i], o1 = p.op_off[i + 1];
    const bool filtered = tid < 0 || (flag & NF_PLP_MASK) != 0; // bam_plp_push: tid < 0 or a masked flag never enters the pileup
    uint64_t ref = 0;
    uint32_t bad = 0, any_ref = 0;
    for (uint64_t o = o0 + (uint64_t)lane; o < o1; o += 64) {
        const uint32_t w = p.ops[o], c = rb_opc(w), l = rb_len(w);
        if (rb_in(RB_REF_MASK, c)) {
            ref += l;
            any_ref = 1;
            if (l == 0) bad = 1; // (resolve_cigar2 steps one op per position: a zero-length M / D / N / = / X is not walked the way it reads)
        }
        if (c > 8u) bad = 1;
    }
    ref = rb_wave_sum_u64(ref);
    bad = rb_wave_or_u32(bad);
    any_ref = rb_wave_or_u32(any_ref);
    // the cases in which htslib's cursor runs into an assertion (or off the cigar): no op at all, no reference-consuming op, a lone
    // op that is not M / = / X; plus what the 32-bit arithmetic of the tile kernel cannot hold
    if (o1 == o0 || !any_ref) bad = 1;
    if (o1 - o0 == 1 && !rb_in(RB_MATCH_MASK, rb_opc(p.ops[o0]))) bad = 1;
    if (pos < 0 || pos > 0x7FFFFFFFll || ref > 0x7FFFFFFFull) bad = 1;
    const uint32_t status = filtered ? RB_RD_FILTERED : (bad ? RB_RD_BAD_CIGAR : RB_RD_OK);
    if (lane == 0) {
        uint64_t end = (uint64_t)pos + ref;
        if (end > 0xFFFFFFFFull) end = 0xFFFFFFFFull;
        p.read_status[i] = status;
        p.rd_end[i] = status == RB_RD_OK ? (uint32_t)end : 0u;
        p.end_key[i] = status == RB_RD_OK ? nf_key(tid, end) : 0ull;
        if (status == RB_RD_BAD_CIGAR) atomicAdd((unsigned long long *)&p.counters->n_bad, 1ull);
        if (i > 0 && nf_key(tid, (uint64_t)pos) < nf_key(p.tid[i - 1], (uint64_t)p.pos[i - 1])) p.counters->unsorted = 1;
    }
}

// ---- inclusive prefix maximum (u64), three launches ----------------------------------------------------------------------
#define NF_SCAN_PER_BLOCK 2048
__device__ __forceinline__ uint64_t nf_wave_max_incl(uint64_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t o = __shfl_up(v, d, 64);
        if (lane >= d) v = v > o ? v : o;
    }
    return v;
}
__global__ __launch_bounds__(256) void rb_k_nf_pmax_partial(const uint64_t *v, uint64_t n, uint64_t *blk) {
    __shared__ uint64_t sh[4];
    const uint64_t base = (uint64_t)blockIdx.x * NF_SCAN_PER_BLOCK;
    uint64_t m = 0;
    for (uint32_t k = threadIdx.x; k < NF_SCAN_PER_BLOCK; k += 256) {
        const uint64_t i = base + k;
        if (i < n) m = m > v[i] ? m : v[i];
    }
    m = nf_wave_max_incl(m, rb_lane());
    if (rb_lane() == 63) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t a = sh[0];
        for (int w = 1; w < 4; w++) a = a > sh[w] ? a : sh[w];
        blk[blockIdx.x] = a;
    }
}
__global__ __launch_bounds__(64) void rb_k_nf_pmax_top(uint64_t *blk, uint64_t n_blocks) { // exclusive prefix maximum of the block maxima
    const int lane = rb_lane();
    uint64_t carry = 0;
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += 64) {
        const uint64_t i = b0 + (uint64_t)lane;
        const uint64_t v = i < n_blocks ? blk[i] : 0;
        uint64_t inc = nf_wave_max_incl(v, lane);
        inc = inc > carry ? inc : carry;
        uint64_t exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = carry;
        if (i < n_blocks) blk[i] = exc;
        carry = __shfl(inc, 63, 64);
    }
}
__global__ __launch_bounds__(256) void rb_k_nf_pmax_apply(uint64_t *v, uint64_t n, const uint64_t *blk) {
    __shared__ uint64_t sh[4];
    const uint64_t base = (uint64_t)blockIdx.x * NF_SCAN_PER_BLOCK + (uint64_t)threadIdx.x * 8u;
    uint64_t x[8];
    uint64_t m = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        x[k] = (base + k < n) ? v[base + k] : 0;
        m = m > x[k] ? m : x[k];
        x[k] = m;
    }
    const int lane = rb_lane(), w = (int)(threadIdx.x >> 6);
    const uint64_t inc = nf_wave_max_incl(m, lane);
    if (lane == 63) sh[w] = inc;
    __syncthreads();
    uint64_t before = blk[blockIdx.x];
    for (int k = 0; k < w; k++) before = before > sh[k] ? before : sh[k];
    uint64_t prev = __shfl_up(inc, 1, 64);
    if (lane == 0) prev = 0;
    before = before > prev ? before : prev;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (base + k < n) v[base + k] = x[k] > before ? x[k] : before;
}

// ---- tiles ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rb_k_nf_tile_count(rb_nf_params p) {
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= p.n_regions) return;
    const uint64_t st = p.rg_st[r], en = p.rg_en[r];
    p.tile_off[r] = en > st ? (en - st + NF_TILE - 1) / NF_TILE : 0;
}

struct nf_tile {
    uint64_t r, st, en, out;
    int32_t tid;
};
__device__ __forceinline__ nf_tile nf_tile_of(const rb_nf_params &p, uint64_t t) {
    uint64_t lo = 0, hi = p.n_regions; // last region with tile_off[r] <= t
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (p.tile_off[mid] <= t) lo = mid; else hi = mid;
    }
    nf_tile T;
    T.r = lo;
    T.tid = p.rg_tid[lo];
    const uint64_t rst = p.rg_st[lo], ren = p.rg_en[lo];
    T.st = rst + (t - p.tile_off[lo]) * NF_TILE;
    T.en = T.st + NF_TILE < ren ? T.st + NF_TILE : ren;
    T.out = p.out_off[lo] + (T.st - rst);
    return T;
}

__global__ __launch_bounds__(256) void rb_k_nf_plan_tiles(rb_nf_params p) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= p.tile_off[p.n_regions]) return;
    const nf_tile T = nf_tile_of(p, t);
    uint64_t lo = 0, hi = 0;
    if (T.tid >= 0 && T.st <= 0xFFFFFFFEull && p.n_reads) {
        const uint64_t k_st = nf_key(T.tid, T.st);
        const uint64_t k_en = nf_key(T.tid, T.en > 0xFFFFFFFFull ? 0xFFFFFFFFull : T.en);
        // first read whose prefix maximum of (tid, end) exceeds (tid, st): nothing before it reaches the tile
        uint64_t a = 0, b = p.n_reads;
        while (a < b) {
            const uint64_t mid = (a + b) >> 1;
            if (p.end_key[mid] > k_st) b = mid; else a = mid + 1;
        }
        lo = a;
        // first read that starts at or behind the tile's end (reads are sorted by (tid, pos))
        a = lo, b = p.n_reads;
        while (a < b) {
            const uint64_t mid = (a + b) >> 1;
            if (nf_key(p.tid[mid], (uint64_t)p.pos[mid]) >= k_en) b = mid; else a = mid + 1;
        }
        hi = a;
    }
    p.tile_lo[t] = lo;
    p.tile_hi[t] = hi;
}

__global__ __launch_bounds__(NF_THREADS) void rb_k_nf_tiles(rb_nf_params p) {
    __shared__ uint32_t cnt[NF_TILE * 2]; // per position: A | C << 16, G | T << 16
    __shared__ int32_t diff[NF_TILE + 8]; // +1 where a read starts covering, -1 where it stops; then the depth
    __shared__ int32_t wsum[NF_WAVES];
    __shared__ uint32_t blk_max, blk_cov;
    const uint64_t t = blockIdx.x;
    if (t >= p.tile_off[p.n_regions]) return;
    const nf_tile T = nf_tile_of(p, t);
    const uint32_t n_pos = (uint32_t)(T.en - T.st);
    for (uint32_t k = threadIdx.x; k < NF_TILE * 2; k += NF_THREADS) cnt[k] = 0;
    for (uint32_t k = threadIdx.x; k < NF_TILE + 8; k += NF_THREADS) diff[k] = 0;
    if (threadIdx.x == 0) blk_max = 0, blk_cov = 0;
    __syncthreads();
    const uint32_t wib = rb_first(threadIdx.x >> 6);
    const int lane = rb_lane();
    const uint64_t lo = p.tile_lo[t], hi = p.tile_hi[t];
    for (uint64_t i = lo + wib; i < hi; i += NF_WAVES) {
        const uint32_t rs = p.read_status[i];
        if (rs == RB_RD_FILTERED || rs == RB_RD_BAD_CIGAR) continue;
        const int64_t pos = p.pos[i];
        const uint64_t rend = p.rd_end[i];
        if (p.tid[i] != T.tid || (uint64_t)pos >= T.en || rend <= T.st) continue; // hts_itr_next: pos < en && endpos > st
        const uint64_t c0 = (uint64_t)pos > T.st ? (uint64_t)pos : T.st, c1 = rend < T.en ? rend : T.en;
        if (lane == 0) {
            atomicAdd(&diff[(uint32_t)(c0 - T.st)], 1);
            atomicAdd(&diff[(uint32_t)(c1 - T.st)], -1);
        }
        const uint64_t o0 = p.op_off[i], o1 = p.op_off[i + 1];
        const uint8_t *__restrict__ sq = p.seq + p.seq_off[i];
        const uint32_t lseq = p.l_seq[i];
        const int64_t rel_st = (int64_t)T.st - pos, rel_en = (int64_t)T.en - pos; // the tile in read-relative reference offsets
        const int64_t idx0 = pos - (int64_t)T.st;                                  // tile index of the read's first base
        uint32_t R = 0, Q = 0;
        bool seq_short = false;
        for (uint64_t o = o0; o < o1; o += 64) {
            const uint32_t w = (o + (uint64_t)lane < o1) ? p.ops[o + (uint64_t)lane] : RB_NULL_OP;
            const uint32_t c = rb_opc(w), len = rb_len(w);
            const uint32_t rl = rb_in(RB_REF_MASK, c) ? len : 0u, ql = rb_in(RB_QRY_MASK, c) ? len : 0u;
            const uint32_t ir = rb_wave_scan_incl(rl), iq = rb_wave_scan_incl(ql);
            const uint32_t r0 = R + ir - rl, q0 = Q + iq - ql; // where my op starts on the reference / in the read
            R += rb_readlane<uint32_t>(ir, 63);
            Q += rb_readlane<uint32_t>(iq, 63);
            if ((int64_t)R <= rel_st) continue; // these 64 ops end before the tile
            const bool hit = rb_in(RB_MATCH_MASK, c) && (int64_t)r0 < rel_en && (int64_t)r0 + (int64_t)len > rel_st;
            uint64_t m = __ballot(hit);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t jr0 = rb_readlane<uint32_t>(r0, j), jlen = rb_readlane<uint32_t>(len, j), jq0 = rb_readlane<uint32_t>(q0, j);
                const int64_t a = (int64_t)jr0 > rel_st ? (int64_t)jr0 : rel_st;
                const int64_t b = (int64_t)jr0 + (int64_t)jlen < rel_en ? (int64_t)jr0 + (int64_t)jlen : rel_en;
                for (int64_t x = a + lane; x < b; x += 64) {
                    const uint32_t q = jq0 + (uint32_t)(x - (int64_t)jr0); // bam_pileup1_t::qpos
                    if (q >= lseq) { // record().seq()[qpos] is out of bounds: the reference panics
                        seq_short = true;
                        continue;
                    }
                    const uint32_t nib = ((uint32_t)sq[q >> 1] >> ((~q & 1u) << 2)) & 15u;
                    if (nib != 0u && (nib & (nib - 1u)) == 0u) { // 1 2 4 8 = A C G T (nucfreq.rs:83-90); N and the IUPAC codes count nothing
                        const uint32_t bs = (uint32_t)__builtin_ctz(nib);
                        atomicAdd(&cnt[(uint32_t)(idx0 + x) * 2u + (bs >> 1)], 1u << ((bs & 1u) << 4));
                    }
                }
            }
            if ((int64_t)R >= rel_en) break;
        }
        if (__ballot(seq_short) != 0 && lane == 0) p.read_status[i] = RB_RD_SEQ_SHORT;
    }
    __syncthreads();
    // depth = prefix sum of the difference array: 8 positions per thread
    {
        const uint32_t b0 = threadIdx.x * 8u;
        int32_t d[8], s = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            s += diff[b0 + k];
            d[k] = s;
        }
        const int32_t inc = (int32_t)rb_wave_scan_incl((uint32_t)s);
        if (lane == 63) wsum[wib] = inc;
        __syncthreads();
        int32_t before = inc - s;
        for (uint32_t k = 0; k < wib; k++) before += wsum[k];
        int32_t mx = 0;
        uint32_t cov = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int32_t dep = before + d[k];
            diff[b0 + k] = dep;
            if (b0 + k < n_pos) {
                mx = mx > dep ? mx : dep;
                cov += dep > 0;
            }
        }
        atomicMax(&blk_max, (uint32_t)mx);
        atomicAdd(&blk_cov, cov);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax((unsigned long long *)&p.counters->max_depth, (unsigned long long)blk_max);
        atomicAdd((unsigned long long *)&p.counters->n_covered, (unsigned long long)blk_cov);
    }
    uint4 *__restrict__ out = reinterpret_cast<uint4 *>(p.counts + 4ull * T.out);
    for (uint32_t k = threadIdx.x; k < n_pos; k += NF_THREADS) {
        const uint32_t c01 = cnt[2 * k], c23 = cnt[2 * k + 1];
        out[k] = make_uint4((c01 & 0xFFFFu) | (diff[k] > 0 ? RB_NF_COVERED : 0u), c01 >> 16, c23 & 0xFFFFu, c23 >> 16);
    }
}

extern "C" size_t rb_nf_tile_positions(void) { return NF_TILE; }
extern "C" size_t rb_nf_scan_blocks(uint64_t n) { return (size_t)((n + NF_SCAN_PER_BLOCK - 1) / NF_SCAN_PER_BLOCK); }

extern "C" hipError_t rb_launch_exclusive_scan(uint64_t *v, uint64_t n, uint64_t *block_sums, uint64_t *total_out, hipStream_t stream);

extern "C" hipError_t rb_launch_nucfreq(const rb_nf_params *pp, hipStream_t stream) {
    const rb_nf_params p = *pp;
    if (p.n_reads) {
        hipLaunchKernelGGL(rb_k_nf_read_spans, dim3((unsigned)((p.n_reads + 3) / 4)), dim3(256), 0, stream, p);
        const uint64_t nb = rb_nf_scan_blocks(p.n_reads);
        hipLaunchKernelGGL(rb_k_nf_pmax_partial, dim3((unsigned)nb), dim3(256), 0, stream, p.end_key, p.n_reads, p.blk);
        hipLaunchKernelGGL(rb_k_nf_pmax_top, dim3(1), dim3(64), 0, stream, p.blk, nb);
        hipLaunchKernelGGL(rb_k_nf_pmax_apply, dim3((unsigned)nb), dim3(256), 0, stream, p.end_key, p.n_reads, p.blk);
    }
    if (!p.n_regions || !p.max_tiles) return hipGetLastError();
    hipLaunchKernelGGL(rb_k_nf_tile_count, dim3((unsigned)((p.n_regions + 255) / 256)), dim3(256), 0, stream, p);
    hipError_t e = rb_launch_exclusive_scan(p.tile_off, p.n_regions, p.blk, p.tile_off + p.n_regions, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rb_k_nf_plan_tiles, dim3((unsigned)((p.max_tiles + 255) / 256)), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(rb_k_nf_tiles, dim3((unsigned)p.max_tiles), dim3(NF_THREADS), 0, stream, p);
    return hipGetLastError();
}
