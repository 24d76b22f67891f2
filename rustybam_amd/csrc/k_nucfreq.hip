// k_nucfreq.hip -- A/C/G/T counts at every covered reference position (`rb nucfreq`), gfx950 / wave64.
//
// Replaces nucfreq::nucfreq (nucfreq.rs:61-95) over the reads region_nucfreq fetches (:111-125) for every 10 kb piece
// main.rs:100-110 cuts a region into.  The reference runs htslib's pileup: position by position, every read that covers
// the position resolves the position inside its CIGAR (sam.c resolve_cigar2) and contributes the base it aligns there
// unless the position falls in a D / N op.  Positions no read covers are not reported.
//
// Here the loop is turned inside out (partition by position range, not by read):
//   rb_k_nf_read_spans  lane per read (the whole wave on a long cigar): reference span of the CIGAR, the pileup's flag filter, the
//                       cases where htslib asserts, sortedness of (tid, pos); writes each read's record and a key (tid << 32 | end)
//   rb_k_nf_pmax_*      inclusive prefix MAXIMUM of the end keys in file order: the first read that can reach a position
//                       is then one binary search away (reads are sorted by start, not by end)
//   rb_k_nf_plan_tiles  thread per tile of NF_TILE positions: its region, and the range of reads that can overlap it
//   rb_k_nf_crowded / _deep_regions / _admit   htslib's cap of 8000 buffered reads, replayed per region only where it can be reached:
//                       a bitmap of the reads each such region's fetch drops (see the comment at rb_k_nf_admit)
//   rb_k_nf_tiles       workgroup per tile: the tile's counters live in LDS (4 x u16 in one u64 per position, or 4 bytes in one dword
//                       where at most 255 reads are in range: then one ds_add_u64 covers two positions; + a coverage
//                       difference array).  Each wave takes every NF_WAVES-th read of the tile's range: the reads' records sit one
//                       per lane, and the reads go by as a stream of chunks of 64 ops, three under way (round 6: the bases of one
//                       parked in LDS while the next is scanned -- wave scans give every op its reference / query start, all lanes
//                       at once their op's share of the tile -- and the one after has its ops requested).  For every match-type op
//                       that overlaps the tile each lane takes 8 consecutive positions: two dwords of the staged bases give a lane
//                       its 8 base codes, a 256-entry table turns two codes into what they add, one ds_add_u64 per two bases (no
//                       branch on the base: N and the IUPAC codes add 0).  At the end a block scan of the difference array gives
//                       the depth (coverage + the htslib depth-cap check), and the tile is written out with 16-byte stores.
// HBM traffic: each read's packed bases once per tile it overlaps (4 bits / base), its CIGAR likewise, 16 B written per
// position.  Bound: HBM (the counters never leave LDS).
#include "rb_device.h"
#include <algorithm>

#ifndef NF_TILE
#define NF_TILE 4096
#endif
#define NF_STAGE_DW (NF_TILE / 8 + 256) // dwords of packed bases staged per read and tile: the tile itself + 2048 inserted bases
#ifndef NF_THREADS
#define NF_THREADS 512
#endif
#define NF_PER_THREAD (NF_TILE / NF_THREADS) // positions per thread in the depth scan
#define NF_WAVES (NF_THREADS / 64)
#define NF_LANE_OPS 4u // reads with at most this many ops are walked by one lane each where a tile is crowded
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
    struct nf_read *hd; // [n_reads] what the tile kernel needs of a read, in one 48-byte record
    uint64_t *tile_off; // [n_regions + 1] exclusive prefix of tiles per region
    uint64_t *blk;      // block partials of the scans
    uint64_t *tile_lo, *tile_hi; // [max_tiles] reads that can overlap the tile
    uint64_t max_tiles;
    // htslib's cap on buffered reads (rb_k_nf_admit): per region the bit offset of its dropped-read bitmap in drop_bits (~0: none)
    uint64_t *drop_off;   // [n_regions]
    uint64_t *drop_bits;  // pool of drop_words 64-bit words; drop_bits[-1] is the pool's cursor
    uint64_t drop_words;
    uint32_t *deep_list;  // [n_regions + 1] regions whose fetch holds more reads than the cap; [n_regions] = how many
    uint32_t flags;       // bit 0: 16-bit counters for every tile (diagnostic)
    struct nf_tdesc *tdesc; // [max_tiles] what a workgroup needs to know about a tile, in one 64-byte record (rb_k_nf_tile_desc)
    uint32_t *wide_list;    // [max_tiles + 2] the tiles of the two builds that walk lists (rb_k_nf_tile_desc): [0] = how many of kind 2, then which,
                            // upwards from [1]; [max_tiles + 1] = how many of kind 0, then which, downwards from [max_tiles]
};

// one read as the tile kernel sees it: a single 48-byte record (one scalar load) instead of eight arrays
struct __attribute__((aligned(16))) nf_read {
    uint32_t pos, end; // reference span [pos, end); end = 0 for reads that take no part (filtered, malformed)
    int32_t tid;
    uint32_t l_seq;
    uint64_t op_off;
    uint32_t n_ops, pad0;
    uint64_t nib0; // index of the read's first base counted in 4-bit units from seq
    uint64_t pad1;
};

__device__ __forceinline__ uint64_t nf_key(int32_t tid, uint64_t pos32) { return ((uint64_t)(uint32_t)tid << 32) | (pos32 & 0xFFFFFFFFull); }

// ---- per-read spans ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rb_k_nf_read_spans(rb_nf_params p) {
    // a wave takes 64 consecutive reads, one per lane (coalesced loads of the per-read arrays); a lane walks a short cigar itself,
    // the reads with longer ones are then done one after the other by the whole wave
    const uint64_t i = ((uint64_t)blockIdx.x * 4u + rb_first(threadIdx.x >> 6)) * 64u + (uint64_t)rb_lane();
    const int lane = rb_lane();
    const bool in = i < p.n_reads;
    const int32_t tid = in ? p.tid[i] : -1;
    const int64_t pos = in ? p.pos[i] : 0;
    const uint32_t flag = in ? p.flag[i] : 0u;
    const uint64_t o0 = in ? p.op_off[i] : 0, o1 = in ? p.op_off[i + 1] : 0;
    const bool filtered = tid < 0 || (flag & NF_PLP_MASK) != 0; // bam_plp_push: tid < 0 or a masked flag never enters the pileup
    uint64_t ref = 0;
    uint32_t bad = 0, any_ref = 0;
    auto one_op = [](uint32_t w, uint64_t &r, uint32_t &b, uint32_t &a) {
        const uint32_t c = rb_opc(w), l = rb_len(w);
        if (rb_in(RB_REF_MASK, c)) {
            r += l;
            a = 1;
            if (l == 0) b = 1; // (resolve_cigar2 steps one op per position: a zero-length M / D / N / = / X is not walked the way it reads)
        }
        if (c > 8u) b = 1;
    };
    const bool short_cigar = in && o1 - o0 <= 8u;
    if (short_cigar)
        for (uint64_t o = o0; o < o1; o++) one_op(p.ops[o], ref, bad, any_ref);
    uint64_t cm = __ballot(in && !short_cigar);
    while (cm) {
        const int l = __builtin_ctzll(cm);
        cm &= cm - 1;
        const uint64_t a0 = rb_readlane<uint64_t>(o0, l), a1 = rb_readlane<uint64_t>(o1, l);
        uint64_t r = 0;
        uint32_t b = 0, a = 0;
        for (uint64_t o = a0 + (uint64_t)lane; o < a1; o += 64) one_op(p.ops[o], r, b, a);
        r = rb_wave_sum_u64(r), b = rb_wave_or_u32(b), a = rb_wave_or_u32(a);
        if (lane == l) ref = r, bad = b, any_ref = a;
    }
    if (!in) return;
    // the cases in which htslib's cursor runs into an assertion (or off the cigar): no op at all, no reference-consuming op, a lone
    // op that is not M / = / X; plus what the 32-bit arithmetic of the tile kernel cannot hold
    if (o1 == o0 || !any_ref) bad = 1;
    if (o1 - o0 == 1 && !rb_in(RB_MATCH_MASK, rb_opc(p.ops[o0]))) bad = 1;
    if (pos < 0 || pos > 0x7FFFFFFFll || ref > 0x7FFFFFFFull || o1 - o0 > 0x7FFFFFFFull) bad = 1;
    const uint32_t status = filtered ? RB_RD_FILTERED : (bad ? RB_RD_BAD_CIGAR : RB_RD_OK);
    uint64_t end = (uint64_t)pos + ref;
    if (end > 0xFFFFFFFFull) end = 0xFFFFFFFFull;
    p.read_status[i] = status;
    nf_read h;
    h.pos = (uint32_t)pos, h.end = status == RB_RD_OK ? (uint32_t)end : 0u;
    h.tid = tid, h.l_seq = p.l_seq[i];
    h.op_off = o0, h.n_ops = (uint32_t)(o1 - o0 < 0xFFFFFFFFull ? o1 - o0 : 0xFFFFFFFFull), h.pad0 = 0;
    h.nib0 = 2ull * p.seq_off[i], h.pad1 = 0;
    p.hd[i] = h;
    p.end_key[i] = status == RB_RD_OK ? nf_key(tid, end) : 0ull;
    if (status == RB_RD_BAD_CIGAR) atomicAdd((unsigned long long *)&p.counters->n_bad, 1ull);
    if (i > 0 && nf_key(tid, (uint64_t)pos) < nf_key(p.tid[i - 1], (uint64_t)p.pos[i - 1])) p.counters->unsorted = 1;
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

// ---- htslib's cap on buffered reads (bam_plp_push: `iter->tid == b->core.tid && iter->pos == b->core.pos && iter->mp->cnt >
//      iter->maxcnt` drops the read; maxcnt = 8000, htslib sam.c) --------------------------------------------------------------------
// The iterator only stands at a read's own start when an earlier read of the fetch starts there too, so the first read of a
// start position always enters.  The j-th (j >= 2) enters unless 8000 reads are buffered: those admitted before it in this fetch
// whose end is >= the position (a read is released while the position behind its end is processed, and the position the iterator
// stands on has not been).  Admission is per fetch -- per region -- and sequential in file order, so it only runs for regions whose
// fetch holds more than 8000 reads at all: one wavefront per such region, the ends of the buffered reads in LDS.
#define NFA_LIVE 15360u // buffered reads the wave can hold (8000 + one per start position reached at the cap)
// Can the cap be reached at all?  An upper bound of what is buffered when read i arrives, whatever the fetch: the reads in front of
// it in the file down to the first one whose prefix maximum of (tid, end) reaches (tid, pos) -- everything before that one has
// ended.  Only if the bound reaches the cap somewhere do the per-region simulations below run.
__global__ __launch_bounds__(256) void rb_k_nf_crowded(rb_nf_params p) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= p.n_reads || i < RB_NF_DEPTH_CAP) return;
    const nf_read h = p.hd[i];
    if (!h.end) return;
    const uint64_t key = nf_key(h.tid, h.pos);
    if (p.end_key[i - RB_NF_DEPTH_CAP] >= key) p.deep_list[p.n_regions + 1] = 1u; // (8000 reads back something may still be open)
}
__global__ __launch_bounds__(256) void rb_k_nf_deep_regions(rb_nf_params p) {
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= p.n_regions) return;
    p.drop_off[r] = ~0ull;
    if (!p.deep_list[p.n_regions + 1]) return;
    const uint64_t t0 = p.tile_off[r], t1 = p.tile_off[r + 1];
    if (t0 == t1) return;
    const uint64_t lo = p.tile_lo[t0], hi = p.tile_hi[t1 - 1];
    if (hi > lo && hi - lo > RB_NF_DEPTH_CAP) p.deep_list[atomicAdd(&p.deep_list[p.n_regions], 1u)] = (uint32_t)r;
}

__global__ __launch_bounds__(64) void rb_k_nf_admit(rb_nf_params p) {
    __shared__ uint32_t live[NFA_LIVE];
    const int lane = rb_lane();
    const uint64_t lt = (1ull << lane) - 1ull;
    const uint32_t n_deep = p.deep_list[p.n_regions];
    for (uint32_t k = blockIdx.x; k < n_deep; k += gridDim.x) {
        const uint64_t r = p.deep_list[k];
        const uint64_t t0 = p.tile_off[r], t1 = p.tile_off[r + 1];
        const uint64_t rlo = p.tile_lo[t0], rhi = p.tile_hi[t1 - 1];
        const int32_t rtid = p.rg_tid[r];
        const uint64_t st = p.rg_st[r], en = p.rg_en[r];
        const uint64_t words = (rhi - rlo + 63) >> 6;
        uint64_t off = 0;
        if (lane == 0) off = atomicAdd((unsigned long long *)(p.drop_bits - 1), (unsigned long long)words);
        off = rb_first64(off);
        if (off + words > p.drop_words) { // (the pool holds 64 regions per read: only regions that overlap en masse get here)
            if (lane == 0) p.counters->cap_overflow = 1;
            continue;
        }
        uint32_t n_live = 0, quota = 0, n_drop = 0;
        int64_t cur_p = -1;
        bool overflow = false;
        for (uint64_t b0 = rlo; b0 < rhi; b0 += 64) { // 64 reads of the fetch = one word of the bitmap
            const uint64_t i = b0 + (uint64_t)lane;
            uint32_t pos = 0, end = 0;
            bool cand = false;
            if (i < rhi) { // hts_itr_next: same contig, pos < en, endpos > st; filtered / unusable reads carry end = 0
                const nf_read h = p.hd[i];
                pos = h.pos, end = h.end;
                cand = h.tid == rtid && (uint64_t)h.pos < en && (uint64_t)h.end > st;
            }
            uint64_t cm = __ballot(cand), dropm = 0;
            while (cm) {
                const uint32_t gp = rb_readlane<uint32_t>(pos, __builtin_ctzll(cm));
                const bool mine = cand && pos == gp;
                const uint64_t grp = __ballot(mine);
                if ((int64_t)gp != cur_p) {
                    // a new start position: release what ended before it (compaction in place: a chunk is read before anything
                    // of it is overwritten), the rest is what the iterator holds when the second read of this position arrives
                    uint32_t wi = 0;
                    for (uint32_t c0 = 0; c0 < n_live; c0 += 64) {
                        const bool in = c0 + (uint32_t)lane < n_live;
                        const uint32_t e = in ? live[c0 + (uint32_t)lane] : 0u;
                        const bool keep = in && e >= gp;
                        const uint64_t km = __ballot(keep);
                        if (keep) live[wi + (uint32_t)__builtin_popcountll(km & lt)] = e;
                        wi += (uint32_t)__builtin_popcountll(km);
                    }
                    n_live = wi;
                    cur_p = gp;
                    quota = n_live < RB_NF_DEPTH_CAP ? RB_NF_DEPTH_CAP - n_live : 1u; // the first read of a position is never at it
                }
                const uint32_t g = (uint32_t)__builtin_popcountll(grp), rank = (uint32_t)__builtin_popcountll(grp & lt);
                const uint32_t a = g < quota ? g : quota;
                const bool admit = mine && rank < quota;
                if (n_live + a > NFA_LIVE) overflow = true;
                else if (admit) live[n_live + rank] = end;
                if (!overflow) n_live += a;
                quota -= a;
                dropm |= __ballot(mine && !admit);
                cm &= ~grp;
            }
            if (lane == 0) p.drop_bits[off + ((b0 - rlo) >> 6)] = dropm;
            n_drop += (uint32_t)__builtin_popcountll(dropm);
        }
        if (lane == 0) {
            if (overflow) p.counters->cap_overflow = 1;
            p.drop_off[r] = off * 64ull;
            if (n_drop) atomicAdd((unsigned long long *)&p.counters->n_dropped, (unsigned long long)n_drop);
        }
    }
}

struct nf_drop {
    uint64_t off, rlo;
    const uint64_t *bits;
};
__device__ __forceinline__ nf_drop nf_drop_of(const rb_nf_params &p, const nf_tile &T) {
    nf_drop d;
    d.off = p.drop_off[T.r], d.rlo = 0, d.bits = p.drop_bits;
    if (d.off != ~0ull) d.rlo = p.tile_lo[p.tile_off[T.r]];
    return d;
}
__device__ __forceinline__ bool nf_dropped(const nf_drop &d, uint64_t i) { // did this region's fetch drop read i at the cap?
    if (d.off == ~0ull) return false;
    const uint64_t b = d.off + (i - d.rlo);
    return (d.bits[b >> 6] >> (b & 63u)) & 1ull;
}

// a tile with at most this many reads in range cannot count past 255 anywhere: its counters are bytes (see rb_k_nf_tiles)
#define NF_U8_MAX_READS 255u
#define NF_D8_MAX_READS 127u // up to here a position's coverage difference is a signed BYTE too (nf_one_tile: the 40 KB build)
// which build of the tile kernel takes a tile: 1 = byte counters + byte differences, 2 = byte counters + 16-bit differences, 0 = 16-bit counters
__device__ __forceinline__ uint32_t nf_tile_kind(const rb_nf_params &p, uint64_t lo, uint64_t hi) {
    if (p.flags & 1u) return 0u;
    return hi - lo <= NF_D8_MAX_READS ? 1u : hi - lo <= NF_U8_MAX_READS ? 2u : 0u;
}

// everything a workgroup needs about a tile in ONE 64-byte record at a wave-uniform address (a scalar load), instead of the chain tile_off ->
// region (a binary search) -> region arrays -> drop_off -> tile_lo of the region's first tile that rb_k_nf_tiles walks at its start
struct __attribute__((aligned(64))) nf_tdesc {
    uint64_t st, out, lo, hi, drop_off, drop_rlo; // first position, first output position, reads [lo, hi), the region's dropped-read bitmap
    uint32_t n_pos;
    int32_t tid;
    uint32_t r, u8; // region; which build takes the tile (nf_tile_kind)
};
// a record loaded at a wave-uniform address, told to the compiler as the wave-uniform value it is
__device__ __forceinline__ uint64_t nf_first64(uint64_t v) { return rb_first64(v); }
__device__ __forceinline__ nf_tdesc nf_uniform(const nf_tdesc &d) {
    nf_tdesc o;
    o.st = nf_first64(d.st), o.out = nf_first64(d.out), o.lo = nf_first64(d.lo), o.hi = nf_first64(d.hi);
    o.drop_off = nf_first64(d.drop_off), o.drop_rlo = nf_first64(d.drop_rlo);
    o.n_pos = rb_first(d.n_pos), o.tid = (int32_t)rb_first((uint32_t)d.tid), o.r = rb_first(d.r), o.u8 = rb_first(d.u8);
    return o;
}
__device__ __forceinline__ nf_read nf_uniform(const nf_read &h) {
    nf_read o;
    o.pos = rb_first(h.pos), o.end = rb_first(h.end), o.tid = (int32_t)rb_first((uint32_t)h.tid), o.l_seq = rb_first(h.l_seq);
    o.op_off = nf_first64(h.op_off), o.n_ops = rb_first(h.n_ops), o.pad0 = 0, o.nib0 = nf_first64(h.nib0), o.pad1 = 0;
    return o;
}
__global__ __launch_bounds__(256) void rb_k_nf_tile_desc(rb_nf_params p) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= p.tile_off[p.n_regions]) { // (the launch over all tiles reads a record for every workgroup: behind the last tile one that says "nobody's")
        if (t < p.max_tiles) p.tdesc[t].u8 = 3u;
        return;
    }
    const nf_tile T = nf_tile_of(p, t);
    const nf_drop d = nf_drop_of(p, T);
    nf_tdesc D;
    D.st = T.st, D.out = T.out, D.lo = p.tile_lo[t], D.hi = p.tile_hi[t], D.drop_off = d.off, D.drop_rlo = d.rlo;
    D.n_pos = (uint32_t)(T.en - T.st), D.tid = T.tid, D.r = (uint32_t)T.r, D.u8 = nf_tile_kind(p, D.lo, D.hi);
    p.tdesc[t] = D;
    // (the builds of the rarer kinds walk lists: round 5 -- the 16-bit build used to be launched over all tiles to find its own)
    if (D.u8 == 2u) p.wide_list[1u + atomicAdd(&p.wide_list[0], 1u)] = (uint32_t)t;
    else if (D.u8 == 0u) p.wide_list[p.max_tiles - atomicAdd(&p.wide_list[p.max_tiles + 1u], 1u)] = (uint32_t)t;
}

// LDS place of tile position i: two dwords (A | C << 16, G | T << 16); 8 guard positions in front (a lane's group of 8 may start
// before the tile), and one dword skipped after every 8 positions so that the lanes of an atomic -- lane l adds to position
// 8 l + k -- are 17 dwords apart and fall on different banks
#define NF_CNT_DW (17 * ((NF_TILE + 16) / 8))
__device__ __forceinline__ uint32_t nf_slot(uint32_t i) { return 2u * (i + 8u) + ((i + 8u) >> 3); }
__device__ __forceinline__ uint32_t nf_swap_nibbles(uint32_t v) { // (two shifts and one v_bfi_b32: written as and / or the compiler makes five instructions of it)
    uint32_t o;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(o) : "s"(0xF0F0F0F0u), "v"(v << 4), "v"(v >> 4));
    return o;
}

// Three builds of the tile kernel (nf_tile_kind):
//   <U8T, D8>   tiles with at most 127 reads in range -- the usual ones with long reads: byte counters, the coverage differences as signed
//               bytes in the counters' own padding, staging buffers of 548 dwords: 40 KB of LDS, FOUR workgroups per CU (round 6); a
//               workgroup per tile, launched over all tiles
//   <U8T, !D8>  128 .. 255 reads in range: byte counters, the differences as 16-bit halves of a dword, staging buffers of 648 dwords: 51 KB,
//               three per CU (round 2's byte build: 4.43 -> 3.6 ms on config 5 against the 16-bit layout)
//   <!U8T>      more reads than that: 16-bit counters, 32-bit differences: 77 KB, two per CU; beyond 512 reads a lane per read
// The last two walk lists of their tiles (rb_k_nf_tile_desc writes them); each build leaves the others' tiles alone.
#define NF_CNT8_DW (10 * ((NF_TILE + 16) / 8))
// -DNF_DIAG (a diagnostics variant, tools/nf_phases.py): every 64th tile's waves add the shader-clock length of their phases to p.blk[8 ..]
#ifdef NF_DIAG
#define NF_STAMP(k) const uint64_t nf_t##k = __builtin_amdgcn_s_memtime()
#define NF_DECL(k) uint64_t nf_t##k = 0
#define NF_SET(k) nf_t##k = __builtin_amdgcn_s_memtime()
#define NF_PHASE(slot, a, b) do { if ((t & 63u) == 0u && (tix & 63u) == 0u) atomicAdd(reinterpret_cast<unsigned long long *>(p.blk) + 8 + (slot), (unsigned long long)(nf_t##b - nf_t##a)); } while (0)
#else
#define NF_STAMP(k) do { } while (0)
#define NF_DECL(k) do { } while (0)
#define NF_SET(k) do { } while (0)
#define NF_PHASE(slot, a, b) do { } while (0)
#endif
#ifndef NF_U8_WPE
#define NF_U8_WPE 8 // waves per SIMD the byte-counter build is compiled for: four workgroups of eight waves a CU
#endif
#ifndef NF_U8_SLACK_DW
#define NF_U8_SLACK_DW 28 // byte-counter tiles: dwords of staging beyond the tile's own 512 (224 inserted bases + alignment; more: the unstaged route)
#endif
#ifndef NF_PIPE
#define NF_PIPE 1 // 0: the wave's reads strictly one after the other (rounds 1 - 5)
#endif
#ifndef NF_STOP
#define NF_STOP 0 // diagnostics (timing only, wrong counts): 1 = no output written, 2 = no LDS atomics (bases staged and decoded, nothing added), 3 = no read touched, 5 = chunks scanned and landed but not counted (9: no bases fetched either, 10: fetched and dropped), 6 = every op's set-up but none of its groups, 7 = only the launch, 8 = no read touched and no depth scan
#endif
// (at most) 64 ops of one read against a tile, between their scan and the counting (nf_one_tile: chunk_scan / chunk_land / chunk_count)
template <int IT>
struct nf_chunk { // (no implicit padding: the compiler copies a struct's padding through memory)
    uint64_t m;      // the lanes whose op lays bases over the tile
    int64_t wd_lo;   // the stretch of packed bases the ops of m cover: its first dword ...
    int32_t ia, ib;  // the lane's op: the tile indices [ia, ib) of its bases ...
    uint32_t qa;     // ... and where in the read the base at ia is (bam_pileup1_t::qpos)
    uint32_t staged; // 1: the stretch fits the wave's buffer: v holds it (loads in flight until chunk_land)
    int32_t n_dw;    // ... and how many dwords (meaningful while it fits: a longer stretch is capped)
    int32_t s;       // 4-bit index of the read's first base relative to the stretch's first dword (mod 2^32)
    uint32_t pad[2];
    uint4 v[IT];
};
template <bool U8T, bool D8>
__device__ __forceinline__ void nf_one_tile(const rb_nf_params &p, const uint64_t t) {
    // the thread's index through an opaque copy: what is derived from it (a dozen lane-times-constant addresses) is then computed per
    // tile -- a few instructions -- instead of being kept in registers across the list-walking builds' loop over tiles, where it cost spills
    uint32_t tix;
    asm volatile("v_mov_b32 %0, %1" : "=v"(tix) : "v"(threadIdx.x));
    NF_STAMP(0);
    NF_DECL(1);
    static_assert(U8T || !D8, "byte differences live in the byte counters' padding");
    constexpr uint32_t STG = U8T ? (uint32_t)(NF_TILE / 8 + (D8 ? NF_U8_SLACK_DW : 128)) : (uint32_t)NF_STAGE_DW; // dwords of a read staged per tile (a longer stretch is read from memory group by group)
    static_assert((STG + 8) % 4 == 0, "a wave's staging buffer starts 16-byte aligned");
    constexpr int STG_IT = (int)((STG + 255u) / 256u);
    constexpr uint32_t CNT_DW = ((U8T ? NF_CNT8_DW : NF_CNT_DW) + 3) / 4 * 4; // (zeroed 16 bytes at a time)
    __shared__ __attribute__((aligned(16))) uint32_t cnt[CNT_DW]; // per position: one dword of four byte counters / A | C << 16, G | T << 16
    __shared__ uint32_t lut[16];        // what a base code adds to its word: 1 4 = A G: 1; 2 8 = C T: 1 << 16; everything else 0 (nucfreq.rs:83-90)
    __shared__ uint32_t lut8[16];       // U8 tiles: 1 2 4 8 = A C G T: 1 << 0, 8, 16, 24
#ifndef NF_LUT_SINGLE
    __shared__ unsigned long long lut16[U8T ? 256 : 1]; // two bases at once: low nibble -> low dword, high nibble -> high dword
    if (U8T && tix < 256) {
        auto one = [](uint32_t n) -> unsigned long long { return n == 1 ? 1ull : n == 2 ? 0x100ull : n == 4 ? 0x10000ull : n == 8 ? 0x1000000ull : 0ull; };
        lut16[tix] = one(tix & 15u) | (one(tix >> 4) << 32);
    }
#endif
    // +1 where a read starts covering, -1 where it stops; then the depth.  U8T (round 6): a signed BYTE per position, and the bytes live
    // in the counters' own padding -- a group of 8 positions is 10 dwords, 8 of counters and 2 that only keep the lanes of an atomic on
    // different banks: 8 bytes, one per position.  They are added as whole 32-bit integers (a borrow of a byte travels into the next one
    // and is taken back when the dword is read: exact while |difference| <= 127, i.e. at most 127 reads in range, NF_D8_MAX_READS;
    // tiles of 128 .. 255 reads keep 16-bit halves in an array of their own: the !D8 build).
    // With that, the eight staging buffers a little shorter and nothing else, a tile is 40 KB of LDS: FOUR workgroups a CU instead of
    // three (51 KB before), eight waves per SIMD -- the kernel's vector ALU idled a quarter of the time for want of a fourth tile in
    // another phase.
    constexpr uint32_t NF_DIFF_DW = D8 ? 4u : ((uint32_t)(U8T ? (NF_TILE + 8) / 2 + 2 : NF_TILE + 8) + 3u) / 4u * 4u; // (!D8, byte counters: two positions a dword, 16-bit halves added as whole integers)
    __shared__ __attribute__((aligned(16))) int32_t diff[NF_DIFF_DW];
    __shared__ uint8_t covb[U8T ? NF_THREADS : 4]; // U8T: which of a thread's 8 positions are covered
    __shared__ int32_t wsum[NF_WAVES];
    __shared__ uint32_t blk_max, blk_cov;
    __shared__ __attribute__((aligned(16))) uint32_t stage_all[NF_WAVES][STG + 8]; // per wave: the bases one read lays over the tile (4 zero dwords in front)
    // (round 4: the tile's facts come as ONE 64-byte record at a wave-uniform address -- rb_k_nf_tile_desc -- instead of a binary search
    //  for the region followed by dependent loads of its arrays, of the region's first tile and of its dropped-read bitmap's place;
    //  round 6: asked for first thing and looked at behind the zeroing -- the record's trip to memory runs under it; rb_k_nf_tile_desc
    //  marks the records behind the last tile, so nothing else has to be fetched to know that a workgroup has no tile)
    const nf_tdesc Draw = p.tdesc[t];
    // ---- zero the tile (nothing of it depends on the record) ----
    for (uint32_t k = tix; k < CNT_DW / 4; k += NF_THREADS) reinterpret_cast<uint4 *>(cnt)[k] = make_uint4(0, 0, 0, 0); // (16 bytes a store)
    if ((tix & 63u) < 4u) stage_all[tix >> 6][tix & 63u] = 0;
    if constexpr (!D8)
        for (uint32_t k = tix; k < NF_DIFF_DW / 4; k += NF_THREADS) reinterpret_cast<uint4 *>(diff)[k] = make_uint4(0, 0, 0, 0);
    if (tix < 16) {
        const uint32_t n = tix;
        lut[n] = (n == 1 || n == 4) ? 1u : (n == 2 || n == 8) ? 0x10000u : 0u;
        lut8[n] = n == 1 ? 1u : n == 2 ? 0x100u : n == 4 ? 0x10000u : n == 8 ? 0x1000000u : 0u;
    }
    if (tix == 0) blk_max = 0, blk_cov = 0;
    const nf_tdesc D = nf_uniform(Draw);
    if (D.u8 != (D8 ? 1u : U8T ? 2u : 0u)) return; // another build's, or nobody's
#if NF_STOP == 7
    if (D.n_pos != 0x7FFFFFFFu) return; // (timing only: what it costs to launch the tiles' workgroups)
#endif
    nf_tile T;
    T.r = D.r, T.st = D.st, T.en = D.st + D.n_pos, T.out = D.out, T.tid = D.tid;
    nf_drop drop;
    drop.off = D.drop_off, drop.rlo = D.drop_rlo, drop.bits = p.drop_bits;
    const uint32_t n_pos = D.n_pos;
#if NF_PIPE
    // the wave's reads (lo + wave + NF_WAVES j): lane j asks for read j's record now, before the tile is zeroed -- the first of the
    // dependent trips (tile -> records -> ops -> bases) runs under the zeroing and the barrier.  (Not for the crowded tiles: lane-per-read.)
    const bool nf_by_wave = U8T || D.hi - D.lo <= 8u * 64u;
    const uint32_t nw = nf_by_wave && D.hi > D.lo + (tix >> 6) ? (uint32_t)((D.hi - D.lo - (tix >> 6) + NF_WAVES - 1) / NF_WAVES) : 0u;
    nf_read hv;
    hv.pos = 0, hv.end = 0, hv.tid = -1, hv.l_seq = 0, hv.op_off = 0, hv.n_ops = 0, hv.pad0 = 0, hv.nib0 = 0, hv.pad1 = 0;
    const uint64_t hv_i = D.lo + (tix >> 6) + (uint64_t)NF_WAVES * (tix & 63u);
    if ((tix & 63u) < nw) hv = p.hd[hv_i];
#endif
    auto diff_add = [&](uint32_t i, int32_t delta) {
        if constexpr (D8) { // position i = byte (i + 8) & 7 of the two spare dwords of its group (i + 8) >> 3
            const uint32_t q = i + 8u;
            atomicAdd(reinterpret_cast<int32_t *>(&cnt[10u * (q >> 3) + 8u + ((q >> 2) & 1u)]), (int32_t)((uint32_t)delta << (8u * (q & 3u))));
        } else if constexpr (U8T) {
            atomicAdd(&diff[i >> 1], (int32_t)((uint32_t)delta << (16u * (i & 1u))));
        } else {
            atomicAdd(&diff[i], delta);
        }
    };
    // (the barrier that makes the zeroing everybody's sits in front of the first LDS atomic: behind the first requests of the wave's reads)
    const uint32_t wib = rb_first(tix >> 6);
    const int lane = (int)(tix & 63u);
    const uint64_t lo = D.lo, hi = D.hi;
    // U8 tiles (at most 255 reads in range, the usual case with long reads): a position is ONE dword of four byte counters, 10 dwords
    // per 8 positions (lane l of an add is 10 l dwords on: all of a half-wave's 8-byte accesses on different banks), and one
    // ds_add_u64 covers two positions -- half the atomics of the 16-bit layout and no choice of word per base
    const uint32_t *__restrict__ sw32 = reinterpret_cast<const uint32_t *>(p.seq);
    uint32_t *stage = stage_all[wib] + 4;
    // ---- one read against the tile, in three pieces so that the pieces of consecutive reads can overlap (the loop further down):
    //      chunk_scan  (at most) 64 ops of the read: where each starts on the reference / in the read, which of them lay bases over the
    //                  tile, and the loads of that stretch of packed bases -- 16 bytes per lane, all in flight, nothing waited for
    //      chunk_land  the stretch arrives and is parked in the wave's LDS buffer in base order
    //      chunk_count every match-type op of the chunk: each lane takes 8 consecutive positions, table reads, LDS atomics
    auto chunk_scan = [&](const nf_read &h, const uint32_t w, uint32_t &R, uint32_t &Q, const uint64_t i) -> nf_chunk<STG_IT> {
        nf_chunk<STG_IT> k;
        const int64_t pos = h.pos;
        const int64_t rel_st = (int64_t)T.st - pos, rel_en = (int64_t)T.en - pos; // the tile in read-relative reference offsets
        const uint32_t c = rb_opc(w), len = rb_len(w);
        const uint32_t rl = rb_in(RB_REF_MASK, c) ? len : 0u, ql = rb_in(RB_QRY_MASK, c) ? len : 0u;
        const uint32_t ir = rb_wave_scan_incl(rl), iq = rb_wave_scan_incl(ql);
        const uint32_t r0 = R + ir - rl, q0 = Q + iq - ql; // where my op starts on the reference / in the read
        const uint32_t Q0 = Q;
        R += rb_readlane<uint32_t>(ir, 63);
        Q += rb_readlane<uint32_t>(iq, 63);
        k.m = 0, k.wd_lo = 0, k.n_dw = 0, k.staged = 0, k.ia = 0, k.ib = 0, k.qa = 0, k.s = 0, k.pad[0] = k.pad[1] = 0;
        // A read of the tile's fetch has pos < T.en and pos + span > T.st with span < 2^31 (rb_k_nf_read_spans), so rel_st lies in
        // (-NF_TILE, 2^31), every op starts and ends below 2^31, and 32 bits hold all of it (rel_en only matters below a span; the wave-
        // uniform tests are 32-bit on purpose: a 64-bit ordered compare has no scalar instruction and lands on the vector ALU).  Behind
        // the last chunk -- null ops, the record of no read -- nothing hits whatever the numbers are.
        const int32_t rs = (int32_t)rel_st, re = (uint64_t)rel_en < 0x7FFFFFFFull ? (int32_t)rel_en : 0x7FFFFFFF;
        if ((int32_t)R <= rs) return k; // these 64 ops end before the tile
        // every lane its op's share of the tile, once and for all lanes at once (round 6; it was scalar arithmetic per op): tile indices
        // [ia, ib) and bam_pileup1_t::qpos of the base at ia
        const int32_t r1 = (int32_t)(r0 + len);
        const bool hit = rb_in(RB_MATCH_MASK, c) && (int32_t)r0 >= 0 && r1 >= (int32_t)r0 && (int32_t)r0 < re && r1 > rs;
        k.m = __ballot(hit);
        if (!k.m) return k;
        const int32_t a = (int32_t)r0 > rs ? (int32_t)r0 : rs, b = r1 < re ? r1 : re;
        const int64_t lseq = (int64_t)h.l_seq;
        int32_t ia = a - rs, ib = b - rs;
        const uint32_t qa = q0 + (uint32_t)(a - (int32_t)r0);
        if ((uint64_t)Q > (uint64_t)h.l_seq || Q < Q0) { // (only a read whose ops ask for more bases than it has -- record().seq()[qpos] out of bounds: the reference panics)
            const int64_t qa64 = (int64_t)q0 + (int64_t)(a - (int32_t)r0);
            bool seq_short = false;
            if (hit && qa64 + (int64_t)(ib - ia) > lseq) {
                seq_short = true;
                ib = lseq > qa64 ? ia + (int32_t)(lseq - qa64) : ia;
            }
            if (__ballot(seq_short) != 0 && lane == 0) p.read_status[i] = RB_RD_SEQ_SHORT;
        }
        k.ia = ia, k.ib = ib, k.qa = qa; // (an op that counts anything has qa < l_seq: 32 bits)
        // the bases these ops lay over the tile are one contiguous stretch of the read: fetch all of it now, 16 bytes per lane and
        // every load in flight at once -- the per-op loops then never wait for HBM.  (A stretch longer than the buffer -- a long
        // insertion inside the tile -- is read from memory group by group instead.)
        {
            const int jf = __builtin_ctzll(k.m), jl = 63 - __builtin_clzll(k.m);
            const int32_t fr0 = rb_readlane<int32_t>((int32_t)r0, jf), lr1 = rb_readlane<int32_t>(r1, jl);
            const uint32_t fq0 = rb_readlane<uint32_t>(q0, jf), lq0 = rb_readlane<uint32_t>(q0, jl), lr0 = rb_readlane<uint32_t>(r0, jl);
            const uint64_t nib0 = h.nib0;
            const uint32_t q_lo = fq0 + (uint32_t)((fr0 > rs ? fr0 : rs) - fr0);
            uint32_t q_hi = lq0 + ((uint32_t)(lr1 < re ? lr1 : re) - lr0);
            q_hi = q_hi < h.l_seq ? q_hi : h.l_seq;
            const uint64_t t_lo = nib0 + q_lo;
            k.wd_lo = ((uint32_t)(t_lo >> 32) != 0u || (uint32_t)t_lo >= 7u) ? (int64_t)((t_lo - 7u) >> 3) : 0;
            k.s = (int32_t)(uint32_t)(nib0 - 8ull * (uint64_t)k.wd_lo); // (4-bit index of the read's first base inside the stretch; small once an op's qa is added)
            const uint64_t n_dw = ((nib0 + q_hi + 7u) >> 3) + 2u - (uint64_t)k.wd_lo; // (+ the second dword of the last group)
            const bool fits = (uint32_t)(n_dw >> 32) == 0u && (uint32_t)n_dw <= STG;
            k.staged = fits && q_hi > q_lo ? 1u : 0u;
#if NF_STOP == 9
            k.staged = 0u; // (timing only: no bases fetched)
#endif
            k.n_dw = fits ? (int32_t)(uint32_t)n_dw : (int32_t)STG + 1;
            if (k.staged) {
#pragma unroll
                for (int r = 0; r < STG_IT; r++) {
                    int32_t idx = 4 * lane + 256 * r;
                    idx = idx < k.n_dw ? idx : (k.n_dw - 1 > 0 ? (k.n_dw - 1) & ~3 : 0); // (past the stretch: re-read its last piece)
                    k.v[r] = rb_load4_unaligned(sw32 + k.wd_lo + idx);
                }
            }
        }
        return k;
    };
    auto chunk_land = [&](const nf_chunk<STG_IT> &k) {
        if (!k.staged) return;
#if NF_STOP == 10
        if (k.v[0].x != 0x12345678u || k.v[STG_IT - 1].w != 0x9ABCDEF0u) return; // (timing only: the bases fetched and dropped)
#endif
#pragma unroll
        for (int r = 0; r < STG_IT; r++)
            if (4 * lane + 256 * r < k.n_dw) // (kept in base order: BAM packs the first base of a byte into its high half)
                *reinterpret_cast<uint4 *>(stage + 4 * lane + 256 * r) =
                    make_uint4(nf_swap_nibbles(k.v[r].x), nf_swap_nibbles(k.v[r].y), nf_swap_nibbles(k.v[r].z), nf_swap_nibbles(k.v[r].w));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto chunk_count = [&](auto U8, const nf_read &h, const nf_chunk<STG_IT> &k) {
        const bool staged = k.staged != 0u;
        uint64_t m = k.m;
#if NF_STOP == 5 || NF_STOP == 9 || NF_STOP == 10
        if (m != 0x123456789ull) return; // (timing only: chunks scanned and landed, nothing counted)
#endif
        // lane l takes the 8 positions [P, P + 8) of an 8-aligned group (P = tile index + 8 = P0 + 8 l, then 512 on per turn); their
        // bases are 8 consecutive 4-bit codes of the read: two dwords (nibbles already in base order), funnel-shifted to the group's
        // first base; x = the 8 codes with the bases outside [ia, ib) made code 0 (they add nothing)
        auto add_group = [&](const uint32_t x, uint32_t *const g32) {
#if NF_STOP == 2
            if (x == 0x12345678u) cnt[x & 1023u] = x; // (keeps x alive)
            return;
#endif
            uint32_t inc[8]; // (the table reads first, all eight in flight, then the atomics)
            if constexpr (decltype(U8)::value) {
                unsigned long long *g = reinterpret_cast<unsigned long long *>(g32);
#ifndef NF_LUT_SINGLE // (round 3: one 8-byte table read per TWO bases -- 256 entries, 2 KB -- instead of two 4-byte reads: 3.74 -> 3.57 ms per call, same box)
                unsigned long long inc2[4];
#pragma unroll
                for (int q = 0; q < 4; q++) inc2[q] = lut16[(x >> (8 * q)) & 255u];
#pragma unroll
                for (int q = 0; q < 4; q++) atomicAdd(g + q, inc2[q]);
#else
#pragma unroll
                for (int q = 0; q < 8; q++) inc[q] = lut8[(x >> (4 * q)) & 15u];
#pragma unroll
                for (int q = 0; q < 4; q++) atomicAdd(g + q, (unsigned long long)inc[2 * q] | ((unsigned long long)inc[2 * q + 1] << 32));
#endif
            } else {
#pragma unroll
                for (int q = 0; q < 8; q++) inc[q] = lut[(x >> (4 * q)) & 15u];
#pragma unroll
                for (int q = 0; q < 8; q++) atomicAdd(g32 + 2 * q + __builtin_amdgcn_ubfe(0x110u, (x >> (4 * q)) & 15u, 1u), inc[q]);
            }
        };
        constexpr uint32_t CNT_PER8 = decltype(U8)::value ? 10u : 17u; // dwords of counters per 8 positions (see nf_slot / NF_CNT8_DW)
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            const int32_t ia = rb_readlane<int32_t>(k.ia, j), ib = rb_readlane<int32_t>(k.ib, j); // tile indices [ia, ib) of this op's bases
            const uint32_t qa = rb_readlane<uint32_t>(k.qa, j);                                   // bam_pileup1_t::qpos of the base at ia
            const int32_t sP0 = (ia + 8) & ~7;
#if NF_STOP == 6
            if (ia != 0x7FFFFFF0) { // (timing only: an op's set-up, none of its groups)
                if (sP0 == 0x7FFFFFF1 + (int32_t)(qa & 1u) + ib) cnt[0] = 1u;
                continue;
            }
#endif
            // what moves from turn to turn is kept as running values (no multiply, no shift inside the loop): lo4 / hi4 = four times the
            // number of positions the group starts before ia / ends behind ib (the masks' shift counts); the lane is in while hi4 < 32
            int32_t lo4 = 4 * (ia + 8 - sP0) - 32 * lane, hi4 = 4 * (sP0 - ib) + 32 * lane;
            uint32_t *g = &cnt[CNT_PER8 * (((uint32_t)sP0 >> 3) + (uint32_t)lane)];
            if (staged) {
                const int32_t an0 = (int32_t)((uint32_t)k.s + qa) - (ia + 8) + sP0; // 4-bit index of position sP0 inside the staged stretch
                const uint32_t *sp = stage + (an0 >> 3) + lane; // ((an0 >> 3) = -1: only masked bases, in front of the buffer -- the zero guard)
                const uint32_t shift4 = (uint32_t)(an0 & 7) * 4u;
                for (; hi4 < 32; lo4 -= 2048, hi4 += 2048, sp += 64, g += CNT_PER8 * 64u) {
                    const uint32_t raw = __builtin_amdgcn_alignbit(sp[1], sp[0], shift4);
                    add_group(raw & (0xFFFFFFFFu << (uint32_t)(lo4 > 0 ? lo4 : 0)) & (0xFFFFFFFFu >> (uint32_t)(hi4 > 0 ? hi4 : 0)), g);
                }
            } else {
                const int64_t An0 = (int64_t)h.nib0 + (int64_t)qa + (int64_t)(sP0 + 8 * lane - 8 - ia); // (>= -7: only masked bases can lie before the buffer)
                for (int64_t An = An0; hi4 < 32; lo4 -= 2048, hi4 += 2048, An += 512, g += CNT_PER8 * 64u) {
                    const int64_t wd = An >> 3;
                    const uint32_t raw = __builtin_amdgcn_alignbit(nf_swap_nibbles(sw32[wd + 1]), wd >= 0 ? nf_swap_nibbles(sw32[wd]) : 0u, (uint32_t)(An & 7) * 4u);
                    add_group(raw & (0xFFFFFFFFu << (uint32_t)(lo4 > 0 ? lo4 : 0)) & (0xFFFFFFFFu >> (uint32_t)(hi4 > 0 ? hi4 : 0)), g);
                }
            }
        }
    };
    // does the tile's fetch hold this read?  (hts_itr_next: pos < en && endpos > st; end = 0: not in the pileup; dropped at the depth cap)
    auto in_tile = [&](const nf_read &h, uint64_t i) -> bool {
        if (h.tid != T.tid || (uint64_t)h.pos >= T.en || (uint64_t)h.end <= T.st) return false;
        return !nf_dropped(drop, i);
    };
    auto cover = [&](const nf_read &h) { // the read's stretch of the tile in the difference array
        const uint64_t c0 = (uint64_t)h.pos > T.st ? (uint64_t)h.pos : T.st, c1 = (uint64_t)h.end < T.en ? (uint64_t)h.end : T.en;
        if (lane == 0) {
            diff_add((uint32_t)(c0 - T.st), 1);
            diff_add((uint32_t)(c1 - T.st), -1);
        }
    };
    // one read from start to end, the whole wave on it, chunk after chunk (w_first: its first 64 ops, already in registers): reads
    // of more than 64 ops, and the reads the crowded tiles leave to the wave
    auto read_by_wave = [&](auto U8, const nf_read &h, uint32_t w_first, uint64_t i) {
#if NF_STOP == 3
        return;
#endif
        if (!in_tile(h, i)) return;
        cover(h);
        const uint64_t o0 = h.op_off, o1 = h.op_off + h.n_ops;
        const int64_t rel_en = (int64_t)T.en - (int64_t)h.pos;
        uint32_t R = 0, Q = 0;
        for (uint64_t o = o0; o < o1; o += 64) {
            const uint32_t w = o == o0 ? w_first : ((o + (uint64_t)lane < o1) ? p.ops[o + (uint64_t)lane] : RB_NULL_OP);
            const nf_chunk<STG_IT> k = chunk_scan(h, w, R, Q, i);
            chunk_land(k);
            chunk_count(U8, h, k);
            if ((int64_t)R >= rel_en) break;
        }
    };
    if (!U8T && hi - lo > 8u * 64u) {
        __syncthreads(); // (the zeroing)
        NF_SET(1);
        // ---- a tile crowded with reads (short reads): 64 reads per wave and turn, one lane per read with at most NF_LANE_OPS ops
        //      -- it walks its ops and drops its bases one by one; the wave scans would idle on 150-base reads -- and the few
        //      reads with longer cigars one after the other with the whole wave ----
        for (uint64_t g = lo + 64u * wib; g < hi; g += 64u * NF_WAVES) {
            const uint64_t i = g + (uint64_t)lane;
            nf_read h;
            h.end = 0, h.n_ops = 0, h.op_off = 0, h.pos = 0, h.tid = -1, h.l_seq = 0, h.nib0 = 0;
            if (i < hi) h = p.hd[i];
            const bool overl = h.tid == T.tid && (uint64_t)h.pos < T.en && (uint64_t)h.end > T.st && !nf_dropped(drop, i);
            const bool simple = overl && h.n_ops <= NF_LANE_OPS;
            uint64_t cm = __ballot(overl && !simple);
            if (simple) {
                const int64_t pos = h.pos;
                const uint64_t c0 = (uint64_t)pos > T.st ? (uint64_t)pos : T.st, c1 = (uint64_t)h.end < T.en ? (uint64_t)h.end : T.en;
                diff_add((uint32_t)(c0 - T.st), 1);
                diff_add((uint32_t)(c1 - T.st), -1);
                const int64_t rel_st = (int64_t)T.st - pos, rel_en = (int64_t)T.en - pos, idx0 = pos - (int64_t)T.st;
                int64_t R = 0, Q = 0;
                bool seq_short = false;
                for (uint32_t j = 0; j < h.n_ops; j++) {
                    const uint32_t w = p.ops[h.op_off + j], c = rb_opc(w);
                    const int64_t len = rb_len(w);
                    if (rb_in(RB_MATCH_MASK, c)) {
                        const int64_t a = R > rel_st ? R : rel_st, e = R + len < rel_en ? R + len : rel_en;
                        int64_t e2 = e;
                        if (Q + (e - R) > (int64_t)h.l_seq) { // record().seq()[qpos] would be out of bounds: the reference panics
                            seq_short = true;
                            e2 = R + ((int64_t)h.l_seq - Q);
                        }
                        // eight bases per load: the aligned dword that holds the base, nibbles swapped into base order
                        uint64_t nb = h.nib0 + (uint64_t)(Q + (a - R)); // 4-bit index of the base at x = a (bam_pileup1_t::qpos + the read's start)
                        uint32_t word = a < e2 ? nf_swap_nibbles(sw32[nb >> 3]) >> ((uint32_t)(nb & 7u) * 4u) : 0u;
                        for (int64_t x = a; x < e2; x++) {
                            const uint32_t nib = word & 15u;
                            atomicAdd(&cnt[nf_slot((uint32_t)(idx0 + x)) + __builtin_amdgcn_ubfe(0x110u, nib, 1u)], lut[nib]);
                            nb++;
                            word = (nb & 7u) ? word >> 4 : (x + 1 < e2 ? nf_swap_nibbles(sw32[nb >> 3]) : 0u);
                        }
                    }
                    if (rb_in(RB_REF_MASK, c)) R += len;
                    if (rb_in(RB_QRY_MASK, c)) Q += len;
                }
                if (seq_short) p.read_status[i] = RB_RD_SEQ_SHORT;
            }
            while (cm) {
                const int l = __builtin_ctzll(cm);
                cm &= cm - 1;
                nf_read hh;
                hh.pos = rb_readlane<uint32_t>(h.pos, l), hh.end = rb_readlane<uint32_t>(h.end, l), hh.tid = rb_readlane<int>(h.tid, l);
                hh.l_seq = rb_readlane<uint32_t>(h.l_seq, l), hh.n_ops = rb_readlane<uint32_t>(h.n_ops, l);
                hh.op_off = rb_readlane<uint64_t>(h.op_off, l), hh.nib0 = rb_readlane<uint64_t>(h.nib0, l);
                const uint32_t wf = (uint32_t)lane < hh.n_ops ? p.ops[hh.op_off + (uint32_t)lane] : RB_NULL_OP;
                read_by_wave(std::false_type{}, hh, wf, g + (uint64_t)l);
            }
        }
    } else {
#if NF_PIPE
        // ---- the wave's reads as a stream of CHUNKS (a read's ops, 64 at a time), three chunks under way at any time (round 6).
        //      The wave's reads are lo + wib + NF_WAVES j, j < nw <= 64: lane j fetched read j's record before the tile was zeroed
        //      (hv), so a record is a row of v_readlane away; the lanes have each entered their own read into the coverage
        //      differences, and `alive` says which reads the tile's fetch holds.  In a turn: the bases of chunk c (requested a turn
        //      ago) are parked in LDS; chunk c + 1 (ops requested a turn ago) is scanned and ITS bases requested; the ops of chunk
        //      c + 2 are requested; then chunk c is counted out of LDS.  Every trip to memory has a chunk's worth of work in front
        //      of it -- before, a read's bases were waited for where they were requested: five reads a wave and tile, two
        //      microseconds each. ----
        bool mine = (uint32_t)lane < nw && in_tile(hv, hv_i);
#if NF_STOP == 3 || NF_STOP == 8
        mine = false;
#endif
        const uint64_t alive = __ballot(mine);
        struct cursor { // a chunk to come: which read, where in its ops, the reference / read bases in front of it
            nf_read h;
            uint64_t i;
            uint32_t j, o, R, Q;
            uint32_t valid, pad;
        };
        auto read_at = [&](uint64_t from_mask) -> cursor { // the first chunk of the first live read among the lanes of from_mask
            cursor c;
            const uint64_t mk = alive & from_mask;
            c.valid = mk != 0 ? 1u : 0u, c.pad = 0;
            c.j = c.valid ? (uint32_t)__builtin_ctzll(mk) : 0u;
            c.o = 0, c.R = 0, c.Q = 0;
            c.i = lo + wib + (uint64_t)NF_WAVES * c.j;
            c.h.pos = rb_readlane<uint32_t>(hv.pos, (int)c.j), c.h.l_seq = rb_readlane<uint32_t>(hv.l_seq, (int)c.j);
            c.h.n_ops = c.valid ? rb_readlane<uint32_t>(hv.n_ops, (int)c.j) : 0u;
            c.h.op_off = rb_readlane<uint64_t>(hv.op_off, (int)c.j), c.h.nib0 = rb_readlane<uint64_t>(hv.nib0, (int)c.j);
            c.h.end = 0, c.h.tid = T.tid, c.h.pad0 = 0, c.h.pad1 = 0; // (what only the fetch's test needed)
            return c;
        };
        auto after = [&](const cursor &c) -> cursor { // the chunk behind c (c scanned: c.R is the reference behind its ops)
            if (c.valid && c.o + 64u < c.h.n_ops && (int64_t)c.R < (int64_t)T.en - (int64_t)c.h.pos) {
                cursor d = c;
                d.o = c.o + 64u;
                return d;
            }
            return read_at(c.valid && c.j < 63u ? ~0ull << (c.j + 1u) : 0ull);
        };
        auto ops_at = [&](const cursor &c) -> uint32_t {
            return c.o + (uint32_t)lane < c.h.n_ops ? p.ops[c.h.op_off + c.o + (uint32_t)lane] : RB_NULL_OP;
        };
        auto empty = []() -> nf_chunk<STG_IT> {
            nf_chunk<STG_IT> k;
            k.m = 0, k.wd_lo = 0, k.ia = 0, k.ib = 0, k.qa = 0, k.staged = 0, k.n_dw = 0, k.s = 0, k.pad[0] = k.pad[1] = 0;
            return k;
        };
        cursor c_cur = read_at(~0ull);
        const uint32_t w_cur = ops_at(c_cur); // (the first read's ops: on their way while the workgroup meets at the barrier)
        __syncthreads();                      // (the zeroing is everybody's: LDS atomics from here on)
        NF_SET(1);
        if (mine) { // (every lane its own read into the coverage differences: the two atomics of up to 64 reads in one go)
            const uint64_t c0 = (uint64_t)hv.pos > T.st ? (uint64_t)hv.pos : T.st, c1 = (uint64_t)hv.end < T.en ? (uint64_t)hv.end : T.en;
            diff_add((uint32_t)(c0 - T.st), 1);
            diff_add((uint32_t)(c1 - T.st), -1);
        }
        nf_chunk<STG_IT> k_cur = empty();
        if (c_cur.valid) k_cur = chunk_scan(c_cur.h, w_cur, c_cur.R, c_cur.Q, c_cur.i);
        cursor c_nxt = after(c_cur);
        uint32_t w_nxt = ops_at(c_nxt);
        NF_STAMP(2);
        NF_PHASE(1, 1, 2);
        while (c_cur.valid) {
            chunk_land(k_cur);
            // (scanned whether there is a chunk or not -- behind the last one the ops are null ops and nothing hits --: every path through
            //  a turn then consumes the ops it asked for, and the compiler's wait in front of the next request is not for everything)
            nf_chunk<STG_IT> k_nxt = chunk_scan(c_nxt.h, w_nxt, c_nxt.R, c_nxt.Q, c_nxt.i);
            const cursor c_nn = after(c_nxt);
            const uint32_t w_nn = ops_at(c_nn);
            chunk_count(std::integral_constant<bool, U8T>{}, c_cur.h, k_cur);
            c_cur = c_nxt, k_cur = k_nxt, c_nxt = c_nn, w_nxt = w_nn;
        }
        NF_STAMP(3);
        NF_PHASE(2, 2, 3);
#else
        __syncthreads(); // (the zeroing)
        NF_SET(1);
        // ---- the wave's reads, one after the other.  Two loads run ahead of the work: the record of the read after next, and the
        //      first 64 ops of the next read (whose record arrived one turn earlier) -- a read then starts with its ops in registers
        //      instead of waiting for three dependent trips to memory ----
        nf_read h_cur, h_nxt;
        h_cur.end = 0, h_nxt.end = 0, h_cur.n_ops = 0, h_nxt.n_ops = 0, h_cur.op_off = 0, h_nxt.op_off = 0;
        uint32_t w_cur = RB_NULL_OP;
        if (lo + wib < hi) h_cur = p.hd[lo + wib];
        if (lo + wib + NF_WAVES < hi) h_nxt = p.hd[lo + wib + NF_WAVES];
        if ((uint32_t)lane < h_cur.n_ops) w_cur = p.ops[h_cur.op_off + (uint32_t)lane];
        for (uint64_t i = lo + wib; i < hi; i += NF_WAVES) {
            nf_read h_nn;
            h_nn.end = 0, h_nn.n_ops = 0, h_nn.op_off = 0;
            if (i + 2 * NF_WAVES < hi) h_nn = p.hd[i + 2 * NF_WAVES];
            uint32_t w_nxt = RB_NULL_OP;
            if ((uint32_t)lane < h_nxt.n_ops) w_nxt = p.ops[h_nxt.op_off + (uint32_t)lane];
            const nf_read h = h_cur;
            const uint32_t w_first = w_cur;
            h_cur = h_nxt, h_nxt = h_nn, w_cur = w_nxt;
            read_by_wave(std::integral_constant<bool, U8T>{}, h, w_first, i);
        }
#endif
    }
    NF_STAMP(3b);
    __syncthreads();
    NF_STAMP(4);
    NF_PHASE(0, 0, 1);
    NF_PHASE(3, 3b, 4);
    // depth = prefix sum of the difference array: NF_PER_THREAD positions per thread
#if NF_STOP != 8
    {
        const uint32_t b0 = tix * NF_PER_THREAD;
        int32_t d[NF_PER_THREAD], s = 0;
        if constexpr (D8) {
            // the thread's 8 positions are one group of the counters (tile position 8 t = slot 8 t + 8): its differences are the 8 bytes of
            // the group's two spare dwords; byte by byte, each one's borrow taken back out of what is left
            static_assert(NF_PER_THREAD == 8, "a thread scans one group of 8 positions");
            const uint2 sp2 = *reinterpret_cast<const uint2 *>(&cnt[10u * (tix + 1u) + 8u]);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                int32_t v = (int32_t)(h ? sp2.y : sp2.x);
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int32_t dk = (int32_t)(int8_t)(uint8_t)((uint32_t)v & 255u);
                    v = (v - dk) >> 8;
                    s += dk;
                    d[4 * h + k] = s;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < NF_PER_THREAD; k++) {
                int32_t dk;
                if constexpr (U8T) { // (b0 is even: positions b0 + k and b0 + k + 1, k even, share a dword)
                    const int32_t v = diff[(b0 + (uint32_t)k) >> 1], lo16 = (int32_t)(int16_t)(uint16_t)((uint32_t)v & 0xFFFFu);
                    dk = (k & 1) ? ((v - lo16) >> 16) : lo16;
                } else {
                    dk = diff[b0 + k];
                }
                s += dk;
                d[k] = s;
            }
        }
        const int32_t inc = (int32_t)rb_wave_scan_incl((uint32_t)s);
        if (lane == 63) wsum[wib] = inc;
        __syncthreads();
        int32_t before = inc - s;
        for (uint32_t k = 0; k < wib; k++) before += wsum[k];
        int32_t mx = 0;
        uint32_t cov = 0, cbits = 0;
#pragma unroll
        for (int k = 0; k < NF_PER_THREAD; k++) {
            const int32_t dep = before + d[k];
            if constexpr (U8T) cbits |= dep > 0 ? (1u << k) : 0u;
            else diff[b0 + k] = dep;
            if (b0 + k < n_pos) {
                mx = mx > dep ? mx : dep;
                cov += dep > 0;
            }
        }
        if constexpr (U8T) covb[tix] = (uint8_t)cbits;
        // (a wave's maximum and sum first: one LDS atomic per wave instead of 64 to one address)
        const uint32_t wmx = rb_readlane<uint32_t>(rb_wave_scan_incl_max_u32((uint32_t)mx), 63), wcov = rb_wave_sum_u32(cov);
        if (lane == 0) {
            atomicMax(&blk_max, wmx);
            atomicAdd(&blk_cov, wcov);
        }
    }
    __syncthreads();
#endif
    NF_STAMP(5);
    NF_PHASE(4, 4, 5);
#if NF_STOP != 4 && !defined(NF_NO_CTR)
    if (tix == 0) {
        atomicMax((unsigned long long *)&p.counters->max_depth, (unsigned long long)blk_max);
        atomicAdd((unsigned long long *)&p.counters->n_covered, (unsigned long long)blk_cov);
    }
#endif
    uint4 *__restrict__ out = reinterpret_cast<uint4 *>(p.counts + 4ull * T.out);
#if NF_STOP == 1
    if (blk_max != 0x7FFFFFFFu) return;
#endif
    if constexpr (U8T) {
        static_assert(NF_PER_THREAD <= 8, "one byte of coverage flags per thread");
        for (uint32_t k = tix; k < n_pos; k += NF_THREADS) {
            const uint32_t v = cnt[10u * ((k + 8u) >> 3) + ((k + 8u) & 7u)];
            const bool covered = (covb[k / NF_PER_THREAD] >> (k % NF_PER_THREAD)) & 1u;
            out[k] = make_uint4((v & 255u) | (covered ? RB_NF_COVERED : 0u), (v >> 8) & 255u, (v >> 16) & 255u, v >> 24);
        }
    } else {
        for (uint32_t k = tix; k < n_pos; k += NF_THREADS) {
            const uint32_t ac = cnt[nf_slot(k)], gt = cnt[nf_slot(k) + 1];
            out[k] = make_uint4((ac & 0xFFFFu) | (diff[k] > 0 ? RB_NF_COVERED : 0u), ac >> 16, gt & 0xFFFFu, gt >> 16);
        }
    }
    NF_STAMP(6);
    NF_PHASE(5, 5, 6);
    NF_PHASE(6, 0, 6);
#ifdef NF_DIAG
    if ((t & 63u) == 0u && (tix & 63u) == 0u) atomicAdd(reinterpret_cast<unsigned long long *>(p.blk) + 8 + 7, 1ull);
#endif
}

// The build of the usual tiles (byte counters, byte differences): a workgroup per tile, the dispatcher hands them out.  The other two walk
// the list of THEIR tiles (none on long-read data at ordinary coverage): workgroups that stay, each asking a cursor for its next entry
// while it works on the current one (a static stride loses to the dispatcher's order: tiles differ in work).
template <bool U8T, bool D8>
__global__ __launch_bounds__(NF_THREADS) __attribute__((amdgpu_waves_per_eu(D8 ? NF_U8_WPE : U8T ? 6 : 4))) void rb_k_nf_tiles(rb_nf_params p) {
    if constexpr (D8) {
        nf_one_tile<true, true>(p, blockIdx.x);
    } else {
        __shared__ uint32_t next_i[2];
        const uint32_t n = U8T ? p.wide_list[0] : p.wide_list[p.max_tiles + 1u];
        unsigned long long *cursor = reinterpret_cast<unsigned long long *>(p.blk) + (U8T ? 0 : 1); // (zeroed by the launch; the scans' partials are done with)
        uint32_t i = blockIdx.x, par = 0;
        while (i < n) {
            if (threadIdx.x == 0) next_i[par] = (uint32_t)atomicAdd(cursor, 1ull);
            nf_one_tile<U8T, false>(p, U8T ? p.wide_list[1u + i] : p.wide_list[p.max_tiles - i]);
            __syncthreads(); // (the next tile zeroes the counters this one's output loop reads; and next_i is everybody's now)
            i = gridDim.x + next_i[par];
            par ^= 1u;
        }
    }
}

extern "C" size_t rb_nf_tile_positions(void) { return NF_TILE; }
extern "C" size_t rb_nf_scan_blocks(uint64_t n) { return (size_t)((n + NF_SCAN_PER_BLOCK - 1) / NF_SCAN_PER_BLOCK); }

extern "C" hipError_t rb_launch_exclusive_scan(uint64_t *v, uint64_t n, uint64_t *block_sums, uint64_t *total_out, hipStream_t stream);
extern "C" hipError_t rb_fill_async(void *dst, int value, size_t bytes, hipStream_t stream);

extern "C" hipError_t rb_launch_nucfreq(const rb_nf_params *pp, hipStream_t stream) {
    const rb_nf_params p = *pp;
    if (p.n_reads) {
        hipLaunchKernelGGL(rb_k_nf_read_spans, dim3((unsigned)((p.n_reads + 255) / 256)), dim3(256), 0, stream, p);
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
    if (p.n_reads > RB_NF_DEPTH_CAP) hipLaunchKernelGGL(rb_k_nf_crowded, dim3((unsigned)((p.n_reads + 255) / 256)), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(rb_k_nf_deep_regions, dim3((unsigned)((p.n_regions + 255) / 256)), dim3(256), 0, stream, p);
    hipLaunchKernelGGL(rb_k_nf_admit, dim3(512), dim3(64), 0, stream, p);
    e = rb_fill_async(p.wide_list, 0, 4, stream);
    if (e == hipSuccess) e = rb_fill_async(p.wide_list + p.max_tiles + 1, 0, 4, stream);
    if (e == hipSuccess) e = rb_fill_async(p.blk, 0, 16, stream); // (the two list walkers' cursors)
#ifdef NF_DIAG
    if (e == hipSuccess) e = rb_fill_async(p.blk + 8, 0, 64, stream);
#endif
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rb_k_nf_tile_desc, dim3((unsigned)((p.max_tiles + 255) / 256)), dim3(256), 0, stream, p);
    hipLaunchKernelGGL((rb_k_nf_tiles<true, true>), dim3((unsigned)p.max_tiles), dim3(NF_THREADS), 0, stream, p);
    // the list walkers: as many workgroups as the device holds at a time (three / two per CU by their LDS)
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    hipLaunchKernelGGL((rb_k_nf_tiles<true, false>), dim3((unsigned)std::min<uint64_t>(p.max_tiles, 3ull * (uint64_t)cus)), dim3(NF_THREADS), 0, stream, p);
    hipLaunchKernelGGL((rb_k_nf_tiles<false, false>), dim3((unsigned)std::min<uint64_t>(p.max_tiles, 2ull * (uint64_t)cus)), dim3(NF_THREADS), 0, stream, p);
    return hipGetLastError();
}
