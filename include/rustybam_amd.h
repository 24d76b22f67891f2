/*
 * rustybam_amd.h -- C ABI of the MI355X (gfx950) CIGAR-walk engine.
 *
 * Drop-in boundary for the per-record CIGAR walk of mrvollger/rustybam v0.1.33.  The
 * reference has no FFI seam; the entry points below are what a Rust host would bind with
 * `extern "C"` in place of the following calls (file:line into the reference's src/):
 *
 *   rb_dev_scan_records   <- PafRecord::check_integrity / infer_n_bases   paf.rs:825-857, :631-654
 *                            (called per record by Paf::from_file         paf.rs:70)
 *                            PafRecord::remove_trailing_indels            paf.rs:656-783
 *                            bamstats::add_stats_from_cigar/stats_from_paf bamstats.rs:107-154, :91-105
 *   rb_dev_liftover       <- liftover::trim_paf_by_rgns                   liftover.rs:134-167
 *                            (= trim_helper :107-132 + trim_paf_rec_to_rgn :17-105, which in turn
 *                             replace aligned_pairs paf.rs:501-538, tpos_to_idx(_match) :541-561,
 *                             subset_cigar/collapse_long_cigar :593-620, paf_overlaps_rgn :622-627)
 *   rb_dev_break          <- liftover::break_paf_on_indels                liftover.rs:182-226
 *   rb_dev_swap           <- paf::paf_swap_query_and_target               paf.rs:1050-1094
 *   rb_dev_overlap_split  <- trim_overlap::trim_overlapping_pafs          trim_overlap.rs:36-86
 *   rb_dev_trim_select    <- the pair scan / selection of Paf::overlapping_paf_recs   paf.rs:223-284
 *                            + PafRecord::truncate_record_by_query        paf.rs:785-823
 *                            (the pass / recursion driver Paf::overlapping_paf_recs, paf.rs:210-305, stays on the host)
 *   rb_dev_nucfreq        <- nucfreq::nucfreq / region_nucfreq            nucfreq.rs:61-95, :111-125
 *
 * Conventions
 *   - Plain C types only.  Every `rb_dev_*` pointer argument is a DEVICE pointer (HBM) owned by
 *     the caller; the call enqueues kernels on the context's HIP stream and returns without
 *     synchronising.  `rb_host_*` take HOST pointers and do H2D / kernels / D2H themselves.
 *   - Return value: RB_OK (0) or a negative rb_error; no exceptions or unwinding cross the ABI.
 *     Where the reference would panic or silently drop a (record, window) pair the outcome is
 *     a per-record / per-hit `status` in the result rows (enum rb_status).
 *   - One context per host thread; a context pins one device and one stream.
 *   - The library never falls back to the CPU: without a usable gfx950 device every compute
 *     entry point fails with RB_E_NO_DEVICE.
 *
 * Data model (all little-endian, in HBM):
 *   ops[]      u32  packed BAM encoding  len << 4 | op   (M0 I1 D2 N3 S4 H5 P6 =7 X8); 16-byte aligned (128 is faster), and
 *              readable -- contents ignored -- up to the next multiple of 32 ops behind the last one.
 *              A length of 2^28 and more (rust-htslib's Cigar holds a u32; a chromosome-long `=` of a plant genome aligned to
 *              itself has one) takes TWO words: (len & (2^28 - 1)) << 4 | op, then the CONTINUATION word
 *              (len >> 28) << 4 | RB_OP_CONT.  Inputs may hold such pairs and outputs (out_ops) hold them wherever a clipped
 *              or merged op is that long; every count of "ops" in this interface (op_off, first_op, n_ops, lead_ops,
 *              trail_ops, out_n) counts WORDS.  A record with a continuation word is not "regular": it takes the general
 *              kernels (same results, slower).  A continuation word with no op in front of it is a word of an unknown code:
 *              it consumes nothing.
 *   op_off[]   u64  [n_rec + 1] exclusive prefix of ops-per-record
 *   t_st,t_en,q_st,q_en u64 [n_rec];  strand u8 ('+' / '-');  contig u32 (dense ids, host keeps names)
 *   windows: contig u32, st u64, en u64   [n_win]  in BED file order
 */
#ifndef RUSTYBAM_AMD_H
#define RUSTYBAM_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RB_ABI_VERSION 1

enum rb_op { RB_OP_M = 0, RB_OP_I = 1, RB_OP_D = 2, RB_OP_N = 3, RB_OP_S = 4, RB_OP_H = 5, RB_OP_P = 6, RB_OP_EQ = 7, RB_OP_X = 8,
             RB_OP_CONT = 14 /* continuation word: bits 28..31 of the length of the op in front of it (see "ops[]" above) */ };

typedef enum rb_error {
    RB_OK = 0,
    RB_E_INVALID = -1,   /* bad argument */
    RB_E_NO_DEVICE = -2, /* no gfx950 device / HIP runtime unusable: there is NO CPU fallback */
    RB_E_HIP = -3,       /* a HIP call failed; see rb_ctx_last_error */
    RB_E_CAPACITY = -4,  /* caller-provided output buffer too small; counters say what is needed */
    RB_E_NOMEM = -5
} rb_error;

/* Per-record and per-hit outcome.  0 = Some(record).  1..5 = the reference returns None (pair is
 * silently dropped).  >= 16 = the reference panics at the cited line. */
typedef enum rb_status {
    RB_ST_OK = 0,
    RB_ST_NONE_INDEL = 1,         /* liftover.rs:52-54   start_idx > end_idx                    */
    RB_ST_NONE_NOMATCH = 2,       /* liftover.rs:65-75                                          */
    RB_ST_NONE_EMPTY = 3,         /* liftover.rs:87-89                                          */
    RB_ST_NONE_INVERTED = 4,      /* liftover.rs:90-96                                          */
    RB_ST_NONE_INTEGRITY = 5,     /* liftover.rs:99-102                                         */
    RB_ST_PANIC_NOTFOUND = 16,    /* liftover.rs:31 / :42, paf.rs:792-793                       */
    RB_ST_PANIC_EMPTY_CIGAR = 17, /* paf.rs:663                                                 */
    RB_ST_PANIC_INTEGRITY_T = 18, /* paf.rs:827 through .unwrap() at paf.rs:70 / :782           */
    RB_ST_PANIC_INTEGRITY_Q = 19, /* paf.rs:839                                                 */
    RB_ST_PANIC_ALL_INDEL = 20,   /* paf.rs:757                                                 */
    RB_ST_PANIC_ASSERT = 21,      /* paf.rs:787-788                                             */
    RB_ST_PANIC_OVERFLOW = 22     /* u32 overflow of a CIGAR length sum (paf.rs:632-647)        */
} rb_status;

/* Rust slice::binary_search duplicate policy, low bit of `bsearch_policy`.
 * RB_LIFT_EARLY_EXIT may be OR-ed in: stop walking a record once all its windows are resolved
 * (same results; the reference always walks every record, so benchmarks leave it off). */
enum { RB_BSEARCH_MODERN = 0 /* rustc >= 1.82 (and < 1.52) */, RB_BSEARCH_LEGACY = 1 /* 1.52 .. 1.81 */, RB_LIFT_EARLY_EXIT = 16,
       /* RB_LIFT_DESCRIPTORS: do not copy the clipped cigars.  A host that still holds each record's cigar (the
        * reference does) only needs to know WHICH ops a clip keeps: for a row with RB_HIT_DESCRIPTOR set,
        * out_ops[out_off .. out_off + 4) = { first kept op (index into the record's ORIGINAL cigar), op count,
        * length of the first kept op, length of the last kept op }; every op in between is unchanged; a one-op clip
        * has length aln_len; RB_HIT_INSIDE rows keep all lengths.  Rows resolved by the generic kernel (irregular
        * cigars whose adjacent ops may merge) still carry real ops.  out_cap must be >= 4 * rows_cap + room for those. */
       RB_LIFT_DESCRIPTORS = 32,
       /* RB_LIFT_FUSED_SCAN (rb_dev_liftover, rb_dev_break): norm_rows is an OUTPUT.  rb_dev_scan_records need not have run: the call
        * looks at the ends of every record (remove_trailing_indels), verifies integrity and regularity while the clip kernel
        * streams the record, and runs the full record scan only for records that fail that check.  Rows of a record whose
        * norm row ends up with status != RB_ST_OK carry that status. */
       RB_LIFT_FUSED_SCAN = 64,
       /* RB_BREAK_ONE_WALK (rb_dev_break): the clip kernel finds the long indels itself while it streams a record (no separate pass
        * that collects the pieces first: the ops are read once).  The caller must look at counters->redo_two_walk afterwards: set,
        * the batch holds something this path does not take (an irregular record, a boundary the fast path cannot
        * resolve), the results are incomplete and the call is to be repeated without this flag (rb_host_break does). */
       RB_BREAK_ONE_WALK = 128,
       /* RB_LIFT_OP_STARTS (rb_dev_liftover, rb_dev_break; not with RB_LIFT_FUSED_SCAN or RB_LIFT_DESCRIPTORS): the batch is one that trim-paf has cut IN PLACE
        * (rb_dev_overlap_split with RB_TRIM_IN_PLACE + rb_dev_apply_pairs): batch->op_off[r] is where record r starts and says nothing
        * about where it ends, every extent comes from norm_rows (first_op, n_ops), which must be the finished rows of that batch.
        * The plan is the one built from the op offsets the batch had BEFORE the passes (records only shrink inside their old
        * extents, so its tiles still hold).  With it the README pipeline trim-paf | break-paf needs no rb_dev_gather_records
        * between the two: the clip kernels stream over the gaps the cuts left between the records of a tile.  A record a pass has
        * MOVED (an irregular record's clip, written behind the ops in use) must lie below batch->n_ops = the plan's op count: if the
        * passes moved any, gather first. */
       RB_LIFT_OP_STARTS = 1 << 20 };

/* rb_norm_row.flags / rb_reduce_row.flags */
enum {
    RB_F_REGULAR = 1u << 0,   /* only M I D N = X ops, every len >= 1, no two adjacent ops of one type, M / = / X at both ends */
    RB_F_STRIPPED = 1u << 1,  /* leading/trailing indels were removed (host appends _TO.<..>.<..>) */
    RB_F_HAS_M = 1u << 2,     /* cigar contains 'M' (bamstats.rs:145 warning)                      */
    RB_F_PROVISIONAL = 1u << 3, /* never visible to callers: row written from the record's ends only, not yet verified */
    RB_F_ENDS_NOT_MATCH = 1u << 4 /* never visible to callers: provisional row whose kept range does not start and end on M / = / X */
};
/* rb_hit_row.flags */
enum {
    RB_HIT_INSIDE = 1u << 0,  /* liftover.rs:23-25: record returned unchanged and keeps its OWN id */
    RB_HIT_GENERIC = 1u << 1, /* resolved by the generic (serial) kernel, informational             */
    RB_HIT_DESCRIPTOR = 1u << 2 /* out_ops holds a 4-word clip descriptor, not ops (RB_LIFT_DESCRIPTORS) */
};

/* ---- result rows (written by the device; 72 / 64 / 64 bytes) ---------------------------------- */
typedef struct rb_reduce_row { /* check_integrity + stats of the ORIGINAL record */
    uint64_t t_bases, q_bases;                  /* infer_n_bases: reference / query consuming      */
    uint32_t nmatch, aln_len;                   /* M+=+X lengths (quirk 9.3.1), all lengths        */
    uint32_t equal, diff, ins, del, matches;    /* bamstats.rs:26-30 (diff = X + M, matches = M)   */
    uint32_t ins_events, del_events;
    float id_by_all, id_by_events, id_by_matches; /* bamstats.rs:138-142, f32                      */
    uint32_t status;                            /* RB_ST_OK / PANIC_INTEGRITY_* / PANIC_OVERFLOW    */
    uint32_t flags;
} rb_reduce_row;

typedef struct rb_norm_row { /* the record after remove_trailing_indels (paf.rs:656-783) */
    uint64_t t_st, t_en, q_st, q_en;
    uint32_t first_op, n_ops;    /* kept op range, relative to the record's first op */
    uint32_t lead_ops, trail_ops;
    uint32_t nmatch, aln_len;    /* of the kept range */
    uint32_t status;
    uint32_t flags;
} rb_norm_row;

typedef struct rb_hit_row { /* one (record, window) pair that passed paf_overlaps_rgn */
    uint32_t rec, win;           /* record index in the batch; window index in BED order (break: piece #) */
    uint16_t status, flags;
    uint32_t out_n;              /* ops in the clipped cigar */
    uint64_t t_st, t_en, q_st, q_en;
    uint32_t nmatch, aln_len;
    uint64_t out_off;            /* first op of the clipped cigar in out_ops[] */
} rb_hit_row;

typedef struct rb_pair_row { /* trim-paf: one (left, right) overlap pair, 128 bytes */
    uint64_t split_idx;          /* trim_overlap.rs:69-76 */
    int32_t split_score;
    uint32_t status;
    uint64_t t_st[2], t_en[2], q_st[2], q_en[2]; /* [0]=left [1]=right after truncate_record_by_query */
    uint32_t nmatch[2], aln_len[2];
    uint64_t out_off[2];
    uint32_t out_n[2];
    uint64_t _pad;
} rb_pair_row;

typedef struct rb_counters { /* device-written job summary, 64 bytes */
    uint64_t n_hits;             /* rows the job wants to write                                  */
    uint64_t out_ops_needed;     /* upper bound of out_ops[] capacity that makes the job fit      */
    uint64_t out_ops_used;       /* highest op index written + 1                                  */
    uint64_t n_generic;          /* hits routed to the generic kernel                             */
    uint32_t overflow;           /* != 0: rows or out_ops capacity exceeded, results incomplete   */
    uint32_t phase[5];           /* diagnostics (debug_skip & 32): shader-clock sums per phase of the clip kernel, units of 16 cycles;
                                    without debug_skip: phase[3] = tiles of short records the call ran (0: none), phase[4] = records of
                                    those tiles that the tile kernel handed back to the per-record kernel (informational) */
    uint32_t brk_scratch_short;  /* RB_BREAK_ONE_WALK only: != 0: a scratch-row cursor ran out before the rows did; n_hits then asks for more */
    uint32_t redo_two_walk;      /* RB_BREAK_ONE_WALK only: != 0: the batch holds what the one-walk path does not take (irregular
                                    records, boundaries only the generic kernel resolves): results incomplete, call again without the flag */
} rb_counters;

/* per-record outcome of rb_dev_parse_cigars */
enum { RB_TEXT_OK = 0, RB_TEXT_BAD = 1 /* the reference's "Unable to parse cigar string." panic */, RB_TEXT_TOO_LONG = 2 /* a length >= 2^28 */,
       RB_TEXT_UNUSUAL = 3 /* a number written with ten or more digits (zero padded, or past u32): not decided on the device, parse it on the host */ };

/* ---- views over caller-owned device memory ---------------------------------------------------- */
typedef struct rb_batch_view {
    uint64_t n_rec, n_ops;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint64_t *t_st, *t_en, *q_st, *q_en;
    const uint8_t *strand;
    const uint32_t *contig;
} rb_batch_view;

typedef struct rb_windows_view {
    uint64_t n_win;
    const uint32_t *contig;
    const uint64_t *st, *en;
} rb_windows_view;

typedef struct rb_ctx rb_ctx;
typedef struct rb_plan rb_plan; /* host-built schedule for one (batch, windows) pair */

/* ---- context ---------------------------------------------------------------------------------- */
int rb_abi_version(void);
int rb_device_count(void);
/* `hip_stream`: a hipStream_t to enqueue on (e.g. torch's current stream), or NULL for a private one. */
int rb_ctx_create(int device, void *hip_stream, rb_ctx **out);
void rb_ctx_destroy(rb_ctx *ctx);
const char *rb_ctx_last_error(const rb_ctx *ctx);
int rb_ctx_sync(rb_ctx *ctx);
void *rb_ctx_stream(rb_ctx *ctx);
/* Measurement aid: when enabled, every rb_dev_liftover / rb_dev_break call brackets its dominant
 * (streaming clip) kernel with HIP events on the context's stream.  rb_ctx_get_timing synchronises
 * and returns the per-call durations in milliseconds, oldest first (ring of the last 256 calls). */
int rb_ctx_set_timing(rb_ctx *ctx, int enabled);
int rb_ctx_get_timing(rb_ctx *ctx, double *ms_out, int cap, int *n_out);

/* Device memory.  Hosts that keep a batch resident across calls should take the batch, workspace, rows and
 * output arenas from here: requests of 256 MB and more are built from 2 MB physical chunks (hipMemCreate)
 * mapped into one virtual range, which spreads a multi-GB array evenly over the HBM channels whatever the
 * driver's free list looks like.  On the headline batch the clip kernel takes 9.3-9.4 ms per launch on such
 * memory against 10.2-11.9 ms on plain hipMalloc memory and 18-20 ms on one physically contiguous block
 * (profiles/r03_alloc_summary.md).  Smaller requests are plain hipMalloc.  RB_ALLOC_MODE in the environment
 * overrides: default (hipMalloc always) | chunks | scatter (chunks in shuffled order) | contiguous.
 * Pointers are 2 MB aligned when chunked, 256 B otherwise; free only with rb_dev_free. */
int rb_dev_alloc(rb_ctx *ctx, size_t bytes, void **dev_ptr);
int rb_dev_free(rb_ctx *ctx, void *dev_ptr);
/* which route a buffer of rb_dev_alloc took: 1 = 2 MB physical chunks, 0 = plain hipMalloc (a small request, or the fallback when the
 * chunked route failed -- the two routes differ by 10-20 % in the clip kernel's time, so bench.py reports it) */
int rb_dev_alloc_mode(rb_ctx *ctx, const void *dev_ptr);
/* ADDRESS SPACE.  A chunked buffer that is FREED gives its memory back, not its virtual range: on this ROCm a kernel's stores into a range
 * the process had mapped before went to the OLD physical pages (tools/alloc_probe2.py), so rb_dev_free retires the range for the life of
 * the process.  A host that allocates and frees 40-75 GB batches therefore spends that much address space per batch; the library counts it
 * and stops using the chunked route (plain hipMalloc instead: slower pages, rb_dev_alloc_mode says 0) once live + retired ranges reach
 * RB_ALLOC_VA_CAP_GB (default 32768 = 32 TB of the 128 TB a process has).  The way around it is not to free:
 *   rb_dev_release     gives a buffer back to the CONTEXT: it stays mapped, keeps its physical pages (and so what rb_dev_alloc_placed
 *                      chose them for), and the next rb_dev_alloc / rb_dev_alloc_placed[_by] of exactly the same size on this context
 *                      returns it -- no address space retired, no 14 us per chunk, no placement measured again (a released PLACED
 *                      buffer satisfies rb_dev_alloc_placed[_by] at once: *kept = -1, no score).  Buffers below 1 MB are simply freed.
 *                      The context's stream is synchronised; work on other streams must be finished.  At most RB_ALLOC_CACHE_GB
 *                      (default 96) gigabytes are held, the oldest are freed first; rb_ctx_destroy frees what is held.
 *   rb_dev_cache_trim  frees held buffers, oldest first, until at most keep_bytes remain.
 *   rb_dev_alloc_stats out[0] live chunked bytes (process), out[1] bytes this context holds for reuse, out[2] retired address space
 *                      (process), out[3] the cap on out[0] + out[2], out[4] requests of this context that fell back to plain hipMalloc. */
int rb_dev_release(rb_ctx *ctx, void *dev_ptr);
int rb_dev_cache_trim(rb_ctx *ctx, uint64_t keep_bytes);
int rb_dev_alloc_stats(rb_ctx *ctx, uint64_t out[5]);
/* Transfers of 8 MB and more go through the context's pinned staging ring (two page-locked 32 MB chunks, hipHostMalloc): the
 * host side of a chunk is copied on several host threads while the DMA of the other chunk runs, so pageable caller memory moves
 * at the link's rate and host_src may be reused as soon as rb_dev_upload returns (the DMAs are queued on the context's stream).
 * Smaller uploads are plain asynchronous copies: host_src must then stay valid until the stream has been synchronised. */
int rb_dev_upload(rb_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);   /* async with respect to the device */
int rb_dev_download(rb_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes); /* synchronises */
int rb_dev_memset(rb_ctx *ctx, void *dev_dst, int value, size_t bytes);              /* async */

/* ---- K1: one pass over every record's ops ----------------------------------------------------- *
 * reduce_rows and/or norm_rows may be NULL.  [n_rec] each. */
int rb_dev_scan_records(rb_ctx *ctx, const rb_batch_view *batch, rb_reduce_row *reduce_rows, rb_norm_row *norm_rows);

/* ---- schedule (host side, no device work besides small uploads) ------------------------------- *
 * Built from HOST copies of the small per-record / per-window arrays:
 *   - canonical output order: contigs by first appearance, then record order (liftover.rs:151-164)
 *   - longest-first launch order (load balance; does not affect results)
 *   - windows grouped by contig in BED order, with a per-contig "monotone" flag
 *   - (round 5) TILES of short records: runs of records that lie one behind the other in ops[], 8 .. 2048 ops each, at most 32 of them and
 *     4064 ops together.  rb_dev_liftover / rb_dev_break stream a tile with ONE wavefront (k_tile.hip) instead of one per record; longer
 *     records keep the per-record kernel, and a tile the tile kernel does not take (an irregular or stripped record, an integrity failure,
 *     more than 64 hits, ...) is handed to it record by record.  Same rows, same clips either way; counters.phase[3..4] tell how it went.
 *     RB_TILE=0 in the environment switches the tiles off, RB_SHORT_MAX=<ops> moves the line (both read here, per plan).
 * `windows` may be NULL (break-paf / stats only).  */
int rb_plan_create(rb_ctx *ctx, uint64_t n_rec, const uint64_t *op_off_host, const uint32_t *contig_host,
                   uint64_t n_win, const uint32_t *w_contig_host, const uint64_t *w_st_host, const uint64_t *w_en_host,
                   rb_plan **out);
void rb_plan_destroy(rb_plan *plan);
/* How rb_plan_create cuts a batch into tiles and orders its schedule, on plain host arrays (no device, no context): sched_out[n_rec] =
 * the records longer than short_max (0: the default, 2048) longest first -- *n_long_out of them --, then the others in memory order;
 * tiles_out[3 t ..] = {first record | pass-through << 31, records, schedule slot of the first record}.  Returns the number of tiles
 * (all of them are counted, tiles_cap of them written), -1 on bad arguments. */
int64_t rb_plan_tiles_host(uint64_t n_rec, const uint64_t *op_off_host, uint64_t short_max, uint32_t *sched_out, uint32_t *tiles_out,
                           uint64_t tiles_cap, uint64_t *n_long_out);
/* bytes of device workspace rb_dev_liftover / rb_dev_break need for this plan and row capacity (the workspace must be
 * 256-byte aligned, as rb_dev_alloc returns it) */
size_t rb_plan_workspace_bytes(const rb_plan *plan, uint64_t rows_cap);
/* diagnostics (bench.py's box block): byte offset, inside that workspace, of [n_rec] u32 words in which the diagnostics build of the
 * clip kernel leaves the 100 MHz clock (low 32 bits) at which the wave of schedule slot w was done.  Written only by calls whose
 * policy carries the undocumented debug bits; the product's kernels never touch the area. */
size_t rb_plan_diag_stamps_offset(const rb_plan *plan, uint64_t rows_cap);
/* out_ops capacity (in ops) with which rb_dev_liftover (for_break = 0) / rb_dev_break (1) can emit every clip while the record
 * streams past: the clipped cigars go to up to 4 positional copies of the batch's op index space (as many as the sorted window
 * lists overlap deep; two for break-paf), and only rows' out_off says where a clip is.  A smaller out_cap still works -- clips
 * without a place are then copied by a second kernel into what room there is, and counters report overflow / out_ops_needed as
 * before -- a larger one is never needed for sorted, at most 4-deep window lists. */
uint64_t rb_plan_out_capacity(const rb_plan *plan, int for_break);

/* ---- liftover --------------------------------------------------------------------------------- *
 * For every record (status OK in norm_rows) and every window of the same contig with
 * t_en > st && t_st < en, in canonical order, writes one rb_hit_row and the clipped cigar.
 *   rows      [rows_cap]      out_ops [out_cap]      counters [1]      workspace [rb_plan_workspace_bytes]
 * On capacity overflow counters->overflow != 0 and counters say what is needed; the caller
 * enlarges and calls again (the host wrapper below does that). */
int rb_dev_liftover(rb_ctx *ctx, const rb_plan *plan, const rb_batch_view *batch, const rb_norm_row *norm_rows,
                    int bsearch_policy, void *workspace, rb_hit_row *rows, uint64_t rows_cap, uint32_t *out_ops,
                    uint64_t out_cap, rb_counters *counters);

/* ---- break-paf -------------------------------------------------------------------------------- *
 * Pieces between indels longer than max_size, record order then piece order; same row shape,
 * rb_hit_row.win = piece ordinal among the candidate windows of the record. */
int rb_dev_break(rb_ctx *ctx, const rb_plan *plan, const rb_batch_view *batch, const rb_norm_row *norm_rows,
                 uint32_t max_size, int bsearch_policy, void *workspace, rb_hit_row *rows, uint64_t rows_cap,
                 uint32_t *out_ops, uint64_t out_cap, rb_counters *counters);

/* ---- invert ----------------------------------------------------------------------------------- *
 * out_ops[op_off[r] .. op_off[r+1]) = I<->D swapped, order reversed when strand[r] == '-'.
 * (The header swap t<->q is a host-side field swap.) */
int rb_dev_swap(rb_ctx *ctx, const rb_batch_view *batch, uint32_t *out_ops);

/* ---- trim-paf: one pass of independent (left, right) overlap pairs ------------------------------ *
 * left[i] / right[i] index records of the batch (left = smaller q_st, paf.rs:252-256).  For each pair:
 * split point = first arg-max of prefix(left scores) + suffix(right scores) over the overlapped query
 * bases (trim_overlap.rs:50-76), then both records are clipped by query range.  pair_out_off[i] = first
 * op of pair i's output in out_ops; the pair needs room for n_ops(left) + n_ops(right) ops.  norm_rows
 * come from rb_dev_scan_records on the same batch.
 * RB_TRIM_IN_PLACE OR-ed into bsearch_policy (out_ops must then be batch->ops, the resident-batch set-up of rb_dev_apply_pairs):
 * a clip by query range keeps a RUN of the record's ops and changes only the lengths of the run's first and last op, so a pair of
 * regular records is not copied -- the two end words are rewritten where they are and rows[].out_off points at the run inside the
 * array (out_n = its length).  The record's ops outside the run stay where they were, no longer part of it; the batch's original
 * CIGARs are gone after the call.  Pairs with an irregular record still write their clips at pair_out_off. */
enum { RB_TRIM_IN_PLACE = 256 };
int rb_dev_overlap_split(rb_ctx *ctx, const rb_batch_view *batch, const rb_norm_row *norm_rows, uint64_t n_pairs,
                         const uint32_t *left, const uint32_t *right, const uint64_t *pair_out_off, int match_score,
                         int diff_score, int indel_score, int bsearch_policy, rb_pair_row *rows, uint32_t *out_ops);

/* rb_dev_trim_reserve (optional): rb_dev_overlap_split keeps, per context, a list of the pairs its first kernel leaves to the ones
 * behind it; the list grows with the largest n_pairs seen, and growing means a stream synchronisation and an allocation inside that
 * call.  A host that knows how many pairs a pass can have (one per query group) sizes the list up front with this. */
int rb_dev_trim_reserve(rb_ctx *ctx, uint64_t n_pairs);

/* ---- trim-paf with the batch resident on the device across the passes of Paf::overlapping_paf_recs (paf.rs:210-305) ----------
 * rb_dev_apply_pairs: the records a pass has cut become the batch's current records.  For every pair k with status RB_ST_OK and
 *     side s (0 = left[k], 1 = right[k]):  op_off[rec] = rows[k].out_off[s] and norm_rows[rec] = the clipped record (coordinates,
 *     nmatch, aln_len, first_op 0, n_ops = out_n; a clip starts and ends on a match op, so the remove_trailing_indels of the next
 *     pass, paf.rs:218-220, finds nothing to strip).  out_off must index the SAME array as the batch's ops: call
 *     rb_dev_overlap_split with out_ops = batch->ops and pair_out_off pointing behind the ops in use.  From then on op_off is a
 *     table of starts, no longer a prefix array: only rb_dev_overlap_split, rb_dev_gather_records and rb_dev_format_cigars -- which
 *     take a record's extent from norm_rows / explicit counts -- may be used on the batch.
 * rb_dev_gather_records: the current records (ops[op_off[r] + first_op ..][n_ops], records whose status is not RB_ST_OK count as
 *     empty) copied into a dense array in record order; new_op_off [n_rec + 1] is its exclusive prefix.  new_ops NULL: sizes only.
 *     scratch: rb_text_scratch_bytes(n_rec) bytes.  The dense array with the coordinates of norm_rows is an ordinary batch again
 *     (e.g. for rb_dev_break: the README pipeline trim-paf | break-paf). */
int rb_dev_apply_pairs(rb_ctx *ctx, uint64_t n_pairs, const uint32_t *left, const uint32_t *right, const rb_pair_row *rows,
                       uint64_t *op_off, rb_norm_row *norm_rows);

/* rb_dev_trim_select: which pairs a pass of Paf::overlapping_paf_recs cuts, found on the device (paf.rs:223-284: the pair scan per
 *     query name over bed::get_overlap, the contained flags :244-249, "pairs by overlap descending, then the first pair of every
 *     query name" :262-284 = per group the pair of largest overlap, among equals the first in scan order).  order [n_rec] = the
 *     records stably sorted by query name (:223), grp_off [n_groups + 1] = where the groups of equal names begin in it (both device
 *     arrays the caller builds once).  Outputs: contained [n_rec] by RECORD (this pass's flags: the ones that count are the last
 *     pass's, :224), the chosen pairs dense in left / right / pair_out_off (room for n_groups each; pair k writes its clips at
 *     pair_out_off[k] >= out_base, n_ops(left) + n_ops(right) apart), and *pass (device memory, 64 bytes): n_pairs, n_deferred (pairs
 *     left for a later pass: the recursion of :286-288 goes on while it is not 0; the exact count as long as no single query group
 *     leaves more than (2^32 - 1) / n_groups pairs, a lower bound above that), ops_end (first op behind this pass's clips).
 *     scratch: rb_trim_select_scratch_bytes(n_groups).  Then rb_dev_overlap_split + rb_dev_apply_pairs on the n_pairs pairs, and
 *     rb_dev_trim_check folds the pair rows' statuses into pass->bad_status (0: every pair was cut). */
typedef struct rb_trim_pass {
    uint64_t n_pairs, n_deferred, ops_end;
    uint32_t bad_status, _pad;
    uint64_t _reserved[4];
} rb_trim_pass; /* 64 B */
size_t rb_trim_select_scratch_bytes(uint64_t n_groups);
int rb_dev_trim_select(rb_ctx *ctx, uint64_t n_rec, uint64_t n_groups, const uint32_t *order, const uint64_t *grp_off,
                       const rb_norm_row *norm_rows, uint64_t out_base, uint8_t *contained, uint32_t *left, uint32_t *right,
                       uint64_t *pair_out_off, rb_trim_pass *pass, void *scratch);
int rb_dev_trim_check(rb_ctx *ctx, uint64_t n_pairs, const rb_pair_row *rows, rb_trim_pass *pass);
int rb_dev_gather_records(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const rb_norm_row *norm_rows,
                          uint64_t *new_op_off, uint32_t *new_ops, void *scratch);

/* ---- host-buffer wrappers (H2D, kernels, D2H; results malloc'ed, free with rb_host_free) ------ */
int rb_host_scan_records(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off,
                         const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en,
                         const uint8_t *strand, rb_reduce_row *reduce_rows, rb_norm_row *norm_rows);
int rb_host_liftover(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                     const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                     const uint32_t *contig, uint64_t n_win, const uint32_t *w_contig, const uint64_t *w_st,
                     const uint64_t *w_en, int bsearch_policy, rb_norm_row *norm_rows_out /* [n_rec] or NULL */,
                     rb_hit_row **rows, uint64_t *n_rows, uint32_t **out_ops, uint64_t *n_out, rb_counters *counters);
int rb_host_break(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                  const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                  uint32_t max_size, int bsearch_policy, rb_norm_row *norm_rows_out, rb_hit_row **rows,
                  uint64_t *n_rows, uint32_t **out_ops, uint64_t *n_out, rb_counters *counters);
int rb_host_swap(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint8_t *strand,
                 uint32_t *out_ops);
/* rows [n_pairs] caller-allocated; out_ops malloc'ed (dense, rows' out_off rebased onto it) */
int rb_host_overlap_split(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                          const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                          uint64_t n_pairs, const uint32_t *left, const uint32_t *right, int match_score, int diff_score,
                          int indel_score, int bsearch_policy, rb_pair_row *rows, uint32_t **out_ops, uint64_t *n_out);
void rb_host_free(void *p);

/* ---- CIGAR text <-> packed ops on the device (the data format either side of the path) ----------
 *
 * rb_dev_parse_cigars   <- CigarString::try_from(value.as_bytes()).expect(..)       paf.rs:398-399
 *     text      device bytes holding the CIGAR strings of n_rec records (the `cg:Z:` values, without the tag);
 *               16-byte aligned, readable up to the next multiple of 16 past the last string
 *     text_off  [n_rec + 1] start of every string;  text_end [n_rec] or NULL (NULL: strings are back to back)
 *     op_off    [n_rec + 1] OUT  exclusive prefix of ops per record (op_off[n_rec] = total)
 *     ops       OUT, capacity ops_cap (total text bytes / 2 is always enough: an op is at least two characters)
 *     status    [n_rec] OUT  RB_TEXT_OK / RB_TEXT_BAD (the reference panics: "Unable to parse cigar string.") /
 *               RB_TEXT_TOO_LONG (a length >= 2^28 cannot be packed; the reference would go on)
 *     scratch   rb_text_scratch_bytes(n_rec) bytes
 * rb_dev_format_cigars  <- impl Display for CigarString inside impl Display for PafRecord   paf.rs:923-944
 *     item i prints ops[first[i] .. first[i] + count[i]) as <len><op>...; first_len / last_len (arrays or NULL,
 *     0 = keep) replace the length of the first / last op -- the clip descriptors of RB_LIFT_DESCRIPTORS:
 *     {first kept op, op count, first length, last length}; a one-op item with both prints first + last - len.
 *     ops_alt   optional second source array: an item whose first[] has bit 63 set takes its ops from ops_alt[]
 *               (clips the generic kernel copied out live in out_ops[], descriptor clips point into the batch)
 *     text_off  [n_items + 1] OUT exclusive prefix of bytes;  text OUT (NULL: sizes only), capacity text_cap
 *               (11 bytes per op suffice)
 * rb_host_liftover_text: liftover from CIGAR text to CIGAR text.  text holds the records' `cg:Z:` values at
 *     [cig_off[r], cig_end[r]); they are parsed on the device (cig_status[r] = RB_TEXT_*; if any is not OK nothing else
 *     is computed), scanned (reduce_out / norm_out, either may be NULL), lifted over the windows in descriptor mode, and the
 *     clipped CIGAR of every hit row is printed on the device: row k's text is row_text[row_text_off[k] .. [k + 1]) (empty
 *     for rows whose status is not RB_ST_OK).  rows / row_text_off / row_text are malloc'ed (rb_host_free).
 */
size_t rb_text_scratch_bytes(uint64_t n);
int rb_dev_parse_cigars(rb_ctx *ctx, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_end, uint64_t n_rec,
                        uint64_t *op_off, uint32_t *ops, uint64_t ops_cap, uint8_t *status, void *scratch);
int rb_dev_format_cigars(rb_ctx *ctx, const uint32_t *ops, const uint32_t *ops_alt, uint64_t n_items, const uint64_t *first,
                         const uint32_t *count, const uint32_t *first_len, const uint32_t *last_len, uint64_t *text_off, uint8_t *text,
                         uint64_t text_cap, void *scratch);
/* host-buffer forms: H2D, kernels, D2H.  *ops / *text are malloc'ed (rb_host_free) */
int rb_host_parse_cigars(rb_ctx *ctx, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_end, uint64_t n_rec,
                         uint64_t *op_off, uint32_t **ops, uint8_t *status);
int rb_host_format_cigars(rb_ctx *ctx, const uint32_t *ops, uint64_t n_ops, uint64_t n_items, const uint64_t *first, const uint32_t *count,
                          const uint32_t *first_len, const uint32_t *last_len, uint64_t *text_off, uint8_t **text);
int rb_host_liftover_text(rb_ctx *ctx, uint64_t n_rec, const uint8_t *text, uint64_t text_bytes, const uint64_t *cig_off,
                          const uint64_t *cig_end, const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en,
                          const uint8_t *strand, const uint32_t *contig, uint64_t n_win, const uint32_t *w_contig, const uint64_t *w_st,
                          const uint64_t *w_en, int bsearch_policy, uint8_t *cig_status, rb_reduce_row *reduce_out, rb_norm_row *norm_out,
                          rb_hit_row **rows, uint64_t *n_rows, uint64_t **row_text_off, uint8_t **row_text, rb_counters *counters);
/* the same around rb_dev_break (break-paf, main.rs:271-281), and the record scan alone (stats --paf, main.rs:50-58) */
int rb_host_break_text(rb_ctx *ctx, uint64_t n_rec, const uint8_t *text, uint64_t text_bytes, const uint64_t *cig_off,
                       const uint64_t *cig_end, const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en,
                       const uint8_t *strand, uint32_t max_size, int bsearch_policy, uint8_t *cig_status, rb_reduce_row *reduce_out,
                       rb_norm_row *norm_out, rb_hit_row **rows, uint64_t *n_rows, uint64_t **row_text_off, uint8_t **row_text,
                       rb_counters *counters);
int rb_host_scan_text(rb_ctx *ctx, uint64_t n_rec, const uint8_t *text, uint64_t text_bytes, const uint64_t *cig_off, const uint64_t *cig_end,
                      const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                      uint8_t *cig_status, rb_reduce_row *reduce_out, rb_norm_row *norm_out);

/* ---- nucfreq: A/C/G/T counts at every covered reference position -----------------
 *
 * rb_dev_nucfreq  <- nucfreq::nucfreq (nucfreq.rs:61-95) over the reads nucfreq::region_nucfreq fetches (:111-125),
 *                    for all the 10 kb pieces main.rs:100-110 cuts the regions into, in one call.
 *   reads   BAM records in FILE ORDER (coordinate sorted: (tid, pos) non-decreasing, unplaced reads with tid -1 last):
 *           ops / op_off  the CIGAR words (BAM's encoding is the packed encoding of this ABI; the host resolves the CG:B,I
 *                         long-cigar convention as htslib does) and their exclusive prefix [n_reads + 1]
 *           seq / seq_off 4-bit bases exactly as in the BAM record (=ACMGRSVTWYHKDBN, high nibble first), read i starting at
 *                         byte seq_off[i] (any byte); seq itself 4-byte aligned and readable for 32 bytes past the last read
 *                         (the kernel fetches 16 bytes at a time);  l_seq bases per read;  tid, pos (0-based leftmost), flag
 *   regions rg_tid / rg_st / rg_en [n_regions]  half-open, 0-based; out_off [n_regions + 1] = exclusive prefix of en - st
 *           (n_positions = out_off[n_regions]); regions may overlap, each is computed on its own
 *   counts  OUT [4 * n_positions] u32: A, C, G, T at position rg_st[r] + k -> counts[4 * (out_off[r] + k) ..].  Bit 31 of the A
 *           word (RB_NF_COVERED) is set where the pileup reports the position at all (some read that passes the flag
 *           filter covers it, deletions and reference skips included): the reference prints only those positions
 *   read_status OUT [n_reads] rb_read_status.  FILTERED = tid < 0 or a flag of htslib's default pileup mask (UNMAP |
 *           SECONDARY | QCFAIL | DUP).  BAD_CIGAR = htslib's cursor would assert (no reference-consuming op, a lone op that
 *           is not M/=/X, a zero-length M/D/N/=/X) or the spans do not fit 31 bits.  SEQ_SHORT = a counted base lies past
 *           l_seq (the reference panics on the index).  Reads that are not OK contribute nothing.
 *   counters OUT: max_depth (reads covering one position, deletions included), n_covered, n_bad, unsorted, n_dropped,
 *           cap_overflow.  htslib's pileup (bam_plp_push) drops a read that starts where the iterator stands while more than
 *           maxcnt = 8000 reads are buffered; that rule is restated on the device per region (= per fetch of the reference),
 *           oracle/rb_oracle.c rbo_nucfreq holds the same restatement: n_dropped counts the (region, read) pairs it removed.
 *           cap_overflow = 1 when the bookkeeping of that rule did not fit (more than 15360 reads buffered at once, or regions
 *           deeper than the cap overlapping each other more than 64-fold): the counts are then not the reference's.  The same
 *           goes for max_depth > 65535 (16-bit counters).  rb_host_nucfreq returns RB_E_INVALID for all three, and for
 *           unsorted input.
 *   ws      rb_nucfreq_workspace_bytes(n_reads, n_regions, n_positions) bytes, 256-byte aligned
 */
typedef struct {
    uint64_t n_reads;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint8_t *seq;
    const uint64_t *seq_off;
    const uint32_t *l_seq;
    const int32_t *tid;
    const int64_t *pos;
    const uint32_t *flag;
} rb_reads_view;
enum rb_read_status { RB_RD_OK = 0, RB_RD_FILTERED = 1, RB_RD_BAD_CIGAR = 2, RB_RD_SEQ_SHORT = 3 };
typedef struct {
    uint64_t max_depth, n_covered, n_bad, unsorted, n_dropped, cap_overflow;
} rb_nucfreq_counters;
#define RB_NF_COVERED 0x80000000u
#define RB_NF_DEPTH_CAP 8000u /* htslib bam_plp_init: maxcnt */
size_t rb_nucfreq_workspace_bytes(uint64_t n_reads, uint64_t n_regions, uint64_t n_positions);
int rb_dev_nucfreq(rb_ctx *ctx, const rb_reads_view *reads, uint64_t n_regions, const int32_t *rg_tid, const uint64_t *rg_st,
                   const uint64_t *rg_en, const uint64_t *out_off, uint64_t n_positions, uint32_t *counts, uint32_t *read_status,
                   rb_nucfreq_counters *counters, void *ws, size_t ws_bytes);
/* host-buffer form: uploads ops[op_off[0] .. op_off[n_reads]) and the sequence bytes the reads span, runs, downloads.
 * counts [4 * sum(en - st)] and read_status [n_reads] are caller-allocated; read_status may be NULL */
int rb_host_nucfreq(rb_ctx *ctx, const rb_reads_view *reads, uint64_t n_regions, const int32_t *rg_tid, const uint64_t *rg_st,
                    const uint64_t *rg_en, uint32_t *counts, uint32_t *read_status, rb_nucfreq_counters *counters);

/* ---- verification aid (tests, bench; not a reference function): digest of hit rows and their clipped CIGARs ---------------
 * *digest += sum over i < n_rows of mix(row i) * (2 * (row_base + i) + 1), wrapping u64 (the caller zeroes *digest).  mix covers
 * rec + rec_base, win, status and -- for RB_ST_OK rows -- the INSIDE flag, out_n, the coordinates, nmatch, aln_len and every op of
 * the clip in order.  It does NOT cover out_off, nor how the clip is stored: a descriptor row (RB_HIT_DESCRIPTOR) is expanded
 * through batch->ops / batch->op_off.  Two runs that produce the same records in the same order have the same digest wherever
 * the ops were placed, and the digests of contiguous record-range shards add up to the digest of the whole batch when each shard
 * passes the rows / records that precede it as row_base / rec_base (the 1/2/4/8-GPU determinism check of bench.py). */
int rb_dev_digest_rows(rb_ctx *ctx, const rb_batch_view *batch, const rb_hit_row *rows, uint64_t n_rows, const uint32_t *out_ops,
                       uint64_t row_base, uint64_t rec_base, uint64_t *digest);

/* ---- diagnostics: the box (bench.py's `box` block; not a reference function) ------------------------------------------- *
 * Moves the clip kernel's memory mix without its instructions: every wave reads a 20 KiB stretch of src[0 .. src_bytes) in the clip
 * kernel's access shape and writes it to dst0 (all of it) and dst1 (a fifth of it); both need src_bytes of room.  reps launches
 * back to back; *ms_out = mean time of one, *mhz_out (may be NULL) = the shader clock held meanwhile, from s_memtime / s_memrealtime
 * stamps around each wave's loop.  scatter != 0: the waves that run at the same time work 5 MB apart, all over the array (as the clip
 * kernel's do: its records run longest first), instead of side by side; 2 / 3: the same two orders with the other load shape (every
 * instruction covering 1 KiB of whole lines instead of 32 contiguous bytes per lane); + 4: the read side alone (no stores), + 8: the
 * write side alone (no loads). */
int rb_dev_box_probe(rb_ctx *ctx, const void *src, uint64_t src_bytes, void *dst0, void *dst1, int reps, int scatter, double *ms_out, double *mhz_out);

/* A buffer that will be WRITTEN at streaming rate (an output arena), placed by measurement: on MI355X the time of a launch that streams
 * its output into a buffer depends on which physical pages the buffer has (same process, same launch, the arena allocated four times
 * over: 9.10 / 9.17 / 9.25 / 11.04 ms; reads do not care).  Allocates up to `tries` candidates of `bytes` with rb_dev_alloc -- fewer
 * when the device lacks the room --, times a store sweep over each, returns the fastest in *out (free it with rb_dev_free) and gives
 * the others back.  sweep_ms (NULL or room for `tries` doubles): the time of each candidate's sweep, -1 where none was made;
 * *kept (may be NULL): the index of the one returned.  tries = 1 is rb_dev_alloc. */
int rb_dev_alloc_placed(rb_ctx *ctx, uint64_t bytes, int tries, void **out, double *sweep_ms, int *kept);
/* The same with the caller's own measure instead of the store sweep: score(candidate, user) is called once per candidate (nothing of
 * the library is locked meanwhile; it may launch on the context and must leave the device idle or its work ordered on the context's
 * stream) and returns a time -- the lowest wins, a negative one ends the search with RB_E_INVALID.  The measure that counts is the
 * launch the buffer is for: the sweep tells a 9.2 ms arena from an 11 ms one, not always a 9.9 ms one from a 10.1 ms one. */
int rb_dev_alloc_placed_by(rb_ctx *ctx, uint64_t bytes, int tries, double (*score)(void *candidate, void *user), void *user, void **out,
                           double *scores, int *kept);

/* ---- synthetic workload generator (bench and tests, not a reference function) -- *
 * Counter-based: ops of record r depend only on (seed, first_record + r, op index).  The host and
 * device versions produce identical bytes.  n_ops per record comes from rb_synth_n_ops. */
uint32_t rb_synth_n_ops(uint64_t seed, uint64_t record, uint32_t lo, uint32_t hi);
/* the "imbalance" shape of SURVEY.md 8(d): log-normal (mu = ln 2000, sigma = 1.35) clipped to [31, 80000] ops, forced odd */
uint32_t rb_synth_n_ops_lognormal(uint64_t seed, uint64_t record);
void rb_synth_fill_ops_host(uint64_t seed, uint64_t first_record, uint64_t n_rec, const uint64_t *op_off, uint32_t *ops);
int rb_dev_synth_fill_ops(rb_ctx *ctx, uint64_t seed, uint64_t first_record, uint64_t n_rec, const uint64_t *op_off_dev,
                          uint32_t *ops_dev);

#ifdef __cplusplus
}
#endif
#endif /* RUSTYBAM_AMD_H */
