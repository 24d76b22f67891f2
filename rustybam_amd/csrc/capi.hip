// capi.hip -- implementation of include/rustybam_amd.h (host side, HIP runtime).
//
// No CPU fallback lives here: every compute entry point needs a gfx950 device and fails with
// RB_E_NO_DEVICE otherwise.  Nothing in this file (or this library) touches oracle/.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <numeric>
#include <string>
#include <thread>
#include <vector>
#include <chrono>
#include <map>
#include <mutex>

#include "../../include/rustybam_amd.h"
#include "rb_lift.h" // rb_lift_params (the device helpers in it are unused here)
#include "rb_trim.h" // rb_trim_params

// ---- kernel-side parameter blocks (must match the .hip files) ----------------------------------
struct rb_scan_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint64_t *t_st, *t_en, *q_st, *q_en;
    const uint8_t *strand;
    rb_reduce_row *reduce_rows;
    rb_norm_row *norm_rows;
    const uint32_t *list;
    const uint64_t *n_list;
};
struct rb_break_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const rb_norm_row *norm;
    const uint32_t *sched;
    uint64_t *hit_off; // indexed by record (break-paf order is record order)
    uint64_t *x_st, *x_en;
    uint64_t rows_cap;
    uint32_t max_size;
    int fill;
    int redo_only;
    void *tmp;
    uint64_t *tmp_off;
    unsigned long long *tmp_cursor;
    uint32_t n_arena;
    uint64_t arena_cap;
    const uint32_t *list;
    const unsigned long long *n_list;
};
// (rb_trim_params: rb_trim.h)
struct rb_swap_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint8_t *strand;
    uint32_t *out_ops;
};

extern "C" hipError_t rb_launch_scan_records(const rb_scan_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_scan_rows(const rb_scan_params *p, void *long_buf, hipStream_t stream);
#ifndef RB_SCAN_ROWS_MEAN_MAX
#define RB_SCAN_ROWS_MEAN_MAX 1536 // ops per record, batch mean, up to which rb_dev_scan_records takes the row form (k_records.hip)
#endif
extern "C" hipError_t rb_launch_peek_norm(const rb_scan_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_count_and_scan(const rb_lift_params *p, uint64_t *block_sums, bool do_count, hipStream_t stream);
struct rb_parse_params {
    uint64_t n_rec;
    const uint8_t *text;
    const uint64_t *text_off;
    const uint64_t *text_end;
    uint64_t *op_off;
    uint32_t *ops;
    uint64_t ops_cap;
    uint8_t *status;
};
struct rb_format_params {
    uint64_t n_items;
    const uint32_t *ops;
    const uint32_t *ops_alt;
    const uint64_t *first;
    const uint32_t *count;
    const uint32_t *first_len;
    const uint32_t *last_len;
    uint64_t *text_off;
    uint8_t *text;
    uint64_t text_cap;
    int plain_ops;
};
extern "C" hipError_t rb_launch_parse_cigars(const rb_parse_params *p, bool fill, hipStream_t stream);
extern "C" hipError_t rb_launch_format_cigars(const rb_format_params *p, bool fill, hipStream_t stream);
extern "C" hipError_t rb_launch_exclusive_scan(uint64_t *v, uint64_t n, uint64_t *block_sums, uint64_t *total_out, hipStream_t stream);
extern "C" hipError_t rb_launch_make_jobs(const rb_lift_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_liftover_stream(const rb_lift_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_liftover_tail(const rb_lift_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_liftover_tiles(const rb_lift_params *p, hipStream_t stream);       // k_tile.hip
extern "C" hipError_t rb_launch_liftover_stream_list(const rb_lift_params *p, hipStream_t stream);
extern "C" uint32_t rb_tile_max_ops(void);
extern "C" uint32_t rb_tile_max_records(void);
#ifndef RB_SHORT_MAX_DEFAULT
#define RB_SHORT_MAX_DEFAULT 2048 // records of up to this many ops go through the tile kernel (profiles/r05_reclen_summary.md)
#endif
extern "C" hipError_t rb_launch_break_gather(const rb_lift_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_break_declined(const rb_lift_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_break_list_declined(const rb_lift_params *p, hipStream_t stream);
extern "C" size_t rb_scan_block_sums_count(uint64_t n_rec);
extern "C" hipError_t rb_launch_break_pieces(const rb_break_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_break_place(const rb_break_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_swap(const rb_swap_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_overlap_split(const rb_trim_params *p, hipStream_t stream);
extern "C" size_t rb_trim_scratch_bytes(uint32_t blocks);
extern "C" hipError_t rb_launch_synth(uint64_t seed, uint64_t first_record, uint64_t n_rec, const uint64_t *op_off, uint32_t *ops, hipStream_t stream);

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
    uint64_t *end_key;
    void *hd;
    uint64_t *tile_off;
    uint64_t *blk;
    uint64_t *tile_lo, *tile_hi;
    uint64_t max_tiles;
    uint64_t *drop_off;
    uint64_t *drop_bits;
    uint64_t drop_words;
    uint32_t *deep_list;
    uint32_t flags;
    void *tdesc;
    uint32_t *wide_list;
};
extern "C" hipError_t rb_launch_nucfreq(const rb_nf_params *p, hipStream_t stream);
struct rb_compact_params {
    uint64_t n_rows;
    rb_hit_row *rows;
    const uint32_t *src;
    uint64_t *off;
    uint32_t *dst;
    int fill;
};
extern "C" hipError_t rb_launch_compact_clips(const rb_compact_params *p, hipStream_t stream);
extern "C" size_t rb_nf_tile_positions(void);
extern "C" size_t rb_nf_scan_blocks(uint64_t n);

#define RB_MAX_ARENA 256

#define RB_TIMING_RING 256
// Filling device memory is done by a kernel of the library (plain 16-byte stores; the runtime's hipMemsetAsync is a kernel too).
// Round 4 wrote it while hunting a fill that "ran" and changed nothing; the cause turned out to be elsewhere -- a virtual range
// given back and taken again, rb_free_vmm -- and the kernel stayed: one code path less that is not the library's own.
__global__ __launch_bounds__(256) void rb_k_fill_bytes(uint8_t *dst, uint32_t v4, size_t n) {
    const uintptr_t a = (uintptr_t)dst;
    const size_t head = n < 16 ? n : (size_t)((16 - (a & 15)) & 15); // bytes in front of the first 16-byte boundary
    const size_t body = (n - head) / 16;                             // whole 16-byte groups
    const size_t tail0 = head + body * 16;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = i; k < body; k += stride) *reinterpret_cast<uint4 *>(dst + head + k * 16) = make_uint4(v4, v4, v4, v4);
    if (i < head) dst[i] = (uint8_t)v4;
    if (i < n - tail0) dst[tail0 + i] = (uint8_t)v4;
}
extern "C" hipError_t rb_fill_async(void *dst, int value, size_t bytes, hipStream_t stream) {
    if (!bytes) return hipSuccess;
    const uint32_t b = (uint32_t)value & 255u, v4 = b | (b << 8) | (b << 16) | (b << 24);
    const size_t groups = bytes / 16 + 1;
    const unsigned blocks = (unsigned)std::min<size_t>((groups + 255) / 256, 65536);
    hipLaunchKernelGGL(rb_k_fill_bytes, dim3(blocks ? blocks : 1), dim3(256), 0, stream, (uint8_t *)dst, v4, bytes);
    return hipGetLastError();
}

struct rb_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // optional HIP-event timing of the dominant (streaming) kernel of each liftover/break call
    bool timing = false;
    std::vector<hipEvent_t> ev_a, ev_b;
    uint64_t timed_calls = 0;
    // trim-paf: slabs of device memory for pairs whose overlap region does not fit LDS (allocated at the first rb_dev_overlap_split)
    void *trim_scratch = nullptr;
    uint32_t trim_scratch_blocks = 0;
    void *trim_pend = nullptr; // [count (256 B) | indices of the pairs the first wave kernel declined]
    uint64_t trim_pend_cap = 0;
    void *scan_list = nullptr; // [count (256 B) | indices of the records the row form of the scan left to the wave-per-record kernel]
    uint64_t scan_list_cap = 0;
    // pinned staging ring of the host-buffer entry points (rb_dev_upload / rb_dev_download): large transfers go through two
    // page-locked chunks (hipHostMalloc), the copy into / out of a chunk on several host threads while the DMA of the other runs
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    bool pin_busy[2] = {false, false};
    unsigned pin_next = 0; // the slot the next upload chunk takes
    uint64_t alloc_fallbacks = 0; // rb_dev_alloc requests that wanted the chunked route and got plain hipMalloc
    // buffers given back with rb_dev_release: still mapped (a chunked buffer keeps its physical pages, hence its placement), handed out
    // again by the next rb_dev_alloc / rb_dev_alloc_placed of the same size on this context
    struct cached_buf { void *p; size_t bytes; bool placed; };
    std::vector<cached_buf> cache;
    size_t cache_bytes = 0;
};
#define RB_PIN_CHUNK ((size_t)32 << 20)
#define RB_PIN_MIN ((size_t)8 << 20) // (round 3: smaller transfers took the runtime's own pageable path; round 4: nothing does, rb_dev_upload says why)

struct rb_plan {
    rb_ctx *ctx = nullptr;
    uint64_t n_rec = 0, n_win = 0, n_ops = 0;
    uint32_t n_contig = 0;
    uint32_t depth = 0; // most windows of a sorted per-contig window list that overlap at one point (0: no sorted list)
    // device arrays
    uint32_t *sched = nullptr, *slot_of = nullptr, *canon_pos = nullptr, *w_orig = nullptr, *ident = nullptr;
    uint64_t *w_st = nullptr, *w_en = nullptr, *wo_st = nullptr, *wo_en = nullptr, *cw_off = nullptr;
    uint8_t *cw_mono = nullptr;
    // short records (k_tile.hip): tiles of consecutive records, three words each {first record | passthrough << 31, records, schedule slot of
    // the first record}; the schedule's slots [0, stream_end) hold the records longer than the tiles take, longest first, the others follow in memory order
    uint32_t *tiles = nullptr;
    uint32_t n_tiles = 0, stream_end = 0;
};

static int fail(rb_ctx *ctx, int code, const char *fmt, ...) {
    if (ctx) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        ctx->err = buf;
    }
    return code;
}
#define HIPCHK(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail((ctx), RB_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

extern "C" int rb_abi_version(void) { return RB_ABI_VERSION; }

extern "C" int rb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int rb_ctx_create(int device, void *hip_stream, rb_ctx **out) {
    if (!out) return RB_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return RB_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return RB_E_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "rustybam_amd: device %d is %s; this library only carries gfx950 (MI355X) code objects\n", device, prop.gcnArchName);
        return RB_E_NO_DEVICE;
    }
    if (hipSetDevice(device) != hipSuccess) return RB_E_NO_DEVICE;
    rb_ctx *c = new rb_ctx();
    c->device = device;
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return RB_E_HIP;
        }
        c->own_stream = true;
    }
    *out = c;
    return RB_OK;
}
extern "C" void rb_ctx_destroy(rb_ctx *ctx) {
    if (!ctx) return;
    for (auto e : ctx->ev_a) hipEventDestroy(e);
    for (auto e : ctx->ev_b) hipEventDestroy(e);
    for (int k = 0; k < 2; k++) {
        if (ctx->pin[k]) hipHostFree(ctx->pin[k]);
        if (ctx->pin_ev[k]) hipEventDestroy(ctx->pin_ev[k]);
    }
    if (ctx->trim_scratch) hipFree(ctx->trim_scratch);
    if (ctx->trim_pend) hipFree(ctx->trim_pend);
    if (ctx->scan_list) hipFree(ctx->scan_list);
    (void)rb_dev_cache_trim(ctx, 0);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}
extern "C" const char *rb_ctx_last_error(const rb_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }
extern "C" int rb_ctx_sync(rb_ctx *ctx) {
    if (!ctx) return RB_E_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RB_OK;
}
extern "C" void *rb_ctx_stream(rb_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int rb_ctx_set_timing(rb_ctx *ctx, int enabled) {
    if (!ctx) return RB_E_INVALID;
    if (enabled && ctx->ev_a.empty()) {
        ctx->ev_a.resize(RB_TIMING_RING);
        ctx->ev_b.resize(RB_TIMING_RING);
        for (int i = 0; i < RB_TIMING_RING; i++) {
            HIPCHK(ctx, hipEventCreate(&ctx->ev_a[i]));
            HIPCHK(ctx, hipEventCreate(&ctx->ev_b[i]));
        }
    }
    ctx->timing = enabled != 0;
    ctx->timed_calls = 0;
    return RB_OK;
}
extern "C" int rb_ctx_get_timing(rb_ctx *ctx, double *ms_out, int cap, int *n_out) {
    if (!ctx || !ms_out || !n_out) return RB_E_INVALID;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t have = ctx->timed_calls < RB_TIMING_RING ? ctx->timed_calls : RB_TIMING_RING;
    int n = 0;
    for (uint64_t k = 0; k < have && n < cap; k++) {
        const uint64_t call = ctx->timed_calls - have + k;
        const size_t slot = (size_t)(call % RB_TIMING_RING);
        float ms = 0;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_a[slot], ctx->ev_b[slot]));
        ms_out[n++] = ms;
    }
    *n_out = n;
    return RB_OK;
}

// A large buffer as separately created 2 MB physical chunks (hipMemCreate) mapped into one virtual range, in the order they were
// created or (RB_ALLOC_MODE=scatter) in a pseudo-random one; rb_dev_free unmaps and releases them (rb_dev_alloc says why).
struct rb_vmm_alloc {
    size_t bytes, chunk;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    bool placed = false; // chosen among candidates by measurement (rb_dev_alloc_placed[_by])
};
static std::map<void *, rb_vmm_alloc> g_vmm;
static std::mutex g_vmm_mu;
// Virtual address space: a freed chunked buffer's range is never given back to the driver (rb_free_vmm says why), so every chunked
// allocation that is FREED costs its size in address space for the life of the process.  Counted, and capped: beyond the cap (default
// 32 TB of the 128 TB a process has, RB_ALLOC_VA_CAP_GB) the chunked route is refused and rb_dev_alloc falls back to plain hipMalloc
// (rb_dev_alloc_mode / rb_dev_alloc_stats say so).  A host that recycles its buffers with rb_dev_release instead of rb_dev_free retires
// nothing.
static size_t g_va_live = 0, g_va_retired = 0; // (under g_vmm_mu)
static std::map<void *, size_t> g_asked;        // (under g_vmm_mu) what rb_dev_alloc was asked for, rounded to 256: buffers of 1 MB and more (the ones rb_dev_release keeps)
static size_t rb_va_cap() { // (read per call: tests move it)
    const char *e = getenv("RB_ALLOC_VA_CAP_GB");
    return (size_t)(e ? strtoull(e, nullptr, 10) : 32768ull) << 30;
}
// -> 0: not one of ours; 1: released; -1: a HIP call failed (*err names it; the pieces that could be given back have been)
static int rb_free_vmm(void *p, int device, hipError_t *err) {
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    auto it = g_vmm.find(p);
    if (it == g_vmm.end()) return 0;
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t e) {
        if (e != hipSuccess && first == hipSuccess) first = e;
    };
    note(hipSetDevice(device));       // (the buffer's device, not whatever is current: kernels on it must have drained)
    note(hipDeviceSynchronize());
    // every chunk is a mapping of its own: HIP promises the unmapping of whole mappings, not of a range that spans several
    const size_t chunk = it->second.chunk, n = it->second.handles.size();
    for (size_t k = 0; k < n; k++) note(hipMemUnmap((char *)p + k * chunk, chunk));
    for (auto &h : it->second.handles) note(hipMemRelease(h));
    // The virtual range is NOT given back (hipMemAddressFree): a later reservation would get it again, and a kernel's stores into a
    // buffer mapped at addresses this process had mapped before went to the OLD physical pages -- a fill kernel that "ran" and changed
    // nothing that three different readers could see (tools/alloc_probe2.py: a 300 MB buffer taken right after a 1 GB one was freed;
    // copies, which do not go through the compute units' translation caches, read and wrote the new pages).  Addresses are not
    // scarce (a batch of 75 GB takes 2^-11 of the 47-bit range); memory is what is returned, chunk by chunk, above.
    g_va_live -= it->second.bytes;
    if (getenv("RB_ALLOC_FREE_VA")) note(hipMemAddressFree(p, it->second.bytes)); // (the probe's switch)
    else g_va_retired += it->second.bytes;
    g_vmm.erase(it);
    if (first != hipSuccess) (void)hipGetLastError();
    *err = first;
    return first == hipSuccess ? 1 : -1;
}
static void *rb_alloc_vmm(int device, size_t bytes, bool shuffle) {
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || !gran) return nullptr;
    // (2 MB: chunks of 4 MB measured 2 % slower, and with 8 MB and 32 MB chunks the first kernel on the buffer took a memory access
    //  fault on this ROCm -- not understood, not used)
    size_t chunk = (size_t)2 << 20;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        if (g_va_live + g_va_retired + n * chunk > rb_va_cap()) return nullptr; // (address space used up by freed buffers: the caller takes plain hipMalloc)
    }
    void *va = nullptr;
    if (hipMemAddressReserve(&va, n * chunk, 0, nullptr, 0) != hipSuccess || !va) return nullptr;
    std::vector<hipMemGenericAllocationHandle_t> h;
    h.reserve(n);
    size_t mapped = 0;
    bool ok = true;
    for (size_t k = 0; k < n && ok; k++) {
        hipMemGenericAllocationHandle_t hk;
        ok = hipMemCreate(&hk, chunk, &prop, 0) == hipSuccess;
        if (ok) h.push_back(hk);
    }
    std::vector<size_t> order(n);
    for (size_t k = 0; k < n; k++) order[k] = k;
    if (shuffle) {
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (size_t k = n; k > 1; k--) {
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            std::swap(order[k - 1], order[(size_t)(x % k)]);
        }
    }
    for (size_t k = 0; k < n && ok; k++) {
        ok = hipMemMap((char *)va + k * chunk, chunk, 0, h[order[k]], 0) == hipSuccess;
        if (ok) mapped = k + 1;
    }
    if (ok) {
        hipMemAccessDesc acc;
        memset(&acc, 0, sizeof acc);
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        ok = hipMemSetAccess(va, n * chunk, &acc, 1) == hipSuccess;
    }
    if (!ok) { // give everything back: the caller falls back to hipMalloc
        (void)hipGetLastError();
        for (size_t k = 0; k < mapped; k++) (void)hipMemUnmap((char *)va + k * chunk, chunk);
        for (auto &hk : h) (void)hipMemRelease(hk);
        (void)hipMemAddressFree(va, n * chunk);
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        g_vmm[va] = rb_vmm_alloc{n * chunk, chunk, h};
        g_va_live += n * chunk;
    }
    return va;
}
extern "C" int rb_dev_alloc(rb_ctx *ctx, size_t bytes, void **dev_ptr) {
    if (!ctx || !dev_ptr) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t want = bytes ? ((bytes + 255) & ~(size_t)255) : 256;
    // How a large buffer is pieced together physically decides the streaming kernels' time: the same clip kernel on the same virtual
    // addresses takes 10.2-11.9 ms in plain hipMalloc memory depending on the allocation (tools/layout_probe.py), 18-20 ms in
    // physically contiguous memory (hipDeviceMallocContiguous), and 9.4 ms -- every time -- when the buffer is made of separately
    // created 2 MB physical chunks mapped side by side (hipMemCreate / hipMemMap; tools/contig_probe.py, round 3).  So requests of
    // 1 GB and more -- a resident batch and its outputs -- are built that way (about 14 us per chunk: 1 s for 75 GB, once per batch);
    // smaller ones, and everything under RB_ALLOC_MODE=default, come from hipMalloc.  RB_ALLOC_MODE=chunks / scatter (the chunks in
    // pseudo-random order: no different) / contiguous force a mode for the probe.
    for (size_t k = 0; k < ctx->cache.size(); k++) // a buffer this context released, of this very size: as it is (still mapped, its pages its own)
        if (ctx->cache[k].bytes == want) {
            *dev_ptr = ctx->cache[k].p;
            ctx->cache_bytes -= want;
            ctx->cache.erase(ctx->cache.begin() + (long)k);
            return RB_OK;
        }
    const char *mode = getenv("RB_ALLOC_MODE");
    // (round 4: from 256 MB up, not 1 GB -- the row arena of the headline batch, 1.02e9 bytes, fell just short of the old threshold and sat
    //  in plain hipMalloc memory beside a chunked batch; RB_ALLOC_CHUNK_MIN_MB moves the threshold for experiments)
    static const size_t chunk_min = (size_t)(getenv("RB_ALLOC_CHUNK_MIN_MB") ? atol(getenv("RB_ALLOC_CHUNK_MIN_MB")) : 256) << 20;
    const bool chunks = mode ? (!strcmp(mode, "scatter") || !strcmp(mode, "chunks")) : want >= chunk_min;
    if (want >= ((size_t)64 << 20) && chunks) {
        void *q = rb_alloc_vmm(ctx->device, want, mode && !strcmp(mode, "scatter"));
        if (q) {
            *dev_ptr = q;
            std::lock_guard<std::mutex> lk(g_vmm_mu);
            g_asked[q] = want;
            return RB_OK;
        }
        const hipError_t why = hipGetLastError();
        ctx->alloc_fallbacks++; // (rb_dev_alloc_mode tells the caller which route a buffer took: the two differ by 10-20 % in the clip kernel's time)
        if (getenv("RB_ALLOC_LOG")) fprintf(stderr, "[rb_dev_alloc] %zu bytes: the chunked route failed (%s), plain hipMalloc instead\n", want, hipGetErrorString(why));
    }
    auto remember = [&]() {
        if (want >= ((size_t)1 << 20)) {
            std::lock_guard<std::mutex> lk(g_vmm_mu);
            g_asked[*dev_ptr] = want;
        }
    };
    if (want >= ((size_t)64 << 20) && mode && !strcmp(mode, "contiguous")) {
        if (hipExtMallocWithFlags(dev_ptr, want, hipDeviceMallocContiguous) == hipSuccess) {
            remember();
            return RB_OK;
        }
        (void)hipGetLastError();
    }
    hipError_t e = hipMalloc(dev_ptr, want);
    if (e != hipSuccess) {
        (void)hipGetLastError(); // reported through the return code: not left behind for the caller's next HIP call to trip over
        return fail(ctx, RB_E_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    }
    remember();
    return RB_OK;
}
// which route a buffer of rb_dev_alloc took: 1 = separately created 2 MB physical chunks, 0 = anything else (plain hipMalloc: a small
// request, a forced mode, the fallback when the chunked route failed or the address-space cap was reached; or a foreign pointer)
extern "C" int rb_dev_alloc_mode(rb_ctx *ctx, const void *dev_ptr) {
    if (!ctx || !dev_ptr) return RB_E_INVALID;
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    return g_vmm.count(const_cast<void *>(dev_ptr)) ? 1 : 0;
}
static size_t rb_buf_bytes(rb_ctx *, void *p, bool *placed) {
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    auto it = g_asked.find(p);
    if (it == g_asked.end()) return 0;
    auto iv = g_vmm.find(p);
    *placed = iv != g_vmm.end() && iv->second.placed;
    return it->second;
}
extern "C" int rb_dev_free(rb_ctx *ctx, void *dev_ptr) {
    if (!ctx) return RB_E_INVALID;
    if (!dev_ptr) return RB_OK;
    {
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        g_asked.erase(dev_ptr);
    }
    hipError_t e = hipSuccess;
    const int mine = rb_free_vmm(dev_ptr, ctx->device, &e);
    if (mine < 0) return fail(ctx, RB_E_HIP, "rb_dev_free (chunked buffer): %s", hipGetErrorString(e));
    if (mine > 0) return RB_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipFree(dev_ptr));
    return RB_OK;
}
// rb_dev_release: the buffer goes back to the CONTEXT, not to the device -- it stays mapped, a chunked buffer keeps its physical pages
// (and with them what rb_dev_alloc_placed chose them for), and the next rb_dev_alloc / rb_dev_alloc_placed of exactly this size on this
// context returns it.  No address space is retired and nothing is unmapped, so a host that allocates and releases the same shapes batch
// after batch runs in constant address space and pays the 14 us per chunk once.  The cache holds at most RB_ALLOC_CACHE_GB (default
// 96) gigabytes: what does not fit is freed (rb_dev_free).  The caller's work on the buffer must have been enqueued on the context's
// stream (it is synchronised here) or be finished.
static size_t rb_cache_cap() {
    static const size_t v = [] { const char *e = getenv("RB_ALLOC_CACHE_GB"); return (size_t)(e ? strtoull(e, nullptr, 10) : 96ull) << 30; }();
    return v;
}
extern "C" int rb_dev_release(rb_ctx *ctx, void *dev_ptr) {
    if (!ctx) return RB_E_INVALID;
    if (!dev_ptr) return RB_OK;
    bool placed = false;
    const size_t bytes = rb_buf_bytes(ctx, dev_ptr, &placed);
    if (!bytes || bytes > rb_cache_cap()) return rb_dev_free(ctx, dev_ptr);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    while (!ctx->cache.empty() && ctx->cache_bytes + bytes > rb_cache_cap()) { // the oldest go first
        void *q = ctx->cache.front().p;
        ctx->cache_bytes -= ctx->cache.front().bytes;
        ctx->cache.erase(ctx->cache.begin());
        const int rc = rb_dev_free(ctx, q);
        if (rc) return rc;
    }
    ctx->cache.push_back({dev_ptr, bytes, placed});
    ctx->cache_bytes += bytes;
    return RB_OK;
}
// frees cached buffers, oldest first, until at most keep_bytes are held
extern "C" int rb_dev_cache_trim(rb_ctx *ctx, uint64_t keep_bytes) {
    if (!ctx) return RB_E_INVALID;
    while (!ctx->cache.empty() && ctx->cache_bytes > keep_bytes) {
        void *q = ctx->cache.front().p;
        ctx->cache_bytes -= ctx->cache.front().bytes;
        ctx->cache.erase(ctx->cache.begin());
        const int rc = rb_dev_free(ctx, q);
        if (rc) return rc;
    }
    return RB_OK;
}
// out[0] = bytes of live chunked buffers (this process), out[1] = bytes this context holds for reuse (rb_dev_release), out[2] = address
// space retired by freed chunked buffers (this process; never returned: rb_free_vmm), out[3] = the cap on out[0] + out[2],
// out[4] = rb_dev_alloc requests of this context that wanted the chunked route and got plain hipMalloc
extern "C" int rb_dev_alloc_stats(rb_ctx *ctx, uint64_t out[5]) {
    if (!ctx || !out) return RB_E_INVALID;
    std::lock_guard<std::mutex> lk(g_vmm_mu);
    out[0] = g_va_live, out[1] = ctx->cache_bytes, out[2] = g_va_retired, out[3] = rb_va_cap(), out[4] = ctx->alloc_fallbacks;
    return RB_OK;
}
static bool pin_ready(rb_ctx *ctx) {
    if (ctx->pin[0]) return true;
    static const bool off = getenv("RB_NO_PINNED") != nullptr; // diagnostics: the runtime's pageable path for everything
    if (off) return false;
    for (int k = 0; k < 2; k++) {
        if (hipHostMalloc(&ctx->pin[k], RB_PIN_CHUNK, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&ctx->pin_ev[k], hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            for (int j = 0; j <= k; j++) {
                if (ctx->pin[j]) hipHostFree(ctx->pin[j]);
                if (ctx->pin_ev[j]) hipEventDestroy(ctx->pin_ev[j]);
                ctx->pin[j] = nullptr, ctx->pin_ev[j] = nullptr;
            }
            return false;
        }
    }
    return true;
}
static void par_memcpy(void *dst, const void *src, size_t n) { // a chunk of the staging ring, on a few host threads
    const unsigned T = n >= ((size_t)16 << 20) ? 8u : (n >= ((size_t)4 << 20) ? 4u : 1u); // (a thread copies 6-8 GB/s; the DMA behind it does 50+)
    if (T == 1) {
        memcpy(dst, src, n);
        return;
    }
    std::vector<std::thread> th;
    const size_t part = ((n + T - 1) / T + 63) & ~(size_t)63; // (rounded UP before the alignment: n / T rounded down left the last n % T bytes of a chunk uncopied
                                                              //  whenever n / T was a multiple of 64 -- found by the md5 of a 19 GB output, round 3)
    for (unsigned t = 1; t < T; t++) {
        const size_t lo = std::min(n, t * part), hi = std::min(n, (t + 1) * part);
        if (hi > lo) th.emplace_back([=]() { memcpy((char *)dst + lo, (const char *)src + lo, hi - lo); });
    }
    memcpy(dst, src, std::min(n, part));
    for (auto &x : th) x.join();
}
extern "C" int rb_dev_upload(rb_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) {
    if (!ctx) return RB_E_INVALID;
    if (!bytes) return RB_OK;
    // Every transfer goes through the context's own page-locked chunks (round 4; small ones used to be handed to the runtime as
    // pageable memory): one path, and the caller's buffer is never pinned behind its back.
    if (!pin_ready(ctx)) { // (no page-locked memory to be had: the runtime's path, synchronously)
        HIPCHK(ctx, hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
        return RB_OK;
    }
    // host_src is consumed chunk by chunk into page-locked memory; when the call returns the caller may reuse it, the DMAs are
    // queued on the context's stream
    // (the slot cursor runs on ACROSS calls: two small uploads in a row take different slots, so the second does not wait for the DMA --
    //  and whatever kernel was queued in front of it -- of the first)
    for (size_t off = 0; off < bytes; off += RB_PIN_CHUNK) {
        const int slot = (int)(ctx->pin_next++ & 1u);
        const size_t n = std::min(RB_PIN_CHUNK, bytes - off);
        if (ctx->pin_busy[slot]) HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[slot]));
        par_memcpy(ctx->pin[slot], (const char *)host_src + off, n);
        HIPCHK(ctx, hipMemcpyAsync((char *)dev_dst + off, ctx->pin[slot], n, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->pin_ev[slot], ctx->stream));
        ctx->pin_busy[slot] = true;
    }
    return RB_OK;
}
extern "C" int rb_dev_download(rb_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
    if (!ctx) return RB_E_INVALID;
    if (!bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return RB_OK;
    }
    if (!pin_ready(ctx)) { // (see rb_dev_upload)
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost));
        return RB_OK;
    }
    // chunk k + 1 is in flight into one page-locked slot while chunk k leaves the other for host_dst
    size_t prev_off = 0, prev_n = 0;
    int prev_slot = -1;
    for (size_t off = 0, k = 0; off < bytes; off += RB_PIN_CHUNK, k++) {
        const int slot = (int)(k & 1);
        const size_t n = std::min(RB_PIN_CHUNK, bytes - off);
        if (ctx->pin_busy[slot]) HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[slot])); // (an upload that used the slot)
        HIPCHK(ctx, hipMemcpyAsync(ctx->pin[slot], (const char *)dev_src + off, n, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipEventRecord(ctx->pin_ev[slot], ctx->stream));
        ctx->pin_busy[slot] = true;
        if (prev_slot >= 0) {
            HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[prev_slot]));
            par_memcpy((char *)host_dst + prev_off, ctx->pin[prev_slot], prev_n);
            ctx->pin_busy[prev_slot] = false;
        }
        prev_off = off, prev_n = n, prev_slot = slot;
    }
    if (prev_slot >= 0) {
        HIPCHK(ctx, hipEventSynchronize(ctx->pin_ev[prev_slot]));
        par_memcpy((char *)host_dst + prev_off, ctx->pin[prev_slot], prev_n);
        ctx->pin_busy[prev_slot] = false;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return RB_OK;
}
extern "C" int rb_dev_memset(rb_ctx *ctx, void *dev_dst, int value, size_t bytes) {
    if (!ctx) return RB_E_INVALID;
    if (bytes) HIPCHK(ctx, rb_fill_async(dev_dst, value, bytes, ctx->stream));
    return RB_OK;
}

// ---- K1 ----------------------------------------------------------------------------------------
extern "C" int rb_dev_scan_records(rb_ctx *ctx, const rb_batch_view *b, rb_reduce_row *reduce_rows, rb_norm_row *norm_rows) {
    if (!ctx || !b) return RB_E_INVALID;
    if (((uintptr_t)b->ops & 15u) != 0) return fail(ctx, RB_E_INVALID, "ops must be 16-byte aligned");
    rb_scan_params p;
    p.n_rec = b->n_rec;
    p.ops = b->ops;
    p.op_off = b->op_off;
    p.t_st = b->t_st;
    p.t_en = b->t_en;
    p.q_st = b->q_st;
    p.q_en = b->q_en;
    p.strand = b->strand;
    p.reduce_rows = reduce_rows;
    p.norm_rows = norm_rows;
    p.list = nullptr;
    p.n_list = nullptr;
    // a batch of short records (config 4's shape: a few hundred ops each) goes through the row form, four records per wavefront, and
    // only what that lists through the wave-per-record kernel; RB_SCAN_ROWS=0 (diagnostics): the wave-per-record kernel for everything
    static const bool rows_off = getenv("RB_SCAN_ROWS") && atoi(getenv("RB_SCAN_ROWS")) == 0;
    if (!rows_off && b->n_rec >= 64 && b->n_ops / b->n_rec <= RB_SCAN_ROWS_MEAN_MAX) {
        if (ctx->scan_list_cap < b->n_rec) { // (grows with the largest batch seen: four bytes per record)
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->scan_list) hipFree(ctx->scan_list);
            ctx->scan_list = nullptr, ctx->scan_list_cap = 0;
            const uint64_t cap = b->n_rec + b->n_rec / 4 + 1024;
            if (hipMalloc(&ctx->scan_list, 256 + cap * 4) == hipSuccess) ctx->scan_list_cap = cap;
            else (void)hipGetLastError();
        }
        if (ctx->scan_list) {
            HIPCHK(ctx, rb_fill_async(ctx->scan_list, 0, 8, ctx->stream));
            HIPCHK(ctx, rb_launch_scan_rows(&p, ctx->scan_list, ctx->stream));
            return RB_OK;
        }
    }
    HIPCHK(ctx, rb_launch_scan_records(&p, ctx->stream));
    return RB_OK;
}

// ---- plan ----------------------------------------------------------------------------------------
template <typename T>
static int upload_vec(rb_ctx *ctx, const std::vector<T> &v, T **dev) {
    int rc = rb_dev_alloc(ctx, v.size() * sizeof(T) + 16, (void **)dev);
    if (rc) return rc;
    if (!v.empty()) { // (through the context's page-locked chunks, like every host transfer of the library: rb_dev_upload says why)
        rc = rb_dev_upload(ctx, *dev, v.data(), v.size() * sizeof(T)); // (the host vector is consumed when this returns; the DMA is queued)
        if (rc) return rc;
    }
    return RB_OK;
}

// Tiles of short records (k_tile.hip): runs of consecutive records of 8 .. short_max ops, at most rb_tile_max_records() of them and
// rb_tile_max_ops() ops together, three words a tile {first record | pass-through << 31, records, schedule slot of the first record}; runs of
// records below 8 ops become pass-through tiles (the per-record kernel takes them: a lane's eight ops would hold three records).  `sched` comes
// in as the longest-first order of all records and leaves as: the records longer than short_max, longest first (*n_long of them: the
// per-record kernel's launch), then the others in MEMORY order -- their slots are never launched as waves of their own, they are where the
// records' jobs lie, so a tile's jobs lie side by side (one coalesced read instead of a gather, and rb_k_make_jobs writes them side by side).
// No tile worth the name (nothing of 8 ops and more below the line): tiles stays empty and sched as it was.
static void rb_cut_tiles(uint64_t n_rec, const uint64_t *op_off, uint64_t short_max, std::vector<uint32_t> &sched, std::vector<uint32_t> &tiles,
                         uint64_t *n_long_out) {
    tiles.clear();
    *n_long_out = 0;
    short_max = std::min<uint64_t>(short_max, rb_tile_max_ops());
    if (short_max < 8 || !n_rec) return;
    const uint64_t max_ops = rb_tile_max_ops(), max_rec = rb_tile_max_records();
    std::vector<uint32_t> t2;
    uint64_t first = 0, cnt = 0, ops = 0, n_long = 0, n_tiled = 0;
    bool tiny = false;
    auto close = [&]() {
        if (cnt) t2.push_back((uint32_t)first | (tiny ? 0x80000000u : 0u)), t2.push_back((uint32_t)cnt), n_tiled += tiny ? 0 : cnt;
        cnt = 0, ops = 0;
    };
    for (uint64_t r = 0; r < n_rec; r++) {
        const uint64_t n = op_off[r + 1] - op_off[r];
        if (n > short_max) {
            close();
            n_long++;
            continue;
        }
        const bool t = n < 8;
        if (cnt && (t != tiny || cnt >= max_rec || (!t && ops + n > max_ops))) close();
        if (!cnt) first = r, tiny = t;
        cnt++, ops += n;
    }
    close();
    if (n_tiled == 0) return;
    std::vector<uint32_t> s2(n_rec);
    uint64_t a = 0, b = n_long;
    for (uint64_t w = 0; w < n_rec; w++)
        if (op_off[(uint64_t)sched[w] + 1] - op_off[sched[w]] > short_max) s2[a++] = sched[w];
    std::vector<uint32_t> slot_short(n_rec, 0);
    for (uint64_t r = 0; r < n_rec; r++)
        if (op_off[r + 1] - op_off[r] <= short_max) slot_short[r] = (uint32_t)b, s2[b++] = (uint32_t)r;
    sched.swap(s2);
    tiles.reserve(t2.size() / 2 * 3);
    for (size_t t = 0; t + 1 < t2.size(); t += 2) tiles.push_back(t2[t]), tiles.push_back(t2[t + 1]), tiles.push_back(slot_short[t2[t] & 0x7FFFFFFFu]);
    *n_long_out = n_long;
}
// the same on plain host arrays, no device needed (tests/test_plan_tiles.py; a host that wants to know how a batch will be cut):
// sched_out [n_rec], tiles_out [3 * tiles_cap]; returns the number of tiles (tiles beyond tiles_cap are counted, not written), -1: bad arguments
extern "C" int64_t rb_plan_tiles_host(uint64_t n_rec, const uint64_t *op_off, uint64_t short_max, uint32_t *sched_out, uint32_t *tiles_out,
                                      uint64_t tiles_cap, uint64_t *n_long_out) {
    if (n_rec && (!op_off || !sched_out)) return -1;
    if (n_rec >= 0xFFFFFFFFull) return -1;
    std::vector<uint32_t> sched(n_rec), tiles;
    std::iota(sched.begin(), sched.end(), 0u);
    std::stable_sort(sched.begin(), sched.end(), [&](uint32_t x, uint32_t y) { return op_off[(uint64_t)x + 1] - op_off[x] > op_off[(uint64_t)y + 1] - op_off[y]; }); // (what rb_plan_create's radix sort gives)
    uint64_t n_long = 0;
    rb_cut_tiles(n_rec, op_off, short_max ? short_max : RB_SHORT_MAX_DEFAULT, sched, tiles, &n_long);
    for (uint64_t i = 0; i < n_rec; i++) sched_out[i] = sched[i];
    const uint64_t nt = tiles.size() / 3;
    for (uint64_t t = 0; t < nt && t < tiles_cap && tiles_out; t++)
        for (int k = 0; k < 3; k++) tiles_out[3 * t + k] = tiles[3 * t + k];
    if (n_long_out) *n_long_out = tiles.empty() ? n_rec : n_long;
    return (int64_t)nt;
}

extern "C" int rb_plan_create(rb_ctx *ctx, uint64_t n_rec, const uint64_t *op_off, const uint32_t *contig, uint64_t n_win,
                              const uint32_t *w_contig, const uint64_t *w_st, const uint64_t *w_en, rb_plan **out) {
    if (!ctx || !out || (n_rec && (!op_off || !contig))) return RB_E_INVALID;
    if (n_rec >= 0xFFFFFFFFull || n_win >= 0xFFFFFFFFull) return fail(ctx, RB_E_INVALID, "too many records/windows for one batch");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rb_plan *pl = new rb_plan();
    pl->ctx = ctx;
    pl->n_rec = n_rec;
    pl->n_win = n_win;
    uint32_t maxc = 0;
    for (uint64_t i = 0; i < n_rec; i++) maxc = std::max(maxc, contig[i]);
    for (uint64_t i = 0; i < n_win; i++) maxc = std::max(maxc, w_contig[i]);
    const uint32_t n_contig = (n_rec || n_win) ? maxc + 1 : 0;
    pl->n_contig = n_contig;
    // canonical order: contigs by first appearance among the records (liftover.rs:151), then record order
    std::vector<uint32_t> canon_pos(n_rec);
    {
        std::vector<uint64_t> rank(n_contig, UINT64_MAX);
        uint64_t nr = 0;
        for (uint64_t i = 0; i < n_rec; i++)
            if (rank[contig[i]] == UINT64_MAX) rank[contig[i]] = nr++;
        std::vector<uint64_t> start(nr + 1, 0);
        for (uint64_t i = 0; i < n_rec; i++) start[rank[contig[i]] + 1]++;
        for (uint64_t k = 0; k < nr; k++) start[k + 1] += start[k];
        for (uint64_t i = 0; i < n_rec; i++) canon_pos[i] = (uint32_t)start[rank[contig[i]]]++;
    }
    // launch order: longest record first (pure load balancing)
    std::vector<uint32_t> sched(n_rec), ident(n_rec);
    std::iota(ident.begin(), ident.end(), 0u);
    {   // stable, descending by op count: an LSD radix sort on the complemented count, 16 bits a pass (a comparison sort of 1e6
        // records with its random reads of op_off took 90 ms of the host's time per plan; this takes a tenth)
        uint64_t max_len = 0;
        std::vector<uint64_t> key(n_rec);
        for (uint64_t i = 0; i < n_rec; i++) key[i] = op_off[i + 1] - op_off[i], max_len = std::max(max_len, key[i]);
        int passes = 0;
        while (passes < 4 && (max_len >> (16 * passes)) != 0) passes++;
        std::vector<uint32_t> a(ident), b(n_rec);
        std::vector<uint64_t> cnt(65537);
        for (int ps = 0; ps < passes; ps++) {
            std::fill(cnt.begin(), cnt.end(), 0);
            const int sh = 16 * ps;
            for (uint64_t i = 0; i < n_rec; i++) cnt[(0xFFFFu - (uint32_t)((key[i] >> sh) & 0xFFFFu)) + 1]++;
            for (int k = 0; k < 65536; k++) cnt[k + 1] += cnt[k];
            for (uint64_t i = 0; i < n_rec; i++) {
                const uint32_t r = a[i];
                b[cnt[0xFFFFu - (uint32_t)((key[r] >> sh) & 0xFFFFu)]++] = r;
            }
            a.swap(b);
        }
        sched = a;
        // RB_SCHED=chunk:<N> (experiments): records in MEMORY order by chunks of N, longest first inside a chunk.  The waves that run at
        // the same time then work on neighbouring records -- a few hundred MB of the ops array and of each output slot -- instead of
        // on records scattered over all of them, and the four waves of a workgroup still get records of one length.
        if (const char *e = getenv("RB_SCHED")) {
            uint64_t N = 0;
            if (!strncmp(e, "chunk:", 6)) N = strtoull(e + 6, nullptr, 10);
            else if (!strcmp(e, "memory")) N = 1;
            if (N >= 1) {
                std::iota(sched.begin(), sched.end(), 0u);
                if (N > 1)
                    for (uint64_t c0 = 0; c0 < n_rec; c0 += N) {
                        const uint64_t c1 = std::min<uint64_t>(n_rec, c0 + N);
                        std::stable_sort(sched.begin() + c0, sched.begin() + c1, [&](uint32_t x, uint32_t y) { return key[x] > key[y]; });
                    }
            }
        }
    }
    // windows grouped by contig, BED order kept inside a contig
    std::vector<uint64_t> cw_off(n_contig + 1, 0), g_st(n_win), g_en(n_win), o_st(n_win), o_en(n_win);
    std::vector<uint32_t> g_orig(n_win);
    std::vector<uint8_t> mono(n_contig ? n_contig : 1, 1);
    for (uint64_t i = 0; i < n_win; i++) cw_off[w_contig[i] + 1]++;
    for (uint32_t c = 0; c < n_contig; c++) cw_off[c + 1] += cw_off[c];
    {
        std::vector<uint64_t> cur(cw_off.begin(), cw_off.end() - (n_contig ? 1 : 0));
        for (uint64_t i = 0; i < n_win; i++) {
            const uint64_t d = cur[w_contig[i]]++;
            g_st[d] = w_st[i];
            g_en[d] = w_en[i];
            g_orig[d] = (uint32_t)i;
            o_st[i] = w_st[i];
            o_en[i] = w_en[i];
        }
    }
    for (uint32_t c = 0; c < n_contig; c++)
        for (uint64_t i = cw_off[c] + 1; i < cw_off[c + 1]; i++)
            if (g_st[i] < g_st[i - 1] || g_en[i] < g_en[i - 1]) mono[c] = 0;
    // how deep the sorted window lists overlap: clip j of a record goes to output slot j mod depth (k_liftover.hip), so that
    // clips sharing a slot follow one another along the record; en == st counts as overlapping (both clips may cut one op)
    pl->n_ops = n_rec ? op_off[n_rec] : 0;
    for (uint32_t c = 0; c < n_contig; c++) {
        if (!mono[c]) continue;
        uint64_t lo = cw_off[c];
        for (uint64_t i = cw_off[c]; i < cw_off[c + 1]; i++) {
            while (lo < i && g_en[lo] < g_st[i]) lo++; // (a window with en < st may not push lo past i)
            pl->depth = std::max<uint32_t>(pl->depth, (uint32_t)std::min<uint64_t>(i - lo + 1, 1u << 20));
        }
    }
    // tiles of short records (k_tile.hip): rb_plan_tiles_host below.  RB_TILE=0 switches the tile kernel off, RB_SHORT_MAX=<ops> moves the
    // line between the two kernels (experiments; RB_SCHED -- a schedule that is not sorted by length -- switches it off as well).
    std::vector<uint32_t> tiles;
    pl->stream_end = (uint32_t)n_rec;
    {
        const char *e = getenv("RB_TILE");
        const bool on = !(e && !strcmp(e, "0")) && !getenv("RB_SCHED");
        uint64_t short_max = RB_SHORT_MAX_DEFAULT;
        if (const char *m = getenv("RB_SHORT_MAX")) short_max = strtoull(m, nullptr, 10);
        if (on && n_rec) {
            uint64_t n_long = 0;
            rb_cut_tiles(n_rec, op_off, short_max, sched, tiles, &n_long);
            if (!tiles.empty()) pl->stream_end = (uint32_t)n_long;
        }
        pl->n_tiles = (uint32_t)(tiles.size() / 3);
    }
    int rc = RB_OK;
    std::vector<uint32_t> slot_of(n_rec);
    for (uint64_t w = 0; w < n_rec; w++) slot_of[sched[w]] = (uint32_t)w;
    if (!rc) rc = upload_vec(ctx, sched, &pl->sched);
    if (!rc) rc = upload_vec(ctx, slot_of, &pl->slot_of);
    if (!rc) rc = upload_vec(ctx, ident, &pl->ident);
    if (!rc) rc = upload_vec(ctx, canon_pos, &pl->canon_pos);
    if (!rc) rc = upload_vec(ctx, g_st, &pl->w_st);
    if (!rc) rc = upload_vec(ctx, g_en, &pl->w_en);
    if (!rc) rc = upload_vec(ctx, g_orig, &pl->w_orig);
    if (!rc) rc = upload_vec(ctx, o_st, &pl->wo_st);
    if (!rc) rc = upload_vec(ctx, o_en, &pl->wo_en);
    if (!rc) rc = upload_vec(ctx, cw_off, &pl->cw_off);
    if (!rc) rc = upload_vec(ctx, mono, &pl->cw_mono);
    if (!rc && pl->n_tiles) rc = upload_vec(ctx, tiles, &pl->tiles);
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(ctx, RB_E_HIP, "plan upload: %s", hipGetErrorString(hipGetLastError())); // (once, for all of them)
    if (rc) {
        rb_plan_destroy(pl);
        return rc;
    }
    *out = pl;
    return RB_OK;
}
extern "C" void rb_plan_destroy(rb_plan *pl) {
    if (!pl) return;
    void *ptrs[] = {pl->sched, pl->slot_of, pl->ident, pl->canon_pos, pl->w_st, pl->w_en, pl->w_orig, pl->wo_st, pl->wo_en, pl->cw_off, pl->cw_mono, pl->tiles};
    for (void *q : ptrs)
        if (q) (void)rb_dev_free(pl->ctx, q); // (rb_dev_alloc memory: a large array is a chunked buffer, which hipFree does not know)
    delete pl;
}

// workspace layout: [hit_off (n_rec+1) u64][win_lo][block sums][arena cursors][jobs n_rec x 64 B][gen_list rows_cap u32][x_st rows_cap u64][x_en rows_cap u64]
struct ws_layout {
    size_t hit_off, win_lo, block_sums, arena, pend_count, pend_list, jobs, gen_list, x_st, x_en, bp_tmp, bp_off, bp_cur, brk_rows, copy_count, copy_list, decl_count, decl_list, gen_cp, diag_stamps, fb_count, fb_list, total;
};
// RB_DEBUG_NO_GEN_CP (diagnostics: the generic kernel walks every record from its first op, no room for checkpoints): read ONCE per
// process -- the workspace layout and the kernel parameter must agree on it
static bool rb_no_gen_cp() {
    static const bool v = getenv("RB_DEBUG_NO_GEN_CP") != nullptr;
    return v;
}
static ws_layout ws_of(uint64_t n_rec, uint64_t rows_cap, uint64_t n_ops) {
    ws_layout w;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o += (bytes + 255) & ~(size_t)255;
        return at;
    };
    w.hit_off = take((n_rec + 2) * 8);
    w.win_lo = take((n_rec + 2) * 4);
    w.block_sums = take(rb_scan_block_sums_count(n_rec) * 8);
    w.arena = take((size_t)RB_MAX_ARENA * RB_ARENA_STRIDE * 8);
    w.pend_count = take(256);
    w.pend_list = take((n_rec + 1) * 4);
    w.jobs = take((n_rec + 1) * sizeof(rb_job));
    w.gen_list = take((rows_cap + 1) * 4);
    w.x_st = take((rows_cap + 1) * 8);
    w.x_en = take((rows_cap + 1) * 8);
    w.bp_tmp = take((rows_cap + 1) * 8); // break-paf: piece windows between the collect pass and their rows
    w.bp_off = take((n_rec + 1) * 8);
    w.bp_cur = take((size_t)RB_MAX_ARENA * 128);
    w.brk_rows = take((rows_cap + 1) * sizeof(rb_hit_row)); // break-paf in one walk: the rows before they are in record order
    w.copy_count = take(256);
    w.copy_list = take((rows_cap + 1) * 16);
    w.decl_count = take(256);
    w.decl_list = take((n_rec + 1) * 4); // break-paf in one walk: the records its clip kernel declined
    w.gen_cp = take(rb_no_gen_cp() ? 256 : (size_t)(n_ops / RB_GCP + n_rec + 2) * sizeof(uint4)); // checkpoints of the records the generic kernel works on
    w.diag_stamps = take((n_rec + 1) * 4); // diagnostics build of the clip kernel: when each record's wave was done
    w.fb_count = take(256);
    w.fb_list = take((n_rec + 1) * 4); // the records of the tiles the tile kernel handed to the per-record kernel
    w.total = o;
    return w;
}
extern "C" size_t rb_plan_workspace_bytes(const rb_plan *plan, uint64_t rows_cap) {
    if (!plan) return 0;
    return ws_of(plan->n_rec, rows_cap, plan->n_ops).total;
}
extern "C" size_t rb_plan_diag_stamps_offset(const rb_plan *plan, uint64_t rows_cap) {
    if (!plan) return 0;
    return ws_of(plan->n_rec, rows_cap, plan->n_ops).diag_stamps;
}

// ops per output slot: the batch's op index space, 32 ops (one 128-byte line) of room per record (records that share a line in the
// input must not share one in the output), a multiple of 32
static uint64_t slot_pad() { // (experiment, RB_SLOT_PAD=<ops>: where slot 1 lies relative to slot 0 below the 2 MB of a physical chunk)
    static const uint64_t v = [] { const char *e = getenv("RB_SLOT_PAD"); return e ? (uint64_t)strtoull(e, nullptr, 10) & ~(uint64_t)31 : 0ull; }();
    return v;
}
static uint64_t slot_stride_of(uint64_t n_ops, uint64_t n_rec) { return ((n_ops + 31) & ~(uint64_t)31) + 32 * n_rec + 64 + slot_pad(); }
extern "C" uint64_t rb_plan_out_capacity(const rb_plan *plan, int for_break) {
    if (!plan) return 0;
    const uint64_t slots = for_break ? std::min<uint32_t>(2u, RB_MS) : std::min<uint32_t>(plan->depth, RB_MS);
    return slots * slot_stride_of(plan->n_ops, plan->n_rec) + plan->n_ops / 16 + 32 * plan->n_rec + (1u << 20);
}

static uint32_t pick_arenas(uint64_t n_rec) {
    uint64_t a = n_rec / 256;
    if (a < 1) a = 1;
    if (a > RB_MAX_ARENA) a = RB_MAX_ARENA;
    return (uint32_t)a;
}

static int lift_common(rb_ctx *ctx, const rb_plan *plan, const rb_batch_view *b, const rb_norm_row *norm, int policy,
                       void *workspace, rb_hit_row *rows, uint64_t rows_cap, uint32_t *out_ops, uint64_t out_cap,
                       rb_counters *counters, bool is_break, uint32_t max_size) {
    if (!ctx || !plan || !b || !norm || !workspace || !counters) return RB_E_INVALID;
    if (b->n_rec != plan->n_rec) return fail(ctx, RB_E_INVALID, "plan was built for %llu records, batch has %llu", (unsigned long long)plan->n_rec, (unsigned long long)b->n_rec);
    if (((uintptr_t)b->ops & 15u) || ((uintptr_t)out_ops & 15u)) return fail(ctx, RB_E_INVALID, "ops/out_ops must be 16-byte aligned");
    if (rows_cap >= 0xFFFFFFFFull) return fail(ctx, RB_E_INVALID, "rows_cap too large");
    if ((uintptr_t)workspace & 255u) return fail(ctx, RB_E_INVALID, "workspace must be 256-byte aligned");
    const ws_layout w = ws_of(plan->n_rec, rows_cap, plan->n_ops);
    char *ws = (char *)workspace;
    rb_lift_params p;
    memset(&p, 0, sizeof p);
    p.n_rec = b->n_rec;
    p.ops = b->ops;
    p.op_off = b->op_off;
    p.contig = b->contig;
    p.strand = b->strand;
    p.norm = norm;
    p.sched = plan->sched;
    p.slot_of = plan->slot_of;
    p.canon_pos = is_break ? plan->ident : plan->canon_pos;
    p.w_st = plan->w_st;
    p.w_en = plan->w_en;
    p.w_orig = plan->w_orig;
    p.wo_st = plan->wo_st;
    p.wo_en = plan->wo_en;
    p.cw_off = plan->cw_off;
    p.cw_mono = plan->cw_mono;
    p.n_contig = plan->n_contig;
    p.hit_off = (uint64_t *)(ws + w.hit_off);
    p.win_lo = (uint32_t *)(ws + w.win_lo);
    p.rows = rows;
    p.rows_cap = rows_cap;
    p.out_ops = out_ops;
    p.out_cap = out_cap;
    p.arena_cur = (unsigned long long *)(ws + w.arena);
    p.n_arena = pick_arenas(b->n_rec);
    p.desc_mode = (policy & RB_LIFT_DESCRIPTORS) ? 1 : 0;
    // out_ops = [slots the streaming kernel emits into | arenas of the generic and the copy kernel] (descriptor mode: [descriptors | arenas]).
    // As many slots as the window lists overlap deep (break-paf: two -- its pieces do not overlap, but neighbours end and begin in
    // the same 128-byte line, and a slot wants its clips four lines apart), as far as out_cap allows; with fewer,
    // or none, the clips that lose their place are copied into the arenas instead: slower, same rows.
    p.slot_stride = slot_stride_of(b->n_ops, b->n_rec);
    const uint32_t want = p.desc_mode ? 0u : std::min<uint32_t>(is_break ? 2u : plan->depth, RB_MS);
    const uint64_t min_arena = 1024ull * p.n_arena;
    uint32_t n_slots = want;
    while (n_slots && (uint64_t)n_slots * p.slot_stride + min_arena > out_cap) n_slots--;
    if (getenv("RB_DEBUG_SLOTS")) n_slots = std::min<uint32_t>(n_slots, (uint32_t)atoi(getenv("RB_DEBUG_SLOTS"))); // tests: force the copy path
    p.n_slots = n_slots;
    p.needed_base = p.desc_mode ? 4 * rows_cap : (uint64_t)want * p.slot_stride;
    p.arena_origin = p.desc_mode ? 4 * rows_cap : (uint64_t)n_slots * p.slot_stride;
    if (p.desc_mode && out_cap < p.arena_origin + 1024) return fail(ctx, RB_E_CAPACITY, "descriptor mode needs out_cap >= 4 * rows_cap + 1024");
    p.arena_size = out_cap > p.arena_origin ? ((out_cap - p.arena_origin) / p.n_arena) & ~(uint64_t)3 : 0;
    p.copy_list = (uint4 *)(ws + w.copy_list);
    // the generic kernel's per-hit descriptors take three arrays that are done with by the time it runs (rb_lift.h)
    p.gj_a = (rb_gja *)(ws + w.brk_rows), p.gj_b = (uint4 *)(ws + w.copy_list), p.gj_c = (uint2 *)(ws + w.bp_tmp);
    p.copy_count = (unsigned long long *)(ws + w.copy_count);
    p.gen_list = (uint32_t *)(ws + w.gen_list);
    p.gen_cp = rb_no_gen_cp() ? nullptr : (uint4 *)(ws + w.gen_cp);
    p.diag_stamps = (uint32_t *)(ws + w.diag_stamps); // (diagnostics: every generic walk from the record's first op, as before round 3)
    p.jobs = (rb_job *)(ws + w.jobs);
    p.fused = (policy & RB_LIFT_FUSED_SCAN) ? 1 : 0;
    p.op_starts = (policy & RB_LIFT_OP_STARTS) ? 1 : 0;
    if (p.op_starts && p.fused) return fail(ctx, RB_E_INVALID, "RB_LIFT_OP_STARTS takes finished norm_rows: not together with RB_LIFT_FUSED_SCAN");
    if (p.op_starts && (policy & RB_LIFT_DESCRIPTORS)) return fail(ctx, RB_E_INVALID, "RB_LIFT_OP_STARTS with RB_LIFT_DESCRIPTORS: a descriptor indexes the record's ORIGINAL cigar, which a batch cut in place no longer is");
    if (p.op_starts && b->n_ops > plan->n_ops) return fail(ctx, RB_E_INVALID, "RB_LIFT_OP_STARTS: batch->n_ops exceeds the plan's op count (gather the batch first)");
    p.norm_w = const_cast<rb_norm_row *>(norm);
    p.pend_list = (uint32_t *)(ws + w.pend_list);
    p.pend_count = (unsigned long long *)(ws + w.pend_count);
    // short records go through the tile kernel (k_tile.hip); the diagnostics builds keep every record on the per-record kernel
    const bool use_tiles = plan->n_tiles != 0 && ((policy >> 8) & 0xFFF) == 0 && !getenv("RB_DEBUG_NO_TILES");
    p.tile_first = use_tiles ? plan->tiles : nullptr;
    p.n_tiles = use_tiles ? plan->n_tiles : 0;
    p.fb_list = (uint32_t *)(ws + w.fb_list);
    p.fb_count = (unsigned long long *)(ws + w.fb_count);
    const uint32_t stream_end = use_tiles ? plan->stream_end : (uint32_t)b->n_rec;
    rb_scan_params sp;
    memset(&sp, 0, sizeof sp);
    if (p.fused) {
        sp.n_rec = b->n_rec;
        sp.ops = b->ops;
        sp.op_off = b->op_off;
        sp.t_st = b->t_st, sp.t_en = b->t_en, sp.q_st = b->q_st, sp.q_en = b->q_en;
        sp.strand = b->strand;
        sp.norm_rows = p.norm_w;
        HIPCHK(ctx, rb_fill_async(p.pend_count, 0, 8, ctx->stream));
        HIPCHK(ctx, rb_launch_peek_norm(&sp, ctx->stream)); // provisional rows from the records' ends
    }
    p.counters = counters;
    p.policy = policy & 1;
    p.early_exit = (policy & RB_LIFT_EARLY_EXIT) ? 1 : 0;
    p.debug_skip = (policy >> 8) & 0xFFF; // diagnostics only, undocumented on purpose
    uint64_t *block_sums = (uint64_t *)(ws + w.block_sums);
    HIPCHK(ctx, rb_fill_async(counters, 0, sizeof(rb_counters), ctx->stream));
    HIPCHK(ctx, rb_fill_async(p.copy_count, 0, 8, ctx->stream));
    HIPCHK(ctx, rb_fill_async(p.arena_cur, 0, (size_t)RB_MAX_ARENA * RB_ARENA_STRIDE * 8, ctx->stream));
    if (b->n_rec == 0) return RB_OK;
    const bool one_walk = is_break && (policy & RB_BREAK_ONE_WALK) && !p.desc_mode;
    if (one_walk) {
        // the clip kernel finds the long indels while it streams (k_liftover.hip, BRK): no pass over the ops before it.  Rows land
        // in scratch (one bump cursor of 256 per record), the piece counts are scanned, rb_k_break_gather orders the rows
        p.brk_mode = 1, p.brk_max = max_size;
        p.rows_final = rows;
        p.rows = (rb_hit_row *)(ws + w.brk_rows);
        p.brk_off = (uint64_t *)(ws + w.bp_off);
        p.brk_cursor = (unsigned long long *)(ws + w.bp_cur);
        p.brk_n_arena = pick_arenas(b->n_rec);
        p.brk_arena_cap = rows_cap / p.brk_n_arena;
        p.brk_decl_list = (uint32_t *)(ws + w.decl_list);
        p.brk_decl_count = (unsigned long long *)(ws + w.decl_count);
        HIPCHK(ctx, rb_fill_async(p.brk_decl_count, 0, 8, ctx->stream));
        HIPCHK(ctx, rb_fill_async(p.brk_cursor, 0, (size_t)RB_MAX_ARENA * 128, ctx->stream));
        HIPCHK(ctx, rb_fill_async(p.hit_off, 0, (size_t)(b->n_rec + 2) * 8, ctx->stream));
        HIPCHK(ctx, rb_fill_async(p.brk_off, 0xFF, (size_t)(b->n_rec + 1) * 8, ctx->stream));
        HIPCHK(ctx, rb_launch_make_jobs(&p, ctx->stream));
        const size_t slot1 = (size_t)(ctx->timed_calls % RB_TIMING_RING);
        if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev_a[slot1], ctx->stream));
        p.wave0 = 0;
        p.wave_end = stream_end;
        HIPCHK(ctx, rb_launch_liftover_stream(&p, ctx->stream));
        if (use_tiles) {
            HIPCHK(ctx, rb_fill_async(p.fb_count, 0, 8, ctx->stream));
            HIPCHK(ctx, rb_launch_liftover_tiles(&p, ctx->stream));
            HIPCHK(ctx, rb_launch_liftover_stream_list(&p, ctx->stream));
        }
        if (p.fused) {
            sp.list = p.pend_list;
            sp.n_list = (const uint64_t *)p.pend_count;
            HIPCHK(ctx, rb_launch_scan_records(&sp, ctx->stream));
        }
        if (ctx->timing) {
            HIPCHK(ctx, hipEventRecord(ctx->ev_b[slot1], ctx->stream));
            ctx->timed_calls++;
        }
        // the records the clip kernel declined, one by one: their pieces counted (list mode) before the scan of the counts, their
        // windows written to their rows' places after it, rows + generic-list entries made, and the generic kernel clips them
        rb_break_params bp;
        memset(&bp, 0, sizeof bp);
        bp.n_rec = b->n_rec, bp.ops = b->ops, bp.op_off = b->op_off, bp.norm = norm, bp.sched = plan->sched, bp.hit_off = p.hit_off;
        bp.x_st = (uint64_t *)(ws + w.x_st), bp.x_en = (uint64_t *)(ws + w.x_en);
        bp.rows_cap = rows_cap, bp.max_size = max_size;
        bp.list = p.brk_decl_list, bp.n_list = p.brk_decl_count;
        bp.fill = 0;
        HIPCHK(ctx, rb_launch_break_list_declined(&p, ctx->stream));
        HIPCHK(ctx, rb_launch_break_pieces(&bp, ctx->stream));
        HIPCHK(ctx, rb_launch_count_and_scan(&p, block_sums, false, ctx->stream));
        bp.fill = 1;
        HIPCHK(ctx, rb_launch_break_pieces(&bp, ctx->stream));
        HIPCHK(ctx, rb_launch_break_gather(&p, ctx->stream)); // (copies the clips without a slot, moves the rows into record order)
        rb_lift_params pg = p; // the generic kernel works on the final rows, with the declined records' windows as explicit windows
        pg.rows = rows, pg.x_st = bp.x_st, pg.x_en = bp.x_en;
        HIPCHK(ctx, rb_launch_break_declined(&pg, ctx->stream));
        return RB_OK;
    }
    if (is_break) {
        rb_break_params bp;
        memset(&bp, 0, sizeof bp);
        bp.n_rec = b->n_rec;
        bp.ops = b->ops;
        bp.op_off = b->op_off;
        bp.norm = norm;
        bp.sched = plan->sched;
        bp.hit_off = p.hit_off;
        bp.x_st = (uint64_t *)(ws + w.x_st);
        bp.x_en = (uint64_t *)(ws + w.x_en);
        bp.rows_cap = rows_cap;
        bp.max_size = max_size;
        // one walk of the ops: count the pieces of every record and keep their windows (tmp[]), scan the counts, move the
        // windows to their rows; a second walk only for records with more pieces than the collect pass keeps
        bp.tmp = ws + w.bp_tmp;
        bp.tmp_off = (uint64_t *)(ws + w.bp_off);
        bp.tmp_cursor = (unsigned long long *)(ws + w.bp_cur);
        bp.n_arena = pick_arenas(b->n_rec);
        bp.arena_cap = rows_cap / bp.n_arena;
        HIPCHK(ctx, rb_fill_async(bp.tmp_cursor, 0, (size_t)RB_MAX_ARENA * 128, ctx->stream));
        bp.fill = 2, bp.redo_only = 0;
        HIPCHK(ctx, rb_launch_break_pieces(&bp, ctx->stream));
        HIPCHK(ctx, rb_launch_count_and_scan(&p, block_sums, false, ctx->stream));
        HIPCHK(ctx, rb_launch_break_place(&bp, ctx->stream));
        bp.fill = 1, bp.redo_only = 1;
        HIPCHK(ctx, rb_launch_break_pieces(&bp, ctx->stream));
        p.x_st = bp.x_st;
        p.x_en = bp.x_en;
    } else {
        HIPCHK(ctx, rb_launch_count_and_scan(&p, block_sums, true, ctx->stream));
    }
    HIPCHK(ctx, rb_launch_make_jobs(&p, ctx->stream));
    const size_t slot = (size_t)(ctx->timed_calls % RB_TIMING_RING);
    if (ctx->timing) HIPCHK(ctx, hipEventRecord(ctx->ev_a[slot], ctx->stream));
    p.wave0 = 0;
    p.wave_end = stream_end;
    HIPCHK(ctx, rb_launch_liftover_stream(&p, ctx->stream));
    if (use_tiles) {
        HIPCHK(ctx, rb_fill_async(p.fb_count, 0, 8, ctx->stream));
        HIPCHK(ctx, rb_launch_liftover_tiles(&p, ctx->stream));
        HIPCHK(ctx, rb_launch_liftover_stream_list(&p, ctx->stream));
    }
    if (p.fused) { // the full record scan for the records the clip kernel handed back (usually none)
        sp.list = p.pend_list;
        sp.n_list = (const uint64_t *)p.pend_count;
        HIPCHK(ctx, rb_launch_scan_records(&sp, ctx->stream));
    }
    if (ctx->timing) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_b[slot], ctx->stream));
        ctx->timed_calls++;
    }
    HIPCHK(ctx, rb_launch_liftover_tail(&p, ctx->stream));
    return RB_OK;
}

extern "C" int rb_dev_liftover(rb_ctx *ctx, const rb_plan *plan, const rb_batch_view *batch, const rb_norm_row *norm_rows,
                               int bsearch_policy, void *workspace, rb_hit_row *rows, uint64_t rows_cap, uint32_t *out_ops,
                               uint64_t out_cap, rb_counters *counters) {
    return lift_common(ctx, plan, batch, norm_rows, bsearch_policy, workspace, rows, rows_cap, out_ops, out_cap, counters, false, 0);
}
extern "C" int rb_dev_break(rb_ctx *ctx, const rb_plan *plan, const rb_batch_view *batch, const rb_norm_row *norm_rows,
                            uint32_t max_size, int bsearch_policy, void *workspace, rb_hit_row *rows, uint64_t rows_cap,
                            uint32_t *out_ops, uint64_t out_cap, rb_counters *counters) {
    return lift_common(ctx, plan, batch, norm_rows, bsearch_policy, workspace, rows, rows_cap, out_ops, out_cap, counters, true, max_size);
}

extern "C" int rb_dev_swap(rb_ctx *ctx, const rb_batch_view *b, uint32_t *out_ops) {
    if (!ctx || !b || !out_ops) return RB_E_INVALID;
    rb_swap_params p;
    p.n_rec = b->n_rec;
    p.ops = b->ops;
    p.op_off = b->op_off;
    p.strand = b->strand;
    p.out_ops = out_ops;
    HIPCHK(ctx, rb_launch_swap(&p, ctx->stream));
    return RB_OK;
}

static int trim_pend_reserve(rb_ctx *ctx, uint64_t n_pairs) {
    if (ctx->trim_pend_cap >= n_pairs) return RB_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->trim_pend) hipFree(ctx->trim_pend);
    ctx->trim_pend = nullptr, ctx->trim_pend_cap = 0;
    const uint64_t cap = n_pairs + n_pairs / 4 + 1024;
    if (hipMalloc(&ctx->trim_pend, 256 + cap * 4) == hipSuccess) ctx->trim_pend_cap = cap;
    else (void)hipGetLastError(); // (without a list the wave-per-pair kernel looks at every pair: slower, same rows)
    return RB_OK;
}
extern "C" int rb_dev_trim_reserve(rb_ctx *ctx, uint64_t n_pairs) {
    if (!ctx) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return trim_pend_reserve(ctx, n_pairs);
}

extern "C" int rb_dev_overlap_split(rb_ctx *ctx, const rb_batch_view *b, const rb_norm_row *norm, uint64_t n_pairs, const uint32_t *left,
                                    const uint32_t *right, const uint64_t *pair_out_off, int match_score, int diff_score,
                                    int indel_score, int policy, rb_pair_row *rows, uint32_t *out_ops) {
    if (!ctx || !b || !norm || (n_pairs && (!left || !right || !pair_out_off || !rows || !out_ops))) return RB_E_INVALID;
    rb_trim_params p;
    p.n_pairs = n_pairs;
    p.ops = b->ops;
    p.op_off = b->op_off;
    p.strand = b->strand;
    p.norm = norm;
    p.left = left;
    p.right = right;
    p.pair_out_off = pair_out_off;
    p.match_score = match_score;
    p.diff_score = diff_score;
    p.indel_score = indel_score;
    p.policy = policy & 1;
    p.rows = rows;
    p.out_ops = out_ops;
    p.only_pending = 0;
    p.in_place = (policy & RB_TRIM_IN_PLACE) ? 1 : 0;
    if (p.in_place && out_ops != b->ops) return fail(ctx, RB_E_INVALID, "RB_TRIM_IN_PLACE: out_ops must be the batch's own ops array");
    if (!ctx->trim_scratch) { // (40 MB, once per context; without it those pairs simply stay with the serial kernel)
        const uint32_t blocks = 48;
        if (hipMalloc(&ctx->trim_scratch, rb_trim_scratch_bytes(blocks)) == hipSuccess) ctx->trim_scratch_blocks = blocks;
        else ctx->trim_scratch = nullptr, (void)hipGetLastError();
    }
    p.scratch = (uint32_t *)ctx->trim_scratch;
    p.scratch_blocks = ctx->trim_scratch_blocks;
    { // (the list grows with the largest pass seen; a few bytes per pair.  rb_dev_trim_reserve takes the growth out of the first pass)
        const int rc = trim_pend_reserve(ctx, n_pairs);
        if (rc) return rc;
    }
    p.pend = nullptr, p.pend_list = nullptr;
    if (ctx->trim_pend) {
        p.pend = (unsigned long long *)ctx->trim_pend;
        p.pend_list = (uint32_t *)((char *)ctx->trim_pend + 256);
        HIPCHK(ctx, rb_fill_async(p.pend, 0, 8, ctx->stream));
    }
    HIPCHK(ctx, rb_launch_overlap_split(&p, ctx->stream));
    return RB_OK;
}

struct rb_apply_params {
    uint64_t n_pairs;
    const uint32_t *left, *right;
    const rb_pair_row *rows;
    uint64_t *op_off;
    rb_norm_row *norm;
};
struct rb_gather_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const rb_norm_row *norm;
    uint64_t *new_off;
    uint32_t *new_ops;
    int fill;
};
struct rb_tsel_params {
    uint64_t n_groups;
    const uint32_t *order;
    const uint64_t *grp_off;
    const rb_norm_row *norm;
    uint8_t *contained;
    uint64_t *slot;
    uint64_t *has;
    uint32_t *cand;
    uint64_t out_base;
    uint32_t *left, *right;
    uint64_t *pair_out_off;
    rb_trim_pass *pass;
};
extern "C" hipError_t rb_launch_trim_select(const rb_tsel_params *p, uint64_t *block_sums, hipStream_t stream);
extern "C" hipError_t rb_launch_trim_check(const rb_pair_row *rows, uint64_t n_pairs, rb_trim_pass *pass, hipStream_t stream);
// scratch of rb_dev_trim_select: [slot (n_groups + 2) u64][has (n_groups + 2) u64][cand 2 n_groups u32][block sums of the scans]
extern "C" size_t rb_trim_select_scratch_bytes(uint64_t n_groups) {
    return 2 * (((n_groups + 2) * 8 + 255) & ~(size_t)255) + ((2 * n_groups * 4 + 255) & ~(size_t)255) + (rb_scan_block_sums_count(n_groups) + 4) * 8 + 256;
}
extern "C" int rb_dev_trim_select(rb_ctx *ctx, uint64_t n_rec, uint64_t n_groups, const uint32_t *order, const uint64_t *grp_off,
                                  const rb_norm_row *norm_rows, uint64_t out_base, uint8_t *contained, uint32_t *left, uint32_t *right,
                                  uint64_t *pair_out_off, rb_trim_pass *pass, void *scratch) {
    if (!ctx || !pass || (n_groups && (!order || !grp_off || !norm_rows || !contained || !left || !right || !pair_out_off || !scratch))) return RB_E_INVALID;
    (void)n_rec;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    rb_tsel_params p;
    memset(&p, 0, sizeof p);
    char *sc = (char *)scratch;
    p.n_groups = n_groups, p.order = order, p.grp_off = grp_off, p.norm = norm_rows, p.contained = contained;
    p.slot = (uint64_t *)sc;
    sc += ((n_groups + 2) * 8 + 255) & ~(size_t)255;
    p.has = (uint64_t *)sc;
    sc += ((n_groups + 2) * 8 + 255) & ~(size_t)255;
    p.cand = (uint32_t *)sc;
    sc += (2 * n_groups * 4 + 255) & ~(size_t)255;
    p.out_base = out_base, p.left = left, p.right = right, p.pair_out_off = pair_out_off, p.pass = pass;
    HIPCHK(ctx, rb_launch_trim_select(&p, (uint64_t *)sc, ctx->stream));
    return RB_OK;
}
extern "C" int rb_dev_trim_check(rb_ctx *ctx, uint64_t n_pairs, const rb_pair_row *rows, rb_trim_pass *pass) {
    if (!ctx || !pass || (n_pairs && !rows)) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, rb_launch_trim_check(rows, n_pairs, pass, ctx->stream));
    return RB_OK;
}
extern "C" hipError_t rb_launch_apply_pairs(const rb_apply_params *p, hipStream_t stream);
extern "C" hipError_t rb_launch_gather_records(const rb_gather_params *p, hipStream_t stream);
extern "C" int rb_dev_apply_pairs(rb_ctx *ctx, uint64_t n_pairs, const uint32_t *left, const uint32_t *right, const rb_pair_row *rows,
                                  uint64_t *op_off, rb_norm_row *norm_rows) {
    if (!ctx || (n_pairs && (!left || !right || !rows || !op_off || !norm_rows))) return RB_E_INVALID;
    rb_apply_params p{n_pairs, left, right, rows, op_off, norm_rows};
    HIPCHK(ctx, rb_launch_apply_pairs(&p, ctx->stream));
    return RB_OK;
}
extern "C" int rb_dev_gather_records(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const rb_norm_row *norm_rows,
                                     uint64_t *new_op_off, uint32_t *new_ops, void *scratch) {
    if (!ctx || !new_op_off || !scratch || (n_rec && (!ops || !op_off || !norm_rows))) return RB_E_INVALID;
    rb_gather_params p{n_rec, ops, op_off, norm_rows, new_op_off, new_ops, 0};
    HIPCHK(ctx, rb_fill_async(new_op_off + n_rec, 0, 8, ctx->stream));
    if (n_rec == 0) return RB_OK;
    HIPCHK(ctx, rb_launch_gather_records(&p, ctx->stream));
    HIPCHK(ctx, rb_launch_exclusive_scan(new_op_off, n_rec, (uint64_t *)scratch, nullptr, ctx->stream));
    if (new_ops) { // (NULL: sizes only -- new_op_off[n_rec] says how many ops the dense batch holds)
        p.fill = 1;
        HIPCHK(ctx, rb_launch_gather_records(&p, ctx->stream));
    }
    return RB_OK;
}

// ---- CIGAR text ----------------------------------------------------------------------------------
extern "C" size_t rb_text_scratch_bytes(uint64_t n) { return (rb_scan_block_sums_count(n) + 4) * 8; }
extern "C" int rb_dev_parse_cigars(rb_ctx *ctx, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_end, uint64_t n_rec,
                                   uint64_t *op_off, uint32_t *ops, uint64_t ops_cap, uint8_t *status, void *scratch) {
    if (!ctx || !text_off || !op_off || !status || !scratch || (n_rec && !text)) return RB_E_INVALID;
    if (((uintptr_t)text & 15u) != 0) return fail(ctx, RB_E_INVALID, "text must be 16-byte aligned");
    rb_parse_params p;
    p.n_rec = n_rec;
    p.text = text;
    p.text_off = text_off;
    p.text_end = text_end;
    p.op_off = op_off;
    p.ops = ops;
    p.ops_cap = ops_cap;
    p.status = status;
    HIPCHK(ctx, rb_fill_async(op_off + n_rec, 0, 8, ctx->stream));
    if (n_rec == 0) return RB_OK;
    HIPCHK(ctx, rb_launch_parse_cigars(&p, false, ctx->stream));
    HIPCHK(ctx, rb_launch_exclusive_scan(op_off, n_rec, (uint64_t *)scratch, nullptr, ctx->stream));
    HIPCHK(ctx, rb_launch_parse_cigars(&p, true, ctx->stream));
    return RB_OK;
}
// offsets_ready: text_off already holds the prefix of a sizes-only call on the same items (the host wrappers size, allocate, fill)
static int format_cigars_impl(rb_ctx *ctx, const uint32_t *ops, const uint32_t *ops_alt, uint64_t n_items, const uint64_t *first,
                              const uint32_t *count, const uint32_t *first_len, const uint32_t *last_len, uint64_t *text_off,
                              uint8_t *text, uint64_t text_cap, void *scratch, bool offsets_ready, bool plain_ops = false) {
    if (!ctx || !text_off || !scratch || (n_items && (!ops || !first || !count))) return RB_E_INVALID;
    rb_format_params p;
    p.plain_ops = plain_ops ? 1 : 0; // (the caller knows that ops[] came out of rb_k_parse_cigars: no continuation words to look for)
    p.n_items = n_items;
    p.ops = ops;
    p.ops_alt = ops_alt;
    p.first = first;
    p.count = count;
    p.first_len = first_len;
    p.last_len = last_len;
    p.text_off = text_off;
    p.text = text;
    p.text_cap = text_cap;
    if (!offsets_ready) HIPCHK(ctx, rb_fill_async(text_off + n_items, 0, 8, ctx->stream));
    if (n_items == 0) return RB_OK;
    if (!offsets_ready) {
        HIPCHK(ctx, rb_launch_format_cigars(&p, false, ctx->stream));
        HIPCHK(ctx, rb_launch_exclusive_scan(text_off, n_items, (uint64_t *)scratch, nullptr, ctx->stream));
    }
    if (text) HIPCHK(ctx, rb_launch_format_cigars(&p, true, ctx->stream));
    return RB_OK;
}
extern "C" int rb_dev_format_cigars(rb_ctx *ctx, const uint32_t *ops, const uint32_t *ops_alt, uint64_t n_items, const uint64_t *first,
                                    const uint32_t *count, const uint32_t *first_len, const uint32_t *last_len, uint64_t *text_off,
                                    uint8_t *text, uint64_t text_cap, void *scratch) {
    return format_cigars_impl(ctx, ops, ops_alt, n_items, first, count, first_len, last_len, text_off, text, text_cap, scratch, false);
}

// gigabyte-sized host buffers: ask for huge pages where the system lets a process opt in (transparent_hugepage = madvise): a
// 2 GB buffer is then a thousand faults and a thousand pages to free instead of half a million of each
static void rb_advise_huge(void *p, size_t bytes) {
    if (!p || bytes < ((size_t)8 << 20)) return;
    const uintptr_t a = ((uintptr_t)p + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1), e = ((uintptr_t)p + bytes) & ~(uintptr_t)(((size_t)2 << 20) - 1);
    if (e > a) (void)madvise((void *)a, (size_t)(e - a), MADV_HUGEPAGE);
}
// ---- host-buffer wrappers ----------------------------------------------------------------------
namespace {
struct DevBatch {
    rb_ctx *ctx;
    rb_batch_view v{};
    std::vector<void *> owned;
    explicit DevBatch(rb_ctx *c) : ctx(c) {}
    ~DevBatch() {
        for (void *q : owned) (void)rb_dev_free(ctx, q);
    }
    template <typename T>
    int up(const T *host, size_t n, const T **dev) {
        void *d = nullptr;
        int rc = rb_dev_alloc(ctx, n * sizeof(T) + 256, &d); // (the kernels read whole 128-byte lines of the ops)
        if (rc) return rc;
        owned.push_back(d);
        if (host && n) rc = rb_dev_upload(ctx, d, host, n * sizeof(T));
        *dev = (const T *)d;
        return rc;
    }
    template <typename T>
    int alloc(size_t n, T **dev) {
        void *d = nullptr;
        int rc = rb_dev_alloc(ctx, n * sizeof(T) + 64, &d);
        if (rc) return rc;
        owned.push_back(d);
        *dev = (T *)d;
        return RB_OK;
    }
    int load(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st, const uint64_t *t_en,
             const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand, const uint32_t *contig) {
        v.n_rec = n_rec;
        v.n_ops = n_rec ? op_off[n_rec] : 0;
        int rc = RB_OK;
        std::vector<uint32_t> zc;
        std::vector<uint8_t> zs;
        if (!contig) {
            zc.assign(n_rec, 0);
            contig = zc.data();
        }
        if (!strand) {
            zs.assign(n_rec, (uint8_t)'+');
            strand = zs.data();
        }
        if (!rc) rc = up(ops, (size_t)v.n_ops + 4, &v.ops);
        if (!rc) rc = up(op_off, (size_t)n_rec + 1, &v.op_off);
        if (!rc) rc = up(t_st, (size_t)n_rec, &v.t_st);
        if (!rc) rc = up(t_en, (size_t)n_rec, &v.t_en);
        if (!rc) rc = up(q_st, (size_t)n_rec, &v.q_st);
        if (!rc) rc = up(q_en, (size_t)n_rec, &v.q_en);
        if (!rc) rc = up(strand, (size_t)n_rec, &v.strand);
        if (!rc) rc = up(contig, (size_t)n_rec, &v.contig);
        if (!rc) rc = rb_ctx_sync(ctx); // the temporaries above must outlive the copies
        return rc;
    }
};
} // namespace

extern "C" void rb_host_free(void *p) { free(p); }

extern "C" int rb_host_scan_records(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                                    const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                                    rb_reduce_row *reduce_rows, rb_norm_row *norm_rows) {
    if (!ctx) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBatch b(ctx);
    int rc = b.load(n_rec, ops, op_off, t_st, t_en, q_st, q_en, strand, nullptr);
    if (rc) return rc;
    rb_reduce_row *d_red = nullptr;
    rb_norm_row *d_norm = nullptr;
    if (reduce_rows && (rc = b.alloc(n_rec, &d_red))) return rc;
    if (norm_rows && (rc = b.alloc(n_rec, &d_norm))) return rc;
    if ((rc = rb_dev_scan_records(ctx, &b.v, d_red, d_norm))) return rc;
    if (reduce_rows && (rc = rb_dev_download(ctx, reduce_rows, d_red, n_rec * sizeof(rb_reduce_row)))) return rc;
    if (norm_rows && (rc = rb_dev_download(ctx, norm_rows, d_norm, n_rec * sizeof(rb_norm_row)))) return rc;
    return rb_ctx_sync(ctx);
}

static double rb_now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
static void rb_lap(const char *what, double &t) {
    static const bool on = getenv("RB_TIMING") != nullptr;
    const double n = rb_now_s();
    if (on) fprintf(stderr, "[rb timing]     %-24s %.3f s\n", what, n - t);
    t = n;
}
static int host_lift(rb_ctx *ctx, bool is_break, uint32_t max_size, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off,
                     const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                     const uint32_t *contig, uint64_t n_win, const uint32_t *w_contig, const uint64_t *w_st, const uint64_t *w_en,
                     int policy, rb_norm_row *norm_out, rb_hit_row **rows, uint64_t *n_rows, uint32_t **out_ops, uint64_t *n_out,
                     rb_counters *counters) {
    if (!ctx || !rows || !n_rows || !out_ops || !n_out) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *rows = nullptr;
    *out_ops = nullptr;
    *n_rows = *n_out = 0;
    DevBatch b(ctx);
    std::vector<uint32_t> zc;
    if (!contig) {
        zc.assign(n_rec, 0);
        contig = zc.data();
    }
    double tl = rb_now_s();
    int rc = b.load(n_rec, ops, op_off, t_st, t_en, q_st, q_en, strand, contig);
    if (rc) return rc;
    rb_lap("H2D batch", tl);
    rb_norm_row *d_norm = nullptr;
    if ((rc = b.alloc(n_rec, &d_norm))) return rc;
    const bool fused = (policy & RB_LIFT_FUSED_SCAN) != 0; // the clip kernel verifies the records itself
    if (!fused) {
        if ((rc = rb_dev_scan_records(ctx, &b.v, nullptr, d_norm))) return rc;
        if (norm_out && (rc = rb_dev_download(ctx, norm_out, d_norm, n_rec * sizeof(rb_norm_row)))) return rc;
        rb_lap("scan_records + norm D2H", tl);
    }
    rb_plan *plan = nullptr;
    if ((rc = rb_plan_create(ctx, n_rec, op_off, contig, n_win, w_contig, w_st, w_en, &plan))) return rc;
    rb_lap("plan", tl);
    rb_counters *d_cnt = nullptr;
    if ((rc = b.alloc(1, &d_cnt))) {
        rb_plan_destroy(plan);
        return rc;
    }
    const uint64_t n_ops = n_rec ? op_off[n_rec] : 0;
    uint64_t rows_cap = 16 * n_rec + n_win + 1024; // a first guess; the counters say what is needed if it is short
    uint64_t out_cap = ((policy & RB_LIFT_DESCRIPTORS) ? n_ops / 4 : rb_plan_out_capacity(plan, is_break ? 1 : 0)) + 16 * rows_cap + 4096;
    rb_counters hc;
    memset(&hc, 0, sizeof hc);
    void *ws = nullptr;
    rb_hit_row *d_rows = nullptr;
    uint32_t *d_out = nullptr;
    bool one_walk = is_break && !(policy & RB_LIFT_DESCRIPTORS) && !getenv("RB_BREAK_TWO_WALK"); // (the env: diagnostics)
    for (int attempt = 0; attempt < 6; attempt++) {
        if (ws) (void)rb_dev_free(ctx, ws);
        if (d_rows) (void)rb_dev_free(ctx, d_rows);
        if (d_out) (void)rb_dev_free(ctx, d_out);
        ws = nullptr;
        d_rows = nullptr;
        d_out = nullptr;
        rc = rb_dev_alloc(ctx, rb_plan_workspace_bytes(plan, rows_cap), &ws);
        if (!rc) rc = rb_dev_alloc(ctx, (rows_cap + 1) * sizeof(rb_hit_row), (void **)&d_rows);
        if (!rc) rc = rb_dev_alloc(ctx, (out_cap + 4) * 4, (void **)&d_out);
        if (rc) break;
        rc = is_break ? rb_dev_break(ctx, plan, &b.v, d_norm, max_size, policy | (one_walk ? RB_BREAK_ONE_WALK : 0), ws, d_rows, rows_cap, d_out,
                                     out_cap, d_cnt)
                      : rb_dev_liftover(ctx, plan, &b.v, d_norm, policy, ws, d_rows, rows_cap, d_out, out_cap, d_cnt);
        if (rc) break;
        rc = rb_dev_download(ctx, &hc, d_cnt, sizeof hc);
        if (rc) break;
        if (one_walk && hc.redo_two_walk) { // something the one-walk path does not take: the same buffers, the two-walk path
            one_walk = false;
            attempt--;
            rc = RB_E_CAPACITY;
            continue;
        }
        if (!hc.overflow) break;
        rows_cap = std::max<uint64_t>(rows_cap, hc.n_hits + 16);
        out_cap = std::max<uint64_t>(out_cap * 2, hc.out_ops_needed + hc.out_ops_needed / 4 + 4096 + 4 * rows_cap);
        rc = RB_E_CAPACITY;
    }
    if (!rc && hc.overflow) rc = RB_E_CAPACITY;
    rb_lap("alloc + kernels", tl);
    if (!rc && fused && norm_out) rc = rb_dev_download(ctx, norm_out, d_norm, n_rec * sizeof(rb_norm_row));
    if (!rc) {
        // the clips packed side by side in row order ON THE DEVICE (the slots of out_ops are as large as the batch: only the clips
        // themselves cross PCIe), rows' out_off rebased onto the dense array
        *n_rows = hc.n_hits;
        *rows = (rb_hit_row *)malloc((size_t)(hc.n_hits + 1) * sizeof(rb_hit_row));
        uint64_t o = 0;
        if (hc.n_hits) {
            uint64_t *d_off = nullptr, *d_blk = nullptr;
            uint32_t *d_dense = nullptr;
            rc = rb_dev_alloc(ctx, (hc.n_hits + 2) * 8, (void **)&d_off);
            if (!rc) rc = rb_dev_alloc(ctx, (rb_scan_block_sums_count(hc.n_hits) + 2) * 8, (void **)&d_blk);
            rb_compact_params cp{hc.n_hits, d_rows, d_out, d_off, nullptr, 0};
            if (!rc && rb_launch_compact_clips(&cp, ctx->stream) != hipSuccess) rc = RB_E_HIP;
            if (!rc && rb_launch_exclusive_scan(d_off, hc.n_hits, d_blk, d_off + hc.n_hits, ctx->stream) != hipSuccess) rc = RB_E_HIP;
            if (!rc) rc = rb_dev_download(ctx, &o, d_off + hc.n_hits, 8);
            if (!rc) rc = rb_dev_alloc(ctx, (o + 4) * 4, (void **)&d_dense);
            cp.dst = d_dense, cp.fill = 1;
            if (!rc && rb_launch_compact_clips(&cp, ctx->stream) != hipSuccess) rc = RB_E_HIP;
            if (!rc) rc = rb_dev_download(ctx, *rows, d_rows, (size_t)hc.n_hits * sizeof(rb_hit_row));
            *out_ops = (uint32_t *)malloc((size_t)(o + 1) * 4);
            if (!rc && o) rc = rb_dev_download(ctx, *out_ops, d_dense, (size_t)o * 4);
            if (d_off) (void)rb_dev_free(ctx, d_off);
            if (d_blk) (void)rb_dev_free(ctx, d_blk);
            if (d_dense) (void)rb_dev_free(ctx, d_dense);
            for (uint64_t i = 0; i < hc.n_hits && !rc; i++) // (rows that carry no clip: the fields the reference has no value for)
                if ((*rows)[i].status != RB_ST_OK) (*rows)[i].out_off = 0, (*rows)[i].out_n = 0;
        } else {
            *out_ops = (uint32_t *)malloc(4);
        }
        *n_out = o;
        if (counters) *counters = hc;
        rb_lap("D2H + compact", tl);
    }
    if (ws) (void)rb_dev_free(ctx, ws);
    if (d_rows) (void)rb_dev_free(ctx, d_rows);
    if (d_out) (void)rb_dev_free(ctx, d_out);
    rb_plan_destroy(plan);
    if (rc) {
        free(*rows);
        free(*out_ops);
        *rows = nullptr;
        *out_ops = nullptr;
        *n_rows = *n_out = 0;
    }
    return rc;
}

extern "C" int rb_host_liftover(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                                const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                                const uint32_t *contig, uint64_t n_win, const uint32_t *w_contig, const uint64_t *w_st,
                                const uint64_t *w_en, int policy, rb_norm_row *norm_out, rb_hit_row **rows, uint64_t *n_rows,
                                uint32_t **out_ops, uint64_t *n_out, rb_counters *counters) {
    return host_lift(ctx, false, 0, n_rec, ops, op_off, t_st, t_en, q_st, q_en, strand, contig, n_win, w_contig, w_st, w_en, policy,
                     norm_out, rows, n_rows, out_ops, n_out, counters);
}
extern "C" int rb_host_break(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                             const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand, uint32_t max_size,
                             int policy, rb_norm_row *norm_out, rb_hit_row **rows, uint64_t *n_rows, uint32_t **out_ops,
                             uint64_t *n_out, rb_counters *counters) {
    return host_lift(ctx, true, max_size, n_rec, ops, op_off, t_st, t_en, q_st, q_en, strand, nullptr, 0, nullptr, nullptr, nullptr,
                     policy, norm_out, rows, n_rows, out_ops, n_out, counters);
}

extern "C" int rb_host_parse_cigars(rb_ctx *ctx, const uint8_t *text, const uint64_t *text_off, const uint64_t *text_end, uint64_t n_rec,
                                    uint64_t *op_off, uint32_t **ops, uint8_t *status) {
    if (!ctx || !text_off || !op_off || !ops || !status) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *ops = nullptr;
    uint64_t text_bytes = 0;
    for (uint64_t r = 0; r < n_rec; r++) text_bytes = std::max(text_bytes, text_end ? text_end[r] : text_off[r + 1]);
    DevBatch b(ctx);
    const uint8_t *d_text = nullptr;
    const uint64_t *d_off = nullptr, *d_end = nullptr;
    uint64_t *d_opoff = nullptr;
    uint32_t *d_ops = nullptr;
    uint8_t *d_status = nullptr;
    void *d_scr = nullptr;
    int rc;
    {
        uint8_t *t = nullptr;
        if ((rc = b.alloc((size_t)text_bytes + 32, &t))) return rc;
        if (text_bytes && (rc = rb_dev_upload(ctx, t, text, (size_t)text_bytes))) return rc;
        d_text = t;
    }
    if ((rc = b.up(text_off, (size_t)n_rec + 1, &d_off))) return rc;
    if (text_end && (rc = b.up(text_end, (size_t)n_rec, &d_end))) return rc;
    const uint64_t ops_cap = text_bytes / 2 + 4;
    if ((rc = b.alloc((size_t)n_rec + 2, &d_opoff))) return rc;
    if ((rc = b.alloc((size_t)ops_cap, &d_ops))) return rc;
    if ((rc = b.alloc((size_t)n_rec + 1, &d_status))) return rc;
    if ((rc = b.alloc(rb_text_scratch_bytes(n_rec), (uint8_t **)&d_scr))) return rc;
    if ((rc = rb_dev_parse_cigars(ctx, d_text, d_off, d_end, n_rec, d_opoff, d_ops, ops_cap, d_status, d_scr))) return rc;
    if ((rc = rb_dev_download(ctx, op_off, d_opoff, ((size_t)n_rec + 1) * 8))) return rc;
    if (n_rec && (rc = rb_dev_download(ctx, status, d_status, (size_t)n_rec))) return rc;
    const uint64_t n_ops = op_off[n_rec];
    *ops = (uint32_t *)malloc(((size_t)n_ops + 4) * 4);
    if (!*ops) return fail(ctx, RB_E_NOMEM, "malloc(%llu ops)", (unsigned long long)n_ops);
    if (n_ops) return rb_dev_download(ctx, *ops, d_ops, (size_t)n_ops * 4);
    return RB_OK;
}
extern "C" int rb_host_format_cigars(rb_ctx *ctx, const uint32_t *ops, uint64_t n_ops, uint64_t n_items, const uint64_t *first,
                                     const uint32_t *count, const uint32_t *first_len, const uint32_t *last_len, uint64_t *text_off,
                                     uint8_t **text) {
    if (!ctx || !text_off || !text || (n_items && (!first || !count))) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *text = nullptr;
    for (uint64_t i = 0; i < n_items; i++)
        if (first[i] + count[i] > n_ops) return fail(ctx, RB_E_INVALID, "item %llu reaches past the ops", (unsigned long long)i);
    DevBatch b(ctx);
    const uint32_t *d_ops = nullptr, *d_count = nullptr, *d_fl = nullptr, *d_ll = nullptr;
    const uint64_t *d_first = nullptr;
    uint64_t *d_toff = nullptr;
    uint8_t *d_text = nullptr;
    void *d_scr = nullptr;
    int rc;
    if ((rc = b.up(ops, (size_t)n_ops, &d_ops))) return rc;
    if ((rc = b.up(first, (size_t)n_items, &d_first))) return rc;
    if ((rc = b.up(count, (size_t)n_items, &d_count))) return rc;
    if (first_len && (rc = b.up(first_len, (size_t)n_items, &d_fl))) return rc;
    if (last_len && (rc = b.up(last_len, (size_t)n_items, &d_ll))) return rc;
    if ((rc = b.alloc((size_t)n_items + 2, &d_toff))) return rc;
    if ((rc = b.alloc(rb_text_scratch_bytes(n_items), (uint8_t **)&d_scr))) return rc;
    if ((rc = rb_dev_format_cigars(ctx, d_ops, nullptr, n_items, d_first, d_count, d_fl, d_ll, d_toff, nullptr, 0, d_scr))) return rc; // sizes
    if ((rc = rb_dev_download(ctx, text_off, d_toff, ((size_t)n_items + 1) * 8))) return rc;
    const uint64_t bytes = text_off[n_items];
    if ((rc = b.alloc((size_t)bytes + 16, &d_text))) return rc;
    if ((rc = format_cigars_impl(ctx, d_ops, nullptr, n_items, d_first, d_count, d_fl, d_ll, d_toff, d_text, bytes, d_scr, true))) return rc;
    *text = (uint8_t *)malloc((size_t)bytes + 16);
    if (!*text) return fail(ctx, RB_E_NOMEM, "malloc(%llu text bytes)", (unsigned long long)bytes);
    rb_advise_huge(*text, (size_t)bytes);
    if (bytes) return rb_dev_download(ctx, *text, d_text, (size_t)bytes);
    return RB_OK;
}

// text in -> text out around the clip kernels; scan_only stops after the record scan (rb_host_scan_text)
static int host_lift_text(rb_ctx *ctx, bool is_break, uint32_t max_size, bool scan_only, uint64_t n_rec, const uint8_t *text,
                          uint64_t text_bytes, const uint64_t *cig_off, const uint64_t *cig_end, const uint64_t *t_st, const uint64_t *t_en,
                          const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand, const uint32_t *contig, uint64_t n_win,
                          const uint32_t *w_contig, const uint64_t *w_st, const uint64_t *w_en, int policy, uint8_t *cig_status,
                          rb_reduce_row *reduce_out, rb_norm_row *norm_out, rb_hit_row **rows, uint64_t *n_rows, uint64_t **row_text_off,
                          uint8_t **row_text, rb_counters *counters) {
    if (!ctx || (n_rec && (!cig_off || !cig_end || !cig_status))) return RB_E_INVALID;
    if (n_rec == 0) { // an empty file: no rows, no text
        if (rows) *rows = nullptr;
        if (n_rows) *n_rows = 0;
        if (row_text) *row_text = nullptr;
        if (row_text_off) {
            *row_text_off = (uint64_t *)calloc(2, 8);
            if (!*row_text_off) return fail(ctx, RB_E_NOMEM, "malloc");
        }
        if (counters) memset(counters, 0, sizeof *counters);
        return RB_OK;
    }
    rb_hit_row *rows_dummy = nullptr;
    uint64_t n_dummy = 0, *off_dummy = nullptr;
    uint8_t *text_dummy = nullptr;
    if (scan_only) rows = &rows_dummy, n_rows = &n_dummy, row_text_off = &off_dummy, row_text = &text_dummy;
    if (!rows || !n_rows || !row_text_off || !row_text) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *rows = nullptr;
    *row_text_off = nullptr;
    *row_text = nullptr;
    *n_rows = 0;
    DevBatch b(ctx);
    int rc;
    double tl = rb_now_s();
    // ---- text -> ops on the device ----
    uint8_t *d_text = nullptr, *d_status = nullptr;
    const uint64_t *d_coff = nullptr, *d_cend = nullptr;
    uint64_t *d_opoff = nullptr;
    uint32_t *d_ops = nullptr;
    void *d_scr = nullptr;
    if ((rc = b.alloc((size_t)text_bytes + 32, &d_text))) return rc;
    if (text_bytes && (rc = rb_dev_upload(ctx, d_text, text, (size_t)text_bytes))) return rc;
    if ((rc = b.up(cig_off, (size_t)n_rec, &d_coff))) return rc;
    if ((rc = b.up(cig_end, (size_t)n_rec, &d_cend))) return rc;
    uint64_t cig_bytes = 0;
    for (uint64_t r = 0; r < n_rec; r++) {
        if (cig_end[r] < cig_off[r] || cig_end[r] > text_bytes) return fail(ctx, RB_E_INVALID, "record %llu: CIGAR range outside the text", (unsigned long long)r);
        cig_bytes += cig_end[r] - cig_off[r];
    }
    const uint64_t ops_cap = cig_bytes / 2 + 8;
    if ((rc = b.alloc((size_t)n_rec + 2, &d_opoff))) return rc;
    if ((rc = b.alloc((size_t)ops_cap + 4, &d_ops))) return rc;
    if ((rc = b.alloc((size_t)n_rec + 1, &d_status))) return rc;
    if ((rc = b.alloc(rb_text_scratch_bytes(n_rec), (uint8_t **)&d_scr))) return rc;
    rb_lap("H2D text", tl);
    if ((rc = rb_dev_parse_cigars(ctx, d_text, d_coff, d_cend, n_rec, d_opoff, d_ops, ops_cap, d_status, d_scr))) return rc;
    if (n_rec && (rc = rb_dev_download(ctx, cig_status, d_status, (size_t)n_rec))) return rc;
    for (uint64_t r = 0; r < n_rec; r++)
        if (cig_status[r] != RB_TEXT_OK) return RB_OK; // the caller reports it (the reference panics at paf.rs:399)
    std::vector<uint64_t> op_off((size_t)n_rec + 1, 0);
    if ((rc = rb_dev_download(ctx, op_off.data(), d_opoff, ((size_t)n_rec + 1) * 8))) return rc;
    rb_lap("parse cigars + op_off D2H", tl);
    // ---- the batch, device-resident ----
    std::vector<uint32_t> zc;
    if (!contig) {
        zc.assign(n_rec, 0);
        contig = zc.data();
    }
    b.v.n_rec = n_rec;
    b.v.n_ops = op_off[n_rec];
    b.v.ops = d_ops;
    b.v.op_off = d_opoff;
    if ((rc = b.up(t_st, (size_t)n_rec, &b.v.t_st))) return rc;
    if ((rc = b.up(t_en, (size_t)n_rec, &b.v.t_en))) return rc;
    if ((rc = b.up(q_st, (size_t)n_rec, &b.v.q_st))) return rc;
    if ((rc = b.up(q_en, (size_t)n_rec, &b.v.q_en))) return rc;
    if ((rc = b.up(strand, (size_t)n_rec, &b.v.strand))) return rc;
    if ((rc = b.up(contig, (size_t)n_rec, &b.v.contig))) return rc;
    rb_norm_row *d_norm = nullptr;
    rb_reduce_row *d_red = nullptr;
    if ((rc = b.alloc(n_rec, &d_norm))) return rc;
    if (reduce_out && (rc = b.alloc(n_rec, &d_red))) return rc;
    if ((rc = rb_dev_scan_records(ctx, &b.v, d_red, d_norm))) return rc;
    if (norm_out && (rc = rb_dev_download(ctx, norm_out, d_norm, n_rec * sizeof(rb_norm_row)))) return rc;
    if (reduce_out && (rc = rb_dev_download(ctx, reduce_out, d_red, n_rec * sizeof(rb_reduce_row)))) return rc;
    rb_lap("scan_records + rows D2H", tl);
    if (scan_only) return RB_OK;
    rb_plan *plan = nullptr;
    if ((rc = rb_plan_create(ctx, n_rec, op_off.data(), contig, n_win, w_contig, w_st, w_en, &plan))) return rc;
    rb_counters *d_cnt = nullptr;
    if ((rc = b.alloc(1, &d_cnt))) {
        rb_plan_destroy(plan);
        return rc;
    }
    policy |= RB_LIFT_DESCRIPTORS;
    uint64_t rows_cap = 16 * n_rec + n_win + 1024;
    uint64_t out_cap = b.v.n_ops / 4 + 16 * rows_cap + 4096;
    rb_counters hc;
    memset(&hc, 0, sizeof hc);
    void *ws = nullptr;
    rb_hit_row *d_rows = nullptr;
    uint32_t *d_out = nullptr;
    bool one_walk = is_break && !(policy & RB_LIFT_DESCRIPTORS) && !getenv("RB_BREAK_TWO_WALK"); // (the env: diagnostics)
    for (int attempt = 0; attempt < 6; attempt++) {
        if (ws) (void)rb_dev_free(ctx, ws);
        if (d_rows) (void)rb_dev_free(ctx, d_rows);
        if (d_out) (void)rb_dev_free(ctx, d_out);
        ws = nullptr, d_rows = nullptr, d_out = nullptr;
        rc = rb_dev_alloc(ctx, rb_plan_workspace_bytes(plan, rows_cap), &ws);
        if (!rc) rc = rb_dev_alloc(ctx, (rows_cap + 1) * sizeof(rb_hit_row), (void **)&d_rows);
        if (!rc) rc = rb_dev_alloc(ctx, (out_cap + 4) * 4, (void **)&d_out);
        if (rc) break;
        rc = is_break ? rb_dev_break(ctx, plan, &b.v, d_norm, max_size, policy | (one_walk ? RB_BREAK_ONE_WALK : 0), ws, d_rows, rows_cap, d_out,
                                     out_cap, d_cnt)
                      : rb_dev_liftover(ctx, plan, &b.v, d_norm, policy, ws, d_rows, rows_cap, d_out, out_cap, d_cnt);
        if (rc) break;
        rc = rb_dev_download(ctx, &hc, d_cnt, sizeof hc);
        if (rc) break;
        if (one_walk && hc.redo_two_walk) { // something the one-walk path does not take: the same buffers, the two-walk path
            one_walk = false;
            attempt--;
            rc = RB_E_CAPACITY;
            continue;
        }
        if (!hc.overflow) break;
        rows_cap = std::max<uint64_t>(rows_cap, hc.n_hits + 16);
        out_cap = std::max<uint64_t>(out_cap * 2, hc.out_ops_needed + hc.out_ops_needed / 4 + 4096 + 4 * rows_cap);
        rc = RB_E_CAPACITY;
    }
    if (!rc && hc.overflow) rc = RB_E_CAPACITY;
    rb_lap("plan + clip kernels", tl);
    // ---- rows to the host, clip items back to the device, text out ----
    const uint64_t nr = hc.n_hits;
    std::vector<uint32_t> desc;
    uint64_t *d_first = nullptr, *d_toff = nullptr;
    uint32_t *d_cnt3 = nullptr; // count | first_len | last_len, three arrays of nr
    uint8_t *d_rtext = nullptr;
    if (!rc) {
        *rows = (rb_hit_row *)malloc((size_t)(nr + 1) * sizeof(rb_hit_row));
        *row_text_off = (uint64_t *)malloc((size_t)(nr + 2) * 8);
        if (!*rows || !*row_text_off) rc = fail(ctx, RB_E_NOMEM, "malloc(rows)");
    }
    if (!rc && nr) rc = rb_dev_download(ctx, *rows, d_rows, (size_t)nr * sizeof(rb_hit_row));
    if (!rc && nr) {
        desc.resize((size_t)nr * 4);
        rc = rb_dev_download(ctx, desc.data(), d_out, (size_t)nr * 16); // descriptor of row k at out_ops[4k]
    }
    rb_lap("rows D2H", tl);
    if (!rc) {
        std::vector<uint64_t> first((size_t)nr + 1, 0);
        std::vector<uint32_t> c3((size_t)nr * 3 + 1, 0);
        for (uint64_t k = 0; k < nr; k++) {
            const rb_hit_row &h = (*rows)[k];
            if (h.status != RB_ST_OK) continue;
            if (h.flags & RB_HIT_DESCRIPTOR) {
                const uint32_t *d = &desc[(size_t)k * 4];
                first[k] = op_off[h.rec] + d[0];
                c3[k] = d[1];
                c3[nr + k] = d[2];
                c3[2 * nr + k] = d[3];
            } else { // copied by the generic kernel
                first[k] = h.out_off | (1ull << 63);
                c3[k] = h.out_n;
            }
        }
        rc = b.alloc((size_t)nr + 1, &d_first);
        if (!rc) rc = b.alloc((size_t)nr * 3 + 1, &d_cnt3);
        if (!rc) rc = b.alloc((size_t)nr + 2, &d_toff);
        void *d_scr2 = nullptr;
        if (!rc) rc = b.alloc(rb_text_scratch_bytes(nr), (uint8_t **)&d_scr2);
        if (!rc && nr) rc = rb_dev_upload(ctx, d_first, first.data(), (size_t)nr * 8);
        if (!rc && nr) rc = rb_dev_upload(ctx, d_cnt3, c3.data(), (size_t)nr * 12);
        if (!rc) rc = rb_ctx_sync(ctx);
        if (!rc) rc = format_cigars_impl(ctx, d_ops, d_out, nr, d_first, d_cnt3, d_cnt3 + nr, d_cnt3 + 2 * nr, d_toff, nullptr, 0, d_scr2, false, true);
        if (!rc) rc = rb_dev_download(ctx, *row_text_off, d_toff, ((size_t)nr + 1) * 8);
        const uint64_t bytes = rc ? 0 : (*row_text_off)[nr];
        if (!rc) rc = b.alloc((size_t)bytes + 16, &d_rtext);
        if (!rc) rc = format_cigars_impl(ctx, d_ops, d_out, nr, d_first, d_cnt3, d_cnt3 + nr, d_cnt3 + 2 * nr, d_toff, d_rtext, bytes, d_scr2, true, true);
        if (!rc) {
            *row_text = (uint8_t *)malloc((size_t)bytes + 16);
            if (!*row_text) rc = fail(ctx, RB_E_NOMEM, "malloc(%llu text bytes)", (unsigned long long)bytes);
            else rb_advise_huge(*row_text, (size_t)bytes);
        }
        if (!rc && bytes) rc = rb_dev_download(ctx, *row_text, d_rtext, (size_t)bytes);
        if (!rc && bytes && getenv("RB_DEBUG_TEXT_CHECK")) { // diagnostics: CIGAR text holds no NUL byte
            const uint8_t *z = (const uint8_t *)memchr(*row_text, 0, (size_t)bytes);
            if (z) {
                const uint64_t at = (uint64_t)(z - *row_text);
                uint64_t row = 0;
                while (row + 1 < nr && (*row_text_off)[row + 1] <= at) row++;
                fprintf(stderr, "[rb text check] NUL at byte %llu of %llu (mod 32 MB: %llu, mod 16: %llu), row %llu of %llu, row text [%llu, %llu), item first %llu count %u\n",
                        (unsigned long long)at, (unsigned long long)bytes, (unsigned long long)(at % RB_PIN_CHUNK), (unsigned long long)(at & 15),
                        (unsigned long long)row, (unsigned long long)nr, (unsigned long long)(*row_text_off)[row], (unsigned long long)(*row_text_off)[row + 1],
                        (unsigned long long)0, 0u);
                std::vector<uint8_t> again((size_t)bytes);
                (void)hipMemcpy(again.data(), d_rtext, (size_t)bytes, hipMemcpyDeviceToHost);
                fprintf(stderr, "[rb text check] plain hipMemcpy of the same buffer: byte is %u (device %s)\n", again[(size_t)at], again[(size_t)at] ? "has the text: the staged download lost it" : "holds the NUL too: the kernel did not write it");
            }
        }
        rb_lap("format cigars + text D2H", tl);
    }
    if (!rc) {
        *n_rows = nr;
        if (counters) *counters = hc;
    } else {
        free(*rows);
        free(*row_text_off);
        free(*row_text);
        *rows = nullptr, *row_text_off = nullptr, *row_text = nullptr;
    }
    if (ws) (void)rb_dev_free(ctx, ws);
    if (d_rows) (void)rb_dev_free(ctx, d_rows);
    if (d_out) (void)rb_dev_free(ctx, d_out);
    rb_plan_destroy(plan);
    return rc;
}

extern "C" int rb_host_liftover_text(rb_ctx *ctx, uint64_t n_rec, const uint8_t *text, uint64_t text_bytes, const uint64_t *cig_off,
                                     const uint64_t *cig_end, const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st,
                                     const uint64_t *q_en, const uint8_t *strand, const uint32_t *contig, uint64_t n_win,
                                     const uint32_t *w_contig, const uint64_t *w_st, const uint64_t *w_en, int policy, uint8_t *cig_status,
                                     rb_reduce_row *reduce_out, rb_norm_row *norm_out, rb_hit_row **rows, uint64_t *n_rows,
                                     uint64_t **row_text_off, uint8_t **row_text, rb_counters *counters) {
    return host_lift_text(ctx, false, 0, false, n_rec, text, text_bytes, cig_off, cig_end, t_st, t_en, q_st, q_en, strand, contig, n_win,
                          w_contig, w_st, w_en, policy, cig_status, reduce_out, norm_out, rows, n_rows, row_text_off, row_text, counters);
}
extern "C" int rb_host_break_text(rb_ctx *ctx, uint64_t n_rec, const uint8_t *text, uint64_t text_bytes, const uint64_t *cig_off,
                                  const uint64_t *cig_end, const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st,
                                  const uint64_t *q_en, const uint8_t *strand, uint32_t max_size, int policy, uint8_t *cig_status,
                                  rb_reduce_row *reduce_out, rb_norm_row *norm_out, rb_hit_row **rows, uint64_t *n_rows,
                                  uint64_t **row_text_off, uint8_t **row_text, rb_counters *counters) {
    return host_lift_text(ctx, true, max_size, false, n_rec, text, text_bytes, cig_off, cig_end, t_st, t_en, q_st, q_en, strand, nullptr, 0,
                          nullptr, nullptr, nullptr, policy, cig_status, reduce_out, norm_out, rows, n_rows, row_text_off, row_text, counters);
}
extern "C" int rb_host_scan_text(rb_ctx *ctx, uint64_t n_rec, const uint8_t *text, uint64_t text_bytes, const uint64_t *cig_off,
                                 const uint64_t *cig_end, const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st,
                                 const uint64_t *q_en, const uint8_t *strand, uint8_t *cig_status, rb_reduce_row *reduce_out,
                                 rb_norm_row *norm_out) {
    return host_lift_text(ctx, false, 0, true, n_rec, text, text_bytes, cig_off, cig_end, t_st, t_en, q_st, q_en, strand, nullptr, 0, nullptr,
                          nullptr, nullptr, 0, cig_status, reduce_out, norm_out, nullptr, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int rb_host_swap(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint8_t *strand,
                            uint32_t *out_ops) {
    if (!ctx) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBatch b(ctx);
    std::vector<uint64_t> z(n_rec, 0);
    int rc = b.load(n_rec, ops, op_off, z.data(), z.data(), z.data(), z.data(), strand, nullptr);
    if (rc) return rc;
    uint32_t *d_out = nullptr;
    const uint64_t n_ops = n_rec ? op_off[n_rec] : 0;
    if ((rc = b.alloc((size_t)n_ops + 4, &d_out))) return rc;
    if ((rc = rb_dev_swap(ctx, &b.v, d_out))) return rc;
    return rb_dev_download(ctx, out_ops, d_out, (size_t)n_ops * 4);
}

extern "C" int rb_host_overlap_split(rb_ctx *ctx, uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                                     const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                                     uint64_t n_pairs, const uint32_t *left, const uint32_t *right, int match_score, int diff_score,
                                     int indel_score, int policy, rb_pair_row *rows, uint32_t **out_ops, uint64_t *n_out) {
    if (!ctx || !out_ops || !n_out || (n_pairs && !rows)) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    *out_ops = nullptr;
    *n_out = 0;
    DevBatch b(ctx);
    int rc = b.load(n_rec, ops, op_off, t_st, t_en, q_st, q_en, strand, nullptr);
    if (rc) return rc;
    rb_norm_row *d_norm = nullptr;
    if ((rc = b.alloc(n_rec, &d_norm))) return rc;
    if ((rc = rb_dev_scan_records(ctx, &b.v, nullptr, d_norm))) return rc;
    std::vector<uint64_t> poff(n_pairs + 1, 0);
    for (uint64_t i = 0; i < n_pairs; i++) {
        if (left[i] >= n_rec || right[i] >= n_rec) return fail(ctx, RB_E_INVALID, "pair %llu references a record outside the batch", (unsigned long long)i);
        poff[i + 1] = poff[i] + (op_off[left[i] + 1] - op_off[left[i]]) + (op_off[right[i] + 1] - op_off[right[i]]);
    }
    const uint32_t *d_left = nullptr, *d_right = nullptr;
    const uint64_t *d_poff = nullptr;
    rb_pair_row *d_rows = nullptr;
    uint32_t *d_out = nullptr;
    if ((rc = b.up(left, (size_t)n_pairs, &d_left))) return rc;
    if ((rc = b.up(right, (size_t)n_pairs, &d_right))) return rc;
    if ((rc = b.up(poff.data(), (size_t)n_pairs + 1, &d_poff))) return rc;
    if ((rc = b.alloc((size_t)n_pairs + 1, &d_rows))) return rc;
    if ((rc = b.alloc((size_t)poff[n_pairs] + 4, &d_out))) return rc;
    if ((rc = rb_dev_overlap_split(ctx, &b.v, d_norm, n_pairs, d_left, d_right, d_poff, match_score, diff_score, indel_score, policy,
                                   d_rows, d_out)))
        return rc;
    if (n_pairs && (rc = rb_dev_download(ctx, rows, d_rows, (size_t)n_pairs * sizeof(rb_pair_row)))) return rc;
    std::vector<uint32_t> raw((size_t)poff[n_pairs] + 4);
    if (poff[n_pairs] && (rc = rb_dev_download(ctx, raw.data(), d_out, (size_t)poff[n_pairs] * 4))) return rc;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n_pairs; i++)
        if (rows[i].status == RB_ST_OK) total += rows[i].out_n[0] + rows[i].out_n[1];
    *out_ops = (uint32_t *)malloc((size_t)(total + 1) * 4);
    if (!*out_ops) return fail(ctx, RB_E_NOMEM, "malloc(%llu ops)", (unsigned long long)total);
    // dense offsets first, then the copies on all host threads
    std::vector<uint64_t> src(2 * (size_t)n_pairs);
    uint64_t o = 0;
    for (uint64_t i = 0; i < n_pairs; i++) {
        for (int s = 0; s < 2; s++) {
            src[2 * i + s] = rows[i].out_off[s];
            if (rows[i].status == RB_ST_OK) {
                rows[i].out_off[s] = o;
                o += rows[i].out_n[s];
            } else {
                rows[i].out_off[s] = 0;
                rows[i].out_n[s] = 0;
            }
        }
    }
    {
        const unsigned T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::thread::hardware_concurrency(), n_pairs / 1024));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t] {
                for (uint64_t i = n_pairs * t / T; i < n_pairs * (t + 1) / T; i++)
                    for (int s = 0; s < 2; s++)
                        if (rows[i].out_n[s]) memcpy(*out_ops + rows[i].out_off[s], raw.data() + src[2 * i + s], (size_t)rows[i].out_n[s] * 4);
            });
        for (auto &x : th) x.join();
    }
    *n_out = o;
    return rb_ctx_sync(ctx);
}

// ---- synthetic workload --------------------------------------------------------------------------
#include "synth.h"
extern "C" uint32_t rb_synth_n_ops(uint64_t seed, uint64_t record, uint32_t lo, uint32_t hi) { return rb_synth_n_ops_impl(seed, record, lo, hi); }
extern "C" uint32_t rb_synth_n_ops_lognormal(uint64_t seed, uint64_t record) { return rb_synth_n_ops_lognormal_impl(seed, record); }
extern "C" void rb_synth_fill_ops_host(uint64_t seed, uint64_t first_record, uint64_t n_rec, const uint64_t *op_off, uint32_t *ops) {
    for (uint64_t r = 0; r < n_rec; r++) {
        const uint64_t n = op_off[r + 1] - op_off[r];
        for (uint64_t j = 0; j < n; j++) ops[op_off[r] + j] = rb_synth_op(seed, first_record + r, j);
    }
}
extern "C" int rb_dev_synth_fill_ops(rb_ctx *ctx, uint64_t seed, uint64_t first_record, uint64_t n_rec, const uint64_t *op_off_dev,
                                     uint32_t *ops_dev) {
    if (!ctx) return RB_E_INVALID;
    HIPCHK(ctx, rb_launch_synth(seed, first_record, n_rec, op_off_dev, ops_dev, ctx->stream));
    return RB_OK;
}

// ---- verification aid: digest of rows + clipped cigars ----------------------------------------------
struct rb_digest_params {
    const uint32_t *ops;
    const uint64_t *op_off;
    const rb_hit_row *rows;
    uint64_t n_rows;
    const uint32_t *out_ops;
    uint64_t row_base, rec_base;
    unsigned long long *digest;
};
extern "C" hipError_t rb_launch_digest_rows(const rb_digest_params *p, hipStream_t stream);
extern "C" int rb_dev_digest_rows(rb_ctx *ctx, const rb_batch_view *batch, const rb_hit_row *rows, uint64_t n_rows, const uint32_t *out_ops,
                                  uint64_t row_base, uint64_t rec_base, uint64_t *digest) {
    if (!ctx || !batch || !digest || (n_rows && (!rows || !out_ops))) return RB_E_INVALID;
    rb_digest_params p;
    p.ops = batch->ops;
    p.op_off = batch->op_off;
    p.rows = rows;
    p.n_rows = n_rows;
    p.out_ops = out_ops;
    p.row_base = row_base;
    p.rec_base = rec_base;
    p.digest = (unsigned long long *)digest;
    HIPCHK(ctx, rb_launch_digest_rows(&p, ctx->stream));
    return RB_OK;
}

// ---- the box: what this GPU moves at the clip kernel's memory mix, and the clock it holds meanwhile (diagnostics for bench.py) ----
extern "C" hipError_t rb_launch_box_probe(const void *src, void *d0, void *d1, uint64_t n_stretch, uint32_t *stamps, int scatter, hipStream_t stream);
extern "C" int rb_dev_box_probe(rb_ctx *ctx, const void *src, uint64_t src_bytes, void *dst0, void *dst1, int reps, int scatter, double *ms_out, double *mhz_out) {
    if (!ctx || !src || !dst0 || !dst1 || !ms_out || reps < 1) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t n_stretch = src_bytes / (10 * 2048);
    if (n_stretch == 0) return RB_E_INVALID;
    uint32_t *stamps = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&stamps, 64));
    hipEvent_t a = nullptr, b = nullptr;
    int rc = RB_OK;
    auto chk = [&](hipError_t e) {
        if (e != hipSuccess && rc == RB_OK) rc = fail(ctx, RB_E_HIP, "rb_dev_box_probe: %s", hipGetErrorString(e));
    };
    chk(hipEventCreate(&a));
    chk(hipEventCreate(&b));
    chk(rb_launch_box_probe(src, dst0, dst1, n_stretch, stamps, scatter, ctx->stream)); // (untimed: first touch)
    chk(rb_fill_async(stamps, 0, 64, ctx->stream));
    chk(hipEventRecord(a, ctx->stream));
    for (int i = 0; i < reps && rc == RB_OK; i++) chk(rb_launch_box_probe(src, dst0, dst1, n_stretch, stamps, scatter, ctx->stream));
    chk(hipEventRecord(b, ctx->stream));
    chk(hipEventSynchronize(b));
    float ms = 0;
    if (rc == RB_OK) chk(hipEventElapsedTime(&ms, a, b));
    uint32_t h[4] = {0, 0, 0, 0};
    if (rc == RB_OK) chk(hipMemcpy(h, stamps, 16, hipMemcpyDeviceToHost));
    *ms_out = ms / reps;
    if (mhz_out) *mhz_out = h[1] ? (double)h[0] * 64.0 / (double)h[1] * 100.0 : 0.0; // cycles per 10 ns tick x 100 MHz
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    (void)hipFree(stamps);
    return rc;
}

// ---- a buffer that will be WRITTEN at the clip kernel's rate, placed by measurement ----
// On MI355X the time of a launch that streams its output into a buffer depends on WHICH physical pages the buffer has: the same
// process, the same launch, the output arena allocated four times over -- 9.10, 9.17, 9.25 and 11.04 ms (profiles/r04_alloc_summary.md).
// Reads do not care; writes do, and the library's store sweep (the box probe's write side alone, over the whole buffer) sees most of
// it.  rb_dev_alloc_placed allocates up to `tries` candidates of `bytes` (fewer when the device has not the room: a candidate is only
// taken while twice its size stays free), times the sweep on each, keeps the fastest and gives the others back.  sweep_ms[i] (may be
// NULL, room for `tries`) = what candidate i took, -1 where none was made; *kept = the index of the one returned.
// rb_dev_alloc_placed_by: the same with the caller's own measure -- score(candidate, user) is called once per candidate, with nothing
// of the library locked, and may launch on the context (bench.py: the clip kernel itself on its batch, which is what the placement
// is for; the store sweep ranks a 9.9 ms arena behind a 10.1 ms one now and then).  The lowest score wins; a negative score ends the
// search with RB_E_INVALID.
extern "C" int rb_dev_alloc_placed_by(rb_ctx *ctx, uint64_t bytes, int tries, double (*score)(void *candidate, void *user), void *user, void **out,
                                      double *scores, int *kept);
namespace {
struct sweep_arg { rb_ctx *ctx; uint64_t half; int rc; };
double sweep_score(void *q, void *user) {
    sweep_arg *a = (sweep_arg *)user;
    double t0 = 0, t1 = 0;
    char *p0 = (char *)q, *p1 = (char *)q + a->half;
    int r = rb_dev_box_probe(a->ctx, p0, a->half, p0, p1, 3, 1 | 8, &t0, nullptr); // (8: no loads -- src is not read)
    if (r == RB_OK) r = rb_dev_box_probe(a->ctx, p1, a->half, p1, p0, 3, 1 | 8, &t1, nullptr);
    if (r != RB_OK) {
        a->rc = r;
        return -1.0;
    }
    return t0 + t1;
}
} // namespace
extern "C" int rb_dev_alloc_placed(rb_ctx *ctx, uint64_t bytes, int tries, void **out, double *sweep_ms, int *kept) {
    if (!ctx || !out || tries < 1) return RB_E_INVALID;
    sweep_arg a{ctx, bytes / 2 / 20480 * 20480, RB_OK}; // (the sweep writes 20 KiB stretches: the buffer as two halves, each one fully written once)
    const bool sweep = a.half >= 20480 && tries > 1;
    const int rc = rb_dev_alloc_placed_by(ctx, bytes, sweep ? tries : 1, sweep ? sweep_score : nullptr, &a, out, sweep_ms, kept);
    for (int i = 1; sweep_ms && !sweep && i < tries; i++) sweep_ms[i] = -1.0;
    return (rc != RB_OK && a.rc != RB_OK) ? a.rc : rc;
}
extern "C" int rb_dev_alloc_placed_by(rb_ctx *ctx, uint64_t bytes, int tries, double (*score)(void *candidate, void *user), void *user, void **out,
                                      double *sweep_ms, int *kept) {
    if (!ctx || !out || tries < 1) return RB_E_INVALID;
    *out = nullptr;
    if (kept) *kept = -1;
    for (int i = 0; sweep_ms && i < tries; i++) sweep_ms[i] = -1.0;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (tries > 1) { // a buffer of this size that this context placed before and released (rb_dev_release): as it is, nothing is measured again
        const size_t want = bytes ? ((bytes + 255) & ~(size_t)255) : 256;
        for (size_t k = 0; k < ctx->cache.size(); k++)
            if (ctx->cache[k].bytes == want && ctx->cache[k].placed) {
                *out = ctx->cache[k].p;
                ctx->cache_bytes -= want;
                ctx->cache.erase(ctx->cache.begin() + (long)k);
                return RB_OK; // (*kept = -1, every score -1: no candidate was made)
            }
    }
    std::vector<void *> cand;
    std::vector<double> ms;
    int rc = RB_OK, best = -1;
    for (int i = 0; i < tries && rc == RB_OK; i++) {
        if (i > 0) { // (never the last of the memory: the winner's neighbours -- workspace, rows -- still have to fit)
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || (uint64_t)fr < 2 * bytes + ((uint64_t)8 << 30)) break;
        }
        void *q = nullptr;
        const auto t_a = std::chrono::steady_clock::now();
        const int r1 = rb_dev_alloc(ctx, bytes, &q);
        if (r1 != RB_OK) {
            if (i == 0) rc = r1;
            break;
        }
        if (getenv("RB_ALLOC_LOG"))
            fprintf(stderr, "[rb_dev_alloc_placed] candidate %d: %.2f s to allocate\n", i, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count());
        cand.push_back(q);
        double t = 0;
        if (score && tries > 1) {
            t = score(q, user);
            if (!(t >= 0)) {
                rc = fail(ctx, RB_E_INVALID, "rb_dev_alloc_placed_by: the score of candidate %d is negative", i);
                break;
            }
        }
        ms.push_back(t);
        if (sweep_ms) sweep_ms[i] = t;
        if (best < 0 || t < ms[(size_t)best]) best = i;
    }
    const auto t_f = std::chrono::steady_clock::now();
    for (size_t i = 0; i < cand.size(); i++)
        if ((int)i != best || rc != RB_OK) {
            const int r3 = rb_dev_free(ctx, cand[i]);
            if (r3 != RB_OK && rc == RB_OK) rc = r3;
        }
    if (getenv("RB_ALLOC_LOG"))
        fprintf(stderr, "[rb_dev_alloc_placed] %zu candidates, kept %d, %.2f s to give the others back\n", cand.size(), best,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_f).count());
    if (rc != RB_OK) return rc;
    *out = cand[(size_t)best];
    if (kept) *kept = best;
    if (cand.size() > 1) { // chosen by measurement: the context's cache keeps that in mind (rb_dev_release)
        std::lock_guard<std::mutex> lk(g_vmm_mu);
        auto it = g_vmm.find(*out);
        if (it != g_vmm.end()) it->second.placed = true;
    }
    return RB_OK;
}

// ---- nucfreq ----------------------------------------------------------------------------------------------------------
namespace {
struct nf_layout {
    size_t end_key, rd_end, tile_off, blk, tile_lo, tile_hi, drop_off, deep_list, drop_pool, tdesc, wide_list, total;
    uint64_t max_tiles, drop_words;
};
nf_layout nf_ws_layout(uint64_t n_reads, uint64_t n_regions, uint64_t n_positions) {
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    nf_layout L;
    L.max_tiles = n_positions / rb_nf_tile_positions() + n_regions;
    size_t o = 0;
    L.end_key = o, o = up(o + ((size_t)n_reads + 1) * 8);
    L.rd_end = o, o = up(o + ((size_t)n_reads + 1) * 48);
    L.tile_off = o, o = up(o + ((size_t)n_regions + 2) * 8);
    L.blk = o, o = up(o + (std::max(rb_nf_scan_blocks(n_reads), rb_scan_block_sums_count(n_regions)) + 2) * 8);
    L.tile_lo = o, o = up(o + ((size_t)L.max_tiles + 1) * 8);
    L.tile_hi = o, o = up(o + ((size_t)L.max_tiles + 1) * 8);
    // the depth cap's bookkeeping: a bitmap of dropped reads per region that can reach the cap, from a pool of 64 bits per read
    L.drop_off = o, o = up(o + ((size_t)n_regions + 1) * 8);
    L.deep_list = o, o = up(o + ((size_t)n_regions + 2) * 4);
    L.drop_words = n_reads + 1024;
    L.drop_pool = o, o = up(o + ((size_t)L.drop_words + 1) * 8); // (word 0: the pool's cursor)
    L.tdesc = o, o = up(o + ((size_t)L.max_tiles + 1) * 64);     // one 64-byte descriptor per tile (rb_k_nf_tile_desc)
    L.wide_list = o, o = up(o + ((size_t)L.max_tiles + 2) * 4);  // the tiles of the 16-bit build: count, then indices
    L.total = o;
    return L;
}
} // namespace

extern "C" size_t rb_nucfreq_workspace_bytes(uint64_t n_reads, uint64_t n_regions, uint64_t n_positions) {
    return nf_ws_layout(n_reads, n_regions, n_positions).total;
}

extern "C" int rb_dev_nucfreq(rb_ctx *ctx, const rb_reads_view *reads, uint64_t n_regions, const int32_t *rg_tid, const uint64_t *rg_st,
                              const uint64_t *rg_en, const uint64_t *out_off, uint64_t n_positions, uint32_t *counts, uint32_t *read_status,
                              rb_nucfreq_counters *counters, void *ws, size_t ws_bytes) {
    if (!ctx || !reads || !counters || (n_regions && (!rg_tid || !rg_st || !rg_en || !out_off)) || (n_positions && !counts)) return RB_E_INVALID;
    if (reads->n_reads && (!reads->ops || !reads->op_off || !reads->seq || !reads->seq_off || !reads->l_seq || !reads->tid || !reads->pos ||
                           !reads->flag || !read_status))
        return RB_E_INVALID;
    const nf_layout L = nf_ws_layout(reads->n_reads, n_regions, n_positions);
    if (!ws || ((uintptr_t)ws & 255) || ws_bytes < L.total) return fail(ctx, RB_E_INVALID, "nucfreq workspace: %zu bytes, 256-byte aligned, needed", L.total);
    if (L.max_tiles >= 0x7FFFFFFFull) return fail(ctx, RB_E_INVALID, "nucfreq: too many positions for one call");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, rb_fill_async(counters, 0, sizeof(rb_nucfreq_counters), ctx->stream));
    uint8_t *w = (uint8_t *)ws;
    rb_nf_params p{};
    p.n_reads = reads->n_reads;
    p.ops = reads->ops, p.op_off = reads->op_off, p.seq = reads->seq, p.seq_off = reads->seq_off, p.l_seq = reads->l_seq;
    p.tid = reads->tid, p.pos = reads->pos, p.flag = reads->flag;
    p.n_regions = n_regions;
    p.rg_tid = rg_tid, p.rg_st = rg_st, p.rg_en = rg_en, p.out_off = out_off;
    p.counts = counts, p.read_status = read_status, p.counters = counters;
    p.end_key = (uint64_t *)(w + L.end_key), p.hd = (void *)(w + L.rd_end), p.tile_off = (uint64_t *)(w + L.tile_off);
    p.blk = (uint64_t *)(w + L.blk), p.tile_lo = (uint64_t *)(w + L.tile_lo), p.tile_hi = (uint64_t *)(w + L.tile_hi);
    p.max_tiles = L.max_tiles;
    p.drop_off = (uint64_t *)(w + L.drop_off), p.deep_list = (uint32_t *)(w + L.deep_list);
    p.drop_bits = (uint64_t *)(w + L.drop_pool) + 1, p.drop_words = L.drop_words;
    static const bool all_atomic = getenv("RB_DEBUG_NF_ATOMIC") != nullptr; // (diagnostic: every tile through the LDS-atomic kernel)
    p.flags = all_atomic ? 1u : 0u;
    p.tdesc = (void *)(w + L.tdesc);
    p.wide_list = (uint32_t *)(w + L.wide_list);
    HIPCHK(ctx, rb_fill_async(p.drop_bits - 1, 0, 8, ctx->stream));
    HIPCHK(ctx, rb_fill_async(p.deep_list + n_regions, 0, 8, ctx->stream)); // (how many deep regions; can the cap be reached at all)
    HIPCHK(ctx, rb_launch_nucfreq(&p, ctx->stream));
    return RB_OK;
}

extern "C" int rb_host_nucfreq(rb_ctx *ctx, const rb_reads_view *reads, uint64_t n_regions, const int32_t *rg_tid, const uint64_t *rg_st,
                               const uint64_t *rg_en, uint32_t *counts, uint32_t *read_status, rb_nucfreq_counters *counters) {
    if (!ctx || !reads || !counters || (n_regions && (!rg_tid || !rg_st || !rg_en))) return RB_E_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = reads->n_reads;
    std::vector<uint64_t> out_off(n_regions + 1, 0);
    for (uint64_t r = 0; r < n_regions; r++) {
        if (rg_en[r] < rg_st[r]) return fail(ctx, RB_E_INVALID, "region %llu ends before it starts", (unsigned long long)r);
        out_off[r + 1] = out_off[r] + (rg_en[r] - rg_st[r]);
    }
    const uint64_t n_pos = out_off[n_regions];
    if (n_pos && !counts) return RB_E_INVALID;
    DevBatch b(ctx);
    rb_reads_view d{};
    d.n_reads = n;
    int rc;
    std::vector<uint64_t> opo, sqo; // (function scope: the asynchronous uploads below read them until the stream is synchronised)
    if (n) {
        // only the part of ops / seq these reads use goes up, offsets rebased
        const uint64_t op0 = reads->op_off[0], op1 = reads->op_off[n];
        uint64_t s0 = ~0ull, s1 = 0;
        for (uint64_t i = 0; i < n; i++) {
            const uint64_t a = reads->seq_off[i], e = a + ((uint64_t)reads->l_seq[i] + 1) / 2;
            s0 = std::min(s0, a), s1 = std::max(s1, e);
        }
        opo.resize(n + 1), sqo.resize(n);
        for (uint64_t i = 0; i <= n; i++) opo[i] = reads->op_off[i] - op0;
        for (uint64_t i = 0; i < n; i++) sqo[i] = reads->seq_off[i] - s0;
        if ((rc = b.up(reads->ops + op0, (size_t)(op1 - op0), &d.ops))) return rc;
        if ((rc = b.up(opo.data(), (size_t)n + 1, &d.op_off))) return rc;
        {   // (32 readable bytes behind the last read, as the kernel asks)
            uint8_t *dsq = nullptr;
            if ((rc = b.alloc((size_t)(s1 - s0) + 32, &dsq))) return rc;
            if (s1 > s0 && (rc = rb_dev_upload(ctx, dsq, reads->seq + s0, (size_t)(s1 - s0)))) return rc;
            d.seq = dsq;
        }
        if ((rc = b.up(sqo.data(), (size_t)n, &d.seq_off))) return rc;
        if ((rc = b.up(reads->l_seq, (size_t)n, &d.l_seq))) return rc;
        if ((rc = b.up(reads->tid, (size_t)n, &d.tid))) return rc;
        if ((rc = b.up(reads->pos, (size_t)n, &d.pos))) return rc;
        if ((rc = b.up(reads->flag, (size_t)n, &d.flag))) return rc;
    }
    const int32_t *d_tid = nullptr;
    const uint64_t *d_st = nullptr, *d_en = nullptr, *d_off = nullptr;
    if ((rc = b.up(rg_tid, (size_t)n_regions, &d_tid))) return rc;
    if ((rc = b.up(rg_st, (size_t)n_regions, &d_st))) return rc;
    if ((rc = b.up(rg_en, (size_t)n_regions, &d_en))) return rc;
    if ((rc = b.up(out_off.data(), (size_t)n_regions + 1, &d_off))) return rc;
    uint32_t *d_counts = nullptr, *d_status = nullptr;
    rb_nucfreq_counters *d_ctr = nullptr;
    uint8_t *d_ws = nullptr;
    if ((rc = b.alloc((size_t)n_pos * 4 + 4, &d_counts))) return rc;
    if ((rc = b.alloc((size_t)n + 1, &d_status))) return rc;
    if ((rc = b.alloc(1, &d_ctr))) return rc;
    const size_t wsb = rb_nucfreq_workspace_bytes(n, n_regions, n_pos);
    if ((rc = b.alloc(wsb, &d_ws))) return rc;
    if ((rc = rb_dev_nucfreq(ctx, &d, n_regions, d_tid, d_st, d_en, d_off, n_pos, d_counts, d_status, d_ctr, d_ws, wsb))) return rc;
    if ((rc = rb_dev_download(ctx, counters, d_ctr, sizeof(*counters)))) return rc;
    if (read_status && n && (rc = rb_dev_download(ctx, read_status, d_status, (size_t)n * 4))) return rc;
    if (n_pos && (rc = rb_dev_download(ctx, counts, d_counts, (size_t)n_pos * 16))) return rc;
    if (counters->unsorted) return fail(ctx, RB_E_INVALID, "nucfreq: the reads are not sorted by (tid, pos)");
    if (counters->cap_overflow)
        return fail(ctx, RB_E_INVALID, "nucfreq: the bookkeeping of htslib's cap of %u buffered reads does not fit (depth %llu)", RB_NF_DEPTH_CAP,
                    (unsigned long long)counters->max_depth);
    if (counters->max_depth > 65535) return fail(ctx, RB_E_INVALID, "nucfreq: depth %llu does not fit the 16-bit counters", (unsigned long long)counters->max_depth);
    return RB_OK;
}
