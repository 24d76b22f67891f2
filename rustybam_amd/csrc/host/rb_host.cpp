// rb_host.cpp -- see rb_host.hpp.  No CIGAR is walked on the CPU here: counting, clipping, splitting and
// swapping all go through the C ABI to the device.
#include "rb_host.hpp"

#include <zlib.h>

#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <exception>
#include <memory>
#include <mutex>
#include <condition_variable>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <functional>
#include <map>
#include <numeric>
#include <unordered_map>
#include <unordered_set>
#include <thread>

namespace rb {

// static chunking over host threads: text decode / encode is what bounds end-to-end throughput
static unsigned host_threads() {
    static const unsigned n = [] {
        const char *e = getenv("RB_THREADS");
        unsigned v = e ? (unsigned)atoi(e) : std::thread::hardware_concurrency();
        return v < 1 ? 1u : (v > 64 ? 64u : v);
    }();
    return n;
}
template <typename F>
static void parallel_chunks(size_t n, F fn) { // fn(chunk_index, lo, hi)
    const unsigned T = (unsigned)std::min<size_t>(host_threads(), n ? n : 1);
    if (T <= 1) {
        fn(0u, (size_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err(T);
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&, t] {
            try {
                fn(t, n * t / T, n * (t + 1) / T);
            } catch (...) {
                err[t] = std::current_exception();
            }
        });
    for (auto &x : th) x.join();
    for (auto &e : err)
        if (e) std::rethrow_exception(e); // the first chunk's failure = the first failing line in file order
}
// RB_TIMING=1: wall time of host phases on stderr
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void lap(const char *what, double &t) {
    static const bool on = getenv("RB_TIMING") != nullptr;
    const double n = now_s();
    if (on) fprintf(stderr, "[rb timing]   %-26s %.3f s\n", what, n - t);
    t = n;
}
unsigned parallel_chunk_count(size_t n) { return (unsigned)std::min<size_t>(host_threads(), n ? n : 1); }

static const char OPCH[] = "MIDNSHP=X";
// Display of packed words (impl Display for CigarString): a continuation word (RB_OP_CONT, rustybam_amd.h) is bits 28.. of the
// length of the op in front of it
template <typename Out>
static inline void append_ops(Out &o, const uint32_t *w, size_t n) {
    char nb[24];
    for (size_t i = 0; i < n; i++) {
        uint64_t len = w[i] >> 4;
        const uint32_t code = w[i] & 15u;
        if (code != RB_OP_CONT && i + 1 < n && (w[i + 1] & 15u) == RB_OP_CONT) len += (uint64_t)((w[++i] >> 4) & 15u) << 28;
        auto r = std::to_chars(nb, nb + sizeof nb, len);
        o.append(nb, r.ptr);
        o.push_back(code < 9u ? OPCH[code] : '?');
    }
}

Engine::Engine(int device) {
    int rc = rb_ctx_create(device, nullptr, &ctx_);
    if (rc != RB_OK) throw std::runtime_error("rustybam_amd: no usable MI355X (gfx950) device (rb_ctx_create = " + std::to_string(rc) + "); there is no CPU fallback");
}
Engine::~Engine() { rb_ctx_destroy(ctx_); }
void Engine::check(int rc, const char *what) const {
    if (rc != RB_OK) throw std::runtime_error(std::string(what) + " failed (" + std::to_string(rc) + "): " + rb_ctx_last_error(ctx_));
}

std::string cigar_to_string(const std::vector<uint32_t> &cigar) {
    std::string s;
    s.reserve(cigar.size() * 5);
    append_ops(s, cigar.data(), cigar.size());
    return s;
}

std::string PafRecord::to_string() const {
    std::string s;
    s.reserve(128 + cigar.size() * 5);
    s += q_name; s += '\t'; s += std::to_string(q_len); s += '\t'; s += std::to_string(q_st); s += '\t';
    s += std::to_string(q_en); s += '\t'; s += strand; s += '\t'; s += t_name; s += '\t'; s += std::to_string(t_len);
    s += '\t'; s += std::to_string(t_st); s += '\t'; s += std::to_string(t_en); s += '\t'; s += std::to_string(nmatch);
    s += '\t'; s += std::to_string(aln_len); s += '\t'; s += std::to_string(mapq); s += "\tid:Z:"; s += id; s += "\tcg:Z:";
    s += cigar_to_string(cigar);
    return s;
}

static bool parse_u64(const char *s, size_t n, uint64_t &out) { // Rust u64::from_str
    size_t i = 0;
    if (n == 0) return false;
    if (s[0] == '+') {
        i = 1;
        if (n == 1) return false;
    }
    uint64_t v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return false;
        const uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (UINT64_MAX - d) / 10) return false;
        v = v * 10 + d;
    }
    out = v;
    return true;
}

// rust-htslib CigarString::try_from as used at paf.rs:398-399; any violation is the .expect() panic
static void parse_cigar(const char *s, size_t n, std::vector<uint32_t> &out) {
    out.clear();
    size_t i = 0;
    while (i < n) {
        size_t j = i;
        uint64_t len = 0;
        while (j < n && s[j] >= '0' && s[j] <= '9') {
            len = len * 10 + (uint64_t)(s[j] - '0');
            if (len > 0xFFFFFFFFull) throw Panic("Unable to parse cigar string.");
            j++;
        }
        if (j == i || j >= n) throw Panic("Unable to parse cigar string.");
        const char *p = (const char *)memchr(OPCH, s[j], 9);
        if (!p) throw Panic("Unable to parse cigar string.");
        out.push_back(((uint32_t)(len & 0x0FFFFFFFull) << 4) | (uint32_t)(p - OPCH));
        if (len >> 28) out.push_back(((uint32_t)(len >> 28) << 4) | (uint32_t)RB_OP_CONT); // (rustybam_amd.h: lengths of 2^28 and more)
        i = j + 1;
    }
}

int paf_record_new(const std::string &line, PafRecord &out) {
    std::vector<std::pair<const char *, size_t>> t;
    const char *p = line.c_str();
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f'; };
    while (*p) {
        while (*p && ws(*p)) p++;
        if (!*p) break;
        const char *q = p;
        while (*q && !ws(*q)) q++;
        t.emplace_back(p, (size_t)(q - p));
        p = q;
    }
    if (t.size() < 12) throw Panic("assertion failed: t.len() >= 12"); // paf.rs:381
    std::vector<uint32_t> cigar;
    for (size_t k = 12; k < t.size(); k++) { // PAF_TAG "(..):(.):(.*)", leftmost match (paf.rs:21, :387-390)
        const char *s = t[k].first;
        const size_t n = t[k].second;
        size_t m = (size_t)-1;
        for (size_t i = 0; i + 5 <= n; i++)
            if (s[i + 2] == ':' && s[i + 4] == ':') {
                m = i;
                break;
            }
        if (m == (size_t)-1) throw Panic("assertion failed: PAF_TAG.is_match(token)");
        if (s[m] == 'c' && s[m + 1] == 'g' && cigar.empty()) parse_cigar(s + m + 5, n - (m + 5), cigar); // paf.rs:395
    }
    uint64_t v[12] = {0};
    static const int numeric[] = {1, 2, 3, 6, 7, 8, 9, 10, 11};
    for (int c : numeric)
        if (!parse_u64(t[c].first, t[c].second, v[c])) return 1;
    if (t[4].second != 1) return 1;
    out = PafRecord();
    out.q_name.assign(t[0].first, t[0].second);
    out.q_len = v[1], out.q_st = v[2], out.q_en = v[3];
    out.strand = t[4].first[0];
    out.t_name.assign(t[5].first, t[5].second);
    out.t_len = v[6], out.t_st = v[7], out.t_en = v[8], out.nmatch = v[9], out.aln_len = v[10], out.mapq = v[11];
    out.cigar.swap(cigar);
    return 0;
}

static bool gz_getline(gzFile f, std::string &line) {
    line.clear();
    char buf[1 << 16];
    for (;;) {
        if (!gzgets(f, buf, sizeof buf)) return !line.empty();
        line += buf;
        if (!line.empty() && line.back() == '\n') break;
    }
    if (!line.empty() && line.back() == '\n') line.pop_back();
    if (!line.empty() && line.back() == '\r') line.pop_back();
    return true;
}

std::vector<Region> parse_bed(const std::string &filename) {
    gzFile f = gzopen(filename.c_str(), "rb");
    if (!f) throw Panic("unable to open bam file."); // bed.rs:175 (sic)
    std::vector<Region> out;
    std::string line;
    while (gz_getline(f, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::vector<std::string> c;
        size_t a = 0;
        for (;;) {
            size_t b = line.find('\t', a);
            c.push_back(line.substr(a, b == std::string::npos ? std::string::npos : b - a));
            if (b == std::string::npos) break;
            a = b + 1;
        }
        uint64_t st, en;
        if (c.size() < 3 || !parse_u64(c[1].c_str(), c[1].size(), st) || !parse_u64(c[2].c_str(), c[2].size(), en)) continue; // warn + skip
        Region r;
        r.name = c[0], r.st = st, r.en = en;
        r.id = c.size() > 3 ? c[3] : (c[0] + ":" + std::to_string(st + 1) + "-" + std::to_string(en)); // bed.rs:150-153
        out.push_back(std::move(r));
    }
    gzclose(f);
    return out;
}

std::string f32_display(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v < 0 ? "-inf" : "inf";
    char buf[128];
    auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::fixed); // shortest round trip, positional
    return std::string(buf, r.ptr);
}

// ---- batches ------------------------------------------------------------------------------------------
struct HostBatch {
    std::vector<uint32_t> ops, contig;
    std::vector<uint64_t> op_off, t_st, t_en, q_st, q_en;
    std::vector<uint8_t> strand;
    std::vector<std::string> contig_names;
    std::unordered_map<std::string, uint32_t> contig_id;
    explicit HostBatch(const std::vector<PafRecord> &recs) { fill(recs, nullptr); }
    HostBatch(const std::vector<PafRecord> &all, const std::vector<uint32_t> &idx) { fill(all, &idx); } // the records all[idx[k]], in that order
    void fill(const std::vector<PafRecord> &all, const std::vector<uint32_t> *idx) {
        const size_t n = idx ? idx->size() : all.size();
        auto rec = [&](size_t i) -> const PafRecord & { return idx ? all[(*idx)[i]] : all[i]; };
        op_off.assign(n + 1, 0);
        size_t total = 0;
        for (size_t i = 0; i < n; i++) {
            total += rec(i).cigar.size();
            op_off[i + 1] = total;
        }
        ops.resize(total + 4, 0);
        t_st.resize(n), t_en.resize(n), q_st.resize(n), q_en.resize(n), strand.resize(n), contig.resize(n);
        parallel_chunks(n, [&](unsigned, size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++)
                if (!rec(i).cigar.empty()) memcpy(&ops[op_off[i]], rec(i).cigar.data(), rec(i).cigar.size() * 4);
        });
        for (size_t i = 0; i < n; i++) {
            const PafRecord &r = rec(i);
            t_st[i] = r.t_st, t_en[i] = r.t_en, q_st[i] = r.q_st, q_en[i] = r.q_en;
            strand[i] = (uint8_t)r.strand;
            auto it = contig_id.find(r.t_name);
            if (it == contig_id.end()) { // dense ids in order of first appearance = canonical contig order
                it = contig_id.emplace(r.t_name, (uint32_t)contig_names.size()).first;
                contig_names.push_back(r.t_name);
            }
            contig[i] = it->second;
        }
    }
    uint64_t n() const { return t_st.size(); }
};

static void panic_on(uint32_t status, const char *what, size_t i) {
    if (status >= RB_ST_PANIC_NOTFOUND) throw Panic(std::string(what) + ": record " + std::to_string(i + 1) + " has status " + std::to_string(status));
}

// the record after remove_trailing_indels: id gains _TO.<lead>.<trail> (paf.rs:726-732)
// the last `count` words of a cigar, last OP first (the order remove_trailing_indels pops them in, paf.rs:704-723); an op and its
// continuation word stay together
static std::vector<uint32_t> popped_trail(const std::vector<uint32_t> &cig, uint32_t count) {
    std::vector<uint32_t> t;
    size_t i = cig.size(), left = count;
    while (left > 0 && i > 0) {
        if ((cig[i - 1] & 15u) == RB_OP_CONT && i >= 2 && left >= 2) {
            t.push_back(cig[i - 2]);
            t.push_back(cig[i - 1]);
            i -= 2, left -= 2;
        } else {
            t.push_back(cig[i - 1]);
            i--, left--;
        }
    }
    return t;
}
static std::string stripped_id(const PafRecord &r, const rb_norm_row &nr) {
    if (!(nr.flags & RB_F_STRIPPED)) return r.id;
    std::vector<uint32_t> lead(r.cigar.begin(), r.cigar.begin() + nr.lead_ops);
    return r.id + "_TO." + cigar_to_string(lead) + "." + cigar_to_string(popped_trail(r.cigar, nr.trail_ops));
}

// `rb --gpus N`: a worker process reads only its own lines of the (plain) input file -- bytes [begin, end), cut at line starts
static uint64_t g_slice_begin = 0, g_slice_end = 0;
static bool g_sliced = false;
void set_input_slice(uint64_t begin, uint64_t end) { g_slice_begin = begin, g_slice_end = end, g_sliced = true; }

// `rb --gpus N trim-paf`: the unit of dependency is the query-name group (paf.rs:223, :235), and the output is ordered by query
// name (the stable sort at :223), so worker k takes the lines whose first column lies in the k-th range of the sorted names:
// [lo, hi) bytewise (Rust's String order), an absent bound = open.  The readers below then keep only those lines.
static bool g_qrange = false, g_q_has_lo = false, g_q_has_hi = false;
static std::string g_q_lo, g_q_hi;
void set_input_query_range(const std::string *lo, const std::string *hi) {
    g_qrange = true, g_q_has_lo = lo != nullptr, g_q_has_hi = hi != nullptr;
    if (lo) g_q_lo = *lo;
    if (hi) g_q_hi = *hi;
}
static inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f'; }
static inline std::string_view first_token(const char *a, const char *e) { // split_ascii_whitespace().next()
    while (a < e && is_ws(*a)) a++;
    const char *q = a;
    while (q < e && !is_ws(*q)) q++;
    return std::string_view(a, (size_t)(q - a));
}
static inline bool in_query_range(std::string_view t) {
    return (!g_q_has_lo || t.compare(g_q_lo) >= 0) && (!g_q_has_hi || t.compare(g_q_hi) < 0);
}
// the kept lines of text [a, e) (whole lines; the last one may lack its newline), appended to `out` with their newlines
static void keep_query_lines(const char *a, const char *e, std::vector<char> &out) {
    const char *run = nullptr; // start of the current run of kept lines
    while (a < e) {
        const char *nl = (const char *)memchr(a, '\n', (size_t)(e - a));
        const char *next = nl ? nl + 1 : e;
        if (in_query_range(first_token(a, nl ? nl : e))) {
            if (!run) run = a;
        } else if (run) {
            out.insert(out.end(), run, a);
            run = nullptr;
        }
        a = next;
    }
    if (run) out.insert(out.end(), run, e);
}

// stdin can be read once: what came out of it is kept, so that a command's text route and -- when a line needs the general parser --
// the route it falls back to both see the input (round 4: `rb trim-paf x | rb break-paf -` took the line-by-line route for 15 GB)
static const std::string &stdin_text() {
    static std::string all;
    static bool done = false;
    if (!done) {
        done = true;
        gzFile f = gzdopen(0, "rb"); // (plain text passes through; gzip and BGZF are inflated)
        if (!f) throw Panic("Failed to open -");
        gzbuffer(f, 1 << 22);
        size_t cap = (size_t)256 << 20, n = 0;
        all.resize(cap);
        for (;;) {
            if (cap - n < ((size_t)16 << 20)) all.resize(cap *= 2);
            const int r = gzread(f, &all[n], (unsigned)std::min<size_t>(cap - n, (size_t)1 << 30));
            if (r <= 0) break;
            n += (size_t)r;
        }
        gzclose(f);
        all.resize(n);
    }
    return all;
}
// the whole (decompressed) text of a PAF file or stdin
static std::string read_all(const std::string &file_name) {
    std::string all;
    bool plain = false;
    if (file_name == "-") {
        all = stdin_text();
        if (g_qrange) {
            std::vector<char> kept;
            keep_query_lines(all.data(), all.data() + all.size(), kept);
            all.assign(kept.data(), kept.size());
        }
        return all;
    }
    if (file_name != "-") { // uncompressed regular file: read it directly (zlib's pass-through mode is 3x slower)
        FILE *fp = fopen(file_name.c_str(), "rb");
        if (!fp) throw Panic("Failed to open " + file_name);
        unsigned char magic[2] = {0, 0};
        const size_t got = fread(magic, 1, 2, fp);
        if (!(got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) && fseek(fp, 0, SEEK_END) == 0) {
            long sz = ftell(fp);
            if (sz >= 0) {
                long from = 0;
                if (g_sliced) from = (long)std::min<uint64_t>(g_slice_begin, (uint64_t)sz), sz = (long)std::min<uint64_t>(g_slice_end, (uint64_t)sz) - from;
                fseek(fp, from, SEEK_SET);
                all.resize((size_t)sz);
                if (fread(&all[0], 1, (size_t)sz, fp) == (size_t)sz) plain = true;
            }
        }
        fclose(fp);
    }
    if (!plain) {
        gzFile f = file_name == "-" ? gzdopen(0, "rb") : gzopen(file_name.c_str(), "rb");
        if (!f) throw Panic("Failed to open " + file_name);
        gzbuffer(f, 1 << 20);
        all.clear();
        std::vector<char> buf(1 << 24);
        int r;
        while ((r = gzread(f, buf.data(), (unsigned)buf.size())) > 0) all.append(buf.data(), (size_t)r);
        gzclose(f);
    }
    if (g_qrange) {
        std::vector<char> kept;
        keep_query_lines(all.data(), all.data() + all.size(), kept);
        all.assign(kept.data(), kept.size());
    }
    return all;
}
std::string read_input_text(const std::string &file_name) { return read_all(file_name); }

// the bytes of a text file in a buffer nobody zero-fills first (std::string::resize would touch 1.5 GB twice), 32 zero bytes behind
// the text (the device reads whole 16-byte groups).  A plain file is read by all host threads at once, each its own slice.
// gigabyte-sized host buffers: huge pages where the system lets a process opt in (transparent_hugepage = madvise)
static void advise_huge(const void *p, size_t bytes) {
    if (!p || bytes < ((size_t)8 << 20)) return;
    const uintptr_t a = ((uintptr_t)p + ((size_t)2 << 20) - 1) & ~(uintptr_t)(((size_t)2 << 20) - 1), e = ((uintptr_t)p + bytes) & ~(uintptr_t)(((size_t)2 << 20) - 1);
    if (e > a) (void)madvise((void *)a, (size_t)(e - a), MADV_HUGEPAGE);
}
struct TextBuf {
    std::unique_ptr<char[]> p;
    size_t n = 0;
    const char *data() const { return p.get(); }
    size_t size() const { return n; }
};
// explicit_range: bytes [begin, end) of a plain file (a chunk of the pipelined text routes); nullptr: the process-wide slice / filter
static TextBuf read_text(const std::string &file_name, bool raw_bytes = false, const std::pair<uint64_t, uint64_t> *explicit_range = nullptr) { // raw_bytes: a gzip file is NOT inflated
    TextBuf b;
    if (explicit_range) {
        const int fd = open(file_name.c_str(), O_RDONLY);
        if (fd < 0) throw Panic("Failed to open " + file_name);
        const size_t from = (size_t)explicit_range->first;
        b.n = (size_t)(explicit_range->second - explicit_range->first);
        b.p.reset(new char[b.n + 32]);
        advise_huge(b.p.get(), b.n);
        memset(b.p.get() + b.n, 0, 32);
        std::atomic<bool> ok{true};
        parallel_chunks((b.n + (1u << 22) - 1) >> 22, [&](unsigned, size_t lo, size_t hi) { // 4 MiB pieces
            size_t a = lo << 22;
            const size_t e = std::min(b.n, hi << 22);
            while (a < e) {
                const ssize_t r = pread(fd, b.p.get() + a, e - a, (off_t)(from + a));
                if (r <= 0) {
                    ok = false;
                    return;
                }
                a += (size_t)r;
            }
        });
        close(fd);
        if (!ok) throw Panic("Failed to read " + file_name);
        return b;
    }
    if (file_name != "-") {
        const int fd = open(file_name.c_str(), O_RDONLY);
        if (fd < 0) throw Panic("Failed to open " + file_name);
        unsigned char magic[2] = {0, 0};
        struct stat st;
        const bool plain = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && (raw_bytes || !(pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b));
        if (plain && g_qrange && !raw_bytes) { // every host thread filters its own run of whole lines, then the kept lines side by side
            const size_t size = (size_t)st.st_size;
            const unsigned T = parallel_chunk_count((size >> 22) + 1);
            std::vector<size_t> cut(T + 1, size);
            cut[0] = 0;
            std::vector<char> probe(1 << 16);
            for (unsigned t = 1; t < T; t++) { // the first line start at or behind size * t / T
                size_t at = std::max(cut[t - 1], size / T * t);
                bool found = at == 0;
                if (!found) at -= 1;
                while (!found && at < size) {
                    const ssize_t r = pread(fd, probe.data(), probe.size(), (off_t)at);
                    if (r <= 0) break;
                    const void *nl = memchr(probe.data(), '\n', (size_t)r);
                    if (nl) at += (size_t)((const char *)nl - probe.data()) + 1, found = true;
                    else at += (size_t)r;
                }
                cut[t] = found ? std::min(at, size) : size;
            }
            std::vector<std::vector<char>> kept(T);
            std::atomic<bool> ok{true};
            parallel_chunks(T, [&](unsigned, size_t lo, size_t hi) {
                for (size_t t = lo; t < hi; t++) {
                    std::vector<char> blk((size_t)8 << 20);
                    size_t pos = cut[t];
                    while (pos < cut[t + 1]) {
                        const size_t want = std::min(blk.size(), cut[t + 1] - pos);
                        size_t got = 0;
                        while (got < want) {
                            const ssize_t r = pread(fd, blk.data() + got, want - got, (off_t)(pos + got));
                            if (r <= 0) { ok = false; return; }
                            got += (size_t)r;
                        }
                        size_t use = got; // whole lines only, unless this is the end of the run
                        if (pos + got < cut[t + 1]) {
                            while (use > 0 && blk[use - 1] != '\n') use--;
                            if (use == 0) { blk.resize(blk.size() * 2); continue; } // a line longer than the block
                        }
                        keep_query_lines(blk.data(), blk.data() + use, kept[t]);
                        pos += use;
                    }
                }
            });
            close(fd);
            if (!ok) throw Panic("Failed to read " + file_name);
            std::vector<size_t> at(T + 1, 0);
            for (unsigned t = 0; t < T; t++) at[t + 1] = at[t] + kept[t].size();
            b.n = at[T];
            b.p.reset(new char[b.n + 32]);
            advise_huge(b.p.get(), b.n);
            memset(b.p.get() + b.n, 0, 32);
            parallel_chunks(T, [&](unsigned, size_t lo, size_t hi) {
                for (size_t t = lo; t < hi; t++)
                    if (!kept[t].empty()) memcpy(b.p.get() + at[t], kept[t].data(), kept[t].size());
            });
            return b;
        }
        if (plain) {
            b.n = (size_t)st.st_size;
            size_t from = 0;
            if (g_sliced && !raw_bytes) from = (size_t)std::min<uint64_t>(g_slice_begin, b.n), b.n = (size_t)std::min<uint64_t>(g_slice_end, b.n) - from;
            b.p.reset(new char[b.n + 32]);
            advise_huge(b.p.get(), b.n);
            memset(b.p.get() + b.n, 0, 32);
            std::atomic<bool> ok{true};
            parallel_chunks((b.n + (1u << 22) - 1) >> 22, [&](unsigned, size_t lo, size_t hi) { // 4 MiB pieces
                size_t a = lo << 22;
                const size_t e = std::min(b.n, hi << 22);
                while (a < e) {
                    const ssize_t r = pread(fd, b.p.get() + a, e - a, (off_t)(from + a));
                    if (r <= 0) {
                        ok = false;
                        return;
                    }
                    a += (size_t)r;
                }
            });
            close(fd);
            if (!ok) throw Panic("Failed to read " + file_name);
            return b;
        }
        close(fd);
        if (raw_bytes) throw Panic("Failed to read " + file_name);
    }
    const std::string all = read_all(file_name); // gzip / stdin
    b.n = all.size();
    b.p.reset(new char[b.n + 32]);
    memcpy(b.p.get(), all.data(), b.n);
    memset(b.p.get() + b.n, 0, 32);
    return b;
}
// `rb --gpus N trim-paf`, parent side (no device): the N - 1 query names that cut the sorted names of a plain file into N ranges
// of about equal text bytes; whole query groups stay together (paf.rs:223).  Fewer cuts come back when there are fewer names.
std::vector<std::string> query_name_cuts(const std::string &file_name, int n) {
    const int fd = open(file_name.c_str(), O_RDONLY);
    if (fd < 0) throw Panic("Failed to open " + file_name);
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); throw Panic("Failed to read " + file_name); }
    const size_t size = (size_t)st.st_size;
    const unsigned T = parallel_chunk_count((size >> 22) + 1);
    std::vector<size_t> cut(T + 1, size);
    cut[0] = 0;
    std::vector<char> probe(1 << 16);
    for (unsigned t = 1; t < T; t++) {
        size_t at = std::max(cut[t - 1], size / T * t);
        bool found = at == 0;
        if (!found) at -= 1;
        while (!found && at < size) {
            const ssize_t r = pread(fd, probe.data(), probe.size(), (off_t)at);
            if (r <= 0) break;
            const void *nl = memchr(probe.data(), '\n', (size_t)r);
            if (nl) at += (size_t)((const char *)nl - probe.data()) + 1, found = true;
            else at += (size_t)r;
        }
        cut[t] = found ? std::min(at, size) : size;
    }
    std::vector<std::unordered_map<std::string, uint64_t>> part(T);
    std::atomic<bool> ok{true};
    parallel_chunks(T, [&](unsigned, size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; t++) {
            std::vector<char> blk((size_t)8 << 20);
            size_t pos = cut[t];
            std::string key;
            while (pos < cut[t + 1]) {
                const size_t want = std::min(blk.size(), cut[t + 1] - pos);
                size_t got = 0;
                while (got < want) {
                    const ssize_t r = pread(fd, blk.data() + got, want - got, (off_t)(pos + got));
                    if (r <= 0) { ok = false; return; }
                    got += (size_t)r;
                }
                size_t use = got;
                if (pos + got < cut[t + 1]) {
                    while (use > 0 && blk[use - 1] != '\n') use--;
                    if (use == 0) { blk.resize(blk.size() * 2); continue; }
                }
                const char *a = blk.data(), *e = a + use;
                while (a < e) {
                    const char *nl = (const char *)memchr(a, '\n', (size_t)(e - a));
                    const char *next = nl ? nl + 1 : e;
                    const std::string_view tk = first_token(a, nl ? nl : e);
                    key.assign(tk.data(), tk.size());
                    part[t][key] += (uint64_t)(next - a);
                    a = next;
                }
                pos += use;
            }
        }
    });
    close(fd);
    if (!ok) throw Panic("Failed to read " + file_name);
    std::map<std::string, uint64_t> all; // bytewise order = Rust's String order
    uint64_t total = 0;
    for (auto &m : part)
        for (auto &kv : m) all[kv.first] += kv.second, total += kv.second;
    std::vector<std::string> cuts;
    uint64_t acc = 0;
    int k = 1;
    for (auto it = all.begin(); it != all.end() && k < n; ++it) {
        // `it` opens range k when the names before it hold k / n of the bytes
        while (k < n && acc >= total / (uint64_t)n * (uint64_t)k && it != all.begin()) {
            if (cuts.empty() || cuts.back() != it->first) cuts.push_back(it->first);
            k++;
        }
        acc += it->second;
    }
    return cuts;
}

// BufRead::lines: split on \n, strip one trailing \r
static std::vector<std::pair<size_t, size_t>> split_lines(std::string_view all) {
    // newline positions, every thread its slice of the bytes; then the (start, length) pairs in file order
    const unsigned T = parallel_chunk_count((all.size() >> 20) + 1);
    std::vector<std::vector<size_t>> nl(T);
    parallel_chunks(T, [&](unsigned, size_t lo, size_t hi) {
        for (size_t t = lo; t < hi; t++) {
            const size_t a0 = all.size() * t / T, a1 = all.size() * (t + 1) / T;
            for (size_t a = a0; a < a1;) {
                const void *q = memchr(all.data() + a, '\n', a1 - a);
                if (!q) break;
                const size_t b = (size_t)((const char *)q - all.data());
                nl[t].push_back(b);
                a = b + 1;
            }
        }
    });
    size_t total = 0;
    for (const auto &v : nl) total += v.size();
    std::vector<std::pair<size_t, size_t>> lines;
    lines.reserve(total + 1);
    size_t a = 0;
    auto add = [&](size_t b) { // line [a, b), b = the newline or the end of the text
        size_t e = b;
        if (e > a && all[e - 1] == '\r') e--;
        lines.emplace_back(a, e - a);
        a = b + 1;
    };
    for (const auto &v : nl)
        for (size_t b : v) add(b);
    if (a < all.size()) add(all.size());
    return lines;
}

static bool paf_from_text_file(Engine &eng, const std::string &file_name, Paf &paf);
Paf Paf::from_file(Engine &eng, const std::string &file_name) {
    if (file_name != "-" && !getenv("RB_GENERAL_PATH")) { // regular files: header columns on the host, the cg:Z: values parsed on the device
        Paf fast;
        if (paf_from_text_file(eng, file_name, fast)) return fast;
    }
    std::string all = read_all(file_name); // lines are parsed in parallel below
    const std::vector<std::pair<size_t, size_t>> lines = split_lines(all);
    const unsigned T = parallel_chunk_count(lines.size());
    std::vector<std::vector<PafRecord>> part(T);
    std::vector<std::vector<size_t>> skipped(T);
    parallel_chunks(lines.size(), [&](unsigned t, size_t lo, size_t hi) {
        part[t].reserve(hi - lo);
        std::string line;
        for (size_t i = lo; i < hi; i++) {
            line.assign(all, lines[i].first, lines[i].second);
            PafRecord rec;
            if (paf_record_new(line, rec) == 0)
                part[t].push_back(std::move(rec));
            else
                skipped[t].push_back(i);
        }
    });
    Paf paf;
    for (unsigned t = 0; t < T; t++) {
        for (size_t i : skipped[t]) fprintf(stderr, "\nUnable to parse PAF record. Skipping line %zu\n", i + 1);
        for (auto &r : part[t]) paf.records.push_back(std::move(r));
    }
    all.clear();
    all.shrink_to_fit();
    // check_integrity().unwrap() (paf.rs:70) for the whole file in one device pass; overwrites nmatch / aln_len
    HostBatch b(paf.records);
    std::vector<rb_reduce_row> red(b.n());
    eng.check(rb_host_scan_records(eng.ctx(), b.n(), b.ops.data(), b.op_off.data(), b.t_st.data(), b.t_en.data(), b.q_st.data(),
                                   b.q_en.data(), b.strand.data(), red.data(), nullptr),
              "rb_host_scan_records");
    for (size_t i = 0; i < paf.records.size(); i++) {
        if (red[i].status != RB_ST_OK) throw Panic("check_integrity: record " + std::to_string(i + 1) + " status " + std::to_string(red[i].status));
        paf.records[i].nmatch = red[i].nmatch;
        paf.records[i].aln_len = red[i].aln_len;
    }
    return paf;
}

std::vector<PafRecord> paf_swap_query_and_target(Engine &eng, const std::vector<PafRecord> &recs) {
    HostBatch b(recs);
    std::vector<uint32_t> out(b.ops.size());
    eng.check(rb_host_swap(eng.ctx(), b.n(), b.ops.data(), b.op_off.data(), b.strand.data(), out.data()), "rb_host_swap");
    std::vector<PafRecord> fl(recs);
    for (size_t i = 0; i < recs.size(); i++) {
        fl[i].t_name = recs[i].q_name, fl[i].t_len = recs[i].q_len, fl[i].t_st = recs[i].q_st, fl[i].t_en = recs[i].q_en;
        fl[i].q_name = recs[i].t_name, fl[i].q_len = recs[i].t_len, fl[i].q_st = recs[i].t_st, fl[i].q_en = recs[i].t_en;
        fl[i].cigar.assign(out.begin() + b.op_off[i], out.begin() + b.op_off[i + 1]);
    }
    return fl;
}

static std::vector<PafRecord> rows_to_records(const std::vector<PafRecord> &src, const std::vector<rb_norm_row> &norm, const rb_hit_row *rows,
                                              uint64_t n_rows, const uint32_t *out, const std::vector<Region> *rgns) {
    for (uint64_t k = 0; k < n_rows; k++)
        if (rows[k].status >= RB_ST_PANIC_NOTFOUND) throw Panic("Problem getting index in cigar: record " + std::to_string(rows[k].rec + 1));
    const unsigned T = parallel_chunk_count((size_t)n_rows);
    std::vector<std::vector<PafRecord>> part(T);
    parallel_chunks((size_t)n_rows, [&](unsigned t, size_t lo, size_t hi) {
        for (size_t k = lo; k < hi; k++) {
            const rb_hit_row &h = rows[k];
            if (h.status != RB_ST_OK) continue; // trim_paf_rec_to_rgn returned None
            const PafRecord &s = src[h.rec];
            PafRecord r;
            r.q_name = s.q_name, r.q_len = s.q_len, r.strand = s.strand, r.t_name = s.t_name, r.t_len = s.t_len, r.mapq = s.mapq;
            r.t_st = h.t_st, r.t_en = h.t_en, r.q_st = h.q_st, r.q_en = h.q_en, r.nmatch = h.nmatch, r.aln_len = h.aln_len;
            r.id = (rgns && !(h.flags & RB_HIT_INSIDE)) ? (*rgns)[h.win].id : stripped_id(s, norm[h.rec]);
            if (h.flags & RB_HIT_DESCRIPTOR) {
                // the clip is described, not copied: {first op of the record's own cigar, count, first length, last length}
                const uint32_t *d = out + h.out_off;
                r.cigar.assign(s.cigar.begin() + d[0], s.cigar.begin() + d[0] + d[1]);
                if (!(h.flags & RB_HIT_INSIDE)) {
                    if (d[1] == 1) {
                        r.cigar[0] = (h.aln_len << 4) | (r.cigar[0] & 15u);
                    } else {
                        r.cigar[0] = (d[2] << 4) | (r.cigar[0] & 15u);
                        r.cigar.back() = (d[3] << 4) | (r.cigar.back() & 15u);
                    }
                }
            } else {
                r.cigar.assign(out + h.out_off, out + h.out_off + h.out_n);
            }
            part[t].push_back(std::move(r));
        }
    });
    std::vector<PafRecord> res;
    size_t total = 0;
    for (auto &p : part) total += p.size();
    res.reserve(total);
    for (auto &p : part)
        for (auto &r : p) res.push_back(std::move(r));
    return res;
}

// `println!("{}", rec)` for every Some(rec) of a liftover / break-paf result, straight from the hit rows: no
// intermediate PafRecord is built (1.3 M of them cost more than the whole device pass), text is encoded on all
// host cores and returned in the reference's output order.
// per chunk of output text: (contig id, byte offset in the chunk) wherever the rows of another contig begin (and at the chunk's start)
typedef std::vector<std::vector<std::pair<uint32_t, size_t>>> RunMarks;
static void marks_to_runs(const RunMarks &marks, const std::vector<std::string> &text, TextRuns &tr) {
    tr.runs.clear();
    for (size_t t = 0; t < text.size(); t++)
        for (size_t m = 0; m < marks[t].size(); m++) {
            const size_t end = m + 1 < marks[t].size() ? marks[t][m + 1].second : text[t].size();
            const uint64_t bytes = end - marks[t][m].second;
            if (!bytes) continue;
            if (!tr.runs.empty() && tr.runs.back().first == marks[t][m].first) tr.runs.back().second += bytes;
            else tr.runs.emplace_back(marks[t][m].first, bytes);
        }
}
static std::vector<std::string> rows_to_text(const std::vector<PafRecord> &src, const std::vector<rb_norm_row> &norm, const rb_hit_row *rows, uint64_t n_rows,
                                const uint32_t *out, const std::vector<Region> *rgns, const uint32_t *contig_of_rec = nullptr, RunMarks *marks = nullptr) {
    for (uint64_t k = 0; k < n_rows; k++)
        if (rows[k].status >= RB_ST_PANIC_NOTFOUND) throw Panic("Problem getting index in cigar: record " + std::to_string(rows[k].rec + 1));
    const unsigned T = parallel_chunk_count((size_t)n_rows);
    std::vector<std::string> part(T);
    if (marks) marks->assign(T, {});
    parallel_chunks((size_t)n_rows, [&](unsigned t, size_t lo, size_t hi) {
        std::string &o = part[t];
        size_t est = 0; // ~2.7 text bytes per op on alignment cigars; one allocation instead of a dozen doublings
        for (size_t k = lo; k < hi; k++) est += rows[k].status == RB_ST_OK ? 160 + (size_t)rows[k].out_n * 3 : 0;
        o.reserve(est);
        advise_huge(o.data(), est);
        char nb[24];
        auto num = [&](uint64_t v) {
            auto r = std::to_chars(nb, nb + sizeof nb, v);
            o.append(nb, r.ptr);
        };
        auto op = [&](uint32_t v) {
            auto r = std::to_chars(nb, nb + sizeof nb, v >> 4);
            o.append(nb, r.ptr);
            o.push_back(OPCH[v & 15u]);
        };
        for (size_t k = lo; k < hi; k++) {
            const rb_hit_row &h = rows[k];
            if (h.status != RB_ST_OK) continue;
            const PafRecord &s = src[h.rec];
            if (marks && ((*marks)[t].empty() || (*marks)[t].back().first != contig_of_rec[h.rec])) (*marks)[t].emplace_back(contig_of_rec[h.rec], o.size());
            o += s.q_name; o += '\t'; num(s.q_len); o += '\t'; num(h.q_st); o += '\t'; num(h.q_en); o += '\t'; o += s.strand; o += '\t';
            o += s.t_name; o += '\t'; num(s.t_len); o += '\t'; num(h.t_st); o += '\t'; num(h.t_en); o += '\t'; num(h.nmatch); o += '\t';
            num(h.aln_len); o += '\t'; num(s.mapq); o += "\tid:Z:";
            o += (rgns && !(h.flags & RB_HIT_INSIDE)) ? (*rgns)[h.win].id : stripped_id(s, norm[h.rec]);
            o += "\tcg:Z:";
            if (h.flags & RB_HIT_DESCRIPTOR) {
                const uint32_t *d = out + h.out_off;
                const uint32_t *c = s.cigar.data() + d[0];
                const uint32_t n = d[1];
                if (h.flags & RB_HIT_INSIDE) {
                    append_ops(o, c, n);
                } else if (n == 1) {
                    op((h.aln_len << 4) | (c[0] & 15u));
                } else {
                    op((d[2] << 4) | (c[0] & 15u));
                    for (uint32_t i = 1; i + 1 < n; i++) op(c[i]);
                    op((d[3] << 4) | (c[n - 1] & 15u));
                }
            } else {
                append_ops(o, out + h.out_off, h.out_n);
            }
            o += '\n';
        }
    });
    return part; // chunks in output order (concatenating 3 GB of text would cost as much as encoding it)
}

// Display of many records at once (text encode in parallel, written in order)
std::vector<std::string> records_to_text(const std::vector<PafRecord> &recs) {
    const unsigned T = parallel_chunk_count(recs.size());
    std::vector<std::string> part(T);
    parallel_chunks(recs.size(), [&](unsigned t, size_t lo, size_t hi) {
        std::string &o = part[t];
        for (size_t i = lo; i < hi; i++) {
            o += recs[i].to_string();
            o += '\n';
        }
    });
    return part;
}

namespace {
struct LiftResult { // rows + clip descriptors of one liftover / break-paf call (freed on destruction)
    std::vector<PafRecord> swapped;
    const std::vector<PafRecord> *recs = nullptr;
    std::vector<rb_norm_row> norm;
    std::vector<uint32_t> contig;          // per record: dense target id, first appearance order
    std::vector<std::string> contig_names; // of the records (windows on other names come after)
    rb_hit_row *rows = nullptr;
    uint32_t *out = nullptr;
    uint64_t n_rows = 0, n_out = 0;
    ~LiftResult() {
        rb_host_free(rows);
        rb_host_free(out);
    }
};
void run_liftover(Engine &eng, const std::vector<Region> &rgns, const std::vector<PafRecord> &paf_recs, bool invert_query, LiftResult &L) {
    if (invert_query) L.swapped = paf_swap_query_and_target(eng, paf_recs);
    L.recs = invert_query ? &L.swapped : &paf_recs;
    const std::vector<PafRecord> &recs = *L.recs;
    double tl = now_s();
    HostBatch b(recs);
    lap("pack batch", tl);
    std::vector<uint32_t> w_contig(rgns.size());
    std::vector<uint64_t> w_st(rgns.size()), w_en(rgns.size());
    for (size_t i = 0; i < rgns.size(); i++) {
        auto it = b.contig_id.find(rgns[i].name);
        if (it == b.contig_id.end()) it = b.contig_id.emplace(rgns[i].name, (uint32_t)b.contig_id.size()).first; // no record there
        w_contig[i] = it->second, w_st[i] = rgns[i].st, w_en[i] = rgns[i].en;
    }
    L.norm.resize(b.n());
    rb_counters cnt;
    eng.check(rb_host_liftover(eng.ctx(), b.n(), b.ops.data(), b.op_off.data(), b.t_st.data(), b.t_en.data(), b.q_st.data(), b.q_en.data(),
                               b.strand.data(), b.contig.data(), rgns.size(), w_contig.data(), w_st.data(), w_en.data(),
                               eng.bsearch_policy | RB_LIFT_DESCRIPTORS | RB_LIFT_FUSED_SCAN, L.norm.data(), &L.rows, &L.n_rows, &L.out, &L.n_out,
                               &cnt),
              "rb_host_liftover"); // (fused scan: aligned_pairs' strip + integrity check happen inside the clip kernel)
    lap("rb_host_liftover", tl);
    L.contig.swap(b.contig), L.contig_names.swap(b.contig_names);
    for (size_t i = 0; i < L.norm.size(); i++) panic_on(L.norm[i].status, "aligned_pairs", i); // liftover.rs:119-121
}
void run_break(Engine &eng, const std::vector<PafRecord> &paf_recs, uint32_t break_length, LiftResult &L) {
    L.recs = &paf_recs;
    HostBatch b(paf_recs);
    L.norm.resize(b.n());
    rb_counters cnt;
    eng.check(rb_host_break(eng.ctx(), b.n(), b.ops.data(), b.op_off.data(), b.t_st.data(), b.t_en.data(), b.q_st.data(), b.q_en.data(),
                            b.strand.data(), break_length, eng.bsearch_policy | RB_LIFT_DESCRIPTORS | RB_LIFT_FUSED_SCAN, L.norm.data(), &L.rows,
                            &L.n_rows, &L.out,
                            &L.n_out, &cnt),
              "rb_host_break");
    for (size_t i = 0; i < L.norm.size(); i++) panic_on(L.norm[i].status, "aligned_pairs", i); // main.rs:275
}
} // namespace

std::vector<PafRecord> trim_paf_by_rgns(Engine &eng, const std::vector<Region> &rgns, const std::vector<PafRecord> &paf_recs, bool invert_query) {
    LiftResult L;
    run_liftover(eng, rgns, paf_recs, invert_query, L);
    return rows_to_records(*L.recs, L.norm, L.rows, L.n_rows, L.out, &rgns);
}
std::vector<std::string> trim_paf_by_rgns_text(Engine &eng, const std::vector<Region> &rgns, const std::vector<PafRecord> &paf_recs, bool invert_query,
                                               TextRuns *runs) {
    LiftResult L;
    run_liftover(eng, rgns, paf_recs, invert_query, L);
    double tl = now_s();
    RunMarks marks;
    std::vector<std::string> text = rows_to_text(*L.recs, L.norm, L.rows, L.n_rows, L.out, &rgns, L.contig.data(), runs ? &marks : nullptr);
    if (runs) {
        runs->contigs = L.contig_names;
        marks_to_runs(marks, text, *runs);
    }
    lap("rows -> text", tl);
    return text;
}

// ---- liftover, text in -> text out ------------------------------------------------------------------------------------
// main.rs:186-214 without --qbed / --largest: Paf::from_file + trim_paf_by_rgns + println! of every record.  The CIGAR text
// (95 % of a PAF's bytes) never becomes host data structures: the `cg:Z:` values are parsed on the device, the clipped CIGARs
// are printed on the device from the clip descriptors, and the host only handles the twelve header columns.
namespace {
struct HeaderOnly { // PafRecord::new (paf.rs:379-430) minus the CIGAR: token ranges inside the file text
    size_t q_name, q_name_n, t_name, t_name_n, cg, cg_n;
    uint64_t q_len, q_st, q_en, t_len, t_st, t_en, mapq;
    char strand;
};
// 0 = ok, 1 = Err(ParsePafColumn) (the line is skipped), 2 = needs the general parser (two cg tags); throws Panic
int paf_header_new(const char *base, size_t a, size_t n, HeaderOnly &h) {
    std::pair<size_t, size_t> t[64];
    size_t nt = 0, cg_tags = 0;
    const char *p = base + a, *end = p + n;
    auto ws = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f'; };
    h.cg = h.cg_n = 0;
    while (p < end) {
        while (p < end && ws(*p)) p++;
        if (p >= end) break;
        const char *q = p;
        while (q < end && !ws(*q)) q++;
        if (nt < 12) {
            t[nt] = {(size_t)(p - base), (size_t)(q - p)};
        } else { // PAF_TAG "(..):(.):(.*)", leftmost match (paf.rs:21, :387-390)
            const size_t len = (size_t)(q - p);
            size_t m = (size_t)-1;
            for (size_t i = 0; i + 5 <= len; i++)
                if (p[i + 2] == ':' && p[i + 4] == ':') {
                    m = i;
                    break;
                }
            if (m == (size_t)-1) throw Panic("assertion failed: PAF_TAG.is_match(token)");
            if (p[m] == 'c' && p[m + 1] == 'g') {
                if (cg_tags++ == 0) h.cg = (size_t)(p - base) + m + 5, h.cg_n = len - (m + 5);
            }
        }
        nt++;
        p = q;
    }
    if (nt < 12) throw Panic("assertion failed: t.len() >= 12"); // paf.rs:381
    if (cg_tags > 1) return 2; // (a second cg tag is parsed when the first one is empty, paf.rs:395: left to the general path)
    uint64_t v[12] = {0};
    static const int numeric[] = {1, 2, 3, 6, 7, 8, 9, 10, 11};
    for (int c : numeric)
        if (!parse_u64(base + t[c].first, t[c].second, v[c])) return 1;
    if (t[4].second != 1) return 1;
    h.q_name = t[0].first, h.q_name_n = t[0].second, h.t_name = t[5].first, h.t_name_n = t[5].second;
    h.q_len = v[1], h.q_st = v[2], h.q_en = v[3], h.t_len = v[6], h.t_st = v[7], h.t_en = v[8], h.mapq = v[11];
    h.strand = base[t[4].first];
    return 0;
}
} // namespace

namespace {
// the file text, where every kept line's columns and cg:Z: value sit in it, and the arrays the ABI wants
struct TextFile {
    TextBuf all;
    size_t text_bytes = 0;
    std::vector<HeaderOnly> recs;
    std::vector<uint64_t> cig_off, cig_end, t_st, t_en, q_st, q_en;
    std::vector<uint8_t> strand;
    std::vector<uint32_t> contig;
    std::unordered_map<std::string_view, uint32_t> contig_id; // dense ids in order of first appearance (canonical contig order)
    std::string_view name(size_t off, size_t n) const { return std::string_view(all.data() + off, n); }
    // false = a line needs the general parser (two cg tags)
    bool load(const std::string &paf_path, const std::pair<uint64_t, uint64_t> *range = nullptr) {
        double tl = now_s();
        all = read_text(paf_path, false, range);
        text_bytes = all.size();
        const std::vector<std::pair<size_t, size_t>> lines = split_lines(std::string_view(all.data(), text_bytes));
        lap("read + split lines", tl);
        const unsigned T = parallel_chunk_count(lines.size());
        std::vector<std::vector<HeaderOnly>> part(T);
        std::vector<std::vector<size_t>> skipped(T);
        std::vector<int> general(T, 0);
        std::vector<std::string> panics(T);
        parallel_chunks(lines.size(), [&](unsigned t, size_t lo, size_t hi) {
            part[t].reserve(hi - lo);
            try {
                for (size_t i = lo; i < hi; i++) {
                    HeaderOnly h;
                    const int rc = paf_header_new(all.data(), lines[i].first, lines[i].second, h);
                    if (rc == 0) part[t].push_back(h);
                    else if (rc == 1) {
                        // PafRecord::new parses the tags before the columns (paf.rs:387-399): a malformed cg:Z: value panics there
                        // ("Unable to parse cigar string.") before a bad numeric column can make from_file skip the line
                        if (h.cg_n) {
                            std::vector<uint32_t> cig;
                            parse_cigar(all.data() + h.cg, h.cg_n, cig);
                        }
                        skipped[t].push_back(i);
                    }
                    else general[t] = 1;
                }
            } catch (const Panic &e) {
                panics[t] = e.what();
            }
        });
        for (unsigned t = 0; t < T; t++)
            if (!panics[t].empty()) throw Panic(panics[t]);
        for (unsigned t = 0; t < T; t++)
            if (general[t]) return false;
        for (unsigned t = 0; t < T; t++) {
            for (size_t i : skipped[t]) fprintf(stderr, "\nUnable to parse PAF record. Skipping line %zu\n", i + 1);
            recs.insert(recs.end(), part[t].begin(), part[t].end());
        }
        const size_t n = recs.size();
        cig_off.resize(n), cig_end.resize(n), t_st.resize(n), t_en.resize(n), q_st.resize(n), q_en.resize(n), strand.resize(n), contig.resize(n);
        for (size_t i = 0; i < n; i++) {
            const HeaderOnly &h = recs[i];
            cig_off[i] = h.cg, cig_end[i] = h.cg + h.cg_n;
            t_st[i] = h.t_st, t_en[i] = h.t_en, q_st[i] = h.q_st, q_en[i] = h.q_en, strand[i] = (uint8_t)h.strand;
            const std::string_view nm = name(h.t_name, h.t_name_n);
            auto it = contig_id.find(nm);
            if (it == contig_id.end()) it = contig_id.emplace(nm, (uint32_t)contig_id.size()).first;
            contig[i] = it->second;
        }
        lap("header columns", tl);
        return true;
    }
    // the panics of Paf::from_file (paf.rs:399, :70), in the reference's order of checks.  false = a CIGAR the device left to
    // the host (zero-padded numbers longer than a lane) turned out to be valid: the caller takes the general path
    bool check_loaded(const std::vector<uint8_t> &cig_status, const std::vector<rb_reduce_row> &red) const {
        bool unusual = false;
        for (size_t i = 0; i < recs.size(); i++) {
            if (cig_status[i] == RB_TEXT_UNUSUAL) {
                std::vector<uint32_t> cig;
                parse_cigar(all.data() + recs[i].cg, recs[i].cg_n, cig); // throws the reference's panic if it is malformed
                unusual = true;
                continue;
            }
            if (cig_status[i] == RB_TEXT_TOO_LONG) { // a length of 2^28 and more: two words per op, made by the line-by-line parser
                unusual = true;
                continue;
            }
            if (cig_status[i] != RB_TEXT_OK) throw Panic("Unable to parse cigar string.");
        }
        if (unusual) return false;
        for (size_t i = 0; i < recs.size(); i++)
            if (red[i].status != RB_ST_OK) throw Panic("check_integrity: record " + std::to_string(i + 1) + " status " + std::to_string(red[i].status));
        return true;
    }
    // "_TO.<lead>.<trail>" of a record whose end indels were stripped (paf.rs:726-732), parsed from its own text (rare)
    std::string stripped_suffix(uint32_t r, const rb_norm_row &nr) const {
        if (!(nr.flags & RB_F_STRIPPED)) return std::string();
        std::vector<uint32_t> cig;
        parse_cigar(all.data() + recs[r].cg, recs[r].cg_n, cig);
        std::vector<uint32_t> lead(cig.begin(), cig.begin() + nr.lead_ops);
        return "_TO." + cigar_to_string(lead) + "." + cigar_to_string(popped_trail(cig, nr.trail_ops));
    }
};
} // namespace
// Paf::from_file for a regular file: read by all host threads, the twelve header columns parsed on the host, every cg:Z: value parsed
// by rb_k_parse_cigars, check_integrity (paf.rs:70) in the same device visit.  false = the file needs the line-by-line parser.
static bool paf_from_text_file(Engine &eng, const std::string &file_name, Paf &paf) {
    TextFile f;
    if (!f.load(file_name)) return false;
    const size_t n = f.recs.size();
    std::vector<uint64_t> op_off(n + 1, 0);
    std::vector<uint8_t> status(n ? n : 1, 0);
    uint32_t *ops = nullptr;
    eng.check(rb_host_parse_cigars(eng.ctx(), (const uint8_t *)f.all.data(), f.cig_off.data(), f.cig_end.data(), n, op_off.data(), &ops, status.data()),
              "rb_host_parse_cigars");
    struct Free {
        uint32_t *p;
        ~Free() { rb_host_free(p); }
    } guard{ops};
    std::vector<rb_reduce_row> red(n);
    bool text_ok = true;
    for (size_t i = 0; i < n; i++) text_ok = text_ok && status[i] == RB_TEXT_OK;
    if (text_ok && n)
        eng.check(rb_host_scan_records(eng.ctx(), n, ops, op_off.data(), f.t_st.data(), f.t_en.data(), f.q_st.data(), f.q_en.data(), f.strand.data(),
                                       red.data(), nullptr),
                  "rb_host_scan_records");
    if (!f.check_loaded(status, red)) return false; // (throws the reference's panics, in its order of checks)
    paf.records.resize(n);
    parallel_chunks(n, [&](unsigned, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const HeaderOnly &h = f.recs[i];
            PafRecord &r = paf.records[i];
            r.q_name.assign(f.all.data() + h.q_name, h.q_name_n);
            r.q_len = h.q_len, r.q_st = h.q_st, r.q_en = h.q_en, r.strand = h.strand;
            r.t_name.assign(f.all.data() + h.t_name, h.t_name_n);
            r.t_len = h.t_len, r.t_st = h.t_st, r.t_en = h.t_en, r.mapq = h.mapq;
            r.nmatch = red[i].nmatch, r.aln_len = red[i].aln_len; // check_integrity overwrites both
            r.cigar.assign(ops + op_off[i], ops + op_off[i + 1]);
        }
    });
    return true;
}
namespace {
struct TextRows { // results of rb_host_liftover_text / rb_host_break_text (freed on destruction)
    rb_hit_row *rows = nullptr;
    uint64_t n_rows = 0, *toff = nullptr;
    uint8_t *text = nullptr;
    ~TextRows() { rb_host_free(rows), rb_host_free(toff), rb_host_free(text); }
};
// `println!("{}", rec)` for every Some(rec): header columns from the file text and the hit rows, CIGAR text from the device
std::vector<std::string> assemble_lines(const TextFile &f, const std::vector<rb_norm_row> &norm, const TextRows &R, const std::vector<Region> *rgns,
                                        RunMarks *marks = nullptr) {
    for (uint64_t k = 0; k < R.n_rows; k++)
        if (R.rows[k].status >= RB_ST_PANIC_NOTFOUND) throw Panic("Problem getting index in cigar: record " + std::to_string(R.rows[k].rec + 1));
    const unsigned TO = parallel_chunk_count((size_t)R.n_rows);
    std::vector<std::string> out_text(TO);
    if (marks) marks->assign(TO, {});
    parallel_chunks((size_t)R.n_rows, [&](unsigned t, size_t lo, size_t hi) {
        std::string &o = out_text[t];
        size_t est = 0;
        for (size_t k = lo; k < hi; k++) est += R.rows[k].status == RB_ST_OK ? 160 + (size_t)(R.toff[k + 1] - R.toff[k]) : 0;
        o.reserve(est);
        advise_huge(o.data(), est);
        char nb[24];
        auto num = [&](uint64_t v) {
            auto r = std::to_chars(nb, nb + sizeof nb, v);
            o.append(nb, r.ptr);
        };
        for (size_t k = lo; k < hi; k++) {
            const rb_hit_row &h = R.rows[k];
            if (h.status != RB_ST_OK) continue;
            const HeaderOnly &s = f.recs[h.rec];
            if (marks && ((*marks)[t].empty() || (*marks)[t].back().first != f.contig[h.rec])) (*marks)[t].emplace_back(f.contig[h.rec], o.size());
            o.append(f.all.data() + s.q_name, s.q_name_n); o += '\t'; num(s.q_len); o += '\t'; num(h.q_st); o += '\t'; num(h.q_en); o += '\t';
            o += s.strand; o += '\t'; o.append(f.all.data() + s.t_name, s.t_name_n); o += '\t'; num(s.t_len); o += '\t'; num(h.t_st); o += '\t';
            num(h.t_en); o += '\t'; num(h.nmatch); o += '\t'; num(h.aln_len); o += '\t'; num(s.mapq); o += "\tid:Z:";
            if (rgns && !(h.flags & RB_HIT_INSIDE)) o += (*rgns)[h.win].id;
            else o += f.stripped_suffix(h.rec, norm[h.rec]); // (a record read from a file has an empty id of its own)
            o += "\tcg:Z:";
            o.append((const char *)R.text + R.toff[k], (size_t)(R.toff[k + 1] - R.toff[k]));
            o += '\n';
        }
    });
    return out_text;
}
} // namespace

bool liftover_file_text(Engine &eng, const std::string &paf_path, const std::vector<Region> &rgns, std::vector<std::string> &out_text, TextRuns *runs) {
    // (one-shot command: the gigabytes behind these two are left to the end of the process instead of being unmapped piece by piece)
    TextFile &f = *new TextFile;
    if (!f.load(paf_path)) return false; // the caller takes the general path
    double tl = now_s();
    const size_t n = f.recs.size();
    if (runs) { // the records' contigs by first appearance (before windows on other names get ids of their own below)
        runs->contigs.assign(f.contig_id.size(), std::string());
        for (const auto &kv : f.contig_id) runs->contigs[kv.second].assign(kv.first.data(), kv.first.size());
    }
    std::vector<uint32_t> w_contig(rgns.size());
    std::vector<uint64_t> w_st(rgns.size()), w_en(rgns.size());
    for (size_t i = 0; i < rgns.size(); i++) {
        auto it = f.contig_id.find(std::string_view(rgns[i].name));
        if (it == f.contig_id.end()) it = f.contig_id.emplace(std::string_view(rgns[i].name), (uint32_t)f.contig_id.size()).first; // no record there
        w_contig[i] = it->second, w_st[i] = rgns[i].st, w_en[i] = rgns[i].en;
    }
    std::vector<uint8_t> cig_status(n ? n : 1);
    std::vector<rb_reduce_row> red(n);
    std::vector<rb_norm_row> norm(n);
    TextRows &R = *new TextRows;
    rb_counters cnt;
    eng.check(rb_host_liftover_text(eng.ctx(), n, (const uint8_t *)f.all.data(), f.text_bytes, f.cig_off.data(), f.cig_end.data(), f.t_st.data(),
                                    f.t_en.data(), f.q_st.data(), f.q_en.data(), f.strand.data(), f.contig.data(), rgns.size(), w_contig.data(),
                                    w_st.data(), w_en.data(), eng.bsearch_policy, cig_status.data(), red.data(), norm.data(), &R.rows,
                                    &R.n_rows, &R.toff, &R.text, &cnt),
              "rb_host_liftover_text");
    lap("rb_host_liftover_text", tl);
    if (!f.check_loaded(cig_status, red)) return false;
    for (size_t i = 0; i < n; i++) panic_on(norm[i].status, "aligned_pairs", i); // liftover.rs:119-121
    RunMarks marks;
    out_text = assemble_lines(f, norm, R, &rgns, runs ? &marks : nullptr);
    if (runs) marks_to_runs(marks, out_text, *runs);
    lap("assemble lines", tl);
    return true;
}

// main.rs:271-281, text in -> text out
bool break_file_text(Engine &eng, const std::string &paf_path, uint32_t break_length, std::vector<std::string> &out_text) {
    // (one-shot command: the gigabytes behind these two are left to the end of the process instead of being unmapped piece by piece)
    TextFile &f = *new TextFile;
    if (!f.load(paf_path)) return false;
    double tl = now_s();
    const size_t n = f.recs.size();
    std::vector<uint8_t> cig_status(n ? n : 1);
    std::vector<rb_reduce_row> red(n);
    std::vector<rb_norm_row> norm(n);
    TextRows &R = *new TextRows;
    rb_counters cnt;
    eng.check(rb_host_break_text(eng.ctx(), n, (const uint8_t *)f.all.data(), f.text_bytes, f.cig_off.data(), f.cig_end.data(), f.t_st.data(),
                                 f.t_en.data(), f.q_st.data(), f.q_en.data(), f.strand.data(), break_length, eng.bsearch_policy, cig_status.data(),
                                 red.data(), norm.data(), &R.rows, &R.n_rows, &R.toff, &R.text, &cnt),
              "rb_host_break_text");
    lap("rb_host_break_text", tl);
    if (!f.check_loaded(cig_status, red)) return false;
    for (size_t i = 0; i < n; i++) panic_on(norm[i].status, "aligned_pairs", i); // main.rs:275
    out_text = assemble_lines(f, norm, R, nullptr);
    lap("assemble lines", tl);
    return true;
}

// ---- liftover / break-paf, text in -> text out, as a PIPELINE over chunks of the file (round 3) ---------------------------------------
// Records are independent (liftover.rs:123-129), so a big file is cut at line starts into chunks of RB_CHUNK_MB (default 512) MB
// and a few host threads, each with a context (= a stream) of its own on the same GPU, take the chunks in turn: read + header
// columns of chunk k + 2 overlap the H2D / kernels / D2H of chunk k + 1 and the line assembly of chunk k; the caller's sink gets the
// chunks' outputs IN ORDER (and writes while the rest is still being computed).  A chunk's output is contig-major within the
// chunk (TextRuns says where the contigs lie); putting chunks together in the reference's order is the sink's business.
// false = not applicable (not a plain regular file, or smaller than two chunks) or a line needs the general parser: nothing was
// handed to the sink in the first case; in the second the sink may have seen some chunks (it is told so by the return value of
// pipeline_started()).
namespace {
std::atomic<bool> g_pipeline_started{false};
}
bool pipeline_started() { return g_pipeline_started.load(); }
bool lift_file_text_pipelined(int device, int bsearch_policy, bool is_break, uint32_t break_length, const std::string &paf_path,
                              const std::vector<Region> &rgns, const std::function<bool(std::vector<std::string> &, TextRuns &)> &sink) {
    g_pipeline_started = false;
    if (paf_path == "-" || g_qrange) return false;
    const int fd = open(paf_path.c_str(), O_RDONLY);
    if (fd < 0) return false; // (the ordinary path reports it)
    struct stat st;
    unsigned char magic[2] = {0, 0};
    const bool plain = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && !(pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
    if (!plain) {
        close(fd);
        return false;
    }
    uint64_t begin = 0, end = (uint64_t)st.st_size;
    if (g_sliced) begin = std::min<uint64_t>(g_slice_begin, end), end = std::min<uint64_t>(g_slice_end, end);
    const char *ce = getenv("RB_CHUNK_MB"), *ck = getenv("RB_CHUNK_KB"); // (KB: tests on the 2 MB fixture)
    const uint64_t chunk = ck ? (uint64_t)std::max(1, atoi(ck)) << 10 : (uint64_t)std::max(1, ce ? atoi(ce) : 384) << 20;
    if (end - begin < 2 * chunk) {
        close(fd);
        return false;
    }
    std::vector<uint64_t> cut{begin};
    std::vector<char> probe(1 << 16);
    // (RB_FIRST_CHUNK_MB: a smaller first chunk -- the writer has something to do sooner; measured, not the default: see profiles/r03_e2e.md)
    const char *fe = getenv("RB_FIRST_CHUNK_MB");
    const uint64_t first = fe ? std::min<uint64_t>(chunk, (uint64_t)std::max(1, atoi(fe)) << 20) : chunk;
    for (uint64_t at = begin + first; at < end; at += chunk) { // the first line start at or behind every multiple of the chunk size
        uint64_t a = std::max(at, cut.back()) - 1;
        bool found = false;
        while (a < end && !found) {
            const ssize_t r = pread(fd, probe.data(), probe.size(), (off_t)a);
            if (r <= 0) break;
            const void *nl = memchr(probe.data(), '\n', (size_t)r);
            if (nl) a += (uint64_t)((const char *)nl - probe.data()) + 1, found = true;
            else a += (uint64_t)r;
        }
        if (!found || a >= end) break;
        if (a > cut.back()) cut.push_back(a);
    }
    cut.push_back(end);
    close(fd);
    const size_t n_chunks = cut.size() - 1;
    struct Result {
        std::vector<std::string> text;
        TextRuns runs;
        std::string panic;
        bool general = false, done = false;
        bool late_panic = false; // the panic came from the liftover stage, not from loading / parsing the chunk
    };
    std::vector<Result> res(n_chunks);
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<size_t> next{0};
    std::atomic<bool> stop{false};
    const char *we = getenv("RB_PIPE_WORKERS");
    // (2 workers: the run is bound by the ONE writer of the output file; the sweep of profiles/r03_e2e.md -- 1 to 6 workers, chunks of
    //  256 MB to 1 GB -- ends within the noise of the box for everything from 1 worker up, and more workers only load the host)
    const unsigned W = (unsigned)std::min<size_t>(n_chunks, (size_t)std::max(1, we ? atoi(we) : 2));
    std::vector<std::thread> workers;
    for (unsigned w = 0; w < W; w++)
        workers.emplace_back([&]() {
            std::unique_ptr<Engine> eng;
            for (;;) {
                const size_t k = next++;
                if (k >= n_chunks || stop) break;
                {   // (at most W + 1 chunks ahead of the sink: their outputs are held in memory)
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || k < W + 1 || (res[k - W - 1].done && res[k - W - 1].text.empty()); });
                }
                Result R;
                bool loaded = false;
                try {
                    if (!eng) {
                        eng.reset(new Engine(device));
                        eng->bsearch_policy = bsearch_policy;
                    }
                    const std::pair<uint64_t, uint64_t> range(cut[k], cut[k + 1]);
                    TextFile f;
                    if (!f.load(paf_path, &range)) {
                        R.general = true;
                    } else {
                        const size_t n = f.recs.size();
                        R.runs.contigs.assign(f.contig_id.size(), std::string());
                        for (const auto &kv : f.contig_id) R.runs.contigs[kv.second].assign(kv.first.data(), kv.first.size());
                        std::vector<uint32_t> w_contig(rgns.size());
                        std::vector<uint64_t> w_st(rgns.size()), w_en(rgns.size());
                        for (size_t i = 0; i < rgns.size(); i++) {
                            auto it = f.contig_id.find(std::string_view(rgns[i].name));
                            if (it == f.contig_id.end()) it = f.contig_id.emplace(std::string_view(rgns[i].name), (uint32_t)f.contig_id.size()).first;
                            w_contig[i] = it->second, w_st[i] = rgns[i].st, w_en[i] = rgns[i].en;
                        }
                        std::vector<uint8_t> cig_status(n ? n : 1);
                        std::vector<rb_reduce_row> red(n);
                        std::vector<rb_norm_row> norm(n);
                        TextRows TR;
                        rb_counters cnt;
                        if (is_break)
                            eng->check(rb_host_break_text(eng->ctx(), n, (const uint8_t *)f.all.data(), f.text_bytes, f.cig_off.data(), f.cig_end.data(),
                                                          f.t_st.data(), f.t_en.data(), f.q_st.data(), f.q_en.data(), f.strand.data(), break_length,
                                                          bsearch_policy, cig_status.data(), red.data(), norm.data(), &TR.rows, &TR.n_rows, &TR.toff,
                                                          &TR.text, &cnt),
                                       "rb_host_break_text");
                        else
                            eng->check(rb_host_liftover_text(eng->ctx(), n, (const uint8_t *)f.all.data(), f.text_bytes, f.cig_off.data(),
                                                             f.cig_end.data(), f.t_st.data(), f.t_en.data(), f.q_st.data(), f.q_en.data(), f.strand.data(),
                                                             f.contig.data(), rgns.size(), w_contig.data(), w_st.data(), w_en.data(), bsearch_policy,
                                                             cig_status.data(), red.data(), norm.data(), &TR.rows, &TR.n_rows, &TR.toff, &TR.text, &cnt),
                                       "rb_host_liftover_text");
                        if (!f.check_loaded(cig_status, red)) {
                            R.general = true;
                        } else {
                            loaded = true;
                            for (size_t i = 0; i < n; i++) panic_on(norm[i].status, "aligned_pairs", i);
                            RunMarks marks;
                            R.text = assemble_lines(f, norm, TR, is_break ? nullptr : &rgns, &marks);
                            marks_to_runs(marks, R.text, R.runs);
                            if (is_break) { // (record order: one run)
                                uint64_t total = 0;
                                for (const std::string &t : R.text) total += t.size();
                                R.runs.contigs.assign(1, std::string());
                                R.runs.runs.assign(1, {0u, total});
                            }
                        }
                    }
                } catch (const Panic &e) {
                    R.panic = e.what();
                    if (R.panic.empty()) R.panic = "panic";
                    R.late_panic = loaded;
                } catch (const std::exception &e) {
                    R.panic = std::string("\x01") + e.what(); // (not a reference panic: the caller rethrows it as a runtime error)
                }
                R.done = true;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    res[k] = std::move(R);
                }
                cv.notify_all();
            }
        });
    bool ok = true;
    std::string panic;
    for (size_t k = 0; k < n_chunks && ok; k++) {
        Result R;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return res[k].done; });
            R = std::move(res[k]);
            res[k].text.clear();
            res[k].done = true;
        }
        cv.notify_all();
        // A panic of the liftover stage in chunk k must not win over a load / parse panic in a later chunk: the reference reads the whole
        // file (Paf::from_file, paf.rs:70) before it lifts anything.  The whole-file route keeps that order: it takes over.
        if (!R.panic.empty() && R.late_panic) { ok = false; break; }
        if (!R.panic.empty()) { panic = R.panic; ok = false; break; }
        if (R.general) { ok = false; break; }
        g_pipeline_started = true;
        if (!sink(R.text, R.runs)) { ok = false; break; } // (the output so far is not the reference's: no point in computing the rest)
    }
    stop = true;
    cv.notify_all();
    for (auto &t : workers) t.join();
    if (!panic.empty()) {
        if (panic[0] == '\x01') throw std::runtime_error(panic.substr(1));
        throw Panic(panic);
    }
    return ok;
}

// main.rs:218-230 (trim-paf), text in -> text out with the batch resident on the device across the passes of
// Paf::overlapping_paf_recs (paf.rs:210-305): the file's cg:Z: values are parsed on the device once; every pass picks its pairs on
// the host from three small arrays (query group, q_st, q_en), cuts them in place (rb_dev_overlap_split writing behind the ops in use,
// rb_dev_apply_pairs) and brings back the pair rows; the CIGARs are printed by the device at the end.  false = the caller takes the
// record-based path (a file the text loader declines, or not enough room for the clips of an unusually long run of passes).
bool trim_file_text(Engine &eng, const std::string &paf_path, int match_score, int diff_score, int indel_score, bool remove_contained,
                    std::vector<std::string> &out_text) {
    TextFile &f = *new TextFile; // (one-shot command: left to the end of the process, like the buffers of the other text routes)
    if (!f.load(paf_path)) return false;
    double tl = now_s();
    const size_t n = f.recs.size();
    if (n == 0) {
        out_text.clear();
        return true;
    }
    rb_ctx *ctx = eng.ctx();
    struct Dev { // device allocations of this call
        rb_ctx *ctx;
        std::vector<void *> owned;
        ~Dev() { for (void *q : owned) rb_dev_free(ctx, q); }
        void *take(Engine &e, size_t bytes) {
            void *d = nullptr;
            e.check(rb_dev_alloc(ctx, bytes + 256, &d), "rb_dev_alloc");
            owned.push_back(d);
            return d;
        }
    } D{ctx, {}};
    auto up = [&](const void *host, size_t bytes) {
        void *d = D.take(eng, bytes);
        if (bytes) eng.check(rb_dev_upload(ctx, d, host, bytes), "rb_dev_upload");
        return d;
    };
    uint64_t cig_bytes = 0;
    for (size_t r = 0; r < n; r++) cig_bytes += f.cig_end[r] - f.cig_off[r];
    const uint64_t ops_bound = cig_bytes / 2 + 8;        // an op is at least two characters
    const uint64_t ops_cap = 3 * ops_bound + 4096;       // the file's ops + room for the clips of the passes (every pass rewrites at most all of them)
    uint8_t *d_text = (uint8_t *)up(f.all.data(), f.text_bytes + 32 <= f.all.size() ? f.text_bytes : f.text_bytes);
    const uint64_t *d_coff = (const uint64_t *)up(f.cig_off.data(), n * 8), *d_cend = (const uint64_t *)up(f.cig_end.data(), n * 8);
    uint64_t *d_opoff = (uint64_t *)D.take(eng, (n + 2) * 8);
    uint32_t *d_ops = (uint32_t *)D.take(eng, (size_t)ops_cap * 4);
    uint8_t *d_status = (uint8_t *)D.take(eng, n + 1);
    void *d_scr = D.take(eng, rb_text_scratch_bytes(std::max<uint64_t>(n, 2 * n)));
    eng.check(rb_dev_parse_cigars(ctx, d_text, d_coff, d_cend, n, d_opoff, d_ops, ops_bound, d_status, d_scr), "rb_dev_parse_cigars");
    std::vector<uint8_t> cig_status(n);
    eng.check(rb_dev_download(ctx, cig_status.data(), d_status, n), "rb_dev_download");
    std::vector<uint64_t> op_off(n + 1);
    eng.check(rb_dev_download(ctx, op_off.data(), d_opoff, (n + 1) * 8), "rb_dev_download");
    rb_batch_view v;
    v.n_rec = n, v.n_ops = op_off[n], v.ops = d_ops, v.op_off = d_opoff;
    v.t_st = (const uint64_t *)up(f.t_st.data(), n * 8), v.t_en = (const uint64_t *)up(f.t_en.data(), n * 8);
    v.q_st = (const uint64_t *)up(f.q_st.data(), n * 8), v.q_en = (const uint64_t *)up(f.q_en.data(), n * 8);
    v.strand = (const uint8_t *)up(f.strand.data(), n), v.contig = (const uint32_t *)up(f.contig.data(), n * 4);
    rb_reduce_row *d_red = (rb_reduce_row *)D.take(eng, n * sizeof(rb_reduce_row));
    rb_norm_row *d_norm = (rb_norm_row *)D.take(eng, n * sizeof(rb_norm_row));
    bool all_ok = true;
    for (size_t r = 0; r < n; r++) all_ok &= cig_status[r] == RB_TEXT_OK;
    std::vector<rb_reduce_row> red(n);
    std::vector<rb_norm_row> norm(n);
    if (all_ok) {
        eng.check(rb_dev_scan_records(ctx, &v, d_red, d_norm), "rb_dev_scan_records");
        eng.check(rb_dev_download(ctx, red.data(), d_red, n * sizeof(rb_reduce_row)), "rb_dev_download");
        eng.check(rb_dev_download(ctx, norm.data(), d_norm, n * sizeof(rb_norm_row)), "rb_dev_download");
    }
    if (!f.check_loaded(cig_status, red)) return false;                                  // the panics of Paf::from_file, in its order
    for (size_t i = 0; i < n; i++) panic_on(norm[i].status, "remove_trailing_indels", i); // paf.rs:218-220
    lap("  text -> ops, scan (device)", tl);
    // records stably ordered by query name (:223); a group = a run of equal names.  Names are hashed into groups in order of first
    // appearance (one pass), only the distinct names are sorted, and the groups are laid out in that order: file order inside a group
    std::vector<uint32_t> order(n);
    std::vector<uint64_t> grp_off;
    {
        auto qname = [&](uint32_t i) { return f.name(f.recs[i].q_name, f.recs[i].q_name_n); };
        std::unordered_map<std::string_view, uint32_t> gid_of;
        gid_of.reserve(n / 2 + 16);
        std::vector<uint32_t> gid(n), first_rec, count;
        for (size_t i = 0; i < n; i++) {
            auto it = gid_of.emplace(qname((uint32_t)i), (uint32_t)first_rec.size());
            if (it.second) first_rec.push_back((uint32_t)i), count.push_back(0);
            gid[i] = it.first->second;
            count[gid[i]]++;
        }
        const size_t G = first_rec.size();
        std::vector<uint32_t> by_name(G);
        std::iota(by_name.begin(), by_name.end(), 0u);
        std::sort(by_name.begin(), by_name.end(), [&](uint32_t x, uint32_t y) { return qname(first_rec[x]) < qname(first_rec[y]); });
        grp_off.assign(G + 1, 0);
        std::vector<uint64_t> at(G);
        for (size_t k = 0; k < G; k++) at[by_name[k]] = grp_off[k], grp_off[k + 1] = grp_off[k] + count[by_name[k]];
        for (size_t i = 0; i < n; i++) order[at[gid[i]]++] = (uint32_t)i;
    }
    const size_t n_groups = grp_off.size() - 1;
    lap("  sort by query name", tl);
    // ---- the passes (paf.rs:286-288), on the device: per pass the pair scan + selection (rb_dev_trim_select), the split + clip of
    //      the chosen pairs in place behind the ops in use (rb_dev_overlap_split, rb_dev_apply_pairs); the host reads 64 bytes ----
    const uint32_t *d_order = (const uint32_t *)up(order.data(), n * 4);
    const uint64_t *d_grp = (const uint64_t *)up(grp_off.data(), (n_groups + 1) * 8);
    uint8_t *d_contained = (uint8_t *)D.take(eng, n + 1);
    eng.check(rb_dev_memset(ctx, d_contained, 0, n + 1), "rb_dev_memset");
    uint32_t *d_left = (uint32_t *)D.take(eng, n_groups * 4 + 4), *d_right = (uint32_t *)D.take(eng, n_groups * 4 + 4);
    uint64_t *d_poff = (uint64_t *)D.take(eng, n_groups * 8 + 8);
    rb_pair_row *d_rows = (rb_pair_row *)D.take(eng, (n_groups + 1) * sizeof(rb_pair_row));
    rb_trim_pass *d_pass = (rb_trim_pass *)D.take(eng, sizeof(rb_trim_pass));
    void *d_sel = D.take(eng, rb_trim_select_scratch_bytes(n_groups));
    uint64_t cursor = (v.n_ops + 31) & ~(uint64_t)31;
    for (int pass = 0;; pass++) {
        if (pass > 100000) throw Panic("trim-paf did not converge");
        eng.check(rb_dev_trim_select(ctx, n, n_groups, d_order, d_grp, d_norm, cursor, d_contained, d_left, d_right, d_poff, d_pass, d_sel), "rb_dev_trim_select");
        rb_trim_pass hp;
        eng.check(rb_dev_download(ctx, &hp, d_pass, sizeof hp), "rb_dev_download");
        if (hp.ops_end + 64 > ops_cap) return false; // (nothing has been printed: the record-based path starts over)
        if (hp.n_pairs) {
            // (regular records are cut where they are -- two words a record -- instead of being copied behind the ops in use; RB_TRIM_COPY=1: the copy)
            static const int in_place = getenv("RB_TRIM_COPY") ? 0 : RB_TRIM_IN_PLACE;
            eng.check(rb_dev_overlap_split(ctx, &v, d_norm, hp.n_pairs, d_left, d_right, d_poff, match_score, diff_score, indel_score, eng.bsearch_policy | in_place,
                                           d_rows, d_ops),
                      "rb_dev_overlap_split");
            eng.check(rb_dev_apply_pairs(ctx, hp.n_pairs, d_left, d_right, d_rows, d_opoff, d_norm), "rb_dev_apply_pairs");
            eng.check(rb_dev_trim_check(ctx, hp.n_pairs, d_rows, d_pass), "rb_dev_trim_check");
            rb_trim_pass chk;
            eng.check(rb_dev_download(ctx, &chk, d_pass, sizeof chk), "rb_dev_download");
            if (chk.bad_status != RB_ST_OK) throw Panic("trim_overlapping_pafs: a pair of pass " + std::to_string(pass) + " has status " + std::to_string(chk.bad_status));
            cursor = (hp.ops_end + 31) & ~(uint64_t)31;
        }
        lap("  pass (pair scan, selection, split + clip: all on the device)", tl);
        if (hp.n_deferred == 0) break; // :286-288
    }
    // the records as the passes left them
    struct Cur { uint64_t off, t_st, t_en, q_st, q_en; uint32_t n, nmatch, aln_len; };
    std::vector<Cur> cur(n);
    std::vector<uint8_t> contained_rec(n);
    {
        std::vector<rb_norm_row> fin(n);
        std::vector<uint64_t> foff(n);
        eng.check(rb_dev_download(ctx, fin.data(), d_norm, n * sizeof(rb_norm_row)), "rb_dev_download");
        eng.check(rb_dev_download(ctx, foff.data(), d_opoff, n * 8), "rb_dev_download");
        eng.check(rb_dev_download(ctx, contained_rec.data(), d_contained, n), "rb_dev_download");
        for (size_t i = 0; i < n; i++)
            cur[i] = {foff[i] + fin[i].first_op, fin[i].t_st, fin[i].t_en, fin[i].q_st, fin[i].q_en, fin[i].n_ops, fin[i].nmatch, fin[i].aln_len};
    }
    std::vector<char> contained(n, 0); // by position in `order`
    for (size_t p0 = 0; p0 < n; p0++) contained[p0] = (char)contained_rec[order[p0]];
    lap("  final rows D2H", tl);
    // ---- print: kept records in the sorted order, CIGAR text from the device ----
    std::vector<uint32_t> keep;
    keep.reserve(n);
    for (size_t p0 = 0; p0 < n; p0++)
        if (!(remove_contained && contained[p0])) keep.push_back(order[p0]); // :289-301 (the flags of the last pass)
    const size_t nk = keep.size();
    std::vector<uint64_t> first(nk);
    std::vector<uint32_t> count(nk);
    uint64_t out_ops_total = 0;
    for (size_t k = 0; k < nk; k++) first[k] = cur[keep[k]].off, count[k] = cur[keep[k]].n, out_ops_total += count[k];
    const uint64_t *d_first = (const uint64_t *)up(first.data(), nk * 8);
    const uint32_t *d_count = (const uint32_t *)up(count.data(), nk * 4);
    uint64_t *d_toff = (uint64_t *)D.take(eng, (nk + 2) * 8);
    const uint64_t text_cap = 11 * out_ops_total + 16;
    uint8_t *d_out = (uint8_t *)D.take(eng, text_cap);
    eng.check(rb_dev_format_cigars(ctx, d_ops, nullptr, nk, d_first, d_count, nullptr, nullptr, d_toff, d_out, text_cap, d_scr), "rb_dev_format_cigars");
    std::vector<uint64_t> &toff = *new std::vector<uint64_t>(nk + 1);
    eng.check(rb_dev_download(ctx, toff.data(), d_toff, (nk + 1) * 8), "rb_dev_download");
    std::vector<uint8_t> &text = *new std::vector<uint8_t>((size_t)toff[nk] + 1);
    if (toff[nk]) eng.check(rb_dev_download(ctx, text.data(), d_out, (size_t)toff[nk]), "rb_dev_download");
    lap("  print cigars (device) + D2H", tl);
    const unsigned TO = parallel_chunk_count(nk);
    out_text.assign(TO, std::string());
    parallel_chunks(nk, [&](unsigned t, size_t lo, size_t hi) {
        std::string &o = out_text[t];
        size_t est = 0;
        for (size_t k = lo; k < hi; k++) est += 160 + (size_t)(toff[k + 1] - toff[k]);
        o.reserve(est);
        advise_huge(o.data(), est);
        char nb[24];
        auto num = [&](uint64_t x) {
            auto r = std::to_chars(nb, nb + sizeof nb, x);
            o.append(nb, r.ptr);
        };
        for (size_t k = lo; k < hi; k++) {
            const uint32_t i = keep[k];
            const HeaderOnly &s = f.recs[i];
            const Cur &c = cur[i];
            o.append(f.all.data() + s.q_name, s.q_name_n); o += '\t'; num(s.q_len); o += '\t'; num(c.q_st); o += '\t'; num(c.q_en); o += '\t';
            o += s.strand; o += '\t'; o.append(f.all.data() + s.t_name, s.t_name_n); o += '\t'; num(s.t_len); o += '\t'; num(c.t_st); o += '\t';
            num(c.t_en); o += '\t'; num(c.nmatch); o += '\t'; num(c.aln_len); o += '\t'; num(s.mapq); o += "\tid:Z:";
            o += f.stripped_suffix(i, norm[i]);
            o += "\tcg:Z:";
            o.append((const char *)text.data() + toff[k], (size_t)(toff[k + 1] - toff[k]));
            o += '\n';
        }
    });
    lap("  assemble lines", tl);
    return true;
}

// main.rs:176-182 (invert), text in -> text out: cg:Z: values parsed, swapped (paf.rs:1050-1094) and printed on the device, the host
// exchanges the query and target columns
bool invert_file_text(Engine &eng, const std::string &paf_path, std::vector<std::string> &out_text) {
    TextFile &f = *new TextFile; // (one-shot command: left to the end of the process, like the buffers of the other text routes)
    if (!f.load(paf_path)) return false;
    double tl = now_s();
    const size_t n = f.recs.size();
    out_text.clear();
    if (n == 0) return true;
    rb_ctx *ctx = eng.ctx();
    struct Dev { // device allocations of this call
        rb_ctx *ctx;
        std::vector<void *> owned;
        ~Dev() { for (void *q : owned) rb_dev_free(ctx, q); }
        void *take(Engine &e, size_t bytes) {
            void *d = nullptr;
            e.check(rb_dev_alloc(ctx, bytes + 256, &d), "rb_dev_alloc");
            owned.push_back(d);
            return d;
        }
    } D{ctx, {}};
    auto up = [&](const void *host, size_t bytes) {
        void *d = D.take(eng, bytes);
        if (bytes) eng.check(rb_dev_upload(ctx, d, host, bytes), "rb_dev_upload");
        return d;
    };
    uint64_t cig_bytes = 0;
    for (size_t r = 0; r < n; r++) cig_bytes += f.cig_end[r] - f.cig_off[r];
    const uint64_t ops_bound = cig_bytes / 2 + 8; // an op is at least two characters
    uint8_t *d_text = (uint8_t *)up(f.all.data(), f.text_bytes);
    const uint64_t *d_coff = (const uint64_t *)up(f.cig_off.data(), n * 8), *d_cend = (const uint64_t *)up(f.cig_end.data(), n * 8);
    uint64_t *d_opoff = (uint64_t *)D.take(eng, (n + 2) * 8);
    uint32_t *d_ops = (uint32_t *)D.take(eng, (size_t)ops_bound * 4 + 256);
    uint8_t *d_status = (uint8_t *)D.take(eng, n + 1);
    void *d_scr = D.take(eng, rb_text_scratch_bytes(n));
    eng.check(rb_dev_parse_cigars(ctx, d_text, d_coff, d_cend, n, d_opoff, d_ops, ops_bound, d_status, d_scr), "rb_dev_parse_cigars");
    std::vector<uint8_t> cig_status(n);
    eng.check(rb_dev_download(ctx, cig_status.data(), d_status, n), "rb_dev_download");
    std::vector<uint64_t> op_off(n + 1);
    eng.check(rb_dev_download(ctx, op_off.data(), d_opoff, (n + 1) * 8), "rb_dev_download");
    rb_batch_view v;
    v.n_rec = n, v.n_ops = op_off[n], v.ops = d_ops, v.op_off = d_opoff;
    v.t_st = (const uint64_t *)up(f.t_st.data(), n * 8), v.t_en = (const uint64_t *)up(f.t_en.data(), n * 8);
    v.q_st = (const uint64_t *)up(f.q_st.data(), n * 8), v.q_en = (const uint64_t *)up(f.q_en.data(), n * 8);
    v.strand = (const uint8_t *)up(f.strand.data(), n), v.contig = (const uint32_t *)up(f.contig.data(), n * 4);
    bool all_ok = true;
    for (size_t r = 0; r < n; r++) all_ok &= cig_status[r] == RB_TEXT_OK;
    std::vector<rb_reduce_row> red(n);
    if (all_ok) {
        rb_reduce_row *d_red = (rb_reduce_row *)D.take(eng, n * sizeof(rb_reduce_row));
        eng.check(rb_dev_scan_records(ctx, &v, d_red, nullptr), "rb_dev_scan_records");
        eng.check(rb_dev_download(ctx, red.data(), d_red, n * sizeof(rb_reduce_row)), "rb_dev_download");
    }
    if (!f.check_loaded(cig_status, red)) return false; // the panics of Paf::from_file, in its order
    lap("  text -> ops, scan (device)", tl);
    uint32_t *d_swapped = (uint32_t *)D.take(eng, (size_t)v.n_ops * 4 + 256);
    eng.check(rb_dev_swap(ctx, &v, d_swapped), "rb_dev_swap");
    std::vector<uint32_t> count(n);
    for (size_t i = 0; i < n; i++) count[i] = (uint32_t)(op_off[i + 1] - op_off[i]);
    const uint32_t *d_count = (const uint32_t *)up(count.data(), n * 4);
    uint64_t *d_toff = (uint64_t *)D.take(eng, (n + 2) * 8);
    const uint64_t text_cap = 11 * v.n_ops + 16;
    uint8_t *d_out = (uint8_t *)D.take(eng, text_cap);
    eng.check(rb_dev_format_cigars(ctx, d_swapped, nullptr, n, d_opoff, d_count, nullptr, nullptr, d_toff, d_out, text_cap, d_scr), "rb_dev_format_cigars");
    std::vector<uint64_t> &toff = *new std::vector<uint64_t>(n + 1);
    eng.check(rb_dev_download(ctx, toff.data(), d_toff, (n + 1) * 8), "rb_dev_download");
    std::vector<uint8_t> &text = *new std::vector<uint8_t>((size_t)toff[n] + 1);
    if (toff[n]) eng.check(rb_dev_download(ctx, text.data(), d_out, (size_t)toff[n]), "rb_dev_download");
    lap("  swap + print cigars (device) + D2H", tl);
    const unsigned TO = parallel_chunk_count(n);
    out_text.assign(TO, std::string());
    parallel_chunks(n, [&](unsigned t, size_t lo, size_t hi) {
        std::string &o = out_text[t];
        size_t est = 0;
        for (size_t k = lo; k < hi; k++) est += 160 + (size_t)(toff[k + 1] - toff[k]);
        o.reserve(est);
        advise_huge(o.data(), est);
        char nb[24];
        auto num = [&](uint64_t x) {
            auto r = std::to_chars(nb, nb + sizeof nb, x);
            o.append(nb, r.ptr);
        };
        for (size_t k = lo; k < hi; k++) {
            const HeaderOnly &s = f.recs[k];
            o.append(f.all.data() + s.t_name, s.t_name_n); o += '\t'; num(s.t_len); o += '\t'; num(s.t_st); o += '\t'; num(s.t_en); o += '\t';
            o += s.strand; o += '\t'; o.append(f.all.data() + s.q_name, s.q_name_n); o += '\t'; num(s.q_len); o += '\t'; num(s.q_st); o += '\t';
            num(s.q_en); o += '\t'; num(red[k].nmatch); o += '\t'; num(red[k].aln_len); o += '\t'; num(s.mapq); o += "\tid:Z:\tcg:Z:";
            o.append((const char *)text.data() + toff[k], (size_t)(toff[k + 1] - toff[k]));
            o += '\n';
        }
    });
    lap("  assemble lines", tl);
    return true;
}

// main.rs:50-58 (stats --paf), text in -> stats lines
bool stats_file_text(Engine &eng, const std::string &paf_path, bool qbed, std::vector<std::string> &out_text) {
    TextFile f;
    if (!f.load(paf_path)) return false;
    const size_t n = f.recs.size();
    std::vector<uint8_t> cig_status(n ? n : 1);
    std::vector<rb_reduce_row> red(n);
    eng.check(rb_host_scan_text(eng.ctx(), n, (const uint8_t *)f.all.data(), f.text_bytes, f.cig_off.data(), f.cig_end.data(), f.t_st.data(),
                                f.t_en.data(), f.q_st.data(), f.q_en.data(), f.strand.data(), cig_status.data(), red.data(), nullptr),
              "rb_host_scan_text");
    if (!f.check_loaded(cig_status, red)) return false;
    for (size_t i = 0; i < n; i++)
        if (red[i].flags & RB_F_HAS_M) { // bamstats.rs:145-153
            fprintf(stderr, "\r⚠ warning: cigar string contains 'M', assuming mismatch since there is no MD tag.");
            break;
        }
    const unsigned TO = parallel_chunk_count(n);
    out_text.assign(TO, std::string());
    parallel_chunks(n, [&](unsigned t, size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const HeaderOnly &r = f.recs[i];
            Stats s;
            s.r_nm.assign(f.all.data() + r.t_name, r.t_name_n), s.r_len = (int64_t)r.t_len, s.r_st = (int64_t)r.t_st, s.r_en = (int64_t)r.t_en;
            s.q_nm.assign(f.all.data() + r.q_name, r.q_name_n), s.q_len = (int64_t)r.q_len, s.q_st = (int64_t)r.q_st, s.q_en = (int64_t)r.q_en;
            s.strand = r.strand;
            s.equal = red[i].equal, s.diff = red[i].diff, s.ins = red[i].ins, s.del = red[i].del, s.matches = red[i].matches;
            s.ins_events = red[i].ins_events, s.del_events = red[i].del_events;
            s.id_by_all = red[i].id_by_all, s.id_by_events = red[i].id_by_events, s.id_by_matches = red[i].id_by_matches;
            out_text[t] += cigar_stats_line(s, qbed);
        }
    });
    return true;
}

std::vector<PafRecord> break_paf_on_indels(Engine &eng, const std::vector<PafRecord> &paf_recs, uint32_t break_length) {
    LiftResult L;
    run_break(eng, paf_recs, break_length, L);
    return rows_to_records(paf_recs, L.norm, L.rows, L.n_rows, L.out, nullptr); // id = paf.id (liftover.rs:194)
}
std::vector<std::string> break_paf_on_indels_text(Engine &eng, const std::vector<PafRecord> &paf_recs, uint32_t break_length) {
    LiftResult L;
    run_break(eng, paf_recs, break_length, L);
    return rows_to_text(paf_recs, L.norm, L.rows, L.n_rows, L.out, nullptr);
}

std::vector<Stats> stats_from_paf(Engine &eng, const std::vector<PafRecord> &recs) {
    HostBatch b(recs);
    std::vector<rb_reduce_row> red(b.n());
    eng.check(rb_host_scan_records(eng.ctx(), b.n(), b.ops.data(), b.op_off.data(), b.t_st.data(), b.t_en.data(), b.q_st.data(), b.q_en.data(),
                                   b.strand.data(), red.data(), nullptr),
              "rb_host_scan_records");
    std::vector<Stats> out(recs.size());
    bool warned = false;
    for (size_t i = 0; i < recs.size(); i++) {
        const PafRecord &r = recs[i];
        Stats &s = out[i];
        s.r_nm = r.t_name, s.r_len = (int64_t)r.t_len, s.r_st = (int64_t)r.t_st, s.r_en = (int64_t)r.t_en;
        s.q_nm = r.q_name, s.q_len = (int64_t)r.q_len, s.q_st = (int64_t)r.q_st, s.q_en = (int64_t)r.q_en;
        s.strand = r.strand;
        s.equal = red[i].equal, s.diff = red[i].diff, s.ins = red[i].ins, s.del = red[i].del, s.matches = red[i].matches;
        s.ins_events = red[i].ins_events, s.del_events = red[i].del_events;
        s.id_by_all = red[i].id_by_all, s.id_by_events = red[i].id_by_events, s.id_by_matches = red[i].id_by_matches;
        if ((red[i].flags & RB_F_HAS_M) && !warned) { // bamstats.rs:145-153
            fprintf(stderr, "\r⚠ warning: cigar string contains 'M', assuming mismatch since there is no MD tag.");
            warned = true;
        }
    }
    return out;
}

void parse_md_for_stats(const std::string &md, uint32_t out[4]) { // regex (\d+)|([A-Z])|(\^[A-Z]+), left to right
    uint32_t m = 0, mm = 0, ic = 0, ib = 0;
    size_t p = 0;
    const size_t n = md.size();
    while (p < n) {
        const char c = md[p];
        if (c >= '0' && c <= '9') {
            uint64_t v = 0;
            while (p < n && md[p] >= '0' && md[p] <= '9') v = v * 10 + (uint64_t)(md[p++] - '0');
            m += (uint32_t)v;
        } else if (c >= 'A' && c <= 'Z') {
            mm++;
            p++;
        } else if (c == '^' && p + 1 < n && md[p + 1] >= 'A' && md[p + 1] <= 'Z') {
            size_t q = p + 1;
            while (q < n && md[q] >= 'A' && md[q] <= 'Z') q++;
            ib += (uint32_t)(q - p) - 1;
            ic++;
            p = q;
        } else {
            p++;
        }
    }
    out[0] = m, out[1] = mm, out[2] = ic, out[3] = ib;
}

namespace {
struct BamRec {
    std::string qname, md;
    bool has_md = false;
    int32_t ref_id = -1;
    int64_t pos = 0;
    uint32_t flag = 0, l_seq = 0;
    size_t cig0 = 0, ncig = 0; // slice of the shared ops array
};
uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// The decompressed bytes of a BAM file.  A BGZF file is a series of independent gzip members of at most 64 KiB, each carrying its
// compressed size (BC subfield) and its uncompressed size (ISIZE): the file is read by all host threads, the members are found by
// hopping from header to header, and every thread inflates its share straight into place.  Anything else (stdin, plain gzip)
// streams through zlib.
struct BamStream {
    std::unique_ptr<uint8_t[]> buf;
    size_t n = 0, pos = 0;
    explicit BamStream(const std::string &path) {
        if (path != "-" && inflate_bgzf(path)) return;
        gzFile f = path == "-" ? gzdopen(0, "rb") : gzopen(path.c_str(), "rb");
        if (!f) throw Panic("Failed to open " + path); // main.rs:63, nucfreq.rs:115
        gzbuffer(f, 1 << 20);
        std::vector<uint8_t> all, chunk(1 << 24);
        int r;
        while ((r = gzread(f, chunk.data(), (unsigned)chunk.size())) > 0) all.insert(all.end(), chunk.begin(), chunk.begin() + r);
        gzclose(f);
        n = all.size();
        buf.reset(new uint8_t[n + 1]);
        if (n) memcpy(buf.get(), all.data(), n);
    }
    bool underflow = false; // a read ran past the loaded bytes
    bool exact(void *dst, size_t k) {
        if (k > n - pos) {
            underflow = true;
            return false;
        }
        memcpy(dst, buf.get() + pos, k);
        pos += k;
        return true;
    }
    struct Member { size_t off, data, clen, out, isize; }; // file offset of the gzip member, of its deflate data; sizes; place in buf
    std::vector<Member> mem;
    // the members found in raw[0 .. sz) (file offset of raw[0] = base): walk the headers, inflate all of them on all host threads
    bool inflate_members(const uint8_t *d, size_t sz, size_t base) {
        mem.clear();
        size_t off = 0, out = 0;
        while (off < sz) {
            if (sz - off < 18 || d[off] != 0x1f || d[off + 1] != 0x8b || d[off + 2] != 8 || !(d[off + 3] & 4)) return false;
            const size_t xlen = (size_t)d[off + 10] | ((size_t)d[off + 11] << 8);
            if (sz - off < 12 + xlen) return false;
            size_t bsize = 0;
            for (size_t x = off + 12; x + 4 <= off + 12 + xlen;) { // extra subfields: SI1 SI2 SLEN data
                const size_t slen = (size_t)d[x + 2] | ((size_t)d[x + 3] << 8);
                if (d[x] == 'B' && d[x + 1] == 'C' && slen == 2 && x + 6 <= off + 12 + xlen) bsize = ((size_t)d[x + 4] | ((size_t)d[x + 5] << 8)) + 1;
                x += 4 + slen;
            }
            if (bsize < 12 + xlen + 8 || bsize > sz - off) return false;
            const size_t isize = le32(d + off + bsize - 4);
            mem.push_back({base + off, off + 12 + xlen, bsize - xlen - 12 - 8, out, isize});
            out += isize;
            off += bsize;
        }
        buf.reset(new uint8_t[out + 1]);
        advise_huge(buf.get(), out);
        n = out, pos = 0;
        std::atomic<bool> ok{true};
        parallel_chunks(mem.size(), [&](unsigned, size_t lo, size_t hi) {
            z_stream zs;
            for (size_t m = lo; m < hi && ok; m++) {
                if (mem[m].isize == 0) continue; // (the empty end-of-file member)
                memset(&zs, 0, sizeof zs);
                if (inflateInit2(&zs, -15) != Z_OK) { ok = false; break; }
                zs.next_in = const_cast<Bytef *>(d + mem[m].data);
                zs.avail_in = (uInt)mem[m].clen;
                zs.next_out = buf.get() + mem[m].out;
                zs.avail_out = (uInt)mem[m].isize;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0) ok = false;
                inflateEnd(&zs);
            }
        });
        return ok;
    }
    bool inflate_bgzf(const std::string &path) {
        TextBuf raw;
        try {
            raw = read_text(path, true);
        } catch (const Panic &) {
            return false;
        }
        return inflate_members((const uint8_t *)raw.data(), raw.size(), 0);
    }
    BamStream() = default;
    // place in buf of a BGZF virtual offset (file offset of a member << 16 | offset inside its data); ~0 if the member is not loaded
    size_t upos(uint64_t v) const {
        const size_t c = (size_t)(v >> 16);
        size_t lo = 0, hi = mem.size();
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (mem[mid].off < c) lo = mid + 1; else hi = mid;
        }
        if (lo >= mem.size() || mem[lo].off != c) return ~(size_t)0;
        return mem[lo].out + (size_t)(v & 0xFFFF);
    }
};
bool gz_exact(BamStream &f, void *buf, size_t n) { return f.exact(buf, n); }
// a header field that must be there: a file that ends inside its header is an error in htslib, a panic in the reference
void bam_need(BamStream &f, void *buf, size_t n, const std::string &path) {
    if (!f.exact(buf, n)) throw Panic("truncated BAM header in " + path);
}
// the fixed part of one BAM record and the places of its variable parts, with the bounds htslib's bam_read1 enforces
// (block_size >= 32; l_read_name + 4 n_cigar + (l_seq + 1) / 2 + l_seq <= block_size - 32; sam.c): a record that breaks them is a
// read error there and a panic at the reference's rec.unwrap() (main.rs:72)
struct BamCore {
    int32_t tid;
    int64_t pos;
    uint32_t l_rn, n_cig, flag, l_seq;
    const uint8_t *cg, *sq;
    size_t aux0;
};
BamCore bam_core(const uint8_t *rec, size_t bs) {
    if (bs < 32) throw Panic("invalid BAM record: block_size < 32");
    BamCore c;
    c.tid = (int32_t)le32(rec);
    c.pos = (int32_t)le32(rec + 4);
    c.l_rn = rec[8];
    c.n_cig = (uint32_t)rec[12] | ((uint32_t)rec[13] << 8);
    c.flag = (uint32_t)rec[14] | ((uint32_t)rec[15] << 8);
    c.l_seq = le32(rec + 16);
    const uint64_t need = 32ull + c.l_rn + 4ull * c.n_cig + ((uint64_t)c.l_seq + 1) / 2 + c.l_seq;
    if (need > bs) throw Panic("invalid BAM record: name, cigar and sequence do not fit the block");
    if (c.l_rn == 0 || rec[32 + c.l_rn - 1] != 0) throw Panic("invalid BAM record: read name is not NUL-terminated");
    c.cg = rec + 32 + c.l_rn;
    c.sq = c.cg + 4 * (size_t)c.n_cig;
    c.aux0 = (size_t)need;
    return c;
}
// the next record's bytes; false at a clean end of file; a file that ends inside a record is a panic (htslib: truncated file)
bool bam_next(BamStream &f, std::vector<uint8_t> &rec) {
    uint8_t b4[4];
    if (f.pos == f.n) return false;
    if (!f.exact(b4, 4)) throw Panic("truncated BAM file (inside a block_size field)");
    const uint32_t bs = le32(b4);
    if (bs > f.n - f.pos) throw Panic("truncated BAM file (inside a record)");
    rec.resize(bs);
    f.exact(rec.data(), bs);
    return true;
}
} // namespace

std::vector<Stats> cigar_stats_bam(Engine &eng, const std::string &path, std::string *trailing_panic) {
    BamStream f(path);
    uint8_t h8[8] = {0}, b4[4] = {0};
    if (!gz_exact(f, h8, 8) || memcmp(h8, "BAM\1", 4) != 0) throw Panic(path + " is not a BAM file");
    if (le32(h8 + 4) > f.n - f.pos) throw Panic("truncated BAM header in " + path); // (sizes are checked before they size anything)
    f.pos += le32(h8 + 4);                                                           // the SAM text is not used
    bam_need(f, b4, 4, path);
    const uint32_t n_ref = le32(b4);
    if ((uint64_t)n_ref * 8 > f.n - f.pos) throw Panic("truncated BAM header in " + path);
    std::vector<std::string> ref_nm(n_ref);
    std::vector<uint32_t> ref_len(n_ref);
    for (uint32_t i = 0; i < n_ref; i++) {
        bam_need(f, b4, 4, path);
        if (le32(b4) > f.n - f.pos) throw Panic("truncated BAM header in " + path);
        std::vector<char> nm(le32(b4) + 1, 0);
        bam_need(f, nm.data(), nm.size() - 1, path);
        ref_nm[i] = nm.data();
        bam_need(f, b4, 4, path);
        ref_len[i] = le32(b4);
    }
    // decode every mapped record; cigars go straight into one packed array (BAM's encoding IS the ABI's)
    std::vector<BamRec> recs;
    std::vector<uint32_t> ops;
    std::vector<uint8_t> rec;
    // The reference prints record by record (main.rs:71-76): what was decoded before a broken record, and the stats of the
    // records before one whose read_pos panics, are printed before the panic.  The first such panic is kept in *trailing_panic
    // (or thrown right away when the caller does not ask for it) and the records before it are returned.
    std::string pending;
    while (pending.empty()) {
        try {
            if (!bam_next(f, rec)) break;
        } catch (const Panic &e) {
            pending = e.what();
            break;
        }
        const size_t bs = rec.size();
        BamCore core;
        try {
            core = bam_core(rec.data(), bs);
        } catch (const Panic &e) {
            pending = e.what();
            break;
        }
        BamRec r;
        r.ref_id = core.tid;
        r.pos = core.pos;
        uint32_t n_cig = core.n_cig;
        r.flag = core.flag;
        r.l_seq = core.l_seq;
        if (r.flag & 4) continue; // main.rs:73 is_unmapped
        r.qname = (const char *)(rec.data() + 32);
        const uint8_t *cg = core.cg;
        const size_t aux0 = core.aux0;
        const uint8_t *cg_tag = nullptr;
        uint32_t cg_n = 0;
        for (size_t p = aux0; p + 3 <= bs;) { // aux fields: MD:Z and the CG:B,I long-cigar convention
            const uint8_t *tag = rec.data() + p;
            const char ty = (char)rec[p + 2];
            p += 3;
            if (ty == 'A' || ty == 'c' || ty == 'C') p += 1;
            else if (ty == 's' || ty == 'S') p += 2;
            else if (ty == 'i' || ty == 'I' || ty == 'f') p += 4;
            else if (ty == 'Z' || ty == 'H') {
                size_t q = p;
                while (q < bs && rec[q]) q++;
                if (tag[0] == 'M' && tag[1] == 'D' && ty == 'Z') {
                    r.md.assign((const char *)rec.data() + p, q - p);
                    r.has_md = true;
                }
                p = q + 1;
            } else if (ty == 'B') {
                if (p + 5 > bs) break;
                const char sub = (char)rec[p];
                const uint32_t cnt = le32(rec.data() + p + 1);
                const size_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4);
                if (p + 5 + es * (uint64_t)cnt > bs) break; // (an array that runs past the record: the aux walk stops, as htslib's does)
                if (tag[0] == 'C' && tag[1] == 'G' && sub == 'I') {
                    cg_tag = rec.data() + p + 5;
                    cg_n = cnt;
                }
                p += 5 + es * (size_t)cnt;
            } else {
                break;
            }
        }
        if (cg_tag && n_cig >= 1 && (le32(cg) & 15u) == RB_OP_S && (le32(cg) >> 4) == r.l_seq) {
            cg = cg_tag;
            n_cig = cg_n;
        }
        r.cig0 = ops.size();
        r.ncig = n_cig;
        for (uint32_t i = 0; i < n_cig; i++) ops.push_back(le32(cg + 4 * (size_t)i));
        recs.push_back(std::move(r));
    }

    const size_t n = recs.size();
    std::vector<uint64_t> op_off(n + 1, 0), zero(n, 0);
    std::vector<uint8_t> strand(n, (uint8_t)'+');
    for (size_t i = 0; i < n; i++) op_off[i + 1] = op_off[i] + recs[i].ncig;
    ops.resize(ops.size() + 4, 0);
    std::vector<rb_reduce_row> red(n);
    eng.check(rb_host_scan_records(eng.ctx(), n, ops.data(), op_off.data(), zero.data(), zero.data(), zero.data(), zero.data(), strand.data(),
                                   red.data(), nullptr),
              "rb_host_scan_records");
    std::vector<Stats> out(n);
    for (size_t i = 0; i < n; i++) try {
        const BamRec &r = recs[i];
        const uint32_t *cg = ops.data() + r.cig0;
        const size_t nc = r.ncig;
        auto opc = [&](size_t k) { return cg[k] & 15u; };
        auto len = [&](size_t k) { return (int64_t)(cg[k] >> 4); };
        Stats &s = out[i];
        s.r_nm = (r.ref_id >= 0 && (uint32_t)r.ref_id < n_ref) ? ref_nm[r.ref_id] : "*";
        s.r_len = (r.ref_id >= 0 && (uint32_t)r.ref_id < n_ref) ? ref_len[r.ref_id] : 0;
        s.r_st = r.pos;
        s.r_en = r.pos + (int64_t)red[i].t_bases; // CigarStringView::end_pos = pos + reference-consuming lengths
        s.q_nm = r.qname;
        const int64_t lead_h = (nc && opc(0) == RB_OP_H) ? len(0) : 0;
        int64_t lead_s = 0;
        if (nc && opc(0) == RB_OP_S) lead_s = len(0);
        else if (nc > 1 && opc(0) == RB_OP_H && opc(1) == RB_OP_S) lead_s = len(1);
        const int64_t trail_h = (nc && opc(nc - 1) == RB_OP_H) ? len(nc - 1) : 0;
        // read_pos(r_en - 1): position in the read of the last reference base = read bases consumed through the last
        // M/=/X op, minus one; anything else at the end of the reference span makes the reference's unwrap() panic
        int64_t q_after = 0;
        size_t k = nc;
        bool ok = false;
        for (size_t j = 0; j < nc; j++) { // rust-htslib read_pos: D / N before any op that describes read sequence is an Err
            const uint32_t o = opc(j);
            if (o == RB_OP_D || o == RB_OP_N) throw Panic("read_pos: 'deletion' found before any operation describing read sequence (" + r.qname + ")");
            if (o == RB_OP_H && j > 0 && j + 1 < nc) throw Panic("read_pos: hard clip between operations (" + r.qname + ")");
            if (o != RB_OP_H && o != RB_OP_P) break;
        }
        while (k > 0) {
            const uint32_t o = opc(k - 1);
            if (o == RB_OP_M || o == RB_OP_EQ || o == RB_OP_X) {
                ok = true;
                break;
            }
            if (o == RB_OP_D || o == RB_OP_N) break;
            if (o == RB_OP_H && !(k == nc || k == 1)) break;
            if (o == RB_OP_S || o == RB_OP_I) q_after += len(k - 1);
            k--;
        }
        if (!ok || red[i].t_bases == 0) throw Panic("called `Option::unwrap()` on a `None` value (read_pos) for " + r.qname);
        const int64_t qpos = (int64_t)red[i].q_bases - q_after - 1; // S and I count as read bases, H does not
        s.q_st = lead_h + lead_s;
        s.q_en = lead_h + 1 + qpos;
        s.q_len = lead_h + (int64_t)r.l_seq + trail_h;
        s.strand = (r.flag & 16) ? '-' : '+';
        if (r.flag & 16) { // bamstats.rs:203-207
            const int64_t t = s.q_st;
            s.q_st = s.q_len - s.q_en;
            s.q_en = s.q_len - t;
        }
        s.equal = red[i].equal, s.diff = red[i].diff, s.ins = red[i].ins, s.del = red[i].del, s.matches = red[i].matches;
        s.ins_events = red[i].ins_events, s.del_events = red[i].del_events;
        s.id_by_all = red[i].id_by_all, s.id_by_events = red[i].id_by_events, s.id_by_matches = red[i].id_by_matches;
        if (s.equal == 0 && s.matches > 0 && r.has_md) { // bamstats.rs:129-142
            uint32_t m4[4];
            parse_md_for_stats(r.md, m4);
            if (m4[0] + m4[1] != s.diff) throw Panic("assertion failed: m_count + mm_count == stats.diff");
            s.equal = m4[0];
            s.diff = m4[1];
            const volatile float e = (float)s.equal;
            const volatile float num = 100.0f * e;
            s.id_by_all = num / (float)(uint32_t)(s.equal + s.diff + s.del + s.ins);
            s.id_by_events = num / (float)(uint32_t)(s.equal + s.diff + s.del_events + s.ins_events);
            s.id_by_matches = num / (float)(uint32_t)(s.equal + s.diff);
        } else if (s.matches > 0 && !r.has_md) { // bamstats.rs:145-153 (once per such record, like the reference)
            fprintf(stderr, "\r⚠ warning: cigar string contains 'M', assuming mismatch since there is no MD tag.");
        }
    } catch (const Panic &e) { // the reference has printed the records before this one (main.rs:71-76)
        out.resize(i);
        pending = e.what();
        break;
    }
    if (!pending.empty()) {
        if (!trailing_panic) throw Panic(pending);
        *trailing_panic = pending;
    }
    return out;
}

std::string cigar_stats_header(bool qbed) {
    std::string s;
    if (qbed)
        s = "#query_name\tquery_start\tquery_end\tquery_length\tstrand\treference_name\treference_start\treference_end\treference_length\t";
    else
        s = "#reference_name\treference_start\treference_end\treference_length\tstrand\tquery_name\tquery_start\tquery_end\tquery_length\t";
    s += "perID_by_matches\tperID_by_events\tperID_by_all\tmatches\tmismatches\tdeletion_events\tinsertion_events\tdeletions\tinsertions\n";
    return s;
}
std::string cigar_stats_line(const Stats &s, bool qbed) {
    std::string o;
    auto four = [&](const std::string &nm, int64_t a, int64_t b, int64_t c) {
        o += nm; o += '\t'; o += std::to_string(a); o += '\t'; o += std::to_string(b); o += '\t'; o += std::to_string(c); o += '\t';
    };
    if (qbed) four(s.q_nm, s.q_st, s.q_en, s.q_len); else four(s.r_nm, s.r_st, s.r_en, s.r_len);
    o += s.strand; o += '\t';
    if (qbed) four(s.r_nm, s.r_st, s.r_en, s.r_len); else four(s.q_nm, s.q_st, s.q_en, s.q_len);
    o += f32_display(s.id_by_matches); o += '\t'; o += f32_display(s.id_by_events); o += '\t'; o += f32_display(s.id_by_all); o += '\t';
    o += std::to_string(s.equal); o += '\t'; o += std::to_string(s.diff); o += '\t'; o += std::to_string(s.del_events); o += '\t';
    o += std::to_string(s.ins_events); o += '\t'; o += std::to_string(s.del); o += '\t'; o += std::to_string(s.ins); o += '\n';
    return o;
}

// ---- trim-paf driver (paf.rs:210-305); the recursion is a loop -----------------------------------------
void Paf::overlapping_paf_recs(Engine &eng, int match_score, int diff_score, int indel_score, bool remove_contained) {
    // remove_trailing_indels (:218-220) runs on every record in every pass of the reference; it changes nothing on a record it
    // has already seen unless a trim rewrote that record in between, so only those ("dirty") go back to the device
    std::vector<uint32_t> dirty(records.size());
    std::iota(dirty.begin(), dirty.end(), 0u);
    double tl = now_s();
    for (int pass = 0; pass < 100000; pass++) {
        if (!dirty.empty()) {
            HostBatch b(records, dirty);
            std::vector<rb_norm_row> norm(b.n());
            eng.check(rb_host_scan_records(eng.ctx(), b.n(), b.ops.data(), b.op_off.data(), b.t_st.data(), b.t_en.data(), b.q_st.data(),
                                           b.q_en.data(), b.strand.data(), nullptr, norm.data()),
                      "rb_host_scan_records");
            for (size_t k = 0; k < dirty.size(); k++) {
                const size_t i = dirty[k];
                panic_on(norm[k].status, "remove_trailing_indels", i);
                PafRecord &r = records[i];
                if (norm[k].flags & RB_F_STRIPPED) {
                    r.id = stripped_id(r, norm[k]);
                    r.cigar.assign(r.cigar.begin() + norm[k].first_op, r.cigar.begin() + norm[k].first_op + norm[k].n_ops);
                }
                r.t_st = norm[k].t_st, r.t_en = norm[k].t_en, r.q_st = norm[k].q_st, r.q_en = norm[k].q_en;
                r.nmatch = norm[k].nmatch, r.aln_len = norm[k].aln_len;
            }
        }
        dirty.clear();
        lap("  pass: strip rewritten records", tl);
        const auto by_q = [](const PafRecord &a, const PafRecord &b) { return a.q_name < b.q_name; };
        if (!std::is_sorted(records.begin(), records.end(), by_q)) std::stable_sort(records.begin(), records.end(), by_q); // :223 (a no-op from the second pass on)
        const size_t n = records.size();
        std::vector<char> contained(n, 0);
        if (n < 2) return; // :227-229
        struct Pair { uint64_t overlap; uint32_t i, j; };
        std::vector<Pair> pairs;
        for (size_t i = 0; i + 1 < n; i++) { // :231-261
            const PafRecord &r1 = records[i];
            for (size_t j = i + 1; j < n && r1.q_name == records[j].q_name; j++) {
                const PafRecord &r2 = records[j];
                const uint64_t mn = std::min(r1.q_en, r2.q_en), mx = std::max(r1.q_st, r2.q_st);
                const uint64_t ov = mn < mx ? 0 : mn - mx;
                if (ov < 1) continue;
                if (ov == r2.q_en - r2.q_st)
                    contained[j] = 1;
                else if (ov == r1.q_en - r1.q_st)
                    contained[i] = 1;
                else if (r1.q_st <= r2.q_st)
                    pairs.push_back({ov, (uint32_t)i, (uint32_t)j});
                else
                    pairs.push_back({ov, (uint32_t)j, (uint32_t)i});
            }
        }
        std::stable_sort(pairs.begin(), pairs.end(), [](const Pair &a, const Pair &b) { return a.overlap > b.overlap; }); // :262
        std::unordered_set<std::string> q_seen;
        std::vector<uint32_t> left, right;
        size_t unseen = 0;
        for (const Pair &pr : pairs) { // :266-284: one pair per query name per pass
            if (q_seen.insert(records[pr.i].q_name).second) {
                left.push_back(pr.i);
                right.push_back(pr.j);
            } else {
                unseen++;
            }
        }
        lap("  pass: sort, pairs, one per query", tl);
        if (!left.empty()) {
            // only the records of this pass's pairs go to the device: sub-batch = left[0], right[0], left[1], right[1], ...
            std::vector<uint32_t> sub(2 * left.size()), sl(left.size()), sr(left.size());
            for (size_t k = 0; k < left.size(); k++) sub[2 * k] = left[k], sub[2 * k + 1] = right[k], sl[k] = (uint32_t)(2 * k), sr[k] = (uint32_t)(2 * k + 1);
            HostBatch b(records, sub);
            std::vector<rb_pair_row> rows(left.size());
            uint32_t *out = nullptr;
            uint64_t n_out = 0;
            eng.check(rb_host_overlap_split(eng.ctx(), b.n(), b.ops.data(), b.op_off.data(), b.t_st.data(), b.t_en.data(), b.q_st.data(),
                                            b.q_en.data(), b.strand.data(), left.size(), sl.data(), sr.data(), match_score, diff_score,
                                            indel_score, eng.bsearch_policy, rows.data(), &out, &n_out),
                      "rb_host_overlap_split");
            for (size_t k = 0; k < left.size(); k++)
                if (rows[k].status != RB_ST_OK) throw Panic("trim_overlapping_pafs: pair " + std::to_string(k) + " status " + std::to_string(rows[k].status));
            parallel_chunks(left.size(), [&](unsigned, size_t lo, size_t hi) { // (a record is in at most one pair of a pass)
                for (size_t k = lo; k < hi; k++) {
                    const uint32_t idx[2] = {left[k], right[k]};
                    for (int s = 0; s < 2; s++) {
                        PafRecord &r = records[idx[s]];
                        r.t_st = rows[k].t_st[s], r.t_en = rows[k].t_en[s], r.q_st = rows[k].q_st[s], r.q_en = rows[k].q_en[s];
                        r.nmatch = rows[k].nmatch[s], r.aln_len = rows[k].aln_len[s];
                        r.cigar.assign(out + rows[k].out_off[s], out + rows[k].out_off[s] + rows[k].out_n[s]);
                    }
                }
            });
            for (size_t k = 0; k < left.size(); k++) dirty.push_back(left[k]), dirty.push_back(right[k]);
            rb_host_free(out);
        }
        lap("  pass: split + clip on the device", tl);
        if (unseen > 0) continue; // :286-288
        if (remove_contained) {   // :289-301
            std::vector<PafRecord> keep;
            for (size_t i = 0; i < n; i++)
                if (!contained[i]) keep.push_back(std::move(records[i]));
            records.swap(keep);
        }
        return;
    }
    throw Panic("trim-paf did not converge");
}


// ---------------------------------------------------------------- nucfreq (main.rs:82-121, nucfreq.rs, bed.rs:88-131, :215-235)
Region parse_region(const std::string &region) {
    // regex (.+):([0-9]+)-([0-9]+), leftmost match with a greedy name: the LAST ":digits-digits" that has a name before it
    const size_t n = region.size();
    for (size_t c = n; c-- > 1;) {
        if (region[c] != ':') continue;
        size_t i = c + 1;
        const size_t a0 = i;
        while (i < n && region[i] >= '0' && region[i] <= '9') i++;
        if (i == a0 || i >= n || region[i] != '-') continue;
        const size_t b0 = ++i;
        while (i < n && region[i] >= '0' && region[i] <= '9') i++;
        if (i == b0) continue;
        uint64_t st = 0, en = 0;
        auto [p1, e1] = std::from_chars(region.data() + a0, region.data() + b0 - 1, st);
        if (e1 != std::errc() || st == 0) throw Panic("called `Result::unwrap()` on an `Err` value: region start " + region); // parse().unwrap() - 1
        st -= 1;
        auto [p2, e2] = std::from_chars(region.data() + b0, region.data() + i, en);
        if (e2 != std::errc()) en = 4294967295ull; // unwrap_or(2^32 - 1)
        (void)p1, (void)p2;
        if (st > en) throw Panic("Region start must be less than end.\n" + region);
        Region r;
        r.name = region.substr(0, c);
        r.st = st, r.en = en;
        r.id = r.name + ":" + std::to_string(st + 1) + "-" + std::to_string(en);
        return r;
    }
    throw Panic("Failed to parse region string.");
}

namespace {
struct BamReads { // every record of the file, in file order: the arrays of rb_reads_view plus what the host needs to pick a slice
    std::vector<std::string> ref_nm;
    std::vector<uint32_t> ref_len;
    std::vector<int32_t> tid;
    std::vector<int64_t> pos;
    std::vector<uint32_t> flag, l_seq, ops;
    std::vector<uint64_t> op_off, seq_off, end_key_pmax; // prefix maximum of tid << 32 | endpos over the reads that enter the pileup
    std::vector<uint8_t> seq;
};
uint64_t nf_key(int32_t tid, uint64_t p) { return ((uint64_t)(uint32_t)tid << 32) | std::min<uint64_t>(p, 0xFFFFFFFFull); }

// the BAM header: reference names and lengths
static void bam_read_header(BamStream &f, const std::string &path, BamReads &B) {
    uint8_t h8[8] = {0}, b4[4] = {0};
    if (!gz_exact(f, h8, 8) || memcmp(h8, "BAM\1", 4) != 0) throw Panic(path + " is not a BAM file");
    if (le32(h8 + 4) > f.n - f.pos) throw Panic("truncated BAM header in " + path); // (sizes are checked before they size anything)
    f.pos += le32(h8 + 4);                                                           // the SAM text is not used
    bam_need(f, b4, 4, path);
    const uint32_t n_ref = le32(b4);
    if ((uint64_t)n_ref * 8 > f.n - f.pos) throw Panic("truncated BAM header in " + path);
    B.ref_nm.resize(n_ref), B.ref_len.resize(n_ref);
    for (uint32_t i = 0; i < n_ref; i++) {
        bam_need(f, b4, 4, path);
        if (le32(b4) > f.n - f.pos) throw Panic("truncated BAM header in " + path);
        std::vector<char> nm(le32(b4) + 1, 0);
        bam_need(f, nm.data(), nm.size() - 1, path);
        B.ref_nm[i] = nm.data();
        bam_need(f, b4, 4, path);
        B.ref_len[i] = le32(b4);
    }
}
// the records from the cursor up to (not including) the one that starts at or behind `stop`
static void bam_read_records(BamStream &f, size_t stop, BamReads &B) {
    std::vector<uint8_t> rec;
    if (B.op_off.empty()) B.op_off.push_back(0);
    uint64_t run = B.end_key_pmax.empty() ? 0 : B.end_key_pmax.back();
    while (f.pos < stop && bam_next(f, rec)) {
        const size_t bs = rec.size();
        const BamCore core = bam_core(rec.data(), bs);
        const int32_t tid = core.tid;
        const int64_t pos = core.pos;
        uint32_t n_cig = core.n_cig;
        const uint32_t flag = core.flag, l_seq = core.l_seq;
        const uint8_t *cg = core.cg;
        const uint8_t *sq = core.sq;
        // htslib resolves the CG:B,I long-cigar convention while reading (bam_tag2cigar): <l_seq>S<ref_len>N + CG tag
        if (n_cig >= 1 && (le32(cg) & 15u) == RB_OP_S && (le32(cg) >> 4) == l_seq) {
            const size_t aux0 = core.aux0;
            for (size_t p = aux0; p + 3 <= bs;) {
                const uint8_t *tag = rec.data() + p;
                const char ty = (char)rec[p + 2];
                p += 3;
                if (ty == 'A' || ty == 'c' || ty == 'C') p += 1;
                else if (ty == 's' || ty == 'S') p += 2;
                else if (ty == 'i' || ty == 'I' || ty == 'f') p += 4;
                else if (ty == 'Z' || ty == 'H') {
                    while (p < bs && rec[p]) p++;
                    p++;
                } else if (ty == 'B') {
                    if (p + 5 > bs) break;
                    const char sub = (char)rec[p];
                    const uint32_t cnt = le32(rec.data() + p + 1);
                    const size_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4);
                    if (p + 5 + es * (uint64_t)cnt > bs) break; // (an array that runs past the record)
                    if (tag[0] == 'C' && tag[1] == 'G' && sub == 'I') {
                        cg = rec.data() + p + 5;
                        n_cig = cnt;
                        break;
                    }
                    p += 5 + es * (size_t)cnt;
                } else break;
            }
        }
        uint64_t ref = 0;
        for (uint32_t i = 0; i < n_cig; i++) {
            const uint32_t w = le32(cg + 4 * (size_t)i);
            B.ops.push_back(w);
            if ((0x18Du >> (w & 15u)) & 1u) ref += w >> 4;
        }
        B.op_off.push_back(B.ops.size());
        B.tid.push_back(tid), B.pos.push_back(pos), B.flag.push_back(flag), B.l_seq.push_back(l_seq);
        B.seq_off.push_back(B.seq.size());
        B.seq.insert(B.seq.end(), sq, sq + (l_seq + 1) / 2);
        // bam_endpos; reads the pileup never admits reach nothing
        const bool enters = tid >= 0 && !(flag & (0x4u | 0x100u | 0x200u | 0x400u));
        const uint64_t endpos = (uint64_t)std::max<int64_t>(pos, 0) + ((flag & 4u) || ref == 0 ? 1 : ref);
        if (enters) run = std::max(run, nf_key(tid, endpos));
        B.end_key_pmax.push_back(run);
    }

}
static void bam_finish(BamReads &B) {
    B.ops.resize(B.ops.size() + 4, 0);
    B.seq.resize(B.seq.size() + 32, 0);
}
void load_bam_reads(const std::string &path, BamReads &B) {
    BamStream f(path);
    bam_read_header(f, path, B);
    bam_read_records(f, ~(size_t)0, B);
    bam_finish(B);
}

// ---- .bai: bins -> chunks of BGZF virtual offsets, and the 16 kb linear index (SAM spec 5.2) ----
struct Bai {
    struct Ref {
        std::unordered_map<uint32_t, std::vector<std::pair<uint64_t, uint64_t>>> bins;
        std::vector<uint64_t> ioffset;
    };
    std::vector<Ref> refs;
    bool load(const std::string &path) {
        FILE *fp = fopen(path.c_str(), "rb");
        if (!fp) return false;
        std::vector<uint8_t> d;
        uint8_t tmp[1 << 16];
        size_t r;
        while ((r = fread(tmp, 1, sizeof tmp, fp)) > 0) d.insert(d.end(), tmp, tmp + r);
        fclose(fp);
        auto u64 = [&](size_t p) { return (uint64_t)le32(&d[p]) | ((uint64_t)le32(&d[p + 4]) << 32); };
        if (d.size() < 8 || memcmp(d.data(), "BAI\1", 4) != 0) return false;
        const uint32_t n_ref = le32(&d[4]);
        size_t p = 8;
        refs.resize(n_ref);
        for (uint32_t i = 0; i < n_ref; i++) {
            if (p + 4 > d.size()) return false;
            const uint32_t n_bin = le32(&d[p]);
            p += 4;
            for (uint32_t b = 0; b < n_bin; b++) {
                if (p + 8 > d.size()) return false;
                const uint32_t bin = le32(&d[p]), n_chunk = le32(&d[p + 4]);
                p += 8;
                if (p + 16 * (size_t)n_chunk > d.size()) return false;
                auto &v = refs[i].bins[bin];
                for (uint32_t c = 0; c < n_chunk; c++, p += 16) v.emplace_back(u64(p), u64(p + 8));
            }
            if (p + 4 > d.size()) return false;
            const uint32_t n_intv = le32(&d[p]);
            p += 4;
            if (p + 8 * (size_t)n_intv > d.size()) return false;
            refs[i].ioffset.resize(n_intv);
            for (uint32_t k = 0; k < n_intv; k++, p += 8) refs[i].ioffset[k] = u64(p);
        }
        return true;
    }
    // a range of virtual offsets that holds every record overlapping [st, en) of tid (a superset); false = no such record
    bool range(int32_t tid, uint64_t st, uint64_t en, uint64_t *vlo, uint64_t *vhi) const {
        if (tid < 0 || (size_t)tid >= refs.size() || en <= st) return false;
        const Ref &R = refs[tid];
        if (en > (1ull << 29)) en = 1ull << 29; // the binning scheme of BAI ends at 512 Mbp
        if (st >= en) return false;
        uint64_t min_off = 0;
        if (!R.ioffset.empty()) min_off = R.ioffset[std::min<size_t>((size_t)(st >> 14), R.ioffset.size() - 1)];
        uint64_t lo = ~0ull, hi = 0;
        const uint64_t e = en - 1;
        auto visit = [&](uint32_t bin) {
            auto it = R.bins.find(bin);
            if (it == R.bins.end()) return;
            for (const auto &c : it->second)
                if (c.second > min_off) lo = std::min(lo, c.first), hi = std::max(hi, c.second);
        };
        visit(0); // reg2bins (SAM spec 5.3)
        for (uint64_t k = 1 + (st >> 26); k <= 1 + (e >> 26); k++) visit((uint32_t)k);
        for (uint64_t k = 9 + (st >> 23); k <= 9 + (e >> 23); k++) visit((uint32_t)k);
        for (uint64_t k = 73 + (st >> 20); k <= 73 + (e >> 20); k++) visit((uint32_t)k);
        for (uint64_t k = 585 + (st >> 17); k <= 585 + (e >> 17); k++) visit((uint32_t)k);
        for (uint64_t k = 4681 + (st >> 14); k <= 4681 + (e >> 14); k++) visit((uint32_t)k);
        if (hi <= lo) return false;
        *vlo = lo, *vhi = hi;
        return true;
    }
};

// the member at file offset c and everything up to and including the member at file offset c_last, read and inflated
static bool bgzf_load_range(int fd, size_t file_size, size_t c, size_t c_last, BamStream &f) {
    if (c > c_last || c_last >= file_size) return false;
    uint8_t hd[18];
    if (pread(fd, hd, 18, (off_t)c_last) != 18 || hd[0] != 0x1f || hd[1] != 0x8b) return false;
    const size_t xlen = (size_t)hd[10] | ((size_t)hd[11] << 8);
    if (xlen != 6 || hd[12] != 'B' || hd[13] != 'C') return false; // (the layout every BGZF writer produces; anything else: full scan)
    const size_t end = c_last + (((size_t)hd[16] | ((size_t)hd[17] << 8)) + 1);
    if (end > file_size) return false;
    std::unique_ptr<uint8_t[]> raw(new uint8_t[end - c]);
    std::atomic<bool> ok{true};
    parallel_chunks((end - c + (1u << 22) - 1) >> 22, [&](unsigned, size_t lo, size_t hi) {
        size_t a = c + (lo << 22);
        const size_t e = std::min(end, c + (hi << 22));
        while (a < e) {
            const ssize_t r = pread(fd, raw.get() + (a - c), e - a, (off_t)a);
            if (r <= 0) {
                ok = false;
                return;
            }
            a += (size_t)r;
        }
    });
    return ok && f.inflate_members(raw.get(), end - c, c);
}

// load only what the regions need, through <bam>.bai.  false = no usable index (the caller reads the whole file)
static bool load_bam_reads_indexed(const std::string &path, const std::vector<Region> &rgns, BamReads &B) {
    if (path == "-" || getenv("RB_NO_BAI")) return false;
    Bai bai;
    if (!bai.load(path + ".bai")) return false;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct Close {
        int fd;
        ~Close() { close(fd); }
    } closer{fd};
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return false;
    const size_t file_size = (size_t)st.st_size;
    // the header: members from the start of the file, more of them until it parses
    BamReads H;
    {
        bool got = false;
        for (size_t want = 1u << 20; !got; want *= 4) {
            // last member that starts inside the first `want` bytes
            size_t off = 0, last = 0;
            uint8_t hd[18];
            while (off < std::min(want, file_size)) {
                if (pread(fd, hd, 18, (off_t)off) != 18 || hd[0] != 0x1f || hd[1] != 0x8b || hd[12] != 'B' || hd[13] != 'C') return false;
                last = off;
                off += ((size_t)hd[16] | ((size_t)hd[17] << 8)) + 1;
            }
            BamStream f;
            if (!bgzf_load_range(fd, file_size, 0, last, f)) return false;
            H = BamReads();
            try {
                bam_read_header(f, path, H);
            } catch (const Panic &) {
                return false;
            }
            if (!f.underflow) got = true;
            else if (want >= file_size) return false;
        }
    }
    if (bai.refs.size() != H.ref_nm.size()) return false;
    // virtual-offset ranges of all regions, merged where their members touch
    std::vector<std::pair<uint64_t, uint64_t>> rg;
    for (const Region &R : rgns) {
        int32_t tid = -1;
        for (size_t k = 0; k < H.ref_nm.size(); k++)
            if (H.ref_nm[k] == R.name) { tid = (int32_t)k; break; }
        if (tid < 0) continue; // (the panic for it comes when its turn to print comes)
        uint64_t lo, hi;
        if (bai.range(tid, R.st, std::min<uint64_t>(R.en, H.ref_len[tid]), &lo, &hi)) rg.emplace_back(lo, hi);
    }
    std::sort(rg.begin(), rg.end());
    B.ref_nm = H.ref_nm, B.ref_len = H.ref_len;
    for (size_t i = 0; i < rg.size();) {
        uint64_t lo = rg[i].first, hi = rg[i].second;
        size_t j = i + 1;
        while (j < rg.size() && (rg[j].first >> 16) <= (hi >> 16)) hi = std::max(hi, rg[j].second), j++; // same or overlapping members
        BamStream f;
        if (!bgzf_load_range(fd, file_size, (size_t)(lo >> 16), (size_t)(hi >> 16), f)) return false;
        const size_t a = f.upos(lo), e = f.upos(hi);
        if (a == ~(size_t)0 || e == ~(size_t)0 || a > e || e > f.n) return false;
        f.pos = a;
        bam_read_records(f, e, B);
        i = j;
    }
    if (B.op_off.empty()) B.op_off.push_back(0);
    bam_finish(B);
    return true;
}

void put_u64(std::string &o, uint64_t v) {
    char b[24];
    auto [p, e] = std::to_chars(b, b + sizeof b, v);
    (void)e;
    o.append(b, (size_t)(p - b));
}
} // namespace

void nucfreq_bam(Engine &eng, const std::string &bam_path, const std::vector<Region> &rgns, bool small,
                 const std::function<void(const std::string &)> &put) {
    double tl = now_s();
    BamReads B;
    if (!load_bam_reads_indexed(bam_path, rgns, B)) {
        B = BamReads();
        load_bam_reads(bam_path, B);
    }
    lap("nucfreq: inflate + decode BAM", tl);
    const uint64_t n_all = B.tid.size();
    for (uint64_t i = 1; i < n_all; i++) // the pileup iterator refuses unsorted input ("the input is not sorted"), p.unwrap() panics
        if (nf_key(B.tid[i], (uint64_t)B.pos[i]) < nf_key(B.tid[i - 1], (uint64_t)B.pos[i - 1]) && B.tid[i] >= 0 && B.pos[i] >= 0)
            throw Panic("nucfreq: the BAM file is not coordinate sorted");
    const uint64_t MED = 1000000; // main.rs:101: one header per piece of this size
    // the work list: every region cut into chunks of at most 32 pieces; consecutive chunks on one contig share a device call
    struct Chunk {
        size_t rg;
        int32_t tid;
        uint64_t b0, b1, c1; // positions [b0, b1) are printed, [b0, c1) computed (nothing lies past the contig's end)
    };
    std::vector<uint32_t> counts, status;
    std::string out;
    auto run_batch = [&](const std::vector<Chunk> &batch) {
        if (batch.empty()) return;
        const int32_t tid = batch[0].tid;
        std::vector<int32_t> rt(batch.size(), tid);
        std::vector<uint64_t> st(batch.size()), en(batch.size()), off(batch.size() + 1, 0);
        uint64_t lo_pos = ~0ull, hi_pos = 0;
        for (size_t k = 0; k < batch.size(); k++) {
            st[k] = batch[k].b0, en[k] = batch[k].c1;
            off[k + 1] = off[k] + (en[k] - st[k]);
            if (en[k] > st[k]) lo_pos = std::min(lo_pos, st[k]), hi_pos = std::max(hi_pos, en[k]);
        }
        counts.assign((size_t)off[batch.size()] * 4 + 4, 0);
        if (hi_pos > lo_pos) { // the slice of reads that can reach any chunk of the batch
            const uint64_t k_st = nf_key(tid, lo_pos), k_en = nf_key(tid, hi_pos);
            const uint64_t i0 = (uint64_t)(std::upper_bound(B.end_key_pmax.begin(), B.end_key_pmax.end(), k_st) - B.end_key_pmax.begin());
            uint64_t lo = i0, hi = n_all;
            while (lo < hi) {
                const uint64_t mid = (lo + hi) >> 1;
                if (nf_key(B.tid[mid], (uint64_t)B.pos[mid]) >= k_en) hi = mid; else lo = mid + 1;
            }
            const uint64_t i1 = std::max(lo, i0);
            if (i1 > i0) {
                rb_nucfreq_counters ctr{};
                rb_reads_view v{};
                v.n_reads = i1 - i0;
                v.ops = B.ops.data(), v.op_off = B.op_off.data() + i0, v.seq = B.seq.data(), v.seq_off = B.seq_off.data() + i0;
                v.l_seq = B.l_seq.data() + i0, v.tid = B.tid.data() + i0, v.pos = B.pos.data() + i0, v.flag = B.flag.data() + i0;
                status.assign((size_t)v.n_reads, 0);
                eng.check(rb_host_nucfreq(eng.ctx(), &v, batch.size(), rt.data(), st.data(), en.data(), counts.data(), status.data(), &ctr), "rb_host_nucfreq");
                for (uint64_t i = 0; i < v.n_reads; i++) {
                    if (status[i] == RB_RD_SEQ_SHORT) throw Panic("index out of bounds: a base of read " + std::to_string(i0 + i) + " lies past its sequence");
                    if (status[i] == RB_RD_BAD_CIGAR) { // htslib's cursor asserts on it -- if a fetch of one of these chunks returns the read
                        const int64_t p0 = B.pos[i0 + i];
                        for (const Chunk &c : batch)
                            if (p0 < (int64_t)c.c1 && p0 + 1 > (int64_t)c.b0) throw Panic("nucfreq: read " + std::to_string(i0 + i) + " has a cigar htslib's pileup cannot walk");
                    }
                }
            }
        }
        lap("nucfreq: device", tl);
        for (size_t k = 0; k < batch.size(); k++) {
            const Chunk &C = batch[k];
            const Region &R = rgns[C.rg];
            const uint32_t *cbase = counts.data() + (size_t)off[k] * 4;
            for (uint64_t m0 = C.b0; m0 < C.b1; m0 += MED) {
                const uint64_t m1 = std::min(m0 + MED, C.b1), me = std::min(m1, C.c1);
                out.clear();
                if (!small) out += "#chr\tstart\tend\tA\tC\tG\tT\tregion_id\n"; // nucfreq.rs:127-131, once per piece
                // the lines of the piece, formatted by all host threads (each its own run of positions), put out in order
                const size_t n_pos = me > m0 ? (size_t)(me - m0) : 0;
                const unsigned T = parallel_chunk_count(n_pos / 4096 + 1);
                std::vector<std::string> part(T);
                std::vector<uint64_t> first_pos(T, ~0ull);
                parallel_chunks(T, [&](unsigned, size_t tlo, size_t thi) {
                    for (size_t t = tlo; t < thi; t++) {
                        std::string &o = part[t];
                        for (uint64_t p = m0 + n_pos * t / T; p < m0 + n_pos * (t + 1) / T; p++) {
                            const uint32_t *c = cbase + (size_t)(p - C.b0) * 4;
                            if (!(c[0] & RB_NF_COVERED)) continue;
                            if (first_pos[t] == ~0ull) first_pos[t] = p;
                            const uint64_t a = c[0] & ~RB_NF_COVERED;
                            if (small) { // nucfreq.rs:139-153
                                uint64_t mc[4] = {a, c[1], c[2], c[3]};
                                std::sort(mc, mc + 4);
                                put_u64(o, mc[3]);
                                o += '\t';
                                put_u64(o, mc[2]);
                                o += '\n';
                            } else { // impl Display for Nucfreq, nucfreq.rs:17-33
                                o += R.name, o += '\t';
                                put_u64(o, p);
                                o += '\t';
                                put_u64(o, (uint32_t)(p + 1));
                                o += '\t';
                                put_u64(o, a);
                                o += '\t';
                                put_u64(o, c[1]);
                                o += '\t';
                                put_u64(o, c[2]);
                                o += '\t';
                                put_u64(o, c[3]);
                                o += '\t', o += R.id, o += '\n';
                            }
                        }
                    }
                });
                if (small) // the "#name pos id" line in front of the piece's first reported position
                    for (unsigned t = 0; t < T; t++)
                        if (first_pos[t] != ~0ull) {
                            out += '#', out += R.name, out += '\t';
                            put_u64(out, first_pos[t]);
                            out += '\t', out += R.id, out += '\n';
                            break;
                        }
                put(out);
                for (unsigned t = 0; t < T; t++) put(part[t]);
            }
        }
        lap("nucfreq: format + write", tl);
    };
    std::vector<Chunk> batch;
    uint64_t batch_pos = 0, batch_lo = 0, batch_hi = 0;
    for (size_t rg = 0; rg < rgns.size(); rg++) {
        const Region &R = rgns[rg];
        int32_t tid = -1;
        for (size_t k = 0; k < B.ref_nm.size(); k++)
            if (B.ref_nm[k] == R.name) { tid = (int32_t)k; break; }
        for (uint64_t b0 = R.st; b0 < R.en; b0 += MED * 32) {
            if (tid < 0) { // nucfreq.rs:121-122 (fetch fails) -- after everything before it has been printed
                run_batch(batch);
                throw Panic("Is this region (" + R.name + ":" + std::to_string(R.st + 1) + "-" + std::to_string(R.en) + ") in your reference/bam?");
            }
            const uint64_t b1 = std::min(b0 + MED * 32, R.en);
            const uint64_t c1 = std::min<uint64_t>(b1, std::max<uint64_t>(b0, std::min<uint64_t>(B.ref_len[tid], 0xFFFFFFFFull)));
            // a chunk joins the open batch while the batch stays on one contig, under 32 M positions and 4096 chunks, and its
            // positions stay within 64 Mbp of each other (the reads in between are uploaded too)
            const bool fits = !batch.empty() && batch[0].tid == tid && batch.size() < 4096 && batch_pos + (c1 - b0) <= 32000000ull &&
                              std::max(batch_hi, c1) - std::min(batch_lo, b0) <= 64000000ull;
            if (!fits) {
                run_batch(batch);
                batch.clear();
                batch_pos = 0, batch_lo = b0, batch_hi = c1;
            }
            batch.push_back({rg, tid, b0, b1, c1});
            batch_pos += c1 - b0;
            batch_lo = std::min(batch_lo, b0), batch_hi = std::max(batch_hi, c1);
        }
    }
    run_batch(batch);
}

// ---------------------------------------------------------------- header-only commands (paf.rs:91-207)
namespace {
struct TqKey {
    std::string t, q;
    bool operator==(const TqKey &o) const { return t == o.t && q == o.q; }
};
struct TqHash {
    size_t operator()(const TqKey &k) const { return std::hash<std::string>()(k.t) * 1000003u ^ std::hash<std::string>()(k.q); }
};
} // namespace

void Paf::filter_aln_pairs(uint64_t paired_len) {
    std::unordered_map<TqKey, uint64_t, TqHash> dict;
    for (const PafRecord &rec : records) dict[TqKey{rec.t_name, rec.q_name}] += rec.t_en - rec.t_st;
    records.erase(std::remove_if(records.begin(), records.end(),
                                 [&](const PafRecord &rec) { return !(paired_len < dict[TqKey{rec.t_name, rec.q_name}]); }),
                  records.end());
}
void Paf::filter_query_len(uint64_t min_query_len) {
    records.erase(std::remove_if(records.begin(), records.end(), [&](const PafRecord &rec) { return !(rec.q_len > min_query_len); }), records.end());
}
void Paf::filter_aln_len(uint64_t min_aln_len) {
    records.erase(std::remove_if(records.begin(), records.end(), [&](const PafRecord &rec) { return !(rec.t_en - rec.t_st > min_aln_len); }),
                  records.end());
}
void Paf::orient() {
    struct Acc {
        int64_t orient = 0;
        uint64_t total_bp = 0, order = 0;
    };
    std::unordered_map<TqKey, Acc, TqHash> dict;
    for (const PafRecord &rec : records) {
        Acc &a = dict[TqKey{rec.t_name, rec.q_name}];
        if (rec.strand == '-') a.orient -= (int64_t)(rec.q_en - rec.q_st);
        else a.orient += (int64_t)(rec.q_en - rec.q_st);
        const uint64_t weight = rec.t_en - rec.t_st;
        a.total_bp += weight;
        a.order += weight * (rec.t_st + rec.t_en) / 2;
    }
    for (PafRecord &rec : records) {
        const Acc &a = dict[TqKey{rec.t_name, rec.q_name}];
        if (a.total_bp == 0) throw Panic("attempt to divide by zero");
        rec.order = a.order / a.total_bp;
        if (a.orient < 0) {
            rec.q_name += "-";
            const uint64_t new_st = rec.q_len - rec.q_en, new_en = rec.q_len - rec.q_st;
            rec.q_st = new_st;
            rec.q_en = new_en;
            rec.strand = rec.strand == '+' ? '-' : '+';
        } else {
            rec.q_name += "+";
        }
    }
}
void Paf::scaffold(uint64_t spacer_size) {
    std::stable_sort(records.begin(), records.end(), [](const PafRecord &a, const PafRecord &b) {
        const int t = a.t_name.compare(b.t_name);
        if (t) return t < 0;
        if (a.order != b.order) return a.order < b.order;
        return a.q_st < b.q_st;
    });
    for (size_t g0 = 0; g0 < records.size();) {
        size_t g1 = g0;
        while (g1 < records.size() && records[g1].t_name == records[g0].t_name) g1++;
        // within one target the records are already in (order, q_st) order, so the reference's second sort (:173-177) keeps them
        std::string scaffold_name;
        {
            std::unordered_map<std::string, char> seen;
            for (size_t i = g0; i < g1; i++)
                if (seen.emplace(records[i].q_name, 1).second) {
                    if (!scaffold_name.empty()) scaffold_name += "::";
                    scaffold_name += records[i].q_name;
                }
        }
        uint64_t scaffold_len = 0;
        for (size_t q0 = g0; q0 < g1;) {
            size_t q1 = q0;
            uint64_t q_min = UINT64_MAX, q_max = 0;
            while (q1 < g1 && records[q1].q_name == records[q0].q_name) {
                q_min = std::min(q_min, records[q1].q_st);
                q_max = std::max(q_max, records[q1].q_en);
                q1++;
            }
            for (size_t i = q0; i < q1; i++) {
                records[i].q_st = records[i].q_st - q_min + scaffold_len;
                records[i].q_en = records[i].q_en - q_min + scaffold_len;
            }
            scaffold_len += (q_max - q_min) + spacer_size;
            q0 = q1;
        }
        scaffold_len -= spacer_size;
        for (size_t i = g0; i < g1; i++) {
            records[i].q_name = scaffold_name;
            records[i].q_len = scaffold_len;
        }
        g0 = g1;
    }
}

} // namespace rb
