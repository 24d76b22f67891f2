// rb_main.cpp -- `rb`-compatible front end for the hot-path subcommands (dispatcher arms main.rs:50-58 stats --paf,
// :176-182 invert, :186-214 liftover, :218-230 trim-paf, :271-281 break-paf) over the MI355X engine, plus the two
// header-only commands that sit between them in the reference's pipelines (:234-249 filter, :253-267 orient).
// Same flag names and defaults as src/cli.rs; exits with 101 where the reference panics.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include <chrono>

#include "rb_host.hpp"
#include "../synth.h"

// RB_TIMING=1: wall time of the host phases on stderr (decode / device + gather / encode / write)
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void lap(const char *what, double &t) {
    static const bool on = getenv("RB_TIMING") != nullptr;
    const double n = now_s();
    if (on) fprintf(stderr, "[rb timing] %-28s %.3f s\n", what, n - t);
    t = n;
}

static int usage() {
    fprintf(stderr,
            "usage: rb [--bsearch modern|legacy] [--device N] [--gpus N] <subcommand> ...\n"
            "  stats [-q|--qbed] [-p|--paf] <PAF or BAM>\n"
            "  liftover -b|--bed <BED> [-q|--qbed] [-l|--largest] [PAF]\n"
            "  break-paf [-m|--max-size 100] [PAF]\n"
            "  trim-paf [-m|--match-score 1] [-d|--diff-score 1] [-i|--indel-score 1] [-r|--remove-contained] [PAF]\n"
            "  invert [PAF]\n"
            "  orient [-s|--scaffold] [-i|--insert 1000000] [PAF]\n"
            "  filter [-p|--paired-len 0] [-a|--aln 0] [-q|--query 0] [PAF]\n"
            "  nucfreq [-r|--region chr:st-en] [-b|--bed <BED>] [-s|--small] <BAM>\n"
            "--gpus N: the PAF records in N shards, one worker process per GPU (liftover, break-paf, stats --paf, invert: contiguous\n"
            "          runs of lines; trim-paf: ranges of the sorted query names); the output is the single-GPU output byte for byte.\n"
            "Every other rustybam subcommand is outside this engine's scope.\n");
    return 2;
}

// a finished run leaves without tearing down gigabytes of buffers and the HIP runtime piece by piece
// (a profiler that reports at exit needs the ordinary exit path: RB_FULL_EXIT=1, or rocprofv3's library in LD_PRELOAD)
static double g_t_main = 0;
[[noreturn]] static void done(int rc) {
    fflush(stdout);
    close(1); // (whoever reads the output sees its end now, not after this process has been taken apart)
    if (getenv("RB_TIMING")) fprintf(stderr, "[rb timing] main entry to last byte written %.3f s\n", now_s() - g_t_main);
    fflush(stderr);
    const char *pre = getenv("LD_PRELOAD");
    if (getenv("RB_FULL_EXIT") || (pre && strstr(pre, "rocprofiler"))) exit(rc);
    _exit(rc);
}
// stdout's buffer lives as long as the process: it must still be there when a panic unwinds out of main's try block and the
// handler (and exit) flush what was printed before the panic (the reference prints the stats header / earlier regions first)
static char g_obuf[1 << 22];
// Gigabytes into a regular file: parallel pwrite does not scale -- a buffered write holds the file's inode lock, so 8 or 32 writers
// of one file take turns (4-5 GB/s into tmpfs whatever their number: 4 s of a 5 s run at the headline size).  The alternative
// tried here: through a shared mapping the pages are made by page faults: the file is grown to its new end (`grow`: only by the one
// process that owns the end -- a single run, or the parent of `--gpus` before its workers write), the byte range is mapped, and
// the segments are copied in by up to 32 threads.  The descriptor of a shell redirection is write-only, so the file is reopened
// read-write through /proc/self/fd; false = not possible here (the caller falls back to pwrite).
// writers of one regular file: ONE by default.  tools/shm_write_probe.c on the GPU box (16 GB into /dev/shm): 1 thread 5.85 GB/s,
// 2: 4.5, 4: 6.0, 8: 3.3, 16: 4.4, 32: 4.2 (what this used to run with), 64: 4.9 -- a buffered write holds the inode lock, more
// writers only queue for it.  RB_WRITE_THREADS overrides (a file system that does scale).
static unsigned write_threads(size_t segments) {
    static const unsigned want = getenv("RB_WRITE_THREADS") ? (unsigned)std::max(1, atoi(getenv("RB_WRITE_THREADS"))) : 1u;
    return (unsigned)std::max<size_t>(1, std::min<size_t>(segments, want));
}
struct WSeg { const char *p; size_t n; off_t at; };
static bool write_segments_mapped(const std::vector<WSeg> &segs, bool grow) {
    // MEASURED AND SHELVED (round 3, 19 GB into /dev/shm): the mapped route is 2x SLOWER than pwrite there (11.7 s against 5.3 s for the
    // whole run: shared-memory page faults serialise harder than the write path does), so it is off unless RB_MMAP_WRITE_MIN is set
    // (tests keep the code alive; a file system whose faults do scale can switch it on)
    static const bool on = getenv("RB_MMAP_WRITE_MIN") != nullptr;
    if (!on || segs.empty()) return false;
    off_t lo = segs[0].at, hi = 0;
    size_t total = 0;
    for (const WSeg &g : segs) lo = std::min(lo, g.at), hi = std::max(hi, g.at + (off_t)g.n), total += g.n;
    static const size_t min_total = getenv("RB_MMAP_WRITE_MIN") ? (size_t)strtoull(getenv("RB_MMAP_WRITE_MIN"), nullptr, 10) : (size_t)64 << 20; // (tests lower it)
    if (total < min_total) return false;
    const int fd = open("/proc/self/fd/1", O_RDWR);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return false; }
    if (st.st_size < hi) {
        if (!grow || ftruncate(fd, hi) != 0) { close(fd); return false; }
    }
    const long pg = sysconf(_SC_PAGESIZE);
    const off_t m0 = lo / pg * pg;
    char *base = (char *)mmap(nullptr, (size_t)(hi - m0), PROT_READ | PROT_WRITE, MAP_SHARED, fd, m0);
    close(fd);
    if (base == MAP_FAILED) return false;
    const unsigned T = write_threads(segs.size());
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++)
        th.emplace_back([&]() {
            for (size_t k = next++; k < segs.size(); k = next++) memcpy(base + (segs[k].at - m0), segs[k].p, segs[k].n);
        });
    for (auto &x : th) x.join();
    munmap(base, (size_t)(hi - m0));
    return true;
}
static void put(const std::string &s) { fwrite(s.data(), 1, s.size(), stdout); }
// the output of a text route: chunks in output order, gigabytes in all.  Into a regular file they go with pwrite from several
// threads, each chunk at its own offset (the page cache takes the pages in parallel); anywhere else with plain write(2) -- no
// second copy through stdio's buffer either way
static void put(const std::vector<std::string> &chunks, bool direct = false) { // direct: never through stdio's buffer (output that may be taken back)
    size_t total = 0;
    for (const std::string &s : chunks) total += s.size();
    if (total < ((size_t)16 << 20) && !direct) {
        for (const std::string &s : chunks) fwrite(s.data(), 1, s.size(), stdout);
        return;
    }
    fflush(stdout);
    struct stat st;
    const int fl = fcntl(1, F_GETFL);
    off_t at = -1;
    if (fstat(1, &st) == 0 && S_ISREG(st.st_mode) && fl >= 0 && !(fl & O_APPEND)) at = lseek(1, 0, SEEK_CUR);
    if (at >= 0) {
        // segments of at most 16 MB (write_threads() says by how many writers)
        std::vector<WSeg> segs;
        off_t o = at;
        for (const std::string &c : chunks)
            for (size_t a = 0; a < c.size(); a += (size_t)16 << 20) {
                const size_t n = std::min(c.size() - a, (size_t)16 << 20);
                segs.push_back({c.data() + a, n, o});
                o += (off_t)n;
            }
        if (write_segments_mapped(segs, true)) {
            lseek(1, o, SEEK_SET);
            return;
        }
        const unsigned T = write_threads(segs.size());
        std::vector<std::thread> th;
        std::vector<int> bad(T, 0);
        std::atomic<size_t> next{0};
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                for (size_t k = next++; k < segs.size(); k = next++) {
                    size_t done_ = 0;
                    while (done_ < segs[k].n) {
                        const ssize_t w = pwrite(1, segs[k].p + done_, segs[k].n - done_, segs[k].at + (off_t)done_);
                        if (w <= 0) { bad[t] = 1; return; }
                        done_ += (size_t)w;
                    }
                }
            });
        for (auto &x : th) x.join();
        lseek(1, o, SEEK_SET);
        for (int b : bad)
            if (b) { perror("rb: write"); _exit(1); }
        return;
    }
    for (const std::string &s : chunks) {
        size_t done_ = 0;
        while (done_ < s.size()) {
            const ssize_t w = write(1, s.data() + done_, s.size() - done_);
            if (w <= 0) { perror("rb: write"); _exit(1); }
            done_ += (size_t)w;
        }
    }
}

// `rb synth-paf` / `rb synth-bed`: the bench workload of SURVEY.md 8(d) as text (not a reference subcommand).
// Same counter-based generator as the device one (csrc/synth.h) and the same header rule as rustybam_amd/workload.py.
static int synth_paf(uint64_t seed, uint64_t first, uint64_t n_rec, bool overlap_window) {
    const uint64_t T_LEN = 248387497ull;
    // records are independent (counter-based generator): all host threads write slices of them, printed in order
    const unsigned T = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 64u));
    const uint64_t slice = 256; // records per work item
    std::vector<std::string> buf(T);
    auto put_u = [](std::string &o, uint64_t v) {
        char tmp[24];
        int k = 24;
        do { tmp[--k] = (char)('0' + v % 10); v /= 10; } while (v);
        o.append(tmp + k, (size_t)(24 - k));
    };
    for (uint64_t base = 0; base < n_rec; base += slice * T) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) {
            th.emplace_back([&, t]() {
                std::string &o = buf[t];
                o.clear();
                const uint64_t lo = base + t * slice, hi = std::min(n_rec, lo + slice);
                std::string cg;
                for (uint64_t i = lo; i < hi; i++) {
                    const uint64_t r = first + i;
                    const uint32_t n = rb_synth_n_ops_impl(seed, r, 1000, 9000);
                    uint64_t R = 0, Q = 0;
                    cg.clear();
                    for (uint32_t j = 0; j < n; j++) {
                        const uint32_t v = rb_synth_op(seed, r, j), op = v & 15u, len = v >> 4;
                        if (op != 1) R += len;
                        if (op != 2) Q += len;
                        put_u(cg, len);
                        cg.push_back("MIDNSHP=X"[op]);
                    }
                    const uint64_t h1 = rb_splitmix64(seed ^ rb_splitmix64(r ^ 0x1111111111111111ull)), h2 = rb_splitmix64(h1), h3 = rb_splitmix64(h2);
                    uint64_t t_st;
                    if (!overlap_window) {
                        t_st = h1 % (T_LEN - (R < T_LEN ? R : T_LEN) + 1);
                    } else {
                        const uint64_t w0 = 12000000ull, w1 = 13000000ull, lo_ = R > w0 ? 0 : w0 - R + 1, hi_ = w1 - 1;
                        t_st = lo_ + h1 % (hi_ - lo_ + 1);
                    }
                    const char strand = (h2 & 1ull) == 0 ? '+' : '-';
                    const uint64_t q_st = h3 % 100001ull, q_en = q_st + Q;
                    o.push_back('q'); put_u(o, r); o.push_back('\t'); put_u(o, q_en + 1000); o.push_back('\t'); put_u(o, q_st); o.push_back('\t');
                    put_u(o, q_en); o.push_back('\t'); o.push_back(strand); o.append("\tchr1\t"); put_u(o, T_LEN); o.push_back('\t'); put_u(o, t_st);
                    o.push_back('\t'); put_u(o, t_st + R); o.append("\t0\t0\t60\ttp:A:P\tcg:Z:"); o.append(cg); o.push_back('\n');
                }
            });
        }
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < T; t++) fwrite(buf[t].data(), 1, buf[t].size(), stdout);
    }
    fflush(stdout);
    return 0;
}
// `rb synth-paf config4 <n_records>`: SURVEY.md 8(d) config 4 as text -- records over the 25 contigs of .test/asm_small.bam's header in
// proportion to their lengths (the 16.5 kb chrM gets none that fit), 300..700 ops a record (mean 500: 5e9 ops at 1e7 records), queries
// q<k> of four records whose consecutive query spans overlap by U[100, 10000] bases with none contained, '+' strand, seed 0x5EED0004.
// Counter-based (a query's four records depend on (seed, k) only), all host threads, printed in order.
static int synth_paf_config4(uint64_t n_rec) {
    static const struct { const char *name; uint64_t len; } C4[25] = {
        {"chr1", 248387497}, {"chr2", 242696747}, {"chr3", 201106605}, {"chr4", 193575430}, {"chr5", 182045437}, {"chr6", 172126870},
        {"chr7", 160567423}, {"chr8", 146259322}, {"chr9", 150617274}, {"chr10", 134758122}, {"chr11", 135127772}, {"chr12", 133324781},
        {"chr13", 114240146}, {"chr14", 101219177}, {"chr15", 100338308}, {"chr16", 96330493}, {"chr17", 84277185}, {"chr18", 80542536},
        {"chr19", 61707359}, {"chr20", 66210247}, {"chr21", 45827691}, {"chr22", 51353906}, {"chrX", 154259625}, {"chrM", 16569}, {"chrY", 57227415}};
    const uint64_t seed = 0x5EED0004ull;
    uint64_t cum[26];
    cum[0] = 0;
    for (int c = 0; c < 25; c++) cum[c + 1] = cum[c] + C4[c].len;
    n_rec = n_rec / 4 * 4;
    const uint64_t n_q = n_rec / 4;
    const unsigned T = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 64u));
    const uint64_t slice = 256; // queries per work item
    std::vector<std::string> buf(T);
    auto put_u = [](std::string &o, uint64_t v) {
        char tmp[24];
        int k = 24;
        do { tmp[--k] = (char)('0' + v % 10); v /= 10; } while (v);
        o.append(tmp + k, (size_t)(24 - k));
    };
    for (uint64_t base = 0; base < n_q; base += slice * T) {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < T; t++) {
            th.emplace_back([&, t]() {
                std::string &o = buf[t];
                o.clear();
                const uint64_t lo = base + t * slice, hi = std::min(n_q, lo + slice);
                std::string cg[4];
                for (uint64_t k = lo; k < hi; k++) {
                    uint64_t R[4], Q[4], q_st[4];
                    for (int j = 0; j < 4; j++) {
                        const uint64_t r = 4 * k + j;
                        const uint32_t n = rb_synth_n_ops_impl(seed, r, 300, 700);
                        R[j] = Q[j] = 0;
                        cg[j].clear();
                        for (uint32_t i = 0; i < n; i++) {
                            const uint32_t v = rb_synth_op(seed, r, i), op = v & 15u, len = v >> 4;
                            if (op != 1) R[j] += len;
                            if (op != 2) Q[j] += len;
                            put_u(cg[j], len);
                            cg[j].push_back("MIDNSHP=X"[op]);
                        }
                    }
                    q_st[0] = 0;
                    for (int j = 1; j < 4; j++) { // the next span starts inside the previous one, U[100, 10000] bases before its end; never contained
                        const uint64_t h = rb_splitmix64(seed ^ rb_splitmix64((4 * k + j) ^ 0x2222222222222222ull));
                        const uint64_t ov = std::min<uint64_t>(100 + h % 9901, std::min(Q[j - 1], Q[j]) / 2);
                        q_st[j] = q_st[j - 1] + Q[j - 1] - ov;
                    }
                    const uint64_t q_len = q_st[3] + Q[3] + 1000;
                    for (int j = 0; j < 4; j++) {
                        const uint64_t r = 4 * k + j;
                        const uint64_t h1 = rb_splitmix64(seed ^ rb_splitmix64(r ^ 0x1111111111111111ull)), h2 = rb_splitmix64(h1);
                        int c = 0;
                        for (int tries = 0; tries < 8; tries++) { // a contig in proportion to its length, long enough for the record
                            const uint64_t x = rb_splitmix64(h1 + (uint64_t)tries) % cum[25];
                            c = 0;
                            while (cum[c + 1] <= x) c++;
                            if (C4[c].len > R[j]) break;
                            c = 0; // (chr1 takes whatever fits nowhere else)
                        }
                        const uint64_t t_st = h2 % (C4[c].len - R[j] + 1);
                        o.push_back('q'); put_u(o, k); o.push_back('\t'); put_u(o, q_len); o.push_back('\t'); put_u(o, q_st[j]); o.push_back('\t');
                        put_u(o, q_st[j] + Q[j]); o.append("\t+\t"); o.append(C4[c].name); o.push_back('\t'); put_u(o, C4[c].len); o.push_back('\t');
                        put_u(o, t_st); o.push_back('\t'); put_u(o, t_st + R[j]); o.append("\t0\t0\t60\ttp:A:P\tcg:Z:"); o.append(cg[j]); o.push_back('\n');
                    }
                }
            });
        }
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < T; t++) fwrite(buf[t].data(), 1, buf[t].size(), stdout);
    }
    fflush(stdout);
    return 0;
}
static int synth_bed(uint64_t n_win) {
    for (uint64_t i = 0; i < n_win; i++) {
        const uint64_t st = i * 82796ull, en = st + 100000ull < 248387497ull ? st + 100000ull : 248387497ull;
        printf("chr1\t%llu\t%llu\n", (unsigned long long)st, (unsigned long long)en);
    }
    return 0;
}

// ---- `rb --gpus N` -----------------------------------------------------------------------------------------------
// PAF records are independent (liftover.rs:123-129 hands them to rayon), so the input is cut into N runs of whole lines of about
// equal bytes (CIGAR text is what weighs) -- for trim-paf, whose unit is the query-name group (paf.rs:223, :235), into N ranges of
// the sorted query names -- and N worker processes are forked BEFORE anything touches the GPU; worker k takes device `device + k`
// and runs the ordinary single-GPU command over its lines.  No data-path collective: the workers' outputs are put together on the
// host.  The parent never initialises HIP and never holds output text: a worker keeps its output in memory (as the single run
// does), tells the parent what PIECES it consists of (a small index over a socket), and is told in which order -- and, when stdout
// is a regular file, at which file offsets -- to write them:
//   concat   break-paf, stats --paf, invert, trim-paf: one piece per worker, shard order = record order (= sorted-name order).
//   contig   liftover: the reference emits contig-major, contigs by first appearance over the WHOLE file (liftover.rs:151-164), and
//            within a contig in record order.  A worker's pieces are its per-contig runs; it also lists every contig of its records
//            in local first-appearance order (also those without output: they still take their rank).  Global rank = first
//            appearance over (shard, local rank); pieces go out by (rank, shard).  `A B | A B` gives `A A B B`, as one GPU does.
//   largest  liftover --largest (main.rs:200-208): stable sort by id, keep the LAST record of maximal target span per id.  Workers
//            reduce per (id, contig) and send one candidate line each; the parent keeps, per id, the greatest span, among equals the
//            last in canonical order (contig rank, shard, local order), and has the winners written in id order.
// Into a regular file the workers pwrite side by side (no byte passes through the parent); otherwise each worker writes its pieces
// in the order it was given into its own pipe and the parent copies exactly the announced byte counts to stdout in global order.
// A worker that fails (a reference panic: exit code 101) does so before it sends its index, as the single run panics before it
// prints: the parent then has nothing written and returns the code of the first shard that failed.
namespace {
enum GatherMode { GATHER_CONCAT, GATHER_CONTIG, GATHER_LARGEST };
struct Piece {
    uint32_t contig = 0; // index into Output::contigs
    std::string key;     // largest: the record's id
    uint64_t aux = 0;    // largest: target span
    uint64_t bytes = 0;
};
struct Output {
    std::vector<std::string> chunks;  // the text, in local output order
    std::vector<std::string> contigs; // liftover: the records' contigs, local first-appearance order
    std::vector<Piece> pieces;        // consecutive byte ranges of the chunks' concatenation (none = one piece holding everything)
};
struct Worker {
    bool on = false, to_file = false;
    int ctl = -1;
} g_worker;

bool write_all(int fd, const void *p, size_t n) {
    const char *c = (const char *)p;
    while (n) {
        const ssize_t w = write(fd, c, n);
        if (w <= 0) return false;
        c += w, n -= (size_t)w;
    }
    return true;
}
bool read_all_fd(int fd, void *p, size_t n) {
    char *c = (char *)p;
    while (n) {
        const ssize_t r = read(fd, c, n);
        if (r <= 0) return false;
        c += r, n -= (size_t)r;
    }
    return true;
}
void msg_u64(std::string &m, uint64_t v) { m.append((const char *)&v, 8); }
void msg_str(std::string &m, const std::string &v) { msg_u64(m, v.size()), m.append(v); }
struct MsgIn {
    std::string buf;
    size_t at = 0;
    bool ok = true;
    uint64_t u64() {
        uint64_t v = 0;
        if (at + 8 > buf.size()) { ok = false; return 0; }
        memcpy(&v, buf.data() + at, 8), at += 8;
        return v;
    }
    std::string str() {
        const uint64_t n = u64();
        if (!ok || at + n > buf.size()) { ok = false; return std::string(); }
        std::string v = buf.substr(at, (size_t)n);
        at += (size_t)n;
        return v;
    }
};
bool send_msg(int fd, const std::string &m) {
    const uint64_t n = m.size();
    return write_all(fd, &n, 8) && write_all(fd, m.data(), m.size());
}
bool recv_msg(int fd, MsgIn &m) {
    uint64_t n = 0;
    if (!read_all_fd(fd, &n, 8) || n > ((uint64_t)1 << 40)) return false;
    m.buf.resize((size_t)n), m.at = 0, m.ok = true;
    return read_all_fd(fd, &m.buf[0], (size_t)n);
}

// worker side: announce the pieces, learn order / offsets, write, leave
[[noreturn]] void worker_emit(Output &o) {
    uint64_t total = 0;
    for (const std::string &c : o.chunks) total += c.size();
    if (o.pieces.empty()) {
        Piece p;
        p.bytes = total;
        o.pieces.push_back(p);
        if (o.contigs.empty()) o.contigs.push_back(std::string());
    }
    std::string m;
    msg_u64(m, o.contigs.size());
    for (const std::string &c : o.contigs) msg_str(m, c);
    msg_u64(m, o.pieces.size());
    for (const Piece &p : o.pieces) msg_u64(m, p.contig), msg_str(m, p.key), msg_u64(m, p.aux), msg_u64(m, p.bytes);
    MsgIn plan;
    if (!send_msg(g_worker.ctl, m) || !recv_msg(g_worker.ctl, plan)) _exit(1); // (the parent gave up: another shard failed)
    std::vector<uint64_t> start(o.pieces.size() + 1, 0), cstart(o.chunks.size() + 1, 0);
    for (size_t k = 0; k < o.pieces.size(); k++) start[k + 1] = start[k] + o.pieces[k].bytes;
    for (size_t k = 0; k < o.chunks.size(); k++) cstart[k + 1] = cstart[k] + o.chunks[k].size();
    struct Seg { const char *p; size_t n; int64_t at; };
    std::vector<Seg> segs;
    bool mapped = false;
    const uint64_t n_writes = plan.u64();
    for (uint64_t w = 0; w < n_writes && plan.ok; w++) {
        const uint64_t k = plan.u64();
        int64_t at = (int64_t)plan.u64();
        if (!plan.ok || k >= o.pieces.size()) _exit(1);
        uint64_t a = start[k];
        const uint64_t e = start[k + 1];
        size_t c = (size_t)(std::upper_bound(cstart.begin(), cstart.end(), a) - cstart.begin()) - 1;
        while (a < e) { // the chunks this piece runs through, in parts of at most 16 MB
            while (c < o.chunks.size() && cstart[c + 1] <= a) c++;
            const uint64_t n = std::min<uint64_t>(std::min(e, cstart[c + 1]) - a, (uint64_t)16 << 20);
            segs.push_back({o.chunks[c].data() + (a - cstart[c]), (size_t)n, at});
            a += n;
            if (at >= 0) at += (int64_t)n;
        }
    }
    bool ok = true;
    if (g_worker.to_file) { // (the parent has grown the file to its final size: every worker can map its ranges)
        std::vector<WSeg> ws;
        for (const Seg &sg : segs) ws.push_back({sg.p, sg.n, (off_t)sg.at});
        mapped = write_segments_mapped(ws, false);
    }
    if (g_worker.to_file && !mapped) {
        const unsigned T = write_threads(segs.size()); // (the page cache scales with the writers)
        std::vector<std::thread> th;
        std::vector<int> bad(T, 0);
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                for (size_t k = t; k < segs.size(); k += T) {
                    size_t d = 0;
                    while (d < segs[k].n) {
                        const ssize_t w = pwrite(1, segs[k].p + d, segs[k].n - d, (off_t)(segs[k].at + (int64_t)d));
                        if (w <= 0) { bad[t] = 1; return; }
                        d += (size_t)w;
                    }
                }
            });
        for (auto &x : th) x.join();
        for (int b : bad) ok = ok && !b;
    } else if (!g_worker.to_file) {
        for (const Seg &sg : segs) ok = ok && write_all(1, sg.p, sg.n);
    }
    if (!ok) perror("rb: write");
    close(1);
    if (getenv("RB_TIMING")) fprintf(stderr, "[rb timing] worker: main entry to last byte written %.3f s\n", now_s() - g_t_main);
    fflush(stderr);
    const char *pre = getenv("LD_PRELOAD");
    if (getenv("RB_FULL_EXIT") || (pre && strstr(pre, "rocprofiler"))) exit(ok ? 0 : 1);
    _exit(ok ? 0 : 1);
}
} // namespace

// what a finished arm does with its text: the single run prints it; a `--gpus` worker hands it to the gather (and does not return)
static void emit(Output &o) {
    if (g_worker.on) worker_emit(o);
    bool in_order = true; // pieces by contig in first-appearance order? (the chunks of the pipelined route need not be)
    for (size_t k = 1; k < o.pieces.size(); k++) in_order = in_order && o.pieces[k].contig >= o.pieces[k - 1].contig;
    if (in_order) {
        put(o.chunks);
        return;
    }
    // contig-major (liftover.rs:151-164): the pieces of contig 0 in the order they came, then contig 1, ...
    std::vector<uint64_t> start(o.pieces.size() + 1, 0), cstart(o.chunks.size() + 1, 0);
    for (size_t k = 0; k < o.pieces.size(); k++) start[k + 1] = start[k] + o.pieces[k].bytes;
    for (size_t k = 0; k < o.chunks.size(); k++) cstart[k + 1] = cstart[k] + o.chunks[k].size();
    std::vector<size_t> order(o.pieces.size());
    for (size_t k = 0; k < order.size(); k++) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return o.pieces[a].contig < o.pieces[b].contig; });
    fflush(stdout);
    for (size_t k : order) {
        uint64_t a = start[k];
        const uint64_t e = start[k + 1];
        size_t c = (size_t)(std::upper_bound(cstart.begin(), cstart.end(), a) - cstart.begin()) - 1;
        while (a < e) {
            while (c < o.chunks.size() && cstart[c + 1] <= a) c++;
            const uint64_t n = std::min(e, cstart[c + 1]) - a;
            if (!write_all(1, o.chunks[c].data() + (a - cstart[c]), (size_t)n)) { perror("rb: write"); _exit(1); }
            a += n;
        }
    }
}
static void emit(std::vector<std::string> &chunks) {
    Output o;
    o.chunks.swap(chunks);
    emit(o);
}

// parent side.  Returns -1 in a worker (which carries on as an ordinary single-GPU run over its share of the input), the exit code
// in the parent.  `header`: printed by the parent before anything else (stats --paf).
static int shard_fork(int n, GatherMode mode, bool by_query, std::string &path, int &device, const std::string &header) {
    int fd = -1;
    bool plain = false;
    if (path != "-") {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) {
            fprintf(stderr, "thread 'main' panicked: Failed to open %s\n", path.c_str());
            return 101;
        }
        unsigned char magic[2] = {0, 0};
        struct stat st;
        plain = fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && !(pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b);
    }
    if (!plain) { // stdin / gzip: the text into an anonymous file every worker inherits
        if (fd >= 0) close(fd);
        fd = memfd_create("rb_input", 0);
        bool ok = fd >= 0;
        try {
            const std::string all = rb::read_input_text(path);
            ok = ok && write_all(fd, all.data(), all.size());
        } catch (const rb::Panic &e) {
            fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
            return 101;
        } // (the text itself is gone here: the anonymous file is the only copy)
        if (!ok) {
            fprintf(stderr, "rb: cannot buffer the input for --gpus\n");
            return 1;
        }
        path = "/proc/self/fd/" + std::to_string(fd);
    }
    struct stat st;
    fstat(fd, &st);
    const uint64_t size = (uint64_t)st.st_size;
    std::vector<uint64_t> cut(n + 1, size);
    std::vector<std::string> qcut;
    if (by_query) {
        try {
            qcut = rb::query_name_cuts(path, n);
        } catch (const rb::Panic &e) {
            fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
            return 101;
        }
        n = (int)qcut.size() + 1; // (fewer names than GPUs: fewer workers)
    } else {
        cut[0] = 0;
        std::vector<char> buf(1 << 16);
        for (int k = 1; k < n; k++) { // the first line start at or behind size * k / n
            uint64_t at = std::max(cut[k - 1], size / (uint64_t)n * (uint64_t)k);
            bool found = at == 0;
            if (!found && at < size) { // (a cut must follow a newline: look from one byte earlier)
                at -= 1;
                while (at < size && !found) {
                    const ssize_t r = pread(fd, buf.data(), buf.size(), (off_t)at);
                    if (r <= 0) break;
                    const void *nl = memchr(buf.data(), '\n', (size_t)r);
                    if (nl) at += (uint64_t)((const char *)nl - buf.data()) + 1, found = true;
                    else at += (uint64_t)r;
                }
            }
            cut[k] = found ? std::min(at, size) : size;
        }
    }
    fflush(stdout);
    if (!header.empty() && !write_all(1, header.data(), header.size())) return 1;
    struct stat so;
    const int fl = fcntl(1, F_GETFL);
    off_t base = -1;
    if (fstat(1, &so) == 0 && S_ISREG(so.st_mode) && fl >= 0 && !(fl & O_APPEND)) base = lseek(1, 0, SEEK_CUR);
    const bool to_file = base >= 0;
    const bool same = getenv("RB_GPUS_SAME_DEVICE") != nullptr; // (diagnostic: every worker on --device, to try the gather on one GPU)
    std::vector<int> rd(n, -1), ctl(n, -1);
    std::vector<pid_t> pid(n, -1);
    for (int k = 0; k < n; k++) {
        int pp[2] = {-1, -1}, sp[2];
        if (socketpair(AF_UNIX, SOCK_STREAM, 0, sp) != 0) return 1;
        if (!to_file && pipe(pp) != 0) return 1;
        const pid_t c = fork();
        if (c < 0) return 1;
        if (c == 0) {
            for (int j = 0; j < k; j++) {
                if (rd[j] >= 0) close(rd[j]);
                close(ctl[j]);
            }
            close(sp[0]);
            if (!to_file) {
                close(pp[0]);
                dup2(pp[1], 1);
                close(pp[1]);
            }
            g_worker.on = true, g_worker.to_file = to_file, g_worker.ctl = sp[1];
            if (!same) device += k;
            if (by_query) rb::set_input_query_range(k > 0 ? &qcut[k - 1] : nullptr, k + 1 < n ? &qcut[k] : nullptr);
            else rb::set_input_slice(cut[k], cut[k + 1]);
            return -1;
        }
        close(sp[1]);
        if (!to_file) close(pp[1]), rd[k] = pp[0];
        ctl[k] = sp[0], pid[k] = c;
    }
    // the workers' indexes
    struct Item { uint64_t rank; int shard; uint64_t idx, bytes; };
    std::vector<std::vector<std::string>> contigs(n);
    std::vector<std::vector<Piece>> pieces(n);
    bool all_ok = true;
    std::vector<char> failed(n, 0);
    for (int k = 0; k < n; k++) {
        MsgIn m;
        bool ok = recv_msg(ctl[k], m);
        if (ok) {
            const uint64_t nc = m.u64();
            for (uint64_t i = 0; i < nc && m.ok; i++) contigs[k].push_back(m.str());
            const uint64_t np = m.u64();
            for (uint64_t i = 0; i < np && m.ok; i++) {
                Piece p;
                p.contig = (uint32_t)m.u64(), p.key = m.str(), p.aux = m.u64(), p.bytes = m.u64();
                if (p.contig >= contigs[k].size()) m.ok = false;
                pieces[k].push_back(std::move(p));
            }
            ok = m.ok;
        }
        all_ok = all_ok && ok;
        failed[k] = !ok;
    }
    auto reap = [&]() { // the exit code of the first shard that failed by itself (the others only stop because the parent gave up)
        int rc = 0, rc_own = 0;
        for (int k = 0; k < n; k++) {
            int st_k = 0;
            waitpid(pid[k], &st_k, 0);
            const int rc_k = WIFEXITED(st_k) ? WEXITSTATUS(st_k) : 1;
            if (rc == 0) rc = rc_k;
            if (rc_own == 0 && failed[k]) rc_own = rc_k ? rc_k : 1;
        }
        return rc_own ? rc_own : rc;
    };
    if (!all_ok) { // a shard failed (it has said why on stderr): nobody writes
        for (int k = 0; k < n; k++) close(ctl[k]);
        const int rc = reap();
        return rc ? rc : 1;
    }
    // the global order
    std::unordered_map<std::string, uint64_t> rank;
    for (int k = 0; k < n; k++)
        for (const std::string &c : contigs[k]) rank.emplace(c, (uint64_t)rank.size());
    std::vector<Item> plan;
    if (mode == GATHER_LARGEST) {
        struct Best { uint64_t aux; Item it; };
        std::map<std::string, Best> best; // bytewise order of the ids = Rust's String order
        for (int k = 0; k < n; k++)
            for (uint64_t i = 0; i < pieces[k].size(); i++) {
                const Piece &p = pieces[k][i];
                const Item it{rank[contigs[k][p.contig]], k, i, p.bytes};
                auto f = best.find(p.key);
                if (f == best.end()) { best.emplace(p.key, Best{p.aux, it}); continue; }
                Best &b = f->second;
                const bool later = std::make_tuple(it.rank, it.shard, it.idx) > std::make_tuple(b.it.rank, b.it.shard, b.it.idx);
                if (p.aux > b.aux || (p.aux == b.aux && later)) b = Best{p.aux, it};
            }
        for (auto &kv : best) plan.push_back(kv.second.it);
    } else {
        for (int k = 0; k < n; k++)
            for (uint64_t i = 0; i < pieces[k].size(); i++)
                plan.push_back({mode == GATHER_CONTIG ? rank[contigs[k][pieces[k][i].contig]] : 0, k, i, pieces[k][i].bytes});
        if (mode == GATHER_CONTIG)
            std::stable_sort(plan.begin(), plan.end(), [](const Item &a, const Item &b) { return a.rank < b.rank; }); // (shard, idx) order kept inside a rank
    }
    std::vector<std::string> msg(n);
    std::vector<uint64_t> count(n, 0);
    uint64_t at = to_file ? (uint64_t)base : 0;
    for (const Item &it : plan) count[it.shard]++;
    for (int k = 0; k < n; k++) msg_u64(msg[k], count[k]);
    for (const Item &it : plan) {
        msg_u64(msg[it.shard], it.idx);
        msg_u64(msg[it.shard], to_file ? at : ~(uint64_t)0);
        at += it.bytes;
    }
    bool ok = true;
    if (to_file) { // the file at its final size before anybody writes: the workers map their ranges of it (write_segments_mapped)
        struct stat sn;
        if (fstat(1, &sn) == 0 && (uint64_t)sn.st_size < at) (void)!ftruncate(1, (off_t)at);
    }
    for (int k = 0; k < n; k++) ok = send_msg(ctl[k], msg[k]) && ok;
    if (!to_file && ok) { // exactly the announced bytes of each piece, in global order
        std::vector<char> b((size_t)4 << 20);
        for (const Item &it : plan) {
            uint64_t left = it.bytes;
            while (left && ok) {
                const ssize_t r = read(rd[it.shard], b.data(), (size_t)std::min<uint64_t>(left, b.size()));
                if (r <= 0) { ok = false; break; }
                ok = write_all(1, b.data(), (size_t)r);
                left -= (uint64_t)r;
            }
        }
    }
    for (int k = 0; k < n; k++) {
        close(ctl[k]);
        if (rd[k] >= 0) close(rd[k]);
    }
    const int rc = reap();
    if (to_file) lseek(1, (off_t)at, SEEK_SET);
    return rc ? rc : (ok ? 0 : 1);
}

// liftover / break-paf on a big plain file: the pipelined text route (rb_host: chunks of the file on a few host threads, each with a
// context of its own).  A single run writing into a regular file streams the chunks' outputs as they arrive -- as long as that is
// the reference's order: contig ranks (first appearance over the records seen so far) must not decrease along the output.  When they
// do (a file that is not sorted by target: the rows of a contig seen earlier must go in FRONT of what has been written), or when a
// line needs the general parser, the file is cut back to where the output began and the caller takes the whole-file route.  Into
// a pipe, and in a `--gpus` worker, the outputs are kept and put in order at the end (emit).  true = the output is complete.
static bool try_pipelined(int device, int policy, bool is_break, uint32_t max_size, const std::string &paf_path, const std::vector<rb::Region> &rgns) {
    if (getenv("RB_NO_PIPELINE")) return false;
    struct stat so;
    const int fl = fcntl(1, F_GETFL);
    fflush(stdout);
    off_t base = -1;
    if (!g_worker.on && fstat(1, &so) == 0 && S_ISREG(so.st_mode) && fl >= 0 && !(fl & O_APPEND)) base = lseek(1, 0, SEEK_CUR);
    const bool streaming = base >= 0;
    Output acc;
    std::unordered_map<std::string, uint32_t> rank; // contig -> first-appearance rank over the chunks so far
    int64_t last_rank = -1;
    bool violated = false;
    auto sink = [&](std::vector<std::string> &text, rb::TextRuns &runs) -> bool {
        std::vector<uint32_t> gid(runs.contigs.size());
        for (size_t c = 0; c < runs.contigs.size(); c++) {
            auto it = rank.emplace(runs.contigs[c], (uint32_t)rank.size());
            if (it.second) acc.contigs.push_back(runs.contigs[c]);
            gid[c] = it.first->second;
        }
        if (violated) return false;
        if (streaming) {
            for (const auto &r : runs.runs) {
                if ((int64_t)gid[r.first] < last_rank) { violated = true; return false; }
                last_rank = gid[r.first];
            }
            put(text, true); // (straight to the file: a rewind must not find earlier chunks waiting in stdio's buffer)
            std::vector<std::string>().swap(text);
            return true;
        }
        for (const auto &r : runs.runs) {
            Piece pc;
            pc.contig = gid[r.first], pc.bytes = r.second;
            acc.pieces.push_back(pc);
        }
        for (std::string &t : text) acc.chunks.push_back(std::move(t));
        return true;
    };
    auto rewind = [&]() { // what was written is not the reference's output: back to where it began
        if (streaming && rb::pipeline_started())
            if (ftruncate(1, base) != 0 || lseek(1, base, SEEK_SET) != base) { perror("rb: cannot rewind the output"); _exit(1); }
    };
    // MEASURED AND SHELVED (round 3): the pages of the output file made ahead of the writes (fallocate beyond the end of the file while
    // the first chunk is still being computed: making the pages is a third of the cost of a write into a page cache -- fallocate 17 GB/s,
    // pwrite into pages that exist 9 GB/s, pwrite that has to make them 5.9 GB/s, tools/shm_write_probe.c).  At the headline size the
    // run got SLOWER, 5.5 s against 4.7 s: fallocate holds the inode lock the first writes then wait for.  Only with RB_PREALLOC=1.
    off_t pre_len = 0;
    std::thread pre;
    struct stat si;
    if (streaming && getenv("RB_PREALLOC") && stat(paf_path.c_str(), &si) == 0 && S_ISREG(si.st_mode) && si.st_size >= ((off_t)1 << 30)) {
        pre_len = (off_t)((double)si.st_size * 1.35);
        pre = std::thread([base, pre_len]() { (void)fallocate(1, FALLOC_FL_KEEP_SIZE, base, pre_len); });
    }
    auto give_back = [&]() { // (after the last write: the pages behind the end of the output)
        if (pre.joinable()) pre.join();
        if (pre_len <= 0) return;
        const off_t end = lseek(1, 0, SEEK_CUR);
        if (end >= 0 && end < base + pre_len) (void)fallocate(1, FALLOC_FL_PUNCH_HOLE | FALLOC_FL_KEEP_SIZE, end, base + pre_len - end);
    };
    bool ok = false;
    try {
        ok = rb::lift_file_text_pipelined(device, policy, is_break, max_size, paf_path, rgns, sink);
    } catch (...) { // (a reference panic in a later chunk: the single run panics before it prints anything)
        rewind();
        give_back();
        throw;
    }
    give_back();
    if (ok && !violated) {
        if (!streaming) {
            if (acc.contigs.empty()) acc.contigs.push_back(std::string());
            if (acc.pieces.empty()) acc.pieces.push_back(Piece());
            emit(acc);
        }
        return true;
    }
    rewind();
    return false;
}

// `rb [--gpus N] regroup [-q] [-l] <PAF>`: not a reference subcommand and no device work -- the lines of a PAF file put in the ORDER
// the hot-path commands emit (so that the `--gpus` gather can be checked on a machine without a GPU, tests/test_rb_gather_cpu.py):
// default = liftover's canonical order with one output line per record (contig-major by first appearance of column 6, then file
// order); -q = trim-paf's order (stable sort by column 1); -l = liftover --largest's reduction with id := column 1, span := col 9 - col 8.
static int regroup(const std::string &path, bool by_query, bool largest_) {
    const std::string all = rb::read_input_text(path);
    struct Ln { std::string_view text, q, t; uint64_t span; uint32_t contig; };
    std::vector<Ln> lines;
    Output o;
    std::unordered_map<std::string, uint32_t> cid;
    for (size_t a = 0; a < all.size();) {
        size_t e = all.find('\n', a);
        if (e == std::string::npos) e = all.size();
        std::string_view ln(all.data() + a, e - a);
        std::vector<std::string_view> col;
        for (size_t i = 0; i < ln.size() && col.size() < 9;) {
            size_t j = ln.find('\t', i);
            if (j == std::string_view::npos) j = ln.size();
            col.push_back(ln.substr(i, j - i));
            i = j + 1;
        }
        if (col.size() >= 9) {
            Ln l{ln, col[0], col[5], strtoull(std::string(col[8]).c_str(), nullptr, 10) - strtoull(std::string(col[7]).c_str(), nullptr, 10), 0};
            auto it = cid.emplace(std::string(l.t), (uint32_t)o.contigs.size());
            if (it.second) o.contigs.push_back(std::string(l.t));
            l.contig = it.first->second;
            lines.push_back(l);
        }
        a = e + 1;
    }
    o.chunks.emplace_back();
    auto add = [&](const Ln &l, const std::string &key, bool new_piece) {
        if (new_piece || o.pieces.empty()) {
            Piece p;
            p.contig = l.contig, p.key = key, p.aux = l.span;
            o.pieces.push_back(p);
        }
        o.chunks[0].append(l.text), o.chunks[0].push_back('\n');
        o.pieces.back().bytes += l.text.size() + 1;
    };
    if (by_query) {
        std::stable_sort(lines.begin(), lines.end(), [](const Ln &x, const Ln &y) { return x.q < y.q; });
        for (const Ln &l : lines) o.chunks[0].append(l.text), o.chunks[0].push_back('\n');
        o.contigs.clear(); // (one piece: worker_emit fills it in)
    } else {
        std::stable_sort(lines.begin(), lines.end(), [](const Ln &x, const Ln &y) { return x.contig < y.contig; });
        if (largest_) {
            std::stable_sort(lines.begin(), lines.end(), [](const Ln &x, const Ln &y) { return x.q < y.q; });
            for (size_t i = 0; i < lines.size();) {
                size_t j = i, best = i;
                for (; j < lines.size() && lines[j].q == lines[i].q && (!g_worker.on || lines[j].contig == lines[i].contig); j++)
                    if (lines[j].span >= lines[best].span) best = j;
                add(lines[best], std::string(lines[best].q), true);
                i = j;
            }
        } else {
            for (size_t i = 0; i < lines.size(); i++) add(lines[i], std::string(), i == 0 || lines[i].contig != lines[i - 1].contig);
        }
        if (o.contigs.empty()) o.contigs.push_back(std::string());
        if (o.pieces.empty()) o.pieces.push_back(Piece());
    }
    if (!g_worker.on) o.pieces.clear();
    emit(o);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv) {
    g_t_main = now_s();
    setvbuf(stdout, g_obuf, _IOFBF, sizeof g_obuf);
    // a one-shot command runs every kernel once over buffers it allocates and frees: plain hipMalloc.  (The library builds buffers of
    // 1 GB and more from 2 MB physical chunks -- 10-15 % on the streaming kernels of a RESIDENT batch, about 14 us per chunk to make:
    // a second per 75 GB, which this front end would pay for 50 ms of kernels.)
    setenv("RB_ALLOC_MODE", "default", 0);
    int a = 1, device = 0, policy = RB_BSEARCH_MODERN, gpus = 1;
    while (a + 1 < argc && argv[a][0] == '-') {
        if (!strcmp(argv[a], "--bsearch")) policy = !strcmp(argv[a + 1], "legacy") ? RB_BSEARCH_LEGACY : RB_BSEARCH_MODERN;
        else if (!strcmp(argv[a], "--device")) device = atoi(argv[a + 1]);
        else if (!strcmp(argv[a], "--gpus")) gpus = atoi(argv[a + 1]);
        else if (!strcmp(argv[a], "-t") || !strcmp(argv[a], "--threads")) { /* accepted, unused */ }
        else break;
        a += 2;
    }
    if (a >= argc) return usage();
    const std::string cmd = argv[a++];
    if (cmd == "synth-paf" || cmd == "synth-bed") { // rb synth-paf <seed> <first_record> <n_records> [overlap] | rb synth-bed <n_windows>
        if (cmd == "synth-bed") return synth_bed(a < argc ? strtoull(argv[a], nullptr, 0) : 3000);
        if (a + 1 < argc && !strcmp(argv[a], "config4")) return synth_paf_config4(strtoull(argv[a + 1], nullptr, 0)); // rb synth-paf config4 <n_records>
        if (a + 2 >= argc) return usage();
        return synth_paf(strtoull(argv[a], nullptr, 0), strtoull(argv[a + 1], nullptr, 0), strtoull(argv[a + 2], nullptr, 0),
                         a + 3 < argc && !strcmp(argv[a + 3], "overlap"));
    }
    std::string paf_path = "-", bed_path;
    bool qbed = false, largest = false, remove_contained = false, is_paf = false;
    int ms = 1, ds = 1, is = 1;
    uint32_t max_size = 100;
    uint64_t paired_len = 0, min_aln = 0, min_query = 0, insert = 1000000;
    bool do_scaffold = false;
    const bool trim = cmd == "trim-paf" || cmd == "trim" || cmd == "tp";
    const bool filter = cmd == "filter", orient = cmd == "orient", nucfreq = cmd == "nucfreq";
    std::string region;
    bool small = false;
    for (; a < argc; a++) {
        const std::string s = argv[a];
        auto next = [&]() -> const char * { return a + 1 < argc ? argv[++a] : ""; };
        if (nucfreq && (s == "-r" || s == "--region")) region = next();
        else if (nucfreq && (s == "-s" || s == "--small")) small = true;
        else if (s == "--paired-len" || (s == "-p" && filter)) paired_len = strtoull(next(), nullptr, 10);
        else if (s == "--query" || (s == "-q" && filter)) min_query = strtoull(next(), nullptr, 10);
        else if (s == "--aln" || (s == "-a" && filter)) min_aln = strtoull(next(), nullptr, 10);
        else if (s == "--insert" || (s == "-i" && orient)) insert = strtoull(next(), nullptr, 10);
        else if (s == "--scaffold" || (s == "-s" && orient)) do_scaffold = true;
        else if (s == "-p" || s == "--paf") is_paf = true;
        else if (s == "-q" || s == "--qbed") qbed = true;
        else if (s == "-l" || s == "--largest") largest = true;
        else if (s == "-r" || s == "--remove-contained") remove_contained = true;
        else if (s == "-b" || s == "--bed") bed_path = next();
        else if (s == "--max-size" || (s == "-m" && !trim)) max_size = (uint32_t)strtoul(next(), nullptr, 10);
        else if (s == "--match-score" || (s == "-m" && trim)) ms = atoi(next());
        else if (s == "-d" || s == "--diff-score") ds = atoi(next());
        else if (s == "-i" || s == "--indel-score") is = atoi(next());
        else paf_path = s;
    }
    if (gpus > 1) {
        const bool lift = cmd == "liftover" || cmd == "lo", brk = cmd == "break-paf" || cmd == "breakpaf" || cmd == "bp";
        const bool stats_paf = cmd == "stats" && is_paf;
        const bool regr = cmd == "regroup";
        if (!((lift && !bed_path.empty()) || brk || stats_paf || cmd == "invert" || trim || regr)) {
            fprintf(stderr, "rb: --gpus shards PAF records: liftover, break-paf, stats --paf, invert, trim-paf\n");
            return 2;
        }
        const GatherMode mode = lift || (regr && !qbed) ? (largest ? GATHER_LARGEST : GATHER_CONTIG) : GATHER_CONCAT;
        const int rc = shard_fork(gpus, mode, trim || (regr && qbed), paf_path, device, stats_paf ? rb::cigar_stats_header(qbed) : std::string());
        if (rc >= 0) return rc;
    }
    if (cmd == "regroup") return regroup(paf_path, qbed, largest);
    try {
        double tl = now_s();
        // text in -> text out (CIGAR text parsed / printed on the device) for regular files; stdin and RB_GENERAL_PATH=1 take the
        // record-based path, which every other arm uses anyway
        const bool text_path = !getenv("RB_GENERAL_PATH"); // (also for stdin: its text is kept, rb_host.cpp stdin_text)
        rb::Engine eng(device);
        lap("device context", tl);
        eng.bsearch_policy = policy;
        if (nucfreq) { // main.rs:82-121: --region first, then the bed file
            std::vector<rb::Region> rgns;
            if (!region.empty()) rgns.push_back(rb::parse_region(region));
            if (!bed_path.empty())
                for (rb::Region &r : rb::parse_bed(bed_path)) rgns.push_back(std::move(r));
            rb::nucfreq_bam(eng, paf_path, rgns, small, [&](const std::string &t) { put(t); });
        } else if (cmd == "stats") {
            if (!g_worker.on) put(rb::cigar_stats_header(qbed)); // (`--gpus`: the parent has printed it)
            if (!is_paf) { // BAM input (main.rs:60-77)
                std::string panic;
                for (const rb::Stats &s : rb::cigar_stats_bam(eng, paf_path, &panic)) put(rb::cigar_stats_line(s, qbed));
                if (!panic.empty()) throw rb::Panic(panic);
            } else {
                std::vector<std::string> text;
                if (text_path && rb::stats_file_text(eng, paf_path, qbed, text)) {
                    emit(text);
                } else {
                    rb::Paf paf = rb::Paf::from_file(eng, paf_path);
                    text.assign(1, std::string());
                    for (const rb::Stats &s : rb::stats_from_paf(eng, paf.records)) text[0] += rb::cigar_stats_line(s, qbed);
                    emit(text);
                }
            }
        } else if (cmd == "invert") {
            std::vector<std::string> text;
            if (!(text_path && rb::invert_file_text(eng, paf_path, text))) {
                rb::Paf paf = rb::Paf::from_file(eng, paf_path);
                text = rb::records_to_text(rb::paf_swap_query_and_target(eng, paf.records));
            }
            emit(text);
        } else if (cmd == "liftover" || cmd == "lo") {
            if (bed_path.empty()) return usage();
            std::vector<rb::Region> rgns = rb::parse_bed(bed_path);
            rb::TextRuns runs;
            auto with_runs = [&](std::vector<std::string> &text) { // a worker of `--gpus` announces where its contigs lie
                Output o;
                o.chunks.swap(text);
                if (g_worker.on) {
                    o.contigs = runs.contigs;
                    for (const auto &r : runs.runs) {
                        Piece p;
                        p.contig = r.first, p.bytes = r.second;
                        o.pieces.push_back(p);
                    }
                    if (o.contigs.empty()) o.contigs.push_back(std::string()); // (no record at all)
                    if (o.pieces.empty()) o.pieces.push_back(Piece());
                }
                return o;
            };
            if (!largest && !qbed && text_path && try_pipelined(device, policy, false, 0, paf_path, rgns)) {
                lap("liftover (text to text, pipelined)", tl);
                done(0);
            }
            if (!largest && !qbed && text_path) { // text in -> text out, CIGAR text handled on the device
                std::vector<std::string> text;
                if (rb::liftover_file_text(eng, paf_path, rgns, text, g_worker.on ? &runs : nullptr)) {
                    lap("liftover (text to text)", tl);
                    Output o = with_runs(text);
                    emit(o);
                    fflush(stdout);
                    lap("write", tl);
                    done(0);
                }
            }
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            lap("decode + check_integrity", tl);
            if (largest) { // main.rs:200-208: stable sort by id, keep the LAST record with maximal target span per id
                std::vector<rb::PafRecord> out = rb::trim_paf_by_rgns(eng, rgns, paf.records, qbed);
                Output o;
                o.chunks.emplace_back();
                std::unordered_map<std::string, uint32_t> cid; // (a worker: the contigs of its records, first appearance order)
                if (g_worker.on)
                    for (const rb::PafRecord &r : paf.records) {
                        const std::string &nm = qbed ? r.q_name : r.t_name;
                        if (cid.emplace(nm, (uint32_t)o.contigs.size()).second) o.contigs.push_back(nm);
                    }
                std::stable_sort(out.begin(), out.end(), [](const rb::PafRecord &x, const rb::PafRecord &y) { return x.id < y.id; });
                for (size_t i = 0; i < out.size();) {
                    // a worker keeps one candidate per (id, contig): equal ids are still in canonical, i.e. contig-major, order
                    size_t j = i, best = i;
                    for (; j < out.size() && out[j].id == out[i].id && (!g_worker.on || out[j].t_name == out[i].t_name); j++)
                        if (out[j].t_en - out[j].t_st >= out[best].t_en - out[best].t_st) best = j;
                    const std::string line = out[best].to_string() + "\n";
                    o.chunks[0] += line;
                    if (g_worker.on) {
                        Piece p;
                        p.contig = cid[out[best].t_name], p.key = out[best].id, p.aux = out[best].t_en - out[best].t_st, p.bytes = line.size();
                        o.pieces.push_back(std::move(p));
                    }
                    i = j;
                }
                if (g_worker.on && o.contigs.empty()) o.contigs.push_back(std::string());
                if (g_worker.on && o.pieces.empty()) o.pieces.push_back(Piece());
                emit(o);
            } else {
                std::vector<std::string> text = rb::trim_paf_by_rgns_text(eng, rgns, paf.records, qbed, g_worker.on ? &runs : nullptr);
                lap("liftover (device + encode)", tl);
                Output o = with_runs(text);
                emit(o);
                fflush(stdout);
                lap("write", tl);
            }
        } else if (cmd == "break-paf" || cmd == "breakpaf" || cmd == "bp") {
            if (text_path && try_pipelined(device, policy, true, max_size, paf_path, std::vector<rb::Region>())) done(0);
            std::vector<std::string> text;
            if (!(text_path && rb::break_file_text(eng, paf_path, max_size, text))) {
                rb::Paf paf = rb::Paf::from_file(eng, paf_path);
                text = rb::break_paf_on_indels_text(eng, paf.records, max_size);
            }
            emit(text);
        } else if (filter) { // main.rs:234-249
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            paf.filter_query_len(min_query);
            paf.filter_aln_len(min_aln);
            paf.filter_aln_pairs(paired_len);
            put(rb::records_to_text(paf.records));
        } else if (orient) { // main.rs:253-267
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            paf.orient();
            if (do_scaffold) paf.scaffold(insert);
            put(rb::records_to_text(paf.records));
        } else if (trim) {
            std::vector<std::string> ttext;
            if (text_path && rb::trim_file_text(eng, paf_path, ms, ds, is, remove_contained, ttext)) {
                lap("trim-paf (text to text)", tl);
                emit(ttext);
                fflush(stdout);
                lap("write", tl);
                done(0);
            }
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            lap("decode + check_integrity", tl);
            paf.overlapping_paf_recs(eng, ms, ds, is, remove_contained);
            lap("overlapping_paf_recs (passes)", tl);
            std::vector<std::string> text = rb::records_to_text(paf.records);
            lap("encode", tl);
            emit(text);
            fflush(stdout);
            lap("write", tl);
        } else {
            return usage();
        }
        done(0);
    } catch (const rb::Panic &e) {
        fflush(stdout);
        fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
        return 101;
    } catch (const std::exception &e) {
        fflush(stdout);
        fprintf(stderr, "rb: %s\n", e.what());
        return 1;
    }
    return 0;
}
