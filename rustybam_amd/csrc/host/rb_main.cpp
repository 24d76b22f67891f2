// rb_main.cpp -- `rb`-compatible front end for the hot-path subcommands (dispatcher arms main.rs:50-58 stats --paf,
// :176-182 invert, :186-214 liftover, :218-230 trim-paf, :271-281 break-paf) over the MI355X engine, plus the two
// header-only commands that sit between them in the reference's pipelines (:234-249 filter, :253-267 orient).
// Same flag names and defaults as src/cli.rs; exits with 101 where the reference panics.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <string>
#include <thread>
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
            "--gpus N: the PAF records in N contiguous shards, one worker process per GPU (liftover without --largest, break-paf,\n"
            "          stats --paf, invert); the output is the single-GPU output byte for byte.\n"
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
static void put(const std::string &s) { fwrite(s.data(), 1, s.size(), stdout); }
// the output of a text route: chunks in output order, gigabytes in all.  Into a regular file they go with pwrite from several
// threads, each chunk at its own offset (the page cache takes the pages in parallel); anywhere else with plain write(2) -- no
// second copy through stdio's buffer either way
static void put(const std::vector<std::string> &chunks) {
    size_t total = 0;
    for (const std::string &s : chunks) total += s.size();
    if (total < ((size_t)16 << 20)) {
        for (const std::string &s : chunks) fwrite(s.data(), 1, s.size(), stdout);
        return;
    }
    fflush(stdout);
    struct stat st;
    const int fl = fcntl(1, F_GETFL);
    off_t at = -1;
    if (fstat(1, &st) == 0 && S_ISREG(st.st_mode) && fl >= 0 && !(fl & O_APPEND)) at = lseek(1, 0, SEEK_CUR);
    if (at >= 0) {
        std::vector<off_t> off(chunks.size());
        off_t o = at;
        for (size_t k = 0; k < chunks.size(); k++) off[k] = o, o += (off_t)chunks[k].size();
        const unsigned T = (unsigned)std::min<size_t>(chunks.size(), 8);
        std::vector<std::thread> th;
        std::vector<int> bad(T, 0);
        for (unsigned t = 0; t < T; t++)
            th.emplace_back([&, t]() {
                for (size_t k = t; k < chunks.size(); k += T) {
                    size_t done_ = 0;
                    while (done_ < chunks[k].size()) {
                        const ssize_t w = pwrite(1, chunks[k].data() + done_, chunks[k].size() - done_, off[k] + (off_t)done_);
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
static int synth_bed(uint64_t n_win) {
    for (uint64_t i = 0; i < n_win; i++) {
        const uint64_t st = i * 82796ull, en = st + 100000ull < 248387497ull ? st + 100000ull : 248387497ull;
        printf("chr1\t%llu\t%llu\n", (unsigned long long)st, (unsigned long long)en);
    }
    return 0;
}

// `rb --gpus N`: SURVEY 8(e) -- PAF records are independent (liftover.rs:123-129 hands them to rayon), so the input is cut into N
// runs of whole lines of about equal bytes (CIGAR text is what weighs) and N worker processes are forked BEFORE anything touches
// the GPU; worker k takes device `device + k`, reads only its lines, and writes to a pipe; the parent never initialises HIP, it
// concatenates the pipes in shard order.  Returns -1 in a worker (which carries on as an ordinary single-GPU run over its slice),
// the exit code in the parent: that of the first shard that failed, after the output of the shards before it and its own.
static int g_rank = 0;
static int shard_fork(int n, std::string &path, int &device) {
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
        std::string all;
        try {
            all = rb::read_input_text(path);
        } catch (const rb::Panic &e) {
            fprintf(stderr, "thread 'main' panicked: %s\n", e.what());
            return 101;
        }
        fd = memfd_create("rb_input", 0);
        size_t off = 0;
        while (fd >= 0 && off < all.size()) {
            const ssize_t w = write(fd, all.data() + off, all.size() - off);
            if (w <= 0) break;
            off += (size_t)w;
        }
        if (fd < 0 || off != all.size()) {
            fprintf(stderr, "rb: cannot buffer the input for --gpus\n");
            return 1;
        }
        path = "/proc/self/fd/" + std::to_string(fd);
    }
    struct stat st;
    fstat(fd, &st);
    const uint64_t size = (uint64_t)st.st_size;
    std::vector<uint64_t> cut(n + 1, size);
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
    const bool same = getenv("RB_GPUS_SAME_DEVICE") != nullptr; // (diagnostic: every worker on --device, to try the gather on one GPU)
    std::vector<int> rd(n, -1);
    std::vector<pid_t> pid(n, -1);
    for (int k = 0; k < n; k++) {
        int pp[2];
        if (pipe(pp) != 0) return 1;
        const pid_t c = fork();
        if (c < 0) return 1;
        if (c == 0) {
            for (int j = 0; j < k; j++) close(rd[j]);
            close(pp[0]);
            dup2(pp[1], 1);
            close(pp[1]);
            g_rank = k;
            if (!same) device += k;
            rb::set_input_slice(cut[k], cut[k + 1]);
            return -1;
        }
        close(pp[1]);
        rd[k] = pp[0], pid[k] = c;
    }
    std::vector<std::string> out(n);
    std::vector<std::thread> th;
    for (int k = 0; k < n; k++)
        th.emplace_back([&, k]() {
            std::vector<char> b(1 << 20);
            ssize_t r;
            while ((r = read(rd[k], b.data(), b.size())) > 0) out[k].append(b.data(), (size_t)r);
            close(rd[k]);
        });
    int rc = 0;
    for (int k = 0; k < n; k++) {
        th[k].join();
        int st_k = 0;
        waitpid(pid[k], &st_k, 0);
        const int rc_k = WIFEXITED(st_k) ? WEXITSTATUS(st_k) : 1;
        if (rc == 0) { // (what a later shard printed after an earlier one failed is not output the single run would have made)
            fwrite(out[k].data(), 1, out[k].size(), stdout);
            rc = rc_k;
        }
    }
    fflush(stdout);
    return rc;
}

int main(int argc, char **argv) {
    g_t_main = now_s();
    setvbuf(stdout, g_obuf, _IOFBF, sizeof g_obuf);
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
        if (!((lift && !largest && !bed_path.empty()) || brk || (cmd == "stats" && is_paf) || cmd == "invert")) {
            fprintf(stderr, "rb: --gpus shards PAF records: liftover (without --largest), break-paf, stats --paf, invert\n");
            return 2;
        }
        const int rc = shard_fork(gpus, paf_path, device);
        if (rc >= 0) return rc;
    }
    try {
        double tl = now_s();
        // text in -> text out (CIGAR text parsed / printed on the device) for regular files; stdin and RB_GENERAL_PATH=1 take the
        // record-based path, which every other arm uses anyway
        const bool text_path = paf_path != "-" && !getenv("RB_GENERAL_PATH");
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
            if (g_rank == 0) put(rb::cigar_stats_header(qbed));
            if (!is_paf) { // BAM input (main.rs:60-77)
                std::string panic;
                for (const rb::Stats &s : rb::cigar_stats_bam(eng, paf_path, &panic)) put(rb::cigar_stats_line(s, qbed));
                if (!panic.empty()) throw rb::Panic(panic);
            } else {
                std::vector<std::string> text;
                if (text_path && rb::stats_file_text(eng, paf_path, qbed, text)) {
                    put(text);
                } else {
                    rb::Paf paf = rb::Paf::from_file(eng, paf_path);
                    for (const rb::Stats &s : rb::stats_from_paf(eng, paf.records)) put(rb::cigar_stats_line(s, qbed));
                }
            }
        } else if (cmd == "invert") {
            std::vector<std::string> text;
            if (text_path && rb::invert_file_text(eng, paf_path, text)) {
                put(text);
            } else {
                rb::Paf paf = rb::Paf::from_file(eng, paf_path);
                put(rb::records_to_text(rb::paf_swap_query_and_target(eng, paf.records)));
            }
        } else if (cmd == "liftover" || cmd == "lo") {
            if (bed_path.empty()) return usage();
            std::vector<rb::Region> rgns = rb::parse_bed(bed_path);
            if (!largest && !qbed && text_path) { // text in -> text out, CIGAR text handled on the device
                std::vector<std::string> text;
                if (rb::liftover_file_text(eng, paf_path, rgns, text)) {
                    lap("liftover (text to text)", tl);
                    put(text);
                    fflush(stdout);
                    lap("write", tl);
                    done(0);
                }
            }
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            lap("decode + check_integrity", tl);
            if (largest) { // main.rs:200-208: stable sort by id, keep the LAST record with maximal target span per id
                std::vector<rb::PafRecord> out = rb::trim_paf_by_rgns(eng, rgns, paf.records, qbed);
                std::stable_sort(out.begin(), out.end(), [](const rb::PafRecord &x, const rb::PafRecord &y) { return x.id < y.id; });
                for (size_t i = 0; i < out.size();) {
                    size_t j = i, best = i;
                    for (; j < out.size() && out[j].id == out[i].id; j++)
                        if (out[j].t_en - out[j].t_st >= out[best].t_en - out[best].t_st) best = j;
                    put(out[best].to_string() + "\n");
                    i = j;
                }
            } else {
                const std::vector<std::string> text = rb::trim_paf_by_rgns_text(eng, rgns, paf.records, qbed);
                lap("liftover (device + encode)", tl);
                put(text);
                fflush(stdout);
                lap("write", tl);
            }
        } else if (cmd == "break-paf" || cmd == "breakpaf" || cmd == "bp") {
            std::vector<std::string> text;
            if (text_path && rb::break_file_text(eng, paf_path, max_size, text)) {
                put(text);
            } else {
                rb::Paf paf = rb::Paf::from_file(eng, paf_path);
                put(rb::break_paf_on_indels_text(eng, paf.records, max_size));
            }
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
                put(ttext);
                fflush(stdout);
                lap("write", tl);
                done(0);
            }
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            lap("decode + check_integrity", tl);
            paf.overlapping_paf_recs(eng, ms, ds, is, remove_contained);
            lap("overlapping_paf_recs (passes)", tl);
            const std::vector<std::string> text = rb::records_to_text(paf.records);
            lap("encode", tl);
            put(text);
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
