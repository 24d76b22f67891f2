// rb_main.cpp -- `rb`-compatible front end for the hot-path subcommands (dispatcher arms main.rs:50-58 stats --paf,
// :176-182 invert, :186-214 liftover, :218-230 trim-paf, :271-281 break-paf) over the MI355X engine.
// Same flag names and defaults as src/cli.rs; exits with 101 where the reference panics.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "rb_host.hpp"

static int usage() {
    fprintf(stderr,
            "usage: rb [--bsearch modern|legacy] [--device N] <subcommand> ...\n"
            "  stats [-q|--qbed] [-p|--paf] <PAF or BAM>\n"
            "  liftover -b|--bed <BED> [-q|--qbed] [-l|--largest] [PAF]\n"
            "  break-paf [-m|--max-size 100] [PAF]\n"
            "  trim-paf [-m|--match-score 1] [-d|--diff-score 1] [-i|--indel-score 1] [-r|--remove-contained] [PAF]\n"
            "  invert [PAF]\n"
            "Every other rustybam subcommand is outside this engine's scope.\n");
    return 2;
}

static void put(const std::string &s) { fwrite(s.data(), 1, s.size(), stdout); }

int main(int argc, char **argv) {
    int a = 1, device = 0, policy = RB_BSEARCH_MODERN;
    while (a + 1 < argc && argv[a][0] == '-') {
        if (!strcmp(argv[a], "--bsearch")) policy = !strcmp(argv[a + 1], "legacy") ? RB_BSEARCH_LEGACY : RB_BSEARCH_MODERN;
        else if (!strcmp(argv[a], "--device")) device = atoi(argv[a + 1]);
        else if (!strcmp(argv[a], "-t") || !strcmp(argv[a], "--threads")) { /* accepted, unused */ }
        else break;
        a += 2;
    }
    if (a >= argc) return usage();
    const std::string cmd = argv[a++];
    std::string paf_path = "-", bed_path;
    bool qbed = false, largest = false, remove_contained = false, is_paf = false;
    int ms = 1, ds = 1, is = 1;
    uint32_t max_size = 100;
    const bool trim = cmd == "trim-paf" || cmd == "trim" || cmd == "tp";
    for (; a < argc; a++) {
        const std::string s = argv[a];
        auto next = [&]() -> const char * { return a + 1 < argc ? argv[++a] : ""; };
        if (s == "-p" || s == "--paf") is_paf = true;
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
    try {
        rb::Engine eng(device);
        eng.bsearch_policy = policy;
        std::vector<char> obuf(1 << 22);
        setvbuf(stdout, obuf.data(), _IOFBF, obuf.size());
        if (cmd == "stats") {
            put(rb::cigar_stats_header(qbed));
            if (!is_paf) { // BAM input (main.rs:60-77)
                for (const rb::Stats &s : rb::cigar_stats_bam(eng, paf_path)) put(rb::cigar_stats_line(s, qbed));
            } else {
                rb::Paf paf = rb::Paf::from_file(eng, paf_path);
                for (const rb::Stats &s : rb::stats_from_paf(eng, paf.records)) put(rb::cigar_stats_line(s, qbed));
            }
        } else if (cmd == "invert") {
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            for (const rb::PafRecord &r : rb::paf_swap_query_and_target(eng, paf.records)) put(r.to_string() + "\n");
        } else if (cmd == "liftover" || cmd == "lo") {
            if (bed_path.empty()) return usage();
            std::vector<rb::Region> rgns = rb::parse_bed(bed_path);
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            std::vector<rb::PafRecord> out = rb::trim_paf_by_rgns(eng, rgns, paf.records, qbed);
            if (largest) { // main.rs:200-208: stable sort by id, keep the LAST record with maximal target span per id
                std::stable_sort(out.begin(), out.end(), [](const rb::PafRecord &x, const rb::PafRecord &y) { return x.id < y.id; });
                for (size_t i = 0; i < out.size();) {
                    size_t j = i, best = i;
                    for (; j < out.size() && out[j].id == out[i].id; j++)
                        if (out[j].t_en - out[j].t_st >= out[best].t_en - out[best].t_st) best = j;
                    put(out[best].to_string() + "\n");
                    i = j;
                }
            } else {
                for (const rb::PafRecord &r : out) put(r.to_string() + "\n");
            }
        } else if (cmd == "break-paf" || cmd == "breakpaf" || cmd == "bp") {
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            for (const rb::PafRecord &r : rb::break_paf_on_indels(eng, paf.records, max_size)) put(r.to_string() + "\n");
        } else if (trim) {
            rb::Paf paf = rb::Paf::from_file(eng, paf_path);
            paf.overlapping_paf_recs(eng, ms, ds, is, remove_contained);
            for (const rb::PafRecord &r : paf.records) put(r.to_string() + "\n");
        } else {
            return usage();
        }
        fflush(stdout);
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
