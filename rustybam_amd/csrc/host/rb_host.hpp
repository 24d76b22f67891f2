// rb_host.hpp -- C++ host mirror of the reference's interface for the CIGAR-walk path, over the C ABI
// (include/rustybam_amd.h).  Same names, argument meaning and error behaviour as the Rust functions it
// stands in for; where the reference panics this layer throws rb::Panic (the rb binary then exits 101,
// Rust's panic exit code).  Text decode/encode stays on the CPU; every CIGAR walk goes to the device.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/rustybam_amd.h"

namespace rb {

struct Panic : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// paf::PafRecord (paf.rs:346-368); cigar is packed len << 4 | op
struct PafRecord {
    std::string q_name;
    uint64_t q_len = 0, q_st = 0, q_en = 0;
    char strand = '+';
    std::string t_name;
    uint64_t t_len = 0, t_st = 0, t_en = 0, nmatch = 0, aln_len = 0, mapq = 0;
    std::vector<uint32_t> cigar;
    std::string id;
    uint64_t order = 0;            // set by Paf::orient (paf.rs:143)
    std::string to_string() const; // impl Display (paf.rs:923-944)
};

// bed::Region (bed.rs:15-21)
struct Region {
    std::string name;
    uint64_t st = 0, en = 0;
    std::string id;
};

// bamstats::Stats (bamstats.rs:16-36)
struct Stats {
    std::string q_nm, r_nm;
    int64_t q_len = 0, q_st = 0, q_en = 0, r_len = 0, r_st = 0, r_en = 0;
    char strand = '+';
    uint32_t equal = 0, diff = 0, ins = 0, del = 0, matches = 0, ins_events = 0, del_events = 0;
    float id_by_all = 0, id_by_events = 0, id_by_matches = 0;
};

class Engine { // one rb_ctx
  public:
    explicit Engine(int device = 0);
    ~Engine();
    rb_ctx *ctx() const { return ctx_; }
    int bsearch_policy = RB_BSEARCH_MODERN;
    void check(int rc, const char *what) const;

  private:
    rb_ctx *ctx_ = nullptr;
};

struct Paf { // paf::Paf (paf.rs:34-37)
    std::vector<PafRecord> records;
    // Paf::from_file (paf.rs:62-78): decode + check_integrity().unwrap() on every record (on the device)
    static Paf from_file(Engine &eng, const std::string &file_name);
    // Paf::overlapping_paf_recs (paf.rs:210-305)
    void overlapping_paf_recs(Engine &eng, int match_score, int diff_score, int indel_score, bool remove_contained);
    // header-only commands that sit between the CIGAR-walk stages of a pipeline (no device work: they never touch a CIGAR)
    void filter_aln_pairs(uint64_t paired_len);  // paf.rs:91-102
    void filter_query_len(uint64_t min_query_len); // paf.rs:104-106
    void filter_aln_len(uint64_t min_aln_len);   // paf.rs:109-111
    void orient();                               // paf.rs:114-157 (panics on a zero-bp (target, query) pair: divide by zero)
    void scaffold(uint64_t spacer_size);         // paf.rs:160-207
};

// `rb --gpus N` (the record shards of liftover.rs:123-129, one worker process per GPU): the readers below then load only bytes
// [begin, end) of a plain input file; read_input_text: the whole decompressed text of a file or stdin ("-")
void set_input_slice(uint64_t begin, uint64_t end);
// `rb --gpus N trim-paf` (query groups are the unit, paf.rs:223): the readers keep only the lines whose query name lies in [lo, hi)
// (bytewise order, nullptr = open); query_name_cuts: the names that cut a plain file's sorted query names into n ranges of about equal bytes
void set_input_query_range(const std::string *lo, const std::string *hi);
std::vector<std::string> query_name_cuts(const std::string &file_name, int n);
// where the contigs lie in a liftover output (canonical order is contig-major, liftover.rs:151-164): what `rb --gpus N` needs to
// put the shards' outputs together.  contigs = target names of the (possibly swapped) records in order of first appearance,
// runs = (index into contigs, bytes of output text) in output order
struct TextRuns {
    std::vector<std::string> contigs;
    std::vector<std::pair<uint32_t, uint64_t>> runs;
};
std::string read_input_text(const std::string &file_name);

std::string cigar_to_string(const std::vector<uint32_t> &cigar);
std::vector<std::string> records_to_text(const std::vector<PafRecord> &recs); // `println!("{}", rec)` for every record, encoded on all host cores; chunks in output order
// PafRecord::new (paf.rs:379-430): 0 = ok, 1 = Err(ParsePafColumn) (caller skips the line); throws Panic
int paf_record_new(const std::string &line, PafRecord &out);
std::vector<Region> parse_bed(const std::string &filename);                    // bed.rs:172-194
std::string f32_display(float v);                                              // Rust `{}` for f32

// paf_swap_query_and_target for a whole record set (paf.rs:1068-1094)
std::vector<PafRecord> paf_swap_query_and_target(Engine &eng, const std::vector<PafRecord> &recs);
// liftover::trim_paf_by_rgns (liftover.rs:134-167), single-thread output order
std::vector<PafRecord> trim_paf_by_rgns(Engine &eng, const std::vector<Region> &rgns, const std::vector<PafRecord> &paf_recs, bool invert_query);
// the same, printed: every Some(rec) as `println!("{}", rec)` would, without materialising the records
std::vector<std::string> trim_paf_by_rgns_text(Engine &eng, const std::vector<Region> &rgns, const std::vector<PafRecord> &paf_recs, bool invert_query,
                                               TextRuns *runs = nullptr);
// main.rs:186-214 without --qbed / --largest, text in -> text out: the CIGAR text is parsed and printed on the device
// (rb_host_liftover_text); false = the file needs the general path (a line with two cg tags), nothing was produced
bool liftover_file_text(Engine &eng, const std::string &paf_path, const std::vector<Region> &rgns, std::vector<std::string> &out_text,
                        TextRuns *runs = nullptr);
// the same two routes as a pipeline over chunks of a big plain file (a few host threads, each with its own context on `device`):
// the sink receives the chunks' outputs in file order while later chunks are still being read / clipped / printed.  A chunk's text
// is contig-major within the chunk (runs); false = not applicable or a line needs the general parser (pipeline_started(): the
// sink has already been given chunks)
bool lift_file_text_pipelined(int device, int bsearch_policy, bool is_break, uint32_t break_length, const std::string &paf_path,
                              const std::vector<Region> &rgns, const std::function<bool(std::vector<std::string> &, TextRuns &)> &sink); // sink: false = stop, the caller takes the whole-file route
bool pipeline_started();
bool break_file_text(Engine &eng, const std::string &paf_path, uint32_t break_length, std::vector<std::string> &out_text); // main.rs:271-281
bool trim_file_text(Engine &eng, const std::string &paf_path, int match_score, int diff_score, int indel_score, bool remove_contained,
                    std::vector<std::string> &out_text); // main.rs:218-230, the batch resident on the device across the passes
bool invert_file_text(Engine &eng, const std::string &paf_path, std::vector<std::string> &out_text); // main.rs:176-182, CIGARs parsed, swapped and printed on the device
bool stats_file_text(Engine &eng, const std::string &paf_path, bool qbed, std::vector<std::string> &out_text);          // main.rs:50-58, lines without the header
std::vector<std::string> break_paf_on_indels_text(Engine &eng, const std::vector<PafRecord> &paf_recs, uint32_t break_length);
// main.rs:274-280: aligned_pairs + liftover::break_paf_on_indels (liftover.rs:182-226) for every record, record order
std::vector<PafRecord> break_paf_on_indels(Engine &eng, const std::vector<PafRecord> &paf_recs, uint32_t break_length);
// bamstats::stats_from_paf (bamstats.rs:91-154) for every record
std::vector<Stats> stats_from_paf(Engine &eng, const std::vector<PafRecord> &paf_recs);
// bamstats::parse_md_for_stats (bamstats.rs:48-79): (match_count, mismatch_count, insertion_count, insertion_bases)
void parse_md_for_stats(const std::string &md, uint32_t out[4]);
// main.rs:60-77 + bamstats::cigar_stats (bamstats.rs:156-222): every mapped record of a BAM file (BGZF through zlib);
// the CIGAR counters come from the device record-scan kernel (BAM cigars are already the packed u32 form)
// trailing_panic: where the reference panics part-way through the file (a broken record, read_pos), the stats of the records
// before it come back and the panic message goes here (NULL: thrown instead) -- the reference has printed them by then
std::vector<Stats> cigar_stats_bam(Engine &eng, const std::string &bam_path, std::string *trailing_panic = nullptr);
std::string cigar_stats_header(bool qbed);              // bamstats.rs:225-236
// bed::parse_region (bed.rs:88-131): "name:st-en", 1-based inclusive start -> 0-based half-open; id = the same text
Region parse_region(const std::string &region);
// main.rs:82-121 + nucfreq.rs: A/C/G/T counts at every covered position of every region, printed piece by piece (1 Mbp) through
// `put`; BGZF/BAM decode on the host, the pileup on the device (rb_host_nucfreq)
void nucfreq_bam(Engine &eng, const std::string &bam_path, const std::vector<Region> &rgns, bool small,
                 const std::function<void(const std::string &)> &put);
std::string cigar_stats_line(const Stats &s, bool qbed); // bamstats.rs:239-270

} // namespace rb
