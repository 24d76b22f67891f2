/*
 * rbo_main.c -- command-line front end of the CPU ORACLE (test infrastructure only).
 * Mirrors the dispatcher arms of the reference for the hot-path subcommands
 * (main.rs:50-58 stats --paf, :176-182 invert, :186-214 liftover, :218-230 trim-paf,
 * :271-281 break-paf) so that file-level outputs can be diffed / digested.
 */
#include "rb_oracle.h"

#include <stdlib.h>
#include <string.h>

static int policy = RBO_BSEARCH_MODERN;

static int cmp_id_stable(const void *a, const void *b) {
    const rbo_rec *const *x = (const rbo_rec *const *)a, *const *y = (const rbo_rec *const *)b;
    int c = strcmp((*x)->id, (*y)->id);
    if (c) return c;
    return (*x < *y) ? -1 : (*x > *y);
}

static int usage(void) {
    fprintf(stderr,
            "usage: rb_oracle [--bsearch modern|legacy] <cmd> ...\n"
            "  stats [--qbed] --paf <paf>\n"
            "  liftover --bed <bed> [--qbed] [--largest] <paf>\n"
            "  break-paf [--max-size N] <paf>\n"
            "  trim-paf [--match-score 1] [--diff-score 1] [--indel-score 1] [--remove-contained] <paf>\n"
            "  invert <paf>\n"
            "  nucfreq [--region chr:st-en] [--bed <bed>] [--small] <bam>\n");
    return 2;
}

int main(int argc, char **argv) {
    int a = 1;
    if (a + 1 < argc && strcmp(argv[a], "--bsearch") == 0) {
        policy = strcmp(argv[a + 1], "legacy") == 0 ? RBO_BSEARCH_LEGACY : RBO_BSEARCH_MODERN;
        a += 2;
    }
    if (a >= argc) return usage();
    const char *cmd = argv[a++];
    const char *paf_path = NULL, *bed_path = NULL;
    int qbed = 0, largest = 0, remove_contained = 0, is_paf = 0;
    int ms = 1, ds = 1, is = 1, scaffold = 0;
    uint32_t max_size = 100;
    uint64_t paired_len = 0, min_aln = 0, min_query = 0, insert = 1000000;
    const char *region = NULL;
    int small = 0;
    const int is_nucfreq = !strcmp(cmd, "nucfreq");
    for (; a < argc; a++) {
        const int is_filter = !strcmp(cmd, "filter");
        if (is_nucfreq && (!strcmp(argv[a], "--region") || !strcmp(argv[a], "-r")) && a + 1 < argc) { region = argv[++a]; continue; }
        if (is_nucfreq && (!strcmp(argv[a], "--small") || !strcmp(argv[a], "-s"))) { small = 1; continue; }
        if (!strcmp(argv[a], "--paf") || (!strcmp(argv[a], "-p") && !is_filter)) { is_paf = 1; continue; }
        else if (!strcmp(argv[a], "--qbed") || (!strcmp(argv[a], "-q") && !is_filter)) qbed = 1;
        else if (!strcmp(argv[a], "--largest") || !strcmp(argv[a], "-l")) largest = 1;
        else if (!strcmp(argv[a], "--remove-contained") || !strcmp(argv[a], "-r")) remove_contained = 1;
        else if ((!strcmp(argv[a], "--bed") || !strcmp(argv[a], "-b")) && a + 1 < argc) bed_path = argv[++a];
        else if ((!strcmp(argv[a], "--max-size") || !strcmp(argv[a], "-m")) && a + 1 < argc) max_size = (uint32_t)strtoul(argv[++a], NULL, 10);
        else if (!strcmp(argv[a], "--match-score") && a + 1 < argc) ms = atoi(argv[++a]);
        else if ((!strcmp(argv[a], "--paired-len") || (!strcmp(argv[a], "-p") && !strcmp(cmd, "filter"))) && a + 1 < argc) paired_len = strtoull(argv[++a], NULL, 10);
        else if ((!strcmp(argv[a], "--aln") || !strcmp(argv[a], "-a")) && a + 1 < argc) min_aln = strtoull(argv[++a], NULL, 10);
        else if ((!strcmp(argv[a], "--query") || (!strcmp(argv[a], "-q") && !strcmp(cmd, "filter"))) && a + 1 < argc) min_query = strtoull(argv[++a], NULL, 10);
        else if ((!strcmp(argv[a], "--insert") || !strcmp(argv[a], "-i")) && a + 1 < argc && !strcmp(cmd, "orient")) insert = strtoull(argv[++a], NULL, 10);
        else if ((!strcmp(argv[a], "--scaffold") || !strcmp(argv[a], "-s")) && !strcmp(cmd, "orient")) scaffold = 1;
        else if (!strcmp(argv[a], "--diff-score") && a + 1 < argc) ds = atoi(argv[++a]);
        else if (!strcmp(argv[a], "--indel-score") && a + 1 < argc) is = atoi(argv[++a]);
        else paf_path = argv[a];
    }
    if (!paf_path) paf_path = "-";
    if (is_nucfreq) { /* main.rs:82-121 */
        int nrc = rbo_bam_nucfreq(paf_path, region, bed_path, small, stdout);
        return nrc ? 101 : 0;
    }
    if (!strcmp(cmd, "stats") && !is_paf) { /* BAM input, main.rs:60-77 */
        int brc = rbo_bam_stats(paf_path, qbed, stdout);
        return brc ? 101 : 0;
    }
    rbo_paf paf;
    int rc = rbo_paf_from_file(paf_path, &paf);
    if (rc) {
        fprintf(stderr, "rb_oracle: cannot load %s (rc %d)\n", paf_path, rc);
        return 101; /* Rust panic exit code */
    }
    if (!strcmp(cmd, "stats")) {
        rbo_print_stats_header(qbed, stdout);
        for (size_t i = 0; i < paf.n; i++) {
            rbo_stats s;
            rbo_stats_from_cigar(paf.recs[i].cigar, paf.recs[i].n_cigar, &s);
            rbo_print_stats(&paf.recs[i], &s, qbed, stdout);
        }
    } else if (!strcmp(cmd, "invert")) {
        for (size_t i = 0; i < paf.n; i++) {
            rbo_rec f;
            rbo_paf_swap_query_and_target(&paf.recs[i], &f);
            rbo_rec_print(&f, stdout);
            rbo_rec_free(&f);
        }
    } else if (!strcmp(cmd, "liftover")) {
        if (!bed_path) return usage();
        rbo_bed bed;
        if (rbo_bed_from_file(bed_path, &bed)) {
            fprintf(stderr, "rb_oracle: cannot read %s\n", bed_path);
            return 101;
        }
        rbo_paf out;
        rc = rbo_trim_paf_by_rgns(&bed, &paf, qbed, policy, &out);
        if (rc) return 101;
        if (largest) { /* main.rs:200-208: stable sort by id, group, max_by_key keeps the LAST maximum */
            const rbo_rec **v = (const rbo_rec **)malloc((out.n + 1) * sizeof(*v));
            for (size_t i = 0; i < out.n; i++) v[i] = &out.recs[i];
            qsort(v, out.n, sizeof(*v), cmp_id_stable);
            size_t i = 0;
            while (i < out.n) {
                size_t j = i, best = i;
                while (j < out.n && strcmp(v[j]->id, v[i]->id) == 0) {
                    if (v[j]->t_en - v[j]->t_st >= v[best]->t_en - v[best]->t_st) best = j;
                    j++;
                }
                rbo_rec_print(v[best], stdout);
                i = j;
            }
            free(v);
        } else {
            for (size_t i = 0; i < out.n; i++) rbo_rec_print(&out.recs[i], stdout);
        }
        rbo_paf_free(&out);
        rbo_bed_free(&bed);
    } else if (!strcmp(cmd, "break-paf")) {
        for (size_t i = 0; i < paf.n; i++) {
            int ap = rbo_aligned_pairs(&paf.recs[i]);
            if (ap) return 101;
            rbo_paf out = {0};
            rc = rbo_break_paf_on_indels(&paf.recs[i], max_size, policy, &out);
            if (rc) return 101;
            for (size_t k = 0; k < out.n; k++) rbo_rec_print(&out.recs[k], stdout);
            rbo_paf_free(&out);
            rbo_rec_drop_aln(&paf.recs[i]);
        }
    } else if (!strcmp(cmd, "filter")) { /* main.rs:234-249 */
        rbo_paf_filter(&paf, paired_len, min_aln, min_query);
        for (size_t i = 0; i < paf.n; i++) rbo_rec_print(&paf.recs[i], stdout);
    } else if (!strcmp(cmd, "orient")) { /* main.rs:253-267 */
        uint64_t *orders = (uint64_t *)malloc((paf.n + 1) * sizeof(uint64_t));
        if (rbo_paf_orient(&paf, orders)) return 101;
        if (scaffold) rbo_paf_scaffold(&paf, orders, insert);
        for (size_t i = 0; i < paf.n; i++) rbo_rec_print(&paf.recs[i], stdout);
        free(orders);
    } else if (!strcmp(cmd, "trim-paf")) {
        rc = rbo_overlapping_paf_recs(&paf, ms, ds, is, remove_contained, policy);
        if (rc) return 101;
        for (size_t i = 0; i < paf.n; i++) rbo_rec_print(&paf.recs[i], stdout);
    } else {
        return usage();
    }
    rbo_paf_free(&paf);
    return 0;
}
