/*
 * rb_oracle.c -- CPU ORACLE (test infrastructure; see rb_oracle.h for the rules).
 *
 * Plain-C, PER-BASE restatement of the reference hot path (mrvollger/rustybam v0.1.33).
 * Citations "paf.rs:NNN" etc. are into /root/reference/src/.
 * Parity pinning: tests/golden/known_answers.json (KA1..KA12) + SURVEY.md 8c digests;
 * unpinned third-party behaviours are listed in rb_oracle.h.
 */
#define _GNU_SOURCE
#include "rb_oracle.h"

#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

/* ------------------------------------------------------------------ small helpers */
static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) {
        fprintf(stderr, "rb_oracle: out of memory (%zu bytes)\n", n);
        abort();
    }
    return p;
}
static void *xrealloc(void *q, size_t n) {
    void *p = realloc(q, n ? n : 1);
    if (!p) {
        fprintf(stderr, "rb_oracle: out of memory (%zu bytes)\n", n);
        abort();
    }
    return p;
}
static char *xstrdup(const char *s) {
    size_t n = strlen(s);
    char *p = (char *)xmalloc(n + 1);
    memcpy(p, s, n + 1);
    return p;
}
static char *xstrndup(const char *s, size_t n) {
    char *p = (char *)xmalloc(n + 1);
    memcpy(p, s, n);
    p[n] = 0;
    return p;
}
void rbo_free(void *p) { free(p); }

static const char OPCHARS[] = "MIDNSHP=X";

/* paf.rs:946-951 */
int rbo_consumes_reference(rbo_cig op) {
    op &= 15;
    return op == RBO_M || op == RBO_D || op == RBO_N || op == RBO_X || op == RBO_EQ;
}
/* paf.rs:958-963 */
int rbo_consumes_query(rbo_cig op) {
    op &= 15;
    return op == RBO_M || op == RBO_I || op == RBO_S || op == RBO_X || op == RBO_EQ;
}
/* paf.rs:973-975 */
int rbo_is_match(rbo_cig op) {
    op &= 15;
    return op == RBO_M || op == RBO_X || op == RBO_EQ;
}
static int is_indel(rbo_cig op) {
    op &= 15;
    return op == RBO_I || op == RBO_D;
}

/* ------------------------------------------------------------------ record lifecycle */
void rbo_rec_init(rbo_rec *r) {
    memset(r, 0, sizeof(*r));
    r->q_name = xstrdup("");
    r->t_name = xstrdup("");
    r->id = xstrdup("");
    r->strand = '+';
}
void rbo_rec_drop_aln(rbo_rec *r) {
    free(r->tpos_aln);
    free(r->qpos_aln);
    free(r->long_cigar);
    r->tpos_aln = r->qpos_aln = NULL;
    r->long_cigar = NULL;
    r->n_aln = 0;
}
void rbo_rec_free(rbo_rec *r) {
    free(r->q_name);
    free(r->t_name);
    free(r->id);
    free(r->cigar);
    rbo_rec_drop_aln(r);
    memset(r, 0, sizeof(*r));
}
static void *memdup(const void *p, size_t n) {
    if (!p || !n) return NULL;
    void *q = xmalloc(n);
    memcpy(q, p, n);
    return q;
}
void rbo_rec_clone(rbo_rec *dst, const rbo_rec *src) {
    *dst = *src;
    dst->q_name = xstrdup(src->q_name);
    dst->t_name = xstrdup(src->t_name);
    dst->id = xstrdup(src->id);
    dst->cigar = (rbo_cig *)memdup(src->cigar, src->n_cigar * sizeof(rbo_cig));
    dst->tpos_aln = (uint64_t *)memdup(src->tpos_aln, src->n_aln * sizeof(uint64_t));
    dst->qpos_aln = (uint64_t *)memdup(src->qpos_aln, src->n_aln * sizeof(uint64_t));
    dst->long_cigar = (uint8_t *)memdup(src->long_cigar, src->n_aln);
}
/* paf.rs:433-456 small_copy: everything but cigar and expansion */
static void rec_small_copy(rbo_rec *dst, const rbo_rec *src) {
    *dst = *src;
    dst->q_name = xstrdup(src->q_name);
    dst->t_name = xstrdup(src->t_name);
    dst->id = xstrdup(src->id);
    dst->cigar = NULL;
    dst->n_cigar = 0;
    dst->tpos_aln = dst->qpos_aln = NULL;
    dst->long_cigar = NULL;
    dst->n_aln = 0;
}
static void paf_push(rbo_paf *p, rbo_rec *r) { /* takes ownership */
    if (p->n == p->cap) {
        p->cap = p->cap ? p->cap * 2 : 16;
        p->recs = (rbo_rec *)xrealloc(p->recs, p->cap * sizeof(rbo_rec));
    }
    p->recs[p->n++] = *r;
    memset(r, 0, sizeof(*r));
}
void rbo_paf_free(rbo_paf *p) {
    for (size_t i = 0; i < p->n; i++) rbo_rec_free(&p->recs[i]);
    free(p->recs);
    memset(p, 0, sizeof(*p));
}
void rbo_bed_free(rbo_bed *b) {
    for (size_t i = 0; i < b->n; i++) {
        free(b->r[i].name);
        free(b->r[i].id);
    }
    free(b->r);
    memset(b, 0, sizeof(*b));
}

/* ------------------------------------------------------------------ cigar text */
/* rust-htslib 0.44.1 CigarString::try_from(&[u8]) as used at paf.rs:398-399:
 * decimal u32 length then one op char of MIDNSHP=X.  Any violation makes the reference
 * panic via .expect(); we return -1. (Missing digits / overflow: parity unpinned.) */
/* rust-htslib CigarString::try_from (paf.rs:398-399): decimal u32 length, one op character */
static int parse_cigar64(const char *s, size_t n, rbo_cig **ops, size_t *n_ops) {
    size_t cap = 16, cnt = 0;
    rbo_cig *v = (rbo_cig *)xmalloc(cap * sizeof(rbo_cig));
    size_t i = 0;
    while (i < n) {
        size_t j = i;
        uint64_t len = 0;
        while (j < n && s[j] >= '0' && s[j] <= '9') {
            len = len * 10 + (uint64_t)(s[j] - '0');
            if (len > 0xFFFFFFFFull) {
                free(v);
                return -1;
            }
            j++;
        }
        if (j == i || j >= n) {
            free(v);
            return -1;
        }
        const char *p = strchr(OPCHARS, s[j]);
        if (!p || !s[j]) {
            free(v);
            return -1;
        }
        if (cnt == cap) {
            cap *= 2;
            v = (rbo_cig *)xrealloc(v, cap * sizeof(rbo_cig));
        }
        v[cnt++] = (len << 4) | (uint64_t)(p - OPCHARS);
        i = j + 1;
    }
    *ops = v;
    *n_ops = cnt;
    return 0;
}
static size_t cigar64_to_string(const rbo_cig *ops, size_t n, char **out) {
    size_t cap = n * 12 + 1, k = 0;
    char *b = (char *)xmalloc(cap);
    for (size_t i = 0; i < n; i++) k += (size_t)sprintf(b + k, "%llu%c", (unsigned long long)(ops[i] >> 4), OPCHARS[ops[i] & 15]);
    b[k] = 0;
    *out = b;
    return k;
}

/* ---- the 32-bit word form of the product's ABI (rb_oracle.h, rbo_cig) ---- */
size_t rbo_words_of(const rbo_cig *ops, size_t n) {
    size_t w = n;
    for (size_t i = 0; i < n; i++) w += (ops[i] >> 4) >> RBO_WORD_LEN_BITS ? 1 : 0;
    return w;
}
size_t rbo_cig_to_words(const rbo_cig *ops, size_t n, uint32_t *out) {
    size_t k = 0;
    for (size_t i = 0; i < n; i++) {
        const uint64_t len = ops[i] >> 4;
        if (out) out[k] = (uint32_t)((len & ((1u << RBO_WORD_LEN_BITS) - 1u)) << 4) | (uint32_t)(ops[i] & 15);
        k++;
        if (len >> RBO_WORD_LEN_BITS) {
            if (out) out[k] = (uint32_t)((len >> RBO_WORD_LEN_BITS) << 4) | RBO_CONT;
            k++;
        }
    }
    return k;
}
int rbo_words_to_cig(const uint32_t *w, size_t n_words, rbo_cig **ops, size_t *n_ops) {
    rbo_cig *v = (rbo_cig *)xmalloc((n_words ? n_words : 1) * sizeof(rbo_cig));
    size_t cnt = 0;
    int fresh = 0; /* the op in front may still take a continuation word */
    for (size_t i = 0; i < n_words; i++) {
        if ((w[i] & 15) == RBO_CONT) {
            if (!fresh || (w[i] >> 4) > 15 || (w[i] >> 4) == 0) {
                free(v);
                return -1;
            }
            v[cnt - 1] += ((uint64_t)(w[i] >> 4) << RBO_WORD_LEN_BITS) << 4;
            fresh = 0;
        } else {
            v[cnt++] = w[i]; /* len << 4 | code as it stands */
            fresh = 1;
        }
    }
    *ops = v;
    *n_ops = cnt;
    return 0;
}
int rbo_parse_cigar(const char *s, size_t n, uint32_t **ops, size_t *n_ops) {
    rbo_cig *v = NULL;
    size_t cnt = 0;
    if (parse_cigar64(s, n, &v, &cnt)) return -1;
    const size_t nw = rbo_cig_to_words(v, cnt, NULL);
    uint32_t *w = (uint32_t *)xmalloc((nw ? nw : 1) * sizeof(uint32_t));
    rbo_cig_to_words(v, cnt, w);
    free(v);
    *ops = w;
    *n_ops = nw;
    return 0;
}
size_t rbo_cigar_to_string(const uint32_t *words, size_t n, char **out) {
    rbo_cig *v = NULL;
    size_t cnt = 0;
    if (rbo_words_to_cig(words, n, &v, &cnt)) {
        *out = xstrdup("");
        return 0;
    }
    const size_t k = cigar64_to_string(v, cnt, out);
    free(v);
    return k;
}

/* Rust u64::from_str: optional '+', then >=1 ASCII digits, no overflow. */
static int parse_u64(const char *s, size_t n, uint64_t *out) {
    size_t i = 0;
    if (n == 0) return 1;
    if (s[0] == '+') {
        i = 1;
        if (n == 1) return 1;
    }
    uint64_t v = 0;
    for (; i < n; i++) {
        if (s[i] < '0' || s[i] > '9') return 1;
        uint64_t d = (uint64_t)(s[i] - '0');
        if (v > (UINT64_MAX - d) / 10) return 1;
        v = v * 10 + d;
    }
    *out = v;
    return 0;
}

/* paf.rs:379-430 PafRecord::new.  0 ok; 1 Err(ParsePafColumn) -> caller skips the line;
 * -1 the reference panics (assert on < 12 columns, tag regex, cigar parse). */
int rbo_rec_from_line(const char *line, rbo_rec *out) {
    /* split_ascii_whitespace */
    size_t ntok = 0, cap = 32;
    const char **tok = (const char **)xmalloc(cap * sizeof(char *));
    size_t *tlen = (size_t *)xmalloc(cap * sizeof(size_t));
    const char *p = line;
    while (*p) {
        while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r' || *p == '\f') p++; /* U+000C incl.; \v is not ASCII whitespace in Rust */
        if (!*p) break;
        const char *q = p;
        while (*q && !(*q == ' ' || *q == '\t' || *q == '\n' || *q == '\r' || *q == '\f')) q++;
        if (ntok == cap) {
            cap *= 2;
            tok = (const char **)xrealloc((void *)tok, cap * sizeof(char *));
            tlen = (size_t *)xrealloc(tlen, cap * sizeof(size_t));
        }
        tok[ntok] = p;
        tlen[ntok] = (size_t)(q - p);
        ntok++;
        p = q;
    }
    int rc = 0;
    rbo_cig *cigar = NULL;
    size_t n_cigar = 0;
    if (ntok < 12) {
        rc = -1;
        goto done;
    }
    for (size_t k = 12; k < ntok; k++) {
        /* PAF_TAG = "(..):(.):(.*)" unanchored, leftmost match (paf.rs:21, :387-390) */
        const char *t = tok[k];
        size_t n = tlen[k];
        size_t m = (size_t)-1;
        for (size_t i = 0; i + 5 <= n; i++) {
            if (t[i + 2] == ':' && t[i + 4] == ':') {
                m = i;
                break;
            }
        }
        if (m == (size_t)-1) {
            rc = -1;
            goto done;
        }
        if (t[m] == 'c' && t[m + 1] == 'g' && n_cigar == 0) { /* paf.rs:395 */
            free(cigar);
            cigar = NULL;
            if (parse_cigar64(t + m + 5, n - (m + 5), &cigar, &n_cigar)) {
                rc = -1;
                goto done;
            }
        }
    }
    {
        uint64_t v[12] = {0};
        static const int numeric[] = {1, 2, 3, 6, 7, 8, 9, 10, 11};
        for (size_t k = 0; k < sizeof(numeric) / sizeof(numeric[0]); k++) {
            int c = numeric[k];
            if (parse_u64(tok[c], tlen[c], &v[c])) {
                rc = 1;
                goto done;
            }
        }
        if (tlen[4] != 1) { /* parse::<char>() */
            rc = 1;
            goto done;
        }
        rbo_rec_init(out);
        free(out->q_name);
        free(out->t_name);
        out->q_name = xstrndup(tok[0], tlen[0]);
        out->q_len = v[1];
        out->q_st = v[2];
        out->q_en = v[3];
        out->strand = tok[4][0];
        out->t_name = xstrndup(tok[5], tlen[5]);
        out->t_len = v[6];
        out->t_st = v[7];
        out->t_en = v[8];
        out->nmatch = v[9];
        out->aln_len = v[10];
        out->mapq = v[11];
        out->cigar = cigar;
        out->n_cigar = n_cigar;
        cigar = NULL;
    }
done:
    free(cigar);
    free((void *)tok);
    free(tlen);
    return rc;
}

/* paf.rs:923-944 Display */
void rbo_rec_print(const rbo_rec *r, FILE *f) {
    char *cg = NULL;
    cigar64_to_string(r->cigar, r->n_cigar, &cg);
    fprintf(f, "%s\t%llu\t%llu\t%llu\t%c\t%s\t%llu\t%llu\t%llu\t%llu\t%llu\t%llu\tid:Z:%s\tcg:Z:%s\n", r->q_name,
            (unsigned long long)r->q_len, (unsigned long long)r->q_st, (unsigned long long)r->q_en, r->strand,
            r->t_name, (unsigned long long)r->t_len, (unsigned long long)r->t_st, (unsigned long long)r->t_en,
            (unsigned long long)r->nmatch, (unsigned long long)r->aln_len, (unsigned long long)r->mapq, r->id, cg);
    free(cg);
}

/* read one line of any length from a gz/plain stream; strips \n and \r\n like BufRead::lines */
static int gz_getline(gzFile f, char **buf, size_t *cap) {
    size_t n = 0;
    for (;;) {
        if (*cap - n < 2) {
            *cap = *cap ? *cap * 2 : 65536;
            *buf = (char *)xrealloc(*buf, *cap);
        }
        if (!gzgets(f, *buf + n, (int)((*cap - n > 0x40000000u) ? 0x40000000u : (*cap - n)))) {
            if (n == 0) return 0;
            break;
        }
        n += strlen(*buf + n);
        if (n && (*buf)[n - 1] == '\n') break;
    }
    if (n && (*buf)[n - 1] == '\n') (*buf)[--n] = 0;
    if (n && (*buf)[n - 1] == '\r') (*buf)[--n] = 0;
    return 1;
}

/* paf.rs:62-78 Paf::from_file.  Negative return = reference would panic. */
int rbo_paf_from_file(const char *path, rbo_paf *out) {
    memset(out, 0, sizeof(*out));
    gzFile f = strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb");
    if (!f) return -2;
    gzbuffer(f, 1 << 20);
    char *buf = NULL;
    size_t cap = 0;
    size_t index = 0;
    int rc = 0;
    while (gz_getline(f, &buf, &cap)) {
        rbo_rec rec;
        int pr = rbo_rec_from_line(buf, &rec);
        if (pr == 0) {
            int ci = rbo_check_integrity(&rec); /* paf.rs:70 unwrap */
            if (ci != RBO_OK) {
                fprintf(stderr, "rb_oracle: record %zu fails check_integrity (status %d): reference panics\n",
                        index + 1, ci);
                rbo_rec_free(&rec);
                rc = -ci;
                break;
            }
            paf_push(out, &rec);
        } else if (pr == 1) {
            fprintf(stderr, "\nUnable to parse PAF record. Skipping line %zu\n", index + 1);
        } else {
            fprintf(stderr, "rb_oracle: line %zu makes the reference panic in PafRecord::new\n", index + 1);
            rc = -1;
            break;
        }
        index++;
    }
    free(buf);
    gzclose(f);
    return rc;
}

/* bed.rs:146-161 default id */
void rbo_region_default_id(rbo_region *r) {
    char tmp[64];
    size_t n = strlen(r->name);
    snprintf(tmp, sizeof tmp, ":%llu-%llu", (unsigned long long)(r->st + 1), (unsigned long long)r->en);
    free(r->id);
    r->id = (char *)xmalloc(n + strlen(tmp) + 1);
    memcpy(r->id, r->name, n);
    strcpy(r->id + n, tmp);
}

/* bed.rs:172-194 parse_bed through bio 1.6.0 io::bed::Reader (tab-delimited csv, no header,
 * '#' comment lines skipped; a row whose start/end does not parse is skipped with a warning).
 * Ragged / other csv corner cases: parity unpinned. */
int rbo_bed_from_file(const char *path, rbo_bed *out) {
    memset(out, 0, sizeof(*out));
    gzFile f = gzopen(path, "rb");
    if (!f) return -2;
    char *buf = NULL;
    size_t cap = 0;
    while (gz_getline(f, &buf, &cap)) {
        if (buf[0] == '#' || buf[0] == 0) continue;
        char *c1 = strchr(buf, '\t');
        if (!c1) continue;
        char *c2 = strchr(c1 + 1, '\t');
        if (!c2) continue;
        char *c3 = strchr(c2 + 1, '\t');
        size_t l3 = c3 ? (size_t)(c3 - (c2 + 1)) : strlen(c2 + 1);
        uint64_t st, en;
        if (parse_u64(c1 + 1, (size_t)(c2 - (c1 + 1)), &st) || parse_u64(c2 + 1, l3, &en)) continue;
        rbo_region r;
        r.name = xstrndup(buf, (size_t)(c1 - buf));
        r.st = st;
        r.en = en;
        r.id = NULL;
        if (c3) {
            char *c4 = strchr(c3 + 1, '\t');
            size_t l4 = c4 ? (size_t)(c4 - (c3 + 1)) : strlen(c3 + 1);
            r.id = xstrndup(c3 + 1, l4);
        } else {
            rbo_region_default_id(&r);
        }
        if (out->n == out->cap) {
            out->cap = out->cap ? out->cap * 2 : 16;
            out->r = (rbo_region *)xrealloc(out->r, out->cap * sizeof(rbo_region));
        }
        out->r[out->n++] = r;
    }
    free(buf);
    gzclose(f);
    return 0;
}

/* ------------------------------------------------------------------ integrity */
/* paf.rs:631-654.  The reference sums in u32 (opt.len() is u32); a sum that does not fit
 * is a debug-build panic / release-build wrap: reported as RBO_PANIC_OVERFLOW. */
int rbo_infer_n_bases(const rbo_rec *r, uint64_t o[4]) {
    uint64_t t = 0, q = 0, m = 0, a = 0;
    for (size_t i = 0; i < r->n_cigar; i++) {
        const rbo_cig op = r->cigar[i];
        const uint64_t len = op >> 4;
        if (rbo_consumes_reference(op)) t += len;
        if (rbo_consumes_query(op)) q += len;
        if (rbo_is_match(op)) m += len;
        a += len;
    }
    o[0] = t;
    o[1] = q;
    o[2] = m;
    o[3] = a;
    return a > 0xFFFFFFFFull ? RBO_PANIC_OVERFLOW : RBO_OK;
}

/* paf.rs:825-857 */
int rbo_check_integrity(rbo_rec *r) {
    uint64_t o[4];
    int rc = rbo_infer_n_bases(r, o);
    if (rc) return rc;
    if (r->t_en < r->t_st || r->t_en - r->t_st != o[0]) return RBO_PANIC_INTEGRITY_T;
    if (r->q_en < r->q_st || r->q_en - r->q_st != o[1]) return RBO_PANIC_INTEGRITY_Q;
    r->nmatch = o[2];
    r->aln_len = o[3];
    return RBO_OK;
}

static void id_append_TO(rbo_rec *r, const rbo_cig *st, size_t nst, const rbo_cig *en, size_t nen) {
    char *a = NULL, *b = NULL;
    cigar64_to_string(st, nst, &a);
    cigar64_to_string(en, nen, &b);
    size_t n = strlen(r->id) + strlen(a) + strlen(b) + 6;
    char *s = (char *)xmalloc(n);
    snprintf(s, n, "%s_TO.%s.%s", r->id, a, b);
    free(r->id);
    r->id = s;
    free(a);
    free(b);
}

/* paf.rs:656-783 */
int rbo_remove_trailing_indels(rbo_rec *r) {
    size_t cigar_len = r->n_cigar;
    if (cigar_len == 0) return RBO_PANIC_EMPTY_CIGAR; /* :663 */
    rbo_cig st_opt = r->cigar[0];
    int64_t remove_st_t = 0, remove_st_q = 0;
    size_t remove_st_opts = 0;
    rbo_cig *removed_st = (rbo_cig *)xmalloc(cigar_len * sizeof(rbo_cig));
    while (is_indel(st_opt)) { /* :668-687 */
        if ((st_opt & 15) == RBO_D) {
            remove_st_t += (int64_t)(st_opt >> 4);
            remove_st_q += 1;
        } else {
            remove_st_q += (int64_t)(st_opt >> 4);
        }
        removed_st[remove_st_opts++] = st_opt;
        if (remove_st_opts < cigar_len)
            st_opt = r->cigar[remove_st_opts];
        else
            break;
    }
    if (remove_st_opts > 1) { /* :690-701 */
        for (size_t i = 0; i + 1 < remove_st_opts; i++) {
            const unsigned pre = (unsigned)(removed_st[i] & 15), cur = (unsigned)(removed_st[i + 1] & 15);
            if ((pre == RBO_D && cur == RBO_I) || (pre == RBO_I && cur == RBO_D)) {
                remove_st_t += 1;
                remove_st_q -= 1;
            }
        }
    }
    rbo_cig en_opt = r->cigar[cigar_len - 1]; /* :704-723 */
    int64_t remove_en_t = 0, remove_en_q = 0;
    size_t remove_en_opts = 0;
    rbo_cig *removed_en = (rbo_cig *)xmalloc(cigar_len * sizeof(rbo_cig));
    while (is_indel(en_opt)) {
        if ((en_opt & 15) == RBO_D)
            remove_en_t += (int64_t)(en_opt >> 4);
        else
            remove_en_q += (int64_t)(en_opt >> 4);
        removed_en[remove_en_opts++] = en_opt;
        if (cigar_len - remove_en_opts > 0)
            en_opt = r->cigar[cigar_len - 1 - remove_en_opts];
        else
            break;
    }
    if (remove_en_opts > 0 || remove_st_opts > 0) /* :726-732 */
        id_append_TO(r, removed_st, remove_st_opts, removed_en, remove_en_opts);
    free(removed_st);
    free(removed_en);
    /* :756-757 */
    if (remove_st_opts + remove_en_opts > cigar_len) return RBO_PANIC_ALL_INDEL;
    size_t new_len = cigar_len - remove_st_opts - remove_en_opts;
    memmove(r->cigar, r->cigar + remove_st_opts, (cigar_len - remove_st_opts) * sizeof(rbo_cig));
    r->n_cigar = new_len;
    /* :760-769 */
    r->t_st += (uint64_t)remove_st_t;
    r->t_en -= (uint64_t)remove_en_t;
    if (r->strand == '-') {
        int64_t t = remove_st_q;
        remove_st_q = remove_en_q;
        remove_en_q = t;
    }
    r->q_st += (uint64_t)remove_st_q;
    r->q_en -= (uint64_t)remove_en_q;
    /* :782 check_integrity().unwrap() */
    return rbo_check_integrity(r);
}

/* paf.rs:501-538 */
int rbo_aligned_pairs(rbo_rec *r) {
    int rc = rbo_remove_trailing_indels(r);
    if (rc) return rc;
    rbo_rec_drop_aln(r);
    size_t total = 0;
    for (size_t i = 0; i < r->n_cigar; i++) total += r->cigar[i] >> 4;
    r->tpos_aln = (uint64_t *)xmalloc(total * sizeof(uint64_t));
    r->qpos_aln = (uint64_t *)xmalloc(total * sizeof(uint64_t));
    r->long_cigar = (uint8_t *)xmalloc(total);
    r->n_aln = total;
    int64_t t_pos = (int64_t)r->t_st - 1;
    int64_t q_pos = (int64_t)r->q_st - 1;
    if (r->strand == '-') q_pos = (int64_t)r->q_en;
    size_t k = 0;
    for (size_t i = 0; i < r->n_cigar; i++) {
        const rbo_cig op = r->cigar[i];
        int moves_t = rbo_consumes_reference(op), moves_q = rbo_consumes_query(op);
        const uint64_t len = op >> 4;
        for (uint64_t j = 0; j < len; j++) {
            r->long_cigar[k] = (uint8_t)(op & 15);
            if (moves_t) t_pos += 1;
            if (moves_q && r->strand == '+') q_pos += 1;
            if (moves_q && r->strand == '-') q_pos -= 1;
            r->tpos_aln[k] = (uint64_t)t_pos;
            r->qpos_aln[k] = (uint64_t)q_pos;
            k++;
        }
    }
    return RBO_OK;
}

/* paf.rs:489-498 */
static void make_long_cigar(rbo_rec *r) {
    size_t total = 0;
    for (size_t i = 0; i < r->n_cigar; i++) total += r->cigar[i] >> 4;
    free(r->long_cigar);
    r->long_cigar = (uint8_t *)xmalloc(total);
    size_t k = 0;
    for (size_t i = 0; i < r->n_cigar; i++)
        for (uint64_t j = 0; j < (r->cigar[i] >> 4); j++) r->long_cigar[k++] = (uint8_t)(r->cigar[i] & 15);
    /* NB the reference does not touch tpos_aln/qpos_aln here; long_cigar.len() may differ from
     * n_aln only if the caller let them go stale; we keep n_aln tied to the position arrays. */
}

/* ------------------------------------------------------------------ Rust slice::binary_search
 * cmp(i) returns <0 / 0 / >0 for Less / Equal / Greater of f(&a[i]).
 * MODERN: core::slice::binary_search_by as of rustc 1.82 (branchless halving).
 * LEGACY: rustc 1.52 .. 1.81 (early return on Equal). */
typedef int (*probe_fn)(const void *ctx, size_t i);
static int rust_bsearch(size_t len, probe_fn f, const void *ctx, int policy, size_t *idx) {
    if (policy == RBO_BSEARCH_MODERN) {
        size_t size = len;
        if (size == 0) return 1;
        size_t base = 0;
        while (size > 1) {
            size_t half = size / 2, mid = base + half;
            int c = f(ctx, mid);
            base = (c > 0) ? base : mid;
            size -= half;
        }
        int c = f(ctx, base);
        if (c == 0) {
            *idx = base;
            return 0;
        }
        *idx = base + (c < 0 ? 1 : 0);
        return 1;
    } else {
        size_t size = len, left = 0, right = len;
        while (left < right) {
            size_t mid = left + size / 2;
            int c = f(ctx, mid);
            if (c < 0)
                left = mid + 1;
            else if (c > 0)
                right = mid;
            else {
                *idx = mid;
                return 0;
            }
            size = right - left;
        }
        *idx = left;
        return 1;
    }
}
typedef struct {
    const uint64_t *a;
    uint64_t key;
    int reverse;
} probe_ctx;
static int probe_u64(const void *c, size_t i) {
    const probe_ctx *p = (const probe_ctx *)c;
    uint64_t v = p->a[i];
    int r = v < p->key ? -1 : (v > p->key ? 1 : 0);
    return p->reverse ? -r : r;
}

/* paf.rs:541-544 */
int rbo_tpos_to_idx(const rbo_rec *r, uint64_t tpos, int policy, size_t *idx) {
    probe_ctx c = {r->tpos_aln, tpos, 0};
    return rust_bsearch(r->n_aln, probe_u64, &c, policy, idx);
}
/* paf.rs:547-561 */
int rbo_tpos_to_idx_match(const rbo_rec *r, uint64_t tpos, int right, int policy, size_t *out) {
    size_t idx;
    if (rbo_tpos_to_idx(r, tpos, policy, &idx)) {
        *out = idx;
        return 1;
    }
    size_t max_idx = r->n_aln;
    if (right) {
        while (idx < max_idx && !rbo_is_match(r->long_cigar[idx])) idx += 1;
    } else {
        while (idx > 0 && !rbo_is_match(r->long_cigar[idx])) idx -= 1;
    }
    *out = idx;
    return 0;
}
/* paf.rs:564-573 */
int rbo_qpos_to_idx(const rbo_rec *r, uint64_t qpos, int policy, size_t *idx) {
    probe_ctx c = {r->qpos_aln, qpos, r->strand == '-'};
    return rust_bsearch(r->n_aln, probe_u64, &c, policy, idx);
}
/* paf.rs:576-590 */
int rbo_qpos_to_idx_match(const rbo_rec *r, uint64_t qpos, int right, int policy, size_t *out) {
    size_t idx;
    if (rbo_qpos_to_idx(r, qpos, policy, &idx)) {
        *out = idx;
        return 1;
    }
    size_t max_idx = r->n_aln;
    if ((right && r->strand == '+') || (!right && r->strand == '-')) {
        while (idx < max_idx && !rbo_is_match(r->long_cigar[idx])) idx += 1;
    } else {
        while (idx > 0 && !rbo_is_match(r->long_cigar[idx])) idx -= 1;
    }
    *out = idx;
    return 0;
}

/* paf.rs:984-998 update_cigar_opt_len: the same op with a new length (every arm of the match keeps the variant) */
rbo_cig rbo_update_cigar_opt_len(rbo_cig op, uint32_t new_opt_len) { return ((rbo_cig)new_opt_len << 4) | (op & 15u); }

/* paf.rs:593-620 subset_cigar + collapse_long_cigar */
static void subset_collapse(const uint8_t *lc, size_t a, size_t b, rbo_cig **ops, size_t *n_ops) {
    size_t cap = 16, cnt = 0;
    rbo_cig *v = (rbo_cig *)xmalloc(cap * sizeof(rbo_cig));
    uint8_t pre = lc[a];
    uint32_t pre_len = 1; /* (u32 in the reference: paf.rs:606) */
    for (size_t i = a + 1; i <= b; i++) {
        if (lc[i] == pre) {
            pre_len++;
        } else {
            if (cnt == cap) {
                cap *= 2;
                v = (rbo_cig *)xrealloc(v, cap * sizeof(rbo_cig));
            }
            v[cnt++] = rbo_update_cigar_opt_len(pre, pre_len); /* paf.rs:612 */
            pre = lc[i];
            pre_len = 1;
        }
    }
    if (cnt == cap) {
        cap += 1;
        v = (rbo_cig *)xrealloc(v, cap * sizeof(rbo_cig));
    }
    v[cnt++] = rbo_update_cigar_opt_len(pre, pre_len); /* paf.rs:618 */
    *ops = v;
    *n_ops = cnt;
}

/* bed.rs:66-71 has_overlap; bed.rs:74-85 get_overlap; bed.rs:215-235 split_region: the three region helpers the path uses, exposed so
 * that the reference's doctest vectors for them (bed.rs:51-64, :199-214) can be held against this restatement */
int rbo_has_overlap(const char *name1, uint64_t st1, uint64_t en1, const char *name2, uint64_t st2, uint64_t en2) {
    if (strcmp(name1, name2) != 0) return 0;
    return en1 > st2 && st1 < en2;
}
uint64_t rbo_get_overlap(const char *name1, uint64_t st1, uint64_t en1, const char *name2, uint64_t st2, uint64_t en2) {
    if (strcmp(name1, name2) != 0) return 0;
    const uint64_t mn = en1 < en2 ? en1 : en2, mx = st1 > st2 ? st1 : st2;
    if (mn < mx) return 0;
    return mn - mx;
}
/* the k-th piece of split_region(rgn, window); returns 0 when there is no such piece */
int rbo_split_region(uint64_t st, uint64_t en, uint64_t window, uint64_t k, uint64_t *pst, uint64_t *pen) {
    uint64_t start = st;
    for (uint64_t i = 0; start < en; i++) {
        uint64_t end = start + window;
        if (end > en) end = en;
        if (i == k) {
            *pst = start, *pen = end;
            return 1;
        }
        start = end;
    }
    return 0;
}

/* paf.rs:622-627 */
static int paf_overlaps_rgn(const rbo_rec *p, const rbo_region *g) { return rbo_has_overlap(p->t_name, p->t_st, p->t_en, g->name, g->st, g->en); }

/* ------------------------------------------------------------------ liftover.rs:17-105
 * returns RBO_OK with *out filled, RBO_NONE_* (reference returns None), or RBO_PANIC_*. */
int rbo_trim_paf_rec_to_rgn(const rbo_region *rgn, const rbo_rec *paf, int policy, rbo_rec *out) {
    memset(out, 0, sizeof(*out));
    if (paf->t_st > rgn->st && paf->t_en < rgn->en) { /* :23-25 paf.clone(), own id */
        rec_small_copy(out, paf);
        out->cigar = (rbo_cig *)memdup(paf->cigar, paf->n_cigar * sizeof(rbo_cig));
        out->n_cigar = paf->n_cigar;
        return RBO_OK;
    }
    rbo_rec tr;
    rec_small_copy(&tr, paf);
    free(tr.id);
    tr.id = xstrdup(rgn->id);
    int rc = RBO_OK;
    size_t start_idx, end_idx;
    tr.t_st = rgn->st > paf->t_st ? rgn->st : paf->t_st; /* :28 */
    if (rbo_tpos_to_idx_match(paf, tr.t_st, 1, policy, &start_idx)) {
        rc = RBO_PANIC_NOTFOUND;
        goto fail;
    }
    tr.t_en = rgn->en < paf->t_en ? rgn->en : paf->t_en; /* :38 */
    if (rbo_tpos_to_idx_match(paf, tr.t_en - 1, 0, policy, &end_idx)) {
        rc = RBO_PANIC_NOTFOUND;
        goto fail;
    }
    if (start_idx > end_idx) { /* :52-54 */
        rc = RBO_NONE_INDEL;
        goto fail;
    }
    tr.t_st = paf->tpos_aln[start_idx]; /* :57-60 */
    tr.q_st = paf->qpos_aln[start_idx];
    tr.t_en = paf->tpos_aln[end_idx];
    tr.q_en = paf->qpos_aln[end_idx];
    subset_collapse(paf->long_cigar, start_idx, end_idx, &tr.cigar, &tr.n_cigar); /* :63 */
    {
        int no_match = 1; /* :66-75 */
        for (size_t i = 0; i < tr.n_cigar; i++)
            if (rbo_is_match(tr.cigar[i])) {
                no_match = 0;
                break;
            }
        if (no_match) {
            rc = RBO_NONE_NOMATCH;
            goto fail;
        }
    }
    if (paf->strand == '-') { /* :77-79 */
        uint64_t t = tr.q_en;
        tr.q_en = tr.q_st;
        tr.q_st = t;
    }
    tr.t_en += 1; /* :81-82 */
    tr.q_en += 1;
    rc = rbo_remove_trailing_indels(&tr); /* :85 (unwrap inside => panic statuses) */
    if (rc) goto fail;
    if (tr.n_cigar == 0) { /* :87-89 */
        rc = RBO_NONE_EMPTY;
        goto fail;
    }
    if (tr.q_st > tr.q_en || tr.t_st > tr.t_en) { /* :90-96 */
        rc = RBO_NONE_INVERTED;
        goto fail;
    }
    if (rbo_check_integrity(&tr) != RBO_OK) { /* :99-102 */
        rc = RBO_NONE_INTEGRITY;
        goto fail;
    }
    *out = tr;
    return RBO_OK;
fail:
    rbo_rec_free(&tr);
    return rc;
}

/* ------------------------------------------------------------------ paf.rs:1050-1094 */
void rbo_paf_swap_query_and_target(const rbo_rec *paf, rbo_rec *fl) {
    rbo_rec_clone(fl, paf);
    free(fl->t_name);
    free(fl->q_name);
    fl->t_name = xstrdup(paf->q_name);
    fl->t_len = paf->q_len;
    fl->t_st = paf->q_st;
    fl->t_en = paf->q_en;
    fl->q_name = xstrdup(paf->t_name);
    fl->q_len = paf->t_len;
    fl->q_st = paf->t_st;
    fl->q_en = paf->t_en;
    for (size_t i = 0; i < fl->n_cigar; i++) {
        rbo_cig op = fl->cigar[i] & 15;
        const rbo_cig len = fl->cigar[i] >> 4;
        if (op == RBO_I)
            op = RBO_D;
        else if (op == RBO_D)
            op = RBO_I;
        fl->cigar[i] = (len << 4) | op;
    }
    if (paf->strand == '-')
        for (size_t i = 0, j = fl->n_cigar; i + 1 < j; i++, j--) {
            rbo_cig t = fl->cigar[i];
            fl->cigar[i] = fl->cigar[j - 1];
            fl->cigar[j - 1] = t;
        }
    if (paf->n_aln) { /* :1089-1091 */
        rbo_rec_drop_aln(fl);
        (void)rbo_aligned_pairs(fl);
    } else {
        rbo_rec_drop_aln(fl);
    }
}

/* ------------------------------------------------------------------ liftover.rs:107-167
 * Single-thread order: contigs by first appearance, then record order, then region order. */
int rbo_trim_paf_by_rgns(const rbo_bed *rgns, const rbo_paf *paf_in, int invert_query, int policy, rbo_paf *out) {
    memset(out, 0, sizeof(*out));
    rbo_paf swapped = {0};
    const rbo_paf *paf = paf_in;
    if (invert_query) {
        for (size_t i = 0; i < paf_in->n; i++) {
            rbo_rec f;
            rbo_paf_swap_query_and_target(&paf_in->recs[i], &f);
            paf_push(&swapped, &f);
        }
        paf = &swapped;
    }
    /* unique names in first-appearance order (:151) */
    size_t n_names = 0;
    const char **names = (const char **)xmalloc((paf->n + 1) * sizeof(char *));
    for (size_t i = 0; i < paf->n; i++) {
        size_t k;
        for (k = 0; k < n_names; k++)
            if (strcmp(names[k], paf->recs[i].t_name) == 0) break;
        if (k == n_names) names[n_names++] = paf->recs[i].t_name;
    }
    int rc = 0;
    for (size_t c = 0; c < n_names && rc == 0; c++) {
        for (size_t i = 0; i < paf->n && rc == 0; i++) {
            if (strcmp(paf->recs[i].t_name, names[c]) != 0) continue;
            rbo_rec cur;
            rbo_rec_clone(&cur, &paf->recs[i]);
            int ap = rbo_aligned_pairs(&cur); /* :119-121 */
            if (ap) {
                fprintf(stderr, "rb_oracle: record %zu: aligned_pairs status %d (reference panics)\n", i + 1, ap);
                rbo_rec_free(&cur);
                rc = -ap;
                break;
            }
            for (size_t g = 0; g < rgns->n; g++) {
                if (strcmp(rgns->r[g].name, names[c]) != 0) continue;
                if (!paf_overlaps_rgn(&cur, &rgns->r[g])) continue;
                rbo_rec t;
                int st = rbo_trim_paf_rec_to_rgn(&rgns->r[g], &cur, policy, &t);
                if (st == RBO_OK) {
                    paf_push(out, &t);
                } else if (st >= RBO_PANIC_NOTFOUND) {
                    fprintf(stderr, "rb_oracle: record %zu region %zu: status %d (reference panics)\n", i + 1,
                            g + 1, st);
                    rc = -st;
                    break;
                }
            }
            rbo_rec_free(&cur);
        }
    }
    free((void *)names);
    rbo_paf_free(&swapped);
    return rc;
}

/* ------------------------------------------------------------------ liftover.rs:182-226
 * `paf` must already have been through aligned_pairs (main.rs:275).  If windows/rows are
 * non-NULL every candidate window is reported with its status (for the flat API). */
typedef struct {
    uint64_t st, en;
    int status;
} piece_t;
static int break_impl(const rbo_rec *paf, uint32_t break_length, int policy, rbo_paf *out, piece_t **pieces,
                      size_t *n_pieces) {
    size_t pcap = 8, pn = 0;
    piece_t *pv = pieces ? (piece_t *)xmalloc(pcap * sizeof(piece_t)) : NULL;
    uint64_t cur_tpos = paf->t_st, pre_tpos = paf->t_st;
    int rc = 0;
    for (size_t i = 0; i <= paf->n_cigar; i++) {
        int last = (i == paf->n_cigar);
        const rbo_cig opt = last ? 0 : paf->cigar[i];
        const uint64_t opt_len = opt >> 4;
        int big = !last && opt_len > break_length && is_indel(opt);
        if (big || last) {
            if (cur_tpos > pre_tpos) {
                rbo_region rgn = {paf->t_name, pre_tpos, cur_tpos, paf->id};
                rbo_rec x;
                int st = rbo_trim_paf_rec_to_rgn(&rgn, paf, policy, &x);
                if (pv) {
                    if (pn == pcap) {
                        pcap *= 2;
                        pv = (piece_t *)xrealloc(pv, pcap * sizeof(piece_t));
                    }
                    pv[pn].st = pre_tpos;
                    pv[pn].en = cur_tpos;
                    pv[pn].status = st;
                    pn++;
                }
                if (st == RBO_OK) {
                    paf_push(out, &x);
                } else if (st >= RBO_PANIC_NOTFOUND && rc == 0) {
                    rc = -st;
                }
            }
            if (last) break;
            pre_tpos = cur_tpos;
            if (rbo_consumes_reference(opt)) pre_tpos += opt_len;
        }
        if (rbo_consumes_reference(opt)) cur_tpos += opt_len;
    }
    if (pieces) {
        *pieces = pv;
        *n_pieces = pn;
    }
    return rc;
}
int rbo_break_paf_on_indels(const rbo_rec *paf, uint32_t break_length, int policy, rbo_paf *out) {
    return break_impl(paf, break_length, policy, out, NULL, NULL);
}

/* ------------------------------------------------------------------ paf.rs:785-823 */
int rbo_truncate_record_by_query(rbo_rec *r, uint64_t new_q_st, uint64_t new_q_en, int policy) {
    if (!(new_q_st >= r->q_st)) return RBO_PANIC_ASSERT;
    if (!(new_q_en <= r->q_en)) return RBO_PANIC_ASSERT;
    if (new_q_en == 0) return RBO_PANIC_ASSERT; /* new_q_en - 1 underflows */
    make_long_cigar(r);
    size_t aln_st, aln_en;
    if (rbo_qpos_to_idx_match(r, new_q_st, 1, policy, &aln_st)) return RBO_PANIC_NOTFOUND;
    if (rbo_qpos_to_idx_match(r, new_q_en - 1, 0, policy, &aln_en)) return RBO_PANIC_NOTFOUND;
    /* the walk to a match may run off the end (idx == len): self.qpos_aln[idx] then panics (paf.rs:795-796) */
    if (aln_st >= r->n_aln || aln_en >= r->n_aln) return RBO_PANIC_NOTFOUND;
    uint64_t nn_q_st = r->qpos_aln[aln_st];
    uint64_t nn_q_en = r->qpos_aln[aln_en] + 1;
    if (aln_st > aln_en) {
        size_t t = aln_st;
        aln_st = aln_en;
        aln_en = t;
    }
    uint64_t new_t_st = r->tpos_aln[aln_st];
    uint64_t new_t_en = r->tpos_aln[aln_en] + 1;
    rbo_cig *ops;
    size_t n_ops;
    subset_collapse(r->long_cigar, aln_st, aln_en, &ops, &n_ops);
    free(r->cigar);
    r->cigar = ops;
    r->n_cigar = n_ops;
    r->t_st = new_t_st;
    r->t_en = new_t_en;
    r->q_st = nn_q_st;
    r->q_en = nn_q_en;
    int rc = rbo_remove_trailing_indels(r);
    if (rc) return rc;
    return rbo_check_integrity(r);
}

/* trim_overlap.rs:6-19 */
static int score_of_qpos(const rbo_rec *rec, uint64_t pos, int ms, int ds, int is, int policy, int *score) {
    size_t idx;
    if (rbo_qpos_to_idx(rec, pos, policy, &idx)) return 1;
    uint8_t op = rec->long_cigar[idx];
    if (op == RBO_EQ)
        *score = ms;
    else if (op == RBO_I || op == RBO_D)
        *score = -is;
    else
        *score = -ds;
    return 0;
}

/* trim_overlap.rs:36-86 */
int rbo_trim_overlapping_pafs(rbo_rec *left, rbo_rec *right, int ms, int ds, int is, int policy,
                              uint64_t *split_idx, int *split_score) {
    uint64_t st_ovl = left->q_st > right->q_st ? left->q_st : right->q_st;
    uint64_t en_ovl = left->q_en < right->q_en ? left->q_en : right->q_en;
    size_t n = en_ovl > st_ovl ? (size_t)(en_ovl - st_ovl) : 0;
    int32_t *l_score = (int32_t *)xmalloc((n + 1) * sizeof(int32_t));
    int32_t *r_score = (int32_t *)xmalloc((n + 1) * sizeof(int32_t));
    l_score[0] = 0;
    for (size_t k = 0; k < n; k++) {
        int ls, rs;
        if (score_of_qpos(left, st_ovl + k, ms, ds, is, policy, &ls) ||
            score_of_qpos(right, st_ovl + k, ms, ds, is, policy, &rs)) {
            free(l_score);
            free(r_score);
            return RBO_PANIC_NOTFOUND; /* .unwrap() at trim_overlap.rs:13 */
        }
        l_score[k + 1] = ls;
        r_score[k] = rs;
    }
    r_score[n] = 0;
    int32_t acc = 0;
    for (size_t k = 0; k <= n; k++) {
        l_score[k] += acc;
        acc = l_score[k];
    }
    acc = 0;
    for (size_t k = n + 1; k-- > 0;) {
        r_score[k] += acc;
        acc = r_score[k];
    }
    uint64_t max_idx = 0;
    int32_t max = 0;
    for (size_t k = 0; k <= n; k++) {
        if (l_score[k] + r_score[k] > max) {
            max = l_score[k] + r_score[k];
            max_idx = k;
        }
    }
    free(l_score);
    free(r_score);
    if (split_idx) *split_idx = max_idx;
    if (split_score) *split_score = max;
    int rc = rbo_truncate_record_by_query(left, left->q_st, st_ovl + max_idx, policy);
    if (rc) return rc;
    return rbo_truncate_record_by_query(right, st_ovl + max_idx, right->q_en, policy);
}

/* stable merge sort of indices by key */
typedef int (*idx_cmp)(const void *ctx, size_t a, size_t b);
static void msort(size_t *v, size_t *tmp, size_t n, idx_cmp cmp, const void *ctx) {
    if (n < 2) return;
    size_t h = n / 2;
    msort(v, tmp, h, cmp, ctx);
    msort(v + h, tmp, n - h, cmp, ctx);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (cmp(ctx, v[j], v[i]) < 0) ? v[j++] : v[i++];
    while (i < h) tmp[k++] = v[i++];
    while (j < n) tmp[k++] = v[j++];
    memcpy(v, tmp, n * sizeof(size_t));
}
static int cmp_qname(const void *ctx, size_t a, size_t b) {
    const rbo_rec *r = (const rbo_rec *)ctx;
    return strcmp(r[a].q_name, r[b].q_name);
}
typedef struct {
    uint64_t overlap;
    size_t i, j;
} ovl_pair;
static int cmp_pair(const void *ctx, size_t a, size_t b) {
    const ovl_pair *p = (const ovl_pair *)ctx;
    uint64_t ka = UINT64_MAX - p[a].overlap, kb = UINT64_MAX - p[b].overlap;
    return ka < kb ? -1 : (ka > kb ? 1 : 0);
}

/* bed.rs:74-85 with paf.rs:459-466 */
static uint64_t query_overlap(const rbo_rec *a, const rbo_rec *b) { return rbo_get_overlap(a->q_name, a->q_st, a->q_en, b->q_name, b->q_st, b->q_en); }

/* paf.rs:210-305 (the recursion is a loop here) */
int rbo_overlapping_paf_recs(rbo_paf *paf, int ms, int ds, int is, int remove_contained, int policy) {
    for (int depth = 0; depth < 100000; depth++) {
        for (size_t i = 0; i < paf->n; i++) { /* :218-220 */
            int rc = rbo_remove_trailing_indels(&paf->recs[i]);
            if (rc) return -rc;
        }
        size_t n = paf->n;
        { /* :223 stable sort by q_name */
            size_t *ix = (size_t *)xmalloc(n * sizeof(size_t)), *tmp = (size_t *)xmalloc(n * sizeof(size_t));
            for (size_t i = 0; i < n; i++) ix[i] = i;
            msort(ix, tmp, n, cmp_qname, paf->recs);
            rbo_rec *nr = (rbo_rec *)xmalloc((paf->cap ? paf->cap : 1) * sizeof(rbo_rec));
            for (size_t i = 0; i < n; i++) nr[i] = paf->recs[ix[i]];
            free(paf->recs);
            paf->recs = nr;
            free(ix);
            free(tmp);
        }
        uint8_t *contained = (uint8_t *)xmalloc(n + 1);
        memset(contained, 0, n + 1);
        if (n < 2) { /* :227-229 */
            free(contained);
            return 0;
        }
        size_t pcap = 64, pn = 0;
        ovl_pair *pairs = (ovl_pair *)xmalloc(pcap * sizeof(ovl_pair));
        for (size_t i = 0; i + 1 < n; i++) { /* :231-261 */
            const rbo_rec *r1 = &paf->recs[i];
            for (size_t j = i + 1; j < n && strcmp(r1->q_name, paf->recs[j].q_name) == 0; j++) {
                const rbo_rec *r2 = &paf->recs[j];
                uint64_t ov = query_overlap(r1, r2);
                if (ov < 1) continue;
                if (ov == r2->q_en - r2->q_st) {
                    contained[j] = 1;
                } else if (ov == r1->q_en - r1->q_st) {
                    contained[i] = 1;
                } else {
                    if (pn == pcap) {
                        pcap *= 2;
                        pairs = (ovl_pair *)xrealloc(pairs, pcap * sizeof(ovl_pair));
                    }
                    pairs[pn].overlap = ov;
                    if (r1->q_st <= r2->q_st) {
                        pairs[pn].i = i;
                        pairs[pn].j = j;
                    } else {
                        pairs[pn].i = j;
                        pairs[pn].j = i;
                    }
                    pn++;
                }
            }
        }
        size_t *order = (size_t *)xmalloc((pn + 1) * sizeof(size_t)), *tmp = (size_t *)xmalloc((pn + 1) * sizeof(size_t));
        for (size_t k = 0; k < pn; k++) order[k] = k;
        msort(order, tmp, pn, cmp_pair, pairs); /* :262 */
        free(tmp);
        /* q_seen: records are sorted by q_name, so "seen" is a per-name flag on the first record index */
        size_t unseen = 0;
        char **seen = (char **)xmalloc((pn + 1) * sizeof(char *));
        size_t n_seen = 0;
        int rc = 0;
        for (size_t k = 0; k < pn && rc == 0; k++) { /* :266-284 */
            size_t i = pairs[order[k]].i, j = pairs[order[k]].j;
            const char *qn = paf->recs[i].q_name;
            int was_seen = 0;
            for (size_t s = 0; s < n_seen; s++)
                if (strcmp(seen[s], qn) == 0) {
                    was_seen = 1;
                    break;
                }
            if (was_seen) {
                unseen++;
                continue;
            }
            rbo_rec left, right;
            rbo_rec_clone(&left, &paf->recs[i]);
            rbo_rec_clone(&right, &paf->recs[j]);
            rc = rbo_aligned_pairs(&left);
            if (!rc) rc = rbo_aligned_pairs(&right);
            if (!rc) rc = rbo_trim_overlapping_pafs(&left, &right, ms, ds, is, policy, NULL, NULL);
            if (rc) {
                rbo_rec_free(&left);
                rbo_rec_free(&right);
                rc = -rc;
                break;
            }
            rbo_rec_drop_aln(&left);
            rbo_rec_drop_aln(&right);
            seen[n_seen++] = xstrdup(qn);
            rbo_rec_free(&paf->recs[i]);
            rbo_rec_free(&paf->recs[j]);
            paf->recs[i] = left;
            paf->recs[j] = right;
        }
        for (size_t s = 0; s < n_seen; s++) free(seen[s]);
        free(seen);
        free(order);
        free(pairs);
        if (rc) {
            free(contained);
            return rc;
        }
        if (unseen > 0) { /* :286-288 recurse */
            free(contained);
            continue;
        }
        if (remove_contained) { /* :289-301 */
            size_t w = 0;
            for (size_t i = 0; i < n; i++) {
                if (contained[i])
                    rbo_rec_free(&paf->recs[i]);
                else
                    paf->recs[w++] = paf->recs[i];
            }
            paf->n = w;
        }
        free(contained);
        return 0;
    }
    return -RBO_PANIC_ASSERT;
}

/* ------------------------------------------------------------------ header-only commands (paf.rs:91-207) */
typedef struct {
    const char *t, *q;
    int64_t orient;
    uint64_t total_bp, order, aln_bp;
} tq_entry;
static tq_entry *tq_find(tq_entry **tab, size_t *n, size_t *cap, const char *t, const char *q) {
    for (size_t i = 0; i < *n; i++)
        if (strcmp((*tab)[i].t, t) == 0 && strcmp((*tab)[i].q, q) == 0) return &(*tab)[i];
    if (*n == *cap) {
        *cap = *cap ? *cap * 2 : 64;
        *tab = (tq_entry *)xrealloc(*tab, *cap * sizeof(tq_entry));
    }
    tq_entry *e = &(*tab)[(*n)++];
    memset(e, 0, sizeof(*e));
    e->t = t;
    e->q = q;
    return e;
}
/* paf.rs:91-111 in the order main.rs:242-244 applies them */
void rbo_paf_filter(rbo_paf *paf, uint64_t paired_len, uint64_t min_aln, uint64_t min_query) {
    size_t w = 0;
    for (size_t i = 0; i < paf->n; i++) { /* filter_query_len */
        if (paf->recs[i].q_len > min_query) paf->recs[w++] = paf->recs[i];
        else rbo_rec_free(&paf->recs[i]);
    }
    paf->n = w;
    w = 0;
    for (size_t i = 0; i < paf->n; i++) { /* filter_aln_len */
        if (paf->recs[i].t_en - paf->recs[i].t_st > min_aln) paf->recs[w++] = paf->recs[i];
        else rbo_rec_free(&paf->recs[i]);
    }
    paf->n = w;
    tq_entry *tab = NULL;
    size_t n = 0, cap = 0;
    for (size_t i = 0; i < paf->n; i++) /* filter_aln_pairs */
        tq_find(&tab, &n, &cap, paf->recs[i].t_name, paf->recs[i].q_name)->aln_bp += paf->recs[i].t_en - paf->recs[i].t_st;
    uint8_t *keep = (uint8_t *)xmalloc(paf->n + 1);
    for (size_t i = 0; i < paf->n; i++) keep[i] = paired_len < tq_find(&tab, &n, &cap, paf->recs[i].t_name, paf->recs[i].q_name)->aln_bp;
    free(tab);
    w = 0;
    for (size_t i = 0; i < paf->n; i++) {
        if (keep[i]) paf->recs[w++] = paf->recs[i];
        else rbo_rec_free(&paf->recs[i]);
    }
    paf->n = w;
    free(keep);
}
/* paf.rs:114-157; `order` of every record is returned through orders[] for scaffold */
int rbo_paf_orient(rbo_paf *paf, uint64_t *orders) {
    tq_entry *tab = NULL;
    size_t n = 0, cap = 0;
    for (size_t i = 0; i < paf->n; i++) {
        rbo_rec *r = &paf->recs[i];
        tq_entry *e = tq_find(&tab, &n, &cap, r->t_name, r->q_name);
        if (r->strand == '-') e->orient -= (int64_t)(r->q_en - r->q_st); else e->orient += (int64_t)(r->q_en - r->q_st);
        uint64_t weight = r->t_en - r->t_st;
        e->total_bp += weight;
        e->order += weight * (r->t_st + r->t_en) / 2;
    }
    /* the table keys point into the records' own strings, which the second loop replaces: resolve all lookups first */
    tq_entry **ent = (tq_entry **)xmalloc((paf->n + 1) * sizeof(tq_entry *));
    for (size_t i = 0; i < paf->n; i++) ent[i] = tq_find(&tab, &n, &cap, paf->recs[i].t_name, paf->recs[i].q_name);
    int rc = 0;
    for (size_t i = 0; i < paf->n && rc == 0; i++) {
        rbo_rec *r = &paf->recs[i];
        tq_entry *e = ent[i];
        if (e->total_bp == 0) { /* attempt to divide by zero */
            rc = -RBO_PANIC_ASSERT;
            break;
        }
        if (orders) orders[i] = e->order / e->total_bp;
        size_t L = strlen(r->q_name);
        char *nn = (char *)xmalloc(L + 2);
        memcpy(nn, r->q_name, L);
        nn[L + 1] = 0;
        if (e->orient < 0) {
            nn[L] = '-';
            uint64_t ns = r->q_len - r->q_en, ne = r->q_len - r->q_st;
            r->q_st = ns;
            r->q_en = ne;
            r->strand = r->strand == '+' ? '-' : '+';
        } else {
            nn[L] = '+';
        }
        /* keep the old name alive until every lookup that points at it is done (they all are: ent[] is resolved) */
        char *old = r->q_name;
        r->q_name = nn;
        for (size_t k = 0; k < n; k++)
            if (tab[k].q == old) tab[k].q = "";
        free(old);
    }
    free(ent);
    free(tab);
    return rc;
}
/* paf.rs:160-207 */
typedef struct {
    const rbo_rec *recs;
    const uint64_t *orders;
} scaf_ctx;
static int cmp_scaf(const void *ctx, size_t a, size_t b) {
    const scaf_ctx *c = (const scaf_ctx *)ctx;
    int t = strcmp(c->recs[a].t_name, c->recs[b].t_name);
    if (t) return t;
    if (c->orders[a] != c->orders[b]) return c->orders[a] < c->orders[b] ? -1 : 1;
    if (c->recs[a].q_st != c->recs[b].q_st) return c->recs[a].q_st < c->recs[b].q_st ? -1 : 1;
    return 0;
}
void rbo_paf_scaffold(rbo_paf *paf, uint64_t *orders, uint64_t spacer) {
    size_t n = paf->n;
    size_t *ix = (size_t *)xmalloc((n + 1) * sizeof(size_t)), *tmp = (size_t *)xmalloc((n + 1) * sizeof(size_t));
    for (size_t i = 0; i < n; i++) ix[i] = i;
    scaf_ctx c = {paf->recs, orders};
    msort(ix, tmp, n, cmp_scaf, &c);
    rbo_rec *nr = (rbo_rec *)xmalloc((paf->cap ? paf->cap : 1) * sizeof(rbo_rec));
    uint64_t *no = (uint64_t *)xmalloc((n + 1) * sizeof(uint64_t));
    for (size_t i = 0; i < n; i++) {
        nr[i] = paf->recs[ix[i]];
        no[i] = orders[ix[i]];
    }
    free(paf->recs);
    paf->recs = nr;
    memcpy(orders, no, n * sizeof(uint64_t));
    free(no);
    free(ix);
    free(tmp);
    for (size_t g0 = 0; g0 < n;) { /* group_by t_name (consecutive) */
        size_t g1 = g0;
        while (g1 < n && strcmp(paf->recs[g1].t_name, paf->recs[g0].t_name) == 0) g1++;
        /* (already sorted by order, q_st inside the group; the second sort at :173-177 is a no-op) */
        size_t cap = 64, len = 0;
        char *name = (char *)xmalloc(cap);
        name[0] = 0;
        for (size_t i = g0; i < g1; i++) { /* unique q_names in order of first appearance, joined by "::" */
            int seen = 0;
            for (size_t k = g0; k < i; k++)
                if (strcmp(paf->recs[k].q_name, paf->recs[i].q_name) == 0) {
                    seen = 1;
                    break;
                }
            if (seen) continue;
            size_t L = strlen(paf->recs[i].q_name);
            while (len + L + 3 > cap) {
                cap *= 2;
                name = (char *)xrealloc(name, cap);
            }
            if (len) {
                memcpy(name + len, "::", 2);
                len += 2;
            }
            memcpy(name + len, paf->recs[i].q_name, L + 1);
            len += L;
        }
        uint64_t scaffold_len = 0;
        for (size_t q0 = g0; q0 < g1;) { /* group_by q_name (consecutive) */
            size_t q1 = q0;
            while (q1 < g1 && strcmp(paf->recs[q1].q_name, paf->recs[q0].q_name) == 0) q1++;
            uint64_t q_min = UINT64_MAX, q_max = 0;
            for (size_t i = q0; i < q1; i++) {
                if (paf->recs[i].q_st < q_min) q_min = paf->recs[i].q_st;
                if (paf->recs[i].q_en > q_max) q_max = paf->recs[i].q_en;
            }
            for (size_t i = q0; i < q1; i++) {
                paf->recs[i].q_st = paf->recs[i].q_st - q_min + scaffold_len;
                paf->recs[i].q_en = paf->recs[i].q_en - q_min + scaffold_len;
            }
            scaffold_len += (q_max - q_min) + spacer;
            q0 = q1;
        }
        scaffold_len -= spacer;
        for (size_t i = g0; i < g1; i++) {
            free(paf->recs[i].q_name);
            paf->recs[i].q_name = xstrdup(name);
            paf->recs[i].q_len = scaffold_len;
        }
        free(name);
        g0 = g1;
    }
}

/* ------------------------------------------------------------------ bamstats.rs:107-154 (md = None) */
void rbo_stats_from_cigar(const rbo_cig *ops, size_t n, rbo_stats *s) {
    memset(s, 0, sizeof(*s));
    for (size_t i = 0; i < n; i++) {
        const uint32_t op = (uint32_t)(ops[i] & 15), val = (uint32_t)(ops[i] >> 4); /* (a Cigar length is u32) */
        switch (op) {
        case RBO_D:
            s->del_events += 1;
            s->del += val;
            break;
        case RBO_I:
            s->ins_events += 1;
            s->ins += val;
            break;
        case RBO_EQ:
            s->equal += val;
            break;
        case RBO_X:
            s->diff += val;
            break;
        case RBO_M:
            s->diff += val;
            s->matches += val;
            break;
        default:
            break;
        }
    }
    /* `100.0 * equal as f32 / (sum) as f32`: u32 sums, each cast to f32, (100*e)/s */
    volatile float e = (float)s->equal;
    volatile float num = 100.0f * e;
    s->id_by_all = num / (float)(uint32_t)(s->equal + s->diff + s->del + s->ins);
    s->id_by_events = num / (float)(uint32_t)(s->equal + s->diff + s->del_events + s->ins_events);
    s->id_by_matches = num / (float)(uint32_t)(s->equal + s->diff);
}

/* Rust `impl Display for f32`: shortest digit string that round-trips, printed positionally
 * (never in exponent form), "NaN", "inf".  (Text format: parity unpinned.) */
size_t rbo_f32_display(float v, char *buf, size_t cap) {
    if (isnan(v)) return (size_t)snprintf(buf, cap, "NaN");
    if (isinf(v)) return (size_t)snprintf(buf, cap, v < 0 ? "-inf" : "inf");
    if (v == 0.0f) return (size_t)snprintf(buf, cap, signbit(v) ? "-0" : "0");
    char digits[32];
    int exp10 = 0, nd = 0, neg = v < 0;
    float a = fabsf(v);
    for (int p = 1; p <= 9; p++) {
        char tmp[64];
        snprintf(tmp, sizeof tmp, "%.*e", p - 1, (double)a);
        int found = strtof(tmp, NULL) == a;
        if (!found) {
            /* asymmetric rounding interval at binade boundaries: try the neighbouring
             * p-digit decimals, as a shortest-digits algorithm would */
            char *e = strchr(tmp, 'e');
            int ex = atoi(e + 1);
            char mant[32];
            int k = 0;
            for (char *c = tmp; c < e; c++)
                if (*c != '.') mant[k++] = *c;
            mant[k] = 0;
            long long m = atoll(mant);
            double best = -1;
            for (int dlt = -1; dlt <= 1; dlt += 2) {
                long long m2 = m + dlt;
                if (m2 <= 0) continue;
                char t2[64];
                snprintf(t2, sizeof t2, "%llde%d", m2, ex - (p - 1));
                if (strtof(t2, NULL) == a) {
                    double err = fabs(strtod(t2, NULL) - (double)a);
                    if (best < 0 || err < best) {
                        best = err;
                        snprintf(tmp, sizeof tmp, "%llde%d", m2, ex - (p - 1));
                        found = 2;
                    }
                }
            }
        }
        if (found) {
            /* normalise to digits + exponent of the first digit */
            char *e = strchr(tmp, 'e');
            int ex = atoi(e + 1);
            nd = 0;
            int seen_point = 0, int_digits = 0;
            for (char *c = tmp; c < e; c++) {
                if (*c == '.') {
                    seen_point = 1;
                    continue;
                }
                digits[nd++] = *c;
                if (!seen_point) int_digits++;
            }
            exp10 = ex + int_digits - 1;
            /* strip leading zeros (none expected) and trailing zeros */
            while (nd > 1 && digits[nd - 1] == '0') nd--;
            digits[nd] = 0;
            break;
        }
    }
    size_t k = 0;
#define PUT(ch)                       \
    do {                              \
        if (k + 1 < cap) buf[k] = (ch); \
        k++;                          \
    } while (0)
    if (neg) PUT('-');
    if (exp10 >= 0) {
        for (int i = 0; i <= exp10; i++) PUT(i < nd ? digits[i] : '0');
        if (nd > exp10 + 1) {
            PUT('.');
            for (int i = exp10 + 1; i < nd; i++) PUT(digits[i]);
        }
    } else {
        PUT('0');
        PUT('.');
        for (int i = 0; i < -exp10 - 1; i++) PUT('0');
        for (int i = 0; i < nd; i++) PUT(digits[i]);
    }
#undef PUT
    if (cap) buf[k < cap ? k : cap - 1] = 0;
    return k;
}

/* bamstats.rs:225-236 */
void rbo_print_stats_header(int qbed, FILE *f) {
    if (qbed) {
        fputs("#query_name\tquery_start\tquery_end\tquery_length\t", f);
        fputs("strand\t", f);
        fputs("reference_name\treference_start\treference_end\treference_length\t", f);
    } else {
        fputs("#reference_name\treference_start\treference_end\treference_length\t", f);
        fputs("strand\t", f);
        fputs("query_name\tquery_start\tquery_end\tquery_length\t", f);
    }
    fputs("perID_by_matches\tperID_by_events\tperID_by_all\tmatches\tmismatches\tdeletion_events\tinsertion_"
          "events\tdeletions\tinsertions\n",
          f);
}
/* bamstats.rs:239-270 with stats_from_paf :91-105 */
void rbo_print_stats(const rbo_rec *r, const rbo_stats *s, int qbed, FILE *f) {
    if (qbed) {
        fprintf(f, "%s\t%lld\t%lld\t%lld\t", r->q_name, (long long)r->q_st, (long long)r->q_en, (long long)r->q_len);
        fprintf(f, "%c\t", r->strand);
        fprintf(f, "%s\t%lld\t%lld\t%lld\t", r->t_name, (long long)r->t_st, (long long)r->t_en, (long long)r->t_len);
    } else {
        fprintf(f, "%s\t%lld\t%lld\t%lld\t", r->t_name, (long long)r->t_st, (long long)r->t_en, (long long)r->t_len);
        fprintf(f, "%c\t", r->strand);
        fprintf(f, "%s\t%lld\t%lld\t%lld\t", r->q_name, (long long)r->q_st, (long long)r->q_en, (long long)r->q_len);
    }
    char a[64], b[64], c[64];
    rbo_f32_display(s->id_by_matches, a, sizeof a);
    rbo_f32_display(s->id_by_events, b, sizeof b);
    rbo_f32_display(s->id_by_all, c, sizeof c);
    fprintf(f, "%s\t%s\t%s\t", a, b, c);
    fprintf(f, "%u\t%u\t%u\t%u\t%u\t%u\n", s->equal, s->diff, s->del_events, s->ins_events, s->del, s->ins);
}

/* ==================================================================================
 * BAM input of `rb stats` (main.rs:60-77, bamstats.rs:156-222).  BGZF is read through zlib (a BGZF file is a
 * multi-member gzip file).  CigarStringView::{end_pos, read_pos, leading/trailing clips} are restated from
 * rust-htslib 0.44.1's published algorithm (only a smoke test exists in the reference: parity unpinned).
 * ================================================================================== */
/* bamstats.rs:48-79: regex (\d+)|([A-Z])|(\^[A-Z]+) scanned left to right */
void rbo_parse_md_for_stats(const char *md, uint32_t out[4]) {
    uint32_t match_count = 0, mismatch_count = 0, insertion_count = 0, insertion_bases = 0;
    const char *p = md;
    while (*p) {
        if (*p >= '0' && *p <= '9') {
            uint64_t v = 0;
            while (*p >= '0' && *p <= '9') v = v * 10 + (uint64_t)(*p++ - '0');
            match_count += (uint32_t)v;
        } else if (*p >= 'A' && *p <= 'Z') {
            mismatch_count += 1;
            p++;
        } else if (*p == '^' && p[1] >= 'A' && p[1] <= 'Z') {
            const char *q = p + 1;
            while (*q >= 'A' && *q <= 'Z') q++;
            insertion_bases += (uint32_t)(q - p) - 1;
            insertion_count += 1;
            p = q;
        } else {
            p++; /* not matched by the regex: skipped */
        }
    }
    out[0] = match_count;
    out[1] = mismatch_count;
    out[2] = insertion_count;
    out[3] = insertion_bases;
}

/* rust-htslib CigarStringView::read_pos(ref_pos, include_softclips = false, include_dels = false).
 * returns 0 = Ok(Some(*qpos)), 1 = Ok(None), -1 = Err */
static int hts_read_pos(const uint32_t *cig, size_t n, int64_t pos, uint32_t ref_pos, uint32_t *qpos_out) {
    uint32_t rpos = (uint32_t)pos, qpos = 0;
    size_t j = 0;
    for (size_t i = 0; i < n; i++) {
        uint32_t op = cig[i] & 15;
        if (op == RBO_M || op == RBO_X || op == RBO_EQ || op == RBO_I || op == RBO_S) {
            j = i;
            break;
        }
        if (op == RBO_D || op == RBO_N) return -1;
        if (op == RBO_H && i > 0 && i + 1 < n) return -1;
        if ((op == RBO_P || op == RBO_H) && i + 1 == n) return 1;
        /* leading H / P: skipped */
    }
    while (rpos <= ref_pos && j < n) {
        uint32_t op = cig[j] & 15, l = cig[j] >> 4;
        if (op == RBO_M || op == RBO_X || op == RBO_EQ) {
            if (rpos <= ref_pos && rpos + l > ref_pos) {
                *qpos_out = qpos + (ref_pos - rpos);
                return 0;
            }
            rpos += l;
            qpos += l;
            j++;
        } else if (op == RBO_S || op == RBO_I) {
            qpos += l;
            j++;
        } else if (op == RBO_N || op == RBO_D) {
            rpos += l;
            j++;
        } else if (op == RBO_P) {
            j++;
        } else { /* H */
            if (j + 1 < n) return -1;
            return 1;
        }
    }
    return 1;
}

typedef struct {
    gzFile f;
} bam_reader;
static int bam_read_exact(gzFile f, void *buf, size_t n) {
    size_t got = 0;
    while (got < n) {
        int r = gzread(f, (char *)buf + got, (unsigned)((n - got) > (1u << 30) ? (1u << 30) : (n - got)));
        if (r <= 0) return 0;
        got += (size_t)r;
    }
    return 1;
}
static uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* walk the aux area: find the MD:Z string and a CG:B,I long cigar */
static void bam_aux_scan(const uint8_t *a, size_t n, const char **md, const uint8_t **cg, uint32_t *cg_n) {
    size_t p = 0;
    *md = NULL;
    *cg = NULL;
    *cg_n = 0;
    while (p + 3 <= n) {
        const uint8_t *tag = a + p;
        char ty = (char)a[p + 2];
        p += 3;
        size_t sz = 0;
        if (ty == 'A' || ty == 'c' || ty == 'C') sz = 1;
        else if (ty == 's' || ty == 'S') sz = 2;
        else if (ty == 'i' || ty == 'I' || ty == 'f') sz = 4;
        else if (ty == 'Z' || ty == 'H') {
            size_t q = p;
            while (q < n && a[q]) q++;
            if (tag[0] == 'M' && tag[1] == 'D' && ty == 'Z') *md = (const char *)(a + p);
            p = q + 1;
            continue;
        } else if (ty == 'B') {
            if (p + 5 > n) return;
            char sub = (char)a[p];
            uint32_t cnt = rd_u32(a + p + 1);
            size_t es = (sub == 'c' || sub == 'C') ? 1 : ((sub == 's' || sub == 'S') ? 2 : 4);
            if (tag[0] == 'C' && tag[1] == 'G' && sub == 'I') {
                *cg = a + p + 5;
                *cg_n = cnt;
            }
            p += 5 + es * (size_t)cnt;
            continue;
        } else {
            return;
        }
        p += sz;
    }
}

/* `rb stats <bam>`: header line + one line per mapped record.  returns 0, or <0 where the reference panics */
int rbo_bam_stats(const char *path, int qbed, FILE *out) {
    gzFile f = strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb");
    if (!f) return -2;
    gzbuffer(f, 1 << 20);
    uint8_t h8[8];
    if (!bam_read_exact(f, h8, 8) || memcmp(h8, "BAM\1", 4) != 0) {
        gzclose(f);
        return -3;
    }
    uint32_t l_text = rd_u32(h8 + 4);
    char *text = (char *)xmalloc(l_text + 1);
    bam_read_exact(f, text, l_text);
    free(text);
    uint8_t b4[4];
    bam_read_exact(f, b4, 4);
    uint32_t n_ref = rd_u32(b4);
    char **ref_nm = (char **)xmalloc((n_ref + 1) * sizeof(char *));
    uint32_t *ref_len = (uint32_t *)xmalloc((n_ref + 1) * sizeof(uint32_t));
    for (uint32_t i = 0; i < n_ref; i++) {
        bam_read_exact(f, b4, 4);
        uint32_t l = rd_u32(b4);
        ref_nm[i] = (char *)xmalloc(l + 1);
        bam_read_exact(f, ref_nm[i], l);
        ref_nm[i][l] = 0;
        bam_read_exact(f, b4, 4);
        ref_len[i] = rd_u32(b4);
    }
    rbo_print_stats_header(qbed, out);
    int rc = 0;
    uint8_t *rec = NULL;
    size_t cap = 0;
    for (;;) {
        /* htslib bam_read1 (sam.c): 0 bytes where a record would start = end of file; a file that ends inside the length field
         * or inside the record, a block_size < 32, or name + cigar + sequence that do not fit the block are read errors, and the
         * reference's rec.unwrap() (main.rs:72) panics -- after the records before it have been printed */
        int got = gzread(f, b4, 4);
        if (got == 0) break;
        if (got != 4) {
            rc = -RBO_PANIC_NOTFOUND;
            break;
        }
        uint32_t bs = rd_u32(b4);
        if (bs > cap) {
            cap = (size_t)bs * 2;
            rec = (uint8_t *)xrealloc(rec, cap);
        }
        if (!bam_read_exact(f, rec, bs) || bs < 32) {
            rc = -RBO_PANIC_NOTFOUND;
            break;
        }
        int32_t refID = (int32_t)rd_u32(rec);
        int64_t pos = (int32_t)rd_u32(rec + 4);
        uint32_t l_rn = rec[8];
        uint32_t n_cig = (uint32_t)rec[12] | ((uint32_t)rec[13] << 8);
        uint32_t flag = (uint32_t)rec[14] | ((uint32_t)rec[15] << 8);
        uint32_t l_seq = rd_u32(rec + 16);
        if (32ull + l_rn + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq > bs || l_rn == 0 || rec[32 + l_rn - 1] != 0) {
            rc = -RBO_PANIC_NOTFOUND;
            break;
        }
        if (flag & 4) continue; /* main.rs:73 */
        const char *qname = (const char *)(rec + 32);
        const uint8_t *cg_raw = rec + 32 + l_rn;
        size_t aux_off = 32 + l_rn + 4 * (size_t)n_cig + (l_seq + 1) / 2 + l_seq;
        const char *md = NULL;
        const uint8_t *cg_tag = NULL;
        uint32_t cg_n = 0;
        if (aux_off <= bs) bam_aux_scan(rec + aux_off, bs - aux_off, &md, &cg_tag, &cg_n);
        uint32_t *cig;
        size_t n;
        if (cg_tag && n_cig >= 1 && (rd_u32(cg_raw) & 15) == RBO_S && (rd_u32(cg_raw) >> 4) == l_seq) { /* htslib: real cigar in CG */
            n = cg_n;
            cig = (uint32_t *)xmalloc((n + 1) * sizeof(uint32_t));
            for (size_t i = 0; i < n; i++) cig[i] = rd_u32(cg_tag + 4 * i);
        } else {
            n = n_cig;
            cig = (uint32_t *)xmalloc((n + 1) * sizeof(uint32_t));
            for (size_t i = 0; i < n; i++) cig[i] = rd_u32(cg_raw + 4 * i);
        }
        /* bamstats.rs:156-207 */
        int64_t r_st = pos, r_en = pos;
        for (size_t i = 0; i < n; i++)
            if (rbo_consumes_reference(cig[i])) r_en += cig[i] >> 4;
        int64_t lead_h = (n && (cig[0] & 15) == RBO_H) ? (cig[0] >> 4) : 0;
        int64_t lead_s = 0;
        if (n && (cig[0] & 15) == RBO_S) lead_s = cig[0] >> 4;
        else if (n > 1 && (cig[0] & 15) == RBO_H && (cig[1] & 15) == RBO_S) lead_s = cig[1] >> 4;
        int64_t trail_h = (n && (cig[n - 1] & 15) == RBO_H) ? (cig[n - 1] >> 4) : 0;
        uint32_t qp = 0;
        int rp = hts_read_pos(cig, n, pos, (uint32_t)r_en - 1, &qp);
        if (rp != 0) { /* .unwrap().unwrap() */
            fprintf(stderr, "rb_oracle: read_pos failed for %s: the reference panics\n", qname);
            free(cig);
            rc = -RBO_PANIC_NOTFOUND;
            break;
        }
        int64_t q_st = lead_h + lead_s;
        int64_t q_en = lead_h + 1 + (int64_t)qp;
        int64_t q_len = lead_h + (int64_t)l_seq + trail_h;
        if (flag & 16) {
            int64_t t = q_st;
            q_st = q_len - q_en;
            q_en = q_len - t;
        }
        rbo_stats s;
        {   /* (BAM words are 28-bit lengths already: widened as they stand) */
            rbo_cig *wide = (rbo_cig *)xmalloc((n ? n : 1) * sizeof(rbo_cig));
            for (size_t k = 0; k < (size_t)n; k++) wide[k] = cig[k];
            rbo_stats_from_cigar(wide, n, &s);
            free(wide);
        }
        if (s.equal == 0 && s.matches > 0 && md) { /* bamstats.rs:129-135 */
            uint32_t m4[4];
            rbo_parse_md_for_stats(md, m4);
            if (m4[0] + m4[1] != s.diff) {
                free(cig);
                rc = -RBO_PANIC_ASSERT;
                break;
            }
            s.equal = m4[0];
            s.diff = m4[1];
            volatile float e = (float)s.equal;
            volatile float num = 100.0f * e;
            s.id_by_all = num / (float)(uint32_t)(s.equal + s.diff + s.del + s.ins);
            s.id_by_events = num / (float)(uint32_t)(s.equal + s.diff + s.del_events + s.ins_events);
            s.id_by_matches = num / (float)(uint32_t)(s.equal + s.diff);
        }
        rbo_rec r;
        rbo_rec_init(&r);
        free(r.q_name);
        free(r.t_name);
        r.q_name = xstrdup(qname);
        r.t_name = xstrdup((refID >= 0 && (uint32_t)refID < n_ref) ? ref_nm[refID] : "*");
        r.t_len = (refID >= 0 && (uint32_t)refID < n_ref) ? ref_len[refID] : 0;
        r.t_st = (uint64_t)r_st;
        r.t_en = (uint64_t)r_en;
        r.q_st = (uint64_t)q_st;
        r.q_en = (uint64_t)q_en;
        r.q_len = (uint64_t)q_len;
        r.strand = (flag & 16) ? '-' : '+';
        rbo_print_stats(&r, &s, qbed, out);
        rbo_rec_free(&r);
        free(cig);
    }
    free(rec);
    for (uint32_t i = 0; i < n_ref; i++) free(ref_nm[i]);
    free(ref_nm);
    free(ref_len);
    gzclose(f);
    return rc;
}

/* ==================================================================================
 * nucfreq (nucfreq.rs:61-95, :111-125; main.rs:82-121): A/C/G/T counts at every covered position of a region.
 *
 * The pileup engine is third-party: rust-htslib 0.44.1 (Cargo.lock:1533-1534) `Read::pileup()` over hts-sys 2.2.0
 * (Cargo.lock:723-724, which vendors htslib) = htslib `bam_plp_init` / `bam_plp_auto` with
 * the default mask (UNMAP | SECONDARY | QCFAIL | DUP) and maxcnt 8000.  It is absent from /root/reference; the
 * published algorithm (htslib sam.c: bam_plp_push, bam_plp64_next, resolve_cigar2) is restated below, position by
 * position with the same per-read cursor.  Only KA13 (nucfreq.rs:41-60) pins it: the depth cap and the behaviour on
 * malformed cigars are "parity unpinned".
 * ================================================================================== */
#define PLP_MASK (0x4u | 0x100u | 0x200u | 0x400u)
#define PLP_MAXCNT 8000

/* bam_endpos: pos + reference length (0 for unmapped reads), at least pos + 1 */
static int64_t plp_endpos(const rbo_read *b) {
    int64_t rlen = 0;
    if (!(b->flag & 4u))
        for (uint32_t k = 0; k < b->n_cigar; k++)
            if (rbo_consumes_reference(b->cigar[k])) rlen += b->cigar[k] >> 4;
    if (rlen == 0) rlen = 1;
    return b->pos + rlen;
}

typedef struct plp_node {
    const rbo_read *b;
    int64_t beg, end;
    int k;        /* cstate: current op (-1 = never processed) */
    int64_t x, y; /* reference position / query offset of the start of op k */
    struct plp_node *next;
} plp_node;

static int plp_is_mdnex(uint32_t op) { return op == RBO_M || op == RBO_D || op == RBO_N || op == RBO_EQ || op == RBO_X; }
static int plp_is_mex(uint32_t op) { return op == RBO_M || op == RBO_EQ || op == RBO_X; }

/* resolve_cigar2: moves the cursor of one read to `pos`; *is_del / *qpos as bam_pileup1_t.  <0 where htslib asserts */
static int plp_resolve(plp_node *s, int64_t pos, int *is_del, int64_t *qpos) {
    const rbo_read *b = s->b;
    const uint32_t *cg = b->cigar;
    int n = (int)b->n_cigar, k;
    if (s->k == -1) {
        if (n == 1) {
            if (plp_is_mex(cg[0] & 15u)) s->k = 0, s->x = b->pos, s->y = 0;
        } else {
            for (k = 0, s->x = b->pos, s->y = 0; k < n; ++k) {
                uint32_t op = cg[k] & 15u, l = cg[k] >> 4;
                if (plp_is_mdnex(op)) break;
                else if (op == RBO_I || op == RBO_S) s->y += l;
            }
            if (k >= n) return -5;
            s->k = k;
        }
        if (s->k < 0) return -5; /* (a lone non-match op: htslib reads cigar[-1]) */
    } else {
        int64_t l = cg[s->k] >> 4;
        if (pos - s->x >= l) {
            if (s->k + 1 >= n) return -5;
            uint32_t op = cg[s->k + 1] & 15u;
            if (plp_is_mdnex(op)) {
                if (plp_is_mex(cg[s->k] & 15u)) s->y += l;
                s->x += l;
                ++s->k;
            } else {
                if (plp_is_mex(cg[s->k] & 15u)) s->y += l;
                s->x += l;
                for (k = s->k + 1; k < n; ++k) {
                    uint32_t o = cg[k] & 15u, ll = cg[k] >> 4;
                    if (plp_is_mdnex(o)) break;
                    else if (o == RBO_I || o == RBO_S) s->y += ll;
                }
                s->k = k;
            }
            if (s->k >= n) return -5;
        }
    }
    uint32_t op = cg[s->k] & 15u;
    *is_del = 0;
    if (plp_is_mex(op)) *qpos = s->y + (pos - s->x);
    else {
        *is_del = 1; /* D and N both set is_del; N also is_refskip: neither is counted (nucfreq.rs:79) */
        *qpos = s->y;
    }
    return 0;
}

/* one fetch + pileup (nucfreq.rs:111-125 then :61-95): rows for the covered positions of [st, en) on tid.
 * returns 0, -1 = unsorted input (the iterator errors, p.unwrap() panics), -4 = base index past the sequence (panic),
 * -5 = htslib assertion */
int rbo_nucfreq(const rbo_read *reads, size_t n_reads, int32_t rtid, uint64_t st, uint64_t en, rbo_nucfreq_row **rows_out, size_t *n_rows) {
    size_t cap = 1024, nr = 0;
    rbo_nucfreq_row *rows = (rbo_nucfreq_row *)xmalloc(cap * sizeof(*rows));
    int rc = 0;
    /* iterator state (bam_plp_init) */
    plp_node *head = (plp_node *)xmalloc(sizeof(plp_node)), *tail = head;
    memset(head, 0, sizeof(*head));
    long mp_cnt = 1;
    int32_t it_tid = 0, max_tid = -1;
    int64_t it_pos = 0, max_pos = -1;
    int is_eof = 0;
    size_t next_read = 0;
    for (;;) {
        /* ---- bam_plp64_next ---- */
        int have = 0, n_plp = 0;
        int32_t o_tid = 0;
        int64_t o_pos = 0;
        uint64_t cnt[4] = {0, 0, 0, 0};
        int finished = 0;
        if (is_eof && head == tail) finished = 1;
        while (!finished && !have && (is_eof || max_tid > it_tid || (max_tid == it_tid && max_pos > it_pos))) {
            n_plp = 0;
            cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
            plp_node **pptr = &head;
            while (*pptr != tail) {
                plp_node *p = *pptr;
                if (p->b->tid < it_tid || (p->b->tid == it_tid && p->end <= it_pos)) {
                    *pptr = p->next;
                    free(p);
                    --mp_cnt;
                } else {
                    if (p->b->tid == it_tid && p->beg <= it_pos) {
                        int is_del = 0;
                        int64_t qpos = 0;
                        int r = plp_resolve(p, it_pos, &is_del, &qpos);
                        if (r < 0) { rc = r; goto done; }
                        ++n_plp;
                        if (!is_del && (uint64_t)it_pos >= st && (uint64_t)it_pos < en && it_tid == rtid) { /* nucfreq.rs:65, :79-91 */
                            if (qpos < 0 || (uint64_t)qpos >= p->b->l_seq) { rc = -4; goto done; }
                            uint8_t nib = (p->b->seq[qpos >> 1] >> ((~qpos & 1) << 2)) & 15u;
                            if (nib == 1) cnt[0]++;
                            else if (nib == 2) cnt[1]++;
                            else if (nib == 4) cnt[2]++;
                            else if (nib == 8) cnt[3]++;
                            /* N: silent; anything else: a warning on stderr, not counted */
                        }
                    }
                    pptr = &(*pptr)->next;
                }
            }
            o_tid = it_tid;
            o_pos = it_pos;
            if (head != tail && it_tid > head->b->tid) { rc = -1; goto done; }
            if (head != tail && it_tid < head->b->tid) {
                it_tid = head->b->tid;
                it_pos = head->beg;
            } else if (head != tail && it_pos < head->beg) {
                it_pos = head->beg;
            } else ++it_pos;
            if (n_plp) have = 1;
            else if (is_eof && head == tail) finished = 1;
        }
        if (have) {
            if (o_tid == rtid && (uint64_t)o_pos >= st && (uint64_t)o_pos < en) {
                if (nr == cap) {
                    cap *= 2;
                    rows = (rbo_nucfreq_row *)xrealloc(rows, cap * sizeof(*rows));
                }
                rows[nr].pos = (uint32_t)o_pos;
                rows[nr].a = cnt[0], rows[nr].c = cnt[1], rows[nr].g = cnt[2], rows[nr].t = cnt[3];
                nr++;
            }
            continue;
        }
        if (finished) break;
        /* ---- bam_plp_auto: read the next record of the fetch and push it ---- */
        const rbo_read *b = NULL;
        while (next_read < n_reads) { /* hts_itr_next on (tid, st, en): same contig, pos < en, endpos > st */
            const rbo_read *c = &reads[next_read++];
            if (c->tid == rtid && c->pos < (int64_t)en && plp_endpos(c) > (int64_t)st) { b = c; break; }
        }
        if (!b) { is_eof = 1; continue; }
        /* ---- bam_plp_push ---- */
        if (b->tid < 0) continue;
        if (b->flag & PLP_MASK) continue;
        if (it_tid == b->tid && it_pos == b->pos && mp_cnt > PLP_MAXCNT) continue;
        tail->b = b;
        tail->beg = b->pos;
        tail->end = plp_endpos(b);
        tail->k = -1, tail->x = tail->y = 0;
        if (b->tid < max_tid || (b->tid == max_tid && tail->beg < max_pos)) { rc = -1; goto done; }
        max_tid = b->tid;
        max_pos = tail->beg;
        if (tail->end > it_pos || tail->b->tid > it_tid) {
            plp_node *nn = (plp_node *)xmalloc(sizeof(plp_node));
            memset(nn, 0, sizeof(*nn));
            ++mp_cnt;
            tail->next = nn;
            tail = nn;
        }
    }
done:
    while (head) {
        plp_node *nx = head == tail ? NULL : head->next;
        free(head);
        head = nx;
    }
    if (rc) {
        free(rows);
        rows = NULL;
        nr = 0;
    }
    *rows_out = rows;
    *n_rows = nr;
    return rc;
}

/* flat-array form: one region; rows go to out_pos[cap] / out_cnt[4 * cap]; returns the row count or <0 */
int64_t rbo_nucfreq_arrays(uint64_t n_reads, const int32_t *tid, const int64_t *pos, const uint32_t *flag, const uint64_t *op_off,
                           const uint32_t *ops, const uint32_t *l_seq, const uint64_t *seq_off, const uint8_t *seq, int32_t rtid,
                           uint64_t st, uint64_t en, uint32_t *out_pos, uint64_t *out_cnt, uint64_t cap) {
    rbo_read *rd = (rbo_read *)xmalloc((n_reads + 1) * sizeof(*rd));
    for (uint64_t i = 0; i < n_reads; i++) {
        rd[i].tid = tid[i], rd[i].pos = pos[i], rd[i].flag = flag[i];
        rd[i].n_cigar = (uint32_t)(op_off[i + 1] - op_off[i]);
        rd[i].cigar = ops + op_off[i];
        rd[i].l_seq = l_seq[i];
        rd[i].seq = seq + seq_off[i];
    }
    rbo_nucfreq_row *rows = NULL;
    size_t n = 0;
    int rc = rbo_nucfreq(rd, n_reads, rtid, st, en, &rows, &n);
    free(rd);
    if (rc) return rc;
    if (n > cap) {
        free(rows);
        return -100;
    }
    for (size_t i = 0; i < n; i++) {
        out_pos[i] = rows[i].pos;
        out_cnt[4 * i] = rows[i].a, out_cnt[4 * i + 1] = rows[i].c, out_cnt[4 * i + 2] = rows[i].g, out_cnt[4 * i + 3] = rows[i].t;
    }
    free(rows);
    return (int64_t)n;
}

/* bed.rs:104-131 parse_region: regex (.+):([0-9]+)-([0-9]+), leftmost match, greedy name */
int rbo_parse_region(const char *s, rbo_region *out) {
    size_t n = strlen(s);
    /* greedy (.+): the LAST position where ":digits-digits" starts, with at least one name character before it */
    for (size_t c = n; c-- > 1;) {
        if (s[c] != ':') continue;
        size_t i = c + 1, a0 = i;
        while (i < n && s[i] >= '0' && s[i] <= '9') i++;
        if (i == a0 || i >= n || s[i] != '-') continue;
        size_t b0 = ++i;
        while (i < n && s[i] >= '0' && s[i] <= '9') i++;
        if (i == b0) continue;
        /* st: parse::<u64>().unwrap() - 1 */
        uint64_t st = 0, en = 0;
        int en_ok = 1;
        for (size_t k = a0; k < b0 - 1; k++) {
            if (st > (UINT64_MAX - (uint64_t)(s[k] - '0')) / 10) return -1;
            st = st * 10 + (uint64_t)(s[k] - '0');
        }
        if (st == 0) return -1; /* 0 - 1 overflows */
        st -= 1;
        for (size_t k = b0; k < i; k++) {
            if (en > (UINT64_MAX - (uint64_t)(s[k] - '0')) / 10) { en_ok = 0; break; }
            en = en * 10 + (uint64_t)(s[k] - '0');
        }
        if (!en_ok) en = 4294967295ull; /* unwrap_or(2^32 - 1) */
        if (st > en) return -1;         /* assert!(st <= en) */
        out->name = (char *)xmalloc(c + 1);
        memcpy(out->name, s, c);
        out->name[c] = 0;
        out->st = st, out->en = en;
        out->id = NULL;
        rbo_region_default_id(out); /* name:st+1-en */
        return 0;
    }
    return -1;
}

/* `rb nucfreq [--region R] [--bed B] [--small] <bam>` (main.rs:82-121) */
int rbo_bam_nucfreq(const char *path, const char *region, const char *bed_path, int small, FILE *out) {
    gzFile f = strcmp(path, "-") == 0 ? gzdopen(0, "rb") : gzopen(path, "rb");
    if (!f) return -2;
    gzbuffer(f, 1 << 20);
    uint8_t h8[8], b4[4];
    if (!bam_read_exact(f, h8, 8) || memcmp(h8, "BAM\1", 4) != 0) {
        gzclose(f);
        return -3;
    }
    uint32_t l_text = rd_u32(h8 + 4);
    char *text = (char *)xmalloc(l_text + 1);
    bam_read_exact(f, text, l_text);
    free(text);
    bam_read_exact(f, b4, 4);
    uint32_t n_ref = rd_u32(b4);
    char **ref_nm = (char **)xmalloc((n_ref + 1) * sizeof(char *));
    for (uint32_t i = 0; i < n_ref; i++) {
        bam_read_exact(f, b4, 4);
        uint32_t l = rd_u32(b4);
        ref_nm[i] = (char *)xmalloc(l + 1);
        bam_read_exact(f, ref_nm[i], l);
        ref_nm[i][l] = 0;
        bam_read_exact(f, b4, 4);
    }
    size_t n = 0, cap = 1024;
    rbo_read *rd = (rbo_read *)xmalloc(cap * sizeof(*rd));
    while (bam_read_exact(f, b4, 4)) {
        uint32_t bs = rd_u32(b4);
        uint8_t *rec = (uint8_t *)xmalloc(bs + 8);
        if (!bam_read_exact(f, rec, bs)) { free(rec); break; }
        if (n == cap) {
            cap *= 2;
            rd = (rbo_read *)xrealloc(rd, cap * sizeof(*rd));
        }
        rbo_read *r = &rd[n++];
        r->tid = (int32_t)rd_u32(rec);
        r->pos = (int32_t)rd_u32(rec + 4);
        uint32_t l_rn = rec[8];
        uint32_t n_cig = (uint32_t)rec[12] | ((uint32_t)rec[13] << 8);
        r->flag = (uint32_t)rec[14] | ((uint32_t)rec[15] << 8);
        r->l_seq = rd_u32(rec + 16);
        const uint8_t *cg_raw = rec + 32 + l_rn;
        r->seq = cg_raw + 4 * (size_t)n_cig;
        size_t aux_off = 32 + l_rn + 4 * (size_t)n_cig + (r->l_seq + 1) / 2 + r->l_seq;
        const char *md = NULL;
        const uint8_t *cg_tag = NULL;
        uint32_t cg_n = 0;
        if (aux_off <= bs) bam_aux_scan(rec + aux_off, bs - aux_off, &md, &cg_tag, &cg_n);
        const uint8_t *src = cg_raw;
        uint32_t nn = n_cig;
        if (cg_tag && n_cig >= 1 && (rd_u32(cg_raw) & 15) == RBO_S && (rd_u32(cg_raw) >> 4) == r->l_seq) src = cg_tag, nn = cg_n;
        uint32_t *cig = (uint32_t *)xmalloc((nn + 1) * sizeof(uint32_t));
        for (uint32_t i = 0; i < nn; i++) cig[i] = rd_u32(src + 4 * (size_t)i);
        r->cigar = cig;
        r->n_cigar = nn;
        /* (rec and cig stay allocated for the life of the process: test tool) */
    }
    gzclose(f);
    /* main.rs:90-98: --region first, then the bed file */
    rbo_bed rg;
    rg.r = NULL, rg.n = rg.cap = 0;
    int rc = 0;
    if (bed_path && rbo_bed_from_file(bed_path, &rg)) return -6;
    size_t n_rg = rg.n + (region ? 1 : 0);
    rbo_region *all = (rbo_region *)xmalloc((n_rg + 1) * sizeof(*all));
    size_t w = 0;
    if (region) {
        if (rbo_parse_region(region, &all[w])) return -7;
        w++;
    }
    for (size_t i = 0; i < rg.n; i++) all[w++] = rg.r[i];
    for (size_t i = 0; i < n_rg && !rc; i++) {
        const rbo_region *R = &all[i];
        int32_t tid = -1;
        for (uint32_t k = 0; k < n_ref; k++)
            if (!strcmp(ref_nm[k], R->name)) { tid = (int32_t)k; break; }
        uint64_t m0, m1;
        for (uint64_t mk = 0; !rc && rbo_split_region(R->st, R->en, 1000000, mk, &m0, &m1); mk++) { /* main.rs:101: split_region(1 Mbp) */
            if (tid < 0) { rc = -8; break; } /* fetch fails: "Is this region in your reference/bam?" */
            if (!small) fprintf(out, "#chr\tstart\tend\tA\tC\tG\tT\tregion_id\n"); /* nucfreq.rs:127-131 */
            int first = 1;
            uint64_t s0, s1;
            for (uint64_t sk = 0; !rc && rbo_split_region(m0, m1, 10000, sk, &s0, &s1); sk++) { /* split_region(10 kbp), one fetch + pileup each */
                rbo_nucfreq_row *rows = NULL;
                size_t nr = 0;
                rc = rbo_nucfreq(rd, n, tid, s0, s1, &rows, &nr);
                if (rc) break;
                for (size_t k = 0; k < nr; k++) {
                    if (small) { /* nucfreq.rs:139-153 (name and id are constant inside one call) */
                        if (first) fprintf(out, "#%s\t%u\t%s\n", R->name, rows[k].pos, R->id);
                        first = 0;
                        uint64_t mc[4] = {rows[k].a, rows[k].c, rows[k].g, rows[k].t};
                        for (int x = 0; x < 4; x++)
                            for (int y = x + 1; y < 4; y++)
                                if (mc[y] < mc[x]) { uint64_t t = mc[x]; mc[x] = mc[y]; mc[y] = t; }
                        fprintf(out, "%llu\t%llu\n", (unsigned long long)mc[3], (unsigned long long)mc[2]);
                    } else {
                        fprintf(out, "%s\t%u\t%u\t%llu\t%llu\t%llu\t%llu\t%s\n", R->name, rows[k].pos, rows[k].pos + 1,
                                (unsigned long long)rows[k].a, (unsigned long long)rows[k].c, (unsigned long long)rows[k].g,
                                (unsigned long long)rows[k].t, R->id);
                    }
                }
                free(rows);
            }
        }
    }
    return rc;
}

/* ==================================================================================
 * Flat-array API
 * ================================================================================== */
/* the words of one record -> r->cigar (a continuation word with no op in front of it stays an op of its own: code 14, consumes
 * nothing -- what the product makes of it too) */
static void cig_from_words(rbo_rec *r, const uint32_t *w, size_t n_words) {
    free(r->cigar);
    r->cigar = NULL;
    r->n_cigar = 0;
    if (rbo_words_to_cig(w, n_words, &r->cigar, &r->n_cigar)) {
        r->cigar = (rbo_cig *)xmalloc((n_words ? n_words : 1) * sizeof(rbo_cig));
        for (size_t k = 0; k < n_words; k++) r->cigar[k] = w[k];
        r->n_cigar = n_words;
    }
}
/* ops [0, k) of a cigar take this many words */
static size_t words_before(const rbo_cig *c, size_t k) { return rbo_words_of(c, k); }
static void rec_from_arrays(rbo_rec *r, uint64_t i, const uint32_t *ops, const uint64_t *op_off,
                            const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en,
                            const uint8_t *strand) {
    rbo_rec_init(r);
    cig_from_words(r, ops + op_off[i], (size_t)(op_off[i + 1] - op_off[i]));
    r->t_st = t_st[i];
    r->t_en = t_en[i];
    r->q_st = q_st[i];
    r->q_en = q_en[i];
    r->strand = strand ? (char)strand[i] : '+';
}

int rbo_reduce_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                      const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, rbo_reduce_row *out) {
    for (uint64_t i = 0; i < n_rec; i++) {
        rbo_rec r;
        rec_from_arrays(&r, i, ops, op_off, t_st, t_en, q_st, q_en, NULL);
        uint64_t o[4];
        rbo_reduce_row *w = &out[i];
        memset(w, 0, sizeof(*w));
        rbo_infer_n_bases(&r, o);
        w->t_bases = o[0];
        w->q_bases = o[1];
        w->nmatch = (uint32_t)o[2];
        w->aln_len = (uint32_t)o[3];
        w->status = (uint32_t)rbo_check_integrity(&r);
        rbo_stats s;
        rbo_stats_from_cigar(r.cigar, r.n_cigar, &s);
        w->equal = s.equal;
        w->diff = s.diff;
        w->ins = s.ins;
        w->del = s.del;
        w->matches = s.matches;
        w->ins_events = s.ins_events;
        w->del_events = s.del_events;
        w->id_by_all = s.id_by_all;
        w->id_by_events = s.id_by_events;
        w->id_by_matches = s.id_by_matches;
        rbo_rec_free(&r);
    }
    return 0;
}

int rbo_normalize_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                         const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                         rbo_norm_row *out) {
    for (uint64_t i = 0; i < n_rec; i++) {
        rbo_rec r;
        rec_from_arrays(&r, i, ops, op_off, t_st, t_en, q_st, q_en, strand);
        rbo_norm_row *w = &out[i];
        memset(w, 0, sizeof(*w));
        size_t n0 = r.n_cigar;
        /* count the stripped ops exactly as paf.rs:668-723 does */
        size_t lead = 0, trail = 0;
        while (lead < n0 && is_indel(r.cigar[lead])) lead++;
        while (trail < n0 && is_indel(r.cigar[n0 - 1 - trail])) trail++;
        const size_t lead_w = words_before(r.cigar, lead < n0 ? lead : n0);                                   /* (words at the boundary) */
        const size_t trail_w = rbo_words_of(r.cigar + (n0 - (trail < n0 ? trail : n0)), trail < n0 ? trail : n0);
        int st = rbo_remove_trailing_indels(&r);
        w->status = (uint32_t)st;
        w->lead_ops = (uint32_t)lead_w;
        w->trail_ops = (uint32_t)trail_w;
        if (st == RBO_OK) {
            w->t_st = r.t_st;
            w->t_en = r.t_en;
            w->q_st = r.q_st;
            w->q_en = r.q_en;
            w->first_op = (uint32_t)lead_w;
            w->n_ops = (uint32_t)rbo_words_of(r.cigar, r.n_cigar);
            w->nmatch = (uint32_t)r.nmatch;
            w->aln_len = (uint32_t)r.aln_len;
        }
        rbo_rec_free(&r);
    }
    return 0;
}

typedef struct {
    rbo_hit_row *rows;
    size_t n_rows, cap_rows;
    uint32_t *ops;
    size_t n_ops, cap_ops;
} hit_buf;
static void hb_push(hit_buf *b, const rbo_hit_row *row, const rbo_cig *cig, size_t n_cig) {
    const size_t n = rbo_cig_to_words(cig, n_cig, NULL);
    if (b->n_rows == b->cap_rows) {
        b->cap_rows = b->cap_rows ? b->cap_rows * 2 : 8;
        b->rows = (rbo_hit_row *)xrealloc(b->rows, b->cap_rows * sizeof(rbo_hit_row));
    }
    if (b->n_ops + n > b->cap_ops) {
        while (b->n_ops + n > b->cap_ops) b->cap_ops = b->cap_ops ? b->cap_ops * 2 : 1024;
        b->ops = (uint32_t *)xrealloc(b->ops, b->cap_ops * sizeof(uint32_t));
    }
    b->rows[b->n_rows] = *row;
    b->rows[b->n_rows].out_off = b->n_ops;
    b->rows[b->n_rows].out_n = (uint32_t)n;
    b->n_rows++;
    if (n) rbo_cig_to_words(cig, n_cig, b->ops + b->n_ops);
    b->n_ops += n;
}
static void row_from_rec(rbo_hit_row *row, const rbo_rec *t) {
    row->t_st = t->t_st;
    row->t_en = t->t_en;
    row->q_st = t->q_st;
    row->q_en = t->q_en;
    row->nmatch = (uint32_t)t->nmatch;
    row->aln_len = (uint32_t)t->aln_len;
}
static int concat_bufs(hit_buf *bufs, uint64_t n, rbo_hit_row **hits, uint64_t *n_hits, uint32_t **out_ops,
                       uint64_t *n_out) {
    size_t tr = 0, to = 0;
    for (uint64_t k = 0; k < n; k++) {
        tr += bufs[k].n_rows;
        to += bufs[k].n_ops;
    }
    rbo_hit_row *H = (rbo_hit_row *)xmalloc(tr * sizeof(rbo_hit_row));
    uint32_t *O = (uint32_t *)xmalloc(to * sizeof(uint32_t));
    size_t r = 0, o = 0;
    for (uint64_t k = 0; k < n; k++) {
        for (size_t j = 0; j < bufs[k].n_rows; j++) {
            H[r] = bufs[k].rows[j];
            H[r].out_off += o;
            r++;
        }
        if (bufs[k].n_ops) memcpy(O + o, bufs[k].ops, bufs[k].n_ops * sizeof(uint32_t));
        o += bufs[k].n_ops;
        free(bufs[k].rows);
        free(bufs[k].ops);
    }
    *hits = H;
    *n_hits = tr;
    *out_ops = O;
    *n_out = to;
    return 0;
}

int rbo_liftover_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                        const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                        const uint32_t *contig, uint64_t n_win, const uint32_t *w_contig, const uint64_t *w_st,
                        const uint64_t *w_en, int policy, int n_threads, rbo_hit_row **hits, uint64_t *n_hits,
                        uint32_t **out_ops, uint64_t *n_out) {
    /* canonical record order: contigs by first appearance (liftover.rs:151), then record order */
    uint64_t *canon = (uint64_t *)xmalloc((n_rec + 1) * sizeof(uint64_t));
    {
        uint32_t maxc = 0;
        for (uint64_t i = 0; i < n_rec; i++)
            if (contig[i] > maxc) maxc = contig[i];
        uint64_t *rank = (uint64_t *)xmalloc(((size_t)maxc + 2) * sizeof(uint64_t));
        uint64_t *cnt = (uint64_t *)xmalloc(((size_t)maxc + 2) * sizeof(uint64_t));
        for (uint32_t c = 0; c <= maxc; c++) rank[c] = UINT64_MAX;
        uint64_t nr = 0;
        for (uint64_t i = 0; i < n_rec; i++)
            if (rank[contig[i]] == UINT64_MAX) rank[contig[i]] = nr++;
        memset(cnt, 0, ((size_t)maxc + 2) * sizeof(uint64_t));
        uint64_t *start = (uint64_t *)xmalloc((nr + 1) * sizeof(uint64_t));
        memset(start, 0, (nr + 1) * sizeof(uint64_t));
        for (uint64_t i = 0; i < n_rec; i++) start[rank[contig[i]] + 1]++;
        for (uint64_t k = 0; k < nr; k++) start[k + 1] += start[k];
        for (uint64_t i = 0; i < n_rec; i++) canon[start[rank[contig[i]]]++] = i;
        free(start);
        free(rank);
        free(cnt);
    }
    hit_buf *bufs = (hit_buf *)xmalloc((n_rec + 1) * sizeof(hit_buf));
    memset(bufs, 0, (n_rec + 1) * sizeof(hit_buf));
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t k = 0; k < (int64_t)n_rec; k++) {
        uint64_t i = canon[k];
        rbo_rec cur;
        rec_from_arrays(&cur, i, ops, op_off, t_st, t_en, q_st, q_en, strand);
        cur.t_name[0] = 0;
        if (rbo_aligned_pairs(&cur) != RBO_OK) { /* reference panics; flat API: record yields no rows */
            rbo_rec_free(&cur);
            continue;
        }
        for (uint64_t g = 0; g < n_win; g++) {
            if (w_contig[g] != contig[i]) continue;
            if (!(cur.t_en > w_st[g] && cur.t_st < w_en[g])) continue; /* paf.rs:622-627 */
            rbo_region rgn = {cur.t_name, w_st[g], w_en[g], (char *)"w"};
            rbo_rec t;
            int st = rbo_trim_paf_rec_to_rgn(&rgn, &cur, policy, &t);
            rbo_hit_row row;
            memset(&row, 0, sizeof row);
            row.rec = (uint32_t)i;
            row.win = (uint32_t)g;
            row.status = (uint32_t)st;
            if (st == RBO_OK) {
                row.flags = (cur.t_st > w_st[g] && cur.t_en < w_en[g]) ? 1u : 0u;
                row_from_rec(&row, &t);
                hb_push(&bufs[k], &row, t.cigar, t.n_cigar);
                rbo_rec_free(&t);
            } else {
                hb_push(&bufs[k], &row, NULL, 0);
            }
        }
        rbo_rec_free(&cur);
    }
    free(canon);
    int rc = concat_bufs(bufs, n_rec, hits, n_hits, out_ops, n_out);
    free(bufs);
    return rc;
}

int rbo_break_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                     const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand,
                     uint32_t max_size, int policy, int n_threads, rbo_hit_row **hits, uint64_t *n_hits,
                     uint32_t **out_ops, uint64_t *n_out) {
    hit_buf *bufs = (hit_buf *)xmalloc((n_rec + 1) * sizeof(hit_buf));
    memset(bufs, 0, (n_rec + 1) * sizeof(hit_buf));
    (void)n_threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
#endif
    for (int64_t i = 0; i < (int64_t)n_rec; i++) {
        rbo_rec cur;
        rec_from_arrays(&cur, (uint64_t)i, ops, op_off, t_st, t_en, q_st, q_en, strand);
        if (rbo_aligned_pairs(&cur) != RBO_OK) {
            rbo_rec_free(&cur);
            continue;
        }
        rbo_paf out = {0};
        piece_t *pieces = NULL;
        size_t np = 0;
        break_impl(&cur, max_size, policy, &out, &pieces, &np);
        size_t oi = 0;
        for (size_t p = 0; p < np; p++) {
            rbo_hit_row row;
            memset(&row, 0, sizeof row);
            row.rec = (uint32_t)i;
            row.win = (uint32_t)p;
            row.status = (uint32_t)pieces[p].status;
            if (pieces[p].status == RBO_OK) {
                row_from_rec(&row, &out.recs[oi]);
                hb_push(&bufs[i], &row, out.recs[oi].cigar, out.recs[oi].n_cigar);
                oi++;
            } else {
                hb_push(&bufs[i], &row, NULL, 0);
            }
        }
        free(pieces);
        rbo_paf_free(&out);
        rbo_rec_free(&cur);
    }
    int rc = concat_bufs(bufs, n_rec, hits, n_hits, out_ops, n_out);
    free(bufs);
    return rc;
}

int rbo_overlap_split_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st,
                             const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en,
                             const uint8_t *strand, uint64_t n_pairs, const uint32_t *left, const uint32_t *right,
                             int ms, int ds, int is, int policy, rbo_pair_row *rows, uint32_t **out_ops,
                             uint64_t *n_out) {
    (void)n_rec;
    uint32_t *O = NULL;
    size_t no = 0, cap = 0;
    for (uint64_t p = 0; p < n_pairs; p++) {
        rbo_rec L, R;
        rec_from_arrays(&L, left[p], ops, op_off, t_st, t_en, q_st, q_en, strand);
        rec_from_arrays(&R, right[p], ops, op_off, t_st, t_en, q_st, q_en, strand);
        rbo_pair_row *w = &rows[p];
        memset(w, 0, sizeof(*w));
        int rc = rbo_aligned_pairs(&L);
        if (!rc) rc = rbo_aligned_pairs(&R);
        int sc = 0;
        if (!rc) rc = rbo_trim_overlapping_pafs(&L, &R, ms, ds, is, policy, &w->split_idx, &sc);
        w->split_score = sc;
        w->status = (uint32_t)rc;
        if (!rc) {
            const rbo_rec *rr[2] = {&L, &R};
            for (int s = 0; s < 2; s++) {
                w->t_st[s] = rr[s]->t_st;
                w->t_en[s] = rr[s]->t_en;
                w->q_st[s] = rr[s]->q_st;
                w->q_en[s] = rr[s]->q_en;
                w->nmatch[s] = (uint32_t)rr[s]->nmatch;
                w->aln_len[s] = (uint32_t)rr[s]->aln_len;
                w->out_off[s] = no;
                const size_t nw = rbo_cig_to_words(rr[s]->cigar, rr[s]->n_cigar, NULL);
                w->out_n[s] = (uint32_t)nw;
                if (no + nw > cap) {
                    while (no + nw > cap) cap = cap ? cap * 2 : 1024;
                    O = (uint32_t *)xrealloc(O, cap * sizeof(uint32_t));
                }
                rbo_cig_to_words(rr[s]->cigar, rr[s]->n_cigar, O + no);
                no += nw;
            }
        }
        rbo_rec_free(&L);
        rbo_rec_free(&R);
    }
    *out_ops = O ? O : (uint32_t *)xmalloc(4);
    *n_out = no;
    return 0;
}

int rbo_swap_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint8_t *strand,
                    uint32_t *out_ops) {
    for (uint64_t i = 0; i < n_rec; i++) {
        rbo_rec r, f;
        uint64_t z = 0;
        rbo_rec_init(&r);
        cig_from_words(&r, ops + op_off[i], (size_t)(op_off[i + 1] - op_off[i]));
        r.strand = (char)strand[i];
        (void)z;
        rbo_paf_swap_query_and_target(&r, &f);
        rbo_cig_to_words(f.cigar, f.n_cigar, out_ops + op_off[i]); /* (as many words as came in) */
        rbo_rec_free(&r);
        rbo_rec_free(&f);
    }
    return 0;
}
