/* rb_opspace.c -- TEST / MEASUREMENT INFRASTRUCTURE, never linked into the product.
 *
 * The second CPU baseline SURVEY.md 8(d) asks for next to the per-base restatement (rb_oracle.c): liftover in OP SPACE on the host's
 * cores -- what a CPU implementation that does not expand CIGARs to per-base vectors achieves.  It is a port to plain C of the
 * formulation the HIP kernels use: exclusive prefixes R / Q / U over a record's ops, a binary
 * search for the op that holds a window boundary, the reference's duplicate ("last equal element", paf.rs:542, modern Rust) and
 * walk-to-match rules (paf.rs:547-561) on the neighbouring ops, coordinates as in liftover.rs:57-82.  Semantics are NOT taken from
 * here: tests/test_oracle_opspace.py holds it to the per-base oracle, row by row and op by op.
 *
 * Scope: what the streaming kernel's fast path takes -- regular records (only M I D N = X, lengths >= 1, no two adjacent ops of
 * one type, a match op at both ends, coordinates consistent with the CIGAR) and the modern binary-search policy.  Anything else
 * returns RBO_OPSPACE_UNSUPPORTED for the whole call (the caller then has only the per-base number).
 * Output order: canonical (contig first appearance -> record -> window in BED order), as rbo_liftover_arrays. */
#include "rb_oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define RBO_OPSPACE_UNSUPPORTED 7

static inline uint32_t o_len(uint32_t v) { return v >> 4; }
static inline uint32_t o_opc(uint32_t v) { return v & 15u; }
static inline uint32_t o_rl(uint32_t v) { return o_opc(v) == 1u ? 0u : o_len(v); }                       /* ref: not I   */
static inline uint32_t o_ql(uint32_t v) { return (o_opc(v) == 2u || o_opc(v) == 3u) ? 0u : o_len(v); }   /* query: not D, N */
static inline int o_ism(uint32_t v) { return (0x181u >> o_opc(v)) & 1u; }                                /* M = X */
static inline int o_regular(uint32_t v) { return (0x18Fu >> o_opc(v)) & 1u; }                            /* M I D N = X */

typedef struct { int st; uint32_t op, part, R, Q, U; } bres; /* st: 1 ok, 2 none */

/* boundary D (= reference offset of the boundary base + 1) of a record with n ops and exclusive prefixes pR/pQ/pU (n + 1 entries) */
static bres resolve(const uint32_t *ops, uint32_t n, const uint32_t *pR, const uint32_t *pQ, const uint32_t *pU, uint32_t D, int is_start) {
    bres o;
    memset(&o, 0, sizeof o);
    const uint32_t Rtot = pR[n];
    if (D == Rtot) { /* the record's last base; the last op is match-type */
        o.st = 1, o.op = n - 1;
        if (is_start) o.part = 1, o.R = Rtot - 1, o.Q = pQ[n] - 1, o.U = pU[n] - 1;
        else o.part = o_len(ops[n - 1]), o.R = Rtot, o.Q = pQ[n], o.U = pU[n];
        return o;
    }
    /* f: the op with pR[f] <= D < pR[f + 1] (first op whose inclusive prefix passes D) */
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (pR[mid + 1] > D) hi = mid; else lo = mid + 1;
    }
    int64_t fi = lo;
    const uint32_t fv = ops[fi], off = D - pR[fi];
    if (is_start) {
        int64_t X;
        if (off > 0) {
            if (o_ism(fv)) { o.st = 1, o.op = (uint32_t)fi, o.part = o_len(fv) - (off - 1), o.R = pR[fi] + off - 1, o.Q = pQ[fi] + off - 1, o.U = pU[fi] + off - 1; return o; }
            X = fi + 1;
        } else {
            if (fi > 0 && o_ism(ops[fi - 1])) { o.st = 1, o.op = (uint32_t)(fi - 1), o.part = 1, o.R = pR[fi] - 1, o.Q = pQ[fi] - 1, o.U = pU[fi] - 1; return o; }
            X = fi;
        }
        for (int64_t i = fi; i < (int64_t)n; i++) /* paf.rs:551-553 */
            if (i >= X && o_ism(ops[i])) { o.st = 1, o.op = (uint32_t)i, o.part = o_len(ops[i]), o.R = pR[i], o.Q = pQ[i], o.U = pU[i]; return o; }
        o.st = 2;
        return o;
    }
    int64_t Y;
    if (off > 0) {
        if (o_ism(fv)) { o.st = 1, o.op = (uint32_t)fi, o.part = off, o.R = D, o.Q = pQ[fi] + off, o.U = pU[fi] + off; return o; }
        Y = fi - 1;
    } else {
        if (fi > 0 && o_ism(ops[fi - 1])) { o.st = 1, o.op = (uint32_t)(fi - 1), o.part = o_len(ops[fi - 1]), o.R = pR[fi], o.Q = pQ[fi], o.U = pU[fi]; return o; }
        Y = fi - 2;
    }
    for (int64_t i = fi - 1; i >= 0; i--) /* paf.rs:555-557: the prefixes at the END of op i */
        if (i <= Y && o_ism(ops[i])) { o.st = 1, o.op = (uint32_t)i, o.part = o_len(ops[i]), o.R = pR[i + 1], o.Q = pQ[i + 1], o.U = pU[i + 1]; return o; }
    o.st = 2;
    return o;
}

int rbo_liftover_opspace_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st, const uint64_t *t_en,
                                const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand, const uint32_t *contig, uint64_t n_win,
                                const uint32_t *w_contig, const uint64_t *w_st, const uint64_t *w_en, int n_threads, rbo_hit_row **hits,
                                uint64_t *n_hits, uint32_t **out_ops, uint64_t *n_out) {
    *hits = NULL, *out_ops = NULL, *n_hits = 0, *n_out = 0;
    /* canonical record order: contigs by first appearance, records in file order inside a contig */
    uint32_t max_c = 0;
    for (uint64_t r = 0; r < n_rec; r++) max_c = contig[r] > max_c ? contig[r] : max_c;
    for (uint64_t w = 0; w < n_win; w++) max_c = w_contig[w] > max_c ? w_contig[w] : max_c;
    const uint64_t nc = (uint64_t)max_c + 1;
    uint64_t *rank = malloc(nc * 8), *cnt = calloc(nc + 1, 8), *order = malloc((n_rec + 1) * 8);
    for (uint64_t c = 0; c < nc; c++) rank[c] = ~0ull;
    uint64_t seen = 0;
    for (uint64_t r = 0; r < n_rec; r++)
        if (rank[contig[r]] == ~0ull) rank[contig[r]] = seen++;
    for (uint64_t r = 0; r < n_rec; r++) cnt[rank[contig[r]] + 1]++;
    for (uint64_t c = 0; c < seen; c++) cnt[c + 1] += cnt[c];
    for (uint64_t r = 0; r < n_rec; r++) order[cnt[rank[contig[r]]]++] = r;
    /* windows grouped by contig, BED order kept */
    uint64_t *wcnt = calloc(nc + 1, 8), *wlist = malloc((n_win + 1) * 8);
    for (uint64_t w = 0; w < n_win; w++) wcnt[w_contig[w] + 1]++;
    for (uint64_t c = 0; c < nc; c++) wcnt[c + 1] += wcnt[c];
    {
        uint64_t *cur = malloc(nc * 8);
        memcpy(cur, wcnt, nc * 8);
        for (uint64_t w = 0; w < n_win; w++) wlist[cur[w_contig[w]]++] = w;
        free(cur);
    }
    /* pass 1: per record (in canonical order) its hits and emitted ops; pass 2 after the prefix sums: the rows */
    uint64_t *hcount = calloc(n_rec + 1, 8), *ocount = calloc(n_rec + 1, 8);
    int bad = 0;
    if (n_threads < 1) n_threads = 1;
    for (int pass = 0; pass < 2 && !bad; pass++) {
        if (pass == 1) {
            uint64_t h = 0, o = 0;
            for (uint64_t k = 0; k < n_rec; k++) {
                const uint64_t a = hcount[k], b = ocount[k];
                hcount[k] = h, ocount[k] = o;
                h += a, o += b;
            }
            hcount[n_rec] = h, ocount[n_rec] = o;
            *hits = calloc(h + 1, sizeof(rbo_hit_row));
            *out_ops = malloc((o + 1) * 4);
            *n_hits = h, *n_out = o;
        }
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads)
        for (uint64_t k = 0; k < n_rec; k++) {
            if (bad) continue;
            const uint64_t r = order[k];
            const uint32_t *c = ops + op_off[r];
            const uint64_t n64 = op_off[r + 1] - op_off[r];
            const uint32_t n = (uint32_t)n64;
            int ok = n64 > 0 && n64 < 0xFFFFFFFFull && o_ism(c[0]) && o_ism(c[n - 1]);
            uint32_t *pR = malloc(((size_t)n + 1) * 12), *pQ = pR + n + 1, *pU = pQ + n + 1;
            uint64_t R = 0, Q = 0, U = 0;
            for (uint32_t i = 0; i < n && ok; i++) {
                const uint32_t v = c[i];
                ok = o_regular(v) && o_len(v) >= 1 && (i == 0 || o_opc(v) != o_opc(c[i - 1]));
                pR[i] = (uint32_t)R, pQ[i] = (uint32_t)Q, pU[i] = (uint32_t)U;
                R += o_rl(v), Q += o_ql(v), U += o_len(v);
            }
            ok = ok && U <= 0xFFFFFFFFull && t_en[r] >= t_st[r] && q_en[r] >= q_st[r] && R == t_en[r] - t_st[r] && Q == q_en[r] - q_st[r];
            if (!ok) {
                bad = 1;
                free(pR);
                continue;
            }
            pR[n] = (uint32_t)R, pQ[n] = (uint32_t)Q, pU[n] = (uint32_t)U;
            const int minus = strand[r] == '-';
            uint64_t nh = 0, no = 0;
            for (uint64_t wi = wcnt[contig[r]]; wi < wcnt[contig[r] + 1]; wi++) {
                const uint64_t w = wlist[wi];
                if (!(t_en[r] > w_st[w] && t_st[r] < w_en[w])) continue; /* paf.rs:622-627 */
                rbo_hit_row row;
                memset(&row, 0, sizeof row);
                row.rec = (uint32_t)r, row.win = (uint32_t)w;
                uint32_t a_op = 0, cnt_ops = 0, pa = 0, pb = 0;
                if (t_st[r] > w_st[w] && t_en[r] < w_en[w]) { /* liftover.rs:23-25 */
                    row.flags = 1, row.t_st = t_st[r], row.t_en = t_en[r], row.q_st = q_st[r], row.q_en = q_en[r];
                    row.nmatch = (uint32_t)(R + Q - U), row.aln_len = (uint32_t)U;
                    cnt_ops = n;
                } else {
                    const uint32_t Ds = (uint32_t)((w_st[w] > t_st[r] ? w_st[w] : t_st[r]) - t_st[r]) + 1u; /* liftover.rs:28 */
                    const uint32_t De = (uint32_t)((w_en[w] < t_en[r] ? w_en[w] : t_en[r]) - t_st[r]);      /* :38-40 */
                    const bres A = resolve(c, n, pR, pQ, pU, Ds, 1), B = resolve(c, n, pR, pQ, pU, De, 0);
                    if (A.st != 1 || B.st != 1 || A.U >= B.U) {
                        row.status = 1; /* RBO_ST_NONE_INDEL: liftover.rs:52-54 */
                    } else {
                        a_op = A.op, cnt_ops = B.op - A.op + 1, pa = A.part, pb = B.part;
                        row.t_st = t_st[r] + A.R, row.t_en = t_st[r] + B.R; /* liftover.rs:57-60, :77-82 */
                        if (!minus) row.q_st = q_st[r] + A.Q, row.q_en = q_st[r] + B.Q;
                        else row.q_st = q_en[r] - B.Q, row.q_en = q_en[r] - A.Q;
                        row.aln_len = B.U - A.U;
                        row.nmatch = (B.R + B.Q - B.U) - (A.R + A.Q - A.U);
                    }
                }
                if (pass == 1) {
                    row.out_n = row.status ? 0 : cnt_ops;
                    row.out_off = ocount[k] + no;
                    if (!row.status) {
                        uint32_t *dst = *out_ops + row.out_off;
                        memcpy(dst, c + a_op, (size_t)cnt_ops * 4);
                        if (!row.flags) {
                            if (cnt_ops == 1) dst[0] = ((pa + pb - o_len(dst[0])) << 4) | o_opc(dst[0]);
                            else dst[0] = (pa << 4) | o_opc(dst[0]), dst[cnt_ops - 1] = (pb << 4) | o_opc(dst[cnt_ops - 1]);
                        }
                    }
                    (*hits)[hcount[k] + nh] = row;
                }
                nh++;
                no += row.status ? 0 : cnt_ops;
            }
            if (pass == 0) hcount[k] = nh, ocount[k] = no;
            free(pR);
        }
    }
    free(rank), free(cnt), free(order), free(wcnt), free(wlist), free(hcount), free(ocount);
    if (bad) {
        free(*hits), free(*out_ops);
        *hits = NULL, *out_ops = NULL, *n_hits = 0, *n_out = 0;
        return RBO_OPSPACE_UNSUPPORTED;
    }
    return 0;
}

/* ---- break-paf (liftover.rs:182-226) in op space: the windows of a record are the stretches of target between its indels longer than
 * max_size -- [pre_tpos, cur_tpos) wherever cur_tpos > pre_tpos (:190-201, :213-224) --, each clipped exactly as a BED window is
 * (trim_paf_rec_to_rgn, :17-105; a piece never strictly contains its record, so the clone of :23-25 does not occur).  Rows in record
 * order, then piece order; win = the piece's ordinal among the record's candidate windows, as rbo_break_arrays.  Same scope as above
 * (regular records, modern policy); held to the per-base oracle by tests/test_oracle_opspace.py. */
int rbo_break_opspace_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *op_off, const uint64_t *t_st, const uint64_t *t_en,
                             const uint64_t *q_st, const uint64_t *q_en, const uint8_t *strand, uint32_t max_size, int n_threads,
                             rbo_hit_row **hits, uint64_t *n_hits, uint32_t **out_ops, uint64_t *n_out) {
    *hits = NULL, *out_ops = NULL, *n_hits = 0, *n_out = 0;
    uint64_t *hcount = calloc(n_rec + 1, 8), *ocount = calloc(n_rec + 1, 8);
    int bad = 0;
    if (n_threads < 1) n_threads = 1;
    for (int pass = 0; pass < 2 && !bad; pass++) {
        if (pass == 1) {
            uint64_t h = 0, o = 0;
            for (uint64_t k = 0; k < n_rec; k++) {
                const uint64_t a = hcount[k], b = ocount[k];
                hcount[k] = h, ocount[k] = o;
                h += a, o += b;
            }
            *hits = calloc(h + 1, sizeof(rbo_hit_row));
            *out_ops = malloc((o + 1) * 4);
            *n_hits = h, *n_out = o;
        }
#pragma omp parallel for schedule(dynamic, 64) num_threads(n_threads)
        for (uint64_t r = 0; r < n_rec; r++) {
            if (bad) continue;
            const uint32_t *c = ops + op_off[r];
            const uint64_t n64 = op_off[r + 1] - op_off[r];
            const uint32_t n = (uint32_t)n64;
            int ok = n64 > 0 && n64 < 0xFFFFFFFFull && o_ism(c[0]) && o_ism(c[n - 1]);
            uint32_t *pR = malloc(((size_t)n + 1) * 12), *pQ = pR + n + 1, *pU = pQ + n + 1;
            uint64_t R = 0, Q = 0, U = 0;
            for (uint32_t i = 0; i < n && ok; i++) {
                const uint32_t v = c[i];
                ok = o_regular(v) && o_len(v) >= 1 && (i == 0 || o_opc(v) != o_opc(c[i - 1]));
                pR[i] = (uint32_t)R, pQ[i] = (uint32_t)Q, pU[i] = (uint32_t)U;
                R += o_rl(v), Q += o_ql(v), U += o_len(v);
            }
            ok = ok && U <= 0xFFFFFFFFull && t_en[r] >= t_st[r] && q_en[r] >= q_st[r] && R == t_en[r] - t_st[r] && Q == q_en[r] - q_st[r];
            if (!ok) {
                bad = 1;
                free(pR);
                continue;
            }
            pR[n] = (uint32_t)R, pQ[n] = (uint32_t)Q, pU[n] = (uint32_t)U;
            const int minus = strand[r] == '-';
            uint64_t nh = 0, no = 0;
            uint64_t pre = 0; /* pre_tpos - t_st; cur_tpos - t_st = pR[i] */
            for (uint32_t i = 0; i <= n; i++) {
                const int last = i == n;
                const uint32_t v = last ? 0u : c[i];
                const int big = !last && (o_opc(v) == 1u || o_opc(v) == 2u) && o_len(v) > max_size; /* liftover.rs:188: Del | Ins */
                if (!big && !last) continue;
                const uint64_t cur = pR[i];
                if (cur > pre) { /* :191, :213 */
                    rbo_hit_row row;
                    memset(&row, 0, sizeof row);
                    row.rec = (uint32_t)r, row.win = (uint32_t)nh;
                    const uint64_t wst = t_st[r] + pre, wen = t_st[r] + cur;
                    const uint32_t Ds = (uint32_t)((wst > t_st[r] ? wst : t_st[r]) - t_st[r]) + 1u; /* liftover.rs:28 */
                    const uint32_t De = (uint32_t)((wen < t_en[r] ? wen : t_en[r]) - t_st[r]);      /* :38-40 */
                    const bres A = resolve(c, n, pR, pQ, pU, Ds, 1), B = resolve(c, n, pR, pQ, pU, De, 0);
                    uint32_t a_op = 0, cnt_ops = 0, pa = 0, pb = 0;
                    if (A.st != 1 || B.st != 1 || A.U >= B.U) {
                        row.status = 1; /* RBO_ST_NONE_INDEL: liftover.rs:52-54 */
                    } else {
                        a_op = A.op, cnt_ops = B.op - A.op + 1, pa = A.part, pb = B.part;
                        row.t_st = t_st[r] + A.R, row.t_en = t_st[r] + B.R;
                        if (!minus) row.q_st = q_st[r] + A.Q, row.q_en = q_st[r] + B.Q;
                        else row.q_st = q_en[r] - B.Q, row.q_en = q_en[r] - A.Q;
                        row.aln_len = B.U - A.U;
                        row.nmatch = (B.R + B.Q - B.U) - (A.R + A.Q - A.U);
                    }
                    if (pass == 1) {
                        row.out_n = row.status ? 0 : cnt_ops;
                        row.out_off = ocount[r] + no;
                        if (!row.status) {
                            uint32_t *dst = *out_ops + row.out_off;
                            memcpy(dst, c + a_op, (size_t)cnt_ops * 4);
                            if (cnt_ops == 1) dst[0] = ((pa + pb - o_len(dst[0])) << 4) | o_opc(dst[0]);
                            else dst[0] = (pa << 4) | o_opc(dst[0]), dst[cnt_ops - 1] = (pb << 4) | o_opc(dst[cnt_ops - 1]);
                        }
                        (*hits)[hcount[r] + nh] = row;
                    }
                    nh++;
                    no += row.status ? 0 : cnt_ops;
                }
                if (last) break;
                pre = cur + o_rl(v); /* :203-206: behind the indel (a deletion's bases are skipped) */
            }
            if (pass == 0) hcount[r] = nh, ocount[r] = no;
            free(pR);
        }
    }
    free(hcount), free(ocount);
    if (bad) {
        free(*hits), free(*out_ops);
        *hits = NULL, *out_ops = NULL, *n_hits = 0, *n_out = 0;
        return RBO_OPSPACE_UNSUPPORTED;
    }
    return 0;
}

/* ---- trim-paf's pair step in op space: trim_overlapping_pafs (trim_overlap.rs:36-86: score_of_qpos :6-19 over the overlapped query
 * bases, the first maximum of prefix(left) + suffix(right) :69-76) followed by truncate_record_by_query on both records
 * (paf.rs:785-823).  A record comes as a VIEW -- where its kept ops begin in ops[], how many they are, and the lengths its first and
 * last op have by now (0: as stored) -- so that a caller can carry the passes of Paf::overlapping_paf_recs over a batch it never
 * rewrites; the answer is the cut as a view again (first kept op, count, new end lengths) with the coordinates, nmatch and aln_len.
 * Per-base semantics in op space: query offset d of a record lies in the query op i with pQ[i] <= d < pQ[i] + len; its
 * unit is pU[i] + (d - pQ[i]), but for the LAST base of the op the last unit in front of the next query op (the D / N run behind it:
 * modern binary search, the last equal element of qpos_aln); a unit scores by the op it lies in.  Written as runs of equal scores over
 * the overlap, left and right merged, the sum evaluated where either changes.  Scope as above: regular records, modern policy; a pair
 * outside it (or one whose cut ends in anything but RB_ST_OK) gets status RBO_PAIR_UNSUPPORTED and is counted in the return value.
 * tests/test_oracle_opspace.py holds it to the per-base oracle. */
typedef struct {
    uint32_t n;
    uint32_t *c, *pR, *pQ, *pU; /* ops with the end lengths applied; exclusive prefixes, n + 1 entries */
    uint64_t t_st, t_en, q_st, q_en;
    int minus;
} pview;
static int pv_load(pview *v, uint32_t *buf, const uint32_t *ops, uint64_t off, uint32_t n, uint32_t fl, uint32_t ll, uint64_t t_st, uint64_t t_en,
                   uint64_t q_st, uint64_t q_en, int minus) {
    v->n = n, v->c = buf, v->pR = buf + n, v->pQ = v->pR + n + 1, v->pU = v->pQ + n + 1;
    v->t_st = t_st, v->t_en = t_en, v->q_st = q_st, v->q_en = q_en, v->minus = minus;
    if (n == 0) return 0;
    memcpy(v->c, ops + off, (size_t)n * 4);
    if (fl) v->c[0] = (fl << 4) | o_opc(v->c[0]);
    if (ll) v->c[n - 1] = (ll << 4) | o_opc(v->c[n - 1]);
    uint64_t R = 0, Q = 0, U = 0;
    int ok = o_ism(v->c[0]) && o_ism(v->c[n - 1]);
    for (uint32_t i = 0; i < n && ok; i++) {
        const uint32_t w = v->c[i];
        ok = o_regular(w) && o_len(w) >= 1 && (i == 0 || o_opc(w) != o_opc(v->c[i - 1]));
        v->pR[i] = (uint32_t)R, v->pQ[i] = (uint32_t)Q, v->pU[i] = (uint32_t)U;
        R += o_rl(w), Q += o_ql(w), U += o_len(w);
    }
    v->pR[n] = (uint32_t)R, v->pQ[n] = (uint32_t)Q, v->pU[n] = (uint32_t)U;
    return ok && U <= 0xFFFFFFFFull && t_en >= t_st && q_en >= q_st && R == t_en - t_st && Q == q_en - q_st;
}
static inline int pv_score(uint32_t opc, int ms, int ds, int is) { return opc == 7u ? ms : ((opc == 1u || opc == 2u) ? -is : -ds); } /* trim_overlap.rs:14-18 */
static uint32_t pv_qop(const pview *v, uint32_t d) { /* the query op that holds query offset d (d < pQ[n]) */
    uint32_t lo = 0, hi = v->n; /* first i with pQ[i + 1] > d */
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (v->pQ[mid + 1] > d) hi = mid; else lo = mid + 1;
    }
    return lo;
}
/* the op whose type the LAST query base of query op i takes: the last op in front of the next query op */
static uint32_t pv_special(const pview *v, uint32_t i) {
    uint32_t j = i;
    while (j + 1 < v->n && o_ql(v->c[j + 1]) == 0) j++;
    return j;
}
/* scores of the query positions st .. en - 1 in rising position order, as runs: len[] / sc[]; returns how many (cap entries at most) */
static size_t pv_runs(const pview *v, uint64_t st, uint64_t en, int ms, int ds, int is, uint32_t *len, int *sc, size_t cap) {
    size_t m = 0;
    uint64_t left = en - st;
    if (!v->minus) {
        uint32_t d = (uint32_t)(st - v->q_st), i = pv_qop(v, d);
        while (left && i < v->n) {
            const uint32_t l = o_len(v->c[i]), o = d - v->pQ[i];
            const int own = pv_score(o_opc(v->c[i]), ms, ds, is), sp = pv_score(o_opc(v->c[pv_special(v, i)]), ms, ds, is);
            uint64_t body = (uint64_t)(l - 1u) - o; /* offsets o .. l - 2 */
            if (body > left) body = left;
            if (body && m < cap) len[m] = (uint32_t)body, sc[m++] = own;
            left -= body;
            if (left && m < cap) len[m] = 1, sc[m++] = sp, left--;
            i++;
            while (i < v->n && o_ql(v->c[i]) == 0) i++;
            if (i < v->n) d = v->pQ[i];
        }
    } else {
        uint32_t d = (uint32_t)(v->q_en - 1 - st), i = pv_qop(v, d); /* op order runs against the positions */
        for (;;) {
            const uint32_t l = o_len(v->c[i]), o = d - v->pQ[i];
            const int own = pv_score(o_opc(v->c[i]), ms, ds, is), sp = pv_score(o_opc(v->c[pv_special(v, i)]), ms, ds, is);
            uint64_t body = (uint64_t)o + 1u; /* offsets o .. 0 */
            if (o == l - 1u) { /* the op's last base (in op order) comes first */
                if (left && m < cap) len[m] = 1, sc[m++] = sp, left--;
                body = l - 1u;
            }
            if (body > left) body = left;
            if (body && m < cap) len[m] = (uint32_t)body, sc[m++] = own;
            left -= body;
            if (!left || i == 0) break;
            uint32_t j = i;
            while (j > 0 && o_ql(v->c[j - 1]) == 0) j--;
            if (j == 0) break;
            i = j - 1;
            d = v->pQ[i] + o_len(v->c[i]) - 1u;
        }
    }
    return left ? (size_t)-1 : m;
}
typedef struct { uint32_t k, j; } punit; /* unit k in op j */
static int pv_q2u(const pview *v, uint64_t p, punit *u) { /* qpos_to_idx (paf.rs:564-574), modern policy */
    if (p < v->q_st || p >= v->q_en) return 0;
    const uint32_t d = (uint32_t)(v->minus ? v->q_en - 1 - p : p - v->q_st), i = pv_qop(v, d);
    if (d + 1u == v->pQ[i] + o_len(v->c[i])) {
        u->j = pv_special(v, i), u->k = v->pU[u->j + 1] - 1u;
    } else {
        u->j = i, u->k = v->pU[i] + (d - v->pQ[i]);
    }
    return 1;
}
static int pv_match(const pview *v, punit *u, int up) { /* the nearest match-type unit at or above / at or below u (paf.rs:576-590) */
    if (o_ism(v->c[u->j])) return 1;
    if (up) {
        for (uint32_t j = u->j + 1; j < v->n; j++)
            if (o_ism(v->c[j])) { u->j = j, u->k = v->pU[j]; return 1; }
        return 0; /* index past the end: the reference panics */
    }
    for (uint32_t j = u->j; j > 0; j--)
        if (o_ism(v->c[j - 1])) { u->j = j - 1, u->k = v->pU[j] - 1u; return 1; }
    return 0;
}
#define RBO_PAIR_UNSUPPORTED 0xFFFFFFFFu
static int pv_cut(const pview *v, uint64_t a, uint64_t b, rbo_pair_clip_row *row, int s) { /* truncate_record_by_query (paf.rs:785-823) */
    if (a < v->q_st || b > v->q_en || b == 0 || b <= a) return 0;
    punit A, B;
    if (!pv_q2u(v, a, &A) || !pv_q2u(v, b - 1, &B)) return 0;
    if (!pv_match(v, &A, !v->minus) || !pv_match(v, &B, v->minus)) return 0; /* :792-796 */
    const uint32_t oa = A.k - v->pU[A.j], ob = B.k - v->pU[B.j];
    const uint64_t qa = v->minus ? v->q_en - 1 - v->pQ[A.j] - oa : v->q_st + v->pQ[A.j] + oa;
    const uint64_t qb = v->minus ? v->q_en - 1 - v->pQ[B.j] - ob : v->q_st + v->pQ[B.j] + ob;
    const uint64_t nq_st = qa, nq_en = qb + 1;
    if (A.k > B.k) { const punit t = A; A = B; B = t; } /* :799-801 */
    const uint64_t nt_st = v->t_st + v->pR[A.j] + (A.k - v->pU[A.j]), nt_en = v->t_st + v->pR[B.j] + (B.k - v->pU[B.j]) + 1; /* :802-803 */
    const uint32_t cnt = B.j - A.j + 1;
    const uint32_t fl = cnt == 1 ? B.k - A.k + 1u : v->pU[A.j + 1] - A.k, ll = cnt == 1 ? fl : B.k - v->pU[B.j] + 1u;
    /* the kept ops, summed one by one (check_integrity, paf.rs:825-857; nmatch = M + = + X) */
    uint64_t R = 0, Q = 0, M = 0, U = 0;
    for (uint32_t j = A.j; j <= B.j; j++) {
        const uint32_t w = v->c[j], l = j == A.j ? fl : (j == B.j ? ll : o_len(w)), t = (l << 4) | o_opc(w);
        R += o_rl(t), Q += o_ql(t), U += l, M += o_ism(t) ? l : 0;
    }
    if (nt_en < nt_st || nq_en < nq_st || R != nt_en - nt_st || Q != nq_en - nq_st) return 0; /* (the reference panics: not this port's business) */
    row->t_st[s] = nt_st, row->t_en[s] = nt_en, row->q_st[s] = nq_st, row->q_en[s] = nq_en;
    row->nmatch[s] = (uint32_t)M, row->aln_len[s] = (uint32_t)U;
    row->first[s] = A.j, row->count[s] = cnt, row->first_len[s] = fl, row->last_len[s] = ll;
    return 1;
}
int64_t rbo_overlap_split_opspace_arrays(uint64_t n_rec, const uint32_t *ops, const uint64_t *rec_off, const uint32_t *rec_n, const uint32_t *first_len,
                                         const uint32_t *last_len, const uint64_t *t_st, const uint64_t *t_en, const uint64_t *q_st, const uint64_t *q_en,
                                         const uint8_t *strand, uint64_t n_pairs, const uint32_t *left, const uint32_t *right, int match_score,
                                         int diff_score, int indel_score, int n_threads, rbo_pair_clip_row *rows) {
    int64_t unsupported = 0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads) reduction(+ : unsupported)
    for (uint64_t k = 0; k < n_pairs; k++) {
        rbo_pair_clip_row *row = &rows[k];
        memset(row, 0, sizeof *row);
        row->status = RBO_PAIR_UNSUPPORTED;
        const uint32_t rl = left[k], rr = right[k];
        if (rl >= n_rec || rr >= n_rec) { unsupported++; continue; }
        const size_t nl = rec_n[rl], nr = rec_n[rr];
        uint32_t *buf = malloc(((nl + 1) * 4 + (nr + 1) * 4 + 8) * 4 + (nl + nr + 8) * 4 * 8); /* two views, then four run arrays of 2 (nl + nr) + 8 entries */
        pview L, R;
        int ok = pv_load(&L, buf, ops, rec_off[rl], (uint32_t)nl, first_len[rl], last_len[rl], t_st[rl], t_en[rl], q_st[rl], q_en[rl], strand[rl] == '-') &&
                 pv_load(&R, buf + (nl + 1) * 4, ops, rec_off[rr], (uint32_t)nr, first_len[rr], last_len[rr], t_st[rr], t_en[rr], q_st[rr], q_en[rr], strand[rr] == '-');
        const uint64_t st = ok ? (L.q_st > R.q_st ? L.q_st : R.q_st) : 0, en = ok ? (L.q_en < R.q_en ? L.q_en : R.q_en) : 0; /* trim_overlap.rs:43-44 */
        ok = ok && en > st;
        if (ok) {
            const size_t cap = 2 * (nl + nr) + 8;
            uint32_t *ll_ = (uint32_t *)(buf + (nl + 1) * 4 + (nr + 1) * 4 + 8), *rl_ = ll_ + cap;
            int *ls_ = (int *)(rl_ + cap), *rs_ = ls_ + cap;
            /* (room: cap entries of each of the four arrays fit the allocation above) */
            const size_t ml = pv_runs(&L, st, en, match_score, diff_score, indel_score, ll_, ls_, cap);
            const size_t mr = pv_runs(&R, st, en, match_score, diff_score, indel_score, rl_, rs_, cap);
            ok = ml != (size_t)-1 && mr != (size_t)-1;
            if (ok) {
                int64_t rsum = 0;
                for (size_t i = 0; i < mr; i++) rsum += (int64_t)rs_[i] * rl_[i];
                int64_t best = 0, P = 0; /* trim_overlap.rs:69-76: first maximum of l[0 .. k) + r[k .. n), starting from (0, index 0) */
                uint64_t best_idx = 0, pos = 0;
                if (rsum > best) best = rsum;
                size_t i = 0, j = 0;
                uint32_t ci = ml ? ll_[0] : 0, cj = mr ? rl_[0] : 0;
                while (i < ml && j < mr) {
                    const uint32_t c = ci < cj ? ci : cj;
                    P += (int64_t)(ls_[i] - rs_[j]) * c, pos += c;
                    if (rsum + P > best) best = rsum + P, best_idx = pos;
                    ci -= c, cj -= c;
                    if (!ci && ++i < ml) ci = ll_[i];
                    if (!cj && ++j < mr) cj = rl_[j];
                }
                row->split_idx = best_idx, row->split_score = (int32_t)best;
                const uint64_t split = st + best_idx;
                ok = pv_cut(&L, L.q_st, split, row, 0) && pv_cut(&R, split, R.q_en, row, 1); /* trim_overlap.rs:77-78 */
            }
        }
        if (ok) row->status = 0; else unsupported++;
        free(buf);
    }
    return unsupported;
}
