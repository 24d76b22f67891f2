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
