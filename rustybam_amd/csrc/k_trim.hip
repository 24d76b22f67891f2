// k_trim.hip -- trim-paf pair kernel for gfx950: trim_overlapping_pafs (trim_overlap.rs:36-86) followed by
// truncate_record_by_query on both records (paf.rs:785-823).
//
// The reference scores every overlapped query base with a binary search into per-base arrays
// (score_of_qpos, trim_overlap.rs:6-19).  In op space the per-base score of a record is piecewise
// constant over its ops, except for the last unit of an op that is followed by non-query ops (the
// duplicate-value run in qpos_aln): rb_qstream yields those runs in increasing query position for either
// strand, the two records' runs are merged, and the first arg-max of prefix(left) + suffix(right)
// (trim_overlap.rs:69-76) is found from run ends.  One thread per pair, serial: the pairs of one pass are
// independent (one pair per query name and pass, paf.rs:264-284); the pass/recursion driver stays on the
// host.  Fully general (all op codes, both binary-search policies).
#include "rb_serial.h"

struct rb_trim_params {
    uint64_t n_pairs;
    const uint32_t *ops;
    const uint64_t *op_off;
    const uint8_t *strand;
    const rb_norm_row *norm;
    const uint32_t *left, *right;
    const uint64_t *pair_out_off; // [n_pairs] first output op of each pair (room for n_left + n_right ops)
    int match_score, diff_score, indel_score;
    int policy;
    rb_pair_row *rows;
    uint32_t *out_ops;
};

struct rb_qstream {
    rb_sview v;
    int policy;
    uint64_t N;
    int32_t ms, ds, is;
    int32_t i;       // current query-consuming op; -1 / n when exhausted
    uint64_t U;      // unit index of op i's first unit
    uint64_t lo, hi; // query positions covered by op i
    uint64_t pos;    // next position to yield

    __device__ int32_t score_of(uint32_t opc) const { // trim_overlap.rs:14-18
        return opc == RB_OP_EQ ? ms : ((opc == RB_OP_I || opc == RB_OP_D) ? -is : -ds);
    }
    __device__ bool is_q(int32_t j) const { return rb_len(v.ops[j]) != 0 && rb_s_qry(rb_opc(v.ops[j])); }

    // op code of the unit that qpos_to_idx returns for the LAST unit (in op order) of op i
    __device__ uint32_t special_type() const {
        const uint32_t own = rb_opc(v.ops[i]);
        uint64_t runU = 0;
        uint32_t last = own;
        for (uint32_t j = (uint32_t)i + 1; j < v.n; j++) {
            const uint32_t opc = rb_opc(v.ops[j]), len = rb_len(v.ops[j]);
            if (len == 0) continue;
            if (rb_s_qry(opc)) break;
            runU += len;
            last = opc;
        }
        if (runU == 0) return own;
        if (policy != RB_BSEARCH_LEGACY) return last; // modern: last equal element
        const uint64_t klo = U + rb_len(v.ops[i]) - 1;
        const uint64_t k = rb_s_legacy_probe(N, klo, klo + runU);
        if (k == klo) return own;
        uint64_t u = klo + 1;
        for (uint32_t j = (uint32_t)i + 1; j < v.n; j++) {
            const uint32_t opc = rb_opc(v.ops[j]), len = rb_len(v.ops[j]);
            if (len == 0) continue;
            if (k < u + len) return opc;
            u += len;
        }
        return last;
    }

    __device__ void next_op() { // move to the op with the next higher query positions
        const uint64_t base = hi + 1;
        if (!v.minus) {
            U += rb_len(v.ops[i]);
            i++;
            while (i < (int32_t)v.n && !is_q(i)) {
                U += rb_len(v.ops[i]);
                i++;
            }
            if (i >= (int32_t)v.n) return;
        } else {
            i--;
            while (i >= 0 && !is_q(i)) {
                U -= rb_len(v.ops[i]);
                i--;
            }
            if (i < 0) return;
            U -= rb_len(v.ops[i]);
        }
        lo = base;
        hi = base + rb_len(v.ops[i]) - 1;
    }

    __device__ void seek(uint64_t p) {
        if (!v.minus) {
            i = 0;
            U = 0;
            while (i < (int32_t)v.n && !is_q(i)) {
                U += rb_len(v.ops[i]);
                i++;
            }
        } else {
            i = (int32_t)v.n - 1;
            U = N;
            while (i >= 0 && !is_q(i)) {
                U -= rb_len(v.ops[i]);
                i--;
            }
            if (i >= 0) U -= rb_len(v.ops[i]);
        }
        if (i < 0 || i >= (int32_t)v.n) return;
        lo = v.q_st;
        hi = lo + rb_len(v.ops[i]) - 1;
        while (p > hi && i >= 0 && i < (int32_t)v.n) next_op();
        pos = p;
    }

    __device__ bool valid() const { return i >= 0 && i < (int32_t)v.n; }

    // current run of equal scores starting at pos
    __device__ void run(uint64_t *count, int32_t *score) const {
        const uint32_t own = rb_opc(v.ops[i]);
        if (!v.minus) {
            if (pos < hi) {
                *count = hi - pos;
                *score = score_of(own);
            } else {
                *count = 1;
                *score = score_of(special_type());
            }
        } else {
            if (pos == lo) {
                *count = 1;
                *score = score_of(special_type());
            } else {
                *count = hi - pos + 1;
                *score = score_of(own);
            }
        }
    }
    __device__ void advance(uint64_t c) {
        pos += c;
        if (pos > hi) next_op();
    }
};

// truncate_record_by_query (paf.rs:785-823).  Writes the clipped cigar to `out`, fills side s of the row.
__device__ uint32_t rb_clip_by_query(const rb_sview &v, uint64_t N, uint64_t new_q_st, uint64_t new_q_en, int policy, uint32_t *out,
                                     rb_pair_row *row, int s, uint64_t out_base) {
    if (!(new_q_st >= v.q_st) || !(new_q_en <= v.q_en) || new_q_en == 0) return RB_ST_PANIC_ASSERT; // :787-788
    uint64_t klo, khi, ks, ke;
    if (rb_s_wrapped_q(v)) { // qpos_aln is not sorted: replay the binary search itself
        if (!rb_s_bsearch_q(v, N, new_q_st, policy, &ks) || !rb_s_bsearch_q(v, N, new_q_en - 1, policy, &ke)) return RB_ST_PANIC_NOTFOUND;
    } else {
        if (!rb_s_qrange(v, new_q_st, &klo, &khi)) return RB_ST_PANIC_NOTFOUND;
        ks = policy == RB_BSEARCH_LEGACY ? rb_s_legacy_probe(N, klo, khi) : khi;
        if (!rb_s_qrange(v, new_q_en - 1, &klo, &khi)) return RB_ST_PANIC_NOTFOUND;
        ke = policy == RB_BSEARCH_LEGACY ? rb_s_legacy_probe(N, klo, khi) : khi;
    }
    // qpos_to_idx_match (paf.rs:576-590): search_right flips on '-'
    uint64_t aln_st = !v.minus ? rb_s_match_ge(v, ks, N) : rb_s_match_le(v, ks);
    uint64_t aln_en = !v.minus ? rb_s_match_le(v, ke) : rb_s_match_ge(v, ke, N);
    if (aln_st >= N || aln_en >= N) return RB_ST_PANIC_NOTFOUND; // index past the end of qpos_aln (:795-796)
    uint32_t oc;
    uint64_t tp, qp_st, qp_en;
    rb_s_unit(v, aln_st, &oc, &tp, &qp_st);
    rb_s_unit(v, aln_en, &oc, &tp, &qp_en);
    uint64_t nq_st = qp_st, nq_en = qp_en + 1; // :795-796
    if (aln_st > aln_en) { // :799-801
        const uint64_t t = aln_st;
        aln_st = aln_en;
        aln_en = t;
    }
    uint64_t t0, t1, qd;
    rb_s_unit(v, aln_st, &oc, &t0, &qd);
    rb_s_unit(v, aln_en, &oc, &t1, &qd);
    uint64_t nt_st = t0, nt_en = t1 + 1; // :802-803
    uint64_t sums[4];
    const uint32_t cnt = rb_s_emit_units(v, aln_st, aln_en, out, sums); // :807-808
    uint32_t first = 0, count = 0, nm = 0, al = 0;
    const uint32_t st = rb_s_strip_indels(out, cnt, v.minus, &nt_st, &nt_en, &nq_st, &nq_en, &first, &count, &nm, &al); // :819-822
    if (st != RB_ST_OK) return st;
    row->t_st[s] = nt_st;
    row->t_en[s] = nt_en;
    row->q_st[s] = nq_st;
    row->q_en[s] = nq_en;
    row->nmatch[s] = nm;
    row->aln_len[s] = al;
    row->out_off[s] = out_base + first;
    row->out_n[s] = count;
    return RB_ST_OK;
}

__global__ __launch_bounds__(64) void rb_k_overlap_split(rb_trim_params p) {
    const uint64_t pi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= p.n_pairs) return;
    rb_pair_row w;
    w.split_idx = 0;
    w.split_score = 0;
    w.status = RB_ST_OK;
    w._pad = 0;
    for (int s = 0; s < 2; s++) {
        w.t_st[s] = w.t_en[s] = w.q_st[s] = w.q_en[s] = 0;
        w.nmatch[s] = w.aln_len[s] = 0;
        w.out_off[s] = 0;
        w.out_n[s] = 0;
    }
    const uint32_t rl = p.left[pi], rr = p.right[pi];
    const rb_norm_row *nl = &p.norm[rl], *nr = &p.norm[rr];
    if (nl->status != RB_ST_OK || nr->status != RB_ST_OK) { // aligned_pairs() panics (paf.rs:273-274, :782)
        w.status = nl->status != RB_ST_OK ? nl->status : nr->status;
        p.rows[pi] = w;
        return;
    }
    rb_sview L, R;
    L.ops = p.ops + p.op_off[rl] + nl->first_op;
    L.n = nl->n_ops;
    L.t_st = nl->t_st, L.t_en = nl->t_en, L.q_st = nl->q_st, L.q_en = nl->q_en;
    L.minus = p.strand[rl] == (uint8_t)'-';
    R.ops = p.ops + p.op_off[rr] + nr->first_op;
    R.n = nr->n_ops;
    R.t_st = nr->t_st, R.t_en = nr->t_en, R.q_st = nr->q_st, R.q_en = nr->q_en;
    R.minus = p.strand[rr] == (uint8_t)'-';
    const uint64_t NL = nl->aln_len, NR = nr->aln_len; // total units of each record

    const uint64_t st_ovl = L.q_st > R.q_st ? L.q_st : R.q_st; // trim_overlap.rs:43-44
    const uint64_t en_ovl = L.q_en < R.q_en ? L.q_en : R.q_en;
    const uint64_t n = en_ovl > st_ovl ? en_ovl - st_ovl : 0;

    rb_qstream A, B;
    A.v = L, A.policy = p.policy, A.N = NL, A.ms = p.match_score, A.ds = p.diff_score, A.is = p.indel_score;
    B.v = R, B.policy = p.policy, B.N = NR, B.ms = p.match_score, B.ds = p.diff_score, B.is = p.indel_score;
    int32_t best = 0;
    uint64_t best_idx = 0;
    if (n > 0 && (rb_s_wrapped_q(L) || rb_s_wrapped_q(R))) {
        // a qpos_aln that is not sorted (see rb_s_wrapped_q): score_of_qpos (trim_overlap.rs:6-19) base by base with the
        // binary search replayed; an Err is the .unwrap() panic.  Slow, and only for this corner.
        auto score = [&](const rb_sview &v, uint64_t N, uint64_t pos, int32_t *out) -> bool {
            uint64_t idx;
            if (!rb_s_bsearch_q(v, N, pos, p.policy, &idx)) return false;
            uint32_t oc;
            uint64_t tp, qp;
            rb_s_unit(v, idx, &oc, &tp, &qp);
            *out = oc == RB_OP_EQ ? p.match_score : ((oc == RB_OP_I || oc == RB_OP_D) ? -p.indel_score : -p.diff_score);
            return true;
        };
        bool ok = true;
        int32_t rsum = 0;
        for (uint64_t k = 0; k < n && ok; k++) {
            int32_t ls, rs;
            ok = score(L, NL, st_ovl + k, &ls) && score(R, NR, st_ovl + k, &rs); // (both are looked up, left first: :50-51)
            if (ok) rsum += rs;
        }
        if (!ok) {
            w.status = RB_ST_PANIC_NOTFOUND;
            p.rows[pi] = w;
            return;
        }
        int32_t lpre = 0, rpre = 0;
        for (uint64_t k = 0; k <= n; k++) { // l_score[k] + r_score[k], first strict maximum (initial 0 at 0)
            const int32_t val = lpre + (rsum - rpre);
            if (val > best) {
                best = val;
                best_idx = k;
            }
            if (k < n) {
                int32_t ls, rs;
                score(L, NL, st_ovl + k, &ls);
                score(R, NR, st_ovl + k, &rs);
                lpre += ls;
                rpre += rs;
            }
        }
    } else if (n > 0) {
        // sum of the right record's scores over the overlap (r_score suffix sum at index 0)
        int32_t rsum = 0;
        B.seek(st_ovl);
        for (uint64_t k = 0; k < n && B.valid();) {
            uint64_t c;
            int32_t sc;
            B.run(&c, &sc);
            if (c > n - k) c = n - k;
            rsum += sc * (int32_t)c;
            k += c;
            B.advance(c);
        }
        // l_score prefix + r_score suffix, first strict maximum over idx 0..n (initial max 0 at idx 0)
        if (rsum > best) best = rsum;
        A.seek(st_ovl);
        B.seek(st_ovl);
        int32_t P = 0; // sum over positions < k of (l - r)
        for (uint64_t k = 0; k < n && A.valid() && B.valid();) {
            uint64_t ca, cb;
            int32_t sa, sb;
            A.run(&ca, &sa);
            B.run(&cb, &sb);
            uint64_t c = ca < cb ? ca : cb;
            if (c > n - k) c = n - k;
            const int32_t d = sa - sb;
            if (d > 0) { // the sum rises through the run: its last index is the only candidate
                const int32_t cand = rsum + P + d * (int32_t)c;
                if (cand > best) {
                    best = cand;
                    best_idx = k + c;
                }
            }
            P += d * (int32_t)c;
            k += c;
            A.advance(c);
            B.advance(c);
        }
    }
    w.split_idx = best_idx;
    w.split_score = best;
    const uint64_t split = st_ovl + best_idx;
    const uint64_t ob = p.pair_out_off[pi];
    uint32_t st = rb_clip_by_query(L, NL, L.q_st, split, p.policy, p.out_ops + ob, &w, 0, ob); // trim_overlap.rs:77
    if (st == RB_ST_OK) {
        const uint64_t ob2 = ob + L.n;
        st = rb_clip_by_query(R, NR, split, R.q_en, p.policy, p.out_ops + ob2, &w, 1, ob2); // :78
    }
    w.status = st;
    p.rows[pi] = w;
}

extern "C" hipError_t rb_launch_overlap_split(const rb_trim_params *p, hipStream_t stream) {
    if (p->n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_overlap_split, dim3((unsigned)((p->n_pairs + 63) / 64)), dim3(64), 0, stream, *p);
    return hipGetLastError();
}
