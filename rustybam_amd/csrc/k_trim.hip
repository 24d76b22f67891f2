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
    int only_pending;
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
    if (p.only_pending && p.rows[pi].status != 0x7FFF0001u) return; // (second launch: what the wave-per-pair kernel left)
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

// ---------------------------------------------------------------------------------------------------------------------------
// Wave-per-pair form for the common case: both records REGULAR (only M I D N = X, every length >= 1, no two adjacent ops of one
// type, a match-type op at both ends), at most RB_TW_CAP ops each, modern binary-search policy.  Every question the serial
// kernel answers by walking the ops becomes a binary search in prefix arrays built once per record with wave scans:
// the op words are staged in LDS together with exclusive prefixes built with wave scans:
//   Qc[i] query bases before op i     (a query offset x lies in the query-consuming op with Qc <= x < Qc + len)
//   SP[i] score of the query bases before op i, in op order (score_of_qpos, trim_overlap.rs:6-19: an op's own score for all its
//         bases but the last one in op order, which -- modern policy: last equal element -- takes the score of the last D / N
//         op that follows before the next query op)
//   checkpoints every 16 ops of the units (U), query bases and reference bases before the op
// A wave-uniform search is one ballot over the checkpoints plus the 16 ops of the chunk side by side; the per-lane searches of
// the split candidates are binary searches in Qc.
// The split (trim_overlap.rs:50-76) is the first strict maximum of f(k) = l[0..k) + r[k..n): f is piecewise linear, so it is
// evaluated only where either record's score changes (op starts and the special last base), all candidates in parallel.
// Pairs this kernel does not take are marked RB_ST_PENDING_INTERNAL and done by rb_k_overlap_split afterwards.
#define RB_TW_CAP 768
#define RB_TW_NCP (RB_TW_CAP / 16 + 1)
#define RB_ST_PENDING_INTERNAL 0x7FFF0001u

struct rb_wrec {
    const uint32_t *ops;
    uint32_t n, ncp; // ops; checkpoints in use = ceil(n / 16) + 1 (the last one holds the totals)
    uint64_t t_st, t_en, q_st, q_en;
    bool minus;
    uint32_t N, Qtot, Rtot;
    int32_t Stot;
    uint32_t *w;            // LDS [n + 1]: the op words; w[n] = a zero-length M (ends every D / N run, contains nothing)
    uint32_t *Qc;           // LDS [n + 1]: query bases before op i
    int32_t *SP;            // LDS [n + 1]: score of the query bases before op i, in op order
    uint32_t *cU, *cQ, *cR; // LDS [ncp]: units / query bases / reference bases before op 16 c
};
struct rb_wpos { // an op (i = n: past the end) and the exclusive prefix of the searched quantity at it
    uint32_t i, w, pre;
};

__device__ __forceinline__ int32_t rb_tw_score(uint32_t opc, int32_t ms, int32_t ds, int32_t is) {
    return opc == RB_OP_EQ ? ms : ((opc == RB_OP_I || opc == RB_OP_D) ? -is : -ds);
}

__device__ void rb_tw_stage(rb_wrec &v, int lane, int32_t ms, int32_t ds, int32_t is) {
    // pass 1: the op words into LDS, every load in flight at once (addresses past the record re-read its last op)
    {
        uint32_t t[RB_TW_CAP / 64];
#pragma unroll
        for (int c = 0; c < RB_TW_CAP / 64; c++) {
            const uint32_t i = (uint32_t)c * 64u + (uint32_t)lane;
            t[c] = v.ops[i < v.n ? i : v.n - 1u];
        }
#pragma unroll
        for (int c = 0; c < RB_TW_CAP / 64; c++) {
            const uint32_t i = (uint32_t)c * 64u + (uint32_t)lane;
            if (i < v.n) v.w[i] = t[c];
        }
        if (lane == 0) v.w[v.n] = RB_OP_M; // length 0
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // pass 2: prefixes, 64 ops at a time; every 16th op leaves a checkpoint
    uint32_t Ub = 0, Qb = 0, Rb = 0;
    int32_t Sb = 0;
    for (uint32_t c0 = 0; c0 < v.n; c0 += 64) {
        const uint32_t i = c0 + (uint32_t)lane;
        const bool in = i < v.n;
        const uint32_t w = in ? v.w[i] : 0u;
        const uint32_t opc = rb_opc(w), len = in ? rb_len(w) : 0u;
        const bool q = in && rb_in(RB_QRY_MASK, opc), r = in && rb_in(RB_REF_MASK, opc);
        int32_t m = 0;
        if (q) { // own score for all bases but the last in op order, which takes the score of the last D / N op of the run behind it
            int32_t sp = rb_tw_score(opc, ms, ds, is);
            const int32_t own = sp;
            for (uint32_t j = i + 1;; j++) { // (regular records: short; the sentinel at w[n] stops it)
                const uint32_t oj = rb_opc(v.w[j]);
                if (rb_in(RB_QRY_MASK, oj)) break;
                sp = rb_tw_score(oj, ms, ds, is);
            }
            m = (int32_t)(len - 1u) * own + sp;
        }
        const uint32_t iu = rb_wave_scan_incl(len), iq = rb_wave_scan_incl(q ? len : 0u), ir = rb_wave_scan_incl(r ? len : 0u);
        const int32_t isc = (int32_t)rb_wave_scan_incl((uint32_t)m);
        if (in) {
            v.Qc[i] = Qb + iq - (q ? len : 0u);
            v.SP[i] = Sb + isc - m;
            if ((i & 15u) == 0u) {
                v.cU[i >> 4] = Ub + iu - len;
                v.cQ[i >> 4] = Qb + iq - (q ? len : 0u);
                v.cR[i >> 4] = Rb + ir - (r ? len : 0u);
            }
        }
        Ub += rb_readlane<uint32_t>(iu, 63);
        Qb += rb_readlane<uint32_t>(iq, 63);
        Rb += rb_readlane<uint32_t>(ir, 63);
        Sb += rb_readlane<int>(isc, 63);
    }
    v.ncp = (v.n + 15u) / 16u + 1u;
    if (lane == 0) {
        v.cU[v.ncp - 1] = Ub, v.cQ[v.ncp - 1] = Qb, v.cR[v.ncp - 1] = Rb;
        v.Qc[v.n] = Qb, v.SP[v.n] = Sb;
    }
    v.N = Ub, v.Qtot = Qb, v.Rtot = Rb, v.Stot = Sb;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// units (KIND 0) / reference bases (KIND 1) before op i: the checkpoint of its chunk + the ops of the chunk in front of it
template <int KIND>
__device__ __forceinline__ uint32_t rb_tw_before(const rb_wrec &v, uint32_t i, int lane) {
    const uint32_t c = i >> 4, j = 16u * c + (uint32_t)lane;
    uint32_t x = 0;
    if (lane < 16 && j < i) {
        const uint32_t w = v.w[j];
        x = (KIND == 0 || rb_in(RB_REF_MASK, rb_opc(w))) ? rb_len(w) : 0u;
    }
    return (KIND == 0 ? v.cU[c] : v.cR[c]) + rb_wave_sum_u32(x);
}
// wave-uniform search: the op that holds unit x (BY_UNIT) / query offset x, found by all lanes at once -- one ballot over the
// checkpoints, then the 16 ops of that chunk side by side
template <bool BY_UNIT>
__device__ rb_wpos rb_tw_find(const rb_wrec &v, uint32_t x, int lane) {
    const uint32_t *cp = BY_UNIT ? v.cU : v.cQ;
    const bool le = (uint32_t)lane + 1u < v.ncp && cp[lane] <= x; // (the totals entry is not a chunk)
    const uint32_t c = (uint32_t)__builtin_popcountll(__ballot(le)) - 1u; // cp[0] = 0 <= x
    const uint32_t i = 16u * c + (uint32_t)lane;
    const bool in = lane < 16 && i < v.n;
    const uint32_t w = in ? v.w[i] : 0u;
    const uint32_t len = in ? rb_len(w) : 0u;
    uint32_t pre;
    bool hit;
    if (BY_UNIT) {
        pre = v.cU[c] + rb_wave_scan_incl(len) - len;
        hit = in && pre <= x && x - pre < len;
    } else {
        pre = in ? v.Qc[i] : 0u;
        hit = in && rb_in(RB_QRY_MASK, rb_opc(w)) && pre <= x && x - pre < len;
    }
    const uint64_t mk = __ballot(hit);
    rb_wpos o;
    if (!mk) {
        o.i = v.n, o.w = RB_NULL_OP, o.pre = BY_UNIT ? v.N : v.Qtot;
        return o;
    }
    const int l = __builtin_ctzll(mk);
    o.i = 16u * c + (uint32_t)l;
    o.w = rb_readlane<uint32_t>(w, l), o.pre = rb_readlane<uint32_t>(pre, l);
    return o;
}
// score of the first x query bases in op order, wave-uniform x
__device__ __forceinline__ int64_t rb_tw_W(const rb_wrec &v, uint32_t x, int lane, int32_t ms, int32_t ds, int32_t is) {
    const rb_wpos o = rb_tw_find<false>(v, x, lane);
    if (o.i >= v.n) return v.Stot;
    return (int64_t)v.SP[o.i] + (int64_t)(x - o.pre) * rb_tw_score(rb_opc(o.w), ms, ds, is);
}
// the same for a per-lane x: binary search in Qc (non-query ops share the value of the query op behind them, and that op comes
// later: the last index with Qc <= x is the query op that holds x)
__device__ int64_t rb_tw_W_lane(const rb_wrec &v, uint32_t x, int32_t ms, int32_t ds, int32_t is) {
    if (x >= v.Qtot) return v.Stot;
    uint32_t lo = 0, hi = v.n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v.Qc[mid] <= x) lo = mid; else hi = mid;
    }
    return (int64_t)v.SP[lo] + (int64_t)(x - v.Qc[lo]) * rb_tw_score(rb_opc(v.w[lo]), ms, ds, is);
}
// score of the query positions [q_st, p) in increasing position order
__device__ __forceinline__ int64_t rb_tw_G(const rb_wrec &v, uint64_t p, int lane, int32_t ms, int32_t ds, int32_t is) {
    return !v.minus ? rb_tw_W(v, (uint32_t)(p - v.q_st), lane, ms, ds, is) : (int64_t)v.Stot - rb_tw_W(v, (uint32_t)(v.q_en - p), lane, ms, ds, is);
}
__device__ __forceinline__ int64_t rb_tw_G_lane(const rb_wrec &v, uint64_t p, int32_t ms, int32_t ds, int32_t is) {
    return !v.minus ? rb_tw_W_lane(v, (uint32_t)(p - v.q_st), ms, ds, is) : (int64_t)v.Stot - rb_tw_W_lane(v, (uint32_t)(v.q_en - p), ms, ds, is);
}

// truncate_record_by_query (paf.rs:785-823) on a staged regular record; same results as rb_clip_by_query
__device__ uint32_t rb_tw_clip(const rb_wrec &v, uint64_t new_q_st, uint64_t new_q_en, uint32_t *out, rb_pair_row *row, int s, uint64_t out_base,
                               int lane) {
    if (!(new_q_st >= v.q_st) || !(new_q_en <= v.q_en) || new_q_en == 0) return RB_ST_PANIC_ASSERT; // :787-788
    const uint32_t n = v.n, N = v.N;
    auto qlast = [&](uint64_t p, uint32_t *k) -> bool { // qpos_to_idx, modern policy: the LAST unit whose qpos equals p
        if (p < v.q_st || p >= v.q_en) return false;
        const uint32_t x = (uint32_t)(v.minus ? v.q_en - 1 - p : p - v.q_st);
        const rb_wpos o = rb_tw_find<false>(v, x, lane);
        const uint32_t j = x - o.pre, len = rb_len(o.w);
        uint32_t u = rb_tw_before<0>(v, o.i, lane) + j;
        if (j + 1u == len) // last base of the op: the D / N units behind it repeat its position
            for (uint32_t i2 = o.i + 1; i2 < n && !rb_in(RB_QRY_MASK, rb_opc(v.w[i2])); i2++) u += rb_len(v.w[i2]);
        *k = u;
        return true;
    };
    auto match_ge = [&](uint32_t k) -> uint32_t { // paf.rs:581-583
        const rb_wpos o = rb_tw_find<true>(v, k, lane);
        if (rb_in(RB_MATCH_MASK, rb_opc(o.w))) return k;
        uint32_t u = o.pre + rb_len(o.w);
        for (uint32_t i = o.i + 1; i < n; i++) {
            if (rb_in(RB_MATCH_MASK, rb_opc(v.w[i]))) return u;
            u += rb_len(v.w[i]);
        }
        return N;
    };
    auto match_le = [&](uint32_t k) -> uint32_t { // paf.rs:585-587
        const rb_wpos o = rb_tw_find<true>(v, k, lane);
        if (rb_in(RB_MATCH_MASK, rb_opc(o.w))) return k;
        uint32_t u = o.pre; // first unit of the op after the candidate
        for (uint32_t i = o.i; i > 0;) {
            i--;
            if (rb_in(RB_MATCH_MASK, rb_opc(v.w[i]))) return u - 1u;
            u -= rb_len(v.w[i]);
        }
        return 0u;
    };
    uint32_t ks, ke;
    if (!qlast(new_q_st, &ks) || !qlast(new_q_en - 1, &ke)) return RB_ST_PANIC_NOTFOUND;
    uint32_t aln_st = !v.minus ? match_ge(ks) : match_le(ks);
    uint32_t aln_en = !v.minus ? match_le(ke) : match_ge(ke);
    if (aln_st >= N || aln_en >= N) return RB_ST_PANIC_NOTFOUND; // :795-796
    rb_wpos oa = rb_tw_find<true>(v, aln_st, lane), ob = rb_tw_find<true>(v, aln_en, lane);
    uint32_t Ra = rb_tw_before<1>(v, oa.i, lane), Rb = rb_tw_before<1>(v, ob.i, lane);
    auto unit = [&](const rb_wpos &o, uint32_t Rpre, uint32_t k, uint64_t *tpos, uint64_t *qpos) {
        const uint32_t off = k - o.pre, opc = rb_opc(o.w), Qpre = v.Qc[o.i];
        const bool r = rb_in(RB_REF_MASK, opc), q = rb_in(RB_QRY_MASK, opc);
        *tpos = r ? v.t_st + Rpre + off : v.t_st + Rpre - 1;
        *qpos = q ? (v.minus ? v.q_en - 1 - Qpre - off : v.q_st + Qpre + off) : (v.minus ? v.q_en - Qpre : v.q_st + Qpre - 1);
    };
    uint64_t tp, qp_st, qp_en;
    unit(oa, Ra, aln_st, &tp, &qp_st);
    unit(ob, Rb, aln_en, &tp, &qp_en);
    const uint64_t nq_st = qp_st, nq_en = qp_en + 1;
    if (aln_st > aln_en) { // :799-801
        const uint32_t t = aln_st;
        aln_st = aln_en;
        aln_en = t;
        const rb_wpos to = oa;
        oa = ob;
        ob = to;
        const uint32_t tr = Ra;
        Ra = Rb;
        Rb = tr;
    }
    uint64_t t0, t1, qd;
    unit(oa, Ra, aln_st, &t0, &qd);
    unit(ob, Rb, aln_en, &t1, &qd);
    const uint64_t nt_st = t0, nt_en = t1 + 1; // :802-803
    // subset_cigar + collapse (:807-808): ops ia..ib with the first / last length cut; adjacent ops differ, nothing merges; both
    // ends are match-type units, so the strip of :819-822 removes nothing
    const uint32_t ia = oa.i, ib = ob.i, cnt = ib - ia + 1;
    uint64_t R = 0, Q = 0, M = 0;
    for (uint32_t j = (uint32_t)lane; j < cnt; j += 64) {
        const uint32_t i = ia + j, opc = rb_opc(v.w[i]);
        uint32_t len = rb_len(v.w[i]);
        if (cnt == 1) len = aln_en - aln_st + 1u;
        else if (j == 0) len = oa.pre + len - aln_st;
        else if (j == cnt - 1) len = aln_en - ob.pre + 1u;
        out[j] = (len << 4) | opc;
        if (rb_in(RB_REF_MASK, opc)) R += len;
        if (rb_in(RB_QRY_MASK, opc)) Q += len;
        if (rb_in(RB_MATCH_MASK, opc)) M += len;
    }
    R = rb_wave_sum_u64(R), Q = rb_wave_sum_u64(Q), M = rb_wave_sum_u64(M);
    if (nt_en < nt_st || nt_en - nt_st != R) return RB_ST_PANIC_INTEGRITY_T;
    if (nq_en < nq_st || nq_en - nq_st != Q) return RB_ST_PANIC_INTEGRITY_Q;
    row->t_st[s] = nt_st;
    row->t_en[s] = nt_en;
    row->q_st[s] = nq_st;
    row->q_en[s] = nq_en;
    row->nmatch[s] = (uint32_t)M;
    row->aln_len[s] = aln_en - aln_st + 1u;
    row->out_off[s] = out_base;
    row->out_n[s] = cnt;
    return RB_ST_OK;
}

__global__ __launch_bounds__(64) void rb_k_overlap_split_wave(rb_trim_params p) {
    __shared__ uint32_t lds_w[2][3][RB_TW_CAP + 1];
    __shared__ uint32_t lds_c[2][3][RB_TW_NCP + 1];
    const uint64_t pi = blockIdx.x;
    if (pi >= p.n_pairs) return;
    const int lane = rb_lane();
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
        if (lane == 0) p.rows[pi] = w;
        return;
    }
    if (p.policy == RB_BSEARCH_LEGACY || !(nl->flags & RB_F_REGULAR) || !(nr->flags & RB_F_REGULAR) || nl->n_ops > RB_TW_CAP ||
        nr->n_ops > RB_TW_CAP || nl->n_ops == 0 || nr->n_ops == 0) {
        if (lane == 0) p.rows[pi].status = RB_ST_PENDING_INTERNAL;
        return;
    }
    const int32_t ms = p.match_score, ds = p.diff_score, is = p.indel_score;
    rb_wrec L, R;
    L.ops = p.ops + p.op_off[rl] + nl->first_op, L.n = nl->n_ops;
    L.t_st = nl->t_st, L.t_en = nl->t_en, L.q_st = nl->q_st, L.q_en = nl->q_en, L.minus = p.strand[rl] == (uint8_t)'-';
    L.w = lds_w[0][0], L.Qc = lds_w[0][1], L.SP = reinterpret_cast<int32_t *>(lds_w[0][2]), L.cU = lds_c[0][0], L.cQ = lds_c[0][1], L.cR = lds_c[0][2];
    R.ops = p.ops + p.op_off[rr] + nr->first_op, R.n = nr->n_ops;
    R.t_st = nr->t_st, R.t_en = nr->t_en, R.q_st = nr->q_st, R.q_en = nr->q_en, R.minus = p.strand[rr] == (uint8_t)'-';
    R.w = lds_w[1][0], R.Qc = lds_w[1][1], R.SP = reinterpret_cast<int32_t *>(lds_w[1][2]), R.cU = lds_c[1][0], R.cQ = lds_c[1][1], R.cR = lds_c[1][2];
    rb_tw_stage(L, lane, ms, ds, is);
    rb_tw_stage(R, lane, ms, ds, is);
    if ((uint64_t)L.Qtot != L.q_en - L.q_st || (uint64_t)R.Qtot != R.q_en - R.q_st) { // (cannot happen for rows that passed the scan)
        if (lane == 0) p.rows[pi].status = RB_ST_PENDING_INTERNAL;
        return;
    }
    const uint64_t st_ovl = L.q_st > R.q_st ? L.q_st : R.q_st; // trim_overlap.rs:43-44
    const uint64_t en_ovl = L.q_en < R.q_en ? L.q_en : R.q_en;
    const uint64_t n_ov = en_ovl > st_ovl ? en_ovl - st_ovl : 0;
    int64_t best = 0;
    uint64_t best_idx = 0;
    if (n_ov > 0) {
        const int64_t gl0 = rb_tw_G(L, st_ovl, lane, ms, ds, is), gr0 = rb_tw_G(R, st_ovl, lane, ms, ds, is), gr1 = rb_tw_G(R, en_ovl, lane, ms, ds, is);
        const int64_t rsum = gr1 - gr0; // f(0)
        if (rsum > best) best = rsum;   // (index stays 0)
        int64_t cb = INT64_MIN;         // best f over the candidates k > 0 of this lane; ties: the smaller k
        uint64_t ck = 0;
        // a candidate is a position where one record's score changes; that record's own sum up to it comes straight from its
        // prefix arrays, only the other record is searched
        auto consider = [&](uint64_t pos, const rb_wrec &other, bool own_is_left, int64_t g_own) {
            if (pos <= st_ovl || pos > en_ovl) return;
            const int64_t g_other = rb_tw_G_lane(other, pos, ms, ds, is);
            const int64_t gl = own_is_left ? g_own : g_other, gr = own_is_left ? g_other : g_own;
            const int64_t f = (gl - gl0) + (gr1 - gr);
            const uint64_t k = pos - st_ovl;
            if (f > cb || (f == cb && k < ck)) cb = f, ck = k;
        };
        auto candidates = [&](const rb_wrec &v, const rb_wrec &other, bool is_left) {
            // ops whose query bases intersect the overlap: a contiguous op range
            const uint32_t xa = (uint32_t)(!v.minus ? st_ovl - v.q_st : v.q_en - en_ovl);
            const uint32_t xb = (uint32_t)(!v.minus ? en_ovl - 1 - v.q_st : v.q_en - 1 - st_ovl);
            const uint32_t ia = rb_tw_find<false>(v, xa, lane).i, ib = rb_tw_find<false>(v, xb, lane).i;
            for (uint32_t i = ia + (uint32_t)lane; i <= ib && i < v.n; i += 64) {
                const uint32_t wv = v.w[i];
                if (!rb_in(RB_QRY_MASK, rb_opc(wv))) continue;
                const uint64_t len = rb_len(wv), Qi = v.Qc[i];
                const int64_t Si = v.SP[i], mi = (int64_t)v.SP[i + 1] - Si, own = rb_tw_score(rb_opc(wv), ms, ds, is);
                const int64_t w0 = Si, w1 = Si + (int64_t)(len - 1) * own, w2 = Si + mi; // W at offsets Qi, Qi + len - 1, Qi + len
                if (!v.minus) {
                    const uint64_t lo = v.q_st + Qi;
                    consider(lo, other, is_left, w0);
                    consider(lo + len - 1, other, is_left, w1); // the special base (the last one in op order) starts
                    consider(lo + len, other, is_left, w2);
                } else {
                    const uint64_t lo = v.q_en - Qi - len; // G(p) = Stot - W(q_en - p)
                    consider(lo, other, is_left, (int64_t)v.Stot - w2);
                    consider(lo + 1, other, is_left, (int64_t)v.Stot - w1); // the special base (lowest position) ends
                    consider(lo + len, other, is_left, (int64_t)v.Stot - w0);
                }
            }
        };
        candidates(L, R, true);
        candidates(R, L, false);
        if (lane == 0) consider(en_ovl, R, true, rb_tw_G_lane(L, en_ovl, ms, ds, is));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int64_t ob = __shfl_xor(cb, off, 64);
            const uint64_t ok = __shfl_xor(ck, off, 64);
            if (ob > cb || (ob == cb && ok < ck)) cb = ob, ck = ok;
        }
        if (cb > best) best = cb, best_idx = ck;
    }
    w.split_idx = best_idx;
    w.split_score = (int32_t)best;
    const uint64_t split = st_ovl + best_idx;
    const uint64_t ob = p.pair_out_off[pi];
    uint32_t st = rb_tw_clip(L, L.q_st, split, p.out_ops + ob, &w, 0, ob, lane); // trim_overlap.rs:77
    if (st == RB_ST_OK) {
        const uint64_t ob2 = ob + L.n;
        st = rb_tw_clip(R, split, R.q_en, p.out_ops + ob2, &w, 1, ob2, lane); // :78
    }
    w.status = st;
    if (lane == 0) p.rows[pi] = w;
}

extern "C" hipError_t rb_launch_overlap_split(const rb_trim_params *p, hipStream_t stream) {
    if (p->n_pairs == 0) return hipSuccess;
    rb_trim_params q = *p;
    static const bool serial_only = getenv("RB_DEBUG_TRIM_SERIAL") != nullptr; // diagnostics: the general kernel for every pair
    q.only_pending = serial_only ? 0 : 1;
    if (!serial_only) hipLaunchKernelGGL(rb_k_overlap_split_wave, dim3((unsigned)p->n_pairs), dim3(64), 0, stream, q);
    hipLaunchKernelGGL(rb_k_overlap_split, dim3((unsigned)((p->n_pairs + 63) / 64)), dim3(64), 0, stream, q);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// between two passes of trim-paf: the clipped records of a pass become the batch's current records (include/rustybam_amd.h,
// rb_dev_apply_pairs), and the current records gathered into a dense batch again (rb_dev_gather_records)
// ------------------------------------------------------------------------------------------------
struct rb_apply_params {
    uint64_t n_pairs;
    const uint32_t *left, *right;
    const rb_pair_row *rows;
    uint64_t *op_off;
    rb_norm_row *norm;
};
__global__ __launch_bounds__(256) void rb_k_apply_pairs(rb_apply_params p) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t k = t >> 1;
    const int s = (int)(t & 1);
    if (k >= p.n_pairs) return;
    const rb_pair_row *row = &p.rows[k];
    if (row->status != RB_ST_OK) return;
    const uint32_t rec = s ? p.right[k] : p.left[k];
    rb_norm_row n = p.norm[rec];
    n.t_st = row->t_st[s], n.t_en = row->t_en[s], n.q_st = row->q_st[s], n.q_en = row->q_en[s];
    n.first_op = 0, n.n_ops = row->out_n[s];
    n.lead_ops = n.trail_ops = 0; // (a clip starts and ends on a match op: remove_trailing_indels finds nothing, paf.rs:218-220)
    n.nmatch = row->nmatch[s], n.aln_len = row->aln_len[s];
    p.norm[rec] = n;
    p.op_off[rec] = row->out_off[s];
}
extern "C" hipError_t rb_launch_apply_pairs(const rb_apply_params *p, hipStream_t stream) {
    if (p->n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_apply_pairs, dim3((unsigned)((2 * p->n_pairs + 255) / 256)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}

struct rb_gather_params {
    uint64_t n_rec;
    const uint32_t *ops;
    const uint64_t *op_off;
    const rb_norm_row *norm;
    uint64_t *new_off; // [n_rec + 1]: counts (fill == 0) then their exclusive prefix
    uint32_t *new_ops;
    int fill;
};
__global__ __launch_bounds__(256) void rb_k_gather_records(rb_gather_params p) {
    if (!p.fill) { // the kept length of every record
        const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (r < p.n_rec) p.new_off[r] = p.norm[r].status == RB_ST_OK ? p.norm[r].n_ops : 0u;
        return;
    }
    const uint64_t r = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (r >= p.n_rec) return;
    const uint64_t n = p.new_off[r + 1] - p.new_off[r];
    const uint32_t *src = p.ops + p.op_off[r] + p.norm[r].first_op;
    uint32_t *dst = p.new_ops + p.new_off[r];
    for (uint64_t j = rb_lane(); j < n; j += 64) dst[j] = src[j];
}
extern "C" hipError_t rb_launch_gather_records(const rb_gather_params *p, hipStream_t stream) {
    if (p->n_rec == 0) return hipSuccess;
    if (!p->fill) hipLaunchKernelGGL(rb_k_gather_records, dim3((unsigned)((p->n_rec + 255) / 256)), dim3(256), 0, stream, *p);
    else hipLaunchKernelGGL(rb_k_gather_records, dim3((unsigned)((p->n_rec + 3) / 4)), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
