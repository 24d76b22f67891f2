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
#include "rb_trim.h"
#include <algorithm>


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
    __device__ bool is_q(int32_t j) const { return rb_wlen(v.ops, (uint32_t)j) != 0 && rb_s_qry(rb_wopc(v.ops, (uint32_t)j)); }

    // op code of the unit that qpos_to_idx returns for the LAST unit (in op order) of op i
    __device__ uint32_t special_type() const {
        const uint32_t own = rb_wopc(v.ops, (uint32_t)i);
        uint64_t runU = 0;
        uint32_t last = own;
        for (uint32_t j = (uint32_t)i + 1; j < v.n; j++) {
            const uint32_t opc = rb_wopc(v.ops, (uint32_t)j), len = rb_wlen(v.ops, (uint32_t)j);
            if (len == 0) continue;
            if (rb_s_qry(opc)) break;
            runU += len;
            last = opc;
        }
        if (runU == 0) return own;
        if (policy != RB_BSEARCH_LEGACY) return last; // modern: last equal element
        const uint64_t klo = U + rb_wlen(v.ops, (uint32_t)i) - 1;
        const uint64_t k = rb_s_legacy_probe(N, klo, klo + runU);
        if (k == klo) return own;
        uint64_t u = klo + 1;
        for (uint32_t j = (uint32_t)i + 1; j < v.n; j++) {
            const uint32_t opc = rb_wopc(v.ops, (uint32_t)j), len = rb_wlen(v.ops, (uint32_t)j);
            if (len == 0) continue;
            if (k < u + len) return opc;
            u += len;
        }
        return last;
    }

    __device__ void next_op() { // move to the op with the next higher query positions
        const uint64_t base = hi + 1;
        if (!v.minus) {
            U += rb_wlen(v.ops, (uint32_t)i);
            i++;
            while (i < (int32_t)v.n && !is_q(i)) {
                U += rb_wlen(v.ops, (uint32_t)i);
                i++;
            }
            if (i >= (int32_t)v.n) return;
        } else {
            i--;
            while (i >= 0 && !is_q(i)) {
                U -= rb_wlen(v.ops, (uint32_t)i);
                i--;
            }
            if (i < 0) return;
            U -= rb_wlen(v.ops, (uint32_t)i);
        }
        lo = base;
        hi = base + rb_wlen(v.ops, (uint32_t)i) - 1;
    }

    __device__ void seek(uint64_t p) {
        if (!v.minus) {
            i = 0;
            U = 0;
            while (i < (int32_t)v.n && !is_q(i)) {
                U += rb_wlen(v.ops, (uint32_t)i);
                i++;
            }
        } else {
            i = (int32_t)v.n - 1;
            U = N;
            while (i >= 0 && !is_q(i)) {
                U -= rb_wlen(v.ops, (uint32_t)i);
                i--;
            }
            if (i >= 0) U -= rb_wlen(v.ops, (uint32_t)i);
        }
        if (i < 0 || i >= (int32_t)v.n) return;
        lo = v.q_st;
        hi = lo + rb_wlen(v.ops, (uint32_t)i) - 1;
        while (p > hi && i >= 0 && i < (int32_t)v.n) next_op();
        pos = p;
    }

    __device__ bool valid() const { return i >= 0 && i < (int32_t)v.n; }

    // current run of equal scores starting at pos
    __device__ void run(uint64_t *count, int32_t *score) const {
        const uint32_t own = rb_wopc(v.ops, (uint32_t)i);
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

__device__ void rb_serial_pair(const rb_trim_params &p, const uint64_t pi);
__global__ __launch_bounds__(64) void rb_k_overlap_split(rb_trim_params p) {
    if (p.only_pending && p.pend_list) { // what the wave kernels left, from their list
        const uint64_t n = *p.pend;
        for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
            const uint64_t pi = p.pend_list[e];
            if (p.rows[pi].status == 0x7FFF0001u) rb_serial_pair(p, pi);
        }
        return;
    }
    const uint64_t pi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pi >= p.n_pairs) return;
    if (p.only_pending && p.rows[pi].status != 0x7FFF0001u) return; // (what the wave-per-pair kernels left, every row looked at)
    rb_serial_pair(p, pi);
}
__device__ void rb_serial_pair(const rb_trim_params &p, const uint64_t pi) {
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
// type, a match-type op at both ends) of any length whose overlap spans at most RB_TW_CAP - 128 ops, modern binary-search policy.  Every question the serial
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
#ifndef RB_TW_CAP
#define RB_TW_CAP 192   // ops of a record's region, first attempt: 5 KB of LDS per pair and 64 VGPRs = 32 pairs per CU.  The kernel waits on
                        // memory, not on the ALUs: 768 ops (8 pairs per CU, whole 500-op records staged) took 20.1 ms per 1.5e6 pairs, this 7.2
#endif
#ifndef RB_TW_CAP1
#define RB_TW_CAP1 1024 // the pairs the first attempt lists, second attempt (26 KB: 6 pairs per CU)
#endif
#define RB_TW_CAP2 6144 // second attempt for the pairs whose overlap spans more ops (one pair per CU at a time; such overlaps are rare)
#define RB_TW_CAP3 32768 // third attempt: the same arrays in device memory (whole-chromosome alignments that overlap by hundreds of kilobases)
#define RB_TW_SLAB_WORDS(CAP) (2u * 3u * ((CAP) + 1u) + 2u * 3u * ((CAP) / 16u + 2u))
#ifndef RB_TW_STOP
#define RB_TW_STOP 0 // diagnostics (tools/prof_c4_decomp.sh): != 0 ends a pair early -- 1 behind the left record's staging, 2 behind both, 3 behind the
                     // searches of the overlap's end ops, 4 behind the split, 5 behind the left clip; the rows are wrong then, only the time is of interest
#endif

// A record of a pair as the wave kernel sees it.  Only the REGION of the record that the overlap can touch is staged in LDS --
// the ops that hold the overlapped query bases plus a 64-op step on either side -- with prefixes that are absolute (counted from
// the record's first op), so a record may be as long as it likes: what lies in front of the region is streamed once for its
// totals, what lies behind it is only copied when the clip keeps it.
struct rb_wrec {
    const uint32_t *ops; // the record's kept ops, in memory
    uint32_t n;          // how many
    uint32_t i0, m, ncp; // region = ops [i0, i0 + m); checkpoints in use = ceil(m / 16) + 1 (the last one holds the prefixes at the region's end)
    uint64_t t_st, t_en, q_st, q_en;
    bool minus;
    bool bad;            // a search left the region: the pair goes to the serial kernel
    uint32_t N, Qtot, Rtot; // totals of the whole record (units from the norm row, bases from the coordinates)
    uint32_t *w;            // LDS [m + 1]: the op words of the region; w[m] = a zero-length M (ends every D / N run, contains nothing)
    uint32_t *Qc;           // LDS [m + 1]: query bases before op i0 + k
    int32_t *SP;            // LDS [m + 1]: score of the query bases of the region before op i0 + k, in op order
    uint32_t *cU, *cQ, *cR; // LDS [ncp]: units / query bases / reference bases before op i0 + 16 c
};
struct rb_wpos { // an op (i = n: none) and the exclusive prefix of the searched quantity at it
    uint32_t i, w, pre;
};


// Stage the region of v that holds the query offsets [xa, xb] (op order).  false: it does not fit RB_TW_CAP ops.
template <int CAP>
__device__ bool rb_tw_stage(rb_wrec &v, int lane, int32_t ms, int32_t ds, int32_t is, uint32_t xa, uint32_t xb) {
    v.bad = false;
    // phase A: 64 ops at a time from the record's first op, totals only, up to the step that holds xb; the region starts one step
    // before the step that holds xa and ends one step behind the one that holds xb
    uint32_t Ub = 0, Qb = 0, Rb = 0, pU = 0, pQ = 0, pR = 0; // prefixes at the current step / at the step before
    uint32_t i0 = 0, bU = 0, bQ = 0, bR = 0, i1 = v.n;
    bool found = v.n <= (uint32_t)CAP; // a record that fits is staged whole: nothing to look for
    // Round 3: the overlap of a pair lies at one END of each record (the left record's last query bases, the right record's first,
    // or the other way round on '-'), so the totals in front of the region are taken from whichever side is nearer: from the
    // record's first op forwards, or from its last op BACKWARDS with the record's totals (known from its row) minus the suffix
    // sums.  Same steps (multiples of 64 ops from op 0), same region, a handful of steps instead of the whole record.
    const bool backwards = !found && xa > v.Qtot - 1u - (xb < v.Qtot ? xb : v.Qtot - 1u);
    // ... and the region is cut to the ops that matter: 16 ops in front of the op that holds xa, 16 behind the op that holds xb
    // (the walks to the next / previous match op and the D / N runs behind a last base stay within a few ops; a search that leaves
    // the region still sends the pair to the serial kernel).  Round 2 took whole 64-op steps on either side: 192 ops staged and
    // scanned for an overlap of a dozen.  The op inside its step is found with one scan of that step.
    auto refine_start = [&](uint32_t c0, uint32_t len, uint32_t ql, uint32_t rl, uint32_t Ub4, uint32_t Qb4, uint32_t Rb4) -> bool {
        const uint32_t iq = rb_wave_scan_incl(ql);
        const uint64_t mk = __ballot(ql != 0u && Qb4 + iq - ql <= xa && xa < Qb4 + iq);
        if (!mk) return false;
        const int la = __builtin_ctzll(mk);
        if (la < 16) return false; // (the margin reaches into the step in front: the caller takes that whole step, as before)
        const uint32_t iu = rb_wave_scan_incl(len), ir = rb_wave_scan_incl(rl);
        const int sl = la - 16;
        i0 = c0 + (uint32_t)sl;
        bU = Ub4 + rb_readlane<uint32_t>(iu - len, sl), bQ = Qb4 + rb_readlane<uint32_t>(iq - ql, sl), bR = Rb4 + rb_readlane<uint32_t>(ir - rl, sl);
        return true;
    };
    auto refine_end = [&](uint32_t c0, uint32_t ql, uint32_t Qb4) {
        const uint32_t iq = rb_wave_scan_incl(ql);
        const uint64_t mk = __ballot(ql != 0u && Qb4 + iq - ql <= xb && xb < Qb4 + iq);
        const uint32_t lb = mk ? (uint32_t)__builtin_ctzll(mk) : 63u;
        i1 = c0 + lb + 17u < v.n ? c0 + lb + 17u : v.n;
    };
    if (backwards) {
        uint32_t Ua = 0, Qa = 0, Ra = 0; // sums over the steps BEHIND the current one
        bool have_b = false, have_a = false;
        for (int64_t c0 = (int64_t)((v.n - 1u) / 64u) * 64; c0 >= 0; c0 -= 64) {
            const uint32_t i = (uint32_t)c0 + (uint32_t)lane;
            const uint32_t w = i < v.n ? v.ops[i] : 0u;
            const uint32_t opc = rb_opc(w), len = i < v.n ? rb_len(w) : 0u;
            const uint32_t ql = rb_in(RB_QRY_MASK, opc) ? len : 0u, rl = rb_in(RB_REF_MASK, opc) ? len : 0u;
            const uint32_t tu = rb_wave_sum_u32(len), tq = rb_wave_sum_u32(ql), tr = rb_wave_sum_u32(rl);
            const uint32_t qb4 = v.Qtot - Qa - tq; // query bases in front of this step
            if (have_a) { // the step in front of the one that holds xa: the region starts here
                i0 = (uint32_t)c0, bU = v.N - Ua - tu, bQ = qb4, bR = v.Rtot - Ra - tr;
                found = true;
                break;
            }
            if (!have_b && qb4 <= xb) have_b = true, refine_end((uint32_t)c0, ql, qb4);
            if (have_b && qb4 <= xa) {
                have_a = true;
                if (refine_start((uint32_t)c0, len, ql, rl, v.N - Ua - tu, qb4, v.Rtot - Ra - tr)) {
                    found = true;
                    break;
                }
                if (c0 == 0) { // (the record's first step: nothing in front of it)
                    i0 = 0, bU = bQ = bR = 0;
                    found = true;
                    break;
                }
            }
            Ua += tu, Qa += tq, Ra += tr;
        }
    }
    bool found_a = false;
    for (uint32_t c0 = 0; c0 < v.n && !found; c0 += 64) {
        const uint32_t i = c0 + (uint32_t)lane;
        const uint32_t w = i < v.n ? v.ops[i] : 0u;
        const uint32_t opc = rb_opc(w), len = i < v.n ? rb_len(w) : 0u;
        const uint32_t ql = rb_in(RB_QRY_MASK, opc) ? len : 0u, rl = rb_in(RB_REF_MASK, opc) ? len : 0u;
        const uint32_t tu = rb_wave_sum_u32(len), tq = rb_wave_sum_u32(ql), tr = rb_wave_sum_u32(rl);
        if (!found_a && Qb + tq > xa) {
            found_a = true;
            if (!refine_start(c0, len, ql, rl, Ub, Qb, Rb) && c0 >= 64u) i0 = c0 - 64u, bU = pU, bQ = pQ, bR = pR;
        }
        if (found_a && Qb + tq > xb) {
            refine_end(c0, ql, Qb);
            found = true;
            break;
        }
        pU = Ub, pQ = Qb, pR = Rb;
        Ub += tu, Qb += tq, Rb += tr;
    }
    if (found_a && !found) found = true, i1 = v.n; // (xb behind the last query base: the region runs to the record's end)
    if (!found || i1 - i0 > (uint32_t)CAP) return false;
    const uint32_t m = i1 - i0;
    v.i0 = i0, v.m = m;
    // phase B: the op words of the region into LDS (the common size: every load in flight at once; addresses past the region
    // re-read its last op)
    if constexpr (CAP <= 1024) {
        uint32_t t[CAP / 64];
#pragma unroll
        for (int c = 0; c < CAP / 64; c++) {
            const uint32_t k = (uint32_t)c * 64u + (uint32_t)lane;
            t[c] = v.ops[i0 + (k < m ? k : m - 1u)];
        }
#pragma unroll
        for (int c = 0; c < CAP / 64; c++) {
            const uint32_t k = (uint32_t)c * 64u + (uint32_t)lane;
            if (k < m) v.w[k] = t[c];
        }
    } else {
        for (uint32_t k = (uint32_t)lane; k < m; k += 64u) v.w[k] = v.ops[i0 + k];
    }
    if (lane == 0) v.w[m] = RB_OP_M; // length 0
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // phase C: prefixes, 64 ops at a time; every 16th op leaves a checkpoint
    Ub = bU, Qb = bQ, Rb = bR;
    int32_t Sb = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += 64) {
        const uint32_t k = c0 + (uint32_t)lane;
        const bool in = k < m;
        const uint32_t w = in ? v.w[k] : 0u;
        const uint32_t opc = rb_opc(w), len = in ? rb_len(w) : 0u;
        const bool q = in && rb_in(RB_QRY_MASK, opc), r = in && rb_in(RB_REF_MASK, opc);
        int32_t mm = 0;
        if (q) { // own score for all bases but the last in op order, which takes the score of the last D / N op of the run behind it
            int32_t sp = rb_tw_score(opc, ms, ds, is);
            const int32_t own = sp;
            for (uint32_t j = k + 1;; j++) { // (regular records: short; the sentinel at w[m] stops it -- a run cut by the region's end
                                             //  lies in the step behind the overlap, where no score is looked up)
                const uint32_t oj = rb_opc(v.w[j]);
                if (rb_in(RB_QRY_MASK, oj)) break;
                sp = rb_tw_score(oj, ms, ds, is);
            }
            mm = (int32_t)(len - 1u) * own + sp;
        }
        const uint32_t iu = rb_wave_scan_incl(len), iq = rb_wave_scan_incl(q ? len : 0u), ir = rb_wave_scan_incl(r ? len : 0u);
        const int32_t isc = (int32_t)rb_wave_scan_incl((uint32_t)mm);
        if (in) {
            v.Qc[k] = Qb + iq - (q ? len : 0u);
            v.SP[k] = Sb + isc - mm;
            if ((k & 15u) == 0u) {
                v.cU[k >> 4] = Ub + iu - len;
                v.cQ[k >> 4] = Qb + iq - (q ? len : 0u);
                v.cR[k >> 4] = Rb + ir - (r ? len : 0u);
            }
        }
        Ub += rb_readlane<uint32_t>(iu, 63);
        Qb += rb_readlane<uint32_t>(iq, 63);
        Rb += rb_readlane<uint32_t>(ir, 63);
        Sb += rb_readlane<int>(isc, 63);
    }
    v.ncp = (m + 15u) / 16u + 1u;
    if (lane == 0) {
        v.cU[v.ncp - 1] = Ub, v.cQ[v.ncp - 1] = Qb, v.cR[v.ncp - 1] = Rb;
        v.Qc[m] = Qb, v.SP[m] = Sb;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// units (KIND 0) / reference bases (KIND 1) before op i of the region: the checkpoint of its chunk + the ops of the chunk in front of it
template <int KIND>
__device__ __forceinline__ uint32_t rb_tw_before(const rb_wrec &v, uint32_t i, int lane) {
    const uint32_t k = i - v.i0, c = k >> 4, j = 16u * c + (uint32_t)lane;
    uint32_t x = 0;
    if (lane < 16 && j < k) {
        const uint32_t w = v.w[j];
        x = (KIND == 0 || rb_in(RB_REF_MASK, rb_opc(w))) ? rb_len(w) : 0u;
    }
    return (KIND == 0 ? v.cU[c] : v.cR[c]) + rb_wave_sum_u32(x);
}
// wave-uniform search inside the region: the op that holds unit x (BY_UNIT) / query offset x, found by all lanes at once -- one
// ballot over the checkpoints, then the 16 ops of that chunk side by side.  i = n: no op of the region holds x.
template <bool BY_UNIT>
__device__ rb_wpos rb_tw_find(const rb_wrec &v, uint32_t x, int lane) {
    const uint32_t *cp = BY_UNIT ? v.cU : v.cQ;
    rb_wpos o;
    o.i = v.n, o.w = RB_NULL_OP, o.pre = 0;
    // the last chunk whose checkpoint is <= x: 64 checkpoints per ballot, a 64-ary search when the region has more chunks
    uint32_t cbase = 0, ccount = v.ncp - 1u; // (the last entry is not a chunk)
    while (ccount > 64u) {
        const uint32_t stride = (ccount + 63u) / 64u, t = (uint32_t)lane * stride;
        const uint64_t mk0 = __ballot(t < ccount && cp[cbase + t] <= x);
        if (!mk0) return o; // in front of the region
        const uint32_t j = (uint32_t)__builtin_popcountll(mk0) - 1u;
        cbase += j * stride;
        ccount = ccount - j * stride < stride ? ccount - j * stride : stride;
    }
    const bool le = (uint32_t)lane < ccount && cp[cbase + (uint32_t)lane] <= x;
    const uint64_t lem = __ballot(le);
    if (!lem) return o; // in front of the region
    const uint32_t c = cbase + (uint32_t)__builtin_popcountll(lem) - 1u;
    const uint32_t k = 16u * c + (uint32_t)lane;
    const bool in = lane < 16 && k < v.m;
    const uint32_t w = in ? v.w[k] : 0u;
    const uint32_t len = in ? rb_len(w) : 0u;
    uint32_t pre;
    bool hit;
    if (BY_UNIT) {
        pre = v.cU[c] + rb_wave_scan_incl(len) - len;
        hit = in && pre <= x && x - pre < len;
    } else {
        pre = in ? v.Qc[k] : 0u;
        hit = in && rb_in(RB_QRY_MASK, rb_opc(w)) && pre <= x && x - pre < len;
    }
    const uint64_t mk = __ballot(hit);
    if (!mk) return o; // behind the region (or behind the record)
    const int l = __builtin_ctzll(mk);
    o.i = v.i0 + 16u * c + (uint32_t)l;
    o.w = rb_readlane<uint32_t>(w, l), o.pre = rb_readlane<uint32_t>(pre, l);
    return o;
}
// score of the region's query bases in front of query offset x (op order), wave-uniform x inside the region (or just behind it)
__device__ __forceinline__ int64_t rb_tw_W(rb_wrec &v, uint32_t x, int lane, int32_t ms, int32_t ds, int32_t is) {
    if (x >= v.Qc[v.m]) return v.SP[v.m];
    const rb_wpos o = rb_tw_find<false>(v, x, lane);
    if (o.i >= v.n) {
        v.bad = true;
        return 0;
    }
    return (int64_t)v.SP[o.i - v.i0] + (int64_t)(x - o.pre) * rb_tw_score(rb_opc(o.w), ms, ds, is);
}
// the same for a per-lane x: binary search in Qc (non-query ops share the value of the query op behind them, and that op comes
// later: the last index with Qc <= x is the query op that holds x)
__device__ int64_t rb_tw_W_lane(const rb_wrec &v, uint32_t x, int32_t ms, int32_t ds, int32_t is) {
    if (x >= v.Qc[v.m]) return v.SP[v.m];
    uint32_t lo = 0, hi = v.m;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v.Qc[mid] <= x) lo = mid; else hi = mid;
    }
    return (int64_t)v.SP[lo] + (int64_t)(x - v.Qc[lo]) * rb_tw_score(rb_opc(v.w[lo]), ms, ds, is);
}
// score of the query positions below p, in increasing position order, up to a constant of the record (only differences are used)
__device__ __forceinline__ int64_t rb_tw_G(rb_wrec &v, uint64_t p, int lane, int32_t ms, int32_t ds, int32_t is) {
    return !v.minus ? rb_tw_W(v, (uint32_t)(p - v.q_st), lane, ms, ds, is) : -rb_tw_W(v, (uint32_t)(v.q_en - p), lane, ms, ds, is);
}
__device__ __forceinline__ int64_t rb_tw_G_lane(const rb_wrec &v, uint64_t p, int32_t ms, int32_t ds, int32_t is) {
    return !v.minus ? rb_tw_W_lane(v, (uint32_t)(p - v.q_st), ms, ds, is) : -rb_tw_W_lane(v, (uint32_t)(v.q_en - p), ms, ds, is);
}

// truncate_record_by_query (paf.rs:785-823) on a staged regular record; same results as rb_clip_by_query.  One end of the new
// query range is the record's own end (trim_overlap.rs:77-78), the other lies in the staged region.
struct rb_wend { // a unit of the record: its index, the op that holds it, and the prefixes before that op
    uint32_t k;   // unit
    rb_wpos o;    // op, its word, units before it
    uint32_t R, Q; // reference / query bases before the op
};
struct rb_wcut { // in-place clip: the two words to rewrite (absolute indices into the ops array) once BOTH clips of the pair stand
    uint64_t at_first, at_last;
    uint32_t w_first, w_last;
};
__device__ uint32_t rb_tw_clip(rb_wrec &v, uint64_t new_q_st, uint64_t new_q_en, uint32_t *out, rb_pair_row *row, int s, uint64_t out_base,
                               int lane, rb_wcut *cut = nullptr, uint64_t rec_base = 0) {
    if (!(new_q_st >= v.q_st) || !(new_q_en <= v.q_en) || new_q_en == 0) return RB_ST_PANIC_ASSERT; // :787-788
    if (new_q_en <= new_q_st) { // an empty range: the serial kernel says what the reference does with it
        v.bad = true;
        return RB_ST_OK;
    }
    const uint32_t n = v.n, N = v.N;
    // the match-type unit truncate_record_by_query ends up at for query position p: qpos_to_idx_match (paf.rs:564-590) = the last
    // unit whose qpos equals p (modern policy), then the nearest match-type unit in the search direction
    auto resolve = [&](uint64_t p, bool search_up, rb_wend *e) -> bool {
        if (p < v.q_st || p >= v.q_en) return false;
        const uint32_t x = (uint32_t)(v.minus ? v.q_en - 1 - p : p - v.q_st);
        if (x < v.Qc[0] || x >= v.Qc[v.m]) {
            // outside the region: only the record's own first / last query base is asked for there.  A regular record starts and
            // ends on a match op; its last base is its last unit, its first base its first unit unless that op has one base and a
            // D / N run behind it (the run repeats the position: left to the serial kernel)
            if (x == 0u) {
                const uint32_t w0 = v.ops[0];
                if (rb_len(w0) < 2u && n > 1u && !rb_in(RB_QRY_MASK, rb_opc(v.ops[1]))) return false;
                e->k = 0, e->o.i = 0, e->o.w = w0, e->o.pre = 0, e->R = 0, e->Q = 0;
                return true;
            }
            if (x + 1u == v.Qtot) {
                const uint32_t wl = v.ops[n - 1u], len = rb_len(wl);
                e->k = N - 1u, e->o.i = n - 1u, e->o.w = wl, e->o.pre = N - len, e->R = v.Rtot - len, e->Q = v.Qtot - len;
                return true;
            }
            return false;
        }
        const rb_wpos o = rb_tw_find<false>(v, x, lane);
        if (o.i >= n) return false;
        const uint32_t j = x - o.pre, len = rb_len(o.w);
        const uint32_t ub = rb_tw_before<0>(v, o.i, lane);
        uint32_t u = ub + j;
        bool moved = false; // the unit left the op that holds the base
        if (j + 1u == len) { // last base of the op: the D / N units behind it repeat its position
            uint32_t k2 = o.i - v.i0 + 1u;
            for (; k2 < v.m && !rb_in(RB_QRY_MASK, rb_opc(v.w[k2])); k2++) u += rb_len(v.w[k2]), moved = true;
            if (k2 >= v.m && v.i0 + v.m < n) return false; // (the run leaves the region)
        }
        // nearest match-type unit, up (paf.rs:581-583) or down (:585-587); the op that holds unit u is the one just found unless the
        // unit moved into the run behind it (round 3: the second search is skipped then -- it was a tenth of a pair's instructions)
        rb_wpos om;
        if (!moved) om.i = o.i, om.w = o.w, om.pre = ub;
        else om = rb_tw_find<true>(v, u, lane);
        if (om.i >= n) return false;
        uint32_t km = u;
        if (!rb_in(RB_MATCH_MASK, rb_opc(om.w))) {
            if (search_up) {
                uint32_t uu = om.pre + rb_len(om.w), k2 = om.i - v.i0 + 1u;
                for (; k2 < v.m && !rb_in(RB_MATCH_MASK, rb_opc(v.w[k2])); k2++) uu += rb_len(v.w[k2]);
                if (k2 >= v.m) return false; // (no match op behind it inside the region; at the record's end the reference panics: serial kernel)
                km = uu;
                om.i = v.i0 + k2, om.w = v.w[k2], om.pre = uu;
            } else {
                uint32_t uu = om.pre, k2 = om.i - v.i0;
                bool got = false;
                while (k2 > 0u) {
                    k2--;
                    if (rb_in(RB_MATCH_MASK, rb_opc(v.w[k2]))) {
                        got = true;
                        break;
                    }
                    uu -= rb_len(v.w[k2]);
                }
                if (!got) return false;
                km = uu - 1u;
                om.i = v.i0 + k2, om.w = v.w[k2], om.pre = uu - rb_len(v.w[k2]);
            }
        }
        e->k = km, e->o = om, e->R = rb_tw_before<1>(v, om.i, lane), e->Q = v.Qc[om.i - v.i0];
        return true;
    };
    rb_wend A, B; // paf.rs:792-796: the start searches up on '+' and down on '-', the end the other way
    if (!resolve(new_q_st, !v.minus, &A) || !resolve(new_q_en - 1, v.minus, &B)) {
        v.bad = true;
        return RB_ST_OK;
    }
    auto unit = [&](const rb_wend &e, uint64_t *tpos, uint64_t *qpos) { // both are match-type units
        const uint32_t off = e.k - e.o.pre;
        *tpos = v.t_st + e.R + off;
        *qpos = v.minus ? v.q_en - 1 - e.Q - off : v.q_st + e.Q + off;
    };
    uint64_t tp, qp_st, qp_en;
    unit(A, &tp, &qp_st);
    unit(B, &tp, &qp_en);
    const uint64_t nq_st = qp_st, nq_en = qp_en + 1;
    if (A.k > B.k) { // :799-801
        const rb_wend t = A;
        A = B;
        B = t;
    }
    uint64_t t0, t1, qd;
    unit(A, &t0, &qd);
    unit(B, &t1, &qd);
    const uint64_t nt_st = t0, nt_en = t1 + 1; // :802-803
    // subset_cigar + collapse (:807-808): ops ia..ib with the first / last length cut; adjacent ops differ, nothing merges; both
    // ends are match-type units, so the strip of :819-822 removes nothing
    // Round 3: the copy is only a copy.  check_integrity of the clipped record (:819-822) compares the sums of its ops with
    // coordinates that were derived from those very prefixes: for a regular record it cannot fail, and nmatch follows from the spans
    // (a match-type op counts in reference, query and units, an I in query and units, a D / N in reference and units, so
    // matches = ref + query - units: the clip kernel's identity).  Round 2 summed three 64-bit totals over every copied op: 470 of a
    // pair's 2640 vector instructions.
    const uint32_t ia = A.o.i, ib = B.o.i, cnt = ib - ia + 1;
    if (cut) { // (in place: nothing is copied; rec_base = where the record's kept ops begin in the ops array)
        const uint32_t lf = cnt == 1 ? B.k - A.k + 1u : A.o.pre + rb_len(A.o.w) - A.k, ll = cnt == 1 ? lf : B.k - B.o.pre + 1u;
        cut->at_first = rec_base + ia, cut->at_last = rec_base + ib;
        cut->w_first = (lf << 4) | rb_opc(A.o.w), cut->w_last = (ll << 4) | rb_opc(B.o.w);
        out_base = rec_base + ia;
    } else {
        for (uint32_t j = (uint32_t)lane; j < cnt; j += 64) {
            const uint32_t wv = v.ops[ia + j];
            uint32_t len = rb_len(wv);
            if (cnt == 1) len = B.k - A.k + 1u;
            else if (j == 0) len = A.o.pre + len - A.k;
            else if (j == cnt - 1) len = B.k - B.o.pre + 1u;
            out[j] = (len << 4) | rb_opc(wv);
        }
    }
    // (what CAN fail is the query side: the new start and end are resolved independently -- up and down --, and when the end lands on
    //  a lower query position than the start the coordinates say end + 1 - start while the ops between the two units still hold
    //  |end - start| + 1 query bases: check_integrity's unwrap panics)
    if (nt_en < nt_st) return RB_ST_PANIC_INTEGRITY_T;
    if (qp_en < qp_st) return RB_ST_PANIC_INTEGRITY_Q;
    const uint32_t units = B.k - A.k + 1u;
    row->t_st[s] = nt_st;
    row->t_en[s] = nt_en;
    row->q_st[s] = nq_st;
    row->q_en[s] = nq_en;
    row->nmatch[s] = (uint32_t)((nt_en - nt_st) + (nq_en - nq_st) - units);
    row->aln_len[s] = units;
    row->out_off[s] = out_base;
    row->out_n[s] = cnt;
    return RB_ST_OK;
}

template <int CAP>
__device__ void rb_tw_pair(const rb_trim_params &p, const uint64_t pi, uint32_t (*lds_w)[3][CAP + 1], uint32_t (*lds_c)[3][CAP / 16 + 2]) {
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
    auto pending = [&](uint32_t why = 0) { // (why: diagnostics, RB_DEBUG_TRIM_NO_SERIAL; the serial kernel rewrites the whole row)
        if (lane == 0) {
            if (p.pend_list && !p.only_pending) p.pend_list[atomicAdd(p.pend, 1ull)] = (uint32_t)pi; // (listed once: by the first attempt)
            p.rows[pi].status = RB_ST_PENDING_INTERNAL, p.rows[pi].split_idx = why;
        }
    };
    if (p.policy == RB_BSEARCH_LEGACY || !(nl->flags & RB_F_REGULAR) || !(nr->flags & RB_F_REGULAR) || nl->n_ops == 0 || nr->n_ops == 0) {
        pending(1);
        return;
    }
    const int32_t ms = p.match_score, ds = p.diff_score, is = p.indel_score;
    rb_wrec L, R;
    L.ops = p.ops + p.op_off[rl] + nl->first_op, L.n = nl->n_ops;
    L.t_st = nl->t_st, L.t_en = nl->t_en, L.q_st = nl->q_st, L.q_en = nl->q_en, L.minus = p.strand[rl] == (uint8_t)'-';
    L.N = nl->aln_len, L.Qtot = (uint32_t)(nl->q_en - nl->q_st), L.Rtot = (uint32_t)(nl->t_en - nl->t_st);
    L.w = lds_w[0][0], L.Qc = lds_w[0][1], L.SP = reinterpret_cast<int32_t *>(lds_w[0][2]), L.cU = lds_c[0][0], L.cQ = lds_c[0][1], L.cR = lds_c[0][2];
    R.ops = p.ops + p.op_off[rr] + nr->first_op, R.n = nr->n_ops;
    R.t_st = nr->t_st, R.t_en = nr->t_en, R.q_st = nr->q_st, R.q_en = nr->q_en, R.minus = p.strand[rr] == (uint8_t)'-';
    R.N = nr->aln_len, R.Qtot = (uint32_t)(nr->q_en - nr->q_st), R.Rtot = (uint32_t)(nr->t_en - nr->t_st);
    R.w = lds_w[1][0], R.Qc = lds_w[1][1], R.SP = reinterpret_cast<int32_t *>(lds_w[1][2]), R.cU = lds_c[1][0], R.cQ = lds_c[1][1], R.cR = lds_c[1][2];
    const uint64_t st_ovl = L.q_st > R.q_st ? L.q_st : R.q_st; // trim_overlap.rs:43-44
    const uint64_t en_ovl = L.q_en < R.q_en ? L.q_en : R.q_en;
    if (en_ovl <= st_ovl || st_ovl < L.q_st || en_ovl > L.q_en || st_ovl < R.q_st || en_ovl > R.q_en) { // (no overlap: the serial kernel says what the reference does)
        pending(2);
        return;
    }
    // query offsets of the overlap in each record's op order
    auto span = [&](const rb_wrec &v, uint32_t *xa, uint32_t *xb) {
        *xa = (uint32_t)(!v.minus ? st_ovl - v.q_st : v.q_en - en_ovl);
        *xb = (uint32_t)(!v.minus ? en_ovl - 1 - v.q_st : v.q_en - 1 - st_ovl);
    };
    uint32_t lxa, lxb, rxa, rxb;
    span(L, &lxa, &lxb);
    span(R, &rxa, &rxb);
#if RB_TW_STOP == 1
    if (!rb_tw_stage<CAP>(L, lane, ms, ds, is, lxa, lxb)) pending(3);
    if (lane == 0) p.rows[pi].split_idx = L.m;
    return;
#endif
    if (!rb_tw_stage<CAP>(L, lane, ms, ds, is, lxa, lxb) || !rb_tw_stage<CAP>(R, lane, ms, ds, is, rxa, rxb)) { // an overlap of more ops than the region holds
        pending(3);
        return;
    }
#if RB_TW_STOP == 2
    if (lane == 0) p.rows[pi].split_idx = L.m + R.m;
    return;
#endif
    int64_t best = 0;
    uint64_t best_idx = 0;
    // the ops that hold the first and the last overlapped query base of each record, searched ONCE (round 3: the scores at the ends
    // of the overlap and the candidate ranges below each searched them again -- seven wave searches of a pair's instructions)
    const rb_wpos La = rb_tw_find<false>(L, lxa, lane), Lb = rb_tw_find<false>(L, lxb, lane), Ra = rb_tw_find<false>(R, rxa, lane), Rb = rb_tw_find<false>(R, rxb, lane);
    if (La.i >= L.n || Lb.i >= L.n || Ra.i >= R.n || Rb.i >= R.n) {
        pending(4);
        return;
    }
#if RB_TW_STOP == 3
    if (lane == 0) p.rows[pi].split_idx = La.i + Lb.i + Ra.i + Rb.i;
    return;
#endif
    // W (score of the query bases in front of offset x, op order) at x = xa and at x = xb + 1, from those ops
    auto W_at_first = [&](const rb_wrec &v, const rb_wpos &o, uint32_t xa) -> int64_t {
        return (int64_t)v.SP[o.i - v.i0] + (int64_t)(xa - o.pre) * rb_tw_score(rb_opc(o.w), ms, ds, is);
    };
    auto W_behind_last = [&](const rb_wrec &v, const rb_wpos &o, uint32_t xb) -> int64_t {
        const uint32_t k = o.i - v.i0;
        return xb + 1u < o.pre + rb_len(o.w) ? (int64_t)v.SP[k] + (int64_t)(xb + 1u - o.pre) * rb_tw_score(rb_opc(o.w), ms, ds, is) : (int64_t)v.SP[k + 1u];
    };
    // G(p) = W(p - q_st) on '+', -W(q_en - p) on '-': st_ovl is offset xa on '+' and xb + 1 on '-', en_ovl the other way round
    auto G_st = [&](const rb_wrec &v, const rb_wpos &oa, const rb_wpos &ob, uint32_t xa, uint32_t xb) -> int64_t {
        return !v.minus ? W_at_first(v, oa, xa) : -W_behind_last(v, ob, xb);
    };
    auto G_en = [&](const rb_wrec &v, const rb_wpos &oa, const rb_wpos &ob, uint32_t xa, uint32_t xb) -> int64_t {
        return !v.minus ? W_behind_last(v, ob, xb) : -W_at_first(v, oa, xa);
    };
    {
        const int64_t gl0 = G_st(L, La, Lb, lxa, lxb), gr0 = G_st(R, Ra, Rb, rxa, rxb), gr1 = G_en(R, Ra, Rb, rxa, rxb);
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
        auto candidates = [&](rb_wrec &v, const rb_wrec &other, bool is_left, uint32_t ia, uint32_t ib) {
            // ops whose query bases intersect the overlap: a contiguous op range [ia, ib]
            for (uint32_t i = ia + (uint32_t)lane; i <= ib; i += 64) {
                const uint32_t k = i - v.i0;
                const uint32_t wv = v.w[k];
                if (!rb_in(RB_QRY_MASK, rb_opc(wv))) continue;
                const uint64_t len = rb_len(wv), Qi = v.Qc[k];
                const int64_t Si = v.SP[k], mi = (int64_t)v.SP[k + 1] - Si, own = rb_tw_score(rb_opc(wv), ms, ds, is);
                const int64_t w0 = Si, w1 = Si + (int64_t)(len - 1) * own, w2 = Si + mi; // W at offsets Qi, Qi + len - 1, Qi + len
                // the score changes where the op starts, where its special last base starts (only if that base scores differently:
                // a D / N run behind the op) and where the op ends -- which is where the next query op starts, so only the last op
                // of the range looks at its end
                const bool special = mi != (int64_t)len * own;
                if (!v.minus) {
                    const uint64_t lo = v.q_st + Qi;
                    consider(lo, other, is_left, w0);
                    if (special) consider(lo + len - 1, other, is_left, w1); // the special base (the last one in op order) starts
                    if (i == ib) consider(lo + len, other, is_left, w2);
                } else {
                    const uint64_t lo = v.q_en - Qi - len; // G(p) = -W(q_en - p); positions fall as the ops go on
                    consider(lo + len, other, is_left, -w0);
                    if (special) consider(lo + 1, other, is_left, -w1); // the special base (lowest position) ends
                    if (i == ib) consider(lo, other, is_left, -w2);
                }
            }
        };
        candidates(L, R, true, La.i, Lb.i);
        candidates(R, L, false, Ra.i, Rb.i);
        if (lane == 0) consider(en_ovl, R, true, G_en(L, La, Lb, lxa, lxb));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int64_t ob = __shfl_xor(cb, off, 64);
            const uint64_t ok = __shfl_xor(ck, off, 64);
            if (ob > cb || (ob == cb && ok < ck)) cb = ob, ck = ok;
        }
        if (cb > best) best = cb, best_idx = ck;
    }
    if (L.bad || R.bad) {
        pending(4);
        return;
    }
    w.split_idx = best_idx;
    w.split_score = (int32_t)best;
#if RB_TW_STOP == 4
    if (lane == 0) p.rows[pi] = w;
    return;
#endif
    const uint64_t split = st_ovl + best_idx;
    const uint64_t ob = p.pair_out_off[pi];
    rb_wcut cutL, cutR;
    const bool inpl = p.in_place != 0;
    uint32_t st = rb_tw_clip(L, L.q_st, split, p.out_ops + ob, &w, 0, ob, lane, inpl ? &cutL : nullptr, (uint64_t)(L.ops - p.ops)); // trim_overlap.rs:77
#if RB_TW_STOP == 5
    if (lane == 0) p.rows[pi] = w;
    return;
#endif
    if (st == RB_ST_OK && !L.bad) {
        const uint64_t ob2 = ob + L.n;
        st = rb_tw_clip(R, split, R.q_en, p.out_ops + ob2, &w, 1, ob2, lane, inpl ? &cutR : nullptr, (uint64_t)(R.ops - p.ops)); // :78
    }
    if (L.bad || R.bad) { // a boundary the region cannot answer: the serial kernel does the pair (it rewrites both clips)
        pending(L.bad ? 5 : 6);
        return;
    }
    if (inpl && st == RB_ST_OK && lane == 0) { // both clips stand: their end words, where they are (first before last: one op -> the same word twice)
        p.out_ops[cutL.at_first] = cutL.w_first, p.out_ops[cutL.at_last] = cutL.w_last;
        p.out_ops[cutR.at_first] = cutR.w_first, p.out_ops[cutR.at_last] = cutR.w_last;
    }
    w.status = st;
    w._pad = 1; // (diagnostic: done by the wave kernel; the serial kernel leaves 0)
    if (lane == 0) p.rows[pi] = w;
}
// the list (or, without one, every row) as the attempts behind the first walk it: a workgroup looks at 64 entries at a time, one per
// lane, and does those that are still pending one after the other (round 6; one entry at a time, two dependent loads each, took a
// 48-workgroup attempt 0.35 ms to find out that a list of 150,000 pairs held nothing for it)
struct rb_tw_walker {
    uint64_t n, e0;
    unsigned long long todo;
    uint32_t pi; // per lane: the entry this lane looked at
};
__device__ __forceinline__ void rb_tw_walk_begin(const rb_trim_params &p, rb_tw_walker &w) {
    w.n = p.pend_list ? rb_first64(*p.pend) : p.n_pairs;
    w.e0 = 0, w.todo = 0ull, w.pi = 0u;
}
__device__ __forceinline__ bool rb_tw_walk_next(const rb_trim_params &p, rb_tw_walker &w, uint64_t *pi) { // wave-uniform
    while (!w.todo) {
        // (workgroup b owns the entries b, b + G, b + 2 G ...: a short list is spread over the workgroups, a long one is looked at 64
        //  entries at a time)
        if (w.e0 * gridDim.x + blockIdx.x >= w.n) return false;
        const uint64_t e = (w.e0 + (uint64_t)rb_lane()) * gridDim.x + blockIdx.x;
        w.pi = e < w.n ? (p.pend_list ? p.pend_list[e] : (uint32_t)e) : 0u;
        w.todo = rb_ballot(e < w.n && p.rows[w.pi].status == RB_ST_PENDING_INTERNAL);
        w.e0 += 64u;
    }
    const int l = __builtin_ctzll(w.todo);
    w.todo &= w.todo - 1ull;
    *pi = (uint64_t)rb_readlane<uint32_t>(w.pi, l);
    return true;
}
// first attempt: a wavefront per pair
#ifndef RB_TW_WPE
#define RB_TW_WPE 8
#endif
template <int CAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RB_TW_WPE))) void rb_k_overlap_split_wave(rb_trim_params p) {
    __shared__ uint32_t lds_w[2][3][CAP + 1];
    __shared__ uint32_t lds_c[2][3][CAP / 16 + 2];
    if (blockIdx.x < p.n_pairs) rb_tw_pair<CAP>(p, blockIdx.x, lds_w, lds_c);
}
// third attempt: the region arrays of a wavefront live in a slab of device memory (same code: the arrays are pointers).  Stores
// and loads of one wavefront go through its CU's vector L1 in program order, so a lane sees what another lane of its own wave
// has stored (wavefront-scope fences are empty on this target for exactly that reason).
template <int CAP>
__global__ __launch_bounds__(64) void rb_k_overlap_split_wave_scratch(rb_trim_params p) {
    if (!p.scratch || blockIdx.x >= p.scratch_blocks) return;
    uint32_t *slab = p.scratch + (size_t)blockIdx.x * RB_TW_SLAB_WORDS(CAP);
    auto *aw = reinterpret_cast<uint32_t (*)[3][CAP + 1]>(slab);
    auto *ac = reinterpret_cast<uint32_t (*)[3][CAP / 16 + 2]>(slab + 2u * 3u * (CAP + 1u));
    rb_tw_walker wk;
    rb_tw_walk_begin(p, wk);
    for (uint64_t pi; rb_tw_walk_next(p, wk, &pi);) {
        rb_tw_pair<CAP>(p, pi, aw, ac);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
}
// second attempt, for the pairs the first one left: a few wavefronts with a large region each walk the list
template <int CAP>
__global__ __launch_bounds__(64) void rb_k_overlap_split_wave_pending(rb_trim_params p) {
    __shared__ uint32_t lds_w[2][3][CAP + 1];
    __shared__ uint32_t lds_c[2][3][CAP / 16 + 2];
    rb_tw_walker wk;
    rb_tw_walk_begin(p, wk);
    for (uint64_t pi; rb_tw_walk_next(p, wk, &pi);) {
        rb_tw_pair<CAP>(p, pi, lds_w, lds_c);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the LDS arrays are reused by the next pair)
        __builtin_amdgcn_wave_barrier();
    }
}
extern "C" hipError_t rb_launch_overlap_split_quad(const rb_trim_params *p, int t, bool from_list, hipStream_t stream); // k_trim4.hip
extern "C" size_t rb_trim_scratch_bytes(uint32_t blocks) { return (size_t)blocks * RB_TW_SLAB_WORDS(RB_TW_CAP3) * 4u; }
extern "C" hipError_t rb_launch_overlap_split(const rb_trim_params *p, hipStream_t stream) {
    if (p->n_pairs == 0) return hipSuccess;
    rb_trim_params q = *p;
    static const bool serial_only = getenv("RB_DEBUG_TRIM_SERIAL") != nullptr; // diagnostics: the general kernel for every pair
    if (!serial_only) {
        q.only_pending = 0;
        // first attempt: four pairs per wavefront (k_trim4.hip); what it lists goes to the wave-per-pair kernel.  Without a list (its
        // allocation failed) the wave-per-pair kernel looks at every pair, as it did before round 6.
        static const char *quad_env = getenv("RB_TRIM_QUAD"); // diagnostics: 0 = off, 4 / 8 = ops per lane of the first attempt's regions
        const int quad_t = quad_env ? atoi(quad_env) : 4;
        if (q.pend_list && quad_t) {
            hipError_t e = rb_launch_overlap_split_quad(&q, quad_t, false, stream);
            if (e != hipSuccess) return e;
            q.only_pending = 2;
            if (quad_t < 8) { // the pairs whose overlap does not fit 64 ops of a record's end: 128
                e = rb_launch_overlap_split_quad(&q, 8, true, stream);
                if (e != hipSuccess) return e;
            }
            const unsigned g0 = (unsigned)(p->n_pairs < 8192 ? p->n_pairs : 8192);
            hipLaunchKernelGGL(rb_k_overlap_split_wave_pending<RB_TW_CAP>, dim3(g0), dim3(64), 0, stream, q);
        } else {
            hipLaunchKernelGGL(rb_k_overlap_split_wave<RB_TW_CAP>, dim3((unsigned)p->n_pairs), dim3(64), 0, stream, q);
        }
        q.only_pending = 2; // (the attempts behind the first walk its list; what they decline is listed already)
        const unsigned g1 = (unsigned)(p->n_pairs < 8192 ? p->n_pairs : 8192);
        hipLaunchKernelGGL(rb_k_overlap_split_wave_pending<RB_TW_CAP1>, dim3(g1), dim3(64), 0, stream, q);
        const unsigned g2 = (unsigned)(p->n_pairs < 2048 ? p->n_pairs : 2048);
        hipLaunchKernelGGL(rb_k_overlap_split_wave_pending<RB_TW_CAP2>, dim3(g2), dim3(64), 0, stream, q);
        if (q.scratch && q.scratch_blocks) {
            const unsigned g3 = (unsigned)(p->n_pairs < q.scratch_blocks ? p->n_pairs : q.scratch_blocks);
            hipLaunchKernelGGL(rb_k_overlap_split_wave_scratch<RB_TW_CAP3>, dim3(g3), dim3(64), 0, stream, q);
        }
    }
    q.only_pending = serial_only ? 0 : 1;
    if (serial_only) q.pend_list = nullptr;
    static const bool no_serial = getenv("RB_DEBUG_TRIM_NO_SERIAL") != nullptr; // diagnostics: leave what the wave kernels declined as it is
    if (no_serial) return hipGetLastError();
    const uint64_t sblocks = (q.only_pending && q.pend_list) ? std::min<uint64_t>((p->n_pairs + 63) / 64, 256) : (p->n_pairs + 63) / 64;
    hipLaunchKernelGGL(rb_k_overlap_split, dim3((unsigned)sblocks), dim3(64), 0, stream, q);
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


// ------------------------------------------------------------------------------------------------
// trim-paf: the pass driver's heavy half on the device (round 3).  Paf::overlapping_paf_recs (paf.rs:223-284) scans, per query
// name, all pairs of records for overlaps on the query (bed::get_overlap, bed.rs:74-85), flags contained records (:244-249),
// sorts ALL pairs by overlap (descending, stable) and then takes the first pair of every query name (:264-284).  Per query group
// that is: the pair with the LARGEST overlap, among equals the FIRST in scan order (i ascending, then j) -- a segmented arg-max, no
// global sort.  Groups are independent, so one pass = one launch: rb_k_trim_select (a thread per group; groups of more than
// RB_TS_BIG records by the whole wave, one after the other), an exclusive scan that gives the chosen pairs dense slots and their
// places in the ops arena, rb_k_trim_place.  The host keeps only the recursion loop (:286-288) and reads 64 bytes per pass.
// ------------------------------------------------------------------------------------------------
struct rb_tsel_params {
    uint64_t n_groups;
    const uint32_t *order;     // [n_rec] records stably sorted by query name
    const uint64_t *grp_off;   // [n_groups + 1] group g = order[grp_off[g] .. grp_off[g + 1])
    const rb_norm_row *norm;   // current coordinates / lengths of every record
    uint8_t *contained;        // [n_rec] by record: the flags of THIS pass (paf.rs:224: reset at every level)
    uint64_t *slot;            // [n_groups + 1] ops(left) + ops(right) of the group's pair (0: none); scanned in place: where its clips go
    uint64_t *has;             // [n_groups + 1] 1 for a group with a pair, else 0; scanned in place: the pair's dense slot
    uint32_t *cand;            // [2 n_groups] the chosen (left, right) records of each group
    uint64_t out_base;
    uint32_t *left, *right;    // dense outputs
    uint64_t *pair_out_off;
    rb_trim_pass *pass;
};
#define RB_TS_BIG 48u

struct rb_tsel_best {
    uint64_t ov;  // overlap (0: none yet)
    uint64_t ord; // scan order i * m + j of the pair that holds it
    uint32_t l, r;
};
__device__ __forceinline__ void rb_tsel_pair(const rb_tsel_params &p, uint32_t ri, uint32_t rj, uint64_t ord, rb_tsel_best &b, uint64_t &n_pairs) {
    const rb_norm_row *a = &p.norm[ri], *c = &p.norm[rj];
    const uint64_t st1 = a->q_st, en1 = a->q_en, st2 = c->q_st, en2 = c->q_en;
    const uint64_t mn = en1 < en2 ? en1 : en2, mx = st1 > st2 ? st1 : st2;
    if (mn <= mx) return;                       // bed.rs:74-85: no overlap
    const uint64_t ov = mn - mx;
    if (ov == en2 - st2) { p.contained[rj] = 1; return; } // paf.rs:244-249
    if (ov == en1 - st1) { p.contained[ri] = 1; return; }
    n_pairs++;
    if (ov > b.ov || (ov == b.ov && ord < b.ord)) {
        b.ov = ov, b.ord = ord;
        if (st1 <= st2) b.l = ri, b.r = rj; // the smaller q_st is "left" (:252-256)
        else b.l = rj, b.r = ri;
    }
}
template <bool BIG_ONLY> // BIG_ONLY: the groups of up to 16 records have been done by rb_k_trim_select_rows
__global__ __launch_bounds__(256) void rb_k_trim_select(rb_tsel_params p) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = rb_lane();
    const uint64_t gg0 = g < p.n_groups ? p.grp_off[g] : 0, gm = g < p.n_groups ? p.grp_off[g + 1] - gg0 : 0;
    const bool live = g < p.n_groups && (!BIG_ONLY || gm > 16u);
    const uint64_t g0 = live ? gg0 : 0, m = live ? gm : 0;
    for (uint64_t k = 0; k < m && m <= RB_TS_BIG; k++) p.contained[p.order[g0 + k]] = 0;
    rb_tsel_best b = {0, 0, 0, 0};
    uint64_t n_pairs = 0;
    if (live && m <= RB_TS_BIG) {
        for (uint64_t i = 0; i + 1 < m; i++) {
            const uint32_t ri = p.order[g0 + i];
            for (uint64_t j = i + 1; j < m; j++) rb_tsel_pair(p, ri, p.order[g0 + j], i * m + j, b, n_pairs);
        }
    }
    // big groups of this wave: all lanes on one group at a time (lane l takes the pairs whose j is l mod 64)
    unsigned long long big = __ballot(live && m > RB_TS_BIG);
    while (big) {
        const int src = __builtin_ctzll(big);
        big &= big - 1ull;
        const uint64_t bg0 = __shfl(g0, src, 64), bm = __shfl(m, src, 64);
        for (uint64_t k = (uint64_t)lane; k < bm; k += 64) p.contained[p.order[bg0 + k]] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        rb_tsel_best bb = {0, 0, 0, 0};
        uint64_t np = 0;
        for (uint64_t i = 0; i + 1 < bm; i++) {
            const uint32_t ri = p.order[bg0 + i];
            for (uint64_t j = i + 1 + (uint64_t)lane; j < bm; j += 64) rb_tsel_pair(p, ri, p.order[bg0 + j], i * bm + j, bb, np);
        }
        for (int off = 32; off > 0; off >>= 1) { // the wave's best: largest overlap, then smallest scan order
            const uint64_t oov = __shfl_xor(bb.ov, off, 64), oord = __shfl_xor(bb.ord, off, 64);
            const uint32_t ol = (uint32_t)__shfl_xor((int)bb.l, off, 64), orr = (uint32_t)__shfl_xor((int)bb.r, off, 64);
            if (oov > bb.ov || (oov == bb.ov && oov != 0 && oord < bb.ord)) bb.ov = oov, bb.ord = oord, bb.l = ol, bb.r = orr;
            np += __shfl_xor(np, off, 64);
        }
        if (lane == src) b = bb, n_pairs = np;
    }
    if (!live) return;
    p.slot[g] = b.ov ? (uint64_t)p.norm[b.l].n_ops + (uint64_t)p.norm[b.r].n_ops : 0ull;
    // has: 1 for a group with a pair, and in the high half the pairs the group leaves for a later pass (one pair per name and pass, :266-284):
    // the scan that gives the pairs their dense slots sums those as well (no atomic: 2.5e6 adds to one word were most of a pass's selection)
    // (a group's share is capped at (2^32 - 1) / n_groups so that the sum cannot leave its 32 bits: n_deferred is 0 exactly when nothing is left,
    //  and the exact count whenever no single group leaves more than that)
    const uint64_t dcap = 0xFFFFFFFFull / p.n_groups, dleft = n_pairs > 1 ? n_pairs - 1 : 0ull;
    p.has[g] = (b.ov ? 1ull : 0ull) | ((dleft < dcap ? dleft : dcap) << 32);
    p.cand[2 * g] = b.l, p.cand[2 * g + 1] = b.r;
}
// The same selection for groups of up to 16 records, a group per ROW of 16 lanes (round 6): lane j holds record j of the group -- ONE read of its
// norm row, where the thread-per-group form above walks every pair with four strided loads --, the outer index i runs row-uniform, record i's
// span reaches the lanes by ds_bpermute, lane j keeps the best pair (i, j) it has seen, and the row's best (largest overlap, then the smallest
// scan order i m + j) falls out of four rotate-and-compare steps.  Groups of more than 16 records are left to the kernel above (big_only).
__global__ __launch_bounds__(256) void rb_k_trim_select_rows(rb_tsel_params p) {
    const int lane = rb_lane();
    const uint32_t gbase = (uint32_t)lane & 48u, gl = (uint32_t)lane & 15u;
    const uint64_t g = ((uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u + ((uint32_t)lane >> 4);
    const bool live = g < p.n_groups;
    const uint64_t g0 = live ? p.grp_off[g] : 0, m64 = live ? p.grp_off[g + 1] - g0 : 0;
    const bool mine = live && m64 <= 16u; // (row-uniform)
    const uint32_t m = mine ? (uint32_t)m64 : 0u;
    const bool have = gl < m;
    const uint32_t r = have ? p.order[g0 + gl] : 0u;
    uint64_t st = 0, en = 0;
    if (have) st = p.norm[r].q_st, en = p.norm[r].q_en;
    bool cont = false;
    rb_tsel_best b = {0, 0, 0, 0};
    uint32_t np = 0; // candidate pairs this lane has seen as their j
    uint32_t w_m = m; // (the wavefront walks as far as its largest group)
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)w_m, off, 64);
        w_m = w_m > o ? w_m : o;
    }
    w_m = rb_first(w_m);
    for (uint32_t i = 0; i + 1u < w_m; i++) {
        // record i of every row's group, to all lanes of the row (rows whose group is shorter see zeros and have no lane behind i)
        const uint32_t ri = rb_row_read(r, gbase, i);
        const uint64_t st1 = ((uint64_t)rb_row_read((uint32_t)(st >> 32), gbase, i) << 32) | rb_row_read((uint32_t)st, gbase, i);
        const uint64_t en1 = ((uint64_t)rb_row_read((uint32_t)(en >> 32), gbase, i) << 32) | rb_row_read((uint32_t)en, gbase, i);
        const bool pair = have && gl > i && i + 1u < m;
        const uint64_t mn = en1 < en ? en1 : en, mx = st1 > st ? st1 : st;
        const bool ovl = pair && mn > mx;                      // bed.rs:74-85
        const uint64_t ov = ovl ? mn - mx : 0;
        const bool c2 = ovl && ov == en - st;                  // paf.rs:244-249: record j is contained
        const bool c1 = ovl && !c2 && ov == en1 - st1;         // ... record i is
        cont |= c2;
        if (rb_row_ballot(c1, gbase) != 0u && gl == i) cont = true;
        if (ovl && !c2 && !c1) {
            np++;
            const uint64_t ord = (uint64_t)i * m + gl;
            if (ov > b.ov || (ov == b.ov && ord < b.ord)) {
                b.ov = ov, b.ord = ord;
                if (st1 <= st) b.l = ri, b.r = r; // the smaller q_st is "left" (:252-256)
                else b.l = r, b.r = ri;
            }
        }
    }
    // the row's best pair and its number of candidates
    uint32_t np_row = rb_row_sum(np);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        const uint64_t oov = ((uint64_t)rb_row_ror((uint32_t)(b.ov >> 32), off) << 32) | rb_row_ror((uint32_t)b.ov, off);
        const uint64_t oord = ((uint64_t)rb_row_ror((uint32_t)(b.ord >> 32), off) << 32) | rb_row_ror((uint32_t)b.ord, off);
        const uint32_t ol = rb_row_ror(b.l, off), orr = rb_row_ror(b.r, off);
        if (oov > b.ov || (oov == b.ov && oov != 0 && oord < b.ord)) b.ov = oov, b.ord = oord, b.l = ol, b.r = orr;
    }
    if (have) p.contained[r] = cont ? 1 : 0;
    if (mine && gl == 0u) {
        p.slot[g] = b.ov ? (uint64_t)p.norm[b.l].n_ops + (uint64_t)p.norm[b.r].n_ops : 0ull;
        const uint64_t dcap = 0xFFFFFFFFull / p.n_groups, dleft = np_row > 1u ? np_row - 1u : 0u; // (the cap: rb_k_trim_select)
        p.has[g] = (b.ov ? 1ull : 0ull) | ((dleft < dcap ? dleft : dcap) << 32); // (high half: pairs left for a later pass)
        p.cand[2 * g] = b.l, p.cand[2 * g + 1] = b.r;
    }
}
__global__ __launch_bounds__(256) void rb_k_trim_place(rb_tsel_params p) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= p.n_groups) return;
    const uint64_t h0 = p.has[g], h1 = p.has[g + 1]; // scanned: low half = pairs in front of the group, high half = deferred pairs in front of it
    const uint64_t k = h0 & 0xFFFFFFFFull, k1 = h1 & 0xFFFFFFFFull;
    if (g + 1 == p.n_groups) p.pass->n_pairs = k1, p.pass->n_deferred = h1 >> 32, p.pass->ops_end = p.out_base + p.slot[g + 1];
    if (k1 == k) return; // no pair in this group
    p.left[k] = p.cand[2 * g], p.right[k] = p.cand[2 * g + 1];
    p.pair_out_off[k] = p.out_base + p.slot[g];
}
// the worst status of a pass's pair rows (0 = every pair was cut), for the host's one read per pass
__global__ __launch_bounds__(256) void rb_k_trim_check(const rb_pair_row *rows, uint64_t n_pairs, rb_trim_pass *pass) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_pairs && rows[k].status != RB_ST_OK) atomicMax(&pass->bad_status, rows[k].status);
}
extern "C" hipError_t rb_launch_exclusive_scan(uint64_t *v, uint64_t n, uint64_t *block_sums, uint64_t *total_out, hipStream_t stream);
extern "C" hipError_t rb_fill_async(void *dst, int value, size_t bytes, hipStream_t stream);
extern "C" hipError_t rb_launch_trim_select(const rb_tsel_params *p, uint64_t *block_sums, hipStream_t stream) {
    hipError_t e = rb_fill_async(p->pass, 0, sizeof(rb_trim_pass), stream); // (the library's own fill kernel: capi.hip says why)
    if (e != hipSuccess) return e;
    if (p->n_groups == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((p->n_groups + 255) / 256);
    static const bool rows_off = getenv("RB_TRIM_SELECT_ROWS") && atoi(getenv("RB_TRIM_SELECT_ROWS")) == 0; // diagnostics: the thread-per-group form for every group
    if (rows_off) {
        hipLaunchKernelGGL(rb_k_trim_select<false>, dim3(blocks), dim3(256), 0, stream, *p);
    } else {
        hipLaunchKernelGGL(rb_k_trim_select_rows, dim3((unsigned)((p->n_groups + 15) / 16)), dim3(256), 0, stream, *p);
        hipLaunchKernelGGL(rb_k_trim_select<true>, dim3(blocks), dim3(256), 0, stream, *p);
    }
    e = rb_launch_exclusive_scan(p->slot, p->n_groups, block_sums, nullptr, stream);
    if (e != hipSuccess) return e;
    e = rb_launch_exclusive_scan(p->has, p->n_groups, block_sums, nullptr, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(rb_k_trim_place, dim3(blocks), dim3(256), 0, stream, *p);
    return hipGetLastError();
}
extern "C" hipError_t rb_launch_trim_check(const rb_pair_row *rows, uint64_t n_pairs, rb_trim_pass *pass, hipStream_t stream) {
    if (n_pairs == 0) return hipSuccess;
    hipLaunchKernelGGL(rb_k_trim_check, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, stream, rows, n_pairs, pass);
    return hipGetLastError();
}
