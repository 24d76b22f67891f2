// k_trim4.hip -- trim-paf pair kernel, FOUR pairs per wavefront (gfx950): trim_overlapping_pafs (trim_overlap.rs:36-86) followed by
// truncate_record_by_query on both records (paf.rs:785-823), for the common case -- both records REGULAR (rb_norm_row.flags), modern
// binary-search policy, and the overlap within 16 T ops of one END of each record.
//
// Why: the pairs of a pass overlap at the ends of their records (the left record's last query bases, the right record's first), a few
// dozen ops; the wave-per-pair kernel (k_trim.hip) gives each of them 64 lanes and ~4100 instructions, most of them wave-uniform, and is
// bound by instruction issue (profiles/r03_c4_summary.md).  Here a pair is a ROW of 16 lanes -- the unit the DPP row operations work
// on, so every scan and sum stays inside the pair's own lanes without LDS -- and one instruction stream serves four pairs:
//   * the region of a record is its first or last 16 T ops, fetched with T / 4 unaligned 16-byte loads per lane, all of them (both
//     records) in flight before anything is looked at; lane j owns ops [j T, (j + 1) T) of the region: prefixes are serial inside a
//     lane and ONE 16-lane DPP scan across the lanes per quantity (the wave kernel scans every 64-op step six times);
//   * the prefixes are absolute (a region at the record's end starts from the record's totals, known from its row, minus its own
//     sums), so what lies outside the region is never read, but for the record's first two ops and its last one, fetched with the rest;
//   * searches by query offset: a ballot over the 16 lanes' chunk bases, then the T ops of that chunk side by side; per-lane searches
//     of the split candidates: a branch-free descent through the query prefixes in LDS.
// Same arrays, same formulas and the same order of exits as rb_tw_pair (k_trim.hip) -- the two are checked against each other and
// against the oracle by tests/test_gpu_trim.py and tests/soak/soak_trim.py.  A pair this kernel does not take (irregular record, legacy
// policy, region too small, a walk that leaves the region, a non-query run of T ops) is listed in pend_list and done by the kernels
// behind it: the wave-per-pair kernel with its larger regions, then the serial one.
#include "rb_trim.h"

// ---- a record's region in LDS -----------------------------------------------------------------------------------------------------
template <int T>
struct rb_q4_slab {
    uint32_t w[16 * T + 4];  // the op words of the region; [m ..] = zero-length M (ends every D / N run, contains nothing)
    uint32_t Qc[16 * T + 4]; // query bases before op i0 + k (absolute); [m ..] = at the region's end
    int32_t SP[16 * T + 4];  // score of the region's query bases before op i0 + k, in op order (rb_tw_stage, k_trim.hip)
    uint32_t pad[28];        // [0, 20): where a record's cut-only state waits during the split, and the left cut's results during the right cut.
                             // Two slabs = 96 T + 80 words: 16 mod 64 for T = 4 and 8 -- the four pairs of a wavefront, reading the same index
                             // of their own slabs, land in four different quarters of the 64 banks.  (The units / reference bases in front of the
                             // 16 chunks are NOT here: lane c holds chunk c's -- rb_qrec.cu / .cr --, a ds_bpermute away: 7.4 KB per wavefront at
                             // T = 4, so that the five wavefronts per SIMD the registers allow all find room)
};
static_assert((2 * sizeof(rb_q4_slab<4>) / 4) % 64 == 16 && (2 * sizeof(rb_q4_slab<8>) / 4) % 64 == 16, "slab stride");

struct rb_qrec { // row-uniform values, one copy per lane
    const uint32_t *ops;
    uint32_t n, i0, m;
    uint64_t t_st, t_en, q_st, q_en;
    bool minus, bad;
    uint32_t N, Qtot, Rtot;
    uint32_t bQ, eQ;           // query bases before the region / before its end
    uint32_t lastq;            // the region's last query op (absolute index); n: none
    uint32_t w0, w1, wl;       // the record's first, second and last op word
#ifdef RB_Q4_DEBUG
    uint32_t dbg[5];
#endif
    uint32_t cq, cu, cr;       // PER LANE: query bases / units / reference bases before this lane's chunk (op i0 + T lane), absolute
};
struct rb_qpos {
    uint32_t i, w, pre;
};
struct rb_qend {
    uint32_t k;
    rb_qpos o;
    uint32_t R, Q;
};
struct rb_qcut {
    uint64_t at_first, at_last;
    uint32_t w_first, w_last;
};

template <int T>
__device__ __forceinline__ void rb_q4_load(rb_qrec &v, uint32_t gl, uint32_t (&t)[T]) {
    const uint32_t *src = v.ops + v.i0;
    if (v.m == 16u * T) { // (the usual case: the record is longer than a region)
#pragma unroll
        for (int c = 0; c < T / 4; c++) {
            const uint4 q = rb_load4_unaligned(src + gl * T + 4 * c);
            t[4 * c] = q.x, t[4 * c + 1] = q.y, t[4 * c + 2] = q.z, t[4 * c + 3] = q.w;
        }
    } else { // a record of fewer ops: word by word, addresses behind it re-read its last op
#pragma unroll
        for (int e = 0; e < T; e++) {
            const uint32_t k = gl * T + (uint32_t)e;
            t[e] = src[k < v.m ? k : v.m - 1u];
        }
    }
    v.w0 = v.ops[0], v.w1 = v.ops[v.n > 1u ? 1u : 0u], v.wl = v.ops[v.n - 1u];
}

// prefixes of the region, into LDS.  false: a chunk of T ops without a query op (the run behind a last base would cross a lane).
template <int T>
__device__ __forceinline__ bool rb_q4_build(rb_qrec &v, rb_q4_slab<T> &S, uint32_t gl, uint32_t gbase, uint32_t (&t)[T], bool from_end, int32_t ms,
                                            int32_t ds, int32_t is) {
    const uint32_t k0 = gl * T;
    uint32_t su = 0, sq = 0, sr = 0;
    bool lead = true;     // still inside the chunk's leading run of non-query ops
    int32_t efirst = 0;   // score of that run's last op
    int32_t hq = -1;      // the chunk's last query op
#pragma unroll
    for (int e = 0; e < T; e++) {
        const bool in = k0 + (uint32_t)e < v.m;
        const uint32_t w = in ? t[e] : 0u; // (behind the region: zero-length M)
        t[e] = w;
        const uint32_t opc = rb_opc(w), len = rb_len(w);
        const bool q = rb_in(RB_QRY_MASK, opc);
        su += len, sq += q ? len : 0u, sr += rb_in(RB_REF_MASK, opc) ? len : 0u;
        const bool nq = lead && !q;
        efirst = nq ? rb_tw_score(opc, ms, ds, is) : efirst;
        lead = nq;
        if (q && in) hq = e;
    }
    if (rb_row_ballot(lead, gbase)) return false;
    // the run behind this lane's last op begins in the next lane (behind the region: none -- see lastq)
    const bool nxt_nq = rb_row_next((uint32_t)(!rb_in(RB_QRY_MASK, rb_opc(t[0])))) != 0u;
    const int32_t nxt_e = (int32_t)rb_row_next((uint32_t)efirst);
    // score of every op's query bases: its own for all but the last one in op order, which takes the score of the last D / N op of the
    // run behind it (modern policy: the last equal element of qpos_aln)
    int32_t mm[T], ss = 0;
    {
        bool nnq = nxt_nq;
        int32_t ne = nxt_e;
#pragma unroll
        for (int e = T - 1; e >= 0; e--) {
            const uint32_t opc = rb_opc(t[e]), len = rb_len(t[e]);
            const int32_t own = rb_tw_score(opc, ms, ds, is);
            if (rb_in(RB_QRY_MASK, opc)) {
                mm[e] = len ? (int32_t)(len - 1u) * own + (nnq ? ne : own) : 0;
                nnq = false;
            } else {
                mm[e] = 0;
                ne = nnq ? ne : own;
                nnq = true;
            }
            ss += mm[e];
        }
    }
    const uint32_t iu = rb_row_scan_incl(su), iq = rb_row_scan_incl(sq), ir = rb_row_scan_incl(sr);
    const int32_t isc = (int32_t)rb_row_scan_incl((uint32_t)ss);
    const uint32_t tu = rb_row_last(iu), tq = rb_row_last(iq), tr = rb_row_last(ir);
    const uint32_t bU = from_end ? v.N - tu : 0u, bQ = from_end ? v.Qtot - tq : 0u, bR = from_end ? v.Rtot - tr : 0u;
    uint32_t cq = bQ + iq - sq;
    int32_t cs = isc - ss;
    v.cq = cq, v.bQ = bQ, v.eQ = bQ + tq;
#ifdef RB_Q4_DEBUG
    v.dbg[0] = bU, v.dbg[1] = bR, v.dbg[2] = tu, v.dbg[3] = tr, v.dbg[4] = tq;
#endif
    v.cu = bU + iu - su, v.cr = bR + ir - sr;
    uint32_t qv[T];
    int32_t sv[T];
#pragma unroll
    for (int e = 0; e < T; e++) {
        qv[e] = cq, sv[e] = cs;
        cq += rb_in(RB_QRY_MASK, rb_opc(t[e])) ? rb_len(t[e]) : 0u, cs += mm[e];
    }
#pragma unroll
    for (int c = 0; c < T / 4; c++) {
        *reinterpret_cast<uint4 *>(&S.w[k0 + 4 * c]) = make_uint4(t[4 * c], t[4 * c + 1], t[4 * c + 2], t[4 * c + 3]);
        *reinterpret_cast<uint4 *>(&S.Qc[k0 + 4 * c]) = make_uint4(qv[4 * c], qv[4 * c + 1], qv[4 * c + 2], qv[4 * c + 3]);
        *reinterpret_cast<int4 *>(&S.SP[k0 + 4 * c]) = make_int4(sv[4 * c], sv[4 * c + 1], sv[4 * c + 2], sv[4 * c + 3]);
    }
    if (gl == 15u) { // the entry behind the last op: the prefixes at the region's end
        S.w[16 * T] = 0u, S.Qc[16 * T] = cq, S.SP[16 * T] = cs;
    }
    // the region's last query op
    const uint32_t hm = rb_row_ballot(hq >= 0, gbase);
    const uint32_t hl = hm ? 31u - (uint32_t)__builtin_clz(hm) : 0u;
    const uint32_t hv = rb_row_read((uint32_t)hq, gbase, hl);
    v.lastq = hm ? v.i0 + hl * T + hv : v.n;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return true;
}

// the query op of the region that holds query offset x (row-uniform); i = n: none
template <int T>
__device__ __forceinline__ rb_qpos rb_q4_find(const rb_qrec &v, const rb_q4_slab<T> &S, uint32_t x, uint32_t gl, uint32_t gbase) {
    rb_qpos o;
    o.i = v.n, o.w = RB_NULL_OP, o.pre = 0;
    const uint32_t lem = rb_row_ballot(v.cq <= x, gbase);
    if (!lem) return o; // in front of the region
    const uint32_t c = 31u - (uint32_t)__builtin_clz(lem);
    const uint32_t k = c * T + (gl < (uint32_t)T ? gl : 0u);
    const uint32_t w = S.w[k], pre = S.Qc[k];
    const bool hit = gl < (uint32_t)T && rb_in(RB_QRY_MASK, rb_opc(w)) && pre <= x && x - pre < rb_len(w);
    const uint32_t hm = rb_row_ballot(hit, gbase);
    if (!hm) return o; // behind the region (or behind the record)
    const uint32_t kk = c * T + (uint32_t)__builtin_ctz(hm);
    o.i = v.i0 + kk, o.w = S.w[kk], o.pre = S.Qc[kk];
    return o;
}
// units (KIND 0) / reference bases (KIND 1) before op i of the region: its chunk's base + the ops of the chunk in front of it
template <int T, int KIND>
__device__ __forceinline__ uint32_t rb_q4_before(const rb_qrec &v, const rb_q4_slab<T> &S, uint32_t i, uint32_t gl, uint32_t gbase) {
    const uint32_t k = i - v.i0, c = k / (uint32_t)T, j = c * T + gl;
    uint32_t x = 0;
    if (gl < (uint32_t)T && j < k) {
        const uint32_t w = S.w[j];
        x = (KIND == 0 || rb_in(RB_REF_MASK, rb_opc(w))) ? rb_len(w) : 0u;
    }
    return rb_row_read(KIND == 0 ? v.cu : v.cr, gbase, c) + rb_row_sum(x);
}
// truncate_record_by_query (paf.rs:785-823) on a staged regular record: rb_tw_clip (k_trim.hip) for a row of 16 lanes
template <int T>
__device__ __forceinline__ uint32_t rb_q4_clip(rb_qrec &v, const rb_q4_slab<T> &S, uint64_t new_q_st, uint64_t new_q_en, uint32_t *out, rb_pair_row *row,
                                               int s, uint64_t out_base, uint32_t gl, uint32_t gbase, bool in_place, rb_qcut &cut, uint64_t rec_base) {
    if (!(new_q_st >= v.q_st) || !(new_q_en <= v.q_en) || new_q_en == 0) return RB_ST_PANIC_ASSERT; // :787-788
    if (new_q_en <= new_q_st) { // an empty range: the serial kernel says what the reference does with it
        v.bad = true;
        return RB_ST_OK;
    }
    const uint32_t n = v.n, N = v.N;
    // the match-type unit truncate_record_by_query ends up at for query position p: qpos_to_idx_match (paf.rs:564-590) = the last
    // unit whose qpos equals p (modern policy), then the nearest match-type unit in the search direction
    auto resolve = [&](uint64_t p, bool search_up, rb_qend *e) -> bool {
        if (p < v.q_st || p >= v.q_en) return false;
        const uint32_t x = (uint32_t)(v.minus ? v.q_en - 1 - p : p - v.q_st);
        if (x < v.bQ || x >= v.eQ) {
            // outside the region: only the record's own first / last query base is asked for there.  A regular record starts and
            // ends on a match op; its last base is its last unit, its first base its first unit unless that op has one base and a
            // D / N run behind it (the run repeats the position: left to the kernels behind this one)
            if (x == 0u) {
                if (rb_len(v.w0) < 2u && n > 1u && !rb_in(RB_QRY_MASK, rb_opc(v.w1))) return false;
                e->k = 0, e->o.i = 0, e->o.w = v.w0, e->o.pre = 0, e->R = 0, e->Q = 0;
                return true;
            }
            if (x + 1u == v.Qtot) {
                const uint32_t len = rb_len(v.wl);
                e->k = N - 1u, e->o.i = n - 1u, e->o.w = v.wl, e->o.pre = N - len, e->R = v.Rtot - len, e->Q = v.Qtot - len;
                return true;
            }
            return false;
        }
        const rb_qpos o = rb_q4_find<T>(v, S, x, gl, gbase);
        if (o.i >= n) return false;
        const uint32_t j = x - o.pre, len = rb_len(o.w);
        const uint32_t ub = rb_q4_before<T, 0>(v, S, o.i, gl, gbase);
        uint32_t u = ub + j;
        rb_qpos om;
        om.i = o.i, om.w = o.w, om.pre = ub;
        if (j + 1u == len) { // last base of the op: the D / N units behind it repeat its position, and the last of them is the unit
            uint32_t k2 = o.i - v.i0 + 1u;
            for (; k2 < v.m && !rb_in(RB_QRY_MASK, rb_opc(S.w[k2])); k2++) {
                const uint32_t w2 = S.w[k2];
                om.i = v.i0 + k2, om.w = w2, om.pre = u + 1u;
                u += rb_len(w2);
            }
            if (k2 >= v.m && v.i0 + v.m < n) return false; // (the run leaves the region)
        }
        // nearest match-type unit, up (paf.rs:581-583) or down (:585-587)
        uint32_t km = u;
        if (!rb_in(RB_MATCH_MASK, rb_opc(om.w))) {
            if (search_up) {
                uint32_t uu = om.pre + rb_len(om.w), k2 = om.i - v.i0 + 1u;
                for (; k2 < v.m && !rb_in(RB_MATCH_MASK, rb_opc(S.w[k2])); k2++) uu += rb_len(S.w[k2]);
                if (k2 >= v.m) return false; // (no match op behind it inside the region; at the record's end the reference panics: serial kernel)
                km = uu;
                om.i = v.i0 + k2, om.w = S.w[k2], om.pre = uu;
            } else {
                uint32_t uu = om.pre, k2 = om.i - v.i0;
                bool got = false;
                while (k2 > 0u) {
                    k2--;
                    if (rb_in(RB_MATCH_MASK, rb_opc(S.w[k2]))) {
                        got = true;
                        break;
                    }
                    uu -= rb_len(S.w[k2]);
                }
                if (!got) return false;
                km = uu - 1u;
                om.i = v.i0 + k2, om.w = S.w[k2], om.pre = uu - rb_len(S.w[k2]);
            }
        }
        e->k = km, e->o = om, e->R = rb_q4_before<T, 1>(v, S, om.i, gl, gbase), e->Q = S.Qc[om.i - v.i0];
        return true;
    };
    rb_qend A, B; // paf.rs:792-796: the start searches up on '+' and down on '-', the end the other way
    if (!resolve(new_q_st, !v.minus, &A) || !resolve(new_q_en - 1, v.minus, &B)) {
        v.bad = true;
        return RB_ST_OK;
    }
    auto unit = [&](const rb_qend &e, uint64_t *tpos, uint64_t *qpos) { // both are match-type units
        const uint32_t off = e.k - e.o.pre;
        *tpos = v.t_st + e.R + off;
        *qpos = v.minus ? v.q_en - 1 - e.Q - off : v.q_st + e.Q + off;
    };
    uint64_t tp, qp_st, qp_en;
    unit(A, &tp, &qp_st);
    unit(B, &tp, &qp_en);
    const uint64_t nq_st = qp_st, nq_en = qp_en + 1;
    if (A.k > B.k) { // :799-801
        const rb_qend t = A;
        A = B;
        B = t;
    }
    uint64_t t0, t1, qd;
    unit(A, &t0, &qd);
    unit(B, &t1, &qd);
    const uint64_t nt_st = t0, nt_en = t1 + 1; // :802-803
    // subset_cigar + collapse (:807-808): ops ia..ib with the first / last length cut; adjacent ops differ, nothing merges; both ends
    // are match-type units, so the strip of :819-822 removes nothing (rb_tw_clip says why nothing is summed)
    const uint32_t ia = A.o.i, ib = B.o.i, cnt = ib - ia + 1;
    const uint32_t lf = cnt == 1 ? B.k - A.k + 1u : A.o.pre + rb_len(A.o.w) - A.k, ll = cnt == 1 ? lf : B.k - B.o.pre + 1u;
    cut.at_first = rec_base + ia, cut.at_last = rec_base + ib;
    cut.w_first = (lf << 4) | rb_opc(A.o.w), cut.w_last = (ll << 4) | rb_opc(B.o.w);
    if (in_place) { // (nothing is copied; rec_base = where the record's kept ops begin in the ops array)
        out_base = rec_base + ia;
    } else {
        for (uint32_t j = gl; j < cnt; j += 16u) {
            const uint32_t wv = v.ops[ia + j];
            out[j] = j == 0 ? ((lf << 4) | rb_opc(wv)) : (j == cnt - 1 ? ((ll << 4) | rb_opc(wv)) : wv);
        }
    }
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

#ifndef RB_Q4_STOP
#define RB_Q4_STOP 0 // diagnostics (tools/r06_quad_decomp.sh): != 0 ends a pair early -- 1 behind the staging of both records, 2 behind the searches
                     // of the overlap's end ops, 3 behind the split, 4 behind the left clip; 9: the whole pair, but a declined pair is listed without the atomic counter
                     // (measured: 2.41 ms with it, 2.39 - 2.41 without, per 2.5e6 pairs); the rows are wrong then, only the time is of interest
#endif
template <int T>
__device__ __forceinline__ void rb_q4_pair(const rb_trim_params &p, const uint64_t pi, const uint32_t gbase, const uint32_t gl, const uint32_t g) {
    __shared__ __attribute__((aligned(16))) rb_q4_slab<T> lds[4][2]; // (here and not in the kernels: as an argument it would be a generic pointer)
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
    const rb_norm_row nl = p.norm[rl], nr = p.norm[rr];
    if (nl.status != RB_ST_OK || nr.status != RB_ST_OK) { // aligned_pairs() panics (paf.rs:273-274, :782)
        w.status = nl.status != RB_ST_OK ? nl.status : nr.status;
        if (gl == 0) p.rows[pi] = w;
        return;
    }
    auto pending = [&](uint32_t why) { // (why: diagnostics, RB_DEBUG_TRIM_NO_SERIAL; whoever does the pair rewrites the whole row)
        if (gl == 0) {
#if RB_Q4_STOP == 9 // (diagnostics, timing only: what the one counter costs -- the list is garbage)
            if (!p.only_pending) p.pend_list[pi] = (uint32_t)pi;
#else
            if (!p.only_pending) p.pend_list[atomicAdd(p.pend, 1ull)] = (uint32_t)pi; // (listed once: by the first attempt)
#endif
            p.rows[pi].status = RB_ST_PENDING_INTERNAL, p.rows[pi].split_idx = why;
        }
    };
    if (p.policy == RB_BSEARCH_LEGACY || !(nl.flags & RB_F_REGULAR) || !(nr.flags & RB_F_REGULAR) || nl.n_ops == 0 || nr.n_ops == 0) {
        pending(1);
        return;
    }
    const int32_t ms = p.match_score, ds = p.diff_score, is = p.indel_score;
    {   // the sums below are 32 bits wide: scores so large that a sum over both records' query bases could leave 2^29 are left to the
        // wave-per-pair kernel, which sums in 64 bits
        const uint32_t a = (uint32_t)(ms < 0 ? -ms : ms), b = (uint32_t)(ds < 0 ? -ds : ds), c = (uint32_t)(is < 0 ? -is : is);
        const uint64_t smax = a > b ? (a > c ? a : c) : (b > c ? b : c);
        if (smax * ((nl.q_en - nl.q_st) + (nr.q_en - nr.q_st)) >= (1ull << 29)) {
            pending(8);
            return;
        }
    }
    rb_qrec L, R;
    L.ops = p.ops + p.op_off[rl] + nl.first_op, L.n = nl.n_ops;
    L.t_st = nl.t_st, L.t_en = nl.t_en, L.q_st = nl.q_st, L.q_en = nl.q_en, L.minus = p.strand[rl] == (uint8_t)'-';
    L.N = nl.aln_len, L.Qtot = (uint32_t)(nl.q_en - nl.q_st), L.Rtot = (uint32_t)(nl.t_en - nl.t_st), L.bad = false;
    R.ops = p.ops + p.op_off[rr] + nr.first_op, R.n = nr.n_ops;
    R.t_st = nr.t_st, R.t_en = nr.t_en, R.q_st = nr.q_st, R.q_en = nr.q_en, R.minus = p.strand[rr] == (uint8_t)'-';
    R.N = nr.aln_len, R.Qtot = (uint32_t)(nr.q_en - nr.q_st), R.Rtot = (uint32_t)(nr.t_en - nr.t_st), R.bad = false;
    const uint64_t st_ovl = L.q_st > R.q_st ? L.q_st : R.q_st; // trim_overlap.rs:43-44
    const uint64_t en_ovl = L.q_en < R.q_en ? L.q_en : R.q_en;
    if (en_ovl <= st_ovl || st_ovl < L.q_st || en_ovl > L.q_en || st_ovl < R.q_st || en_ovl > R.q_en) { // (no overlap: the serial kernel says what the reference does)
        pending(2);
        return;
    }
    // query offsets of the overlap in each record's op order, and the end of the record they lie at
    auto span = [&](rb_qrec &v, uint32_t *xa, uint32_t *xb) -> bool {
        *xa = (uint32_t)(!v.minus ? st_ovl - v.q_st : v.q_en - en_ovl);
        *xb = (uint32_t)(!v.minus ? en_ovl - 1 - v.q_st : v.q_en - 1 - st_ovl);
        const bool from_end = v.n > 16u * T && *xa > v.Qtot - 1u - (*xb < v.Qtot ? *xb : v.Qtot - 1u);
        v.m = v.n < 16u * T ? v.n : 16u * T;
        v.i0 = from_end ? v.n - v.m : 0u;
        return from_end;
    };
    uint32_t lxa, lxb, rxa, rxb;
    const bool lfe = span(L, &lxa, &lxb), rfe = span(R, &rxa, &rxb);
    rb_q4_slab<T> &SL = lds[g][0], &SR = lds[g][1];
    uint32_t tl[T], tr[T];
    rb_q4_load<T>(L, gl, tl);
    rb_q4_load<T>(R, gl, tr);
    if (!rb_q4_build<T>(L, SL, gl, gbase, tl, lfe, ms, ds, is) || !rb_q4_build<T>(R, SR, gl, gbase, tr, rfe, ms, ds, is)) {
        pending(3);
        return;
    }
    // What only the CUTS need of a record -- where its ops lie, its coordinates, its totals, its first two ops and its last -- waits in the slab's
    // unused words while the split is found: held in registers through the searches it was a third of the kernel's 110 VGPRs, i.e. the
    // difference between four and five wavefronts per SIMD.  (Lane 0 of the row writes, every lane reads it back in front of the record's cut.)
    auto stash = [&](const rb_qrec &v, rb_q4_slab<T> &S) {
        if (gl == 0u) {
            const uint64_t base = (uint64_t)(v.ops - p.ops);
            uint32_t *z = S.pad;
            *reinterpret_cast<uint4 *>(z) = make_uint4((uint32_t)base, (uint32_t)(base >> 32), (uint32_t)v.t_st, (uint32_t)(v.t_st >> 32));
            *reinterpret_cast<uint4 *>(z + 4) = make_uint4((uint32_t)v.q_st, (uint32_t)(v.q_st >> 32), (uint32_t)v.q_en, (uint32_t)(v.q_en >> 32));
            *reinterpret_cast<uint4 *>(z + 8) = make_uint4(v.N, v.Qtot, v.Rtot, v.w0);
            z[12] = v.w1, z[13] = v.wl;
        }
    };
    auto unstash = [&](rb_qrec &v, const rb_q4_slab<T> &S) {
        asm volatile("" ::: "memory"); // (read where they are needed, not before)
        const uint32_t *z = S.pad;
        const uint4 a = *reinterpret_cast<const uint4 *>(z), b = *reinterpret_cast<const uint4 *>(z + 4), c = *reinterpret_cast<const uint4 *>(z + 8);
        v.ops = p.ops + (((uint64_t)a.y << 32) | a.x), v.t_st = ((uint64_t)a.w << 32) | a.z;
        v.q_st = ((uint64_t)b.y << 32) | b.x, v.q_en = ((uint64_t)b.w << 32) | b.z;
        v.N = c.x, v.Qtot = c.y, v.Rtot = c.z, v.w0 = c.w, v.w1 = z[12], v.wl = z[13];
    };
    stash(L, SL);
    stash(R, SR);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#if RB_Q4_STOP == 1
    if (gl == 0) p.rows[pi].split_idx = L.eQ + R.eQ;
    return;
#endif
    int32_t best = 0;
    uint32_t best_idx = 0;
    // the ops that hold the first and the last overlapped query base of each record
    const rb_qpos La = rb_q4_find<T>(L, SL, lxa, gl, gbase), Lb = rb_q4_find<T>(L, SL, lxb, gl, gbase);
    const rb_qpos Ra = rb_q4_find<T>(R, SR, rxa, gl, gbase), Rb = rb_q4_find<T>(R, SR, rxb, gl, gbase);
    if (La.i >= L.n || Lb.i >= L.n || Ra.i >= R.n || Rb.i >= R.n) { // the overlap is not inside the regions
        pending(4);
        return;
    }
    // the run behind the last overlapped base must end inside the region (its score is that base's score)
    if ((Lb.i >= L.lastq && L.i0 + L.m < L.n) || (Rb.i >= R.lastq && R.i0 + R.m < R.n)) {
        pending(7);
        return;
    }
#if RB_Q4_STOP == 2
    if (gl == 0) p.rows[pi].split_idx = La.i + Lb.i + Ra.i + Rb.i;
    return;
#endif
    // Everything below works in k = bases behind st_ovl (0 .. n), 32 bits, and on query offsets in op order, so that the two strands
    // share one instruction stream: the boundary k of the overlap is query offset x = xa + k on '+' and xb + 1 - k on '-' (op order
    // runs against the positions there), and G(k) = W(x) on '+', -W(x) on '-', W = score of the query bases in front of offset x.
    const uint32_t n_ovl = (uint32_t)(en_ovl - st_ovl);
    auto W_ends = [&](const rb_qrec &v, const rb_q4_slab<T> &S, const rb_qpos &oa, const rb_qpos &ob, uint32_t xa, uint32_t xb, int32_t *g0, int32_t *gn) {
        const int32_t Wa = S.SP[oa.i - v.i0] + (xa - oa.pre) * rb_tw_score(rb_opc(oa.w), ms, ds, is); // W(xa)
        const uint32_t kb = ob.i - v.i0;
        const int32_t Wb = xb + 1u < ob.pre + rb_len(ob.w) ? S.SP[kb] + (xb + 1u - ob.pre) * rb_tw_score(rb_opc(ob.w), ms, ds, is)
                                                            : S.SP[kb + 1u];                                 // W(xb + 1)
        *g0 = v.minus ? -Wb : Wa, *gn = v.minus ? -Wa : Wb;
    };
    {
        int32_t gl0, gl1, gr0, gr1;
        W_ends(L, SL, La, Lb, lxa, lxb, &gl0, &gl1);
        W_ends(R, SR, Ra, Rb, rxa, rxb, &gr0, &gr1);
        const int32_t rsum = gr1 - gr0; // f(0)
        if (rsum > best) best = rsum;   // (index stays 0)
        int32_t cb = INT32_MIN;         // best f over the candidates k > 0 of this lane; ties: the smaller k
        uint32_t ck = 0;
        if (gl == 0u) cb = gl1 - gl0, ck = n_ovl; // f(n): the left record's whole overlap
        // a candidate is a boundary where one record's score changes -- the end of a query op and, where the op's last base scores
        // differently (a D / N run behind it), the boundary in front of that base; that record's own sum up to it comes straight
        // from its prefix arrays, only the other record is searched.  Both records' candidates go through ONE loop, free of branches
        // but for a rare second search, so that the two searches of a step -- chains of dependent LDS reads -- run side by side.
        struct hit_t { // the other record at ITS offset xo
            int32_t W;
            uint32_t j, len;
            int32_t own;
            bool inside;
        };
        // the last index with Qc <= xo holds xo (the entries behind the region hold eQ): a 4-ary descent, three reads per level
        auto search = [&](const rb_qrec &o, const rb_q4_slab<T> &SO, uint32_t xo) -> hit_t {
            uint32_t lo = 0;
            rb_static_for<3>([&](auto lv) { // strides 4 T, T, T / 4
                constexpr uint32_t st = (4u * T) >> (2 * decltype(lv)::value);
                const uint32_t a = SO.Qc[lo + st], b = SO.Qc[lo + 2u * st], c = SO.Qc[lo + 3u * st];
                lo += ((a <= xo ? 1u : 0u) + (b <= xo ? 1u : 0u) + (c <= xo ? 1u : 0u)) * st;
            });
            if constexpr (T == 8) lo += SO.Qc[lo + 1u] <= xo ? 1u : 0u; // (128 = 2 * 4^3)
            hit_t h;
            const uint32_t wo = SO.w[lo];
            h.j = xo - SO.Qc[lo], h.len = rb_len(wo), h.own = rb_tw_score(rb_opc(wo), ms, ds, is), h.inside = xo < o.eQ;
            h.W = h.inside ? SO.SP[lo] + h.j * h.own : SO.SP[16 * T];
            return h;
        };
        struct late_t { // a special boundary the end's search did not answer
            bool need;
            uint32_t ks;
            int32_t W_own;
        };
        auto eval = [&](const rb_qrec &v, const rb_q4_slab<T> &S, uint32_t vxa, uint32_t vxb, const rb_qrec &o, const rb_q4_slab<T> &SO, uint32_t oxa, uint32_t oxb,
                        bool is_left, uint32_t i, uint32_t ib) -> late_t {
            auto update = [&](bool act, uint32_t k, int32_t W_own, int32_t Wo) {
                const int32_t g_own = v.minus ? -W_own : W_own, g_oth = o.minus ? -Wo : Wo;
                const int32_t f = is_left ? (g_own - gl0) + (gr1 - g_oth) : (g_oth - gl0) + (gr1 - g_own);
                const bool better = act && (f > cb || (f == cb && k < ck));
                cb = better ? f : cb, ck = better ? k : ck;
            };
            const bool in = i <= ib;
            const uint32_t kk = in ? i - v.i0 : 0u;
            const uint32_t wv = S.w[kk], len = rb_len(wv), xe = S.Qc[kk] + len;
            const int32_t Si = S.SP[kk], Sn = S.SP[kk + 1u], own = rb_tw_score(rb_opc(wv), ms, ds, is);
            const bool q = in && rb_in(RB_QRY_MASK, rb_opc(wv));
            const int32_t body = (int32_t)(len - 1u) * own; // (as the prefixes were built: 32 bits)
            // the op's end: boundary ke, the other record's offset there
            const uint32_t ke = v.minus ? vxb + 1u - xe : xe - vxa;
            const bool e_ok = q && ke - 1u < n_ovl; // 0 < k <= n
            const hit_t h = search(o, SO, e_ok ? (o.minus ? oxb + 1u - ke : oxa + ke) : o.bQ);
            update(e_ok, ke, Sn, h.W);
            // the boundary in front of the op's last base, where that base scores differently: one base towards the op's start,
            // which is one base down the other record's offsets on the same strand and one base up on the other -- the hit of the
            // end's search answers it unless the step leaves the other record's op (then it is searched: below)
            const bool same = v.minus == o.minus;
            late_t l;
            l.ks = v.minus ? ke + 1u : ke - 1u;
            l.W_own = Si + (len - 1u) * own;
            const bool sp = q && Sn - Si - body != own && l.ks - 1u < n_ovl;
            const bool easy = e_ok && h.inside && (same ? h.j >= 1u : h.j + 1u < h.len);
            update(sp && easy, l.ks, l.W_own, same ? h.W - h.own : h.W + h.own);
            l.need = sp && !easy;
            return l;
        };
        auto late = [&](const late_t &l, const rb_qrec &v, const rb_qrec &o, const rb_q4_slab<T> &SO, uint32_t oxa, uint32_t oxb, bool is_left) {
            if (!l.need) return;
            const int32_t Wo = search(o, SO, o.minus ? oxb + 1u - l.ks : oxa + l.ks).W;
            const int32_t g_own = v.minus ? -l.W_own : l.W_own, g_oth = o.minus ? -Wo : Wo;
            const int32_t f = is_left ? (g_own - gl0) + (gr1 - g_oth) : (g_oth - gl0) + (gr1 - g_own);
            if (f > cb || (f == cb && l.ks < ck)) cb = f, ck = l.ks;
        };
        {
            const uint32_t cl = Lb.i - La.i, cr = Rb.i - Ra.i, steps = (cl > cr ? cl : cr) / 16u + 1u;
            for (uint32_t it = 0, d = gl; it < steps; it++, d += 16u) {
                const late_t ll = eval(L, SL, lxa, lxb, R, SR, rxa, rxb, true, La.i + d, Lb.i);
                const late_t lr = eval(R, SR, rxa, rxb, L, SL, lxa, lxb, false, Ra.i + d, Rb.i);
                late(ll, L, R, SR, rxa, rxb, true);
                late(lr, R, L, SL, lxa, lxb, false);
            }
        }
        rb_static_for<4>([&](auto s_) { // rotate by 8, 4, 2, 1: every lane ends with the row's best
            constexpr int off = 8 >> decltype(s_)::value;
            const int32_t ob = (int32_t)rb_row_ror((uint32_t)cb, off);
            const uint32_t ok = rb_row_ror(ck, off);
            if (ob > cb || (ob == cb && ok < ck)) cb = ob, ck = ok;
        });
        if (cb > best) best = cb, best_idx = ck;
    }
#if RB_Q4_STOP == 3
    if (gl == 0) p.rows[pi].split_idx = best_idx;
    return;
#endif
    w.split_idx = best_idx;
    w.split_score = best;
    const uint64_t split = st_ovl + best_idx;
    const bool inpl = p.in_place != 0;
    const uint64_t ob = inpl ? 0ull : p.pair_out_off[pi];
    rb_qcut cutL, cutR;
    unstash(L, SL);
    uint32_t st = rb_q4_clip<T>(L, SL, L.q_st, split, p.out_ops + ob, &w, 0, ob, gl, gbase, inpl, cutL, (uint64_t)(L.ops - p.ops)); // trim_overlap.rs:77
#if RB_Q4_STOP == 4
    if (gl == 0) p.rows[pi] = w;
    return;
#endif
    // ... and what the left cut left (its half of the row, its two end words) waits there while the right record is cut
    if (gl == 0u) {
        uint32_t *y = SL.pad;
        *reinterpret_cast<uint4 *>(y) = make_uint4((uint32_t)w.t_st[0], (uint32_t)(w.t_st[0] >> 32), (uint32_t)w.t_en[0], (uint32_t)(w.t_en[0] >> 32));
        *reinterpret_cast<uint4 *>(y + 4) = make_uint4((uint32_t)w.q_st[0], (uint32_t)(w.q_st[0] >> 32), (uint32_t)w.q_en[0], (uint32_t)(w.q_en[0] >> 32));
        *reinterpret_cast<uint4 *>(y + 8) = make_uint4(w.nmatch[0], w.aln_len[0], (uint32_t)w.out_off[0], (uint32_t)(w.out_off[0] >> 32));
        *reinterpret_cast<uint4 *>(y + 12) = make_uint4(w.out_n[0], cutL.w_first, cutL.w_last, (uint32_t)cutL.at_first);
        *reinterpret_cast<uint4 *>(y + 16) = make_uint4((uint32_t)(cutL.at_first >> 32), (uint32_t)cutL.at_last, (uint32_t)(cutL.at_last >> 32), 0u);
    }
    unstash(R, SR);
    if (st == RB_ST_OK && !L.bad) {
        const uint64_t ob2 = ob + L.n;
        st = rb_q4_clip<T>(R, SR, split, R.q_en, p.out_ops + ob2, &w, 1, ob2, gl, gbase, inpl, cutR, (uint64_t)(R.ops - p.ops)); // :78
    }
    if (L.bad || R.bad) { // a boundary the region cannot answer
        pending(L.bad ? 5 : 6);
        return;
    }
    if (gl == 0) {
        {   // the left cut's half of the row and its end words, back from the slab
            asm volatile("" ::: "memory");
            const uint32_t *y = SL.pad;
            const uint4 a = *reinterpret_cast<const uint4 *>(y), b = *reinterpret_cast<const uint4 *>(y + 4), c = *reinterpret_cast<const uint4 *>(y + 8),
                        d = *reinterpret_cast<const uint4 *>(y + 12), e = *reinterpret_cast<const uint4 *>(y + 16);
            w.t_st[0] = ((uint64_t)a.y << 32) | a.x, w.t_en[0] = ((uint64_t)a.w << 32) | a.z, w.q_st[0] = ((uint64_t)b.y << 32) | b.x, w.q_en[0] = ((uint64_t)b.w << 32) | b.z;
            w.nmatch[0] = c.x, w.aln_len[0] = c.y, w.out_off[0] = ((uint64_t)c.w << 32) | c.z, w.out_n[0] = d.x;
            cutL.w_first = d.y, cutL.w_last = d.z, cutL.at_first = ((uint64_t)e.x << 32) | d.w, cutL.at_last = ((uint64_t)e.z << 32) | e.y;
        }
        if (inpl && st == RB_ST_OK) { // both clips stand: their end words, where they are (first before last: one op -> the same word twice)
            p.out_ops[cutL.at_first] = cutL.w_first, p.out_ops[cutL.at_last] = cutL.w_last;
            p.out_ops[cutR.at_first] = cutR.w_first, p.out_ops[cutR.at_last] = cutR.w_last;
        }
        w.status = st;
        w._pad = 1; // (diagnostic: done by a wave kernel; the serial kernel leaves 0)
#ifdef RB_Q4_DEBUG
        w.t_st[0] = ((uint64_t)L.dbg[0] << 32) | L.dbg[1], w.t_en[0] = ((uint64_t)L.dbg[2] << 32) | L.dbg[3], w.q_st[0] = ((uint64_t)L.dbg[4] << 32) | L.N, w.q_en[0] = ((uint64_t)L.i0 << 32) | L.m;
#endif
        p.rows[pi] = w;
    }
}

// first attempt: pairs 4 b .. 4 b + 3 to workgroup b
template <int T>
__global__ __launch_bounds__(64) void rb_k_overlap_split_quad(rb_trim_params p) {
    const uint32_t lane = (uint32_t)rb_lane(), gbase = lane & 48u, gl = lane & 15u, g = lane >> 4;
    const uint64_t pi = (uint64_t)blockIdx.x * 4u + g;
    if (pi < p.n_pairs) rb_q4_pair<T>(p, pi, gbase, gl, g);
}
// second attempt, with larger regions, for the pairs the first one listed: the workgroups walk the list, four entries at a time
template <int T>
__global__ __launch_bounds__(64) void rb_k_overlap_split_quad_list(rb_trim_params p) {
    const uint32_t lane = (uint32_t)rb_lane(), gbase = lane & 48u, gl = lane & 15u, g = lane >> 4;
    const uint64_t n = *p.pend;
    for (uint64_t e = (uint64_t)blockIdx.x * 4u + g; e < n; e += (uint64_t)gridDim.x * 4u) {
        const uint64_t pi = p.pend_list[e];
        if (p.rows[pi].status == RB_ST_PENDING_INTERNAL) rb_q4_pair<T>(p, pi, gbase, gl, g);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // (the slabs are reused by the next entry)
        __builtin_amdgcn_wave_barrier();
    }
}
extern "C" hipError_t rb_launch_overlap_split_quad(const rb_trim_params *p, int t, bool from_list, hipStream_t stream) {
    if (p->n_pairs == 0) return hipSuccess;
    const unsigned blocks = (unsigned)((p->n_pairs + 3) / 4);
    if (from_list) {
        const unsigned g = blocks < 16384u ? blocks : 16384u;
        if (t == 4) hipLaunchKernelGGL(rb_k_overlap_split_quad_list<4>, dim3(g), dim3(64), 0, stream, *p);
        else hipLaunchKernelGGL(rb_k_overlap_split_quad_list<8>, dim3(g), dim3(64), 0, stream, *p);
    } else if (t == 4) {
        hipLaunchKernelGGL(rb_k_overlap_split_quad<4>, dim3(blocks), dim3(64), 0, stream, *p);
    } else {
        hipLaunchKernelGGL(rb_k_overlap_split_quad<8>, dim3(blocks), dim3(64), 0, stream, *p);
    }
    return hipGetLastError();
}
