// rb_serial.h -- per-thread (serial) evaluation of the reference's per-base semantics in op space.
//
// The reference materialises tpos_aln / qpos_aln / long_cigar (paf.rs:501-538) and indexes them by
// "unit" (one entry per aligned base or indel base).  These helpers answer the same questions by
// walking the packed ops; they are fully general (all nine op codes, zero-length ops, any adjacency)
// and are used where generality matters more than speed: the trim-paf pair kernel and the clip-by-
// query step.  One thread per work item; O(n_ops) per question.
#pragma once
#include "rb_device.h"

struct rb_sview { // one record after remove_trailing_indels
    const uint32_t *ops;
    uint32_t n;
    uint64_t t_st, t_en, q_st, q_en;
    bool minus;
};

__device__ __forceinline__ bool rb_s_ref(uint32_t opc) { return opc <= 8u && rb_in(RB_REF_MASK, opc); }
__device__ __forceinline__ bool rb_s_qry(uint32_t opc) { return opc <= 8u && rb_in(RB_QRY_MASK, opc); }
__device__ __forceinline__ bool rb_s_match(uint32_t opc) { return opc <= 8u && rb_in(RB_MATCH_MASK, opc); }

__device__ inline uint64_t rb_s_units(const rb_sview &v) {
    uint64_t N = 0;
    for (uint32_t i = 0; i < v.n; i++) N += rb_wlen(v.ops, i);
    return N;
}

// legacy Rust binary_search (1.52..1.81) on a virtual array whose equal range is [klo, khi]
__device__ inline uint64_t rb_s_legacy_probe(uint64_t N, uint64_t klo, uint64_t khi) {
    uint64_t size = N, left = 0, right = N;
    while (left < right) {
        const uint64_t mid = left + size / 2;
        if (mid < klo)
            left = mid + 1;
        else if (mid > khi)
            right = mid;
        else
            return mid;
        size = right - left;
    }
    return klo;
}

// units k with qpos_aln[k] == p form a contiguous range; returns false when there is none
// (paf.rs:505-534: '+' counts up from q_st - 1, '-' counts down from q_en; non-query units repeat the value)
__device__ inline bool rb_s_qrange(const rb_sview &v, uint64_t p, uint64_t *klo, uint64_t *khi) {
    bool found = false;
    uint64_t U = 0;
    int64_t qpos = v.minus ? (int64_t)v.q_en : (int64_t)v.q_st - 1;
    const int64_t pp = (int64_t)p;
    for (uint32_t i = 0; i < v.n; i++) {
        const uint32_t opc = rb_wopc(v.ops, i), len = rb_wlen(v.ops, i);
        if (len == 0) continue;
        if (rb_s_qry(opc)) {
            if (!v.minus) {
                if (pp > qpos && pp <= qpos + (int64_t)len) {
                    const uint64_t u = U + (uint64_t)(pp - qpos - 1);
                    if (!found) *klo = u;
                    *khi = u;
                    found = true;
                }
                qpos += len;
            } else {
                if (pp < qpos && pp >= qpos - (int64_t)len) {
                    const uint64_t u = U + (uint64_t)(qpos - 1 - pp);
                    if (!found) *klo = u;
                    *khi = u;
                    found = true;
                }
                qpos -= len;
            }
        } else if (pp == qpos && qpos >= 0) {
            if (!found) *klo = U;
            *khi = U + len - 1;
            found = true;
        }
        U += len;
    }
    return found;
}

// what unit k holds: op code, tpos, qpos (as the reference's arrays would)
__device__ inline void rb_s_unit(const rb_sview &v, uint64_t k, uint32_t *opc_out, uint64_t *tpos_out, uint64_t *qpos_out) {
    uint64_t U = 0;
    int64_t tpos = (int64_t)v.t_st - 1;
    int64_t qpos = v.minus ? (int64_t)v.q_en : (int64_t)v.q_st - 1;
    for (uint32_t i = 0; i < v.n; i++) {
        const uint32_t opc = rb_wopc(v.ops, i), len = rb_wlen(v.ops, i);
        if (len == 0) continue;
        const bool r = rb_s_ref(opc), q = rb_s_qry(opc);
        if (k < U + len) {
            const int64_t off = (int64_t)(k - U);
            *opc_out = opc;
            *tpos_out = (uint64_t)(r ? tpos + 1 + off : tpos);
            *qpos_out = (uint64_t)(q ? (v.minus ? qpos - 1 - off : qpos + 1 + off) : qpos);
            return;
        }
        if (r) tpos += len;
        if (q) qpos += v.minus ? -(int64_t)len : (int64_t)len;
        U += len;
    }
    *opc_out = RB_NULL_OP;
    *tpos_out = *qpos_out = 0;
}

// qpos_aln of a '+' record whose query start is 0 and whose first ops consume no query begins with units at
// q_pos = -1, i.e. u64::MAX (paf.rs:506, :532): the array is then NOT sorted and what slice::binary_search_by returns is
// whatever its probe sequence leads to.  rb_s_bsearch_q replays that sequence on the virtual array (paf.rs:564-573; the
// comparator is reversed on '-'), for both generations of the Rust standard library.  false = Err.
__device__ inline bool rb_s_wrapped_q(const rb_sview &v) {
    if (v.minus || v.q_st != 0) return false;
    for (uint32_t i = 0; i < v.n; i++) {
        if (rb_wlen(v.ops, i) == 0) continue;
        return !rb_s_qry(rb_wopc(v.ops, i));
    }
    return false;
}
__device__ inline bool rb_s_bsearch_q(const rb_sview &v, uint64_t N, uint64_t key, int policy, uint64_t *idx) {
    auto cmp = [&](uint64_t mid) -> int {
        uint32_t oc;
        uint64_t tp, qp;
        rb_s_unit(v, mid, &oc, &tp, &qp);
        const int c = qp < key ? -1 : (qp > key ? 1 : 0);
        return v.minus ? -c : c;
    };
    if (policy != RB_BSEARCH_LEGACY) { // rustc >= 1.82
        uint64_t size = N;
        if (size == 0) return false;
        uint64_t base = 0;
        while (size > 1) {
            const uint64_t half = size / 2, mid = base + half;
            base = cmp(mid) > 0 ? base : mid;
            size -= half;
        }
        *idx = base;
        return cmp(base) == 0;
    }
    uint64_t size = N, left = 0, right = N; // 1.52 .. 1.81
    while (left < right) {
        const uint64_t mid = left + size / 2;
        const int c = cmp(mid);
        if (c < 0) left = mid + 1;
        else if (c > 0) right = mid;
        else {
            *idx = mid;
            return true;
        }
        size = right - left;
    }
    return false;
}

// first match-type unit >= k (N if none): the walk of paf.rs:551-553 / :581-583
__device__ inline uint64_t rb_s_match_ge(const rb_sview &v, uint64_t k, uint64_t N) {
    uint64_t U = 0;
    for (uint32_t i = 0; i < v.n; i++) {
        const uint32_t opc = rb_wopc(v.ops, i), len = rb_wlen(v.ops, i);
        if (len == 0) continue;
        if (rb_s_match(opc) && U + len > k) return k > U ? k : U;
        U += len;
    }
    return N;
}
// last match-type unit <= k (0 if none): the walk of paf.rs:555-557 / :585-587
__device__ inline uint64_t rb_s_match_le(const rb_sview &v, uint64_t k) {
    uint64_t U = 0, best = 0;
    for (uint32_t i = 0; i < v.n; i++) {
        const uint32_t opc = rb_wopc(v.ops, i), len = rb_wlen(v.ops, i);
        if (len == 0) continue;
        if (U > k) break;
        if (rb_s_match(opc)) best = (U + len - 1) < k ? (U + len - 1) : k;
        U += len;
    }
    return best;
}

// run-length-merged ops of units [a, b] (paf.rs:593-620) written to out; returns the op count and the
// reference / query / match / unit sums of what was written
__device__ inline uint32_t rb_s_emit_units(const rb_sview &v, uint64_t a, uint64_t b, uint32_t *out, uint64_t sums[4]) {
    uint64_t U = 0;
    uint32_t prev = RB_NULL_OP, run = 0, cnt = 0;
    sums[0] = sums[1] = sums[2] = sums[3] = 0;
    for (uint32_t i = 0; i < v.n; i++) {
        const uint32_t opc = rb_wopc(v.ops, i), len = rb_wlen(v.ops, i);
        if (len == 0) continue;
        const uint64_t u0 = U, u1 = U + len - 1;
        U += len;
        if (u1 < a) continue;
        if (u0 > b) break;
        const uint64_t c0 = u0 > a ? u0 : a, c1 = u1 < b ? u1 : b;
        const uint32_t piece = (uint32_t)(c1 - c0 + 1);
        if (rb_s_ref(opc)) sums[0] += piece;
        if (rb_s_qry(opc)) sums[1] += piece;
        if (rb_s_match(opc)) sums[2] += piece;
        sums[3] += piece;
        if (opc != prev) {
            if (prev != RB_NULL_OP) cnt += rb_emit_run(out + cnt, run, prev);
            prev = opc;
            run = piece;
        } else {
            run += piece;
        }
    }
    if (prev != RB_NULL_OP) cnt += rb_emit_run(out + cnt, run, prev);
    return cnt;
}

// remove_trailing_indels (paf.rs:656-783) applied to an op list in memory.  Shifts the coordinates,
// returns the kept range [*first, *first + *count) and the status of the check_integrity().unwrap().
__device__ inline uint32_t rb_s_strip_indels(const uint32_t *ops, uint32_t n, bool minus, uint64_t *t_st, uint64_t *t_en, uint64_t *q_st,
                                             uint64_t *q_en, uint32_t *first, uint32_t *count, uint32_t *nmatch, uint32_t *aln_len) {
    if (n == 0) return RB_ST_PANIC_EMPTY_CIGAR;
    uint64_t lead = 0, rm_st_t = 0, rm_st_q = 0;
    uint32_t prev = RB_NULL_OP;
    while (lead < n) {
        uint32_t opc, len;
        const uint32_t words = rb_op_fwd(ops, n, lead, &opc, &len);
        if (!rb_in(RB_INDEL_MASK, opc)) break;
        if (opc == RB_OP_D) {
            rm_st_t += len;
            rm_st_q += 1;
        } else {
            rm_st_q += len;
        }
        if (prev != RB_NULL_OP && prev != opc) {
            rm_st_t += 1;
            rm_st_q -= 1;
        }
        prev = opc;
        lead += words;
    }
    uint64_t trail = 0, rm_en_t = 0, rm_en_q = 0;
    while (trail < n) {
        uint32_t opc, len;
        const uint32_t words = rb_op_bwd(ops, n - 1 - trail, &opc, &len);
        if (!rb_in(RB_INDEL_MASK, opc)) break;
        if (opc == RB_OP_D) rm_en_t += len; else rm_en_q += len;
        trail += words;
    }
    if (lead + trail > n) return RB_ST_PANIC_ALL_INDEL;
    *t_st += rm_st_t;
    *t_en -= rm_en_t;
    if (minus) {
        const uint64_t t = rm_st_q;
        rm_st_q = rm_en_q;
        rm_en_q = t;
    }
    *q_st += rm_st_q;
    *q_en -= rm_en_q;
    uint64_t R = 0, Q = 0, M = 0, U = 0;
    for (uint64_t i = lead; i < n - trail; i++) {
        const uint32_t opc = rb_wopc(ops, (uint32_t)i), len = rb_wlen(ops, (uint32_t)i);
        if (rb_s_ref(opc)) R += len;
        if (rb_s_qry(opc)) Q += len;
        if (rb_s_match(opc)) M += len;
        U += len;
    }
    *first = (uint32_t)lead;
    *count = (uint32_t)(n - lead - trail);
    *nmatch = (uint32_t)M;
    *aln_len = (uint32_t)U;
    if (U > 0xFFFFFFFFull) return RB_ST_PANIC_OVERFLOW;
    if (*t_en < *t_st || *t_en - *t_st != R) return RB_ST_PANIC_INTEGRITY_T;
    if (*q_en < *q_st || *q_en - *q_st != Q) return RB_ST_PANIC_INTEGRITY_Q;
    return RB_ST_OK;
}
