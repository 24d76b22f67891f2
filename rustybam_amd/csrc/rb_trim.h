// rb_trim.h -- what the trim-paf pair kernels share (k_trim.hip: the wave-per-pair and the serial kernel; k_trim4.hip: four pairs per
// wavefront) and what capi.hip fills in.
#pragma once
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
    uint32_t *scratch;       // device memory for the third attempt of the wave kernel (regions too large for LDS), or NULL
    uint32_t scratch_blocks; // slabs in it
    // pairs the first wave kernel declines, so that the attempts behind it do not have to look at every row: pend[0] = how many,
    // pend_list[0 .. n_pairs) their indices (NULL: every row is looked at)
    unsigned long long *pend;
    uint32_t *pend_list;
    // RB_TRIM_IN_PLACE (out_ops is the batch's own ops array): a regular record the wave kernel clips is not copied -- a clip by query
    // keeps a run of the record's ops and changes the lengths of the run's first and last op only, so those two words are rewritten
    // where they are and the row points at the run (out_off = its place in the array).  Pairs the serial kernel does still write
    // their clips at pair_out_off.
    int in_place;
};

#define RB_ST_PENDING_INTERNAL 0x7FFF0001u // a pair row's status while a kernel further down the chain still has to do the pair

__device__ __forceinline__ int32_t rb_tw_score(uint32_t opc, int32_t ms, int32_t ds, int32_t is) { // trim_overlap.rs:14-18
    return opc == RB_OP_EQ ? ms : ((opc == RB_OP_I || opc == RB_OP_D) ? -is : -ds);
}
