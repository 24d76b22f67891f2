"""SURVEY 8d config 4, scaled: the two CIGAR-walk stages of the README pipeline (`trim-paf | break-paf --max-size 100`) on
synthetic records of ~500 ops, 4 records per query whose consecutive query spans overlap by U[100, 10000] bases.

Host-buffer entry points (rb_host_overlap_split, rb_host_break): the wall times include PCIe; run under
`rocprofv3 --kernel-trace --stats` for the kernel times (profiles/r01_c4_summary.md).  No oracle here: parity of both stages is
the business of tests/ and tests/soak/.

  python tools/bench_config4.py [--records 1000000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
SEED = 0x5EED0004


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=1_000_000)
    a = ap.parse_args()
    import rustybam_amd
    from rustybam_amd import workload as wl, capi
    eng = rustybam_amd.Engine(0)
    n = a.records // 4 * 4
    t0 = time.time()
    nops = wl.n_ops(SEED, 0, n, 300, 700)
    off = wl.op_offsets(nops)
    ops = capi.synth_fill_ops_host(SEED, 0, off)
    strand = np.full(n, ord("+"), np.uint8)
    z = np.zeros(n, np.uint64)
    red, _ = eng.scan_records(ops, off, z, z, z, z, strand)
    tb, qb = red["t_bases"].astype(np.uint64), red["q_bases"].astype(np.uint64)
    rng = np.random.default_rng(SEED)
    # 4 records per query: each starts `ov` bases before the previous one ends (ov < both lengths: nothing contained)
    q_st = np.zeros(n, np.uint64)
    ov = rng.integers(100, 10001, n).astype(np.uint64)
    for j in range(1, 4):
        prev_en = q_st[j - 1::4][:n // 4] + qb[j - 1::4]
        o = np.minimum(ov[j::4], np.minimum(qb[j - 1::4], qb[j::4]) // np.uint64(2))
        q_st[j::4] = prev_en - o
    q_en = q_st + qb
    t_st = rng.integers(0, 200_000_000, n).astype(np.uint64)
    t_en = t_st + tb
    left = np.arange(n, dtype=np.uint32)[np.arange(n) % 4 != 3]
    right = left + 1
    gen = time.time() - t0
    t0 = time.time()
    rows, out = eng.overlap_split(ops, off, t_st, t_en, q_st, q_en, strand, left, right)
    t_trim = time.time() - t0
    ok = int((rows["status"] == 0).sum())
    t0 = time.time()
    hits, bout, norm, cnt = eng.break_paf(ops, off, t_st, t_en, q_st, q_en, strand, 100)
    t_break = time.time() - t0
    # algorithmic bytes of the pair pass: both records of a pair read once (4 B per op) + 128 B row written + 4 B per emitted op
    nops64 = nops.astype(np.int64)
    pair_in = int((nops64[left] + nops64[right]).sum())
    pair_out = int(rows["out_n"].astype(np.int64).sum())
    pair_bytes = 4 * pair_in + 128 * len(left) + 4 * pair_out
    print(json.dumps({"workload": f"config4 scaled: {n} records, {int(off[-1])} ops, {len(left)} overlapping pairs, seed 0x5eed0004",
                      "trim_pair_pass_algorithmic_bytes": pair_bytes, "trim_pair_ops_in": pair_in, "trim_pair_ops_out": pair_out,
                      "trim_pairs": len(left), "trim_pairs_ok": ok, "trim_wall_s": round(t_trim, 3), "trim_pairs_per_s_wall": len(left) / t_trim,
                      "break_pieces": int(len(hits)), "break_wall_s": round(t_break, 3), "break_records_per_s_wall": n / t_break,
                      "setup_s": round(gen, 2)}))


if __name__ == "__main__":
    main()
