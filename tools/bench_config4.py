"""the two CIGAR-walk stages of the README pipeline (`trim-paf | break-paf --max-size 100`) on synthetic records
of 300-700 ops, 4 records per query whose consecutive query spans overlap by U[100, 10000] bases (1e7 records = the full size).

The batch is generated in HBM and stays there: the passes of Paf::overlapping_paf_recs run on the device (rb_dev_trim_select: pair
scan + selection; rb_dev_overlap_split + rb_dev_apply_pairs: split + clip in place), rb_dev_gather_records makes the batch dense,
rb_dev_break cuts it.  Reported: wall time of the trim passes and of break-paf (inputs resident, one 64-byte read per pass), the
pair kernels' time under HIP events and their roofline (algorithmic bytes of a pair pass: both records read once, 4 B per op, +
128 B row written + 4 B per emitted op).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split
(tools/prof_c4.sh).  No oracle here: parity of both stages is the business of tests/ (test_gpu_fullsize.py runs this workload).

  python tools/bench_config4.py [--records 10000000] [--host-buffers]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
SEED = 0x5EED0004


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=10_000_000)
    ap.add_argument("--host-buffers", action="store_true", help="also time rb_host_overlap_split on host arrays (PCIe-inclusive, 2e6 records at most)")
    a = ap.parse_args()
    import torch
    import rustybam_amd
    from rustybam_amd import workload as wl, capi, trim_driver
    from devutil import DevBatch
    dev = torch.device("cuda", 0)
    torch.cuda.set_stream(torch.cuda.Stream(dev))
    eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
    n = a.records // 4 * 4
    t0 = time.time()
    nops = wl.n_ops(SEED, 0, n, 300, 700)
    off = wl.op_offsets(nops)
    total_ops = int(off[-1])
    i64 = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to(dev)  # noqa: E731
    d_off = i64(off)
    d_ops = torch.empty(total_ops + 64, dtype=torch.int32, device=dev)
    eng.dev_synth_fill_ops(SEED, 0, n, d_off.data_ptr(), d_ops.data_ptr())
    zeros = torch.zeros(n, dtype=torch.int64, device=dev)
    d_red = torch.empty(n * 72, dtype=torch.uint8, device=dev)
    v0 = eng.batch_view(n, total_ops, d_ops.data_ptr(), d_off.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(), zeros.data_ptr(),
                        torch.full((n,), ord("+"), dtype=torch.uint8, device=dev).data_ptr(), torch.zeros(n, dtype=torch.int32, device=dev).data_ptr())
    torch.cuda.synchronize()
    eng.dev_scan_records(v0, d_red.data_ptr(), 0)
    torch.cuda.synchronize()
    red = d_red.cpu().numpy().view(rustybam_amd.REDUCE_DT)
    tb, qb = red["t_bases"].astype(np.uint64), red["q_bases"].astype(np.uint64)
    del d_red, red
    rng = np.random.default_rng(SEED)
    # 4 records per query: each starts `ov` bases before the previous one ends (ov < both lengths: nothing contained)
    q_st = np.zeros(n, np.uint64)
    ov = rng.integers(100, 10001, n).astype(np.uint64)
    for j in range(1, 4):
        prev_en = q_st[j - 1::4] + qb[j - 1::4]
        q_st[j::4] = prev_en - np.minimum(ov[j::4], np.minimum(qb[j - 1::4], qb[j::4]) // np.uint64(2))
    q_en = q_st + qb
    t_st = rng.integers(0, 200_000_000, n).astype(np.uint64)
    t_en = t_st + tb
    strand = np.where(rng.integers(0, 2, n) == 0, ord("+"), ord("-")).astype(np.uint8)
    group = np.arange(n) // 4
    T = trim_driver.ResidentTrim(eng, torch, dev, d_ops, off, t_st, t_en, q_st, q_en, strand, group, room_factor=1.6)
    del d_ops
    gen = time.time() - t0
    # ---- trim-paf: the passes, device-resident ----
    # (warm-up, as bench.py has one: the first trim call of a context allocates its scratch -- 40 MB for the pairs whose region does not fit
    #  LDS -- and the first launch of every kernel is not a launch like the others; eight records of the same batch, thrown away)
    W_ = trim_driver.ResidentTrim(eng, torch, dev, T.d_ops[: int(off[8])].clone(), off[:9], t_st[:8], t_en[:8], q_st[:8], q_en[:8], strand[:8], group[:8], room_factor=1.6)
    try:
        W_.run((1, 1, 1), rustybam_amd.BSEARCH_MODERN)
    except RuntimeError:
        if not os.environ.get("RB_C4_ONE_PASS"):  # (a diagnostic variant that ends its pairs early leaves rows nobody can use)
            raise
    W_.release()
    del W_
    T._pass_buffers()  # (the query groups and the passes' device buffers: part of having the batch resident, not of the passes)
    eng.set_timing(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if os.environ.get("RB_C4_ONE_PASS"):  # (diagnostics, tools/prof_c4_decomp.sh: one pass of a library variant whose rows are wrong, then out)
        try:
            T.run((1, 1, 1), rustybam_amd.BSEARCH_MODERN, max_passes=1)
        except Exception as e:  # (more passes wanted, or rows a stopped variant left unfinished)
            print(f"one pass: {e}", file=sys.stderr)
        torch.cuda.synchronize()
        print(json.dumps({"one_pass": True}))
        return
    ev = []
    T.run((1, 1, 1), rustybam_amd.BSEARCH_MODERN, fetch=False, events=ev)  # (the batch goes on on the device: nothing of it is fetched inside the stage)
    torch.cuda.synchronize()
    t_trim = time.perf_counter() - t0
    eng.set_timing(False)
    T.fetch()  # (the rows the checks below read: outside the stage)
    k_sel, k_pair, k_apply = sum(e[1] for e in ev), sum(e[2] for e in ev), sum(e[3] for e in ev)
    # algorithmic bytes of the pair passes: every pair reads both of its records as they are at that pass (bounded by their original
    # lengths: counted from the ops in use), writes a 128-byte row and the two clipped records
    pairs = T.pairs_done
    d_new, new_off, norm = T.gather()
    nops64 = nops.astype(np.int64)
    left = np.arange(n)[np.arange(n) % 4 != 3]
    pair_in = int((nops64[left] + nops64[left + 1]).sum())        # (upper bound: the middle records have been cut once already in later passes)
    pair_out = int(2 * norm["n_ops"].astype(np.int64).sum() - norm["n_ops"][0::4].astype(np.int64).sum() - norm["n_ops"][3::4].astype(np.int64).sum())
    pair_bytes = 4 * pair_in + 128 * pairs + 4 * pair_out
    # ... and what MOVES on a resident batch (RB_TRIM_IN_PLACE): the kept run of a regular record stays where it is -- only the two end
    # words of each clip are rewritten --, so the clips are not written: ops read + rows + 2 words per clip (round-3 review: the figure
    # above prices 4 B per emitted op that this route no longer emits)
    pair_bytes_moved = 4 * pair_in + 128 * pairs + 2 * 2 * 4 * pairs
    # ---- break-paf --max-size 100 on the trimmed batch ----
    d_c = [torch.from_numpy(np.ascontiguousarray(norm[k]).view(np.int64)).to(dev) for k in ("t_st", "t_en", "q_st", "q_en")]
    B = DevBatch.from_device(torch, eng, dev, d_new, int(new_off[-1]), new_off, d_c, torch.from_numpy(strand).to(dev))
    T.release()
    torch.cuda.empty_cache()
    B.run(None, max_size=100, rows_cap=4 * n, policy=rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN | rustybam_amd.BREAK_ONE_WALK)  # sizing
    B.last = None  # (the sizing run's buffers go back to torch's allocator: the timed run reuses them instead of asking the driver for 45 GB)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rows, out, cnt = B.run(None, max_size=100, rows_cap=4 * n, policy=rustybam_amd.BSEARCH_MODERN | rustybam_amd.LIFT_FUSED_SCAN | rustybam_amd.BREAK_ONE_WALK)
    torch.cuda.synchronize()
    t_break = time.perf_counter() - t0
    res = {"workload": f"config 4: {n} records, {total_ops} ops, {n // 4} query groups of 4, seed 0x5eed0004; batch resident in HBM",
           "trim_passes": T.passes, "trim_pairs": pairs, "trim_wall_s": round(t_trim, 4), "trim_pairs_per_s_wall": pairs / t_trim,
           "trim_records_per_s_wall": n / t_trim,
           "trim_pair_pass_algorithmic_bytes": pair_bytes, "trim_pair_ops_in": pair_in, "trim_pair_ops_out": pair_out,
           "trim_roofline_on_wall": {"bound": "hbm", "achieved": round(pair_bytes / t_trim / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(pair_bytes / t_trim / 8e12, 4),
                                     "note": "whole trim-paf stage (selection + pair kernels + apply + the host's reads) over the pair passes' algorithmic bytes"},
           "trim_pair_pass_moved_bytes": pair_bytes_moved,
           "trim_roofline_moved_on_wall": {"bound": "hbm", "achieved": round(pair_bytes_moved / t_trim / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                           "frac": round(pair_bytes_moved / t_trim / 8e12, 4),
                                           "note": "the same time over the bytes the in-place route really moves (ops read + rows + two words per clip): the honest fraction"},
           "trim_kernels_ms": {"selection": round(k_sel, 3), "pair_kernels": round(k_pair, 3), "apply_and_check": round(k_apply, 3),
                               "per_pass": [[int(e[0]), round(e[1], 3), round(e[2], 3), round(e[3], 3)] for e in ev],
                               "note": "HIP events on the engine's stream around every pass: pairs, ms of rb_dev_trim_select, of rb_dev_overlap_split "
                                       "(the wave-per-pair kernel and its retries), of rb_dev_apply_pairs + rb_dev_trim_check"},
           "trim_wall_over_kernels": round(t_trim * 1e3 / max(1e-9, k_sel + k_pair + k_apply), 2),
           "trim_roofline_moved_on_pair_kernels": {"bound": "hbm", "achieved": round(pair_bytes_moved / max(1e-9, k_pair * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                                   "frac": round(pair_bytes_moved / max(1e-9, k_pair * 1e-3) / 8e12, 4),
                                                   "note": "the bytes the in-place route moves over the pair kernels' own time (HIP events)"},
           "pairs_by_wave_kernel": T.pairs_by_wave,
           "break_pieces": int(rows.shape[0]), "break_wall_s": round(t_break, 4), "break_records_per_s_wall": n / t_break,
           "break_one_walk": not bool(cnt["redo_two_walk"]), "setup_s": round(gen, 2)}
    if a.host_buffers and n <= 2_000_000:
        ops_h = capi.synth_fill_ops_host(SEED, 0, off)
        t0 = time.time()
        rows_h, _ = eng.overlap_split(ops_h, off, t_st, t_en, q_st, q_en, strand, left.astype(np.uint32), (left + 1).astype(np.uint32))
        res["host_buffer_overlap_split_s"] = round(time.time() - t0, 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
