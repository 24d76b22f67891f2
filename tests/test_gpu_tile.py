"""The tile kernel (rustybam_amd/csrc/k_tile.hip): one wavefront streams a run of consecutive SHORT records as if they were one record.
Same rows and clipped CIGARs as the per-record kernel (RB_TILE=0) and as the per-base oracle; the cases here are the ones the tile
form adds to the reference's semantics (liftover.rs:17-132, :182-226 are per record, and so are the oracle and the per-record kernel):
record boundaries inside a lane's eight ops, tiles that are full by ops / by records / by hits, records the tile kernel must hand
back (irregular, stripped ends, integrity failures, window lists that are not sorted), and the batch shape of BASELINE config 4."""
import os

import numpy as np
import pytest

import rustybam_amd
from rustybam_amd import capi
from rbtest_util import batch_args, compare_hits, random_batch, random_windows, sums

pytestmark = pytest.mark.gpu
FUSED = rustybam_amd.LIFT_FUSED_SCAN


class tile_env:
    """RB_TILE / RB_SHORT_MAX for the plans made inside the block (rb_plan_create reads them per call)"""

    def __init__(self, on=True, short_max=None):
        self.kv = {"RB_TILE": None if on else "0", "RB_SHORT_MAX": None if short_max is None else str(short_max)}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def synth_batch(engine, seed, n_rec, lo, hi, span=2_600_000, n_contig=1):
    """records of the library's generator (what bench.py runs), headers from the record scan"""
    n = capi.synth_n_ops(seed, 0, n_rec, lo, hi)
    off = np.zeros(n_rec + 1, np.uint64)
    off[1:] = np.cumsum(n)
    ops = capi.synth_fill_ops_host(seed, 0, off)
    z = np.zeros(n_rec, np.uint64)
    strand = np.where(np.arange(n_rec) % 3 == 0, ord("-"), ord("+")).astype(np.uint8)
    red, _ = engine.scan_records(ops, off, z, z, z, z, strand)
    rng = np.random.default_rng(seed & 0xFFFF)
    t_st = rng.integers(0, span, n_rec).astype(np.uint64)
    q_st = rng.integers(0, 100_000, n_rec).astype(np.uint64)
    return dict(ops=ops, op_off=off, t_st=t_st, t_en=t_st + red["t_bases"], q_st=q_st, q_en=q_st + red["q_bases"], strand=strand,
                contig=(np.arange(n_rec) % n_contig).astype(np.uint32))


def sliding(span=2_800_000, step=82_796, width=100_000, n_contig=1):
    st = np.arange(0, span, step, dtype=np.uint64)
    st = np.tile(st, n_contig)
    wc = np.repeat(np.arange(n_contig, dtype=np.uint32), len(st) // n_contig)
    return wc, st, st + np.uint64(width)


def lift_both(engine, oracle, b, w, policy, what, expect_back=None):
    """liftover through the tile kernel == through the per-record kernel == the oracle; -> counters of the tile run"""
    ob = oracle.Batch(*batch_args(b), b["contig"])
    orows, oops = oracle.liftover(ob, *w, policy=policy & 1)
    with tile_env(True):
        rows, ops, norm, cnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=policy)
    with tile_env(False):
        rows0, ops0, norm0, cnt0 = engine.liftover(*batch_args(b), b["contig"], *w, policy=policy)
    assert int(cnt0["phase"][3]) == 0
    assert np.array_equal(norm["status"], norm0["status"]), what
    for k in ("t_st", "t_en", "q_st", "q_en", "first_op", "n_ops", "nmatch", "aln_len", "flags"):
        okn = norm0["status"] == 0
        assert np.array_equal(norm[k][okn], norm0[k][okn]), f"{what}: norm.{k}"
    keep = (norm["status"] == 0)[rows["rec"]] if len(rows) else np.zeros(0, bool)
    compare_hits(rows[keep], ops, orows, oops, what + " (tiles)")
    compare_hits(rows0[keep], ops0, orows, oops, what + " (per record)")
    assert np.array_equal(rows["flags"] & 2, rows0["flags"] & 2), f"{what}: the same hits go to the generic kernel"
    if expect_back is not None:
        assert int(cnt["phase"][3]) > 0, f"{what}: no tiles were made"
        back = int(cnt["phase"][4])
        assert (back == 0) if expect_back == 0 else (back >= expect_back), f"{what}: {back} records handed back, expected {expect_back}"
    return cnt


def break_both(engine, oracle, b, max_size, policy, what, expect_back=None):
    ob = oracle.Batch(*batch_args(b), b["contig"])
    orows, oops = oracle.break_paf(ob, max_size)
    with tile_env(True):
        rows, ops, norm, cnt = engine.break_paf(*batch_args(b), max_size, policy=policy)
    with tile_env(False):
        rows0, ops0, norm0, cnt0 = engine.break_paf(*batch_args(b), max_size, policy=policy)
    compare_hits(rows, ops, orows, oops, what + " (tiles)")
    compare_hits(rows0, ops0, orows, oops, what + " (per record)")
    if expect_back is not None:
        assert int(cnt["phase"][3]) > 0, f"{what}: no tiles were made"
        back = int(cnt["phase"][4])
        assert (back == 0) if expect_back == 0 else (back >= expect_back), f"{what}: {back} records handed back, expected {expect_back}"
    return cnt


@pytest.mark.parametrize("policy", [FUSED, 0, FUSED | rustybam_amd.BSEARCH_LEGACY])
def test_config4_shape_sample_liftover(engine, oracle, policy):
    """5,000 records of BASELINE config 4's shape (300-700 ops, seed 0x5EED0004) under 100 kb sliding windows: tiles of six to thirteen
    records, every one taken by the tile kernel"""
    b = synth_batch(engine, 0x5EED0004, 5000, 300, 700)
    lift_both(engine, oracle, b, sliding(), policy, "config 4 shape", expect_back=0)


@pytest.mark.parametrize("max_size", [100, 0, 10])
def test_config4_shape_sample_break(engine, oracle, max_size):
    b = synth_batch(engine, 0x5EED0004, 5000, 300, 700)
    pol = FUSED | rustybam_amd.BREAK_ONE_WALK
    # --max-size 0 cuts at every indel: tiles of 4000 ops hold more than 64 pieces and go back to the per-record kernel, which is the point
    break_both(engine, oracle, b, max_size, pol, f"config 4 shape, break {max_size}", expect_back=0 if max_size == 100 else None)
    break_both(engine, oracle, b, max_size, FUSED, f"config 4 shape, break {max_size}, two walks")


@pytest.mark.parametrize("lo,hi", [(8, 9), (8, 40), (31, 33), (63, 130), (500, 520), (1000, 2048)])
def test_record_lengths_around_the_tile_geometry(engine, oracle, lo, hi):
    """records of 8 ops (the least a tile takes) up to the longest (RB_SHORT_MAX): a lane's eight ops hold the end of one record and
    the start of the next at every offset, 32 records fill a tile before 4064 ops do, two records fill it"""
    b = synth_batch(engine, 0x7117 + lo, 900, lo, hi, span=400_000)
    w = sliding(span=600_000, step=8_279, width=10_000)
    lift_both(engine, oracle, b, w, FUSED, f"lengths {lo}-{hi}", expect_back=None)
    break_both(engine, oracle, b, 5, FUSED | rustybam_amd.BREAK_ONE_WALK, f"lengths {lo}-{hi}")


def test_tiny_records_pass_through(engine, oracle):
    """records below 8 ops never enter a tile (a lane's eight ops would hold three records): pass-through tiles hand them to the
    per-record kernel, between tiles of longer records"""
    rng = np.random.default_rng(8)
    b = random_batch(rng, 1200, "regular", n_contig=2, max_ops=14, long_frac=0.05)
    w = random_windows(rng, b, 80, monotone=True)
    cnt = lift_both(engine, oracle, b, w, FUSED, "tiny records")
    n = np.diff(b["op_off"].astype(np.int64))
    assert int(cnt["phase"][4]) >= int((n < 8).sum())


def test_dense_windows_hand_tiles_back(engine, oracle):
    """more than 64 hits in a tile: the tile goes to the per-record kernel; 1 kb windows over records of 20 kb"""
    b = synth_batch(engine, 0xD3115E, 300, 100, 140, span=200_000)
    w = sliding(span=260_000, step=1_000, width=1_500)
    lift_both(engine, oracle, b, w, FUSED, "dense windows", expect_back=100)


def test_tiles_are_cut_where_their_hits_fill_the_lanes(engine, oracle):
    """a tile of 31 records with three hits each holds more hits than the tile kernel has lanes (64): it keeps the records in front whose
    hits fit and hands the others to the per-record kernel -- a third of the batch, not all of it"""
    b = synth_batch(engine, 0xC07, 620, 100, 140, span=400_000)          # records of about 23 kb ...
    w = sliding(span=500_000, step=9_000, width=10_000)                   # ... under windows every 9 kb: about 3.5 hits a record
    cnt = lift_both(engine, oracle, b, w, FUSED, "tiles cut by hits")
    back = int(cnt["phase"][4])
    assert int(cnt["phase"][3]) > 0 and 620 // 8 < back < 620 * 2 // 3, back


@pytest.mark.parametrize("mode", ["indel_ends", "wild", "mixed", "spliced"])
def test_records_the_tile_kernel_hands_back(engine, oracle, mode):
    """stripped end indels, irregular CIGARs, integrity failures inside tiles: the whole tile is handed back, results as before"""
    rng = np.random.default_rng(77)
    b = random_batch(rng, 700, mode, n_contig=2, max_ops=60, long_frac=0.2, break_frac=0.05 if mode == "mixed" else 0.0)
    w = random_windows(rng, b, 120, monotone=True)
    for pol in (FUSED, 0):
        with tile_env(True):
            rows, ops, norm, cnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=pol)
        with tile_env(False):
            rows0, ops0, norm0, cnt0 = engine.liftover(*batch_args(b), b["contig"], *w, policy=pol)
        assert len(rows) == len(rows0) and np.array_equal(norm["status"], norm0["status"])
        for k in ("rec", "win", "status", "out_n", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len"):
            assert np.array_equal(rows[k], rows0[k]), f"{mode}: {k}"
        for g, o in zip(rows, rows0):
            if g["status"] == 0:
                assert np.array_equal(ops[int(g["out_off"]):int(g["out_off"]) + int(g["out_n"])], ops0[int(o["out_off"]):int(o["out_off"]) + int(o["out_n"])])
    if mode == "spliced":  # (N ops are regular: these tiles stay with the tile kernel)
        assert int(cnt["phase"][3]) > 0


def test_one_bad_record_in_a_tile(engine, oracle):
    """a record whose CIGAR does not sum to its header in the middle of a tile (check_integrity, paf.rs:825-857): the record gets its
    status, its neighbours their rows.  (Round 6: a record the tile kernel cannot take at SET-UP -- irregular, stripped -- cuts the tile to
    its longest run of records it can take; this one is found by the fused scan behind the stream, and the tile goes back as a whole.)"""
    b = synth_batch(engine, 0xBAD, 64, 100, 160, span=100_000)
    b["t_en"][17] += np.uint64(3)
    b["q_en"][40] += np.uint64(1)
    w = sliding(span=200_000, step=8_279, width=10_000)
    with tile_env(True):
        rows, ops, norm, cnt = engine.liftover(*batch_args(b), b["contig"], *w, policy=FUSED)
    with tile_env(False):
        rows0, ops0, norm0, cnt0 = engine.liftover(*batch_args(b), b["contig"], *w, policy=FUSED)
    assert norm["status"][17] != 0 and norm["status"][40] != 0 and np.array_equal(norm["status"], norm0["status"])
    assert int((norm["status"] != 0).sum()) == 2
    assert int(cnt["phase"][4]) >= 2
    assert len(rows) == len(rows0)
    ok = (norm["status"] == 0)[rows["rec"]]
    for k in ("rec", "win", "status", "t_st", "t_en", "q_st", "q_en", "nmatch", "aln_len", "out_n"):
        assert np.array_equal(rows[k][ok], rows0[k][ok]), k
    good = np.nonzero(norm["status"] == 0)[0]
    sub = {k: (v[good] if k not in ("ops", "op_off") else v) for k, v in b.items()}
    n = np.diff(b["op_off"].astype(np.int64))
    sub["ops"] = np.concatenate([b["ops"][int(b["op_off"][r]):int(b["op_off"][r + 1])] for r in good])
    sub["op_off"] = np.concatenate([[0], np.cumsum(n[good])]).astype(np.uint64)
    orows, oops = oracle.liftover(oracle.Batch(*batch_args(sub), sub["contig"]), *w)
    grows = rows[ok].copy()
    grows["rec"] = np.searchsorted(good, grows["rec"])
    compare_hits(grows, ops, orows, oops, "the good records of a tile with two bad ones")


def test_windows_not_sorted_and_many_contigs(engine, oracle):
    """neighbouring records of a tile on different contigs, one contig's window list not sorted (those records are not the tile kernel's)"""
    b = synth_batch(engine, 0xC0117, 600, 40, 90, span=300_000, n_contig=3)
    rng = np.random.default_rng(5)
    wc, ws, we = sliding(span=400_000, step=8_279, width=10_000, n_contig=3)
    sel = np.nonzero(wc == 1)[0]
    perm = rng.permutation(sel)
    ws[sel], we[sel] = ws[perm], we[perm]
    lift_both(engine, oracle, b, (wc, ws, we), FUSED, "contigs", expect_back=100)


def test_no_slots_and_one_slot(engine, oracle):
    """RB_DEBUG_SLOTS=0 / 1: clips that find no place in a slot are copied by rb_k_copy_clips -- from tiles too"""
    b = synth_batch(engine, 0x5107, 400, 60, 200, span=300_000)
    w = sliding(span=400_000, step=41_398, width=50_000)
    for slots in ("0", "1"):
        os.environ["RB_DEBUG_SLOTS"] = slots
        try:
            lift_both(engine, oracle, b, w, FUSED, f"slots {slots}", expect_back=0)  # (windows wide apart: a tile of 31 records stays below 64 hits)
        finally:
            del os.environ["RB_DEBUG_SLOTS"]


def test_windows_inside_one_op_and_on_record_ends(engine, oracle):
    """boundaries on the first / last base of a record, windows inside one op, windows that end where the next record of the tile begins"""
    t0 = 1000
    cig = []
    for k in range(40):
        c = [(50 + k, 7), (3, 1), (20, 8), (5, 2), (100, 7), (1, 1), (30 + k, 7), (2, 2), (9, 8), (77, 7)]
        cig.append(np.array([(ln << 4) | op for ln, op in c], np.uint32))
    off = np.zeros(41, np.uint64)
    off[1:] = np.cumsum([len(c) for c in cig])
    ops = np.concatenate(cig)
    R = np.array([sums(c)[0] for c in cig], np.uint64)
    Q = np.array([sums(c)[1] for c in cig], np.uint64)
    t_st = np.uint64(t0) + np.arange(40, dtype=np.uint64) * np.uint64(1000)
    b = dict(ops=ops, op_off=off, t_st=t_st, t_en=t_st + R, q_st=np.full(40, 7, np.uint64), q_en=np.uint64(7) + Q,
             strand=np.where(np.arange(40) % 2 == 0, ord("+"), ord("-")).astype(np.uint8), contig=np.zeros(40, np.uint32))
    offs = [0, 1, 49, 50, 53, 60, 100, 200, 290, 296, 300, 400, 500]  # (record spans are 297 .. 375 reference bases)
    for width in (1, 2, 30, 400):
        # two windows per record, the offsets taken in turn: 64 hits in the tile of 32 records -- as many as a tile holds
        st = np.array(sorted(int(t_st[k]) + offs[(2 * k + j) % 13] for k in range(40) for j in (0, 1)), np.uint64)
        w = (np.zeros(len(st), np.uint32), st, st + np.uint64(width))
        lift_both(engine, oracle, b, w, FUSED, f"record ends, width {width}", expect_back=0)
        lift_both(engine, oracle, b, w, rustybam_amd.BSEARCH_LEGACY, f"record ends, width {width}, legacy", expect_back=0)
    for ms in (0, 1, 2, 4):
        # (--max-size 4: two pieces per record, 64 in the tile of 32; below that the tile holds more pieces than lanes and is handed back)
        break_both(engine, oracle, b, ms, FUSED | rustybam_amd.BREAK_ONE_WALK, f"ten-op records, break {ms}", expect_back=0 if ms == 4 else 32)


def test_a_tile_is_cut_around_a_record_it_cannot_take(engine, oracle):
    """round 6 (the advisor's second finding of round 5): one irregular record used to send its whole tile -- up to 32 records -- to the
    per-record kernel.  The tile is cut to its longest run of records it can take; the irregular record and the shorter side go back.
    Results as without tiles and as the oracle's; fewer records handed back than the tiles hold."""
    n_rec = 96
    b = synth_batch(engine, 0xC07, n_rec, 100, 160, span=150_000)
    off = b["op_off"].astype(np.int64)
    bad = [5, 41, 42, 90]
    new_ops, new_off = [], [0]
    for r in range(n_rec):  # op 3 of a bad record split in two ops of its type: irregular (paf.rs:602-620 merges them on the way out), same bases
        seg = b["ops"][int(off[r]):int(off[r + 1])].tolist()
        if r in bad:
            k = next(i for i in range(3, len(seg)) if (seg[i] >> 4) >= 2)
            w = seg[k]
            seg[k:k + 1] = [((w >> 4) - 1) << 4 | (w & 15), 1 << 4 | (w & 15)]
        new_ops += seg
        new_off.append(len(new_ops))
    b2 = dict(b)
    b2["ops"] = np.array(new_ops, np.uint32)
    b2["op_off"] = np.array(new_off, np.uint64)
    w = sliding(span=300_000, step=49_999, width=12_000)  # (sparse enough that no tile is cut by its hits)
    # (with the record scan done beforehand the set-up knows which records are irregular; with the FUSED scan it only knows what the peek at a
    #  record's ends shows -- an irregularity in the middle is found behind the stream, and that tile still goes back as a whole)
    c = lift_both(engine, oracle, b2, w, 0, "tiles cut around irregular records")
    tiles, back = int(c["phase"][3]), int(c["phase"][4])
    assert tiles > 0 and len(bad) <= back < n_rec // 2, (tiles, back)
    lift_both(engine, oracle, b2, w, FUSED, "irregular records in the middle of tiles, fused scan", expect_back=len(bad))
    c = break_both(engine, oracle, b2, 50, 0, "break-paf, tiles cut around irregular records")
    assert int(c["phase"][3]) > 0 and len(bad) <= int(c["phase"][4]) < n_rec // 2, (int(c["phase"][3]), int(c["phase"][4]))
