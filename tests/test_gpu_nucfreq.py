"""GPU parity of rb_dev_nucfreq (k_nucfreq.hip) against the oracle's pileup restatement: bit-exact counts and coverage."""
import os

import numpy as np
import pytest

import rustybam_amd
from nf_util import QRY_OPS, Reads, read_bam, random_reads, check_regions

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_ka13_fixture(engine, oracle):
    names, lens, rd = read_bam(f"{GOLD}/test_nucfreq.bam")
    counts, status, ctr = check_regions(engine, oracle, rd, [(0, 1, 102), (0, 0, lens[0] if lens[0] < 50000 else 50000)])
    mx = (counts[:101] & 0x7FFFFFFF).max(axis=1)
    assert set(mx.tolist()) <= {0, 2}          # nucfreq.rs:41-60
    assert ctr["max_depth"] == 2 and ctr["unsorted"] == 0 and ctr["n_bad"] == 0


@pytest.mark.parametrize("bam", ["asm_small.bam", "stats.bam"])
def test_bam_fixtures(engine, oracle, bam):
    """whole-contig alignments (40k-op cigars spanning megabases) and a read set: windows around the tile edges"""
    names, lens, rd = read_bam(f"{GOLD}/{bam}")
    ok = np.flatnonzero((rd.tid >= 0) & ((rd.flag & 0x704) == 0))
    regions = []
    rng = np.random.default_rng(5)
    for i in rng.choice(ok, size=min(6, len(ok)), replace=False).tolist():
        p = int(rd.pos[i])
        regions.append((int(rd.tid[i]), max(p - 100, 0), p + 9000))
    regions.append((int(rd.tid[ok[0]]), 0, 5000))
    check_regions(engine, oracle, rd, regions)


@pytest.mark.parametrize("seed", range(8))
def test_random_reads_all_op_types(engine, oracle, seed):
    rng = np.random.default_rng(77 + seed)
    rd = random_reads(rng, 300, n_contig=3, span=30000, long_frac=0.1)
    regions = []
    for _ in range(6):
        t = int(rng.integers(0, 3))
        st = int(rng.integers(0, 35000))
        regions.append((t, st, st + int(rng.integers(1, 20000))))
    regions += [(0, 4095, 4097), (1, 0, 4096), (2, 8192, 8193), (0, 60000, 60010)]   # tile edges, one position, beyond every read
    counts, status, ctr = check_regions(engine, oracle, rd, regions)
    filt = (rd.flag & 0x704) != 0
    assert np.array_equal(status == rustybam_amd.RD_FILTERED, filt)
    assert ctr["unsorted"] == 0


def test_deep_pile_and_overlapping_regions(engine, oracle):
    rng = np.random.default_rng(9)
    rd = random_reads(rng, 3000, n_contig=1, span=2000, long_frac=0.0, odd_flags=False, max_ops=8)
    counts, status, ctr = check_regions(engine, oracle, rd, [(0, 0, 6000), (0, 1000, 3000), (0, 1000, 3000)])
    assert ctr["max_depth"] > 500


def test_empty_inputs(engine):
    rd = Reads([], [], [], [], [])
    counts, status, ctr = engine.nucfreq(*rd.args(), [0], [10], [500])
    assert counts.shape == (490, 4) and not counts.any() and ctr["n_covered"] == 0
    rd = Reads([0], [5], [0], [[(10 << 4)]], [[1] * 10])
    counts, status, ctr = engine.nucfreq(*rd.args(), [], [], [])
    assert counts.shape[0] == 0 and status.tolist() == [rustybam_amd.RD_OK]
    counts, status, ctr = engine.nucfreq(*rd.args(), [0, 0], [7, 9], [7, 12])   # an empty region among the regions
    assert (counts[:, 0] & 0x7FFFFFFF).tolist() == [1, 1, 1] and ctr["n_covered"] == 3


def test_bad_reads_are_flagged_not_counted(engine, oracle):
    M, I, D, S = 0, 1, 2, 4
    cig = lambda *x: [(l << 4) | o for l, o in x]
    rd = Reads([0, 0, 0, 0, 0, 0], [10, 20, 30, 40, 50, 60], [0] * 6,
               [cig((5, M)), cig((5, D)), cig((5, S), (3, I)), cig((3, M), (0, D), (3, M)), [], cig((8, M))],
               [[1] * 5, [], [2] * 8, [4] * 6, [], [8] * 4])
    counts, status, ctr = engine.nucfreq(*rd.args(), [0], [0], [100])
    assert status.tolist() == [rustybam_amd.RD_OK, rustybam_amd.RD_BAD_CIGAR, rustybam_amd.RD_BAD_CIGAR, rustybam_amd.RD_BAD_CIGAR,
                               rustybam_amd.RD_BAD_CIGAR, rustybam_amd.RD_SEQ_SHORT]
    assert ctr["n_bad"] == 4
    c = counts & 0x7FFFFFFF
    assert c[10:15, 0].tolist() == [1] * 5 and c[60:64, 3].tolist() == [1] * 4 and c[64:68].sum() == 0
    assert oracle.nucfreq(*rd.slice(5, 6).args(), 0, 0, 100)[0] == -4   # the reference panics on the same read


def test_unsorted_is_refused(engine):
    rd = Reads([0, 0], [50, 10], [0, 0], [[(10 << 4)], [(10 << 4)]], [[1] * 10, [1] * 10])
    with pytest.raises(rustybam_amd.RbError):
        engine.nucfreq(*rd.args(), [0], [0], [100])


def test_depth_cap_one_start_position(engine, oracle):
    """htslib's pileup buffers at most 8000 reads (bam_plp_push drops a read that starts where the iterator stands while more are
    buffered): 8100 reads on one start position -> the first 8000 count"""
    n = 8100
    rng = np.random.default_rng(3)
    seqs = rng.choice([1, 2, 4, 8], size=(n, 4)).tolist()
    rd = Reads([0] * n, [5] * n, [0] * n, [[(4 << 4)]] * n, seqs)
    counts, status, ctr = check_regions(engine, oracle, rd, [(0, 0, 100)])
    assert ctr["max_depth"] == 8000 and ctr["n_dropped"] == 100 and ctr["cap_overflow"] == 0
    assert int((counts[5] & 0x7FFFFFFF).sum()) == 8000
    rd = rd.slice(0, 7000)
    counts, status, ctr = engine.nucfreq(*rd.args(), [0], [0], [100])
    assert ctr["max_depth"] == 7000 and ctr["n_dropped"] == 0 and int((counts[5:9] & 0x7FFFFFFF).sum()) == 4 * 7000


def test_depth_cap_spread_starts_and_regions(engine, oracle):
    """a pile that crosses the cap over many start positions; what is dropped depends on the fetch (the region): the same reads
    through three regions in one call.  A long read dropped at the cap must also stay out of the far tiles it reaches"""
    rng = np.random.default_rng(4)
    n = 14000
    pos = np.sort(rng.integers(0, 40, n)).tolist()
    cigs, seqs = [], []
    for k in range(n):
        l = int(rng.integers(20, 60))
        if rng.random() < 0.1:
            d = int(rng.integers(1, 5))
            cigs.append([((l // 2) << 4), (d << 4) | 2, ((l - l // 2) << 4)])
        else:
            cigs.append([(l << 4)])
        seqs.append(rng.choice([1, 2, 4, 8, 15], size=l).tolist())
    # two long reads that start late on a crowded position: the second of them is dropped there
    at = max(i for i in range(n) if pos[i] == 25) + 1
    for _ in range(2):
        pos.insert(at, 25)
        cigs.insert(at, [(9000 << 4)])
        seqs.insert(at, rng.choice([1, 2, 4, 8], size=9000).tolist())
    n += 2
    rd = Reads([0] * n, pos, [0] * n, cigs, seqs)
    counts, status, ctr = check_regions(engine, oracle, rd, [(0, 0, 9500), (0, 30, 35), (0, 45, 200)])
    assert ctr["max_depth"] >= 8000 and ctr["n_dropped"] > 1000 and ctr["cap_overflow"] == 0
    assert (status == 0).all()


def test_long_insertions_inside_a_tile(engine, oracle):
    """the tile kernel stages the stretch of a read that lies over a tile (the usual build: the tile's own 4096 bases plus 224 of slack;
    the 16-bit build 6144); a longer stretch -- a long insertion between two matches of the same tile -- is read from memory group by group
    instead: both routes, next to each other, insertions on either side of the slack"""
    rng = np.random.default_rng(11)
    M, I, D = 0, 1, 2
    cigs, seqs, poss = [], [], []
    for k, ins in enumerate([10, 3000, 3700, 3790, 3830, 3900, 5900, 6200, 9000, 20000, 7000]):
        c = [(int(rng.integers(50, 300)) << 4) | M, (ins << 4) | I, (int(rng.integers(50, 300)) << 4) | M, (7 << 4) | D, (120 << 4) | M]
        q = sum(w >> 4 for w in c if (w & 15) in (0, 1))
        cigs.append(c)
        seqs.append(rng.choice([1, 2, 4, 8, 15], size=q).tolist())
        poss.append(100 + 37 * k)
    rd = Reads([0] * len(cigs), poss, [0] * len(cigs), cigs, seqs)
    check_regions(engine, oracle, rd, [(0, 0, 1200), (0, 300, 301), (0, 0, 5000)])
    # reads that cover a whole tile: the stretch is the tile's 4096 bases plus the insertion -- on either side of the slack (224 bases, less
    # what the stretch's alignment to a dword takes)
    cigs, seqs, poss = [], [], []
    for k, ins in enumerate([1, 100, 200, 210, 216, 220, 224, 228, 236, 300, 1100]):
        c = [((2300 + 13 * k) << 4) | M, (ins << 4) | I, (2600 << 4) | M, (5 << 4) | D, (900 << 4) | M]
        q = sum(w >> 4 for w in c if (w & 15) in (0, 1))
        cigs.append(c), seqs.append(rng.choice([1, 2, 4, 8, 15], size=q).tolist()), poss.append(3 * k)
    rd = Reads([0] * len(cigs), poss, [0] * len(cigs), cigs, seqs)
    check_regions(engine, oracle, rd, [(0, 0, 8192), (0, 4090, 4100)])


def test_crowded_tile_lane_per_read_path(engine, oracle):
    """more than 512 reads over one tile: the tile kernel gives every read of at most four ops to one lane.  All op kinds that
    fit four ops, both routes mixed (some reads have longer cigars), and the sequence-too-short flag of that route"""
    rng = np.random.default_rng(21)
    M, I, D, N, S, EQ, X = 0, 1, 2, 3, 4, 7, 8
    shapes = [[(40, M)], [(10, EQ), (1, X), (12, EQ)], [(5, M), (2, D), (7, M)], [(3, S), (9, M), (1, I), (6, M)], [(4, M), (30, N), (5, M)],
              [(3, M), (1, I), (3, M), (1, D), (3, M), (2, I), (4, M)]]          # the last one has 7 ops: whole-wave route
    cigs, seqs, poss = [], [], []
    for k in range(900):
        c = [(l << 4) | o for l, o in shapes[int(rng.integers(0, len(shapes)))]]
        q = sum(w >> 4 for w in c if (w & 15) in (0, 1, 4, 7, 8))
        cigs.append(c)
        seqs.append(rng.choice([1, 2, 4, 8, 15, 3], size=q).tolist())
        poss.append(int(rng.integers(100, 1500)))
    order = np.argsort(poss, kind="stable")
    rd = Reads([0] * 900, [poss[i] for i in order], [0] * 900, [cigs[i] for i in order], [seqs[i] for i in order])
    counts, status, ctr = check_regions(engine, oracle, rd, [(0, 0, 2000), (0, 700, 701), (0, 4000, 4100)])
    assert ctr["max_depth"] > 5 and (status == 0).all()
    # the same reads, a few of them with fewer bases than their cigar consumes
    short = set(rng.choice(900, size=12, replace=False).tolist())
    sorted_seqs = [seqs[i] for i in order]
    seqs2 = [s[:max(len(s) - 3, 0)] if i in short else s for i, s in enumerate(sorted_seqs)]
    rd2 = Reads([0] * 900, [poss[i] for i in order], [0] * 900, [cigs[i] for i in order], seqs2)
    counts, status, ctr = engine.nucfreq(*rd2.args(), [0], [0], [2000])
    assert set(np.flatnonzero(status == rustybam_amd.RD_SEQ_SHORT).tolist()) == short


def test_chunk_boundaries_and_short_sequences(engine, oracle):
    """round 6: a wave's reads go by as a stream of 64-op chunks, three under way.  Reads of 63 .. 200 ops (one chunk, exactly one, one
    op into the next, three and a bit), several per wave and tile, tiles that start and end inside chunks -- against the oracle; then
    the same reads with sequences cut short somewhere inside (record().seq()[qpos] out of bounds: the reference panics, the device
    counts what is there and flags the read): the counts must be those of the reads with the missing bases present as code 0."""
    rng = np.random.default_rng(606)
    M, I, D, EQ, X = 0, 1, 2, 7, 8
    cigs, seqs, poss = [], [], []
    for k, n_ops in enumerate([63, 64, 65, 127, 128, 129, 200, 64, 65, 190, 1, 2, 66, 130] * 3):
        c = []
        for j in range(n_ops):
            o = [M, I, M, D, EQ, X][j % 6] if n_ops > 2 else M
            c.append((int(rng.integers(1, 8) if o in (I, D) else rng.integers(20, 160)) << 4) | o)
        if (c[0] & 15) in (I, D):
            c[0] = (c[0] & ~15) | M
        if (c[-1] & 15) in (I, D):
            c[-1] = (c[-1] & ~15) | M
        q = sum(w >> 4 for w in c if (w & 15) in QRY_OPS)
        cigs.append(c), seqs.append(rng.choice([1, 2, 4, 8, 15], size=q).tolist()), poss.append(int(rng.integers(0, 9000)))
    order = np.argsort(poss, kind="stable")
    cigs, seqs, poss = [cigs[i] for i in order], [seqs[i] for i in order], [poss[i] for i in order]
    regions = [(0, 0, 50000), (0, 4000, 4200), (0, 8191, 12289)]
    full = Reads([0] * len(cigs), poss, [0] * len(cigs), cigs, seqs)
    counts_full, status, _ = check_regions(engine, oracle, full, regions)
    assert (status == rustybam_amd.RD_OK).all()
    # cut every third read's sequence short; expected = the full-length read with code 0 (counts nothing) where the bases are missing
    cut = {i: int(rng.integers(0, len(seqs[i]))) for i in range(0, len(seqs), 3) if len(seqs[i]) > 1}
    padded = Reads([0] * len(cigs), poss, [0] * len(cigs), cigs, [s[:cut[i]] + [0] * (len(s) - cut[i]) if i in cut else s for i, s in enumerate(seqs)])
    short = Reads([0] * len(cigs), poss, [0] * len(cigs), cigs, [s[:cut[i]] if i in cut else s for i, s in enumerate(seqs)])
    counts_pad, _, _ = check_regions(engine, oracle, padded, regions)
    rg = np.array(regions, np.int64)
    counts_short, status_short, _ = engine.nucfreq(*short.args(), rg[:, 0], rg[:, 1], rg[:, 2])
    assert np.array_equal(counts_short, counts_pad)
    want = np.array([rustybam_amd.RD_SEQ_SHORT if i in cut else rustybam_amd.RD_OK for i in range(len(cigs))])
    assert np.array_equal(status_short, want)


@pytest.mark.parametrize("n", [126, 127, 128, 129, 255, 256, 257])
def test_byte_counters_and_byte_differences_at_their_limit(engine, oracle, n):
    """round 6: a tile with at most 127 reads in range keeps a position's coverage difference as a signed byte in the counters' padding
    (borrows between the bytes of a dword taken back at the read).  n identical reads: +n where they start, -n behind their end, every
    counter n; 127 is the last tile of that build, 128 the first of the one with byte counters and 16-bit differences, 255 its last, 256 the
    first of the 16-bit counters.  Next to them reads that end and start on neighbouring positions (differences of both signs inside one
    dword)."""
    M, D = 0, 2
    cig = [(300 << 4) | M, (5 << 4) | D, (200 << 4) | M]
    cigs = [cig] * n
    seqs = [[1, 2, 4, 8] * 125] * n
    poss = [1001] * n
    k = (n - 1) // 2   # a second pile that ends where a third one starts, one position apart
    cigs += [[(100 << 4) | M]] * k + [[(64 << 4) | M]] * (n - 1 - k)
    seqs += [[8] * 100] * k + [[2] * 64] * (n - 1 - k)
    poss += [6000] * k + [6101] * (n - 1 - k)
    rd = Reads([0] * len(cigs), poss, [0] * len(cigs), cigs, seqs)
    counts, status, ctr = check_regions(engine, oracle, rd, [(0, 0, 8192), (0, 1000, 1002), (0, 1505, 1507)])
    assert ctr["max_depth"] == n and (status == rustybam_amd.RD_OK).all()


@pytest.mark.parametrize("seed,n", [(0, 150), (1, 200), (2, 250), (3, 300)])
def test_random_reads_128_to_255_per_tile(engine, oracle, seed, n):
    """the middle build of the tile kernel (byte counters, 16-bit coverage differences: tiles with 128 .. 255 reads in range) walks a
    list of its tiles, as the 16-bit build does: random reads of every op type packed so that tiles of all three kinds lie side by side"""
    rng = np.random.default_rng(4242 + seed)
    rd = random_reads(rng, n, n_contig=1, span=5000, long_frac=0.1, max_ops=30)
    regions = [(0, 0, 16000), (0, 4000, 4200), (0, 4096, 8192)]
    counts, status, ctr = check_regions(engine, oracle, rd, regions)
    filt = (rd.flag & 0x704) != 0
    assert np.array_equal(status == rustybam_amd.RD_FILTERED, filt)
