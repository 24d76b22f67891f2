"""How rb_plan_create cuts a batch into tiles of short records and orders its schedule (rb_plan_tiles_host: the same code on host arrays, no
device): the tile kernel (rustybam_amd/csrc/k_tile.hip) relies on every property checked here -- consecutive records, 8 .. short_max ops each,
at most 32 records and 4064 ops a tile, the jobs of a tile side by side in the schedule, records below 8 ops in pass-through tiles, longer
records in front of the schedule, longest first."""
import ctypes as C

import numpy as np
import pytest

from rustybam_amd import capi

MAX_OPS, MAX_REC = 4064, 32


def cut(n_ops, short_max=0):
    L = capi.lib()
    L.rb_plan_tiles_host.restype = C.c_int64
    n = len(n_ops)
    off = np.zeros(n + 1, np.uint64)
    off[1:] = np.cumsum(n_ops)
    sched = np.zeros(max(n, 1), np.uint32)
    tiles = np.zeros(3 * max(n, 1), np.uint32)
    n_long = C.c_uint64(0)
    nt = L.rb_plan_tiles_host(C.c_uint64(n), off.ctypes.data_as(C.c_void_p), C.c_uint64(short_max), sched.ctypes.data_as(C.c_void_p),
                              tiles.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.byref(n_long))
    assert nt >= 0
    return sched[:n], tiles[: 3 * nt].reshape(-1, 3), int(n_long.value), off


def check(n_ops, short_max=2048):
    n_ops = np.asarray(n_ops, np.uint64)
    sched, tiles, n_long, off = cut(n_ops, short_max)
    short_max = min(short_max, MAX_OPS)                                              # (the line is capped at what a tile holds)
    n = len(n_ops)
    assert sorted(sched.tolist()) == list(range(n))                                  # a permutation
    if len(tiles) == 0:                                                              # nothing to tile: longest first, all of them
        assert n_long == n and (np.diff(n_ops[sched].astype(np.int64)) <= 0).all()
        assert not ((n_ops >= 8) & (n_ops <= short_max)).any()
        return sched, tiles
    long_ = n_ops > short_max
    assert n_long == int(long_.sum())
    head = sched[:n_long]
    assert long_[head].all() and (np.diff(n_ops[head].astype(np.int64)) <= 0).all()   # the per-record kernel's launch: longest first
    tail = sched[n_long:]
    assert (np.diff(tail.astype(np.int64)) > 0).all() and not long_[tail].any()      # the others in memory order
    slot_of = np.zeros(n, np.int64)
    slot_of[sched] = np.arange(n)
    seen = np.zeros(n, bool)
    prev_end = 0
    for first, cnt, slot0 in tiles.tolist():
        tiny = bool(first >> 31)
        first &= 0x7FFFFFFF
        assert 1 <= cnt <= MAX_REC and first >= prev_end                             # tiles follow one another along the batch
        prev_end = first + cnt
        rs = np.arange(first, first + cnt)
        assert not seen[rs].any()
        seen[rs] = True
        m = n_ops[rs]
        if tiny:
            assert (m < 8).all()
        else:
            assert (m >= 8).all() and (m <= short_max).all() and int(m.sum()) <= MAX_OPS
        assert np.array_equal(slot_of[rs], slot0 + np.arange(cnt))                   # the tile's jobs side by side
    assert np.array_equal(seen, ~long_)                                              # every short record in exactly one tile
    return sched, tiles


@pytest.mark.parametrize("seed", range(6))
def test_random_batches(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 4000))
    kind = rng.choice(4, n, p=[.55, .15, .2, .1])
    n_ops = np.where(kind == 0, rng.integers(8, 700, n), np.where(kind == 1, rng.integers(0, 8, n), np.where(kind == 2, rng.integers(700, 2049, n), rng.integers(2049, 90000, n))))
    check(n_ops)
    check(n_ops, short_max=int(rng.integers(8, 4064)))


def test_shapes_that_fill_a_tile_exactly():
    check([8] * 100)                                     # 32 records fill a tile long before 4064 ops do
    check([2032, 2032, 2032, 2032, 1, 2032])             # two records are a tile; a tiny record in between is a tile of its own kind
    check([2048] * 7)                                    # one record of 2048 ops and the next do not fit together: 4096 > 4064
    check([4064], short_max=4064)
    check([4065], short_max=9999)                        # (the line is capped at what a tile holds: this record stays per record)
    s, t = check([500] * 8 + [3000] + [500] * 9)         # a long record closes the tile in front of it
    assert [x[1] for x in t.tolist()] == [8, 8, 1]
    s, t = check([7] * 70)                               # tiny records only: nothing for the tile kernel, the schedule stays whole
    assert len(t) == 0


def test_empty_and_degenerate():
    sched, tiles, n_long, _ = cut(np.zeros(0, np.uint64))
    assert len(sched) == 0 and len(tiles) == 0
    check([0, 0, 0])
    check([100000, 5, 100000])
    check([9])
