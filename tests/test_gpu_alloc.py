"""rb_dev_alloc / rb_dev_free (include/rustybam_amd.h): requests of 256 MB and more are built from 2 MB physical chunks mapped into one
virtual range ("Where the batch lives"); every mode of RB_ALLOC_MODE must behave like plain device memory."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MB = 1 << 20


def _roundtrip(engine, ptr, at, data):
    L = engine.L
    assert L.rb_dev_upload(engine.ctx, C.c_void_p(ptr + at), C.c_void_p(data.ctypes.data), C.c_size_t(data.nbytes)) == 0
    back = np.zeros_like(data)
    assert L.rb_dev_download(engine.ctx, C.c_void_p(back.ctypes.data), C.c_void_p(ptr + at), C.c_size_t(data.nbytes)) == 0
    assert np.array_equal(back, data), at


@pytest.mark.parametrize("mode,size", [(None, (1 << 30) + 4096), (None, 300 * MB), ("chunks", 130 * MB + 12345), ("scatter", 70 * MB),
                                       ("default", (1 << 30) + 4096), ("contiguous", 96 * MB)])
def test_allocations_behave_like_device_memory(engine, mode, size):
    old = os.environ.get("RB_ALLOC_MODE")
    try:
        if mode is None:
            os.environ.pop("RB_ALLOC_MODE", None)
        else:
            os.environ["RB_ALLOC_MODE"] = mode
        rng = np.random.default_rng(size & 0xFFFF)
        ptrs = []
        for _ in range(2):                                     # two live at once, then freed and taken again
            p = engine.dev_alloc(size)
            assert p % 256 == 0
            ptrs.append(p)
            blob = rng.integers(0, 256, 5 * MB + 77, dtype=np.uint8)
            for at in (0, 2 * MB - 1000, size // 2 - 3, size - blob.nbytes):      # across chunk borders, first and last byte
                _roundtrip(engine, p, at, blob)
        assert abs(ptrs[0] - ptrs[1]) >= size
        # a kernel of the library on the memory: a memset, then what it wrote
        assert engine.L.rb_dev_memset(engine.ctx, C.c_void_p(ptrs[0] + size - 3 * MB), 0x5A, C.c_size_t(3 * MB)) == 0
        back = np.zeros(3 * MB, np.uint8)
        assert engine.L.rb_dev_download(engine.ctx, C.c_void_p(back.ctypes.data), C.c_void_p(ptrs[0] + size - 3 * MB), C.c_size_t(3 * MB)) == 0
        assert (back == 0x5A).all()
        for p in ptrs:
            engine.dev_free(p)
        p = engine.dev_alloc(size)
        _roundtrip(engine, p, size - 4096, rng.integers(0, 256, 4096, dtype=np.uint8))
        engine.dev_free(p)
    finally:
        if old is None:
            os.environ.pop("RB_ALLOC_MODE", None)
        else:
            os.environ["RB_ALLOC_MODE"] = old


def test_torch_sees_library_memory(engine):
    torch = pytest.importorskip("torch")
    from rustybam_amd import capi
    buf = capi.DevBuf(engine, torch, (1 << 28) + 5, torch.int32, device=torch.device("cuda", 0))     # 1 GB + 20 bytes: the chunked route
    t = buf.t
    assert t.numel() == (1 << 28) + 5 and t.device.type == "cuda" and t.data_ptr() == buf.ptr
    t.fill_(7)
    t[1 << 27] = 11
    assert int(t.sum().item()) == 7 * ((1 << 28) + 5) + 4
    buf.free()
    assert buf.ptr == 0 and buf.t is None


def test_chunked_buffers_give_their_memory_back(engine):
    """A chunked buffer is many mappings; rb_dev_free unmaps them one by one and releases every chunk (a spanning hipMemUnmap is not
    promised to).  Ten rounds of 1.5 GB: the device's free memory comes back to within a few chunks every time."""
    torch = pytest.importorskip("torch")
    size = (3 << 29) + 12345
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(10):
        p = engine.dev_alloc(size)
        assert engine.L.rb_dev_memset(engine.ctx, C.c_void_p(p), 1, C.c_size_t(size)) == 0
        free_mid, _ = torch.cuda.mem_get_info()
        assert free0 - free_mid >= size - 64 * MB
        engine.dev_free(p)
        free1, _ = torch.cuda.mem_get_info()
        assert abs(free0 - free1) <= 16 * MB, (free0, free1)


def test_placed_allocation_keeps_one_candidate_and_gives_the_others_back(engine):
    """rb_dev_alloc_placed: up to `tries` candidates, a store sweep over each, the fastest kept, the others freed; the buffer it returns
    is ordinary library memory."""
    torch = pytest.importorskip("torch")
    size = (3 << 29) + 4096                                                  # 1.5 GB: the chunked route
    free0 = torch.cuda.mem_get_info()[0]
    ptr, sweeps, kept = engine.dev_alloc_placed(size, 3)
    assert len(sweeps) == 3 and 0 <= kept < 3 and all(s > 0 for s in sweeps) and sweeps[kept] == min(sweeps)
    assert engine.L.rb_dev_alloc_mode(engine.ctx, C.c_void_p(ptr)) == 1
    assert free0 - torch.cuda.mem_get_info()[0] < size + 64 * MB           # one candidate's worth is held, not three
    rng = np.random.default_rng(7)
    blob = rng.integers(0, 256, 5 * MB + 77, dtype=np.uint8)
    for at in (0, 2 * MB - 1000, size - blob.nbytes):
        _roundtrip(engine, ptr, at, blob)
    engine.dev_free(ptr)
    assert free0 - torch.cuda.mem_get_info()[0] < 64 * MB
    ptr, sweeps, kept = engine.dev_alloc_placed(size, 1)                     # one try = rb_dev_alloc (no sweep)
    assert kept == 0 and len(sweeps) == 1
    engine.dev_free(ptr)


def test_placed_allocation_by_the_callers_own_measure(engine):
    """rb_dev_alloc_placed_by: the caller scores every candidate (here: a list of made-up times, and a launch of the library on the
    candidate); the lowest score is kept, an exception in the score ends the search and gives everything back"""
    torch = pytest.importorskip("torch")
    size = (3 << 29) + 4096
    free0 = torch.cuda.mem_get_info()[0]
    seen = []

    def score(ptr):
        seen.append(ptr)
        assert engine.L.rb_dev_memset(engine.ctx, C.c_void_p(ptr), 0x11, C.c_size_t(4 * MB)) == 0     # (it may launch on the context)
        return [5.0, 2.0, 9.0][len(seen) - 1]
    ptr, scores, kept = engine.dev_alloc_placed(size, 3, score)
    assert kept == 1 and scores == [5.0, 2.0, 9.0] and ptr == seen[1] and len(set(seen)) == 3
    back = np.zeros(4 * MB, np.uint8)
    assert engine.L.rb_dev_download(engine.ctx, C.c_void_p(back.ctypes.data), C.c_void_p(ptr), C.c_size_t(4 * MB)) == 0 and (back == 0x11).all()
    engine.dev_free(ptr)

    def bad(ptr):
        raise ValueError("no")
    with pytest.raises(ValueError):
        engine.dev_alloc_placed(size, 3, bad)
    assert free0 - torch.cuda.mem_get_info()[0] < 64 * MB


def _pattern_check(engine, ptr, size, k):
    """a kernel of the library writes the buffer's two ends and its middle; the copy engine reads them back"""
    for at in (0, size // 2 // 4096 * 4096, size - 4 * MB):
        assert engine.L.rb_dev_memset(engine.ctx, C.c_void_p(ptr + at), (k * 7 + 3) & 255, C.c_size_t(4 * MB)) == 0
        back = np.zeros(4 * MB, np.uint8)
        assert engine.L.rb_dev_download(engine.ctx, C.c_void_p(back.ctypes.data), C.c_void_p(ptr + at), C.c_size_t(4 * MB)) == 0
        assert (back == ((k * 7 + 3) & 255)).all(), (k, at)


def test_release_recycles_buffers_in_constant_address_space(engine):
    """rb_dev_release: 300 rounds of a 1 GB buffer allocated, written by a kernel, read back and released -- the same mapping comes back
    every time: no address space is retired, no memory moves, and what the kernel wrote is what a copy reads (every round)"""
    torch = pytest.importorskip("torch")
    size = (1 << 30) + 4096
    engine.dev_cache_trim(0)
    s0 = engine.dev_alloc_stats()
    first = None
    for k in range(300):
        p = engine.dev_alloc(size)
        first = first or p
        assert p == first
        if k % 10 == 0 or k == 299:
            _pattern_check(engine, p, size, k)
        else:
            assert engine.L.rb_dev_memset(engine.ctx, C.c_void_p(p + (k % 200) * 4 * MB), k & 255, C.c_size_t(4 * MB)) == 0
        engine.dev_release(p)
    s1 = engine.dev_alloc_stats()
    assert s1["retired_va"] == s0["retired_va"] and s1["cached"] == (size + 255) // 256 * 256
    assert s1["live"] - s0["live"] <= size + 2 * MB
    free_held = torch.cuda.mem_get_info()[0]
    engine.dev_cache_trim(0)
    s2 = engine.dev_alloc_stats()
    assert s2["cached"] == 0 and s2["retired_va"] - s1["retired_va"] >= size and torch.cuda.mem_get_info()[0] - free_held >= size - 64 * MB


def test_freed_buffers_retire_a_bounded_amount_of_address_space(engine):
    """rb_dev_free retires the virtual range of a chunked buffer (a later mapping at the same addresses was seen to take stores into the
    OLD pages): 300 rounds of 1 GB cost at most 300 x (1 GB + one chunk) of address space, counted by rb_dev_alloc_stats and far below
    the cap; every round's kernel writes are what a copy reads back; the device's memory comes back every time."""
    torch = pytest.importorskip("torch")
    size = (1 << 30) + 4096
    engine.dev_cache_trim(0)
    s0 = engine.dev_alloc_stats()
    free0 = torch.cuda.mem_get_info()[0]
    seen = set()
    for k in range(300):
        p = engine.dev_alloc(size)
        assert p not in seen
        seen.add(p)
        assert engine.L.rb_dev_alloc_mode(engine.ctx, C.c_void_p(p)) == 1
        if k % 25 == 0 or k == 299:
            _pattern_check(engine, p, size, k)
        else:
            assert engine.L.rb_dev_memset(engine.ctx, C.c_void_p(p), k & 255, C.c_size_t(2 * MB)) == 0
            back = np.zeros(4096, np.uint8)
            assert engine.L.rb_dev_download(engine.ctx, C.c_void_p(back.ctypes.data), C.c_void_p(p + MB), C.c_size_t(4096)) == 0 and (back == (k & 255)).all()
        engine.dev_free(p)
    s1 = engine.dev_alloc_stats()
    grown = s1["retired_va"] - s0["retired_va"]
    assert 300 * size <= grown <= 300 * (size + 2 * MB), grown
    assert s1["live"] == s0["live"] and s1["live"] + s1["retired_va"] < s1["va_cap"] // 8
    assert abs(free0 - torch.cuda.mem_get_info()[0]) <= 64 * MB


def test_address_space_cap_falls_back_to_plain_memory(engine):
    """beyond RB_ALLOC_VA_CAP_GB (live + retired ranges) the chunked route is refused: rb_dev_alloc still serves the request, from plain
    hipMalloc, and says so"""
    size = (1 << 30) + 4096
    s0 = engine.dev_alloc_stats()
    old = os.environ.get("RB_ALLOC_VA_CAP_GB")
    os.environ["RB_ALLOC_VA_CAP_GB"] = str((s0["live"] + s0["retired_va"]) // (1 << 30) + 1)   # (room for less than one more gigabyte)
    try:
        p = engine.dev_alloc(size)
        assert engine.L.rb_dev_alloc_mode(engine.ctx, C.c_void_p(p)) == 0
        assert engine.dev_alloc_stats()["fallbacks"] == s0["fallbacks"] + 1
        _pattern_check(engine, p, size, 5)
        engine.dev_free(p)
    finally:
        if old is None:
            del os.environ["RB_ALLOC_VA_CAP_GB"]
        else:
            os.environ["RB_ALLOC_VA_CAP_GB"] = old
    p = engine.dev_alloc(size)
    assert engine.L.rb_dev_alloc_mode(engine.ctx, C.c_void_p(p)) == 1
    engine.dev_free(p)


def test_a_released_placed_buffer_is_not_measured_again(engine):
    size = (3 << 29) + 4096
    engine.dev_cache_trim(0)
    ptr, sweeps, kept = engine.dev_alloc_placed(size, 3)
    assert kept >= 0 and len([s for s in sweeps if s > 0]) == 3
    engine.dev_release(ptr)
    ptr2, sweeps2, kept2 = engine.dev_alloc_placed(size, 3)
    assert ptr2 == ptr and kept2 == -1 and all(s < 0 for s in sweeps2)
    engine.dev_release(ptr2)
    p3 = engine.dev_alloc(size)                       # (a plain request of the size takes it as well)
    assert p3 == ptr
    engine.dev_free(p3)
    assert engine.dev_alloc_stats()["cached"] == 0
