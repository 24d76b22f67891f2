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
