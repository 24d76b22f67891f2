"""Does hipMemsetAsync fill a range that spans several hipMemMap'ed chunks?  (rb_dev_alloc builds buffers from 2 MB chunks; a flaky
tests/test_gpu_alloc.py run read back old bytes behind an rb_dev_memset.)  python tools/memset_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (first: its copy of the HIP runtime serves the process)
import rustybam_amd

MB = 1 << 20
eng = rustybam_amd.Engine(0)
L = eng.L
for mode in ("chunks", "default"):
    os.environ["RB_ALLOC_MODE"] = mode
    size = 300 * MB
    p = eng.dev_alloc(size)
    bad = 0
    rng = np.random.default_rng(1)
    for i in range(300):
        n = int(rng.integers(1, 6 * MB))
        off = int(rng.integers(0, size - n))
        blob = rng.integers(0, 256, n, dtype=np.uint8)
        assert L.rb_dev_upload(eng.ctx, C.c_void_p(p + off), C.c_void_p(blob.ctypes.data), C.c_size_t(n)) == 0
        v = int(rng.integers(0, 256))
        assert L.rb_dev_memset(eng.ctx, C.c_void_p(p + off), v, C.c_size_t(n)) == 0
        back = np.zeros(n, np.uint8)
        assert L.rb_dev_download(eng.ctx, C.c_void_p(back.ctypes.data), C.c_void_p(p + off), C.c_size_t(n)) == 0
        if not (back == v).all():
            bad += 1
            w = np.nonzero(back != v)[0]
            print(mode, "iteration", i, "off", off, "n", n, "first wrong byte", int(w[0]), "wrong bytes", len(w), "chunk phase", (off + int(w[0])) % (2 * MB))
    print(mode, "bad", bad, "of 300")
    eng.dev_free(p)
