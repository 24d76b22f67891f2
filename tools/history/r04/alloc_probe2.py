import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import rustybam_amd
MB = 1 << 20
eng = rustybam_amd.Engine(0)
L = eng.L
def roundtrip(p, at, data):
    assert L.rb_dev_upload(eng.ctx, C.c_void_p(p + at), C.c_void_p(data.ctypes.data), C.c_size_t(data.nbytes)) == 0
    back = np.zeros_like(data)
    assert L.rb_dev_download(eng.ctx, C.c_void_p(back.ctypes.data), C.c_void_p(p + at), C.c_size_t(data.nbytes)) == 0
    return bool(np.array_equal(back, data))
def rd(n, ptr):
    b = np.zeros(n, np.uint8)
    assert L.rb_dev_download(eng.ctx, C.c_void_p(b.ctypes.data), C.c_void_p(ptr), C.c_size_t(n)) == 0
    return b
for mode, size in ((None, (1 << 30) + 4096), (None, 300 * MB), ("chunks", 130 * MB + 12345), ("default", 300 * MB)):
    if mode is None: os.environ.pop("RB_ALLOC_MODE", None)
    else: os.environ["RB_ALLOC_MODE"] = mode
    rng = np.random.default_rng(size & 0xFFFF)
    ptrs = []
    for _ in range(2):
        p = eng.dev_alloc(size); ptrs.append(p)
        blob = rng.integers(0, 256, 5 * MB + 77, dtype=np.uint8)
        print(mode, size, "roundtrips", [roundtrip(p, at, blob) for at in (0, 2 * MB - 1000, size // 2 - 3, size - blob.nbytes)])
    dst = ptrs[0] + size - 3 * MB
    assert L.rb_dev_memset(eng.ctx, C.c_void_p(dst), 0x5A, C.c_size_t(3 * MB)) == 0
    a = rd(3 * MB, dst)
    print("  after memset: 0x5A", int((a == 0x5A).sum()), "of", 3 * MB, "first", a[:3], "last", a[-3:])
    a = rd(3 * MB, dst)
    print("  again:        0x5A", int((a == 0x5A).sum()))
    class V: pass
    v = V(); v.__cuda_array_interface__ = {"shape": (3 * MB,), "typestr": "|u1", "data": (dst, False), "version": 3}
    tc = torch.as_tensor(v, device="cuda").cpu().numpy()
    print("  torch:        0x5A", int((tc == 0x5A).sum()))
    for p in ptrs: eng.dev_free(p)
