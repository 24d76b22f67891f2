"""rb_dev_scan_records, timing only, on three batch shapes of 5e9 ops (device resident); prints a digest of the rows so that variants can be compared"""
import os, sys, time, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch, rustybam_amd
from rustybam_amd import capi
eng = rustybam_amd.Engine(0)
dev = torch.device("cuda:0")
for (n_rec, lo, hi, seed) in ((10_000_000, 300, 700, 0x5EED0004), (1_000_000, 4000, 6000, 0x5EED0003), (20_000_000, 150, 350, 0x5EED0004)):
    n = capi.synth_n_ops(seed, 0, n_rec, lo, hi)
    off = np.zeros(n_rec + 1, np.uint64); off[1:] = np.cumsum(n)
    total = int(off[-1])
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    d_ops = torch.empty(total + 64, dtype=torch.int32, device=dev)
    eng.dev_synth_fill_ops(seed, 0, n_rec, d_off.data_ptr(), d_ops.data_ptr())
    z = torch.zeros(n_rec, dtype=torch.int64, device=dev)
    strand = torch.full((n_rec,), ord("+"), dtype=torch.uint8, device=dev)
    contig = torch.zeros(n_rec, dtype=torch.int32, device=dev)
    view = eng.batch_view(n_rec, total, d_ops.data_ptr(), d_off.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), strand.data_ptr(), contig.data_ptr())
    red = torch.zeros(n_rec * capi.REDUCE_DT.itemsize, dtype=torch.uint8, device=dev)
    norm = torch.zeros(n_rec * capi.NORM_DT.itemsize, dtype=torch.uint8, device=dev)
    for _ in range(2):
        eng.dev_scan_records(view, red.data_ptr(), norm.data_ptr())
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        eng.dev_scan_records(view, red.data_ptr(), norm.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) * 1e3 / 5
    dg = zlib.crc32(red.cpu().numpy().tobytes()) ^ zlib.crc32(norm.cpu().numpy().tobytes())
    print(f"{n_rec} records of {lo}-{hi} ops ({total/1e9:.2f}e9 ops): {ms:.3f} ms ({total*4/ms/1e9*1e3/8000:.3f} of 8 TB/s on the ops), rows crc {dg:08x}")
    del d_ops, red, norm
