import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["RB_DEBUG_TRIM_NO_SERIAL"] = "1"
import torch
import rustybam_amd
from rustybam_amd import trim_driver
from rbtest_util import read_paf
r = read_paf("/root/repo/tests/golden/asm_small.paf")
names = {q: i for i, q in enumerate(sorted(set(r.q_name)))}
group = np.array([names[q] for q in r.q_name])
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
eng = rustybam_amd.Engine(0, torch.cuda.current_stream().cuda_stream)
T = trim_driver.ResidentTrim(eng, torch, dev, r.ops, r.op_off, r.t_st, r.t_en, r.q_st, r.q_en, r.strand, group)
left, right, unseen, contained = trim_driver.select_pairs(T.order, T.grp_sorted, T.q_st, T.q_en)
need = T.cur_n[left] + T.cur_n[right]
poff = T.cursor + np.r_[0, np.cumsum(need)[:-1]].astype(np.uint64)
d_l = torch.from_numpy(left.view(np.int32)).to(dev); d_r = torch.from_numpy(right.view(np.int32)).to(dev); d_po = torch.from_numpy(poff.view(np.int64)).to(dev)
d_rows = torch.zeros(len(left) * 128, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
eng.dev_overlap_split(T.view, T.d_norm.data_ptr(), len(left), d_l.data_ptr(), d_r.data_ptr(), d_po.data_ptr(), (1, 1, 1), 0, d_rows.data_ptr(), T.d_ops.data_ptr())
torch.cuda.synchronize()
rows = d_rows.cpu().numpy().view(rustybam_amd.PAIR_DT)
norm = T.d_norm.cpu().numpy().view(rustybam_amd.NORM_DT)
pend = rows["status"] == 0x7FFF0001
print("pairs", len(rows), "pending", int(pend.sum()))
for i in np.nonzero(pend)[0]:
    l, rr = int(left[i]), int(right[i])
    ov = min(int(T.q_en[l]), int(T.q_en[rr])) - max(int(T.q_st[l]), int(T.q_st[rr]))
    print("  why", int(rows["split_idx"][i]), "L n_ops", int(norm["n_ops"][l]), "flags", int(norm["flags"][l]), chr(r.strand[l]), "R n_ops", int(norm["n_ops"][rr]), "flags", int(norm["flags"][rr]), chr(r.strand[rr]), "overlap", ov,
          "Lq", int(T.q_st[l]), int(T.q_en[l]), "Rq", int(T.q_st[rr]), int(T.q_en[rr]))
