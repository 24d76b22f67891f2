"""Quantile table of the log-normal op-count distribution of SURVEY.md 8(d)'s secondary config-2 workload (mu = ln 2000, sigma = 1.35,
clipped to [31, 80000]): 257 integer quantiles at k / 256, so that the generator is integer-only and the same bytes come out of the C
twin (rustybam_amd/csrc/synth.h) and the numpy twin (rustybam_amd/workload.py).  Prints the table; both files hold a copy."""
import math

from scipy.stats import norm

MU, SIGMA, LO, HI = math.log(2000.0), 1.35, 31, 80000
q = []
for k in range(257):
    p = min(max(k / 256.0, 1e-9), 1 - 1e-9)
    v = math.exp(MU + SIGMA * norm.ppf(p))
    q.append(int(min(max(round(v), LO), HI)))
q[0], q[256] = LO, HI
for i in range(0, 257, 12):
    print("    " + ", ".join(str(x) for x in q[i:i + 12]) + ",")
