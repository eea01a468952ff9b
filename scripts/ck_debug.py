"""Debug aid: one uniform batch through the forced lane kernel against the CPU checker; prints which fields differ.  python scripts/ck_debug.py tl ql [n] [strategy]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as ol
from mgl_amd import smithwaterman as sw
tl, ql = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 130
strategy = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(5)
alpha = np.frombuffer(b"ACGT", np.uint8)
ts, qs = [], []
for k in range(n):
    t = alpha[rng.integers(0, 4, tl)]
    src = np.resize(t[int(rng.integers(0, max(1, tl // 2))):], ql + 50).copy()
    if k % 3 == 1 and ql > 8:
        at = int(rng.integers(2, ql - 2)); src = np.concatenate([src[:at], src[at + 3:]])
    if k % 3 == 2 and ql > 8:
        at = int(rng.integers(2, ql - 2)); src = np.concatenate([src[:at], alpha[rng.integers(0, 4, 2)], src[at:]])
    sub = rng.random(len(src)) < 0.03
    src[sub] = alpha[rng.integers(0, 4, int(sub.sum()))]
    ts.append(t.tobytes()); qs.append(src[:ql].tobytes())
a = sw.MicrosoftSmithWaterman(0); a.set_lane_kernel(2)
params = (200, -150, 260, 11)
res = a.align_batch(ts, qs, params, strategy)
print("kernel", a.fill_kernel_name(a.timing()))
off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=4)
bad_sc = [k for k in range(n) if (res.scores[k] != sc[k]).any()]
bad_cg = [k for k in range(n) if res.cigars[k] != cg[k] or res.offsets[k] != off[k]]
print("score mismatches", len(bad_sc), bad_sc[:10]); print("cigar/offset mismatches", len(bad_cg), bad_cg[:10])
for k in bad_cg[:6]:
    print(k, "got", res.offsets[k], res.cigars[k], "want", off[k], cg[k], "scores", list(res.scores[k]), list(sc[k]))
