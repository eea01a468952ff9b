"""Differential run of the checkpointed lane kernel (sw_dp16_lane_ck_kernel, forced onto small batches) against the reference's own
code (oracle/_ref through tests/oracle_lib.py): random geometries, parameter sets and strategies; related pairs with substitutions and
gaps of up to 80 cells, unrelated pairs.  python scripts/lane_ck_fuzz.py [geometries] [pairs per geometry]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as ol
from mgl_amd import smithwaterman as sw

n_geo = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(20260104)
alpha = np.frombuffer(b"ACGT", np.uint8)
PARAMS = [(200, -150, 260, 11), (25, -50, 110, 6), (1, -1, 1, 1), (5, -4, 10, 1), (10, -30, 40, 1), (100, -100, 300, 10), (3, -1, 4, 3)]
a = sw.MicrosoftSmithWaterman(0)
a.set_lane_kernel(2)
total = bad = ck_runs = other = 0
t00 = time.time()
for gi in range(n_geo):
    tl = int(rng.integers(1, 700)) if gi % 3 else int(rng.choice([32, 64, 256, 512, 33, 31, 17]))
    ql = int(rng.integers(1, 320)) if gi % 4 else int(rng.choice([32, 33, 64, 150, 31, 1, 8]))
    params = PARAMS[int(rng.integers(0, len(PARAMS)))]
    strategy = ol.STRATEGIES[int(rng.integers(0, 4))]
    ts, qs = [], []
    for k in range(n):
        t = alpha[rng.integers(0, 4, tl)]
        if k % 6 == 5:
            q = alpha[rng.integers(0, 4, ql)]
        else:
            src = np.resize(t[int(rng.integers(0, max(1, tl // 2))):], ql + 200).copy()
            if k % 6 in (0, 1) and ql > 4:
                g, at = int(rng.integers(1, 80)), int(rng.integers(1, ql - 1))
                src = np.concatenate([src[:at], src[at + g:]])
            elif k % 6 == 2 and ql > 4:
                g, at = int(rng.integers(1, 80)), int(rng.integers(1, ql - 1))
                src = np.concatenate([src[:at], alpha[rng.integers(0, 4, g)], src[at:]])
            sub = rng.random(len(src)) < 0.04
            src[sub] = alpha[rng.integers(0, 4, int(sub.sum()))]
            q = src[:ql]
        ts.append(t.tobytes()); qs.append(q.tobytes())
    res = a.align_batch(ts, qs, params, strategy)
    k7 = a.timing().fill_kernel == 7
    ck_runs += k7; other += not k7
    off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=16)
    m = int((res.offsets != off).sum() + (res.scores != sc).any(axis=1).sum() + sum(x != y for x, y in zip(res.cigars, cg)))
    total += n; bad += m
    if m or gi % 10 == 0:
        print(f"geometry {gi}: {tl} x {ql}, params {params}, strategy {strategy}: kernel {a.fill_kernel_name(a.timing())}, mismatches {m}", flush=True)
print(f"TOTAL {total} pairs over {n_geo} geometries, {bad} mismatches; {ck_runs} batches on sw_dp16_lane_ck_kernel, {other} elsewhere "
      f"(16-row strips or scores beyond 16 bits); {time.time()-t00:.0f} s")
sys.exit(1 if bad else 0)
