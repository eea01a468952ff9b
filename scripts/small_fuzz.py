"""Differential run of small_pair() (sw_small_pair.h: one wave per pair, scores kept in LDS, the walk reads its moves off the scores) --
through sw_small_kernel forced onto every batch and, for a sample of each batch, through the one-pair entry's mailboxes
(sw_service_kernel) -- against the reference's own code (oracle/_ref through tests/oracle_lib.py): random geometries up to 512 rows,
parameter sets whose scores fit 16 bits and ones that need the 32-bit form, every strategy; related pairs with substitutions and gaps,
unrelated pairs, ragged lengths.  python scripts/small_fuzz.py [geometries] [pairs per geometry]"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as ol
from mgl_amd import _lib, smithwaterman as sw

n_geo = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rng = np.random.default_rng(20261004)
PARAMS = [(200, -150, 260, 11), (1, -1, 1, 1), (5, -4, 10, 1), (10, -30, 40, 1), (1000, -800, 1500, 50), (3, -3, 0, 0), (25, -50, 110, 6), (100, -100, 300, 10)]
alpha = np.frombuffer(b"ACGTN", np.uint8)
a = sw.MicrosoftSmithWaterman(0)
a.set_small_kernel(2)
L = _lib.lib()
total = bad = small = mail = 0
for gi in range(n_geo):
    tl = int(rng.integers(1, 513))
    ql = int(rng.integers(1, min(700, max(2, 36000 // max(tl, 8))) + 1))
    params = PARAMS[gi % len(PARAMS)]
    strategy = ol.STRATEGIES[(gi // 3) % 4]
    ragged = gi % 4 == 1
    ts, qs = [], []
    for k in range(n):
        tlk = int(rng.integers(max(1, tl // 2), tl + 1)) if ragged else tl
        qlk = int(rng.integers(max(1, ql // 2), ql + 1)) if ragged else ql
        t = alpha[rng.integers(0, 4 if k % 9 else 5, tlk)]
        if k % 5 == 4:
            q = alpha[rng.integers(0, 4, qlk)]
        else:
            src = np.resize(t[int(rng.integers(0, max(1, tlk // 3))):], qlk + 100).copy()
            gap, at = int(rng.integers(1, 40)), int(rng.integers(1, max(2, qlk - 1)))
            if k % 5 in (0, 1): src = np.concatenate([src[:at], src[at + gap:]])
            elif k % 5 == 2: src = np.concatenate([src[:at], alpha[rng.integers(0, 4, gap)], src[at:]])
            sub = rng.random(len(src)) < 0.04
            src[sub] = alpha[rng.integers(0, 4, int(sub.sum()))]
            q = src[:qlk]
        if k % 7 == 3:   # bytes outside ACGT in the QUERY alone (a target of ACGT is staged as base codes: such a byte must differ from every base)
            q = q.copy()
            q[rng.integers(0, len(q), 1 + len(q) // 40)] = np.frombuffer(b"NaRc", np.uint8)[rng.integers(0, 4, 1 + len(q) // 40)]
        ts.append(t.tobytes()); qs.append(q.tobytes())
    res = a.align_batch(ts, qs, params, strategy)
    took = a.timing().fill_kernel == 8
    small += took
    off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=8)
    wrong = [k for k in range(n) if int(res.offsets[k]) != int(off[k]) or (res.scores[k] != sc[k]).any() or res.cigars[k] != cg[k]]
    # a sample through the one-pair entry (mailboxes where the pair fits one, else the coalescer)
    c0, l0 = C.c_int64(), C.c_int64(); L.mgl_sw_service_stats(C.byref(c0), C.byref(l0))
    for k in range(0, n, 10):
        cigar, o, ez = sw.align(ts[k], qs[k], params, strategy)
        if cigar != cg[k] or o != int(off[k]) or tuple(ez) != tuple(int(x) for x in sc[k]): wrong.append(("one-pair", k))
    c1, l1 = C.c_int64(), C.c_int64(); L.mgl_sw_service_stats(C.byref(c1), C.byref(l1))
    mail += c1.value - c0.value
    total += n; bad += len(wrong)
    if wrong or gi % 15 == 0:
        print(f"geometry {gi}: {tl} x {ql}{' ragged' if ragged else ''}, params {params}, strategy {strategy}: {'sw_small_kernel' if took else 'another kernel'}, mismatches {len(wrong)} {wrong[:3]}", flush=True)
print(f"TOTAL {total} pairs over {n_geo} geometries, {bad} mismatches; {small} batches on sw_small_kernel; {mail} one-pair calls through mailboxes")
sys.exit(1 if bad else 0)
