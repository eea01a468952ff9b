"""Packed-int16 kernel at the edge of its range guard (dp16_range_ok, sw_dp16.hip): random parameter sets, the largest
geometry the guard admits for them, sequences that drive H to its extremes (all match, all mismatch, periodic, random),
all four overhang strategies, compared with the CPU restatement.  GPU box:  python scripts/range_fuzz.py [cases] [seed]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as ol
from mgl_amd.smithwaterman import MicrosoftSmithWaterman



def range_ok(tl, ql, match, mismatch, gopen, gext):   # mirrors dp16_range_ok
    if match <= 0 or gopen < gext:
        return False
    top = match * min(tl, ql) + gext * (tl + ql)
    low = -3 * gopen - (match - mismatch) - 2 * gext - 64
    return 32767 - top + low >= -32768 and match - mismatch <= 30000 and gopen <= 10000 and gext <= 5000 and match + 2 * gext <= 30000


def largest_ql(tl, p):
    lo, hi = 0, 4000
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if range_ok(tl, mid, *p):
            lo = mid
        else:
            hi = mid - 1
    return lo


def run(cases, seed, log=print):
    """returns (mismatching batches, batches that took the packed kernel)"""
    rng = np.random.default_rng(seed)
    a = MicrosoftSmithWaterman(0)
    a.set_small_kernel(1)   # eight pairs per batch: without this a default context runs one wave per pair (sw_small.hip), not the packed kernel under test
    bad = packed = 0
    for case in range(cases):
        gext = int(rng.integers(0, 40))
        p = (int(rng.integers(1, 400)), -int(rng.integers(1, 3000)), gext + int(rng.integers(0, 3000)), gext)
        tl = int(rng.choice([16, 40, 100, 256, 300, 1000, 2000]))
        ql = min(largest_ql(tl, p), 3000)
        if ql < 2:
            continue
        ql -= int(rng.integers(0, 2))
        al = np.frombuffer(b"AC", np.uint8)
        ts = [b"A" * tl, b"A" * tl, (b"AC" * tl)[:tl], al[rng.integers(0, 2, tl)].tobytes(), b"A" * (tl // 2) + b"C" * (tl - tl // 2)]
        qs = [b"A" * ql, b"C" * ql, (b"CA" * ql)[:ql], al[rng.integers(0, 2, ql)].tobytes(), b"C" * (ql // 2) + b"A" * (ql - ql // 2)]
        ts, qs = ts + ts[:3], qs + qs[:3]   # eight pairs: one wave
        for strategy in ol.STRATEGIES:
            res = a.align_batch(ts, qs, p, strategy)
            packed += a.timing().packed16
            off, sc, cg = ol.oracle_align_batch(ts, qs, p, strategy, nthreads=8)
            if not ((res.offsets == off).all() and (res.scores == sc).all() and res.cigars == cg):
                bad += 1
                log("MISMATCH", p, tl, ql, strategy)
        if case % 20 == 0:
            log(f"case {case}: params {p} tl {tl} ql {ql}: ok so far ({bad} bad)")
    a.close()
    return bad, packed


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    bad, packed = run(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 5, lambda *x: print(*x, flush=True))
    print(f"{cases} parameter sets x 4 strategies x 8 pairs at the guard's edge: {bad} mismatching batches; packed kernel taken in {packed} batches")
    sys.exit(1 if bad else 0)
