"""Debug helper: compare the GPU backtrack matrix with the oracle's for failing golden cases."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import golden_io, oracle_lib as ol
from mgl_amd import smithwaterman as sw

suite = sys.argv[1] if len(sys.argv) > 1 else "random"
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = golden_io.load(suite)
bad = 0
for g in rows:
    btr, ez = sw.backtrack_matrix(g.t, g.q, g.params, g.strategy)
    o = ol.oracle_align(g.t, g.q, g.params, g.strategy, want_btr=True)
    d = np.argwhere(btr[1:, 1:] != o["btr"][1:, 1:])
    cig, off, _ = sw.align(g.t, g.q, g.params, g.strategy)
    if len(d) or tuple(ez) != g.score or cig != g.cigar:
        bad += 1
        print("CASE tl=%d ql=%d params=%s strat=%d" % (len(g.t), len(g.q), g.params, g.strategy))
        print("  score gpu", tuple(ez), "gold", g.score, "cigar gpu", cig, "gold", g.cigar)
        print("  ndiff", len(d), "first", [(int(a) + 1, int(b) + 1, int(btr[a + 1, b + 1]), int(o["btr"][a + 1, b + 1])) for a, b in d[:12]])
        if bad >= limit:
            break
print("bad", bad)
