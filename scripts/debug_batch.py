import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from collections import defaultdict
import numpy as np
import golden_io, oracle_lib as ol
from mgl_amd import smithwaterman as sw

a = sw.MicrosoftSmithWaterman(0)
rows = golden_io.load(sys.argv[1] if len(sys.argv) > 1 else "random")
groups = defaultdict(list)
for g in rows:
    groups[(g.params, g.strategy)].append(g)
shown = 0
for (params, strategy), gs in groups.items():
    res = a.align_batch([g.t for g in gs], [g.q for g in gs], params, strategy)
    for k, g in enumerate(gs):
        if res.cigars[k] != g.cigar or int(res.offsets[k]) != g.offset or tuple(int(x) for x in res.scores[k]) != g.score:
            w = k // 4 * 4
            mates = gs[w:w + 4]
            print("FAIL k=%d (lane grp %d) params=%s strat=%d" % (k, k % 4, params, strategy))
            print("   mates (tl,ql):", [(len(m.t), len(m.q)) for m in mates])
            print("   gpu:", res.cigars[k], int(res.offsets[k]), tuple(int(x) for x in res.scores[k]))
            print("   ref:", g.cigar, g.offset, g.score)
            r2 = a.align_batch([m.t for m in mates], [m.q for m in mates], params, strategy)
            print("   4-batch:", [r2.cigars[i] == m.cigar for i, m in enumerate(mates)])
            i = k - w
            btr = a.expand_slot(i, len(g.t), len(g.q))
            o = ol.oracle_align(g.t, g.q, params, strategy, want_btr=True)
            d = np.argwhere(btr[1:, 1:] != o["btr"][1:, 1:])
            print("   ndiff", len(d), [(int(x) + 1, int(y) + 1, int(btr[x + 1, y + 1]), int(o["btr"][x + 1, y + 1])) for x, y in d[:20]])
            shown += 1
            if shown >= 8:
                sys.exit(0)
print("done, failures shown:", shown)
