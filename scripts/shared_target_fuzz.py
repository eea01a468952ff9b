"""Fuzz of the shared-target protein kernel (MGL_SW_FLAG_SHARED_TARGET) against the CPU restatement's extension: random tiles (target length,
query length, pair count of the last tile), random gap penalties inside the byte table's range, the four overhang strategies, BLOSUM62 and
a random symmetric matrix.  python scripts/shared_target_fuzz.py [rounds] [seed]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
import oracle_lib as ol
from mgl_amd import protein, smithwaterman as sw
from test_gpu_matrix import _shared_batch, _tiles, oracle_matrix_batch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda", 0)
a = sw.MicrosoftSmithWaterman(0)
code, blosum = protein.blosum62()
pairs = cells = 0
t0 = time.time()
for r in range(rounds):
    if r % 3 == 2:   # a random symmetric matrix with entries in [-8, 12]
        m = rng.integers(-8, 13, (32, 32)); mat = np.triu(m) + np.triu(m, 1).T; mat = mat.astype(np.int8)
    else:
        mat = blosum
    smin = int(mat.min())
    e = int(rng.integers(0, 6)); o = int(rng.integers(max(e, -smin - e, 1), 40))   # 0 <= S + e + o for every entry
    strategy = ol.STRATEGIES[r % 4]
    nt = int(rng.integers(6, 20))
    shapes = [(int(rng.integers(1, 1400)) if rng.random() < 0.8 else int(rng.integers(1, 40)), int(rng.integers(1, 420)) if rng.random() < 0.85 else int(rng.integers(1, 9))) for _ in range(nt)]
    ts, qs = _tiles(rng, shapes, int(rng.integers(1, 129)))
    stride = 2 * (max(len(t) for t in ts) + max(len(q) for q in qs)) + 16
    stride = (stride + 3) // 4 * 4
    b = _shared_batch(ts, qs, dev, stride)
    slots = None if r % 2 else str(int(rng.integers(2, 9)))
    if slots: os.environ["MGL_SW_DEBUG_LANE_SLOTS"] = slots
    protein.run_matrix(b, a, code, mat, o, e, strategy, shared_target=True)
    torch.cuda.synchronize()
    os.environ.pop("MGL_SW_DEBUG_LANE_SLOTS", None)
    kern = a.fill_kernel_name(a.timing())
    off, sc, cg = oracle_matrix_batch(ts, qs, code, mat, o, e, strategy, stride)
    ok = int((b.status != 0).sum()) == 0 and (b.offsets.cpu().numpy() == off).all() and (b.scores.cpu().numpy() == sc).all() and b.cigar_strings() == cg
    n_cells = sum(len(t) * len(q) for t, q in zip(ts, qs))
    pairs += len(ts); cells += n_cells
    print(f"round {r}: {nt} tiles, {len(ts)} pairs, {n_cells/1e6:.0f} M cells, gap {o}/{e}, strategy {strategy}, slots {slots or 'chip'}, {kern}: {'identical' if ok else 'MISMATCH'}", flush=True)
    assert ok and kern == "sw_dp16_lane_matrix_kernel"
print(f"{rounds} rounds, {pairs} pairs, {cells/1e9:.2f} G cells against the CPU restatement's extension: 0 mismatches ({time.time()-t0:.0f} s)")
