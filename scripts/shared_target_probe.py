"""The shared-target protein kernel (MGL_SW_FLAG_SHARED_TARGET, sw_dp16_lane_matrix.hip) against the packed kernel of the grouped
path on the protein bench's workload, pair for pair, and against the CPU restatement's extension on a sample.
python scripts/shared_target_probe.py [queries] [db] [workspace GiB]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
from mgl_amd import protein, smithwaterman as sw

Q = int(sys.argv[1]) if len(sys.argv) > 1 else 400
D = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
WS = float(sys.argv[3]) if len(sys.argv) > 3 else 128
QL = 300
rng = np.random.default_rng(42)
code, mat = protein.blosum62()
lens = np.clip(np.exp(rng.normal(5.7, 0.55, size=D)).astype(np.int64), 40, 2000)
lens.sort()
db_off = np.zeros(D + 1, np.int64); np.cumsum(lens, out=db_off[1:])
db = protein.random_proteins(rng, 1, int(db_off[-1]))[0]
queries = protein.random_proteins(rng, Q, QL)
for k in range(0, Q, 5):
    d = int(rng.integers(0, D))
    if lens[d] >= QL:
        s = int(rng.integers(0, lens[d] - QL + 1)); frag = db[db_off[d] + s: db_off[d] + s + QL].copy()
        mut = rng.random(QL) < 0.4
        frag[mut] = protein.random_proteins(rng, 1, int(mut.sum()))[0]
        queries[k] = frag
dev = torch.device("cuda", 0)
ds = protein.DatabaseSearch(db, db_off, queries, dev)
print(f'shared-target tiles up to {ds.shared_max_tl} residues; longer targets: {0 if ds.long is None else ds.long.n} pairs through the packed kernel', flush=True)
a = sw.MicrosoftSmithWaterman(0)
a.set_workspace(int(WS * (1 << 30)))
protein.run_matrix(ds.shared, a, code, mat, 11, 1, shared_target=True); torch.cuda.synchronize()
print("shared part: kernel", a.fill_kernel_name(a.timing()), flush=True)
ds.run(a, code, mat); torch.cuda.synchronize()
bad = int((ds.shared.status != 0).sum())
print(f"shared part: {ds.shared.n} pairs, {bad} with a status (CIGAR longer than the slot, or a broken promise)", flush=True)
# the same pairs through the grouped path (no promise of a shared target)
ref = protein.IndexedBatch(ds.shared.targets, ds.shared.t_off, ds.shared.t_len, ds.shared.queries, ds.shared.q_off, ds.shared.q_len, ds.shared.max_tl, QL, 256)
protein.run_matrix(ref, a, code, mat, 11, 1, grouped=True); torch.cuda.synchronize()
print("grouped path:", a.fill_kernel_name(a.timing()), flush=True)
same = torch.equal(ref.scores, ds.shared.scores) and torch.equal(ref.offsets, ds.shared.offsets) and torch.equal(ref.status, ds.shared.status)
ok = ref.status == 0
same_c = torch.equal(ref.cigar_len[ok], ds.shared.cigar_len[ok]) and torch.equal(ref.cigars[ok], ds.shared.cigars[ok])
print(f"scores / offsets / status identical: {same}; CIGARs identical: {same_c}", flush=True)
if not same:
    diff = (ref.scores != ds.shared.scores).any(dim=1).nonzero().flatten()
    print("first differing pairs:", diff[:10].tolist(), "of", diff.numel())
    k = int(diff[0]); print(ref.scores[k].tolist(), ds.shared.scores[k].tolist(), int(ds.shared.t_len[k]))
for label, run in (("shared target", lambda: ds.run(a, code, mat)), ("shared part alone", lambda: protein.run_matrix(ds.shared, a, code, mat, 11, 1, shared_target=True)),
                   ("grouped", lambda: protein.run_matrix(ref, a, code, mat, 11, 1, grouped=True))):
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    cells = ds.cells if label.startswith("shared target") else int(lens[lens <= ds.shared_max_tl].sum()) * ds.Qs * QL
    print(f"{label}: {dt*1e3:.1f} ms per pass = {cells/dt/1e9:.0f} GCUPS", flush=True)
assert same and same_c
