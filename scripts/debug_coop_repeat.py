"""Repeat one long-read batch and compare every pass with the first (hunting a non-deterministic result)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from mgl_amd import device_batch, synth
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, concat
n = int(sys.argv[1]); passes = int(sys.argv[2]); length = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
rng = synth.rng_for(11)
base = [synth.ont_pair(rng, length) for _ in range(min(n, 32))]
ts = [base[k % len(base)][0].tobytes() for k in range(n)]; qs = [base[k % len(base)][1].tobytes() for k in range(n)]
td, toff = concat(ts); qd, qoff = concat(qs)
b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=2 * (length + 2000))
a = MicrosoftSmithWaterman(0)
a.set_workspace(int(float(os.environ.get("WS_GIB", "32")) * (1 << 30)))
if os.environ.get("MGL_COOP_W"): a.set_cooperative(int(os.environ["MGL_COOP_W"]))
b.run(a); torch.cuda.synchronize()
ref = (b.offsets.clone(), b.scores.clone(), b.cigars.clone())
# pairs k and k+32 are the same problem: check inside the pass too
def inconsistent():
    bad = []
    for k in range(32, n):
        if not (torch.equal(b.scores[k], b.scores[k % 32]) and int(b.offsets[k]) == int(b.offsets[k % 32]) and torch.equal(b.cigars[k], b.cigars[k % 32])):
            bad.append(k)
    return bad
print("pass 0 internal inconsistencies:", inconsistent()[:10], flush=True)
for p in range(1, passes):
    b.run(a); torch.cuda.synchronize()
    ds = (b.scores != ref[1]).any(dim=1); do = b.offsets != ref[0]; dc = (b.cigars != ref[2]).any(dim=1)
    if int(ds.sum() + do.sum() + dc.sum()):
        idx = torch.nonzero(ds | do | dc).flatten().tolist()
        print(f"pass {p}: {len(idx)} pairs differ: {idx[:8]} scores_differ={int(ds.sum())} offsets={int(do.sum())} cigars={int(dc.sum())}", flush=True)
        k = idx[0]
        print("   ref score", ref[1][k].tolist(), "now", b.scores[k].tolist(), "status", int(b.status[k]))
        ca = bytes(ref[2][k].cpu().numpy()).rstrip(b"\0"); cb = bytes(b.cigars[k].cpu().numpy()).rstrip(b"\0")
        m = next((i for i in range(min(len(ca), len(cb))) if ca[i] != cb[i]), -1)
        print("   cigar len", len(ca), len(cb), "first diff at", m, ca[max(0,m-20):m+20], cb[max(0,m-20):m+20])
print("done", flush=True)
