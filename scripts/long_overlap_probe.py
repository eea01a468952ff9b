"""Long reads, pass after pass: one context and stream (fill, walk, fill, walk ...) against two contexts on two streams taking the passes in
turn (the walk of one pass -- one wave per pair -- beside the fill of the next).  python scripts/long_overlap_probe.py [pairs] [length] [passes]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch, synth
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, concat

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
length = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rng = synth.rng_for(11)
base = [synth.ont_pair(rng, length) for _ in range(32)]
ts = [base[k % 32][0].tobytes() for k in range(n)]
qs = [base[k % 32][1].tobytes() for k in range(n)]
td, toff = concat(ts); qd, qoff = concat(qs)
stride = 2 * (length + 2000)
b0 = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=stride)
b1 = device_batch.DeviceBatch(b0.targets, b0.t_off, b0.queries, b0.q_off, b0.max_tl, b0.max_ql, b0.cigar_stride)
als = [MicrosoftSmithWaterman(0), MicrosoftSmithWaterman(0)]
for a in als:
    a.set_workspace(110 << 30)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
cells = b0.cells
torch.cuda.synchronize()
def run(two):
    for a, b in zip(als, (b0, b1)):
        b.run(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(passes):
        i = k & 1 if two else 0
        with torch.cuda.stream(streams[i]):
            (b0, b1)[i].run(als[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes
for rep in range(2):
    for two in (False, True):
        dt = run(two)
        print(f"{n} pairs of {length}, {'two contexts on two streams in turn' if two else 'one context, one stream':36s}: {dt*1e3:.1f} ms per pass = {cells/dt/1e9:.0f} GCUPS", flush=True)
assert torch.equal(b0.scores, b1.scores) and torch.equal(b0.cigars, b1.cigars) and int((b0.status != 0).sum()) == 0
print("both contexts' results identical")
