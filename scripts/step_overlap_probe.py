"""Consecutive launches of one rank's shard of BASELINE configs[2] (1.25 M pairs of 256 x 150): one context and stream -- every launch's
tail (its last tiles, most wave slots idle) before the next launch starts -- against two contexts on two streams taking the steps in turn,
the next grid's workgroups moving into the wave slots the last one's tail leaves free.  python scripts/step_overlap_probe.py [pairs] [steps]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
b0 = device_batch.window_batch(42, n, dev)
b1 = device_batch.DeviceBatch(b0.targets, b0.t_off, b0.queries, b0.q_off, b0.max_tl, b0.max_ql, b0.cigar_stride, uniform=b0.uniform)  # the same inputs, results of its own
als = [MicrosoftSmithWaterman(0), MicrosoftSmithWaterman(0)]
for a in als:
    a.set_workspace(8 << 30)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
def run(two):
    for a, b in zip(als, (b0, b1)):
        b.run(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        i = k & 1 if two else 0
        with torch.cuda.stream(streams[i]):
            (b0, b1)[i].run(als[i])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
for rep in range(2):
    for two in (False, True):
        dt = run(two)
        print(f"{n} pairs per step, {'two contexts on two streams in turn' if two else 'one context, one stream':36s}: {dt*1e3:.2f} ms per step = {n*256*150/dt/1e9:.0f} GCUPS", flush=True)
assert torch.equal(b0.scores, b1.scores) and torch.equal(b0.cigars, b1.cigars)
print("both contexts' results identical")
