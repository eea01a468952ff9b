"""Where does the direct host form's time go?  The bench batch device resident with its RESULT arrays in pinned host memory (the waves
write them over the link in whole lines out of LDS) against the same with results in device memory.  python scripts/zero_copy_out_probe.py [pairs]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
a = MicrosoftSmithWaterman(0)
a.set_workspace(8 << 30)
b, _ascii = device_batch.window_batch_2bit(42, n, dev)
del _ascii
def timed(label):
    b.run(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        b.run(a)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{label}: {dt*1e3:.2f} ms per call = {n*256*150/dt/1e9:.0f} GCUPS", flush=True)
timed("results in device memory       ")
keep = (b.offsets, b.scores, b.cigars, b.cigar_len, b.status)
for which in ("cigars", "all"):
    if which == "cigars":
        b.cigars = torch.empty((n, 64), dtype=torch.uint8, pin_memory=True)
    else:
        b.offsets = torch.empty(n, dtype=torch.int32, pin_memory=True)
        b.scores = torch.empty((n, 6), dtype=torch.int32, pin_memory=True)
        b.cigar_len = torch.empty(n, dtype=torch.int32, pin_memory=True)
        b.status = torch.empty(n, dtype=torch.int32, pin_memory=True)
    timed(f"results in pinned host memory ({which:6s})")
