"""One launch size of the one-wave-per-pair kernel (sw_small_kernel), a few launches: what scripts/prof_r04.sh profiles.
Usage: python scripts/small_kernel_probe.py [pairs] [tl] [ql]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ql = int(sys.argv[3]) if len(sys.argv) > 3 else 150
b = device_batch.window_batch(42, n, torch.device("cuda", 0), window=tl, read_len=ql)
a = MicrosoftSmithWaterman(0)
a.set_small_kernel(2)
b.run(a); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    b.run(a)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
assert a.timing().fill_kernel == 8
print(f"{n} pairs {tl}x{ql} through sw_small_kernel: {dt*1e6:.1f} us per launch = {n*tl*ql/dt/1e9:.0f} GCUPS")
