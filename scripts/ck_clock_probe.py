"""The shader clock the chip holds under sw_dp16_lane_ck_kernel (profiling level 2: every wave stamps s_memtime and the 100 MHz s_memrealtime
around its life) and the launch's duration, for the library MGL_SW_LIB names.  python scripts/ck_clock_probe.py [pairs]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
b = device_batch.window_batch(42, n, dev)
a = MicrosoftSmithWaterman(0)
a.set_workspace(8 << 30)
for _ in range(2):
    b.run(a); torch.cuda.synchronize()
for rep in range(3):
    a.set_profiling(2)
    b.run(a); torch.cuda.synchronize()
    tm = a.timing()
    print(f"{os.environ.get('MGL_SW_LIB', 'shipped build')}: {n} pairs: fill {tm.dp_ms:.2f} ms, clock {tm.clock_mhz} MHz", flush=True)
