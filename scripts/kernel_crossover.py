"""Where the two packed kernels cross: eight pairs per wave (sw_dp16_kernel) vs 128 per wave (sw_dp16_lane_kernel), device
resident uniform batches of growing size.  Usage: python scripts/kernel_crossover.py [tl] [ql]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

tl = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ql = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = torch.device("cuda", 0)
for n in (16384, 32768, 65536, 131072, 262144, 393216, 524288, 786432, 1048576):
    b = device_batch.window_batch(42, n, dev, window=tl, read_len=ql)
    line = f"{n:8d} pairs:"
    for mode, name in ((1, "wave8"), (2, "lane ")):
        a = MicrosoftSmithWaterman(0)
        a.set_workspace(200 << 30)
        a.set_lane_kernel(mode)
        b.run(a); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            b.run(a)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        line += f"  {name} {dt*1e3:8.3f} ms {n*tl*ql/dt/1e9:7.0f} GCUPS"
        a.close()
    print(line, flush=True)
    del b
