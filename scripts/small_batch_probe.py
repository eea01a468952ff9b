"""Small batches: the one-wave-per-pair kernel that fills, walks and writes the text in one launch (sw_small_kernel) against the
kernels a batch of that size takes otherwise, device-resident batches, wall time per call including the synchronisation.
`uniform` = the caller promises one geometry (the packed kernel is then available to the general path).
Usage: python scripts/small_batch_probe.py [tl] [ql]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

tl = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ql = int(sys.argv[2]) if len(sys.argv) > 2 else 150
dev = torch.device("cuda", 0)
for n in (1, 16, 64, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
    b = device_batch.window_batch(42, n, dev, window=tl, read_len=ql)
    line = f"{n:6d} pairs:"
    for uniform in (True, False):
        b.uniform = uniform
        for mode, name in ((2, "small"), (1, "other")):
            a = MicrosoftSmithWaterman(0)
            a.set_small_kernel(mode)
            b.run(a); torch.cuda.synchronize()
            reps = 20 if n <= 4096 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                b.run(a)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            line += f"  {'uni' if uniform else 'mix'} {name} k{a.timing().fill_kernel} {dt*1e6:8.1f} us {n*tl*ql/dt/1e9:6.0f} GCUPS"
            a.close()
    print(line, flush=True)
    del b
