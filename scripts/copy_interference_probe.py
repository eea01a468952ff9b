"""Does host -> device DMA traffic beside it slow the headline kernel?  Device-resident 2-bit batch, timed alone and with 0.54 GB of pinned
host memory copied in (2 MB pieces, a second stream) during every call.  python scripts/copy_interference_probe.py"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

n = 10_000_000
dev = torch.device("cuda", 0)
a = MicrosoftSmithWaterman(0)
a.set_workspace(8 << 30)
pb, _ = device_batch.window_batch_2bit(42, n, dev)
del _
src = torch.empty(540 << 20, dtype=torch.uint8).pin_memory()
dst = torch.empty(540 << 20, dtype=torch.uint8, device=dev)
out_h = torch.empty(960 << 20, dtype=torch.uint8).pin_memory()
out_d = torch.empty(960 << 20, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
def timed(label, copies, reps=6):
    pb.run(a); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if copies:
            with torch.cuda.stream(side):
                step = 2 << 20
                if copies == "h2d":
                    for o in range(0, src.numel(), step):
                        dst[o:o + step].copy_(src[o:o + step], non_blocking=True)
                else:
                    for o in range(0, out_h.numel(), step):
                        out_h[o:o + step].copy_(out_d[o:o + step], non_blocking=True)
        pb.run(a)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label}: {dt*1e3:.2f} ms per call = {n*256*150/dt/1e9:.0f} GCUPS", flush=True)
for _ in range(2):
    timed("alone", None)
    timed("with 0.54 GB host -> device beside it", "h2d")
    timed("with 0.96 GB device -> host beside it", "d2h")
