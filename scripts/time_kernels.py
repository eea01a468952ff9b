"""Time sw_dp_kernel / sw_traceback_kernel on the bench workload (HIP events inside the library)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

name = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
tl = int(os.environ.get("TL", 256)); ql = int(os.environ.get("QL", 150))
dev = torch.device("cuda", 0)
a = MicrosoftSmithWaterman(0)
a.set_workspace(8 << 30)
b = device_batch.window_batch(42, pairs, dev, window=tl, read_len=ql)
b.run(a); torch.cuda.synchronize()
a.set_profiling(2)
best = None
for _ in range(3):
    b.run(a); torch.cuda.synchronize()
    t = a.timing()
    if best is None or t.dp_ms < best[0]:
        best = (t.dp_ms, t.tb_ms, t.clock_mhz, t.packed16)
cells = pairs * tl * ql
print(f"{name:24s} dp_ms={best[0]:9.3f} tb_ms={best[1]:8.3f} dp_gcups={cells / best[0] / 1e6:9.1f} clock_mhz={best[2]} packed16={best[3]}", flush=True)
