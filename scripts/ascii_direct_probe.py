"""The ASCII host entry's direct form on the bench workload (10 M pairs 256 x 150, arrays registered by the caller), PCIe inclusive, for a few
chunk sizes (MGL_SW_DEBUG_DIRECT_CHUNK) -- with MGL_SW_DEBUG_HOST_TIMING=1 the library says when the inputs had landed and when the grid ended.
python scripts/ascii_direct_probe.py [pairs] [chunk ...]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from mgl_amd import _lib, device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, GATK_PARAMETERS, SWOverhangStrategy
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
chunks = [int(x) for x in sys.argv[2:]] or [131072]
dev = torch.device("cuda", 0)
b = device_batch.window_batch(42, n, dev)
a = MicrosoftSmithWaterman(0)
a.set_workspace(12 << 30)
b.run(a); torch.cuda.synchronize()
want = (b.offsets.cpu().numpy(), b.cigars.cpu().numpy().reshape(-1))
pin = lambda x: torch.from_numpy(np.ascontiguousarray(x)).pin_memory().numpy()
t, q = pin(b.targets.cpu().numpy()), pin(b.queries.cpu().numpy())
toff, qoff = b.t_off.cpu().numpy(), b.q_off.cpu().numpy()
off, sc, cg, ln = pin(np.zeros(n, np.int32)), pin(np.zeros((n, 6), np.int32)), pin(np.zeros(n * 64, np.uint8)), pin(np.zeros(n, np.int32))
L = _lib.lib()
m, x, o, e = GATK_PARAMETERS
def call():
    rc = L.mgl_sw_align_batch(a.ctx, n, t.ctypes.data, toff.ctypes.data, q.ctypes.data, qoff.ctypes.data, m, x, o, e, int(SWOverhangStrategy.SOFTCLIP), off.ctypes.data,
                              sc.ctypes.data, cg.ctypes.data, 64, ln.ctypes.data)
    assert rc == 0, rc
for c in chunks:
    os.environ["MGL_SW_DEBUG_DIRECT_CHUNK"] = str(c)
    call()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
    ok = bool((off == want[0]).all() and (cg == want[1]).all())
    print(f"chunk {c}: {[round(v*1e3,2) for v in ts]} ms -> best {min(ts)*1e3:.2f} ms = {n*256*150/min(ts)/1e9:.0f} GCUPS, {a.timing().dp_launches} launch(es), identical to device resident: {ok}", flush=True)
