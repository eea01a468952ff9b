"""Where the host-buffer entry (mgl_sw_align_batch) spends its time on a uniform batch: PCIe inclusive, pageable memory.
Usage: MGL_SW_DEBUG_HOST_TIMING=1 python scripts/host_entry_probe.py [pairs] [lane_mode 0|1|2] [workspace GiB]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from mgl_amd import _lib, device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ws = float(sys.argv[3]) if len(sys.argv) > 3 else 0
b = device_batch.window_batch(42, n, torch.device("cuda", 0))
t, q, toff, qoff = b.targets.cpu().numpy(), b.queries.cpu().numpy(), b.t_off.cpu().numpy(), b.q_off.cpu().numpy()
del b
a = MicrosoftSmithWaterman(0)
if ws:
    a.set_workspace(int(ws * (1 << 30)))
a.set_lane_kernel(mode)
off, sc, cg, ln = np.zeros(n, np.int32), np.zeros((n, 6), np.int32), np.zeros(n * 64, np.uint8), np.zeros(n, np.int32)
L = _lib.lib()
for rep in range(3):
    t0 = time.perf_counter()
    rc = L.mgl_sw_align_batch(a.ctx, n, t.ctypes.data, toff.ctypes.data, q.ctypes.data, qoff.ctypes.data, 200, -150, 260, 11, 1,
                              off.ctypes.data, sc.ctypes.data, cg.ctypes.data, 64, ln.ctypes.data)
    dt = time.perf_counter() - t0
    assert rc == 0
    tm = a.timing()
    print(f"lane_mode {mode} ws {ws}: {dt*1e3:.1f} ms = {n*256*150/dt/1e9:.0f} GCUPS; {a.fill_kernel_name(tm)} x {tm.dp_launches}", flush=True)
