"""The direct form of mgl_sw_align_batch_2bit on the bench batch: arrays registered by the library (numpy) against arrays the caller pinned
itself (torch pin_memory = hipHostMalloc); MGL_SW_DEBUG_HOST_TIMING=1 prints when the first / last inputs landed and when the grid ended.
python scripts/direct_probe.py [pairs]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, GATK_PARAMETERS, SWOverhangStrategy

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
a = MicrosoftSmithWaterman(0)
a.set_workspace(12 << 30)
pb, _ascii = device_batch.window_batch_2bit(42, n, dev)
del _ascii
tl, ql, stride = 256, 150, 64
host = [pb.target_bases.cpu(), pb.t_start.cpu(), pb.query_bases.cpu(), pb.q_start.cpu()]
def run(label, arrays, outs, reps=4):
    G, win, Q, qst = arrays
    call = lambda: a.align_packed_2bit(G, 1 << 24, win, None, Q, n * ql, qst, None, tl, ql, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP, stride, out=outs)
    call()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    dt = (time.perf_counter() - t0) / reps
    a.set_profiling(1)
    call()
    tm = a.timing()
    a.set_profiling(0)
    print(f"{label}: {dt*1e3:.2f} ms per call = {n*tl*ql/dt/1e9:.0f} GCUPS ({tm.dp_launches} launch(es), the last one's kernel(s) {tm.dp_ms:.2f} ms by HIP events)", flush=True)
# (1) numpy arrays registered through the library
np_in = [x.numpy().copy() for x in host]
np_out = (np.zeros(n, np.int32), np.zeros((n, 6), np.int32), np.zeros(n * stride, np.uint8), np.zeros(n, np.int32))
for x in np_in + list(np_out):
    a.register_host_buffer(x)
run("numpy arrays, mgl_sw_register_host_buffer", np_in, np_out)
ref = [x.copy() for x in np_out]
for x in np_in + list(np_out):
    a.unregister_host_buffer(x)
# (2) arrays the caller pinned itself
pin_in = [x.pin_memory().numpy() for x in host]
pin_out_t = (torch.zeros(n, dtype=torch.int32).pin_memory(), torch.zeros((n, 6), dtype=torch.int32).pin_memory(),
             torch.zeros(n * stride, dtype=torch.uint8).pin_memory(), torch.zeros(n, dtype=torch.int32).pin_memory())
pin_out = tuple(x.numpy() for x in pin_out_t)
run("arrays pinned by the caller (hipHostMalloc)", pin_in, pin_out)
assert all((x == y).all() for x, y in zip(ref, pin_out)), "the two runs differ"
os.environ["MGL_SW_DEBUG_HOST_DIRECT"] = "0"
run("the same through the chunked form          ", pin_in, pin_out)
print("identical results")
