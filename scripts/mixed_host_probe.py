"""Mixed read lengths FROM HOST MEMORY through the 2-bit packed host entry (mgl_sw_align_batch_2bit without the uniform flag): 4 M reads of
100-150 bases against 256-base windows into one packed genome, every array page-locked by the caller, PCIe inclusive (inputs in, every
result back in the caller's arrays), against the same pairs device resident.  python scripts/mixed_host_probe.py [pairs] [lo] [--json]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, GATK_PARAMETERS, SWOverhangStrategy

pos = [x for x in sys.argv[1:] if not x.startswith("--")]
n = int(pos[0]) if len(pos) > 0 else 4_000_000
lo = int(pos[1]) if len(pos) > 1 else 100
dev = torch.device("cuda", 0)
pb, ascii_twin = device_batch.window_batch_2bit(42, n, dev, window=256, read_len=150)
g = torch.Generator(device=dev); g.manual_seed(1)
ql = torch.randint(lo, 151, (n,), generator=g, device=dev, dtype=torch.int32)
tl = torch.full((n,), 256, dtype=torch.int32, device=dev)
cells = int((ql.to(torch.int64) * 256).sum())
a = MicrosoftSmithWaterman(0)
a.set_workspace(int(os.environ.get("WS_GIB", "12")) << 30)
pin = lambda x: torch.from_numpy(np.ascontiguousarray(x)).pin_memory().numpy()
G, Q = pin(pb.target_bases.cpu().numpy()), pin(pb.query_bases.cpu().numpy())
ts, qs = pin(pb.t_start.cpu().numpy()), pin(pb.q_start.cpu().numpy())
tlh, qlh = pin(tl.cpu().numpy()), pin(ql.cpu().numpy())
out = (pin(np.zeros(n, np.int32)), pin(np.zeros((n, 6), np.int32)), pin(np.zeros(n * 64, np.uint8)), pin(np.zeros(n, np.int32)))
nbases_t, nbases_q = int(G.size) * 4, n * 150

def host_call():
    a.align_packed_2bit(G, nbases_t, ts, tlh, Q, nbases_q, qs, qlh, 256, 150, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP, 64, out=out)

host_call()
times = []
for _ in range(3):
    t0 = time.perf_counter(); host_call(); times.append(time.perf_counter() - t0)
dt = min(times)
tm = a.timing()
print(f"host, 2-bit packed, pinned, mixed lengths U[{lo},150]: {dt*1e3:.1f} ms (best of 3; {[round(x*1e3,1) for x in times]}) = {cells/dt/1e9:.1f} GCUPS, {tm.dp_launches} fill launches", flush=True)
# device resident, same pairs (no promise): the reference for "what the link costs"
dp = device_batch.PackedBatch(pb.target_bases, pb.t_start, tl, pb.query_bases, pb.q_start, ql, 256, 150, 64)
dp.run(a); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): dp.run(a)
torch.cuda.synchronize()
dd = (time.perf_counter() - t0) / 3
print(f"device resident, same pairs: {dd*1e3:.1f} ms = {cells/dd/1e9:.1f} GCUPS", flush=True)
ok = bool((dp.offsets.cpu().numpy() == out[0]).all() and (dp.scores.cpu().numpy() == out[1]).all() and (dp.cigars.cpu().numpy().reshape(-1) == out[2]).all())
print("host results identical to the device-resident run" if ok else "MISMATCH between host and device-resident results", flush=True)
if "--json" in sys.argv:
    import json
    print(json.dumps({"host_2bit_pinned_gcups": round(cells / dt / 1e9, 1), "host_ms": round(dt * 1e3, 2), "device_resident_gcups": round(cells / dd / 1e9, 1),
                      "fill_launches": int(tm.dp_launches), "identical": ok}))
