"""Timeline of one launch of sw_dp16_lane_ck_kernel from a -DMGL_CK_TRACE build (scripts/build_variant.sh trace sw_dp16_lane_ck.hip -DMGL_CK_TRACE):
    MGL_SW_LIB=build/variants/lib_trace.so python scripts/ck_trace.py [pairs] [slots]
per tile {slot, hardware id, start, end}: tile durations by generation, how long SIMDs stood with one wave or none, the tail."""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from mgl_amd import _lib, device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
dev = torch.device("cuda", 0)
b = device_batch.window_batch(42, n, dev, window=256, read_len=150)
a = MicrosoftSmithWaterman(0)
a.set_workspace(int(os.environ.get("WS_GIB", "8")) << 30)
direct = os.environ.get("DIRECT") == "1"   # the direct form of the packed host entry (one gated launch, results into pinned host arrays)
if direct:
    from mgl_amd.smithwaterman import GATK_PARAMETERS, SWOverhangStrategy
    pb, _x = device_batch.window_batch_2bit(42, n, dev)
    del _x
    hin = [x.cpu().pin_memory().numpy() for x in (pb.target_bases, pb.t_start, pb.query_bases, pb.q_start)]
    hout = tuple(x.pin_memory().numpy() for x in (torch.zeros(n, dtype=torch.int32), torch.zeros((n, 6), dtype=torch.int32), torch.zeros(n * 64, dtype=torch.uint8), torch.zeros(n, dtype=torch.int32)))
    run = lambda: a.align_packed_2bit(hin[0], 1 << 24, hin[1], None, hin[2], n * 150, hin[3], None, 256, 150, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP, 64, out=hout)
else:
    run = lambda: b.run(a)
for rep in range(3):
    run(); torch.cuda.synchronize()
L = _lib.lib()
path = "/tmp/ck_trace.bin"
assert L.mgl_ck_trace_dump(path.encode()) == 0
for rep in range(int(os.environ.get("REPS", "8"))):   # back to back, as a timed loop runs them: the trace is the LAST launch's
    run()
torch.cuda.synchronize()
assert L.mgl_ck_trace_dump(path.encode()) == 0
tiles = (n + 127) // 128
t = np.fromfile(path, dtype=np.uint64).reshape(-1, 4)[:tiles]
slot, hw, t0, t1 = t[:, 0].astype(np.int64), t[:, 1], t[:, 2].astype(np.int64), t[:, 3].astype(np.int64)
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0   # microseconds
dur = t1 - t0
simd = ((hw >> 32) & 15).astype(np.int64) << 20 | ((hw >> 8) & 0xff).astype(np.int64) << 4 | ((hw >> 4) & 3).astype(np.int64)  # xcc | se,sh,cu | simd
print(f"{tiles} tiles, {len(np.unique(slot))} slots, {len(np.unique(simd))} SIMDs; launch {t1.max()/1e3:.2f} ms; tile duration mean {dur.mean():.0f} us, "
      f"p5 {np.percentile(dur,5):.0f}, p50 {np.percentile(dur,50):.0f}, p95 {np.percentile(dur,95):.0f}, max {dur.max():.0f}")
# the k-th tile of every slot
order = np.lexsort((t0, slot))
k = np.zeros(tiles, np.int64)
s_sorted = slot[order]
first = np.r_[True, s_sorted[1:] != s_sorted[:-1]]
idx = np.arange(tiles)
start_of_run = np.maximum.accumulate(np.where(first, idx, 0))
k[order] = idx - start_of_run
for g in range(int(k.max()) + 1):
    m = k == g
    print(f"  generation {g}: {m.sum():5d} tiles, start {t0[m].min()/1e3:6.2f} .. {t0[m].max()/1e3:6.2f} ms, end {t1[m].min()/1e3:6.2f} .. {t1[m].max()/1e3:6.2f} ms, duration mean {dur[m].mean():6.0f} us (p5 {np.percentile(dur[m],5):.0f}, p95 {np.percentile(dur[m],95):.0f})")
# time a slot spent between its tiles (the gate of the direct host form; a tile's own duration starts behind it)
o2 = np.lexsort((t0, slot))
gap = t0[o2][1:] - t1[o2][:-1]
same = slot[o2][1:] == slot[o2][:-1]
print(f"  between a slot's tiles: total {gap[same].sum()/1e3:.1f} ms over all slots = {gap[same].sum()/len(np.unique(slot))/1e3:.3f} ms per slot; first tiles start {t0[k == 0].mean()/1e3:.3f} ms after the earliest (max {t0[k == 0].max()/1e3:.3f})")
per_slot = np.bincount(slot)
print("  tiles per slot:", dict(zip(*np.unique(per_slot[per_slot > 0], return_counts=True))))
# occupancy of the SIMDs over time
ev = np.concatenate([np.stack([t0, np.ones(tiles)], 1), np.stack([t1, -np.ones(tiles)], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
occ = np.cumsum(ev[:, 1])
T = t1.max()
grid = np.linspace(0, T, 41)
pos = np.searchsorted(ev[:, 0], grid, side="right") - 1
print("  waves alive at", " ".join(f"{g/1e3:.1f}ms:{int(occ[max(p,0)])}" for g, p in zip(grid[::4], pos[::4])))
busy = (dur.sum()) / (T * 2 * len(np.unique(simd)))
print(f"  wave-slot utilisation (sum of tile durations / (launch x 2 slots x SIMDs)): {busy:.3f}")
