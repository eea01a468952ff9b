"""Variable-length reads (trimmed reads, U[lo, 150] bases) against 256-base windows: the same pairs through the int32
kernel as an ordinary mixed batch and through the packed-int16 kernel as a geometry-grouped batch
(device_batch.GroupedBatch, MGL_SW_FLAG_GROUPED_GEOMETRY).  python scripts/grouped_bench.py [pairs] [lo]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

pos = [x for x in sys.argv[1:] if not x.startswith("--")]
n = int(pos[0]) if len(pos) > 0 else 4_000_000
lo = int(pos[1]) if len(pos) > 1 else 100
dev = torch.device("cuda", 0)
b = device_batch.window_batch(42, n, dev, window=256, read_len=150)
g = torch.Generator(device=dev); g.manual_seed(1)
ql = torch.randint(lo, 151, (n,), generator=g, device=dev, dtype=torch.int32)
tl = torch.full((n,), 256, dtype=torch.int32, device=dev)
t_start, q_start = b.t_off[:-1].contiguous(), b.q_off[:-1].contiguous()
a = MicrosoftSmithWaterman(0)
if os.environ.get("WS_GIB"):   # default: the context's own (a quarter of the device's memory); 8 GiB gives 21 chunks of 190 k pairs
    a.set_workspace(int(os.environ["WS_GIB"]) << 30)
cells = int((ql.to(torch.int64) * 256).sum())

figures = {}
def timed(run, label):
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    tm = a.timing()
    figures[label.split(":")[0].split(",")[0]] = round(cells / dt / 1e9, 1)
    print(f"{label}: {dt*1e3:.1f} ms = {cells/dt/1e9:.1f} GCUPS, {n/dt/1e6:.2f} M reads/s (packed16={tm.packed16}, {tm.dp_launches} chunk(s))", flush=True)

t0 = time.perf_counter()
gb = device_batch.GroupedBatch(b.targets, t_start, tl, b.queries, q_start, ql)
torch.cuda.synchronize()
print(f"{n} pairs, read length U[{lo},150]: grouping (sort + pad on the GPU) {1e3*(time.perf_counter()-t0):.1f} ms, {gb.n - n} padding slots")
timed(lambda: gb.run(a), "grouped geometry, packed int16")
# the same pairs as an ordinary mixed batch: indexed entry without the promise -> int32 kernel
import ctypes as C
from mgl_amd import _lib
mixed = device_batch.GroupedBatch.__new__(device_batch.GroupedBatch)
mixed.__dict__.update(gb.__dict__)
def run_mixed():
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().mgl_sw_align_batch_device_indexed(a.ctx, C.c_void_p(st.cuda_stream), gb.n, gb.targets.data_ptr(), gb.t_off.data_ptr(),
        gb.t_len.data_ptr(), gb.queries.data_ptr(), gb.q_off.data_ptr(), gb.q_len.data_ptr(), gb.max_tl, gb.max_ql, 200, -150, 260, 11, 1,
        gb.offsets.data_ptr(), gb.scores.data_ptr(), gb.cigars.data_ptr(), gb.cigar_stride, gb.cigar_len.data_ptr(), gb.status.data_ptr(), 0)
    assert rc == 0
ref = (gb.offsets.clone(), gb.scores.clone(), gb.cigars.clone())
timed(run_mixed, "same slots, no promise (sorted on the device by the library; MGL_SW_DEBUG_AUTO_GROUP=0: int32 kernel)")
assert torch.equal(ref[0], gb.offsets) and torch.equal(ref[1], gb.scores) and torch.equal(ref[2], gb.cigars)
print("identical results")
# the pairs in their original (unsorted) order, no promise: mgl_sw_align_batch_device_indexed sorts every chunk on the device
u_off = torch.zeros(n, dtype=torch.int32, device=dev); u_sc = torch.zeros((n, 6), dtype=torch.int32, device=dev)
u_cg = torch.zeros((n, 64), dtype=torch.uint8, device=dev); u_ln = torch.zeros(n, dtype=torch.int32, device=dev); u_st = torch.zeros(n, dtype=torch.int32, device=dev)
def run_unsorted():
    st = torch.cuda.current_stream(dev)
    rc = _lib.lib().mgl_sw_align_batch_device_indexed(a.ctx, C.c_void_p(st.cuda_stream), n, b.targets.data_ptr(), t_start.data_ptr(),
        tl.data_ptr(), b.queries.data_ptr(), q_start.data_ptr(), ql.data_ptr(), 256, 150, 200, -150, 260, 11, 1,
        u_off.data_ptr(), u_sc.data_ptr(), u_cg.data_ptr(), 64, u_ln.data_ptr(), u_st.data_ptr(), 0)
    assert rc == 0
timed(run_unsorted, "original order, no promise (device-resident, sorted by the library)")
g0 = gb.gather()
assert torch.equal(g0[0], u_off) and torch.equal(g0[1], u_sc)
print("identical to the grouped batch, pair for pair")

# the same reads as a HOST batch of mixed lengths with no flag and no sorting by the caller: mgl_sw_align_batch sorts every
# chunk by geometry itself (PCIe inclusive: pageable host memory in, every result back on the host)
import numpy as np
ql_h = ql.cpu().numpy().astype(np.int64)
reads = b.queries.view(n, 150).cpu().numpy()
keep = np.arange(150)[None, :] < ql_h[:, None]
qd = reads[keep]                                   # ragged reads, concatenated
qoff = np.concatenate([[0], np.cumsum(ql_h)]).astype(np.int64)
td = b.targets.cpu().numpy()
toff = b.t_off.cpu().numpy()
off = np.zeros(n, np.int32); sc = np.zeros((n, 6), np.int32); cg = np.zeros(n * 64, np.uint8); ln = np.zeros(n, np.int32)
L = _lib.lib()
for env, label in (("1", "host batch of mixed lengths, sorted per chunk by the library"), ("0", "same, MGL_SW_DEBUG_AUTO_GROUP=0 (int32 kernel)")):
    if env == "0":
        break  # (the switch is read once per process: run the script again with MGL_SW_DEBUG_AUTO_GROUP=0 for the other line)
    for rep in range(3):
        t0 = time.perf_counter()
        rc = L.mgl_sw_align_batch(a.ctx, n, td.ctypes.data, toff.ctypes.data, qd.ctypes.data, qoff.ctypes.data, 200, -150, 260, 11, 1,
                                  off.ctypes.data, sc.ctypes.data, cg.ctypes.data, 64, ln.ctypes.data)
        dt = time.perf_counter() - t0
        assert rc == 0
    print(f"{label} (MGL_SW_DEBUG_AUTO_GROUP={os.environ.get('MGL_SW_DEBUG_AUTO_GROUP', '1')}): {dt*1e3:.1f} ms = {cells/dt/1e9:.1f} GCUPS, "
          f"{n/dt/1e6:.2f} M reads/s (packed16={a.timing().packed16})", flush=True)
# same answers as the device-resident grouped run (caller order)
g_off, g_sc = gb.gather()[0].cpu().numpy(), gb.gather()[1].cpu().numpy()
assert (g_off == off).all() and (g_sc == sc).all()
print("host batch identical to the grouped device batch")
# ... and as the packed entry takes them (round 5): 2-bit packed bases, windows into one packed genome, every array page-locked by the caller,
# mgl_sw_align_batch_2bit WITHOUT the uniform flag -- the chunks sorted on the device, inputs in and every result back inside the wall time
host2 = None
try:
    pb, _twin = device_batch.window_batch_2bit(42, n, dev, window=256, read_len=150)
    del _twin
    pin = lambda x: torch.from_numpy(np.ascontiguousarray(x)).pin_memory().numpy()
    G2, Q2 = pin(pb.target_bases.cpu().numpy()), pin(pb.query_bases.cpu().numpy())
    ts2, qs2 = pin(pb.t_start.cpu().numpy()), pin(pb.q_start.cpu().numpy())
    tl2, ql2 = pin(np.full(n, 256, np.int32)), pin(ql.cpu().numpy().astype(np.int32))
    out2 = (pin(np.zeros(n, np.int32)), pin(np.zeros((n, 6), np.int32)), pin(np.zeros(n * 64, np.uint8)), pin(np.zeros(n, np.int32)))
    from mgl_amd.smithwaterman import GATK_PARAMETERS, SWOverhangStrategy
    call2 = lambda: a.align_packed_2bit(G2, int(G2.size) * 4, ts2, tl2, Q2, n * 150, qs2, ql2, 256, 150, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP, 64, out=out2)
    call2()
    t2 = []
    for _ in range(3):
        t0 = time.perf_counter(); call2(); t2.append(time.perf_counter() - t0)
    dt2 = sum(t2) / len(t2)
    same2 = bool((out2[0] == g_off).all() and (out2[1] == g_sc).all())
    host2 = {"gcups": round(cells / dt2 / 1e9, 1), "ms_per_call": round(dt2 * 1e3, 2), "fill_launches": int(a.timing().dp_launches), "identical_to_device_resident": same2}
    print(f"host batch of mixed lengths, 2-bit packed, page-locked arrays, sorted on the device: {dt2*1e3:.1f} ms = {cells/dt2/1e9:.1f} GCUPS ({host2['fill_launches']} chunks; identical: {same2})", flush=True)
    assert same2
except Exception as e:  # noqa: BLE001
    host2 = {"error": repr(e)[:200]}
if "--json" in sys.argv:
    import json
    print(json.dumps({"gcups": figures.get("original order"), "pairs": n, "read_lengths": [lo, 150], "window": 256,
                      "device_resident_no_promise_gcups": figures.get("original order"),
                      "device_resident_sorted_by_caller_with_promise_gcups": figures.get("grouped geometry"),
                      "host_buffers_pcie_inclusive_gcups": (host2 or {}).get("gcups"),
                      "host_buffers_pcie_inclusive": dict(host2 or {}, form="mgl_sw_align_batch_2bit without the uniform flag: 2-bit packed bases, page-locked arrays, chunks sorted on the device"),
                      "host_ascii_pageable_gcups": round(cells / dt / 1e9, 1),
                      "note": "mixed geometries: the library sorts every chunk by (tl, ql) itself -- counting sort on the GPU for device-resident "
                              "batches, on the host for host buffers -- whole waves of one geometry through the lane kernel, full blocks of eight through the packed kernel, the rest through int32"}))
