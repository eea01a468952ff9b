#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --memory-copy-trace run of scripts/host_entry_probe.py into a per-call timeline:
for the LAST mgl_sw_align_batch call, when each kernel and each large copy ran, how long the GPU computed, how long the
link was busy in each direction and how much of the copy time lay under a kernel."""
import csv, glob, sys
csv.field_size_limit(1 << 30)
d = sys.argv[1]
def rows(pat):
    out = []
    for f in glob.glob(f"{d}/**/*{pat}", recursive=True):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out
K = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1]) for r in rows("kernel_trace.csv")]
C = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"]) for r in rows("memory_copy_trace.csv")]
K = [k for k in K if k[2].startswith("sw_")]
K.sort(); C.sort()
# the calls are separated by gaps > 60 ms without our kernels (the Python side of the probe); take the last burst
bursts, cur = [], [K[0]]
for k in K[1:]:
    if k[0] - max(x[1] for x in cur) > 60e6:
        bursts.append(cur); cur = []
    cur.append(k)
bursts.append(cur)
B = bursts[-1]
k0, k1 = min(k[0] for k in B), max(k[1] for k in B)
Cb = [c for c in C if c[1] > k0 - 30e6 and c[0] < k1 + 30e6]
t0 = min([k0] + [c[0] for c in Cb])
t1 = max([k1] + [c[1] for c in Cb])
def union(iv):
    iv = sorted(iv); tot = 0; ce = None; cs = None; merged = []
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None: merged.append((cs, ce))
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None: merged.append((cs, ce))
    return merged
def length(m): return sum(e - s for s, e in m)
def overlap(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: tot += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
fill = union([(k[0], k[1]) for k in B if "dp" in k[2]])
tb = union([(k[0], k[1]) for k in B if "traceback" in k[2]])
anyk = union([(k[0], k[1]) for k in B])
h2d = union([(c[0], c[1]) for c in Cb if "HOST_TO_DEVICE" in c[2].upper() or c[2].upper().startswith("H2D")])
d2h = union([(c[0], c[1]) for c in Cb if "DEVICE_TO_HOST" in c[2].upper() or c[2].upper().startswith("D2H")])
ms = lambda x: x / 1e6
print(f"last call: {ms(t1 - t0):.1f} ms from the first copy to the last copy / kernel end; {len(B)} kernels, {len(Cb)} copies")
print(f"  kernels busy {ms(length(anyk)):.1f} ms (fill {ms(length(fill)):.1f}, traceback {ms(length(tb)):.1f}); idle gaps inside the call {ms((t1 - t0) - length(anyk)):.1f} ms")
print(f"  host->device busy {ms(length(h2d)):.1f} ms, of which under a kernel {ms(overlap(h2d, anyk)):.1f} ms")
print(f"  device->host busy {ms(length(d2h)):.1f} ms, of which under a kernel {ms(overlap(d2h, anyk)):.1f} ms")
print("  kernel timeline (ms from the start of the call): name start end")
names = {}
for k in B:
    print(f"    {k[2]:28s} {ms(k[0] - t0):8.2f} {ms(k[1] - t0):8.2f}")
print("  copies longer than 1 ms: direction start end")
for c in Cb:
    if c[1] - c[0] > 1e6:
        print(f"    {c[2].replace('MEMORY_COPY_', ''):18s} {ms(c[0] - t0):8.2f} {ms(c[1] - t0):8.2f}")
