"""Differential run over long pairs with the library's own kernel choice (strip kernel, 16-bit workgroup kernel, int32 workgroup
kernel, by geometry and parameters): batches of random pairs -- related (ONT-style noise, with a long deletion or insertion now and
then) or unrelated, lengths drawn per batch from a different range -- against the reference's own code (oracle/_ref, AVX2 path) on
the host cores: offsets and CIGAR bytes must be identical.  python scripts/long_fuzz.py [batches] [pairs per batch] [passes]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import oracle_lib as ol
from bench import host_cores
from mgl_amd import device_batch, synth
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, concat, _lib

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 18
per = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rng = synth.rng_for(31337)
lib = ol.ref(); cores = host_cores()
PARAMS = [(200, -150, 260, 11), (100, -100, 300, 10), (25, -50, 110, 6), (400, -300, 500, 20)]
RANGES = [(1500, 4095), (4096, 4400), (4096, 8192), (8000, 12288), (9500, 10500), (12289, 16384), (3000, 16500), (600, 2500), (15000, 20000)]
if len(sys.argv) > 3 and sys.argv[3] == "passes":  # targets beyond 16 384 rows: several passes of the strip kernel (round 4)
    RANGES = [(16385, 17000), (17000, 24000), (24000, 33000), (32700, 40000)]
a = MicrosoftSmithWaterman(0)
a.set_workspace(64 << 30)
total = bad = 0
kernels = {}
for bi in range(nb):
    lo, hi = RANGES[bi % len(RANGES)]
    params = PARAMS[bi % len(PARAMS)]
    strategy = ol.STRATEGIES[(bi // 2) % 4]
    ts, qs = [], []
    for k in range(per):
        n = int(rng.integers(lo, hi + 1))
        t, q = synth.ont_pair(rng, n)
        r = rng.random()
        if r < 0.2:                       # unrelated, different length
            q = synth.random_genome(rng, int(rng.integers(max(50, lo // 3), hi + 1)))
        elif r < 0.4:                     # a long deletion
            cut = int(rng.integers(0, max(1, len(q) - 600))); q = np.concatenate([q[:cut], q[cut + int(rng.integers(100, 600)):]])
        elif r < 0.5:                     # a long insertion
            cut = int(rng.integers(0, len(q))); q = np.concatenate([q[:cut], synth.random_genome(rng, int(rng.integers(100, 500))), q[cut:]])
        ts.append(t.tobytes()); qs.append(q.tobytes())
    td, toff = concat(ts); qd, qoff = concat(qs)
    stride = 2 * (max(max(map(len, ts)), max(map(len, qs))) + 64)
    off = np.zeros(per, np.int32); cg = np.zeros(per * stride, np.uint8); ln = np.zeros(per, np.int32)
    t0 = time.perf_counter()
    rc = lib.ref_align_batch(per, td.ctypes.data, toff.ctypes.data, qd.ctypes.data, qoff.ctypes.data, *params, strategy, 1, cores,
                             off.ctypes.data, cg.ctypes.data, stride, ln.ctypes.data)
    assert rc == 0
    t_cpu = time.perf_counter() - t0
    b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=stride)
    b.run(a, params, strategy); torch.cuda.synchronize()
    name = _lib.FILL_KERNEL_NAMES[a.timing().fill_kernel]
    kernels[name] = kernels.get(name, 0) + per
    go, gc = b.offsets.cpu().numpy(), b.cigars.cpu().numpy()
    mism = int((go != off).sum() + (gc != cg.reshape(per, stride)).any(axis=1).sum()) + int((b.status != 0).sum())
    bad += mism; total += per
    print(f"batch {bi}: {per} pairs, lengths {lo}..{hi}, params {params}, strategy {strategy}: {name}, mismatches vs reference {mism} "
          f"(reference on {cores} threads: {t_cpu:.1f} s)", flush=True)
print(f"TOTAL {total} pairs, {bad} mismatches; pairs per fill kernel: {kernels}")
sys.exit(1 if bad else 0)
