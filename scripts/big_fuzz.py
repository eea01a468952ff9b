"""One-off differential run at scale: ragged random pairs (tl 1..400, ql 1..300, related / unrelated, several parameter
sets, all four strategies) through (a) the mixed int32 kernel, (b) the geometry-grouped packed kernel, (c) the
cooperative kernel, against the reference's own code (oracle/_ref, AVX2 where ql >= 8) on the host cores: offsets and
CIGARs must be identical.  python scripts/big_fuzz.py [pairs per configuration]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import oracle_lib as ol
from bench import host_cores
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(2026)
dev = torch.device("cuda", 0)
lib = ol.ref(); cores = host_cores()
PARAMS = [(200, -150, 260, 11), (25, -50, 110, 6), (10, -15, 30, 5), (3, -1, 4, 3), (1, -4, 6, 1)]
total = bad = 0
for ci, (params, strategy) in enumerate([(p, s) for p in PARAMS for s in ol.STRATEGIES][:12]):
    tl = rng.integers(1, 401, size=n); ql = rng.integers(1, 301, size=n)
    if ci % 3 == 0:   # few distinct geometries: the grouped (packed) path gets real buckets
        tl = rng.choice([64, 200, 256], size=n); ql = rng.choice([36, 101, 150], size=n)
    toff = np.zeros(n + 1, np.int64); np.cumsum(tl, out=toff[1:])
    qoff = np.zeros(n + 1, np.int64); np.cumsum(ql, out=qoff[1:])
    alpha = np.frombuffer(b"ACGT" if ci % 2 else b"ACGTN", np.uint8)
    t = alpha[rng.integers(0, len(alpha), size=int(toff[-1]))]
    q = alpha[rng.integers(0, len(alpha), size=int(qoff[-1]))]
    # two thirds of the queries are noisy copies of a piece of their target
    for k in np.nonzero(rng.random(n) < 0.66)[0]:
        m = min(tl[k], ql[k]); s0 = rng.integers(0, tl[k] - m + 1)
        seg = t[toff[k] + s0: toff[k] + s0 + m].copy()
        flips = rng.random(m) < 0.05
        seg[flips] = alpha[rng.integers(0, len(alpha), size=int(flips.sum()))]
        q[qoff[k]: qoff[k] + m] = seg
    stride = 2 * 400 + 64
    off = np.zeros(n, np.int32); cg = np.zeros(n * stride, np.uint8); ln = np.zeros(n, np.int32)
    t0 = time.perf_counter()
    rc = lib.ref_align_batch(n, t.ctypes.data, toff.ctypes.data, q.ctypes.data, qoff.ctypes.data, *params, strategy, 1, cores,
                             off.ctypes.data, cg.ctypes.data, stride, ln.ctypes.data)
    assert rc == 0
    t_cpu = time.perf_counter() - t0
    cg = cg.reshape(n, stride)
    tt, qq = torch.from_numpy(t).to(dev), torch.from_numpy(q).to(dev)
    ts, tls = torch.from_numpy(toff[:-1].copy()).to(dev), torch.from_numpy(tl.astype(np.int32)).to(dev)
    qs, qls = torch.from_numpy(qoff[:-1].copy()).to(dev), torch.from_numpy(ql.astype(np.int32)).to(dev)
    results = {}
    for name, coop in (("int32 mixed", 1), ("cooperative", 3)):
        a = MicrosoftSmithWaterman(0); a.set_cooperative(coop); a.set_workspace(8 << 30)
        b = device_batch.from_host(t, toff, q, qoff, dev, cigar_stride=stride)
        b.run(a, params, strategy); torch.cuda.synchronize()
        results[name] = (b.offsets.cpu().numpy(), b.cigars.cpu().numpy()); a.close(); del b
    a = MicrosoftSmithWaterman(0); a.set_workspace(8 << 30)
    gb = device_batch.GroupedBatch(tt, ts, tls, qq, qs, qls, cigar_stride=stride, min_bucket=8)
    gb.run(a, params, strategy); torch.cuda.synchronize()
    o, _, c, _, st = gb.gather(); results[f"grouped ({gb.n_grouped} packed-eligible slots, {gb.n_rest} mixed)"] = (o.cpu().numpy(), c.cpu().numpy())
    assert int((st != 0).sum()) == 0
    a.close()
    for name, (go, gc) in results.items():
        mism = int((go != off).sum() + (gc != cg).any(axis=1).sum())
        bad += mism; total += n
        print(f"params {params} strategy {strategy}: {name}: {n} pairs, mismatches vs reference {mism}", flush=True)
    print(f"   (reference on {cores} threads: {t_cpu:.1f} s)", flush=True)
print(f"TOTAL {total} comparisons, {bad} mismatches")
sys.exit(1 if bad else 0)
