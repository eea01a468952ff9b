"""PairHMM throughput on synthetic GATK-like regions, device-resident inputs:
`regions` active regions, each with `reads` Illumina-style reads (read_len bases, per-base qualities) against
`haps` haplotypes of about hap_len bases (variants of one local reference); every read of a region against every
haplotype of that region.  Metric: DP cells per second (read_len * hap_len per pair) and pairs per second; the CPU
baseline is the scalar restatement (oracle/, pinned by the reference's known answers) on all host cores.

  python scripts/pairhmm_bench.py [--regions 2000] [--reads 100] [--haps 8] [--read-len 150] [--hap-len 300]
"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
from mgl_amd import pairhmm

ap = argparse.ArgumentParser()
ap.add_argument("--regions", type=int, default=2000)
ap.add_argument("--reads", type=int, default=100)
ap.add_argument("--haps", type=int, default=8)
ap.add_argument("--read-len", type=int, default=150)
ap.add_argument("--hap-len", type=int, default=300)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--double", action="store_true")
ap.add_argument("--cpu-seconds", type=float, default=10.0)
ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--json", action="store_true", help="print one JSON line with the figures and a roofline object (bench.py reads it)")
args = ap.parse_args()

rng = np.random.default_rng(42)
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
G, NR, NH, RL, HL = args.regions, args.reads, args.haps, args.read_len, args.hap_len
# haplotypes: per region one local reference, each haplotype = that reference with a few SNVs and a +-20 length change
ref = BASES[rng.integers(0, 4, size=(G, HL + 40))]
hap_len = rng.integers(HL - 20, HL + 21, size=(G, NH))
hap_rows = []
for g in range(G):
    for h in range(NH):
        x = ref[g, : hap_len[g, h]].copy()
        snv = rng.integers(0, len(x), size=3)
        x[snv] = BASES[rng.integers(0, 4, size=3)]
        hap_rows.append(x)
haps = np.concatenate(hap_rows)
hap_off = np.zeros(G * NH + 1, dtype=np.int64); np.cumsum([len(x) for x in hap_rows], out=hap_off[1:])
# reads: copies of the region's reference with 1 % substitutions, qualities Q6..Q41, indel GOP 45, GCP 10
starts = rng.integers(0, HL - 20 - RL, size=(G, NR))
idx = starts[:, :, None] + np.arange(RL)[None, None, :]
bases = np.take_along_axis(ref[:, None, :].repeat(NR, axis=1), idx, axis=2)
sub = rng.random(bases.shape) < 0.01
bases = np.where(sub, BASES[rng.integers(0, 4, size=bases.shape)], bases).astype(np.uint8)
qual = rng.integers(6, 42, size=bases.shape, dtype=np.uint8)
gop = np.full(bases.shape, 45, np.uint8); gcp = np.full(bases.shape, 10, np.uint8)
reads = np.stack([bases, qual, gop, gop, gcp], axis=2).reshape(-1)   # per read: bases|qual|ins|del|gcp
read_off = np.arange(G * NR + 1, dtype=np.int64) * RL
pr = np.repeat(np.arange(G * NR, dtype=np.int32), NH)
ph = (np.repeat(np.arange(G, dtype=np.int32), NR * NH) * NH + np.tile(np.arange(NH, dtype=np.int32), G * NR))
n_pairs = len(pr)
cells = int((RL * (hap_off[ph + 1] - hap_off[ph])).sum())

dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(a).to(dev)
d = [t(reads), t(read_off), t(haps), t(hap_off), t(pr), t(ph)]
out = torch.zeros(n_pairs, dtype=torch.float64, device=dev)
used = torch.zeros(n_pairs, dtype=torch.int32, device=dev)
hmm = pairhmm.MicrosoftPairHmm(0)
assert hmm.load()
hmm.initialize(pairhmm.PairHMMNativeArguments(args.double, 1))
run = lambda: hmm.compute_pairs_device(*d, RL, int(hap_len.max()), out, used)
run(); torch.cuda.synchronize()
hmm.set_profiling(1)
t0 = time.perf_counter()
for _ in range(args.steps):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
tm = hmm.timing()
print(f"PairHMM {'double' if args.double else 'float + double rescue'}: {n_pairs} pairs ({G} regions x {NR} reads x {NH} haplotypes, "
      f"{RL} x ~{HL}), {dt*1e3:.2f} ms per pass = {cells/dt/1e9:.1f} GCUPS, {n_pairs/dt/1e6:.2f} M pairs/s "
      f"(float kernel {tm.float_ms:.2f} ms, double kernel {tm.double_ms:.2f} ms, rescued {int(used.sum())})", flush=True)
if args.json:
    import json
    # fp32 vector roofline: 13 flops per cell of the recurrence (compute_prob_scalar.cc:39-43: M 5, X 3, Y 3, sum 2) against
    # the 157.3 TFLOP/s fp32 VALU peak (MI355X_MICROARCH.md); HBM bytes per pair for comparison: the read's five tracks and
    # the haplotype (both shared by many pairs: an upper bound), two int32 indices, the float64 result, the int32 flag
    k_s = (tm.float_ms + tm.double_ms) / 1e3
    hbm = int((5 * RL + (hap_off[ph + 1] - hap_off[ph]) + 8 + 8 + 4).sum())
    print(json.dumps({"gcups": round(cells / dt / 1e9, 1), "pairs": int(n_pairs), "pairs_per_s": round(n_pairs / dt, 1), "ms_per_pass": round(dt * 1e3, 3),
                      "kernel_ms": {"float": round(tm.float_ms, 3), "double_rescue": round(tm.double_ms, 3)},
                      "roofline": {"bound": "valu_fp32", "achieved": round(13 * cells / k_s / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": round(13 * cells / k_s / 1e12 / 157.3, 4), "flops_per_cell": 13,
                                   "hbm": {"algorithmic_bytes_per_pass_upper_bound": hbm, "achieved_gb_s": round(hbm / k_s / 1e9, 1), "frac_of_8_tb_s": round(hbm / k_s / 8e12, 4)},
                                   "note": "shuffle- and issue-bound wavefront (14.3 VALU instructions per cell, no MFMA shape); inputs resident"}}),
          flush=True)
if not args.no_cpu:
    import pairhmm_oracle_lib as pol
    from bench import host_cores
    cores = host_cores()
    m = min(n_pairs, 20000)
    t0 = time.perf_counter(); want, _ = pol.compute_pairs(reads, read_off, haps, hap_off, pr[:m], ph[:m], args.double, cores); dtc = time.perf_counter() - t0
    m = int(min(n_pairs, max(m, m * args.cpu_seconds / dtc)))
    t0 = time.perf_counter(); want, wused = pol.compute_pairs(reads, read_off, haps, hap_off, pr[:m], ph[:m], args.double, cores); dtc = time.perf_counter() - t0
    ccells = int((RL * (hap_off[ph[:m] + 1] - hap_off[ph[:m]])).sum())
    got = out[:m].cpu().numpy()
    print(f"CPU baseline (scalar restatement of compute_prob_scalar.cc, {cores} threads): {m} pairs in {dtc:.1f} s = {ccells/dtc/1e9:.3f} GCUPS; "
          f"max |log10 difference| vs GPU: {np.abs(got - want).max():.2e}", flush=True)
