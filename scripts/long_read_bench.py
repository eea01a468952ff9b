"""BASELINE.json configs[3] shape: ONT-style ~10 kb x ~10 kb pairs (5 % sub /
5 % ins / 5 % del), full matrix, on-device traceback and full CIGAR (scores exceed 16 bits: the fill kernel works in
16 bits relative to moving baselines and checks its window, sw_dp_coop.hip).  The traceback needs tl*ql/2
bytes per pair (50 MB), so the number of pairs in flight is set by the workspace; the cooperative fill kernel
(sw_dp_coop.hip) fills the chip with 256 of them.

  python scripts/long_read_bench.py --pairs 2048 --workspace-gib 220 --seconds 30 --cpu-pairs 64
"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
from mgl_amd import device_batch, synth
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, GATK_PARAMETERS, SWOverhangStrategy, concat

ap = argparse.ArgumentParser()
ap.add_argument("pairs", nargs="?", type=int, default=256)
ap.add_argument("workspace_gib", nargs="?", type=float, default=32)
ap.add_argument("length", nargs="?", type=int, default=10000)
ap.add_argument("check", nargs="?", type=int, default=2, help="pairs compared with the scalar oracle")
ap.add_argument("--seconds", type=float, default=0, help="repeat the batch until this much GPU time has been spent")
ap.add_argument("--cpu-pairs", type=int, default=0, help="time the reference's CPU path (oracle/_ref) on this many pairs")
ap.add_argument("--distinct", type=int, default=32, help="distinct synthetic pairs (the batch cycles through them)")
ap.add_argument("--in-flight-seconds", type=float, default=0, help="after the timed passes: this many seconds of passes with TWO in flight (a second context and stream)")
ap.add_argument("--json", action="store_true", help="print one JSON line with the figures and a roofline object (bench.py reads it)")
args = ap.parse_args()
PARAMS = (200, -150, 260, 11)  # GATK_PARAMETERS after the sign normalisation of the JNI boundary
n, length = args.pairs, args.length

rng = synth.rng_for(11)
base = [synth.ont_pair(rng, length) for _ in range(min(n, args.distinct))]
ts = [base[k % len(base)][0].tobytes() for k in range(n)]
qs = [base[k % len(base)][1].tobytes() for k in range(n)]
td, toff = concat(ts); qd, qoff = concat(qs)
stride = 2 * (length + 2000)
b = device_batch.from_host(td, toff, qd, qoff, "cuda:0", cigar_stride=stride)
a = MicrosoftSmithWaterman(0)
a.set_workspace(int(args.workspace_gib * (1 << 30)))
if os.environ.get("MGL_STRIP"):
    a.set_strip_kernel(int(os.environ["MGL_STRIP"]))   # 1 = never, 2 = whenever eligible
if os.environ.get("MGL_STORED"):
    a.set_lane_checkpoint(1)   # the strip kernel with the flags of every cell stored (default: none stored, the walk recomputes blocks)
if os.environ.get("MGL_COOP_W"):
    a.set_cooperative(int(os.environ["MGL_COOP_W"]))
cells = b.cells
b.run(a); torch.cuda.synchronize()
a.set_profiling(1)
reps, t0 = 0, time.perf_counter()
while True:
    b.run(a); torch.cuda.synchronize()
    reps += 1
    dt = time.perf_counter() - t0
    if dt >= args.seconds:
        break
tm = a.timing()
print(f"{n} pairs of ~{length} x {length}, {reps} pass(es): {dt*1e3/reps:.1f} ms per pass = {cells*reps/dt/1e9:.1f} GCUPS over "
      f"{dt:.1f} s (last pass: fill {tm.dp_ms:.1f} ms in {tm.dp_launches} launches, traceback {tm.tb_ms:.1f} ms, "
      f"traceback workspace {tm.tb_bytes/2**30:.1f} GiB)", flush=True)
assert int((b.status != 0).sum()) == 0
two_in_flight = None
if args.in_flight_seconds > 0:
    # pass after pass with TWO in flight: a second context (its own workspace) and result arrays, two streams taking the passes in turn --
    # the walk of one pass (one wave per pair) runs beside the fill of the next
    a.set_profiling(0)
    a2 = MicrosoftSmithWaterman(0)
    a2.set_workspace(int(args.workspace_gib * (1 << 30)))
    b2 = device_batch.DeviceBatch(b.targets, b.t_off, b.queries, b.q_off, b.max_tl, b.max_ql, b.cigar_stride)
    lanes = [(a, b, torch.cuda.Stream()), (a2, b2, torch.cuda.Stream())]
    torch.cuda.synchronize()
    for al, bt, st in lanes:
        with torch.cuda.stream(st):
            bt.run(al)
    torch.cuda.synchronize()
    k, t1 = 0, time.perf_counter()
    while True:
        for al, bt, st in lanes:
            with torch.cuda.stream(st):
                bt.run(al)
        k += 2
        if k % 8 == 0:  # (the host's clock: keep the queues a few passes deep, not hundreds)
            torch.cuda.synchronize()
            if time.perf_counter() - t1 >= args.in_flight_seconds:
                break
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t1
    assert torch.equal(b.scores, b2.scores) and torch.equal(b.cigars, b2.cigars) and int((b2.status != 0).sum()) == 0
    two_in_flight = {"gcups": round(cells * k / dt2 / 1e9, 1), "ms_per_pass": round(dt2 * 1e3 / k, 2), "passes": k, "seconds": round(dt2, 1)}
    print(f"two passes in flight (two contexts, two streams in turn): {dt2*1e3/k:.1f} ms per pass = {cells*k/dt2/1e9:.1f} GCUPS over {dt2:.1f} s; "
          f"both contexts' results identical", flush=True)
    a2.close()
if args.json:
    import json
    # HBM roofline of the fill kernel (north_star's roofline; the kernel is VALU-issue bound): algorithmic bytes per pair =
    # both sequences + offsets + the 4-bit-per-cell traceback spilled to HBM + the fill record
    # (the traceback counts only where it is spilled to HBM -- the strip kernel's default form keeps none)
    spilled = a.slot_layout(0) != 6
    out_bytes = float(b.cigar_len.float().mean().item()) + 36
    alg = sum(len(ts[k]) + len(qs[k]) + 16 + out_bytes + (len(ts[k]) * len(qs[k]) // 2 + 32 if spilled else 0) for k in range(n))
    fill_s = tm.dp_ms / 1e3
    # HBM bytes per pass by the PMC passes of the shipped build (profiles/pmc_traffic.json: WRITE_SIZE + FETCH_SIZE of the fill and the walk,
    # per 10 kb pair; other lengths: none taken)
    traffic, traffic_src = None, None
    try:
        sys.path.insert(0, os.path.join(R, "scripts"))
        import src_hash  # (an entry's counters are printed only when they were taken on the kernel sources this run is built from)
        for r_ in json.load(open(os.path.join(R, "profiles", "pmc_traffic.json")))["sw_dp16_strip_kernel"]:
            if (r_["tl"], r_["ql"]) == (length, length) and not r_.get("superseded") and not spilled and tm.fill_kernel == 6 and "hbm_bytes_per_pair" in r_:
                ok_, why_ = src_hash.check(r_)
                traffic, traffic_src = (int(r_["hbm_bytes_per_pair"] * n), r_["source"]) if ok_ else (None, "profiles/pmc_traffic.json: " + why_)
                break
    except (OSError, KeyError, ValueError, ImportError):
        pass
    walk_name = "sw_traceback_wave_kernel" if spilled else "sw_strip_ck_walk_kernel"
    print(json.dumps({"gcups": round(cells * reps / dt / 1e9, 1), "pairs": n, "length": length, "passes": reps, "seconds": round(dt, 1),
                      "ms_per_pass": round(dt * 1e3 / reps, 2), "two_passes_in_flight": two_in_flight,
                      "kernel_ms": {a.fill_kernel_name(tm): round(tm.dp_ms, 2), walk_name: round(tm.tb_ms, 2), "launches": tm.dp_launches},
                      "traceback": "4 bits per cell in HBM" if spilled else "none stored: kept rows and checkpoints (10 MB per 10 kb pair), the walk recomputes the blocks the path crosses",
                      "roofline": {"bound": "hbm", "kernel": a.fill_kernel_name(tm), "achieved": round(alg / fill_s / 1e9, 1), "peak": 8000.0,
                                   "unit": "GB/s", "frac": round(alg / fill_s / 1e9 / 8000.0, 5), "traffic": traffic, "traffic_source": traffic_src,
                                   "traffic_frac_of_peak": None if traffic is None else round(traffic / (dt / reps) / 8e12, 4),
                                   "algorithmic_bytes_per_pass": round(alg), "kernel_gcups": round(cells / fill_s / 1e9, 1),
                                   "if_traceback_were_spilled_frac": round(sum(len(ts[k]) * len(qs[k]) // 2 for k in range(n)) / fill_s / 1e9 / 8000.0, 4),
                                   "note": ("one strip of 17-32 rows per lane-half, the lane kernel's column code with per-strip 16-bit baselines; VALU-issue "
                                            "bound; " + ("no flags stored (score-only column code: 8 instructions per two cells with base codes, 9 with the byte compare), profiles/r04_m_final_pmc.txt"
                                                         if not spilled else "flags of every cell stored, profiles/r02_d_strip_kernel.txt")
                                            if tm.fill_kernel == 6 else
                                            "packed int16 wavefront, 128 rows per wave, VALU-issue bound (36 VALU instructions per 128-cell step = 87 % of "
                                            "the issue peak, profiles/r02_b_long_reads.txt)" if tm.fill_kernel == 5 else
                                            "int32 wavefront, VALU-issue bound (about 22 instructions per 64-cell step)")}}),
          flush=True)
import oracle_lib as ol
if args.check:
    idx = list(range(min(args.check, len(base))))
    cg = b.cigar_strings(idx)
    for k in idx:
        o = ol.oracle_align(ts[k], qs[k], PARAMS, ol.SOFTCLIP)
        assert (int(b.offsets[k]), cg[k], tuple(int(x) for x in b.scores[k])) == (o["offset"], o["cigar"], o["score"]), k
    print(f"checked {len(idx)} pairs against the oracle: identical", flush=True)
if args.cpu_pairs and ol.have_ref():
    sys.path.insert(0, R)
    from bench import host_cores
    m = min(args.cpu_pairs, n)
    cores = host_cores()
    lib = ol.ref()
    off = np.zeros(m, np.int32); cgb = np.zeros(m * stride, np.uint8); ln = np.zeros(m, np.int32)
    mt, mx, mo, me = PARAMS
    t0 = time.perf_counter()
    rc = lib.ref_align_batch(m, td.ctypes.data, toff.ctypes.data, qd.ctypes.data, qoff.ctypes.data, mt, mx, mo, me,
                             int(SWOverhangStrategy.SOFTCLIP), 1, cores, off.ctypes.data, cgb.ctypes.data, stride, ln.ctypes.data)
    dtc = time.perf_counter() - t0
    assert rc == 0
    ccells = int(sum(len(ts[k]) * len(qs[k]) for k in range(m)))
    g_off = b.offsets[:m].cpu().numpy(); g_cg = b.cigars[:m].cpu().numpy()
    mism = int((g_off != off).sum() + (g_cg != cgb.reshape(m, stride)).any(axis=1).sum())
    print(f"CPU baseline: mgl align_avx via oracle/_ref, {m} pairs on {cores} threads: {dtc:.2f} s = {ccells/dtc/1e9:.2f} GCUPS; "
          f"mismatches vs GPU: {mism}", flush=True)
