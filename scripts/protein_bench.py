"""BASELINE.json configs[4] shape: protein Smith-Waterman with BLOSUM62 (gap open 11, extend 1), 300-residue
queries against a 5 000-sequence database, every query against every database sequence (a subset of the 1 M
queries: queries x 5 000 pairs per pass).  The reference has no such path: this
measures the substitution-matrix extension; results are checked against the CPU restatement's extension on a sample.

--layout shared (default, round 5): mgl_amd.protein.DatabaseSearch -- per database sequence, its queries in tiles of 128 pairs that
share it (MGL_SW_FLAG_SHARED_TARGET, sw_dp16_lane_matrix_kernel), the queries beyond whole tiles and the targets too long for a
region of the workspace through the packed kernel.  --layout grouped: round 4's, pair = d * Q + q, blocks of eight of one geometry.

  python scripts/protein_bench.py [--queries 400] [--db 5000]
"""
import argparse, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import torch
from mgl_amd import protein, smithwaterman as sw

ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=400)
ap.add_argument("--db", type=int, default=5000)
ap.add_argument("--query-len", type=int, default=300)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--stride", type=int, default=256, help="CIGAR bytes kept per pair (longer ones are flagged, not written)")
ap.add_argument("--workspace-gib", type=float, default=64, help="shared layout: the regions of 3 072 wave slots, each sized by the tile it starts on, take 41 GB on this database; grouped layout at 64: 2 079 GCUPS")
ap.add_argument("--layout", choices=("shared", "grouped"), default="shared")
ap.add_argument("--check", type=int, default=200)
ap.add_argument("--score-only", action="store_true", help="MGL_SW_FLAG_SCORE_ONLY: the database-search pre-filter mode")
ap.add_argument("--seconds", type=float, default=0.0, help="repeat the pass for at least this long (SURVEY 8d: a seeded subset sized to >= 30 s) instead of --steps times")
ap.add_argument("--cpu-seconds", type=float, default=0.0, help="time the CPU restatement's matrix extension on all host cores for about this long (0: skip)")
ap.add_argument("--json", action="store_true", help="one JSON line with the figures, a roofline object and the CPU figure (bench.py reads it)")
args = ap.parse_args()
rng = np.random.default_rng(42)
code, mat = protein.blosum62()
# database: log-normal lengths (median ~300, clipped to [40, 2000]), sorted by length so that the four pairs of a wave
# have like geometry; a fifth of the queries are diverged fragments of database sequences, the rest unrelated
lens = np.clip(np.exp(rng.normal(5.7, 0.55, size=args.db)).astype(np.int64), 40, 2000)
lens.sort()
db_off = np.zeros(args.db + 1, np.int64); np.cumsum(lens, out=db_off[1:])
db = protein.random_proteins(rng, 1, int(db_off[-1]))[0]
Q, QL = args.queries, args.query_len
queries = protein.random_proteins(rng, Q, QL)
for k in range(0, Q, 5):
    d = int(rng.integers(0, args.db))
    if lens[d] >= QL:
        s = int(rng.integers(0, lens[d] - QL + 1)); frag = db[db_off[d] + s: db_off[d] + s + QL].copy()
        mut = rng.random(QL) < 0.4
        frag[mut] = protein.random_proteins(rng, 1, int(mut.sum()))[0]
        queries[k] = frag
n = Q * args.db
dev = torch.device("cuda", 0)
cells = int(lens.sum()) * Q * QL
a = sw.MicrosoftSmithWaterman(0)
a.set_workspace(int(args.workspace_gib * (1 << 30)))
shared = args.layout == "shared" and not os.environ.get("MGL_PROTEIN_INT32")
if shared:
    ds = protein.DatabaseSearch(db, db_off, queries, dev, args.stride)
    run_pass = lambda: ds.run(a, code, mat, 11, 1, score_only=args.score_only)
    batches = ds.batches()
    where = ds.where
else:
    t_start = torch.from_numpy(np.repeat(db_off[:-1], Q)).to(dev)                 # pair index = d * Q + q
    t_len = torch.from_numpy(np.repeat(lens, Q).astype(np.int32)).to(dev)
    q_start = torch.from_numpy(np.tile(np.arange(Q, dtype=np.int64) * QL, args.db)).to(dev)
    q_len = torch.full((n,), QL, dtype=torch.int32, device=dev)
    b = protein.IndexedBatch(torch.from_numpy(db).to(dev), t_start, t_len, torch.from_numpy(queries.reshape(-1)).to(dev), q_start, q_len,
                             int(lens.max()), QL, args.stride)
    grouped = (Q % 8 == 0) and not os.environ.get("MGL_PROTEIN_INT32")
    run_pass = lambda: protein.run_matrix(b, a, code, mat, 11, 1, grouped=grouped, score_only=args.score_only)
    batches = [b]
    where = lambda d, q: (b, d * Q + q)
run_pass(); torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 0
while steps < args.steps or time.perf_counter() - t0 < args.seconds:
    run_pass()
    steps += 1
    if args.seconds and steps % 8 == 0:
        torch.cuda.synchronize()   # (the clock above is the host's: keep the queue a few passes deep, not hundreds)
torch.cuda.synchronize()
total_s = time.perf_counter() - t0
dt = total_s / steps
a.set_profiling(3)   # one more pass for the kernels' own times, summed over its calls
run_pass(); torch.cuda.synchronize()
tm = a.timing()
a.set_profiling(0)
over = sum(int((x.status != 0).sum()) for x in batches)
kernels = f"sw_dp16_lane_matrix{'_score' if args.score_only else ''}_kernel (tiles of 128 pairs that share their target) + sw_dp16_matrix{'_score' if args.score_only else ''}_kernel (the rest)" if shared else ("sw_dp16_matrix_kernel" if tm.packed16 else "sw_dp_matrix_kernel")
layout_note = ""
if shared:
    layout_note = (f"; {ds.shared.n} pairs in tiles on targets up to {ds.shared_max_tl} residues, {0 if ds.rest is None else ds.rest.n} beyond whole tiles, "
                   f"{0 if ds.long is None else ds.long.n} on longer targets")
print(f"protein SW (BLOSUM62, 11/1, SOFTCLIP, {kernels}{', score only' if args.score_only else ''}): {Q} queries of {QL} aa x {args.db} database sequences (mean {lens.mean():.0f} aa) = "
      f"{n} pairs, {dt*1e3:.1f} ms per pass = {cells/dt/1e9:.1f} GCUPS, {n/dt/1e6:.2f} M alignments/s (fill {tm.dp_ms:.1f} ms in "
      f"{tm.dp_launches} launches, traceback {tm.tb_ms:.1f} ms; {over} CIGARs longer than {args.stride} bytes flagged{layout_note})", flush=True)
if args.check:
    import ctypes as C
    import oracle_lib as ol
    idx = rng.choice(n, size=args.check, replace=False)
    L = ol.oracle()
    L.swo_align_matrix.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
    host = {id(x): (x.offsets.cpu().numpy(), x.scores.cpu().numpy(), x.status.cpu().numpy(), None, x.cigar_len.cpu().numpy()) for x in batches}
    for k in idx:
        d, q = divmod(int(k), Q)
        bx, p_ = where(d, q)
        offs, scs, st, cgs, lns = host[id(bx)]
        t = db[db_off[d]:db_off[d + 1]].tobytes(); qq = queries[q].tobytes()
        buf = C.create_string_buffer(8192); ln, off = C.c_int(), C.c_int(); ez = (C.c_int32 * 6)()
        assert L.swo_align_matrix(t, len(t), qq, len(qq), code.ctypes.data, mat.ctypes.data, 11, 1, 1, buf, 8192, C.byref(ln), C.byref(off), ez) == 0
        assert tuple(scs[p_]) == tuple(ez), k
        if st[p_] == 0 and not args.score_only:
            row = cgs[p_] if cgs is not None else bx.cigars[p_].cpu().numpy()
            assert offs[p_] == off.value and row[:lns[p_]].tobytes() == buf.raw[:ln.value], k
    print(f"checked {len(idx)} random pairs against the CPU restatement's extension: identical", flush=True)
cpu = None
if args.cpu_seconds > 0:
    # CPU figure: the restatement's substitution-matrix extension (oracle/sw_oracle.c swo_align_matrix -- there is NO reference path for this
    # workload), one alignment per task on every host core, on a seeded sample sized to the time asked for
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import oracle_lib as ol
    from bench import host_cores
    L = ol.oracle()
    L.swo_align_matrix.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                   C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
    cores = host_cores()
    def one(k):
        d, q = divmod(int(k), Q)
        t = db[db_off[d]:db_off[d + 1]].tobytes(); qq = queries[q].tobytes()
        buf = C.create_string_buffer(8192); ln, off = C.c_int(), C.c_int(); ez = (C.c_int32 * 6)()
        assert L.swo_align_matrix(t, len(t), qq, len(qq), code.ctypes.data, mat.ctypes.data, 11, 1, 1, buf, 8192, C.byref(ln), C.byref(off), ez) == 0
        return len(t) * len(qq)
    sample = rng.choice(n, size=min(n, 25000 * cores), replace=False)
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter(); probe = sum(ex.map(one, sample[:40 * cores])); dtp = time.perf_counter() - t0
        m = int(min(len(sample), max(40 * cores, 40 * cores * args.cpu_seconds / dtp)))
        t0 = time.perf_counter(); ccells = sum(ex.map(one, sample[:m], chunksize=8)); dtc = time.perf_counter() - t0
    cpu = {"gcups": round(ccells / dtc / 1e9, 3), "cores": cores, "pairs": m, "seconds": round(dtc, 1)}
    print(f"CPU figure (the restatement's matrix extension, {cores} threads; no reference path exists): {m} alignments in {dtc:.1f} s = {ccells/dtc/1e9:.3f} GCUPS", flush=True)
if args.json:
    import json
    # SURVEY 8d's bytes per unit for what this kernel does: residues as shipped (one byte each) + the four index words + results, and --
    # this kernel DOES spill its traceback -- tl * ql / 2 bytes of flags
    k_s = (tm.dp_ms + tm.tb_ms) / 1e3 if tm.dp_ms > 0 else dt
    mean_tl = float(lens.mean())
    alg = mean_tl + QL + 24 + 36 + (0 if args.score_only else mean_tl * QL / 2)
    traffic, traffic_why = None, "no PMC entry"
    pmc_key = "sw_dp16_lane_matrix_kernel" if shared else "sw_dp16_matrix_kernel"
    try:
        sys.path.insert(0, os.path.join(R, "scripts"))
        import src_hash
        for r_ in json.load(open(os.path.join(R, "profiles", "pmc_traffic.json"))).get(pmc_key, []):
            if not r_.get("superseded") and (shared or tm.packed16) and not args.score_only and (Q, args.db, QL) == (400, 5000, 300) and r_.get("workspace_gib", args.workspace_gib) == args.workspace_gib:
                ok_, traffic_why = src_hash.check(r_)
                traffic = int(r_["hbm_bytes_per_pair"] * n) if ok_ else None
                break
    except (OSError, KeyError, ValueError, ImportError) as e_:
        traffic_why = repr(e_)
    if shared:
        note = ("tiles of 128 pairs that share their target run two pairs per lane (sw_dp16_lane_matrix_kernel): a strip of 32 target rows is the same 32 residues for every lane, so a "
                "column's scores are one 32-byte row of a per-strip profile in LDS (four 16-byte reads per column and lane instead of 64 gathers), a cell's score one v_perm_b32 and one add "
                "(the table's bias is the gap's o - e, which the recurrence subtracts anyway); the flags are spilled (tl x ql / 2 bytes per alignment) into the wave slot's own region and "
                "walked by the same wave: one launch of a persistent grid.  The 16 queries per database sequence beyond three whole tiles and the targets too long for a region of the "
                "workspace take round 4's packed kernel (counters: profiles/r05_b_protein_pmc.txt).  Round 4's layout (--layout grouped): 2 079 GCUPS")
    else:
        note = ("the traceback is spilled (four flags per cell, tl x ql / 2 bytes per alignment).  Counters (profiles/r05_b_protein_pmc.txt, rocprofv3 --pmc, the 74-launch build): "
                "SQ_INSTS_VALU 3.79e10 per pass = 23.6 per two-cell step (the DNA form: 19.6; the two table look-ups are LDS gathers) -> VALU issue 51 % of the "
                "pass; SQ_LDS_IDX_ACTIVE 4.15e10 cycles per pass = 56 % of it, half of them bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.51: 64 lanes "
                "gathering out of a 2 KB table); nine waves per CU (12 B of LDS per query residue and group of two pairs); WRITE_SIZE 105 GB per pass = 0.87-1.06 TB/s.  "
                "Neither pipe is full: the launches were -- 3 379 waves on 2 304 slots ran 1.47 rounds in the time of two; whole rounds per chunk and a 64 GiB "
                "workspace: 1 684 -> 2 079 GCUPS")
    if args.score_only:
        note = "MGL_SW_FLAG_SCORE_ONLY: all six ScoreMax fields, no flags formed or stored, no regions, no walk (7 VALU instructions per two-cell step in the shared-target kernel).  " + note
    out = {"gcups": round(cells / dt / 1e9, 1), "pairs_per_pass": int(n), "alignments_per_s": round(n / dt, 1), "ms_per_pass": round(dt * 1e3, 3), "passes": steps,
           "seconds": round(total_s, 1), "kernel": kernels, "layout": args.layout if shared or args.layout == "grouped" else "grouped", "workspace_gib": args.workspace_gib,
           "kernel_ms": {"fill": round(tm.dp_ms, 3), "traceback": round(tm.tb_ms, 3), "launches": int(tm.dp_launches)},
           "roofline": {"bound": "hbm", "achieved": round(alg * n / dt / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(alg * n / dt / 8e12, 4), "traffic": traffic, "traffic_per": "pass", "counters": traffic_why,
                        "algorithmic_bytes_per_alignment": round(alg, 1), "note": note},
           "parity": "no reference path exists for this workload (mgl scores by byte equality only, sw.cpp:55): checked against the CPU restatement's extension, parity with the reference neither pinned nor claimed"}
    if shared:
        out["pairs"] = {"in_tiles_that_share_their_target": int(ds.shared.n), "tile_targets_up_to": int(ds.shared_max_tl), "beyond_whole_tiles": 0 if ds.rest is None else int(ds.rest.n),
                        "on_longer_targets": 0 if ds.long is None else int(ds.long.n)}
    if cpu:
        out["cpu_no_reference_path"] = cpu
    print(json.dumps(out), flush=True)
