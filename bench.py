#!/usr/bin/env python3
"""bench.py -- GCUPS of the MI355X Smith-Waterman path on BASELINE.json's configs[1].

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (fill kernel + traceback kernel, every pair to offset,
ScoreMax and CIGAR text) over one batch of synthetic input that is already resident in HBM:
by default 10 M Illumina-style 150 bp reads, each against its own 256-base reference window
(BASELINE.json configs[1], SURVEY.md 8d "Config 2"), GATK parameters, SOFTCLIP, full matrix.
With N > 1 the ONE seeded 10 M-pair workload is sharded contiguously over the ranks (strong scaling, BASELINE.json
configs[2] / SURVEY.md 8d config 3: 1.25 M pairs per GPU at N = 8; each rank generates only its shard), there is no
data-path collective, and the step ends with the RCCL gather of the int32 scores onto rank 0 (north_star);
`--scaling weak` gives every rank its own --pairs instead.  With N > 1 a rank keeps TWO steps in flight (`--steps-in-flight`,
`config.steps_in_flight`: two contexts on two streams take the steps in turn, so that the tail of one shard's launch -- a tenth
of it at 1.25 M pairs -- runs beside the start of the next; the line also carries rank 0's step time one at a time); on one GPU the
default is one in flight, and the roofline's launch durations are those of launches that have the chip to themselves.

Rank 0 prints ONE JSON line.  `value` = whole-job GCUPS = sum(tl*ql) over all ranks and steps /
max-over-ranks wall time.  `roofline` prices the dominant kernel (sw_dp_kernel) against HBM
bandwidth with the algorithmic bytes of DESIGN.md; `cpu_baseline` is the reference's own AVX2
path (oracle/_ref, built from /root/reference in the authoring container) timed on this host's
cores on a bounded sample of the same batch and cross-checked against the GPU results.
"""
import argparse
import contextlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mgl_amd import device_batch, dist  # noqa: E402
from mgl_amd.smithwaterman import GATK_PARAMETERS, MicrosoftSmithWaterman, SWOverhangStrategy  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
DEFAULT_WORKSPACE_GIB = 8.0  # the 10 M-pair batch of configs[1] is ONE launch: a persistent grid of 2 048 waves, 2 MB of workspace each + 32 B per pair = 4.5 GB (round 3: 208 GiB)
PCIE_WORKSPACE_GIB = 12.0    # the host entries run two launches side by side (two workspace halves): 2 048 wave slots each
TL1000_WORKSPACE_GIB = 24.0  # tl = 1000: 7.3 MB per wave slot


def algorithmic_bytes_per_pair(tl, ql, packed2=False, traceback_spilled=False, cigar_bytes=0.0):
    """SURVEY.md 8d "algorithmic bytes per unit of work", per pair, for what the kernel in question does: in = the two
    sequences as shipped (ASCII like the reference's ByteBuffer, or 2-bit packed) + the two int64 descriptors this ABI
    takes per pair (16 B); out = offset (4 B) + the six ScoreMax fields (24 B) + CIGAR length and status (8 B) + the
    CIGAR bytes actually written; traceback = tl*ql/2 bytes + the 32-byte fill record ONLY for a kernel that spills it
    to HBM (sw_dp16_lane_ck_kernel keeps none: what it writes on top is schedule and shows in `traffic`)."""
    seq = (tl + 3) // 4 + (ql + 3) // 4 if packed2 else tl + ql
    return seq + 16 + 36 + cigar_bytes + ((tl * ql) // 2 + 32 if traceback_spilled else 0)


def host_cores():
    """Cores this process may actually use: the affinity mask capped by the cgroup CPU quota
    (the GPU box exposes 256 logical CPUs but grants a 16-CPU quota per GPU)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pmc_entry(tl, ql, kernel):
    """(entry, current, why) -- the committed rocprofv3 PMC passes for this kernel and geometry (profiles/pmc_traffic.json), and whether
    they were taken on the kernel sources THIS run is built from: the entry carries a hash of mgl_amd/csrc/<its .hip files + the headers
    they include>, recomputed here (scripts/src_hash.py; no git needed on the GPU box).  A stale or unhashed entry is never printed as a
    number: `traffic` / `valu` become null with `why` beside them."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import src_hash

        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        for r in rec[kernel]:  # (the newest entry of a geometry first; what it replaces carries "superseded")
            if (r["tl"], r["ql"]) == (tl, ql) and not r.get("superseded"):
                ok, why = src_hash.check(r)
                return r, ok, why
    except (OSError, KeyError, ValueError, ImportError) as e:
        return None, False, f"no PMC entry readable ({e!r})"
    return None, False, "no PMC pass was taken for this kernel and geometry"


def cpu_baseline(batch, target_seconds=15.0, max_pairs=10_000_000):
    """Time the reference's CPU path on this host; returns the cpu_baseline JSON object."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    cores = host_cores()
    tl, ql = batch.max_tl, batch.max_ql
    m, x, o, e = GATK_PARAMETERS
    stride = batch.cigar_stride

    def fetch(n):
        t = batch.targets[: n * tl].cpu().numpy()
        q = batch.queries[: n * ql].cpu().numpy()
        toff = np.arange(n + 1, dtype=np.int64) * tl
        qoff = np.arange(n + 1, dtype=np.int64) * ql
        return t, toff, q, qoff

    use_ref = ol.have_ref()
    if use_ref:
        lib = ol.ref()
        kind = "reference"
        label = "mgl align_avx (AVX2) via oracle/_ref" if lib.ref_has_avx2() else "mgl align_scalar via oracle/_ref"
    else:
        lib = ol.oracle()
        kind = "port"
        label = "oracle/sw_oracle.c scalar restatement"

    def run(n):
        t, toff, q, qoff = fetch(n)
        off = np.zeros(n, np.int32)
        cg = np.zeros(n * stride, np.uint8)
        ln = np.zeros(n, np.int32)
        t0 = time.perf_counter()
        if use_ref:
            rc = lib.ref_align_batch(n, t.ctypes.data, toff.ctypes.data, q.ctypes.data, qoff.ctypes.data, m, x, o, e,
                                     int(SWOverhangStrategy.SOFTCLIP), 1, cores, off.ctypes.data, cg.ctypes.data,
                                     stride, ln.ctypes.data)
        else:
            rc = lib.swo_align_batch(n, t.ctypes.data, toff.ctypes.data, q.ctypes.data, qoff.ctypes.data, m, x, o, e,
                                     int(SWOverhangStrategy.SOFTCLIP), cores, off.ctypes.data, None, cg.ctypes.data,
                                     stride, ln.ctypes.data)
        dt = time.perf_counter() - t0
        assert rc == 0, rc
        return dt, off, cg.reshape(n, stride), ln

    probe = min(batch.n, 20_000)
    dt, *_ = run(probe)
    n = int(min(batch.n, max_pairs, max(probe, probe * target_seconds / max(dt, 1e-6))))
    dt, off, cg, ln = run(n)
    # cross-check the CPU answers against what the GPU wrote for the same pairs
    g_off = batch.offsets[:n].cpu().numpy()
    g_cg = batch.cigars[:n].cpu().numpy()
    mism = int((g_off != off).sum() + (g_cg != cg).any(axis=1).sum())
    return {
        "value": round(n * tl * ql / dt / 1e9, 3), "unit": "GCUPS", "cores": cores, "kind": kind,
        "sample": f"first {n} pairs of the same batch ({tl}x{ql}), {label}, {cores} threads, {dt:.1f} s",
        "reads_per_s": round(n / dt, 1), "mismatches_vs_gpu": mism,
    }


def file_batch(args, dev):
    """--dataset bam / fastx: pairs read from files (mgl_amd/formats.py), tiled to --pairs on the device."""
    from mgl_amd import formats

    if args.dataset == "bam":
        path = os.path.join(ROOT, "tests", "golden", "HiSeq.1mb.1RG.2k_lines.bam")
        ts, qs, _ = formats.bam_pairs(path, window=args.tl)
        label = f"real Illumina reads (tests/golden/{os.path.basename(path)}, {len(ts)} distinct pairs, tiled)"
    else:
        assert args.fasta and args.fastq, "--dataset fastx needs --fasta and --fastq"
        _, _, genome = next(iter(formats.read_fasta(args.fasta)))
        ts, qs = [], []
        for _name, desc, seq, _qual in formats.read_fastq(args.fastq):
            pos = int(dict(kv.split("=") for kv in desc.split() if "=" in kv)["pos"])
            ts.append(genome[pos:pos + args.tl])
            qs.append(seq)
        label = f"{os.path.basename(args.fasta)} + {os.path.basename(args.fastq)} ({len(ts)} distinct pairs, tiled)"
    ql = len(qs[0])
    assert all(len(t) == args.tl for t in ts) and all(len(q) == ql for q in qs), "one geometry per bench batch"
    T = torch.from_numpy(np.frombuffer(b"".join(ts), dtype=np.uint8).reshape(len(ts), args.tl).copy()).to(dev)
    Q = torch.from_numpy(np.frombuffer(b"".join(qs), dtype=np.uint8).reshape(len(qs), ql).copy()).to(dev)
    idx = torch.arange(args.pairs, device=dev) % len(ts)
    t_off = torch.arange(args.pairs + 1, device=dev, dtype=torch.int64) * args.tl
    q_off = torch.arange(args.pairs + 1, device=dev, dtype=torch.int64) * ql
    # real reads can need long CIGARs: the Java side's buffer size, 2 * max(tl, ql) (MicrosoftSmithWaterman.java:71)
    b = device_batch.DeviceBatch(T[idx].reshape(-1), t_off, Q[idx].reshape(-1), q_off, args.tl, ql, 2 * max(args.tl, ql),
                                 uniform=True)
    return b, label


def secondary_workloads():
    """The other workloads of BASELINE.json / SURVEY.md 8f, each run once as its own short script after the timed
    region (never part of `value`); failures are reported, not raised."""
    import re
    import subprocess

    out = {}
    runs = {
        # "a seeded subset sized to >= 30 s of GPU time": 4 608 pairs of ~10 kb x ~10 kb per pass (one launch: six
        # rounds of the chip for the strip kernel with four waves per pair, 256 CUs x 3 workgroups; 37 GiB of kept rows and checkpoints),
        # passes repeated for 30 s; CPU baseline (the reference's align_avx) on 64 of the pairs.  (Rounds 1-3 ran 2 304 pairs per pass:
        # 4 086 GCUPS on this build against 4 357 -- the walk kernel, one wave per pair, fills the chip only from ~ 4 000 pairs on.)
        "long_reads_10kb (configs[3] shape, 4608 pairs per pass, >= 30 s)": ["scripts/long_read_bench.py", "4608", "110", "10000", "1", "--seconds", "30", "--in-flight-seconds", "6",
                                                                             "--cpu-pairs", "64", "--json"],
        "mixed_read_lengths (4 M reads of 100-150 bases x 256-base windows, no geometry promise from the caller)": ["scripts/grouped_bench.py", "4000000", "100", "--json"],
        "pairhmm_150x300 (SURVEY 8f rank 3, 1.6 M pairs)": ["scripts/pairhmm_bench.py", "--steps", "3", "--cpu-seconds", "3", "--json"],
        # configs[4] to the standard of configs[3]: a seeded subset (400 queries x 5 000 database sequences = 2 M alignments per pass) repeated for
        # >= 30 s, with a roofline object and a CPU figure (the restatement's matrix extension: there is no reference path to time)
        "protein_blosum62 (configs[4] shape, 2 M alignments per pass, >= 30 s, no reference path: parity unpinned)": ["scripts/protein_bench.py", "--steps", "2", "--seconds", "30", "--check", "50", "--workspace-gib", "64",
                                                                                                             "--cpu-seconds", "10", "--json"],
        # ... and its pre-filter mode (MGL_SW_FLAG_SCORE_ONLY: all six ScoreMax fields, no CIGAR), the same subset for >= 10 s
        "protein_blosum62_score_only (the same 2 M alignments per pass, scores only, >= 10 s)": ["scripts/protein_bench.py", "--steps", "2", "--seconds", "10", "--check", "50", "--workspace-gib", "64",
                                                                                                "--score-only", "--json"],
    }
    for name, cmd in runs.items():
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, cmd[0])] + cmd[1:], capture_output=True, text=True, timeout=400)
            m = re.search(r"= ([0-9.]+) GCUPS", r.stdout)
            if r.returncode == 0 and m:
                out[name] = {"gcups": float(m.group(1)), "checked": ("identical" in r.stdout) or None}
                for ln in r.stdout.splitlines():   # the script's own JSON line (figures over the whole run + its roofline object)
                    if ln.startswith("{"):
                        out[name].update(json.loads(ln))
                mm = re.search(r"mismatches vs GPU: (\d+)", r.stdout)
                if mm:
                    out[name]["cpu_mismatches_vs_gpu"] = int(mm.group(1))
                d = re.search(r"max \|log10 difference\| vs GPU: ([0-9.e+-]+)", r.stdout)
                c = re.search(r"CPU baseline[^:]*: .* = ([0-9.]+) GCUPS", r.stdout)
                if d:
                    out[name]["max_abs_log10_diff_vs_cpu"] = float(d.group(1))
                    out[name]["checked"] = float(d.group(1)) < 1e-5
                if c:
                    if name.startswith("pairhmm"):
                        # NOT mgl's AVX2 float path (its sources need TBB, absent from this image, and cannot be built here): the scalar C
                        # restatement of compute_prob_scalar.cc under oracle/, pinned by the reference's 104 known answers
                        out[name]["cpu_gcups_scalar_restatement"] = float(c.group(1))
                        out[name]["cpu_baseline_kind"] = "port (scalar restatement of compute_prob_scalar.cc; the reference's AVX2 path is not buildable here: TBB absent)"
                    else:
                        out[name]["cpu_gcups"] = float(c.group(1))
            else:
                out[name] = {"error": (r.stderr or r.stdout)[-200:]}
        except Exception as e:  # noqa: BLE001 -- a secondary line must never take the headline down
            out[name] = {"error": repr(e)[:200]}
    # the reference's own calling pattern: one pair per call from 16 native threads through the
    # front-end of mgl_sw_align, and one active region per call through the JNI-shaped PairHMM entry -- latency, not throughput
    try:
        exe = os.path.join(ROOT, "tests", "cpp", "coalesce_bench")
        if os.path.exists(exe):
            # mailboxes (sw_service.hip: a resident wave per calling thread, no launch per call), then the coalescer alone
            line = {}
            for key, env in (("pairs_per_s", {}), ("coalescer_only_pairs_per_s", {"MGL_SW_SERVICE_SLOTS": "0"})):
                r = subprocess.run([exe, "16", "3000", "50"], capture_output=True, text=True, timeout=120, env=dict(os.environ, **env))
                m = re.search(r"front-end: ([0-9.]+) pairs/s .*?(\d+) through mailboxes .* wrong results (\d+)", r.stdout)
                if m:
                    line[key] = float(m.group(1))
                    line["checked"] = line.get("checked", True) and m.group(3) == "0"
                    if not env:
                        line["through_mailboxes"] = int(m.group(2))
            if line:
                out["one_pair_per_call, 16 native threads (mgl_sw_align, 256x150)"] = line
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pairhmm_region_latency.py")], capture_output=True, text=True, timeout=120)
        m = re.search(r"= 800 pairs: ([0-9.]+) us per call", r.stdout)
        if m:
            out["pairhmm, one region per call (100 reads x 8 haplotypes, host buffers)"] = {"us_per_call": float(m.group(1))}
    except Exception as e:  # noqa: BLE001
        out["latency_probes"] = {"error": repr(e)[:200]}
    return out


def pcie_inclusive_leg(aligner, batch, args):
    """SURVEY.md 8d's wall-time definition: the same pairs through the host-buffer entries -- inputs in host memory (the
    reference is always called with host buffers, MicrosoftSmithWaterman.java:66-86), H2D, kernels, D2H of every result --
    timed around the one blocking call and cross-checked against the arrays the device-resident headline run left in HBM.
    Three forms: ASCII bases in pageable memory (the reference's own contract: 4.2 GB in), the same arrays REGISTERED
    (mgl_sw_register_host_buffer), and 2-bit packed bases with windows into one packed genome (SURVEY 8d config 2's wire format:
    0.54 GB in) -- in arrays the caller page-locked itself (the entry's direct form: one gated launch, results written in place) and,
    inside that object, in arrays registered where they lie.  Never `value`."""
    from mgl_amd import _lib

    n, tl, ql, stride = batch.n, batch.max_tl, batch.max_ql, batch.cigar_stride
    aligner.set_workspace(int(max(args.workspace_gib, PCIE_WORKSPACE_GIB) * (1 << 30)))
    t, q = batch.targets.cpu().numpy(), batch.queries.cpu().numpy()
    toff, qoff = batch.t_off.cpu().numpy(), batch.q_off.cpu().numpy()
    off, sc = np.zeros(n, np.int32), np.zeros((n, 6), np.int32)
    cg, ln = np.zeros(n * stride, np.uint8), np.zeros(n, np.int32)
    m, x, o, e = GATK_PARAMETERS
    L = _lib.lib()
    want = (batch.offsets.cpu().numpy(), batch.cigars.cpu().numpy(), batch.scores.cpu().numpy())

    def mismatches():
        return int((want[0] != off).sum() + (want[1] != cg.reshape(n, stride)).any(axis=1).sum() + (want[2] != sc).any(axis=1).sum())

    def timed(call):
        call()                                 # untimed: the context's staging buffers exist from here on
        off[:] = -1
        times = []
        for _ in range(max(1, args.steps)):
            t0 = time.perf_counter()
            call()
            times.append(time.perf_counter() - t0)
        dt = sum(times) / len(times)
        return {"ms_per_step": round(dt * 1e3, 3), "gcups": round(n * tl * ql / dt / 1e9, 2), "reads_per_s": round(n / dt, 1),
                "calls": len(times), "mismatches_vs_headline": mismatches()}

    def ascii_call():
        rc = L.mgl_sw_align_batch(aligner.ctx, n, t.ctypes.data, toff.ctypes.data, q.ctypes.data, qoff.ctypes.data, m, x, o, e,
                                  int(SWOverhangStrategy.SOFTCLIP), off.ctypes.data, sc.ctypes.data, cg.ctypes.data, stride,
                                  ln.ctypes.data)
        assert rc == 0, rc

    out = timed(ascii_call)
    out.update({"bytes_in": int(t.nbytes + q.nbytes + toff.nbytes + qoff.nbytes),
                "bytes_out": int(off.nbytes + sc.nbytes + cg.nbytes + ln.nbytes), "host_memory": "pageable", "input": "ascii"})
    regs = [t, toff, q, qoff, off, sc, cg, ln]
    try:
        for a_ in regs:
            aligner.register_host_buffer(a_)
        r = timed(ascii_call)
        r.update({"host_memory": "registered by the caller (mgl_sw_register_host_buffer)", "input": "ascii", "bytes_in": out["bytes_in"]})
        out["registered"] = r
        for a_ in regs[:4]:
            aligner.unregister_host_buffer(a_)
        # 2-bit packed: the genome once, windows by base offset, reads packed back to back
        if getattr(batch, "win", None) is not None:
            g = torch.Generator(device=batch.targets.device)
            g.manual_seed(int(args.seed))
            genome = torch.randint(0, 4, (1 << 24,), generator=g, device=batch.targets.device, dtype=torch.uint8)
            lut = torch.zeros(256, dtype=torch.uint8, device=batch.targets.device)
            for k, ch in enumerate(b"ACGT"):
                lut[ch] = k
            G = device_batch._pack2bit_torch(genome).cpu().numpy()
            Q = device_batch._pack2bit_torch(lut[batch.queries.long()]).cpu().numpy() if (n * ql) % 4 == 0 else None
            if Q is not None:
                win = batch.win.cpu().numpy().astype(np.int64)
                qst = (np.arange(n, dtype=np.int64) * ql)
                packed = [G, win, Q, qst]
                for a_ in packed:
                    aligner.register_host_buffer(a_)

                def packed_call():
                    aligner.align_packed_2bit(G, 1 << 24, win, None, Q, n * ql, qst, None, tl, ql, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP,
                                              stride, out=(off, sc, cg, ln))

                r_reg = timed(packed_call)
                for a_ in packed:
                    aligner.unregister_host_buffer(a_)
                # the same call on arrays the CALLER page-locked with the runtime's own allocator (hipHostMalloc through torch's pinned
                # tensors: large pages for the device's page tables; hipHostRegister pins an array's 4 KB pages where they lie -- measured
                # 3-5 ms per 10 M pairs apart).  Every array pinned: the entry takes its DIRECT form -- one gated launch of the persistent
                # grid beside the copy engines, results written by the waves into these arrays
                pin = lambda x: torch.from_numpy(x).pin_memory().numpy()
                G, win, Q, qst = pin(G), pin(win), pin(Q), pin(qst)
                poff, psc, pcg, pln = pin(off), pin(sc), pin(cg), pin(ln)
                save = (off, sc, cg, ln)
                off, sc, cg, ln = poff, psc, pcg, pln

                def pinned_call():
                    aligner.align_packed_2bit(G, 1 << 24, win, None, Q, n * ql, qst, None, tl, ql, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP,
                                              stride, out=(off, sc, cg, ln))

                r = timed(pinned_call)
                tm_ = aligner.timing()
                # ... and with TWO calls in flight: a second context and result arrays, a host thread each (the entry blocks; ctypes
                # releases the interpreter) -- the next call's gated grid moves into the wave slots the last call's tail leaves
                # free, and its first inputs cross the link while the last call's last tiles compute
                two_calls = None
                try:
                    import threading
                    al2 = MicrosoftSmithWaterman(batch.targets.device.index or 0)
                    al2.set_workspace(int(args.workspace_gib * (1 << 30)))
                    out2 = tuple(pin(np.zeros_like(x)) for x in (off, sc, cg, ln))
                    calls = max(2, min(args.steps, 6))

                    def worker(al, outs, k):
                        for _ in range(k):
                            al.align_packed_2bit(G, 1 << 24, win, None, Q, n * ql, qst, None, tl, ql, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP, stride, out=outs)

                    worker(al2, out2, 1)  # untimed: the second context's buffers exist from here on
                    th = [threading.Thread(target=worker, args=(aligner, (off, sc, cg, ln), calls)), threading.Thread(target=worker, args=(al2, out2, calls))]
                    t2 = time.perf_counter()
                    for t_ in th:
                        t_.start()
                    for t_ in th:
                        t_.join()
                    dt2 = (time.perf_counter() - t2) / (2 * calls)
                    same = bool((out2[0] == off).all() and (out2[2] == cg).all() and (out2[1] == sc).all())
                    two_calls = {"ms_per_call": round(dt2 * 1e3, 3), "gcups": round(n * tl * ql / dt2 / 1e9, 2), "calls": 2 * calls, "both_identical": same,
                                 "mismatches_vs_headline": mismatches()}
                    al2.close()
                    del out2
                except Exception as e2_:  # noqa: BLE001
                    two_calls = {"error": repr(e2_)[:200]}
                r.update({"host_memory": "page-locked by the caller (hipHostMalloc)", "input": "2bit (one packed genome + window offsets, packed reads)",
                          "form": "direct: one gated launch, results written into the caller's arrays by the waves" if tm_.dp_launches == 1 else f"chunked ({tm_.dp_launches} launches)",
                          "bytes_in": int(G.nbytes + win.nbytes + Q.nbytes + qst.nbytes), "bytes_out": out["bytes_out"],
                          "two_calls_in_flight (two contexts, a host thread each)": two_calls,
                          "arrays_registered_in_place (hipHostRegister, 4 KB pages)": {k: r_reg[k] for k in ("ms_per_step", "gcups", "mismatches_vs_headline")}})
                out["packed_2bit"] = r
                off, sc, cg, ln = save
        for a_ in regs[4:]:
            aligner.unregister_host_buffer(a_)
    except Exception as e_:  # noqa: BLE001 -- the extra forms must not take the pageable line down
        out["registered_error"] = repr(e_)[:200]
    return out


def tl1000_leg(aligner, args, dev):
    """the tl = 1000 variant of config 2 (same cell count: pairs/4 windows of 1000 bases), device resident
    like the headline, cross-checked against the CPU checker on a sample."""
    n = max(8, args.pairs * args.tl // 1000 // 8 * 8)
    aligner.set_workspace(int(max(args.workspace_gib, TL1000_WORKSPACE_GIB) * (1 << 30)))
    b = device_batch.window_batch(args.seed, n, dev, window=1000, read_len=args.ql)
    b.run(aligner, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(max(1, args.steps)):
        b.run(aligner, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / max(1, args.steps)
    out = {"pairs": n, "target_len": 1000, "query_len": args.ql, "ms_per_step": round(dt * 1e3, 3),
           "gcups": round(n * 1000 * args.ql / dt / 1e9, 2), "reads_per_s": round(n / dt, 1),
           "cigar_overflows": int((b.status != 0).sum().item())}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol

    idx = list(range(0, n, max(1, n // 4000)))[:4000]
    ts, qs = b.host_pairs(idx)
    woff, wsc, wcg = ol.oracle_align_batch(ts, qs, GATK_PARAMETERS, ol.SOFTCLIP, nthreads=host_cores())
    out["sample_checked"] = len(idx)
    out["mismatches_vs_cpu"] = int((b.offsets[idx].cpu().numpy() != woff).sum() + (b.scores[idx].cpu().numpy() != wsc).any(axis=1).sum()
                                   + sum(a != c for a, c in zip(b.cigar_strings(idx), wcg)))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=10_000_000,
                    help="pairs per step: of the whole job with --scaling strong (default), per GPU with --scaling weak")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default, BASELINE configs[2]): ONE seeded --pairs workload, rank r aligns the contiguous "
                         "shard dist.shard_range(pairs, r, N) of it; weak: every rank its own --pairs pairs")
    ap.add_argument("--tl", type=int, default=256, help="reference window length")
    ap.add_argument("--ql", type=int, default=150, help="read length")
    ap.add_argument("--workspace-gib", type=float, default=None,
                    help=f"workspace per GPU for the headline; default: {DEFAULT_WORKSPACE_GIB:.0f} GiB (the 10 M-pair batch is one launch of a persistent grid), "
                         "or what the card has free beside the batch if that is less")
    ap.add_argument("--no-cpu", action="store_true", help="skip the host-CPU baseline leg")
    ap.add_argument("--steps-in-flight", type=int, choices=(1, 2), default=None,
                    help="2: two contexts on two streams take the steps in turn, so that the next step's grid moves into the wave slots the "
                         "last one's tail leaves free (default with more than one rank, where a step is a shard of 1.25 M pairs and its tail "
                         "a tenth of it); 1: one context, one stream (default on one GPU: the launch durations of the roofline stay those of "
                         "launches that have the chip to themselves)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads (long reads, PairHMM, protein)")
    ap.add_argument("--no-extra", action="store_true", help="skip the PCIe-inclusive and tl=1000 legs (SURVEY 8d)")
    ap.add_argument("--dump-scores", help="rank 0 writes the gathered int32 score vector of the last step here (.npy)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--input", choices=("ascii", "2bit"), default="ascii",
                    help="ascii: concatenated bytes (the reference's ByteBuffer contract); 2bit: one 2-bit packed "
                         "genome with target windows addressed by base offset + packed reads (SURVEY 8d config 2)")
    ap.add_argument("--dataset", choices=("synthetic", "bam", "fastx"), default="synthetic",
                    help="synthetic: BASELINE configs[1] (default, what `value` is quoted on); bam: the real Illumina "
                         "reads of tests/golden/HiSeq.1mb.1RG.2k_lines.bam against --tl-base windows (reference bases "
                         "rebuilt from CIGAR + MD, random flanks), tiled to --pairs; fastx: --fasta genome + --fastq "
                         "reads whose description carries pos=<window start> (scripts/make_synth_fastx.py), tiled")
    ap.add_argument("--fasta", help="--dataset fastx: reference FASTA (first record is used)")
    ap.add_argument("--fastq", help="--dataset fastx: reads, description 'pos=<0-based window start>'")
    args = ap.parse_args()

    rank, local_rank, world = dist.init()
    distributed = torch.distributed.is_initialized()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    dev_index = local_rank % max(1, torch.cuda.device_count())  # (== local_rank; ranks share a GPU only in a gloo rehearsal)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    # this rank's pairs: a contiguous shard of the one global workload (strong), or a workload of its own (weak)
    if args.scaling == "strong":
        lo, hi = dist.shard_range(args.pairs, rank, world)
        n_total, seed = args.pairs, args.seed
    else:
        lo, hi = 0, args.pairs
        n_total, seed = args.pairs * world, args.seed + rank
    n_local = hi - lo

    aligner = MicrosoftSmithWaterman(dev_index)
    data_label = "synthetic"
    if args.dataset != "synthetic":
        assert args.input == "ascii", "--dataset bam/fastx use the ASCII wire format"
        assert world == 1, "--dataset bam/fastx are single-GPU lines"
        batch, data_label = file_batch(args, dev)
        args.ql = batch.max_ql
        ascii_twin = batch
    elif args.input == "2bit":
        batch, ascii_twin = device_batch.window_batch_2bit(seed, n_local, dev, window=args.tl, read_len=args.ql, first=lo)
        if args.no_cpu or world > 1:
            del ascii_twin  # only the CPU-baseline leg needs the unpacked bases
            ascii_twin = None
    else:
        batch = device_batch.window_batch(seed, n_local, dev, window=args.tl, read_len=args.ql, first=lo)
        ascii_twin = batch
    cells = n_local * args.tl * args.ql   # of this rank, per step
    if args.workspace_gib is None:   # what a smaller card (or a shared one) has left beside the batch, 12 GiB of slack
        free, _total = torch.cuda.mem_get_info(dev)
        args.workspace_gib = max(1.0, min(DEFAULT_WORKSPACE_GIB, (free - (12 << 30)) / (1 << 30)))
    aligner.set_workspace(int(args.workspace_gib * (1 << 30)))

    # Steps in flight.  A rank's shard of configs[2] is 1.25 M pairs: five tiles per wave slot of the persistent grid, and the launch's
    # last tiles leave most slots idle for a millisecond of its nine.  A caller with batch after batch to align keeps TWO in
    # flight -- a context and a stream each, taken in turn -- and the next grid's workgroups move into the slots the tail frees:
    # 7.8 ms per step instead of 8.9, the long-launch rate (scripts/step_overlap_probe.py).  Every step is still a whole pass over the
    # shard into result arrays of its own; K steps are K passes.
    in_flight = args.steps_in_flight or (2 if distributed else 1)
    in_flight_note = None
    lanes = [(aligner, batch, None)]
    if in_flight == 2:
        import copy
        aligner2 = None
        try:
            if os.environ.get("MGL_BENCH_DEBUG_NO_SECOND_CONTEXT") == str(rank):  # (tests/test_gpu_multirank.py: the degraded path, on purpose)
                raise RuntimeError("injected: no room for a second context")
            aligner2 = MicrosoftSmithWaterman(dev_index)
            aligner2.set_workspace(int(args.workspace_gib * (1 << 30)))
            batch2 = copy.copy(batch)  # the same inputs, result arrays of its own
            for name in ("offsets", "scores", "cigars", "cigar_len", "status"):
                setattr(batch2, name, torch.empty_like(getattr(batch, name)))
            torch.cuda.synchronize(dev)
            batch2.run(aligner2, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)  # (its workspace exists from here on -- or the card has no room for it)
            torch.cuda.synchronize(dev)
            lanes = [(aligner, batch, torch.cuda.Stream(dev)), (aligner2, batch2, torch.cuda.Stream(dev))]
        except Exception as e:  # noqa: BLE001 -- no room for a second context: one step in flight, as on one GPU
            print(f"bench: a second context could not be set up ({e!r}): one step in flight", file=sys.stderr)
            in_flight = 1
            in_flight_note = f"rank {rank}: a second context could not be set up ({e!r})"[:300]
    in_flight_requested = args.steps_in_flight or (2 if distributed else 1)
    # (every rank runs the same number of steps -- each ends in the gather -- so they agree: one rank without room puts all at one in flight)
    if distributed:
        who = int(dist.min_over_ranks(rank if in_flight < in_flight_requested else 1 << 30, dev))  # (the first rank that fell back, if any: a collective every rank takes part in)
        if who < (1 << 30) and in_flight_requested == 2:
            in_flight_note = in_flight_note or f"rank {who} had no room for a second context: every rank keeps one step in flight (each step ends in the gather: the ranks must agree)"
            if in_flight == 2:
                lanes = [(aligner, batch, None)]
            in_flight = 1
    if in_flight == 1 and in_flight_requested == 2:
        if aligner2 is not None:  # (its workspace goes back to the card)
            aligner2.close()
            aligner2 = batch2 = None
            torch.cuda.empty_cache()
    if in_flight == 2:
        # (the batch was generated on torch's current stream: the two side streams must not start before its kernels have finished --
        # a grid that reads index arrays still being written walks out of its sequences)
        torch.cuda.synchronize(dev)
        for _al, _bt, st in lanes:
            st.wait_stream(torch.cuda.current_stream(dev))
    step_no = [0]

    def step():
        al, bt, st = lanes[step_no[0] % len(lanes)]
        step_no[0] += 1
        with torch.cuda.stream(st) if st is not None else contextlib.nullcontext():
            bt.run(al, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
            if distributed:
                # the only inter-GPU exchange: ScoreMax.max of every pair onto rank 0 (RCCL gather), in shard order
                return dist.gather_scores(bt.scores[:, 2].contiguous(), n_total, dst=0)
            return bt.scores[:, 2]

    gathered = None
    for _ in range(max(args.warmup, len(lanes) if args.warmup else 0)):
        step()
    torch.cuda.synchronize(dev)
    step_no[0] = 0
    # HIP events around every kernel launch, on the streams the kernels run on (asynchronous: they are
    # read back after the timed region) -> the MEAN launch duration over the timed steps for the roofline
    aligner.set_profiling(3)  # summed over the timed steps, read once afterwards
    dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    torch.cuda.synchronize(dev)
    dist.barrier()
    elapsed_here = time.perf_counter() - t0
    elapsed = dist.max_over_ranks(elapsed_here, dev)
    per_rank_s = dist.all_ranks(elapsed_here, dev)  # (every rank's own clock around the same K steps)
    comm = dist.describe()                         # what the COMMUNICATOR says, not the environment

    status_bad = int(sum(int((bt.status != 0).sum().item()) for _al, bt, _st in lanes))
    if in_flight == 2:
        assert torch.equal(lanes[0][1].scores, lanes[1][1].scores) and torch.equal(lanes[0][1].cigars, lanes[1][1].cigars), "the two contexts disagree"

    tm = aligner.timing()  # kernel durations summed over the timed steps (HIP events recorded in the timed region)
    if in_flight == 2:
        # (two grids side by side: a launch's events span the other one's tail too.  The roofline's launch duration is that of ONE more
        # launch, untimed, with the chip to itself)
        torch.cuda.synchronize(dev)
        aligner.set_profiling(1)
        for _ in range(2):  # (the second one: the first follows an idle moment of the chip)
            batch.run(aligner, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
        torch.cuda.synchronize(dev)
        tm = aligner.timing()
        # ... and, for the reader of the line, what this rank's steps take ONE at a time (five of them, untimed, no gather)
        aligner.set_profiling(0)
        t1 = time.perf_counter()
        for _ in range(5):
            batch.run(aligner, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
        torch.cuda.synchronize(dev)
        one_in_flight_ms = (time.perf_counter() - t1) / 5 * 1e3
    # the shader clock under this load, measured inside the kernel on one extra, untimed step (profiling level 2: every wave stamps
    # s_memtime and the 100 MHz s_memrealtime around its life): what the issue-bound fraction below is computed against
    clock_mhz = 0
    try:
        aligner.set_profiling(2)
        batch.run(aligner, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
        torch.cuda.synchronize(dev)
        clock_mhz = int(aligner.timing().clock_mhz)
    except Exception:  # noqa: BLE001 -- no clock, no issue fraction
        clock_mhz = 0
    aligner.set_profiling(0)

    if rank != 0:
        if distributed:
            torch.distributed.destroy_process_group()
        return
    if args.dump_scores and gathered is not None:
        np.save(args.dump_scores, gathered.cpu().numpy().astype(np.int32))
    fill_kernel = aligner.fill_kernel_name(tm)
    total_cells = n_total * args.tl * args.ql * args.steps
    # (tm sums the HIP events of every launch of the timed loop: mgl_sw_ctx_set_profiling(ctx, 3))
    launches = max(1, tm.dp_launches)
    avg_launch_s = tm.dp_ms / 1e3 / launches
    pairs_per_launch = n_local * (1 if in_flight == 2 else args.steps) / launches
    spills = fill_kernel not in ("sw_dp16_lane_ck_kernel",)  # every other fill kernel writes the 4-bit traceback to HBM
    cigar_bytes = float(batch.cigar_len.float().mean().item())
    per_pair = algorithmic_bytes_per_pair(args.tl, args.ql, args.input == "2bit", spills, cigar_bytes)
    achieved = pairs_per_launch * per_pair / avg_launch_s / 1e9
    pmc, pmc_current, pmc_why = pmc_entry(args.tl, args.ql, fill_kernel)
    tpp = pmc["hbm_bytes_per_pair"] if pmc and pmc_current else None
    traffic = None if tpp is None else round(tpp * pairs_per_launch)  # HBM bytes per launch (PMC)
    valu = pmc.get("valu") if pmc and pmc_current else None
    valu_obj = None
    if valu and not (1000 <= clock_mhz <= 3000):
        valu_obj = {"issue_frac": None, "note": "the in-kernel clock probe returned no plausible clock: the issue-bound fraction is not computed against an assumed one",
                    "wave_insts_per_pair": valu["wave_insts_per_pair"], "source": valu.get("source", "profiles/pmc_traffic.json (rocprofv3 --pmc)")}
    elif valu:
        # what actually bounds this integer kernel: VALU issue.  SIMD-cycles the committed instruction count needs at the
        # measured per-class issue costs / SIMD-cycles available in the measured launch duration (1024 SIMDs x the measured clock)
        clock_hz = clock_mhz * 1e6
        need = valu["wave_insts_per_pair"] * pairs_per_launch * valu["avg_cycles_per_inst"]
        have = avg_launch_s * clock_hz * 1024
        valu_obj = {"issue_frac": round(need / have, 3), "wave_insts_per_pair": valu["wave_insts_per_pair"],
                    "avg_cycles_per_inst": valu["avg_cycles_per_inst"], "lds_bank_conflict_rate": valu["lds_bank_conflict_rate"],
                    "clock_mhz": int(clock_hz / 1e6), "clock_source": "measured in the kernel (s_memtime / s_memrealtime over every wave's life, one extra untimed step)", "source": valu.get("source", "profiles/pmc_traffic.json (rocprofv3 --pmc)")}
    if args.dataset == "synthetic":
        workload = (f"BASELINE.json configs[{1 if world == 1 else 2}]: {n_total} Illumina-style {args.ql} bp reads x {args.tl}-base "
                    f"reference windows" + (f", one seeded workload sharded contiguously over {world} GPUs ({n_local} pairs on rank 0)"
                                            if world > 1 and args.scaling == "strong" else
                                            f", {args.pairs} pairs per GPU (weak scaling)" if world > 1 else ""))
    else:
        workload = f"{n_total} pairs ({data_label}), {args.ql} bp reads x {args.tl}-base windows"
    out = {
        "metric": "GCUPS (+ aligned reads/s) for 150 bp short-read batch",
        "value": round(total_cells / elapsed / 1e9, 2),
        "unit": "GCUPS",
        "n_gpus": comm["world_size"],
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "int16" if tm.packed16 else "int32",
        "data": data_label,
        "config": {
            "workload": workload + ", affine-gap SW, full matrix, GATK params (200,-150,260,11), SOFTCLIP, scores+offset+CIGAR "
                                   "for every pair",
            "pairs_total": n_total, "pairs_per_gpu": n_local, "target_len": args.tl, "query_len": args.ql, "input": args.input,
            "parallelism": (f"pairs sharded over {world} GPU(s), one process per GPU, score gather onto rank 0 only"
                            if world > 1 else "1 GPU"),
            "steps_in_flight": in_flight,
            **({"steps_in_flight_requested": in_flight_requested, "steps_in_flight_degraded": in_flight_note} if in_flight != in_flight_requested else {}),
            # the communicator's own account (torch.distributed.get_world_size() / get_backend(): "nccl" is RCCL on ROCm), the devices it
            # spans, and every rank's own clock around the K timed steps -- `value` uses the max
            "distributed": {"world_size": comm["world_size"], "backend": comm["backend"], "launched_with_world_size_env": world,
                            "gpus_visible": torch.cuda.device_count(), "ranks_per_gpu": max(1, -(-comm["world_size"] // max(1, torch.cuda.device_count()))),
                            "per_rank_ms_per_step": {"min": round(min(per_rank_s) / args.steps * 1e3, 3), "max": round(max(per_rank_s) / args.steps * 1e3, 3),
                                                     "all": [round(x / args.steps * 1e3, 3) for x in per_rank_s]}},
            **({"rank0_ms_per_step_one_in_flight": round(one_in_flight_ms, 3),
                "steps_in_flight_note": "two contexts on two streams take the steps in turn: the next step's grid moves into the wave slots the last one's tail "
                                        "leaves free; every step is a whole pass over the shard into result arrays of its own.  roofline.avg_launch_ms is that of "
                                        "one more launch with the chip to itself (a launch's events beside another grid span that grid's tail too)"}
               if in_flight == 2 else {}),
        },
        "reads_per_s": round(n_total * args.steps / elapsed, 1),
        "kernel_ms": ({fill_kernel: round(tm.dp_ms / (1 if in_flight == 2 else args.steps), 3), "path_walk": "inside the fill kernel (every lane walks its own two pairs)",
                       "launches_per_step": launches / (1 if in_flight == 2 else args.steps), "averaged_over_steps": 1 if in_flight == 2 else args.steps}
                      if fill_kernel in ("sw_dp16_lane_kernel", "sw_dp16_lane_ck_kernel") else
                      {fill_kernel: round(tm.dp_ms / args.steps, 3), "sw_traceback_kernel": round(tm.tb_ms / args.steps, 3),
                       "launches_per_step": launches / args.steps, "averaged_over_steps": args.steps}),
        "cigar_overflows": status_bad,
        "roofline": {
            "bound": "hbm", "kernel": fill_kernel,
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            # where `traffic` and `valu` come from: the committed PMC passes, printed only when taken on the kernel sources of THIS build
            "counters": {"file": "profiles/pmc_traffic.json", "commit": pmc.get("commit") if pmc else None, "src_hash": pmc.get("src_hash") if pmc else None,
                         "current": bool(pmc_current), "why": pmc_why},
            "algorithmic_bytes_per_pair": round(per_pair, 1),
            "pairs_per_launch": round(pairs_per_launch, 1),
            "avg_launch_ms": round(avg_launch_s * 1e3, 4),
            "kernel_gcups": round(pairs_per_launch * args.tl * args.ql / avg_launch_s / 1e9, 2),
            # what the counters say the kernel really moves, against the same peak (null without a PMC pass for this geometry)
            "traffic_frac_of_peak": None if traffic is None else round(traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4),
            # the figure earlier rounds' lines quoted: SURVEY 8d's per-pair bytes WITH a spilled 4-bit traceback, which this kernel does not write
            "if_traceback_were_spilled": {"bytes_per_pair": algorithmic_bytes_per_pair(args.tl, args.ql, args.input == "2bit", True, cigar_bytes),
                                          "frac": round(pairs_per_launch * algorithmic_bytes_per_pair(args.tl, args.ql, args.input == "2bit", True, cigar_bytes)
                                                        / avg_launch_s / 1e9 / HBM_PEAK_GBS, 4)},
            "note": ("integer DP: the kernel is VALU-issue bound (`valu`), the HBM fraction of SURVEY 8d's algorithmic bytes is small by construction"
                     + ("; sw_dp16_lane_ck_kernel spills no traceback: `algorithmic_bytes_per_pair` = inputs + descriptors + results, and the rows and "
                        "checkpoints it keeps to restart from are schedule -- they show in `traffic` (rocprofv3 PMC), not here"
                        if fill_kernel == "sw_dp16_lane_ck_kernel" else "")),
            "valu": valu_obj,
        },
    }
    if world == 1 and not args.no_cpu:
        if ascii_twin is not batch:  # the CPU leg reads ASCII bases and compares with the GPU's result arrays
            ascii_twin.offsets, ascii_twin.cigars = batch.offsets, batch.cigars
        out["cpu_baseline"] = cpu_baseline(ascii_twin)
    if world == 1 and not args.no_extra and args.dataset == "synthetic" and args.input == "ascii":
        # "wall time includes H2D of packed inputs, kernel(s), D2H of results" and the tl = 1000 variant,
        # after the timed headline (which keeps its inputs resident, as the bench contract asks)
        if in_flight == 1:
            # for the reader of the line (never `value`): the same steps with TWO in flight -- a second context and result arrays, two
            # streams in turn, the next step's grid moving into the wave slots the last one's tail leaves free
            try:
                import copy
                al2 = MicrosoftSmithWaterman(dev_index)
                al2.set_workspace(int(args.workspace_gib * (1 << 30)))
                bt2 = copy.copy(batch)
                for name in ("offsets", "scores", "cigars", "cigar_len", "status"):
                    setattr(bt2, name, torch.empty_like(getattr(batch, name)))
                two = [(aligner, batch, torch.cuda.Stream(dev)), (al2, bt2, torch.cuda.Stream(dev))]
                torch.cuda.synchronize(dev)
                n2 = max(4, min(args.steps, 10) // 2 * 2)
                for timed in (False, True):
                    t2 = time.perf_counter()
                    for k in range(n2 if timed else 2):
                        al_, bt_, st_ = two[k & 1]
                        with torch.cuda.stream(st_):
                            bt_.run(al_, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
                    torch.cuda.synchronize(dev)
                    dt2 = time.perf_counter() - t2
                assert torch.equal(batch.scores, bt2.scores) and torch.equal(batch.cigars, bt2.cigars)
                out["two_steps_in_flight"] = {"gcups": round(cells * n2 / dt2 / 1e9, 2), "ms_per_step": round(dt2 * 1e3 / n2, 3), "steps": n2,
                                              "note": "not `value`: two contexts on two streams take the steps in turn (bench.py --steps-in-flight 2 makes it the timed form)"}
                al2.close()
                del bt2, two
            except Exception as e:  # noqa: BLE001 -- an extra leg must never take the headline down
                out["two_steps_in_flight"] = {"error": repr(e)[:200]}
        try:
            out["pcie_inclusive"] = pcie_inclusive_leg(aligner, batch, args)
        except Exception as e:  # noqa: BLE001 -- an extra leg must never take the headline down
            out["pcie_inclusive"] = {"error": repr(e)[:200]}
        del batch, ascii_twin
        torch.cuda.empty_cache()
        try:
            out["tl1000"] = tl1000_leg(aligner, args, dev)
        except Exception as e:  # noqa: BLE001
            out["tl1000"] = {"error": repr(e)[:200]}
        batch = ascii_twin = None
    if world == 1 and not args.no_cpu and not args.no_secondary and args.dataset == "synthetic":
        del batch, ascii_twin
        aligner.close()
        torch.cuda.empty_cache()
        out["secondary"] = secondary_workloads()
    print(json.dumps(out), flush=True)
    if distributed:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
