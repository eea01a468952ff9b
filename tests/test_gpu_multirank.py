"""The N > 1 path of bench.py end to end on ONE GPU: two ranks launched by torch.distributed.run share the device and
exchange over gloo (RCCL refuses two ranks on one GPU; MGL_DIST_BACKEND=gloo, mgl_amd/dist.py) -- shards, seeds, barrier,
max-over-ranks timing, the score gather onto rank 0 and the single JSON line are the code the 8-GPU run executes.
BASELINE.json configs[2]: ONE seeded workload sharded contiguously; the gathered score vector must be the one a single
rank computes for the whole workload, and the CPU checker's on a sample.  Also the library-level multi-device entry
(mgl_sw_align_batch_multi) with two contexts / host threads on the one GPU."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = 200_000


def _bench(nproc, extra, tmp_path, tag, backend="gloo"):
    dump = str(tmp_path / f"scores_{tag}.npy")
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MGL_DIST_BACKEND=backend, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--pairs", str(PAIRS), "--steps", "2",
           "--warmup", "1", "--no-secondary", "--no-extra", "--no-cpu", "--dump-scores", dump] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]      # rank 0 alone prints the line
    return json.loads(lines[0]), np.load(dump)


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_over_gloo(tmp_path):
    import torch

    import oracle_lib as ol
    from mgl_amd import device_batch

    d2, s2 = _bench(2, [], tmp_path, "n2")
    assert d2["n_gpus"] == 2 and d2["steps"] == 2 and d2["warmup"] == 1 and d2["scaling"] == "strong"
    assert d2["value"] > 0 and d2["unit"] == "GCUPS"
    assert d2["config"]["pairs_total"] == PAIRS and d2["config"]["pairs_per_gpu"] == PAIRS // 2
    assert d2["cigar_overflows"] == 0
    assert "cpu_baseline" not in d2                # timed on rank 0 at N = 1 only
    assert d2["roofline"]["bound"] == "hbm" and 0 < d2["roofline"]["frac"] < 1
    # the same global workload on one rank: identical gathered vector (shard order, shard seeds, padding of the gather)
    d1, s1 = _bench(1, [], tmp_path, "n1")
    assert d1["n_gpus"] == 1 and d1["config"]["pairs_total"] == PAIRS
    assert s1.shape == s2.shape == (PAIRS,) and s1.dtype == np.int32
    assert (s1 == s2).all()
    # ... and the CPU checker's ScoreMax.max on a sample of that workload, both halves of the split included
    b = device_batch.window_batch(42, PAIRS, torch.device("cuda", 0))
    idx = sorted(set(range(0, PAIRS, 397)) | {PAIRS // 2 - 1, PAIRS // 2, PAIRS - 1})
    ts, qs = b.host_pairs(idx)
    _, wsc, _ = ol.oracle_align_batch(ts, qs, (200, -150, 260, 11), ol.SOFTCLIP, nthreads=8)
    assert (s2[idx] == wsc[:, 2]).all()
    # weak scaling stays available behind the flag: every rank its own PAIRS pairs
    dw, sw_ = _bench(2, ["--scaling", "weak"], tmp_path, "weak")
    assert dw["scaling"] == "weak" and dw["config"]["pairs_total"] == 2 * PAIRS and sw_.shape == (2 * PAIRS,)
    assert (sw_[:PAIRS] == s1).all()              # rank 0's batch is the seed-42 workload


@pytest.mark.gpu
def test_four_ranks_two_steps_in_flight_each_and_the_degraded_path(tmp_path):
    """The shape `bench.py --gpus 8` runs in, rehearsed as far as this pool allows: FOUR ranks over gloo on the one GPU (a box admits six
    processes on its card: this one, the launcher's children -- eight ranks would be refused), each with TWO steps in flight (two contexts,
    two streams: eight contexts on the card), the line carrying what the COMMUNICATOR reports and every rank's own step time.  Then the
    same with rank 2 unable to set up its second context: every rank falls back to one step in flight (the ranks must agree: each step
    ends in the gather), and the line says so instead of a stderr print."""
    global PAIRS
    keep, PAIRS = PAIRS, 120_000
    try:
        d4, s4 = _bench(4, [], tmp_path, "n4")
        d1, s1 = _bench(1, [], tmp_path, "n1b")
        env_keep = os.environ.get("MGL_BENCH_DEBUG_NO_SECOND_CONTEXT")
        os.environ["MGL_BENCH_DEBUG_NO_SECOND_CONTEXT"] = "2"
        try:
            dd, sd = _bench(4, [], tmp_path, "n4d")
        finally:
            if env_keep is None:
                del os.environ["MGL_BENCH_DEBUG_NO_SECOND_CONTEXT"]
            else:
                os.environ["MGL_BENCH_DEBUG_NO_SECOND_CONTEXT"] = env_keep
    finally:
        PAIRS = keep
    c = d4["config"]
    assert d4["n_gpus"] == 4 and c["distributed"] == dict(c["distributed"], world_size=4, backend="gloo", launched_with_world_size_env=4, gpus_visible=1, ranks_per_gpu=4)
    assert c["steps_in_flight"] == 2 and "steps_in_flight_degraded" not in c
    pr = c["distributed"]["per_rank_ms_per_step"]
    assert len(pr["all"]) == 4 and pr["min"] == min(pr["all"]) and pr["max"] == max(pr["all"]) and abs(pr["max"] - d4["ms_per_step"]) < 0.01
    assert c["pairs_per_gpu"] == 30_000 and d4["cigar_overflows"] == 0
    assert (s4 == s1).all() and s4.shape == (120_000,)
    c = dd["config"]
    assert c["steps_in_flight"] == 1 and c["steps_in_flight_requested"] == 2 and "rank 2" in c["steps_in_flight_degraded"]
    assert (sd == s1).all()
    assert d1["config"]["distributed"]["world_size"] == 1 and d1["roofline"]["counters"]["file"] == "profiles/pmc_traffic.json"


@pytest.mark.gpu
def test_one_rank_over_rccl(tmp_path):
    """The `nccl` branch of mgl_amd/dist.py (RCCL: init_process_group(device_id=...), the gather, barrier(device_ids=...), the
    max-over-ranks all-reduce) executed once on the one GPU of this box: torchrun with ONE rank and the backend the 8-GPU run
    uses.  Its gathered score vector must be the vector of a plain one-process run of the same workload."""
    d, s_rccl = _bench(1, [], tmp_path, "rccl", backend="nccl")
    assert d["n_gpus"] == 1 and d["config"]["pairs_total"] == PAIRS and d["cigar_overflows"] == 0
    dump = str(tmp_path / "scores_plain.npy")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--pairs", str(PAIRS), "--steps", "1", "--warmup", "0",
                        "--no-secondary", "--no-extra", "--no-cpu", "--dump-scores", dump], capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    s_plain = np.load(dump)
    assert s_rccl.shape == s_plain.shape == (PAIRS,) and (s_rccl == s_plain).all()


@pytest.mark.gpu
def test_multi_device_entry_two_contexts_on_one_gpu():
    """mgl_sw_align_batch_multi with the device list (0, 0): two contexts, two host threads, shards balanced by cells --
    results in the caller's order, identical to the single-context entry and the CPU checker; per-pair status; errors."""
    import oracle_lib as ol
    from mgl_amd import _lib, smithwaterman as sw

    rng = np.random.default_rng(17)
    alpha = np.frombuffer(b"ACGT", np.uint8)
    ts, qs = [], []
    for k in range(3000):
        tl = int(rng.integers(20, 400)) * (6 if k < 300 else 1)   # a heavy head: the split is far from the middle
        ql = int(rng.integers(8, 200))
        t = alpha[rng.integers(0, 4, tl)]
        a = int(rng.integers(0, max(1, tl - ql)))
        q = np.resize(t[a:a + ql], ql).copy()
        q[rng.integers(0, ql)] = alpha[rng.integers(0, 4)]
        ts.append(t.tobytes())
        qs.append(q.tobytes())
    params = (200, -150, 260, 11)
    with sw.MultiGpuSmithWaterman([0, 0]) as m, sw.MicrosoftSmithWaterman(0) as one:
        for strategy in (ol.SOFTCLIP, ol.LEAD_INDEL):
            res = m.align_batch(ts, qs, params, strategy, cigar_stride=1024)
            first = m.last_shards()
            assert first[0] == 0 and first[2] == 3000 and 0 < first[1] < 1200 and first[1] % 8 == 0
            ref = one.align_batch(ts, qs, params, strategy, cigar_stride=1024)
            off, sc, cg = ol.oracle_align_batch(ts, qs, params, strategy, nthreads=8)
            assert (res.offsets == off).all() and (res.scores == sc).all() and list(res.cigars) == cg
            assert (ref.offsets == off).all() and list(ref.cigars) == cg
        # a uniform batch keeps its packed kernel on both shards
        us = [alpha[rng.integers(0, 4, 256)].tobytes() for _ in range(4000)]
        uq = [u[40:190] for u in us]
        res = m.align_batch(us, uq, params, ol.SOFTCLIP)
        assert (res.offsets == 40).all() and set(res.cigars) == {"150M"}
        # per-pair status instead of a failed call when one CIGAR does not fit its slot
        res, st = m.align_packed(*sw.concat(ts[:64]), *sw.concat(qs[:64]), params, ol.SOFTCLIP, cigar_stride=4, per_pair_status=True)
        assert set(st.tolist()) <= {0, _lib.ERR_CIGAR_OVERFLOW} and (st != 0).any()
        with pytest.raises(_lib.MglSwError) as e:
            m.align_packed(*sw.concat(ts[:64]), *sw.concat(qs[:64]), params, ol.SOFTCLIP, cigar_stride=4)
        assert e.value.status == _lib.ERR_CIGAR_OVERFLOW
