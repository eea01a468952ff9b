"""The N > 1 path of bench.py end to end on ONE GPU: two ranks launched by torch.distributed.run share the device and
exchange over gloo (RCCL refuses two ranks on one GPU; MGL_DIST_BACKEND=gloo, mgl_amd/dist.py) -- shards, seeds, barrier,
max-over-ranks timing, the score gather onto rank 0 and the single JSON line are the code the 8-GPU run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_share_the_gpu_over_gloo():
    import socket

    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MGL_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--pairs", "200000", "--steps", "2",
           "--warmup", "1", "--no-secondary"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]      # rank 0 alone prints the line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["unit"] == "GCUPS" and d["config"]["pairs_per_gpu"] == 200000
    assert d["cigar_overflows"] == 0
    assert "cpu_baseline" not in d                # timed on rank 0 at N = 1 only
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
