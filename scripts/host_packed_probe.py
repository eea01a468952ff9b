"""PCIe-inclusive rate of the host entries on the bench workload: ASCII pageable / ASCII registered / 2-bit registered
(bench.py's pcie_inclusive leg alone).  python scripts/host_packed_probe.py [pairs]   (MGL_SW_LANE_CHUNK_ROUNDS: chunk size experiment)"""
import argparse, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
a = MicrosoftSmithWaterman(0)
a.set_workspace(int(bench.DEFAULT_WORKSPACE_GIB * (1 << 30)))
b = device_batch.window_batch(42, n, dev)
b.run(a); torch.cuda.synchronize()
args = argparse.Namespace(steps=int(sys.argv[2]) if len(sys.argv) > 2 else 2, seed=42, workspace_gib=bench.DEFAULT_WORKSPACE_GIB)
out = bench.pcie_inclusive_leg(a, b, args)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: v[kk] for kk in ("ms_per_step", "gcups", "mismatches_vs_headline")}) for k, v in out.items()
                  if k in ("ms_per_step", "gcups", "registered", "packed_2bit", "registered_error")}))
