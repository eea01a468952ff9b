"""The host entries behind a process that has created its own streams first (torch.cuda.set_device before the library's context exists)
or after (argument `early`: the context is created before torch touches the device): the runtime spreads streams over four hardware
queues, and which of the context's streams share one depends on that order.  Prints the packed / registered ASCII / pageable ASCII
PCIe-inclusive GCUPS.  python scripts/stream_order_probe.py [early]"""
import argparse, os, sys, time
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import numpy as np, torch
import bench
from mgl_amd import device_batch, dist
from mgl_amd.smithwaterman import MicrosoftSmithWaterman, GATK_PARAMETERS, SWOverhangStrategy
args = argparse.Namespace(steps=3, seed=42, pairs=10_000_000, tl=256, ql=150)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
aligner = MicrosoftSmithWaterman(0)
if len(sys.argv) > 1: aligner.set_workspace(1 << 30)   # (creates the context -- its streams -- before torch has made any)
batch = device_batch.window_batch(42, 10_000_000, dev, window=256, read_len=150, first=0)
free, _ = torch.cuda.mem_get_info(dev)
ws = max(1.0, min(bench.DEFAULT_WORKSPACE_GIB, (free - (12 << 30)) / (1 << 30)))
print("workspace GiB", ws, file=sys.stderr)
aligner.set_workspace(int(ws * (1 << 30)))
def step():
    batch.run(aligner, GATK_PARAMETERS, SWOverhangStrategy.SOFTCLIP)
    return batch.scores[:, 2]
step()
def leg(tag):
    o = bench.pcie_inclusive_leg(aligner, batch, args)
    print(tag, o["packed_2bit"]["gcups"], o["registered"]["gcups"], o["gcups"], file=sys.stderr, flush=True)
leg("A (after one run)")
