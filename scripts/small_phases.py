"""Where a wave of sw_small_kernel spends its time: needs the -DMGL_SMALL_PHASES build (bash scripts/build_variant.sh small_phases
sw_small.hip -DMGL_SMALL_PHASES; MGL_SW_LIB=build/variants/lib_small_phases.so python scripts/small_phases.py [pairs] [tl] [ql] [host]).
`host`: the batch goes through the host entry (pinned staging buffer read in place over the link), else it is device resident."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from mgl_amd import _lib, device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
tl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ql = int(sys.argv[3]) if len(sys.argv) > 3 else 150
host = len(sys.argv) > 4
a = MicrosoftSmithWaterman(0)
a.set_small_kernel(2)
b = device_batch.window_batch(42, n, torch.device("cuda", 0), window=tl, read_len=ql)
if host:
    td, toff, qd, qoff = (x.cpu().numpy() for x in (b.targets, b.t_off, b.queries, b.q_off))
    for _ in range(20):
        a.align_packed(td, toff, qd, qoff, (200, -150, 260, 11))
else:
    for _ in range(20):
        b.run(a); torch.cuda.synchronize()
assert a.timing().fill_kernel == 8
_lib.lib().mgl_small_phases_dump()
