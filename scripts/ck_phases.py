"""Where a wave of sw_dp16_lane_ck_kernel spends its time: needs the -DMGL_CK_PHASES build (scripts/build_variant.sh phases
sw_dp16_lane_ck.hip -DMGL_CK_PHASES; MGL_SW_LIB=build/variants/lib_phases.so python scripts/ck_phases.py [pairs] [tl] [ql])."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import _lib, device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_097_152
tl = int(sys.argv[2]) if len(sys.argv) > 2 else 256
ql = int(sys.argv[3]) if len(sys.argv) > 3 else 150
b = device_batch.window_batch(42, n, torch.device("cuda", 0), window=tl, read_len=ql)
a = MicrosoftSmithWaterman(0)
a.set_workspace(128 << 30)
b.run(a); torch.cuda.synchronize()
_lib.lib().mgl_ck_phases_dump()
