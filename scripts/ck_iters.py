"""Experiment: blocks per wave / per pair of the checkpointed lane kernel (MGL_SW_CK_DEBUG=4 writes them into status[])."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mgl_amd import device_batch
from mgl_amd.smithwaterman import MicrosoftSmithWaterman
n, tl, ql = 262144, int(sys.argv[1]) if len(sys.argv) > 1 else 256, 150
b = device_batch.window_batch(42, n, torch.device("cuda", 0), window=tl, read_len=ql)
a = MicrosoftSmithWaterman(0)
a.set_lane_kernel(2)
b.run(a); torch.cuda.synchronize()
st = b.status.cpu().numpy()
it, mine = st & 0xffff, st >> 16
print("wave iterations: mean %.2f max %d; rounds per wave: mean %.2f max %d" % (it.mean(), it.max(), mine.mean(), mine.max()))
import numpy as np
print("hist per pair", np.bincount(mine)[:40])
print("hist per wave", np.bincount(it[::128])[:40])
