"""A/B of the headline call between two builds of the library ON ONE BOX (boxes differ by 2-3 %, so a comparison across gpurun
calls cannot tell a 2 % regression from the box): each .so is opened with plain ctypes (no mgl_amd._lib: an older build fails its
ABI check by design), the same device-resident seed-42 batch goes through mgl_sw_align_batch_device, alternating A B A B.
python scripts/ab_headline.py libA.so libB.so [pairs] [rounds]"""
import ctypes as C
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch

from mgl_amd import device_batch

paths = sys.argv[1:3]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2_097_152
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 6
dev = torch.device("cuda", 0)
b = device_batch.window_batch(42, n, dev, window=256, read_len=150)
st = torch.cuda.current_stream(dev)


def opened(path):
    L = C.CDLL(path)
    L.mgl_sw_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.mgl_sw_ctx_set_workspace.argtypes = [C.c_void_p, C.c_int64]
    L.mgl_sw_align_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    ctx = C.c_void_p()
    assert L.mgl_sw_ctx_create(0, C.byref(ctx)) == 0
    if os.environ.get("WS_GIB"):
        assert L.mgl_sw_ctx_set_workspace(ctx, int(os.environ["WS_GIB"]) << 30) == 0

    def run():
        rc = L.mgl_sw_align_batch_device(ctx, st.cuda_stream, n, b.targets.data_ptr(), b.t_off.data_ptr(), b.queries.data_ptr(), b.q_off.data_ptr(),
                                         256, 150, 200, -150, 260, 11, 1, b.offsets.data_ptr(), b.scores.data_ptr(), b.cigars.data_ptr(), b.cigar_stride,
                                         b.cigar_len.data_ptr(), b.status.data_ptr(), 1)
        assert rc == 0, rc
    return L.mgl_sw_version(), run


libs = [opened(p) for p in paths]
ref = None
for ver, run in libs:
    run(); run(); torch.cuda.synchronize()
    got = (b.offsets.clone(), b.scores.clone(), b.cigars.clone())
    if ref is not None:
        assert all(torch.equal(x, y) for x, y in zip(ref, got)), "the two builds disagree"
    ref = got
times = [[] for _ in libs]
for r in range(rounds):
    for k, (ver, run) in enumerate(libs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            run()
        torch.cuda.synchronize()
        times[k].append((time.perf_counter() - t0) / 10)
cells = n * 256 * 150
for k, (ver, _) in enumerate(libs):
    t = sorted(times[k])
    print(f"{os.path.basename(paths[k])} (MGL_SW_VERSION {ver}): median {cells/t[len(t)//2]/1e9:.0f} GCUPS, best {cells/t[0]/1e9:.0f}, worst {cells/t[-1]/1e9:.0f} "
          f"over {rounds} rounds of 10 calls, {n} pairs 256x150, identical results", flush=True)
