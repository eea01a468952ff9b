"""CPU estimate of how many block recomputations the checkpointed lane kernel's walk needs per wave (sw_dp16_lane_ck.hip, pass 2):
the bench workload's pairs through the CPU checker, each path replayed against the 16-row x 32-column block grid with the rule of
PathWalk::verify_apply (a stretch up to the next recorded row is free when it is purely diagonal).  python scripts/walk_rounds_sim.py [pairs]"""
import os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import oracle_lib as ol
from mgl_amd import device_batch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
device_batch.WORKLOAD_BLOCK = max(n, 1 << 14)
b = device_batch.window_batch(42, n, torch.device("cpu"))
ts, qs = b.host_pairs(list(range(n)))
off, sc, cg = ol.oracle_align_batch(ts, qs, (200, -150, 260, 11), ol.SOFTCLIP, nthreads=8)
rounds_v, rounds_b, rounds_old = [], [], []
for k in range(n):
    I, J = int(sc[k][3]), int(sc[k][4])  # max_t, max_q
    ops = [(int(a), o) for a, o in re.findall(r"(\d+)([MIDS])", cg[k])]
    while ops and ops[-1][1] == "S": ops.pop()
    moves = []  # from the end of the alignment backwards
    for ln, op in reversed(ops):
        if op == "S": break
        moves += [op] * ln
    pos = 0; nb = 0; nv = 0; blocks_old = set()
    i, j = I, J
    # old rule: every block the path touches
    ii, jj = I, J
    for mv in moves:
        blocks_old.add(((ii - 1) >> 4, (jj - 1) >> 5))
        if mv == "M": ii -= 1; jj -= 1
        elif mv == "D": ii -= 1
        else: jj -= 1
    while pos < len(moves) and i > 0 and j > 0:
        L = min(i - (((i - 1) >> 4) << 4), j)
        if all(m == "M" for m in moves[pos:pos + L]) and pos + L <= len(moves):
            pos += L; i -= L; j -= L; nv += 1
            continue
        nb += 1  # flags of block (band, column block): walk until the path leaves it
        kb, cb = (i - 1) >> 4, (j - 1) >> 5
        while pos < len(moves) and i > 0 and j > 0 and (i - 1) >> 4 == kb and (j - 1) >> 5 == cb:
            mv = moves[pos]; pos += 1
            if mv == "M": i -= 1; j -= 1
            elif mv == "D": i -= 1
            else: j -= 1
    rounds_v.append(nv); rounds_b.append(nb); rounds_old.append(len(blocks_old))
rv, rb, ro = np.array(rounds_v), np.array(rounds_b), np.array(rounds_old)
w = n // 128
print(f"{n} pairs: per pair  verify rounds mean {rv.mean():.2f}  block rounds mean {rb.mean():.3f} (old rule: {ro.mean():.2f})")
print(f"per wave of 128: verify max mean {rv[:w*128].reshape(w,128).max(1).mean():.2f}  block max mean {rb[:w*128].reshape(w,128).max(1).mean():.2f} "
      f"(old rule {ro[:w*128].reshape(w,128).max(1).mean():.2f}); histogram of block rounds per pair: {np.bincount(rb)[:8]}")
