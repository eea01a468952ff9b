"""CPU-side checks of the C-ABI library and the host mirror: the library loads, exports every
symbol include/mgl_sw.h declares, and fails loudly (never silently computes on the CPU) when no
GPU is present.  No compute calls are made without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from mgl_amd import _lib, smithwaterman as sw, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mgl_sw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mgl_sw_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = header_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/mgl_sw.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    assert L.mgl_sw_version() >= 100


def test_library_exports_nothing_undeclared():
    """Every dynamic mgl_sw_* / mgl_pairhmm_* symbol of the two libraries is declared in include/ (library-internal
    helpers have hidden visibility)."""
    import subprocess

    from mgl_amd import pairhmm

    for path, header, prefix in ((_lib.LIB_PATH, "mgl_sw.h", "mgl_sw_"), (pairhmm.LIB_PATH, "mgl_pairhmm.h", "mgl_pairhmm_")):
        _lib.lib()
        pairhmm.lib()
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith(prefix)})
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        declared = sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, text)))
        assert exported == declared, (set(exported) ^ set(declared))


def test_header_constants_match_reference_codes():
    text = open(os.path.join(ROOT, "include", "mgl_sw.h")).read()
    # sw_common.h:22-25,33
    for name, val in (("MGL_SW_OS_SOFTCLIP", "0x01"), ("MGL_SW_OS_INDEL", "0x02"), ("MGL_SW_OS_LEAD_ID", "0x04"),
                      ("MGL_SW_OS_IGNORE", "0x08"), ("MGL_SW_NEG_INF", "(-0x40000000)")):
        assert re.search(rf"#define {name} {re.escape(val)}", text), name
    assert [int(s) for s in sw.SWOverhangStrategy] == [1, 2, 4, 8]
    assert C.sizeof(_lib.Score) == 24


def test_param_normalisation_matches_jni_boundary():
    # ..._MicrosoftSmithWaterman.cpp:51-55
    L = _lib.lib()
    for given in ((200, -150, 260, 11), (-200, 150, -260, -11), (200, 150, 260, 11), (200, -150, -260, -11)):
        v = [C.c_int(x) for x in given]
        L.mgl_sw_normalize_params(*[C.byref(x) for x in v])
        assert [x.value for x in v] == [200, -150, 260, 11]


def test_strerror_and_limits():
    L = _lib.lib()
    assert L.mgl_sw_strerror(0) == b"ok"
    assert b"HIP" in L.mgl_sw_strerror(_lib.ERR_DEVICE)
    assert 1000 < L.mgl_sw_max_lds_query_len() < 100000
    assert L.mgl_sw_max_query_len() == 1 << 24


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _lib.lib()
    assert L.mgl_sw_device_count() == 0
    h = C.c_void_p()
    assert L.mgl_sw_ctx_create(0, C.byref(h)) == _lib.ERR_DEVICE
    assert not sw.MicrosoftSmithWaterman().load()
    with pytest.raises(_lib.MglSwError) as e:
        sw.align(b"ACGT", b"ACGT")
    assert e.value.status == _lib.ERR_DEVICE
    with pytest.raises(_lib.MglSwError):
        sw.backtrack_matrix(b"ACGT", b"ACGT")


def test_bad_arguments_rejected_before_any_device_work():
    L = _lib.lib()
    buf = C.create_string_buffer(16)
    ln, off = C.c_int(), C.c_int()
    assert L.mgl_sw_align(b"", 0, b"A", 1, 1, -1, 1, 1, 1, buf, 16, C.byref(ln), C.byref(off), None) == _lib.ERR_BAD_ARG
    assert L.mgl_sw_align(b"A", 1, b"A", 1, 1, -1, 1, 1, 3, buf, 16, C.byref(ln), C.byref(off), None) == _lib.ERR_BAD_ARG
    assert L.mgl_sw_ctx_set_workspace(None, 1 << 30) == _lib.ERR_BAD_ARG


def test_product_never_imports_the_oracle():
    """The shipped package and the native sources must not reference oracle/ in any way."""
    pkg = os.path.join(ROOT, "mgl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "sw_oracle" not in text and "libmgl_ref" not in text, f
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower() or f.endswith(".md")


def test_concat_layout():
    d, off = sw.concat([b"ACG", b"T", b"GGTT"])
    assert d.tobytes() == b"ACGTGGTT" and off.tolist() == [0, 3, 4, 8]


def test_synthetic_workloads_are_deterministic():
    ref1, reads1 = synth.config1()
    ref2, reads2 = synth.config1()
    assert (ref1 == ref2).all() and (reads1 == reads2).all() and reads1.shape == (1000, 150)
    assert set(np.unique(reads1).tolist()) <= set(b"ACGT")
    # error model sanity: ~1 % substitutions => most reads differ from their source in a few bases
    g, ws, rd = synth.window_batch(3, 64, genome_len=1 << 14)
    assert rd.shape == (64, 150) and (ws >= 0).all() and (ws + 256 <= len(g)).all()


def test_group_by_geometry_helper():
    """mgl_sw_group_by_geometry (host only): a permutation whose front part is full blocks of eight pairs with one
    geometry each, fewer than eight leftovers per geometry behind it; both sort strategies; degenerate inputs."""
    import ctypes as C

    import numpy as np

    L = _lib.lib()
    rng = np.random.default_rng(5)
    for n, t_hi, q_hi in ((0, 1, 1), (1, 5, 5), (7, 1, 1), (8, 1, 1), (1000, 3, 4), (20011, 300, 200), (5000, 1 << 20, 1 << 12)):
        tl = rng.integers(1, t_hi + 1, n).astype(np.int32)
        ql = rng.integers(1, q_hi + 1, n).astype(np.int32)
        order = np.full(max(n, 1), -1, np.int64)
        ng = C.c_int64(-1)
        assert L.mgl_sw_group_by_geometry(n, tl.ctypes.data, ql.ctypes.data, order.ctypes.data, C.byref(ng)) == 0
        order = order[:n]
        assert sorted(order.tolist()) == list(range(n))
        g = ng.value
        assert 0 <= g <= n and g % 8 == 0
        blocks = order[:g].reshape(-1, 8)
        assert (tl[blocks] == tl[blocks][:, :1]).all() and (ql[blocks] == ql[blocks][:, :1]).all()
        key = tl[order[g:]].astype(np.int64) << 32 | ql[order[g:]]
        assert (np.diff(key) >= 0).all()                         # the rest is sorted by geometry
        if len(key):
            assert np.unique(key, return_counts=True)[1].max() < 8    # nothing left that would fill a block
        total = np.unique(tl.astype(np.int64) << 32 | ql, return_counts=True)[1] if n else np.zeros(0, np.int64)
        assert g == int((total // 8 * 8).sum())
    assert L.mgl_sw_group_by_geometry(-1, None, None, None, C.byref(C.c_int64())) == _lib.ERR_BAD_ARG


def test_shard_by_cells_helper():
    """mgl_sw_shard_by_cells (the multi-device entry's sharding rule, host only): contiguous, aligned, balanced by
    sum tl * ql rather than by pair count."""
    L = _lib.lib()
    rng = np.random.default_rng(3)
    for n, parts, align in ((0, 4, 8), (5, 4, 8), (1000, 1, 8), (1000, 8, 8), (100_003, 8, 8), (4096, 3, 1)):
        tl = rng.integers(1, 400, n)
        ql = rng.integers(1, 300, n)
        if n > 100:
            tl[: n // 4] *= 20   # a heavy head: equal pair counts would be badly unbalanced
        toff = np.concatenate([[0], np.cumsum(tl)]).astype(np.int64)
        qoff = np.concatenate([[0], np.cumsum(ql)]).astype(np.int64)
        first = np.full(parts + 1, -1, np.int64)
        assert L.mgl_sw_shard_by_cells(n, toff.ctypes.data, qoff.ctypes.data, parts, align, first.ctypes.data) == 0
        assert first[0] == 0 and first[-1] == n and (np.diff(first) >= 0).all()
        assert all(f % align == 0 for f in first[1:-1])
        if n > 1000:
            cells = np.add.reduceat((tl * ql).astype(np.float64), first[:-1]) if parts > 1 else np.array([float((tl * ql).sum())])
            assert cells.max() / cells.mean() < 1.02
    assert L.mgl_sw_shard_by_cells(-1, None, None, 2, 8, np.zeros(3, np.int64).ctypes.data) == _lib.ERR_BAD_ARG
    h = C.c_void_p()
    import torch

    if not torch.cuda.is_available():
        assert L.mgl_sw_multi_create(2, None, C.byref(h)) == _lib.ERR_DEVICE


def test_python_mirror_of_the_header_constants():
    """mgl_amd/_lib.py repeats the header's flags, status codes and fill-kernel ids: the numbers must be the header's (no GPU, no library)."""
    import re

    from mgl_amd import _lib

    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mgl_sw.h")).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"#define\s+(MGL_SW_[A-Z0-9_]+)\s+(0x[0-9a-fA-F]+|\d+)\b", text)}
    for name in ("UNIFORM_GEOMETRY", "BINARY_CIGAR", "GROUPED_GEOMETRY", "SCORE_ONLY", "SHARED_TARGET"):
        assert getattr(_lib, "FLAG_" + name) == defines["MGL_SW_FLAG_" + name], name
    flags = [defines[k] for k in defines if k.startswith("MGL_SW_FLAG_")]
    assert len(set(flags)) == len(flags) and all(f & (f - 1) == 0 for f in flags)   # one bit each
    assert _lib.ABI_VERSION == defines["MGL_SW_VERSION"]
    kernels = sorted((v, k) for k, v in defines.items() if k.startswith("MGL_SW_KERNEL_"))
    assert [v for v, _ in kernels] == list(range(len(kernels))) and len(_lib.FILL_KERNEL_NAMES) == len(kernels)
    for v, k in kernels:   # MGL_SW_KERNEL_LANE16_CK -> sw_dp16_lane_ck_kernel, ...
        assert re.search(rf"#define {k} {v}\s+/\* {re.escape(_lib.FILL_KERNEL_NAMES[v])}\b", text), (k, _lib.FILL_KERNEL_NAMES[v])
