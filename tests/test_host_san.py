"""CPU sanitizers over the library's host side (GPU sanitizers are not available on this pool): sw_capi.cpp, sw_batcher.cpp
and sw_multi.cpp are built against a fake HIP runtime + a fake device backend (tests/cpp/fake_hip, fake_device.cpp -- the
CPU checker stands where the kernels stand) with -fsanitize=address,undefined and again with -fsanitize=thread, and
tests/cpp/host_san_driver.cpp drives them: chunked mixed batches (the per-chunk sort by geometry on its helper thread,
dest map, per-pair status), the lane-kernel path, the multi-device entry, 48 threads through the coalescing front-end and
through the mailbox service in front of it (sw_service.cpp against a fake grid of threads that speaks the waves' protocol)
incl. an injected device-side failure.  Also the restatement itself under ASan."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def _run(binary, env_extra):
    subprocess.check_call(["make", "-s", "-C", CPP, os.path.join(CPP, binary)])
    env = dict(os.environ, **env_extra)
    r = subprocess.run([os.path.join(CPP, binary)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "host-san ok" in r.stdout
    assert "ERROR: " not in r.stderr and "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_host_side_under_asan_ubsan():
    _run("host_san_asan", {"ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "halt_on_error=1"})


def test_host_side_under_tsan():
    # (a grid of the mailbox service lives 20 ms by default: under the thread sanitizer that is a handful of calls, and every launch of
    # the fake grid creates and joins 32 threads; 200 ms leaves the launches to the growing grid and the pauses of the driver)
    _run("host_san_tsan", {"TSAN_OPTIONS": "halt_on_error=1", "MGL_SW_SERVICE_LIFE_MS": "200"})


def test_restatement_under_asan():
    """oracle/libsw_oracle_asan.so (oracle/Makefile) through a few hundred golden records in a child interpreter with the
    ASan runtime preloaded."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "libsw_oracle_asan.so")])
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    code = (
        "import sys, ctypes as C; sys.path.insert(0, %r); import golden_io\n"
        "L = C.CDLL(%r)\n"
        "rows = golden_io.load('known') + golden_io.load('shapes')[:300] + golden_io.load('ties')[:100]\n"
        "for g in rows:\n"
        "    buf = C.create_string_buffer(12 * (len(g.t) + len(g.q) + 4)); ln = C.c_int(); off = C.c_int(); ez = (C.c_int32 * 6)()\n"
        "    rc = L.swo_align(g.t, len(g.t), g.q, len(g.q), *g.params, g.strategy, buf, len(buf), C.byref(ln), C.byref(off), ez, None)\n"
        "    assert rc == 0 and off.value == g.offset and tuple(ez) == g.score, g\n"
        "    assert g.cigar.startswith('sha1:') or buf.raw[:ln.value].decode() == g.cigar\n"
        "print('asan oracle ok', len(rows))\n" % (os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle", "libsw_oracle_asan.so")))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run(["python3", "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "asan oracle ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
