"""The JNI exports (mgl_amd/csrc/jni_exports.cpp, pairhmm_jni_exports.cpp): this image has no JDK, so the product
libraries are built without them; here they are compiled against a minimal test-only jni.h (tests/cpp/jni_stub/, never on
the product include path) and called through a fake JNIEnv with the Java side's buffer contract
(/root/reference/src/main/java/com/microsoft/mgl/smithwaterman/MicrosoftSmithWaterman.java:66-86,
.../pairhmm/MicrosoftPairHmm.java:62-112)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "cpp", "jni_harness")
EXPORTS = [
    "Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_initNative",
    "Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_alignNative",
    "Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_doneNative",
    "Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_initNative",
    "Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_computeLikelihoodsNative",
    "Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_doneNative",
]


def build():
    from mgl_amd import _lib, pairhmm

    _lib.lib()
    pairhmm.lib()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), HARNESS])


def test_jni_exports_compile_against_the_stub_and_have_the_reference_names():
    """Type-check of both JNI files (-Wall -Wextra -Werror) and the six mangled names GATK's loader binds
    (com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman.h:15-32, com_microsoft_mgl_pairhmm_MicrosoftPairHmm.h)."""
    build()
    out = subprocess.run(["nm", "--defined-only", HARNESS], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if ln.split()}
    for name in EXPORTS:
        assert name in defined, name
    # the stub header must never be reachable from the product build
    mk = open(os.path.join(ROOT, "mgl_amd", "csrc", "Makefile")).read()
    assert "jni_stub" not in mk


def test_jni_failure_becomes_a_java_exception_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    build()
    r = subprocess.run([HARNESS], input="GATTACA TTAC 200 -150 260 11 1\n", capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("EXC java/lang/RuntimeException"), r.stdout + r.stderr
    r = subprocess.run([HARNESS, "pairhmm"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.count("EXC java/lang/RuntimeException") == 2, r.stdout + r.stderr


@pytest.mark.gpu
def test_align_native_through_a_fake_jnienv_matches_the_goldens():
    """Java_..._alignNative with direct-buffer stand-ins: reads buffer = target then query, zero-filled CIGAR buffer of
    2 * max(tl, ql) bytes, text read back the way Java's new String(bytes).trim() reads it."""
    import golden_io

    build()
    rows = golden_io.load("known") + golden_io.load("window")[:400] + [g for g in golden_io.load("shapes") if len(g.q) >= 1][:200]
    rows = [g for g in rows if b" " not in g.t and b" " not in g.q and 2 * max(len(g.t), len(g.q)) >= len(g.cigar)]
    assert len(rows) > 500
    text = "".join(f"{g.t.decode()} {g.q.decode()} {g.params[0]} {g.params[1]} {g.params[2]} {g.params[3]} {g.strategy}\n" for g in rows)
    r = subprocess.run([HARNESS], input=text, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == len(rows)
    for ln, g in zip(lines, rows):
        f = ln.split()
        assert f[0] != "EXC", ln
        assert (int(f[0]), f[1] if len(f) > 1 else "") == (g.offset, g.cigar), (ln, g.t, g.q, g.strategy)


@pytest.mark.gpu
def test_pairhmm_native_through_a_fake_jnienv():
    build()
    r = subprocess.run([HARNESS, "pairhmm"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    vals = [float(x) for x in r.stdout.split()]
    assert len(vals) == 2 and all(abs(v - (-6.022797e-01)) < 1e-5 for v in vals), r.stdout   # MicrosoftPairHmmUnitTest.java:49
