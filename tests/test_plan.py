"""The planner's decisions, pinned (mgl_sw_explain: no GPU needed -- a default context on a 256-CU device): the kernel, the chunking
and the streams for every BASELINE.json config and for each kernel's crossover, so that a change of a threshold shows up here and
not as a silent change of a bench line."""
import pytest

from mgl_amd import _lib

GATK = (200, -150, 260, 11)
UNIFORM, SCORE_ONLY = _lib.FLAG_UNIFORM_GEOMETRY, _lib.FLAG_SCORE_ONLY
BENCH_WS = 208 << 30   # bench.py's workspace
DP32, DP16, DP32_64, COOP, LANE16, COOP16, STRIP16, LANE16_CK, SMALL = range(9)


def lane_ck_region(tl, ql):
    """sw_device.h: lane_ck_words * 4 + lane_ck_scratch_bytes -- what one wave slot of sw_dp16_lane_ck_kernel keeps."""
    strips, blocks = (tl + 31) // 32, (ql + 31) // 32
    words = ((strips + 1) * (ql + 1) * 2 + strips * ql * 2 + strips * (blocks - 1) * 64 + 32 * 4) * 64
    return words * 4 + ((ql + 3) // 4 + strips * 8) * 2 * 64 * 4


REGION_256x150 = lane_ck_region(256, 150)


def plan(**kw):
    return _lib.explain(**kw)


def test_symbol_and_struct():
    p = plan(n=8, max_tl=10, max_ql=10)
    assert p.fill_kernel == SMALL and p.chunks == 1 and p.chunk_pairs == 8


def test_small_batches_take_one_wave_per_pair_in_one_launch():
    """sw_small.hip: up to MGL_SW_SMALL_BATCH_PAIRS pairs, targets of at most 512 rows, the matrix of scores within a workgroup's LDS."""
    p = plan(n=16, max_tl=256, max_ql=150, parameters=GATK)    # a coalesced batch of alignNative calls
    assert p.fill_kernel == SMALL and p.waves_per_pair == 1 and p.fused_walk == 1 and p.traceback == 1 and p.workspace_bytes == 0
    # (measured crossovers, scripts/small_batch_probe.py: 256 x 150 pairs, 241 us against 348 at 4 096 pairs, 464 against 363 at 8 192 with one
    # geometry promised -- 455 against 523 without)
    assert plan(n=8192, max_tl=256, max_ql=150, parameters=GATK).fill_kernel == SMALL
    assert plan(n=8193, max_tl=256, max_ql=150, parameters=GATK).fill_kernel != SMALL
    assert plan(n=5120, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM).fill_kernel == SMALL
    assert plan(n=5121, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM).fill_kernel != SMALL
    assert plan(n=16, max_tl=513, max_ql=100, parameters=GATK).fill_kernel != SMALL, "more than 512 rows"
    assert plan(n=16, max_tl=512, max_ql=400, parameters=GATK).fill_kernel != SMALL, "the scores do not fit LDS"
    assert plan(n=16, max_tl=256, max_ql=150, parameters=GATK, flags=SCORE_ONLY).fill_kernel != SMALL


def test_configs1_headline_is_one_launch_of_the_checkpointed_lane_kernel():
    for packed in (False, True):   # ASCII and the 2-bit wire format take the same kernel
        p = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, packed2=packed, workspace=BENCH_WS)
        assert p.fill_kernel == LANE16_CK and p.precision_bits == 16 and p.rows == 32
        assert p.traceback == 1 and p.fused_walk == 1, "no flags stored; every lane walks its own two pairs"
        assert p.chunks == 1 and p.chunk_pairs == 10_000_000 and p.fill_streams == 1 and p.workspace_halves == 1
        # a persistent grid: one region (WaveMem + the staged sequences, 2.0 MB at 256 x 150) per wave SLOT -- two waves per SIMD, 2 048 on
        # 256 CUs -- and a 32-byte record per pair, however many pairs the launch holds (round 3: a region per 128 pairs, 208 GiB)
        assert p.resident_waves == 256 * 8 and p.workspace_bytes_per_pair == 32
        assert p.workspace_fixed_bytes == p.resident_waves * REGION_256x150 and p.workspace_bytes == p.workspace_fixed_bytes + 32 * 10_000_000
        assert p.workspace_bytes < (5 << 30)


def test_headline_fits_a_workspace_of_8_gib_and_less():
    """The whole 10 M-pair batch is ONE launch from 6 GiB on; below that the grid shrinks (regions take at most three quarters of the
    workspace), and a workspace that cannot hold one wave per SIMD leaves the batch to the eight-pairs-per-wave kernel."""
    p = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=8 << 30)
    assert p.fill_kernel == LANE16_CK and p.chunks == 1 and p.chunk_pairs == 10_000_000 and p.resident_waves == 2048
    p = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=4 << 30)
    assert p.fill_kernel == LANE16_CK and p.chunks == 1 and p.resident_waves == (3 << 30) // REGION_256x150 and 1024 <= p.resident_waves < 2048
    p = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=2 << 30)
    assert p.fill_kernel == DP16, "fewer than 1 024 wave slots"
    # a launch of fewer tiles than the chip has slots keeps a region per tile
    p = plan(n=131_072, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=8 << 30)
    assert p.fill_kernel == LANE16_CK and p.resident_waves == 1024 and p.workspace_fixed_bytes == 1024 * REGION_256x150
    # the tl = 1000 variant: 7.4 MB per slot
    p = plan(n=2_560_000, max_tl=1000, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=24 << 30)
    assert p.fill_kernel == LANE16_CK and p.chunks == 1 and p.resident_waves == 2048


def test_configs2_one_rank_of_eight():
    p = plan(n=1_250_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=BENCH_WS)
    assert p.fill_kernel == LANE16_CK and p.chunks == 1 and p.chunk_pairs == 1_250_000


def test_configs0_plumbing_case_is_the_int32_kernel():
    p = plan(n=1000, max_tl=1000, max_ql=160, parameters=GATK)    # ragged reads, no promise, too few to sort
    assert p.fill_kernel == DP32 and p.precision_bits == 32 and p.rows == 16 and p.traceback == 0 and p.fused_walk == 0


def test_configs3_long_reads_take_the_strip_kernel():
    p = plan(n=2304, max_tl=10300, max_ql=10300, parameters=GATK, workspace=BENCH_WS)
    assert p.fill_kernel == STRIP16 and p.waves_per_pair == 4 and 20 <= p.rows <= 32 and p.precision_bits == 16
    assert p.chunk_pairs == 2304 and p.chunks == 1, "17 MB of kept rows per pair: the whole batch is one launch (three chunks measured slower)"
    p = plan(n=2304, max_tl=10300, max_ql=10300, parameters=GATK, workspace=16 << 30)
    assert p.fill_kernel == STRIP16 and p.chunks > 1 and p.workspace_halves == 2, "a smaller workspace: chunks, two halves in flight"
    # beyond the 16 384 rows that four waves' strips hold: round 4 takes the target in two passes of 512 strips (31 rows each: whole
    # bands of two strips per pass); round 3 ran such pairs on the workgroup kernel with every flag stored (967 GCUPS at 30 kb; now 4 620)
    p = plan(n=8, max_tl=31000, max_ql=30500, parameters=GATK, workspace=BENCH_WS)
    assert p.fill_kernel == STRIP16 and p.waves_per_pair == 4 and p.rows == 31 and p.traceback == 1
    # ... and queries whose bytes no longer fit the LDS carve keep the workgroup kernel
    p = plan(n=8, max_tl=31000, max_ql=70000, parameters=GATK, workspace=BENCH_WS)
    assert p.fill_kernel == COOP16


def test_tl1000_variant():
    p = plan(n=2_560_000, max_tl=1000, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=BENCH_WS)
    assert p.fill_kernel == LANE16_CK and p.chunks == 1


def test_lane_kernel_crossover():
    """scripts/kernel_crossover.py: the lane kernel overtakes the eight-pairs-per-wave kernel between 65 536 and 131 072 pairs."""
    small = plan(n=65_536, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=BENCH_WS)
    large = plan(n=131_072, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=BENCH_WS)
    assert small.fill_kernel == DP16 and small.traceback == 0 and small.fused_walk == 0
    assert large.fill_kernel == LANE16_CK
    # a workspace that cuts the batch below the crossover keeps the smaller kernel
    p = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, workspace=1 << 30)
    assert p.fill_kernel == DP16 and p.chunks > 1 and p.workspace_halves == 2


def test_rows_that_do_not_fill_strips_of_32():
    p = plan(n=1_000_000, max_tl=33, max_ql=8, parameters=GATK, flags=UNIFORM, workspace=BENCH_WS)
    assert p.fill_kernel == LANE16 and p.rows == 16, "16-row strips exist only with every flag stored"
    p = plan(n=1_000_000, max_tl=33, max_ql=8, parameters=GATK, flags=UNIFORM, packed2=True, workspace=BENCH_WS)
    assert p.fill_kernel == DP16, "... which reads ASCII only: a 2-bit batch of that shape takes the eight-pairs-per-wave kernel"


def test_score_only_hint():
    p = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM | SCORE_ONLY, workspace=BENCH_WS)
    assert p.fill_kernel == LANE16 and p.traceback == 2 and p.chunks == 1


def test_scores_beyond_16_bits_take_the_int32_kernels():
    p = plan(n=1_000_000, max_tl=256, max_ql=150, parameters=(2000, -1500, 2600, 110), flags=UNIFORM)
    assert p.fill_kernel == DP32 and p.precision_bits == 32
    p = plan(n=1_000_000, max_tl=256, max_ql=2000, parameters=(2000, -1500, 2600, 110), flags=UNIFORM)
    assert p.fill_kernel == DP32_64 and p.rows == 64


def test_mixed_geometries_are_sorted_by_the_library():
    ws = 72 << 30   # a context's default: a quarter of the card
    dev = plan(n=4_000_000, max_tl=256, max_ql=150, parameters=GATK, workspace=ws)              # device resident, no promise
    assert dev.sorted_by_library == 1 and dev.fill_kernel == LANE16_CK and dev.precision_bits == 16, "whole waves of one geometry: the lane kernel"
    host = plan(n=4_000_000, max_tl=256, max_ql=150, parameters=GATK, entry=1, workspace=ws)     # host buffers
    assert host.sorted_by_library == 2 and host.workspace_halves == 2 and host.fill_kernel == LANE16_CK
    # round 4: a sorted chunk is sized by the lane kernel's wave slots (a region each) + an area for the left-over pairs + a record per pair,
    # not by 19 KB of left-over traceback per pair: ONE chunk in 8 GiB (round 3: 21) and in 4 GiB
    for w, slots in ((8 << 30, 2048), (4 << 30, (3 << 30) // REGION_256x150)):
        p = plan(n=4_000_000, max_tl=256, max_ql=150, parameters=GATK, workspace=w)
        assert (p.sorted_by_library, p.fill_kernel, p.chunks, p.chunk_pairs, p.workspace_halves) == (1, LANE16_CK, 1, 4_000_000, 1)
        assert p.resident_waves == slots and p.workspace_bytes <= w and p.workspace_bytes_per_pair == 32
    assert (dev.chunks, dev.workspace_halves) == (1, 1)
    tight = plan(n=4_000_000, max_tl=256, max_ql=150, parameters=GATK, workspace=2 << 30)        # 2 GiB: fewer than half the chip's slots -- chunks below the lane kernel's crossover
    assert tight.sorted_by_library == 1 and tight.fill_kernel == DP16
    few = plan(n=500, max_tl=256, max_ql=150, parameters=GATK)
    assert few.sorted_by_library == 0 and few.fill_kernel == SMALL, "one wave per pair needs no common geometry"
    few = plan(n=500, max_tl=600, max_ql=150, parameters=GATK)
    assert few.sorted_by_library == 0 and few.fill_kernel == DP32


def test_a_promise_of_blocks_of_eight_does_not_keep_a_large_batch_from_the_lane_kernel():
    """MGL_SW_FLAG_GROUPED_GEOMETRY (a caller who has sorted by read length): round 3 pinned such a batch to the eight-pairs-per-wave
    kernel -- 3 013 GCUPS where the caller who promised nothing got 4 883.  Where the library's own sort would feed the lane kernel
    the promise is set aside; a batch too small for lane launches keeps it."""
    ws = 72 << 30
    grouped = _lib.FLAG_GROUPED_GEOMETRY
    big = plan(n=4_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=grouped, workspace=ws)
    none = plan(n=4_000_000, max_tl=256, max_ql=150, parameters=GATK, workspace=ws)
    assert big.fill_kernel == LANE16_CK and big.sorted_by_library == 1
    assert (big.chunk_pairs, big.chunks, big.workspace_halves) == (none.chunk_pairs, none.chunks, none.workspace_halves)
    small = plan(n=100_000, max_tl=256, max_ql=150, parameters=GATK, flags=grouped, workspace=ws)
    assert small.fill_kernel == DP16 and small.sorted_by_library == 0
    tight = plan(n=4_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=grouped, workspace=2 << 30)     # 2 GiB: chunks below the lane launches' threshold
    assert tight.fill_kernel == DP16 and tight.sorted_by_library == 0


def test_host_entries_pipeline_chunks_on_two_streams():
    a = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, entry=1, workspace=BENCH_WS)
    assert a.fill_kernel == LANE16_CK and a.fill_streams == 2 and a.workspace_halves == 2
    assert a.chunk_pairs == 256 * 8 * 128, "ASCII inputs: one round of the chip per chunk (two waves per SIMD, 128 pairs per wave)"
    b = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, entry=1, packed2=True, workspace=BENCH_WS)
    assert b.chunk_pairs == 8 * 256 * 8 * 128 and b.chunks == 11, "2-bit inputs: chunks of 1, 2, 4, 8, 8, 7, 4, 2, 1, 1 rounds and the rest (38.15 rounds in all)"
    # the same pipeline in a workspace of 8 GiB (round 3 needed 208 GiB for it: a region per 128 pairs): two launches side by side,
    # a persistent grid each, their regions at most three quarters of a half
    c = plan(n=10_000_000, max_tl=256, max_ql=150, parameters=GATK, flags=UNIFORM, entry=1, packed2=True, workspace=8 << 30)
    assert c.fill_kernel == LANE16_CK and c.chunk_pairs == b.chunk_pairs and c.chunks == 11 and c.workspace_halves == 2
    assert c.resident_waves == (3 << 30) // REGION_256x150 and c.workspace_bytes <= (8 << 30)


def test_explain_reports_what_the_call_would_refuse():
    with pytest.raises(_lib.MglSwError) as e:
        plan(n=1, max_tl=1 << 20, max_ql=1 << 20, parameters=GATK)
    assert e.value.status == _lib.ERR_UNSUPPORTED
    with pytest.raises(_lib.MglSwError):
        plan(n=0, max_tl=10, max_ql=10)
