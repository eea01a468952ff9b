/*
 * mgl_sw.h -- C ABI of libmgl_sw_hip.so, the MI355X (gfx950) implementation of
 * mgl's Smith-Waterman affine-gap alignment core.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or framework
 * types.  Every entry point names the reference interface it stands in for
 * (paths relative to /root/reference/src/main/native/mgl_sw/).  Results are
 * bit-exact with the reference CPU path (score, CIGAR, offset, traceback).
 *
 * Error convention (the reference has none -- it never checks anything,
 * SURVEY.md 8b): every function returns an mgl_sw_status; 0 is success.
 * There is NO CPU fallback inside this library: without a usable HIP device
 * every compute entry point returns MGL_SW_ERR_DEVICE.
 *
 * Threading: like the reference's alignNative (stateless, re-entrant,
 * ..._MicrosoftSmithWaterman.cpp:44-71) every function may be called from any
 * thread.  An mgl_sw_ctx serialises the calls made on it; use one ctx per
 * host thread (or per GPU) for concurrency.  mgl_sw_align() goes through the
 * coalescing front-end (mgl_sw_set_coalescing, on by default); with that
 * switched off it uses a thread-local ctx.
 */
#ifndef MGL_SW_H
#define MGL_SW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 102 (round 5): mgl_sw_plan has its round-4 fields workspace_fixed_bytes / resident_waves UNDER A NEW NUMBER (round 4 grew the struct and
 * kept 101), mgl_sw_explain_sized and mgl_sw_ctx_check are new.  A caller built against another number must not pass its mgl_sw_plan to
 * mgl_sw_explain (the library writes sizeof(mgl_sw_plan) of ITS header): compare mgl_sw_version() with MGL_SW_VERSION at load time -- the
 * Python mirror does (mgl_amd/_lib.py) -- or call mgl_sw_explain_sized, which never writes beyond the size it is given. */
#define MGL_SW_VERSION 102

/* overhang strategies: sw_common.h:22-25 (= MicrosoftSmithWaterman.java:39-56) */
#define MGL_SW_OS_SOFTCLIP 0x01
#define MGL_SW_OS_INDEL 0x02
#define MGL_SW_OS_LEAD_ID 0x04
#define MGL_SW_OS_IGNORE 0x08

#define MGL_SW_NEG_INF (-0x40000000) /* sw_common.h:33 */

typedef enum mgl_sw_status {
    MGL_SW_OK = 0,
    MGL_SW_ERR_BAD_ARG = 1,        /* null pointer, length < 1, unknown strategy */
    MGL_SW_ERR_CIGAR_OVERFLOW = 2, /* a CIGAR did not fit; *cigar_len holds the size needed */
    MGL_SW_ERR_NOMEM = 3,          /* host or device allocation failed */
    MGL_SW_ERR_DEVICE = 4,         /* no HIP device / HIP runtime error (see mgl_sw_last_error) */
    MGL_SW_ERR_UNSUPPORTED = 5     /* geometry outside what the kernels cover (see mgl_sw_max_query_len), or scores that
                                      would leave the 32-bit range (parameters x lengths >= 2^30) */
} mgl_sw_status;

/* ScoreMax, sw_common.h:36-40: best score of the last column (mqe, row mqe_t,
 * ties -> larger row), best of last column U last row (max, max_t, max_q) and
 * seg_length = ql - max_q when a last-row cell won. */
typedef struct mgl_sw_score {
    int32_t mqe, mqe_t;
    int32_t max, max_t, max_q;
    int32_t seg_length;
} mgl_sw_score;

/* What one kernel pass cost, for bench.py's roofline (HIP events recorded on
 * the stream the kernels ran on, summed over the chunks of the last call). */
typedef struct mgl_sw_timing {
    float dp_ms;        /* sw_dp_kernel (matrix fill + traceback bits)     */
    float tb_ms;        /* sw_traceback_kernel (path walk + CIGAR text)    */
    int32_t dp_launches, tb_launches;
    int64_t cells;      /* sum tl*ql of the last call                      */
    int64_t tb_bytes;   /* traceback bytes written to HBM by the last call */
    int32_t packed16;   /* 1 when the packed-int16 fill kernel (sw_dp16_kernel) ran */
    int32_t clock_mhz;  /* profiling level 2: shader clock seen inside the fill kernel (s_memtime / s_memrealtime) */
    int32_t fill_kernel; /* which fill kernel the last chunk ran: MGL_SW_KERNEL_* */
    int32_t reserved;
} mgl_sw_timing;
#define MGL_SW_KERNEL_DP32 0      /* sw_dp_kernel: int32, 16 rows x four pairs per wave          */
#define MGL_SW_KERNEL_DP16 1      /* sw_dp16_kernel: packed int16, two pairs per lane, systolic  */
#define MGL_SW_KERNEL_DP32_64 2   /* sw_dp64_kernel: int32, 64 rows, one pair per wave           */
#define MGL_SW_KERNEL_COOP 3      /* sw_dp_coop_kernel: one pair per workgroup (long reads)      */
#define MGL_SW_KERNEL_LANE16 4    /* sw_dp16_lane_kernel: packed int16, two pairs per LANE       */
#define MGL_SW_KERNEL_COOP16 5    /* sw_dp_coop16_kernel: long reads, packed int16, 128 rows/wave */
#define MGL_SW_KERNEL_STRIP16 6   /* sw_dp16_strip_kernel: long reads, one 32-row strip per lane-half */
#define MGL_SW_KERNEL_LANE16_CK 7 /* sw_dp16_lane_ck_kernel: the lane kernel, checkpoints instead of stored flags */
#define MGL_SW_KERNEL_SMALL 8     /* sw_small_kernel: small batches, one wave per pair, scores kept in LDS, fill + walk in one launch */
#define MGL_SW_KERNEL_LANE16_MATRIX 9 /* sw_dp16_lane_matrix_kernel: substitution matrix, two pairs per lane, tiles that share their target */

/* What the library WOULD do with a batch: the planner's decisions, without running anything (mgl_sw_explain). */
typedef struct mgl_sw_plan {
    int32_t fill_kernel;        /* MGL_SW_KERNEL_*: the kernel of a uniform batch, or of the bulk of a sorted mixed one */
    int32_t precision_bits;     /* 16 (packed int16 behind the range guard) or 32 */
    int32_t rows;               /* target rows per stripe / strip of that kernel */
    int32_t waves_per_block;
    int32_t waves_per_pair;     /* long-read kernels: waves that share one pair; else 0 */
    int32_t traceback;          /* 0 = four flags per cell stored in HBM, 1 = none stored (checkpoints; the walk recomputes), 2 = score only */
    int32_t fused_walk;         /* 1 = every lane walks its own paths inside the fill kernel, 0 = traceback kernel on an auxiliary stream */
    int32_t sorted_by_library;  /* mixed geometries: 0 = not sorted (int32 kernel), 1 = counting sort on the device, 2 = on the host */
    int32_t fill_streams;       /* 1, or 2 when consecutive chunks alternate between two streams (host entries, lane kernel) */
    int32_t workspace_halves;   /* 1, or 2 when chunk k's walk overlaps chunk k+1's fill */
    int64_t chunk_pairs;        /* pairs per launch */
    int64_t chunks;             /* launches of the fill kernel for this batch */
    int64_t workspace_bytes_per_pair;
    int64_t workspace_bytes;    /* of the context's workspace that this batch would use (per-pair bytes of a chunk + the fixed part, per half) */
    int64_t workspace_fixed_bytes; /* of those, the part that does not grow with the batch: the regions of sw_dp16_lane_ck_kernel's persistent grid, one per wave slot */
    int64_t resident_waves;     /* wave slots of that grid (0: the kernel is launched one wave or workgroup per unit of work) */
} mgl_sw_plan;

typedef struct mgl_sw_ctx mgl_sw_ctx; /* opaque: one GPU, its workspace and stream */

int mgl_sw_version(void);
const char *mgl_sw_strerror(int status);
/* number of HIP devices visible (0 when there is none or no runtime) */
int mgl_sw_device_count(void);
/* longest query accepted (2^24; the matrix must also stay below 2^34 cells) */
int mgl_sw_max_query_len(void);
/* longest query whose stripe carry fits LDS with one wave per pair; longer ones run one pair per workgroup
 * (sw_dp_coop.hip), the waves handing the carry on through small LDS rings */
int mgl_sw_max_lds_query_len(void);

int mgl_sw_ctx_create(int device, mgl_sw_ctx **out);
void mgl_sw_ctx_destroy(mgl_sw_ctx *ctx);
/* text of the last HIP / argument error seen on this ctx ("" if none) */
const char *mgl_sw_last_error(const mgl_sw_ctx *ctx);
/* cap on the device traceback workspace (bytes); default: a quarter of the device's memory (72 GB on an
 * MI355X), at least 4 GiB; memory is only reserved as batches need it.  Batches are processed in chunks that fit. */
int mgl_sw_ctx_set_workspace(mgl_sw_ctx *ctx, int64_t bytes);
/* fill-kernel arithmetic: 0 (default) = per batch, the packed-int16 kernel when every pair has
 * the same tl and ql and the score range fits 16 bits, else int32; long reads: the 16-bit one-pair-per-workgroup
 * kernel, which checks its own score window and redoes a pair in 32 bits if it must, when the scoring parameters make
 * that worthwhile; 32 = always int32; 16 = like 0, but the long-read kernel is tried whenever its constants fit at
 * all (tests of the fall-back).  Results are bit-identical either way. */
int mgl_sw_ctx_set_precision(mgl_sw_ctx *ctx, int bits);
/* where the int32 fill kernel keeps its stripe carry: 0 (default) = LDS whenever the query fits, 1 = always the
 * HBM scratch used for long queries (for tests; results are identical) */
int mgl_sw_ctx_set_carry_memory(mgl_sw_ctx *ctx, int mode);
/* target rows per stripe (= lanes per pair) of the int32 fill kernel: 0 (default) = 16 (four pairs per wave) for
 * queries below 1024 bases, 64 (one pair per wave) from there on; 16 / 64 force one (tests; identical results) */
int mgl_sw_ctx_set_stripe_rows(mgl_sw_ctx *ctx, int rows);
/* long reads: one pair per workgroup, its waves pipelined over the pair's 64-row stripes (sw_dp_coop.hip).
 * 0 (default) = taken when the query is too long for the one-wave-per-pair LDS carve (see
 * mgl_sw_max_lds_query_len); 1 = never; 2..16 = always, with that many waves per pair (tests; identical
 * results) */
int mgl_sw_ctx_set_cooperative(mgl_sw_ctx *ctx, int mode);
/* queries of 1 024 bases and more (targets of any length: beyond 16 384 rows in several passes): one pair per workgroup, every lane-half a strip of 17 .. 32
 * target rows kept in registers (sw_dp16_strip.hip), per-strip 16-bit baselines.  0 (default) = taken for such batches when the
 * scoring parameters fit its static 16-bit window and enough of its strip slots would be busy; 1 = never; 2 = whenever
 * eligible, whatever the lengths (tests; identical results) */
int mgl_sw_ctx_set_strip_kernel(mgl_sw_ctx *ctx, int mode);
/* uniform batches whose scores fit 16 bits: which packed kernel runs.  0 (default) = by launch size: from 262 144 pairs on
 * the two-pairs-per-LANE kernel (sw_dp16_lane.hip: 128 pairs per wave, nothing shared between lanes), below that the
 * two-pairs-per-lane-of-a-16-lane-group kernel (sw_dp16.hip: eight pairs per wave); 1 = never the lane kernel;
 * 2 = the lane kernel whenever the batch is eligible (tests; results are identical) */
int mgl_sw_ctx_set_lane_kernel(mgl_sw_ctx *ctx, int mode);
/* the lane kernel stores no traceback by default (sw_dp16_lane_ck.hip): its fill keeps the carry row leaving every 16 target rows and
 * the lanes' state every 32 query columns, and the path walk recomputes the 16 x 32 blocks it crosses (same flags, same results, about
 * a fifth of the matrix twice instead of eight flag instructions for every cell).  0 (default) = that form whenever the lane kernel
 * runs with 32-row strips and writes CIGARs, and for the geometries of a batch of mixed lengths that fill whole waves of 128 pairs
 * (chunks sorted by geometry, from two rounds of the chip on); 1 = never (the flags of every cell are stored: needed before
 * mgl_sw_ctx_expand_slot); 2 = same as 0 (tests) */
int mgl_sw_ctx_set_lane_checkpoint(mgl_sw_ctx *ctx, int mode);
/* small batches are latency bound: up to MGL_SW_SMALL_BATCH_PAIRS pairs (MGL_SW_SMALL_BATCH_PAIRS_MIXED without a promise of one geometry:
 * measured crossovers, scripts/small_batch_probe.py) whose targets have at most 512 rows and whose matrix of kept
 * scores fits a workgroup's LDS (256 x 150, 400 x 190, ...) run one wave per pair in ONE launch that fills, walks and writes the text
 * (sw_small.hip; nothing of the workspace is touched).  0 (default) = those batches, on a context none of whose other kernel
 * choices has been forced; 1 = never; 2 = every batch whose bounds allow it, whatever its size and the other settings (the
 * coalescing front-end of mgl_sw_align) */
int mgl_sw_ctx_set_small_kernel(mgl_sw_ctx *ctx, int mode);
#define MGL_SW_SMALL_BATCH_PAIRS 5120
#define MGL_SW_SMALL_BATCH_PAIRS_MIXED 8192
/* 1 = HIP events around every kernel launch of a call, on the streams the kernels run on, read back by
 * mgl_sw_ctx_get_timing (the call itself stays asynchronous); 2 = additionally stamp the shader clock
 * inside the fill kernel (diagnostic; a few extra instructions per workgroup); 3 = as 1, summed over every call until
 * mgl_sw_ctx_get_timing reads and clears it (dp_ms / dp_launches = the mean launch duration over a timed loop); 0 = off */
int mgl_sw_ctx_set_profiling(mgl_sw_ctx *ctx, int enable);
/* The plan for a batch of n pairs up to max_tl x max_ql with these parameters, as the entry named by `entry` would run it on this
 * context (its workspace limit and forced modes included) -- nothing is launched, allocated or copied.  flags: MGL_SW_FLAG_*;
 * packed2: the sequences are 2-bit packed; entry: 0 = device-resident (mgl_sw_align_batch_device*), 1 = host buffers
 * (mgl_sw_align_batch / _status / _2bit).  ctx may be NULL: a default context on a 256-CU device (what the CPU tests pin), with
 * workspace_limit bytes of workspace (0: the default, or the context's own limit when ctx is given).  Planned with a CIGAR stride of
 * 64 bytes (the only decision the stride enters is whether a small batch's text fits the one-wave-per-pair kernel's LDS: a call with
 * a much larger stride may leave that kernel where this plan names it).  Returns the status the call itself would return from its
 * planning (MGL_SW_ERR_UNSUPPORTED, ...). */
int mgl_sw_explain(mgl_sw_ctx *ctx, int64_t workspace_limit, int64_t n, int max_tl, int max_ql, int match, int mismatch, int gopen,
                   int gext, int strategy, int flags, int packed2, int entry, mgl_sw_plan *out);
/* ... the same for a caller whose mgl_sw_plan may be older or newer than the library's: at most out_size bytes of *out are written (the
 * struct only ever grows at its end), and what the library does not know of a larger struct is zeroed. */
int mgl_sw_explain_sized(mgl_sw_ctx *ctx, int64_t workspace_limit, int64_t n, int max_tl, int max_ql, int match, int mismatch, int gopen,
                         int gext, int strategy, int flags, int packed2, int entry, mgl_sw_plan *out, size_t out_size);
int mgl_sw_ctx_get_timing(mgl_sw_ctx *ctx, mgl_sw_timing *out); /* waits for the last call's kernels */
/* What a kernel found out about itself AFTER the call that enqueued it returned (the device entries do not wait for their kernels):
 * today, a persistent grid of sw_dp16_lane_ck_kernel that drew a tile number no launch of its size can draw -- its counter did not
 * stand at zero, tiles may be undone.  Synchronise the stream, then ask: MGL_SW_OK, or MGL_SW_ERR_DEVICE (sticky: every later call on
 * the context reports it too; destroy the context).  The host entries ask by themselves before they return.  Waits for nothing. */
int mgl_sw_ctx_check(mgl_sw_ctx *ctx);

/* Sign normalisation of the JNI boundary
 * (..._MicrosoftSmithWaterman.cpp:51-55): match > 0, mismatch < 0, open > 0,
 * ext > 0 whatever signs the caller used.  All align entry points apply it. */
void mgl_sw_normalize_params(int *match, int *mismatch, int *gopen, int *gext);

/*
 * One pair.  Replaces align_avx (sw_avx.h:6) / align_scalar (sw_scalar.h:9)
 * and the body of alignNative (..._MicrosoftSmithWaterman.cpp:44-71):
 * cigar receives *cigar_len ASCII bytes, no terminator (like cigar.copy(),
 * .cpp:65); *offset is the alignment offset they return; *ez (optional) the
 * ScoreMax the reference keeps local (sw_avx.cpp:9).
 */
int mgl_sw_align(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen,
                 int gext, int strategy, char *cigar, int cigar_cap, int *cigar_len, int *offset,
                 mgl_sw_score *ez);

/*
 * Coalescing of concurrent one-pair calls (the way GATK drives alignNative: many threads, one pair each,
 * MicrosoftSmithWaterman.java:66-86).  Every mgl_sw_align call (hence the JNI export) is parked and merged
 * with the calls of other threads that use the same parameters and strategy into one device batch, flushed
 * when as many calls are waiting as the previous batch held (a lone caller never waits), when max_batch are,
 * or when the oldest has waited max_wait_us.  Results are exactly those of the direct call.  ON by default
 * (max_batch 4096, max_wait_us 50; environment: MGL_SW_COALESCE_US, -1 = off, and MGL_SW_COALESCE_BATCH);
 * max_batch = 0 switches it off: every call is then its own device round trip on the calling thread's context.
 */
int mgl_sw_set_coalescing(int max_batch, int max_wait_us);
/* device batches flushed / pairs served by the coalescer so far */
int mgl_sw_coalescing_stats(int64_t *batches, int64_t *pairs);
/*
 * ... and in front of the coalescer, for pairs of the size GATK sends (targets up to 512 bases, queries up to 2 048, the matrix of
 * scores within a workgroup's LDS: 256 x 150, 400 x 190, ...): every calling thread leases a MAILBOX -- its request half in device
 * memory the host stores into over the large BAR where the platform has one (MGL_SW_SERVICE_BAR=0: never), else in pinned host
 * memory; its reply half in pinned host memory -- and one resident wave that serves it (sw_service.hip).  A call writes its pair into the mailbox and spins until the wave hands the
 * result back: no kernel launch, no stream synchronisation and no other thread on the request path (this replaces the
 * launch-per-call of ..._MicrosoftSmithWaterman.cpp:44-71's callers).  A wave ends by itself when its mailbox has been quiet for
 * idle_us (default 1 000; environment MGL_SW_SERVICE_IDLE_US) or after MGL_SW_SERVICE_LIFE_MS (default 20) -- a resident
 * kernel holds up device-wide synchronisation for that long at most -- and the next call launches it again.  `slots` mailboxes at
 * most (default 64, environment MGL_SW_SERVICE_SLOTS, at most 128, and never more than half the device's CUs: a mailbox's wave holds
 * its carve of LDS -- 64 KB, the full 160 KB only once a pair has needed it -- for as long as the grid lives); threads beyond that,
 * pairs that do not fit, and any call whose wave does not answer within 20 s (the service is switched off from then on) take the
 * coalescer.  slots = 0 switches the service off; idle_us = 0 keeps the current value.  Follows mgl_sw_set_coalescing: with
 * coalescing off every call is a direct call.
 */
int mgl_sw_set_service(int slots, int idle_us);
/* calls served through mailboxes / launches of the service kernel so far */
int mgl_sw_service_stats(int64_t *calls, int64_t *launches);

/*
 * Batch, host buffers.  Pair k is targets[t_off[k] .. t_off[k+1]) against
 * queries[q_off[k] .. q_off[k+1]) (raw bytes, compared for equality exactly
 * as sw.cpp:55).  One parameter set and strategy per batch.  cigar_out is
 * n * cigar_stride bytes; each pair's slot is zero padded (the contract of
 * the Java side's zero-filled direct buffer, MicrosoftSmithWaterman.java:71-85).
 * score_out and cigar_len_out may be NULL.
 */
int mgl_sw_align_batch(mgl_sw_ctx *ctx, int64_t n, const uint8_t *targets, const int64_t *t_off,
                       const uint8_t *queries, const int64_t *q_off, int match, int mismatch,
                       int gopen, int gext, int strategy, int32_t *offset_out,
                       mgl_sw_score *score_out, char *cigar_out, int cigar_stride,
                       int32_t *cigar_len_out);

/*
 * The same with a per-pair status array (int32[n], mgl_sw_status values): with status_out non-NULL a CIGAR that
 * does not fit its slot -- or a device-side failure of one pair -- is reported there and does not fail the call
 * (*cigar_len_out still receives the size needed).  status_out == NULL is exactly mgl_sw_align_batch.
 */
int mgl_sw_align_batch_status(mgl_sw_ctx *ctx, int64_t n, const uint8_t *targets, const int64_t *t_off,
                              const uint8_t *queries, const int64_t *q_off, int match, int mismatch,
                              int gopen, int gext, int strategy, int32_t *offset_out,
                              mgl_sw_score *score_out, char *cigar_out, int cigar_stride,
                              int32_t *cigar_len_out, int32_t *status_out);

/*
 * Several GPUs from ONE process ("single process, one host thread per device").  Pairs are independent
 * -- alignNative holds no state, ..._MicrosoftSmithWaterman.cpp:44-71 -- so mgl_sw_align_batch_multi cuts the host
 * batch into contiguous shards, one per device of the set, balanced by the DP cells (sum tl * ql) they hold, and runs
 * every shard through mgl_sw_align_batch_status on that device's own context from its own host thread.  Results are
 * written straight into the caller's arrays (shards are disjoint slices); no inter-GPU traffic.  Arguments and error
 * behaviour are those of mgl_sw_align_batch_status.  `devices` = n_devices HIP ordinals (NULL: 0 .. n_devices-1; an
 * ordinal may be listed more than once -- each entry gets its own context and host thread).
 */
typedef struct mgl_sw_multi mgl_sw_multi;
int mgl_sw_multi_create(int n_devices, const int *devices, mgl_sw_multi **out);
void mgl_sw_multi_destroy(mgl_sw_multi *m);
int mgl_sw_multi_device_count(const mgl_sw_multi *m);
/* the context of entry `index` (to tune it with the mgl_sw_ctx_set_* calls); owned by the set */
mgl_sw_ctx *mgl_sw_multi_ctx(mgl_sw_multi *m, int index);
int mgl_sw_multi_set_workspace(mgl_sw_multi *m, int64_t bytes_per_device);
const char *mgl_sw_multi_last_error(const mgl_sw_multi *m);
int mgl_sw_align_batch_multi(mgl_sw_multi *m, int64_t n, const uint8_t *targets, const int64_t *t_off,
                             const uint8_t *queries, const int64_t *q_off, int match, int mismatch,
                             int gopen, int gext, int strategy, int32_t *offset_out,
                             mgl_sw_score *score_out, char *cigar_out, int cigar_stride,
                             int32_t *cigar_len_out, int32_t *status_out);
/* first pair of every shard of the last mgl_sw_align_batch_multi call: first_out[0 .. n_devices] */
int mgl_sw_multi_last_shards(mgl_sw_multi *m, int64_t *first_out);
/* The sharding rule by itself (host only, no device needed): contiguous parts of pairs 0 .. n-1 with equal shares of
 * sum tl * ql, every boundary a multiple of `align` pairs; first_out[0 .. parts], first_out[parts] == n. */
int mgl_sw_shard_by_cells(int64_t n, const int64_t *t_off, const int64_t *q_off, int parts, int64_t align,
                          int64_t *first_out);

/*
 * Batch, device-resident: every pointer is a device pointer on ctx's GPU and
 * the work is enqueued on `stream` (a hipStream_t; NULL = the null stream)
 * without synchronising -- unless profiling is enabled.  max_tl / max_ql are
 * upper bounds of the pair lengths (they size the workspace).  status_out
 * (optional, int32[n]) receives a per-pair mgl_sw_status (0 or
 * MGL_SW_ERR_CIGAR_OVERFLOW).  flags: MGL_SW_FLAG_UNIFORM_GEOMETRY promises that every
 * pair has exactly tl == max_tl and ql == max_ql (enables the packed-int16 fill kernels; the
 * host-buffer entry detects this by itself).  Without a promise a batch of 1024 pairs or more whose
 * (max_tl, max_ql) grid has at most 2^20 cells is sorted by geometry on the device, chunk by chunk
 * (counting sort, sw_regroup_*_kernel): blocks of eight pairs of one geometry go through the packed
 * kernel, the few left over through the int32 kernel, every result lands at its pair's own index.  The
 * host reads one word per chunk back to size those launches, so such a call waits for the sorts (not for
 * the alignments) before it returns; MGL_SW_AUTO_GROUP=0 in the environment keeps the int32 kernel.
 */
#define MGL_SW_FLAG_UNIFORM_GEOMETRY 0x1
/* MGL_SW_FLAG_BINARY_CIGAR: the CIGAR slot receives BAM-style little-endian uint32 elements
 * (length << 4 | op, op M=0 I=1 D=2 S=4) instead of the text of sw.cpp:251-252; cigar_len is then in bytes
 * (4 per element) and cigar_stride should be a multiple of 4.  Same elements, same order. */
#define MGL_SW_FLAG_BINARY_CIGAR 0x2
/* MGL_SW_FLAG_GROUPED_GEOMETRY: a promise that every aligned block of eight consecutive pairs (pairs 8k .. 8k+7;
 * the last block may be shorter) has one (tl, ql) -- e.g. a batch of variable-length reads sorted by length and
 * padded per length to a multiple of eight.  Such a batch is eligible for the packed-int16 kernel like a uniform
 * one (each wave of that kernel works on one block).  Results are undefined if the promise is broken. */
#define MGL_SW_FLAG_GROUPED_GEOMETRY 0x4
/* MGL_SW_FLAG_SCORE_ONLY: the caller only wants d_score_out (all six ScoreMax fields, bit-identical to the full
 * call).  A hint: batches that run on the packed-int16 kernel then skip the traceback flags and the path walk
 * (offsets are written as 0, cigar_len as 0, the CIGAR slots are left untouched); every other batch runs the full
 * path.  Not a reference feature (align_* always builds the CIGAR): a pre-filter mode for database searches. */
#define MGL_SW_FLAG_SCORE_ONLY 0x8
/* MGL_SW_FLAG_SHARED_TARGET (mgl_sw_align_batch_device_matrix only; ignored elsewhere): a promise that every aligned block of 128
 * consecutive pairs (pairs 128k .. 128k+127; the last block may be shorter) shares ONE target -- the same d_t_off and d_t_len -- and
 * one query length: a database search laid out database sequence by database sequence.  Such a batch runs on a kernel that gives every
 * lane two pairs and looks the scores of a column up as one row of a per-strip profile (sw_dp16_lane_matrix.hip); gap penalties and
 * matrix must satisfy 0 <= S + gopen + gext <= 255 for every entry and the score range 16 bits, else the flag is read as
 * MGL_SW_FLAG_GROUPED_GEOMETRY.  A block that breaks the promise is NOT computed: its pairs get MGL_SW_ERR_BAD_ARG in d_status_out.
 * The call looks at every block's lengths before it launches (8 bytes per block back to the host: it waits for `stream` once) and sizes its
 * workspace by the largest blocks; the blocks may come in any order.
 * Combines with MGL_SW_FLAG_SCORE_ONLY (needs d_score_out) and MGL_SW_FLAG_BINARY_CIGAR. */
#define MGL_SW_FLAG_SHARED_TARGET 0x10
int mgl_sw_align_batch_device(mgl_sw_ctx *ctx, void *stream, int64_t n, const uint8_t *d_targets,
                              const int64_t *d_t_off, const uint8_t *d_queries,
                              const int64_t *d_q_off, int max_tl, int max_ql, int match,
                              int mismatch, int gopen, int gext, int strategy,
                              int32_t *d_offset_out, mgl_sw_score *d_score_out, char *d_cigar_out,
                              int cigar_stride, int32_t *d_cigar_len_out, int32_t *d_status_out,
                              int flags);

/*
 * Batch, device-resident, 2-bit packed bases -- the wire format for ACGT data instead of the
 * ASCII-in-ByteBuffer of MicrosoftSmithWaterman.java:73-75.  d_target_bases / d_query_bases hold four bases per
 * byte, base k of an array in bits 2*(k%4) of byte k/4 (A=0 C=1 G=2 T=3 by convention; the kernels only test
 * equality, exactly like sw.cpp:55 does on bytes).  Pair p is the d_t_len[p] bases starting at BASE index
 * d_t_start[p] against the d_q_len[p] bases starting at d_q_start[p] -- target windows may overlap, e.g. windows
 * into one packed genome.  With MGL_SW_FLAG_UNIFORM_GEOMETRY the length arrays may be NULL (every pair max_tl x
 * max_ql).  Everything else as mgl_sw_align_batch_device; results equal those of the ASCII entries on the
 * unpacked sequences.
 */
int mgl_sw_align_batch_device_2bit(mgl_sw_ctx *ctx, void *stream, int64_t n, const uint8_t *d_target_bases,
                                   const int64_t *d_t_start, const int32_t *d_t_len,
                                   const uint8_t *d_query_bases, const int64_t *d_q_start,
                                   const int32_t *d_q_len, int max_tl, int max_ql, int match, int mismatch,
                                   int gopen, int gext, int strategy, int32_t *d_offset_out,
                                   mgl_sw_score *d_score_out, char *d_cigar_out, int cigar_stride,
                                   int32_t *d_cigar_len_out, int32_t *d_status_out, int flags);

/*
 * The same wire format from HOST memory (replaces the ASCII-in-ByteBuffer contract of MicrosoftSmithWaterman.java:73-75 for callers
 * that hold packed bases): target_base_count / query_base_count are the numbers of bases the two packed arrays hold.  The packed
 * arrays travel chunk by chunk beside the kernels when the pairs' start positions ascend (reads packed back to back), or whole
 * before the first chunk (windows into one genome, in any order); results leave as in mgl_sw_align_batch_status (status_out may
 * be NULL: then a CIGAR that does not fit fails the call).  Everything else as mgl_sw_align_batch_device_2bit.
 */
int mgl_sw_align_batch_2bit(mgl_sw_ctx *ctx, int64_t n, const uint8_t *target_bases, int64_t target_base_count,
                            const int64_t *t_start, const int32_t *t_len, const uint8_t *query_bases,
                            int64_t query_base_count, const int64_t *q_start, const int32_t *q_len, int max_tl, int max_ql,
                            int match, int mismatch, int gopen, int gext, int strategy, int32_t *offset_out,
                            mgl_sw_score *score_out, char *cigar_out, int cigar_stride, int32_t *cigar_len_out,
                            int32_t *status_out, int flags);

/*
 * Page-lock arrays the caller passes to the host-buffer entries (mgl_sw_align_batch, _status, _2bit) again and again --
 * hipHostRegister: the direct ByteBuffers of a JVM are a natural fit.  Copies from registered input arrays are asynchronous DMA
 * (the entry no longer blocks inside pageable copies), and when EVERY output array of a call lies in registered memory the
 * results are copied straight into it instead of through the context's own pinned ring.  Registration costs milliseconds per
 * gigabyte: do it once, not per call.  Unregister before freeing the memory (waits for the context's copies to finish).
 */
int mgl_sw_register_host_buffer(mgl_sw_ctx *ctx, void *ptr, size_t bytes);
int mgl_sw_unregister_host_buffer(mgl_sw_ctx *ctx, void *ptr);

/*
 * ASCII bases addressed by (start, length) per pair instead of consecutive offsets: pair k =
 * d_targets[d_t_start[k] .. + d_t_len[k]) against d_queries[d_q_start[k] .. + d_q_len[k]).  Sequences may be shared
 * or reordered without moving bytes (e.g. to satisfy MGL_SW_FLAG_GROUPED_GEOMETRY by sorting index arrays).
 * Everything else as mgl_sw_align_batch_device; outputs are indexed by k.
 */
int mgl_sw_align_batch_device_indexed(mgl_sw_ctx *ctx, void *stream, int64_t n, const uint8_t *d_targets,
                                      const int64_t *d_t_start, const int32_t *d_t_len, const uint8_t *d_queries,
                                      const int64_t *d_q_start, const int32_t *d_q_len, int max_tl, int max_ql,
                                      int match, int mismatch, int gopen, int gext, int strategy,
                                      int32_t *d_offset_out, mgl_sw_score *d_score_out, char *d_cigar_out,
                                      int cigar_stride, int32_t *d_cigar_len_out, int32_t *d_status_out, int flags);

/*
 * Substitution-matrix scoring ("protein" mode).  NOT in the reference, which scores
 * by byte equality only (sw.cpp:55): the same recurrence, overhang strategies and traceback with
 *     diag = H[i-1][j-1] + matrix[code[t[i-1]] * 32 + code[q[j-1]]]
 * code: 256 bytes -> 0..31, matrix: 32 x 32 int8 (both HOST pointers, copied per call); gopen / gext as in the
 * other entries.  d_t_len / d_q_len (optional, int32 per pair): with them d_t_off[k] / d_q_off[k] are per-pair START
 * positions, so one database sequence can serve many pairs; NULL = pair k is [off[k], off[k+1]).  Uniform / grouped batches (flags) whose score range fits 16 bits
 * take the packed kernel, all others the int32 kernel; queries up to about 3 300 residues (MGL_SW_ERR_UNSUPPORTED
 * beyond), targets any length.  No parity claim exists for this mode: it is validated against the CPU restatement's own extension and
 * an independent textbook DP (tests/test_gpu_matrix.py).
 */
int mgl_sw_align_batch_device_matrix(mgl_sw_ctx *ctx, void *stream, int64_t n, const uint8_t *d_targets,
                                     const int64_t *d_t_off, const int32_t *d_t_len, const uint8_t *d_queries,
                                     const int64_t *d_q_off, const int32_t *d_q_len, int max_tl, int max_ql,
                                     const int8_t *matrix, const uint8_t *code, int gopen,
                                     int gext, int strategy, int32_t *d_offset_out, mgl_sw_score *d_score_out,
                                     char *d_cigar_out, int cigar_stride, int32_t *d_cigar_len_out,
                                     int32_t *d_status_out, int flags);

/*
 * Logical backtrack matrix of one pair, the reference's calculateMatrix
 * (sw_scalar.h:7 / sw.cpp:5-146): btr is (tl+1)*(ql+1) int32 row-major with
 * row 0 / column 0 zero, +k = k rows up (deletion run), -k = k columns left
 * (insertion run), 0 = diagonal; *ez as the reference fills it.  Expanded on
 * the GPU from the 4-bit-per-cell device traceback; a parity / debugging
 * entry, not a fast path.
 */
int mgl_sw_backtrack_matrix(const char *t, int tl, const char *q, int ql, int match, int mismatch,
                            int gopen, int gext, int strategy, int32_t *btr, mgl_sw_score *ez);

/*
 * Traceback + CIGAR text from a caller-supplied logical backtrack matrix and
 * ScoreMax: the reference's calculateCigar (sw_scalar.h:8 / sw.cpp:149-255) with
 * n = tl+1, m = ql+1.  The walk runs on the GPU (one lane).
 */
int mgl_sw_cigar_from_backtrack(const int32_t *btr, int tl, int ql, int strategy, const mgl_sw_score *ez,
                                char *cigar, int cigar_cap, int *cigar_len, int *offset);

/*
 * One band of the matrix fill over the caller's arrays: the reference's calculateMatrix_avx (sw_avx.h:7 /
 * sw_avx.cpp:110-322), the stage its driver runs once per band of default_bw = 8 target rows (sw_avx.cpp:71-80).
 * Layouts are the reference's, all host pointers: `target` one base per int32, zero padded to a
 * multiple of default_bw; `query` reversed, query[default_bw + query_length - 1 - j] = q[j], default_bw zeros on each
 * side; `gap` indexed like `query` (vertical run lengths per column); `score` / `step` = H and E of the row above the
 * band for columns 0 .. query_length (on return: of the band's last row); the band's backtrack values go to
 * bcktrack[(query_length + default_bw - 1) * default_bw * band_count + (j - 1 + J) * default_bw + J] for row J of the
 * band and column j -- cells of that region outside the matrix are left as the caller had them (the reference leaves
 * garbage there, and in the padding entries of `gap`; after a band of fewer than default_bw rows -- the last one --
 * `gap` holds the run lengths of the band's last real row where the reference leaves those of its padding rows:
 * nothing reads it any more); ez->mqe / ez->mqe_t are updated with the band's last-column
 * values ('>=': later rows win).  A compatibility entry for callers that drive the band loop themselves (one small
 * kernel and a round trip per band), not a fast path.
 */
int mgl_sw_band_fill(const int32_t *target, int target_length, const int32_t *query, int query_length,
                     int32_t *bcktrack, int band_count, int default_bw, int actual_bw, int32_t *score,
                     int32_t *step, int32_t *gap, int match, int mismatch, int gopen, int gext, int strategy,
                     mgl_sw_score *ez);

/*
 * Same expansion for pair `slot` of the LAST chunk a batch call processed on ctx
 * (slot = pair index when the whole batch fitted one chunk).  The traceback
 * workspace is only valid until the next call on ctx.  tl / ql must be that
 * pair's lengths.  Parity / debugging entry.  MGL_SW_ERR_UNSUPPORTED when the last call ran the
 * checkpointed lane kernel, which stores no traceback (large uniform batches by default:
 * mgl_sw_ctx_set_lane_checkpoint(ctx, 1) before the batch call keeps the flags of every cell).
 */
int mgl_sw_ctx_expand_slot(mgl_sw_ctx *ctx, int64_t slot, int tl, int ql, int32_t *btr);
/* Traceback layout pair `slot` of the last chunk was filled in: 0 = an int32 kernel, 1 = sw_dp16_kernel, 2 =
 * sw_dp16_lane_kernel, 3 = sw_dp_coop16_kernel (which reports 0 for a pair it had to redo in 32 bits).  Tests. */
int mgl_sw_ctx_slot_layout(mgl_sw_ctx *ctx, int64_t slot, int *layout);

/*
 * Host helper for MGL_SW_FLAG_GROUPED_GEOMETRY (no device work, usable without a GPU): a permutation of 0 .. n-1 that
 * puts pairs of equal (t_len, q_len) next to each other.  order_out[0 .. *n_grouped_out) are FULL blocks of eight pairs
 * with one geometry each (a multiple of eight entries: pass them, through the (start, length) arrays of
 * mgl_sw_align_batch_device_indexed / _2bit permuted by order_out, with the flag set); the rest -- fewer than eight
 * pairs per distinct geometry -- follows, sorted by geometry (an ordinary mixed batch, no flag).  O(n) when the lengths
 * span at most 2^22 distinct (t_len, q_len) cells, else O(n log n).
 */
int mgl_sw_group_by_geometry(int64_t n, const int32_t *t_len, const int32_t *q_len, int64_t *order_out, int64_t *n_grouped_out);

#ifdef __cplusplus
}
#endif
#endif /* MGL_SW_H */
