/*
 * avx_compat.h -- force-included (-include) when compiling the reference's
 * UNMODIFIED sw_avx.cpp on glibc/GCC.  Test infrastructure only.
 *
 * The reference's AVX2 file was written against MSVC (sw_common.h:13-14):
 *   (a) it calls aligned_alloc(size, alignment) in _aligned_malloc argument
 *       order (sw_avx.cpp:17,26,33,38,40,42); C11 order is (alignment, size);
 *   (b) it uses the aligned _mm256_load/store_si256 forms on addresses that
 *       are only 4-byte aligned (sw_avx.cpp:161-162,173,183,...), which MSVC
 *       emits as unaligned moves.
 * Without these three macros the unmodified source aborts at run time here.
 * Nothing the image lacks is being stood in for: <stdlib.h> and <x86intrin.h>
 * are the system headers; only the argument order / alignment assumption is
 * adapted.  The scalar path (sw.cpp) is built with no shim at all and is the
 * authoritative pin; the AVX2 build is only ever used after asserting that it
 * agrees with the scalar build (tests/golden/make_golden.py).
 */
#include <stdlib.h>
#include <x86intrin.h>
static inline void *mgl_ref_aligned_alloc(size_t size, size_t alignment)
{
    return ::aligned_alloc(alignment, (size + alignment - 1) / alignment * alignment);
}
#define aligned_alloc(a, b) mgl_ref_aligned_alloc((a), (b))
#define _mm256_load_si256 _mm256_loadu_si256
#define _mm256_store_si256 _mm256_storeu_si256
