// pairhmm_jni_exports.cpp -- the three JNI symbols GATK's loader binds in libmgl_pairhmm.so
// (/root/reference/src/main/native/mgl_pairhmm/com_microsoft_mgl_pairhmm_MicrosoftPairHmm.h; signatures
// (ZI)V, (Ljava/nio/IntBuffer;Ljava/nio/ByteBuffer;Ljava/nio/ByteBuffer;Ljava/nio/DoubleBuffer;)V, ()V),
// implemented over the C ABI of include/mgl_pairhmm.h.
//
// Compiled only where a JDK provides <jni.h> (none in the authoring image: the file then contributes nothing to
// the library).  The Java side is unchanged: MicrosoftPairHmm.java:62-112 fills four direct buffers (lengths,
// reads, haplotypes, likelihoods) and the native side fills likelihoods[r * nHaplotypes + h]
// (com_microsoft_mgl_pairhmm_MicrosoftPairHmm.cc:77-222).  See INTEGRATION.md section 6.
#if defined(__has_include)
#if __has_include(<jni.h>)
#define MGL_PAIRHMM_HAVE_JNI 1
#endif
#endif

#ifdef MGL_PAIRHMM_HAVE_JNI
#include <jni.h>

#include <atomic>
#include <cstdlib>

#include "../../include/mgl_pairhmm.h"

namespace {
// initNative is static in the reference (globals g_use_double, …PairHmm.cc:36-38); here the flag is process-wide too, the GPU
// context is per calling thread (calls on one context are serialised; regions computed by different threads overlap on the GPU)
std::atomic<int> g_use_double{0};

struct Holder {
    mgl_pairhmm_ctx *ctx = nullptr;
    int use_double = -1;
    ~Holder() { mgl_pairhmm_ctx_destroy(ctx); }
};

mgl_pairhmm_ctx *context()
{
    static thread_local Holder h;
    if (!h.ctx) {
        int dev = 0;
        if (const char *e = getenv("MGL_PAIRHMM_DEVICE")) dev = atoi(e);
        if (mgl_pairhmm_ctx_create(dev, &h.ctx) != MGL_PAIRHMM_OK) return nullptr;
    }
    const int want = g_use_double.load();
    if (h.use_double != want) {
        mgl_pairhmm_initialize(h.ctx, want, 1);
        h.use_double = want;
    }
    return h.ctx;
}
} // namespace

extern "C" {

JNIEXPORT void JNICALL Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_initNative(JNIEnv *, jclass, jboolean use_double,
                                                                                  jint max_threads)
{
    (void)max_threads;                          // ignored by the reference as well (…PairHmm.cc:50-70)
    g_use_double.store(use_double ? 1 : 0);     // …PairHmm.cc:50-53; picked up by every thread's context at its next call
}

JNIEXPORT void JNICALL Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_computeLikelihoodsNative(JNIEnv *env, jobject,
                                                                                                jobject lengthBuffer,
                                                                                                jobject readsBuffer,
                                                                                                jobject haplotypesBuffer,
                                                                                                jobject likelihoodBuffer)
{
    mgl_pairhmm_ctx *ctx = context();
    int rc = MGL_PAIRHMM_ERR_DEVICE;
    if (ctx)
        rc = mgl_pairhmm_compute_likelihoods(ctx, static_cast<const int32_t *>(env->GetDirectBufferAddress(lengthBuffer)), // :84
                                             static_cast<const uint8_t *>(env->GetDirectBufferAddress(readsBuffer)),      // :92
                                             static_cast<const uint8_t *>(env->GetDirectBufferAddress(haplotypesBuffer)), // :110
                                             static_cast<double *>(env->GetDirectBufferAddress(likelihoodBuffer)));       // :121
    if (rc != MGL_PAIRHMM_OK) {
        // the reference cannot fail; surface ours as a Java exception instead of wrong likelihoods
        jclass ex = env->FindClass("java/lang/RuntimeException");
        if (ex) env->ThrowNew(ex, ctx ? mgl_pairhmm_last_error(ctx) : mgl_pairhmm_strerror(rc));
    }
}

JNIEXPORT void JNICALL Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_doneNative(JNIEnv *, jobject)
{
    // a no-op in the reference (…PairHmm.cc:224-226); the per-thread contexts are released when their threads end
}

} // extern "C"
#endif // MGL_PAIRHMM_HAVE_JNI
