// jni_exports.cpp -- the three JNI symbols GATK's loader binds in libmgl_sw.so
// (/root/reference/src/main/native/mgl_sw/com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman.h:15-32,
// JNI signature (Ljava/nio/ByteBuffer;Ljava/nio/ByteBuffer;IIIIIII)I), implemented over the C ABI.
//
// Compiled only where a JDK provides <jni.h> (none in the authoring image: the file then contributes
// nothing to the library).  The Java side is unchanged: MicrosoftSmithWaterman.java:73-86 passes one
// direct buffer holding target bytes followed by query bytes and a zero-filled CIGAR buffer of
// 2*max(refLength, altLength) bytes; the native side writes cigar.length() bytes with no terminator
// (.cpp:65) and returns the offset.  See INTEGRATION.md.
#if defined(__has_include)
#if __has_include(<jni.h>)
#define MGL_SW_HAVE_JNI 1
#endif
#endif

#ifdef MGL_SW_HAVE_JNI
#include <jni.h>

#include "../../include/mgl_sw.h"

extern "C" {

JNIEXPORT void JNICALL Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_initNative(JNIEnv *, jclass) {}

JNIEXPORT jint JNICALL Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_alignNative(
    JNIEnv *env, jclass, jobject readsBuffer, jobject cigarBuffer, jint targetLength, jint queryLength, jint match,
    jint mismatch, jint gapOpen, jint gapExtend, jint overhangStrategy)
{
    const char *target = static_cast<const char *>(env->GetDirectBufferAddress(readsBuffer)); // .cpp:48
    const char *query = target + targetLength;                                                // .cpp:49
    char *cigar = static_cast<char *>(env->GetDirectBufferAddress(cigarBuffer));
    const jlong cap = env->GetDirectBufferCapacity(cigarBuffer);
    int len = 0, offset = 0;
    // sign normalisation (.cpp:51-55) happens inside mgl_sw_align
    const int rc = mgl_sw_align(target, targetLength, query, queryLength, match, mismatch, gapOpen, gapExtend,
                                overhangStrategy, cigar, (int)cap, &len, &offset, nullptr);
    if (rc != MGL_SW_OK) {
        // the reference cannot fail; surface ours as a Java exception instead of a wrong alignment
        jclass ex = env->FindClass("java/lang/RuntimeException");
        if (ex) env->ThrowNew(ex, mgl_sw_strerror(rc));
        return 0;
    }
    return offset;
}

JNIEXPORT void JNICALL Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_doneNative(JNIEnv *, jclass) {}

} // extern "C"
#endif // MGL_SW_HAVE_JNI
