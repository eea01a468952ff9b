/*
 * jni.h -- a MINIMAL, TEST-ONLY stand-in for the JDK's header: just enough of the Java Native Interface's C++ face to
 * compile mgl_amd/csrc/jni_exports.cpp and pairhmm_jni_exports.cpp in an image without a JDK and to call the exported
 * functions through a fake JNIEnv (tests/cpp/jni_harness.cpp).  It is never on the product's include path and never
 * shipped: the product library is built with the real <jni.h> of a JDK.  Type names and
 * member signatures follow the JNI specification (jint = 32-bit, jlong = 64-bit, objects are opaque pointers, the
 * JNIEnv is a struct of member functions taking the spec's arguments).
 */
#ifndef MGL_TEST_JNI_STUB_H
#define MGL_TEST_JNI_STUB_H

#include <stdint.h>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNIIMPORT

typedef int32_t jint;
typedef int64_t jlong;
typedef uint8_t jboolean;
typedef int8_t jbyte;

class _jobject {};
class _jclass : public _jobject {};
typedef _jobject *jobject;
typedef _jclass *jclass;

/* what the fake environment hands out as a direct java.nio.Buffer: its address and capacity */
struct MglFakeDirectBuffer : _jobject {
    void *address;
    jlong capacity;
};

struct JNIEnv_ {
    /* the four entry points the exports use (JNI spec: NIO support, class operations, exceptions) */
    void *GetDirectBufferAddress(jobject buf) { return buf ? static_cast<MglFakeDirectBuffer *>(buf)->address : nullptr; }
    jlong GetDirectBufferCapacity(jobject buf) { return buf ? static_cast<MglFakeDirectBuffer *>(buf)->capacity : -1; }
    jclass FindClass(const char *name)
    {
        last_class = name;
        return &a_class;
    }
    jint ThrowNew(jclass, const char *message)
    {
        thrown = true;
        thrown_class = last_class;
        thrown_message = message ? message : "";
        return 0;
    }
    /* what the harness inspects afterwards */
    bool thrown = false;
    const char *last_class = "", *thrown_class = "", *thrown_message = "";
    _jclass a_class;
};
typedef JNIEnv_ JNIEnv;

#endif
