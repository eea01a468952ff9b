// jni_harness.cpp -- drives the JNI exports of mgl_amd/csrc/jni_exports.cpp and pairhmm_jni_exports.cpp exactly as the
// Java side does, through a fake JNIEnv (tests/cpp/jni_stub/jni.h).  TEST CODE: the two export files are compiled INTO
// this binary against the stub header (the product library carries them only when built with a JDK) and linked with the
// product libraries.
//
// Buffer contract reproduced from /root/reference/src/main/java/com/microsoft/mgl/smithwaterman/MicrosoftSmithWaterman.java:66-86:
// readsBuffer = refLength target bytes followed by altLength query bytes (:73-75); cigarBuffer = a zero-filled direct
// buffer of 2 * max(refLength, altLength) bytes (:71,77); Java reads it back with new String(bytes).trim() (:83-85).
//
//   sw      : stdin lines "t q match mismatch open ext strategy" -> "<offset> <cigar>" or "EXC <class> <message>"
//   pairhmm : the reference's simpleTest pair (MicrosoftPairHmmUnitTest.java:22-56) -> "<log10 likelihood>"
#include <jni.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

extern "C" {
void Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_initNative(JNIEnv *, jclass);
jint Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_alignNative(JNIEnv *, jclass, jobject, jobject, jint, jint, jint, jint,
                                                                             jint, jint, jint);
void Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_doneNative(JNIEnv *, jclass);
void Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_initNative(JNIEnv *, jclass, jboolean, jint);
void Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_computeLikelihoodsNative(JNIEnv *, jobject, jobject, jobject, jobject, jobject);
void Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_doneNative(JNIEnv *, jobject);
}

static int run_sw()
{
    JNIEnv env;
    _jclass cls;
    Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_initNative(&env, &cls);
    std::string t, q;
    int match, mismatch, gopen, gext, strategy;
    while (std::cin >> t >> q >> match >> mismatch >> gopen >> gext >> strategy) {
        std::vector<char> reads(t.size() + q.size());                       // .java:73-75
        memcpy(reads.data(), t.data(), t.size());
        memcpy(reads.data() + t.size(), q.data(), q.size());
        std::vector<char> cigar(2 * std::max(t.size(), q.size()), 0);        // .java:71,77: allocateDirect zero-fills
        MglFakeDirectBuffer rb, cb;
        rb.address = reads.data();
        rb.capacity = (jlong)reads.size();
        cb.address = cigar.data();
        cb.capacity = (jlong)cigar.size();
        env.thrown = false;
        const jint off = Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_alignNative(
            &env, &cls, &rb, &cb, (jint)t.size(), (jint)q.size(), match, mismatch, gopen, gext, strategy);
        if (env.thrown) {
            std::printf("EXC %s %s\n", env.thrown_class, env.thrown_message);
            continue;
        }
        size_t len = cigar.size();                                           // new String(bytes).trim(): .java:83-85
        while (len > 0 && cigar[len - 1] == 0) --len;
        std::printf("%d %s\n", (int)off, std::string(cigar.data(), len).c_str());
    }
    Java_com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman_doneNative(&env, &cls);
    return 0;
}

static int run_pairhmm()
{
    // one read, one haplotype, packed as MicrosoftPairHmm.java:62-112 packs them: reads = bases, quals, insertion /
    // deletion / overall GCP per read
    JNIEnv env;
    _jclass cls;
    _jobject self;
    for (int use_double = 0; use_double < 2; ++use_double) {
        Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_initNative(&env, &cls, (jboolean)use_double, 1);
        const char *bases = "ACGT";
        int32_t lengths[4] = {1, 4, 1, 4};                                  // nReads, read lengths, nHaplotypes, haplotype lengths (.java:70-88)
        std::vector<uint8_t> reads;
        for (int k = 0; k < 4; ++k) reads.push_back((uint8_t)bases[k]);
        for (int part = 0; part < 4; ++part)
            for (int k = 0; k < 4; ++k) reads.push_back((uint8_t)'+');          // simpleTest passes "++++" as is (MicrosoftPairHmmUnitTest.java:44-48)
        std::vector<uint8_t> haps(bases, bases + 4);
        double out[1] = {0.0};
        MglFakeDirectBuffer lb, rb, hb, ob;
        lb.address = lengths;
        lb.capacity = 4;
        rb.address = reads.data();
        rb.capacity = (jlong)reads.size();
        hb.address = haps.data();
        hb.capacity = 4;
        ob.address = out;
        ob.capacity = 1;
        env.thrown = false;
        Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_computeLikelihoodsNative(&env, &self, &lb, &rb, &hb, &ob);
        if (env.thrown)
            std::printf("EXC %s %s\n", env.thrown_class, env.thrown_message);
        else
            std::printf("%.9e\n", out[0]);
    }
    Java_com_microsoft_mgl_pairhmm_MicrosoftPairHmm_doneNative(&env, &self);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "pairhmm") return run_pairhmm();
    return run_sw();
}
