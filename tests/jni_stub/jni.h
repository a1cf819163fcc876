/* minimal stand-in of jni.h for a syntax check of the shim (the image has no JDK) */
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef void* jobject; typedef void* jclass; typedef void* jstring;
struct JNINativeInterface_; typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ { void* (*GetDirectBufferAddress)(JNIEnv*, jobject); jlong (*GetDirectBufferCapacity)(JNIEnv*, jobject); jstring (*NewStringUTF)(JNIEnv*, const char*); jobject (*NewDirectByteBuffer)(JNIEnv*, void*, jlong); };
#define JNIEXPORT
#define JNICALL
