/*
 * gpx_jni.c — the JNI shim between gigapaxos's Java host and the C-ABI of include/gpx.h /
 * include/gpx_wire.h.
 *
 * Compiled only where a JDK exists (none in the build image or on the GPU box: `java -version` is
 * "command not found" on both), hence the guard:
 *
 *   gcc -shared -fPIC -DGPX_HAVE_JNI -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       gigapaxos_amd/jni/gpx_jni.c -Lgigapaxos_amd/csrc -lgpx_hip \
 *       -Wl,-rpath,'$ORIGIN/../csrc' -o gigapaxos_amd/jni/libgpx_jni.so
 *
 * Without GPX_HAVE_JNI the file compiles to an empty translation unit plus a self-check of the
 * argument counts against the header (tests/test_abi_symbols.py builds it that way), so the shim
 * cannot silently drift from include/gpx.h.
 *
 * Java side: edu.umass.cs.gigapaxos.gpx.GpxEngine.  Every column is a direct
 * ByteBuffer in native byte order, owned by the caller and valid only for the call - the contract
 * of include/gpx.h.  One submitting thread per engine at a time (ConsumerTask.java:163-174).
 * Reference seam each entry replaces: PaxosManager.handlePaxosPacket -> PaxosInstanceStateMachine.
 * handlePaxosMessage (PaxosManager.java:1126-1204, PaxosInstanceStateMachine.java:411-583).
 */
#include <stdint.h>

#include "../../include/gpx.h"
#include "../../include/gpx_wire.h"

#ifdef GPX_HAVE_JNI
#include <jni.h>

#define H(h) ((gpx_engine*)(intptr_t)(h))
#define B(b) ((b) ? (*env)->GetDirectBufferAddress(env, (b)) : 0)
#define JFN(ret, name) JNIEXPORT ret JNICALL Java_edu_umass_cs_gigapaxos_gpx_GpxEngine_##name

JFN(jlong, create)(JNIEnv* env, jclass c, jint myId, jint maxGroups, jint kmax, jint window, jint maxBatch,
                   jint device) {
  (void)env, (void)c;
  gpx_config cfg = {myId, maxGroups, kmax, window, maxBatch, device, GPX_F_ACCEPTS_FROM_DISK, 0};
  gpx_engine* h = 0;
  /* a handle (never negative as a jlong: a user-space pointer) or the negative GPX_E* code; the text is in
   * lastError() */
  const int rc = gpx_engine_create(&cfg, &h);
  return rc == GPX_OK ? (jlong)(intptr_t)h : (jlong)rc;
}
JFN(jint, destroy)(JNIEnv* env, jclass c, jlong h) {
  (void)env, (void)c;
  return gpx_engine_destroy(H(h));
}
JFN(jstring, lastError)(JNIEnv* env, jclass c) {
  (void)c;
  return (*env)->NewStringUTF(env, gpx_last_error());
}
/* pins a direct ByteBuffer for DMA once, after allocation (gpx_host_register) */
JFN(jint, hostRegister)(JNIEnv* env, jclass c, jlong h, jobject buf) {
  (void)c;
  if (!buf) return GPX_EINVAL;
  void* p = (*env)->GetDirectBufferAddress(env, buf); /* NULL for a heap (non-direct) buffer */
  const jlong cap = (*env)->GetDirectBufferCapacity(env, buf); /* -1 for one */
  if (!p || cap <= 0) return GPX_EINVAL;
  return gpx_host_register(H(h), p, (size_t)cap);
}
/* a direct ByteBuffer over memory allocated for the DMA engines (gpx_host_alloc = hipHostMalloc): the batch columns of
 * a host that wants the link's full rate without pinning JVM memory afterwards; hostFree(buffer) gives it back */
JFN(jobject, hostAlloc)(JNIEnv* env, jclass c, jlong h, jlong bytes) {
  (void)c;
  void* p = NULL;
  if (bytes <= 0 || gpx_host_alloc(H(h), (size_t)bytes, &p) != GPX_OK) return NULL;
  jobject buf = (*env)->NewDirectByteBuffer(env, p, bytes);
  if (!buf) (void)gpx_host_free(H(h), p); /* (no direct-buffer support, or out of memory: the block must not stay until destroy) */
  return buf;
}
JFN(jint, hostFree)(JNIEnv* env, jclass c, jlong h, jobject buf) {
  (void)c;
  if (!buf) return GPX_EINVAL;
  return gpx_host_free(H(h), (*env)->GetDirectBufferAddress(env, buf));
}
/* PaxosManager.createPaxosInstance(Map, ...) batch create (PaxosManager.java:664-691) */
JFN(jint, groupCreate)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject members, jobject k,
                       jobject hriRows, jobject status) {
  (void)c;
  return gpx_group_create(H(h), n, B(gidx), B(members), B(k), B(hriRows), B(status));
}
/* PISM.tryPause / PaxosManager.kill (PISM:2004-2035, PaxosManager.java:2162) */
JFN(jint, groupRetire)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jint mode, jobject hriRows,
                       jobject status) {
  (void)c;
  return gpx_group_retire(H(h), n, B(gidx), mode, B(hriRows), B(status));
}
/* RequestBatcher.process -> PM.proposeBatched -> PCS.propose (RequestBatcher.java:79-81, PCS:233-263) */
JFN(jint, proposeBatch)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject isStop, jobject slot,
                        jobject bnum, jobject bcoord, jobject medianCp, jobject status) {
  (void)c;
  return gpx_propose_batch(H(h), n, B(gidx), B(isStop), B(slot), B(bnum), B(bcoord), B(medianCp), B(status));
}
/* PISM.handleAccept (PISM:1080-1166) */
JFN(jint, acceptBatch)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject bnum, jobject bcoord,
                       jobject slot, jobject medianCp, jobject aFlags, jobject rBnum, jobject rBcoord,
                       jobject rMaxCp, jobject rFlags, jobject status, jobject xGidx, jobject xFirst,
                       jobject xCount, jobject nRuns) {
  (void)c;
  return gpx_accept_batch(H(h), n, B(gidx), B(bnum), B(bcoord), B(slot), B(medianCp), B(aFlags), B(rBnum),
                          B(rBcoord), B(rMaxCp), B(rFlags), B(status), B(xGidx), B(xFirst), B(xCount),
                          B(nRuns));
}
/* PISM.handleBatchedAcceptReply / handleAcceptReply (PISM:1248-1419): one call per
 * PaxosPacketBatcher dequeue */
JFN(jint, acceptReplyBatch)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject bnum,
                            jobject bcoord, jobject slot, jobject acceptor, jobject maxCp, jobject dGidx,
                            jobject dSlot, jobject dBnum, jobject dBcoord, jobject dMedian, jobject dKind,
                            jobject nOut, jobject status) {
  (void)c;
  return gpx_accept_reply_batch(H(h), n, B(gidx), B(bnum), B(bcoord), B(slot), B(acceptor), B(maxCp),
                                B(dGidx), B(dSlot), B(dBnum), B(dBcoord), B(dMedian), B(dKind), B(nOut),
                                B(status));
}
/* The asynchronous twins (include/gpx.h): the call returns once its copies and kernels are queued; every
 * buffer stays untouched until engineWait(ticket) returns.  A ticket comes back through a one-element direct
 * LongBuffer. */
JFN(jint, proposeBatchAsync)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject isStop, jobject slot,
                             jobject bnum, jobject bcoord, jobject medianCp, jobject status, jobject ticket) {
  (void)c;
  return gpx_propose_batch_async(H(h), n, B(gidx), B(isStop), B(slot), B(bnum), B(bcoord), B(medianCp), B(status),
                                 (gpx_ticket*)B(ticket));
}
/* bnum == null && bcoord == null: every vote carries (commonBnum, commonBcoord) */
JFN(jint, acceptReplyBatchAsync)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject bnum,
                                 jobject bcoord, jint commonBnum, jint commonBcoord, jobject slot,
                                 jobject acceptor, jobject maxCp, jobject dGidx, jobject dSlot, jobject dBnum,
                                 jobject dBcoord, jobject dMedian, jobject dKind, jobject nOut, jobject status,
                                 jobject ticket) {
  (void)c;
  return gpx_accept_reply_batch_async(H(h), n, B(gidx), B(bnum), B(bcoord), commonBnum, commonBcoord, B(slot),
                                      B(acceptor), B(maxCp), B(dGidx), B(dSlot), B(dBnum), B(dBcoord), B(dMedian),
                                      B(dKind), B(nOut), B(status), (gpx_ticket*)B(ticket));
}
JFN(jint, acceptBatchAsync)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject bnum, jobject bcoord,
                            jobject slot, jobject medianCp, jobject aFlags, jobject rBnum, jobject rBcoord,
                            jobject rMaxCp, jobject rFlags, jobject status, jobject xGidx, jobject xFirst,
                            jobject xCount, jobject nRuns, jobject ticket) {
  (void)c;
  return gpx_accept_batch_async(H(h), n, B(gidx), B(bnum), B(bcoord), B(slot), B(medianCp), B(aFlags), B(rBnum),
                                B(rBcoord), B(rMaxCp), B(rFlags), B(status), B(xGidx), B(xFirst), B(xCount),
                                B(nRuns), (gpx_ticket*)B(ticket));
}
JFN(jint, commitBatchAsync)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject bnum, jobject bcoord,
                            jobject slot, jobject medianCp, jobject cKind, jobject status, jobject xGidx,
                            jobject xFirst, jobject xCount, jobject nRuns, jobject ticket) {
  (void)c;
  return gpx_commit_batch_async(H(h), n, B(gidx), B(bnum), B(bcoord), B(slot), B(medianCp), B(cKind), B(status),
                                B(xGidx), B(xFirst), B(xCount), B(nRuns), (gpx_ticket*)B(ticket));
}
JFN(jint, engineWait)(JNIEnv* env, jclass c, jlong h, jlong ticket) {
  (void)env, (void)c;
  return gpx_engine_wait(H(h), (gpx_ticket)ticket);
}
JFN(jint, setOrderedBatches)(JNIEnv* env, jclass c, jlong h, jint mask) {
  (void)env, (void)c;
  return gpx_engine_set_ordered_batches(H(h), mask);
}
/* PISM.handleBatchedCommit / handleCommittedRequest (PISM:1432-1528) */
JFN(jint, commitBatch)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject bnum, jobject bcoord,
                       jobject slot, jobject medianCp, jobject cKind, jobject status, jobject xGidx,
                       jobject xFirst, jobject xCount, jobject nRuns) {
  (void)c;
  return gpx_commit_batch(H(h), n, B(gidx), B(bnum), B(bcoord), B(slot), B(medianCp), B(cKind), B(status),
                          B(xGidx), B(xFirst), B(xCount), B(nRuns));
}
/* PISM.handlePrepare (PISM:900-1006) */
JFN(jint, prepareBatch)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject bnum, jobject bcoord,
                        jobject firstSlot, jobject rBnum, jobject rBcoord, jobject rGc, jobject rFlags,
                        jobject pMask, jobject pSlot, jobject pBnum, jobject pBcoord, jobject status) {
  (void)c;
  return gpx_prepare_batch(H(h), n, B(gidx), B(bnum), B(bcoord), B(firstSlot), B(rBnum), B(rBcoord), B(rGc),
                           B(rFlags), B(pMask), B(pSlot), B(pBnum), B(pBcoord), B(status));
}
/* PISM.checkRunForCoordinator's decision over all groups (PISM:2090-2176) */
JFN(jint, electionScan)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject down, jint nDown,
                        jobject longDead, jint nLongDead, jint force, jobject run, jobject pBnum,
                        jobject pFirst, jobject status) {
  (void)c;
  return gpx_election_scan(H(h), n, B(gidx), B(down), nDown, B(longDead), nLongDead, force, B(run),
                           B(pBnum), B(pFirst), B(status));
}
/* PISM.tryMakeCoordinator -> PaxosCoordinator.makeCoordinator (PISM:2178-2183) */
JFN(jint, electionBegin)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject bnum,
                         jobject eStatus) {
  (void)c;
  return gpx_election_begin(H(h), n, B(gidx), B(bnum), B(eStatus));
}
/* PISM.handlePrepareReply (PISM:1008-1068) */
JFN(jint, prepareReplyBatch)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jobject acceptor,
                             jobject rBnum, jobject rBcoord, jobject firstSlot, jobject pvOff,
                             jobject pvSlot, jobject pvBnum, jobject pvBcoord, jobject pvHandle,
                             jobject pvFlags, jobject vKind, jobject eCount, jobject eMedian,
                             jobject eSlot, jobject eKind, jobject eHandle, jobject eFlags,
                             jobject status) {
  (void)c;
  return gpx_prepare_reply_batch(H(h), n, B(gidx), B(acceptor), B(rBnum), B(rBcoord), B(firstSlot),
                                 B(pvOff), B(pvSlot), B(pvBnum), B(pvBcoord), B(pvHandle), B(pvFlags),
                                 B(vKind), B(eCount), B(eMedian), B(eSlot), B(eKind), B(eHandle),
                                 B(eFlags), B(status));
}
/* getMissingCommittedSlots / shouldSync (PaxosAcceptor.java:405-438, PISM:2341-2364) */
JFN(jint, gapScan)(JNIEnv* env, jclass c, jlong h, jint n, jobject gidx, jint threshold, jint syncMode,
                   jint sizeLimit, jobject firstSlot, jobject maxCommitted, jobject missing,
                   jobject shouldSync, jobject status) {
  (void)c;
  return gpx_gap_scan(H(h), n, B(gidx), threshold, syncMode, sizeLimit, B(firstSlot), B(maxCommitted),
                      B(missing), B(shouldSync), B(status));
}
#endif /* GPX_HAVE_JNI */

/* header drift check, compiled with or without a JDK: the shim's calls above must match these
 * prototypes (a mismatch in include/gpx.h breaks this translation unit) */
typedef int (*gpx_jni_check_ar)(gpx_engine*, int32_t, const int32_t*, const int32_t*, const int32_t*,
                                const int32_t*, const int32_t*, const int32_t*, int32_t*, int32_t*,
                                int32_t*, int32_t*, int32_t*, uint8_t*, int32_t*, uint8_t*);
static gpx_jni_check_ar gpx_jni_check_ar_ = gpx_accept_reply_batch;
void* gpx_jni_selfcheck(void) { return (void*)gpx_jni_check_ar_; }
