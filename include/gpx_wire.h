/*
 * gpx_wire.h — wire frames <-> structure-of-arrays, on the device (SURVEY.md §8(f) row 1, §8(a)
 * rows a5 / a10 / a15 / a16).
 *
 * The reference byteifies exactly four packet types (PaxosPacketDemultiplexerFast.toPaxosPacket,
 * paxosutil/PaxosPacketDemultiplexerFast.java:66-103): REQUEST, ACCEPT, BATCHED_COMMIT and
 * BATCHED_ACCEPT_REPLY, all big-endian java.nio.ByteBuffer layouts behind the common PaxosPacket
 * header (paxospackets/PaxosPacket.java:443-476):
 *
 *   int 90 (PAXOS_PACKET) | int paxosPacketType | int version | byte idLen | idLen bytes paxosID
 *
 * Today every such frame becomes a Java object tree (TreeMap<Integer,Long> per BatchedAcceptReply,
 * TreeSet per BatchedCommit, a String per paxosID) before PaxosManager.handlePaxosPacket
 * (PaxosManager.java:1126-1204) looks the instance up by name and checks its version.  Here the
 * raw frames of one NIO read burst go to HBM as ONE byte buffer + an offset array and come back as
 * the SoA columns the engine's batch calls take; the paxosID -> group row lookup and the version
 * check happen inside the same kernels (a device-resident open-addressing table keyed by the
 * paxosID bytes replaces MultiArrayMap.get, PaxosManager.java:1816-1832).  The opposite direction
 * packs the engine's decisions into BATCHED_COMMIT frames with PaxosPacketBatcher's coalescing rule.
 *
 * Conventions as in gpx.h.  Plain entry points take HOST pointers, the *_dev twins DEVICE pointers
 * (asynchronous on the engine's back-end stream; the column structs themselves live in host memory
 * and hold device pointers).  Request VALUES never cross: `frame` columns tell the host which frame
 * a record came from, so it can keep (gidx, slot) -> the frame's bytes.
 */
#ifndef GPX_WIRE_H
#define GPX_WIRE_H

#include "gpx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* PaxosPacket.PaxosPacketType ints (PaxosPacket.java:202-287) */
#define GPX_WT_PAXOS_PACKET 90
#define GPX_WT_REQUEST 1
#define GPX_WT_ACCEPT 3
#define GPX_WT_BATCHED_ACCEPT_REPLY 34
#define GPX_WT_BATCHED_COMMIT 35

/* per-frame status */
#define GPX_W_OK 0
#define GPX_W_NOGROUP 1     /* getInstance(paxosID) == null (PaxosManager.java:1153-1162) */
#define GPX_W_VERSION 2     /* pism.getVersion() != request.getVersion() (PaxosManager.java:1162) */
#define GPX_W_MALFORMED 3   /* the ByteBuffer constructor would throw (underflow, negative array \
                               size), or the frame is not a PAXOS_PACKET; also a non-ascending  \
                               slot list longer than GPX_W_MAX_UNSORTED */
#define GPX_W_UNSUPPORTED 4 /* a PaxosPacketType the fast demultiplexer does not byteify */
#define GPX_W_CAPACITY 5    /* the frame's records did not fit the output columns */

#define GPX_W_MAX_NAME 127      /* paxosIDLength is a signed byte (PaxosPacket.java:451) */
#define GPX_W_MAX_UNSORTED 1024 /* slot lists are ascending on the wire (TreeMap / TreeSet      \
                                   iteration); a list that is not is still accepted, sorted and  \
                                   de-duplicated like the Java constructors would, up to here */

/* ---- group names (PaxosManager.pinstances, PaxosManager.java:1816-1832) ------ */

/*
 * Binds paxosID i (bytes names[name_off[i] .. name_off[i+1]), ISO-8859-1, 1..127 bytes) to group
 * row gidx[i].  Names inside one call must be pairwise distinct (PaxosManager.createPaxosInstance
 * rejects duplicates against its own table first).  status[i]: GPX_S_OK, GPX_S_EXISTS (name or row
 * already bound), GPX_S_NOGROUP (gidx out of range / bad length).  Host pointers.
 */
int gpx_names_bind(gpx_engine* h, int32_t n, const uint8_t* names, const int32_t* name_off,
                   const int32_t* gidx, uint8_t* status);
/* unbinds the names of these rows (kill / pause); status: GPX_S_OK / GPX_S_NOGROUP */
int gpx_names_unbind(gpx_engine* h, int32_t n, const int32_t* gidx, uint8_t* status);
/* device-side lookup of n names (diagnostics / tests): gidx_out[i] = row or -1 */
int gpx_names_lookup(gpx_engine* h, int32_t n, const uint8_t* names, const int32_t* name_off,
                     int32_t* gidx_out);

/*
 * replaces: PISM.roundRobinCoordinator(paxosID, members, ballotnum)
 * (PaxosInstanceStateMachine.java:2251-2256): members[Math.abs(ballotnum + paxosID.hashCode()) %
 * members.length] from the bound name's String.hashCode, for n group rows.  out[i] = the node id,
 * or INT32_MIN where the Java would throw (Math.abs(Integer.MIN_VALUE) stays negative) or the row
 * has no name / no group.  Host pointers.
 */
int gpx_names_coordinator(gpx_engine* h, int32_t n, const int32_t* gidx, int32_t ballotnum,
                          int32_t* out);

/*
 * Dense row allocator for the caller's (paxosID, version) -> gidx map: LIFO free list over
 * [0, max_groups) (host side, control plane).  Returns GPX_ECAPACITY when fewer than n rows are
 * free (nothing allocated).
 */
int gpx_rows_alloc(gpx_engine* h, int32_t n, int32_t* gidx_out);
int gpx_rows_free(gpx_engine* h, int32_t n, const int32_t* gidx);

/* ---- decode: frames -> SoA ---------------------------------------------------- */

/* one vote per slot of a BATCHED_ACCEPT_REPLY (BatchedAcceptReply.java:103-117): the columns of
 * gpx_accept_reply_batch */
typedef struct gpx_wire_votes {
  int32_t cap;
  int32_t *gidx, *bnum, *bcoord, *slot, *acceptor, *max_cp;
  int32_t* frame; /* nullable: index of the source frame */
} gpx_wire_votes;
/* one record per slot of a BATCHED_COMMIT (BatchedCommit.java:156-170): the columns of
 * gpx_commit_batch, kind = 0 (meta-commit: the value comes from the stored ACCEPT) */
typedef struct gpx_wire_commits {
  int32_t cap;
  int32_t *gidx, *bnum, *bcoord, *slot, *median_cp;
  uint8_t* kind;
  int32_t* frame;
} gpx_wire_commits;
/* one record per ACCEPT (AcceptPacket.java:87-135 over RequestPacket.java:956-1020): the columns
 * of gpx_accept_batch plus the sender to reply to and the request id */
typedef struct gpx_wire_accepts {
  int32_t cap;
  int32_t *gidx, *bnum, *bcoord, *slot, *median_cp;
  uint8_t* flags; /* GPX_A_STOP */
  int32_t* sender;
  int64_t* req_id;
  int32_t* frame;
} gpx_wire_accepts;
/* one record per REQUEST (a client request or an already batched one = ONE proposal): the columns
 * of gpx_propose_batch */
typedef struct gpx_wire_requests {
  int32_t cap;
  int32_t* gidx;
  uint8_t* is_stop; /* RequestPacket.isStopRequest(): own flag or any batched request's */
  int64_t* req_id;
  int32_t* frame;
} gpx_wire_requests;

typedef struct gpx_wire_counts {
  int32_t n_votes, n_commits, n_accepts, n_requests; /* records the frames hold (may exceed cap) */
  int32_t n_bad_frames;                              /* frames with status != GPX_W_OK */
  int32_t reserved[3];
} gpx_wire_counts;

/*
 * replaces: PaxosPacketDemultiplexerFast.toPaxosPacket + the four ByteBuffer constructors +
 * PaxosManager.handlePaxosPacket's lookup / version check, for n_frames frames at once.
 * Frame i = frames[frame_off[i] .. frame_off[i+1]).  Records leave in frame order, the slots of one
 * frame ascending (TreeMap / TreeSet order).  Frames whose status is not GPX_W_OK contribute no
 * record (the reference drops such a packet).  f_gidx[i] = the group row (or -1), f_type[i] = the
 * PaxosPacketType int (or -1).  Any column struct may be NULL when the caller knows the burst holds
 * no such packets (their frames then get GPX_W_CAPACITY).
 */
int gpx_wire_decode(gpx_engine* h, int32_t n_frames, const uint8_t* frames,
                    const int64_t* frame_off, uint8_t* f_status, int32_t* f_gidx, int32_t* f_type,
                    const gpx_wire_votes* votes, const gpx_wire_commits* commits,
                    const gpx_wire_accepts* accepts, const gpx_wire_requests* requests,
                    gpx_wire_counts* counts);
int gpx_wire_decode_dev(gpx_engine* h, int32_t n_frames, const uint8_t* frames,
                        const int64_t* frame_off, uint8_t* f_status, int32_t* f_gidx,
                        int32_t* f_type, const gpx_wire_votes* votes,
                        const gpx_wire_commits* commits, const gpx_wire_accepts* accepts,
                        const gpx_wire_requests* requests, gpx_wire_counts* counts /* device */);

/* ---- encode: decisions -> BATCHED_COMMIT frames -------------------------------- */

/*
 * replaces: PaxosPacketBatcher.coalesce -> fuseBatchedCommits -> enqueueImpl(BatchedCommit) ->
 * dequeueImplC -> BatchedCommit.toBytes (PaxosPacketBatcher.java:121-156, 231-243, 389-414;
 * BatchedCommit.java:184-215) for the decisions of one gpx_accept_reply_batch call: all DECISION
 * rows of one (group, ballot) become ONE frame - slots ascending and distinct (TreeSet),
 * medianCheckpointedSlot folded with `b - cur > 0` in row order (BatchedCommit.java:104-112),
 * group = the members other than this node, ascending (SHORT_CIRCUIT_LOCAL, Util.arrayToIntSet).
 * PREEMPTED rows are not coalescable and are skipped.  Rows must be grouped by gidx (as the
 * engine emits them).  Frames leave ordered by their first row; frame f occupies
 * out[frame_off[f] .. frame_off[f] + frame_len[f]) (frame_off is 4-byte aligned), f_gidx[f] names
 * its group; *n_frames <= n, *n_bytes = bytes used.  n_dev (nullable, device int32) overrides n
 * with a count that is still in device memory (gpx_accept_reply_batch_dev's n_out).
 */
int gpx_wire_pack_commits(gpx_engine* h, int32_t n, const int32_t* d_gidx, const int32_t* d_slot,
                          const int32_t* d_bnum, const int32_t* d_bcoord,
                          const int32_t* d_median_cp, const uint8_t* d_kind, uint8_t* out,
                          int64_t cap_bytes, int64_t* frame_off, int32_t* frame_len,
                          int32_t* f_gidx, int32_t* n_frames, int64_t* n_bytes);
int gpx_wire_pack_commits_dev(gpx_engine* h, int32_t n, const int32_t* n_dev,
                              const int32_t* d_gidx, const int32_t* d_slot, const int32_t* d_bnum,
                              const int32_t* d_bcoord, const int32_t* d_median_cp,
                              const uint8_t* d_kind, uint8_t* out, int64_t cap_bytes,
                              int64_t* frame_off, int32_t* frame_len, int32_t* f_gidx,
                              int32_t* n_frames /* device */, int64_t* n_bytes /* device */);

/* ---- encode: accept replies -> BATCHED_ACCEPT_REPLY frames ----------------------- */

/*
 * replaces: PaxosPacketBatcher.coalesce -> enqueueImpl(AcceptReplyPacket) -> dequeueImplAR ->
 * BatchedAcceptReply.toBytes (PaxosPacketBatcher.java:121-137, 211-224, 353-388;
 * BatchedAcceptReply.java:119-173) for the replies of one gpx_accept_batch call.  Record i is the
 * reply to ACCEPT i: (gidx, slot) from the accept, (r_bnum, r_bcoord, r_maxcp, status) as
 * gpx_accept_batch returned them, sender / req_id (nullable) as gpx_wire_decode gave them.  A reply
 * exists iff status[i] == GPX_S_OK; it is coalescable iff its ballot's coordinator is the
 * ACCEPT's sender (allPositiveAcceptReplies, PaxosPacketBatcher.java:438-446).  All coalescable
 * replies of one (group, reply ballot) become ONE frame: acceptor = this node, the ballot, the
 * slot / maxCheckpointedSlot / requestID of the FIRST such reply in array order
 * (BatchedAcceptReply.java:49-54), then the TreeMap slot -> requestID (ascending, a repeated slot
 * keeps the last request id).  Frames leave grouped by gidx ascending, the ballots of one group in
 * first-appearance order; f_dest[f] = the ballot's coordinator, the node the frame goes to.
 * Engine limits: one PASS coalesces at most 256 replies and 4 distinct reply ballots of one group
 * (the reference's maps are unbounded).  The host-pointer call runs further passes over the replies
 * that were over a limit, in array order, until none is left: their frames follow the first pass's
 * (a group with 600 replies in one ballot leaves as three frames of 256 + 256 + 88 slots instead of
 * the reference's one).  The _dev call runs ONE pass and marks such replies unbatched[i] = 2: the
 * caller passes them again.  unbatched[i] (nullable) = 1 for an existing reply that can not be
 * packed at all (not coalescable, unnamed group): the host sends it as a plain ACCEPT_REPLY, which
 * is what the reference does with BATCHED_ACCEPT_REPLIES off.  frame_off is 4-byte aligned.
 */
int gpx_wire_pack_accept_replies(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* slot,
                                 const int32_t* sender, const int64_t* req_id,
                                 const int32_t* r_bnum, const int32_t* r_bcoord,
                                 const int32_t* r_maxcp, const uint8_t* status, uint8_t* unbatched,
                                 uint8_t* out, int64_t cap_bytes, int64_t* frame_off,
                                 int32_t* frame_len, int32_t* f_gidx, int32_t* f_dest,
                                 int32_t* n_frames, int64_t* n_bytes);
int gpx_wire_pack_accept_replies_dev(gpx_engine* h, int32_t n, const int32_t* gidx,
                                     const int32_t* slot, const int32_t* sender,
                                     const int64_t* req_id, const int32_t* r_bnum,
                                     const int32_t* r_bcoord, const int32_t* r_maxcp,
                                     const uint8_t* status, uint8_t* unbatched, uint8_t* out,
                                     int64_t cap_bytes, int64_t* frame_off, int32_t* frame_len,
                                     int32_t* f_gidx, int32_t* f_dest, int32_t* n_frames /* device */,
                                     int64_t* n_bytes /* device */);

/* ---- what leaves together: the batcher's payload bound and cross-group batching ------------- */

/*
 * replaces: PaxosPacketBatcher.dequeueImpl's payload bound and process() -> batch()
 * (PaxosPacketBatcher.java:182-209, 268-303) for the frames the pack calls produced.  Pure host
 * function (no engine, no device work: the messenger that sends the frames walks them anyway).
 * Frame f, in dequeue order (the reference drains accept replies, then commits, then accepts, then
 * requests into ONE list under one running estimate), carries est[f] = RequestPacket.SIZE_ESTIMATE *
 * Batched*.size() (a request: lengthEstimate()) and dest_key[f] = a key that is equal iff the
 * recipient sets are equal (BATCHED_ACCEPT_REPLY: the coordinator id, f_dest; BATCHED_COMMIT /
 * BATCHED_ACCEPT: an id of the group's member set).
 *   dequeueImpl: frames are taken `while (lengthEstimate < max_payload)` - the test precedes the
 *     add, so the frame that crosses the bound still leaves with this dequeue - burst[f] = the
 *     dequeue (ConsumerTask iteration) frame f leaves in.
 *   process(): a dequeue of MORE than min_batch tasks (PC.MIN_PP_BATCH_SIZE = 3) with
 *     batch_across_groups (PC.BATCH_ACROSS_GROUPS = true) is regrouped by recipient set in
 *     first-appearance order (LinkedHashMap<Set<Integer>, BatchedPaxosPacket>): envelope[f] = index
 *     of its BatchedPaxosPacket inside the burst, position[f] = its index inside that packet;
 *     otherwise every task is sent on its own: envelope[f] = -1, position[f] = 0.
 * *n_bursts = number of dequeues.  (BatchedPaxosPacket itself is a JSON envelope: its bytes are the
 * Java host's, PaxosPacket.java:443-476 has no byteified form for it.)
 */
int gpx_wire_plan_send(int32_t n_frames, const int64_t* est, const int64_t* dest_key,
                       int64_t max_payload, int32_t min_batch, int32_t batch_across_groups,
                       int32_t* burst, int32_t* envelope, int32_t* position, int32_t* n_bursts);

#ifdef __cplusplus
}
#endif
#endif /* GPX_WIRE_H */
