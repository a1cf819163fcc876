/*
 * gpx.h — C-ABI of the MI355X batched-consensus engine.
 *
 * This is the drop-in boundary for ONE hot path of MobilityFirst/gigapaxos: the
 * per-group PaxosInstanceStateMachine propose / accept / accept-reply / commit
 * pipeline.  The reference (100 % Java) has no FFI for this path; the seam is the
 * Java call PaxosManager.handlePaxosPacket -> PaxosInstanceStateMachine
 * .handlePaxosMessage (PaxosManager.java:1126-1204, PaxosInstanceStateMachine
 * .java:411-583).  Every entry point below replaces one branch of that switch
 * for a whole batch of (group, packet) records at once, and is what a JNI stub
 * in RequestBatcher.process / PaxosPacketBatcher.dequeueImpl / PaxosManager
 * .handlePaxosPacket would bind.
 *
 * Conventions
 *   - plain pointers + sizes; no C++ / torch types; no exception crosses the ABI.
 *   - every function returns 0 or a negative GPX_E* code (whole call rejected,
 *     no state changed).  Per-record soft errors go to a `status` column
 *     (GPX_S_*): such a record is DROPPED exactly like a lost / version-
 *     mismatched packet in the reference (PaxosInstanceStateMachine.java:441-460)
 *     and no state of its group changes.
 *   - all buffers are caller-owned and only valid for the duration of the call.
 *   - "gidx" is a dense group index in [0, max_groups): the Java host keeps the
 *     (paxosID, version) -> gidx map (it replaces MultiArrayMap,
 *     PaxosManager.java:1816-1832).  Values (request payloads) never cross the
 *     boundary; the host keeps (gidx, slot) -> RequestPacket.
 *   - ORDER: within one call, records of the same group are applied in array
 *     order, exactly as if handlePaxosMessage had been called once per record in
 *     that order.  Records of different groups are independent
 *     (PaxosManager.java:3170-3171).  Compacted outputs (decisions, exec runs)
 *     leave GROUPED BY GIDX ASCENDING; the entries of one group keep the array
 *     order of the records that produced them.  (The reference's next stage
 *     files outgoing decisions under their paxosID anyway:
 *     PaxosPacketBatcher.java:121-156; per group the decided stream is exactly
 *     the sequential one, across groups the reference defines no order.)
 *   - one submitting thread per engine at a time (the ConsumerTask single-
 *     consumer discipline, ConsumerTask.java:163-174); different engines are
 *     fully concurrent, any number of them per GPU.  (The one-launch kernels of
 *     small ordered batches exchange their verdict between workgroups that must
 *     all be resident.  The library enforces that itself: it counts the streams
 *     the live engines of a device launch on (engines given ONE stream with
 *     gpx_engine_set_stream are serialised by it and count once) and only
 *     launches such a kernel with a grid that fits the device together with one
 *     on every other stream - per CU at most two workgroups and at most one
 *     fewer than hipOccupancyMaxActiveBlocksPerMultiprocessor promises for that
 *     kernel, x CUs / streams; where that share gets small a call takes the
 *     two-launch form, with the same results.  The streams of OTHER PROCESSES on
 *     the device count too: every process keeps a file
 *     /dev/shm/gpx_engines/<PCI bus id>.<pid> with its number of streams, read
 *     by the others at most every 50 ms (files of dead processes are removed).
 *     Processes that do not share /dev/shm - separate containers - cannot see
 *     each other: tell each of them with GPX_DEVICE_SHARERS=<processes>, which
 *     then replaces the files.  A waiter that still starves gives up after ten
 *     seconds (GPX_XCHG_TIMEOUT_MS) and leaves a verdict that makes every
 *     workgroup arriving later refuse its records; the call during which that
 *     happens returns GPX_EDEVICE itself - from its own wait
 *     (the host-pointer calls, gpx_engine_wait, gpx_engine_sync, and
 *     gpx_group_create / retire / snapshot, which refuse to read or build on
 *     such a table) - and so does every later call: never a hang, never
 *     GPX_OK over partial outputs.  DESIGN.md 3.)
 *   - the plain entry points take HOST pointers (what a JNI direct ByteBuffer
 *     gives); the *_dev twins take DEVICE pointers, run asynchronously on the
 *     engine's stream (gpx_engine_set_stream) and leave counts in device memory
 *     (used when the batch is already resident in HBM).
 */
#ifndef GPX_H
#define GPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_ABI_VERSION 1
#define GPX_KMAX_LIMIT 16 /* PC.MAX_GROUP_SIZE = 16, PaxosConfig.java:532 */

/* whole-call errors */
#define GPX_OK 0
#define GPX_EINVAL (-1)    /* bad argument / null handle */
#define GPX_ECAPACITY (-2) /* n > max_batch, gidx space exhausted */
#define GPX_EDEVICE (-3)   /* HIP runtime error (gpx_last_error has text) */
#define GPX_ENOMEM (-4)

/* per-record status column (uint8) */
#define GPX_S_OK 0
#define GPX_S_NOGROUP 1 /* gidx out of range / no such instance: PaxosManager.java:1162-1194 */
#define GPX_S_STOPPED 2 /* acceptor stopped: PaxosInstanceStateMachine.java:456-460 */
#define GPX_S_WINDOW 3  /* slot outside the engine's fixed window; record not applied */
#define GPX_S_FORWARD 4 /* propose only: no coordinator here, forward to `bcoord`
                           (PaxosInstanceStateMachine.java:854-860) */
#define GPX_S_REFUSED 5 /* propose only: proposal after a stop, nothing sent
                           (PaxosCoordinatorState.java:235-239) */
#define GPX_S_EXISTS 6  /* group_create on a live gidx */
#define GPX_S_BUSY 7    /* group_retire(GPX_RETIRE_PAUSE) on a group that is not
                           caught up (PaxosInstanceStateMachine.java:2004-2035) */
#define GPX_S_UNORDERED 9 /* the batch broke the gpx_engine_set_ordered_batches promise at or before this record: refused */
#define GPX_S_PREACTIVE 8 /* propose only: the coordinator here is still being elected; the
                            proposal got slot `slot` but no ACCEPT goes out yet
                            (PaxosCoordinatorState.java:254-261) */

/* decision kinds (d_kind) */
#define GPX_D_DECISION 1  /* PValuePacket.makeDecision, PaxosCoordinatorState.java:630-635 */
#define GPX_D_PREEMPTED 2 /* PaxosCoordinatorState.java:661-675 */

/* accept-reply flag bits (r_flags) */
#define GPX_R_TOLOG 1  /* PaxosInstanceStateMachine.java:1146-1149 */
#define GPX_R_STORED 2 /* accept was put into acceptedProposals, PaxosAcceptor.java:315-316 */

/* accept flag bits (a_flags) / propose is_stop / commit kind bits */
#define GPX_A_STOP 1       /* request is a stop request */
#define GPX_C_HASVALUE 1   /* commit record is a full DECISION (value at host) rather
                              than a BATCHED_COMMIT slot */
#define GPX_C_STOP 2       /* (only with GPX_C_HASVALUE) the decision is a stop */

/* engine flags */
#define GPX_F_ACCEPTS_FROM_DISK 1u /* PaxosAcceptor.GET_ACCEPTED_PVALUES_FROM_DISK
                                      (PaxosAcceptor.java:75-76): default true */

/* group_retire modes */
#define GPX_RETIRE_PAUSE 0 /* only if caught up (tryPause) */
#define GPX_RETIRE_KILL 1  /* unconditional (PaxosManager.kill, PaxosManager.java:2162) */

typedef struct gpx_engine gpx_engine;

typedef struct gpx_config {
  int32_t my_id;      /* this node's integer id (PaxosInstanceStateMachine.getMyID) */
  int32_t max_groups; /* capacity of the gidx space (PC.PINSTANCES_CAPACITY) */
  int32_t kmax;       /* largest replica-group size that will be created (1..16) */
  int32_t window;     /* W: slots tracked per group per map; power of two, 4..64 */
  int32_t max_batch;  /* largest n of any batch call */
  int32_t device;     /* HIP device ordinal, -1 = current device */
  uint32_t flags;     /* GPX_F_* */
  uint32_t reserved;
} gpx_config;

/*
 * One group's pausable state == HotRestoreInfo (HotRestoreInfo.java:35-84) minus
 * the name.  gpx_group_create applies it verbatim (regular creation
 * and createHRI give different rows; the host decides which).
 */
typedef struct gpx_hri {
  int32_t version;
  int32_t acc_slot;   /* accSlot */
  int32_t acc_bnum;   /* accBallot.ballotNumber */
  int32_t acc_bcoord; /* accBallot.coordinatorID */
  int32_t acc_gc_slot;
  int32_t has_coord;  /* coordBallot != null */
  int32_t coord_bnum;
  int32_t coord_bcoord;
  int32_t next_proposal_slot;
  int32_t node_slots[GPX_KMAX_LIMIT]; /* first k entries used */
} gpx_hri;

/* ---- lifecycle ------------------------------------------------------------ */

/* replaces: new PaxosManager(...)'s instance table (PaxosManager.java:380-400) */
int gpx_engine_create(const gpx_config* cfg, gpx_engine** out);
int gpx_engine_destroy(gpx_engine* h);
/* text of the last GPX_EDEVICE on this thread ("" if none) */
const char* gpx_last_error(void);
/* run subsequent *_dev calls on this hipStream_t (NULL = the engine's own stream) */
int gpx_engine_set_stream(gpx_engine* h, void* hip_stream);
/*
 * Pins a caller-owned host buffer for DMA (hipHostRegister): the host-pointer entry points copy
 * their columns with asynchronous DMA from / to registered memory instead of the runtime's staged
 * copies of pageable memory.  A JNI host registers the memory of its direct ByteBuffers once, after
 * allocating them.  Purely an optimisation: unregistered pointers keep working.
 * Memory that is pinned already (hipHostMalloc, the caller's own hipHostRegister) is accepted and noted; the engine
 * never unpins it.  ONLY memory the engine was told about this way (or got from gpx_host_alloc) is written through a
 * device mapping by the asynchronous calls.  The device is synchronised before the pinning changes.  Prefer a few
 * large, page-aligned blocks (or gpx_host_alloc) to many small ones.
 * Host memory that is NOT pinned is copied in 512 KB pieces: the HIP runtime stages such pieces through its own pinned
 * buffer, where it would pin the caller's pages in place for the length of a larger copy - the path in which every GPU
 * page fault this repository has on file was raised.  Pageable calls therefore run at about half the
 * registered rate; nothing else changes for them.
 */
int gpx_host_register(gpx_engine* h, void* ptr, size_t bytes);
/*
 * Takes the pinning away again.  Waits first for everything the engine has in flight (its stream, the
 * asynchronous calls' copy-in stream and every copy-out stream): a call whose ticket was not waited for may still
 * be reading the block or writing through its device mapping.  Tickets stay valid.  gpx_engine_destroy
 * unregisters whatever is still registered through this engine (after the same wait).  An output buffer that
 * lies inside a registered block must fit in it with its full capacity (n entries): a buffer that runs over the
 * block's end is treated as unregistered memory (the copy path).
 */
int gpx_host_unregister(gpx_engine* h, void* ptr);
/*
 * Host memory allocated FOR the DMA engines (hipHostMalloc) instead of pinned afterwards: what a JNI host wraps
 * with NewDirectByteBuffer for its batch columns.  Treated like a registered block by every
 * entry point; gpx_host_free (which waits like gpx_host_unregister) or gpx_engine_destroy gives it back
 * (gpx_host_unregister on such a block: GPX_EINVAL).
 * bench.py's end_to_end.link reports what either kind of memory reaches on the box.
 */
int gpx_host_alloc(gpx_engine* h, size_t bytes, void** out);
int gpx_host_free(gpx_engine* h, void* ptr);
/* block until everything submitted to the engine has finished: its stream and the asynchronous calls' copy streams */
int gpx_engine_sync(gpx_engine* h);
/*
 * Promise about the batches of later calls (mask of GPX_ORDERED_*; 0 = none, the default).
 * Inside the pipeline a batch is usually the previous stage's output and already GROUPED BY GROUP:
 * RequestBatcher hands over one batched request per group (gidx strictly ascending), the ACCEPTs
 * follow that proposal batch, the commits are the decisions, which leave gpx_accept_reply_batch
 * grouped by gidx ascending (ORDER above).  The engine recognises such a batch on the device and
 * applies it without partitioning it; without a promise it also has to launch the partition path,
 * which then returns at once.  With GPX_ORDERED_PROPOSE (gidx in range and strictly ascending: every
 * group at most once) / GPX_ORDERED_ACCEPT / GPX_ORDERED_COMMIT (gidx in range and non-decreasing:
 * the records of a group adjacent, in their order) only the direct path is launched - one kernel per call.
 * The promise is VERIFIED on the device, inside that kernel: the FIRST VIOLATION of a batch is the first index
 * that is out of range or whose gidx is lower than (PROPOSE: not higher than) its predecessor's.  The records
 * before it are applied as usual; the records from it on are refused - status GPX_S_UNORDERED, all their outputs
 * zero, no state change - exactly like a lost tail of the batch (the caller sends them again, in order).  The first
 * violation is always the start of a run of equal gidx (equal neighbours are no violation), so no RUN is cut in two.
 * A group that comes back behind a descent is a second run: in gidx = [1,1,3,2,1] the first violation is index 3,
 * records 0-2 are applied (group 1's first run among them) and records 3-4 refused (group 1's second run among them) -
 * to Paxos the refused records are a lost tail, per group as for the batch.  (Until round 4 such a batch was refused
 * whole, which took a launch of its own for the verdict; applying the verified prefix keeps the guarantee that
 * matters - nothing is applied out of order, nothing silently - without it.)
 * Results of a batch that keeps the promise are identical with and without it.
 */
#define GPX_ORDERED_PROPOSE 1
#define GPX_ORDERED_ACCEPT 2
#define GPX_ORDERED_COMMIT 4
/*
 * Output form of the *_dev calls (mask bit, not a promise).  The compacted outputs of a call - execution runs,
 * decisions - are first PARKED at the indices of the records that produced them; a usual batch leaves them dense
 * already (every commit executes one slot: run i belongs to record i; ACCEPTs release nothing; every group of a
 * vote batch decides once), and only an unusual one needs a compaction pass.  By default the engine launches that
 * pass behind every call (it returns at once for a usual batch) so that the columns are ALWAYS dense and
 * *n_runs / *n_out >= 0 - at 5-6 us per idle launch.  With GPX_LAZY_OUTPUTS it does not: *n_runs / *n_out < 0
 * then says "this batch's outputs are still parked" and the caller - who reads the count anyway - calls
 * gpx_compact_last_dev, which makes the columns of the engine's most recent call dense and rewrites the count
 * (>= 0).  No other batch call may come in between: gpx_compact_last_dev then returns GPX_EINVAL (the parked
 * outputs of the earlier call are gone with its scratch).  The host-pointer calls never leave outputs parked.
 */
#define GPX_LAZY_OUTPUTS 32
/*
 * Accept replies.  What reaches a coordinator is the concatenation of what each acceptor sent, and an
 * acceptor's replies leave gpx_accept_batch in the order of the ACCEPT batch - grouped by group, groups
 * ascending: the vote batch is a few ASCENDING RUNS (one per acceptor), the shape the reference walks as one
 * TreeMap per acceptor, group and ballot (PISM.handleBatchedAcceptReply, PaxosInstanceStateMachine.java:
 * 1370-1419).  The engine applies such a batch without partitioning it: every group's votes are found in
 * every run and replayed in array order (run 0's before run 1's: arrival order).
 *   GPX_ORDERED_REPLY_RUNS  promise: gidx in range and at most GPX_REPLY_RUNS_MAX non-decreasing runs.
 *                           Verified on the device; a batch that breaks it is refused whole (every vote
 *                           GPX_S_UNORDERED, n_out = 0, no state change), like the other promises.
 *   GPX_TRY_REPLY_RUNS      hint, not a promise: the shape is checked on the device and any other batch
 *                           goes through the partition pipeline as before (a few idle launches dearer).
 * Results are identical on every path.  (A call of at most 1,024 votes that is not under the promise takes one launch
 * of one workgroup whatever its order - gigapaxos_amd/csrc/gpx_small.hip.h; a runs call of at most 65,536 votes one
 * launch too.  GPX_ORDERED_REPLY_RUNS keeps its refusal at every size.)
 */
#define GPX_ORDERED_REPLY_RUNS 8
#define GPX_TRY_REPLY_RUNS 16
#define GPX_REPLY_RUNS_MAX 16 /* = GPX_KMAX_LIMIT acceptors */
int gpx_engine_set_ordered_batches(gpx_engine* h, int32_t mask);
/* GPX_LAZY_OUTPUTS: compacts the outputs of the engine's most recent gpx_accept_batch_dev / gpx_commit_batch_dev /
 * gpx_accept_reply_batch_dev call in place (the same device columns) and rewrites its count word; a no-op when they
 * are dense already.  Asynchronous, on the engine's stream. */
int gpx_compact_last_dev(gpx_engine* h);

/*
 * replaces: PaxosManager.createPaxosInstance(Map nameStates, gms) batch create
 * (PaxosManager.java:664-691) -> new PaxosInstanceStateMachine(..., hri, ...)
 * -> hotRestore (PaxosInstanceStateMachine.java:677-690).
 * members: n x kmax int32, each row ascending (PaxosInstanceStateMachine.java:205),
 * unused tail ignored; k[i] = group size.
 * A coordinator is created only when rows[i].has_coord and coord_bcoord == my_id
 * (PaxosInstanceStateMachine.java:682-684).
 */
int gpx_group_create(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* members,
                     const uint8_t* k, const gpx_hri* rows, uint8_t* status);

/*
 * replaces: PaxosInstanceStateMachine.tryPause (PISM:2004-2035) / PaxosManager.kill.
 * rows (nullable) receives the HotRestoreInfo of each retired group.
 */
int gpx_group_retire(gpx_engine* h, int32_t n, const int32_t* gidx, int32_t mode, gpx_hri* rows,
                     uint8_t* status);

/* read-only snapshot of the HotRestoreInfo rows (no state change) */
int gpx_group_snapshot(gpx_engine* h, int32_t n, const int32_t* gidx, gpx_hri* rows,
                       uint8_t* status);

/*
 * Canonical dump of ONE group's full protocol state for parity checks
 * (int32 words; layout in docs/HISTORY.md §state-dump).  Returns the number of words
 * written (<= cap) or a negative error.
 */
int gpx_group_dump(gpx_engine* h, int32_t gidx, int32_t* buf, int32_t cap);

/* ---- data path (host pointers) ---------------------------------------------- */

/*
 * replaces: RequestBatcher.process -> PaxosManager.proposeBatched -> PISM.handleRequest
 * -> handleProposal -> PaxosCoordinatorState.propose + initCommander
 * (RequestBatcher.java:79-81, PaxosInstanceStateMachine.java:767-888,
 *  PaxosCoordinatorState.java:233-263, 841-851).
 * One record = one (already batched) RequestPacket for group gidx[i].
 * is_stop (nullable): 1 if the request is a stop request.
 * status[i]==GPX_S_OK: an ACCEPT(bnum,bcoord,slot,median_cp) must be multicast.
 */
int gpx_propose_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                      int32_t* slot, int32_t* bnum, int32_t* bcoord, int32_t* median_cp,
                      uint8_t* status);

/*
 * replaces: PISM.handleAccept (PaxosInstanceStateMachine.java:1080-1166) ->
 * PaxosAcceptor.acceptAndUpdateBallot (PaxosAcceptor.java:302-322), then
 * reconstructDecision -> handleCommittedRequest -> extractExecuteAndCheckpoint.
 * Dense outputs (one per record): the ACCEPT_REPLY (r_bnum, r_bcoord, slot echoed by
 * caller, r_maxcp = acceptor slot - 1), r_flags = GPX_R_*, status.
 * Compacted outputs: in-order execution runs (x_gidx, x_first, x_count), *n_runs of them.
 */
int gpx_accept_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                     const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                     const uint8_t* a_flags, int32_t* r_bnum, int32_t* r_bcoord,
                     int32_t* r_maxcp, uint8_t* r_flags, uint8_t* status, int32_t* x_gidx,
                     int32_t* x_first, int32_t* x_count, int32_t* n_runs);

/*
 * replaces: PISM.handleBatchedAcceptReply / handleAcceptReply
 * (PaxosInstanceStateMachine.java:1248-1419) -> PaxosCoordinator.handleAcceptReply
 * (PaxosCoordinator.java:210-250) -> PaxosCoordinatorState.handleAcceptReplyMyBallot /
 * handleAcceptReplyHigherBallot (PaxosCoordinatorState.java:597-683).
 * One record = one vote (one slot of a BATCHED_ACCEPT_REPLY).  Compacted outputs: one
 * entry per DECISION / PREEMPTED, grouped by gidx (ORDER above); *n_out of them (<= n).
 * status (nullable): per-vote GPX_S_*.
 * The output columns hold n entries each and overlap neither each other nor the inputs; entries at and beyond
 * *n_out are unspecified after the call (the device path writes a steady-state batch's outputs in place while it is
 * still reading votes, and may leave predicted entries behind when the batch turns out not to be one: DESIGN.md 3.2).
 */
int gpx_accept_reply_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                           const int32_t* bcoord, const int32_t* slot, const int32_t* acceptor,
                           const int32_t* max_cp, int32_t* d_gidx, int32_t* d_slot,
                           int32_t* d_bnum, int32_t* d_bcoord, int32_t* d_median_cp,
                           uint8_t* d_kind, int32_t* n_out, uint8_t* status);

/*
 * replaces: PISM.handleBatchedCommit / handleCommittedRequest
 * (PaxosInstanceStateMachine.java:1432-1528) -> extractExecuteAndCheckpoint
 * (:1619-1701) -> PaxosAcceptor.putAndRemoveNextExecutable (PaxosAcceptor.java:325-366).
 * One record = one committed slot.  c_kind (nullable) = GPX_C_* bits.
 * Compacted outputs: execution runs as in gpx_accept_batch.
 */
int gpx_commit_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                     const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                     const uint8_t* c_kind, uint8_t* status, int32_t* x_gidx, int32_t* x_first,
                     int32_t* x_count, int32_t* n_runs);

/* ---- data path (device pointers, asynchronous on the engine stream) ---------- */
/* Same semantics; every pointer is device memory; n_out / n_runs are device int32. */

int gpx_propose_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                          int32_t* slot, int32_t* bnum, int32_t* bcoord, int32_t* median_cp,
                          uint8_t* status);
int gpx_accept_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                         const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                         const uint8_t* a_flags, int32_t* r_bnum, int32_t* r_bcoord,
                         int32_t* r_maxcp, uint8_t* r_flags, uint8_t* status, int32_t* x_gidx,
                         int32_t* x_first, int32_t* x_count, int32_t* n_runs);
int gpx_accept_reply_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx,
                               const int32_t* bnum, const int32_t* bcoord, const int32_t* slot,
                               const int32_t* acceptor, const int32_t* max_cp, int32_t* d_gidx,
                               int32_t* d_slot, int32_t* d_bnum, int32_t* d_bcoord,
                               int32_t* d_median_cp, uint8_t* d_kind, int32_t* n_out,
                               uint8_t* status);
int gpx_commit_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                         const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                         const uint8_t* c_kind, uint8_t* status, int32_t* x_gidx,
                         int32_t* x_first, int32_t* x_count, int32_t* n_runs);

/* ---- data path (host pointers, ASYNCHRONOUS: copies in, kernels and copies out overlap across calls) ---- */
/*
 * The plain host-pointer calls above are synchronous: H2D of every input column, the kernels, D2H of the
 * outputs, one after the other, and the caller waits - a 3 M-vote accept-reply call is 95 % PCIe time and
 * uses one direction of the link at a time.  The *_async twins return as soon as the work is QUEUED:
 *   - inputs go to the device on a copy stream of their own, the kernels wait for them on the engine's
 *     stream (calls are applied in submission order, like every other call), the dense outputs and the
 *     output COUNT come back on a third stream;
 *   - gpx_engine_wait(ticket) blocks until that call is through, then fetches exactly *n_out / *n_runs
 *     entries of the compacted outputs (never the full capacity) and fills in the count.
 * While call N's outputs travel to the host, call N + 1's inputs travel to the device and its kernels run:
 * both directions of the link are busy.  Up to GPX_ASYNC_DEPTH calls may be in flight; one more returns
 * GPX_EBUSY (wait for the oldest ticket first).  EVERY buffer of a call - inputs and outputs - must stay
 * valid and untouched until its gpx_engine_wait returns (a JNI caller keeps a ring of direct ByteBuffers),
 * and should be pinned (gpx_host_register): the runtime copies pageable memory synchronously, which
 * serialises everything again.  Tickets may be waited for in any order; gpx_engine_destroy and
 * gpx_engine_sync complete the device work but not the host-side fetch of a ticket nobody waited for.
 * Results are those of the synchronous calls.
 *
 * gpx_accept_reply_batch_async takes bnum == NULL && bcoord == NULL to say "every vote of this batch
 * carries the ballot (common_bnum, common_bcoord)" - what a coordinator in its steady state receives
 * (PISM.handleBatchedAcceptReply unpacks one ballot per BatchedAcceptReply, PaxosInstanceStateMachine.java:
 * 1380-1387): 16 instead of 24 bytes per vote cross the link.
 */
#define GPX_ASYNC_DEPTH 4     /* calls in flight an engine takes by default ... */
#define GPX_ASYNC_DEPTH_MAX 8 /* ... and at most: environment GPX_ASYNC_DEPTH=n at engine creation (each call in flight holds
                                 a set of device columns of max_batch entries; six = three steps of two calls: bench.py's
                                 end-to-end leg, whose copy-out is slower than its copy-in) */
#define GPX_EBUSY (-5) /* the engine's depth of calls is in flight, or the ticket is unknown / already waited for */
typedef uint64_t gpx_ticket;
int gpx_propose_batch_async(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop, int32_t* slot,
                            int32_t* bnum, int32_t* bcoord, int32_t* median_cp, uint8_t* status,
                            gpx_ticket* ticket);
int gpx_accept_batch_async(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                           const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                           const uint8_t* a_flags, int32_t* r_bnum, int32_t* r_bcoord, int32_t* r_maxcp,
                           uint8_t* r_flags, uint8_t* status, int32_t* x_gidx, int32_t* x_first,
                           int32_t* x_count, int32_t* n_runs, gpx_ticket* ticket);
int gpx_accept_reply_batch_async(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                                 const int32_t* bcoord, int32_t common_bnum, int32_t common_bcoord,
                                 const int32_t* slot, const int32_t* acceptor, const int32_t* max_cp,
                                 int32_t* d_gidx, int32_t* d_slot, int32_t* d_bnum, int32_t* d_bcoord,
                                 int32_t* d_median_cp, uint8_t* d_kind, int32_t* n_out, uint8_t* status,
                                 gpx_ticket* ticket);
int gpx_commit_batch_async(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                           const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                           const uint8_t* c_kind, uint8_t* status, int32_t* x_gidx, int32_t* x_first,
                           int32_t* x_count, int32_t* n_runs, gpx_ticket* ticket);
int gpx_engine_wait(gpx_engine* h, gpx_ticket ticket);

/* ---- view change, acceptor side ------------------------------------------------ */

#define GPX_P_NACK 1  /* the acceptor's ballot is higher than the prepare's: no pvalues returned */
#define GPX_P_TOLOG 2 /* the acceptor's ballot went up: the PREPARE must be logged before the reply
                         leaves (LogMessagingTask, PaxosInstanceStateMachine.java:985-993) */
/*
 * replaces: PISM.handlePrepare (PaxosInstanceStateMachine.java:900-1006) ->
 * PaxosAcceptor.handlePrepare (PaxosAcceptor.java:239-273) + pruneAcceptedProposals (:283-293) +
 * getMaxGCSlotFirstUndecidedSlot (:275-280) for n PREPAREs at once (a coordinator node dies and
 * every group it led runs an election: each surviving acceptor receives one PREPARE per group).
 * Record i = PreparePacket(ballot (bnum, bcoord), first_slot = firstUndecidedSlot) for gidx[i].
 * Dense outputs: the PREPARE_REPLY's ballot (r_bnum, r_bcoord), r_gc = max(acceptedGCSlot,
 * firstUndecidedSlot - 1) (wraparound-aware), r_flags = GPX_P_*, and the in-memory accepted
 * pvalues with slot - firstUndecidedSlot >= 0 of an acknowledged prepare as `window` planes of n
 * entries each: p_mask[i] bit w set => (p_slot[w * n + i], p_bnum[w * n + i], p_bcoord[w * n + i])
 * is one (the host adds the disk-logged accepts, GET_ACCEPTED_PVALUES_FROM_DISK, and the values).
 * status: GPX_S_OK, GPX_S_STOPPED (no reply), GPX_S_NOGROUP.  Only the acceptor's ballot changes.
 * Not modelled: the re-send of this node's own pending higher prepare (coordinators are always
 * active here, PaxosCoordinator.getPendingBallot == null).
 */
int gpx_prepare_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                      const int32_t* bcoord, const int32_t* first_slot, int32_t* r_bnum,
                      int32_t* r_bcoord, int32_t* r_gc, uint8_t* r_flags, uint64_t* p_mask,
                      int32_t* p_slot, int32_t* p_bnum, int32_t* p_bcoord, uint8_t* status);
int gpx_prepare_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                          const int32_t* bcoord, const int32_t* first_slot, int32_t* r_bnum,
                          int32_t* r_bcoord, int32_t* r_gc, uint8_t* r_flags, uint64_t* p_mask,
                          int32_t* p_slot, int32_t* p_bnum, int32_t* p_bcoord, uint8_t* status);

/* ---- view change, coordinator side ---------------------------------------------- */

/* gpx_election_begin status */
#define GPX_EB_PREPARING 0 /* new coordinator state created, waiting for PREPARE replies */
#define GPX_EB_ACTIVE 1    /* ballot number 0: active at once (PaxosCoordinator.java:74-76) */
#define GPX_EB_RESEND 2    /* same ballot, still not active: send the PREPARE again */
#define GPX_EB_UNCHANGED 3 /* a coordinator with a ballot at least as high exists */
/*
 * replaces: PISM.tryMakeCoordinator -> PaxosCoordinator.makeCoordinator(c, bnum, myID, members,
 * paxosState.getSlot(), false) (PaxosInstanceStateMachine.java:2178-2183,
 * PaxosCoordinator.java:66-89) for the groups gpx_election_scan selected: a coordinator with a
 * lower ballot (or none) is replaced by a fresh PaxosCoordinatorState(bnum, myID, acceptor slot,
 * members, null) - nextProposalSlot = the acceptor's slot, nodeSlotNumbers = -1, no proposals -
 * whose prepare() arms waitforMyBallot.  gidx must be pairwise distinct (GPX_EINVAL otherwise; the
 * _dev twin cannot check and requires it).  e_status: GPX_EB_* (255 for a missing group).  Host pointers.
 */
int gpx_election_begin(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                       uint8_t* e_status);

/* gpx_prepare_reply_batch: what a reply did (v_kind) and what an entry of the lists is (e_kind) */
#define GPX_V_IGNORED 0   /* no coordinator, not waiting, lower ballot, non-member or repeated acceptor
                             (PaxosCoordinatorState.canIgnorePrepareReply, PCS:285-316) */
#define GPX_V_RECORDED 1  /* counted; no majority yet */
#define GPX_V_ELECTED 2   /* majority: the coordinator is active now; e_* = the ACCEPTs to multicast */
#define GPX_V_PREEMPTED 3 /* higher ballot: the coordinator is gone; e_* = its pre-active proposals,
                             to be forwarded to the reply ballot's coordinator (PISM:1042-1048) */
#define GPX_E_CARRY 1     /* a pvalue carried over from the replies: value = the one with e_handle */
#define GPX_E_NOOP 2      /* no-op for a slot neither carried over nor proposed (PCS:407-411) */
#define GPX_E_PREACTIVE 3 /* a proposal made while not active: e_handle = its handle */
#define GPX_E_NEWSTOP 4   /* the stop request processStop appends (PCS:512-516) */
#define GPX_PV_STOP 1     /* pv_flags / e_flags: the request is a stop request */
#define GPX_PV_NOOP 2     /* pv_flags: the carried request value is the NO_OP value */
/*
 * replaces: PISM.handlePrepareReply (PaxosInstanceStateMachine.java:1008-1068) ->
 * PaxosCoordinator.getPreActivesIfPreempted / handlePrepareReply (PaxosCoordinator.java:264-310)
 * -> PaxosCoordinatorState.isPreemptable, isPrepareAcceptedByMajority (carry-over of the highest
 * ballot pvalue per slot), combinePValuesOntoProposals, reproposePreemptedProposals, processStop,
 * spawnCommandersForProposals, setCoordinatorActive (PaxosCoordinatorState.java:271-587).
 * Record i = PrepareReplyPacket(acceptor, ballot (r_bnum, r_bcoord), accepted pvalues, minSlot =
 * the reply's gcSlot) for gidx[i]; its accepted pvalues are entries pv_off[i] .. pv_off[i+1] of the
 * pv_* columns ({slot, ballot, flags} + pv_handle, the caller's 64-bit key of the value: two
 * requests are RequestPacket.equals iff their handles are equal).
 * Outputs per record: v_kind; for ELECTED / PREEMPTED e_count[i] list entries in `window` planes
 * of n entries (entry j of record i at [j * n + i], ascending slot): e_slot, e_kind, e_handle,
 * e_flags; for ELECTED e_median[i] = the medianCheckpointedSlot of the ACCEPTs
 * (getMajorityCommittedSlot at initCommander, PCS:841-851).  status: GPX_S_OK, GPX_S_NOGROUP,
 * GPX_S_STOPPED is not applied (PISM handles prepare replies of a stopped instance alike),
 * GPX_S_WINDOW = the carried-over range and the re-proposed pre-actives do not fit `window` slots:
 * the record is dropped whole (nothing recorded) and the host must run this election itself.
 * Java assertions are taken as disabled (production): processStop's stop / request conversions
 * are unreachable then, because ProposalStateAtCoordinator re-stamps every pvalue with the
 * coordinator's own ballot (PCS:153-157), so only its final "append a stop" step acts.
 */
int gpx_prepare_reply_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* acceptor,
                            const int32_t* r_bnum, const int32_t* r_bcoord, const int32_t* first_slot,
                            const int32_t* pv_off, const int32_t* pv_slot, const int32_t* pv_bnum,
                            const int32_t* pv_bcoord, const int64_t* pv_handle,
                            const uint8_t* pv_flags, uint8_t* v_kind, int32_t* e_count,
                            int32_t* e_median, int32_t* e_slot, uint8_t* e_kind, int64_t* e_handle,
                            uint8_t* e_flags, uint8_t* status);

/*
 * gpx_propose_batch with the caller's 64-bit handle of each request (nullable).  Only needed for
 * groups whose coordinator is being elected: their proposals are kept as pre-active proposals
 * (status GPX_S_PREACTIVE) and come back, by handle, from gpx_prepare_reply_batch.
 */
int gpx_propose_batch_h(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                        const int64_t* handle, int32_t* slot, int32_t* bnum, int32_t* bcoord,
                        int32_t* median_cp, uint8_t* status);

/*
 * Device-pointer twins of the three calls above (conventions of the other *_dev calls: asynchronous
 * on the engine's streams, columns resident in HBM).  gpx_prepare_reply_batch_dev takes the number
 * of pvalue entries as pv_total (= pv_off[n]); the offsets are not validated on the host - a record
 * whose slice falls outside [0, pv_total] is refused with GPX_S_WINDOW.
 */
int gpx_election_begin_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                           uint8_t* e_status);
int gpx_propose_batch_h_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                            const int64_t* handle, int32_t* slot, int32_t* bnum, int32_t* bcoord,
                            int32_t* median_cp, uint8_t* status);
int gpx_prepare_reply_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* acceptor,
                                const int32_t* r_bnum, const int32_t* r_bcoord,
                                const int32_t* first_slot, const int32_t* pv_off, int32_t pv_total,
                                const int32_t* pv_slot, const int32_t* pv_bnum,
                                const int32_t* pv_bcoord, const int64_t* pv_handle,
                                const uint8_t* pv_flags, uint8_t* v_kind, int32_t* e_count,
                                int32_t* e_median, int32_t* e_slot, uint8_t* e_kind,
                                int64_t* e_handle, uint8_t* e_flags, uint8_t* status);

/* ---- retransmission: what is waiting for replies ------------------------------- */

#define GPX_POKE_NONE 0
#define GPX_POKE_ACCEPT 1  /* the active coordinator is commandering the acceptor's next slot */
#define GPX_POKE_PREPARE 2 /* the coordinator is not active: its PREPARE is outstanding */
/*
 * replaces: the state half of the per-message preamble (PaxosInstanceStateMachine.java:480-492):
 * pokeLocalCoordinator (PISM:2268-2279) -> PaxosCoordinator.reissueAcceptIfWaitingTooLong(c,
 * paxosState.getSlot()) (PaxosCoordinator.java:334-350) -> isCommandering(slot) + reInitCommander
 * (PaxosCoordinatorState.java:741-750, 841-851), and the PREPARE resend branch of
 * checkRunForCoordinator (PISM:2169-2181) -> remakeCoordinator -> prepare() (PCS:214-220).
 * For group gidx[i] (gidx == NULL: groups 0 .. n-1): poke[i] = GPX_POKE_ACCEPT with the ACCEPT to
 * multicast again {slot = the acceptor's slot, ballot, median_cp = getMajorityCommittedSlot() NOW,
 * p_flags = GPX_PV_STOP bit, heard = members that already replied (bit per member index)} iff the
 * coordinator is active and still holds a proposal for that slot - only the head-of-line slot is ever
 * re-sent; GPX_POKE_PREPARE with {ballot, slot = the acceptor's slot = PreparePacket.firstUndecidedSlot,
 * heard = waitforMyBallot's mask} iff a coordinator exists and is not active.  The clocks stay with
 * the host (testAndSetWaitingTooLong's ACCEPT_TIMEOUT / PREPARE_TIMEOUT with exponential backoff, PCS:
 * 715-739): it calls this for the groups whose batch timer expired.  No state changes.  Host pointers.
 */
int gpx_poke_scan(gpx_engine* h, int32_t n, const int32_t* gidx, uint8_t* poke, int32_t* slot,
                  int32_t* bnum, int32_t* bcoord, int32_t* median_cp, uint8_t* p_flags,
                  uint32_t* heard, uint8_t* status);

/* ---- view change: who runs for coordinator ------------------------------------ */

#define GPX_RUN_NO 0
#define GPX_RUN_MINE 1      /* I am my acceptor's ballot coordinator but hold no such coordinator */
#define GPX_RUN_NEXT 2      /* the ballot coordinator is down and I am next in line */
#define GPX_RUN_LONGDEAD 3  /* it has been down for long (lastCoordinatorLongDead) */
#define GPX_RUN_FORCED 4    /* forceRun */
/*
 * replaces: the decision of PISM.checkRunForCoordinator (PaxosInstanceStateMachine.java:2090-2176)
 * for n groups at once - the scan a node runs when the failure detector declares another node dead
 * (PaxosManager.isNodeUp / lastCoordinatorLongDead: passed in as the lists of node ids that are
 * down / long dead).  For group gidx[i] (gidx == NULL: groups 0 .. n-1) with acceptor ballot
 * (b, c):   run iff  !(coordinator != null && its ballot >= (b, c))  &&  ( c == me  ||
 * ( c is down && ( me == next member after c (members ascending, wrapping)  ||  c is long dead ) ) )
 * or force.  run[i] = GPX_RUN_*; for a running group the PREPARE to multicast is ballot
 * (p_bnum[i] = b + 1, my_id) with firstUndecidedSlot p_first[i] = the acceptor's slot
 * (PISM:2153-2160).  status: GPX_S_OK / GPX_S_NOGROUP.  Not modelled (time based, host side):
 * ranRecently, waitingTooLong (PREPARE retransmission), notRunYet; when the ballot coordinator is not
 * a member the Java picks a random member as "next" - here nobody is next.  No state changes: the
 * host hands a running group to its own coordinator code.  Host pointers.
 */
int gpx_election_scan(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* down_nodes,
                      int32_t n_down, const int32_t* long_dead_nodes, int32_t n_long_dead,
                      int32_t force, uint8_t* run, int32_t* p_bnum, int32_t* p_first,
                      uint8_t* status);

/* ---- batching of client requests (RequestBatcher) ----------------------------- */

/*
 * replaces: RequestBatcher.enqueueImpl / dequeueImpl (RequestBatcher.java:111-129, 163-239) for a
 * whole burst of client requests.  Record i = one RequestPacket for group gidx[i] with
 * est_bytes[i] = lengthEstimate() and weight[i] = batchSize() + 1 (NULL = 1), is_stop nullable.
 * Per group, in array order (the per-paxosID FIFO), requests are split greedily into consecutive
 * batches: a request opens a new batch when adding it would push the byte total over max_bytes
 * (min(NIOTransport.MAX_PAYLOAD_SIZE, SQLPaxosLogger.MAX_LOG_MESSAGE_SIZE)) or the size total over
 * max_size (PC.MAX_BATCH_SIZE = 2000); a batch head is taken whatever its own size.
 * Dense output: leader[i] = index of the request record i is latched onto (i itself for a head;
 * -1 if gidx[i] is out of range, status[i] = GPX_S_NOGROUP).  Compacted output, grouped by gidx
 * ascending, the batches of one group in FIFO order: b_gidx, b_leader, b_count,
 * b_bytes, b_size (the totals), b_stop (any member is a stop request), *n_batches of them.
 * One batch = one record of gpx_propose_batch.
 */
int gpx_request_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* est_bytes,
                      const int32_t* weight, const uint8_t* is_stop, int32_t max_bytes,
                      int32_t max_size, int32_t* leader, uint8_t* status, int32_t* b_gidx,
                      int32_t* b_leader, int32_t* b_count, int32_t* b_bytes, int32_t* b_size,
                      uint8_t* b_stop, int32_t* n_batches);
int gpx_request_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* est_bytes,
                          const int32_t* weight, const uint8_t* is_stop, int32_t max_bytes,
                          int32_t max_size, int32_t* leader, uint8_t* status, int32_t* b_gidx,
                          int32_t* b_leader, int32_t* b_count, int32_t* b_bytes, int32_t* b_size,
                          uint8_t* b_stop, int32_t* n_batches);

/* ---- gap detection ------------------------------------------------------------ */

#define GPX_SYNC_DEFAULT 0  /* SyncMode.DEFAULT_SYNC */
#define GPX_SYNC_TO_PAUSE 1 /* SyncMode.SYNC_TO_PAUSE */
#define GPX_SYNC_FORCE 2    /* SyncMode.FORCE_SYNC */
/*
 * replaces: PaxosAcceptor.getMaxCommittedSlot / getMissingCommittedSlots
 * (PaxosAcceptor.java:405-438) and PISM.shouldSync (PaxosInstanceStateMachine.java:2341-2364,
 * the decision behind syncLongDecisionGaps :1550-1570) for n groups at once.  Outputs per listed
 * group: first_slot = getSlot(), max_committed = getMaxCommittedSlot(), should_sync (with
 * `threshold` = PaxosManager.getOutOfOrderLimit() / getMaxSyncDecisionsGap()), and missing = a
 * bit mask over first_slot + j, j < min(size_limit, 64): bit set = slot neither committed with its
 * value nor meta-committed with a stored accept.  status: GPX_S_OK, GPX_S_STOPPED (missing = 0, as
 * the Java returns null), GPX_S_NOGROUP.  The time test (canSync) and the SYNC_DECISIONS request
 * packet stay in the host.  Host pointers.
 */
int gpx_gap_scan(gpx_engine* h, int32_t n, const int32_t* gidx, int32_t threshold,
                 int32_t sync_mode, int32_t size_limit, int32_t* first_slot,
                 int32_t* max_committed, uint64_t* missing, uint8_t* should_sync, uint8_t* status);

/* ---- sharding across the GPUs of a node ------------------------------------------------------ */

#define GPX_ROUTE_MAX_SHARDS 16
#define GPX_ROUTE_MAX_COLS 8
/*
 * device = fmix32(gidx) % n_shards (murmur3 finaliser); "the host bins each batch by
 * device".  This does the binning on the device for a batch already resident in HBM: a STABLE
 * partition of n_cols int32 columns (cols[0] = the global group index) by shard - the records of a
 * shard keep the batch's order, so the per-group ordering contract survives the routing - with the
 * group index rewritten to the shard-local dense index through g2l ([n_groups_global] device table,
 * NULL = keep the global index).  out_cols[k] receives column k, shard-major: shard s occupies
 * [shard_off[s], shard_off[s + 1]) (shard_off: n_shards + 1 device ints).  An index outside
 * [0, n_groups_global) goes to shard 0 as -1 (dropped there with GPX_S_NOGROUP).  cols / out_cols are
 * HOST arrays of device pointers.  No group state is touched; runs on the engine's stream.
 */
int gpx_route_batch_dev(gpx_engine* h, int32_t n, int32_t n_cols, const int32_t* const* cols,
                        const int32_t* g2l, int32_t n_groups_global, int32_t n_shards,
                        int32_t* const* out_cols, int32_t* shard_off);

/* ---- telemetry --------------------------------------------------------------- */

/* cumulative counters since engine creation: votes, decisions, dropped records */
int gpx_engine_counters(gpx_engine* h, uint64_t out[3]);

/*
 * Which way the accept-reply calls of the tiled front end delivered their compacted outputs since engine creation
 * (DESIGN.md 3.2): out[0] = calls whose per-bucket kernel wrote every output in its predicted place (a steady-state
 * batch: k_emit_dec16 only published the count), out[1] = calls compacted from the staging.  Telemetry of the device
 * path only - the results are the same either way; tests/test_inplace_gpu.py asserts which one a shape takes.
 */
int gpx_engine_path_counters(gpx_engine* h, uint64_t out[2]);

/*
 * Per-kernel timing with hipEvents on the launch stream.  enable=1 brackets every
 * kernel launch with events (adds host overhead: never enable inside a timed
 * throughput region).  gpx_profile_read copies up to cap entries
 * (name, launches, total_ms) and returns the number of distinct kernels.
 */
typedef struct gpx_kernel_stat {
  char name[48];
  uint64_t launches;
  double total_ms;
} gpx_kernel_stat;
int gpx_profile_enable(gpx_engine* h, int32_t enable);
int gpx_profile_read(gpx_engine* h, gpx_kernel_stat* out, int32_t cap);

int gpx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */
