// Host side above the C-ABI, in C++: the mirror of the reference's PaxosManager surface for the
// accept / decide path (the reference is Java and no JDK exists in the build image or on the GPU box,
// so this is where the host code lives; INTEGRATION.md shows the JNI binding a Java host would use
// instead).  Names and meaning follow the reference:
//
//   gpx::Replicable      edu.umass.cs.gigapaxos.interfaces.Replicable (execute / checkpoint / restore)
//   gpx::Messenger       what PaxosManager.send hands MessagingTasks to (PaxosManager.java:2098-2128)
//   gpx::PaxosManager    PaxosManager: createPaxosInstance (PM:611-810), propose (PM:1206-1260),
//                        handleIncomingPacket -> handlePaxosPacket (PM:1126-1204), kill (PM:2162)
//
// One PaxosManager = one node id = one engine handle.  Everything the per-group Java objects did
// (PaxosInstanceStateMachine, PaxosAcceptor, PaxosCoordinatorState, PaxosPacketBatcher's coalescing,
// the byte decoding of the four byteified packet types) happens behind include/gpx.h and
// include/gpx_wire.h; this layer only moves frames, keeps the request VALUES (the engine never sees
// them) and performs the application upcalls in slot order.
#pragma once

#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "gpx.h"
#include "gpx_wire.h"

namespace gpx {

using Frame = std::vector<uint8_t>;

/* RequestPacket as the application sees it (paxospackets/RequestPacket.java:60-160) */
struct Request {
  std::string paxosID;
  int64_t requestID = 0;
  std::string requestValue;
  bool stop = false;
  int32_t entryReplica = -1;
  int32_t slot = 0; /* the slot it was decided in */
  bool isNoop() const { return requestValue == "NO_OP"; } /* interfaces/Request.NO_OP */
};

/* interfaces/Replicable.java: the replicated application */
class Replicable {
 public:
  virtual ~Replicable() = default;
  /* execute(Request, doNotReplyToClient): must be deterministic; false = retry (PISM:1755-1842) */
  virtual bool execute(const Request& request, bool doNotReplyToClient) = 0;
  virtual std::string checkpoint(const std::string& name) = 0;
  virtual bool restore(const std::string& name, const std::string& state) = 0;
};

/* the transport under PaxosManager.send: unicast of one byteified packet to a node id */
class Messenger {
 public:
  virtual ~Messenger() = default;
  virtual void send(int32_t nodeID, Frame&& frame) = 0;
};

/* AbstractPaxosLogger's batched logging as the accept path sees it (AbstractPaxosLogger.java:656-716):
 * the ACCEPTs of a batch that must be durable before their replies leave (GPX_R_TOLOG, PISM:1146-1149)
 * are handed over in one piece; the replies of that batch are released when the batch is durable.
 * The engine never waits: the next batches are processed while the log write is in flight. */
class Logger {
 public:
  virtual ~Logger() = default;
  virtual uint64_t logBatch(const std::vector<const Frame*>& records) = 0; /* returns the batch's ticket (> 0) */
  virtual uint64_t durable() = 0;                                          /* highest ticket that is durable */
};

/* a log whose batches become durable a fixed number of polls later (deterministic: tests) */
class DelayLogger : public Logger {
 public:
  explicit DelayLogger(int polls) : delay_(polls) {}
  uint64_t logBatch(const std::vector<const Frame*>& records) override;
  uint64_t durable() override;
  uint64_t records = 0, bytes = 0;

 private:
  int delay_;
  uint64_t next_ = 1, polls_ = 0;
  std::deque<std::pair<uint64_t, uint64_t>> pending_; /* (ticket, poll it was logged at) */
  uint64_t durable_ = 0;
};

/* an append-only file written and fdatasync'ed by its own thread */
class FileLogger : public Logger {
 public:
  /* throws std::runtime_error when the file cannot be opened */
  explicit FileLogger(const std::string& path);
  ~FileLogger() override;
  uint64_t logBatch(const std::vector<const Frame*>& records) override;
  uint64_t durable() override;
  /* a write or fdatasync failed: durable() will not advance any more (held replies stay held) */
  bool failed() const;

 private:
  struct Impl;
  Impl* impl_;
};

/* PaxosManager.propose's callback (ExecutedCallback): runs at the entry replica once the request has
 * been executed there */
using ExecutedCallback = std::function<void(const Request&)>;

struct Options {
  int32_t maxGroups = 1 << 16; /* PC.PINSTANCES_CAPACITY */
  int32_t kmax = 3;            /* largest replica group */
  int32_t window = 8;
  int32_t maxBatch = 1 << 18;
  int32_t device = -1;
  /* RequestBatcher (RequestBatcher.java:40-239): requests of one group that are queued together
   * become ONE proposal, up to these limits (PC.MAX_BATCH_SIZE; min(NIO payload, log message size)) */
  /* PISM.syncLongDecisionGaps (PISM:1550-1570): a group whose newest commit is this many slots ahead
   * of its next undecided slot asks the commit's coordinator for the missing decisions */
  int32_t checkpointInterval = 400; /* PC.CHECKPOINT_INTERVAL: app.checkpoint(name) every so many slots */
  int32_t syncGapThreshold = 2;
  int32_t decisionLogSlots = 64; /* executed decisions kept per group to answer such requests */
  Logger* logger = nullptr; /* nullptr = logging off (DISABLE_LOGGING): replies leave at once */
  bool batchRequests = true;
  int32_t maxBatchSize = 2000;
  int32_t maxBatchBytes = 1 << 20;
};

struct Stats {
  uint64_t proposed = 0, forwarded = 0, accepts = 0, votes = 0, decisions = 0, commits = 0, executed = 0;
  uint64_t dropped_frames = 0, refused = 0, engine_calls = 0;
  uint64_t pauses = 0, unpauses = 0, checkpoints = 0, callbacks = 0;
  uint64_t accepts_resent = 0, prepares_resent = 0;
  uint64_t logged_accepts = 0, log_batches = 0, held_replies = 0;
  uint64_t sync_requests = 0, sync_decisions_sent = 0, sync_decisions_applied = 0;
  uint64_t batched_requests = 0; /* requests that rode in another request's proposal */
  uint64_t elections_started = 0, elections_won = 0, elections_lost = 0, prepares = 0, carried_over = 0, deferred = 0,
           noops = 0, preactive = 0;
};

class PaxosManager {
 public:
  PaxosManager(int32_t myID, Replicable* app, Messenger* messenger, const Options& opt);
  ~PaxosManager();
  PaxosManager(const PaxosManager&) = delete;
  PaxosManager& operator=(const PaxosManager&) = delete;

  /* PaxosManager.createPaxosInstance(paxosID, version, gms, app, initialState) (PM:611-810):
   * members sorted ascending; the ballot-0 coordinator is roundRobinCoordinator(paxosID, members, 0)
   * (PISM:2251-2256), created active; false if the name exists or the table is full */
  bool createPaxosInstance(const std::string& paxosID, const std::vector<int32_t>& members,
                           const std::string& initialState = std::string());
  /* many at once (PM:664-691) */
  int createPaxosInstances(const std::vector<std::string>& paxosIDs, const std::vector<int32_t>& members);
  /* PaxosManager.propose(paxosID, requestValue, callback) (PM:1206-1260): this node is the entry
   * replica; returns the request id, 0 if there is no such instance here */
  int64_t propose(const std::string& paxosID, const std::string& requestValue, bool stop = false,
                  ExecutedCallback callback = nullptr);
  /* a byteified packet from the network (PaxosManager.handleIncomingPacket -> handlePaxosPacket) */
  void handleIncomingPacket(const uint8_t* frame, size_t len);
  void handleIncomingPacket(Frame&& frame);
  /* PaxosManager.kill(paxosID) (PM:2162-2192) */
  bool kill(const std::string& paxosID);
  /* PaxosManager.pause (Deactivator -> PISM.tryPause, PM:2284-2412, PISM:2004-2035): the group's
   * HotRestoreInfo leaves the device table and its row is free for another group; only a caught-up
   * group with no accepted value outstanding is paused.  A packet or a request for a paused name
   * brings it back (PaxosManager.getInstance -> unpause, PM:1816-1832), pausing idle groups first if
   * the table is full. */
  bool pause(const std::string& paxosID);
  /* the retransmission timer firing (ACCEPT_TIMEOUT / PREPARE_TIMEOUT, PCS:715-739): re-multicasts
   * the ACCEPT of every group's head-of-line proposal that is still waiting for a majority
   * (pokeLocalCoordinator, PISM:2268-2279) and the PREPAREs of elections still running; returns the
   * number of packets re-sent */
  size_t poke();
  size_t pausedCount() const { return paused_.size(); }
  /* the failure detector's verdict (FailureDetection -> PaxosManager.isNodeUp == false): runs
   * checkRunForCoordinator over every instance (PISM:2090-2176) and multicasts the PREPAREs of the
   * groups this node must run for; returns their number */
  size_t nodeDown(int32_t nodeID);

  /* one pass of the node's pipeline over everything queued so far (the reference spreads this over
   * its demultiplexer pool, RequestBatcher, PaxosPacketBatcher and the per-instance monitors):
   * decode -> propose -> accept -> accept-reply -> commit -> in-order execution -> outgoing frames.
   * Returns the number of frames and requests it consumed (0 = idle). */
  size_t process();

  int32_t myID() const { return myID_; }
  const Stats& stats() const { return stats_; }
  const char* lastError() const { return err_.c_str(); }

 private:
  struct Instance {
    int32_t gidx;
    int32_t version;
    std::vector<int32_t> members;
  };
  struct StoredAccept { /* acceptedProposals' value side: the ACCEPT frame itself */
    int32_t bnum, bcoord;
    Frame frame;
  };
  static uint64_t key(int32_t gidx, int32_t slot) { return ((uint64_t)(uint32_t)gidx << 32) | (uint32_t)slot; }

  /* an ACCEPT this node issues (after propose or after winning an election): multicast at once, my
   * own copy short-circuited into the accept phase of the same pass */
  struct OutAccept {
    int32_t gidx, bnum, bcoord, slot, median;
    uint8_t flags;
    int64_t requestID;
    Frame frame;
  };
  bool check(int rc, const char* what);
  void executeRuns(int32_t nRuns, const int32_t* xg, const int32_t* xf, const int32_t* xc);
  void sendToMembers(const Instance& in, const Frame& frame, bool includeSelf);
  void issueAccept(std::vector<OutAccept>& out, int32_t gidx, const Frame& requestFrame, int64_t requestID,
                   bool stop, int32_t slot, int32_t bnum, int32_t bcoord, int32_t median);
  struct Paused { /* what Deactivator keeps of a paused instance: its HotRestoreInfo */
    gpx_hri hri;
    std::vector<int32_t> members;
    int32_t version;
  };
  bool unpause(const std::string& paxosID);
  bool makeRoom(int32_t rows);
  size_t processRun();
  bool handleSyncRequests(std::vector<Frame>& reqs);
  bool handleDecisions(std::vector<Frame>& decisions);
  bool syncGaps(const std::vector<int32_t>& gidx, const std::vector<int32_t>& bcoord);
  bool handlePrepares(std::vector<Frame>& prepares);
  bool handlePrepareReplies(std::vector<Frame>& replies, std::vector<OutAccept>& out);

  int32_t myID_;
  Replicable* app_;
  Messenger* messenger_;
  Options opt_;
  gpx_engine* engine_ = nullptr;
  std::unordered_map<std::string, Instance> pinstances_; /* PaxosManager.pinstances */
  std::vector<std::string> rowName_;                      /* gidx -> paxosID */
  std::unordered_map<uint64_t, StoredAccept> accepted_;   /* (gidx, slot) -> the stored ACCEPT */
  std::deque<Frame> inbox_;                               /* frames from the network and from myself */
  std::deque<Frame> requests_;                            /* REQUEST frames of local clients */
  std::deque<Frame> deferred_; /* requests the engine's proposal window had no room for: retried first */
  size_t redeferred_ = 0;      /* ... and how many of this pass's RETRIES went back there */
  /* view change: request bytes by (gidx, requestID) - proposals made while not active, and the
   * pvalues the PREPARE replies carried */
  std::map<std::pair<int32_t, int64_t>, Frame> preactive_; /* (gidx, request handle) -> request frame */
  /* carryoverProposals' value side (PCS:271-391): per (gidx, SLOT) the stored ACCEPT frame of the
   * highest ballot seen in the prepare replies of the running election - keyed by slot, not by
   * request id: NO_OP and STOP pvalues share id 0 and one request can sit in two slots with
   * different contents (alone / as a batch head) */
  struct Carried {
    int32_t bnum, bcoord;
    Frame frame;
  };
  std::map<std::pair<int32_t, int32_t>, Carried> carried_;
  std::vector<int32_t> downNodes_;
  std::unordered_map<int64_t, ExecutedCallback> callbacks_; /* my clients' requests, by request id */
  struct Held { /* a reply waiting for its batch's log write */
    uint64_t ticket;
    int32_t dest;
    Frame frame;
  };
  std::deque<Held> held_;
  size_t releaseHeld();
  /* what the logger's getLoggedDecisions would return: the last decisions executed here */
  /* keyed by NAME: a group's log stays with it while it is paused (it can still answer sync requests)
   * and never leaks to the next group that gets its row */
  std::unordered_map<std::string, std::map<int32_t, StoredAccept>> decided_;
  void forgetRow(int32_t gidx); /* host-side bookkeeping keyed by a row that is being vacated */
  std::unordered_map<uint64_t, uint64_t> syncAsked_; /* (gidx, slot) -> the pass a sync was last requested in */
  std::unordered_map<std::string, Paused> paused_;
  std::vector<uint64_t> lastActive_; /* per row: the pass that last touched the group */
  std::vector<int32_t> liveAccepts_; /* per row: accepted values the host still holds */
  std::deque<Frame> retry_;          /* frames that found their group paused */
  uint64_t pass_ = 0;
  int32_t liveRows_ = 0;
  int64_t nextRequestID_;
  Stats stats_;
  std::string err_;
};

/* ---- byte-level helpers shared with the example cluster (what a reference node puts on the wire) */

/* RequestPacket.toBytes (RequestPacket.java:819-948) of a plain client request */
Frame makeRequestFrame(const std::string& paxosID, int32_t version, int64_t requestID,
                       const std::string& value, bool stop, int32_t entryReplica);
/* AcceptPacket.toBytes (AcceptPacket.java:95-135): the request bytes re-typed ACCEPT + the 22-byte tail */
Frame makeAcceptFrame(const Frame& requestFrame, int32_t slot, int32_t bnum, int32_t bcoord,
                      int32_t medianCheckpointedSlot, int32_t sender);
/* PREPARE / PREPARE_REPLY: the reference ships these two as JSON only (no toBytes); the byte layout
 * here is this host layer's own - PaxosPacket header, then {bnum, bcoord, firstUndecidedSlot} and
 * {acceptor, bnum, bcoord, firstSlot, n, n x {slot, bnum, bcoord, len, the ACCEPT frame}} */
constexpr int32_t kTypePrepare = 2, kTypePrepareReply = 7; /* PaxosPacketType ints (PaxosPacket.java:202-230) */
/* SYNC_DECISIONS_REQUEST and DECISION, JSON-only as well: {sender, n, n x slot} and
 * {slot, bnum, bcoord, medianCheckpointedSlot, len, the ACCEPT frame} */
constexpr int32_t kTypeSyncDecisions = 32, kTypeDecision = 6;
/* the request inside a REQUEST / ACCEPT frame; false if the bytes do not parse */
bool parseRequest(const Frame& frame, Request* out);
/* ... and with the requests batched into it (RequestPacket.getRequestPackets, :1239-1245): itself
 * first, then batched[] in order */
bool parseRequests(const Frame& frame, std::vector<Request>* out);
/* first.latchToBatch(rest) on the bytes (RequestPacket.java:1090-1100, toBytes :930-948) */
Frame latchToBatch(const Frame& first, const std::vector<const Frame*>& rest);
int32_t batchSizeOf(const Frame& requestFrame); /* RequestPacket.batchSize() */
/* java.lang.String.hashCode of an ISO-8859-1 string; PISM.roundRobinCoordinator (PISM:2251-2256) */
int32_t javaStringHash(const std::string& s);
int32_t roundRobinCoordinator(const std::string& paxosID, const std::vector<int32_t>& members, int32_t ballotnum);

}  // namespace gpx
