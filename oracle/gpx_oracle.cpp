/*
 * gpx_oracle.cpp — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A deliberately plain, one-object-per-group, std::map-based restatement of the
 * gigapaxos accept / accept-reply / commit hot path, used ONLY as the checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in the
 * product path (gigapaxos_amd/, libgpx_hip.so) may link, import or call it.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/edu/umass/cs/gigapaxos/).  Java semantics reproduced on
 * purpose: 32-bit two's-complement wraparound in `a - b < 0` comparisons, plain
 * `<` where the reference uses it (PaxosCoordinatorState.java:815), TreeMap
 * signed-ascending iteration order, integer-division majority.
 *
 * PARITY STATUS: "parity unpinned by reference fixtures" for the decided
 * (group, slot, ballot, medianCheckpointedSlot) stream — the reference ships no
 * golden vectors for this path and no JVM exists in the build container, so the
 * Java cannot be run to produce any.  What IS pinned: the
 * known-answer assertions the reference's own self-tests make
 * (WaitforUtility.main, PaxosCoordinatorState.main's accept-reply section,
 * PaxosAcceptor.testAcceptor's monotone-ballot property, HotRestoreInfoTest),
 * re-expressed in tests/test_oracle_kat.py against this file.  Beyond those: every row of
 * SURVEY.md 8 is read from the Java a second time, in Python and not from this file
 * (tests/acc_enum_common.py, pcs_enum_common.py, round_model.py, wire_model.py, docs/HISTORY.md 7b);
 * this file and the engine are both held to those readings, and scripts/oracle_mutants.py shows
 * that they notice a wrong oracle (66 single-fault copies of this file: 62 noticed, 4 equivalent).
 * Two readings of the same text can share a misreading: the status above stands until a JVM
 * produces reference outputs (scripts/make_ref_fixtures.sh).
 *
 * Modelling assumptions (same as the engine, stated in docs/HISTORY.md):
 *   - steady state: no wall-clock event fires (no checkRunForCoordinator election,
 *     no accept retransmit; PaxosInstanceStateMachine.java:480-492 preamble is a
 *     no-op), i.e. BOOTSTRAP_COORD_DETERMINISTIC-like behaviour;
 *   - coordinators are always active (ballot-0 / hot-restored coordinators are
 *     created active: PaxosCoordinator.java:91-104, 122-131);
 *   - DIGEST_REQUESTS=false, BATCHED_COMMITS=true, GC_MAJORITY_EXECUTED=true,
 *     FORWARD_PREEMPTED_REQUESTS=false, EXECUTE_UPON_ACCEPT=false (defaults,
 *     PaxosConfig.java:435-453, 466, 788, 882, 927).
 */
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <initializer_list>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "../include/gpx.h"
#include "../include/gpx_wire.h"

namespace {

/* Java `a - b` on ints: wraps. */
static inline int32_t jsub(int32_t a, int32_t b) {
  return (int32_t)((uint32_t)a - (uint32_t)b);
}

/* paxosutil/Ballot.java:60-73 */
struct Ballot {
  int32_t num = 0, coord = 0;
  int compareTo(const Ballot& b) const {
    if (num != b.num) return jsub(num, b.num); /* "will handle wraparounds correctly" */
    return jsub(coord, b.coord);
  }
  bool equals(const Ballot& b) const { return compareTo(b) == 0; } /* :76-78 */
};

/* paxosutil/WaitforUtility.java:36-115 */
struct WaitforUtility {
  const std::vector<int32_t>* members;
  std::vector<bool> responded;
  int heardCount = 0;
  explicit WaitforUtility(const std::vector<int32_t>* m) : members(m), responded(m->size(), false) {}
  int getIndex(int32_t node) const { /* :108-115 — LAST matching index */
    int index = -1;
    for (size_t i = 0; i < members->size(); i++)
      if ((*members)[i] == node) index = (int)i;
    return index;
  }
  bool updateHeardFrom(int32_t node) { /* :51-62 */
    bool changed = false;
    int index = getIndex(node);
    if (index >= 0 && index < (int)members->size()) {
      if (!responded[index]) {
        changed = true;
        heardCount++;
      }
      responded[index] = true;
    }
    return changed;
  }
  bool heardFromMajority() const { return heardCount > (int)members->size() / 2; } /* :64-68 */
  bool contains(int32_t node) const { return getIndex(node) >= 0; }
};

/* An accepted pvalue as the acceptor remembers it (paxospackets/AcceptPacket.java:36-66,
 * PValuePacket.java:38-74).  The value itself stays with the host. */
struct Accepted {
  Ballot ballot;
  int32_t slot;
  bool stop;
};

/* A committed decision (PValuePacket after makeDecision, PValuePacket.java:135-143). */
struct Decision {
  Ballot ballot;
  int32_t slot;
  int32_t median;
  bool hasValue; /* RequestPacket.hasRequestValue(): false for BATCHED_COMMIT placeholders */
  bool stop;
};

/* PaxosAcceptor.java:94-110 */
struct PaxosAcceptor {
  int32_t _slot = 0;
  int32_t ballotNum = -1, ballotCoord = -1;
  int32_t acceptedGCSlot = -1;
  bool stopped = false;
  std::map<int32_t, Accepted> acceptedProposals; /* NullIfEmptyMap == TreeMap */
  std::map<int32_t, Decision> committedRequests;
  bool fromDisk = true; /* GET_ACCEPTED_PVALUES_FROM_DISK, PaxosAcceptor.java:75-76 */

  Ballot getBallot() const { return Ballot{ballotNum, ballotCoord}; }

  /* PaxosAcceptor.java:476-494 */
  void garbageCollectAccepted(int32_t gcSlot) {
    if (jsub(_slot, gcSlot) <= 0) gcSlot = jsub(_slot, 1);
    if (jsub(gcSlot, acceptedGCSlot) > 0) {
      acceptedGCSlot = gcSlot;
      for (auto it = acceptedProposals.begin(); it != acceptedProposals.end();) {
        if (jsub(it->first, gcSlot) <= 0)
          it = acceptedProposals.erase(it);
        else
          ++it;
      }
    }
    garbageCollectDecisions(gcSlot);
  }
  /* PaxosAcceptor.java:496-506 */
  void garbageCollectDecisions(int32_t slot) {
    if (jsub(slot, _slot) >= 0) return;
    for (auto it = committedRequests.begin(); it != committedRequests.end();) {
      if (jsub(slot, it->first) > 0)
        it = committedRequests.erase(it);
      else
        ++it;
    }
  }
  /* PaxosAcceptor.java:302-322.  Returns false iff stopped (Java returns null). */
  bool acceptAndUpdateBallot(const Accepted& accept, int32_t medianCP, Ballot* out, bool* stored) {
    if (stopped) return false;
    *stored = false;
    if (accept.ballot.compareTo(getBallot()) >= 0) {
      ballotNum = accept.ballot.num;
      ballotCoord = accept.ballot.coord;
      if (jsub(accept.slot, acceptedGCSlot) > 0) {
        acceptedProposals[accept.slot] = accept;
        *stored = true;
      }
    }
    garbageCollectAccepted(medianCP);
    *out = getBallot();
    return true;
  }
  /* PaxosAcceptor.java:369-385 */
  bool reconstructDecision(int32_t slot, Decision* out) const {
    auto c = committedRequests.find(slot);
    if (c != committedRequests.end()) {
      if (c->second.hasValue) {
        *out = c->second;
        return true;
      }
      auto a = acceptedProposals.find(slot);
      if (a != acceptedProposals.end() && a->second.ballot.equals(c->second.ballot)) {
        *out = Decision{a->second.ballot, slot, c->second.median, true, a->second.stop};
        return true;
      }
    }
    return false;
  }
  /* PaxosAcceptor.java:462-474 */
  void executed(int32_t s, bool stop) {
    /* s == _slot by construction; the Java asserts/suicides otherwise */
    _slot = (int32_t)((uint32_t)_slot + 1u);
    if (stop) stopped = true;
    if (stopped) committedRequests.clear();
    (void)s;
  }
  /* PaxosAcceptor.java:325-366.  `decision` may be null (poke). Returns true and
   * fills *next when an in-order executable decision was extracted. */
  bool putAndRemoveNextExecutable(const Decision* decision, Decision* next) {
    if (stopped) return false;
    Decision tmp;
    if (decision == nullptr) {
      auto c = committedRequests.find(_slot);
      if (c == committedRequests.end()) return false;
      tmp = c->second;
      decision = &tmp;
    }
    garbageCollectAccepted(decision->median);
    if (jsub(decision->slot, _slot) >= 0) {
      auto c = committedRequests.find(decision->slot);
      if (c == committedRequests.end() || !c->second.hasValue)
        committedRequests[decision->slot] = *decision;
    }
    bool haveNext = false;
    if (committedRequests.count(_slot)) {
      haveNext = reconstructDecision(_slot, next);
      if (haveNext && next->hasValue) {
        committedRequests.erase(_slot);
        executed(next->slot, next->stop);
      }
    }
    if (haveNext && fromDisk) acceptedProposals.erase(next->slot);
    return haveNext;
  }
};

/* PaxosCoordinatorState.java:69-181 (only what the accept phase touches) */
struct ProposalState {
  bool stop;
  WaitforUtility waitfor;
  int64_t handle = 0; /* the caller's key of the request value (RequestPacket.equals <=> same handle) */
  int kind = 0;       /* 0 proposed here, else GPX_E_CARRY / GPX_E_NOOP / GPX_E_NEWSTOP (view change) */
};
/* a pvalue reported in a PREPARE_REPLY (PCS.carryoverProposals, PaxosCoordinatorState.java:88-94) */
struct Carryover {
  Ballot ballot;
  int64_t handle;
  bool stop, noop;
};
struct PaxosCoordinatorState {
  Ballot myBallot;
  int32_t nextProposalSlotNumber = 0;
  bool active = false;
  std::vector<int32_t> nodeSlotNumbers;
  std::map<int32_t, ProposalState> myProposals;
  std::unique_ptr<WaitforUtility> waitforMyBallot;   /* != null while running for coordinator */
  std::map<int32_t, Carryover> carryoverProposals;

  /* PaxosCoordinatorState.java:867-875 (Arrays.sort = signed ascending) */
  int32_t getMedianMinus() const {
    std::vector<int32_t> copy(nodeSlotNumbers);
    std::sort(copy.begin(), copy.end());
    size_t medianMinus = copy.size() % 2 == 0 ? copy.size() / 2 - 1 : copy.size() / 2;
    return copy[medianMinus];
  }
  int32_t getMajorityCommittedSlot() const { return getMedianMinus(); } /* :859-861 */

  /* PaxosCoordinatorState.java:809-825 — note the PLAIN `<` */
  void recordSlotNumber(const std::vector<int32_t>& members, int32_t acceptor, int32_t maxCP) {
    for (size_t i = 0; i < members.size(); i++)
      if (members[i] == acceptor)
        if (nodeSlotNumbers[i] < maxCP) nodeSlotNumbers[i] = maxCP;
  }
  /* PaxosCoordinatorState.java:233-263 + initCommander :841-851.
   * returns 0 = ACCEPT issued, 1 = refused (after stop), 2 = inserted but not active (pre-active) */
  int propose(const std::vector<int32_t>& members, bool stop, int32_t* slot, int32_t* median,
              int64_t handle = 0, int kind = 0) {
    auto prev = myProposals.find(jsub(nextProposalSlotNumber, 1));
    if (prev != myProposals.end() && prev->second.stop) return 1;
    int32_t s = nextProposalSlotNumber;
    nextProposalSlotNumber = (int32_t)((uint32_t)nextProposalSlotNumber + 1u);
    ProposalState ps{stop, WaitforUtility(&members)};
    ps.handle = handle;
    ps.kind = kind;
    myProposals.emplace(s, ps);
    *slot = s;
    *median = getMajorityCommittedSlot();
    return active ? 0 : 2;
  }

  /* ---- phase 1b (PaxosCoordinatorState.java:265-587) ---- */
  bool isPreemptable(const Ballot& b) const { return b.compareTo(myBallot) > 0; } /* :271-279 */
  bool canIgnorePrepareReply(const Ballot& b, int32_t acceptor,
                             const std::vector<int32_t>& members) const { /* :285-316 */
    if (!waitforMyBallot) return true;
    if (b.compareTo(myBallot) < 0) return true;
    bool member = false;
    for (int32_t m : members) member = member || m == acceptor;
    int idx = waitforMyBallot->getIndex(acceptor);
    if (!member || (idx >= 0 && waitforMyBallot->responded[idx])) return true;
    return false;
  }
  /* recordSlotNumber(members, PrepareReplyPacket) :786-803 - wraparound-aware */
  void recordMinSlot(const std::vector<int32_t>& members, int32_t acceptor, int32_t minSlot) {
    for (size_t i = 0; i < members.size(); i++)
      if (members[i] == acceptor && jsub(nodeSlotNumbers[i], minSlot) < 0) nodeSlotNumbers[i] = minSlot;
  }
  /* isPrepareAcceptedByMajority :329-381 after canIgnorePrepareReply said no */
  bool recordPrepareReply(const std::vector<int32_t>& members, int32_t acceptor, int32_t minSlot,
                          int32_t npv, const int32_t* pslot, const int32_t* pbnum,
                          const int32_t* pbcoord, const int64_t* phandle, const uint8_t* pflags) {
    recordMinSlot(members, acceptor, minSlot);
    for (int32_t j = 0; j < npv; j++) {
      const Ballot b{pbnum[j], pbcoord[j]};
      auto ex = carryoverProposals.find(pslot[j]);
      if (ex == carryoverProposals.end() || b.compareTo(ex->second.ballot) > 0)
        carryoverProposals[pslot[j]] = Carryover{b, phandle ? phandle[j] : 0,
                                                 pflags && (pflags[j] & GPX_PV_STOP),
                                                 pflags && (pflags[j] & GPX_PV_NOOP)};
    }
    waitforMyBallot->updateHeardFrom(acceptor);
    return waitforMyBallot->heardFromMajority();
  }
  /* combinePValuesOntoProposals :393-444 + reproposePreemptedProposals :460-468 + processStop
   * :478-517 (assertions off: its conversions are unreachable, every ProposalStateAtCoordinator
   * carries the coordinator's own ballot, :153-157) */
  void combinePValuesOntoProposals(const std::vector<int32_t>& members) {
    if (carryoverProposals.empty()) return;
    int32_t maxCarry = carryoverProposals.begin()->first; /* getMaxPValueSlot :899-910 */
    for (auto& kv : carryoverProposals)
      if (jsub(kv.first, maxCarry) > 0) maxCarry = kv.first;
    int32_t maxMin = nodeSlotNumbers[0]; /* getMaxMinCarryoverSlot :917-927 */
    for (int32_t v : nodeSlotNumbers)
      if (jsub(v, maxMin) > 0) maxMin = v;
    std::map<int32_t, ProposalState> preActives;
    preActives.swap(myProposals);
    for (int32_t cur = maxMin; jsub(cur, maxCarry) <= 0; cur = (int32_t)((uint32_t)cur + 1u)) {
      auto co = carryoverProposals.find(cur);
      auto pa = preActives.find(cur);
      if (co != carryoverProposals.end()) {
        ProposalState ps{co->second.stop, WaitforUtility(&members)};
        ps.handle = co->second.handle;
        ps.kind = GPX_E_CARRY;
        myProposals.emplace(cur, ps);
      } else if (pa == preActives.end()) {
        ProposalState ps{false, WaitforUtility(&members)};
        ps.kind = GPX_E_NOOP;
        myProposals.emplace(cur, ps);
      } else {
        bool dup = false; /* isDuplicate :446-452: RequestPacket.equals over the carry-overs */
        for (auto& kv : carryoverProposals) dup = dup || kv.second.handle == pa->second.handle;
        if (!dup) myProposals.emplace(cur, pa->second);
        preActives.erase(pa);
      }
    }
    nextProposalSlotNumber = (int32_t)((uint32_t)maxCarry + 1u);
    for (auto& kv : preActives) { /* reproposePreemptedProposals: TreeMap order */
      int32_t sl, md;
      propose(members, kv.second.stop, &sl, &md, kv.second.handle, kv.second.kind);
    }
    bool stopExists = false; /* processStop */
    for (auto& kv : myProposals) stopExists = stopExists || kv.second.stop;
    if (stopExists) {
      auto last = myProposals.find(jsub(nextProposalSlotNumber, 1));
      if (last != myProposals.end() && !last->second.stop) {
        int32_t sl, md;
        propose(members, true, &sl, &md, 0, GPX_E_NEWSTOP); /* new RequestPacket(0, STOP, true) */
      }
    }
  }
  void setCoordinatorActive() { /* :568-578 */
    active = true;
    waitforMyBallot.reset();
    carryoverProposals.clear();
  }
  /* PaxosCoordinatorState.java:597-640; returns true on first majority */
  bool handleAcceptReplyMyBallot(const std::vector<int32_t>& members, int32_t slot,
                                 int32_t acceptor, int32_t maxCP, int32_t* median) {
    recordSlotNumber(members, acceptor, maxCP);
    auto p = myProposals.find(slot);
    if (p == myProposals.end()) return false;
    p->second.waitfor.updateHeardFrom(acceptor);
    if (p->second.waitfor.heardFromMajority()) {
      *median = getMajorityCommittedSlot();
      myProposals.erase(p);
      return true;
    }
    return false;
  }
  /* PaxosCoordinatorState.java:661-675; true if a proposal was preempted */
  bool handleAcceptReplyHigherBallot(int32_t slot) { return myProposals.erase(slot) > 0; }
  bool preemptedFully() const { return myProposals.empty(); } /* :677-683 */
};

/* PaxosInstanceStateMachine.java:193-238 (the fields the accept phase touches) */
struct Group {
  int32_t version = 0;
  std::vector<int32_t> members;
  PaxosAcceptor paxosState;
  std::unique_ptr<PaxosCoordinatorState> coordinator;
};

struct ExecRun {
  int32_t first, count;
};

struct Engine {
  gpx_config cfg;
  std::vector<std::unique_ptr<Group>> groups;
  uint64_t counters[3] = {0, 0, 0};
  int32_t ordered_mask = 0; /* gpx_engine_set_ordered_batches: engine contract, not in the reference */
  /* PaxosManager.pinstances (paxosID -> instance), PaxosManager.java:1816-1832; wire oracle only */
  std::map<std::string, int32_t> name2row;
  std::vector<std::string> row2name;
  std::vector<int32_t> free_rows;
  bool free_init = false;

  Group* get(int32_t g) {
    if (g < 0 || g >= cfg.max_groups) return nullptr;
    return groups[g].get();
  }

  /* PaxosInstanceStateMachine.java:1619-1701 (protocol part: no app upcall here,
   * the executed slots are reported to the caller who performs Replicable.execute) */
  ExecRun extractExecuteAndCheckpoint(Group& g, const Decision* logged) {
    ExecRun run{g.paxosState._slot, 0};
    if (g.paxosState.stopped) return run;
    Decision inorder;
    while (g.paxosState.putAndRemoveNextExecutable(logged, &inorder)) {
      run.count++;
      if (inorder.stop) break; /* :1678-1685 (copyEpochFinalCheckpointState assumed ok) */
    }
    return run;
  }
  /* PaxosInstanceStateMachine.java:1432-1478 */
  ExecRun handleCommittedRequest(Group& g, const Decision& committed) {
    return extractExecuteAndCheckpoint(g, &committed);
  }
};

static thread_local char g_err[8] = "";

static void fill_hri(const Group& g, gpx_hri* r) {
  /* PaxosInstanceStateMachine.java:2011-2020 */
  std::memset(r, 0, sizeof(*r));
  r->version = g.version;
  r->acc_slot = g.paxosState._slot;
  r->acc_bnum = g.paxosState.ballotNum;
  r->acc_bcoord = g.paxosState.ballotCoord;
  r->acc_gc_slot = g.paxosState.acceptedGCSlot;
  if (g.coordinator && g.coordinator->active) {
    r->has_coord = 1;
    r->coord_bnum = g.coordinator->myBallot.num;
    r->coord_bcoord = g.coordinator->myBallot.coord;
    r->next_proposal_slot = g.coordinator->nextProposalSlotNumber;
    for (size_t i = 0; i < g.members.size(); i++) r->node_slots[i] = g.coordinator->nodeSlotNumbers[i];
  } else {
    r->has_coord = 0;
    r->next_proposal_slot = -1; /* getNextProposalSlotIfActive, PaxosCoordinator.java:375-377 */
  }
}

}  // namespace

extern "C" {

int orc_abi_version(void) { return GPX_ABI_VERSION; }
const char* orc_last_error(void) { return g_err; }

int orc_engine_create(const gpx_config* cfg, gpx_engine** out) {
  if (!cfg || !out || cfg->max_groups <= 0 || cfg->kmax < 1 || cfg->kmax > GPX_KMAX_LIMIT)
    return GPX_EINVAL;
  /* window: the engine's ring size W (a power of two) - the three refusals it implies (GPX_S_WINDOW of a
   * proposal, an ACCEPT, a commit) are restated below so that engine == oracle holds at the limits too;
   * window = 0 (oracle only) = no limits: the reference's unbounded maps, for the known-answer tests */
  if (cfg->window < 0 || (cfg->window & (cfg->window - 1))) return GPX_EINVAL;
  Engine* e = new Engine();
  e->cfg = *cfg;
  e->groups.resize((size_t)cfg->max_groups);
  *out = reinterpret_cast<gpx_engine*>(e);
  return GPX_OK;
}
int orc_engine_destroy(gpx_engine* h) {
  delete reinterpret_cast<Engine*>(h);
  return GPX_OK;
}
int orc_engine_sync(gpx_engine*) { return GPX_OK; }
int orc_engine_set_ordered_batches(gpx_engine* h, int32_t mask) {
  if (!h || (mask & ~(GPX_ORDERED_PROPOSE | GPX_ORDERED_ACCEPT | GPX_ORDERED_COMMIT | GPX_ORDERED_REPLY_RUNS |
                      GPX_TRY_REPLY_RUNS)))
    return GPX_EINVAL;
  reinterpret_cast<Engine*>(h)->ordered_mask = mask;
  return GPX_OK;
}
/* the promise of gpx_engine_set_ordered_batches: gidx in range and ascending (strictly: every group
 * at most once).  Returns the first index that breaks it - out of range, or a descent (strict: a non-ascent)
 * INTO it - or n: the records before that index are applied, the records from it on are refused
 * (GPX_S_UNORDERED), include/gpx.h.  An engine contract, not in the reference. */
static int32_t batch_first_unordered(const Engine* e, int32_t n, const int32_t* gidx, bool strict) {
  for (int32_t i = 0; i < n; i++) {
    if (gidx[i] < 0 || gidx[i] >= e->cfg.max_groups) return i;
    if (i > 0 && (strict ? gidx[i - 1] >= gidx[i] : gidx[i - 1] > gidx[i])) return i;
  }
  return n;
}
/* the promise GPX_ORDERED_REPLY_RUNS: gidx in range and at most GPX_REPLY_RUNS_MAX non-decreasing runs (the
 * concatenated replies of the acceptors).  GPX_TRY_REPLY_RUNS is only a hint to the engine: nothing to do. */
static bool batch_is_few_runs(const Engine* e, int32_t n, const int32_t* gidx) {
  int32_t descents = 0;
  for (int32_t i = 0; i < n; i++) {
    if (gidx[i] < 0 || gidx[i] >= e->cfg.max_groups) return false;
    if (i + 1 < n && gidx[i] > gidx[i + 1] && ++descents > GPX_REPLY_RUNS_MAX - 1) return false;
  }
  return true;
}
int orc_host_register(gpx_engine* h, void* p, size_t n) { return h && p && n ? GPX_OK : GPX_EINVAL; }
int orc_host_unregister(gpx_engine* h, void* p) { return h && p ? GPX_OK : GPX_EINVAL; }
/* (gpx_host_alloc / gpx_host_free: plain memory here - the oracle has no DMA engine to please) */
int orc_host_alloc(gpx_engine* h, size_t n, void** out) {
  if (!h || !n || !out) return GPX_EINVAL;
  *out = malloc(n);
  return *out ? GPX_OK : GPX_ENOMEM;
}
int orc_host_free(gpx_engine* h, void* p) {
  if (!h || !p) return GPX_EINVAL;
  free(p);
  return GPX_OK;
}
int orc_engine_counters(gpx_engine* h, uint64_t out[3]) {
  if (!h) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  for (int i = 0; i < 3; i++) out[i] = e->counters[i];
  return GPX_OK;
}

/* PaxosInstanceStateMachine.java:677-690 hotRestore; PaxosAcceptor.java:121-134;
 * PaxosCoordinator.java:122-131 */
int orc_group_create(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* members,
                     const uint8_t* k, const gpx_hri* rows, uint8_t* status) {
  if (!h || n < 0) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  for (int32_t i = 0; i < n; i++) {
    int32_t g = gidx[i];
    if (g < 0 || g >= e->cfg.max_groups || k[i] < 1 || k[i] > e->cfg.kmax) {
      if (status) status[i] = GPX_S_NOGROUP;
      continue;
    }
    if (e->groups[g]) {
      if (status) status[i] = GPX_S_EXISTS;
      continue;
    }
    auto grp = std::make_unique<Group>();
    grp->version = rows[i].version;
    grp->members.assign(members + (size_t)i * e->cfg.kmax, members + (size_t)i * e->cfg.kmax + k[i]);
    grp->paxosState.ballotNum = rows[i].acc_bnum;
    grp->paxosState.ballotCoord = rows[i].acc_bcoord;
    grp->paxosState._slot = rows[i].acc_slot;
    grp->paxosState.acceptedGCSlot = rows[i].acc_gc_slot;
    grp->paxosState.fromDisk = (e->cfg.flags & GPX_F_ACCEPTS_FROM_DISK) != 0;
    if (rows[i].has_coord && rows[i].coord_bcoord == e->cfg.my_id) {
      auto c = std::make_unique<PaxosCoordinatorState>();
      c->myBallot = Ballot{rows[i].coord_bnum, rows[i].coord_bcoord};
      c->nextProposalSlotNumber = rows[i].next_proposal_slot;
      c->active = true;
      c->nodeSlotNumbers.assign(rows[i].node_slots, rows[i].node_slots + k[i]);
      grp->coordinator = std::move(c);
    }
    e->groups[g] = std::move(grp);
    if (status) status[i] = GPX_S_OK;
  }
  return GPX_OK;
}

/* PaxosInstanceStateMachine.java:2004-2035 tryPause / PaxosManager.java:2162 kill */
int orc_group_retire(gpx_engine* h, int32_t n, const int32_t* gidx, int32_t mode, gpx_hri* rows,
                     uint8_t* status) {
  if (!h || n < 0) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  for (int32_t i = 0; i < n; i++) {
    Group* g = e->get(gidx[i]);
    if (rows) std::memset(&rows[i], 0, sizeof(gpx_hri));
    if (!g) {
      if (status) status[i] = GPX_S_NOGROUP;
      continue;
    }
    if (mode == GPX_RETIRE_PAUSE) {
      /* PaxosAcceptor.caughtUp :451-459, PaxosCoordinatorState.caughtUp :758-761 */
      bool caughtUp = g->paxosState.committedRequests.empty() &&
                      (g->paxosState.acceptedProposals.empty() || g->paxosState.fromDisk) &&
                      (!g->coordinator || g->coordinator->myProposals.empty());
      if (!caughtUp) {
        if (status) status[i] = GPX_S_BUSY;
        continue;
      }
    }
    if (rows) fill_hri(*g, &rows[i]);
    e->groups[gidx[i]].reset();
    if (status) status[i] = GPX_S_OK;
  }
  return GPX_OK;
}

int orc_group_snapshot(gpx_engine* h, int32_t n, const int32_t* gidx, gpx_hri* rows,
                       uint8_t* status) {
  if (!h || n < 0 || !rows) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  for (int32_t i = 0; i < n; i++) {
    Group* g = e->get(gidx[i]);
    std::memset(&rows[i], 0, sizeof(gpx_hri));
    if (!g) {
      if (status) status[i] = GPX_S_NOGROUP;
      continue;
    }
    fill_hri(*g, &rows[i]);
    if (status) status[i] = GPX_S_OK;
  }
  return GPX_OK;
}

/* canonical state dump: see docs/HISTORY.md §state-dump */
int orc_group_dump(gpx_engine* h, int32_t gidx, int32_t* buf, int32_t cap) {
  if (!h || !buf) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  std::vector<int32_t> w;
  Group* g = e->get(gidx);
  w.push_back(g ? 1 : 0);
  if (g) {
    w.push_back(g->version);
    w.push_back((int32_t)g->members.size());
    for (int32_t m : g->members) w.push_back(m);
    const PaxosAcceptor& a = g->paxosState;
    w.push_back(a._slot);
    w.push_back(a.ballotNum);
    w.push_back(a.ballotCoord);
    w.push_back(a.acceptedGCSlot);
    w.push_back(a.stopped ? 1 : 0);
    w.push_back((int32_t)a.acceptedProposals.size());
    for (auto& kv : a.acceptedProposals) {
      w.push_back(kv.first);
      w.push_back(kv.second.ballot.num);
      w.push_back(kv.second.ballot.coord);
      w.push_back(kv.second.stop ? 1 : 0);
    }
    w.push_back((int32_t)a.committedRequests.size());
    for (auto& kv : a.committedRequests) {
      w.push_back(kv.first);
      w.push_back(kv.second.ballot.num);
      w.push_back(kv.second.ballot.coord);
      w.push_back(kv.second.median);
      w.push_back(kv.second.hasValue ? 1 : 0);
      w.push_back(kv.second.stop ? 1 : 0);
    }
    w.push_back(g->coordinator ? 1 : 0);
    if (g->coordinator) {
      const PaxosCoordinatorState& c = *g->coordinator;
      w.push_back(c.myBallot.num);
      w.push_back(c.myBallot.coord);
      w.push_back(c.nextProposalSlotNumber);
      for (int32_t s : c.nodeSlotNumbers) w.push_back(s);
      w.push_back((int32_t)c.myProposals.size());
      for (auto& kv : c.myProposals) {
        w.push_back(kv.first);
        w.push_back(kv.second.stop ? 1 : 0);
        int32_t mask = 0;
        for (size_t i = 0; i < kv.second.waitfor.responded.size(); i++)
          if (kv.second.waitfor.responded[i]) mask |= (1 << i);
        w.push_back(mask);
      }
      /* view change: active flag; while running for coordinator the heard-from mask, the
       * pre-active proposals' handles and the carried-over pvalues */
      w.push_back(c.active ? 1 : 0);
      if (!c.active) {
        int32_t mask = 0;
        if (c.waitforMyBallot)
          for (size_t i = 0; i < c.waitforMyBallot->responded.size(); i++)
            if (c.waitforMyBallot->responded[i]) mask |= (1 << i);
        w.push_back(c.waitforMyBallot ? 1 : 0);
        w.push_back(mask);
        for (auto& kv : c.myProposals) {
          w.push_back((int32_t)(uint32_t)kv.second.handle);
          w.push_back((int32_t)((uint64_t)kv.second.handle >> 32));
        }
        w.push_back((int32_t)c.carryoverProposals.size());
        for (auto& kv : c.carryoverProposals) {
          w.push_back(kv.first);
          w.push_back(kv.second.ballot.num);
          w.push_back(kv.second.ballot.coord);
          w.push_back((kv.second.stop ? GPX_PV_STOP : 0) | (kv.second.noop ? GPX_PV_NOOP : 0));
          w.push_back((int32_t)(uint32_t)kv.second.handle);
          w.push_back((int32_t)((uint64_t)kv.second.handle >> 32));
        }
      }
    }
  }
  if ((int32_t)w.size() > cap) return GPX_ECAPACITY;
  std::memcpy(buf, w.data(), w.size() * sizeof(int32_t));
  return (int32_t)w.size();
}

/* PaxosInstanceStateMachine.java:767-888 handleRequest -> handleProposal */
int orc_propose_batch_h(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                        const int64_t* handle, int32_t* slot, int32_t* bnum, int32_t* bcoord,
                        int32_t* median_cp, uint8_t* status);
int orc_propose_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                      int32_t* slot, int32_t* bnum, int32_t* bcoord, int32_t* median_cp,
                      uint8_t* status) {
  return orc_propose_batch_h(h, n, gidx, is_stop, nullptr, slot, bnum, bcoord, median_cp, status);
}
int orc_propose_batch_h(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                        const int64_t* handle, int32_t* slot, int32_t* bnum, int32_t* bcoord,
                        int32_t* median_cp, uint8_t* status) {
  if (!h || n < 0) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  const int32_t n_all = n;
  if (e->ordered_mask & GPX_ORDERED_PROPOSE) n = batch_first_unordered(e, n, gidx, true);
  for (int32_t i = n; i < n_all; i++) {
    slot[i] = bnum[i] = bcoord[i] = median_cp[i] = 0;
    status[i] = GPX_S_UNORDERED;
  }
  for (int32_t i = 0; i < n; i++) {
    slot[i] = bnum[i] = bcoord[i] = median_cp[i] = 0;
    Group* g = e->get(gidx[i]);
    if (!g) {
      status[i] = GPX_S_NOGROUP;
      e->counters[2]++;
      continue;
    }
    if (g->paxosState.stopped) { /* :456-460 */
      status[i] = GPX_S_STOPPED;
      e->counters[2]++;
      continue;
    }
    /* PaxosCoordinator.exists(c, paxosState.getBallot()) :213-219, PISM:825-826 */
    if (g->coordinator && g->coordinator->myBallot.compareTo(g->paxosState.getBallot()) >= 0) {
      int32_t s = 0, m = 0;
      {
        /* engine limit, not in the reference (whose map is unbounded): myProposals is a ring of
         * `window` slots - a proposal whose slot - window is still outstanding is refused */
        const PaxosCoordinatorState& c = *g->coordinator;
        auto prev = c.myProposals.find(jsub(c.nextProposalSlotNumber, 1));
        const bool after_stop = prev != c.myProposals.end() && prev->second.stop;
        if (!after_stop && e->cfg.window > 0 &&
            c.myProposals.count(jsub(c.nextProposalSlotNumber, e->cfg.window))) {
          status[i] = GPX_S_WINDOW;
          e->counters[2]++;
          continue;
        }
      }
      int rc = g->coordinator->propose(g->members, is_stop && (is_stop[i] & 1), &s, &m,
                                       handle ? handle[i] : 0);
      if (rc == 1) {
        status[i] = GPX_S_REFUSED;
        continue;
      }
      slot[i] = s;
      bnum[i] = g->coordinator->myBallot.num;
      bcoord[i] = g->coordinator->myBallot.coord;
      median_cp[i] = rc == 0 ? m : 0;
      status[i] = rc == 0 ? GPX_S_OK : GPX_S_PREACTIVE; /* not active: no ACCEPT yet (:254-261) */
    } else {
      /* :854-860 unicast to paxosState.getBallotCoord() */
      bnum[i] = g->paxosState.ballotNum;
      bcoord[i] = g->paxosState.ballotCoord;
      status[i] = GPX_S_FORWARD;
    }
  }
  return GPX_OK;
}

/* Output order contract of include/gpx.h: compacted outputs leave grouped by gidx ascending, the
 * entries of one group in the array order of the records that produced them.  The per-record
 * loops below collect outputs in arrival order (what sequential handlePaxosMessage calls give);
 * this is the regrouping by group that the reference's own next stage performs when it files
 * outgoing decisions under their paxosID (PaxosPacketBatcher.java:121-156) — a STABLE sort by
 * gidx, so nothing inside a group moves. */
static void regroup_by_gidx(int32_t m, int32_t* gidx, std::initializer_list<int32_t*> cols,
                            uint8_t* kind) {
  if (m <= 0) return; /* memcpy from an empty vector's null data() is undefined */
  std::vector<int32_t> order(m);
  for (int32_t i = 0; i < m; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return gidx[a] < gidx[b]; });
  std::vector<int32_t> tmp(m);
  auto permute = [&](int32_t* col) {
    for (int32_t i = 0; i < m; i++) tmp[i] = col[order[i]];
    std::memcpy(col, tmp.data(), (size_t)m * sizeof(int32_t));
  };
  for (int32_t* c : cols) permute(c);
  if (kind) {
    std::vector<uint8_t> tk(m);
    for (int32_t i = 0; i < m; i++) tk[i] = kind[order[i]];
    std::memcpy(kind, tk.data(), (size_t)m);
  }
  permute(gidx);
}

/* PaxosInstanceStateMachine.java:1080-1166 handleAccept */
int orc_accept_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                     const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                     const uint8_t* a_flags, int32_t* r_bnum, int32_t* r_bcoord,
                     int32_t* r_maxcp, uint8_t* r_flags, uint8_t* status, int32_t* x_gidx,
                     int32_t* x_first, int32_t* x_count, int32_t* n_runs) {
  if (!h || n < 0) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  int32_t runs = 0;
  const int32_t n_all = n;
  if (e->ordered_mask & GPX_ORDERED_ACCEPT) n = batch_first_unordered(e, n, gidx, false);
  for (int32_t i = n; i < n_all; i++) {
    r_bnum[i] = r_bcoord[i] = r_maxcp[i] = 0;
    r_flags[i] = 0;
    status[i] = GPX_S_UNORDERED;
  }
  for (int32_t i = 0; i < n; i++) {
    r_bnum[i] = r_bcoord[i] = r_maxcp[i] = 0;
    r_flags[i] = 0;
    Group* g = e->get(gidx[i]);
    if (!g) {
      status[i] = GPX_S_NOGROUP;
      e->counters[2]++;
      continue;
    }
    if (g->paxosState.stopped) {
      status[i] = GPX_S_STOPPED;
      e->counters[2]++;
      continue;
    }
    Accepted accept{Ballot{bnum[i], bcoord[i]}, slot[i], a_flags && (a_flags[i] & GPX_A_STOP)};
    {
      /* engine limit, not in the reference (whose map is unbounded): acceptedProposals is a ring of `window`
       * entries indexed by slot & (window - 1) - an ACCEPT that would be stored where ANOTHER live accepted
       * slot sits is dropped whole (GPX_S_WINDOW: no ballot change, no reply), like a lost packet */
      const PaxosAcceptor& pa = g->paxosState;
      const bool will_store = accept.ballot.compareTo(pa.getBallot()) >= 0 && jsub(accept.slot, pa.acceptedGCSlot) > 0;
      bool clash = false;
      if (will_store && e->cfg.window > 0)
        for (const auto& kv : pa.acceptedProposals)
          clash |= kv.first != accept.slot && ((kv.first ^ accept.slot) & (e->cfg.window - 1)) == 0;
      if (clash) {
        status[i] = GPX_S_WINDOW;
        e->counters[2]++;
        continue;
      }
    }
    /* :1122 PValuePacket prev = paxosState.getAccept(accept.slot) — BEFORE accepting */
    bool havePrev = false;
    Ballot prevBallot;
    auto pit = g->paxosState.acceptedProposals.find(accept.slot);
    if (pit != g->paxosState.acceptedProposals.end()) {
      havePrev = true;
      prevBallot = pit->second.ballot;
    }
    Ballot ballot;
    bool stored = false;
    g->paxosState.acceptAndUpdateBallot(accept, median_cp[i], &ballot, &stored);
    /* :1139-1143 reply(myID, ballot, slot, getSlot()-1) — GC_MAJORITY_EXECUTED */
    r_bnum[i] = ballot.num;
    r_bcoord[i] = ballot.coord;
    r_maxcp[i] = jsub(g->paxosState._slot, 1);
    /* :1146-1149 */
    bool toLog = accept.ballot.compareTo(ballot) >= 0 &&
                 jsub(accept.slot, g->paxosState.acceptedGCSlot) > 0 &&
                 (!havePrev || prevBallot.compareTo(accept.ballot) < 0);
    r_flags[i] = (uint8_t)((toLog ? GPX_R_TOLOG : 0) | (stored ? GPX_R_STORED : 0));
    status[i] = GPX_S_OK;
    /* :1158-1161 might release some meta-commits */
    Decision recon;
    if (g->paxosState.reconstructDecision(accept.slot, &recon)) {
      ExecRun run = e->handleCommittedRequest(*g, recon);
      if (run.count > 0) {
        x_gidx[runs] = gidx[i];
        x_first[runs] = run.first;
        x_count[runs] = run.count;
        runs++;
      }
    }
  }
  regroup_by_gidx(runs, x_gidx, {x_first, x_count}, nullptr);
  *n_runs = runs;
  return GPX_OK;
}

/* PaxosInstanceStateMachine.java:1248-1419 handleAcceptReply (per vote) */
int orc_accept_reply_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                           const int32_t* bcoord, const int32_t* slot, const int32_t* acceptor,
                           const int32_t* max_cp, int32_t* d_gidx, int32_t* d_slot,
                           int32_t* d_bnum, int32_t* d_bcoord, int32_t* d_median_cp,
                           uint8_t* d_kind, int32_t* n_out, uint8_t* status) {
  if (!h || n < 0) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  int32_t out = 0;
  if ((e->ordered_mask & GPX_ORDERED_REPLY_RUNS) && !batch_is_few_runs(e, n, gidx)) {
    if (status)
      for (int32_t i = 0; i < n; i++) status[i] = GPX_S_UNORDERED;
    *n_out = 0;
    return GPX_OK;
  }
  for (int32_t i = 0; i < n; i++) {
    e->counters[0]++;
    Group* g = e->get(gidx[i]);
    if (!g) {
      if (status) status[i] = GPX_S_NOGROUP;
      e->counters[2]++;
      continue;
    }
    if (g->paxosState.stopped) {
      if (status) status[i] = GPX_S_STOPPED;
      e->counters[2]++;
      continue;
    }
    if (status) status[i] = GPX_S_OK;
    /* PaxosCoordinator.handleAcceptReply, PaxosCoordinator.java:210-250 */
    PaxosCoordinatorState* c = g->coordinator.get();
    if (c && c->active) {
      Ballot rb{bnum[i], bcoord[i]};
      int cmp = rb.compareTo(c->myBallot);
      if (cmp > 0) {
        if (c->handleAcceptReplyHigherBallot(slot[i])) {
          d_gidx[out] = gidx[i];
          d_slot[out] = slot[i];
          d_bnum[out] = c->myBallot.num; /* preempted pvalue keeps MY ballot */
          d_bcoord[out] = c->myBallot.coord;
          d_median_cp[out] = -1; /* PValuePacket.medianCheckpointedSlot default, :80 */
          d_kind[out] = GPX_D_PREEMPTED;
          out++;
          e->counters[1]++;
        }
      } else if (cmp == 0) {
        int32_t median = 0;
        if (c->handleAcceptReplyMyBallot(g->members, slot[i], acceptor[i], max_cp[i], &median)) {
          d_gidx[out] = gidx[i];
          d_slot[out] = slot[i];
          d_bnum[out] = c->myBallot.num;
          d_bcoord[out] = c->myBallot.coord;
          d_median_cp[out] = median;
          d_kind[out] = GPX_D_DECISION;
          out++;
          e->counters[1]++;
        }
      }
    }
    /* PISM:1273, 1361-1364 + PaxosCoordinator.isPreemptedFully (PaxosCoordinator.java:110-115):
     * for ANY coordinator, active or still being elected */
    if (c && Ballot{bnum[i], bcoord[i]}.compareTo(c->myBallot) > 0 && c->preemptedFully())
      g->coordinator.reset();
  }
  regroup_by_gidx(out, d_gidx, {d_slot, d_bnum, d_bcoord, d_median_cp}, d_kind);
  *n_out = out;
  return GPX_OK;
}

/* PaxosInstanceStateMachine.java:1480-1528 handleBatchedCommit (per slot) and
 * :1432-1478 handleCommittedRequest (GPX_C_HASVALUE records) */
int orc_commit_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                     const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                     const uint8_t* c_kind, uint8_t* status, int32_t* x_gidx, int32_t* x_first,
                     int32_t* x_count, int32_t* n_runs) {
  if (!h || n < 0) return GPX_EINVAL;
  Engine* e = reinterpret_cast<Engine*>(h);
  int32_t runs = 0;
  const int32_t n_all = n;
  if (e->ordered_mask & GPX_ORDERED_COMMIT) n = batch_first_unordered(e, n, gidx, false);
  for (int32_t i = n; i < n_all; i++) status[i] = GPX_S_UNORDERED;
  for (int32_t i = 0; i < n; i++) {
    Group* g = e->get(gidx[i]);
    if (!g) {
      status[i] = GPX_S_NOGROUP;
      e->counters[2]++;
      continue;
    }
    if (g->paxosState.stopped) {
      status[i] = GPX_S_STOPPED;
      e->counters[2]++;
      continue;
    }
    /* engine limit (not in the reference): committedRequests is a ring of `window` slots from the next slot
     * to execute - a commit further ahead is dropped (GPX_S_WINDOW), like a lost packet */
    if (e->cfg.window > 0 && jsub(slot[i], g->paxosState._slot) >= e->cfg.window) {
      status[i] = GPX_S_WINDOW;
      e->counters[2]++;
      continue;
    }
    status[i] = GPX_S_OK;
    Ballot b{bnum[i], bcoord[i]};
    uint8_t kind = c_kind ? c_kind[i] : 0;
    Decision d;
    if (kind & GPX_C_HASVALUE) {
      d = Decision{b, slot[i], median_cp[i], true, (kind & GPX_C_STOP) != 0};
    } else {
      /* :1488-1524 */
      auto a = g->paxosState.acceptedProposals.find(slot[i]);
      if (a != g->paxosState.acceptedProposals.end() && a->second.ballot.equals(b))
        d = Decision{a->second.ballot, slot[i], median_cp[i], true, a->second.stop};
      else
        d = Decision{b, slot[i], median_cp[i], false, false}; /* placeholder */
    }
    ExecRun run = e->handleCommittedRequest(*g, d);
    if (run.count > 0) {
      x_gidx[runs] = gidx[i];
      x_first[runs] = run.first;
      x_count[runs] = run.count;
      runs++;
    }
  }
  regroup_by_gidx(runs, x_gidx, {x_first, x_count}, nullptr);
  *n_runs = runs;
  return GPX_OK;
}

/* ---- small pure functions exported for the known-answer tests ---------------- */

/* Ballot.compareTo sign: -1/0/1 */
int orc_ballot_compare(int32_t n1, int32_t c1, int32_t n2, int32_t c2) {
  int r = Ballot{n1, c1}.compareTo(Ballot{n2, c2});
  return r < 0 ? -1 : (r > 0 ? 1 : 0);
}

/* WaitforUtility driven as in WaitforUtility.main: feed nodes in order; out[i] =
 * (changed << 1) | majority after node i. */
int orc_waitfor_trace(const int32_t* members, int32_t k, const int32_t* nodes, int32_t n,
                      uint8_t* out) {
  std::vector<int32_t> m(members, members + k);
  WaitforUtility w(&m);
  for (int32_t i = 0; i < n; i++) {
    bool ch = w.updateHeardFrom(nodes[i]);
    out[i] = (uint8_t)((ch ? 2 : 0) | (w.heardFromMajority() ? 1 : 0));
  }
  return w.heardCount;
}

/* PaxosCoordinatorState.getMedianMinus */
int32_t orc_median_minus(const int32_t* a, int32_t k) {
  PaxosCoordinatorState c;
  c.nodeSlotNumbers.assign(a, a + k);
  return c.getMedianMinus();
}

/* PISM.roundRobinCoordinator (PaxosInstanceStateMachine.java:2251-2256) with Java
 * String.hashCode (s[0]*31^(n-1)+...) over ISO-8859-1/ASCII bytes and Java
 * Math.abs / % semantics.  Returns the member, or INT32_MIN if Java would throw
 * (negative index when ballotnum + hash == Integer.MIN_VALUE). */
int32_t orc_round_robin_coordinator(const char* paxos_id, const int32_t* members, int32_t k,
                                    int32_t ballotnum) {
  uint32_t hsh = 0;
  for (const unsigned char* p = (const unsigned char*)paxos_id; *p; ++p) hsh = 31u * hsh + *p;
  int32_t x = (int32_t)((uint32_t)ballotnum + hsh);
  int32_t ax = x < 0 ? (int32_t)(0u - (uint32_t)x) : x; /* Math.abs(MIN_VALUE) == MIN_VALUE */
  int32_t idx = ax % k; /* sign follows dividend, like Java */
  if (idx < 0) return INT32_MIN;
  return members[idx];
}

} /* extern "C" */

/* wire codec oracle (same translation unit: it reads the Engine's groups) */
#include "gpx_wire_oracle.inc"
