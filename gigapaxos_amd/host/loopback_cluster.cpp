// In-process replica cluster over gpx::PaxosManager: N nodes, one engine each, frames carried by an
// in-memory messenger (what tests/loopback_1_group and TESTPaxosMain do over 127.0.0.1 in the
// reference).  Clients send requests to random entry replicas; a request that lands on a
// non-coordinator is forwarded (PISM:854-860).  The app mirrors TESTPaxosApp's invariant
// (testing/TESTPaxosApp.java:190): the request's slot is the group's sequence number, and every
// replica ends with the same (count, hash chain) per group.
//
//   gpx_loopback_cluster [--nodes 3] [--groups 1000] [--rounds 20] [--seed 1] [--value-bytes 64]
//                        [--stop-last] [--entry any|coordinator] [--kill-round r [--kill-node i]]
//                        [--burst b] [--no-batching] [--capacity c] [--drop-commits permille]
// --drop-commits p: the network loses p of 1000 BATCHED_COMMIT frames; a replica that sees newer
// commits while an older slot is undecided (gpx_gap_scan) asks the coordinator for the decisions it
// missed (two extra loss-free rounds at the end let the tail catch up).
// --drop-accepts p: p of 1000 ACCEPT and BATCHED_ACCEPT_REPLY frames are lost; whenever the cluster
// goes quiet with proposals outstanding the retransmission timers fire (PaxosManager::poke).
// --log-delay n / --log-file prefix: logging on - the ACCEPTs that must be durable before their replies
// leave go to a logger (durable n polls later / an fdatasync'ed file per node written by its own
// thread); the replies of a batch wait for its log write, later batches are processed meanwhile.
// --capacity c: the engines' group tables hold only c < groups rows: idle groups are paused (their
// HotRestoreInfo kept by the manager) and come back when a packet or request names them.  The
// table must hold the groups that are busy at the same time: use --active a (a < c) so that a
// round touches only a (rotating) window of a groups.
// --burst b: b requests per group per round, queued together: RequestBatcher latches the requests
// of one group that meet at a replica into one proposal (turned off by --no-batching).
// --kill-round r: in round r node i (default 0) dies after ONE pipeline pass - ACCEPTs are in flight,
// no reply has been processed; the survivors' failure detectors fire (PaxosManager::nodeDown), the
// next member in line runs for coordinator in every group the dead node coordinated, takes a few
// client requests while not yet elected, carries the accepted values over and the cluster goes on.
// Prints one JSON line; exit code 1 if the replicas disagree or a request was lost.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <climits>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "gpx_host.hpp"

namespace {

struct GroupState {
  int64_t seqnum = 0;  /* requests executed */
  int32_t lastSlot = 0; /* slots start at 1 (initial-state checkpoint) */
  uint64_t hash = 1469598103934665603ull;
  bool stopped = false;
};

class HashChainApp : public gpx::Replicable {
 public:
  std::map<std::string, GroupState> state;
  uint64_t outOfOrder = 0;
  bool execute(const gpx::Request& r, bool) override {
    GroupState& g = state[r.paxosID];
    g.seqnum++;
    /* TESTPaxosApp's `assert state.seqnum == requestPacket.slot`, for batches too: requests arrive in
     * slot order, every slot is seen, the requests of one batch share theirs */
    if (r.slot != g.lastSlot && r.slot != g.lastSlot + 1) outOfOrder++;
    g.lastSlot = r.slot;
    uint64_t h = g.hash;
    auto mix = [&](const void* p, size_t n) {
      const unsigned char* c = (const unsigned char*)p;
      for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 1099511628211ull;
    };
    mix(&r.requestID, sizeof(r.requestID));
    mix(r.requestValue.data(), r.requestValue.size());
    g.hash = h;
    if (r.stop) g.stopped = true;
    return true;
  }
  std::string checkpoint(const std::string& name) override { return std::to_string(state[name].seqnum); }
  bool restore(const std::string&, const std::string&) override { return true; }
};

class LoopbackMessenger : public gpx::Messenger {
 public:
  std::map<int32_t, gpx::PaxosManager*> nodes;
  uint64_t frames = 0, bytes = 0, lost = 0, rng = 88172645463325252ull;
  int dropCommitsPermille = 0, dropAcceptsPermille = 0;
  void send(int32_t nodeID, gpx::Frame&& f) override {
    auto it = nodes.find(nodeID);
    if (it == nodes.end()) return; /* a dead node: the frame is lost */
    const int type = f.size() >= 8 && f[4] == 0 && f[5] == 0 && f[6] == 0 ? f[7] : -1;
    if (dropAcceptsPermille > 0 && (type == GPX_WT_ACCEPT || type == GPX_WT_BATCHED_ACCEPT_REPLY)) {
      rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17;
      if ((int)(rng % 1000) < dropAcceptsPermille) {
        lost++;
        return;
      }
    }
    if (dropCommitsPermille > 0 && type == GPX_WT_BATCHED_COMMIT) {
      rng ^= rng << 13, rng ^= rng >> 7, rng ^= rng << 17;
      if ((int)(rng % 1000) < dropCommitsPermille) {
        lost++;
        return;
      }
    }
    frames++;
    bytes += f.size();
    it->second->handleIncomingPacket(std::move(f));
  }
};

uint64_t xorshift(uint64_t& s) {
  s ^= s >> 12;
  s ^= s << 25;
  s ^= s >> 27;
  return s * 2685821657736338717ull;
}

}  // namespace

int main(int argc, char** argv) {
  int nNodes = 3, G = 1000, R = 20, valueBytes = 64;
  uint64_t seed = 1;
  bool stopLast = false, entryAny = true;
  int killRound = -1, killNode = 0, burst = 1, capacity = 0, active = 0, dropCommits = 0, dropAccepts = 0;
  int logDelay = -1;
  std::string logFile;
  bool batching = true, dumpFrames = false;
  for (int i = 1; i < argc; i++) {
    auto is = [&](const char* f) { return std::strcmp(argv[i], f) == 0; };
    if (is("--nodes") && i + 1 < argc) nNodes = std::atoi(argv[++i]);
    else if (is("--groups") && i + 1 < argc) G = std::atoi(argv[++i]);
    else if (is("--rounds") && i + 1 < argc) R = std::atoi(argv[++i]);
    else if (is("--seed") && i + 1 < argc) seed = std::strtoull(argv[++i], nullptr, 10);
    else if (is("--value-bytes") && i + 1 < argc) valueBytes = std::atoi(argv[++i]);
    else if (is("--stop-last")) stopLast = true;
    else if (is("--kill-round") && i + 1 < argc) killRound = std::atoi(argv[++i]);
    else if (is("--kill-node") && i + 1 < argc) killNode = std::atoi(argv[++i]);
    else if (is("--burst") && i + 1 < argc) burst = std::atoi(argv[++i]);
    else if (is("--no-batching")) batching = false;
    else if (is("--dump-frames")) dumpFrames = true;
    else if (is("--capacity") && i + 1 < argc) capacity = std::atoi(argv[++i]);
    else if (is("--active") && i + 1 < argc) active = std::atoi(argv[++i]);
    else if (is("--drop-commits") && i + 1 < argc) dropCommits = std::atoi(argv[++i]);
    else if (is("--drop-accepts") && i + 1 < argc) dropAccepts = std::atoi(argv[++i]);
    else if (is("--log-delay") && i + 1 < argc) logDelay = std::atoi(argv[++i]);
    else if (is("--log-file") && i + 1 < argc) logFile = argv[++i];
    else if (is("--entry") && i + 1 < argc) entryAny = std::strcmp(argv[++i], "any") == 0;
    else {
      std::fprintf(stderr, "unknown argument %s\n", argv[i]);
      return 2;
    }
  }
  if (dumpFrames) { /* the byte builders, for comparison with an independent restatement of toBytes() */
    auto hex = [](const gpx::Frame& f) {
      std::string s;
      char b[3];
      for (uint8_t c : f) std::snprintf(b, sizeof(b), "%02x", c), s += b;
      return s;
    };
    const gpx::Frame rq = gpx::makeRequestFrame("TESTPaxosApp7", 0, 0x1122334455667788ll, "hello-value", false, 101);
    const gpx::Frame st = gpx::makeRequestFrame("g", 3, -5, "", true, 100);
    const gpx::Frame ac = gpx::makeAcceptFrame(rq, 42, 2, 101, 40, 101);
    const gpx::Frame bt = gpx::latchToBatch(rq, {&st, &rq});
    std::vector<gpx::Request> rs;
    gpx::parseRequests(bt, &rs);
    /* RequestPacket.main (RequestPacket.java:1531-1563): "asd999" latched with 25 more stop requests,
     * to bytes and back */
    std::vector<gpx::Frame> subs;
    std::vector<const gpx::Frame*> subp;
    for (int i = 0; i < 25; i++) subs.push_back(gpx::makeRequestFrame("pid", 0, 1000 + i, "asd" + std::to_string(i), true, 100));
    for (auto& f : subs) subp.push_back(&f);
    const gpx::Frame big = gpx::latchToBatch(gpx::makeRequestFrame("pid", 0, 999, "asd999", true, 100), subp);
    std::vector<gpx::Request> back;
    bool same = gpx::parseRequests(big, &back) && back.size() == 26 && back[0].requestValue == "asd999";
    for (size_t i = 1; same && i < back.size(); i++)
      same = back[i].requestValue == "asd" + std::to_string(i - 1) && back[i].stop && back[i].requestID == 999 + (int64_t)i;
    std::printf("{\"rp_main\": \"%s\", \"rp_main_roundtrip\": %s, ", hex(big).c_str(), same ? "true" : "false");
    std::printf("\"request\": \"%s\", \"stop\": \"%s\", \"accept\": \"%s\", \"batched\": \"%s\", \"batch_size\": %d, "
                "\"parsed\": %zu, \"coordinator\": %d, \"hash\": %d}\n",
                hex(rq).c_str(), hex(st).c_str(), hex(ac).c_str(), hex(bt).c_str(), gpx::batchSizeOf(bt), rs.size(),
                gpx::roundRobinCoordinator("TESTPaxosApp7", {100, 101, 102}, 0), gpx::javaStringHash("hello"));
    return 0;
  }
  std::vector<int32_t> ids;
  for (int i = 0; i < nNodes; i++) ids.push_back(100 + i);
  LoopbackMessenger net;
  std::vector<std::unique_ptr<HashChainApp>> apps;
  std::vector<std::unique_ptr<gpx::PaxosManager>> pms;
  gpx::Options opt;
  opt.maxGroups = capacity > 0 ? capacity : G + 16;
  opt.kmax = nNodes < 3 ? 3 : nNodes;
  opt.maxBatch = std::max(1 << 16, 8 * G * burst);
  opt.batchRequests = batching;
  opt.checkpointInterval = 4; /* small, so that short runs checkpoint too */
  std::vector<std::unique_ptr<gpx::Logger>> loggers;
  for (int i = 0; i < nNodes; i++) {
    apps.emplace_back(new HashChainApp());
    opt.logger = nullptr;
    if (!logFile.empty())
      loggers.emplace_back(new gpx::FileLogger(logFile + "." + std::to_string(ids[(size_t)i])));
    else if (logDelay >= 0)
      loggers.emplace_back(new gpx::DelayLogger(logDelay));
    if (!logFile.empty() || logDelay >= 0) opt.logger = loggers.back().get();
    pms.emplace_back(new gpx::PaxosManager(ids[(size_t)i], apps.back().get(), &net, opt));
    net.nodes[ids[(size_t)i]] = pms.back().get();
  }
  std::vector<std::string> names;
  for (int g = 0; g < G; g++) names.push_back("TESTPaxosApp" + std::to_string(g));
  for (auto& pm : pms) {
    int made = 0;
    const size_t chunk = capacity > 0 ? (size_t)std::max(1, capacity / 4) : names.size();
    for (size_t o = 0; o < names.size(); o += chunk) {
      std::vector<std::string> part(names.begin() + (long)o, names.begin() + (long)std::min(names.size(), o + chunk));
      made += pm->createPaxosInstances(part, ids);
      pm->process(); /* a pass goes by: the groups just made count as idle for the next chunk */
    }
    if (made != G) {
      std::fprintf(stderr, "createPaxosInstances failed on node %d: %s\n", pm->myID(), pm->lastError());
      return 2;
    }
  }
  std::vector<bool> alive((size_t)nNodes, true);
  auto drain = [&]() { /* until no node has anything left to do */
    for (int fired = 0;;) {
      size_t work = 0;
      for (int i = 0; i < nNodes; i++)
        if (alive[(size_t)i]) work += pms[(size_t)i]->process();
      if (work) continue;
      /* quiet: the retransmission timers fire (only a lossy network leaves anything to re-send) */
      if ((dropAccepts > 0 || dropCommits > 0) && fired++ < 400)
        for (int i = 0; i < nNodes; i++)
          if (alive[(size_t)i]) work += pms[(size_t)i]->poke();
      if (work == 0) break;
    }
  };
  uint64_t sentKillRound = 0, acked = 0; /* acked: requests whose entry replica executed them */
  uint64_t rng = seed * 0x9E3779B97F4A7C15ull + 1;
  uint64_t sent = 0;
  std::string value((size_t)valueBytes, 'x');
  net.dropCommitsPermille = dropCommits;
  net.dropAcceptsPermille = dropAccepts;
  for (int r = 0; r < R; r++) {
    if (dropCommits > 0 && r >= R - 2) net.dropCommitsPermille = 0; /* the tail catches up */
    const int nAct = active > 0 && active < G ? active : G;
    for (int gb = 0; gb < nAct * burst; gb++) {
      const int g = (int)(((long)r * nAct + gb % nAct) % G); /* a window of groups that moves every round */
      const uint64_t x = xorshift(rng);
      for (size_t b = 0; b < value.size() && b < 8; b++) value[b] = (char)('a' + ((x >> (8 * b)) & 15));
      size_t entry = entryAny ? (size_t)(x % (uint64_t)nNodes) : 0;
      while (!alive[entry]) entry = (entry + 1) % (size_t)nNodes;
      if (!entryAny) { /* the coordinator itself */
        const int32_t c = gpx::roundRobinCoordinator(names[(size_t)g], ids, 0);
        for (size_t i = 0; i < ids.size(); i++)
          if (ids[i] == c && alive[i]) entry = i;
      }
      const bool stop = stopLast && r == R - 1;
      if (pms[entry]->propose(names[(size_t)g], value, stop, [&acked](const gpx::Request&) { acked++; })) {
        sent++;
        if (r == killRound) sentKillRound++;
      }
    }
    if (r == killRound && killNode >= 0 && killNode < nNodes && alive[(size_t)killNode]) {
      for (int i = 0; i < nNodes; i++) pms[(size_t)i]->process(); /* ONE pass each: ACCEPTs in flight */
      alive[(size_t)killNode] = false;
      net.nodes.erase(ids[(size_t)killNode]); /* frames to it are lost from now on */
      for (int i = 0; i < nNodes; i++)
        if (alive[(size_t)i]) pms[(size_t)i]->nodeDown(ids[(size_t)killNode]);
      /* clients keep sending while the elections run: every third group, at the node next in line */
      const size_t cand = (size_t)((killNode + 1) % nNodes);
      for (int g = 0; g < G; g += 3) { /* as REQUEST packets off the network: queued behind the PREPAREs */
        pms[cand]->handleIncomingPacket(gpx::makeRequestFrame(names[(size_t)g], 0, ((int64_t)1 << 50) + g, value,
                                                              false, ids[cand]));
        sent++, sentKillRound++;
      }
    }
    drain();
  }
  /* a replica that lost everything about a group's last slot learns of it with the group's next traffic.
   * That can happen where only some groups are touched per round, and - with bursts longer than the
   * engine's window - where the engine itself drops an ACCEPT (ring index held by an older live accept) AND
   * the commit (further ahead than the committed window) of a lagging replica: one more request for every
   * group, window by window, over a network that is whole again */
  int catchUp = 0;
  if ((dropAccepts > 0 || dropCommits > 0) && ((active > 0 && active < G) || burst > 8)) {
    catchUp = 1;
    net.dropAcceptsPermille = net.dropCommitsPermille = 0;
    const int step = (active > 0 && active < G) ? active : G;
    for (int w = 0; w < G; w += step) {
      for (int g = w; g < std::min(G, w + step); g++) {
        size_t entry = 0;
        while (!alive[entry]) entry++;
        if (pms[entry]->propose(names[(size_t)g], value, false, [&acked](const gpx::Request&) { acked++; })) sent++;
      }
      drain();
    }
  }
  /* verdict: the survivors must agree; without a failure every request is executed everywhere, with
   * one only the requests of the failure round may be lost (they died with the node or were sent
   * to it) */
  bool ok = true;
  uint64_t digest0 = 0, executed0 = 0;
  bool first = true;
  for (int i = 0; i < nNodes; i++) {
    if (!alive[(size_t)i]) continue;
    uint64_t d = 0, ex = 0;
    for (auto& kv : apps[(size_t)i]->state) {
      d = (d ^ kv.second.hash) * 1099511628211ull + (uint64_t)kv.second.seqnum;
      ex += (uint64_t)kv.second.seqnum;
      if (killRound < 0 && active <= 0 && kv.second.seqnum != (int64_t)R * burst + catchUp) ok = false;
      if (stopLast && !kv.second.stopped) ok = false;
    }
    if (first) digest0 = d, executed0 = ex, first = false;
    if (std::getenv("GPX_CLUSTER_DEBUG"))
      std::fprintf(stderr, "node %d: groups %zu outOfOrder %llu digest %016llx executed %llu sent %llu (kill round %llu)\n",
                   ids[(size_t)i], apps[(size_t)i]->state.size(), (unsigned long long)apps[(size_t)i]->outOfOrder,
                   (unsigned long long)d, (unsigned long long)ex, (unsigned long long)sent, (unsigned long long)sentKillRound);
    if ((active <= 0 && (int)apps[(size_t)i]->state.size() != G) || apps[(size_t)i]->outOfOrder || d != digest0) ok = false;
    if (killRound < 0 ? ex != sent : ex + sentKillRound < sent) ok = false;
  }
  uint64_t digest[1] = {digest0};
  if (!ok && std::getenv("GPX_CLUSTER_DEBUG")) { /* which groups disagree, and where each replica stands */
    int shown = 0;
    for (auto& name : names) {
      int64_t lo = INT64_MAX, hi = -1;
      for (int i = 0; i < nNodes; i++)
        if (alive[(size_t)i]) {
          const int64_t q = apps[(size_t)i]->state[name].seqnum;
          lo = std::min(lo, q), hi = std::max(hi, q);
        }
      if (lo == hi || shown++ > 12) continue;
      std::fprintf(stderr, "%s:", name.c_str());
      for (int i = 0; i < nNodes; i++)
        if (alive[(size_t)i])
          std::fprintf(stderr, " node %d seq %lld slot %d |", ids[(size_t)i], (long long)apps[(size_t)i]->state[name].seqnum,
                       apps[(size_t)i]->state[name].lastSlot);
      std::fprintf(stderr, "\n");
    }
  }
  std::printf("{\"nodes\": %d, \"groups\": %d, \"rounds\": %d, \"requests\": %" PRIu64 ", \"executed_per_node\": %" PRIu64
              ", \"state_digest\": \"%016" PRIx64 "\", \"frames\": %" PRIu64 ", \"bytes\": %" PRIu64 ", \"frames_lost\": %" PRIu64 ", \"client_acks\": %" PRIu64 ", \"ok\": %s, \"per_node\": [",
              nNodes, G, R, sent, executed0, digest[0], net.frames, net.bytes, net.lost, acked, ok ? "true" : "false");
  for (int i = 0; i < nNodes; i++) {
    const gpx::Stats& s = pms[(size_t)i]->stats();
    std::printf("%s{\"id\": %d, \"alive\": %s, \"checkpoints\": %" PRIu64 ", \"pauses\": %" PRIu64 ", \"unpauses\": %" PRIu64 ", \"paused_now\": %zu, \"proposed\": %" PRIu64 ", \"batched_requests\": %" PRIu64 ", \"forwarded\": %" PRIu64 ", \"accepts\": %" PRIu64
                ", \"votes\": %" PRIu64 ", \"decisions\": %" PRIu64 ", \"commits\": %" PRIu64 ", \"executed\": %" PRIu64
                ", \"refused\": %" PRIu64 ", \"dropped_frames\": %" PRIu64 ", \"engine_calls\": %" PRIu64
                ", \"elections_started\": %" PRIu64 ", \"elections_won\": %" PRIu64 ", \"elections_lost\": %" PRIu64
                ", \"prepares\": %" PRIu64 ", \"carried_over\": %" PRIu64 ", \"noops\": %" PRIu64 ", \"preactive\": %" PRIu64
                ", \"logged_accepts\": %" PRIu64 ", \"log_batches\": %" PRIu64 ", \"held_replies\": %" PRIu64
                ", \"accepts_resent\": %" PRIu64 ", \"sync_requests\": %" PRIu64 ", \"sync_decisions_sent\": %" PRIu64 ", \"sync_decisions_applied\": %" PRIu64 "}",
                i ? ", " : "", pms[(size_t)i]->myID(), alive[(size_t)i] ? "true" : "false", s.checkpoints, s.pauses, s.unpauses,
                pms[(size_t)i]->pausedCount(), s.proposed, s.batched_requests, s.forwarded,
                s.accepts, s.votes, s.decisions, s.commits, s.executed, s.refused, s.dropped_frames, s.engine_calls,
                s.elections_started, s.elections_won, s.elections_lost, s.prepares, s.carried_over, s.noops,
                s.preactive, s.logged_accepts, s.log_batches, s.held_replies, s.accepts_resent, s.sync_requests, s.sync_decisions_sent, s.sync_decisions_applied);
  }
  std::printf("]}\n");
  return ok ? 0 : 1;
}
