// gpx::PaxosManager - see gpx_host.hpp.  Everything protocol goes through the C-ABI (include/gpx.h,
// include/gpx_wire.h); this file moves frames, keeps request values and performs the upcalls.
#include "gpx_host.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cerrno>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <thread>

namespace gpx {

namespace {

/* big-endian java.nio.ByteBuffer writes */
void put8(Frame& f, int v) { f.push_back((uint8_t)v); }
void put16(Frame& f, int v) {
  f.push_back((uint8_t)(v >> 8));
  f.push_back((uint8_t)v);
}
void put32(Frame& f, int32_t v) {
  for (int s = 24; s >= 0; s -= 8) f.push_back((uint8_t)((uint32_t)v >> s));
}
void put64(Frame& f, int64_t v) {
  for (int s = 56; s >= 0; s -= 8) f.push_back((uint8_t)((uint64_t)v >> s));
}
int32_t jsub32(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); } /* Java int a - b */
/* Ballot.compareTo (paxosutil/Ballot.java:60-73) */
int32_t ballotCmp(int32_t n1, int32_t c1, int32_t n2, int32_t c2) { return n1 != n2 ? jsub32(n1, n2) : jsub32(c1, c2); }
int32_t get32(const uint8_t* p) {
  return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]);
}
int64_t get64(const uint8_t* p) {
  uint64_t v = 0;
  for (int i = 0; i < 8; i++) v = (v << 8) | p[i];
  return (int64_t)v;
}
/* PaxosPacket header (PaxosPacket.java:461-476) */
void putHeader(Frame& f, int32_t type, int32_t version, const std::string& paxosID) {
  put32(f, GPX_WT_PAXOS_PACKET);
  put32(f, type);
  put32(f, version);
  put8(f, (int)paxosID.size());
  f.insert(f.end(), paxosID.begin(), paxosID.end());
}
/* fixed part of RequestPacket.toBytes between the header and the digest length:
 * requestID 8 | stop 1 | client address 4 + 2 | listen address 4 + 2 | entryReplica 4 | entryTime 8 |
 * shouldReturnRequestValue 1 | forwardCount 4 | broadcasted 1 */
constexpr size_t kReqFixed = 8 + 1 + 6 + 6 + 4 + 8 + 1 + 4 + 1;
constexpr size_t kAcceptTail = 4 + 4 + 4 + 1 + 4 + 1 + 4; /* AcceptPacket.java:95-135 */

/* BatchedAcceptReply.toBytes of ONE reply (BatchedAcceptReply.java:119-173): what goes out for a
 * reply gpx_wire_pack_accept_replies left unbatched */
Frame makeSingleAcceptReply(const std::string& paxosID, int32_t version, int32_t acceptor, int32_t bnum,
                            int32_t bcoord, int32_t slot, int32_t maxCheckpointedSlot, int64_t requestID) {
  Frame f;
  putHeader(f, GPX_WT_BATCHED_ACCEPT_REPLY, version, paxosID);
  put32(f, acceptor);
  put32(f, bnum);
  put32(f, bcoord);
  put32(f, slot);
  put32(f, maxCheckpointedSlot);
  put64(f, requestID);
  put8(f, 0);
  put32(f, 1);
  put32(f, slot);
  put64(f, requestID);
  return f;
}

}  // namespace

/* ---- loggers -------------------------------------------------------------------------------- */

uint64_t DelayLogger::logBatch(const std::vector<const Frame*>& recs) {
  for (const Frame* f : recs) records++, bytes += f->size();
  pending_.push_back({next_, polls_});
  return next_++;
}
uint64_t DelayLogger::durable() {
  polls_++;
  while (!pending_.empty() && polls_ - pending_.front().second > (uint64_t)delay_) {
    durable_ = pending_.front().first;
    pending_.pop_front();
  }
  return durable_;
}

struct FileLogger::Impl {
  int fd = -1;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::pair<uint64_t, std::vector<uint8_t>>> queue; /* (ticket, the batch's bytes) */
  std::atomic<uint64_t> durable{0};
  std::atomic<bool> failed{false}; /* a write or fdatasync failed: nothing after it is durable */
  uint64_t next = 1;
  bool stop = false;
  std::thread worker;
  void run() {
    for (;;) {
      std::pair<uint64_t, std::vector<uint8_t>> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !queue.empty(); });
        if (queue.empty()) return;
        job = std::move(queue.front());
        queue.pop_front();
      }
      /* a batch is durable only when every byte was written AND fdatasync succeeded; EINTR is
       * retried; any other failure latches `failed` and `durable` never moves again, so the manager
       * stops releasing the replies it holds (reply-after-log, AbstractPaxosLogger.java:691-715) */
      size_t off = 0;
      bool ok = fd >= 0 && !failed.load(std::memory_order_acquire);
      while (ok && off < job.second.size()) {
        const ssize_t w = ::write(fd, job.second.data() + off, job.second.size() - off);
        if (w < 0 && errno == EINTR) continue;
        if (w <= 0) {
          ok = false;
          break;
        }
        off += (size_t)w;
      }
      while (ok) {
        if (::fdatasync(fd) == 0) break;
        if (errno != EINTR) ok = false;
      }
      if (!ok) {
        failed.store(true, std::memory_order_release);
        continue; /* keep draining the queue so that the destructor can join; nothing becomes durable */
      }
      durable.store(job.first, std::memory_order_release);
    }
  }
};
FileLogger::FileLogger(const std::string& path) : impl_(new Impl) {
  impl_->fd = ::open(path.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
  if (impl_->fd < 0) {
    const int err = errno;
    delete impl_;
    impl_ = nullptr;
    throw std::runtime_error("FileLogger: cannot open " + path + ": " + std::strerror(err));
  }
  impl_->worker = std::thread([this] { impl_->run(); });
}
FileLogger::~FileLogger() {
  {
    std::lock_guard<std::mutex> lk(impl_->mu);
    impl_->stop = true;
  }
  impl_->cv.notify_all();
  impl_->worker.join();
  if (impl_->fd >= 0) ::close(impl_->fd);
  delete impl_;
}
uint64_t FileLogger::logBatch(const std::vector<const Frame*>& recs) {
  std::vector<uint8_t> bytes;
  for (const Frame* f : recs) { /* length-prefixed records, one write + one fdatasync per batch */
    const uint32_t n = (uint32_t)f->size();
    for (int sft = 24; sft >= 0; sft -= 8) bytes.push_back((uint8_t)(n >> sft));
    bytes.insert(bytes.end(), f->begin(), f->end());
  }
  uint64_t t;
  {
    std::lock_guard<std::mutex> lk(impl_->mu);
    t = impl_->next++;
    impl_->queue.emplace_back(t, std::move(bytes));
  }
  impl_->cv.notify_one();
  return t;
}
uint64_t FileLogger::durable() { return impl_->durable.load(std::memory_order_acquire); }
bool FileLogger::failed() const { return impl_->failed.load(std::memory_order_acquire); }

int32_t javaStringHash(const std::string& s) {
  uint32_t h = 0;
  for (unsigned char c : s) h = 31u * h + c;
  return (int32_t)h;
}

int32_t roundRobinCoordinator(const std::string& paxosID, const std::vector<int32_t>& members, int32_t ballotnum) {
  /* members[Math.abs(ballotnum + paxosID.hashCode()) % members.length] (PISM:2251-2256) */
  const int32_t x = (int32_t)((uint32_t)ballotnum + (uint32_t)javaStringHash(paxosID));
  const int64_t a = x == INT32_MIN ? (int64_t)INT32_MIN : (x < 0 ? -(int64_t)x : (int64_t)x);
  int64_t idx = a % (int64_t)members.size(); /* sign follows the dividend, as in Java */
  if (idx < 0) return INT32_MIN;             /* the Java would throw ArrayIndexOutOfBounds */
  return members[(size_t)idx];
}

Frame makeRequestFrame(const std::string& paxosID, int32_t version, int64_t requestID, const std::string& value,
                       bool stop, int32_t entryReplica) {
  Frame f;
  f.reserve(13 + paxosID.size() + kReqFixed + 16 + value.size());
  putHeader(f, GPX_WT_REQUEST, version, paxosID);
  put64(f, requestID);
  put8(f, stop ? 1 : 0);
  put32(f, 0), put16(f, 0); /* client address */
  put32(f, 0), put16(f, 0); /* listen address */
  put32(f, entryReplica);
  put64(f, 0); /* entryTime */
  put8(f, 0);  /* shouldReturnRequestValue */
  put32(f, 0); /* forwardCount */
  put8(f, 0);  /* broadcasted */
  put32(f, 0); /* digest */
  put32(f, (int32_t)value.size());
  f.insert(f.end(), value.begin(), value.end());
  put32(f, 0); /* response */
  put32(f, 0); /* batched */
  return f;
}

Frame makeAcceptFrame(const Frame& req, int32_t slot, int32_t bnum, int32_t bcoord, int32_t median, int32_t sender) {
  Frame f;
  f.reserve(req.size() + kAcceptTail);
  f = req;
  const int32_t t = GPX_WT_ACCEPT; /* the packet type int of the header */
  for (int i = 0; i < 4; i++) f[4 + i] = (uint8_t)((uint32_t)t >> (24 - 8 * i));
  put32(f, slot);
  put32(f, bnum);
  put32(f, bcoord);
  put8(f, 0); /* recovery */
  put32(f, median);
  put8(f, 0); /* noCoalesce */
  put32(f, sender);
  return f;
}

/* one RequestPacket at f[at ..); *end = the offset of its batched-count field */
static bool parseRequestAt(const uint8_t* f, size_t size, size_t at, Request* out, size_t* end) {
  if (size < at + 13) return false;
  const size_t idLen = f[at + 12];
  size_t p = at + 13 + idLen;
  if (size < p + kReqFixed + 8) return false;
  out->paxosID.assign((const char*)&f[at + 13], idLen);
  out->requestID = get64(&f[p]);
  out->stop = f[p + 8] != 0;
  out->entryReplica = get32(&f[p + 8 + 1 + 12]);
  p += kReqFixed;
  const int32_t dl = get32(&f[p]);
  p += 4;
  if (dl < 0 || size < p + (size_t)dl + 4) return false;
  p += (size_t)dl;
  const int32_t vl = get32(&f[p]);
  p += 4;
  if (vl < 0 || size < p + (size_t)vl + 4) return false;
  out->requestValue.assign((const char*)&f[p], (size_t)vl);
  p += (size_t)vl;
  const int32_t rl = get32(&f[p]); /* response */
  p += 4;
  if (rl < 0 || size < p + (size_t)rl + 4) return false;
  *end = p + (size_t)rl;
  return true;
}

/* RequestPacket.isStopRequest(): its own flag or any batched request's (RequestPacket.java:371-378) */
static bool frameIsStop(const Frame& f, bool* noop = nullptr) {
  std::vector<Request> rs;
  if (!parseRequests(f, &rs)) return false;
  bool stop = false;
  for (auto& r : rs) stop = stop || r.stop;
  if (noop) *noop = rs.size() == 1 && rs[0].isNoop();
  return stop;
}

bool parseRequest(const Frame& f, Request* out) {
  size_t end = 0;
  return parseRequestAt(f.data(), f.size(), 0, out, &end);
}

bool parseRequests(const Frame& f, std::vector<Request>* out) {
  Request head;
  size_t p = 0;
  if (!parseRequestAt(f.data(), f.size(), 0, &head, &p)) return false;
  out->push_back(head);
  const int32_t nb = get32(&f[p]);
  p += 4;
  for (int32_t j = 0; j < nb; j++) {
    if (f.size() < p + 4) return false;
    const int32_t len = get32(&f[p]);
    p += 4;
    Request r;
    size_t e = 0;
    if (len < 0 || f.size() < p + (size_t)len || !parseRequestAt(f.data(), p + (size_t)len, p, &r, &e)) return false;
    out->push_back(r);
    p += (size_t)len;
  }
  return true;
}

/* the unbatched requests a frame holds, each as a stand-alone byte array (RequestPacket.toArray,
 * :1117-1127): its head with an empty batched array, then its batched elements as they are */
static bool flattenRequests(const Frame& f, std::vector<Frame>* out) {
  Request head;
  size_t p = 0;
  if (!parseRequestAt(f.data(), f.size(), 0, &head, &p) || f.size() < p + 4) return false;
  Frame h(f.begin(), f.begin() + (long)p);
  put32(h, 0);
  out->push_back(std::move(h));
  const int32_t nb = get32(&f[p]);
  p += 4;
  for (int32_t j = 0; j < nb; j++) {
    if (f.size() < p + 4) return false;
    const int32_t len = get32(&f[p]);
    p += 4;
    if (len < 0 || f.size() < p + (size_t)len) return false;
    out->emplace_back(f.begin() + (long)p, f.begin() + (long)(p + (size_t)len));
    p += (size_t)len;
  }
  return true;
}

int32_t batchSizeOf(const Frame& f) {
  Request head;
  size_t p = 0;
  if (!parseRequestAt(f.data(), f.size(), 0, &head, &p) || f.size() < p + 4) return 0;
  return get32(&f[p]);
}

Frame latchToBatch(const Frame& first, const std::vector<const Frame*>& rest) {
  /* first flatten out the argument; batched = concatenate(this.batched, allThreaded) */
  std::vector<Frame> all;
  if (!flattenRequests(first, &all)) return first;
  for (const Frame* r : rest) flattenRequests(*r, &all);
  Frame f(all[0].begin(), all[0].end() - 4); /* the head up to its batched-count field */
  put32(f, (int32_t)all.size() - 1);
  for (size_t i = 1; i < all.size(); i++) {
    put32(f, (int32_t)all[i].size());
    f.insert(f.end(), all[i].begin(), all[i].end());
  }
  return f;
}

/* ------------------------------------------------------------------------------------------ */

PaxosManager::PaxosManager(int32_t myID, Replicable* app, Messenger* messenger, const Options& opt)
    : myID_(myID), app_(app), messenger_(messenger), opt_(opt), nextRequestID_(((int64_t)myID << 40) + 1) {
  gpx_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.my_id = myID;
  cfg.max_groups = opt.maxGroups;
  cfg.kmax = opt.kmax;
  cfg.window = opt.window;
  cfg.max_batch = opt.maxBatch;
  cfg.device = opt.device;
  cfg.flags = GPX_F_ACCEPTS_FROM_DISK;
  if (!check(gpx_engine_create(&cfg, &engine_), "gpx_engine_create")) engine_ = nullptr;
  rowName_.resize((size_t)opt.maxGroups);
  lastActive_.assign((size_t)opt.maxGroups, 0);
  liveAccepts_.assign((size_t)opt.maxGroups, 0);
}

PaxosManager::~PaxosManager() {
  if (engine_) gpx_engine_destroy(engine_);
}

bool PaxosManager::check(int rc, const char* what) {
  if (rc >= 0) return true;
  char buf[256];
  std::snprintf(buf, sizeof(buf), "%s failed rc=%d %s", what, rc, gpx_last_error());
  err_ = buf;
  std::fprintf(stderr, "gpx::PaxosManager(%d): %s\n", myID_, buf);
  return false;
}

bool PaxosManager::createPaxosInstance(const std::string& paxosID, const std::vector<int32_t>& members,
                                       const std::string& initialState) {
  if (!initialState.empty()) app_->restore(paxosID, initialState);
  return createPaxosInstances({paxosID}, members) == 1;
}

int PaxosManager::createPaxosInstances(const std::vector<std::string>& ids, const std::vector<int32_t>& gms) {
  if (!engine_ || ids.empty()) return 0;
  std::vector<int32_t> members(gms);
  std::sort(members.begin(), members.end()); /* PISM:205 */
  const int32_t k = (int32_t)members.size();
  if (k < 1 || k > opt_.kmax) return 0;
  std::vector<std::string> fresh;
  for (auto& id : ids)
    if (!id.empty() && id.size() <= GPX_W_MAX_NAME && !pinstances_.count(id)) fresh.push_back(id);
  const int32_t n = (int32_t)fresh.size();
  if (n == 0) return 0;
  std::vector<int32_t> gidx((size_t)n);
  pass_++; /* groups made by earlier calls count as idle for this one */
  if (!makeRoom(n)) return 0;
  if (!check(gpx_rows_alloc(engine_, n, gidx.data()), "gpx_rows_alloc")) return 0;
  std::vector<int32_t> mem((size_t)n * opt_.kmax, 0);
  std::vector<uint8_t> ks((size_t)n, (uint8_t)k), st((size_t)n, 0);
  std::vector<gpx_hri> rows((size_t)n);
  for (int32_t i = 0; i < n; i++) {
    for (int32_t j = 0; j < k; j++) mem[(size_t)i * opt_.kmax + j] = members[(size_t)j];
    /* regular creation with an initial-state checkpoint (PISM:612-618, 656-668, 692-699): acceptor
     * slot 1, gcSlot 0, ballot (0, coordinator); the coordinator object only on the coordinator */
    const int32_t coord = roundRobinCoordinator(fresh[(size_t)i], members, 0);
    gpx_hri& r = rows[(size_t)i];
    std::memset(&r, 0, sizeof(r));
    r.acc_slot = 1;
    r.acc_bcoord = coord;
    r.acc_gc_slot = 0;
    r.has_coord = coord == myID_ ? 1 : 0;
    r.coord_bcoord = coord;
    r.next_proposal_slot = r.has_coord ? 1 : -1;
    for (int32_t j = 0; j < k; j++) r.node_slots[j] = r.has_coord ? -1 : 0;
  }
  if (!check(gpx_group_create(engine_, n, gidx.data(), mem.data(), ks.data(), rows.data(), st.data()),
             "gpx_group_create"))
    return 0;
  std::vector<uint8_t> names;
  std::vector<int32_t> off((size_t)n + 1, 0);
  for (int32_t i = 0; i < n; i++) {
    names.insert(names.end(), fresh[(size_t)i].begin(), fresh[(size_t)i].end());
    off[(size_t)i + 1] = (int32_t)names.size();
  }
  std::vector<uint8_t> bst((size_t)n, 0);
  if (!check(gpx_names_bind(engine_, n, names.data(), off.data(), gidx.data(), bst.data()), "gpx_names_bind"))
    return 0;
  int made = 0;
  for (int32_t i = 0; i < n; i++) {
    if (st[(size_t)i] != GPX_S_OK || bst[(size_t)i] != GPX_S_OK) continue;
    pinstances_[fresh[(size_t)i]] = Instance{gidx[(size_t)i], 0, members};
    rowName_[(size_t)gidx[(size_t)i]] = fresh[(size_t)i];
    lastActive_[(size_t)gidx[(size_t)i]] = pass_;
    liveAccepts_[(size_t)gidx[(size_t)i]] = 0;
    liveRows_++;
    made++;
  }
  return made;
}

bool PaxosManager::kill(const std::string& paxosID) {
  auto it = pinstances_.find(paxosID);
  if (it == pinstances_.end() || !engine_) return false;
  const int32_t g = it->second.gidx;
  uint8_t st = 0;
  check(gpx_group_retire(engine_, 1, &g, GPX_RETIRE_KILL, nullptr, &st), "gpx_group_retire");
  check(gpx_names_unbind(engine_, 1, &g, &st), "gpx_names_unbind");
  check(gpx_rows_free(engine_, 1, &g), "gpx_rows_free");
  forgetRow(g);
  decided_.erase(paxosID);
  rowName_[(size_t)g].clear();
  liveRows_--;
  pinstances_.erase(it);
  return true;
}

void PaxosManager::forgetRow(int32_t g) {
  for (auto a = accepted_.begin(); a != accepted_.end();)
    a = (int32_t)(a->first >> 32) == g ? accepted_.erase(a) : std::next(a);
  for (auto a = syncAsked_.begin(); a != syncAsked_.end();)
    a = (int32_t)(a->first >> 32) == g ? syncAsked_.erase(a) : std::next(a);
  for (auto c = carried_.lower_bound({g, INT32_MIN}); c != carried_.end() && c->first.first == g;) c = carried_.erase(c);
  for (auto* m : {&preactive_})
    for (auto a = m->lower_bound({g, INT64_MIN}); a != m->end() && a->first.first == g;) a = m->erase(a);
  liveAccepts_[(size_t)g] = 0;
}

bool PaxosManager::pause(const std::string& paxosID) {
  auto it = pinstances_.find(paxosID);
  if (it == pinstances_.end() || !engine_) return false;
  const int32_t g = it->second.gidx;
  if (liveAccepts_[(size_t)g] != 0) return false; /* accepted values live only here: keep the group */
  Paused p;
  uint8_t st = 0;
  if (!check(gpx_group_retire(engine_, 1, &g, GPX_RETIRE_PAUSE, &p.hri, &st), "gpx_group_retire")) return false;
  stats_.engine_calls++;
  if (st != GPX_S_OK) return false; /* GPX_S_BUSY: not caught up (tryPause refuses) */
  check(gpx_names_unbind(engine_, 1, &g, &st), "gpx_names_unbind");
  check(gpx_rows_free(engine_, 1, &g), "gpx_rows_free");
  p.members = it->second.members;
  p.version = it->second.version;
  paused_[paxosID] = std::move(p);
  forgetRow(g);
  rowName_[(size_t)g].clear();
  liveRows_--;
  pinstances_.erase(it);
  stats_.pauses++;
  return true;
}

/* frees `rows` rows of the device table by pausing the groups idle the longest */
bool PaxosManager::makeRoom(int32_t rows) {
  if (liveRows_ + rows <= opt_.maxGroups) return true;
  std::vector<std::pair<uint64_t, int32_t>> idle;
  for (auto& kv : pinstances_)
    if (lastActive_[(size_t)kv.second.gidx] < pass_ && liveAccepts_[(size_t)kv.second.gidx] == 0)
      idle.push_back({lastActive_[(size_t)kv.second.gidx], kv.second.gidx});
  std::sort(idle.begin(), idle.end());
  for (auto& c : idle) {
    if (liveRows_ + rows <= opt_.maxGroups) break;
    pause(std::string(rowName_[(size_t)c.second]));
  }
  return liveRows_ + rows <= opt_.maxGroups;
}

bool PaxosManager::unpause(const std::string& paxosID) {
  auto it = paused_.find(paxosID);
  if (it == paused_.end() || !engine_) return false;
  if (!makeRoom(1)) return false;
  int32_t g = -1;
  if (!check(gpx_rows_alloc(engine_, 1, &g), "gpx_rows_alloc")) return false;
  const Paused& p = it->second;
  std::vector<int32_t> mem((size_t)opt_.kmax, 0);
  for (size_t j = 0; j < p.members.size(); j++) mem[j] = p.members[j];
  const uint8_t k = (uint8_t)p.members.size();
  uint8_t st = 0;
  /* hotRestore(hri) (PISM:677-690) */
  if (!check(gpx_group_create(engine_, 1, &g, mem.data(), &k, &p.hri, &st), "gpx_group_create") || st != GPX_S_OK)
    return false;
  const int32_t off[2] = {0, (int32_t)paxosID.size()};
  if (!check(gpx_names_bind(engine_, 1, (const uint8_t*)paxosID.data(), off, &g, &st), "gpx_names_bind")) return false;
  stats_.engine_calls += 2;
  pinstances_[paxosID] = Instance{g, p.version, p.members};
  rowName_[(size_t)g] = paxosID;
  lastActive_[(size_t)g] = pass_;
  liveAccepts_[(size_t)g] = 0;
  liveRows_++;
  paused_.erase(it);
  stats_.unpauses++;
  return true;
}

int64_t PaxosManager::propose(const std::string& paxosID, const std::string& value, bool stop,
                              ExecutedCallback callback) {
  auto it = pinstances_.find(paxosID);
  int32_t version = 0;
  if (it != pinstances_.end()) {
    version = it->second.version;
  } else {
    auto pz = paused_.find(paxosID); /* comes back when the request is processed */
    if (pz == paused_.end()) return 0;
    version = pz->second.version;
  }
  const int64_t id = nextRequestID_++;
  requests_.push_back(makeRequestFrame(paxosID, version, id, value, stop, myID_));
  if (callback) callbacks_[id] = std::move(callback);
  return id;
}

void PaxosManager::handleIncomingPacket(const uint8_t* frame, size_t len) { inbox_.emplace_back(frame, frame + len); }
void PaxosManager::handleIncomingPacket(Frame&& frame) { inbox_.push_back(std::move(frame)); }

void PaxosManager::sendToMembers(const Instance& in, const Frame& frame, bool includeSelf) {
  /* loopback first, then the others ascending (SHORT_CIRCUIT_LOCAL, PaxosManager.java:2116-2128) */
  if (includeSelf) inbox_.push_back(frame);
  for (int32_t m : in.members)
    if (m != myID_) messenger_->send(m, Frame(frame));
}

void PaxosManager::executeRuns(int32_t nRuns, const int32_t* xg, const int32_t* xf, const int32_t* xc) {
  /* extractExecuteAndCheckpoint's upcalls (PISM:1619-1701, 1755-1842): per run, slot order */
  for (int32_t r = 0; r < nRuns; r++) {
    for (int32_t j = 0; j < xc[r]; j++) {
      const int32_t slot = (int32_t)((uint32_t)xf[r] + (uint32_t)j);
      auto a = accepted_.find(key(xg[r], slot));
      if (a == accepted_.end()) {
        err_ = "decided slot without a stored ACCEPT";
        std::fprintf(stderr, "gpx::PaxosManager(%d): %s (gidx %d slot %d)\n", myID_, err_.c_str(), xg[r], slot);
        continue;
      }
      std::vector<Request> reqs; /* the request and the requests batched into it, in that order */
      if (parseRequests(a->second.frame, &reqs))
        for (Request& req : reqs) {
          req.slot = slot;
          /* doNotReplyToClient unless this node is the entry replica (PISM:1800-1806); no-ops are fed
           * to the application as they are to TESTPaxosApp */
          for (int tries = 0; tries < 3 && !app_->execute(req, req.entryReplica != myID_); tries++) {
          }
          stats_.executed++;
          if (req.entryReplica == myID_) { /* PaxosManager.executed: the entry replica answers its client */
            auto cb = callbacks_.find(req.requestID);
            if (cb != callbacks_.end()) {
              cb->second(req);
              callbacks_.erase(cb);
              stats_.callbacks++;
            }
          }
          /* consistentCheckpoint every CHECKPOINT_INTERVAL slots and on a stop (PISM:1711-1723, 2037-2041) */
          if (&req == &reqs.back() && (req.stop || (opt_.checkpointInterval > 0 && slot % opt_.checkpointInterval == 0))) {
            (void)app_->checkpoint(req.paxosID);
            stats_.checkpoints++;
          }
        }
      /* the decision stays available to replicas that missed its commit (the logger's job in the
       * reference), a bounded number of slots back */
      syncAsked_.erase(key(xg[r], slot));
      auto& log = decided_[rowName_[(size_t)xg[r]]];
      log[slot] = std::move(a->second);
      log.erase((int32_t)((uint32_t)slot - (uint32_t)opt_.decisionLogSlots));
      accepted_.erase(a); /* acceptedProposals.remove(slot) on execution (PaxosAcceptor.java:357-359) */
      liveAccepts_[(size_t)xg[r]]--;
    }
  }
}

void PaxosManager::issueAccept(std::vector<OutAccept>& out, int32_t gidx, const Frame& requestFrame,
                               int64_t requestID, bool stop, int32_t slot, int32_t bnum, int32_t bcoord,
                               int32_t median) {
  const Instance& in = pinstances_.at(rowName_[(size_t)gidx]);
  Frame acc = makeAcceptFrame(requestFrame, slot, bnum, bcoord, median, myID_);
  for (int32_t m : in.members)
    if (m != myID_) messenger_->send(m, Frame(acc));
  out.push_back(OutAccept{gidx, bnum, bcoord, slot, median, (uint8_t)(stop ? GPX_A_STOP : 0), requestID,
                          std::move(acc)});
}

size_t PaxosManager::nodeDown(int32_t nodeID) {
  if (!engine_) return 0;
  /* lastCoordinatorLongDead (PISM:2090-2176): a node reported down again, or already down when another
   * node fails, has been dead for long - any member may then run, not only the next in line (else a
   * group whose next-in-line member is down too would never get a coordinator) */
  std::vector<int32_t> longDead;
  for (int32_t d : downNodes_) longDead.push_back(d);
  if (std::find(downNodes_.begin(), downNodes_.end(), nodeID) == downNodes_.end()) downNodes_.push_back(nodeID);
  /* checkRunForCoordinator's decision for every instance at once */
  const int32_t n = opt_.maxGroups;
  std::vector<uint8_t> run((size_t)n), st((size_t)n);
  std::vector<int32_t> pb((size_t)n), pf((size_t)n);
  if (!check(gpx_election_scan(engine_, n, nullptr, downNodes_.data(), (int32_t)downNodes_.size(),
                               longDead.empty() ? nullptr : longDead.data(), (int32_t)longDead.size(), 0,
                               run.data(), pb.data(), pf.data(), st.data()),
             "gpx_election_scan"))
    return 0;
  stats_.engine_calls++;
  std::vector<int32_t> g, b;
  for (int32_t i = 0; i < n; i++)
    if (st[(size_t)i] == GPX_S_OK && run[(size_t)i] != GPX_RUN_NO) g.push_back(i), b.push_back(pb[(size_t)i]);
  if (g.empty()) return 0;
  std::vector<uint8_t> es(g.size());
  if (!check(gpx_election_begin(engine_, (int32_t)g.size(), g.data(), b.data(), es.data()), "gpx_election_begin"))
    return 0;
  stats_.engine_calls++;
  size_t started = 0;
  for (size_t i = 0; i < g.size(); i++) {
    if (es[i] != GPX_EB_PREPARING && es[i] != GPX_EB_RESEND) continue;
    /* PreparePacket(newBallot, paxosState.getSlot()) to all members, myself included (PISM:2153-2160) */
    const std::string& name = rowName_[(size_t)g[i]];
    Frame f;
    putHeader(f, kTypePrepare, 0, name);
    put32(f, b[i]);
    put32(f, myID_);
    put32(f, pf[(size_t)g[i]]);
    sendToMembers(pinstances_.at(name), f, true);
    started++;
  }
  stats_.elections_started += started;
  return started;
}

size_t PaxosManager::poke() {
  if (!engine_) return 0;
  const int32_t n = opt_.maxGroups;
  std::vector<uint8_t> pk((size_t)n), fl((size_t)n), st((size_t)n);
  std::vector<int32_t> sl((size_t)n), bn((size_t)n), bc((size_t)n), md((size_t)n);
  std::vector<uint32_t> heard((size_t)n);
  if (!check(gpx_poke_scan(engine_, n, nullptr, pk.data(), sl.data(), bn.data(), bc.data(), md.data(), fl.data(),
                           heard.data(), st.data()),
             "gpx_poke_scan"))
    return 0;
  stats_.engine_calls++;
  size_t resent = 0;
  for (int32_t g = 0; g < n; g++) {
    if (st[(size_t)g] != GPX_S_OK || pk[(size_t)g] == GPX_POKE_NONE) continue;
    const Instance& in = pinstances_.at(rowName_[(size_t)g]);
    if (pk[(size_t)g] == GPX_POKE_ACCEPT) {
      /* reInitCommander: the same pvalue, the median as it is now; to the members not heard from */
      auto a = accepted_.find(key(g, sl[(size_t)g]));
      if (a == accepted_.end()) continue;
      Frame req(a->second.frame.begin(), a->second.frame.end() - (long)kAcceptTail);
      Frame acc = makeAcceptFrame(req, sl[(size_t)g], bn[(size_t)g], bc[(size_t)g], md[(size_t)g], myID_);
      for (size_t j = 0; j < in.members.size(); j++)
        if (in.members[j] != myID_ && !((heard[(size_t)g] >> j) & 1u)) messenger_->send(in.members[j], Frame(acc)), resent++;
      stats_.accepts_resent++;
    } else { /* GPX_POKE_PREPARE */
      Frame f;
      putHeader(f, kTypePrepare, 0, rowName_[(size_t)g]);
      put32(f, bn[(size_t)g]), put32(f, bc[(size_t)g]), put32(f, sl[(size_t)g]);
      for (size_t j = 0; j < in.members.size(); j++)
        if (in.members[j] != myID_ && !((heard[(size_t)g] >> j) & 1u)) messenger_->send(in.members[j], Frame(f)), resent++;
      stats_.prepares_resent++;
    }
  }
  /* ... and the periodic decision sync (SyncMode.FORCE_SYNC on a poke, PISM:2341-2364): slots that
   * are committed here without their value, or missing below a commit, are asked for again - from
   * every other member, whoever still holds the decision answers */
  std::vector<int32_t> live;
  for (auto& kv : pinstances_) live.push_back(kv.second.gidx);
  std::sort(live.begin(), live.end());
  const int32_t m = (int32_t)live.size();
  if (m > 0) {
    /* stored ACCEPT values of slots below the acceptor's slot are garbage (late or re-sent ACCEPTs of
     * executed slots, decisions the engine refused): the mirror of the engine's own cleanup at
     * execution (PaxosAcceptor.java:357-359) - without it a group with such a leftover never pauses */
    std::vector<gpx_hri> rows((size_t)m);
    std::vector<uint8_t> rst((size_t)m);
    if (check(gpx_group_snapshot(engine_, m, live.data(), rows.data(), rst.data()), "gpx_group_snapshot")) {
      stats_.engine_calls++;
      std::vector<int32_t> accSlot((size_t)opt_.maxGroups, 0);
      std::vector<uint8_t> known((size_t)opt_.maxGroups, 0);
      for (int32_t i = 0; i < m; i++)
        if (rst[(size_t)i] == GPX_S_OK) accSlot[(size_t)live[(size_t)i]] = rows[(size_t)i].acc_slot, known[(size_t)live[(size_t)i]] = 1;
      for (auto a = accepted_.begin(); a != accepted_.end();) {
        const int32_t g = (int32_t)(a->first >> 32), slot = (int32_t)(uint32_t)a->first;
        if (known[(size_t)g] && jsub32(slot, accSlot[(size_t)g]) < 0) {
          a = accepted_.erase(a);
          liveAccepts_[(size_t)g]--;
        } else {
          ++a;
        }
      }
    }
    std::vector<int32_t> first((size_t)m), maxc((size_t)m);
    std::vector<uint64_t> missing((size_t)m);
    std::vector<uint8_t> sync((size_t)m), gst((size_t)m);
    if (check(gpx_gap_scan(engine_, m, live.data(), opt_.syncGapThreshold, GPX_SYNC_FORCE, 64, first.data(),
                           maxc.data(), missing.data(), sync.data(), gst.data()),
              "gpx_gap_scan")) {
      stats_.engine_calls++;
      for (int32_t i = 0; i < m; i++) {
        /* something is committed at or beyond my next slot and I have not executed it: the missing
         * slots below the newest commit, or - none missing - my next slot itself, whose commit may
         * be a placeholder without a value (requestMissingDecisions, PISM:2292-2300) */
        if (gst[(size_t)i] != GPX_S_OK) continue;
        /* ... or an accepted value is still waiting for its decision here: its commit may have been
         * lost with nothing newer to reveal the gap */
        if (jsub32(maxc[(size_t)i], first[(size_t)i]) < 0 && liveAccepts_[(size_t)live[(size_t)i]] == 0) continue;
        uint64_t mask = missing[(size_t)i];
        if (!mask) mask = 1;
        Frame f;
        putHeader(f, kTypeSyncDecisions, 0, rowName_[(size_t)live[(size_t)i]]);
        put32(f, myID_);
        put32(f, __builtin_popcountll(mask));
        for (int j = 0; j < 64; j++)
          if ((mask >> j) & 1ull) put32(f, (int32_t)((uint32_t)first[(size_t)i] + (uint32_t)j));
        const Instance& in = pinstances_.at(rowName_[(size_t)live[(size_t)i]]);
        for (int32_t mem : in.members)
          if (mem != myID_) messenger_->send(mem, Frame(f)), resent++;
        stats_.sync_requests++;
      }
    }
  }
  return resent;
}

/* PISM.handlePrepare (PISM:900-1006): adopt a higher ballot, answer with the accepted pvalues */
bool PaxosManager::handlePrepares(std::vector<Frame>& prepares) {
  const int32_t n = (int32_t)prepares.size();
  std::vector<int32_t> g, bn, bc, fs;
  for (auto& f : prepares) {
    const size_t idLen = f.size() > 12 ? f[12] : 0, p = 13 + idLen;
    auto it = f.size() >= p + 12 ? pinstances_.find(std::string((const char*)&f[13], idLen)) : pinstances_.end();
    if (it == pinstances_.end()) {
      if (f.size() >= p + 12 && paused_.count(std::string((const char*)&f[13], idLen)))
        retry_.push_back(std::move(f));
      else
        stats_.dropped_frames++;
      continue;
    }
    g.push_back(it->second.gidx), bn.push_back(get32(&f[p])), bc.push_back(get32(&f[p + 4]));
    fs.push_back(get32(&f[p + 8]));
  }
  const int32_t m = (int32_t)g.size();
  if (m == 0) return n > 0;
  const size_t W = (size_t)opt_.window;
  std::vector<int32_t> rb((size_t)m), rc((size_t)m), rg((size_t)m), ps((size_t)m * W), pb((size_t)m * W),
      pc((size_t)m * W);
  std::vector<uint8_t> rf((size_t)m), st((size_t)m);
  std::vector<uint64_t> mask((size_t)m);
  if (!check(gpx_prepare_batch(engine_, m, g.data(), bn.data(), bc.data(), fs.data(), rb.data(), rc.data(),
                               rg.data(), rf.data(), mask.data(), ps.data(), pb.data(), pc.data(), st.data()),
             "gpx_prepare_batch"))
    return false;
  stats_.engine_calls++;
  stats_.prepares += (uint64_t)m;
  for (int32_t i = 0; i < m; i++) {
    if (st[(size_t)i] != GPX_S_OK) continue;
    /* PrepareReplyPacket(myID, the acceptor's ballot now, accepted pvalues, gcSlot): a NACK carries
     * the higher ballot and nothing else, which makes the candidate resign */
    Frame f, body;
    int32_t cnt = 0;
    for (size_t w = 0; w < W; w++) {
      if (!((mask[(size_t)i] >> w) & 1ull)) continue;
      const size_t q = w * (size_t)m + (size_t)i;
      auto a = accepted_.find(key(g[(size_t)i], ps[q]));
      if (a == accepted_.end()) continue; /* value not at hand: the logger's job in the reference */
      put32(body, ps[q]), put32(body, pb[q]), put32(body, pc[q]);
      put32(body, (int32_t)a->second.frame.size());
      body.insert(body.end(), a->second.frame.begin(), a->second.frame.end());
      cnt++;
    }
    /* ... plus what the reference reads back from its log (getLoggedAccepts, PISM:953-975): with
     * GET_ACCEPTED_PVALUES_FROM_DISK an executed slot's accept leaves memory but not the log, and a
     * candidate whose firstUndecidedSlot is at or below it must still hear about it - or it would
     * fill the slot with something else */
    if (!(rf[(size_t)i] & GPX_P_NACK)) {
      auto lg = decided_.find(rowName_[(size_t)g[(size_t)i]]);
      if (lg != decided_.end())
        for (auto& kv : lg->second) {
          if (jsub32(kv.first, fs[(size_t)i]) < 0) continue;
          put32(body, kv.first), put32(body, kv.second.bnum), put32(body, kv.second.bcoord);
          put32(body, (int32_t)kv.second.frame.size());
          body.insert(body.end(), kv.second.frame.begin(), kv.second.frame.end());
          cnt++;
        }
    }
    putHeader(f, kTypePrepareReply, 0, rowName_[(size_t)g[(size_t)i]]);
    put32(f, myID_), put32(f, rb[(size_t)i]), put32(f, rc[(size_t)i]);
    put32(f, (int32_t)((uint32_t)rg[(size_t)i] + 1u)); /* firstSlot = gcSlot + 1 */
    put32(f, cnt);
    f.insert(f.end(), body.begin(), body.end());
    /* a PREPARE that raised my ballot is logged before its reply leaves (GPX_P_TOLOG, PISM:985-1000) */
    if (opt_.logger && (rf[(size_t)i] & GPX_P_TOLOG)) {
      Frame rec; /* what has to survive: the ballot promised for this group */
      putHeader(rec, kTypePrepare, 0, rowName_[(size_t)g[(size_t)i]]);
      put32(rec, rb[(size_t)i]), put32(rec, rc[(size_t)i]), put32(rec, fs[(size_t)i]);
      const uint64_t ticket = opt_.logger->logBatch({&rec});
      held_.push_back(Held{ticket, bc[(size_t)i], std::move(f)});
      stats_.held_replies++;
      stats_.log_batches++;
    } else if (bc[(size_t)i] == myID_) {
      inbox_.push_back(std::move(f));
    } else {
      messenger_->send(bc[(size_t)i], std::move(f)); /* to the PREPARE's sender */
    }
  }
  return true;
}

/* PISM.handlePrepareReply (PISM:1008-1068): record, and on a majority re-issue what was carried over */
bool PaxosManager::handlePrepareReplies(std::vector<Frame>& replies, std::vector<OutAccept>& out) {
  std::vector<int32_t> g, acc, rb, rc, fs, off(1, 0), ps, pb, pc;
  std::vector<int64_t> ph;
  std::vector<uint8_t> pfl;
  std::vector<Frame> pvFrames; /* frame of pvalue entry j (parallel to ps / pb / pc) */
  for (auto& f : replies) {
    const size_t idLen = f.size() > 12 ? f[12] : 0;
    size_t p = 13 + idLen;
    auto it = f.size() >= p + 20 ? pinstances_.find(std::string((const char*)&f[13], idLen)) : pinstances_.end();
    if (it == pinstances_.end()) {
      if (f.size() >= p + 20 && paused_.count(std::string((const char*)&f[13], idLen)))
        retry_.push_back(std::move(f));
      else
        stats_.dropped_frames++;
      continue;
    }
    const int32_t gi = it->second.gidx;
    g.push_back(gi), acc.push_back(get32(&f[p])), rb.push_back(get32(&f[p + 4])), rc.push_back(get32(&f[p + 8]));
    fs.push_back(get32(&f[p + 12]));
    const int32_t cnt = get32(&f[p + 16]);
    p += 20;
    for (int32_t j = 0; j < cnt && p + 16 <= f.size(); j++) {
      const int32_t len = get32(&f[p + 12]);
      if (len < 0 || p + 16 + (size_t)len > f.size()) break;
      Frame pv(f.begin() + (long)(p + 16), f.begin() + (long)(p + 16 + (size_t)len));
      Request rq;
      if (parseRequest(pv, &rq)) {
        ps.push_back(get32(&f[p])), pb.push_back(get32(&f[p + 4])), pc.push_back(get32(&f[p + 8]));
        ph.push_back(rq.requestID);
        bool noop = false;
        const bool stop = frameIsStop(pv, &noop);
        pfl.push_back((uint8_t)((stop ? GPX_PV_STOP : 0) | (noop ? GPX_PV_NOOP : 0)));
        pvFrames.push_back(std::move(pv));
      }
      p += 16 + (size_t)len;
    }
    off.push_back((int32_t)ps.size());
  }
  const int32_t n = (int32_t)g.size();
  if (n == 0) return true;
  const size_t W = (size_t)opt_.window;
  std::vector<uint8_t> vk((size_t)n), st((size_t)n), ek((size_t)n * W), ef((size_t)n * W);
  std::vector<int32_t> ec((size_t)n), em((size_t)n), es((size_t)n * W);
  std::vector<int64_t> eh((size_t)n * W);
  ps.push_back(0), pb.push_back(0), pc.push_back(0), ph.push_back(0), pfl.push_back(0); /* never null */
  if (!check(gpx_prepare_reply_batch(engine_, n, g.data(), acc.data(), rb.data(), rc.data(), fs.data(), off.data(),
                                     ps.data(), pb.data(), pc.data(), ph.data(), pfl.data(), vk.data(), ec.data(),
                                     em.data(), es.data(), ek.data(), eh.data(), ef.data(), st.data()),
             "gpx_prepare_reply_batch"))
    return false;
  stats_.engine_calls++;
  auto purgeCarried = [&](int32_t gi) {
    for (auto c = carried_.lower_bound({gi, INT32_MIN}); c != carried_.end() && c->first.first == gi;)
      c = carried_.erase(c);
  };
  for (int32_t i = 0; i < n; i++) {
    const int32_t gi = g[(size_t)i];
    const std::string& name = rowName_[(size_t)gi];
    /* only a reply the engine counted contributes carried-over values: per slot the pvalue of the
     * highest ballot, exactly the engine's rule (PCS:349-372); ignored / dropped replies leave nothing */
    if (st[(size_t)i] == GPX_S_OK && (vk[(size_t)i] == GPX_V_RECORDED || vk[(size_t)i] == GPX_V_ELECTED)) {
      for (int32_t q = off[(size_t)i]; q < off[(size_t)i + 1]; q++) {
        auto ins = carried_.find({gi, ps[(size_t)q]});
        if (ins == carried_.end())
          carried_[{gi, ps[(size_t)q]}] = Carried{pb[(size_t)q], pc[(size_t)q], std::move(pvFrames[(size_t)q])};
        else if (ballotCmp(pb[(size_t)q], pc[(size_t)q], ins->second.bnum, ins->second.bcoord) > 0)
          ins->second = Carried{pb[(size_t)q], pc[(size_t)q], std::move(pvFrames[(size_t)q])};
      }
    }
    if (vk[(size_t)i] == GPX_V_PREEMPTED) {
      purgeCarried(gi); /* the election is over: its carried-over values with it */
      /* hand the pre-active requests to the coordinator I deferred to (PISM:1042-1048) */
      for (int32_t j = 0; j < ec[(size_t)i]; j++) {
        auto pa = preactive_.find({gi, eh[(size_t)j * (size_t)n + (size_t)i]});
        if (pa == preactive_.end()) continue;
        if (rc[(size_t)i] != myID_) messenger_->send(rc[(size_t)i], std::move(pa->second));
        preactive_.erase(pa);
      }
      stats_.elections_lost++;
    } else if (vk[(size_t)i] == GPX_V_ELECTED) {
      /* spawnCommandersForProposals: one ACCEPT per entry, all in my new ballot (rb, me) */
      for (int32_t j = 0; j < ec[(size_t)i]; j++) {
        const size_t q = (size_t)j * (size_t)n + (size_t)i;
        const bool stop = (ef[q] & GPX_PV_STOP) != 0;
        Frame req;
        int64_t id = eh[q];
        if (ek[q] == GPX_E_CARRY) {
          auto cf = carried_.find({gi, es[q]}); /* by SLOT */
          if (cf == carried_.end()) continue;
          req.assign(cf->second.frame.begin(), cf->second.frame.end() - (long)kAcceptTail); /* the request part */
          stats_.carried_over++;
        } else if (ek[q] == GPX_E_PREACTIVE) {
          auto pa = preactive_.find({gi, id});
          if (pa == preactive_.end()) continue;
          req = std::move(pa->second);
          preactive_.erase(pa);
        } else if (ek[q] == GPX_E_NOOP) {
          req = makeRequestFrame(name, 0, 0, "NO_OP", false, myID_); /* makeNoopPValue (PCS:878-893) */
          id = 0;
          stats_.noops++;
        } else { /* GPX_E_NEWSTOP: new RequestPacket(0, STOP, true) (PCS:512-516) */
          req = makeRequestFrame(name, 0, 0, "STOP", true, myID_);
          id = 0;
        }
        issueAccept(out, gi, req, id, stop, es[q], rb[(size_t)i], myID_, em[(size_t)i]);
      }
      purgeCarried(gi);
      stats_.elections_won++;
    }
  }
  return true;
}

/* PISM.syncLongDecisionGaps -> requestMissingDecisions (PISM:1550-1570, 2284-2330): which of these
 * groups have a long gap, and which slots are missing */
bool PaxosManager::syncGaps(const std::vector<int32_t>& gidx, const std::vector<int32_t>& bcoord) {
  const int32_t n = (int32_t)gidx.size();
  if (n == 0) return true;
  std::vector<int32_t> first((size_t)n), maxc((size_t)n);
  std::vector<uint64_t> missing((size_t)n);
  std::vector<uint8_t> sync((size_t)n), st((size_t)n);
  if (!check(gpx_gap_scan(engine_, n, gidx.data(), opt_.syncGapThreshold, GPX_SYNC_DEFAULT, 64, first.data(),
                          maxc.data(), missing.data(), sync.data(), st.data()),
             "gpx_gap_scan"))
    return false;
  stats_.engine_calls++;
  for (int32_t i = 0; i < n; i++) {
    if (st[(size_t)i] != GPX_S_OK || !sync[(size_t)i] || !missing[(size_t)i] || bcoord[(size_t)i] == myID_) continue;
    std::vector<int32_t> slots;
    for (int j = 0; j < 64; j++)
      if ((missing[(size_t)i] >> j) & 1ull) {
        const int32_t s = (int32_t)((uint32_t)first[(size_t)i] + (uint32_t)j);
        uint64_t& asked = syncAsked_[key(gidx[(size_t)i], s)]; /* canSync: not again right away */
        if (asked == 0 || pass_ - asked > 8) slots.push_back(s), asked = pass_;
      }
    if (slots.empty()) continue;
    Frame f;
    putHeader(f, kTypeSyncDecisions, 0, rowName_[(size_t)gidx[(size_t)i]]);
    put32(f, myID_);
    put32(f, (int32_t)slots.size());
    for (int32_t s : slots) put32(f, s);
    messenger_->send(bcoord[(size_t)i], std::move(f));
    stats_.sync_requests++;
  }
  return true;
}

/* PISM.handleSyncDecisionsPacket (PISM:2372-2440): answer with the decisions I still hold */
bool PaxosManager::handleSyncRequests(std::vector<Frame>& reqs) {
  for (auto& f : reqs) {
    const size_t idLen = f.size() > 12 ? f[12] : 0, p = 13 + idLen;
    if (f.size() < p + 8) continue;
    const std::string name((const char*)&f[13], idLen);
    auto lg = decided_.find(name); /* live or paused: the log is kept by name */
    if (lg == decided_.end()) continue;
    const int32_t sender = get32(&f[p]), cnt = get32(&f[p + 4]);
    for (int32_t j = 0; j < cnt && p + 8 + 4 * (size_t)j + 4 <= f.size(); j++) {
      const int32_t slot = get32(&f[p + 8 + 4 * (size_t)j]);
      auto d = lg->second.find(slot);
      if (d == lg->second.end()) continue;
      Frame out;
      putHeader(out, kTypeDecision, 0, name);
      put32(out, slot), put32(out, d->second.bnum), put32(out, d->second.bcoord);
      put32(out, (int32_t)((uint32_t)slot - 1u)); /* a median that is safe for anyone who lacks this slot */
      put32(out, (int32_t)d->second.frame.size());
      out.insert(out.end(), d->second.frame.begin(), d->second.frame.end());
      messenger_->send(sender, std::move(out));
      stats_.sync_decisions_sent++;
    }
  }
  return true;
}

/* a full DECISION (value included): handleCommittedRequest (PISM:1432-1478) */
bool PaxosManager::handleDecisions(std::vector<Frame>& decisions) {
  std::vector<int32_t> g, bn, bc, sl, md;
  std::vector<uint8_t> kind;
  for (auto& f : decisions) {
    const size_t idLen = f.size() > 12 ? f[12] : 0, p = 13 + idLen;
    if (f.size() < p + 20) continue;
    auto it = pinstances_.find(std::string((const char*)&f[13], idLen));
    if (it == pinstances_.end()) {
      if (paused_.count(std::string((const char*)&f[13], idLen))) retry_.push_back(std::move(f));
      continue;
    }
    const int32_t len = get32(&f[p + 16]);
    if (len < 0 || f.size() < p + 20 + (size_t)len) continue;
    const int32_t gi = it->second.gidx, slot = get32(&f[p]);
    Frame val(f.begin() + (long)(p + 20), f.begin() + (long)(p + 20 + (size_t)len));
    const bool stop = frameIsStop(val);
    auto ins = accepted_.insert_or_assign(key(gi, slot), StoredAccept{get32(&f[p + 4]), get32(&f[p + 8]), std::move(val)});
    if (ins.second) liveAccepts_[(size_t)gi]++;
    g.push_back(gi), sl.push_back(slot), bn.push_back(get32(&f[p + 4])), bc.push_back(get32(&f[p + 8]));
    md.push_back(get32(&f[p + 12]));
    kind.push_back((uint8_t)(GPX_C_HASVALUE | (stop ? GPX_C_STOP : 0)));
    lastActive_[(size_t)gi] = pass_;
  }
  const int32_t n = (int32_t)g.size();
  if (n == 0) return true;
  std::vector<uint8_t> st((size_t)n);
  std::vector<int32_t> xg((size_t)n), xf((size_t)n), xc((size_t)n);
  int32_t nRuns = 0;
  if (!check(gpx_commit_batch(engine_, n, g.data(), bn.data(), bc.data(), sl.data(), md.data(), kind.data(), st.data(),
                              xg.data(), xf.data(), xc.data(), &nRuns),
             "gpx_commit_batch"))
    return false;
  stats_.engine_calls++;
  stats_.sync_decisions_applied += (uint64_t)n;
  executeRuns(nRuns, xg.data(), xf.data(), xc.data());
  /* a decision for a slot already executed here left its value behind: drop it again */
  for (int32_t i = 0; i < n; i++) {
    auto a = accepted_.find(key(g[(size_t)i], sl[(size_t)i]));
    auto lg = decided_.find(rowName_[(size_t)g[(size_t)i]]);
    if (a != accepted_.end() && lg != decided_.end() && lg->second.count(sl[(size_t)i])) {
      accepted_.erase(a);
      liveAccepts_[(size_t)g[(size_t)i]]--;
    }
  }
  return true;
}

size_t PaxosManager::processRun() {
  if (!engine_) return 0;
  /* The longest run of queued frames of one kind - PREPAREs, PREPARE replies, or the four byteified
   * types - bounded by the engine's batch capacity: a whole run goes through the engine as one
   * batch, and runs are taken in arrival order (a PREPARE that arrived after an ACCEPT is handled
   * after it, as the reference's per-packet handling would). */
  auto kindOf = [](const Frame& f) {
    const int32_t t = f.size() >= 8 ? get32(&f[4]) : -1;
    return t == kTypePrepare ? 1 : t == kTypePrepareReply ? 2 : t == kTypeSyncDecisions ? 3 : t == kTypeDecision ? 4 : 0;
  };
  std::vector<Frame> frames;
  const size_t maxFrames = (size_t)std::max(1, opt_.maxBatch / 4);
  const int kind = inbox_.empty() ? 0 : kindOf(inbox_.front());
  while (!inbox_.empty() && frames.size() < maxFrames && kindOf(inbox_.front()) == kind) {
    frames.push_back(std::move(inbox_.front()));
    inbox_.pop_front();
  }
  size_t fromDeferred = 0; /* retries of requests the proposal window had no room for */
  redeferred_ = 0;
  const size_t firstDeferred = frames.size(); /* frames[firstDeferred .. firstDeferred + fromDeferred) are the retries */
  if (kind == 0 && inbox_.empty()) {
    while (!deferred_.empty() && frames.size() < maxFrames) {
      frames.push_back(std::move(deferred_.front()));
      deferred_.pop_front();
      fromDeferred++;
    }
    while (!requests_.empty() && frames.size() < maxFrames) {
      frames.push_back(std::move(requests_.front()));
      requests_.pop_front();
    }
  }
  if (frames.empty()) return 0;
  pass_++;
  const size_t consumed = frames.size();
  std::vector<OutAccept> outAccepts; /* my own ACCEPTs of this pass: short-circuited, never decoded */
  if (kind == 1) { /* not among the byteified types: handled from the host layer's own layout */
    handlePrepares(frames);
    return consumed;
  }
  if (kind == 2) {
    if (!handlePrepareReplies(frames, outAccepts)) return consumed;
    frames.clear();
  }
  if (kind == 3) {
    handleSyncRequests(frames);
    return consumed;
  }
  if (kind == 4) {
    handleDecisions(frames);
    return consumed;
  }
  const int32_t nF = (int32_t)frames.size();
  if (nF == 0 && outAccepts.empty()) return consumed;
  std::vector<int64_t> off((size_t)nF + 1, 0);
  for (int32_t i = 0; i < nF; i++) off[(size_t)i + 1] = off[(size_t)i] + (int64_t)frames[(size_t)i].size();
  std::vector<uint8_t> buf((size_t)off[(size_t)nF]);
  for (int32_t i = 0; i < nF; i++)
    std::memcpy(buf.data() + off[(size_t)i], frames[(size_t)i].data(), frames[(size_t)i].size());

  /* ---- bytes -> columns (PaxosPacketDemultiplexerFast + the ByteBuffer constructors + getInstance) */
  const int32_t capV = (int32_t)std::min<int64_t>(opt_.maxBatch, off[(size_t)nF] / 12 + nF);
  const int32_t capC = (int32_t)std::min<int64_t>(opt_.maxBatch, off[(size_t)nF] / 4 + nF);
  struct Cols {
    std::vector<int32_t> gidx, bnum, bcoord, slot, x, y, frame;
    std::vector<uint8_t> f;
    std::vector<int64_t> id;
    explicit Cols(int32_t cap)
        : gidx((size_t)cap), bnum((size_t)cap), bcoord((size_t)cap), slot((size_t)cap), x((size_t)cap),
          y((size_t)cap), frame((size_t)cap), f((size_t)cap), id((size_t)cap) {}
  };
  Cols v(std::max(capV, 1)), c(std::max(capC, 1)), a(std::max(nF, 1)), q(std::max(nF, 1));
  gpx_wire_votes V{capV, v.gidx.data(), v.bnum.data(), v.bcoord.data(), v.slot.data(), v.x.data(), v.y.data(),
                   v.frame.data()};
  gpx_wire_commits C{capC, c.gidx.data(), c.bnum.data(), c.bcoord.data(), c.slot.data(), c.x.data(), c.f.data(),
                     c.frame.data()};
  gpx_wire_accepts A{nF, a.gidx.data(), a.bnum.data(), a.bcoord.data(), a.slot.data(), a.x.data(), a.f.data(),
                     a.y.data(), a.id.data(), a.frame.data()};
  gpx_wire_requests Q{nF, q.gidx.data(), q.f.data(), q.id.data(), q.frame.data()};
  std::vector<uint8_t> fst((size_t)nF);
  std::vector<int32_t> fg((size_t)nF), ft((size_t)nF);
  gpx_wire_counts cnt;
  std::memset(&cnt, 0, sizeof(cnt));
  if (nF > 0) {
    if (!check(gpx_wire_decode(engine_, nF, buf.data(), off.data(), fst.data(), fg.data(), ft.data(), &V, &C, &A,
                               &Q, &cnt),
               "gpx_wire_decode"))
      return 0;
    stats_.engine_calls++;
  }
  for (int32_t i = nF - 1; i >= 0; i--) { /* back to front: push_front keeps their order */
    if (fst[(size_t)i] == GPX_W_CAPACITY) {
      inbox_.push_front(std::move(frames[(size_t)i])); /* did not fit the columns: next pass */
      frames[(size_t)i].clear();
    } else if (fst[(size_t)i] == GPX_W_NOGROUP && frames[(size_t)i].size() > 13 &&
               paused_.count(std::string((const char*)&frames[(size_t)i][13],
                                         std::min<size_t>(frames[(size_t)i][12], frames[(size_t)i].size() - 13)))) {
      retry_.push_front(std::move(frames[(size_t)i])); /* its group is paused: unpause, then again */
      frames[(size_t)i].clear();
    } else if (fst[(size_t)i] != GPX_W_OK) {
      stats_.dropped_frames++; /* the reference drops such a packet */
    } else {
      lastActive_[(size_t)fg[(size_t)i]] = pass_;
    }
  }
  const int32_t nV = std::min(cnt.n_votes, capV), nC = std::min(cnt.n_commits, capC);
  const int32_t nA = std::min(cnt.n_accepts, nF), nQ = std::min(cnt.n_requests, nF);
  std::vector<int32_t> xg, xf, xc;
  int32_t nRuns = 0;
  auto runsFor = [&](int32_t n) {
    xg.assign((size_t)std::max(n, 1), 0);
    xf.assign((size_t)std::max(n, 1), 0);
    xc.assign((size_t)std::max(n, 1), 0);
    nRuns = 0;
  };

  /* ---- REQUEST -> handleProposal (PISM:818-888): ACCEPT multicast, forward to the coordinator, or -
   * while my election is running - a pre-active proposal */
  if (nQ > 0) {
    /* RequestBatcher: the requests of one group queued together ride in ONE proposal */
    std::vector<int32_t> pg(q.gidx.begin(), q.gidx.begin() + nQ);
    std::vector<uint8_t> pstop(q.f.begin(), q.f.begin() + nQ);
    std::vector<int64_t> pid(q.id.begin(), q.id.begin() + nQ);
    std::vector<Frame> latched; /* owns the batched frames */
    std::vector<Frame*> pframe((size_t)nQ);
    std::vector<uint8_t> pretry((size_t)nQ); /* the proposal is a retry out of deferred_ (a batch: its leader is) */
    for (int32_t i = 0; i < nQ; i++) {
      pframe[(size_t)i] = &frames[(size_t)q.frame[(size_t)i]];
      pretry[(size_t)i] = (size_t)q.frame[(size_t)i] - firstDeferred < fromDeferred;
    }
    int32_t nP = nQ;
    if (opt_.batchRequests && nQ > 1) {
      std::vector<int32_t> est((size_t)nQ), leader((size_t)nQ), bg((size_t)nQ), bl((size_t)nQ), bcnt((size_t)nQ),
          bbytes((size_t)nQ), bsize((size_t)nQ);
      std::vector<uint8_t> st((size_t)nQ), bstop((size_t)nQ);
      std::vector<int32_t> weight((size_t)nQ);
      for (int32_t i = 0; i < nQ; i++) {
        est[(size_t)i] = (int32_t)pframe[(size_t)i]->size();
        weight[(size_t)i] = batchSizeOf(*pframe[(size_t)i]) + 1; /* a forwarded request may be a batch already */
      }
      int32_t nB = 0;
      if (!check(gpx_request_batch(engine_, nQ, q.gidx.data(), est.data(), weight.data(), q.f.data(), opt_.maxBatchBytes,
                                   opt_.maxBatchSize, leader.data(), st.data(), bg.data(), bl.data(), bcnt.data(),
                                   bbytes.data(), bsize.data(), bstop.data(), &nB),
                 "gpx_request_batch"))
        return 0;
      stats_.engine_calls++;
      std::vector<std::vector<const Frame*>> rest((size_t)nQ);
      for (int32_t i = 0; i < nQ; i++)
        if (leader[(size_t)i] >= 0 && leader[(size_t)i] != i) rest[(size_t)leader[(size_t)i]].push_back(pframe[(size_t)i]);
      latched.reserve((size_t)nB);
      std::vector<int32_t> ng;
      std::vector<uint8_t> ns;
      std::vector<int64_t> ni;
      std::vector<Frame*> nf;
      std::vector<uint8_t> nr;
      for (int32_t b = 0; b < nB; b++) {
        const int32_t l = bl[(size_t)b];
        Frame* fr = pframe[(size_t)l];
        if (bcnt[(size_t)b] > 1) {
          latched.push_back(latchToBatch(*fr, rest[(size_t)l])); /* first.latchToBatch(...) */
          fr = &latched.back();
          stats_.batched_requests += (uint64_t)(bsize[(size_t)b] - weight[(size_t)l]);
        }
        ng.push_back(bg[(size_t)b]), ns.push_back(bstop[(size_t)b]), ni.push_back(q.id[(size_t)l]), nf.push_back(fr);
        nr.push_back(pretry[(size_t)l]);
      }
      for (int32_t i = 0; i < nQ; i++) /* requests for groups this node does not have: dropped */
        if (leader[(size_t)i] < 0) stats_.refused++;
      pg.swap(ng), pstop.swap(ns), pid.swap(ni), pframe.swap(nf), pretry.swap(nr);
      nP = nB;
    }
    std::vector<int32_t> slot((size_t)nP), bn((size_t)nP), bc((size_t)nP), med((size_t)nP);
    std::vector<uint8_t> st((size_t)nP);
    if (nP > 0 && !check(gpx_propose_batch_h(engine_, nP, pg.data(), pstop.data(), pid.data(), slot.data(),
                                             bn.data(), bc.data(), med.data(), st.data()),
                         "gpx_propose_batch_h"))
      return 0;
    stats_.engine_calls++;
    for (int32_t i = 0; i < nP; i++) {
      Frame& rf = *pframe[(size_t)i];
      if (st[(size_t)i] == GPX_S_OK) {
        issueAccept(outAccepts, pg[(size_t)i], rf, pid[(size_t)i], pstop[(size_t)i] != 0, slot[(size_t)i],
                    bn[(size_t)i], bc[(size_t)i], med[(size_t)i]);
        stats_.proposed++;
      } else if (st[(size_t)i] == GPX_S_PREACTIVE) {
        preactive_[{pg[(size_t)i], pid[(size_t)i]}] = std::move(rf); /* comes back by handle */
        stats_.preactive++;
      } else if (st[(size_t)i] == GPX_S_FORWARD && bc[(size_t)i] != myID_) {
        messenger_->send(bc[(size_t)i], Frame(rf)); /* unicast to paxosState.getBallotCoord() */
        stats_.forwarded++;
      } else if (st[(size_t)i] == GPX_S_WINDOW) {
        /* the reference's myProposals is unbounded; the engine's window is not: the request waits
         * for decisions to free a slot and is proposed again on a later pass - never dropped */
        deferred_.push_back(std::move(rf));
        stats_.deferred++;
        if (pretry[(size_t)i]) redeferred_++; /* only a retry refused AGAIN is "no progress"; a fresh request deferred
                                                for the first time left the queue it came from */
      } else {
        stats_.refused++; /* STOPPED / NOGROUP / proposal after a stop: the reference drops these too */
      }
    }
  }

  /* ---- ACCEPT -> handleAccept (PISM:1080-1166): store, reply (coalesced per ballot), maybe execute */
  const int32_t nLA = (int32_t)outAccepts.size();
  const int32_t nAll = nLA + nA;
  if (nAll > 0) {
    std::vector<int32_t> g((size_t)nAll), bn((size_t)nAll), bc((size_t)nAll), sl((size_t)nAll), md((size_t)nAll),
        snd((size_t)nAll);
    std::vector<uint8_t> fl((size_t)nAll);
    std::vector<int64_t> id((size_t)nAll);
    std::vector<const Frame*> src((size_t)nAll);
    for (int32_t i = 0; i < nLA; i++) { /* loopback first */
      const OutAccept& oa = outAccepts[(size_t)i];
      g[(size_t)i] = oa.gidx, bn[(size_t)i] = oa.bnum, bc[(size_t)i] = oa.bcoord, sl[(size_t)i] = oa.slot;
      md[(size_t)i] = oa.median, fl[(size_t)i] = oa.flags, snd[(size_t)i] = myID_, id[(size_t)i] = oa.requestID;
      src[(size_t)i] = &oa.frame;
    }
    for (int32_t i = 0; i < nA; i++) {
      const size_t o = (size_t)(nLA + i), s = (size_t)i;
      g[o] = a.gidx[s], bn[o] = a.bnum[s], bc[o] = a.bcoord[s], sl[o] = a.slot[s], md[o] = a.x[s], fl[o] = a.f[s];
      snd[o] = a.y[s], id[o] = a.id[s], src[o] = &frames[(size_t)a.frame[s]];
    }
    std::vector<int32_t> rb((size_t)nAll), rc((size_t)nAll), rm((size_t)nAll);
    std::vector<uint8_t> rf((size_t)nAll), st((size_t)nAll), unb((size_t)nAll);
    runsFor(nAll);
    if (!check(gpx_accept_batch(engine_, nAll, g.data(), bn.data(), bc.data(), sl.data(), md.data(), fl.data(),
                                rb.data(), rc.data(), rm.data(), rf.data(), st.data(), xg.data(), xf.data(),
                                xc.data(), &nRuns),
               "gpx_accept_batch"))
      return 0;
    stats_.engine_calls++;
    stats_.accepts += (uint64_t)nAll;
    for (int32_t i = 0; i < nAll; i++)
      if (st[(size_t)i] == GPX_S_OK && (rf[(size_t)i] & GPX_R_STORED)) {
        auto ins = accepted_.insert_or_assign(key(g[(size_t)i], sl[(size_t)i]),
                                              StoredAccept{bn[(size_t)i], bc[(size_t)i], *src[(size_t)i]});
        if (ins.second) liveAccepts_[(size_t)g[(size_t)i]]++;
      }
    executeRuns(nRuns, xg.data(), xf.data(), xc.data()); /* meta-commits an ACCEPT released (:1158-1161) */
    const int64_t capB = (int64_t)nAll * 192 + 1024;
    std::vector<uint8_t> out((size_t)capB);
    std::vector<int64_t> fo((size_t)nAll);
    std::vector<int32_t> flen((size_t)nAll), fgi((size_t)nAll), fd((size_t)nAll);
    int32_t nf = 0;
    int64_t nb = 0;
    if (!check(gpx_wire_pack_accept_replies(engine_, nAll, g.data(), sl.data(), snd.data(), id.data(), rb.data(),
                                            rc.data(), rm.data(), st.data(), unb.data(), out.data(), capB,
                                            fo.data(), flen.data(), fgi.data(), fd.data(), &nf, &nb),
               "gpx_wire_pack_accept_replies"))
      return 0;
    stats_.engine_calls++;
    /* the ACCEPTs that must be logged before their replies leave (toLog, PISM:1146-1149): one log
     * batch; the replies of this call wait for it while the engine goes on with the next batches */
    uint64_t ticket = 0;
    if (opt_.logger) {
      std::vector<const Frame*> tolog;
      for (int32_t i = 0; i < nAll; i++)
        if (st[(size_t)i] == GPX_S_OK && (rf[(size_t)i] & GPX_R_TOLOG)) tolog.push_back(src[(size_t)i]);
      if (!tolog.empty()) {
        ticket = opt_.logger->logBatch(tolog);
        stats_.logged_accepts += (uint64_t)tolog.size();
        stats_.log_batches++;
      }
    }
    for (int32_t f = 0; f < nf; f++) {
      Frame fr(out.begin() + fo[(size_t)f], out.begin() + fo[(size_t)f] + flen[(size_t)f]);
      if (ticket) {
        held_.push_back(Held{ticket, fd[(size_t)f], std::move(fr)});
        stats_.held_replies++;
      } else if (fd[(size_t)f] == myID_) {
        inbox_.push_back(std::move(fr));
      } else {
        messenger_->send(fd[(size_t)f], std::move(fr));
      }
    }
    for (int32_t i = 0; i < nAll; i++)
      if (unb[(size_t)i]) { /* e.g. a reply in a higher ballot than the sender's: back to the sender */
        Frame fr = makeSingleAcceptReply(rowName_[(size_t)g[(size_t)i]], 0, myID_, rb[(size_t)i], rc[(size_t)i],
                                         sl[(size_t)i], rm[(size_t)i], id[(size_t)i]);
        if (ticket) { /* same rule as the packed frames: nothing of this call leaves before its log batch */
          held_.push_back(Held{ticket, snd[(size_t)i], std::move(fr)});
          stats_.held_replies++;
        } else if (snd[(size_t)i] == myID_)
          inbox_.push_back(std::move(fr));
        else
          messenger_->send(snd[(size_t)i], std::move(fr));
      }
  }

  /* ---- BATCHED_ACCEPT_REPLY -> handleBatchedAcceptReply (PISM:1370-1419): decisions */
  if (nV > 0) {
    std::vector<int32_t> dg((size_t)nV), ds((size_t)nV), db((size_t)nV), dc((size_t)nV), dm((size_t)nV);
    std::vector<uint8_t> dk((size_t)nV), st((size_t)nV);
    int32_t nOut = 0;
    if (!check(gpx_accept_reply_batch(engine_, nV, v.gidx.data(), v.bnum.data(), v.bcoord.data(), v.slot.data(),
                                      v.x.data(), v.y.data(), dg.data(), ds.data(), db.data(), dc.data(), dm.data(),
                                      dk.data(), &nOut, st.data()),
               "gpx_accept_reply_batch"))
      return 0;
    stats_.engine_calls++;
    stats_.votes += (uint64_t)nV;
    if (nOut > 0) {
      /* BATCHED_COMMIT to the other members, coalesced per (group, ballot) */
      const int64_t capB = (int64_t)nOut * 128 + 1024;
      std::vector<uint8_t> out((size_t)capB);
      std::vector<int64_t> fo((size_t)nOut);
      std::vector<int32_t> flen((size_t)nOut), fgi((size_t)nOut);
      int32_t nf = 0;
      int64_t nb = 0;
      if (!check(gpx_wire_pack_commits(engine_, nOut, dg.data(), ds.data(), db.data(), dc.data(), dm.data(),
                                       dk.data(), out.data(), capB, fo.data(), flen.data(), fgi.data(), &nf, &nb),
                 "gpx_wire_pack_commits"))
        return 0;
      stats_.engine_calls++;
      for (int32_t f = 0; f < nf; f++) {
        const Instance& in = pinstances_.at(rowName_[(size_t)fgi[(size_t)f]]);
        Frame fr(out.begin() + fo[(size_t)f], out.begin() + fo[(size_t)f] + flen[(size_t)f]);
        sendToMembers(in, fr, false);
      }
      /* my own copy of a decision is handled as a full DECISION (local short circuit) */
      std::vector<int32_t> lg, lb, lc, ls, lm;
      std::vector<uint8_t> lk;
      for (int32_t i = 0; i < nOut; i++) {
        if (dk[(size_t)i] != GPX_D_DECISION) continue; /* PREEMPTED: dropped (FORWARD_PREEMPTED_REQUESTS off) */
        uint8_t kind = GPX_C_HASVALUE;
        auto sa = accepted_.find(key(dg[(size_t)i], ds[(size_t)i]));
        if (sa != accepted_.end() && frameIsStop(sa->second.frame)) kind |= GPX_C_STOP;
        lg.push_back(dg[(size_t)i]), lb.push_back(db[(size_t)i]), lc.push_back(dc[(size_t)i]);
        ls.push_back(ds[(size_t)i]), lm.push_back(dm[(size_t)i]), lk.push_back(kind);
        stats_.decisions++;
      }
      const int32_t nL = (int32_t)lg.size();
      if (nL > 0) {
        std::vector<uint8_t> cst((size_t)nL);
        runsFor(nL);
        if (!check(gpx_commit_batch(engine_, nL, lg.data(), lb.data(), lc.data(), ls.data(), lm.data(), lk.data(),
                                    cst.data(), xg.data(), xf.data(), xc.data(), &nRuns),
                   "gpx_commit_batch"))
          return 0;
        stats_.engine_calls++;
        executeRuns(nRuns, xg.data(), xf.data(), xc.data());
      }
    }
  }

  /* ---- BATCHED_COMMIT -> handleBatchedCommit (PISM:1480-1528) -> in-order execution */
  if (nC > 0) {
    std::vector<uint8_t> cst((size_t)nC);
    runsFor(nC);
    if (!check(gpx_commit_batch(engine_, nC, c.gidx.data(), c.bnum.data(), c.bcoord.data(), c.slot.data(),
                                c.x.data(), c.f.data(), cst.data(), xg.data(), xf.data(), xc.data(), &nRuns),
               "gpx_commit_batch"))
      return 0;
    stats_.engine_calls++;
    stats_.commits += (uint64_t)nC;
    executeRuns(nRuns, xg.data(), xf.data(), xc.data());
    /* did a commit of mine get lost on the way?  the groups these commits name, once each */
    std::vector<int32_t> sg, sb;
    for (int32_t i = 0; i < nC; i++)
      if (i == 0 || c.gidx[(size_t)i] != c.gidx[(size_t)i - 1]) sg.push_back(c.gidx[(size_t)i]), sb.push_back(c.bcoord[(size_t)i]);
    syncGaps(sg, sb);
  }
  /* a retry that the window refused again is not progress: a caller that loops while there is work
   * (and fires its retransmission timers only when there is none) would spin on it for ever when the
   * ACCEPTs or replies that would free the window were lost */
  return consumed - std::min(consumed, std::min(fromDeferred, redeferred_));
}

size_t PaxosManager::releaseHeld() {
  if (held_.empty()) return 0;
  const uint64_t d = opt_.logger->durable();
  size_t n = 0;
  while (!held_.empty() && held_.front().ticket <= d) { /* log-batch order */
    Held h = std::move(held_.front());
    held_.pop_front();
    if (h.dest == myID_)
      inbox_.push_back(std::move(h.frame));
    else
      messenger_->send(h.dest, std::move(h.frame));
    n++;
  }
  return n;
}

size_t PaxosManager::process() {
  const size_t released = releaseHeld();
  size_t consumed = processRun() + released;
  if (consumed == 0 && !held_.empty()) { /* only the log write is outstanding: wait for it a little */
    std::this_thread::sleep_for(std::chrono::microseconds(50));
    consumed = 1;
  }
  /* frames that found their group paused: bring the groups back (getInstance -> unpause), pausing
   * idle ones if the table is full, and queue the frames again.  A frame whose group cannot come
   * back yet (every row holds a group with work in flight) waits for a later pass. */
  if (!retry_.empty()) {
    if (consumed == 0) pass_++; /* nothing else to do: every group without work in flight is idle */
    std::deque<Frame> again, wait;
    again.swap(retry_);
    while (!again.empty()) {
      Frame f = std::move(again.back());
      again.pop_back();
      const std::string name((const char*)&f[13], std::min<size_t>(f[12], f.size() - 13));
      if (paused_.count(name) && !unpause(name)) {
        wait.push_front(std::move(f));
        continue;
      }
      inbox_.push_front(std::move(f));
      consumed++;
    }
    retry_.swap(wait);
  }
  return consumed;
}

}  // namespace gpx
