/*
 * gpx_engine.hip — host side of libgpx_hip.so: device-resident group state, batch
 * sequencing on a HIP stream and the extern "C" entry points of include/gpx.h.
 *
 * Build (gfx950 only, no other target, no compatibility paths):
 *   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o libgpx_hip.so gpx_engine.hip
 */
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <dirent.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <cerrno>
#include <chrono>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <algorithm>
#include <initializer_list>

#include "gpx_kernels.hip.h"
#include "gpx_ar16.hip.h"
#include "gpx_direct.hip.h"
#include "gpx_one.hip.h"
#include "gpx_runs.hip.h"
#include "gpx_small.hip.h"
#include "gpx_route.hip.h"
#include "gpx_wire.hip.h"
#include "gpx_elect.hip.h"

#define GPX_STAGE_N 32768 /* host-pointer calls up to this many records cross PCIe as one block each way */
#define GPX_STAGE_BYTES ((size_t)GPX_STAGE_N * 48 + 4096)
/* e->fs[].chunk_cnt: one count per 1024-record chunk of the largest batch, then DirectStage's st_total and mark.
 * Their words do not move with the batch size: a word that once held a chunk count can never be read as a mark. */
#define GPX_CHUNK_CNT_TOTAL(N) ((size_t)(N) / GPX_DCHUNK + 4)
#define GPX_CHUNK_CNT_MARK(N) ((size_t)(N) / GPX_DCHUNK + 5)
#define GPX_CHUNK_CNT_WORDS(N) ((size_t)(N) / GPX_DCHUNK + 8)
namespace {

thread_local char g_err[256] = "";

#define HIPCHK(expr)                                                                      \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      snprintf(g_err, sizeof(g_err), "%s:%d %s -> %s", __FILE__, __LINE__, #expr,         \
               hipGetErrorString(_e));                                                    \
      return GPX_EDEVICE;                                                                 \
    }                                                                                     \
  } while (0)
/* unchecked on purpose: ordering / teardown calls inside helpers that cannot return an error; a
 * failure stays pending in the runtime and surfaces in the closing HIPCHK(hipGetLastError()) of the
 * entry point that called the helper */
#define HIPQ(call) ((void)(call))

struct PendingEvent {
  const char* name;
  hipEvent_t start, stop;
};

/* The streams that the live engines of each device (in this process) launch on: the exchange kernels (grid_exchange,
 * k_ar_runs<.., SMALL>) are only launched with grids that stay resident when EVERY such stream has one of them running
 * at the same moment.  Engines that share a stream (gpx_engine_set_stream with one handle: their launches are
 * serialised by the stream) count once. */
std::mutex g_live_mu;
std::map<int, std::map<hipStream_t, int>> g_live_streams;
/* ... and of the OTHER processes on the device (three processes that each believe they are alone size their
 * grids at two workgroups per CU each, none is resident whole, all give up).  Every process keeps a file
 * /dev/shm/gpx_engines/<PCI bus id>.<pid> holding the number of streams its engines launch on; the others' files are
 * summed (files of dead pids are removed), at most every 50 ms.  Processes that do not share /dev/shm (separate
 * containers) cannot see each other: GPX_DEVICE_SHARERS says it for them (include/gpx.h). */
struct ProcRegistry {
  std::string dir = "/dev/shm/gpx_engines", mine;
  int foreign = 0;
  std::chrono::steady_clock::time_point sampled{};
  bool ok = false, tried = false;
};
std::map<int, ProcRegistry> g_proc_reg; /* by device; under g_live_mu */
ProcRegistry& proc_registry(int device) {
  ProcRegistry& R = g_proc_reg[device];
  if (!R.tried) {
    R.tried = true;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, device) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(bus, sizeof(bus), "dev%d", device);
    }
    if (const char* d = getenv("GPX_REGISTRY_DIR")) R.dir = d; /* tests */
    if (mkdir(R.dir.c_str(), 0777) == 0 || errno == EEXIST) {
      (void)chmod(R.dir.c_str(), 0777);
      R.mine = R.dir + "/" + bus + "." + std::to_string((long)getpid());
      R.ok = true;
    }
  }
  return R;
}
void proc_registry_publish(int device, int nstreams) { /* g_live_mu held */
  ProcRegistry& R = proc_registry(device);
  if (!R.ok) return;
  if (nstreams <= 0) {
    (void)unlink(R.mine.c_str());
    return;
  }
  if (FILE* fp = fopen((R.mine + ".tmp").c_str(), "w")) {
    fprintf(fp, "%d\n", nstreams);
    fclose(fp);
    (void)rename((R.mine + ".tmp").c_str(), R.mine.c_str());
  }
}
int proc_registry_foreign(int device) { /* g_live_mu held */
  ProcRegistry& R = proc_registry(device);
  if (!R.ok) return 0;
  const auto now = std::chrono::steady_clock::now();
  if (R.sampled.time_since_epoch().count() && now - R.sampled < std::chrono::milliseconds(50)) return R.foreign;
  R.sampled = now;
  int sum = 0;
  const std::string base = R.mine.substr(R.dir.size() + 1);
  const std::string prefix = base.substr(0, base.rfind('.') + 1);
  if (DIR* d = opendir(R.dir.c_str())) {
    while (struct dirent* de = readdir(d)) {
      const std::string name = de->d_name;
      if (name.compare(0, prefix.size(), prefix) != 0 || name == base || name.size() > 4 && name.substr(name.size() - 4) == ".tmp") continue;
      const long pid = atol(name.c_str() + prefix.size());
      const std::string path = R.dir + "/" + name;
      if (pid <= 0 || (kill((pid_t)pid, 0) != 0 && errno == ESRCH)) { /* its process is gone (it did not get to unlink) */
        (void)unlink(path.c_str());
        continue;
      }
      int k = 0;
      if (FILE* fp = fopen(path.c_str(), "r")) {
        if (fscanf(fp, "%d", &k) != 1) k = 1;
        fclose(fp);
      }
      sum += std::max(k, 1);
    }
    closedir(d);
  }
  R.foreign = sum;
  return sum;
}

void live_add(int device, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  g_live_streams[device][s]++;
  proc_registry_publish(device, (int)g_live_streams[device].size());
}
void live_drop(int device, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  auto& m = g_live_streams[device];
  auto it = m.find(s);
  if (it != m.end() && --it->second <= 0) m.erase(it);
  proc_registry_publish(device, (int)m.size());
  if (m.empty()) g_proc_reg.erase(device); /* (the next engine of this process looks the directory up afresh) */
}

}  // namespace

struct gpx_engine {
  gpx_config cfg;
  int device = 0;
  DevState S{};
  DevScratch X{};
  hipStream_t own_stream = nullptr;  /* when the caller gave none */
  hipStream_t user_stream = nullptr; /* gpx_engine_set_stream */
  /* ONE stream carries every launch of a call (round 2 measured the two-stream overlap of call N + 1's
   * partition with call N's per-bucket kernels: slower, every kernel of the pipeline fills the chip - the
   * mode is gone, scripts/experiments/r03_removed_alternatives.patch); sF / sB / stream all name it */
  hipStream_t sF = nullptr, sB = nullptr, stream = nullptr;
  /* scratch of the partition front end and of the direct / runs paths */
  struct FrontSet {
    int32_t *bucket_tot = nullptr, *tile_rel = nullptr, *bucket_off = nullptr;
    Rec* rec = nullptr;
    uint32_t* unsorted = nullptr;
    int32_t* chunk_cnt = nullptr; /* direct / runs paths: outputs per 1024-record chunk */
  } fs[1];
  uint64_t call_seq = 0;
  std::vector<void*> allocs;
  /* device staging for the host-pointer entry points */
  int32_t* st_i32[12] = {};
  uint8_t* st_u8[4] = {};
  int32_t* st_count = nullptr;
  int64_t* st_handle = nullptr; /* gpx_propose_batch_h, allocated on first use */
  /* arena of the host-pointer calls' temporary device buffers (TmpDev, gpx_wire_host.inc) */
  char* arena = nullptr;
  size_t arena_cap = 0, arena_used = 0, arena_demand = 0, arena_want = 0;
  bool arena_busy = false;
  /* profiling */
  bool profiling = false;
  std::vector<PendingEvent> pending;
  std::map<std::string, std::pair<uint64_t, double>> prof;
  size_t bucket_lds = 0;      /* dynamic LDS of the current per-bucket launch */
  int32_t lds_recs_max = 0;   /* staging capacity the engine was sized for (kmax records per group) */
  int32_t lds_recs_hw = 0;    /* ... and the most a workgroup can stage (oversized calls) */
  int bucket_threads = 256;
  int32_t ordered_mask = 0; /* gpx_engine_set_ordered_batches */
  int32_t env_mask = 0;     /* GPX_TRY_RUNS=1 (test switch): GPX_TRY_REPLY_RUNS on every engine, whatever the caller sets */
  /* host-pointer calls of small batches: ONE staged H2D and ONE D2H per call (pinned host block <->
   * device block) instead of a copy per column */
  char *hs_in = nullptr, *hs_out = nullptr; /* pinned host */
  char *ds_in = nullptr, *ds_out = nullptr; /* device */
  int32_t* route_cnt = nullptr; /* gpx_route_batch_dev: [tiles][shards], allocated on first use */
  /* tickets of the single-launch kernels for small ordered batches (k_ac_small) */
  unsigned long long* small_tickets = nullptr;
  uint32_t small_epoch = 0;
  uint32_t* small_draw = nullptr; /* chunk tickets drawn so far by every k_ac_small launch (never reset) */
  uint32_t small_drawn = 0;       /* ... as the host counts them: a launch's chunks are draw - small_drawn */
  /* [max_batch] == a call's epoch: record i of that call holds a parked output (direct and sorted-runs
   * paths: no per-record clearing store, no clearing pass) */
  uint32_t* rec_tag = nullptr;
  /* ordered batches in one launch (gpx_one.hip.h): order words, arrival counters, the call counter of their epochs */
  unsigned long long* one_words = nullptr; /* the verdict word of k_one_check (gpx_one.hip.h) */
  uint32_t one_epoch = 0;
  uint32_t* runs_arrive = nullptr; /* k_runs_check's arrival counters (first use) */
  /* accept-reply calls of at most this many votes, in any order, take ONE launch of one workgroup (gpx_small.hip.h;
   * GPX_SAR_MAX_N: tuning, 0 = every call takes the partition pipeline) */
  int32_t sar_max_n = GPX_SAR_MAX_N;
  /* GPX_LAZY_OUTPUTS: what gpx_compact_last_dev needs to finish the most recent call (kind 0: nothing pending) */
  struct LastCall {
    int kind = 0; /* 1 ACCEPT, 2 COMMIT (k_ac_one), 3 accept replies (k_ar_runs) */
    int32_t n = 0, nchunks = 0;
    DevScratch X{};
    const int32_t* gidx = nullptr;
    DirectStage D{};
    int32_t *x_gidx = nullptr, *x_first = nullptr, *x_count = nullptr, *count = nullptr;
    RunsStage rs{};
    RunsInfo* info = nullptr;
    uint64_t seq = 0; /* call_seq right after the call: gpx_compact_last_dev refuses once another batch call came in between */
    bool stale = false; /* a pending compaction was overtaken by another batch call (begin_front) */
  } last;
  int lazy_override = -1; /* 1: the host-pointer twins compact on demand themselves; 0: asynchronous calls need dense columns */
  /* one-launch calls (gpx_one.hip.h: judges that meet at arrival counters): the device's CUs and the processes sharing
   * it (GPX_DEVICE_SHARERS, default 1) decide which grids need no tickets; the host-mapped word a waiter that gave up
   * writes (DevScratch.xabort) */
  int cus = 0, sharers = 1;
  uint32_t xchg_test_skew = 0;          /* GPX_XCHG_TEST_SKEW (tests/test_many_engines_gpu.py): see xchg_ctl */
  bool sharers_set = false;             /* GPX_DEVICE_SHARERS given: it replaces the registry of other processes */
  std::map<const void*, int> occ;       /* hipOccupancyMaxActiveBlocksPerMultiprocessor of the exchange kernels */
  uint32_t xchg_timeout_ms = GPX_XCHG_TIMEOUT_MS; /* GPX_XCHG_TIMEOUT_MS: how long an exchange kernel's pollers wait */
  /* largest batch, in workgroups of 256 records, that takes the one-launch form: PROPOSE, ACCEPT / COMMIT, reply runs
   * (GPX_PERS_CHUNKS=a,b,c for tuning).  The exchange among the workgroups grows with their number, the check kernel of
   * the two-launch form does not: profiles/r06_one_launch_propose_sizes.txt */
  int pers_max_chunks[3] = {128, 192, 256};
  bool one_launch = true;            /* GPX_XCHG_SLOTS=0 (comparison builds, tests): the check kernel + the work kernel instead */
  uint32_t gx_arrive = 0; /* grid_exchange's arrival counters as this engine's launches have left them */
  bool registered_live = false;
  uint32_t* h_abort = nullptr;
  /* host blocks handed out by gpx_host_alloc (hipHostMalloc): freed by gpx_host_free or at destroy */
  std::vector<void*> host_blocks;
  /* accept replies as a few sorted runs (gpx_runs.hip.h): allocated on first use */
  RunsInfo* runs_info = nullptr; /* [2], used alternately */
  uint64_t runs_seq = 0;
  int32_t lds16_max = 0, lds16_hw = 0; /* LDS staging capacities (votes) of k_bucket_ar16 */
  /* accept-reply calls partition at most 4 M groups per pass (4096 buckets of 1024 groups): a bigger
   * table is covered by ar_passes passes over ascending group ranges, each skipping the other ranges'
   * votes; shift16 / nbk16 = the bucket geometry of those passes (== X's when one pass suffices) */
  int32_t ar_passes = 1, shift16 = 0, nbk16 = 0;
  bool ac16 = false;          /* ACCEPT / COMMIT partition path on 16-byte records too (single pass only) */
  int32_t* ar_chain = nullptr; /* [2] running output count between passes */
  /* the tiled front end of accept-reply calls (gpx_tiles.hip.h; GPX_AR_TILES=0 keeps the partition front end for every
   * call): allocated on first use */
  bool ar_tiles = false;
  bool ar_in_place = true; /* GPX_AR_INPLACE=0: the per-bucket kernel stages only, k_emit_dec16 always compacts (PlaceCols, gpx_ar16.hip.h) */
  TileArea tile_area{};
  int32_t tile_force = 0;   /* GPX_TILE_T (tuning): votes per scatter workgroup, 0 = chosen per call */
  int32_t tile_threads = 0; /* GPX_TILE_NT (tuning): 512 or 1024 threads per scatter workgroup, 0 = chosen per call */
  I4* reply_rows = nullptr;    /* [max_batch] packed ACCEPT_REPLY rows of the partition path (first use) */
  size_t lds_pad = 0;         /* GPX_LDS_PAD (tuning): extra dynamic LDS per bucket workgroup */
  /* asynchronous host-pointer calls (gpx_*_batch_async / gpx_engine_wait): GPX_ASYNC_DEPTH sets of device
   * columns, each with its own copy-out stream; one copy-in stream; allocated on first use */
  struct AsyncSet {
    int32_t* i32[11] = {};
    uint8_t* u8[3] = {};
    int32_t* cnt = nullptr;   /* device: the call's output count */
    int32_t* h_cnt = nullptr; /* pinned host copy of it */
    hipStream_t s_out = nullptr;
    hipEvent_t ev_in = nullptr, ev_k = nullptr, ev_cnt = nullptr;
    bool ready = false, busy = false;
    bool direct = false; /* the compacted outputs were written to the caller's buffers by k_copy_out */
    uint64_t ticket = 0;
    int ncols = 0;              /* compacted int32 output columns still to fetch (count-dependent) */
    int32_t* host_col[6] = {};  /* ... their host destinations and device sources */
    const int32_t* dev_col[6] = {};
    uint8_t* host_kind = nullptr; /* accept replies: d_kind */
    const uint8_t* dev_kind = nullptr;
    int32_t* host_count = nullptr; /* n_out / n_runs of the caller */
    int32_t n = 0;                 /* records of the call: the capacity of every output column (gpx.h) */
  } as[GPX_ASYNC_DEPTH_MAX];
  int async_depth = GPX_ASYNC_DEPTH; /* sets in use (GPX_ASYNC_DEPTH=n, up to GPX_ASYNC_DEPTH_MAX) */
  hipStream_t s_in = nullptr;
  uint64_t async_seq = 0;
  /* host blocks registered through gpx_host_register (base, bytes): the extent check of mapped_host, the
   * drain before gpx_host_unregister, and what gpx_engine_destroy still has to unregister */
  std::vector<std::pair<char*, size_t>> registered;
  std::map<char*, bool> registered_own; /* block -> pinned by this engine (false: it was pinned already; not ours to unpin) */
  bool async_in_engine = false, async_fill_memset = false; /* experiments: GPX_ASYNC_IN, GPX_ASYNC_FILL */
  bool async_no_direct = false; /* GPX_ASYNC_DIRECT=0: compacted outputs fetched by gpx_engine_wait even from registered memory */
  bool async_kernel_in = false; /* GPX_ASYNC_COPYIN=kernel (experiment): inputs read by k_copy_in from registered memory */
  /* wire codec (gpx_wire_host.inc): paxosID table, row free list, scratch - allocated on first use */
  DevNames N{};
  int64_t nm_tomb = 0;
  std::vector<int32_t> free_rows;
  bool free_init = false;
  int32_t *w_cnt = nullptr, *w_tile = nullptr, *w_err = nullptr;
  long long* w_tile_b = nullptr;
  unsigned long long* w_look = nullptr; /* [4][tiles] look-back words of the one-launch decode */
  uint32_t* w_ticket = nullptr;
  uint32_t w_epoch = 0;
  int wire_tile = 512; /* frames per workgroup of k_wire_decode1 (GPX_WD_TILE = 256 / 512) */
  uint8_t* w_stage = nullptr;      /* staging of BATCHED_ACCEPT_REPLY frames, 188 B per reply */
  long long* w_bucket_bytes = nullptr;
  int32_t* w_ones = nullptr;       /* a column of ones (gpx_request_batch without weights) */
};

namespace {

template <typename T>
int dev_alloc(gpx_engine* e, T** p, size_t count, bool zero) {
  void* q = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  hipError_t err = hipMalloc(&q, bytes);
  if (err != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "hipMalloc(%zu) -> %s", bytes, hipGetErrorString(err));
    return GPX_ENOMEM;
  }
  e->allocs.push_back(q);
  if (zero) {
    /* null-stream memset + wait: the engine's streams are non-blocking, so nothing else orders the
     * fill before the first kernel that uses the buffer (lazily allocated tables) */
    err = hipMemset(q, 0, bytes);
    if (err == hipSuccess) err = hipStreamSynchronize(nullptr);
    if (err != hipSuccess) {
      snprintf(g_err, sizeof(g_err), "hipMemset -> %s", hipGetErrorString(err));
      return GPX_EDEVICE;
    }
  }
  *p = (T*)q;
  return GPX_OK;
}

inline int grid_for(int64_t n) { return (int)((n + GPX_BLOCK - 1) / GPX_BLOCK); }

/* brackets a launch with events when profiling is on */
struct LaunchScope {
  gpx_engine* e;
  PendingEvent pe{};
  bool on;
  LaunchScope(gpx_engine* e_, const char* name) : e(e_), on(e_->profiling) {
    if (on) {
      pe.name = name;
      HIPQ(hipEventCreate(&pe.start));
      HIPQ(hipEventCreate(&pe.stop));
      HIPQ(hipEventRecord(pe.start, e->stream));
    }
  }
  ~LaunchScope() {
    if (on) {
      HIPQ(hipEventRecord(pe.stop, e->stream));
      e->pending.push_back(pe);
    }
  }
};

int flush_profile(gpx_engine* e) {
  if (e->pending.empty()) return GPX_OK;
  HIPCHK(hipStreamSynchronize(e->sF));
  HIPCHK(hipStreamSynchronize(e->sB));
  for (auto& pe : e->pending) {
    float ms = 0.f;
    HIPQ(hipEventElapsedTime(&ms, pe.start, pe.stop));
    auto& slot = e->prof[pe.name];
    slot.first += 1;
    slot.second += ms;
    HIPQ(hipEventDestroy(pe.start));
    HIPQ(hipEventDestroy(pe.stop));
  }
  e->pending.clear();
  return GPX_OK;
}

#define LAUNCH_L(e, name, kernel, grid, lds_bytes, ...)                                          \
  do {                                                                                           \
    LaunchScope _ls(e, name);                                                                    \
    hipLaunchKernelGGL(kernel, grid, dim3(GPX_BLOCK), (size_t)(lds_bytes), (e)->stream, __VA_ARGS__); \
  } while (0)
#define LAUNCH(e, name, kernel, grid, ...) LAUNCH_L(e, name, kernel, dim3(grid), 0, __VA_ARGS__)
/* per-bucket kernels: one workgroup per bucket, one lane per group (up to 1024) */
#define LAUNCH_B(e, name, kernel, ...)                                                            \
  do {                                                                                            \
    LaunchScope _ls(e, name);                                                                     \
    hipLaunchKernelGGL(kernel, dim3((e)->X.nbk), dim3((e)->bucket_threads), (e)->bucket_lds,      \
                       (e)->stream, __VA_ARGS__);                                                 \
  } while (0)
/* streaming kernels: GPX_FBLOCK threads */
#define LAUNCH_F(e, name, kernel, grid, lds_bytes, ...)                                           \
  do {                                                                                            \
    LaunchScope _ls(e, name);                                                                     \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(GPX_FBLOCK), (size_t)(lds_bytes), (e)->stream,    \
                       __VA_ARGS__);                                                              \
  } while (0)

#define LAUNCH_OC(e, name, kernel, grid, lds_bytes, ...)                                          \
  do {                                                                                            \
    LaunchScope _ls(e, name);                                                                     \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(GPX_OC_BLOCK), (size_t)(lds_bytes), (e)->stream,  \
                       __VA_ARGS__);                                                              \
  } while (0)

inline int ntiles_for(int64_t n) { return (int)((n + GPX_TILE - 1) / GPX_TILE); }
/* k_hist / k_scatter grids: 8 XCD slices of ceil(ntiles / 8) tiles (tile_of_block) */
inline int tile_grid(int ntiles) { return 8 * ((ntiles + 7) / 8); }
inline bool aligned16(std::initializer_list<const void*> ps) {
  for (const void* p : ps)
    if ((uintptr_t)p & 15) return false;
  return true;
}

void apply_streams(gpx_engine* e) {
  e->sF = e->sB = e->stream = e->user_stream ? e->user_stream : e->own_stream;
}

/* Opens a batch call: the call's epoch (what *X.unsorted is compared with). */
int begin_front(gpx_engine* e) {
  /* whatever an earlier call left parked can no longer be compacted: gpx_compact_last_dev says so once */
  e->last.stale = e->last.kind != 0 || e->last.stale;
  e->last.kind = 0;
  e->X.epoch = (uint32_t)(e->call_seq + 1);
  if (e->X.epoch == 0) { /* 2^32 calls: restart the epochs from cleared words */
    HIPQ(hipStreamSynchronize(e->stream));
    HIPQ(hipMemset(e->fs[0].unsorted, 0, sizeof(uint32_t)));
    HIPQ(hipMemset(e->rec_tag, 0, sizeof(uint32_t) * (size_t)e->cfg.max_batch));
    e->X.epoch = 1;
    e->call_seq = 0;
  }
  return 0;
}
/* LDS staging area of the per-bucket kernel, sized for THIS batch: mean records per bucket
 * + 25 % + 128, so that a thin batch (e.g. one proposal per group on an engine sized for 5-vote
 * rounds) still gets many workgroups per CU.  Buckets above it take the global-memory path. */
void begin_back(gpx_engine* e, int, int32_t n, bool v16 = false) {
  const int32_t rmax = v16 ? e->lds16_max : e->lds_recs_max;
  const int32_t rhw = v16 ? e->lds16_hw : e->lds_recs_hw;
  int64_t want = (int64_t)n / std::max(1, e->X.nbk);
  want = (want + want / 4 + 128 + 63) / 64 * 64;
  /* a call that carries more than the round the engine was sized for (several slots per group in
   * one batch) may take the LDS the hardware allows rather than fall back to the global-memory
   * path for every bucket */
  const int64_t mean = (int64_t)n / std::max(1, e->X.nbk);
  const bool oversized = mean + mean / 8 > rmax;
  const int64_t cap = oversized ? rhw : rmax;
  e->X.lds_recs = (int32_t)std::max<int64_t>(256, std::min<int64_t>(want, cap));
  /* beyond that too: nearly every bucket is regrouped in global memory - do not hold LDS it
   * will not use */
  if (oversized && mean > rhw + rhw / 4) e->X.lds_recs = 256;
  e->bucket_lds = (v16 ? GPX_BUCKET16_LDS_BYTES(e->X.gb, e->X.lds_recs)
                       : GPX_BUCKET_LDS_BYTES(e->X.gb, e->X.lds_recs)) + e->lds_pad;
}
void end_call(gpx_engine* e, int) { e->call_seq++; }

/* the verdict word of an ordered batch (gpx_one.hip.h) and a fresh, ascending epoch for it */
OneCtl one_ctl(gpx_engine* e) {
  if (++e->one_epoch == 0) { /* 2^32 launches: start the epochs again from a cleared word (the arrival counters behind it stay) */
    HIPQ(hipMemsetAsync(e->one_words, 0, sizeof(unsigned long long) * GPX_ONE_TICKETS, e->stream));
    e->one_epoch = 1;
  }
  return OneCtl{e->one_words, e->one_epoch};
}
/* are the compacted outputs of this call left parked when the batch is unusual (GPX_LAZY_OUTPUTS)? */
bool lazy_outputs(const gpx_engine* e) {
  if (e->lazy_override >= 0) return e->lazy_override != 0;
  return (e->ordered_mask & GPX_LAZY_OUTPUTS) != 0;
}
/* the compaction pass of an ordered ACCEPT / COMMIT batch that went through k_ac_one: a usual batch has nothing
 * for it to do (D.mark is not raised) */
void launch_one_compaction(gpx_engine* e, const gpx_engine::LastCall& L) {
  const DevScratch X0 = e->X;
  e->X = L.X;
  {
    LaunchScope _ls(e, "k_one_count");
    hipLaunchKernelGGL(k_one_count, dim3(L.nchunks), dim3(GPX_DCHUNK), 0, e->stream, e->X, L.n, L.D);
  }
  if (L.kind == 2) {
    {
      LaunchScope _ls(e, "k_emit_runs_direct");
      hipLaunchKernelGGL(k_emit_runs_direct<true>, dim3(std::min(L.nchunks, GPX_EMIT_GRID)), dim3(GPX_DCHUNK), 0, e->stream, e->X, L.n,
                         L.nchunks, L.gidx, L.D, L.x_gidx, L.x_first, L.x_count, L.count, 0, 1);
    }
    LAUNCH(e, "k_copy_runs", k_copy_runs, 256, e->X, L.D, L.x_gidx, L.x_first, L.x_count);
  } else {
    LaunchScope _ls(e, "k_emit_runs_direct");
    hipLaunchKernelGGL(k_emit_runs_direct<false>, dim3(std::min(L.nchunks, GPX_EMIT_GRID)), dim3(GPX_DCHUNK), 0, e->stream, e->X, L.n,
                       L.nchunks, L.gidx, L.D, L.x_gidx, L.x_first, L.x_count, L.count, 0, 1);
  }
  e->X = X0;
}

/* bucket partition front end, part 1: per-bucket record counts of the batch */
void front_hist(gpx_engine* e, int32_t n, const int32_t* gidx, uint8_t* status, int is_votes,
                int check_order = 0) {
  const int ntiles = ntiles_for(n);
  /* one histogram workgroup per `hsub` scatter tiles: about one workgroup per CU */
  /* (measured: 5 sub-tiles per histogram workgroup for 3 M votes - fewer returning atomics per bucket
   * word, and the slices of a workgroup's sub-tiles are adjacent in every bucket region, which the
   * scatter's L2 likes: k_scatter_ar16 53 -> 49 us) */
  int hsub = std::max(1, std::min(GPX_HSUB_MAX, (ntiles + 75) / 150));
  if (const char* hs = getenv("GPX_HSUB")) hsub = std::max(1, std::min(GPX_HSUB_MAX, atoi(hs))); /* tuning */
  const int nsuper = (ntiles + hsub - 1) / hsub;
  const size_t lds = (size_t)hsub * e->X.nbk * sizeof(int32_t);
  if (aligned16({gidx}))
    LAUNCH_F(e, "k_hist", k_hist<true>, tile_grid(nsuper), lds, n, ntiles, gidx, e->S.G, e->X, status,
             is_votes, check_order, hsub);
  else
    LAUNCH_F(e, "k_hist", k_hist<false>, tile_grid(nsuper), lds, n, ntiles, gidx, e->S.G, e->X, status,
             is_votes, check_order, hsub);
}

/* k_scatter_ac with 16-byte column loads when the caller's columns allow it */
void launch_scatter_ac(gpx_engine* e, int32_t n, const int32_t* gidx, const int32_t* bnum,
                       const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                       const uint8_t* flags, int32_t* r_bnum, int32_t* r_bcoord, int32_t* r_maxcp,
                       uint8_t* r_flags, int32_t only_unsorted = 0) {
  const int ntiles = ntiles_for(n);
  const bool vec = aligned16({gidx, bnum, bcoord, slot, median_cp}) && !((uintptr_t)flags & 3);
  if (vec)
    LAUNCH_F(e, "k_scatter_ac", k_scatter_ac<true>, tile_grid(ntiles), (size_t)e->X.nbk * 4, n, ntiles,
             e->S.G, e->X, gidx, bnum, bcoord, slot, median_cp, flags, r_bnum, r_bcoord, r_maxcp, r_flags,
             only_unsorted);
  else
    LAUNCH_F(e, "k_scatter_ac", k_scatter_ac<false>, tile_grid(ntiles), (size_t)e->X.nbk * 4, n, ntiles,
             e->S.G, e->X, gidx, bnum, bcoord, slot, median_cp, flags, r_bnum, r_bcoord, r_maxcp, r_flags,
             only_unsorted);
}

int check_batch(gpx_engine* h, int32_t n) {
  if (!h || n < 0) return GPX_EINVAL;
  if (n > h->cfg.max_batch) return GPX_ECAPACITY;
  if (h->h_abort && *(volatile uint32_t*)h->h_abort) {
    /* a workgroup of a one-launch kernel waited two seconds at grid_exchange for workgroups that never became resident
     * (somebody this process cannot see holds the device's CUs): that call applied only part of its batch */
    snprintf(g_err, sizeof(g_err), "an exchange kernel gave up waiting for its grid (call epoch %u): engine state is "
             "incomplete; set GPX_DEVICE_SHARERS to the number of processes sharing the device", *(volatile uint32_t*)h->h_abort);
    return GPX_EDEVICE;
  }
  return GPX_OK;
}

/* A synchronous call's wait for its stream, and the question every such wait is followed by: did a workgroup of an
 * exchange kernel give up during THIS call (check_batch reads the host-mapped word)?  Then the call's outputs are not a
 * prefix of anything and the call itself - not the next one - says GPX_EDEVICE. */
#define SYNC_CHECKED(h, stream_)                                \
  do {                                                          \
    HIPCHK(hipStreamSynchronize(stream_));                      \
    if (int _rc_abort = check_batch((h), 0)) return _rc_abort;  \
  } while (0)

/* the streams that may have a one-launch kernel - workgroups waiting for each other at grid_exchange - on this engine's
 * device at the same moment: those of this process's engines, plus either what GPX_DEVICE_SHARERS says about other
 * processes (explicit: the count of processes, each taken to launch on as many streams as this one) or what their
 * registry files say */
int xchg_share(const gpx_engine* e) {
  std::lock_guard<std::mutex> lk(g_live_mu);
  int live = 1;
  auto it = g_live_streams.find(e->device);
  if (it != g_live_streams.end()) live = std::max<int>(1, (int)it->second.size());
  if (e->sharers_set) return live * e->sharers;
  return live + proc_registry_foreign(e->device);
}
/* May a call over `nchunks` chunks take the ONE-launch form (gpx_one.hip.h) with this kernel?  Only with a grid that is
 * resident whatever else is launching: per CU at most two workgroups AND at most one fewer than the occupancy query
 * promises for THIS kernel (the hardware admits one fewer than the query says for some register counts:
 * MI355X_MICROARCH.md, residency), divided by the sharers.  Then: the grid (a multiple of 16: every counter line gets
 * the same number of arrivals) and what the cumulative arrival counters read afterwards. */
bool xchg_ctl(gpx_engine* e, int nchunks, GridXchg* Q, int* grid, const void* kernel, int block) {
  const int g16 = (std::max(nchunks, 1) + GPX_GX_LINES - 1) / GPX_GX_LINES * GPX_GX_LINES;
  if (!e->one_launch) return false;
  int per_cu = 2;
  {
    auto it = e->occ.find(kernel);
    if (it == e->occ.end()) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, 0) != hipSuccess) {
        (void)hipGetLastError();
        nb = 0;
      }
      it = e->occ.emplace(kernel, nb).first;
    }
    per_cu = std::min(2, it->second - 1);
  }
  if (per_cu < 1 || (int64_t)g16 * xchg_share(e) > (int64_t)per_cu * e->cus) return false;
  const OneCtl C = one_ctl(e);
  *grid = g16;
  Q->arrive = (uint32_t*)(e->one_words + GPX_ONE_TICKETS);
  Q->verdict = C.verdict;
  Q->epoch = C.epoch;
  e->gx_arrive += (uint32_t)(g16 / GPX_GX_LINES);
  Q->arrive_target = e->gx_arrive + e->xchg_test_skew; /* (skew: test hook - a target no launch reaches: every poller gives up) */
  Q->timeout_ms = e->xchg_timeout_ms;
  return true;
}

/* A copy between the device and HOST memory.  Memory the runtime knows as pinned (a block given to gpx_host_register, one
 * from gpx_host_alloc, or pinned elsewhere) goes out as one DMA.  PAGEABLE memory goes out in pieces of 512 KB: above
 * ~1 MB the HIP runtime would pin the caller's pages for the length of the copy instead of staging them through its own
 * buffer (profiles/r06_copy_path_probe.txt), and every GPU page fault this code base has on file was raised while the main
 * thread sat in exactly that transient pinning (profiles/r06_abort_backtrace.txt; once under gpx_group_create's 64 MB
 * chunks).  A piece is staged; the engine never asks the runtime to pin memory it was not given.  Pageable calls pay for
 * it in rate (about half of the pinned path's) - the fast host paths are the registered / gpx_host_alloc ones anyway. */
constexpr size_t GPX_PAGEABLE_PIECE = (size_t)512 << 10;
bool host_is_pinned(gpx_engine* e, const void* p) {
  for (auto& r : e->registered)
    if ((const char*)p >= r.first && (const char*)p < r.first + r.second) return true;
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError(); /* unknown to the runtime: plain pageable memory */
    return false;
  }
  return at.type == hipMemoryTypeHost;
}
hipError_t xfer(gpx_engine* e, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  if (bytes <= GPX_PAGEABLE_PIECE || host_is_pinned(e, kind == hipMemcpyHostToDevice ? src : (const void*)dst))
    return hipMemcpyAsync(dst, src, bytes, kind, st);
  for (size_t o = 0; o < bytes; o += GPX_PAGEABLE_PIECE) {
    const hipError_t rc = hipMemcpyAsync((char*)dst + o, (const char*)src + o, std::min(GPX_PAGEABLE_PIECE, bytes - o), kind, st);
    if (rc != hipSuccess) return rc;
  }
  return hipSuccess;
}

template <int KMAX>
void launch_bucket_ar16(gpx_engine* e, const Stage16& O, const VoteCols& in, uint8_t* status) {
  LAUNCH_B(e, "k_bucket_ar16", (k_bucket16<B16_AR, KMAX>), e->S, e->X, O, in, AcceptOut{}, status);
}
template <int KMAX>
void launch_bucket_propose(gpx_engine* e, int32_t* slot, int32_t* bnum, int32_t* bcoord,
                           int32_t* median, uint8_t* status, const int64_t* handle) {
  LAUNCH_B(e, "k_bucket_propose", (k_bucket_propose<KMAX>), e->S, e->X, slot, bnum, bcoord, median,
           status, handle);
}

/* ---- staged host-pointer calls -------------------------------------------------------- */
/* A small batch (at most GPX_STAGE_N records) crosses PCIe as ONE block each way: the caller's columns
 * are packed into a pinned host block (a few KB .. 1 MB of memcpy), one hipMemcpyAsync takes it to the
 * device, the _dev twin runs on slices of the device blocks, one hipMemcpyAsync brings every output
 * column back (full capacity: counts are not known before), one stream wait.  A copy per column
 * costs ~8 us of runtime each - 14 of them were most of a tiny call's ~130 us. */
struct Stage {
  gpx_engine* e;
  size_t in_off = 0, out_off = 0;
  explicit Stage(gpx_engine* e_) : e(e_) {}
  static size_t a16(size_t b) { return (b + 15) & ~(size_t)15; }
  /* packs one input column; returns its device address (null column -> null) */
  template <typename T>
  const T* in(const T* host, size_t count) {
    if (!host) return nullptr;
    const size_t b = count * sizeof(T);
    memcpy(e->hs_in + in_off, host, b);
    const T* d = (const T*)(e->ds_in + in_off);
    in_off += a16(b);
    return d;
  }
  /* reserves one output column; *host_view = where it will be in the pinned block after finish() */
  template <typename T>
  T* out(size_t count, const T** host_view) {
    T* d = (T*)(e->ds_out + out_off);
    *host_view = (const T*)(e->hs_out + out_off);
    out_off += a16(count * sizeof(T));
    return d;
  }
  int upload() {
    if (in_off) HIPCHK(hipMemcpyAsync(e->ds_in, e->hs_in, in_off, hipMemcpyHostToDevice, e->sF));
    return GPX_OK;
  }
  int finish() {
    HIPCHK(hipMemcpyAsync(e->hs_out, e->ds_out, out_off, hipMemcpyDeviceToHost, e->sB));
    HIPCHK(hipStreamSynchronize(e->sB));
    return check_batch(e, 0); /* (an exchange kernel of THIS call gave up: GPX_EDEVICE now, not on the next call) */
  }
};

}  // namespace

extern "C" {

int gpx_abi_version(void) { return GPX_ABI_VERSION; }
const char* gpx_last_error(void) { return g_err; }

int gpx_engine_create(const gpx_config* cfg, gpx_engine** out) {
  g_err[0] = 0;
  if (!cfg || !out) return GPX_EINVAL;
  if (cfg->max_groups <= 0 || cfg->kmax < 1 || cfg->kmax > GPX_KMAX_LIMIT || cfg->max_batch <= 0)
    return GPX_EINVAL;
  if (cfg->window < 4 || cfg->window > 64 || (cfg->window & (cfg->window - 1))) return GPX_EINVAL;
  /* A GPU that has just been powered up (fresh box, "device(s) in a low-power state") can answer the
   * first runtime call with hipErrorNoDevice for a moment - seen once on the MI355X pool.  Wait for it, but
   * only where a ROCm driver is present at all (/dev/kfd): a host without a GPU fails at once. */
  int ndev = 0;
  hipError_t de = hipErrorNoDevice;
  const int attempts = access("/dev/kfd", F_OK) == 0 ? 15 : 1;
  for (int attempt = 0; attempt < attempts; attempt++) {
    de = hipGetDeviceCount(&ndev);
    if (de == hipSuccess && ndev > 0) break;
    (void)hipGetLastError();
    if (de != hipErrorNoDevice && de != hipSuccess) break; /* not the transient answer */
    if (attempt + 1 < attempts) usleep(200 * 1000);
  }
  if (de != hipSuccess || ndev <= 0) {
    snprintf(g_err, sizeof(g_err), "no HIP device visible (%s)", de == hipSuccess ? "count 0" : hipGetErrorString(de));
    return GPX_EDEVICE;
  }
  gpx_engine* e = new gpx_engine();
  e->cfg = *cfg;
  /* a failing runtime call inside creation gives back everything allocated so far */
#define HIPCHK_CREATE(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      snprintf(g_err, sizeof(g_err), "%s:%d %s -> %s", __FILE__, __LINE__, #expr,            \
               hipGetErrorString(_e));                                                       \
      gpx_engine_destroy(e);                                                                 \
      return GPX_EDEVICE;                                                                    \
    }                                                                                        \
  } while (0)
  if (cfg->device >= 0) {
    HIPCHK_CREATE(hipSetDevice(cfg->device));
    e->device = cfg->device;
  } else {
    HIPCHK_CREATE(hipGetDevice(&e->device));
  }
  HIPCHK_CREATE(hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking));
  apply_streams(e);
  const size_t G = (size_t)cfg->max_groups, W = (size_t)cfg->window, K = (size_t)cfg->kmax;
  const size_t N = (size_t)cfg->max_batch;
  DevState& S = e->S;
  S.G = cfg->max_groups;
  S.kmax = cfg->kmax;
  S.W = cfg->window;
  S.my_id = cfg->my_id;
  S.flags = cfg->flags;
  int rc = GPX_OK;
#define A(ptr, count, zero)                                   \
  if ((rc = dev_alloc(e, &(ptr), (count), (zero))) != GPX_OK) { \
    gpx_engine_destroy(e);                                    \
    return rc;                                                \
  }
  A(S.g_flags, G, true);
  A(S.g_version, G, true);
  A(S.a_slot, G, true);
  A(S.a_bnum, G, true);
  A(S.a_bcoord, G, true);
  A(S.a_gc, G, true);
  A(S.c_bnum, G, true);
  A(S.c_bcoord, G, true);
  A(S.c_next, G, true);
  A(S.c_pcount, G, true);
  A(S.members, K * G, true);
  A(S.node_slots, K * G, true);
  A(S.p_ring, W * G, true);
  A(S.acc_ring, W * G, true);
  A(S.com_ring, W * G, true);
  DevScratch& X = e->X;
  /* buckets of 256 groups (one lane per group, whole bucket staged in LDS, 4 workgroups per
   * CU); beyond 1M groups the buckets grow so that there are at most GPX_MAX_BUCKETS */
  auto nbk_for = [&](int sh) { return (int64_t)((G + ((size_t)1 << sh) - 1) >> sh); };
  /* 512 groups per bucket where the group count allows at least a few hundred buckets (measured on
   * MI355X, 1 M groups: 3 M votes partition 10 % faster over 1,954 buckets than over 3,907 and the
   * per-bucket kernels run as fast with 512 lanes as with 256); small tables keep 256 */
  X.shift = nbk_for(GPX_MIN_SHIFT + 1) >= 256 ? GPX_MIN_SHIFT + 1 : GPX_MIN_SHIFT;
  while (nbk_for(X.shift) > GPX_MAX_BUCKETS) X.shift++;
  if (const char* sh = getenv("GPX_BUCKET_SHIFT")) { /* tuning knob */
    const int v = atoi(sh);
    if (v >= GPX_MIN_SHIFT && v <= 20 && nbk_for(v) <= GPX_MAX_BUCKETS) X.shift = v;
  }
  X.gb = 1 << X.shift;
  X.nbk = (int32_t)nbk_for(X.shift);
  X.g_base = 0;
  X.g_end = cfg->max_groups;
  e->bucket_threads = std::min(1024, X.gb);
  /* LDS staging capacity: the expected kmax * gb records of a full round + ~7 sigma, within
   * 8 records per thread and the CU's 160 KiB; 960 (K <= 3, gb 256) keeps 4 workgroups per CU */
  {
    int64_t want = (int64_t)cfg->kmax * X.gb;
    want += want / 4 + 64;
    if (cfg->kmax <= 3 && X.gb == 256) want = 960;
    const int64_t lds_cap = ((int64_t)160 * 1024 - 1024 - (int64_t)X.gb * 8) / (8 + 4 * GPX_PAY_WORDS);
    want = std::min<int64_t>(want, std::min<int64_t>(lds_cap, (int64_t)GPX_BUCKET_ITEMS * e->bucket_threads));
    if (const char* lr = getenv("GPX_LDS_RECS")) want = std::min<int64_t>(want, std::max(64, atoi(lr)));
    X.lds_recs = (int32_t)want;
    e->lds_recs_max = X.lds_recs;
    e->lds_recs_hw = (int32_t)std::max<int64_t>(want, std::min<int64_t>(lds_cap, (int64_t)GPX_BUCKET_ITEMS * e->bucket_threads));
  }
  { /* 16-byte vote records: kmax votes per group of a full round + 25 %; the hardware limit beyond */
    const int64_t cap16 = ((int64_t)160 * 1024 - 1024 - (int64_t)X.gb * 8) / 16;
    int64_t want = (int64_t)cfg->kmax * X.gb;
    want += want / 4 + 128;
    e->lds16_max = (int32_t)std::min<int64_t>(want, cap16);
    e->lds16_hw = (int32_t)cap16;
    e->ac16 = X.shift <= V16_MAX_SHIFT && e->bucket_threads == X.gb;
    if (X.shift <= V16_MAX_SHIFT) {
      e->shift16 = X.shift;
      e->nbk16 = X.nbk;
      e->ar_passes = 1;
    } else { /* more than 4 M groups: passes over ranges of 4096 x 1024 groups */
      e->shift16 = V16_MAX_SHIFT;
      const int64_t range = (int64_t)GPX_MAX_BUCKETS << V16_MAX_SHIFT;
      e->ar_passes = (int32_t)(((int64_t)G + range - 1) / range);
      e->nbk16 = GPX_MAX_BUCKETS;
      const int64_t gb16 = (int64_t)1 << V16_MAX_SHIFT;
      const int64_t cap16b = ((int64_t)160 * 1024 - 1024 - gb16 * 8) / 16;
      int64_t want16 = (int64_t)cfg->kmax * gb16;
      want16 += want16 / 4 + 128;
      e->lds16_max = (int32_t)std::min<int64_t>(want16, cap16b);
      e->lds16_hw = (int32_t)cap16b;
    }
    if ((rc = dev_alloc(e, &e->ar_chain, 2, true)) != GPX_OK) {
      gpx_engine_destroy(e);
      return rc;
    }
  }
  if (const char* lp = getenv("GPX_LDS_PAD")) e->lds_pad = (size_t)std::max(0, atoi(lp));
  if (const char* tr = getenv("GPX_TRY_RUNS")) e->env_mask = atoi(tr) ? GPX_TRY_REPLY_RUNS : 0;
  e->ar_tiles = true; /* GPX_AR_TILES=0: the partition front end for every shuffled call (comparison runs) */
  if (const char* sl = getenv("GPX_AR_TILES")) e->ar_tiles = atoi(sl) != 0;
  if (const char* sl = getenv("GPX_AR_INPLACE")) e->ar_in_place = atoi(sl) != 0;
  if (const char* tt = getenv("GPX_TILE_T")) e->tile_force = atoi(tt);
  if (const char* tt = getenv("GPX_TILE_NT")) e->tile_threads = atoi(tt);
  e->tile_area.xcd_rows = 1;
  if (const char* xr = getenv("GPX_TILE_XCD_ROWS")) e->tile_area.xcd_rows = atoi(xr) != 0;
#ifdef GPX_TL_TRACE
  if (const char* ab = getenv("GPX_TL_ABLATE")) e->tile_area.xcd_rows |= atoi(ab) << 1; /* (wrong results: timing only) */
#endif
  if (const char* sv = getenv("GPX_SAR_MAX_N")) e->sar_max_n = std::max(0, std::min(GPX_SAR_MAX_N, atoi(sv)));
  e->sar_max_n = std::min(e->sar_max_n, cfg->max_batch); /* its keys live in X.perm: [max_batch] entries */
  e->ordered_mask = e->env_mask;
  e->bucket_lds = GPX_BUCKET_LDS_BYTES(X.gb, X.lds_recs) + e->lds_pad;
  const size_t bucket_lds_hw = GPX_BUCKET_LDS_BYTES(X.gb, e->lds_recs_hw) + e->lds_pad;
  if (bucket_lds_hw > 64 * 1024) { /* more dynamic LDS than the default limit: opt in per kernel */
    const void* fns[] = {(const void*)k_bucket_propose<4>,
                         (const void*)k_bucket_propose<8>, (const void*)k_bucket_propose<16>,
                         (const void*)k_bucket_accept,     (const void*)k_bucket_commit,
                         (const void*)k_bucket_pack_ar,    (const void*)k_bucket_reqbatch,
                         (const void*)k_bucket_prepare,
                         (const void*)k_bucket_prepare_reply<4, 64>,
                         (const void*)k_bucket_prepare_reply<8, 64>,
                         (const void*)k_bucket_prepare_reply<16, 64>};
    for (const void* f : fns)
      HIPCHK_CREATE(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bucket_lds_hw));
  }
  {
    if ((rc = dev_alloc(e, &e->small_tickets, 2 * (GPX_SMALL_DIRECT_MAX_N / GPX_DCHUNK), true)) != GPX_OK ||
        (rc = dev_alloc(e, &e->small_draw, 1, true)) != GPX_OK) {
      gpx_engine_destroy(e);
      return rc;
    }
    if (hipHostMalloc((void**)&e->hs_in, GPX_STAGE_BYTES, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&e->hs_out, GPX_STAGE_BYTES, hipHostMallocDefault) != hipSuccess) {
      snprintf(g_err, sizeof(g_err), "hipHostMalloc(%zu) failed", (size_t)GPX_STAGE_BYTES);
      gpx_engine_destroy(e);
      return GPX_ENOMEM;
    }
    if ((rc = dev_alloc(e, &e->ds_in, GPX_STAGE_BYTES, false)) != GPX_OK ||
        (rc = dev_alloc(e, &e->ds_out, GPX_STAGE_BYTES, false)) != GPX_OK) {
      gpx_engine_destroy(e);
      return rc;
    }
  }
  {
    const size_t hw16 = GPX_BUCKET16_LDS_BYTES((size_t)1 << e->shift16, e->lds16_hw) + e->lds_pad;
    const void* fns[] = {(const void*)k_bucket16<B16_AR, 4>, (const void*)k_bucket_ar16_k5, (const void*)k_bucket16<B16_AR, 8>,
                         (const void*)k_bucket16<B16_AR, 16>, (const void*)k_bucket16<B16_ACCEPT, 4>,
                         (const void*)k_bucket16<B16_COMMIT, 4>};
    for (const void* f : fns)
      HIPCHK_CREATE(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hw16));
  }
  { /* the one-launch small accept-reply kernel stages GPX_SAR_LDS_BYTES = 44 KiB (gpx_small.hip.h) */
    const void* fns[] = {(const void*)k_ar_tiny<4>, (const void*)k_ar_tiny<8>, (const void*)k_ar_tiny<16>};
    for (const void* f : fns)
      HIPCHK_CREATE(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)GPX_SAR_LDS_BYTES));
  }
  { /* k_hist may take up to GPX_HSUB_MAX histograms of dynamic LDS */
    const int hl = GPX_HSUB_MAX * std::max(X.nbk, e->nbk16) * (int)sizeof(int32_t);
    if (hl > 64 * 1024) {
      HIPCHK_CREATE(hipFuncSetAttribute((const void*)k_hist<true>, hipFuncAttributeMaxDynamicSharedMemorySize, hl));
      HIPCHK_CREATE(hipFuncSetAttribute((const void*)k_hist<false>, hipFuncAttributeMaxDynamicSharedMemorySize, hl));
    }
  }
  const size_t nbk_alloc = (size_t)std::max(X.nbk, e->nbk16); /* accept-reply passes may use more, smaller buckets */
  for (auto& f : e->fs) { /* double-buffered front-end scratch */
    A(f.bucket_tot, nbk_alloc, true);
    A(f.tile_rel, ((N + GPX_TILE - 1) / GPX_TILE) * nbk_alloc, false);
    A(f.bucket_off, nbk_alloc + 1, true);
    A(f.rec, N, false);
    A(f.unsorted, 1, true);
    A(f.chunk_cnt, GPX_CHUNK_CNT_WORDS(N), true); /* the counts, then st_total and mark at FIXED words behind them */
  }
  X.bucket_tot = e->fs[0].bucket_tot;
  X.tile_rel = e->fs[0].tile_rel;
  X.bucket_off = e->fs[0].bucket_off;
  X.rec = e->fs[0].rec;
  X.unsorted = e->fs[0].unsorted;
  X.epoch = 1;
  A(X.rank2, N, false);
  A(X.perm, N, false);
  A(X.o_rec, N, false);
  A(X.bucket_nout, nbk_alloc, true);
  A(e->rec_tag, N, true);
  A(e->one_words, GPX_ONE_TICKETS + GPX_GX_LINES * 16, true); /* the verdict word, then grid_exchange's arrival counters (a 128-byte line each) */
  A(X.counters, 3, true);
  for (int i = 0; i < 12; i++) A(e->st_i32[i], N, false);
  for (int i = 0; i < 4; i++) A(e->st_u8[i], N, false);
  A(e->st_count, 4, true);
#undef A
  {
    HIPCHK_CREATE(hipDeviceGetAttribute(&e->cus, hipDeviceAttributeMultiprocessorCount, e->device));
    if (const char* sv = getenv("GPX_XCHG_SLOTS")) e->one_launch = atoi(sv) != 0; /* test switch: 0 = the two-launch forms */
    if (const char* sv = getenv("GPX_PERS_CHUNKS")) {
      int a = 0, b = 0, c = 0;
      if (sscanf(sv, "%d,%d,%d", &a, &b, &c) == 3) e->pers_max_chunks[0] = a, e->pers_max_chunks[1] = b, e->pers_max_chunks[2] = c;
    }
    if (const char* sh = getenv("GPX_DEVICE_SHARERS")) e->sharers = std::max(1, atoi(sh)), e->sharers_set = true;
    if (const char* tm = getenv("GPX_XCHG_TIMEOUT_MS")) e->xchg_timeout_ms = (uint32_t)std::max(1, atoi(tm));
    if (const char* sk = getenv("GPX_XCHG_TEST_SKEW")) e->xchg_test_skew = (uint32_t)std::max(0, atoi(sk));
    if (const char* ad = getenv("GPX_ASYNC_DEPTH")) e->async_depth = std::max(1, std::min(GPX_ASYNC_DEPTH_MAX, atoi(ad)));
    HIPCHK_CREATE(hipHostMalloc((void**)&e->h_abort, 64, hipHostMallocMapped));
    memset(e->h_abort, 0, 64);
    void* dptr = nullptr;
    HIPCHK_CREATE(hipHostGetDevicePointer(&dptr, e->h_abort, 0));
    X.xabort = (uint32_t*)dptr;
  }
  HIPCHK_CREATE(hipDeviceSynchronize());
#undef HIPCHK_CREATE
  live_add(e->device, e->stream);
  e->registered_live = true;
  *out = e;
  return GPX_OK;
}

/* every stream a call of this engine may have queued work on: the engine's stream, the copy-in stream and the
 * copy-out stream of every asynchronous set.  After this nothing of the engine is in flight. */
static void drain_all(gpx_engine* h) {
  if (h->s_in) HIPQ(hipStreamSynchronize(h->s_in));
  if (h->sF) HIPQ(hipStreamSynchronize(h->sF));
  if (h->sB && h->sB != h->sF) HIPQ(hipStreamSynchronize(h->sB));
  for (auto& a : h->as)
    if (a.s_out) HIPQ(hipStreamSynchronize(a.s_out));
}

int gpx_engine_destroy(gpx_engine* h) {
  if (!h) return GPX_EINVAL;
  /* first: nothing in flight - copies queued for a ticket nobody waited for read the AsyncSet columns and
   * write through the caller's registered mappings; only then may either go away */
  drain_all(h);
  if (h->registered_live) {
    live_drop(h->device, h->user_stream ? h->user_stream : h->own_stream);
    h->registered_live = false;
  }
  if (!h->registered_own.empty()) HIPQ(hipDeviceSynchronize());
  for (auto& sg : h->registered_own) /* (gpx_host_alloc blocks are in `registered` only) */
    if (sg.second) HIPQ(hipHostUnregister(sg.first));
  h->registered_own.clear();
  h->registered.clear();
  for (void* b : h->host_blocks) HIPQ(hipHostFree(b));
  h->host_blocks.clear();
  if (h->h_abort) HIPQ(hipHostFree(h->h_abort));
  for (auto& pe : h->pending) {
    HIPQ(hipEventDestroy(pe.start));
    HIPQ(hipEventDestroy(pe.stop));
  }
  for (void* p : h->allocs) HIPQ(hipFree(p));
  if (h->hs_in) HIPQ(hipHostFree(h->hs_in));
  if (h->hs_out) HIPQ(hipHostFree(h->hs_out));
  if (h->arena) HIPQ(hipFree(h->arena));
  for (auto& a : h->as) {
    if (a.h_cnt) HIPQ(hipHostFree(a.h_cnt));
    if (a.ev_in) HIPQ(hipEventDestroy(a.ev_in));
    if (a.ev_k) HIPQ(hipEventDestroy(a.ev_k));
    if (a.ev_cnt) HIPQ(hipEventDestroy(a.ev_cnt));
    if (a.s_out) HIPQ(hipStreamDestroy(a.s_out));
  }
  if (h->s_in) HIPQ(hipStreamDestroy(h->s_in));
  if (h->own_stream) HIPQ(hipStreamDestroy(h->own_stream));
  (void)hipGetLastError();
  delete h;
  return GPX_OK;
}

int gpx_engine_set_stream(gpx_engine* h, void* hip_stream) {
  if (!h) return GPX_EINVAL;
  HIPCHK(hipStreamSynchronize(h->sF));
  HIPCHK(hipStreamSynchronize(h->sB));
  if (h->registered_live) live_drop(h->device, h->user_stream ? h->user_stream : h->own_stream);
  h->user_stream = (hipStream_t)hip_stream;
  apply_streams(h);
  if (h->registered_live) live_add(h->device, h->stream);
  return GPX_OK;
}

int gpx_engine_set_ordered_batches(gpx_engine* h, int32_t mask) {
  if (!h || (mask & ~(GPX_ORDERED_PROPOSE | GPX_ORDERED_ACCEPT | GPX_ORDERED_COMMIT | GPX_ORDERED_REPLY_RUNS |
                      GPX_TRY_REPLY_RUNS | GPX_LAZY_OUTPUTS)))
    return GPX_EINVAL;
  h->ordered_mask = mask | h->env_mask;
  return GPX_OK;
}

int gpx_host_register(gpx_engine* h, void* ptr, size_t bytes) {
  if (!h || !ptr || !bytes) return GPX_EINVAL;
  if (h->registered_own.count((char*)ptr)) return GPX_EINVAL; /* already registered through this engine */
  /* nothing of the process is in flight while the pinning changes (the runtime pins pageable memory itself around
   * large copies: whatever it still holds for earlier copies of this range has been let go) */
  HIPCHK(hipDeviceSynchronize());
  /* The caller's exact range.  (Round 6 tried whole pages, each pinned once per process: the runtime then takes
   * NEIGHBOURING heap memory on the rounded pages for registered memory and refuses copies that straddle the
   * boundary - hipMemcpyAsync: invalid argument in five tests.  Not kept.) */
  hipError_t err = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  bool own = true;
  if (err == hipErrorHostMemoryAlreadyRegistered) { /* pinned memory the caller got elsewhere: known from now on, not ours to unpin */
    (void)hipGetLastError();
    err = hipSuccess;
    own = false;
  }
  HIPCHK(err);
  h->registered_own[(char*)ptr] = own;
  h->registered.emplace_back((char*)ptr, bytes);
  return GPX_OK;
}
/* An asynchronous call may still be writing through the block's device mapping (k_copy_out on a set's copy-out
 * stream) or reading it (the copy-in stream's DMA): the mapping is only taken away once every stream of the
 * engine has drained.  The tickets stay valid - gpx_engine_wait on them returns at once afterwards. */
int gpx_host_unregister(gpx_engine* h, void* ptr) {
  if (!h || !ptr) return GPX_EINVAL;
  if (std::find(h->host_blocks.begin(), h->host_blocks.end(), ptr) != h->host_blocks.end())
    return GPX_EINVAL; /* a gpx_host_alloc block: gpx_host_free gives it back */
  auto sg = h->registered_own.find((char*)ptr);
  if (sg == h->registered_own.end()) return GPX_EINVAL; /* not a block gpx_host_register was given */
  drain_all(h);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize()); /* (other streams of the process may hold DMA on these pages too: nothing may) */
  if (sg->second) HIPCHK(hipHostUnregister(ptr));
  h->registered_own.erase(sg);
  for (size_t i = 0; i < h->registered.size(); i++)
    if (h->registered[i].first == (char*)ptr) {
      h->registered.erase(h->registered.begin() + (long)i);
      break;
    }
  return GPX_OK;
}

/* Host memory the DMA engines reach at the link's full rate (hipHostMalloc): what a JNI host wraps in a direct
 * ByteBuffer (NewDirectByteBuffer) for its batch columns instead of registering JVM memory afterwards.  Known to
 * the asynchronous calls like a registered block (mapped_host). */
int gpx_host_alloc(gpx_engine* h, size_t bytes, void** out) {
  if (!h || !out || !bytes) return GPX_EINVAL;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    snprintf(g_err, sizeof(g_err), "hipHostMalloc(%zu) failed", bytes);
    return GPX_ENOMEM;
  }
  h->host_blocks.push_back(p);
  h->registered.emplace_back((char*)p, bytes);
  *out = p;
  return GPX_OK;
}
int gpx_host_free(gpx_engine* h, void* ptr) {
  if (!h || !ptr) return GPX_EINVAL;
  auto it = std::find(h->host_blocks.begin(), h->host_blocks.end(), ptr);
  if (it == h->host_blocks.end()) return GPX_EINVAL;
  drain_all(h); /* as gpx_host_unregister: nothing of the engine may still read or write the block */
  HIPCHK(hipGetLastError());
  h->host_blocks.erase(it);
  for (size_t i = 0; i < h->registered.size(); i++)
    if (h->registered[i].first == (char*)ptr) {
      h->registered.erase(h->registered.begin() + (long)i);
      break;
    }
  HIPCHK(hipHostFree(ptr));
  return GPX_OK;
}

int gpx_engine_sync(gpx_engine* h) {
  if (!h) return GPX_EINVAL;
  HIPCHK(hipStreamSynchronize(h->sF));
  HIPCHK(hipStreamSynchronize(h->sB));
  /* the asynchronous calls' copy streams too: "completes the device work" includes their copies (the tickets
   * still have to be waited for - gpx_engine_wait hands over the counts) */
  if (h->s_in) HIPCHK(hipStreamSynchronize(h->s_in));
  for (auto& a : h->as)
    if (a.s_out) HIPCHK(hipStreamSynchronize(a.s_out));
  return check_batch(h, 0); /* (an exchange kernel that gave up: GPX_EDEVICE) */
}

int gpx_engine_counters(gpx_engine* h, uint64_t out[3]) {
  if (!h || !out) return GPX_EINVAL;
  HIPCHK(hipStreamSynchronize(h->sF));
  HIPCHK(hipStreamSynchronize(h->sB));
  unsigned long long tmp[3];
  HIPCHK(hipMemcpy(tmp, h->X.counters, sizeof(tmp), hipMemcpyDeviceToHost));
  for (int i = 0; i < 3; i++) out[i] = tmp[i];
  return GPX_OK;
}

int gpx_engine_path_counters(gpx_engine* h, uint64_t out[2]) {
  if (!h || !out) return GPX_EINVAL;
  out[0] = out[1] = 0;
  if (!h->tile_area.ref) return GPX_OK; /* no call has gone through the tiled front end yet */
  HIPCHK(hipStreamSynchronize(h->sF));
  HIPCHK(hipStreamSynchronize(h->sB));
  int32_t tmp[2];
  HIPCHK(hipMemcpy(tmp, h->tile_area.ref + 5, sizeof(tmp), hipMemcpyDeviceToHost));
  out[0] = (uint64_t)(uint32_t)tmp[0];
  out[1] = (uint64_t)(uint32_t)tmp[1];
  return GPX_OK;
}

int gpx_profile_enable(gpx_engine* h, int32_t enable) {
  if (!h) return GPX_EINVAL;
  int rc = flush_profile(h);
  if (rc != GPX_OK) return rc;
  h->profiling = enable != 0;
  if (enable == 2) h->prof.clear(); /* 2 = enable and reset */
  return GPX_OK;
}

int gpx_profile_read(gpx_engine* h, gpx_kernel_stat* out, int32_t cap) {
  if (!h) return GPX_EINVAL;
  int rc = flush_profile(h);
  if (rc != GPX_OK) return rc;
  int32_t i = 0;
  for (auto& kv : h->prof) {
    if (i < cap && out) {
      memset(&out[i], 0, sizeof(out[i]));
      strncpy(out[i].name, kv.first.c_str(), sizeof(out[i].name) - 1);
      out[i].launches = kv.second.first;
      out[i].total_ms = kv.second.second;
    }
    i++;
  }
  return i;
}

/* ---- sharding ----------------------------------------------------------------------- */

int gpx_route_batch_dev(gpx_engine* h, int32_t n, int32_t n_cols, const int32_t* const* cols,
                        const int32_t* g2l, int32_t n_groups_global, int32_t n_shards,
                        int32_t* const* out_cols, int32_t* shard_off) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (!cols || !out_cols || !shard_off || n_cols < 1 || n_cols > GPX_ROUTE_MAX_COLS || n_shards < 1 ||
      n_shards > GPX_ROUTE_MAX_SHARDS || n_groups_global <= 0)
    return GPX_EINVAL;
  gpx_engine* e = h;
  e->stream = e->sB;
  if (!e->route_cnt) {
    const size_t tiles = ((size_t)e->cfg.max_batch + GPX_ROUTE_TILE - 1) / GPX_ROUTE_TILE;
    rc = dev_alloc(e, &e->route_cnt, tiles * GPX_ROUTE_MAX_SHARDS, false);
    if (rc != GPX_OK) return rc;
  }
  RouteCols C{};
  C.ncols = n_cols;
  for (int k = 0; k < n_cols; k++) {
    if (!cols[k] || !out_cols[k]) return GPX_EINVAL;
    C.in[k] = cols[k];
    C.out[k] = out_cols[k];
  }
  const int ntiles = (n + GPX_ROUTE_TILE - 1) / GPX_ROUTE_TILE;
  if (n == 0) {
    HIPCHK(hipMemsetAsync(shard_off, 0, sizeof(int32_t) * (size_t)(n_shards + 1), e->stream));
    return GPX_OK;
  }
  {
    LaunchScope _ls(e, "k_route_count");
    hipLaunchKernelGGL(k_route_count, dim3(ntiles), dim3(GPX_ROUTE_NT), 0, e->stream, n, cols[0], n_groups_global,
                       n_shards, e->route_cnt);
  }
  {
    LaunchScope _ls(e, "k_route_offsets");
    hipLaunchKernelGGL(k_route_offsets, dim3(1), dim3(GPX_ROUTE_NT), 0, e->stream, ntiles, n_shards, e->route_cnt, shard_off);
  }
  {
    LaunchScope _ls(e, "k_route_scatter");
    hipLaunchKernelGGL(k_route_scatter, dim3(ntiles), dim3(GPX_ROUTE_NT), 0, e->stream, n, n_groups_global, n_shards,
                       C, g2l, (const int32_t*)e->route_cnt, (const int32_t*)shard_off);
  }
  HIPCHK(hipGetLastError());
  return GPX_OK;
}

/* ---- device-pointer data path ------------------------------------------------- */

/* The tiled front end (gpx_tiles.hip.h) for one pass over every bucket: k_scatter_tiles, k_bucket_ar16_tiles,
 * k_emit_dec16.  false: this call's shape is not one it takes (the caller goes on with k_hist + k_scatter_ar16). */
struct TileShape {
  int32_t T, NT; /* votes and threads per scatter workgroup */
};
#ifdef GPX_TL_TRACE
static unsigned long long* tl_trace_dev = nullptr;
static void tl_trace_begin(gpx_engine* e) {
  if (!tl_trace_dev) {
    HIPQ(hipMalloc(&tl_trace_dev, TL_TRACE_ROWS * 8 * sizeof(unsigned long long)));
    HIPQ(hipMemcpyToSymbol(HIP_SYMBOL(g_tl_trace), &tl_trace_dev, sizeof(tl_trace_dev)));
  }
  HIPQ(hipMemsetAsync(tl_trace_dev, 0, TL_TRACE_ROWS * 8 * sizeof(unsigned long long), e->stream));
}
static void tl_trace_end(gpx_engine* e) { /* every call: the last one wins */
  const char* tf = getenv("GPX_TL_TRACE_FILE");
  if (!tf) return;
  HIPQ(hipStreamSynchronize(e->stream));
  std::vector<unsigned long long> tr((size_t)TL_TRACE_ROWS * 8);
  HIPQ(hipMemcpy(tr.data(), tl_trace_dev, tr.size() * 8, hipMemcpyDeviceToHost));
  if (FILE* fp = fopen(tf, "wb")) {
    fwrite(tr.data(), 8, tr.size(), fp);
    fclose(fp);
  }
}
#endif
/* Votes per scatter workgroup.  A workgroup's LDS holds its whole tile (8 bytes per vote + the bucket counters), the
 * kernel lasts as long as its busiest CU - workgroups go round the CUs, and two of them on one CU take twice as long as
 * one: they run in phase -, and every tile costs the per-bucket kernel a run to look up.  The model is the measured time
 * of one round of tiles per shape (profiles/r06_tile_shapes.txt: 27.9 / 17.3 / 10.0 us for 12,288 / 8,192 / 4,096 votes)
 * times the rounds, plus 5 ns per tile for the per-bucket kernel (733 tiles instead of 245: +2.3 us).  It picks what was
 * measured best on every shape of that file: 3 M votes - 245 x 12,288 (scatter 27.9 us against 33.2 for 367 x 8,192: the
 * second round), 5 M votes - 611 x 8,192 (51.8 against 55.9 for 407 x 12,288: three short rounds against two long ones),
 * 1.5 M votes - 184 x 8,192, 625,000 votes - 153 x 4,096. */
static TileShape tile_shape(const gpx_engine* e, int32_t n, int32_t nbk) {
  if (e->tile_force) return TileShape{e->tile_force, e->tile_threads ? e->tile_threads : (e->tile_force <= 8192 ? 512 : 1024)};
  const int64_t cus = std::max(1, e->cus);
  /* (16,384-vote tiles of 1024 threads spill registers - kept as a forced shape only) */
  static const struct { TileShape s; int64_t round_ns; } cand[] = {{{12288, 1024}, 27900}, {{8192, 1024}, 17300}, {{4096, 512}, 10000}};
  TileShape best = cand[2].s;
  int64_t best_cost = INT64_MAX;
  for (const auto& c : cand) {
    if (GPX_TL_LDS_BYTES(nbk, c.s.T, c.s.NT) > (size_t)158 * 1024) continue;
    const int64_t nwg = ((int64_t)n + c.s.T - 1) / c.s.T;
    if (nwg > GPX_TL_MAXWG) continue;
    const int64_t cost = (nwg + cus - 1) / cus * c.round_ns + nwg * 5;
    if (cost < best_cost) best_cost = cost, best = c.s;
  }
  return best;
}
static bool ar_tiles_call(gpx_engine* e, int32_t n, const int32_t* gidx, const int32_t* bnum, const int32_t* bcoord,
                          const int32_t* slot, const int32_t* acceptor, const int32_t* max_cp, int32_t* d_gidx,
                          int32_t* d_slot, int32_t* d_bnum, int32_t* d_bcoord, int32_t* d_median_cp, uint8_t* d_kind,
                          int32_t* n_out, uint8_t* status) {
  const int32_t nbk = e->nbk16;
  if (nbk > GPX_MAX_BUCKETS || e->shift16 > 10 || e->shift16 < 8) return false;
  const TileShape ts = tile_shape(e, n, nbk);
  const int32_t T = ts.T;
  if ((T != 4096 && T != 8192 && T != 12288 && T != 16384) || (ts.NT != 512 && ts.NT != 1024) || T % (ts.NT * 4)) return false;
  const int32_t nwg = (n + T - 1) / T;
  if (nwg > GPX_TL_MAXWG) return false;
  const size_t N = (size_t)e->cfg.max_batch;
  TileArea& A = e->tile_area;
  if (!A.recs) {
    /* every tile of the largest call, whole: max_batch votes in the smallest tiles, rounded up to tiles */
    const size_t cap = (N + 4095) / 4096 * 4096 + 16384;
    const size_t nwg_max = std::min<size_t>((N + 4095) / 4096, GPX_TL_MAXWG);
    const size_t pad = (nwg_max + 7) / 8 * 8; /* rows of A.off (8 bytes per tile) start on 64-byte lines */
    if (dev_alloc(e, &A.recs, cap, false) != GPX_OK || dev_alloc(e, &A.ext, cap, false) != GPX_OK ||
        dev_alloc(e, &A.off, ((size_t)nbk / 4 + 2) * pad * 4, true) != GPX_OK || dev_alloc(e, &A.ref, 8, true) != GPX_OK) {
      A.recs = nullptr;
      e->ar_tiles = false; /* no room: the partition front end from now on */
      return false;
    }
    A.nwg_pad = (int32_t)pad;
    const int maxlds = (int)std::min<size_t>(GPX_TL_LDS_BYTES(nbk, 16384, 1024), (size_t)158 * 1024);
    HIPQ(hipFuncSetAttribute((const void*)k_scatter_tiles<1024, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    HIPQ(hipFuncSetAttribute((const void*)k_scatter_tiles<1024, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    HIPQ(hipFuncSetAttribute((const void*)k_scatter_tiles<1024, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    HIPQ(hipFuncSetAttribute((const void*)k_scatter_tiles<1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    HIPQ(hipFuncSetAttribute((const void*)k_scatter_tiles<512, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    HIPQ(hipFuncSetAttribute((const void*)k_scatter_tiles<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, maxlds));
    /* (the per-bucket kernels keep 6.2 KB of static LDS - the tiles' run starts and prefix) */
    const size_t hw16 = std::min<size_t>(GPX_BUCKET16_LDS_BYTES((size_t)1 << e->shift16, e->lds16_hw) + e->lds_pad,
                                         (size_t)GPX_TL_BUCKET_DYN_MAX);
    HIPQ(hipFuncSetAttribute((const void*)k_bucket_ar16_tiles_k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hw16));
    HIPQ(hipFuncSetAttribute((const void*)k_bucket_ar16_tiles_k5, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hw16));
    HIPQ(hipFuncSetAttribute((const void*)k_bucket_ar16_tiles<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hw16));
    HIPQ(hipFuncSetAttribute((const void*)k_bucket_ar16_tiles<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hw16));
  }
  A.nwg = nwg;
  A.tile = T;
  const DevScratch X0 = e->X;
  const int threads0 = e->bucket_threads;
  e->X.shift = e->shift16;
  e->X.gb = 1 << e->shift16;
  e->X.nbk = nbk;
  e->bucket_threads = e->X.gb;
  begin_back(e, 0, n, true);
  { /* room for the per-bucket kernel's slotted placement (gpx_ar16.hip.h): a row of the staging arrays per vote rank */
    const int32_t kslot = e->cfg.kmax <= 4 ? 4 : 8;
    if ((int64_t)n / nbk >= e->X.gb / 4 && e->X.lds_recs < kslot * e->X.gb) {
      e->X.lds_recs = kslot * e->X.gb;
      e->bucket_lds = GPX_BUCKET16_LDS_BYTES(e->X.gb, e->X.lds_recs) + e->lds_pad;
    }
  }
  if (e->bucket_lds > (size_t)GPX_TL_BUCKET_DYN_MAX) { /* the static LDS must fit beside the staging */
    e->X.lds_recs = (int32_t)(((size_t)GPX_TL_BUCKET_DYN_MAX - e->lds_pad - (size_t)e->X.gb * 8) / 16);
    e->bucket_lds = GPX_BUCKET16_LDS_BYTES(e->X.gb, e->X.lds_recs) + e->lds_pad;
  }
  const size_t lds = GPX_TL_LDS_BYTES(nbk, T, ts.NT);
#ifdef GPX_TL_TRACE
  tl_trace_begin(e);
#endif
  {
    LaunchScope _ls(e, "k_scatter_tiles");
#define GPX_LAUNCH_TILES(NT_, R4_)                                                                                       \
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scatter_tiles<NT_, R4_>), dim3(tile_grid(A.nwg)), dim3(NT_), lds, e->stream, n, e->S.G, \
                     e->X, A, gidx, bnum, bcoord, slot, acceptor, max_cp, status)
    const int r4 = T / (ts.NT * 4);
    if (ts.NT == 1024 && r4 == 4) GPX_LAUNCH_TILES(1024, 4);
    else if (ts.NT == 1024 && r4 == 3) GPX_LAUNCH_TILES(1024, 3);
    else if (ts.NT == 1024 && r4 == 2) GPX_LAUNCH_TILES(1024, 2);
    else if (ts.NT == 1024 && r4 == 1) GPX_LAUNCH_TILES(1024, 1);
    else if (ts.NT == 512 && r4 == 4) GPX_LAUNCH_TILES(512, 4);
    else GPX_LAUNCH_TILES(512, 2);
#undef GPX_LAUNCH_TILES
  }
  const Stage16 O{(int32_t*)e->X.o_rec, (int64_t)N};
  const VoteCols in{bnum, bcoord, acceptor, slot, max_cp};
  /* votes per output the per-bucket kernel predicts its place with: what the last compacted call left in the
   * host-mapped word (read without waiting: stale is fine), the replica count until then - every replica answers */
  int32_t ip_div = 0;
  if (e->ar_in_place && n < (1 << 26)) {
    const uint32_t learnt = e->h_abort ? ((volatile uint32_t*)e->h_abort)[GPX_IP_LEARN_WORD] : 0u;
    ip_div = learnt >= 1 && learnt <= 64 ? (int32_t)learnt : std::max(1, e->cfg.kmax);
  }
  const PlaceCols IP{d_gidx, d_slot, d_bnum, d_bcoord, d_median_cp, d_kind, ip_div,
                     ip_div > 1 ? (uint32_t)((1ull << 32) / (uint64_t)ip_div) + 1u : 0u};
  {
    LaunchScope _ls(e, e->cfg.kmax <= 4 ? "k_bucket_ar16_tiles_k4" : e->cfg.kmax <= 5 ? "k_bucket_ar16_tiles_k5" : "k_bucket_ar16_tiles");
    const dim3 grid(A.xcd_rows ? tile_grid(nbk) : nbk), block(e->bucket_threads);
    if (e->cfg.kmax <= 4)
      hipLaunchKernelGGL(k_bucket_ar16_tiles_k4, grid, block, e->bucket_lds, e->stream, e->S, e->X, O, in, status, A, IP);
    else if (e->cfg.kmax <= 5)
      hipLaunchKernelGGL(k_bucket_ar16_tiles_k5, grid, block, e->bucket_lds, e->stream, e->S, e->X, O, in, status, A, IP);
    else if (e->cfg.kmax <= 8)
      hipLaunchKernelGGL(k_bucket_ar16_tiles<8>, grid, block, e->bucket_lds, e->stream, e->S, e->X, O, in, status, A, IP);
    else
      hipLaunchKernelGGL(k_bucket_ar16_tiles<16>, grid, block, e->bucket_lds, e->stream, e->S, e->X, O, in, status, A, IP);
  }
  LAUNCH(e, "k_emit_dec16", k_emit_dec16, e->X.nbk, e->X, O, d_gidx, d_slot, d_bnum, d_bcoord, d_median_cp, d_kind, n_out,
         &e->X.counters[1], (const int32_t*)nullptr, (int32_t*)nullptr, ip_div ? A.ref : (int32_t*)nullptr, n,
         ip_div ? (const int32_t*)e->S.c_bnum : (const int32_t*)nullptr, (const int32_t*)e->S.c_bcoord);
#ifdef GPX_TL_TRACE
  tl_trace_end(e);
#endif
  const int32_t lds_recs = e->X.lds_recs;
  const int32_t gate = e->X.gate;
  e->X = X0;
  e->X.lds_recs = lds_recs;
  e->X.gate = gate;
  e->bucket_threads = threads0;
  return true;
}

/* the partition pipeline of an accept-reply batch: k_hist, k_scatter_ar16, k_bucket_ar16, k_emit_dec16, one
 * pass per range of groups; e->ar_chain holds the number of outputs written so far between the passes */
static void ar_partition(gpx_engine* e, int32_t n, const int32_t* gidx, const int32_t* bnum, const int32_t* bcoord,
                         const int32_t* slot, const int32_t* acceptor, const int32_t* max_cp, int32_t* d_gidx,
                         int32_t* d_slot, int32_t* d_bnum, int32_t* d_bcoord, int32_t* d_median_cp,
                         uint8_t* d_kind, int32_t* n_out, uint8_t* status) {
  const int ntiles = ntiles_for(n);
  const bool vec = aligned16({gidx, bnum, bcoord, slot, acceptor, max_cp});
  /* 16-byte vote records (gpx_ar16.hip.h); the back end may re-read bnum / bcoord / acceptor.
   * One pass per range of at most 4 M groups (ranges ascending, so the concatenated outputs stay
   * grouped by gidx ascending); a table of up to 4 M groups is one pass over everything. */
  const DevScratch X0 = e->X;
  const int threads0 = e->bucket_threads;
  /* passes over ascending group ranges: as many as the table needs (at most 4096 buckets of at most 1024
   * groups per pass) and as many as keep a bucket's expected votes inside what one workgroup can stage in
   * LDS - a call that brings many rounds of votes at once (more than ~19 per group) would otherwise regroup
   * every bucket in global memory, an order of magnitude slower than reading the columns once more */
  const int64_t NB = ((int64_t)e->S.G + ((int64_t)1 << e->shift16) - 1) >> e->shift16;
  int64_t want_passes = e->ar_passes;
  {
    const double m = (double)n / (double)NB;
    const int64_t pc = (int64_t)((m + 5.0 * sqrt(m)) / (double)e->lds16_hw) + 1;
    want_passes = std::min<int64_t>(NB, std::max<int64_t>(want_passes, pc));
  }
  const int64_t bpp = (NB + want_passes - 1) / want_passes; /* buckets per pass, <= GPX_MAX_BUCKETS */
  const int32_t passes = (int32_t)((NB + bpp - 1) / bpp);
  if (e->ar_tiles && passes == 1 && vec && ar_tiles_call(e, n, gidx, bnum, bcoord, slot, acceptor, max_cp, d_gidx, d_slot, d_bnum,
                                                        d_bcoord, d_median_cp, d_kind, n_out, status))
    return;
  const size_t N = (size_t)e->cfg.max_batch;
  int32_t* o32 = (int32_t*)e->X.o_rec; /* N x 32 bytes: five int columns + one byte column */
  const Stage16 O{o32, (int64_t)N};
  const VoteCols in{bnum, bcoord, acceptor};
  for (int32_t p = 0; p < passes; p++) {
    e->X.shift = e->shift16;
    e->X.gb = 1 << e->shift16;
    e->X.nbk = e->nbk16;
    e->bucket_threads = e->X.gb;
    if (passes > 1) {
      const int64_t range = bpp << e->shift16;
      e->X.g_base = (int32_t)(p * range);
      e->X.g_end = (int32_t)std::min<int64_t>(e->S.G, (p + 1) * range);
      e->X.nbk = (int32_t)((e->X.g_end - e->X.g_base + e->X.gb - 1) >> e->shift16);
    }
    /* LDS staging sized for what one pass sees */
    if (p == 0) begin_back(e, 0, n / passes, true);
    /* status prefill, vote and out-of-table counters: once, by the first pass */
    front_hist(e, n, gidx, p == 0 ? status : nullptr, p == 0 ? 1 : -1, e->X.gate ? 2 : 0);
    if (vec)
      LAUNCH_F(e, "k_scatter_ar16", k_scatter_ar16<true>, tile_grid(ntiles), (size_t)e->X.nbk * 4, n,
               ntiles, e->S.G, e->X, gidx, bnum, bcoord, slot, acceptor, max_cp);
    else
      LAUNCH_F(e, "k_scatter_ar16", k_scatter_ar16<false>, tile_grid(ntiles), (size_t)e->X.nbk * 4, n,
               ntiles, e->S.G, e->X, gidx, bnum, bcoord, slot, acceptor, max_cp);
    if (e->cfg.kmax <= 4)
      launch_bucket_ar16<4>(e, O, in, status);
    else if (e->cfg.kmax <= 5) /* five replicas (BASELINE config #4): its own kernel, held to 80 VGPRs */
      LAUNCH_B(e, "k_bucket_ar16", k_bucket_ar16_k5, e->S, e->X, O, in, AcceptOut{}, status);
    else if (e->cfg.kmax <= 8)
      launch_bucket_ar16<8>(e, O, in, status);
    else
      launch_bucket_ar16<16>(e, O, in, status);
    LAUNCH(e, "k_emit_dec16", k_emit_dec16, e->X.nbk, e->X, O, d_gidx, d_slot, d_bnum, d_bcoord,
           d_median_cp, d_kind, n_out, &e->X.counters[1],
           (const int32_t*)(p > 0 ? e->ar_chain + (p & 1) : nullptr),
           passes > 1 ? e->ar_chain + ((p + 1) & 1) : (int32_t*)nullptr);
    /* what begin_back computed for this call survives the restore below */
    const int32_t lds_recs = e->X.lds_recs;
    const int32_t gate = e->X.gate;
    e->X = X0;
    e->X.lds_recs = lds_recs;
    e->X.gate = gate;
    e->bucket_threads = threads0;
  }
}

#ifdef GPX_SAR_TRACE
/* timeline build: the stamps of the call's workgroups (256 x 16 words) cleared before the launch, written to
 * GPX_SAR_TRACE_FILE after it (every call: the last one wins) */
static unsigned long long* sar_trace_dev = nullptr;
static void sar_trace_begin(gpx_engine* e) {
  if (!sar_trace_dev) {
    HIPQ(hipMalloc(&sar_trace_dev, 256 * 16 * sizeof(unsigned long long)));
    HIPQ(hipMemcpyToSymbol(HIP_SYMBOL(g_sar_trace), &sar_trace_dev, sizeof(sar_trace_dev)));
  }
  HIPQ(hipMemsetAsync(sar_trace_dev, 0, 256 * 16 * sizeof(unsigned long long), e->stream));
}
static void sar_trace_end(gpx_engine* e) {
  const char* tf = getenv("GPX_SAR_TRACE_FILE");
  if (!tf) return;
  HIPQ(hipStreamSynchronize(e->stream));
  std::vector<unsigned long long> tr(256 * 16);
  HIPQ(hipMemcpy(tr.data(), sar_trace_dev, tr.size() * 8, hipMemcpyDeviceToHost));
  if (FILE* fp = fopen(tf, "wb")) {
    fwrite(tr.data(), 8, tr.size(), fp);
    fclose(fp);
  }
}
#endif

int gpx_accept_reply_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx,
                               const int32_t* bnum, const int32_t* bcoord, const int32_t* slot,
                               const int32_t* acceptor, const int32_t* max_cp, int32_t* d_gidx,
                               int32_t* d_slot, int32_t* d_bnum, int32_t* d_bcoord,
                               int32_t* d_median_cp, uint8_t* d_kind, int32_t* n_out,
                               uint8_t* status) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (n == 0) {
    HIPCHK(hipMemsetAsync(n_out, 0, sizeof(int32_t), h->sB));
    return GPX_OK;
  }
  gpx_engine* e = h;
  const int fs = begin_front(e);
  e->X.gate = 0;
  /* (i) a few sorted runs - the concatenated replies of the acceptors (gpx_runs.hip.h): no partition.
   * GPX_ORDERED_REPLY_RUNS: the caller promises that shape, only this path is launched and a batch that
   * breaks the promise is refused whole; GPX_TRY_REPLY_RUNS: a hint - the shape is checked on the device and
   * the partition pipeline, launched behind, takes any other batch (its kernels return at once otherwise). */
  const bool runs_promised = (e->ordered_mask & GPX_ORDERED_REPLY_RUNS) != 0;
  const bool runs_try = runs_promised || (e->ordered_mask & GPX_TRY_REPLY_RUNS) != 0;
  /* (0) a tiny call, whatever its order: ONE launch of one workgroup (gpx_small.hip.h).  Not under the
   * GPX_ORDERED_REPLY_RUNS promise, whose breach must be refused: this kernel would apply such a batch. */
  if (!runs_promised && n <= e->sar_max_n && e->S.G <= GPX_SAR_MAX_G) {
    e->last.kind = 0; /* dense outputs, always */
#ifdef GPX_SAR_TRACE
    sar_trace_begin(e);
#endif
    {
      LaunchScope _ls(e, "k_ar_tiny");
#define GPX_LAUNCH_AR_TINY(KM)                                                                                          \
  hipLaunchKernelGGL(k_ar_tiny<KM>, dim3(1), dim3(GPX_SAR_BLOCK), GPX_SAR_LDS_BYTES, e->stream, e->S, e->X, n, gidx, bnum, \
                     bcoord, slot, acceptor, max_cp, d_gidx, d_slot, d_bnum, d_bcoord, d_median_cp, d_kind, n_out, status)
      if (e->cfg.kmax <= 4)
        GPX_LAUNCH_AR_TINY(4);
      else if (e->cfg.kmax <= 8)
        GPX_LAUNCH_AR_TINY(8);
      else
        GPX_LAUNCH_AR_TINY(16);
#undef GPX_LAUNCH_AR_TINY
    }
#ifdef GPX_SAR_TRACE
    sar_trace_end(e);
#endif
    end_call(e, fs);
    HIPCHK(hipGetLastError());
    return GPX_OK;
  }
  if (runs_try) {
    const size_t N = (size_t)e->cfg.max_batch;
    if (!e->runs_info && (rc = dev_alloc(e, &e->runs_info, 2, true)) != GPX_OK) return rc;
    RunsInfo* info = e->runs_info + (e->runs_seq & 1);
    RunsInfo* next_info = e->runs_info + ((e->runs_seq + 1) & 1);
    e->runs_seq++;
    const int nchunks = (n + GPX_DCHUNK - 1) / GPX_DCHUNK;
    /* outputs are parked in the caller's own columns (n entries each: gpx.h); the staging block is only used
     * by batches that are not REGULAR (gpx_runs.hip.h) */
    const RunsStage st{DecCols{d_gidx, d_slot, d_bnum, d_bcoord, d_median_cp, d_kind}, e->rec_tag, e->fs[0].chunk_cnt,
                       Stage16{(int32_t*)e->X.o_rec, (int64_t)N}};
    const int32_t refuse = runs_promised ? 1 : 0;
    if (!e->runs_arrive &&
        (rc = dev_alloc(e, &e->runs_arrive,
                        /* k_runs_check: one workgroup per 4,096 records; k_ar_runs<.., SMALL>: its end-of-kernel counters */
                        (size_t)32 * (3 + N / GPX_RBLOCK / 16 + 1),
                        true)) != GPX_OK)
      return rc;
    /* ONE launch where the grid is resident for sure (xchg_ctl): every workgroup judges its own records and all of them
     * meet once (gpx_one.hip.h); else the check kernel and the work kernel */
    const int nch = (n + GPX_RBLOCK - 1) / GPX_RBLOCK;
    int pg = nch;
    GridXchg Q{};
    const void* runs_kernel =
        e->cfg.kmax <= 4 ? (n <= 65536 ? (const void*)k_ar_runs<4, true, true> : (const void*)k_ar_runs<4, true, false>)
        : e->cfg.kmax <= 8 ? (n <= 65536 ? (const void*)k_ar_runs<8, true, true> : (const void*)k_ar_runs<8, true, false>)
                           : (n <= 65536 ? (const void*)k_ar_runs<16, true, true> : (const void*)k_ar_runs<16, true, false>);
    const bool small = nch <= e->pers_max_chunks[2] && xchg_ctl(e, nch, &Q, &pg, runs_kernel, GPX_RBLOCK);
    if (!small)
      LAUNCH_OC(e, "k_runs_check", k_runs_check, (n + GPX_OC_BLOCK * GPX_RC_ITEMS - 1) / (GPX_OC_BLOCK * GPX_RC_ITEMS), 0, n,
                gidx, e->S.G, e->X, status, info, next_info, st.chunk_cnt, nchunks, e->runs_arrive, n_out, &e->X.counters[1],
                refuse);
#ifdef GPX_SAR_TRACE
    if (small) sar_trace_begin(e);
#endif
    {
      LaunchScope _ls(e, small ? "k_ar_runs_pers" : "k_ar_runs");
      const dim3 grid(small ? pg : nch);
#define GPX_LAUNCH_AR_RUNS(KM)                                                                                              \
  do {                                                                                                                      \
    if (small && n <= 65536) /* a chain of round trips, not bytes: the first chunk's state requested before the exchange */ \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ar_runs<KM, true, true>), grid, dim3(GPX_RBLOCK), 0, e->stream, e->S, e->X, n, gidx, \
                         bnum, bcoord, slot, acceptor, max_cp, status, st, info, refuse, n_out, next_info, e->runs_arrive,  \
                         &e->X.counters[1], Q);                                                                             \
    else if (small)                                                                                                         \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ar_runs<KM, true>), grid, dim3(GPX_RBLOCK), 0, e->stream, e->S, e->X, n, gidx, bnum, \
                         bcoord, slot, acceptor, max_cp, status, st, info, refuse, n_out, next_info, e->runs_arrive,        \
                         &e->X.counters[1], Q);                                                                             \
    else                                                                                                                    \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(k_ar_runs<KM, false>), grid, dim3(GPX_RBLOCK), 0, e->stream, e->S, e->X, n, gidx, bnum, \
                         bcoord, slot, acceptor, max_cp, status, st, info, refuse, n_out, (RunsInfo*)nullptr,               \
                         (uint32_t*)nullptr, (unsigned long long*)nullptr, Q);                                              \
  } while (0)
      if (e->cfg.kmax <= 4)
        GPX_LAUNCH_AR_RUNS(4);
      else if (e->cfg.kmax <= 8)
        GPX_LAUNCH_AR_RUNS(8);
      else
        GPX_LAUNCH_AR_RUNS(16);
#undef GPX_LAUNCH_AR_RUNS
    }
#ifdef GPX_SAR_TRACE
    if (small && runs_promised) sar_trace_end(e);
#endif
    /* a REGULAR batch is finished: k_ar_runs' last workgroup has published its count.  The compaction pass of any
     * other batch follows at once - or, under the promise with GPX_LAZY_OUTPUTS, when the caller asks for it */
    e->last.kind = 0;
    if (runs_promised && lazy_outputs(e)) {
      gpx_engine::LastCall& L = e->last;
      L.kind = small ? 4 : 3, L.n = n, L.nchunks = nchunks, L.X = e->X, L.rs = st, L.info = info, L.count = n_out;
      L.seq = e->call_seq + 1, L.stale = false;
    } else {
      if (small) {
        LaunchScope _ls(e, "k_runs_count");
        hipLaunchKernelGGL(k_runs_count, dim3(nchunks), dim3(GPX_DCHUNK), 0, e->stream, e->X, n, st, (const RunsInfo*)info);
      }
      {
        LaunchScope _ls(e, "k_emit_dec_runs");
        hipLaunchKernelGGL(k_emit_dec_runs, dim3(std::min(nchunks, GPX_EMIT_GRID)), dim3(GPX_DCHUNK), 0, e->stream, e->X, n, nchunks, st, info, n_out,
                           &e->X.counters[1], refuse, 1);
      }
      LAUNCH(e, "k_merge_runs", k_merge_runs, 256, e->X, n, st, (const RunsInfo*)info);
    }
    if (runs_promised) {
      end_call(e, fs);
      HIPCHK(hipGetLastError());
      return GPX_OK;
    }
    e->X.gate = 1; /* the partition kernels below run only if k_runs_check raised *X.unsorted */
  }
  {
    /* (iii) the partition pipeline */
    ar_partition(e, n, gidx, bnum, bcoord, slot, acceptor, max_cp, d_gidx, d_slot, d_bnum, d_bcoord, d_median_cp,
                 d_kind, n_out, status);
  }
  e->X.gate = 0;
  end_call(e, fs);
  HIPCHK(hipGetLastError());
  return GPX_OK;
}

int gpx_accept_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                         const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                         const uint8_t* a_flags, int32_t* r_bnum, int32_t* r_bcoord,
                         int32_t* r_maxcp, uint8_t* r_flags, uint8_t* status, int32_t* x_gidx,
                         int32_t* x_first, int32_t* x_count, int32_t* n_runs) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (n == 0) {
    HIPCHK(hipMemsetAsync(n_runs, 0, sizeof(int32_t), h->sB));
    return GPX_OK;
  }
  gpx_engine* e = h;
  const int fs = begin_front(e);
  /* a batch grouped by group (gidx non-decreasing, in range) is applied directly; anything else is
   * partitioned first.  Device-side choice: both back ends are launched, one of them returns at once. */
  const int nchunks = (n + GPX_DCHUNK - 1) / GPX_DCHUNK;
  int32_t* st32 = (int32_t*)e->X.o_rec; /* the two paths never both stage outputs: shared scratch */
  const DirectStage D{st32, st32 + (size_t)e->cfg.max_batch, e->rec_tag, e->fs[fs].chunk_cnt, x_gidx, x_first, x_count,
                      st32 + 2 * (size_t)e->cfg.max_batch, e->fs[fs].chunk_cnt + GPX_CHUNK_CNT_TOTAL(e->cfg.max_batch),
                      (uint32_t*)(e->fs[fs].chunk_cnt + GPX_CHUNK_CNT_MARK(e->cfg.max_batch))};
  /* at most 65,536 records on one stream: order check, direct application and run compaction in ONE
   * launch (k_ac_small: tickets instead of chunk counters) */
  const bool fused = n <= GPX_SMALL_DIRECT_MAX_N;
  const bool promised = (e->ordered_mask & GPX_ORDERED_ACCEPT) != 0;
  e->last.kind = 0;
  if (promised && (!fused || lazy_outputs(e))) {
    /* the promise: the work kernel alone where its grid is resident for sure - its workgroups exchange the verdict among
     * themselves (k_ac_pers, xchg_ctl: 131,072 records for one engine on an MI355X) - else the verdict (k_one_check,
     * which also writes the usual batch's count) and ONE work kernel (gpx_one.hip.h); the compaction pass follows unless
     * the caller asked for lazy outputs.  At most 65,536 records without lazy outputs: k_ac_small's in-kernel run
     * compaction is the one launch */
    const int nch = (n + GPX_DBLOCK - 1) / GPX_DBLOCK;
    GridXchg Q;
    int pg;
    if (nch <= e->pers_max_chunks[1] && xchg_ctl(e, nch, &Q, &pg, (const void*)k_ac_pers<false>, GPX_DBLOCK)) { /* ONE launch: a grid that is resident for sure, the verdict exchanged among its workgroups */
      LaunchScope _ls(e, "k_ac_pers");
      hipLaunchKernelGGL(k_ac_pers<false>, dim3(pg), dim3(GPX_DBLOCK), 0, e->stream, e->S, e->X, Q, n, gidx, bnum, bcoord, slot,
                         median_cp, a_flags, r_bnum, r_bcoord, r_maxcp, r_flags, status, D, n_runs, 0);
    } else {
      const OneCtl C = one_ctl(e);
      {
        LaunchScope _lc(e, "k_one_check");
        hipLaunchKernelGGL(k_one_check<false>, dim3((n + GPX_DBLOCK * 8 - 1) / (GPX_DBLOCK * 8)), dim3(GPX_DBLOCK), 0, e->stream, n, gidx,
                           e->S.G, C, n_runs, 0);
      }
      LaunchScope _ls(e, "k_ac_one");
      hipLaunchKernelGGL(k_ac_one<false>, dim3((n + GPX_DBLOCK - 1) / GPX_DBLOCK), dim3(GPX_DBLOCK), 0, e->stream, e->S, e->X, C, n,
                         gidx, bnum, bcoord, slot, median_cp, a_flags, r_bnum, r_bcoord, r_maxcp, r_flags, status, D, n_runs);
    }
    gpx_engine::LastCall& L = e->last;
    L.kind = 1, L.n = n, L.nchunks = nchunks, L.X = e->X, L.gidx = gidx, L.D = D;
    L.x_gidx = x_gidx, L.x_first = x_first, L.x_count = x_count, L.count = n_runs;
    L.seq = e->call_seq + 1, L.stale = false;
    if (!lazy_outputs(e)) {
      launch_one_compaction(e, L);
      L.kind = 0;
    }
    end_call(e, fs);
    HIPCHK(hipGetLastError());
    return GPX_OK;
  }
  if (!fused)
    LAUNCH_OC(e, "k_order_check", k_order_check<false>, (n + GPX_OC_BLOCK * 8 - 1) / (GPX_OC_BLOCK * 8), 0, n, gidx,
              e->S.G, e->X, status, D.chunk_cnt, nchunks);
  /* fused: FIRST in the stream - the partition path launched behind it reads its verdict */
  if (fused) {
    if (++e->small_epoch == 0) {
      HIPQ(hipMemsetAsync(e->small_tickets, 0, 2 * (GPX_SMALL_DIRECT_MAX_N / GPX_DCHUNK) * sizeof(unsigned long long), e->stream));
      e->small_epoch = 1;
    }
    LaunchScope _ls(e, "k_ac_small");
    hipLaunchKernelGGL(k_ac_small<false>, dim3(nchunks), dim3(GPX_DCHUNK), 0, e->stream, e->S, e->X, n, gidx, bnum,
                       bcoord, slot, median_cp, a_flags, r_bnum, r_bcoord, r_maxcp, r_flags, status, D, x_gidx,
                       x_first, x_count, n_runs, e->small_tickets, e->small_epoch, promised ? 1 : 0, e->small_draw,
                       e->small_drawn);
    e->small_drawn += (uint32_t)nchunks;
  }

  const size_t Nmax = (size_t)e->cfg.max_batch;
  if (!promised && e->ac16 && !e->reply_rows && (rc = dev_alloc(e, &e->reply_rows, Nmax, false)) != GPX_OK) return rc;
  const Stage16 O16{st32, (int64_t)Nmax};
  if (!promised) {
    front_hist(e, n, gidx, fused ? nullptr : status, 0, 2);
    if (e->ac16) {
      const int ntiles = ntiles_for(n);
      if (aligned16({gidx, bnum, bcoord, slot, median_cp}) && !((uintptr_t)a_flags & 3))
        LAUNCH_F(e, "k_scatter_ac16", k_scatter_ac16<true>, tile_grid(ntiles), (size_t)e->X.nbk * 4, n, ntiles, e->S.G,
                 e->X, gidx, bnum, bcoord, slot, median_cp, a_flags, r_bnum, r_bcoord, r_maxcp, r_flags,
                 e->reply_rows);
      else
        LAUNCH_F(e, "k_scatter_ac16", k_scatter_ac16<false>, tile_grid(ntiles), (size_t)e->X.nbk * 4, n, ntiles, e->S.G,
                 e->X, gidx, bnum, bcoord, slot, median_cp, a_flags, r_bnum, r_bcoord, r_maxcp, r_flags,
                 e->reply_rows);
    } else {
      launch_scatter_ac(e, n, gidx, bnum, bcoord, slot, median_cp, a_flags, r_bnum, r_bcoord, r_maxcp, r_flags, 1);
    }
  }
  begin_back(e, fs, n, e->ac16);
  if (!fused) {
    {
      LaunchScope _ls(e, "k_ac_direct");
      hipLaunchKernelGGL(k_ac_direct<false>, dim3((n + GPX_DBLOCK - 1) / GPX_DBLOCK), dim3(GPX_DBLOCK), 0, e->stream, e->S, e->X, n, gidx,
                         bnum, bcoord, slot, median_cp, a_flags, r_bnum, r_bcoord, r_maxcp, r_flags, status, D,
                         promised ? 1 : 0);
    }
    {
      LaunchScope _ls(e, "k_emit_runs_direct");
      hipLaunchKernelGGL(k_emit_runs_direct<false>, dim3(std::min(nchunks, GPX_EMIT_GRID)), dim3(GPX_DCHUNK), 0, e->stream, e->X, n, nchunks, gidx, D,
                         x_gidx, x_first, x_count, n_runs, promised ? 1 : 0);
    }
  }
  if (!promised && e->ac16) {
    /* unordered batch: 16-byte records through the partition (gpx_ar16.hip.h); the back end may
     * re-read bnum / bcoord of records in another ballot than the batch's first */
    LAUNCH_B(e, "k_bucket_accept16", (k_bucket16<B16_ACCEPT, 4>), e->S, e->X, O16, VoteCols{bnum, bcoord, nullptr},
             AcceptOut{r_bnum, r_bcoord, r_maxcp, r_flags, e->reply_rows}, status);
    LAUNCH(e, "k_unpack_replies", k_unpack_replies, grid_for(n), e->X, n, (const I4*)e->reply_rows, r_bnum, r_bcoord,
           r_maxcp, r_flags);
    LAUNCH(e, "k_emit_runs16", k_emit_runs16, e->X.nbk, e->X, O16, x_gidx, x_first, x_count, n_runs);
  } else if (!promised) {
    LAUNCH_B(e, "k_bucket_accept", k_bucket_accept, e->S, e->X, r_bnum, r_bcoord, r_maxcp, r_flags, status);
    LAUNCH(e, "k_emit_runs", k_emit_runs, e->X.nbk, e->X, x_gidx, x_first, x_count, n_runs);
  }
  end_call(e, fs);
  HIPCHK(hipGetLastError());
  return GPX_OK;
}

int gpx_commit_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                         const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                         const uint8_t* c_kind, uint8_t* status, int32_t* x_gidx,
                         int32_t* x_first, int32_t* x_count, int32_t* n_runs) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (n == 0) {
    HIPCHK(hipMemsetAsync(n_runs, 0, sizeof(int32_t), h->sB));
    return GPX_OK;
  }
  gpx_engine* e = h;
  const int fs = begin_front(e);
  /* decisions leave the accept-reply call grouped by gidx: such a commit batch is applied directly */
  const int nchunks = (n + GPX_DCHUNK - 1) / GPX_DCHUNK;
  int32_t* st32 = (int32_t*)e->X.o_rec;
  const DirectStage D{st32, st32 + (size_t)e->cfg.max_batch, e->rec_tag, e->fs[fs].chunk_cnt, x_gidx, x_first, x_count,
                      st32 + 2 * (size_t)e->cfg.max_batch, e->fs[fs].chunk_cnt + GPX_CHUNK_CNT_TOTAL(e->cfg.max_batch),
                      (uint32_t*)(e->fs[fs].chunk_cnt + GPX_CHUNK_CNT_MARK(e->cfg.max_batch))};
  const bool fused = n <= GPX_SMALL_DIRECT_MAX_N; /* one launch: k_ac_small */
  const bool promised = (e->ordered_mask & GPX_ORDERED_COMMIT) != 0;
  e->last.kind = 0;
  if (promised && (!fused || lazy_outputs(e))) { /* check + one work kernel (gpx_one.hip.h), like the ACCEPT call */
    const int nch = (n + GPX_DBLOCK - 1) / GPX_DBLOCK;
    GridXchg Q;
    int pg;
    if (nch <= e->pers_max_chunks[1] && xchg_ctl(e, nch, &Q, &pg, (const void*)k_ac_pers<true>, GPX_DBLOCK)) {
      LaunchScope _ls(e, "k_ac_pers");
      hipLaunchKernelGGL(k_ac_pers<true>, dim3(pg), dim3(GPX_DBLOCK), 0, e->stream, e->S, e->X, Q, n, gidx, bnum, bcoord, slot,
                         median_cp, c_kind, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (uint8_t*)nullptr, status, D,
                         n_runs, n);
    } else {
      const OneCtl C = one_ctl(e);
      {
        LaunchScope _lc(e, "k_one_check");
        hipLaunchKernelGGL(k_one_check<false>, dim3((n + GPX_DBLOCK * 8 - 1) / (GPX_DBLOCK * 8)), dim3(GPX_DBLOCK), 0, e->stream, n, gidx,
                           e->S.G, C, n_runs, n);
      }
      LaunchScope _ls(e, "k_ac_one");
      hipLaunchKernelGGL(k_ac_one<true>, dim3((n + GPX_DBLOCK - 1) / GPX_DBLOCK), dim3(GPX_DBLOCK), 0, e->stream, e->S, e->X, C, n,
                         gidx, bnum, bcoord, slot, median_cp, c_kind, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                         (uint8_t*)nullptr, status, D, n_runs);
    }
    gpx_engine::LastCall& L = e->last;
    L.kind = 2, L.n = n, L.nchunks = nchunks, L.X = e->X, L.gidx = gidx, L.D = D;
    L.x_gidx = x_gidx, L.x_first = x_first, L.x_count = x_count, L.count = n_runs;
    L.seq = e->call_seq + 1, L.stale = false;
    if (!lazy_outputs(e)) {
      launch_one_compaction(e, L);
      L.kind = 0;
    }
    end_call(e, fs);
    HIPCHK(hipGetLastError());
    return GPX_OK;
  }
  if (!fused)
    LAUNCH_OC(e, "k_order_check", k_order_check<false>, (n + GPX_OC_BLOCK * 8 - 1) / (GPX_OC_BLOCK * 8), 0, n, gidx,
              e->S.G, e->X, status, D.chunk_cnt, nchunks);
  if (fused) {
    if (++e->small_epoch == 0) {
      HIPQ(hipMemsetAsync(e->small_tickets, 0, 2 * (GPX_SMALL_DIRECT_MAX_N / GPX_DCHUNK) * sizeof(unsigned long long), e->stream));
      e->small_epoch = 1;
    }
    LaunchScope _ls(e, "k_ac_small");
    hipLaunchKernelGGL(k_ac_small<true>, dim3(nchunks), dim3(GPX_DCHUNK), 0, e->stream, e->S, e->X, n, gidx, bnum,
                       bcoord, slot, median_cp, c_kind, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                       (uint8_t*)nullptr, status, D, x_gidx, x_first, x_count, n_runs, e->small_tickets,
                       e->small_epoch, promised ? 1 : 0, e->small_draw, e->small_drawn);
    e->small_drawn += (uint32_t)nchunks;
  }

  const size_t Nmax = (size_t)e->cfg.max_batch;
  const Stage16 O16{st32, (int64_t)Nmax};
  if (!promised) {
    front_hist(e, n, gidx, fused ? nullptr : status, 0, 2);
    if (e->ac16) {
      const int ntiles = ntiles_for(n);
      if (aligned16({gidx, bnum, bcoord, slot, median_cp}) && !((uintptr_t)c_kind & 3))
        LAUNCH_F(e, "k_scatter_ac16", k_scatter_ac16<true>, tile_grid(ntiles), (size_t)e->X.nbk * 4, n, ntiles, e->S.G,
                 e->X, gidx, bnum, bcoord, slot, median_cp, c_kind, (int32_t*)nullptr, (int32_t*)nullptr,
                 (int32_t*)nullptr, (uint8_t*)nullptr, (I4*)nullptr);
      else
        LAUNCH_F(e, "k_scatter_ac16", k_scatter_ac16<false>, tile_grid(ntiles), (size_t)e->X.nbk * 4, n, ntiles, e->S.G,
                 e->X, gidx, bnum, bcoord, slot, median_cp, c_kind, (int32_t*)nullptr, (int32_t*)nullptr,
                 (int32_t*)nullptr, (uint8_t*)nullptr, (I4*)nullptr);
    } else {
      launch_scatter_ac(e, n, gidx, bnum, bcoord, slot, median_cp, c_kind, nullptr, nullptr, nullptr, nullptr, 1);
    }
  }
  begin_back(e, fs, n, e->ac16);
  if (!fused) {
    {
      LaunchScope _ls(e, "k_ac_direct");
      hipLaunchKernelGGL(k_ac_direct<true>, dim3((n + GPX_DBLOCK - 1) / GPX_DBLOCK), dim3(GPX_DBLOCK), 0, e->stream, e->S, e->X, n, gidx,
                         bnum, bcoord, slot, median_cp, c_kind, (int32_t*)nullptr, (int32_t*)nullptr,
                         (int32_t*)nullptr, (uint8_t*)nullptr, status, D, promised ? 1 : 0);
    }
    {
      LaunchScope _ls(e, "k_emit_runs_direct");
      hipLaunchKernelGGL(k_emit_runs_direct<true>, dim3(std::min(nchunks, GPX_EMIT_GRID)), dim3(GPX_DCHUNK), 0, e->stream, e->X, n, nchunks, gidx, D,
                         x_gidx, x_first, x_count, n_runs, promised ? 1 : 0);
    }
    LAUNCH(e, "k_copy_runs", k_copy_runs, 256, e->X, D, x_gidx, x_first, x_count);
  }
  if (!promised && e->ac16) {
    LAUNCH_B(e, "k_bucket_commit16", (k_bucket16<B16_COMMIT, 4>), e->S, e->X, O16, VoteCols{bnum, bcoord, nullptr},
             AcceptOut{}, status);
    LAUNCH(e, "k_emit_runs16", k_emit_runs16, e->X.nbk, e->X, O16, x_gidx, x_first, x_count, n_runs);
  } else if (!promised) {
    LAUNCH_B(e, "k_bucket_commit", k_bucket_commit, e->S, e->X, status);
    LAUNCH(e, "k_emit_runs", k_emit_runs, e->X.nbk, e->X, x_gidx, x_first, x_count, n_runs);
  }
  end_call(e, fs);
  HIPCHK(hipGetLastError());
  return GPX_OK;
}

int gpx_compact_last_dev(gpx_engine* h) {
  if (!h) return GPX_EINVAL;
  gpx_engine* e = h;
  gpx_engine::LastCall& L = e->last;
  if (L.stale || (L.kind && L.seq != e->call_seq)) { /* another batch call came in between: its scratch and columns are gone */
    L.kind = 0;
    L.stale = false;
    snprintf(g_err, sizeof(g_err), "gpx_compact_last_dev: the call it belongs to is no longer the engine's most recent");
    return GPX_EINVAL;
  }
  if (L.kind == 1 || L.kind == 2) {
    launch_one_compaction(e, L);
  } else if (L.kind == 3 || L.kind == 4) {
    const DevScratch X0 = e->X;
    e->X = L.X;
    if (L.kind == 4) { /* a small call counted nothing per chunk (k_ar_runs<.., SMALL>) */
      LaunchScope _ls(e, "k_runs_count");
      hipLaunchKernelGGL(k_runs_count, dim3(L.nchunks), dim3(GPX_DCHUNK), 0, e->stream, e->X, L.n, L.rs, (const RunsInfo*)L.info);
    }
    {
      LaunchScope _ls(e, "k_emit_dec_runs");
      hipLaunchKernelGGL(k_emit_dec_runs, dim3(std::min(L.nchunks, GPX_EMIT_GRID)), dim3(GPX_DCHUNK), 0, e->stream, e->X, L.n, L.nchunks,
                         L.rs, L.info, L.count, &e->X.counters[1], 1, 1);
    }
    LAUNCH(e, "k_merge_runs", k_merge_runs, 256, e->X, L.n, L.rs, (const RunsInfo*)L.info);
    e->X = X0;
  }
  L.kind = 0;
  HIPCHK(hipGetLastError());
  return GPX_OK;
}

/* handle: device pointer or null (gpx_propose_batch_h) */
static int propose_dev_impl(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                            const int64_t* handle, int32_t* slot, int32_t* bnum, int32_t* bcoord,
                            int32_t* median_cp, uint8_t* status) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (n == 0) return GPX_OK;
  gpx_engine* e = h;
  const int fs = begin_front(e);
  const bool promised = (e->ordered_mask & GPX_ORDERED_PROPOSE) != 0;
  const int32_t refuse = promised ? 1 : 0;
  /* at most 65,536 requests on one stream: order check and direct application in one launch */
  const bool fused = n <= GPX_SMALL_DIRECT_MAX_N;
  if (promised) { /* (gpx_one.hip.h) */
    e->stream = e->sB;
    const int nch = grid_for(n);
    GridXchg Q;
    int pg;
    const void* pers_kernel = e->cfg.kmax <= 4 ? (const void*)k_propose_pers<4>
                              : e->cfg.kmax <= 8 ? (const void*)k_propose_pers<8> : (const void*)k_propose_pers<16>;
    /* ONE launch - a grid that is resident for sure, the verdict exchanged among its workgroups - up to 128 chunks
     * (32,768 proposals): measured per step at 32,000 groups 0.0317 ms against 0.034 for check + work kernel, at 65,000
     * 0.035 against 0.032, at 125,000 0.048 against 0.045 - the exchange grows with the arrivals
     * (profiles/r06_one_launch_propose_sizes.txt) */
    if (nch <= e->pers_max_chunks[0] && xchg_ctl(e, nch, &Q, &pg, pers_kernel, GPX_BLOCK)) {
      if (e->cfg.kmax <= 4)
        LAUNCH(e, "k_propose_pers", k_propose_pers<4>, pg, e->S, e->X, Q, n, gidx, is_stop, slot, bnum, bcoord, median_cp,
               status, handle);
      else if (e->cfg.kmax <= 8)
        LAUNCH(e, "k_propose_pers", k_propose_pers<8>, pg, e->S, e->X, Q, n, gidx, is_stop, slot, bnum, bcoord, median_cp,
               status, handle);
      else
        LAUNCH(e, "k_propose_pers", k_propose_pers<16>, pg, e->S, e->X, Q, n, gidx, is_stop, slot, bnum, bcoord, median_cp,
               status, handle);
    } else { /* the verdict, then the application without a status prefill pass */
      const OneCtl C = one_ctl(e);
      {
        LaunchScope _lc(e, "k_one_check");
        hipLaunchKernelGGL(k_one_check<true>, dim3((n + GPX_DBLOCK * 8 - 1) / (GPX_DBLOCK * 8)), dim3(GPX_DBLOCK), 0, e->stream, n, gidx, e->S.G,
                           C, (int32_t*)nullptr, 0);
      }
      if (e->cfg.kmax <= 4)
        LAUNCH(e, "k_propose_one", k_propose_one<4>, nch, e->S, e->X, C, n, gidx, is_stop, slot, bnum, bcoord, median_cp,
               status, handle);
      else if (e->cfg.kmax <= 8)
        LAUNCH(e, "k_propose_one", k_propose_one<8>, nch, e->S, e->X, C, n, gidx, is_stop, slot, bnum, bcoord, median_cp,
               status, handle);
      else
        LAUNCH(e, "k_propose_one", k_propose_one<16>, nch, e->S, e->X, C, n, gidx, is_stop, slot, bnum, bcoord, median_cp,
               status, handle);
    }
    end_call(e, fs);
    HIPCHK(hipGetLastError());
    return GPX_OK;
  }
  if (fused) {
    e->stream = e->sB;
    if (e->cfg.kmax <= 4)
      LAUNCH(e, "k_propose_small", k_propose_small<4>, grid_for(n), e->S, e->X, n, gidx, is_stop, slot, bnum, bcoord,
             median_cp, status, handle, refuse);
    else if (e->cfg.kmax <= 8)
      LAUNCH(e, "k_propose_small", k_propose_small<8>, grid_for(n), e->S, e->X, n, gidx, is_stop, slot, bnum, bcoord,
             median_cp, status, handle, refuse);
    else
      LAUNCH(e, "k_propose_small", k_propose_small<16>, grid_for(n), e->S, e->X, n, gidx, is_stop, slot, bnum, bcoord,
             median_cp, status, handle, refuse);
  }
  /* order check + status prefill (one pass over gidx); the partition front end only does work for a
   * batch that is NOT strictly ascending */
  if (!fused)
  LAUNCH_OC(e, "k_order_check", k_order_check<true>, (n + GPX_OC_BLOCK * 8 - 1) / (GPX_OC_BLOCK * 8), 0, n, gidx,
           e->S.G, e->X, status, (int32_t*)nullptr, 0);
  if (!promised) {
    front_hist(e, n, gidx, fused ? nullptr : status, 0, 2);
    const int ntiles = ntiles_for(n);
    LAUNCH_F(e, "k_scatter_pr", k_scatter_pr, tile_grid(ntiles), (size_t)e->X.nbk * 4, n, ntiles, e->S.G, e->X,
             gidx, is_stop, slot, bnum, bcoord, median_cp);
  }
  begin_back(e, fs, n);
  /* exactly one of the two back ends does the work (device-side choice on *X.unsorted, set by
   * k_order_check): strictly ascending batch -> k_propose_direct, anything else -> k_bucket_propose;
   * under the GPX_ORDERED_PROPOSE promise only the direct one exists and the other case is refused */
  if (e->cfg.kmax <= 4) {
    if (!fused)
      LAUNCH(e, "k_propose_direct", k_propose_direct<4>, grid_for(n), e->S, e->X, n, gidx, is_stop, slot,
             bnum, bcoord, median_cp, status, handle, refuse);
    if (!promised) launch_bucket_propose<4>(e, slot, bnum, bcoord, median_cp, status, handle);
  } else if (e->cfg.kmax <= 8) {
    if (!fused)
      LAUNCH(e, "k_propose_direct", k_propose_direct<8>, grid_for(n), e->S, e->X, n, gidx, is_stop, slot,
             bnum, bcoord, median_cp, status, handle, refuse);
    if (!promised) launch_bucket_propose<8>(e, slot, bnum, bcoord, median_cp, status, handle);
  } else {
    if (!fused)
      LAUNCH(e, "k_propose_direct", k_propose_direct<16>, grid_for(n), e->S, e->X, n, gidx, is_stop, slot,
             bnum, bcoord, median_cp, status, handle, refuse);
    if (!promised) launch_bucket_propose<16>(e, slot, bnum, bcoord, median_cp, status, handle);
  }
  end_call(e, fs);
  HIPCHK(hipGetLastError());
  return GPX_OK;
}

int gpx_propose_batch_dev(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                          int32_t* slot, int32_t* bnum, int32_t* bcoord, int32_t* median_cp,
                          uint8_t* status) {
  return propose_dev_impl(h, n, gidx, is_stop, nullptr, slot, bnum, bcoord, median_cp, status);
}

/* ---- host-pointer data path ---------------------------------------------------- */
/* H2D into the engine's staging columns, the _dev twin, D2H of the results.        */

#define H2D(dst, src, bytes) HIPCHK(xfer(h, (dst), (src), (bytes), hipMemcpyHostToDevice, h->sF))
/* lifecycle calls change group state: everything on the back-end stream, in call order */
#define H2D_B(dst, src, bytes) HIPCHK(xfer(h, (dst), (src), (bytes), hipMemcpyHostToDevice, h->sB))
#define D2H(dst, src, bytes) HIPCHK(xfer(h, (dst), (src), (bytes), hipMemcpyDeviceToHost, h->sB))

int gpx_propose_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                      int32_t* slot, int32_t* bnum, int32_t* bcoord, int32_t* median_cp,
                      uint8_t* status) {
  return gpx_propose_batch_h(h, n, gidx, is_stop, nullptr, slot, bnum, bcoord, median_cp, status);
}

int gpx_propose_batch_h(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop,
                        const int64_t* handle, int32_t* slot, int32_t* bnum, int32_t* bcoord,
                        int32_t* median_cp, uint8_t* status) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (n == 0) return GPX_OK;
  if (!gidx || !slot || !bnum || !bcoord || !median_cp || !status) return GPX_EINVAL;
  const size_t b4 = (size_t)n * 4;
  if (n <= GPX_STAGE_N) { /* one block each way */
    Stage st(h);
    const int32_t* dg = st.in(gidx, (size_t)n);
    const uint8_t* ds = st.in(is_stop, (size_t)n);
    const int64_t* dh = st.in(handle, (size_t)n);
    const int32_t *v_slot, *v_bnum, *v_bcoord, *v_med;
    const uint8_t* v_st;
    int32_t* o_slot = st.out<int32_t>((size_t)n, &v_slot);
    int32_t* o_bnum = st.out<int32_t>((size_t)n, &v_bnum);
    int32_t* o_bcoord = st.out<int32_t>((size_t)n, &v_bcoord);
    int32_t* o_med = st.out<int32_t>((size_t)n, &v_med);
    uint8_t* o_st = st.out<uint8_t>((size_t)n, &v_st);
    if ((rc = st.upload()) != GPX_OK) return rc;
    rc = propose_dev_impl(h, n, dg, ds, dh, o_slot, o_bnum, o_bcoord, o_med, o_st);
    if (rc != GPX_OK) return rc;
    if ((rc = st.finish()) != GPX_OK) return rc;
    memcpy(slot, v_slot, b4);
    memcpy(bnum, v_bnum, b4);
    memcpy(bcoord, v_bcoord, b4);
    memcpy(median_cp, v_med, b4);
    memcpy(status, v_st, (size_t)n);
    return GPX_OK;
  }
  H2D(h->st_i32[0], gidx, b4);
  if (is_stop) H2D(h->st_u8[0], is_stop, (size_t)n);
  /* staging column of the 64-bit handles: allocated on first use */
  int64_t* d_handle = nullptr;
  if (handle) {
    if (!h->st_handle) {
      rc = dev_alloc(h, &h->st_handle, (size_t)h->cfg.max_batch, false);
      if (rc != GPX_OK) return rc;
    }
    d_handle = h->st_handle;
    H2D(d_handle, handle, b4 * 2);
  }
  rc = propose_dev_impl(h, n, h->st_i32[0], is_stop ? h->st_u8[0] : nullptr, d_handle, h->st_i32[1],
                        h->st_i32[2], h->st_i32[3], h->st_i32[4], h->st_u8[1]);
  if (rc != GPX_OK) return rc;
  D2H(slot, h->st_i32[1], b4);
  D2H(bnum, h->st_i32[2], b4);
  D2H(bcoord, h->st_i32[3], b4);
  D2H(median_cp, h->st_i32[4], b4);
  D2H(status, h->st_u8[1], (size_t)n);
  SYNC_CHECKED(h, h->sB);
  return GPX_OK;
}

int gpx_accept_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                     const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                     const uint8_t* a_flags, int32_t* r_bnum, int32_t* r_bcoord,
                     int32_t* r_maxcp, uint8_t* r_flags, uint8_t* status, int32_t* x_gidx,
                     int32_t* x_first, int32_t* x_count, int32_t* n_runs) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (!n_runs) return GPX_EINVAL;
  *n_runs = 0;
  if (n == 0) return GPX_OK;
  const size_t b4 = (size_t)n * 4;
  if (n <= GPX_STAGE_N) { /* one block each way */
    Stage st(h);
    const int32_t* dg = st.in(gidx, (size_t)n);
    const int32_t* db = st.in(bnum, (size_t)n);
    const int32_t* dc = st.in(bcoord, (size_t)n);
    const int32_t* dsl = st.in(slot, (size_t)n);
    const int32_t* dm = st.in(median_cp, (size_t)n);
    const uint8_t* df = st.in(a_flags, (size_t)n);
    const int32_t *v_rb, *v_rc, *v_rm, *v_xg, *v_xf, *v_xc, *v_nr;
    const uint8_t *v_rf, *v_st;
    int32_t* o_rb = st.out<int32_t>((size_t)n, &v_rb);
    int32_t* o_rc = st.out<int32_t>((size_t)n, &v_rc);
    int32_t* o_rm = st.out<int32_t>((size_t)n, &v_rm);
    uint8_t* o_rf = st.out<uint8_t>((size_t)n, &v_rf);
    uint8_t* o_st = st.out<uint8_t>((size_t)n, &v_st);
    int32_t* o_xg = st.out<int32_t>((size_t)n, &v_xg);
    int32_t* o_xf = st.out<int32_t>((size_t)n, &v_xf);
    int32_t* o_xc = st.out<int32_t>((size_t)n, &v_xc);
    int32_t* o_nr = st.out<int32_t>(4, &v_nr);
    if ((rc = st.upload()) != GPX_OK) return rc;
    h->lazy_override = 0; /* one block comes back, whatever the batch: its columns must be dense (no GPX_LAZY_OUTPUTS here) */
    rc = gpx_accept_batch_dev(h, n, dg, db, dc, dsl, dm, df, o_rb, o_rc, o_rm, o_rf, o_st, o_xg, o_xf, o_xc, o_nr);
    h->lazy_override = -1;
    h->last.kind = 0;
    if (rc != GPX_OK) return rc;
    if ((rc = st.finish()) != GPX_OK) return rc;
    memcpy(r_bnum, v_rb, b4);
    memcpy(r_bcoord, v_rc, b4);
    memcpy(r_maxcp, v_rm, b4);
    memcpy(r_flags, v_rf, (size_t)n);
    memcpy(status, v_st, (size_t)n);
    *n_runs = v_nr[0];
    const size_t m4 = (size_t)(*n_runs) * 4;
    memcpy(x_gidx, v_xg, m4);
    memcpy(x_first, v_xf, m4);
    memcpy(x_count, v_xc, m4);
    return GPX_OK;
  }
  H2D(h->st_i32[0], gidx, b4);
  H2D(h->st_i32[1], bnum, b4);
  H2D(h->st_i32[2], bcoord, b4);
  H2D(h->st_i32[3], slot, b4);
  H2D(h->st_i32[4], median_cp, b4);
  if (a_flags) H2D(h->st_u8[0], a_flags, (size_t)n);
  h->lazy_override = 1; /* the count comes to the host anyway: compaction only for a batch that needs it */
  rc = gpx_accept_batch_dev(h, n, h->st_i32[0], h->st_i32[1], h->st_i32[2], h->st_i32[3],
                            h->st_i32[4], a_flags ? h->st_u8[0] : nullptr, h->st_i32[5],
                            h->st_i32[6], h->st_i32[7], h->st_u8[1], h->st_u8[2], h->st_i32[8],
                            h->st_i32[9], h->st_i32[10], h->st_count);
  h->lazy_override = -1;
  if (rc != GPX_OK) return rc;
  D2H(n_runs, h->st_count, 4);
  if (h->last.kind) {
    SYNC_CHECKED(h, h->sB);
    if (*n_runs < 0) {
      if ((rc = gpx_compact_last_dev(h)) != GPX_OK) return rc;
      D2H(n_runs, h->st_count, 4);
    }
    h->last.kind = 0;
  }
  D2H(r_bnum, h->st_i32[5], b4);
  D2H(r_bcoord, h->st_i32[6], b4);
  D2H(r_maxcp, h->st_i32[7], b4);
  D2H(r_flags, h->st_u8[1], (size_t)n);
  D2H(status, h->st_u8[2], (size_t)n);
  SYNC_CHECKED(h, h->sB);
  const size_t m4 = (size_t)(*n_runs) * 4;
  if (m4) {
    D2H(x_gidx, h->st_i32[8], m4);
    D2H(x_first, h->st_i32[9], m4);
    D2H(x_count, h->st_i32[10], m4);
    SYNC_CHECKED(h, h->sB);
  }
  return GPX_OK;
}

int gpx_accept_reply_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                           const int32_t* bcoord, const int32_t* slot, const int32_t* acceptor,
                           const int32_t* max_cp, int32_t* d_gidx, int32_t* d_slot,
                           int32_t* d_bnum, int32_t* d_bcoord, int32_t* d_median_cp,
                           uint8_t* d_kind, int32_t* n_out, uint8_t* status) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (!n_out) return GPX_EINVAL;
  *n_out = 0;
  if (n == 0) return GPX_OK;
  const size_t b4 = (size_t)n * 4;
  if (n <= GPX_STAGE_N) { /* one block each way */
    Stage st(h);
    const int32_t* dg = st.in(gidx, (size_t)n);
    const int32_t* db = st.in(bnum, (size_t)n);
    const int32_t* dc = st.in(bcoord, (size_t)n);
    const int32_t* dsl = st.in(slot, (size_t)n);
    const int32_t* da = st.in(acceptor, (size_t)n);
    const int32_t* dm = st.in(max_cp, (size_t)n);
    const int32_t *v_g, *v_s, *v_b, *v_c, *v_m, *v_no;
    const uint8_t *v_k, *v_st;
    int32_t* o_g = st.out<int32_t>((size_t)n, &v_g);
    int32_t* o_s = st.out<int32_t>((size_t)n, &v_s);
    int32_t* o_b = st.out<int32_t>((size_t)n, &v_b);
    int32_t* o_c = st.out<int32_t>((size_t)n, &v_c);
    int32_t* o_m = st.out<int32_t>((size_t)n, &v_m);
    uint8_t* o_k = st.out<uint8_t>((size_t)n, &v_k);
    uint8_t* o_st = st.out<uint8_t>((size_t)n, &v_st);
    int32_t* o_no = st.out<int32_t>(4, &v_no);
    if ((rc = st.upload()) != GPX_OK) return rc;
    h->lazy_override = 0; /* one block comes back, whatever the batch: its columns must be dense (no GPX_LAZY_OUTPUTS here) */
    rc = gpx_accept_reply_batch_dev(h, n, dg, db, dc, dsl, da, dm, o_g, o_s, o_b, o_c, o_m, o_k, o_no, o_st);
    h->lazy_override = -1;
    h->last.kind = 0;
    if (rc != GPX_OK) return rc;
    if ((rc = st.finish()) != GPX_OK) return rc;
    *n_out = v_no[0];
    const size_t m = (size_t)(*n_out);
    memcpy(d_gidx, v_g, m * 4);
    memcpy(d_slot, v_s, m * 4);
    memcpy(d_bnum, v_b, m * 4);
    memcpy(d_bcoord, v_c, m * 4);
    memcpy(d_median_cp, v_m, m * 4);
    memcpy(d_kind, v_k, m);
    if (status) memcpy(status, v_st, (size_t)n);
    return GPX_OK;
  }
  H2D(h->st_i32[0], gidx, b4);
  H2D(h->st_i32[1], bnum, b4);
  H2D(h->st_i32[2], bcoord, b4);
  H2D(h->st_i32[3], slot, b4);
  H2D(h->st_i32[4], acceptor, b4);
  H2D(h->st_i32[5], max_cp, b4);
  h->lazy_override = 1; /* the count comes to the host anyway: compaction only for a batch that needs it */
  rc = gpx_accept_reply_batch_dev(h, n, h->st_i32[0], h->st_i32[1], h->st_i32[2], h->st_i32[3],
                                  h->st_i32[4], h->st_i32[5], h->st_i32[6], h->st_i32[7],
                                  h->st_i32[8], h->st_i32[9], h->st_i32[10], h->st_u8[0],
                                  h->st_count, h->st_u8[1]);
  h->lazy_override = -1;
  if (rc != GPX_OK) return rc;
  D2H(n_out, h->st_count, 4);
  if (status) D2H(status, h->st_u8[1], (size_t)n);
  SYNC_CHECKED(h, h->sB);
  if (h->last.kind) {
    if (*n_out < 0) {
      if ((rc = gpx_compact_last_dev(h)) != GPX_OK) return rc;
      D2H(n_out, h->st_count, 4);
      SYNC_CHECKED(h, h->sB);
    }
    h->last.kind = 0;
  }
  const size_t m = (size_t)(*n_out);
  if (m) {
    D2H(d_gidx, h->st_i32[6], m * 4);
    D2H(d_slot, h->st_i32[7], m * 4);
    D2H(d_bnum, h->st_i32[8], m * 4);
    D2H(d_bcoord, h->st_i32[9], m * 4);
    D2H(d_median_cp, h->st_i32[10], m * 4);
    D2H(d_kind, h->st_u8[0], m);
    SYNC_CHECKED(h, h->sB);
  }
  return GPX_OK;
}

int gpx_commit_batch(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                     const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                     const uint8_t* c_kind, uint8_t* status, int32_t* x_gidx, int32_t* x_first,
                     int32_t* x_count, int32_t* n_runs) {
  int rc = check_batch(h, n);
  if (rc != GPX_OK) return rc;
  if (!n_runs) return GPX_EINVAL;
  *n_runs = 0;
  if (n == 0) return GPX_OK;
  const size_t b4 = (size_t)n * 4;
  if (n <= GPX_STAGE_N) { /* one block each way */
    Stage st(h);
    const int32_t* dg = st.in(gidx, (size_t)n);
    const int32_t* db = st.in(bnum, (size_t)n);
    const int32_t* dc = st.in(bcoord, (size_t)n);
    const int32_t* dsl = st.in(slot, (size_t)n);
    const int32_t* dm = st.in(median_cp, (size_t)n);
    const uint8_t* dk = st.in(c_kind, (size_t)n);
    const int32_t *v_xg, *v_xf, *v_xc, *v_nr;
    const uint8_t* v_st;
    uint8_t* o_st = st.out<uint8_t>((size_t)n, &v_st);
    int32_t* o_xg = st.out<int32_t>((size_t)n, &v_xg);
    int32_t* o_xf = st.out<int32_t>((size_t)n, &v_xf);
    int32_t* o_xc = st.out<int32_t>((size_t)n, &v_xc);
    int32_t* o_nr = st.out<int32_t>(4, &v_nr);
    if ((rc = st.upload()) != GPX_OK) return rc;
    h->lazy_override = 0; /* one block comes back, whatever the batch: its columns must be dense (no GPX_LAZY_OUTPUTS here) */
    rc = gpx_commit_batch_dev(h, n, dg, db, dc, dsl, dm, dk, o_st, o_xg, o_xf, o_xc, o_nr);
    h->lazy_override = -1;
    h->last.kind = 0;
    if (rc != GPX_OK) return rc;
    if ((rc = st.finish()) != GPX_OK) return rc;
    memcpy(status, v_st, (size_t)n);
    *n_runs = v_nr[0];
    const size_t m4 = (size_t)(*n_runs) * 4;
    memcpy(x_gidx, v_xg, m4);
    memcpy(x_first, v_xf, m4);
    memcpy(x_count, v_xc, m4);
    return GPX_OK;
  }
  H2D(h->st_i32[0], gidx, b4);
  H2D(h->st_i32[1], bnum, b4);
  H2D(h->st_i32[2], bcoord, b4);
  H2D(h->st_i32[3], slot, b4);
  H2D(h->st_i32[4], median_cp, b4);
  if (c_kind) H2D(h->st_u8[0], c_kind, (size_t)n);
  h->lazy_override = 1;
  rc = gpx_commit_batch_dev(h, n, h->st_i32[0], h->st_i32[1], h->st_i32[2], h->st_i32[3],
                            h->st_i32[4], c_kind ? h->st_u8[0] : nullptr, h->st_u8[1],
                            h->st_i32[5], h->st_i32[6], h->st_i32[7], h->st_count);
  h->lazy_override = -1;
  if (rc != GPX_OK) return rc;
  D2H(n_runs, h->st_count, 4);
  if (h->last.kind) {
    SYNC_CHECKED(h, h->sB);
    if (*n_runs < 0) {
      if ((rc = gpx_compact_last_dev(h)) != GPX_OK) return rc;
      D2H(n_runs, h->st_count, 4);
    }
    h->last.kind = 0;
  }
  D2H(status, h->st_u8[1], (size_t)n);
  SYNC_CHECKED(h, h->sB);
  const size_t m4 = (size_t)(*n_runs) * 4;
  if (m4) {
    D2H(x_gidx, h->st_i32[5], m4);
    D2H(x_first, h->st_i32[6], m4);
    D2H(x_count, h->st_i32[7], m4);
    SYNC_CHECKED(h, h->sB);
  }
  return GPX_OK;
}

} /* extern "C" */

/* ---- asynchronous host-pointer data path ---------------------------------------------- */
/* (include/gpx.h: inputs on a copy-in stream, kernels on the engine's stream behind them, dense outputs and
 * the count on the set's copy-out stream; gpx_engine_wait fetches exactly `count` compacted entries) */

/* Compacted outputs of an asynchronous call straight into the caller's (registered, device-mapped) host
 * buffers: the count is known on the DEVICE when this runs, so exactly `count` entries cross the link without
 * a host round trip for the count in between - the copy-out of call N is on its way while the host is still
 * queueing call N + 1.  Plain coalesced 4-byte stores (posted PCIe writes). */
struct CopyOut {
  int ncols, nb;
  const int32_t* src[6];
  int32_t* dst[6];
  const uint8_t* bsrc[2]; /* byte columns */
  uint8_t* bdst[2];
  int32_t* count_dst; /* the caller's n_out / n_runs, or null */
  int32_t fixed_n;    /* entries to copy when there is no device count (dense per-record outputs) */
};
__global__ __launch_bounds__(256) void k_copy_out(const int32_t* __restrict__ count, CopyOut C) {
  const int32_t m = count ? *count : C.fixed_n;
  if (C.count_dst && blockIdx.x == 0 && threadIdx.x == 0) *C.count_dst = m;
  /* 16 bytes per lane and store where the columns allow it (round 5: the link takes a kernel's writes to host memory
   * faster as 1 KB per wave and column than as 256 bytes; the tail and unaligned columns entry by entry) */
  bool vec = true;
#pragma unroll
  for (int k = 0; k < 6; k++)
    if (k < C.ncols) vec = vec && !(((uintptr_t)C.dst[k] | (uintptr_t)C.src[k]) & 15);
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (k < C.nb) vec = vec && !(((uintptr_t)C.bdst[k] | (uintptr_t)C.bsrc[k]) & 3);
  const int32_t mv = vec ? (m & ~3) : 0;
  for (int32_t i = (blockIdx.x * 256 + threadIdx.x) * 4; i < mv; i += gridDim.x * 256 * 4) {
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < C.ncols) *(I4*)(C.dst[k] + i) = *(const I4*)(C.src[k] + i);
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (k < C.nb) *(uint32_t*)(C.bdst[k] + i) = *(const uint32_t*)(C.bsrc[k] + i);
  }
  for (int32_t i = mv + blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
#pragma unroll
    for (int k = 0; k < 6; k++)
      if (k < C.ncols) C.dst[k][i] = C.src[k][i];
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (k < C.nb) C.bdst[k][i] = C.bsrc[k][i];
  }
}

/* ... and the other way (experiment, GPX_ASYNC_COPYIN=kernel): the input columns read straight out of the
 * caller's registered host buffers by a kernel on the copy-in stream.  Measured on the MI355X box
 * (scripts/bench_async_path.py, profiles/r03_async_variants.txt): slower than the runtime's DMA copies - 2.0 ms
 * per step against 1.45 ms - so the default stays hipMemcpyAsync per column. */
struct CopyIn {
  int32_t n;
  int ncols;
  const int32_t* src[6];
  int32_t* dst[6];
  const uint8_t* bsrc; /* one byte column (flags), or null */
  uint8_t* bdst;
};
__global__ __launch_bounds__(256) void k_copy_in(CopyIn C) {
  const int64_t nv = C.n >> 2; /* whole 16-byte vectors; the device columns are 16-byte aligned */
  for (int k = 0; k < 6; k++) {
    if (k >= C.ncols) break;
    const int32_t* __restrict__ s = C.src[k];
    int32_t* __restrict__ d = C.dst[k];
    if (!((uintptr_t)s & 15)) {
      for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nv; v += (int64_t)gridDim.x * 256)
        ((I4*)d)[v] = ((const I4*)s)[v];
      for (int64_t i = (nv << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < C.n; i += (int64_t)gridDim.x * 256) d[i] = s[i];
    } else {
      for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < C.n; i += (int64_t)gridDim.x * 256) d[i] = s[i];
    }
  }
  if (C.bsrc)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < C.n; i += (int64_t)gridDim.x * 256) C.bdst[i] = C.bsrc[i];
}

namespace {

/* the device address of a host buffer of `bytes` bytes INSIDE a block this engine knows to be pinned - given to
 * gpx_host_register or got from gpx_host_alloc - or null (the caller then takes the copy path, which works for any host
 * memory).  The buffer must fit in its block: a kernel running over the end of a mapping faults on the GPU.
 * Memory the engine was never told about is NOT written through a mapping, whatever the runtime says about it (until
 * round 6 hipPointerGetAttributes' "host memory" was taken on its word; scripts/probe_runtime_pins.py shows that the
 * runtime does not report its own transient pinnings that way, so this was not the fault of profiles/
 * r06_abort_backtrace.txt - but one rule is easier to keep than two).  Pinned memory from elsewhere is registered like
 * any other (gpx_host_register notes that it is pinned already). */
void* mapped_host(gpx_engine* e, void* p, size_t bytes) {
  if (!p) return nullptr;
  bool known = false;
  for (auto& r : e->registered)
    if ((char*)p >= r.first && (char*)p < r.first + r.second) {
      if ((char*)p + bytes > r.first + r.second) return nullptr;
      known = true;
      break;
    }
  if (!known) return nullptr;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return d;
}

/* an asynchronous call failed after some of its copies or kernels were queued: no ticket will be issued, so the
 * caller has nothing to wait on - wait here, so that it may reuse its buffers and the set's columns are quiet */
int async_fail(gpx_engine* e, gpx_engine::AsyncSet& a, int rc) {
  if (e->s_in) HIPQ(hipStreamSynchronize(e->s_in));
  HIPQ(hipStreamSynchronize(e->sB));
  if (a.s_out) HIPQ(hipStreamSynchronize(a.s_out));
  (void)hipGetLastError();
  return rc;
}

int async_begin(gpx_engine* e, int32_t n, gpx_engine::AsyncSet** out) {
  int rc = check_batch(e, n);
  if (rc != GPX_OK) return rc;
  gpx_engine::AsyncSet& a = e->as[e->async_seq % (uint64_t)e->async_depth];
  if (a.busy) return GPX_EBUSY;
  if (!e->s_in) {
    /* GPX_ASYNC_IN=engine (experiment): inputs on the engine's own stream instead of a copy stream */
    const char* v = getenv("GPX_ASYNC_IN");
    e->async_in_engine = v && !strcmp(v, "engine");
    const char* f = getenv("GPX_ASYNC_FILL");
    e->async_fill_memset = f && !strcmp(f, "memset");
    const char* dd = getenv("GPX_ASYNC_DIRECT");
    e->async_no_direct = dd && !strcmp(dd, "0");
    const char* ci = getenv("GPX_ASYNC_COPYIN");
    e->async_kernel_in = ci && !strcmp(ci, "kernel");
    HIPCHK(hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
  }
  if (!a.ready) {
    const size_t N = (size_t)e->cfg.max_batch;
    for (auto& p : a.i32)
      if ((rc = dev_alloc(e, &p, N, false)) != GPX_OK) return rc;
    for (auto& p : a.u8)
      if ((rc = dev_alloc(e, &p, N, false)) != GPX_OK) return rc;
    if ((rc = dev_alloc(e, &a.cnt, 4, true)) != GPX_OK) return rc;
    HIPCHK(hipHostMalloc((void**)&a.h_cnt, 64, hipHostMallocDefault));
    HIPCHK(hipStreamCreateWithFlags(&a.s_out, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&a.ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&a.ev_k, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&a.ev_cnt, hipEventDisableTiming));
    a.ready = true;
  }
  a.ncols = 0;
  a.n = n;
  a.host_kind = nullptr;
  a.dev_kind = nullptr;
  a.host_count = nullptr;
  a.h_cnt[0] = 0;
  *out = &a;
  return GPX_OK;
}
/* queues the input columns of a call: a DMA copy per column (from pageable memory the runtime stages it) */
int async_inputs(gpx_engine* e, int32_t n, int ncols, const int32_t* const* hsrc, int32_t* const* ddst,
                 const uint8_t* hb, uint8_t* db) {
  hipStream_t st = e->async_in_engine ? e->sB : e->s_in;
  CopyIn C{};
  C.n = n;
  C.ncols = ncols;
  bool ok = e->async_kernel_in;
  for (int k = 0; k < ncols && ok; k++) {
    C.src[k] = (const int32_t*)mapped_host(e, (void*)hsrc[k], (size_t)n * 4);
    C.dst[k] = ddst[k];
    ok = C.src[k] != nullptr;
  }
  if (ok && hb) {
    C.bsrc = (const uint8_t*)mapped_host(e, (void*)hb, (size_t)n);
    C.bdst = db;
    ok = C.bsrc != nullptr;
  }
  if (ok) {
    hipLaunchKernelGGL(k_copy_in, dim3(1024), dim3(256), 0, st, C);
    return GPX_OK;
  }
  for (int k = 0; k < ncols; k++) HIPCHK(xfer(e, ddst[k], hsrc[k], (size_t)n * 4, hipMemcpyHostToDevice, st));
  if (hb) HIPCHK(xfer(e, db, hb, (size_t)n, hipMemcpyHostToDevice, st));
  return GPX_OK;
}
/* inputs are on their way: the kernels (engine stream) wait for them */
int async_inputs_done(gpx_engine* e, gpx_engine::AsyncSet& a) {
  if (e->async_in_engine) return GPX_OK;
  HIPCHK(hipEventRecord(a.ev_in, e->s_in));
  HIPCHK(hipStreamWaitEvent(e->sB, a.ev_in, 0));
  return GPX_OK;
}
/* kernels are queued: the copy-out stream waits for them */
int async_kernels_done(gpx_engine* e, gpx_engine::AsyncSet& a) {
  HIPCHK(hipEventRecord(a.ev_k, e->sB));
  HIPCHK(hipStreamWaitEvent(a.s_out, a.ev_k, 0));
  return GPX_OK;
}
/* dense per-record outputs (n entries each): one k_copy_out into registered memory, else a copy per column */
int async_dense_out(gpx_engine* e, gpx_engine::AsyncSet& a, int32_t n, int ncols, int32_t* const* hdst,
                    const int32_t* const* dsrc, int nb, uint8_t* const* hb, const uint8_t* const* db) {
  CopyOut C{};
  C.ncols = ncols;
  C.nb = nb;
  C.fixed_n = n;
  bool ok = !e->async_no_direct;
  for (int k = 0; k < ncols && ok; k++) {
    C.src[k] = dsrc[k];
    C.dst[k] = (int32_t*)mapped_host(e, hdst[k], (size_t)n * 4);
    ok = C.dst[k] != nullptr;
  }
  for (int k = 0; k < nb && ok; k++) {
    C.bsrc[k] = db[k];
    C.bdst[k] = (uint8_t*)mapped_host(e, hb[k], (size_t)n);
    ok = C.bdst[k] != nullptr;
  }
  /* (a kernel writing through the host mapping moves about 31 GB/s where a lone DMA copy moves 48 with the other direction
   * busy - but DMA copies for the big dense columns, tried in round 5, queue behind the copy-in stream's DMA: 2.35 ms per
   * step against 1.33, profiles/r05_bench_e2e_dense_dma.json) */
  if (ok) {
    hipLaunchKernelGGL(k_copy_out, dim3(512), dim3(256), 0, a.s_out, (const int32_t*)nullptr, C);
    return GPX_OK;
  }
  for (int k = 0; k < ncols; k++) HIPCHK(xfer(e, hdst[k], dsrc[k], (size_t)n * 4, hipMemcpyDeviceToHost, a.s_out));
  for (int k = 0; k < nb; k++) HIPCHK(xfer(e, hb[k], db[k], (size_t)n, hipMemcpyDeviceToHost, a.s_out));
  return GPX_OK;
}

int async_submit(gpx_engine* e, gpx_engine::AsyncSet& a, bool with_count, gpx_ticket* ticket) {
  a.direct = false;
  if (with_count && a.ncols > 0 && !e->async_no_direct) {
    /* every compacted output column in registered memory: a kernel writes exactly `count` entries there */
    CopyOut C{};
    C.ncols = a.ncols;
    bool ok = true;
    for (int k = 0; k < a.ncols && ok; k++) {
      C.src[k] = a.dev_col[k];
      C.dst[k] = (int32_t*)mapped_host(e, a.host_col[k], (size_t)a.n * 4);
      ok = C.dst[k] != nullptr;
    }
    if (ok && a.host_kind) {
      C.nb = 1;
      C.bsrc[0] = a.dev_kind;
      C.bdst[0] = (uint8_t*)mapped_host(e, a.host_kind, (size_t)a.n);
      ok = C.bdst[0] != nullptr;
    }
    C.count_dst = ok ? (int32_t*)mapped_host(e, a.host_count, 4) : nullptr;
    if (ok && C.count_dst) {
      hipLaunchKernelGGL(k_copy_out, dim3(512), dim3(256), 0, a.s_out, (const int32_t*)a.cnt, C);
      a.direct = true;
    }
  }
  if (with_count && !a.direct) HIPCHK(hipMemcpyAsync(a.h_cnt, a.cnt, sizeof(int32_t), hipMemcpyDeviceToHost, a.s_out));
  HIPCHK(hipEventRecord(a.ev_cnt, a.s_out));
  a.busy = true;
  a.ticket = ++e->async_seq; /* > 0; the next call takes the next set */
  *ticket = a.ticket;
  return GPX_OK;
}
#define A_OUT(dst, src, bytes) HIPCHK(xfer(h, (dst), (src), (bytes), hipMemcpyDeviceToHost, a.s_out))

}  // namespace

extern "C" {

int gpx_propose_batch_async(gpx_engine* h, int32_t n, const int32_t* gidx, const uint8_t* is_stop, int32_t* slot,
                            int32_t* bnum, int32_t* bcoord, int32_t* median_cp, uint8_t* status,
                            gpx_ticket* ticket) {
  if (!h || !ticket || (n > 0 && (!gidx || !slot || !bnum || !bcoord || !median_cp || !status))) return GPX_EINVAL;
  gpx_engine::AsyncSet* ap = nullptr;
  int rc = async_begin(h, n, &ap);
  if (rc != GPX_OK) return rc;
  gpx_engine::AsyncSet& a = *ap;
  if (n > 0) {
    {
      const int32_t* hs[1] = {gidx};
      int32_t* dd[1] = {a.i32[0]};
      if ((rc = async_inputs(h, n, 1, hs, dd, is_stop, a.u8[0])) != GPX_OK) return async_fail(h, a, rc);
    }
    if ((rc = async_inputs_done(h, a)) != GPX_OK) return async_fail(h, a, rc);
    rc = propose_dev_impl(h, n, a.i32[0], is_stop ? a.u8[0] : nullptr, nullptr, a.i32[1], a.i32[2], a.i32[3],
                          a.i32[4], a.u8[1]);
    if (rc != GPX_OK) return async_fail(h, a, rc);
    if ((rc = async_kernels_done(h, a)) != GPX_OK) return async_fail(h, a, rc);
    {
      int32_t* hd[4] = {slot, bnum, bcoord, median_cp};
      const int32_t* ds[4] = {a.i32[1], a.i32[2], a.i32[3], a.i32[4]};
      uint8_t* hb[1] = {status};
      const uint8_t* db[1] = {a.u8[1]};
      if ((rc = async_dense_out(h, a, n, 4, hd, ds, 1, hb, db)) != GPX_OK) return async_fail(h, a, rc);
    }
  }
  rc = async_submit(h, a, false, ticket);
  return rc == GPX_OK ? rc : async_fail(h, a, rc);
}

int gpx_accept_batch_async(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                           const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                           const uint8_t* a_flags, int32_t* r_bnum, int32_t* r_bcoord, int32_t* r_maxcp,
                           uint8_t* r_flags, uint8_t* status, int32_t* x_gidx, int32_t* x_first,
                           int32_t* x_count, int32_t* n_runs, gpx_ticket* ticket) {
  if (!h || !ticket || !n_runs) return GPX_EINVAL;
  if (n > 0 && (!gidx || !bnum || !bcoord || !slot || !median_cp || !r_bnum || !r_bcoord || !r_maxcp || !r_flags ||
                !status || !x_gidx || !x_first || !x_count))
    return GPX_EINVAL;
  gpx_engine::AsyncSet* ap = nullptr;
  int rc = async_begin(h, n, &ap);
  if (rc != GPX_OK) return rc;
  gpx_engine::AsyncSet& a = *ap;
  a.host_count = n_runs;
  *n_runs = 0;
  if (n > 0) {
    {
      const int32_t* hs[5] = {gidx, bnum, bcoord, slot, median_cp};
      int32_t* dd[5] = {a.i32[0], a.i32[1], a.i32[2], a.i32[3], a.i32[4]};
      if ((rc = async_inputs(h, n, 5, hs, dd, a_flags, a.u8[0])) != GPX_OK) return async_fail(h, a, rc);
    }
    if ((rc = async_inputs_done(h, a)) != GPX_OK) return async_fail(h, a, rc);
    h->lazy_override = 0; /* k_copy_out reads the count on the device: dense columns, always */
    rc = gpx_accept_batch_dev(h, n, a.i32[0], a.i32[1], a.i32[2], a.i32[3], a.i32[4], a_flags ? a.u8[0] : nullptr,
                              a.i32[5], a.i32[6], a.i32[7], a.u8[1], a.u8[2], a.i32[8], a.i32[9], a.i32[10], a.cnt);
    h->lazy_override = -1;
    if (rc != GPX_OK) return async_fail(h, a, rc);
    if ((rc = async_kernels_done(h, a)) != GPX_OK) return async_fail(h, a, rc);
    {
      int32_t* hd[3] = {r_bnum, r_bcoord, r_maxcp};
      const int32_t* ds[3] = {a.i32[5], a.i32[6], a.i32[7]};
      uint8_t* hb[2] = {r_flags, status};
      const uint8_t* db[2] = {a.u8[1], a.u8[2]};
      if ((rc = async_dense_out(h, a, n, 3, hd, ds, 2, hb, db)) != GPX_OK) return async_fail(h, a, rc);
    }
    a.ncols = 3;
    a.host_col[0] = x_gidx, a.host_col[1] = x_first, a.host_col[2] = x_count;
    a.dev_col[0] = a.i32[8], a.dev_col[1] = a.i32[9], a.dev_col[2] = a.i32[10];
  }
  rc = async_submit(h, a, n > 0, ticket);
  return rc == GPX_OK ? rc : async_fail(h, a, rc);
}

int gpx_accept_reply_batch_async(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                                 const int32_t* bcoord, int32_t common_bnum, int32_t common_bcoord,
                                 const int32_t* slot, const int32_t* acceptor, const int32_t* max_cp,
                                 int32_t* d_gidx, int32_t* d_slot, int32_t* d_bnum, int32_t* d_bcoord,
                                 int32_t* d_median_cp, uint8_t* d_kind, int32_t* n_out, uint8_t* status,
                                 gpx_ticket* ticket) {
  if (!h || !ticket || !n_out || (bnum == nullptr) != (bcoord == nullptr)) return GPX_EINVAL;
  if (n > 0 && (!gidx || !slot || !acceptor || !max_cp || !d_gidx || !d_slot || !d_bnum || !d_bcoord ||
                !d_median_cp || !d_kind))
    return GPX_EINVAL;
  gpx_engine::AsyncSet* ap = nullptr;
  int rc = async_begin(h, n, &ap);
  if (rc != GPX_OK) return rc;
  gpx_engine::AsyncSet& a = *ap;
  a.host_count = n_out;
  *n_out = 0;
  if (n > 0) {
    if (bnum) {
      const int32_t* hs[6] = {gidx, bnum, bcoord, slot, acceptor, max_cp};
      int32_t* dd[6] = {a.i32[0], a.i32[1], a.i32[2], a.i32[3], a.i32[4], a.i32[5]};
      if ((rc = async_inputs(h, n, 6, hs, dd, nullptr, nullptr)) != GPX_OK) return async_fail(h, a, rc);
    } else { /* one ballot for the whole batch: the two columns are made on the device */
      const int32_t* hs[4] = {gidx, slot, acceptor, max_cp};
      int32_t* dd[4] = {a.i32[0], a.i32[3], a.i32[4], a.i32[5]};
      if ((rc = async_inputs(h, n, 4, hs, dd, nullptr, nullptr)) != GPX_OK) return async_fail(h, a, rc);
      hipStream_t fs = h->async_in_engine ? h->sB : h->s_in;
      if (h->async_fill_memset) {
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)a.i32[1], common_bnum, (size_t)n, fs));
        HIPCHK(hipMemsetD32Async((hipDeviceptr_t)a.i32[2], common_bcoord, (size_t)n, fs));
      } else {
        hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(n)), dim3(GPX_BLOCK), 0, fs, n, common_bnum, a.i32[1]);
        hipLaunchKernelGGL(k_fill_i32, dim3(grid_for(n)), dim3(GPX_BLOCK), 0, fs, n, common_bcoord, a.i32[2]);
      }
    }
    if ((rc = async_inputs_done(h, a)) != GPX_OK) return async_fail(h, a, rc);
    h->lazy_override = 0; /* k_copy_out reads the count on the device: dense columns, always */
    rc = gpx_accept_reply_batch_dev(h, n, a.i32[0], a.i32[1], a.i32[2], a.i32[3], a.i32[4], a.i32[5], a.i32[6],
                                    a.i32[7], a.i32[8], a.i32[9], a.i32[10], a.u8[0], a.cnt, a.u8[1]);
    h->lazy_override = -1;
    if (rc != GPX_OK) return async_fail(h, a, rc);
    if ((rc = async_kernels_done(h, a)) != GPX_OK) return async_fail(h, a, rc);
    if (status) {
      uint8_t* hb[1] = {status};
      const uint8_t* db[1] = {a.u8[1]};
      if ((rc = async_dense_out(h, a, n, 0, nullptr, nullptr, 1, hb, db)) != GPX_OK) return async_fail(h, a, rc);
    }
    a.ncols = 5;
    int32_t* hc[5] = {d_gidx, d_slot, d_bnum, d_bcoord, d_median_cp};
    for (int k = 0; k < 5; k++) {
      a.host_col[k] = hc[k];
      a.dev_col[k] = a.i32[6 + k];
    }
    a.host_kind = d_kind;
    a.dev_kind = a.u8[0];
  }
  rc = async_submit(h, a, n > 0, ticket);
  return rc == GPX_OK ? rc : async_fail(h, a, rc);
}

int gpx_commit_batch_async(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* bnum,
                           const int32_t* bcoord, const int32_t* slot, const int32_t* median_cp,
                           const uint8_t* c_kind, uint8_t* status, int32_t* x_gidx, int32_t* x_first,
                           int32_t* x_count, int32_t* n_runs, gpx_ticket* ticket) {
  if (!h || !ticket || !n_runs) return GPX_EINVAL;
  if (n > 0 && (!gidx || !bnum || !bcoord || !slot || !median_cp || !status || !x_gidx || !x_first || !x_count))
    return GPX_EINVAL;
  gpx_engine::AsyncSet* ap = nullptr;
  int rc = async_begin(h, n, &ap);
  if (rc != GPX_OK) return rc;
  gpx_engine::AsyncSet& a = *ap;
  a.host_count = n_runs;
  *n_runs = 0;
  if (n > 0) {
    {
      const int32_t* hs[5] = {gidx, bnum, bcoord, slot, median_cp};
      int32_t* dd[5] = {a.i32[0], a.i32[1], a.i32[2], a.i32[3], a.i32[4]};
      if ((rc = async_inputs(h, n, 5, hs, dd, c_kind, a.u8[0])) != GPX_OK) return async_fail(h, a, rc);
    }
    if ((rc = async_inputs_done(h, a)) != GPX_OK) return async_fail(h, a, rc);
    h->lazy_override = 0; /* k_copy_out reads the count on the device: dense columns, always */
    rc = gpx_commit_batch_dev(h, n, a.i32[0], a.i32[1], a.i32[2], a.i32[3], a.i32[4], c_kind ? a.u8[0] : nullptr,
                              a.u8[1], a.i32[5], a.i32[6], a.i32[7], a.cnt);
    h->lazy_override = -1;
    if (rc != GPX_OK) return async_fail(h, a, rc);
    if ((rc = async_kernels_done(h, a)) != GPX_OK) return async_fail(h, a, rc);
    {
      uint8_t* hb[1] = {status};
      const uint8_t* db[1] = {a.u8[1]};
      if ((rc = async_dense_out(h, a, n, 0, nullptr, nullptr, 1, hb, db)) != GPX_OK) return async_fail(h, a, rc);
    }
    a.ncols = 3;
    a.host_col[0] = x_gidx, a.host_col[1] = x_first, a.host_col[2] = x_count;
    a.dev_col[0] = a.i32[5], a.dev_col[1] = a.i32[6], a.dev_col[2] = a.i32[7];
  }
  rc = async_submit(h, a, n > 0, ticket);
  return rc == GPX_OK ? rc : async_fail(h, a, rc);
}

int gpx_engine_wait(gpx_engine* h, gpx_ticket ticket) {
  if (!h) return GPX_EINVAL;
  for (auto& a : h->as) {
    if (!a.busy || a.ticket != ticket) continue;
    HIPCHK(hipEventSynchronize(a.ev_cnt)); /* dense outputs and the count are on the host */
    if (int rc_abort = check_batch(h, 0)) { /* an exchange kernel of this call (or one before it) gave up: nothing to hand over */
      a.busy = false;
      return rc_abort;
    }
    if (a.host_count && !a.direct) {
      const int32_t m = a.ncols ? a.h_cnt[0] : 0;
      *a.host_count = m;
      if (m > 0) { /* exactly m compacted entries, not the capacity */
        for (int k = 0; k < a.ncols; k++) A_OUT(a.host_col[k], a.dev_col[k], (size_t)m * 4);
        if (a.host_kind) A_OUT(a.host_kind, a.dev_kind, (size_t)m);
        SYNC_CHECKED(h, a.s_out);
      }
    }
    a.busy = false;
    return GPX_OK;
  }
  return GPX_EBUSY; /* unknown, or already waited for */
}

} /* extern "C" */
#undef A_OUT

extern "C" {

/* ---- lifecycle ------------------------------------------------------------------ */

int gpx_group_create(gpx_engine* h, int32_t n, const int32_t* gidx, const int32_t* members,
                     const uint8_t* k, const gpx_hri* rows, uint8_t* status) {
  if (!h || n < 0) return GPX_EINVAL;
  if (n == 0) return GPX_OK;
  if (!gidx || !members || !k || !rows) return GPX_EINVAL;
  /* chunked through temporary device buffers: creation is not on the data path */
  const int32_t chunk = 1 << 18;
  int32_t *d_g = nullptr, *d_m = nullptr;
  uint8_t *d_k = nullptr, *d_s = nullptr;
  gpx_hri* d_r = nullptr;
  const int32_t c0 = std::min(n, chunk);
  HIPCHK(hipMalloc((void**)&d_g, (size_t)c0 * 4));
  HIPCHK(hipMalloc((void**)&d_m, (size_t)c0 * 4 * h->cfg.kmax));
  HIPCHK(hipMalloc((void**)&d_k, (size_t)c0));
  HIPCHK(hipMalloc((void**)&d_s, (size_t)c0));
  HIPCHK(hipMalloc((void**)&d_r, (size_t)c0 * sizeof(gpx_hri)));
  int rc = GPX_OK;
  for (int32_t o = 0; o < n && rc == GPX_OK; o += chunk) {
    const int32_t c = std::min(chunk, n - o);
    H2D_B(d_g, gidx + o, (size_t)c * 4);
    H2D_B(d_m, members + (size_t)o * h->cfg.kmax, (size_t)c * 4 * h->cfg.kmax);
    H2D_B(d_k, k + o, (size_t)c);
    H2D_B(d_r, rows + o, (size_t)c * sizeof(gpx_hri));
    LAUNCH(h, "k_group_create", k_group_create, grid_for(c), h->S, c, (const int32_t*)d_g,
           (const int32_t*)d_m, (const uint8_t*)d_k, (const gpx_hri*)d_r, d_s, NameCopies{h->N.rows, (uint8_t*)h->N.tab});
    if (status) D2H(status + o, d_s, (size_t)c);
    HIPCHK(hipStreamSynchronize(h->sB));
    rc = check_batch(h, 0); /* (an exchange kernel that gave up before this call: the table is not to be built on) */
  }
  (void)hipFree(d_g);
  (void)hipFree(d_m);
  (void)hipFree(d_k);
  (void)hipFree(d_s);
  (void)hipFree(d_r);
  return rc;
}

static int retire_impl(gpx_engine* h, int32_t n, const int32_t* gidx, int32_t mode, gpx_hri* rows,
                       uint8_t* status) {
  if (!h || n < 0) return GPX_EINVAL;
  if (n == 0) return GPX_OK;
  if (!gidx) return GPX_EINVAL;
  const int32_t chunk = 1 << 18;
  int32_t* d_g = nullptr;
  uint8_t* d_s = nullptr;
  gpx_hri* d_r = nullptr;
  const int32_t c0 = std::min(n, chunk);
  HIPCHK(hipMalloc((void**)&d_g, (size_t)c0 * 4));
  HIPCHK(hipMalloc((void**)&d_s, (size_t)c0));
  HIPCHK(hipMalloc((void**)&d_r, (size_t)c0 * sizeof(gpx_hri)));
  int rc = GPX_OK;
  for (int32_t o = 0; o < n && rc == GPX_OK; o += chunk) {
    const int32_t c = std::min(chunk, n - o);
    H2D_B(d_g, gidx + o, (size_t)c * 4);
    LAUNCH(h, "k_group_retire", k_group_retire, grid_for(c), h->S, c, (const int32_t*)d_g, mode, d_r,
           d_s, NameCopies{h->N.rows, (uint8_t*)h->N.tab});
    if (rows) D2H(rows + o, d_r, (size_t)c * sizeof(gpx_hri));
    if (status) D2H(status + o, d_s, (size_t)c);
    HIPCHK(hipStreamSynchronize(h->sB));
    rc = check_batch(h, 0); /* a snapshot of a table an exchange kernel left half-written is not a snapshot */
  }
  (void)hipFree(d_g);
  (void)hipFree(d_s);
  (void)hipFree(d_r);
  return rc;
}

int gpx_group_retire(gpx_engine* h, int32_t n, const int32_t* gidx, int32_t mode, gpx_hri* rows,
                     uint8_t* status) {
  if (mode != GPX_RETIRE_PAUSE && mode != GPX_RETIRE_KILL) return GPX_EINVAL;
  return retire_impl(h, n, gidx, mode, rows, status);
}

int gpx_group_snapshot(gpx_engine* h, int32_t n, const int32_t* gidx, gpx_hri* rows,
                       uint8_t* status) {
  if (!rows) return GPX_EINVAL;
  return retire_impl(h, n, gidx, 2, rows, status);
}

/* canonical dump (same word layout as the oracle's orc_group_dump; docs/HISTORY.md §state-dump) */
int gpx_group_dump(gpx_engine* h, int32_t gidx, int32_t* buf, int32_t cap) {
  if (!h || !buf) return GPX_EINVAL;
  SYNC_CHECKED(h, h->sB);
  std::vector<int32_t> w;
  const DevState& S = h->S;
  auto rd32 = [&](const void* base, int64_t idx, int32_t* out) -> hipError_t {
    return hipMemcpy(out, (const char*)base + idx * 4, 4, hipMemcpyDeviceToHost);
  };
  if (gidx < 0 || gidx >= S.G) {
    if (cap < 1) return GPX_ECAPACITY;
    buf[0] = 0;
    return 1;
  }
  int32_t gf_i = 0;
  HIPCHK(rd32(S.g_flags, gidx, &gf_i));
  const uint32_t gf = (uint32_t)gf_i;
  w.push_back((gf & GF_EXISTS) ? 1 : 0);
  if (gf & GF_EXISTS) {
    const int32_t k = (int32_t)GF_K(gf);
    int32_t v = 0;
    HIPCHK(rd32(S.g_version, gidx, &v));
    w.push_back(v);
    w.push_back(k);
    for (int j = 0; j < k; j++) {
      HIPCHK(rd32(S.members, (int64_t)j * S.G + gidx, &v));
      w.push_back(v);
    }
    const void* accf[4] = {S.a_slot, S.a_bnum, S.a_bcoord, S.a_gc};
    for (auto p : accf) {
      HIPCHK(rd32(p, gidx, &v));
      w.push_back(v);
    }
    w.push_back((gf & GF_STOPPED) ? 1 : 0);
    struct Ent {
      int32_t slot, a, b, c;
      uint8_t f;
    };
    /* both rings' flag bytes live in the accepted ring's fourth word (AccView): bits 0-7 / 8-15 */
    auto read_ring = [&](const I4* ring, int shift, std::vector<Ent>& out) -> int {
      for (int32_t x = 0; x < S.W; x++) {
        I4 fe;
        HIPCHK(hipMemcpy(&fe, S.acc_ring + ((int64_t)x * S.G + gidx), sizeof(I4), hipMemcpyDeviceToHost));
        const uint8_t f = (uint8_t)(((uint32_t)fe.w >> shift) & 0xffu);
        if (!(f & RF_PRESENT)) continue;
        I4 r = fe;
        if (ring != S.acc_ring)
          HIPCHK(hipMemcpy(&r, ring + ((int64_t)x * S.G + gidx), sizeof(I4), hipMemcpyDeviceToHost));
        out.push_back(Ent{r.x, r.y, r.z, r.w, f});
      }
      std::sort(out.begin(), out.end(), [](const Ent& p, const Ent& q) { return p.slot < q.slot; });
      return GPX_OK;
    };
    std::vector<Ent> acc, com;
    int rc = read_ring(S.acc_ring, 0, acc);
    if (rc != GPX_OK) return rc;
    rc = read_ring(S.com_ring, CF_SHIFT, com);
    if (rc != GPX_OK) return rc;
    w.push_back((int32_t)acc.size());
    for (auto& en : acc) {
      w.push_back(en.slot);
      w.push_back(en.a);
      w.push_back(en.b);
      w.push_back((en.f & RF_STOP) ? 1 : 0);
    }
    w.push_back((int32_t)com.size());
    for (auto& en : com) {
      w.push_back(en.slot);
      w.push_back(en.a);
      w.push_back(en.b);
      w.push_back(en.c);
      w.push_back((en.f & RF_HASVALUE) ? 1 : 0);
      w.push_back((en.f & RF_STOP) ? 1 : 0);
    }
    w.push_back((gf & GF_HASCOORD) ? 1 : 0);
    if (gf & GF_HASCOORD) {
      int32_t next = 0;
      HIPCHK(rd32(S.c_bnum, gidx, &v));
      w.push_back(v);
      HIPCHK(rd32(S.c_bcoord, gidx, &v));
      w.push_back(v);
      HIPCHK(rd32(S.c_next, gidx, &next));
      w.push_back(next);
      for (int j = 0; j < k; j++) {
        HIPCHK(rd32(S.node_slots, (int64_t)j * S.G + gidx, &v));
        w.push_back(v);
      }
      /* myProposals: slots next-W .. next-1 that are present, ascending signed order */
      std::vector<std::pair<int32_t, uint32_t>> props;
      for (int32_t d = S.W; d >= 1; d--) {
        const int32_t s = (int32_t)((uint32_t)next - (uint32_t)d);
        int32_t ev = 0;
        HIPCHK(rd32(S.p_ring, (int64_t)(s & (S.W - 1)) * S.G + gidx, &ev));
        if ((uint32_t)ev & PR_PRESENT) props.push_back({s, (uint32_t)ev});
      }
      std::sort(props.begin(), props.end(),
                [](const std::pair<int32_t, uint32_t>& p, const std::pair<int32_t, uint32_t>& q) {
                  return p.first < q.first;
                });
      w.push_back((int32_t)props.size());
      for (auto& pr : props) {
        w.push_back(pr.first);
        w.push_back((pr.second & PR_STOP) ? 1 : 0);
        w.push_back((int32_t)(pr.second & 0xffffu));
      }
      /* view change: active flag; while running for coordinator the heard-from mask, the
       * pre-active proposals' handles and the carried-over pvalues */
      const bool preparing = (gf & GF_PREPARING) != 0;
      w.push_back(preparing ? 0 : 1);
      if (preparing) {
        auto rd64 = [&](const int64_t* base, int64_t idx, int64_t* out) -> hipError_t {
          return hipMemcpy(out, base + idx, 8, hipMemcpyDeviceToHost);
        };
        w.push_back(1); /* waitforMyBallot armed */
        HIPCHK(rd32(S.c_wait, gidx, &v));
        w.push_back(v);
        for (auto& pr : props) {
          int64_t hv = 0;
          HIPCHK(rd64(S.p_handle, (int64_t)(pr.first & (S.W - 1)) * S.G + gidx, &hv));
          w.push_back((int32_t)(uint32_t)hv);
          w.push_back((int32_t)((uint64_t)hv >> 32));
        }
        struct Co {
          I4 v;
          int64_t hv;
        };
        std::vector<Co> cos;
        for (int32_t x = 0; x < S.W; x++) {
          Co c;
          HIPCHK(hipMemcpy(&c.v, S.co_ring + ((int64_t)x * S.G + gidx), sizeof(I4), hipMemcpyDeviceToHost));
          if (!(c.v.w & CO_PRESENT)) continue;
          HIPCHK(rd64(S.co_handle, (int64_t)x * S.G + gidx, &c.hv));
          cos.push_back(c);
        }
        std::sort(cos.begin(), cos.end(), [](const Co& p, const Co& q) { return p.v.x < q.v.x; });
        w.push_back((int32_t)cos.size());
        for (auto& c : cos) {
          w.push_back(c.v.x);
          w.push_back(c.v.y);
          w.push_back(c.v.z);
          w.push_back(c.v.w & (GPX_PV_STOP | GPX_PV_NOOP));
          w.push_back((int32_t)(uint32_t)c.hv);
          w.push_back((int32_t)((uint64_t)c.hv >> 32));
        }
      }
    }
  }
  if ((int32_t)w.size() > cap) return GPX_ECAPACITY;
  memcpy(buf, w.data(), w.size() * sizeof(int32_t));
  return (int32_t)w.size();
}

} /* extern "C" */

#include "gpx_wire_host.inc"
#include "gpx_elect_host.inc"
