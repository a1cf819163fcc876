/*
 * gpx_wire.hip.h — CDNA4 (gfx950) kernels of the wire codec (include/gpx_wire.h): device-resident
 * paxosID table, frames -> SoA decode, decisions -> BATCHED_COMMIT frames.
 *
 * Byte / integer work, HBM-bound; one lane per frame (a frame is ~50-100 bytes: neighbouring
 * lanes read neighbouring cache lines).  Every function cites the reference code it replaces
 * (paths relative to /root/reference/src/edu/umass/cs/gigapaxos/).
 */
#pragma once
#include "gpx_kernels.hip.h"
#include "../../include/gpx_wire.h"

/* One row per group (160 bytes):
 *   int32 String.hashCode | uint8 length (0 = no name) | uint8 group exists | 2 pad | int32 version |
 *   name bytes (up to 127) | ... | int32 table slot of the name at byte 144 (-1: none).
 * "exists" and "version" are COPIES of the engine's g_flags & GF_EXISTS / g_version, kept by the only
 * kernels that change them (k_group_create, k_group_retire) and by k_names_bind.
 * The TABLE is open addressing over 128-byte BUCKETS - one cache line each - of four entries: four keys
 * {row + 1, String.hashCode} in the first 32 bytes, then four payloads {length | exists << 8, version, the
 * first 16 name bytes} of 24 bytes.  A lookup reads the keys of its home bucket (one line from HBM: one
 * round trip) and, for a key whose hash matches, that entry's payload (the same line: an L1 hit); with
 * at most one entry per bucket on average, 99.6 % of the names are in their home bucket, so the
 * instance test of PaxosManager.handlePaxosPacket (getInstance + version, PaxosManager.java:1153-1162)
 * is one HBM access for almost every frame of a WAVE - which is what counts: round 3's timeline
 * (scripts/ubench/wire_trace.sh) showed the entry's round trip at 2 us and the wave's wait for its lane
 * with the LONGEST probe sequence at 9 us under linear probing over single entries at load 0.5 (a
 * sequence of four or five entries somewhere among 64 lanes is the rule).  Names longer than 16 bytes
 * compare their tail in the row.  Inserts take the first empty way in probe order and never reuse a
 * tombstone, so a lookup ends at the first bucket that still has an empty way. */
#define NM_STRIDE 160
#define NM_EXISTS 5
#define NM_VERSION 8
#define NM_NAME 12 /* offset of the name bytes inside a row */
#define NM_SLOT 144 /* offset of the row's table entry index (bucket * 4 + way) */
#define NM_HOT 16  /* name bytes inside a table entry */
#define NM_WAYS 4
#define NM_BUCKET 128 /* bytes */
#define NM_KEYS 32    /* bytes of keys at the head of a bucket */
#define NM_PAYLOAD 24 /* bytes per payload */
static_assert(NM_BUCKET == GPX_NAME_BUCKET_BYTES && NM_KEYS == GPX_NAME_BUCKET_KEYS && NM_PAYLOAD == GPX_NAME_PAYLOAD_BYTES &&
                  NM_STRIDE == GPX_NAME_ROW_STRIDE && NM_EXISTS == GPX_NAME_ROW_EXISTS &&
                  NM_VERSION == GPX_NAME_ROW_VERSION && NM_SLOT == GPX_NAME_ROW_SLOT,
              "k_group_create / k_group_retire write the name rows and table entries");
#define GPX_W_MAX_DEPTH 6   /* nesting of batched RequestPackets the walker follows */
#define GPX_W_MAX_SEG 256   /* rows of one group a BATCHED_COMMIT frame may span (>= 2 * window) */

/* device mirror of PaxosManager.pinstances' key side: open addressing over the paxosID bytes */
struct DevNames {
  int32_t cap;    /* table entries = 4 * buckets; buckets a power of two >= G */
  uint8_t* tab;   /* [cap / 4][NM_BUCKET] */
  uint8_t* rows;  /* [G][NM_STRIDE] */
  __device__ __forceinline__ uint8_t* row(int32_t g) const { return rows + (int64_t)g * NM_STRIDE; }
  __device__ __forceinline__ int32_t hash(int32_t g) const { return *(const int32_t*)row(g); }
  __device__ __forceinline__ int32_t len(int32_t g) const { return (int32_t)row(g)[4]; }
  __device__ __forceinline__ const uint8_t* name(int32_t g) const { return row(g) + NM_NAME; }
  __device__ __forceinline__ int32_t& slot(int32_t g) const { return *(int32_t*)(row(g) + NM_SLOT); }
  __device__ __forceinline__ uint32_t bmask() const { return ((uint32_t)cap >> 2) - 1u; }
  __device__ __forceinline__ uint8_t* bucket(uint32_t b) const { return tab + (int64_t)b * NM_BUCKET; }
  __device__ __forceinline__ int32_t* key(int32_t s) const { /* {row + 1 (0 empty, -1 tombstone), hashCode} */
    return (int32_t*)(bucket((uint32_t)s >> 2) + (s & 3) * 8);
  }
  __device__ __forceinline__ uint32_t* payload(int32_t s) const { /* {length | exists << 8, version, name[4]} */
    return (uint32_t*)(bucket((uint32_t)s >> 2) + NM_KEYS + (s & 3) * NM_PAYLOAD);
  }
  /* the rest of entry s, from row g (which holds the whole name and the group's copies) */
  __device__ __forceinline__ void fill(int32_t s, int32_t g) const {
    const uint8_t* r = row(g);
    uint32_t q[4] = {0, 0, 0, 0};
    const int32_t len = (int32_t)r[4];
    for (int32_t i = 0; i < NM_HOT && i < len; i++) q[i >> 2] |= (uint32_t)r[NM_NAME + i] << (8 * (i & 3));
    key(s)[1] = *(const int32_t*)r;
    uint32_t* pl = payload(s);
    pl[0] = (uint32_t)len | ((uint32_t)r[NM_EXISTS] << 8);
    pl[1] = *(const uint32_t*)(r + NM_VERSION);
    pl[2] = q[0];
    pl[3] = q[1];
    pl[4] = q[2];
    pl[5] = q[3];
    slot(g) = s;
  }
};

/* Frame bytes are parsed either in place (generic pointer) or from the LDS staging area.  The
 * parsers are templates over the pointer type so that the staged case compiles to ds_read_u8
 * (a generic pointer into LDS becomes a flat load: every byte then waits on both memory counters) */
typedef const uint8_t* GenBytes;
typedef const __attribute__((address_space(3))) uint8_t* LdsBytes;

__device__ __forceinline__ uint32_t w_fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85ebca6bu;
  h ^= h >> 13;
  h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}
/* String.hashCode of an ISO-8859-1 string: s[0]*31^(n-1) + ... + s[n-1] (int arithmetic) */
template <class BP>
__device__ __forceinline__ int32_t w_java_hash(BP p, int32_t n) {
  uint32_t h = 0;
  for (int32_t i = 0; i < n; i++) h = 31u * h + (uint32_t)p[i];
  return (int32_t)h;
}
template <class BP>
__device__ __forceinline__ bool w_bytes_eq(const uint8_t* a, BP b, int32_t n) {
  for (int32_t i = 0; i < n; i++)
    if (a[i] != b[i]) return false;
  return true;
}
/* one whole bucket in registers: eight 16-byte loads of one 128-byte line, all requested together (the
 * payload as a second, dependent access - even of the same line - was a second round trip for the wave: the
 * L1 holds 128 lines, the CU's waves have 1,500 in flight) */
struct NameKeys {
  uint4 q[8];
  template <int W>
  __device__ __forceinline__ int32_t v() const { return (int32_t)(W == 0 ? q[0].x : W == 1 ? q[0].z : W == 2 ? q[1].x : q[1].z); }
  template <int W>
  __device__ __forceinline__ int32_t h() const { return (int32_t)(W == 0 ? q[0].y : W == 1 ? q[0].w : W == 2 ? q[1].y : q[1].w); }
  /* dword D (0 .. 5) of payload W: dword 8 + 6 W + D of the bucket */
  template <int I>
  __device__ __forceinline__ uint32_t dw() const {
    return (I & 3) == 0 ? q[I >> 2].x : (I & 3) == 1 ? q[I >> 2].y : (I & 3) == 2 ? q[I >> 2].z : q[I >> 2].w;
  }
};
__device__ __forceinline__ NameKeys names_keys(const DevNames& N, uint32_t b) {
  const uint4* kp = (const uint4*)N.bucket(b);
  NameKeys K;
#pragma unroll
  for (int i = 0; i < 8; i++) K.q[i] = kp[i];
  return K;
}
/* way W of the bucket in K: -2 not this name, else its row (exists / version filled in) */
template <int W, class BP>
__device__ __forceinline__ int32_t names_way(const DevNames& N, const NameKeys& K, BP p, int32_t len, int32_t hash,
                                             const uint32_t* q, bool* exists, int32_t* version) {
  const int32_t v = K.v<W>();
  if (!(v > 0 && K.h<W>() == hash)) return -2;
  const uint32_t meta = K.dw<8 + 6 * W>();
  const int32_t ver = (int32_t)K.dw<8 + 6 * W + 1>();
  if ((int32_t)(meta & 0xffu) != len || K.dw<8 + 6 * W + 2>() != q[0] || K.dw<8 + 6 * W + 3>() != q[1] ||
      K.dw<8 + 6 * W + 4>() != q[2] || K.dw<8 + 6 * W + 5>() != q[3])
    return -2;
  const int32_t g = v - 1;
  /* written BEFORE the tail is compared (a caller looks at them only when a row comes back): with the two
   * stores behind the loop they were lost for names longer than 16 bytes (hipcc 7.2; the decode fuzz caught it) */
  if (exists) *exists = ((meta >> 8) & 0xffu) != 0;
  if (version) *version = ver;
  for (int32_t i = NM_HOT; i < len; i++)
    if (N.name(g)[i] != p[i]) return -2;
  return g;
}
/* MultiArrayMap.get(paxosID) (PaxosManager.getInstance, PaxosManager.java:1816-1832): row or -1, walking the
 * buckets from the name's home bucket on.  The first 16 name bytes of the probe are packed into four dwords
 * once. */
template <class BP>
__device__ __forceinline__ int32_t names_find(const DevNames& N, BP p, int32_t len, int32_t hash,
                                              bool* exists = nullptr, int32_t* version = nullptr) {
  if (!N.tab) return -1;
  uint32_t b = w_fmix32((uint32_t)hash) & N.bmask();
  uint32_t q[4] = {0, 0, 0, 0};
#pragma unroll
  for (int32_t i = 0; i < NM_HOT; i++)
    if (i < len) q[i >> 2] |= (uint32_t)p[i] << (8 * (i & 3));
  const uint32_t bm = N.bmask();
  for (uint32_t probe = 0; probe <= bm; probe++) {
    const NameKeys K = names_keys(N, b);
    int32_t g;
    if ((g = names_way<0>(N, K, p, len, hash, q, exists, version)) != -2) return g;
    if ((g = names_way<1>(N, K, p, len, hash, q, exists, version)) != -2) return g;
    if ((g = names_way<2>(N, K, p, len, hash, q, exists, version)) != -2) return g;
    if ((g = names_way<3>(N, K, p, len, hash, q, exists, version)) != -2) return g;
    /* inserts take the first empty way in probe order: the name is not beyond an empty way */
    if (K.v<0>() == 0 || K.v<1>() == 0 || K.v<2>() == 0 || K.v<3>() == 0) return -1;
    b = (b + 1) & bm;
  }
  return -1;
}
__global__ __launch_bounds__(GPX_BLOCK) void k_names_bind(DevState S, DevNames N, int32_t G, int32_t n,
                                                         const uint8_t* __restrict__ names,
                                                         const int32_t* __restrict__ name_off,
                                                         const int32_t* __restrict__ gidx,
                                                         uint8_t* __restrict__ status) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx[i];
  const int32_t len = name_off[i + 1] - name_off[i];
  if ((uint32_t)g >= (uint32_t)G || len < 1 || len > GPX_W_MAX_NAME) {
    status[i] = GPX_S_NOGROUP;
    return;
  }
  if (N.len(g) != 0) { /* the row already carries a name */
    status[i] = GPX_S_EXISTS;
    return;
  }
  const uint8_t* p = names + name_off[i];
  const int32_t h = w_java_hash(p, len);
  uint8_t* row = N.row(g);
  for (int32_t b = 0; b < len; b++) row[NM_NAME + b] = p[b];
  *(int32_t*)row = h;
  row[NM_EXISTS] = (g < S.G && (S.g_flags[g] & GF_EXISTS)) ? 1 : 0;
  *(int32_t*)(row + NM_VERSION) = g < S.G ? S.g_version[g] : 0;
  row[4] = (uint8_t)len;
  __threadfence(); /* the row is complete before the table can point at it */
  const uint32_t mask = (uint32_t)N.cap - 1u;
  uint32_t s = (w_fmix32((uint32_t)h) & N.bmask()) << 2; /* way 0 of the home bucket; then way by way, bucket by bucket */
  for (int32_t probe = 0; probe < N.cap; probe++) {
    int32_t v = N.key((int32_t)s)[0];
    if (v == 0) {
      v = atomicCAS(&N.key((int32_t)s)[0], 0, g + 1);
      if (v == 0) { /* the entry is this row's: the rest of it (readers are later kernels) */
        N.fill((int32_t)s, g);
        status[i] = GPX_S_OK;
        return;
      }
      __threadfence();
    }
    if (v > 0 && v - 1 != g) {
      const int32_t o = v - 1;
      if (N.hash(o) == h && N.len(o) == len && w_bytes_eq(N.name(o), p, len)) {
        N.row(g)[4] = 0; /* name already bound to another row */
        status[i] = GPX_S_EXISTS;
        return;
      }
    }
    s = (s + 1) & mask;
  }
  N.row(g)[4] = 0;
  status[i] = GPX_S_NOGROUP; /* table full: cannot happen with cap >= 2 G */
}

__global__ __launch_bounds__(GPX_BLOCK) void k_names_unbind(DevNames N, int32_t G, int32_t n,
                                                           const int32_t* __restrict__ gidx,
                                                           uint8_t* __restrict__ status) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx[i];
  if ((uint32_t)g >= (uint32_t)G || N.len(g) == 0) {
    if (status) status[i] = GPX_S_NOGROUP;
    return;
  }
  const int32_t s = N.slot(g);
  if (s >= 0 && s < N.cap && N.key(s)[0] == g + 1) N.key(s)[0] = -1; /* tombstone: later probes walk over it */
  N.slot(g) = -1;
  N.row(g)[4] = 0;
  if (status) status[i] = GPX_S_OK;
}

/* rebuild after many unbinds: table cleared by the host, every named row re-inserted */
__global__ __launch_bounds__(GPX_BLOCK) void k_names_reinsert(DevNames N, int32_t G) {
  const int32_t g = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (g >= G || N.len(g) == 0) return;
  const uint32_t mask = (uint32_t)N.cap - 1u;
  uint32_t s = (w_fmix32((uint32_t)N.hash(g)) & N.bmask()) << 2;
  for (int32_t probe = 0; probe < N.cap; probe++) {
    if (N.key((int32_t)s)[0] == 0 && atomicCAS(&N.key((int32_t)s)[0], 0, g + 1) == 0) {
      N.fill((int32_t)s, g);
      return;
    }
    s = (s + 1) & mask;
  }
}

__global__ __launch_bounds__(GPX_BLOCK) void k_names_lookup(DevNames N, int32_t n,
                                                           const uint8_t* __restrict__ names,
                                                           const int32_t* __restrict__ name_off,
                                                           int32_t* __restrict__ out) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t len = name_off[i + 1] - name_off[i];
  const uint8_t* p = names + name_off[i];
  out[i] = (len >= 1 && len <= GPX_W_MAX_NAME) ? names_find(N, p, len, w_java_hash(p, len)) : -1;
}

/* PISM.roundRobinCoordinator (PaxosInstanceStateMachine.java:2251-2256) */
__global__ __launch_bounds__(GPX_BLOCK) void k_names_coordinator(DevState S, DevNames N, int32_t n,
                                                                const int32_t* __restrict__ gidx,
                                                                int32_t ballotnum,
                                                                int32_t* __restrict__ out) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx[i];
  int32_t r = INT32_MIN;
  if ((uint32_t)g < (uint32_t)S.G && (S.g_flags[g] & GF_EXISTS) && N.tab && N.len(g) != 0) {
    const int32_t k = (int32_t)GF_K(S.g_flags[g]);
    const int32_t x = (int32_t)((uint32_t)ballotnum + (uint32_t)N.hash(g));
    const int32_t ax = x < 0 ? (int32_t)(0u - (uint32_t)x) : x; /* Math.abs */
    const int32_t idx = ax % k;                                   /* sign follows the dividend */
    if (idx >= 0) r = S.members[(int64_t)idx * S.G + g];
  }
  out[i] = r;
}

/* ------------------------------------------------------------------------- */
/* decode                                                                       */

/* java.nio.ByteBuffer.getInt at an arbitrary byte position (big-endian) */
template <class BP>
__device__ __forceinline__ int32_t w_be32(BP p) {
  return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) |
                   (uint32_t)p[3]);
}
/* the staged case: two aligned LDS words + v_alignbyte instead of four ds_read_u8 (the staging area
 * always has one readable word past the tile's last byte) */
#ifndef GPX_NO_LDS_BE32
template <>
__device__ __forceinline__ int32_t w_be32<LdsBytes>(LdsBytes p) {
  typedef const __attribute__((address_space(3))) uint32_t* LdsWords;
  const uint32_t a = (uint32_t)(uintptr_t)p;
  LdsWords q = (LdsWords)(uintptr_t)(a & ~3u);
  const uint32_t lo = q[0], hi = q[1];
  return (int32_t)__builtin_bswap32(__builtin_amdgcn_alignbyte(hi, lo, a & 3u));
}
#endif
template <class BP>
__device__ __forceinline__ int64_t w_be64(BP p) {
  return (int64_t)(((uint64_t)(uint32_t)w_be32(p) << 32) | (uint64_t)(uint32_t)w_be32(p + 4));
}

/* PaxosPacket.PaxosPacketType.getPaxosPacketType(int) != null (PaxosPacket.java:202-300) */
__device__ __forceinline__ bool w_known_type(int32_t t) {
  return (t >= 1 && t <= 9) || t == 13 || t == 21 || t == 23 || (t >= 32 && t <= 37) || t == 90 ||
         t == 9999;
}

/* One RequestPacket body starting at its PaxosPacket header, inside the buffer [.., end):
 * RequestPacket(ByteBuffer) up to and including numBatched (RequestPacket.java:956-1005).
 * Returns false where the constructor would throw. */
template <class BP>
__device__ __forceinline__ bool w_request_fixed(BP p, int64_t& pos, int64_t end,
                                                int32_t& num_batched, bool& stop, int64_t& req_id) {
  if (pos + 13 > end) return false;
  const int32_t idl = (int32_t)(int8_t)p[pos + 12];
  if (idl < 0 || pos + 13 + idl > end) return false;
  pos += 13 + idl;
  /* requestID 8, stop 1, addresses 12, entryReplica 4, entryTime 8, shouldReturn 1,
   * forwardCount 4, broadcasted 1 = 39, then int digestLength */
  if (pos + 39 + 4 > end) return false;
  req_id = w_be64(p + pos);
  stop = p[pos + 8] == 1;
  pos += 39;
  const int32_t dl = w_be32(p + pos);
  pos += 4;
  if (dl > 0) {
    if (pos + dl > end) return false;
    pos += dl;
  }
#pragma unroll
  for (int q = 0; q < 2; q++) { /* requestValue, responseValue: new byte[len] */
    if (pos + 4 > end) return false;
    const int32_t vl = w_be32(p + pos);
    pos += 4;
    if (vl < 0 || pos + vl > end) return false;
    pos += vl;
  }
  if (pos + 4 > end) return false;
  num_batched = w_be32(p + pos);
  pos += 4;
  return num_batched >= 0; /* new RequestPacket[numBatched] */
}

/* Whole RequestPacket incl. its batched sub-requests, each a complete RequestPacket byte array
 * parsed by its own constructor (RequestPacket.java:1006-1019).  On success pos = first byte after
 * the request (where an ACCEPT's slot / ballot tail starts); stop_any = isStopRequest()
 * (RequestPacket.java:1069-1080). */
template <class BP>
__device__ bool w_walk_request(BP p, int64_t& pos, int64_t end, bool& stop_any, int64_t& req_id) {
  int64_t bend[GPX_W_MAX_DEPTH + 1];
  int32_t rem[GPX_W_MAX_DEPTH + 1];
  int32_t nb = 0;
  bool st = false;
  if (!w_request_fixed(p, pos, end, nb, st, req_id)) return false;
  stop_any = st;
  bend[0] = end;
  int d = 0;
  if (nb > 0) {
    d = 1;
    rem[1] = nb;
  }
  while (d >= 1) {
    /* next element of the list at level d; it lives in the buffer of level d - 1 */
    rem[d]--;
    if (pos + 4 > bend[d - 1]) return false;
    const int32_t len = w_be32(p + pos);
    pos += 4;
    if (len < 0 || pos + len > bend[d - 1]) return false;
    bend[d] = pos + len;
    int64_t rid;
    if (!w_request_fixed(p, pos, bend[d], nb, st, rid)) return false;
    stop_any = stop_any || st;
    if (nb > 0) {
      if (d + 1 > GPX_W_MAX_DEPTH) return false; /* engine limit on nesting */
      d++;
      rem[d] = nb;
      continue;
    }
    pos = bend[d]; /* element done (trailing bytes of an element are ignored) */
    while (d >= 1 && rem[d] == 0) { /* a finished list completes the element that holds it */
      d--;
      if (d >= 1) pos = bend[d];
    }
  }
  return true;
}

struct WFrame {
  int32_t st, type, gidx, cnt, cls; /* cls: 0 votes 1 commits 2 accepts 3 requests, -1 none */
  int32_t hdr;                      /* bytes of the PaxosPacket header incl. the paxosID */
  int32_t raw_n;                    /* entries of the slot list on the wire */
  bool ascending;
  bool stop;
  int64_t req_id, tail;             /* ACCEPT: byte position of the slot / ballot tail */
#ifdef GPX_WD_TRACE
  unsigned long long t_pre;
#endif
};

/* strictly ascending signed order = what TreeMap / TreeSet iteration put on the wire */
template <class BP>
__device__ __forceinline__ bool w_list_ascending(BP e, int32_t n, int32_t stride) {
  int32_t prev = 0;
  for (int32_t j = 0; j < n; j++) {
    const int32_t s = w_be32(e + (int64_t)j * stride);
    if (j > 0 && !(prev < s)) return false;
    prev = s;
  }
  return true;
}
/* TreeSet size of a non-ascending list (n <= GPX_W_MAX_UNSORTED) */
template <class BP>
__device__ __forceinline__ int32_t w_list_distinct(BP e, int32_t n, int32_t stride) {
  int32_t c = 0;
  for (int32_t j = 0; j < n; j++) {
    const int32_t s = w_be32(e + (int64_t)j * stride);
    bool seen = false;
    for (int32_t q = 0; q < j && !seen; q++) seen = w_be32(e + (int64_t)q * stride) == s;
    c += !seen;
  }
  return c;
}

/* PaxosPacketDemultiplexerFast.toPaxosPacket (paxosutil/PaxosPacketDemultiplexerFast.java:66-103)
 * + the four ByteBuffer constructors + PaxosManager.handlePaxosPacket's getInstance / version test
 * (PaxosManager.java:1153-1162), for one frame. */
template <class BP>
__device__ void w_parse(const DevState& S, const DevNames& N, BP p, int64_t L, WFrame& f) {
  f.st = GPX_W_MALFORMED;
  f.type = -1;
  f.gidx = -1;
  f.cnt = 0;
  f.cls = -1;
  f.hdr = 0;
  f.raw_n = 0;
  f.ascending = true;
  f.stop = false;
  f.req_id = 0;
  f.tail = 0;
  if (L < 8) return; /* BufferUnderflowException */
  if (w_be32(p) != GPX_WT_PAXOS_PACKET) return; /* type == null -> fatal(bytes) */
  const int32_t t = w_be32(p + 4);
  if (!w_known_type(t)) return;
  if (t != GPX_WT_REQUEST && t != GPX_WT_ACCEPT && t != GPX_WT_BATCHED_COMMIT &&
      t != GPX_WT_BATCHED_ACCEPT_REPLY) {
    f.st = GPX_W_UNSUPPORTED; /* assert(false): paxosPacket stays null */
    f.type = t;
    return;
  }
  /* PaxosPacket(ByteBuffer) (PaxosPacket.java:443-458) */
  if (L < 13) return;
  const int32_t version = w_be32(p + 8);
  const int32_t idl = (int32_t)(int8_t)p[12];
  if (idl < 0 || 13 + (int64_t)idl > L) return; /* NegativeArraySize / underflow */
  const int64_t hdr = 13 + idl;
  f.hdr = (int32_t)hdr;
  if (t == GPX_WT_BATCHED_ACCEPT_REPLY) {
    if (idl == 0) return;          /* getPaxosID().getBytes on null (AcceptReplyPacket.java:153) */
    if (hdr + 29 + 4 > L) return;  /* AcceptReplyPacket fields + numSlots */
    const int32_t n = w_be32(p + hdr + 29);
    if (n > 0) {
      if (hdr + 33 + 12 * (int64_t)n > L) return;
      f.raw_n = n;
      f.ascending = w_list_ascending(p + hdr + 33, n, 12);
      if (!f.ascending && n > GPX_W_MAX_UNSORTED) return;
      f.cnt = f.ascending ? n : w_list_distinct(p + hdr + 33, n, 12);
    }
    f.cls = 0;
  } else if (t == GPX_WT_BATCHED_COMMIT) {
    if (hdr + 16 > L) return; /* ballot, medianCheckpointedSlot, numSlots */
    const int32_t n = w_be32(p + hdr + 12);
    int64_t gpos = hdr + 16;
    if (n > 0) {
      if (hdr + 16 + 4 * (int64_t)n > L) return;
      f.raw_n = n;
      f.ascending = w_list_ascending(p + hdr + 16, n, 4);
      if (!f.ascending && n > GPX_W_MAX_UNSORTED) return;
      f.cnt = f.ascending ? n : w_list_distinct(p + hdr + 16, n, 4);
      gpos += 4 * (int64_t)n;
    }
    if (gpos + 4 > L) return; /* groupSize */
    const int32_t gs = w_be32(p + gpos);
    if (gs > 0 && gpos + 4 + 4 * (int64_t)gs > L) return;
    f.cls = 1;
  } else {
    int64_t pos = 0;
    bool stop_any = false;
    int64_t rid = 0;
    if (!w_walk_request(p, pos, L, stop_any, rid)) return;
    f.stop = stop_any;
    f.req_id = rid;
    if (t == GPX_WT_ACCEPT) {
      if (pos + 22 > L) return; /* slot 4, ballot 8, recovery 1, median 4, noCoalesce 1, sender 4 */
      f.tail = pos;
      f.cls = 2;
    } else {
      f.cls = 3;
    }
    f.cnt = 1;
  }
  f.type = t;
  /* getInstance(paxosID) and the version check */
  int32_t g = -1;
  bool exists = false;
  int32_t gver = 0;
#ifdef GPX_WD_NOLOOKUP
  if (idl > 0) {
    g = (int32_t)(w_fmix32((uint32_t)w_java_hash(p + 13, idl)) & (uint32_t)(S.G - 1));
    exists = true;
    gver = version;
  }
#else
#ifdef GPX_WD_TRACE
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); /* the parse's own loads are back */
  f.t_pre = wall_clock64();
#endif
  if (idl > 0) g = names_find(N, p + 13, idl, w_java_hash(p + 13, idl), &exists, &gver);
#endif
  if (g < 0 || !exists) {
    f.st = GPX_W_NOGROUP;
    f.cnt = 0;
    return;
  }
  f.gidx = g;
  if (gver != version) {
    f.st = GPX_W_VERSION;
    f.cnt = 0;
#ifdef GPX_WD_DBGVER
    f.gidx = 1000000 + gver * 1000 + version;
#endif
    return;
  }
  f.st = GPX_W_OK;
}

/* the caller's column structs, by value (device pointers; cap = 0 when the struct was NULL) */
struct WireOut {
  gpx_wire_votes V;
  gpx_wire_commits C;
  gpx_wire_accepts A;
  gpx_wire_requests R;
};

/* The 256 frames of a workgroup's tile are contiguous in the burst: copy their bytes to LDS with
 * coalesced dword loads and let every lane parse its frame from there (a lane walking ~70 bytes
 * of its own frame in global memory touches a different cache line than its neighbours on every
 * load: measured 8x slower).  A tile longer than the staging area (frames with request values) is
 * staged in overlapping WINDOWS of GPX_W_STAGE_BYTES at half-window steps: a frame of at most half a
 * window lies entirely inside one of them; only a frame longer than that is parsed in place.
 * fn(staged, pos) runs exactly once per live lane, with pos = the frame's byte offset into the
 * staging area (staged) or into `frames`; fn must not contain a barrier. */
#define GPX_W_STAGE_BYTES (24 * 1024)
#ifndef GPX_WIRE_WINDOWS
#define GPX_WIRE_WINDOWS 0
#endif
#define GPX_W_STAGE_WORDS (GPX_W_STAGE_BYTES / 4 + 4 + 1) /* + the lead of an unaligned tile + one readable word */
/* Copy the bytes [a0, a0 + nbytes) (a0 dword-aligned) into the 16-byte-aligned staging area.  EVERY load
 * of a lane - its (up to seven) 16-byte chunks, the odd words before the first and after the last whole
 * chunk, a tail byte - is issued before the first is stored, so the copy is ONE memory round trip (the
 * timeline of round 3, scripts/ubench/wire_trace.sh: with four chunks in flight per pass and the odd
 * words in passes of their own the staging of a 512-frame tile was five dependent round trips, 12 of
 * the tile's 47 us).  LDS byte lead + k = byte k, where lead = a0 & 15 keeps 16-byte chunks aligned on
 * both sides; nothing before a0 or past the last byte is read.  Returns lead; the caller's barrier
 * follows. */
/* The chunk array is ZEROED before the loads (round 3 ISA reading, profiles/r03_wire_stage_isa.txt; measured in round
 * 4, profiles/r04_pending_visit.txt): hipcc 7.2 keeps v[] in SCRATCH when its elements are only assigned under the
 * bounds test, and waits for every chunk before the next load - seven round trips instead of the one the loop is
 * written for.  With the elements zeroed first the seven loads are issued back to back and one wait precedes the
 * stores: decode of 2 M frames 0.248 -> 0.197 ms.  (Also: the burst lies in GLOBAL memory - an address made from an
 * integer is generic, and generic loads count against the LDS counter too, so that every LDS wait waits for them as
 * well: the loads name address space 1.) */
template <int BLOCK = GPX_BLOCK>
__device__ __forceinline__ int32_t wire_stage(uint32_t* lds, uintptr_t a0, int64_t nbytes) {
  constexpr int NCH = GPX_W_STAGE_BYTES / 16 / GPX_BLOCK + 1; /* chunks per lane of a full staging area */
  const uintptr_t a16 = a0 & ~(uintptr_t)15;
  const int32_t lead = (int32_t)(a0 - a16);
  const int32_t total = lead + (int32_t)nbytes;
  const int32_t c_lo = (lead + 15) >> 4, c_hi = total >> 4; /* whole chunks inside [lead, total) */
  const uint4* src16 = (const uint4*)a16;
  uint4* dst16 = (uint4*)lds;
  const uint32_t* src = (const uint32_t*)a16;
  const int32_t w_hi = total >> 2;
  const int32_t head_end = c_lo * 4 < w_hi ? c_lo * 4 : w_hi;
  const int32_t hw = (lead >> 2) + (int32_t)threadIdx.x;                     /* a word before the first whole chunk */
  const int32_t t_lo = c_hi * 4 > head_end ? c_hi * 4 : head_end;
  const int32_t tw = t_lo + (int32_t)threadIdx.x;                            /* a word after the last one */
  const bool tbyte = (int32_t)threadIdx.x < (total & 3);                     /* a tail byte */
  for (int32_t c0 = c_lo; c0 < c_hi; c0 += NCH * BLOCK) {                    /* one pass for a staging area's worth */
    uint4 v[NCH];
    uint32_t hv = 0, tv = 0;
    uint8_t bv = 0;
    const bool odd = c0 == c_lo;
#pragma unroll
    for (int k = 0; k < NCH; k++) {
      const int32_t c = c0 + k * BLOCK + (int32_t)threadIdx.x;
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      v[k] = make_uint4(0u, 0u, 0u, 0u);
      if (c < c_hi) {
        const u32x4 t = ((const __attribute__((address_space(1))) u32x4*)a16)[c];
        v[k] = make_uint4(t.x, t.y, t.z, t.w);
      }
    }
    if (odd && hw < head_end) hv = src[hw];
    if (odd && tw < w_hi) tv = src[tw];
    if (odd && tbyte) bv = ((const uint8_t*)src)[(w_hi << 2) + threadIdx.x];
#pragma unroll
    for (int k = 0; k < NCH; k++) {
      const int32_t c = c0 + k * BLOCK + (int32_t)threadIdx.x;
      if (c < c_hi) dst16[c] = v[k];
    }
    if (odd && hw < head_end) lds[hw] = hv;
    if (odd && tw < w_hi) lds[tw] = tv;
    if (odd && tbyte) ((uint8_t*)lds)[(w_hi << 2) + threadIdx.x] = bv;
  }
  if (c_lo >= c_hi) { /* less than one whole chunk: only the odd words */
    if (hw < head_end) lds[hw] = src[hw];
    if (tw < w_hi) lds[tw] = src[tw];
    if (tbyte) ((uint8_t*)lds)[(w_hi << 2) + threadIdx.x] = ((const uint8_t*)src)[(w_hi << 2) + threadIdx.x];
  }
  return lead;
}
/* the r-th smallest distinct slot of a non-ascending list: smallest value greater than `last` */
template <class BP>
__device__ __forceinline__ int32_t w_next_slot(BP e, int32_t n, int32_t stride, int64_t last) {
  int64_t best = (int64_t)1 << 40;
  for (int32_t j = 0; j < n; j++) {
    const int64_t s = w_be32(e + (int64_t)j * stride);
    if (s > last && s < best) best = s;
  }
  return (int32_t)best;
}

/* every frame writes its records at its class offset (frame order; slots ascending).
 * Nothing is validated again: pass 1 left the frame's class, group row, list order and tail. */
/* the records of one frame, written at its class offset */
template <class BP>
__device__ __forceinline__ void wire_emit(const WireOut& O, int32_t i, BP p, int32_t g, int64_t tail,
                                          int32_t flags, int32_t cls, int32_t cnt, int32_t off) {
  const int64_t hdr = 13 + (int64_t)p[12];
  const bool ascending = (flags & 8) != 0;
  if (cls == 0) {
    /* BatchedAcceptReply: acceptor, ballot, (slotNumber), maxCheckpointedSlot once; slots
     * (PISM.handleBatchedAcceptReply iterates getAcceptedSlots(): TreeMap key order) */
    const int32_t acceptor = w_be32(p + hdr), bnum = w_be32(p + hdr + 4);
    const int32_t bcoord = w_be32(p + hdr + 8), maxcp = w_be32(p + hdr + 16);
    const int32_t raw_n = w_be32(p + hdr + 29);
    const BP e = p + hdr + 33;
    int64_t last = -((int64_t)1 << 40);
    for (int32_t r = 0; r < cnt; r++) {
      const int32_t s = ascending ? w_be32(e + (int64_t)r * 12) : w_next_slot(e, raw_n, 12, last);
      last = s;
      const int32_t o = off + r;
      O.V.gidx[o] = g;
      O.V.bnum[o] = bnum;
      O.V.bcoord[o] = bcoord;
      O.V.slot[o] = s;
      O.V.acceptor[o] = acceptor;
      O.V.max_cp[o] = maxcp;
      if (O.V.frame) O.V.frame[o] = i;
    }
  } else if (cls == 1) {
    const int32_t bnum = w_be32(p + hdr), bcoord = w_be32(p + hdr + 4);
    const int32_t median = w_be32(p + hdr + 8);
    const int32_t raw_n = w_be32(p + hdr + 12);
    const BP e = p + hdr + 16;
    int64_t last = -((int64_t)1 << 40);
    for (int32_t r = 0; r < cnt; r++) {
      const int32_t s = ascending ? w_be32(e + (int64_t)r * 4) : w_next_slot(e, raw_n, 4, last);
      last = s;
      const int32_t o = off + r;
      O.C.gidx[o] = g;
      O.C.bnum[o] = bnum;
      O.C.bcoord[o] = bcoord;
      O.C.slot[o] = s;
      O.C.median_cp[o] = median;
      O.C.kind[o] = 0; /* meta-commit: PISM.handleBatchedCommit finds the value in the stored ACCEPT */
      if (O.C.frame) O.C.frame[o] = i;
    }
  } else if (cls == 2) {
    const BP t = p + tail;
    O.A.gidx[off] = g;
    O.A.slot[off] = w_be32(t);
    O.A.bnum[off] = w_be32(t + 4);
    O.A.bcoord[off] = w_be32(t + 8);
    O.A.median_cp[off] = w_be32(t + 13);
    O.A.sender[off] = w_be32(t + 18);
    O.A.flags[off] = (flags & 16) ? GPX_A_STOP : 0;
    O.A.req_id[off] = w_be64(p + hdr);
    if (O.A.frame) O.A.frame[off] = i;
  } else {
    O.R.gidx[off] = g;
    O.R.is_stop[off] = (flags & 16) ? 1 : 0;
    O.R.req_id[off] = w_be64(p + hdr);
    if (O.R.frame) O.R.frame[off] = i;
  }
}

/* decode in ONE launch: parse, place and emit while the tile's bytes are still in LDS.
 * The class offsets of a tile (records of every earlier frame, per class) come from a decoupled
 * look-back over per-tile words {epoch | state | count}: a tile publishes its own counts (AGGREGATE)
 * as soon as it has parsed, then wave c of the workgroup walks back over class c's words, 64 tiles
 * per step, until it meets a tile that already knows its inclusive PREFIX.  Tiles are handed out by a
 * ticket, so every tile a workgroup waits for has been taken by a workgroup that is running (or
 * done): no deadlock whatever the dispatch order.  A word carries its count AND its state, and
 * nothing else travels between tiles, so the loads and stores are RELAXED agent-scope atomics: with
 * release / acquire every store wrote the XCD's L2 back and every poll invalidated it - measured 15x
 * slower than the three-launch form.  Frames are read from HBM once (the three-launch
 * form above reads them twice and keeps 13 bytes of per-frame scratch in between). */
struct WireLook {
  unsigned long long* state; /* [4][ntiles]  epoch << 40 | state << 38 | count */
  uint32_t* ticket;          /* next tile; the workgroup that draws the last one puts 0 back */
  uint32_t epoch;            /* 24 bits, never 0 (the words are zeroed when it wraps) */
#ifdef GPX_WD_TRACE /* timeline build (scripts/ubench/wire_trace.sh): never shipped */
  unsigned long long* trace; /* [ntiles][8] wall_clock64 stamps of thread 0 */
#endif
};
#ifdef GPX_WD_TRACE
#define WD_STAMP(k)                                                              \
  do {                                                                           \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                  \
    if (threadIdx.x == 0) K.trace[(int64_t)tile * 16 + (k)] = wall_clock64();     \
  } while (0)
#else
#define WD_STAMP(k) do { } while (0)
#endif
#define WL_AGG 1ull
#define WL_PRE 2ull
#define WL_VAL_MASK ((1ull << 38) - 1)
__device__ __forceinline__ unsigned long long wl_word(uint32_t epoch, unsigned long long st, uint32_t v) {
  return ((unsigned long long)epoch << 40) | (st << 38) | (unsigned long long)v;
}
/* exclusive prefix of class words before `tile`; called by one whole wave.  Polling is what this
 * costs: ~6,000 waves spinning on 64 words each flood the L2 request path (measured: four words per
 * lane made the kernel 35 % slower), so a wave first waits on ONE word - its nearest predecessor's,
 * published last of all it needs in the usual case - and only then reads 64 at a time. */
#ifndef GPX_WL_WIDE
#define GPX_WL_WIDE 1 /* words per lane and round trip (4 = 256 tiles per step: measured slower) */
#endif
__device__ __forceinline__ unsigned long long wl_lookback64(const unsigned long long* __restrict__ st, int32_t tile,
                                                 uint32_t epoch) {
  const int32_t lane = (int32_t)(threadIdx.x & 63);
  for (;;) { /* the nearest predecessor has parsed */
    const unsigned long long v = __hip_atomic_load(&st[tile - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(v >> 40) == epoch) break;
    __builtin_amdgcn_s_sleep(8);
  }
  unsigned long long excl = 0;
  for (int32_t hi0 = tile - 1; hi0 >= 0; hi0 -= 64 * GPX_WL_WIDE) {
    /* GPX_WL_WIDE x 64 words requested together (the older ones are published long since: one round trip for
     * 256 tiles; the walk's rate - tiles per round trip - is what bounds the whole kernel) */
    unsigned long long vv[GPX_WL_WIDE];
#pragma unroll
    for (int k = 0; k < GPX_WL_WIDE; k++) {
      const int32_t j = hi0 - 64 * k - lane;
      vv[k] = j >= 0 ? __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    }
#pragma unroll
    for (int k = 0; k < GPX_WL_WIDE; k++) {
      const int32_t hi = hi0 - 64 * k;
      if (hi < 0) break;
      const int32_t j = hi - lane; /* lane 0 = the nearest predecessor of this sub-step */
      unsigned long long v = vv[k];
      bool need = j >= 0 && (uint32_t)(v >> 40) != epoch;
      unsigned long long pre_mask;
      for (;;) {
        /* the walk stops at the nearest tile with a PREFIX: only lanes nearer than it must be valid */
        pre_mask = __ballot(!need && j >= 0 && ((v >> 38) & 3ull) == WL_PRE);
        const unsigned long long wait_mask = __ballot(need);
        if (pre_mask) {
          const unsigned long long nearer = (pre_mask & (0ull - pre_mask)) - 1ull;
          if ((wait_mask & nearer) == 0) break;
        } else if (wait_mask == 0) {
          break;
        }
        __builtin_amdgcn_s_sleep(8);
        if (need) {
          v = __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(v >> 40) == epoch) need = false;
        }
      }
      const int32_t first_pre = pre_mask ? (__ffsll((long long)pre_mask) - 1) : 64;
      unsigned long long x = (j >= 0 && lane <= first_pre) ? (v & WL_VAL_MASK) : 0ull;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, d, 64);
        const uint32_t hi32 = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), d, 64);
        x += ((unsigned long long)hi32 << 32) | lo;
      }
      excl += x;
      if (pre_mask) return excl;
    }
  }
  return excl;
}

__device__ __forceinline__ uint32_t wl_lookback(const unsigned long long* __restrict__ st, int32_t tile,
                                                 uint32_t epoch) {
  return (uint32_t)wl_lookback64(st, tile, epoch);
}

template <int WB>
__global__ __launch_bounds__(WB) __attribute__((amdgpu_waves_per_eu(6))) void k_wire_decode1(DevState S, DevNames N, WireLook K, WireOut O,
                                                           int32_t nf, int32_t ntiles,
                                                           const uint8_t* __restrict__ frames,
                                                           const int64_t* __restrict__ frame_off,
                                                           uint8_t* __restrict__ f_status,
                                                           int32_t* __restrict__ f_gidx,
                                                           int32_t* __restrict__ f_type,
                                                           gpx_wire_counts* counts) {
  constexpr int64_t STAGE_BYTES = (int64_t)GPX_W_STAGE_BYTES * (WB / GPX_BLOCK); /* 96 bytes per frame */
  __shared__ __attribute__((aligned(16))) uint32_t stage[STAGE_BYTES / 4 + 4 + 1];
  __shared__ int32_t s_tile;
  __shared__ uint32_t s_tot[4], s_base[4];
#ifdef GPX_WD_TRACE
  const unsigned long long t_entry = wall_clock64();
#endif
  if (threadIdx.x == 0) {
    const uint32_t t = atomicAdd(K.ticket, 1u);
    if (t == (uint32_t)ntiles - 1u) *K.ticket = 0u; /* every tile is taken: ready for the next call */
    s_tile = (int32_t)t;
  }
  __syncthreads();
  const int32_t tile = s_tile;
#ifdef GPX_WD_TRACE
  if (threadIdx.x == 0) K.trace[(int64_t)tile * 16] = t_entry;
#endif
  WD_STAMP(1); /* ticket drawn */
  const int32_t i = tile * WB + (int32_t)threadIdx.x;
  /* stage the tile (one window; a frame that does not lie inside it is read in place) */
  const int32_t t0 = tile * WB;
  const int32_t t1 = t0 + WB < nf ? t0 + WB : nf;
  const int64_t b0 = frame_off[t0], b1 = frame_off[t1];
  const uintptr_t a0 = (uintptr_t)(frames + b0) & ~(uintptr_t)3;
  const int64_t span = (int64_t)((uintptr_t)(frames + b1) - a0);
  const bool live = i < nf;
  const int64_t f0 = live ? frame_off[i] : b0, f1 = live ? frame_off[i + 1] : b0;
  const int64_t r0 = (int64_t)((uintptr_t)(frames + f0) - a0), r1 = (int64_t)((uintptr_t)(frames + f1) - a0);
  const int64_t nbytes = span < 0 ? 0 : (span < STAGE_BYTES ? span : STAGE_BYTES);
#ifdef GPX_WD_NOSTAGE /* ablation builds (scripts/ubench/wire_ablation.sh): never shipped */
  const int32_t lead = 0;
#else
  const int32_t lead = wire_stage<WB>(stage, a0, nbytes);
#endif
  __syncthreads();
  WD_STAMP(2); /* tile staged */
  const bool staged = live && r0 >= 0 && r1 >= r0 && r1 <= nbytes;
  WFrame f;
  f.st = GPX_W_OK;
  f.cnt = 0;
  f.cls = -1;
  f.gidx = -1;
  f.type = -1;
  f.tail = 0;
  f.ascending = true;
  f.stop = false;
#ifdef GPX_WD_NOPARSE
  if (live) {
    f.cls = 0;
    f.cnt = 1;
    f.gidx = i & (S.G - 1);
    f.type = GPX_WT_BATCHED_ACCEPT_REPLY;
  }
#else
  if (live) {
    if (staged)
      w_parse<LdsBytes>(S, N, (LdsBytes)stage + r0 + lead, f1 - f0, f);
    else
      w_parse<GenBytes>(S, N, frames + f0, f1 - f0, f);
  }
#endif
  WD_STAMP(3); /* thread 0 parsed and looked up */
#ifdef GPX_WD_TRACE
  if (threadIdx.x == 0) K.trace[(int64_t)tile * 16 + 7] = f.t_pre; /* ... and when its lookup began */
#endif
  const int32_t cls = (live && f.st == GPX_W_OK) ? f.cls : -1;
  const int32_t cnt = cls >= 0 ? f.cnt : 0;
  /* records of this tile per class; my offset inside the tile */
  int32_t off = 0;
#pragma unroll
  for (int c = 0; c < 4; c++) {
    int32_t tot;
    const int32_t ex = block_exscan_n<WB>(cls == c ? cnt : 0, &tot);
    if (cls == c) off = ex;
    if (threadIdx.x == 0) s_tot[c] = (uint32_t)tot;
  }
  __syncthreads();
  WD_STAMP(4); /* every lane parsed, tile counts known */
  if (threadIdx.x < 256) { /* wave c: class c's words */
    const int32_t c = (int32_t)(threadIdx.x >> 6);
    unsigned long long* st = K.state + (int64_t)c * ntiles;
    const uint32_t mine = s_tot[c];
    if ((threadIdx.x & 63) == 0)
      __hip_atomic_store(&st[tile], wl_word(K.epoch, tile == 0 ? WL_PRE : WL_AGG, mine), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
#ifdef GPX_WD_NOLOOKBACK
    const uint32_t excl = (uint32_t)tile * WB;
#else
    const uint32_t excl = tile == 0 ? 0u : wl_lookback(st, tile, K.epoch);
#endif
    if ((threadIdx.x & 63) == 0) {
      if (tile != 0)
        __hip_atomic_store(&st[tile], wl_word(K.epoch, WL_PRE, excl + mine), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      s_base[c] = excl;
      if (tile == ntiles - 1) (&counts->n_votes)[c] = (int32_t)(excl + mine);
    }
  }
  __syncthreads();
  WD_STAMP(5); /* look-back done */
  bool over = false;
  if (cls >= 0) {
    off += (int32_t)s_base[cls];
    const int32_t cap = cls == 0 ? O.V.cap : cls == 1 ? O.C.cap : cls == 2 ? O.A.cap : O.R.cap;
    over = (int64_t)off + cnt > cap;
  }
  if (live) {
    f_status[i] = over ? (uint8_t)GPX_W_CAPACITY : (uint8_t)f.st;
    if (f_gidx) f_gidx[i] = f.gidx;
    if (f_type) f_type[i] = f.type;
  }
  int32_t bad;
  block_exscan_n<WB>((live && (f.st != GPX_W_OK || over)) ? 1 : 0, &bad);
  if (threadIdx.x == 0 && bad) atomicAdd(&counts->n_bad_frames, bad);
#ifndef GPX_WD_NOEMIT
  if (cls >= 0 && !over) {
    const int32_t flags = (cls & 7) | (f.ascending ? 8 : 0) | (f.stop ? 16 : 0);
    if (staged)
      wire_emit<LdsBytes>(O, i, (LdsBytes)stage + r0 + lead, f.gidx, f.tail, flags, cls, cnt, off);
    else
      wire_emit<GenBytes>(O, i, frames + f0, f.gidx, f.tail, flags, cls, cnt, off);
  }
#endif
#ifdef GPX_WD_TRACE
  __syncthreads();
  WD_STAMP(6); /* records written */
#endif
}

/* ------------------------------------------------------------------------- */
/* encode: decisions -> BATCHED_COMMIT frames                                    */

struct PackIn {
  int32_t n;
  const int32_t* n_dev; /* nullable: the row count still lives in device memory */
  const int32_t *gidx, *slot, *bnum, *bcoord, *median;
  const uint8_t* kind;
};
struct PackScratch {
  int32_t* err;      /* [1] set when one group's block of rows exceeds GPX_W_MAX_SEG */
  int32_t* size;     /* [rows] padded frame size of a head row, else 0 */
  long long* tile_b; /* [ntiles] bytes per tile, then exclusive bases */
  int32_t* tile_f;   /* [ntiles] frames per tile, then exclusive bases */
  int32_t ntiles;
};

__device__ __forceinline__ int32_t pack_rows(const PackIn& P) {
  int32_t n = P.n;
  if (P.n_dev) {
    const int32_t m = *P.n_dev;
    n = m < n ? (m < 0 ? 0 : m) : n;
  }
  return n;
}
__device__ __forceinline__ bool pack_same(const PackIn& P, int32_t i, int32_t j) {
  return P.kind[j] == GPX_D_DECISION && P.bnum[j] == P.bnum[i] && P.bcoord[j] == P.bcoord[i];
}
/* Row i opens a frame iff it is the first DECISION of its (group, ballot) inside the group's block
 * of rows (PaxosPacketBatcher.enqueueImpl(BatchedCommit): one entry per paxosID and ballot,
 * PaxosPacketBatcher.java:139-156). */
__device__ __forceinline__ bool pack_is_head(const PackIn& P, const PackScratch& X, int32_t n,
                                             int32_t i) {
  const int32_t g = P.gidx[i];
  bool head = P.kind[i] == GPX_D_DECISION; /* allCoalescableDecisions (:451-457) */
  int32_t j = i - 1;
  for (; j >= 0 && i - j < GPX_W_MAX_SEG && P.gidx[j] == g; j--)
    if (pack_same(P, i, j)) head = false;
  /* contract: the engine emits at most `window` rows per group and call; a longer block is
   * refused (bounded work per lane whatever the caller passes) */
  if (j >= 0 && i - j >= GPX_W_MAX_SEG && P.gidx[j] == g) *X.err = 1;
  return head;
}

/* big-endian byte stream into 4-byte aligned memory through a one-word accumulator */
struct BEWriter {
  uint32_t* w;
  uint32_t acc;
  int32_t k;
  __device__ __forceinline__ void init(uint8_t* p) {
    w = (uint32_t*)p;
    acc = 0;
    k = 0;
  }
  __device__ __forceinline__ void put8(uint32_t b) {
    acc |= (b & 0xffu) << (8 * k);
    if (++k == 4) {
      *w++ = acc;
      acc = 0;
      k = 0;
    }
  }
  /* ByteBuffer.putInt at any byte offset: one funnel shift and one word out (round 4; until then four put8 -
   * everything behind the 13-byte header and the name is unaligned) */
  __device__ __forceinline__ void put32(int32_t v) { putraw32(__builtin_bswap32((uint32_t)v)); }
  /* four bytes that are already in memory order (a word of the name) */
  __device__ __forceinline__ void putraw32(uint32_t u) {
    const unsigned long long wide = (unsigned long long)acc | ((unsigned long long)u << (8 * k));
    *w++ = (uint32_t)wide;
    acc = (uint32_t)(wide >> 32); /* the k bytes that did not fit */
  }
  __device__ __forceinline__ void flush() {
    if (k) *w = acc;
  }
};

/* pass 1: frame sizes of the head rows */
/* padded frame size of row i if it opens a frame, else 0 */
__device__ __forceinline__ int32_t pack_frame_size(const DevState& S, const DevNames& N, const PackIn& P,
                                                   const PackScratch& X, int32_t n, int32_t i) {
  int32_t size = 0;
  if (i < n && pack_is_head(P, X, n, i)) {
    const int32_t g = P.gidx[i];
    if ((uint32_t)g < (uint32_t)S.G && N.tab && N.len(g) != 0 && N.row(g)[NM_EXISTS]) {
      /* TreeSet of the slots of this (group, ballot) */
      int32_t m = 0;
      for (int32_t j = i; j < n && j - i < GPX_W_MAX_SEG && P.gidx[j] == g; j++) {
        if (!pack_same(P, i, j)) continue;
        bool seen = false;
        for (int32_t q = i; q < j && !seen; q++) seen = pack_same(P, i, q) && P.slot[q] == P.slot[j];
        m += !seen;
      }
      const uint32_t gf = S.g_flags[g];
      const int32_t k = (int32_t)GF_K(gf);
      int32_t gs = 0;
      for (int32_t q = 0; q < k; q++) gs += S.members[(int64_t)q * S.G + g] != S.my_id;
      /* SIZEOF_PAXOSPACKET_FIXED + idLen + SIZEOF_BATCHEDCOMMIT_FIXED + 4 (n + 1 + g + 1)
       * (BatchedCommit.java:197-203), rounded up so that every frame starts 4-byte aligned */
      size = (13 + N.len(g) + 12 + 4 * (m + 1 + gs + 1) + 3) & ~3;
    }
  }
  return size;
}

/* What a row that is ALONE for its group (the usual decision batch: one decided slot per group) needs,
 * requested in two waves of independent loads - its neighbours' groups and its own kind first, then the
 * group's name row header, flags and members - instead of the five or six dependent round trips of the
 * general walk (round 3's decode timeline: a round trip costs 4-5 us under load, the bytes nothing). */
struct PackAlone {
  int32_t g, idl, gs, k;
  bool alone, frame; /* frame: the row opens a frame (a DECISION of an existing, named group) */
  uint4 r0, r1;      /* MEMBERS: the first 32 bytes of the group's name row */
};
template <bool MEMBERS>
__device__ __forceinline__ PackAlone pack_alone(const DevState& S, const DevNames& N, const PackIn& P, int32_t n,
                                                int32_t i, int32_t* mem /* [GPX_KMAX_LIMIT] or null */) {
  PackAlone A;
  A.g = -1;
  A.idl = A.gs = A.k = 0;
  A.alone = A.frame = false;
  A.r0 = A.r1 = make_uint4(0, 0, 0, 0);
  if (i >= n) return A;
  const int32_t g = P.gidx[i];
  const int32_t gp = i > 0 ? P.gidx[i - 1] : ~g, gn = i + 1 < n ? P.gidx[i + 1] : ~g;
  const int32_t kind = (int32_t)P.kind[i];
  A.g = g;
  A.alone = gp != g && gn != g;
  if (!A.alone || kind != GPX_D_DECISION || (uint32_t)g >= (uint32_t)S.G || !N.tab) return A;
  uint32_t hdr; /* length | exists << 8 */
  if (MEMBERS) { /* the writer wants the header's neighbours too: version and the first name bytes */
    const uint4* row = (const uint4*)N.row(g);
    A.r0 = row[0];
    A.r1 = row[1];
    hdr = A.r0.y;
  } else {
    hdr = *(const uint32_t*)(N.row(g) + 4);
  }
  const uint32_t gf = S.g_flags[g];
  int32_t mm[GPX_KMAX_LIMIT];
#pragma unroll
  for (int q = 0; q < GPX_KMAX_LIMIT; q++) mm[q] = q < S.kmax ? S.members[(int64_t)q * S.G + g] : 0;
  A.idl = (int32_t)(hdr & 0xffu);
  A.frame = A.idl != 0 && ((hdr >> 8) & 0xffu) != 0;
  A.k = (int32_t)GF_K(gf);
#pragma unroll
  for (int q = 0; q < GPX_KMAX_LIMIT; q++) {
    A.gs += (q < A.k && mm[q] != S.my_id) ? 1 : 0;
    if (MEMBERS) mem[q] = mm[q];
  }
  return A;
}

__global__ __launch_bounds__(GPX_BLOCK) void k_pack_scan(DevState S, DevNames N, PackIn P,
                                                        PackScratch X) {
  const int32_t n = pack_rows(P);
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  const PackAlone A = pack_alone<false>(S, N, P, n, i, nullptr);
  /* alone: one slot, (13 + idLen) + 12 + 4 (1 + 1 + g + 1) as in pack_frame_size */
  const int32_t size = A.alone ? (A.frame ? (13 + A.idl + 12 + 4 * (1 + 1 + A.gs + 1) + 3) & ~3 : 0)
                               : pack_frame_size(S, N, P, X, n, i);
  if (i < n) X.size[i] = size;
  int32_t tb, tf;
  block_exscan(size, &tb);
  block_exscan(size ? 1 : 0, &tf);
  if (threadIdx.x == 0) {
    X.tile_b[blockIdx.x] = tb;
    X.tile_f[blockIdx.x] = tf;
  }
}

/* BatchedCommit.toBytes (BatchedCommit.java:184-215) of head row i through a big-endian writer;
 * returns the frame's length in bytes */
template <class WR>
__device__ __forceinline__ int32_t pack_commit_frame(const DevState& S, const DevNames& N, const PackIn& P,
                                                     int32_t n, int32_t i, WR& w) {
  const int32_t g = P.gidx[i];
  /* the row's header (with its copy of the group's version) and first 20 name bytes: one 32-byte
   * access, two 16-byte loads (a byte loop over global memory is a chain of dependent loads) */
  const uint4* row = (const uint4*)N.row(g);
  const uint4 r0 = row[0], r1 = row[1];
  const int32_t idl = (int32_t)(r0.y & 0xffu);
  const uint32_t nw[5] = {r0.w, r1.x, r1.y, r1.z, r1.w};
  w.put32(GPX_WT_PAXOS_PACKET);       /* PaxosPacket.toBytes (PaxosPacket.java:461-476) */
  w.put32(GPX_WT_BATCHED_COMMIT);
  w.put32(r0.z);
  w.put8((uint32_t)idl);
  const uint8_t* nm = N.name(g);
#pragma unroll
  for (int32_t b = 0; b < NM_HOT; b++)
    if (b < idl) w.put8(nw[b >> 2] >> (8 * (b & 3)));
  for (int32_t b = NM_HOT; b < idl; b++) w.put8(nm[b]);
  w.put32(P.bnum[i]);
  w.put32(P.bcoord[i]);
  /* medianCheckpointedSlot: addCommit keeps the later one when `b - cur > 0` (:104-112) */
  int32_t med = P.median[i];
  int32_t m = 0;
  int32_t jend = i;
  for (int32_t j = i; j < n && j - i < GPX_W_MAX_SEG && P.gidx[j] == g; j++) {
    jend = j + 1;
    if (!pack_same(P, i, j)) continue;
    if (j > i && jsub(P.median[j], med) > 0) med = P.median[j];
    bool seen = false;
    for (int32_t q = i; q < j && !seen; q++) seen = pack_same(P, i, q) && P.slot[q] == P.slot[j];
    m += !seen;
  }
  w.put32(med);
  w.put32(m);
  long long last = -(1ll << 40);
  for (int32_t r = 0; r < m; r++) { /* TreeSet iteration: ascending */
    long long best = 1ll << 40;
    for (int32_t j = i; j < jend; j++)
      if (pack_same(P, i, j) && (long long)P.slot[j] > last && (long long)P.slot[j] < best)
        best = P.slot[j];
    w.put32((int32_t)best);
    last = best;
  }
  /* group = Util.arrayToIntSet(Util.filter(recipients, myID)) (PaxosPacketBatcher.java:391-395):
   * members are kept ascending, so this is TreeSet order */
  const int32_t k = (int32_t)GF_K(S.g_flags[g]);
  int32_t gs = 0;
  for (int32_t q = 0; q < k; q++) gs += S.members[(int64_t)q * S.G + g] != S.my_id;
  w.put32(gs);
  for (int32_t q = 0; q < k; q++) {
    const int32_t mem = S.members[(int64_t)q * S.G + g];
    if (mem != S.my_id) w.put32(mem);
  }
  w.flush();
  return 13 + idl + 12 + 4 * (m + 1 + gs + 1);
}

/* the frame of a row that is alone for its group, from what pack_alone and the caller already hold */
template <class WR>
__device__ __forceinline__ int32_t pack_commit_frame_alone(const DevState& S, const DevNames& N, const PackAlone& A,
                                                           const int32_t* mem, uint4 r0, uint4 r1, int32_t bnum,
                                                           int32_t bcoord, int32_t median, int32_t slot, WR& w) {
  const int32_t idl = A.idl;
  const uint32_t nw[4] = {r0.w, r1.x, r1.y, r1.z};
  w.put32(GPX_WT_PAXOS_PACKET); /* PaxosPacket.toBytes (PaxosPacket.java:461-476) */
  w.put32(GPX_WT_BATCHED_COMMIT);
  w.put32(r0.z);
  w.put8((uint32_t)idl);
  const uint8_t* nm = N.name(A.g);
#pragma unroll
  for (int32_t b = 0; b < NM_HOT; b += 4) { /* whole words of the name while they last, then its odd bytes */
    if (b + 4 <= idl) {
      w.putraw32(nw[b >> 2]);
    } else {
#pragma unroll
      for (int32_t q = 0; q < 4; q++)
        if (b + q < idl) w.put8(nw[b >> 2] >> (8 * q));
    }
  }
  for (int32_t b = NM_HOT; b < idl; b++) w.put8(nm[b]);
  w.put32(bnum);
  w.put32(bcoord);
  w.put32(median);
  w.put32(1);
  w.put32(slot);
  w.put32(A.gs);
#pragma unroll
  for (int q = 0; q < GPX_KMAX_LIMIT; q++)
    if (q < A.k && mem[q] != S.my_id) w.put32(mem[q]);
  w.flush();
  return 13 + idl + 12 + 4 * (1 + 1 + A.gs + 1);
}

/* the same accumulator over an LDS staging area */
struct BEWriterLds {
  uint32_t* w;
  uint32_t acc;
  int32_t k;
  __device__ __forceinline__ void put8(uint32_t b) {
    acc |= (b & 0xffu) << (8 * k);
    if (++k == 4) {
      *w++ = acc;
      acc = 0;
      k = 0;
    }
  }
  __device__ __forceinline__ void put32(int32_t v) { putraw32(__builtin_bswap32((uint32_t)v)); }
  __device__ __forceinline__ void putraw32(uint32_t u) { /* as BEWriter's */
    const unsigned long long wide = (unsigned long long)acc | ((unsigned long long)u << (8 * k));
    *w++ = (uint32_t)wide;
    acc = (uint32_t)(wide >> 32);
  }
  __device__ __forceinline__ void flush() {
    if (k) *w = acc;
  }
};

/* pass 3: the frames of a tile are contiguous in the output (offsets are prefix sums): every lane
 * builds its frame in LDS, then the workgroup flushes the tile with coalesced dword stores - a lane
 * writing its own ~56-byte frame to global memory word by word issues 14 stores that each touch a
 * different sector than its neighbours' (measured: 238 us per 1 M frames, 0.24 TB/s).  A tile
 * bigger than the staging area (long names, many slots) is written in place as before. */
#define GPX_PACK_STAGE_BYTES (24 * 1024)
__global__ __launch_bounds__(GPX_BLOCK) void k_pack_write(DevState S, DevNames N, PackIn P,
                                                         PackScratch X, uint8_t* __restrict__ out,
                                                         long long cap_bytes,
                                                         long long* __restrict__ frame_off,
                                                         int32_t* __restrict__ frame_len,
                                                         int32_t* __restrict__ f_gidx, int32_t* __restrict__ n_frames,
                                                         long long* __restrict__ n_bytes) {
  __shared__ uint32_t stage[GPX_PACK_STAGE_BYTES / 4];
  __shared__ long long s_b[GPX_BLOCK / 64];
  __shared__ int32_t s_f[GPX_BLOCK / 64];
  const int32_t n = pack_rows(P);
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  /* requested first, all independent of one another: the row's own columns, its frame size, the totals of
   * the tiles before this one ... */
  const bool in = i < n;
  const int32_t a_bnum = in ? P.bnum[i] : 0, a_bcoord = in ? P.bcoord[i] : 0, a_median = in ? P.median[i] : 0;
  const int32_t a_slot = in ? P.slot[i] : 0;
  const int32_t size = in ? X.size[i] : 0;
  /* (this tile's base = the bytes and frames of the tiles before it; a separate one-workgroup scan kernel
   * used to turn k_pack_scan's totals into bases: 10 us of the call) */
  long long bb = 0;
  int32_t bf = 0;
  for (int32_t t0 = threadIdx.x; t0 < (int32_t)blockIdx.x; t0 += 8 * GPX_BLOCK) { /* eight pairs in flight */
    long long vb[8];
    int32_t vf[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int32_t t = t0 + k * GPX_BLOCK;
      const bool ok = t < (int32_t)blockIdx.x;
      vb[k] = ok ? X.tile_b[t] : 0;
      vf[k] = ok ? X.tile_f[t] : 0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
      bb += vb[k];
      bf += vf[k];
    }
  }
  /* ... then what an alone row's frame is made of (pack_alone: the second wave of loads) */
  int32_t mem[GPX_KMAX_LIMIT];
  const PackAlone A = pack_alone<true>(S, N, P, n, i, mem);
  const bool fast = A.alone && A.frame;
  const uint4 r0 = A.r0, r1 = A.r1;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)bb, d, 64);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)((unsigned long long)bb >> 32), d, 64);
    bb += (long long)(((unsigned long long)hi << 32) | lo);
    bf += __shfl_xor(bf, d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    s_b[threadIdx.x >> 6] = bb;
    s_f[threadIdx.x >> 6] = bf;
  }
  int32_t tb, tf;
  const int32_t eb = block_exscan(size, &tb); /* (its barriers publish s_b / s_f) */
  const int32_t ef = block_exscan(size ? 1 : 0, &tf);
  long long tile0 = 0;
  int32_t frame0 = 0;
#pragma unroll
  for (int w = 0; w < GPX_BLOCK / 64; w++) {
    tile0 += s_b[w];
    frame0 += s_f[w];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { /* totals to the caller */
    *n_frames = *X.err ? -1 : frame0 + tf;
    *n_bytes = tile0 + tb;
  }
  const bool staged = tb <= GPX_PACK_STAGE_BYTES && tile0 + tb <= cap_bytes; /* workgroup-uniform */
  const long long off = tile0 + eb;
  const int32_t fi = frame0 + ef;
  if (size && (staged || off + size <= cap_bytes)) { /* else: the host sees n_bytes > cap_bytes */
    int32_t len;
    if (staged) {
      BEWriterLds w;
      w.w = stage + (eb >> 2);
      w.acc = 0;
      w.k = 0;
      len = fast ? pack_commit_frame_alone(S, N, A, mem, r0, r1, a_bnum, a_bcoord, a_median, a_slot, w)
                 : pack_commit_frame(S, N, P, n, i, w);
    } else {
      BEWriter w;
      w.init(out + off);
      len = fast ? pack_commit_frame_alone(S, N, A, mem, r0, r1, a_bnum, a_bcoord, a_median, a_slot, w)
                 : pack_commit_frame(S, N, P, n, i, w);
    }
    frame_off[fi] = off;
    frame_len[fi] = len;
    f_gidx[fi] = P.gidx[i];
  }
  if (staged) {
    __syncthreads();
    uint32_t* dst = (uint32_t*)(out + tile0); /* tile0 is a multiple of 4 */
    for (int32_t wi = threadIdx.x; wi < (tb >> 2); wi += GPX_BLOCK) dst[wi] = stage[wi];
  }
}


/* ------------------------------------------------------------------------- */
/* encode: accept replies -> BATCHED_ACCEPT_REPLY frames                         */
/* replaces PaxosPacketBatcher.enqueueImpl(AcceptReplyPacket) / dequeueImplAR
 * (PaxosPacketBatcher.java:121-137, 211-224) + BatchedAcceptReply.toBytes
 * (BatchedAcceptReply.java:119-173) for the replies of one gpx_accept_batch call.
 *
 * The replies are regrouped by group with the engine's own front end (k_hist + k_scatter_ac carry
 * {slot, maxCP, reply ballot} per record); one workgroup per bucket then stages the bucket in LDS
 * (bucket_prepare) and ONE LANE PER GROUP walks its replies in arrival order exactly like the
 * batcher's HashMap<paxosID, HashMap<Ballot, BatchedAcceptReply>> would see them.  Frames are
 * first written to a per-bucket staging area (a bucket's bytes are bounded by 188 per reply), then
 * k_emit_frames compacts them, buckets in order, into the caller's buffer. */
#define GPX_W_BAR_MAX_RECS 256  /* replies of one group coalesced per call; the rest leave unbatched */
#define GPX_W_BAR_MAX_BALLOTS 4 /* distinct reply ballots of one group coalesced per call */
#define GPX_W_BAR_REC_BYTES 188 /* >= 13 + 127 + 29 + 4 + 12, 4-byte aligned: staging per reply */

struct PackArIn {
  const int32_t* sender;  /* nullable: ACCEPT's sender; a reply whose ballot coordinator differs is
                             not coalescable (allPositiveAcceptReplies, PaxosPacketBatcher.java:438-446) */
  const long long* req_id; /* nullable */
  const uint8_t* status;   /* the accept call's status column: GPX_S_OK = a reply exists */
  uint8_t* unbatched;      /* nullable out: 1 = reply exists but was not packed */
  uint8_t* stage;          /* [max_batch * GPX_W_BAR_REC_BYTES] */
  long long* bucket_bytes; /* [nbk] */
};

__global__ __launch_bounds__(1024) void k_bucket_pack_ar(DevState S, DevScratch X, DevNames N,
                                                         PackArIn P) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  BucketView bv;
  const int32_t b = blockIdx.x;
  if (!bucket_prepare(X, lds, &bv, []() {})) {
    if (threadIdx.x == 0) P.bucket_bytes[b] = 0;
    return;
  }
  const int32_t boff = X.bucket_off[b];
  const int32_t g0 = b << X.shift;
  const int32_t nt = (int32_t)blockDim.x;
  /* pass 1: per group, the distinct reply ballots (first-appearance order), their slot counts,
   * the group's bytes and frames.  Thread t owns groups t, t + nt, ... */
  int32_t my_bytes = 0, my_frames = 0;
  const int32_t per = X.gb / nt;
  for (int32_t qq = 0; qq < per; qq++) {
    const int32_t l = (int32_t)threadIdx.x * per + qq;
    const int32_t c = bv.lcnt[l];
    const int32_t g = g0 + l;
    if (c == 0 || g >= S.G) continue;
    unsigned long long* keys = bv.keys + bv.loff[l];
    if (c <= GPX_SMALL_SEG) { /* arrival order in place (longer segments were sorted cooperatively) */
      for (int32_t i = 1; i < c; i++) {
        const unsigned long long x = keys[i];
        int32_t p = i - 1;
        while (p >= 0 && keys[p] > x) {
          keys[p + 1] = keys[p];
          p--;
        }
        keys[p + 1] = x;
      }
    }
    const bool named = N.tab && N.len(g) != 0 && N.row(g)[NM_EXISTS];
    const int32_t cc = c < GPX_W_BAR_MAX_RECS ? c : GPX_W_BAR_MAX_RECS;
    int32_t nbal = 0, bn[GPX_W_BAR_MAX_BALLOTS], bc[GPX_W_BAR_MAX_BALLOTS];
    for (int32_t t = 0; t < c; t++) {
      const unsigned long long k = keys[t];
      const int32_t ix = (int32_t)(k >> 32);
      const int32_t* pay = bv.pay + (int64_t)(uint32_t)k * bv.rs;
      const bool has_reply = P.status[ix] == GPX_S_OK;
      const int32_t rbn = pay[3 * bv.fs], rbc = pay[4 * bv.fs];
      const bool coalescable = has_reply && named && (!P.sender || P.sender[ix] == rbc);
      bool ok = coalescable && t < cc;
      int32_t q = -1;
      if (ok) {
        for (int32_t z = 0; z < nbal; z++)
          if (bn[z] == rbn && bc[z] == rbc) q = z;
        if (q < 0 && nbal < GPX_W_BAR_MAX_BALLOTS) {
          q = nbal++;
          bn[q] = rbn;
          bc[q] = rbc;
        }
        ok = q >= 0;
      }
      /* remember the verdict in the record's third payload word (unused so far): ballot index
       * or -1 */
      const_cast<int32_t*>(pay)[2 * bv.fs] = ok ? q : -1;
      /* 2 = coalescable but over a per-pass limit: the next pass packs it (gpx_wire.h) */
      if (P.unbatched) P.unbatched[ix] = (coalescable && !ok) ? 2 : (has_reply && !ok) ? 1 : 0;
    }
    for (int32_t q = 0; q < nbal; q++) {
      int32_t m = 0; /* TreeMap size: distinct slots of this ballot */
      for (int32_t t = 0; t < cc; t++) {
        const int32_t* pt = bv.pay + (int64_t)(uint32_t)keys[t] * bv.rs;
        if (pt[2 * bv.fs] != q) continue;
        bool seen = false;
        for (int32_t u = 0; u < t && !seen; u++) {
          const int32_t* pu = bv.pay + (int64_t)(uint32_t)keys[u] * bv.rs;
          seen = pu[2 * bv.fs] == q && pu[0] == pt[0];
        }
        m += !seen;
      }
      my_bytes += (13 + N.len(g) + 29 + 4 + 12 * m + 3) & ~3;
      my_frames++;
    }
  }
  __syncthreads();
  int32_t tot_b, tot_f;
  int32_t ex_b = block_exscan_rt(my_bytes, &tot_b);
  int32_t ex_f = block_exscan_rt(my_frames, &tot_f);
  if (threadIdx.x == 0) {
    P.bucket_bytes[b] = tot_b;
    X.bucket_nout[b] = tot_f;
  }
  /* pass 2: the frames, into the bucket's staging area; one table row per frame in o_rec */
  uint8_t* stage = P.stage + (int64_t)boff * GPX_W_BAR_REC_BYTES;
  Out* ftab = X.o_rec + boff;
  for (int32_t qq = 0; qq < per; qq++) {
    const int32_t l = (int32_t)threadIdx.x * per + qq;
    const int32_t c = bv.lcnt[l];
    const int32_t g = g0 + l;
    if (c == 0 || g >= S.G) continue;
    const unsigned long long* keys = bv.keys + bv.loff[l];
    const int32_t cc = c < GPX_W_BAR_MAX_RECS ? c : GPX_W_BAR_MAX_RECS;
    const int32_t idl = N.tab ? N.len(g) : 0;
    for (int32_t q = 0; q < GPX_W_BAR_MAX_BALLOTS; q++) {
      int32_t head = -1;
      for (int32_t t = 0; t < cc && head < 0; t++)
        if ((bv.pay + (int64_t)(uint32_t)keys[t] * bv.rs)[2 * bv.fs] == q) head = t;
      if (head < 0) break; /* ballot indices are dense */
      const unsigned long long kh = keys[head];
      const int32_t* ph = bv.pay + (int64_t)(uint32_t)kh * bv.rs;
      const int32_t ixh = (int32_t)(kh >> 32);
      BEWriter w;
      w.init(stage + ex_b);
      w.put32(GPX_WT_PAXOS_PACKET);
      w.put32(GPX_WT_BATCHED_ACCEPT_REPLY);
      w.put32(S.g_version[g]);
      w.put8((uint32_t)idl);
      const uint8_t* nm = N.name(g);
      for (int32_t z = 0; z < idl; z++) w.put8(nm[z]);
      /* new BatchedAcceptReply(first reply): acceptor, ballot, its slot, its maxCheckpointedSlot,
       * its requestID, undigestRequest = false (BatchedAcceptReply.java:49-54,
       * AcceptReplyPacket.toBytes :170-180) */
      w.put32(S.my_id);
      w.put32(ph[3 * bv.fs]);
      w.put32(ph[4 * bv.fs]);
      w.put32(ph[0]);
      w.put32(ph[bv.fs]);
      const long long rq0 = P.req_id ? P.req_id[ixh] : 0;
      w.put32((int32_t)(rq0 >> 32));
      w.put32((int32_t)rq0);
      w.put8(0);
      int32_t m = 0;
      for (int32_t t = 0; t < cc; t++) {
        const int32_t* pt = bv.pay + (int64_t)(uint32_t)keys[t] * bv.rs;
        if (pt[2 * bv.fs] != q) continue;
        bool seen = false;
        for (int32_t u = 0; u < t && !seen; u++) {
          const int32_t* pu = bv.pay + (int64_t)(uint32_t)keys[u] * bv.rs;
          seen = pu[2 * bv.fs] == q && pu[0] == pt[0];
        }
        m += !seen;
      }
      w.put32(m);
      long long last = -(1ll << 40);
      for (int32_t r = 0; r < m; r++) { /* TreeMap iteration; put() keeps the LAST request id */
        long long best = 1ll << 40;
        int32_t who = -1;
        for (int32_t t = 0; t < cc; t++) {
          const int32_t* pt = bv.pay + (int64_t)(uint32_t)keys[t] * bv.rs;
          if (pt[2 * bv.fs] != q) continue;
          const long long s = pt[0];
          if (s > last && s <= best) {
            best = s;
            who = t;
          }
        }
        w.put32((int32_t)best);
        const long long rq = P.req_id ? P.req_id[(int32_t)(keys[who] >> 32)] : 0;
        w.put32((int32_t)(rq >> 32));
        w.put32((int32_t)rq);
        last = best;
      }
      w.flush();
      const int32_t len = 13 + idl + 29 + 4 + 12 * m;
      ftab[ex_f] = mk_out(g, len, ex_b, ph[4 * bv.fs], 0, 0); /* gidx, len, rel. offset, dest */
      ex_b += (len + 3) & ~3;
      ex_f++;
    }
  }
}

/* per bucket: copy its staged frames behind those of the buckets before it; frame table */
__global__ __launch_bounds__(GPX_BLOCK) void k_emit_frames(DevScratch X, PackArIn P,
                                                          uint8_t* __restrict__ out,
                                                          long long cap_bytes,
                                                          long long* __restrict__ frame_off,
                                                          int32_t* __restrict__ frame_len,
                                                          int32_t* __restrict__ f_gidx,
                                                          int32_t* __restrict__ f_dest,
                                                          int32_t* n_frames, long long* n_bytes) {
  __shared__ long long red[GPX_BLOCK / 64];
  const int32_t b = blockIdx.x;
  const int32_t fbase = emit_base(X, n_frames, nullptr);
  long long before = 0;
  for (int32_t t = threadIdx.x; t < b; t += GPX_BLOCK) before += P.bucket_bytes[t];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = before;
  __syncthreads();
  long long bbase = 0;
  for (int w = 0; w < GPX_BLOCK / 64; w++) bbase += red[w];
  const long long mine = P.bucket_bytes[b];
  if (b == (int32_t)gridDim.x - 1 && threadIdx.x == 0) *n_bytes = bbase + mine;
  if (bbase + mine > cap_bytes) return; /* the caller sees n_bytes > cap_bytes */
  const int32_t boff = X.bucket_off[b];
  const uint32_t* src = (const uint32_t*)(P.stage + (int64_t)boff * GPX_W_BAR_REC_BYTES);
  uint32_t* dst = (uint32_t*)(out + bbase);
  for (int32_t i = threadIdx.x; i < (int32_t)(mine >> 2); i += GPX_BLOCK) dst[i] = src[i];
  const int32_t nfr = X.bucket_nout[b];
  const Out* ftab = X.o_rec + boff;
  for (int32_t f = threadIdx.x; f < nfr; f += GPX_BLOCK) {
    const Out r = ftab[f];
    frame_off[fbase + f] = bbase + r.x;
    frame_len[fbase + f] = r.slot;
    f_gidx[fbase + f] = r.gidx;
    f_dest[fbase + f] = r.y;
  }
}
