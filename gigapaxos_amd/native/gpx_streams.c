/*
 * gpx_streams.c — SURVEY.md 8(d)'s synthetic accept-reply stream, as specified: "PRNG: xorshift64* seeded
 * 0x9E3779B97F4A7C15 ^ (config_id << 32) ^ r", "acceptor = members[pi(j)] with pi a per-(g, r) permutation", "global
 * order = Fisher-Yates shuffle of the round's records", "adversarial mix: 1 % duplicated votes, 0.5 % stale (bcoord - 1),
 * 0.1 % higher-ballot (bnum = 1) votes".  What the survey leaves open is pinned here and in gigapaxos_amd/streams.py so
 * that anybody can regenerate the stream bit for bit:
 *   next()        x ^= x >> 12; x ^= x << 25; x ^= x >> 27; return x * 0x2545F4914F6CDD1D   (Vigna's xorshift64*)
 *   below(m)      (next() * m) >> 64 with a 128-bit product            (uniform enough for m < 2^32, no rejection)
 *   pi            groups in index order, for each one a Fisher-Yates over 0 .. K-1: for j = K-1 .. 1: swap(j, below(j + 1))
 *   mix           appended behind the G*K votes, in this order: n/100 duplicates, n/200 stale, n/1000 higher-ballot
 *                 votes, each a copy of vote below(n) with the one field changed (at least one of each kind)
 *   global order  one Fisher-Yates over all records: for i = total-1 .. 1: swap(i, below(i + 1))
 * Plain C (gcc), host only: data generation for bench.py and the tests - no device code, nothing of the engine.
 */
#include <stdint.h>
#include <stdlib.h>

static inline uint64_t xs_next(uint64_t* s) {
  uint64_t x = *s;
  x ^= x >> 12;
  x ^= x << 25;
  x ^= x >> 27;
  *s = x;
  return x * 0x2545F4914F6CDD1DULL;
}
static inline uint64_t xs_below(uint64_t* s, uint64_t m) { return (uint64_t)(((unsigned __int128)xs_next(s) * m) >> 64); }

/* columns of capacity gpx_stream_capacity(G, K, mix); returns the number of votes written */
int64_t gpx_stream_capacity(int64_t G, int32_t K, int32_t mix) {
  const int64_t n = G * (int64_t)K;
  if (!mix || n <= 0) return n > 0 ? n : 0;
  const int64_t nd = n / 100 > 0 ? n / 100 : 1, ns = n / 200 > 0 ? n / 200 : 1, nh = n / 1000 > 0 ? n / 1000 : 1;
  return n + nd + ns + nh;
}
int64_t gpx_stream_vote_round(int64_t G, const int32_t* groups /* nullable: 0 .. G-1 */, int32_t K, const int32_t* members,
                              int32_t round, int32_t coordinator, int32_t config_id, int32_t shuffled, int32_t mix,
                              int32_t* gidx, int32_t* bnum, int32_t* bcoord, int32_t* slot, int32_t* acceptor,
                              int32_t* max_cp) {
  if (G < 0 || K < 1 || K > 16 || !members) return -1;
  uint64_t s = 0x9E3779B97F4A7C15ULL ^ ((uint64_t)(uint32_t)config_id << 32) ^ (uint64_t)(uint32_t)round;
  if (!s) s = 0x9E3779B97F4A7C15ULL; /* (the all-zero state is the generator's fixed point) */
  const int64_t n = G * (int64_t)K;
  int64_t w = 0;
  for (int64_t g = 0; g < G; g++) {
    int32_t pi[16];
    for (int j = 0; j < K; j++) pi[j] = j;
    for (int j = K - 1; j >= 1; j--) {
      const int t = (int)xs_below(&s, (uint64_t)j + 1);
      const int32_t x = pi[j];
      pi[j] = pi[t];
      pi[t] = x;
    }
    for (int j = 0; j < K; j++, w++) {
      gidx[w] = groups ? groups[g] : (int32_t)g;
      bnum[w] = 0;
      bcoord[w] = coordinator;
      slot[w] = round + 1;
      acceptor[w] = members[pi[j]];
      max_cp[w] = round;
    }
  }
  if (mix && n > 0) {
    const int64_t nd = n / 100 > 0 ? n / 100 : 1, ns = n / 200 > 0 ? n / 200 : 1, nh = n / 1000 > 0 ? n / 1000 : 1;
    for (int64_t q = 0; q < nd + ns + nh; q++, w++) {
      const int64_t p = (int64_t)xs_below(&s, (uint64_t)n);
      gidx[w] = gidx[p], bnum[w] = bnum[p], bcoord[w] = bcoord[p], slot[w] = slot[p], acceptor[w] = acceptor[p],
      max_cp[w] = max_cp[p];
      if (q >= nd && q < nd + ns) bcoord[w] -= 1; /* stale ballot */
      if (q >= nd + ns) bnum[w] = 1;              /* higher ballot */
    }
  }
  if (shuffled) {
    for (int64_t i = w - 1; i >= 1; i--) {
      const int64_t t = (int64_t)xs_below(&s, (uint64_t)i + 1);
      int32_t x;
#define GPX_SWAP(col) x = col[i], col[i] = col[t], col[t] = x
      GPX_SWAP(gidx);
      GPX_SWAP(bnum);
      GPX_SWAP(bcoord);
      GPX_SWAP(slot);
      GPX_SWAP(acceptor);
      GPX_SWAP(max_cp);
#undef GPX_SWAP
    }
  }
  return w;
}
