// View change, coordinator side: running for coordinator and the prepare-reply phase
// (phase 1b) of PaxosCoordinatorState, batched over groups.
//
//   gpx_election_begin       PISM.tryMakeCoordinator -> PaxosCoordinator.makeCoordinator
//                            (PaxosInstanceStateMachine.java:2178-2183, PaxosCoordinator.java:66-89)
//   gpx_prepare_reply_batch  PISM.handlePrepareReply (PISM:1008-1068) ->
//                            PaxosCoordinatorState.isPreemptable / canIgnorePrepareReply /
//                            isPrepareAcceptedByMajority / combinePValuesOntoProposals /
//                            reproposePreemptedProposals / processStop / spawnCommandersForProposals
//                            / setCoordinatorActive (PaxosCoordinatorState.java:271-587)
//
// A node failure starts an election in every group the dead node coordinated - at a million groups
// per node that is a burst of millions of PREPARE replies, each a small, branchy, strictly
// per-group computation: one lane per group again, over the same bucket front end as the other
// calls (k_hist -> k_scatter_ac -> one workgroup per bucket, records replayed in arrival order).
// This is the cold path: the per-lane working set (the carried-over pvalues of one group, the
// proposal list being built) sits in scratch-backed arrays of `window` entries, which the hot
// kernels never pay for.
//
// State while a group is GF_PREPARING (DevState): c_wait = waitforMyBallot's heard-from mask,
// node_slots = the minimum slots the replies named (-1 = not heard), co_ring/co_handle = the
// carried-over pvalue of slot s at ring index s & (W-1), p_ring/p_handle = the pre-active proposals
// (slots c_next - c_pcount .. c_next - 1, contiguous: nothing removes a proposal before the
// coordinator is active).
#pragma once

struct PReplyIn {
  const int32_t* pv_off;
  int32_t pv_total;
  const int32_t *pv_slot, *pv_bnum, *pv_bcoord;
  const int64_t* pv_handle;
  const uint8_t* pv_flags;
};
struct PReplyOut {
  int32_t n;
  uint8_t* v_kind;
  int32_t *e_count, *e_median, *e_slot; /* e_*: [W][n] */
  uint8_t* e_kind;
  int64_t* e_handle;
  uint8_t* e_flags;
  uint8_t* status;
};

/* makeCoordinator(c, bnum, myID, members, paxosState.getSlot(), recovery = false) */
__global__ __launch_bounds__(GPX_BLOCK) void k_election_begin(DevState S, int32_t n,
                                                             const int32_t* __restrict__ gidx,
                                                             const int32_t* __restrict__ bnum,
                                                             uint8_t* __restrict__ e_status) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx[i];
  if ((uint32_t)g >= (uint32_t)S.G || !(S.g_flags[g] & GF_EXISTS)) {
    e_status[i] = 255;
    return;
  }
  const uint32_t gf = S.g_flags[g];
  const int32_t k = (int32_t)GF_K(gf);
  const bool has = (gf & GF_HASCOORD) != 0;
  const int32_t cmp = has ? ballot_cmp(S.c_bnum[g], S.c_bcoord[g], bnum[i], S.my_id) : -1;
  if (!has || cmp < 0) {
    /* new PaxosCoordinatorState(bnum, myID, slot, members, null) (PCS:166-181): the previous
     * coordinator's proposals are dropped (prev == null) */
    S.c_bnum[g] = bnum[i];
    S.c_bcoord[g] = S.my_id;
    S.c_next[g] = S.a_slot[g];
    S.c_pcount[g] = 0;
    S.c_wait[g] = 0;
    for (int32_t j = 0; j < k; j++) S.node_slots[(int64_t)j * S.G + g] = -1;
    for (int32_t w = 0; w < S.W; w++) {
      const int64_t o = (int64_t)w * S.G + g;
      S.p_ring[o] = 0;
      S.co_ring[o] = I4{0, 0, 0, 0};
    }
    if (bnum[i] == 0) { /* initial coordinator status assumed, not explicitly prepared (:74-76) */
      S.g_flags[g] = (gf | GF_HASCOORD) & ~GF_PREPARING;
      e_status[i] = GPX_EB_ACTIVE;
    } else { /* prepare(members) arms waitforMyBallot (PCS:214-220) */
      S.g_flags[g] = gf | GF_HASCOORD | GF_PREPARING;
      e_status[i] = GPX_EB_PREPARING;
    }
  } else if (cmp == 0 && (gf & GF_PREPARING)) {
    e_status[i] = GPX_EB_RESEND; /* same ballot, not active: resend prepare (:80-83) */
  } else {
    e_status[i] = GPX_EB_UNCHANGED;
  }
}

/* pokeLocalCoordinator / PREPARE resend, minus the clocks (include/gpx.h gpx_poke_scan) */
template <int KMAX>
__global__ __launch_bounds__(GPX_BLOCK) void k_poke_scan(DevState S, int32_t n,
                                                        const int32_t* __restrict__ gidx,
                                                        uint8_t* __restrict__ poke,
                                                        int32_t* __restrict__ slot,
                                                        int32_t* __restrict__ bnum,
                                                        int32_t* __restrict__ bcoord,
                                                        int32_t* __restrict__ median_cp,
                                                        uint8_t* __restrict__ p_flags,
                                                        uint32_t* __restrict__ heard,
                                                        uint8_t* __restrict__ status) {
  const int32_t i = blockIdx.x * GPX_BLOCK + threadIdx.x;
  if (i >= n) return;
  const int32_t g = gidx ? gidx[i] : i;
  uint8_t pk = GPX_POKE_NONE, fl = 0, st = GPX_S_OK;
  int32_t sl = 0, bn = 0, bc = 0, med = 0;
  uint32_t hd = 0;
  if ((uint32_t)g >= (uint32_t)S.G || !(S.g_flags[g] & GF_EXISTS)) {
    st = GPX_S_NOGROUP;
  } else {
    const uint32_t gf = S.g_flags[g];
    if (gf & GF_HASCOORD) {
      const int32_t s = S.a_slot[g];
      if (gf & GF_PREPARING) {
        pk = GPX_POKE_PREPARE;
        hd = S.c_wait[g];
      } else {
        const int32_t d = jsub(S.c_next[g], s); /* in myProposals' window iff 1 <= d <= W */
        const uint32_t e = (d >= 1 && d <= S.W) ? S.p_ring[(int64_t)(s & (S.W - 1)) * S.G + g] : 0u;
        if (e & PR_PRESENT) {
          const int32_t k = (int32_t)GF_K(gf);
          int32_t ns[KMAX];
#pragma unroll
          for (int q = 0; q < KMAX; q++) ns[q] = (q < k) ? S.node_slots[(int64_t)q * S.G + g] : 0;
          pk = GPX_POKE_ACCEPT;
          med = median_minus<KMAX>(ns, k);
          fl = (e & PR_STOP) ? GPX_PV_STOP : 0;
          hd = e & 0xffffu;
        }
      }
      if (pk != GPX_POKE_NONE) {
        sl = s;
        bn = S.c_bnum[g];
        bc = S.c_bcoord[g];
      }
    }
  }
  poke[i] = pk;
  slot[i] = sl;
  bnum[i] = bn;
  bcoord[i] = bc;
  median_cp[i] = med;
  p_flags[i] = fl;
  heard[i] = hd;
  status[i] = st;
}

/* fe[] entry of the proposal list being built */
#define FE_PRESENT 0x1000u
#define FE_STOP 0x1u
#define FE_KIND(e) (((e) >> 4) & 0xfu)
#define FE_MAKE(kind, stop) (FE_PRESENT | ((uint32_t)(kind) << 4) | ((stop) ? FE_STOP : 0u))

/* positions 0 .. cnt-1 hold slots lo .. lo+cnt-1 (wrapping); visit them in ascending SIGNED slot
 * order, the iteration order of the reference's TreeMap<Integer, ...> */
template <class F>
__device__ __forceinline__ void for_signed_order(int32_t lo, int32_t cnt, F f) {
  if (cnt <= 0) return;
  const int32_t hi = (int32_t)((uint32_t)lo + (uint32_t)(cnt - 1));
  int32_t split = 0; /* first position on the negative side of the wrap */
  if (hi < lo) split = (int32_t)(0x80000000u - (uint32_t)lo);
  for (int32_t j = split; j < cnt; j++) f(j);
  for (int32_t j = 0; j < split; j++) f(j);
}

/* WMAX >= window: size of the per-lane arrays (scratch-backed; the host instantiates 64 only).
 * __forceinline__ like every other per-group function here: as a __noinline__ call (an earlier
 * build) the kernel faulted intermittently on the GPU box (SIGSEGV inside elect_group under rocgdb,
 * a silent abort outside it) with the same source that runs clean inlined. */
/* GPX_ELECT_BOUNDS (debug build, scripts/repro_noinline_fault.sh): every index into the per-lane arrays of
 * elect_group is checked - a violation prints its source line and traps, which the runtime reports as an
 * exception of the kernel, not as a memory fault: tells an out-of-bounds scratch index (ours) from a fault
 * in the call / scratch machinery (the toolchain's) */
#ifdef GPX_ELECT_BOUNDS
#define EI(i) elect_idx((int32_t)(i), WMAX, __LINE__)
__device__ __forceinline__ int32_t elect_idx(int32_t i, int32_t lim, int line) {
  if ((uint32_t)i >= (uint32_t)lim) {
    printf("elect_group: index %d outside [0, %d) at gpx_elect.hip.h:%d\n", i, lim, line);
    __builtin_trap();
  }
  return i;
}
#else
#define EI(i) (i)
#endif
template <int KMAX, int WMAX>
#ifdef GPX_ELECT_NOINLINE /* repro build of round 1's intermittent fault: scripts/repro_noinline_fault.sh */
#define GPX_ELECT_INLINE __attribute__((noinline))
#else
#define GPX_ELECT_INLINE __forceinline__
#endif
__device__ GPX_ELECT_INLINE void elect_group(const DevState& S, const DevScratch& X, int32_t g,
                                         GroupIter& it, const PReplyIn& I, const PReplyOut& O) {
  const int32_t G = S.G, W = S.W, Wm = W - 1, n = O.n;
  uint32_t gf = S.g_flags[g];
  const uint32_t gf0 = gf;
  const bool exists = (gf & GF_EXISTS) != 0;
  const int32_t k = (int32_t)GF_K(gf);
  const bool had = exists && (gf & GF_HASCOORD);
  const int32_t my_bnum = had ? S.c_bnum[g] : 0, my_bcoord = had ? S.c_bcoord[g] : 0;
  int32_t next = had ? S.c_next[g] : 0, pcount = had ? S.c_pcount[g] : 0;
  uint32_t wait = (had && (gf & GF_PREPARING)) ? S.c_wait[g] : 0u;
  int32_t mem[KMAX], ns[KMAX];
#pragma unroll
  for (int q = 0; q < KMAX; q++) {
    mem[q] = (had && q < k) ? S.members[(int64_t)q * G + g] : 0;
    ns[q] = (had && q < k) ? S.node_slots[(int64_t)q * G + g] : 0;
  }
  bool ns_dirty = false, wait_dirty = false, prop_dirty = false;
  /* this group's carried-over pvalues, by ring index */
  int32_t cs[WMAX], cb[WMAX], cc[WMAX], cf[WMAX];
  int64_t ch[WMAX];
  bool co_loaded = false;
  unsigned long long cmask = 0, co_dirty = 0;
  unsigned long long n_drop = 0;

  /* the pre-active proposals as list entries (slots next - pcount .. next - 1) */
  auto list_preactives = [&](int32_t ix) -> int32_t {
    int32_t cnt = 0;
    const int32_t lo = jsub(next, pcount);
    for_signed_order(lo, pcount, [&](int32_t j) {
      const int32_t s = (int32_t)((uint32_t)lo + (uint32_t)j);
      const int64_t o = (int64_t)(s & Wm) * G + g;
      const uint32_t pe = S.p_ring[o];
      const int64_t q = (int64_t)cnt * n + ix;
      O.e_slot[q] = s;
      O.e_kind[q] = GPX_E_PREACTIVE;
      O.e_handle[q] = S.p_handle[o];
      O.e_flags[q] = (pe & PR_STOP) ? GPX_PV_STOP : 0;
      cnt++;
    });
    return cnt;
  };

  Rec r;
  while (it.next(r)) {
    const int32_t ix = r.idx, acceptor = r.a, first = r.b;
    uint8_t vk = GPX_V_IGNORED;
    int32_t ecount = 0, emed = 0;
    if (!exists) {
      O.status[ix] = GPX_S_NOGROUP;
      n_drop++;
    } else if ((gf & GF_HASCOORD) && (gf & GF_PREPARING)) {
      const int32_t cmp = ballot_cmp(r.bnum, r.bcoord, my_bnum, my_bcoord);
      if (cmp > 0) {
        /* getPreActivesIfPreempted: resign, hand the pre-actives over (PISM:1042-1048) */
        ecount = list_preactives(ix);
        for (int32_t d = 1; d <= pcount; d++) S.p_ring[(int64_t)(jsub(next, d) & Wm) * G + g] = 0;
        pcount = 0;
        prop_dirty = true;
        gf &= ~(GF_HASCOORD | GF_PREPARING);
        vk = GPX_V_PREEMPTED;
      } else if (cmp == 0) {
        int32_t midx = -1;
#pragma unroll
        for (int q = 0; q < KMAX; q++)
          if (q < k && mem[q] == acceptor) midx = q;
        /* canIgnorePrepareReply: non-member or already heard from (PCS:308-313) */
        if (midx >= 0 && !((wait >> midx) & 1u)) {
          if (!co_loaded) {
            for (int32_t w = 0; w < W; w++) {
              const I4 v = S.co_ring[(int64_t)w * G + g];
              cs[EI(w)] = v.x;
              cb[EI(w)] = v.y;
              cc[EI(w)] = v.z;
              cf[EI(w)] = v.w;
              ch[EI(w)] = S.co_handle[(int64_t)w * G + g];
              if (v.w & CO_PRESENT) cmask |= 1ull << w;
            }
            co_loaded = true;
          }
          const int32_t o = I.pv_off[ix], m = I.pv_off[ix + 1] - o;
          /* engine limit: carried slots must not collide in the ring (reply dropped whole); a slice
           * outside the pvalue columns (device-pointer callers: nothing validated it) likewise */
          bool clash = o < 0 || m < 0 || (int64_t)o + m > (int64_t)I.pv_total;
          if (!clash) {
            int32_t claimed[WMAX];
            unsigned long long nm = 0;
            for (int32_t j = 0; j < m; j++) {
              const int32_t s = I.pv_slot[o + j], x = s & Wm;
              if ((nm >> x) & 1ull) {
                if (claimed[EI(x)] != s) clash = true;
              } else if (((cmask >> x) & 1ull) && cs[EI(x)] != s) {
                clash = true;
              }
              claimed[EI(x)] = s;
              nm |= 1ull << x;
            }
          }
          if (clash) {
            O.status[ix] = GPX_S_WINDOW;
            n_drop++;
          } else {
            vk = GPX_V_RECORDED;
            /* PrepareReplyPacket.getMinSlot (PrepareReplyPacket.java:151-164) */
            int32_t ms = first;
            for (int32_t j = 0; j < m; j++) {
              const int32_t s = I.pv_slot[o + j];
              if (jsub(s, ms) < 0) ms = s;
            }
            /* recordSlotNumber(members, preply) (PCS:786-803): wraparound-aware */
#pragma unroll
            for (int q = 0; q < KMAX; q++)
              if (q < k && mem[q] == acceptor && jsub(ns[q], ms) < 0) {
                ns[q] = ms;
                ns_dirty = true;
              }
            /* pmax: per slot the pvalue of the highest ballot (PCS:345-368) */
            for (int32_t j = 0; j < m; j++) {
              const int32_t s = I.pv_slot[o + j], x = s & Wm;
              const int32_t bn = I.pv_bnum[o + j], bc = I.pv_bcoord[o + j];
              if (!((cmask >> x) & 1ull) || ballot_cmp(bn, bc, cb[EI(x)], cc[EI(x)]) > 0) {
                cs[EI(x)] = s;
                cb[EI(x)] = bn;
                cc[EI(x)] = bc;
                cf[EI(x)] = CO_PRESENT | (I.pv_flags ? (int32_t)(I.pv_flags[o + j] & (GPX_PV_STOP | GPX_PV_NOOP)) : 0);
                ch[EI(x)] = I.pv_handle ? I.pv_handle[o + j] : 0;
                cmask |= 1ull << x;
                co_dirty |= 1ull << x;
              }
            }
            wait |= 1u << midx; /* updateHeardFrom */
            wait_dirty = true;
            if (__popc(wait & 0xffffu) > k / 2) {
              /* heardFromMajority: combinePValuesOntoProposals into fe[] / fh[], position j =
               * slot lo + j; committed to p_ring only if it fits `window` slots */
              uint32_t fe[WMAX];
              int64_t fh[WMAX];
              bool fits = true;
              int32_t lo = jsub(next, pcount), pos = pcount;
              if (cmask == 0) {
                /* nothing carried over: myProposals stays as it is (PCS:394-395) */
                ecount = list_preactives(ix);
              } else {
                int32_t maxCarry = 0, maxMin = ns[0];
                bool any = false;
                for (int32_t w = 0; w < W; w++)
                  if ((cmask >> w) & 1ull) {
                    if (!any || jsub(cs[EI(w)], maxCarry) > 0) maxCarry = cs[EI(w)];
                    any = true;
                  }
#pragma unroll
                for (int q = 0; q < KMAX; q++)
                  if (q < k && jsub(ns[q], maxMin) > 0) maxMin = ns[q];
                const int32_t R = jsub(maxCarry, maxMin);
                /* R == INT32_MIN: exactly 2^31 apart (an unheard node's -1 against slot
                 * Integer.MAX_VALUE): the reference's loop would run 2^31 times; refused */
                if (R >= W || R == INT32_MIN) {
                  fits = false;
                } else {
                  const int32_t pre_lo = jsub(next, pcount);
                  unsigned long long removed = 0; /* pre-actives that kept (or lost as duplicates) their slot */
                  bool stop_exists = false, last_stop = false;
                  lo = R >= 0 ? maxMin : (int32_t)((uint32_t)maxCarry + 1u);
                  pos = 0;
                  for (int32_t j = 0; j <= R; j++) {
                    const int32_t cur = (int32_t)((uint32_t)maxMin + (uint32_t)j), x = cur & Wm;
                    const int32_t d = jsub(cur, pre_lo);
                    uint32_t en = 0;
                    int64_t hn = 0;
                    if (((cmask >> x) & 1ull) && cs[EI(x)] == cur) {
                      en = FE_MAKE(GPX_E_CARRY, cf[EI(x)] & GPX_PV_STOP); /* received pvalues dominate */
                      hn = ch[EI(x)];
                    } else if (!(d >= 0 && d < pcount)) {
                      en = FE_MAKE(GPX_E_NOOP, 0); /* neither received nor pre-active */
                    } else {
                      const int64_t po = (int64_t)x * G + g;
                      const uint32_t pe = S.p_ring[po];
                      const int64_t ph = S.p_handle[po];
                      bool dup = false; /* isDuplicate: RequestPacket.equals over the carry-overs */
                      for (int32_t w = 0; w < W; w++) dup |= ((cmask >> w) & 1ull) && ch[EI(w)] == ph;
                      if (!dup) {
                        en = FE_MAKE(GPX_E_PREACTIVE, pe & PR_STOP);
                        hn = ph;
                      }
                      removed |= 1ull << d; /* remove even if duplicate */
                    }
                    fe[EI(pos)] = en;
                    fh[EI(pos)] = hn;
                    if (en & FE_PRESENT) {
                      stop_exists |= (en & FE_STOP) != 0;
                    }
                    last_stop = (en & FE_PRESENT) && (en & FE_STOP);
                    pos++;
                  }
                  /* reproposePreemptedProposals (PCS:460-468) through propose() (:233-241) */
                  auto append = [&](uint32_t en, int64_t hn) {
                    if (pos > 0 && last_stop) return; /* nothing goes after a stop */
                    if (pos < W) {
                      fe[EI(pos)] = en;
                      fh[EI(pos)] = hn;
                    }
                    pos++;
                    stop_exists |= (en & FE_STOP) != 0;
                    last_stop = (en & FE_STOP) != 0;
                  };
                  for_signed_order(pre_lo, pcount, [&](int32_t dd) {
                    if ((removed >> dd) & 1ull) return;
                    const int32_t s = (int32_t)((uint32_t)pre_lo + (uint32_t)dd);
                    const int64_t po = (int64_t)(s & Wm) * G + g;
                    append(FE_MAKE(GPX_E_PREACTIVE, S.p_ring[po] & PR_STOP), S.p_handle[po]);
                  });
                  /* processStop's last step (PCS:512-516): a stop exists but is not the last */
                  if (stop_exists && pos > 0 && !last_stop) append(FE_MAKE(GPX_E_NEWSTOP, 1), 0);
                  fits = pos <= W;
                  if (fits) {
                    for (int32_t d = 1; d <= pcount; d++) S.p_ring[(int64_t)(jsub(next, d) & Wm) * G + g] = 0;
                    int32_t live = 0;
                    for (int32_t j = 0; j < pos; j++)
                      if (fe[EI(j)] & FE_PRESENT) {
                        const int32_t s = (int32_t)((uint32_t)lo + (uint32_t)j);
                        S.p_ring[(int64_t)(s & Wm) * G + g] = PR_PRESENT | ((fe[EI(j)] & FE_STOP) ? PR_STOP : 0u);
                        live++;
                      }
                    next = (int32_t)((uint32_t)lo + (uint32_t)pos);
                    pcount = live;
                    prop_dirty = true;
                    /* spawnCommandersForProposals: TreeMap order */
                    for_signed_order(lo, pos, [&](int32_t j) {
                      if (!(fe[EI(j)] & FE_PRESENT)) return;
                      const int64_t q = (int64_t)ecount * n + ix;
                      O.e_slot[q] = (int32_t)((uint32_t)lo + (uint32_t)j);
                      O.e_kind[q] = (uint8_t)FE_KIND(fe[EI(j)]);
                      O.e_handle[q] = fh[EI(j)];
                      O.e_flags[q] = (fe[EI(j)] & FE_STOP) ? GPX_PV_STOP : 0;
                      ecount++;
                    });
                  }
                }
              }
              if (fits) {
                emed = median_minus<KMAX>(ns, k); /* initCommander: getMajorityCommittedSlot */
                gf &= ~GF_PREPARING;               /* setCoordinatorActive */
                vk = GPX_V_ELECTED;
              } else {
                O.status[ix] = GPX_S_WINDOW; /* recorded, but the view change cannot complete here */
                n_drop++;
              }
            }
          }
        }
      }
    }
    O.v_kind[ix] = vk;
    O.e_count[ix] = ecount;
    O.e_median[ix] = emed;
  }
  if (co_dirty) {
    for (int32_t w = 0; w < W; w++)
      if ((co_dirty >> w) & 1ull) {
        S.co_ring[(int64_t)w * G + g] = I4{cs[EI(w)], cb[EI(w)], cc[EI(w)], cf[EI(w)]};
        S.co_handle[(int64_t)w * G + g] = ch[EI(w)];
      }
  }
  if (ns_dirty) {
#pragma unroll
    for (int q = 0; q < KMAX; q++)
      if (q < k) S.node_slots[(int64_t)q * G + g] = ns[q];
  }
  if (wait_dirty) S.c_wait[g] = wait;
  if (prop_dirty) {
    S.c_next[g] = next;
    S.c_pcount[g] = pcount;
  }
  if (gf != gf0) S.g_flags[g] = gf;
  if (n_drop) atomicAdd(&X.counters[2], n_drop);
}

/* Record payload (k_scatter_ac): a = acceptor, b = firstSlot, bnum / bcoord = the reply's ballot;
 * the accepted pvalues are read through pv_off[idx]. */
template <int KMAX, int WMAX>
__global__ __launch_bounds__(1024) void k_bucket_prepare_reply(DevState S, DevScratch X, PReplyIn I,
                                                               PReplyOut O) {
  extern __shared__ __attribute__((aligned(16))) int32_t lds[];
  BucketView bv;
  if (!bucket_prepare(X, lds, &bv, []() {})) return;
  const int32_t g0 = blockIdx.x << X.shift;
  for (int32_t l = threadIdx.x; l < X.gb; l += (int32_t)blockDim.x) {
    const int32_t c = bv.lcnt[l];
    const int32_t g = g0 + l;
    if (c == 0 || g >= S.G) continue;
    GroupIter it;
    it.init(bv, l, c);
    elect_group<KMAX, WMAX>(S, X, g, it, I, O);
  }
}
