#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE config #3, one process per GPU.

Workload (per GPU, weak scaling): 1,000,000 Paxos groups x 3 replicas, this engine the
coordinator of every group.  One STEP = one consensus round of the coordinator hot path over the
whole shard: gpx_propose_batch_dev (1 M proposals) + gpx_accept_reply_batch_dev (3 M shuffled
synthetic accept-reply votes -> 1 M decisions).  Every input column is resident in HBM before
the timed region; outputs stay in HBM.

Prints ONE JSON line (rank 0) with the driver's contract fields plus `roofline` and
`cpu_baseline` (the CPU oracle = a restatement of the reference algorithm, "port", timed on a
bounded sample of the same workload on the host cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The parity / CPU-sample legs move pageable numpy arrays with torch: through the runtime's staging buffer, not by pinning
# the arrays' pages per copy (tests/conftest.py has the why; no timed region copies pageable memory - the columns are resident
# in HBM, the end-to-end legs use gpx_host_alloc blocks).  Read when the runtime initialises: before torch is imported.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def pin_to_gpu_numa_node(device_index: int):
    """Runs the calling process on the cores next to the GPU (and so first-touches its host buffers there): a
    DMA from the far socket's memory runs at half the link's rate.  Returns the previous affinity (or None)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        cpus = set()
        for part in open("/sys/bus/pci/devices/%s/local_cpulist" % bdf).read().strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                cpus |= set(range(int(lo), int(hi or lo) + 1))
        prev = os.sched_getaffinity(0)
        if cpus and cpus & prev:
            os.sched_setaffinity(0, cpus & prev)
            return prev
    except (OSError, AttributeError, ValueError, RuntimeError):
        pass
    return None


def csrc_sha16() -> str:
    """Fingerprint of the kernel sources (gigapaxos_amd/csrc/*.hip, *.h, *.inc): profiles/pmc_traffic.meta.json holds the
    one the committed counter passes were taken on, so that a line can say whether its `traffic` is this library's."""
    import hashlib
    d = os.path.join(ROOT, "gigapaxos_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".inc")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def alg_bytes_per_vote(k: int) -> float:
    """A(K) = 48 + 4K + 20/K bytes per accept-reply vote."""
    return 48.0 + 4.0 * k + 20.0 / k


STRONG_GROUPS, STRONG_K = 1_000_000, 5  # BASELINE config #4: the ONE group space the metric is quoted on

# the engine's profile labels name the launch (k_bucket_ar16), rocprofv3 the kernel (k_bucket16<0, 4>)
PMC_ALIAS = {"k_bucket_ar16": "k_bucket16", "k_bucket_accept16": "k_bucket16", "k_bucket_commit16": "k_bucket16",
             "k_ar_runs_pers": "k_ar_runs"}


def gen_round(args, streams, G, members, r, cfg_id, mix):
    """Round r of the synthetic accept-reply stream in the shape the run asked for (every leg draws from here)."""
    if args.runs:
        return streams.vote_round_runs(G, members, r, 100, config_id=cfg_id, mix=mix)
    if stream_name(args, streams) == "survey-8d":
        return streams.vote_round_survey(G, members, r, 100, config_id=cfg_id, shuffled=not args.sorted, mix=mix)
    return streams.vote_round(G, members, r, 100, config_id=cfg_id, shuffled=not args.sorted, mix=mix)


def stream_name(args, streams):
    """Which generator fills the rounds (reported as config.stream): SURVEY.md 8(d)'s own (xorshift64* + Fisher-Yates,
    gigapaxos_amd/native/gpx_streams.c) unless --stream pcg64 asks for the numpy one the tests use or the small C
    library was not built; the runs layout (--runs) has no shuffle to specify and stays with numpy."""
    if args.runs:
        return "pcg64-runs"
    if args.stream == "survey" and streams.native_streams() is not None:
        return "survey-8d"
    return "pcg64"


def link_peaks(torch, dev, eng, mbytes=64, reps=6):
    """What the host link of THIS box moves, measured in the same run as `end_to_end` (its denominator): plain
    hipMemcpyAsync of `mbytes` MB, host -> device and device -> host, alone and both at once, from memory allocated
    for DMA (hipHostMalloc: torch's pinned allocator, = gpx_host_alloc) and from ordinary pages pinned afterwards
    (hipHostRegister through gpx_host_register: what registering a JVM direct buffer gives)."""
    n = mbytes * (1 << 20) // 4
    d_in = torch.empty(n, dtype=torch.int32, device=dev)
    d_out = torch.ones(n, dtype=torch.int32, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def timed(fn_a, fn_b=None):
        best = 0.0
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(s1):
                fn_a()
            if fn_b:
                with torch.cuda.stream(s2):
                    fn_b()
            s1.synchronize()
            s2.synchronize()
            best = max(best, n * 4 / (time.perf_counter() - t0) / 1e9)
        return round(best, 1)

    out = {}
    reg_a, reg_b = np.zeros(n, np.int32), np.zeros(n, np.int32)
    eng.host_register(reg_a, reg_b)
    kinds = {"hipHostMalloc": (torch.empty(n, dtype=torch.int32).pin_memory(), torch.empty(n, dtype=torch.int32).pin_memory()),
             "hipHostRegister": (torch.from_numpy(reg_a), torch.from_numpy(reg_b))}
    for name, (h_a, h_b) in kinds.items():
        h_a.fill_(1)
        pinned = bool(h_a.is_pinned())
        h2d = lambda: d_in.copy_(h_a, non_blocking=True)  # noqa: E731
        d2h = lambda: h_b.copy_(d_out, non_blocking=True)  # noqa: E731
        out[name] = {"h2d_GBps": timed(h2d), "d2h_GBps": timed(d2h), "seen_as_pinned_by_the_runtime": pinned}
        both = timed(h2d, d2h)  # n * 4 bytes each way in the measured time
        out[name]["both_directions_each_GBps"] = both
    eng.host_unregister(reg_a, reg_b)
    out["bytes_per_copy"] = n * 4
    return out


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher around it: re-run this command line under
    torch.distributed.run, N ranks on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve).
    Rank 0's JSON line goes to this process's stdout unchanged."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class CoordinatorLeg:
    """One timed measurement of the coordinator hot path on this rank's engine: set-up (engine, groups, the
    synthetic accept-reply rounds resident in HBM), then run_timed() = the driver's contract (warm-up, barrier +
    synchronize on both sides, exactly `steps` steps, MAX over ranks)."""

    def __init__(self, args, torch, dist, dev, local_rank, rank, world, groups, K, split_global, rounds, mix=None,
                 order=None):
        from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK
        self.torch, self.dist, self.world, self.rank, self.dev = torch, dist, world, rank, dev
        self.coll = world > 1 or args.force_collectives  # (--force-collectives: the N > 1 branch with a world of one)
        # collectives carry telemetry only: over RCCL on the ranks' GPUs, or - ranks sharing ONE device
        # (--same-device) - over gloo on host tensors
        self.cdev = torch.device("cpu") if args.same_device else dev
        mix = args.mix if mix is None else mix
        self.mix, self.K, self.rounds = mix, K, rounds
        G = G_global = groups
        if split_global:
            # this rank's shard of the one global space; its engine indexes the shard densely (ShardMap.local),
            # its stream is generated for exactly its groups - shards are independent (PaxosManager.java:3170-3171)
            from gigapaxos_amd.sharding import ShardMap
            G = int(ShardMap(G_global, world).counts[rank])
        self.G, self.G_global = G, G_global
        self.members = members = list(range(100, 100 + K))
        nv_round = G * K + (G * K // 100 + G * K // 200 + G * K // 1000 if mix else 0)
        if args.runs and mix:
            nv_round = G * K + G * K // 40  # duplicates are drawn per vote: a little slack
        self.nv_round = nv_round
        self.eng = eng = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=nv_round + 1024, device=local_rank)
        # a dedicated (non-default) torch stream carries every engine launch, so torch.cuda.Event
        # and the engine's own hipEvents observe the same stream
        self.tstream = tstream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(tstream)
        assert tstream.cuda_stream != 0
        eng.set_stream(tstream.cuda_stream)
        # the proposal batch is one request per group in gidx order (what RequestBatcher hands over): declared,
        # verified on the device, so the partition path is not even launched for it (--no-promise: mask 0).
        # --runs: the votes are the acceptors' replies concatenated; a regular round's outputs are dense as parked, so
        # the compaction launches are left to gpx_compact_last_dev (GPX_LAZY_OUTPUTS) - and never needed here
        eng.set_ordered_batches(self.promise_mask(args))
        self.mem = mem = np.tile(np.array(members, np.int32), (G, 1))
        assert (eng.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()

        # ---- synthetic stream, resident in HBM before timing ------------------------------
        # rank r owns shard r of a (world*G)-group space; streams are seeded per (config, rank, round)
        self.cfg_id = cfg_id = (3 if K == 3 else 4) + 16 * rank
        # A pool of independently shuffled rounds supplies (gidx, ballot, acceptor); the two columns that
        # depend on the round number - slot = r + 1 and max_cp = r for every vote of round r - are
        # filled on the device, so any --steps fits in memory and start-up time.
        # (round 6: every round of the default run is its own seeded round, as SURVEY 8(d) says - the pool holds
        # warmup + 3 x steps + profile rounds up to --pool; rounds 1-5 reused 8 shuffles modulo 8)
        pool_n = min(rounds, max(1, args.pool))
        self.pool_rounds = pool_n
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:  # (the C generator releases the GIL)
            host_rounds_ = list(ex.map(lambda r: gen_round(args, streams, G, members, r, cfg_id, mix), range(pool_n)))
        pool = [[torch.from_numpy(c).to(dev) for c in cols] for cols in host_rounds_]
        del host_rounds_
        self.nv = nv = int(pool[0][0].shape[0])
        vote_cols = []
        for r in range(rounds):
            c = pool[r % pool_n]
            slot_r = c[3] if r < pool_n else torch.full((nv,), r + 1, dtype=torch.int32, device=dev)
            maxcp_r = c[5] if r < pool_n else torch.full((nv,), r, dtype=torch.int32, device=dev)
            vote_cols.append([c[0], c[1], c[2], slot_r, c[4], maxcp_r])
        g_all = torch.arange(G, dtype=torch.int32, device=dev)
        i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)  # noqa: E731
        u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)  # noqa: E731
        p_slot, p_bnum, p_bcoord, p_med, p_st = i32(G), i32(G), i32(G), i32(G), u8(G)
        d_g, d_s, d_b, d_c, d_m, d_k = i32(nv), i32(nv), i32(nv), i32(nv), i32(nv), u8(nv)
        v_st = u8(nv)
        self.n_out = n_out = torch.zeros(rounds, dtype=torch.int32, device=dev)
        self.p_st, self.d_k, self.d_s = p_st, d_k, d_s
        P = lambda t: t.data_ptr()  # noqa: E731

        def propose_call(r):
            eng.call_dev("propose_batch", G, P(g_all), 0, P(p_slot), P(p_bnum), P(p_bcoord), P(p_med), P(p_st))

        def reply_call(r):
            c = vote_cols[r]
            eng.call_dev("accept_reply_batch", nv, P(c[0]), P(c[1]), P(c[2]), P(c[3]), P(c[4]), P(c[5]),
                         P(d_g), P(d_s), P(d_b), P(d_c), P(d_m), P(d_k), n_out[r:].data_ptr(), P(v_st))

        def step(r):
            propose_call(r)
            reply_call(r)
        self.step, self.propose_call, self.reply_call = step, propose_call, reply_call

        def prepared(r0, r1):
            """The same two calls for the rounds [r0, r1) with their ctypes arguments built beforehand: (propose(r),
            reply(r)).  The two-engine leg issues four calls per step - what Python spends per call on converting sixteen
            arguments would otherwise bound it; a compiled host pays nothing of the kind."""
            import ctypes as C
            lib, VP = eng.lib, C.c_void_p
            f_p, f_r = lib.fn["propose_batch_dev"], lib.fn["accept_reply_batch_dev"]
            a_p = (eng.h, G, VP(P(g_all)), None, VP(P(p_slot)), VP(P(p_bnum)), VP(P(p_bcoord)), VP(P(p_med)), VP(P(p_st)))
            a_r = {r: (eng.h, nv, *[VP(P(t)) for t in vote_cols[r]], VP(P(d_g)), VP(P(d_s)), VP(P(d_b)), VP(P(d_c)), VP(P(d_m)),
                       VP(P(d_k)), VP(n_out[r:].data_ptr()), VP(P(v_st))) for r in range(r0, r1)}

            def propose_prepared(r):
                lib.check(f_p(*a_p), "propose_batch_dev")

            def reply_prepared(r):
                lib.check(f_r(*a_r[r]), "accept_reply_batch_dev")
            return propose_prepared, reply_prepared
        self.prepared = prepared
        self._keep = (vote_cols, g_all, p_slot, p_bnum, p_bcoord, p_med, d_g, d_b, d_c, d_m, v_st)

    @staticmethod
    def promise_mask(args):
        from gigapaxos_amd import ORDERED_PROPOSE, ORDERED_REPLY_RUNS, LAZY_OUTPUTS
        if args.no_promise:
            return 0
        return ORDERED_PROPOSE | ((ORDERED_REPLY_RUNS | LAZY_OUTPUTS) if args.runs else 0)

    def barrier(self):
        if self.coll:
            self.dist.barrier()

    def profile_calls(self, first_round, psteps):
        """Per-kernel launch times (hipEvents on the launch stream, gpx_profile_enable) of `psteps` further steps, kept
        apart per CALL: {"propose_batch": {kernel: (launches, ms)}, "accept_reply_batch": {...}}."""
        eng, torch = self.eng, self.torch
        per_call = {"propose_batch": {}, "accept_reply_batch": {}}
        eng.sync()
        for r in range(first_round, first_round + psteps):
            for name, call in (("propose_batch", self.propose_call), ("accept_reply_batch", self.reply_call)):
                eng.profile(2)
                call(r)
                torch.cuda.synchronize()
                for k, (n, ms) in eng.profile_read().items():
                    a = per_call[name].setdefault(k, [0, 0.0])
                    a[0] += n
                    a[1] += ms
        eng.profile(0)
        return per_call

    def run_timed(self, warmup, steps, reps=1):
        """The driver's contract, `reps` times over: warm-up once, then per repetition barrier + synchronize, EXACTLY
        `steps` steps, synchronize + barrier, MAX over ranks.  self.elapsed = the MEDIAN repetition (what `value` and
        `ms_per_step` are computed from), self.elapsed_all = every repetition's."""
        torch, dist, eng, step, dev, world, G = self.torch, self.dist, self.eng, self.step, self.dev, self.world, self.G
        for r in range(warmup):
            step(r)
        self.elapsed_all, self.gpu_ms_all = [], []
        for rep in range(reps):
            first = warmup + rep * steps
            eng.sync()
            torch.cuda.synchronize()
            self.barrier()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            ev0.record()
            for r in range(first, first + steps):
                step(r)
            ev1.record()
            eng.sync()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0  # this rank's K steps, all ranks started together; MAX below
            self.barrier()
            torch.cuda.synchronize()
            self.gpu_ms_all.append(ev0.elapsed_time(ev1))
            if self.coll:
                t = torch.tensor([elapsed], dtype=torch.float64, device=self.cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                elapsed = float(t.item())
            self.elapsed_all.append(elapsed)
        mid = sorted(range(reps), key=lambda i: self.elapsed_all[i])[reps // 2]
        self.elapsed, self.gpu_ms = self.elapsed_all[mid], self.gpu_ms_all[mid]
        # which way the accept-reply calls delivered their outputs so far (warm-up + timed steps):
        # (written in place by the per-bucket kernel, compacted from the staging by k_emit_dec16)
        self.path_counters = eng.path_counters()
        self.timed_rounds = (warmup + mid * steps, warmup + (mid + 1) * steps)  # the median repetition's rounds
        steps_all = reps * steps

        # ---- checks outside the timed region ----------------------------------------------
        counts = self.n_out[: warmup + steps_all].cpu().numpy()
        if not self.mix:
            assert (counts == G).all(), f"expected {G} decisions per round, got {counts[:8]}"
            assert bool((self.p_st == 0).all()) and bool((self.d_k[:G] == 1).all()) \
                and bool((self.d_s[:G] == warmup + steps_all).all())
        decisions_local = int(counts[self.timed_rounds[0]:self.timed_rounds[1]].sum())
        self.shard_counters = None
        if self.coll:
            t = torch.tensor([decisions_local], dtype=torch.int64, device=self.cdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.decisions_total = int(t.item())
            # optional telemetry exchange: shard load counters over RCCL (not on the decide path)
            ctr = torch.tensor(eng.counters(), dtype=torch.int64, device=self.cdev)
            allc = [torch.zeros_like(ctr) for _ in range(world)]
            dist.all_gather(allc, ctr)
            self.shard_counters = [c.tolist() for c in allc]
            t = torch.tensor([self.nv * steps], dtype=torch.int64, device=self.cdev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.votes_total = int(t.item())
        else:
            self.decisions_total = decisions_local
            self.votes_total = self.nv * steps

    def close(self):
        self.eng.sync()
        self.eng.close()
        self._keep = None


def two_engines_leg(args, torch, dist, dev, local_rank, G, K, warmup, steps, reps):
    """Side figure, never `value`: the SAME table as two independent shards on this GPU - two engines of G / 2 groups, each on
    its own stream with its own seeded shuffled stream, the step = both engines' propose_batch + accept_reply_batch (G
    proposals, K * G votes, G decisions as in the headline step).  Inside one engine the kernels of a step are serial; a second
    engine fills the phases in which the first leaves the memory system idle (profiles/
    r06_two_engines_two_streams.txt).  Shards are independent (PaxosManager.java:3170-3171): no exchange between them."""
    prev = torch.cuda.current_stream()
    sizes = [G - G // 2, G // 2]
    legs = [CoordinatorLeg(args, torch, dist, dev, local_rank, 40 + e, 1, sizes[e], K, False, rounds=warmup + reps * steps)
            for e in range(2)]
    try:
        A, B = legs
        for r in range(warmup):
            for L in legs:
                L.step(r)
        (a_p, a_r), (b_p, b_r) = (L.prepared(warmup, warmup + reps * steps) for L in legs)
        per = []
        for rep in range(reps):
            first = warmup + rep * steps
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            # B runs half a step behind A, so that one engine's proposals have the other's accept replies beside them
            # (every call of the region is issued inside it: `steps` proposes and `steps` reply batches per engine)
            b_p(first)
            for r in range(first, first + steps):
                a_p(r)
                b_r(r)
                a_r(r)
                if r + 1 < first + steps:
                    b_p(r + 1)
            torch.cuda.synchronize()
            per.append(time.perf_counter() - t0)
        mid = sorted(range(reps), key=lambda i: per[i])[reps // 2]
        lo, hi = warmup + mid * steps, warmup + (mid + 1) * steps
        decisions = votes = 0
        for L in legs:
            counts = L.n_out[: warmup + reps * steps].cpu().numpy()
            if not L.mix:
                assert (counts == L.G).all(), f"expected {L.G} decisions per round and engine, got {counts[:8]}"
            decisions += int(counts[lo:hi].sum())
            votes += L.nv * steps
        el = per[mid]
        return {"engines": 2, "streams": 2, "groups_per_engine": sizes, "ms_per_step": round(el * 1e3 / steps, 4),
                "ms_per_step_all": [round(x * 1e3 / steps, 4) for x in per],
                "value": round(decisions / el, 1), "unit": "decisions/s", "votes_per_sec": round(votes / el, 1),
                "workload": "the headline's table as two independent shards on ONE GPU: two engines, each on its own stream "
                            "with its own seeded shuffled stream; step = both engines' propose_batch + accept_reply_batch "
                            "(the headline step's proposals, votes and decisions); a side figure - `value` above is ONE engine"}
    finally:
        for L in legs:
            L.close()
        torch.cuda.set_stream(prev)


def end_to_end_leg(args, torch, dev, local_rank, G, K, nv, nv_round, members, mem, cfg_id):
    """The step through the HOST-pointer entry points, the rate a JNI caller gets: the plain (synchronous) calls, then
    the asynchronous ones two steps deep, and - the denominator - what this box's link moves with plain copies."""
    import ctypes as C
    from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK
    from gigapaxos_amd._abi import _p

    prev_affinity = pin_to_gpu_numa_node(local_rank)  # the batcher thread sits next to its GPU
    hostmalloc = args.e2e_memory == "hostmalloc"
    common = not args.mix  # clean stream: every vote carries the ballot (0, 100) - the common-ballot form, 16 B per vote

    def fresh_engine():
        e = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=nv_round + 1024, device=local_rank)
        assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        return e

    def hbuf(e, n, dtype=np.int32):
        """a host column for engine e: DMA memory from gpx_host_alloc, or numpy pages pinned with gpx_host_register"""
        if hostmalloc:
            return e.host_alloc(n, dtype)
        a = np.zeros(max(n, 1), dtype)[:n]
        e.host_register(a)
        return a

    def host_rounds(e, n_rounds):
        out = []
        for r in range(n_rounds):
            cols = []
            for c in gen_round(args, streams, G, members, r, cfg_id, args.mix):
                b = hbuf(e, c.shape[0])
                b[:] = c
                cols.append(b)
            out.append(cols)
        return out

    # -- the plain calls, one after the other
    ee = fresh_engine()
    link = link_peaks(torch, dev, ee)
    e2e_rounds = 4
    hcols = host_rounds(ee, e2e_rounds)
    hg = hbuf(ee, G)
    hg[:] = np.arange(G, dtype=np.int32)
    ho = [hbuf(ee, G) for _ in range(4)] + [hbuf(ee, G, np.uint8)]
    hd = [hbuf(ee, nv) for _ in range(5)] + [hbuf(ee, nv, np.uint8)]
    hno, hst = hbuf(ee, 1), hbuf(ee, nv, np.uint8)
    fn_p, fn_a = ee.lib.fn["propose_batch"], ee.lib.fn["accept_reply_batch"]

    def host_step(r):
        rc = fn_p(ee.h, G, _p(hg), None, _p(ho[0]), _p(ho[1]), _p(ho[2]), _p(ho[3]), _p(ho[4]))
        c = hcols[r]
        rc |= fn_a(ee.h, nv, _p(c[0]), _p(c[1]), _p(c[2]), _p(c[3]), _p(c[4]), _p(c[5]), _p(hd[0]), _p(hd[1]),
                   _p(hd[2]), _p(hd[3]), _p(hd[4]), _p(hd[5]), _p(hno), _p(hst))
        assert rc == 0
    host_step(0)
    te = time.perf_counter()
    for r in range(1, e2e_rounds):
        host_step(r)
    te_sync = (time.perf_counter() - te) / (e2e_rounds - 1)
    assert args.mix or int(hno[0]) == G
    del hcols, hg, ho, hd, hno, hst, host_step  # views of gpx_host_alloc memory: it goes away with the engine
    ee.close()  # (unpins / frees everything that was pinned or allocated through it)

    # -- the asynchronous calls (gpx_*_batch_async + gpx_engine_wait), two steps in flight: the inputs of step r + 1
    # travel to the device while the outputs of step r travel back - both directions of the link busy.
    # slot / max_cp of a host round are only right once per engine: a fresh one, ten rounds
    n_async = 12
    steps_in_flight = 3  # the copy-out (a kernel writing through the host mapping) is slower than the DMA copy-in: with two
    # steps in flight the next submit waits for it; three keep the inbound link busy (six calls: GPX_ASYNC_DEPTH=6)
    os.environ["GPX_ASYNC_DEPTH"] = str(2 * steps_in_flight)
    ee = fresh_engine()
    del os.environ["GPX_ASYNC_DEPTH"]
    hcols = host_rounds(ee, n_async)
    hg = hbuf(ee, G)
    hg[:] = np.arange(G, dtype=np.int32)
    fn_pa, fn_aa, fn_w = (ee.lib.fn[k] for k in ("propose_batch_async", "accept_reply_batch_async", "engine_wait"))
    ring = []
    for _ in range(steps_in_flight):
        o = [hbuf(ee, G) for _ in range(4)] + [hbuf(ee, G, np.uint8)]
        d = [hbuf(ee, nv) for _ in range(5)] + [hbuf(ee, nv, np.uint8)]
        ring.append((o, d, hbuf(ee, 1), hbuf(ee, nv, np.uint8)))

    def submit(r):
        o, d, no, st = ring[r % steps_in_flight]
        c = hcols[r % n_async]
        tp, ta = C.c_uint64(0), C.c_uint64(0)
        rc = fn_pa(ee.h, G, _p(hg), None, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(o[4]), C.byref(tp))
        rc |= fn_aa(ee.h, nv, _p(c[0]), None if common else _p(c[1]), None if common else _p(c[2]), 0, 100, _p(c[3]),
                    _p(c[4]), _p(c[5]), _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(d[4]), _p(d[5]), _p(no), _p(st),
                    C.byref(ta))
        assert rc == 0
        return tp, ta

    def wait(t):
        assert fn_w(ee.h, t[0]) == 0 and fn_w(ee.h, t[1]) == 0
    for t in [submit(0) for _ in range(steps_in_flight)]:  # warm: every set of device columns allocated (the repeated
        wait(t)                                             # round only brings late votes; one more slot stays outstanding)
    from collections import deque
    te = time.perf_counter()
    flying = deque()
    for r in range(1, n_async):
        flying.append(submit(r))
        if len(flying) == steps_in_flight:
            wait(flying.popleft())
    while flying:
        wait(flying.popleft())
    te = (time.perf_counter() - te) / (n_async - 1)
    n_dec = int(ring[(n_async - 1) % steps_in_flight][2][0])
    assert args.mix or n_dec == G
    b_in = G * 4 + nv * (16 if common else 24)
    b_out = G * 17 + nv + n_dec * 21 + 4
    del hcols, hg, ring, submit, o, d  # views of gpx_host_alloc memory: it goes away with the engine
    ee.close()
    if prev_affinity is not None:
        os.sched_setaffinity(0, prev_affinity)
    kind = "hipHostMalloc" if hostmalloc else "hipHostRegister"
    peak_in, peak_out = link[kind]["both_directions_each_GBps"], link[kind]["both_directions_each_GBps"]
    # the step moves b_in in and b_out out concurrently: the bounding direction is the one closer to what the link
    # gives each way when both are busy
    in_frac, out_frac = (b_in / te / 1e9) / max(peak_in, 1e-9), (b_out / te / 1e9) / max(peak_out, 1e-9)
    return {"ms_per_step": round(te * 1e3, 4), "decisions_per_sec": round(n_dec / te, 1),
            "votes_per_sec": round(nv / te, 1), "bytes_over_pcie_per_step": int(b_in + b_out),
            "pcie_in_GBps": round(b_in / te / 1e9, 1), "pcie_out_GBps": round(b_out / te / 1e9, 1),
            "host_memory": kind + (" (gpx_host_alloc)" if hostmalloc else " (gpx_host_register on numpy pages)"),
            "link": link,
            "achieved_over_link_peak": {"in": round(in_frac, 3), "out": round(out_frac, 3),
                                        "peak_used": "link.%s.both_directions_each_GBps (same run, same box)" % kind},
            "common_ballot_form": bool(common),
            "synchronous_calls_ms_per_step": round(te_sync * 1e3, 4),
            "host_thread_pinned_to_gpu_numa_node": prev_affinity is not None,
            "steps_in_flight": steps_in_flight,
            "path": "gpx_propose_batch_async + gpx_accept_reply_batch_async + gpx_engine_wait with HOST "
                    "pointers, three steps in flight (GPX_ASYNC_DEPTH=6): H2D of the next steps beside the kernels and the D2H of step r; "
                    "synchronous_calls_ms_per_step = the plain calls, one after the other"}


def wire_end_to_end_leg(args, torch, dev, local_rank, G, K, link, steps=6, in_flight=3):
    """The same step as the NIO thread would hand it over: the remote acceptors' BATCHED_ACCEPT_REPLY
    frames in host memory (BatchedAcceptReply.java:103-173: one frame per group and acceptor, one slot each) -> copy in
    -> gpx_wire_decode_dev -> gpx_accept_reply_batch_dev -> gpx_wire_pack_commits_dev -> BATCHED_COMMIT frames
    (BatchedCommit.java:184-215) copied back to host memory, `in_flight` steps deep on three streams (copy in / the
    engine's / copy out).  The coordinator's own vote never was a frame: K - 1 frames per group come in.  Frames are
    built beforehand (a messenger's work, not the engine's)."""
    from gigapaxos_amd import Engine, hri_create, load_hip, S_OK
    from gigapaxos_amd import wire as W
    P = lambda t: t.data_ptr()  # noqa: E731
    members = list(range(100, 100 + K))
    nfr = (K - 1) * G
    eng = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=nfr + 1024, device=local_rank)
    we = W.WireEngine(eng)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    assert (eng.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
    names = W.fixed_names(np.arange(G))
    nb, noff = np.ascontiguousarray(names.reshape(-1)), (np.arange(G + 1, dtype=np.int32) * names.shape[1])
    st, rows = np.zeros(G, np.uint8), np.arange(G, dtype=np.int32)
    we.lib.check(we.lib.fn["names_bind"](eng.h, G, nb.ctypes.data, noff.ctypes.data, rows.ctypes.data, st.ctypes.data), "names_bind")
    assert (st == S_OK).all()
    s_k, s_in, s_out = (torch.cuda.Stream(device=dev) for _ in range(3))
    eng.set_stream(s_k.cuda_stream)
    rng = np.random.default_rng(6)
    # host side: the frames of every step (pinned), the frames coming back
    h_buf, h_off = [], []
    for r in range(steps):
        order = rng.permutation(nfr)
        buf, off = W.bar_frames_single_slot(names[(order % G).astype(np.int64)], 0, np.asarray(members[1:], np.int32)[order // G], 0, 100, r, r + 1)
        h_buf.append(torch.from_numpy(buf).pin_memory())
        h_off.append(torch.from_numpy(off).pin_memory())
    frame_bytes, off_bytes = int(h_buf[0].numel()), int(h_off[0].numel()) * 8
    cap_bytes = 64 * G
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)  # noqa: E731
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)  # noqa: E731
    g_all = torch.arange(G, dtype=torch.int32, device=dev)
    p_out = [i32(G) for _ in range(4)] + [u8(G)]
    sets = []
    for _ in range(in_flight):
        sets.append(dict(
            d_buf=torch.empty(frame_bytes, dtype=torch.uint8, device=dev), d_off=torch.empty(nfr + 1, dtype=torch.int64, device=dev),
            fst=u8(nfr), fg=i32(nfr), ft=i32(nfr), vcols=[i32(nfr) for _ in range(7)],
            counts=torch.zeros(8, dtype=torch.int32, device=dev), dcols=[i32(nfr) for _ in range(5)] + [u8(nfr)],
            n_out=torch.zeros(1, dtype=torch.int32, device=dev), vst=u8(nfr), out=u8(cap_bytes),
            foff=torch.empty(G, dtype=torch.int64, device=dev), flen=i32(G), fgi=i32(G),
            nfo=torch.zeros(1, dtype=torch.int32, device=dev), nbo=torch.zeros(1, dtype=torch.int64, device=dev),
            h_out=torch.empty(cap_bytes, dtype=torch.uint8).pin_memory(), h_nfo=torch.zeros(1, dtype=torch.int32).pin_memory(),
            h_nbo=torch.zeros(1, dtype=torch.int64).pin_memory(),
            ev_in=torch.cuda.Event(), ev_k=torch.cuda.Event(), ev_out=torch.cuda.Event(), used=False))
    out_frame_bytes = [0]

    def submit(r):
        S = sets[r % in_flight]
        with torch.cuda.stream(s_in):
            if S["used"]:
                s_in.wait_event(S["ev_k"])  # the kernels that read this set's frames are done
            S["d_buf"].copy_(h_buf[r], non_blocking=True)
            S["d_off"].copy_(h_off[r], non_blocking=True)
            S["ev_in"].record(s_in)
        with torch.cuda.stream(s_k):
            s_k.wait_event(S["ev_in"])
            if S["used"]:
                s_k.wait_event(S["ev_out"])  # ... and its frames of the step before have left
            eng.call_dev("propose_batch", G, P(g_all), 0, *[P(t) for t in p_out])
            W.decode_dev(we, nfr, P(S["d_buf"]), P(S["d_off"]), P(S["fst"]), P(S["fg"]), P(S["ft"]),
                         votes=(nfr, [P(c) for c in S["vcols"]]), counts_ptr=P(S["counts"]))
            eng.call_dev("accept_reply_batch", nfr, *[P(S["vcols"][i]) for i in range(6)], *[P(c) for c in S["dcols"]],
                         P(S["n_out"]), P(S["vst"]))
            W.pack_commits_dev(we, nfr, P(S["n_out"]), [P(c) for c in S["dcols"]], P(S["out"]), cap_bytes, P(S["foff"]),
                               P(S["flen"]), P(S["fgi"]), P(S["nfo"]), P(S["nbo"]))
            S["ev_k"].record(s_k)
        with torch.cuda.stream(s_out):
            s_out.wait_event(S["ev_k"])
            nbytes = out_frame_bytes[0] or cap_bytes  # (known after the first step: every round packs the same frames)
            S["h_out"][:nbytes].copy_(S["out"][:nbytes], non_blocking=True)
            S["h_nfo"].copy_(S["nfo"], non_blocking=True)
            S["h_nbo"].copy_(S["nbo"], non_blocking=True)
            S["ev_out"].record(s_out)
        S["used"] = True
        return S

    def wait(S):
        S["ev_out"].synchronize()
        assert int(S["h_nfo"][0]) == G, (int(S["h_nfo"][0]), G)
        return int(S["h_nbo"][0])
    out_frame_bytes[0] = wait(submit(0))  # warm: every buffer touched, the outgoing size known
    from collections import deque
    t0 = time.perf_counter()
    flying = deque()
    for r in range(1, steps):
        flying.append(submit(r))
        if len(flying) == in_flight:
            wait(flying.popleft())
    while flying:
        wait(flying.popleft())
    te = (time.perf_counter() - t0) / (steps - 1)
    torch.cuda.synchronize()
    eng.close()
    b_in, b_out = frame_bytes + off_bytes, out_frame_bytes[0] + 12
    peak = link["hipHostMalloc"]["both_directions_each_GBps"]
    return {"ms_per_step": round(te * 1e3, 4), "decisions_per_sec": round(G / te, 1), "votes_per_sec": round(nfr / te, 1),
            "frames_in": nfr, "frame_bytes_in": frame_bytes, "frame_offsets_bytes_in": off_bytes,
            "frames_out": G, "frame_bytes_out": out_frame_bytes[0],
            "pcie_in_GBps": round(b_in / te / 1e9, 1), "pcie_out_GBps": round(b_out / te / 1e9, 1),
            "achieved_over_link_peak": {"in": round(b_in / te / 1e9 / max(peak, 1e-9), 3), "out": round(b_out / te / 1e9 / max(peak, 1e-9), 3),
                                        "peak_used": "link.hipHostMalloc.both_directions_each_GBps (same run, same box)"},
            "steps_in_flight": in_flight,
            "path": "pinned host frames -> hipMemcpyAsync -> gpx_wire_decode_dev -> gpx_accept_reply_batch_dev -> "
                    "gpx_wire_pack_commits_dev -> hipMemcpyAsync -> pinned host frames; %d BATCHED_ACCEPT_REPLY frames of one slot "
                    "(%d bytes each, + 8 bytes of offset) in, %d BATCHED_COMMIT frames out per step" %
                    (nfr, frame_bytes // max(nfr, 1), G)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--groups", type=int, default=1_000_000, help="groups per GPU")
    ap.add_argument("--k", type=int, default=3, help="replicas per group")
    ap.add_argument("--sorted", action="store_true", help="votes sorted by group instead of shuffled")
    ap.add_argument("--runs", action="store_true",
                    help="side figure, not the headline: the votes as K ascending runs (the acceptors' replies "
                         "concatenated, what a coordinator really receives) under the GPX_ORDERED_REPLY_RUNS "
                         "promise - the sorted-runs path, no partition")
    ap.add_argument("--mix", action="store_true", help="adversarial mix (dups / stale / higher ballot)")
    ap.add_argument("--stream", choices=("survey", "pcg64"), default="survey",
                    help="survey: SURVEY.md 8(d)'s generator (xorshift64* + Fisher-Yates, in C); pcg64: the numpy one of the tests")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--cpu-rounds", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline leg (0 = every host core)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg")
    ap.add_argument("--two-engines", action="store_true",
                    help="add the side leg `two_engines_per_gpu`: the table as two shards (two engines, two streams) on the "
                         "GPU (off by default: the default command launches the headline's kernels at the headline's sizes "
                         "only, so that a kernel trace of it averages what the line reports)")
    ap.add_argument("--no-parity-check", action="store_true", help="skip the engine-vs-oracle replay of the CPU sample")
    ap.add_argument("--cpu-mt-passes", type=int, default=16, help="passes over the sample in the multi-threaded leg")
    ap.add_argument("--split-global", action="store_true",
                    help="BASELINE config #4's shape: ONE space of --groups groups hash-sharded over the ranks "
                         "(fmix32(gidx) %% world, gigapaxos_amd.sharding.ShardMap), each rank an independent "
                         "engine over its shard: total work fixed -> strong scaling")
    ap.add_argument("--dry-run-ranks", action="store_true",
                    help="launcher check: start the ranks (gloo), agree on the world, print n_gpus; no engine, no GPU")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="N > 1 only: skip the second timed leg on BASELINE config #4's fixed space (the `strong` object)")
    ap.add_argument("--no-promise", action="store_true",
                    help="do not declare the proposal batches ordered (gpx_engine_set_ordered_batches)")
    ap.add_argument("--timed-regions", type=int, default=3,
                    help="the timed region (exactly --steps steps between barrier + synchronize) is run this many times; "
                         "ms_per_step / value are the median region's, ms_per_step_spread has them all")
    ap.add_argument("--pool", type=int, default=96,
                    help="independently seeded rounds kept in HBM (72 MB each at 1 M groups x 3): every step of the default "
                         "run has its own; a longer run reuses them modulo this")
    ap.add_argument("--force-collectives", action="store_true",
                    help="take the N > 1 branch with a world of ONE rank: dist.init_process_group('nccl'), the barriers, the "
                         "device-tensor all_reduce / all_gather and the strong-scaling leg really execute over RCCL on one "
                         "GPU (tests/test_bench_collectives_gpu.py)")
    ap.add_argument("--no-wire-leg", action="store_true", help="skip end_to_end.wire (frames in host memory -> frames back)")
    ap.add_argument("--same-device", action="store_true",
                    help="N > 1 on a box with ONE GPU: every rank runs on device 0 and the ranks talk over gloo.  Runs "
                         "the whole N > 1 code path (two timed legs, max-over-ranks, the counters' all_gather); the "
                         "figures are N processes sharing one GPU - NOT a scaling measurement")
    ap.add_argument("--stamp-traffic", action="store_true",
                    help="write profiles/pmc_traffic.meta.json for the current kernel sources (after refreshing "
                         "profiles/pmc_traffic.json from `scripts/gpu_visit.sh TAG traffic`) and exit")
    ap.add_argument("--e2e-memory", choices=("registered", "hostmalloc"), default="hostmalloc",
                    help="host buffers of the end-to-end leg: numpy pages pinned with gpx_host_register, or memory "
                         "from gpx_host_alloc (hipHostMalloc)")
    args = ap.parse_args()
    if args.stamp_traffic:
        meta = {"csrc_sha16": csrc_sha16(), "what": "fingerprint of gigapaxos_amd/csrc/*.hip, *.h, *.inc at the time "
                "profiles/pmc_traffic.json's counter passes were taken (bench.py csrc_sha16)"}
        json.dump(meta, open(os.path.join(ROOT, "profiles", "pmc_traffic.meta.json"), "w"), indent=1)
        print(json.dumps(meta))
        return 0

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        # `python bench.py --gpus N` by hand: become the launcher the driver would have used, one rank per GPU
        # (torch.distributed.run, rendezvous on 127.0.0.1); the ranks re-enter main() with WORLD_SIZE set
        return launch_ranks(args.gpus)
    world = int(world_env or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_env is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher starts one rank per GPU")

    import torch
    import torch.distributed as dist

    if args.dry_run_ranks:
        # launcher check without an engine or a GPU (tests/test_bench_launcher.py): the ranks rendezvous over
        # gloo, agree on the world, and rank 0 prints the line's launcher-dependent fields
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            t = torch.tensor([rank + 1], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            assert int(t.item()) == world * (world + 1) // 2
            from gigapaxos_amd.sharding import ShardMap
            shard_groups = [int(c) for c in ShardMap(STRONG_GROUPS, world).counts]
            dist.barrier()
            dist.destroy_process_group()
        else:
            shard_groups = [STRONG_GROUPS]
        if rank == 0:
            print(json.dumps({"metric": "decided_ops_per_sec", "n_gpus": world, "dry_run_ranks": True,
                              "ranks_started": world, "steps": args.steps, "warmup": args.warmup,
                              "scaling": "strong" if args.split_global else "weak",
                              "strong_shard_groups": shard_groups}))
        return 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    if args.same_device:
        local_rank = 0
        # the engines of the other ranks are invisible to this process: the exchange kernels' grids are sized for a
        # device shared by `world` processes (include/gpx.h)
        os.environ["GPX_DEVICE_SHARERS"] = str(world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.zeros(1, device=dev)  # wake the device before the HIP library's own runtime looks for it
    torch.cuda.synchronize()
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.same_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK

    K = args.k
    steps, warmup, psteps = args.steps, args.warmup, args.profile_steps
    REPS = max(1, args.timed_regions)
    collectives = world > 1 or args.force_collectives
    leg = CoordinatorLeg(args, torch, dist, dev, local_rank, rank, world, args.groups, K, args.split_global,
                         rounds=warmup + REPS * steps + psteps)
    leg.run_timed(warmup, steps, REPS)
    eng, step, rounds = leg.eng, leg.step, leg.rounds
    leg_pool_rounds = leg.pool_rounds
    G, G_global, nv, nv_round, members, mem, cfg_id = leg.G, leg.G_global, leg.nv, leg.nv_round, leg.members, leg.mem, leg.cfg_id
    elapsed, gpu_ms, decisions_total, votes_total = leg.elapsed, leg.gpu_ms, leg.decisions_total, leg.votes_total

    # ---- per-kernel timing with hipEvents on the launch stream (profile pass), kept apart per call ---------
    roofline = None
    if psteps > 0:
        per_call = leg.profile_calls(warmup + REPS * steps, psteps)
        kstats = {}
        for call in per_call.values():
            for k, (nl, ms) in call.items():
                a = kstats.setdefault(k, [0, 0.0])
                a[0] += nl
                a[1] += ms
        ar = per_call["accept_reply_batch"]
        # the dominant kernel of the unit's call (one accept-reply vote): by total time
        dom_name, (dom_launches, dom_ms) = max(ar.items(), key=lambda kv: kv[1][1])
        avg_ms = dom_ms / max(dom_launches, 1)
        alg = alg_bytes_per_vote(K) * nv
        achieved = alg / (avg_ms * 1e-3) / 1e9
        call_ms = sum(v[1] for v in ar.values()) / psteps
        call_achieved = alg / (call_ms * 1e-3) / 1e9
        pipe_ms = sum(v[1] for v in kstats.values()) / psteps
        pipe_achieved = alg / (pipe_ms * 1e-3) / 1e9
        # HBM traffic: PMC counters cannot be read from inside this process; they come from separate
        # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this same command, summarised by
        # scripts/rocprof_summary.py into profiles/pmc_traffic.json (2*FETCH_SIZE + WRITE_SIZE, the gfx950
        # correction of MI355X_MICROARCH.md section HBM).  Only the headline shape has such a file.
        traffic = traffic_raw = traffic_call = None
        traffic_call_kernels = None
        traffic_same_source = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            try:  # were the committed passes taken on THESE kernel sources?  (stamped by bench.py --stamp-traffic)
                meta = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.meta.json")))
                traffic_same_source = meta.get("csrc_sha16") == csrc_sha16()
            except (OSError, ValueError):
                pass

            def pmc_of(name):
                kname = PMC_ALIAS.get(name, name)
                cands = [(int(k.split("@")[1]), v) for k, v in pmc.items() if k.split("@")[0].split("<")[0] == kname]
                return max(cands, key=lambda kv: kv[0])[1] if cands else None
            if G == 1_000_000 and K == 3 and not args.mix and not args.sorted and not args.no_promise:
                v = pmc_of(dom_name)
                if v:
                    traffic, traffic_raw = int(v["hbm_bytes_corrected"]), int(v["hbm_bytes_raw"])
                parts = {k: pmc_of(k) for k in ar}
                if all(parts.values()):
                    traffic_call_kernels = {k: int(v["hbm_bytes_corrected"]) for k, v in parts.items()}
                    traffic_call = sum(traffic_call_kernels.values())
        except (OSError, ValueError, KeyError):
            pass
        roofline = {
            "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_raw_counters": traffic_raw,
            "traffic_source": None if traffic is None else
            "profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
            "(scripts/gpu_visit.sh TAG traffic), not counted in this run",
            # True: the passes were taken on the kernel sources this library was built from; False: a kernel source has
            # changed since (re-run the passes); None: no stamp
            "traffic_taken_on_these_kernel_sources": None if traffic is None else traffic_same_source,
            "avg_kernel_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(alg),
            # `frac` is the prescribed per-kernel figure: the whole call's algorithmic bytes over ONE kernel's time.
            # The unit - one accept-reply vote - is finished by all kernels of its call: `call_frac` divides the
            # same bytes by their sum, `traffic_call` is what the counters saw them move together
            "call_kernels": sorted(ar), "call_ms": round(call_ms, 4), "call_achieved": round(call_achieved, 1),
            "call_frac": round(call_achieved / HBM_PEAK_GBS, 4), "traffic_call": traffic_call,
            "traffic_call_kernels": traffic_call_kernels,
            "pipeline_ms_per_step": round(pipe_ms, 4),
            "pipeline_achieved": round(pipe_achieved, 1),
            "pipeline_frac": round(pipe_achieved / HBM_PEAK_GBS, 4),
            "kernels_ms_per_step": {k: round(v[1] / psteps, 4) for k, v in sorted(kstats.items())},
        }

    # ---- N > 1: the shape BASELINE's metric is quoted on - ONE space of 1 M groups, 5 replicas, hash-sharded over
    # the ranks (config #4): total work fixed, so this is the strong-scaling figure beside the weak `value` -------
    strong = None
    if collectives and not args.split_global and not args.no_strong_leg:
        leg.close()
        sleg = CoordinatorLeg(args, torch, dist, dev, local_rank, rank, world, STRONG_GROUPS, STRONG_K, True,
                              rounds=warmup + steps, mix=False)
        sleg.run_timed(warmup, steps)
        strong = {"value": round(sleg.decisions_total / sleg.elapsed, 1), "unit": "decisions/s",
                  "ms_per_step": round(sleg.elapsed * 1e3 / steps, 4), "scaling": "strong",
                  "votes_per_sec": round(sleg.votes_total / sleg.elapsed, 1),
                  "groups_total": STRONG_GROUPS, "replicas": STRONG_K, "groups_rank0": sleg.G,
                  "workload": "BASELINE config #4: %d Paxos groups x %d replicas hash-sharded over %d GPUs "
                              "(fmix32(gidx) %% n), independent shards, same step" % (STRONG_GROUPS, STRONG_K, world)}
        sleg.close()

    # ---- end to end through the HOST-pointer entry points (what a JNI caller with direct ByteBuffers
    # gets): every input column crosses PCIe in, every output column comes back; never `value` -------
    end_to_end = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        end_to_end = end_to_end_leg(args, torch, dev, local_rank, G, K, nv, nv_round, members, mem, cfg_id)
        if not args.no_wire_leg and not args.mix and not args.runs and not args.sorted:
            end_to_end["wire"] = wire_end_to_end_leg(args, torch, dev, local_rank, G, K, end_to_end["link"])

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores ---
    cpu_baseline = None
    parity_checked = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests.oracle_binding import load_oracle

        eo = Engine(load_oracle(), 100, G, kmax=K, window=8)
        assert (eo.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        cpu_rounds = args.cpu_rounds
        cols_cpu = [gen_round(args, streams, G, members, r, cfg_id, args.mix) for r in range(cpu_rounds)]
        gnp = np.arange(G, dtype=np.int32)
        from gigapaxos_amd import LAZY_OUTPUTS
        mask = CoordinatorLeg.promise_mask(args)
        eo.set_ordered_batches(mask & ~LAZY_OUTPUTS)  # the same promises (their refusals are part of the answer)
        tc = time.perf_counter()
        ndec = 0
        oracle_dec, oracle_prop = [], []
        for r in range(cpu_rounds):
            oracle_prop.append(eo.propose(gnp))
            d = eo.accept_reply(*cols_cpu[r])
            ndec += d.gidx.shape[0]
            oracle_dec.append(d)
        tcpu = time.perf_counter() - tc
        if not args.no_parity_check:
            # what is timed is what is checked: the sample's rounds through a fresh HIP engine by the SAME entry points
            # as the timed leg - gpx_propose_batch_dev / gpx_accept_reply_batch_dev on device columns, on a torch
            # stream, under the same gpx_engine_set_ordered_batches mask - all five proposal columns, the decided
            # stream, per-vote status, HotRestoreInfo rows and counters against the oracle's (outside every timed region)
            ep = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=nv_round + 1024, device=local_rank)
            assert (ep.create_groups(gnp, mem, K, hri_create(G, K, 100)) == S_OK).all()
            ep.set_stream(leg.tstream.cuda_stream)
            ep.set_ordered_batches(mask)
            P = lambda t: t.data_ptr()  # noqa: E731
            i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)  # noqa: E731
            u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)  # noqa: E731
            g_dev = torch.arange(G, dtype=torch.int32, device=dev)
            ok = True
            kernels_seen = set()
            for r in range(cpu_rounds):
                vc = [torch.from_numpy(np.ascontiguousarray(c)).to(dev) for c in cols_cpu[r]]
                n_r = int(vc[0].shape[0])
                pr = [i32(G) for _ in range(4)] + [u8(G)]
                dd = [i32(n_r) for _ in range(5)] + [u8(n_r)]
                no, vst = torch.zeros(1, dtype=torch.int32, device=dev), u8(n_r)
                torch.cuda.synchronize()
                ep.profile(2)
                ep.call_dev("propose_batch", G, P(g_dev), 0, *[P(t) for t in pr])
                ep.call_dev("accept_reply_batch", n_r, *[P(t) for t in vc], *[P(t) for t in dd], P(no), P(vst))
                ep.sync()
                kernels_seen |= set(ep.profile_read())
                ep.profile(0)
                if int(no.item()) < 0:  # GPX_LAZY_OUTPUTS: an unusual batch's outputs are still parked
                    ep.compact_last_dev()
                    ep.sync()
                m = int(no.item())
                got = np.stack([t[:m].cpu().numpy().astype(np.int32) for t in dd], axis=1)
                want = oracle_dec[r].as_tuple_array()
                ok = ok and got.shape == want.shape and bool((got == want).all()) \
                    and bool((vst.cpu().numpy() == oracle_dec[r].status).all()) \
                    and all(bool((x.cpu().numpy() == y).all()) for x, y in zip(pr, oracle_prop[r]))
            ok = ok and ep.snapshot(gnp)[0].tobytes() == eo.snapshot(gnp)[0].tobytes()
            ok = ok and ep.counters() == eo.counters()
            ep.close()
            timed_kernels = set(roofline["kernels_ms_per_step"]) if roofline else None
            parity_checked = {"rounds": cpu_rounds, "groups": G, "votes_per_round": int(cols_cpu[0][0].shape[0]),
                              "entry_points": "gpx_propose_batch_dev + gpx_accept_reply_batch_dev (device columns, the "
                                              "timed leg's stream and gpx_engine_set_ordered_batches mask %d)" % mask,
                              "kernels": sorted(kernels_seen),
                              "same_kernels_as_timed_leg": None if timed_kernels is None else kernels_seen == timed_kernels,
                              "compared": "proposals (slot, bnum, bcoord, median_cp, status), decisions (gidx, slot, bnum, "
                                          "bcoord, median_cp, kind), per-vote status, HotRestoreInfo rows of every group, "
                                          "counters", "ok": bool(ok)}
            assert ok, "HIP engine and oracle disagree on the bench workload"
        del oracle_dec, oracle_prop
        single = {"decisions_per_sec": round(ndec / tcpu, 1),
                  "votes_per_sec": round(cols_cpu[0][0].shape[0] * cpu_rounds / tcpu, 1), "seconds": round(tcpu, 2)}
        # the same oracle on T host threads, thread t owning the groups with gidx % T == t (groups are
        # independent: what PaxosManager's demultiplexer thread pool exploits, PACKET_DEMULTIPLEXER_THREADS).
        # The baseline is the BEST the host does: T swept over {32, 64, 128, all cores} (std::map nodes of very
        # many threads fight over the memory system: fewer threads can be faster), every figure in the line.
        import threading

        ncpu = os.cpu_count() or 1
        lib_o = load_oracle()

        def mt_leg(T, passes):
            parts = []  # one stable partition of every sample round by owner thread (gidx % T)
            for cols in cols_cpu:
                key = cols[0] % T
                order = np.argsort(key, kind="stable")
                bounds = np.searchsorted(key[order], np.arange(T + 1))
                parts.append((order, bounds))
            shards = []
            for t in range(T):
                gs = np.arange(t, G, T, dtype=np.int32)
                es = Engine(lib_o, 100, max(1, gs.shape[0]), kmax=K, window=8)
                if gs.shape[0]:
                    assert (es.create_groups(np.arange(gs.shape[0], dtype=np.int32), mem[gs], K,
                                             hri_create(gs.shape[0], K, 100)) == S_OK).all()
                rounds_t = []
                for cols, (order, bounds) in zip(cols_cpu, parts):
                    sel = order[bounds[t]:bounds[t + 1]]
                    rounds_t.append([np.ascontiguousarray(cols[0][sel] // T)] +
                                    [np.ascontiguousarray(c[sel]) for c in cols[1:]])
                shards.append((es, np.arange(gs.shape[0], dtype=np.int32), rounds_t))
            counts_t = [0] * T

            def work(t):
                es, gl, rounds_t = shards[t]
                for _ in range(passes):  # the sample's rounds again with slot / max_cp moved on
                    for cols in rounds_t:
                        es.propose(gl)
                        counts_t[t] += es.accept_reply(*cols).gidx.shape[0]
                    for cols in rounds_t:
                        cols[3] += cpu_rounds
                        cols[5] += cpu_rounds

            threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
            tm = time.perf_counter()
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            tmt = time.perf_counter() - tm
            for es, _, _ in shards:
                es.close()
            assert args.mix or sum(counts_t) == ndec * passes
            return sum(counts_t) / tmt, tmt

        if args.cpu_threads > 0:
            sweep_T = [max(1, min(args.cpu_threads, ncpu))]
        else:
            sweep_T = sorted({t for t in (32, 64, 128, ncpu) if t <= ncpu} or {ncpu})
        passes = max(1, args.cpu_mt_passes // max(1, len(sweep_T) // 2))
        sweep = {}
        for T in sweep_T:
            sweep[T] = mt_leg(T, passes)
        best_T = max(sweep, key=lambda t: sweep[t][0])
        cpu_baseline = {
            "value": round(sweep[best_T][0], 1), "unit": "decisions/s", "cores": best_T, "kind": "port",
            "votes_per_sec": round(sweep[best_T][0] * cols_cpu[0][0].shape[0] / max(ndec / cpu_rounds, 1), 1),
            "host_cores_available": ncpu,
            "threads_sweep_decisions_per_sec": {str(t): round(v[0], 1) for t, v in sorted(sweep.items())},
            "sample": f"{cpu_rounds * passes} rounds of the same workload ({G} groups, {cols_cpu[0][0].shape[0]} votes/round) "
                      f"per thread count, C++ oracle (std::map restatement of the Java; not the JVM), groups partitioned "
                      f"gidx % T; value = the best of T in {sorted(sweep)}",
            "seconds": round(sum(v[1] for v in sweep.values()), 2),
            "single_thread": single,
        }
        eo.close()

    # ---- side figure: the same table as two shards on this GPU, two engines on two streams (never `value`) -------------
    two_engines = None
    if rank == 0 and world == 1 and not collectives and not args.same_device and args.two_engines \
            and not args.split_global and not args.runs and G >= 2048:
        try:
            two_engines = two_engines_leg(args, torch, dist, dev, local_rank, G, K, warmup, steps, REPS)
        except Exception as ex:  # noqa: BLE001  (a side leg never takes the line with it)
            two_engines = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0:
        out = {
            "metric": "decided_ops_per_sec",
            "value": round(decisions_total / elapsed, 1),
            "unit": "decisions/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(elapsed * 1e3 / steps, 4),
            # the timed region (barrier + synchronize, exactly `steps` steps, synchronize + barrier, MAX over ranks) run
            # `timed_regions` times on consecutive rounds; `ms_per_step` and `value` are the MEDIAN region's
            "ms_per_step_spread": {"timed_regions": REPS, "min": round(min(leg.elapsed_all) * 1e3 / steps, 4),
                                   "median": round(elapsed * 1e3 / steps, 4), "max": round(max(leg.elapsed_all) * 1e3 / steps, 4),
                                   "all": [round(x * 1e3 / steps, 4) for x in leg.elapsed_all]},
            "higher_is_better": True,
            "scaling": "strong" if args.split_global else "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE config #4: %d Paxos groups x %d replicas hash-sharded over %d GPU(s) "
                             "(fmix32(gidx) %% n), independent shards" % (G_global, K, world) if args.split_global else
                             "BASELINE config #3: %d Paxos groups x %d replicas per GPU" % (G, K)) +
                            ", synthetic %s accept-reply stream%s; step = propose_batch(G) + accept_reply_batch(K*G "
                            "votes), inputs resident in HBM"
                            % ("sorted" if args.sorted else "shuffled", " + adversarial mix" if args.mix else ""),
                "groups_per_gpu": G, "groups_total": G_global if args.split_global else G * world,
                "replicas": K, "votes_per_step_per_gpu": nv,
                "stream": {"survey-8d": "SURVEY.md 8(d): xorshift64* seeded 0x9E3779B97F4A7C15 ^ (config << 32) ^ round, "
                                        "Fisher-Yates per group and over the round (gigapaxos_amd/native/gpx_streams.c)",
                           "pcg64": "numpy PCG64 with the same seed (gigapaxos_amd/streams.py vote_round)",
                           "pcg64-runs": "numpy PCG64, K ascending runs (gigapaxos_amd/streams.py vote_round_runs)"
                           }[stream_name(args, streams)],
                "rounds_seeded_individually": leg_pool_rounds,
                "ordered_proposals_promise": not args.no_promise,
                "ordered_batches_mask": CoordinatorLeg.promise_mask(args),
                "parallelism": "groups sharded across GPUs, no collective on the decide path",
            },
            "votes_per_sec": round(votes_total / elapsed, 1),
            "votes_per_sec_per_gpu": round(votes_total / elapsed / world, 1),
            "gpu_ms_per_step_rank0": round(gpu_ms / steps, 4),
            # gpx_engine_path_counters after the warm-up and the timed regions (rank 0): accept-reply calls whose outputs
            # the per-bucket kernel wrote in their final place / calls k_emit_dec16 compacted from the staging
            "accept_reply_outputs": {"in_place_calls": leg.path_counters[0], "compacted_calls": leg.path_counters[1]},
            "roofline": roofline,
            "strong": strong,
            "end_to_end": end_to_end,
            "two_engines_per_gpu": two_engines,
            "cpu_baseline": cpu_baseline,
            "parity_checked": parity_checked,
        }
        if args.force_collectives:
            out["collectives"] = {"backend": dist.get_backend(), "world": world, "forced": True,
                                  "note": "the N > 1 branch executed with a world of one rank: init_process_group, barriers, "
                                          "all_reduce(MAX / SUM) and all_gather on device tensors over RCCL, the strong leg",
                                  "shard_counters": leg.shard_counters}
        if args.same_device:
            out["same_device"] = {"ranks_on_device_0": world, "collectives": "gloo (host tensors)",
                                  "note": "ONE GPU shared by %d processes: exercises the N > 1 code path (both timed legs, "
                                          "max-over-ranks, the counters' all_gather); NOT a scaling figure" % world}
            out["shard_counters"] = leg.shard_counters
        print(json.dumps(out))
    if world > 1 or args.force_collectives:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
