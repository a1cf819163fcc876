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


def alg_bytes_per_vote(k: int) -> float:
    """A(K) = 48 + 4K + 20/K bytes per accept-reply vote."""
    return 48.0 + 4.0 * k + 20.0 / k


STRONG_GROUPS, STRONG_K = 1_000_000, 5  # BASELINE config #4: the ONE group space the metric is quoted on


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
        if not args.no_promise:
            # the proposal batch is one request per group in gidx order (what RequestBatcher hands over):
            # declared, verified on the device, so the partition path is not even launched for it
            from gigapaxos_amd import ORDERED_PROPOSE, ORDERED_REPLY_RUNS
            eng.set_ordered_batches(ORDERED_PROPOSE | (ORDERED_REPLY_RUNS if args.runs else 0))
        self.mem = mem = np.tile(np.array(members, np.int32), (G, 1))
        assert (eng.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()

        # ---- synthetic stream, resident in HBM before timing ------------------------------
        # rank r owns shard r of a (world*G)-group space; streams are seeded per (config, rank, round)
        self.cfg_id = cfg_id = (3 if K == 3 else 4) + 16 * rank
        # A pool of independently shuffled rounds supplies (gidx, ballot, acceptor); the two columns that
        # depend on the round number - slot = r + 1 and max_cp = r for every vote of round r - are
        # filled on the device, so any --steps fits in memory and start-up time.
        pool_n = min(rounds, 8)
        pool = []
        for r in range(pool_n):
            cols = (streams.vote_round_runs(G, members, r, 100, config_id=cfg_id, mix=mix) if args.runs else
                    streams.vote_round(G, members, r, 100, config_id=cfg_id, shuffled=not args.sorted, mix=mix))
            pool.append([torch.from_numpy(c).to(dev) for c in cols])
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

        def step(r):
            eng.call_dev("propose_batch", G, P(g_all), 0, P(p_slot), P(p_bnum), P(p_bcoord), P(p_med), P(p_st))
            c = vote_cols[r]
            eng.call_dev("accept_reply_batch", nv, P(c[0]), P(c[1]), P(c[2]), P(c[3]), P(c[4]), P(c[5]),
                         P(d_g), P(d_s), P(d_b), P(d_c), P(d_m), P(d_k), n_out[r:].data_ptr(), P(v_st))
        self.step = step
        self._keep = (vote_cols, g_all, p_slot, p_bnum, p_bcoord, p_med, d_g, d_b, d_c, d_m, v_st)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def run_timed(self, warmup, steps):
        torch, dist, eng, step, dev, world, G = self.torch, self.dist, self.eng, self.step, self.dev, self.world, self.G
        for r in range(warmup):
            step(r)
        eng.sync()
        torch.cuda.synchronize()
        self.barrier()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for r in range(warmup, warmup + steps):
            step(r)
        ev1.record()
        eng.sync()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0  # this rank's K steps, all ranks started together; MAX below
        self.barrier()
        torch.cuda.synchronize()
        self.gpu_ms = ev0.elapsed_time(ev1)
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        self.elapsed = elapsed

        # ---- checks outside the timed region ----------------------------------------------
        counts = self.n_out[: warmup + steps].cpu().numpy()
        if not self.mix:
            assert (counts == G).all(), f"expected {G} decisions per round, got {counts[:8]}"
            assert bool((self.p_st == 0).all()) and bool((self.d_k[:G] == 1).all()) \
                and bool((self.d_s[:G] == warmup + steps).all())
        decisions_local = int(counts[warmup:].sum())
        self.shard_counters = None
        if world > 1:
            t = torch.tensor([decisions_local], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.decisions_total = int(t.item())
            # optional telemetry exchange: shard load counters over RCCL (not on the decide path)
            ctr = torch.tensor(eng.counters(), dtype=torch.int64, device=dev)
            allc = [torch.zeros_like(ctr) for _ in range(world)]
            dist.all_gather(allc, ctr)
            self.shard_counters = [c.tolist() for c in allc]
            t = torch.tensor([self.nv * steps], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.votes_total = int(t.item())
        else:
            self.decisions_total = decisions_local
            self.votes_total = self.nv * steps

    def close(self):
        self.eng.sync()
        self.eng.close()
        self._keep = None


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
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--cpu-rounds", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline leg (0 = every host core)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-pointer (PCIe-inclusive) leg")
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
    args = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    torch.zeros(1, device=dev)  # wake the device before the HIP library's own runtime looks for it
    torch.cuda.synchronize()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK

    K = args.k
    steps, warmup, psteps = args.steps, args.warmup, args.profile_steps
    leg = CoordinatorLeg(args, torch, dist, dev, local_rank, rank, world, args.groups, K, args.split_global,
                         rounds=warmup + steps + psteps)
    leg.run_timed(warmup, steps)
    eng, step, rounds = leg.eng, leg.step, leg.rounds
    G, G_global, nv, nv_round, members, mem, cfg_id = leg.G, leg.G_global, leg.nv, leg.nv_round, leg.members, leg.mem, leg.cfg_id
    elapsed, gpu_ms, decisions_total, votes_total = leg.elapsed, leg.gpu_ms, leg.decisions_total, leg.votes_total

    # ---- per-kernel timing with hipEvents on the launch stream (profile pass) ---------
    roofline = None
    kstats = {}
    if psteps > 0:
        eng.sync()
        eng.profile(2)
        for r in range(warmup + steps, rounds):
            step(r)
        torch.cuda.synchronize()
        kstats = eng.profile_read()
        eng.profile(0)
        ar_kernels = {k: v for k, v in kstats.items() if k not in ("k_apply_propose", "k_fill_pr")}
        dom = max(kstats.items(), key=lambda kv: kv[1][1])
        # launches of shared front-end kernels are split between propose and accept-reply:
        # the dominant kernel is identified by total time; its avg duration = total / launches
        dom_name, (dom_launches, dom_ms) = dom
        avg_ms = dom_ms / max(dom_launches, 1)
        units = nv if dom_name not in ("k_apply_propose", "k_fill_pr") else G
        alg = alg_bytes_per_vote(K) * units
        achieved = alg / (avg_ms * 1e-3) / 1e9
        pipe_ms = sum(v[1] for v in kstats.values()) / psteps
        pipe_achieved = alg_bytes_per_vote(K) * nv / (pipe_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process;
        # they come from separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this
        # same command, summarised by scripts/rocprof_summary.py into profiles/pmc_traffic.json
        # (2*FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md §HBM).
        traffic = traffic_raw = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            # the engine's profile labels name the launch (k_bucket_ar16), rocprofv3 the kernel (k_bucket16<0, 4>)
            alias = {"k_bucket_ar16": "k_bucket16", "k_bucket_accept16": "k_bucket16", "k_bucket_commit16": "k_bucket16"}
            kname = alias.get(dom_name, dom_name)
            cands = [(int(k.split("@")[1]), v) for k, v in pmc.items() if k.split("@")[0].split("<")[0] == kname]
            if cands and G == 1_000_000 and K == 3 and not args.mix and not args.sorted:
                _, v = max(cands, key=lambda kv: kv[0])
                traffic, traffic_raw = int(v["hbm_bytes_corrected"]), int(v["hbm_bytes_raw"])
        except (OSError, ValueError, KeyError):
            pass
        roofline = {
            "bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_raw_counters": traffic_raw,
            "traffic_source": None if traffic is None else
            "profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command "
            "(scripts/gpu_visit.sh TAG traffic), not counted in this run",
            "avg_kernel_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(alg),
            "pipeline_ms_per_step": round(pipe_ms, 4),
            "pipeline_achieved": round(pipe_achieved, 1),
            "pipeline_frac": round(pipe_achieved / HBM_PEAK_GBS, 4),
            "kernels_ms_per_step": {k: round(v[1] / psteps, 4) for k, v in sorted(kstats.items())},
        }

    # ---- N > 1: the shape BASELINE's metric is quoted on - ONE space of 1 M groups, 5 replicas, hash-sharded over
    # the ranks (config #4): total work fixed, so this is the strong-scaling figure beside the weak `value` -------
    strong = None
    if world > 1 and not args.split_global and not args.no_strong_leg:
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
        from gigapaxos_amd._abi import _p
        prev_affinity = pin_to_gpu_numa_node(local_rank)  # the batcher thread sits next to its GPU
        ee = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=nv_round + 1024, device=local_rank)
        assert (ee.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        e2e_rounds = 4
        hcols = [[np.ascontiguousarray(c) for c in streams.vote_round(G, members, r, 100, config_id=cfg_id,
                                                                        shuffled=not args.sorted, mix=args.mix)]
                 for r in range(e2e_rounds)]
        hg = np.arange(G, dtype=np.int32)
        ho = [np.zeros(G, np.int32) for _ in range(4)] + [np.zeros(G, np.uint8)]
        hd = [np.zeros(nv, np.int32) for _ in range(5)] + [np.zeros(nv, np.uint8)]
        hno, hst = np.zeros(1, np.int32), np.zeros(nv, np.uint8)
        pinned_rounds = hcols
        pinned = [hg, hno, hst] + ho + hd + [c for cols in hcols for c in cols]
        ee.host_register(*pinned)  # a JNI host registers its direct ByteBuffers once (gpx_host_register)
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
        # the asynchronous calls (gpx_*_batch_async + gpx_engine_wait): two steps in flight, so the inputs of step
        # r + 1 travel to the device while the outputs of step r travel back - both directions of the link busy;
        # in the clean stream every vote carries the ballot (0, 100): the common-ballot form, 16 B per vote
        import ctypes as C
        fn_pa, fn_aa, fn_w = (ee.lib.fn[k] for k in ("propose_batch_async", "accept_reply_batch_async", "engine_wait"))
        ring = []
        for _ in range(2):
            o = [np.zeros(G, np.int32) for _ in range(4)] + [np.zeros(G, np.uint8)]
            d = [np.zeros(nv, np.int32) for _ in range(5)] + [np.zeros(nv, np.uint8)]
            no, st = np.zeros(1, np.int32), np.zeros(nv, np.uint8)
            ring.append((o, d, no, st))  # (pinned below, through the engine that uses them)
        common = not args.mix

        def submit(r):
            o, d, no, st = ring[r & 1]
            c = hcols[r % e2e_rounds]
            tp, ta = C.c_uint64(0), C.c_uint64(0)
            rc = fn_pa(ee.h, G, _p(hg), None, _p(o[0]), _p(o[1]), _p(o[2]), _p(o[3]), _p(o[4]), C.byref(tp))
            rc |= fn_aa(ee.h, nv, _p(c[0]), None if common else _p(c[1]), None if common else _p(c[2]), 0, 100, _p(c[3]),
                        _p(c[4]), _p(c[5]), _p(d[0]), _p(d[1]), _p(d[2]), _p(d[3]), _p(d[4]), _p(d[5]), _p(no), _p(st),
                        C.byref(ta))
            assert rc == 0
            return tp, ta

        def wait(t):
            assert fn_w(ee.h, t[0]) == 0 and fn_w(ee.h, t[1]) == 0
        n_async = 10
        # slot / max_cp of the host rounds are only right for the first e2e_rounds rounds of an engine: a fresh one
        ee.close()
        ee = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=nv_round + 1024, device=local_rank)
        assert (ee.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        hcols = [[np.ascontiguousarray(c) for c in streams.vote_round(G, members, r, 100, config_id=cfg_id,
                                                                        shuffled=not args.sorted, mix=args.mix)]
                 for r in range(n_async)]
        # (gpx_engine_destroy took away every pinning made through the first engine: round 4's unregister-on-destroy)
        pinned_rounds = hcols
        ee.host_register(hg, *[c for cols_ in hcols for c in cols_])
        for o, d, no, st in ring:
            ee.host_register(*(o + d + [no, st]))
        e2e_rounds = n_async
        for t in [submit(0), submit(0)]:  # warm: all four sets of device columns allocated (the repeated
            wait(t)                       # round only brings late votes; one more slot stays outstanding)
        te = time.perf_counter()
        prev = submit(1)
        for r in range(2, n_async):
            cur = submit(r)
            wait(prev)
            prev = cur
        wait(prev)
        te = (time.perf_counter() - te) / (n_async - 1)
        n_dec = int(ring[(n_async - 1) & 1][2][0])
        assert args.mix or n_dec == G
        b_in = G * 4 + nv * (16 if common else 24)
        b_out = G * 17 + nv + n_dec * 21 + 4
        end_to_end = {"ms_per_step": round(te * 1e3, 4), "decisions_per_sec": round(n_dec / te, 1),
                      "votes_per_sec": round(nv / te, 1), "bytes_over_pcie_per_step": int(b_in + b_out),
                      "pcie_in_GBps": round(b_in / te / 1e9, 1), "pcie_out_GBps": round(b_out / te / 1e9, 1),
                      "common_ballot_form": bool(common),
                      "synchronous_calls_ms_per_step": round(te_sync * 1e3, 4),
                      "path": "gpx_propose_batch_async + gpx_accept_reply_batch_async + gpx_engine_wait with HOST "
                              "pointers (registered memory), two steps in flight: H2D of step r + 1 beside the kernels "
                              "and the D2H of step r; synchronous_calls_ms_per_step = the plain calls, one after the other"}
        for o, d, no, st in ring:
            ee.host_unregister(*(o + d + [no, st]))
        ee.host_unregister(hg, *[c for cols_ in pinned_rounds for c in cols_])
        ee.close()
        end_to_end["host_thread_pinned_to_gpu_numa_node"] = prev_affinity is not None
        if prev_affinity is not None:
            os.sched_setaffinity(0, prev_affinity)

    # ---- CPU baseline: the oracle (port of the reference algorithm) on the host cores ---
    cpu_baseline = None
    parity_checked = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests.oracle_binding import load_oracle

        eo = Engine(load_oracle(), 100, G, kmax=K, window=8)
        assert (eo.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        cpu_rounds = args.cpu_rounds
        cols_cpu = [streams.vote_round(G, members, r, 100, config_id=cfg_id, shuffled=not args.sorted,
                                       mix=args.mix) for r in range(cpu_rounds)]
        gnp = np.arange(G, dtype=np.int32)
        tc = time.perf_counter()
        ndec = 0
        oracle_dec = []
        for r in range(cpu_rounds):
            eo.propose(gnp)
            d = eo.accept_reply(*cols_cpu[r])
            ndec += d.gidx.shape[0]
            oracle_dec.append(d)
        tcpu = time.perf_counter() - tc
        if not args.no_parity_check:
            # what is timed is also what is checked: the same rounds through a fresh HIP engine
            # (C-ABI, host pointers), decided stream / per-vote status / HotRestoreInfo rows against
            # the oracle's (outside every timed region)
            ep = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=nv_round + 1024, device=local_rank)
            assert (ep.create_groups(gnp, mem, K, hri_create(G, K, 100)) == S_OK).all()
            ok = True
            for r in range(cpu_rounds):
                ep.propose(gnp)
                d = ep.accept_reply(*cols_cpu[r])
                ok = ok and d.as_tuple_array().shape == oracle_dec[r].as_tuple_array().shape \
                    and bool((d.as_tuple_array() == oracle_dec[r].as_tuple_array()).all()) \
                    and bool((d.status == oracle_dec[r].status).all())
            ok = ok and ep.snapshot(gnp)[0].tobytes() == eo.snapshot(gnp)[0].tobytes()
            ok = ok and ep.counters() == eo.counters()
            ep.close()
            parity_checked = {"rounds": cpu_rounds, "groups": G, "votes_per_round": int(cols_cpu[0][0].shape[0]),
                              "compared": "decisions (gidx, slot, bnum, bcoord, median_cp, kind), per-vote status, "
                                          "HotRestoreInfo rows of every group, counters", "ok": bool(ok)}
            assert ok, "HIP engine and oracle disagree on the bench workload"
        del oracle_dec
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

    if rank == 0:
        out = {
            "metric": "decided_ops_per_sec",
            "value": round(decisions_total / elapsed, 1),
            "unit": "decisions/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(elapsed * 1e3 / steps, 4),
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
                "ordered_proposals_promise": not args.no_promise,
                "parallelism": "groups sharded across GPUs, no collective on the decide path",
            },
            "votes_per_sec": round(votes_total / elapsed, 1),
            "votes_per_sec_per_gpu": round(votes_total / elapsed / world, 1),
            "gpu_ms_per_step_rank0": round(gpu_ms / steps, 4),
            "roofline": roofline,
            "strong": strong,
            "end_to_end": end_to_end,
            "cpu_baseline": cpu_baseline,
            "parity_checked": parity_checked,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
