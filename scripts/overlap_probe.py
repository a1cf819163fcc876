#!/usr/bin/env python
"""What would running the PROPOSE kernels beside the ACCEPT-REPLY kernels buy?  (A question for DESIGN 7 "next": inside one
engine the two calls of a step are serial - the second reads what the first wrote - but the accept-reply FRONT end, the
scatter, reads only the call's inputs.)  Measured here with what exists: TWO engines of G groups each on one MI355X, each
with its own stream; per iteration both do propose + accept_reply (so the work is two headline steps), issued

  serial      both engines on ONE stream:            A.propose A.reply B.propose B.reply
  staggered   A and B on their own streams, B half a step behind:  A.propose | B.reply  then  A.reply | B.propose

so that in the staggered form one engine's propose kernels always have the other engine's accept-reply kernels beside them.
Prints milliseconds per headline step (elapsed / (2 x iterations)) for both forms; not the judged bench line."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, streams, S_OK, ORDERED_PROPOSE  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=16)
    ap.add_argument("--engines", type=int, default=2, help="other than 2: E engines, every one a step per iteration (no stagger)")
    a = ap.parse_args()
    G, K = a.groups, 3
    members = [100, 101, 102]
    dev = torch.device("cuda:0")
    P = lambda t: t.data_ptr()  # noqa: E731
    mem = np.tile(np.array(members, np.int32), (G, 1))
    pool = 3 * (a.iters + 1) + 1  # every round of an engine's run is its own (slot numbers follow the proposals)
    # every engine its OWN seeded rounds: two engines reading the same columns would find the second reading in the 256 MB
    # memory-side cache (the first form of this probe did, and flattered the two-stream figure by 6-10 %)
    rounds2 = [[[torch.from_numpy(c).to(dev) for c in streams.vote_round_survey(G, members, r, 100, config_id=3 + 16 * side)]
                for r in range(pool)] for side in range(2)]
    n = int(rounds2[0][0][0].shape[0])
    g = torch.arange(G, dtype=torch.int32, device=dev)

    class Side:
        def __init__(self, stream, side):
            self.rounds = rounds2[side]
            self.e = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=G * K + 4096)
            assert (self.e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
            self.e.set_ordered_batches(ORDERED_PROPOSE)
            self.e.set_stream(stream.cuda_stream)
            self.p = [torch.empty(G, dtype=torch.int32, device=dev) for _ in range(4)] + [torch.empty(G, dtype=torch.uint8, device=dev)]
            self.d = [torch.empty(n + 64, dtype=torch.int32, device=dev) for _ in range(5)] + [torch.empty(n + 64, dtype=torch.uint8, device=dev)]
            self.no, self.st = torch.zeros(1, dtype=torch.int32, device=dev), torch.empty(n + 64, dtype=torch.uint8, device=dev)
            self.r = 0

        def propose(self):
            self.e.call_dev("propose_batch", G, P(g), 0, *[P(t) for t in self.p])

        def reply(self):
            c = self.rounds[self.r]
            self.r += 1
            self.e.call_dev("accept_reply_batch", n, *[P(t) for t in c], *[P(t) for t in self.d], P(self.no), P(self.st))

    def timed(form):
        s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        A, B = Side(s1, 0), Side(s1 if form == "serial" else s2, 1)
        ev0, ev1, evb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
        res = []
        for rep in range(3):
            # B half a step behind A (its first propose is outside the timed region in both forms: same work inside)
            B.propose()
            torch.cuda.synchronize()
            ev0.record(s1)
            for _ in range(a.iters):
                A.propose()
                B.reply()
                A.reply()
                B.propose()
            if form != "serial":
                evb.record(s2)
                s1.wait_event(evb)
            ev1.record(s1)
            torch.cuda.synchronize()
            B.reply()  # (B's last propose gets its replies: the next repetition starts from a clean round)
            torch.cuda.synchronize()
            assert int(A.no) == G and int(B.no) == G, (int(A.no), int(B.no))
            res.append(ev0.elapsed_time(ev1) / (2 * a.iters))
        A.e.close()
        B.e.close()
        return res

    def timed_many(E, form):
        """E engines (every one on its own stream, or all on one): per iteration every engine one step, issued engine by engine
        (no deliberate stagger: independent streams drift apart by themselves); ms per iteration = per E x G groups"""
        ss = [torch.cuda.Stream(device=dev) for _ in range(E)]
        sides = [Side(ss[0] if form == "serial" else ss[e], e % 2) for e in range(E)]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res = []
        for rep in range(3):
            torch.cuda.synchronize()
            ev0.record(ss[0])
            for _ in range(a.iters):
                for sd in sides:
                    sd.propose()
                for sd in sides:
                    sd.reply()
            for e in range(1, E):
                if form != "serial":
                    evb = torch.cuda.Event()
                    evb.record(ss[e])
                    ss[0].wait_event(evb)
            ev1.record(ss[0])
            torch.cuda.synchronize()
            assert all(int(sd.no) == G for sd in sides)
            res.append(ev0.elapsed_time(ev1) / a.iters)
        for sd in sides:
            sd.e.close()
        return res

    if a.engines != 2:
        for form in ("serial", "streams", "serial", "streams"):
            r = timed_many(a.engines, form)
            print(f"{a.engines} engines of {G} groups, {form:8s} ms per step of all engines: " + " ".join(f"{x:.4f}" for x in r), flush=True)
        return
    for form in ("serial", "staggered", "serial", "staggered"):
        r = timed(form)
        print(f"{form:10s} ms per headline step: " + " ".join(f"{x:.4f}" for x in r), flush=True)


if __name__ == "__main__":
    main()
