#!/usr/bin/env python
"""Several engines on ONE MI355X (DESIGN 5 / INTEGRATION 4: how to hold more than ~2 M groups per GPU
at full speed).  E engines of 1 M groups x 3 replicas each, every engine the coordinator of its own
groups; one step = for every engine propose(1 M) + accept_reply(3 M shuffled votes), all columns
resident in HBM, the engines' calls issued round-robin on one caller stream (each engine pipelines
its own front and back end on its own two streams).  Not the judged bench line."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, S_OK  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engines", type=int, default=16)
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=6)
    args = ap.parse_args()
    E, G, K = args.engines, args.groups, 3
    dev = torch.device("cuda:0")
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    P = lambda t_: t_.data_ptr()  # noqa: E731
    ids = [100, 101, 102]
    mem = np.tile(np.array(ids, np.int32), (G, 1))
    engs = []
    for _ in range(E):
        e = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=K * G + 1024)
        assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
        e.set_stream(ts.cuda_stream)
        engs.append(e)
    g_all = torch.arange(G, dtype=torch.int32, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(7)
    pm = torch.randperm(K * G, device=dev, generator=gen)
    acc_col = torch.cat([torch.full((G,), nid, dtype=torch.int32, device=dev) for nid in ids])
    v_g, v_acc = g_all.repeat(K)[pm].contiguous(), acc_col[pm].contiguous()
    v_bn = torch.zeros(K * G, dtype=torch.int32, device=dev)
    v_bc = torch.full((K * G,), 100, dtype=torch.int32, device=dev)
    v_slot, v_cp = (torch.empty(K * G, dtype=torch.int32, device=dev) for _ in range(2))
    # outputs are per engine (the calls of different engines overlap)
    outs = [dict(p=[torch.empty(G, dtype=torch.int32, device=dev) for _ in range(4)]
                 + [torch.empty(G, dtype=torch.uint8, device=dev)],
                 d=[torch.empty(K * G, dtype=torch.int32, device=dev) for _ in range(5)]
                 + [torch.empty(K * G, dtype=torch.uint8, device=dev)],
                 n=torch.zeros(1, dtype=torch.int32, device=dev),
                 st=torch.empty(K * G, dtype=torch.uint8, device=dev)) for _ in range(E)]
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    total_ms = 0.0
    for step in range(args.steps + 1):
        v_slot.fill_(step + 1)
        v_cp.fill_(step)
        if step == 1:
            ev0.record()
        for e, o in zip(engs, outs):
            e.call_dev("propose_batch", G, P(g_all), 0, *[P(x) for x in o["p"]])
            e.call_dev("accept_reply_batch", K * G, P(v_g), P(v_bn), P(v_bc), P(v_slot), P(v_acc), P(v_cp),
                       *[P(x) for x in o["d"]], P(o["n"]), P(o["st"]))
    ev1.record()
    for e in engs:
        e.sync()
    torch.cuda.synchronize()
    for o in outs:
        assert int(o["n"]) == G
    ms = ev0.elapsed_time(ev1) / args.steps
    print(json.dumps({"engines": E, "groups_per_engine": G, "groups_on_gpu": E * G, "ms_per_step": round(ms, 4),
                      "decisions_per_sec": round(E * G / ms * 1e3), "votes_per_sec": round(E * K * G / ms * 1e3)}))
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
