#!/usr/bin/env python
"""Wire-path throughput on one MI355X (not the judged bench line; numbers go to docs/HISTORY.md §wire).

Coordinator side of one round at BASELINE config #3's size, frames resident in HBM:
  2 remote acceptors x G BATCHED_ACCEPT_REPLY frames (one slot each)  -> gpx_wire_decode_dev
  -> gpx_accept_reply_batch_dev (2 G votes -> G decisions)             -> gpx_wire_pack_commits_dev
Per-phase GPU time with torch events on the engine's stream."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, S_OK  # noqa: E402
from gigapaxos_amd import wire as W  # noqa: E402

NOCHECK = os.environ.get("GPX_BENCH_NOCHECK") == "1"  # ablation builds (scripts/ubench/wire_ablation.sh) decode wrong on purpose


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=1_000_000)
    ap.add_argument("--rounds", type=int, default=6)
    args = ap.parse_args()
    G, K = args.groups, 3
    dev = torch.device("cuda:0")
    nfr = 2 * G
    eng = Engine(load_hip(), 100, G, kmax=K, window=8, max_batch=nfr + 1024)
    we = W.WireEngine(eng)
    mem = np.tile(np.array([100, 101, 102], np.int32), (G, 1))
    assert (eng.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
    names = W.fixed_names(np.arange(G))
    nb, noff = names.reshape(-1), (np.arange(G + 1, dtype=np.int32) * names.shape[1])
    st = np.zeros(G, np.uint8)
    rows = np.arange(G, dtype=np.int32)
    nb = np.ascontiguousarray(nb)
    we.lib.check(we.lib.fn["names_bind"](eng.h, G, nb.ctypes.data, noff.ctypes.data, rows.ctypes.data,
                                         st.ctypes.data), "names_bind")
    assert (st == S_OK).all()
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    eng.set_stream(ts.cuda_stream)
    rng = np.random.default_rng(0)
    i32 = lambda n: torch.empty(n, dtype=torch.int32, device=dev)  # noqa: E731
    u8 = lambda n: torch.empty(n, dtype=torch.uint8, device=dev)  # noqa: E731
    P = lambda t: t.data_ptr()  # noqa: E731
    g_all = torch.arange(G, dtype=torch.int32, device=dev)
    p_out = [i32(G) for _ in range(4)] + [u8(G)]
    fst, fg, ft = u8(nfr), i32(nfr), i32(nfr)
    vcols = [i32(nfr) for _ in range(7)]
    counts = torch.zeros(8, dtype=torch.int32, device=dev)
    dcols = [i32(nfr) for _ in range(5)] + [u8(nfr)]
    n_out, vst = torch.zeros(1, dtype=torch.int32, device=dev), u8(nfr)
    cap_bytes = 64 * G
    out = u8(cap_bytes)
    foff = torch.empty(G, dtype=torch.int64, device=dev)
    flen, fgi = i32(G), i32(G)
    nfo, nbo = torch.zeros(1, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    t_dec = t_ar = t_pack = 0.0
    frame_bytes = 0
    for r in range(args.rounds):
        order = rng.permutation(nfr)
        gsel = (order % G).astype(np.int64)
        acc = np.where(order < G, 101, 102)
        buf, off = W.bar_frames_single_slot(names[gsel], 0, acc, 0, 100, r, r + 1)
        frame_bytes = int(off[-1])
        d_buf, d_off = torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev)
        eng.call_dev("propose_batch", G, P(g_all), 0, *[P(t) for t in p_out])
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        W.decode_dev(we, nfr, P(d_buf), P(d_off), P(fst), P(fg), P(ft),
                     votes=(nfr, [P(c) for c in vcols]), counts_ptr=P(counts))
        ev[1].record()
        eng.call_dev("accept_reply_batch", nfr, *[P(vcols[i]) for i in (0, 1, 2, 3, 4, 5)],
                     *[P(c) for c in dcols], P(n_out), P(vst))
        ev[2].record()
        W.pack_commits_dev(we, nfr, P(n_out), [P(c) for c in dcols], P(out), cap_bytes, P(foff), P(flen), P(fgi),
                           P(nfo), P(nbo))
        ev[3].record()
        eng.sync()
        torch.cuda.synchronize()
        assert NOCHECK or (int(counts[0]) == nfr and int(counts[4]) == 0 and int(n_out) == G and int(nfo) == G)
        if r > 0:
            t_dec += ev[0].elapsed_time(ev[1])
            t_ar += ev[1].elapsed_time(ev[2])
            t_pack += ev[2].elapsed_time(ev[3])
    k = args.rounds - 1
    out_bytes = int(nbo)
    # per-kernel split (hipEvents around every launch: a separate, untimed round)
    eng.profile(2)
    eng.call_dev("propose_batch", G, P(g_all), 0, *[P(t) for t in p_out])
    buf, off = W.bar_frames_single_slot(names[gsel], 0, acc, 0, 100, args.rounds, args.rounds + 1)
    d_buf, d_off = torch.from_numpy(buf).to(dev), torch.from_numpy(off).to(dev)
    W.decode_dev(we, nfr, P(d_buf), P(d_off), P(fst), P(fg), P(ft), votes=(nfr, [P(c) for c in vcols]),
                 counts_ptr=P(counts))
    eng.call_dev("accept_reply_batch", nfr, *[P(vcols[i]) for i in (0, 1, 2, 3, 4, 5)],
                 *[P(c) for c in dcols], P(n_out), P(vst))
    W.pack_commits_dev(we, nfr, P(n_out), [P(c) for c in dcols], P(out), cap_bytes, P(foff), P(flen), P(fgi),
                       P(nfo), P(nbo))
    eng.sync()
    kern = {kk: round(v[1] * 1e3, 1) for kk, v in eng.profile_read().items()}
    eng.profile(0)
    # acceptor side: G ACCEPT frames carrying a 64-byte request value each (BASELINE config #2's request
    # size): the frames of a 256-frame tile no longer fit one staging window
    abuf, aoff = W.accept_frames_fixed(names[rng.permutation(G)], 0, 1, 0, 100, 0, 100, value_len=64)
    d_abuf, d_aoff = torch.from_numpy(abuf).to(dev), torch.from_numpy(aoff).to(dev)
    acols = [i32(G) for _ in range(5)] + [u8(G), i32(G), torch.empty(G, dtype=torch.int64, device=dev), i32(G)]
    t_acc = 0.0
    for r in range(args.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        W.decode_dev(we, G, P(d_abuf), P(d_aoff), P(fst), P(fg), P(ft), accepts=(G, [P(c) for c in acols]),
                     counts_ptr=P(counts))
        e1.record()
        eng.sync()
        torch.cuda.synchronize()
        assert NOCHECK or (int(counts[2]) == G and int(counts[4]) == 0), counts.tolist()
        if r > 0:
            t_acc += e0.elapsed_time(e1)
    res = {
        "accept_frames": G, "accept_frame_bytes": int(aoff[-1]), "accept_decode_ms": round(t_acc / k, 4),
        "accept_decode_GBps_frame_bytes": round(int(aoff[-1]) / (t_acc / k) * 1e-6, 1),
        "groups": G, "frames_in": nfr, "frame_bytes_in": frame_bytes, "frames_out": G, "frame_bytes_out": out_bytes,
        "decode_ms": round(t_dec / k, 4), "accept_reply_ms": round(t_ar / k, 4), "pack_commits_ms": round(t_pack / k, 4),
        "decode_frames_per_s": round(nfr / (t_dec / k) * 1e3, 1),
        "decode_GBps_frame_bytes": round(frame_bytes / (t_dec / k) * 1e-6, 1),
        "pack_frames_per_s": round(G / (t_pack / k) * 1e3, 1),
        "end_to_end_decisions_per_s": round(G / ((t_dec + t_ar + t_pack) / k) * 1e3, 1),
        "kernels_us": kern,
    }
    print(json.dumps(res))


if __name__ == "__main__":
    main()
