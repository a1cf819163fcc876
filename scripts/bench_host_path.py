#!/usr/bin/env python
"""The HOST-pointer path at the bench size (what a JNI host with direct ByteBuffers gets): one step =
gpx_propose_batch(1 M) + gpx_accept_reply_batch(3 M shuffled votes) with every column in host
memory, pageable vs registered (gpx_host_register).  PCIe-inclusive; never the judged `value`."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gigapaxos_amd import Engine, hri_create, load_hip, S_OK  # noqa: E402


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def run(registered, G=1_000_000, K=3, steps=8):
    lib = load_hip()
    e = Engine(lib, 100, G, kmax=K, window=8, max_batch=K * G + 1024)
    mem = np.tile(np.array([100, 101, 102], np.int32), (G, 1))
    assert (e.create_groups(np.arange(G, dtype=np.int32), mem, K, hri_create(G, K, 100)) == S_OK).all()
    rng = np.random.default_rng(1)
    g_all = np.arange(G, dtype=np.int32)
    pm = rng.permutation(K * G)
    n = K * G
    v_g = np.tile(g_all, K)[pm].copy()
    v_acc = np.repeat(np.array([100, 101, 102], np.int32), G)[pm].copy()
    v_bn, v_bc = np.zeros(n, np.int32), np.full(n, 100, np.int32)
    v_slot, v_cp = np.zeros(n, np.int32), np.zeros(n, np.int32)
    p_out = [np.zeros(G, np.int32) for _ in range(4)] + [np.zeros(G, np.uint8)]
    d_out = [np.zeros(n, np.int32) for _ in range(5)] + [np.zeros(n, np.uint8)]
    n_out, v_st = np.zeros(1, np.int32), np.zeros(n, np.uint8)
    bufs = [g_all, v_g, v_acc, v_bn, v_bc, v_slot, v_cp, v_st] + p_out + d_out
    if registered:
        e.host_register(*bufs)
    t = []
    for step in range(steps + 1):
        v_slot[:] = step + 1
        v_cp[:] = step
        t0 = time.perf_counter()
        lib.check(lib.fn["propose_batch"](e.h, G, _p(g_all), None, *[_p(x) for x in p_out]), "propose_batch")
        lib.check(lib.fn["accept_reply_batch"](e.h, n, _p(v_g), _p(v_bn), _p(v_bc), _p(v_slot), _p(v_acc), _p(v_cp),
                                               *[_p(x) for x in d_out], _p(n_out), _p(v_st)), "accept_reply_batch")
        t.append(time.perf_counter() - t0)
        assert int(n_out[0]) == G
    if registered:
        e.host_unregister(*bufs)
    e.close()
    ms = float(np.median(t[1:])) * 1e3
    return {"ms_per_step": round(ms, 3), "votes_per_sec": round(n / ms * 1e3), "decisions_per_sec": round(G / ms * 1e3),
            "bytes_over_pcie_per_step": int(G * 4 + G * 17 + n * 24 + n * 1 + G * 21)}


def main():
    print(json.dumps({"pageable": run(False), "registered": run(True)}))


if __name__ == "__main__":
    main()
