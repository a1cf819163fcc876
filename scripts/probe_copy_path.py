#!/usr/bin/env python
"""Which way does the HIP runtime take a PAGEABLE host <-> device copy of a given size: through its own pinned staging
buffer, or by pinning the caller's pages for the duration of the copy (DmaBlitManager::hsaCopyStagedOrPinned ->
VirtualGPU::addPinnedMem - the path the main thread sat in at every one of the GPU page faults on file,
profiles/r06_abort_backtrace.txt)?  Runs itself once per (environment, size) with AMD_LOG_LEVEL=4 and counts the runtime's
own "HSA Copy Using Pinned / Staging resource" lines; also times the copies.

    python scripts/probe_copy_path.py            # the table
    python scripts/probe_copy_path.py child MB   # one size (used by the parent)
"""
import os
import subprocess
import sys
import time


def child(mb):
    import numpy as np
    import torch

    dev = torch.device("cuda:0")
    torch.zeros(1, device=dev)
    n = int(mb * (1 << 20)) // 4
    a = np.arange(n, dtype=np.int32)
    b = np.empty_like(a)
    torch.cuda.synchronize()
    sys.stderr.write("PROBE-BEGIN\n")
    sys.stderr.flush()
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        t = torch.from_numpy(a).to(dev)
        torch.from_numpy(b).copy_(t)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    sys.stderr.write("PROBE-END\n")
    sys.stderr.flush()
    assert (a == b).all()
    print(f"{dt * 1e3:.3f}")


def main():
    envs = {
        "default": {},
        "GPU_PINNED_MIN_XFER_SIZE=1048576": {"GPU_PINNED_MIN_XFER_SIZE": "1048576"},
    }
    print(f"{'environment':36s} {'MB':>6s} {'pinned':>7s} {'staging':>8s} {'ms / (H2D + D2H)':>18s}")
    for name, extra in envs.items():
        for mb in (0.0625, 0.5, 2, 16, 64, 200):
            env = dict(os.environ, AMD_LOG_LEVEL="4", **extra)
            p = subprocess.run([sys.executable, __file__, "child", str(mb)], env=env, capture_output=True, text=True,
                               timeout=300)
            err = p.stderr
            seg = err.split("PROBE-BEGIN")[-1].split("PROBE-END")[0] if "PROBE-BEGIN" in err else ""
            pinned = seg.count("Copy Using Pinned")
            staging = seg.count("Copy Using Staging")
            print(f"{name:36s} {mb:6g} {pinned:7d} {staging:8d} {p.stdout.strip() or ('rc=%d' % p.returncode):>18s}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(float(sys.argv[2]))
    else:
        main()
