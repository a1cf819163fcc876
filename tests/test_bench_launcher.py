"""`bench.py --gpus N` really starts N ranks.  Run with --dry-run-ranks: the launcher, the rendezvous (gloo here) and the
line's launcher-dependent fields, without an engine or a GPU.  Two ways in: by hand (bench.py re-runs itself under
torch.distributed.run) and the driver's own command line (torch.distributed.run around bench.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out       # rank 0 prints ONE JSON line, the other ranks none
    return json.loads(lines[0])


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_flag_spawns_that_many_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks", "--steps", "7", "--warmup", "2"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_started"] == 2 and line["steps"] == 7 and line["warmup"] == 2
    # config #4's fixed space really is split: the two shards add up to the one million groups
    assert len(line["strong_shard_groups"]) == 2 and sum(line["strong_shard_groups"]) == 1_000_000


def test_driver_command_line():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run-ranks"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2


def test_single_rank_stays_in_process():
    r = subprocess.run([sys.executable, BENCH, "--dry-run-ranks"], capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 1


def test_world_and_flag_must_agree():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-ranks"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_same_device_flag_goes_through_the_launcher():
    """`--gpus 2 --same-device` (round 5: N ranks on ONE GPU over gloo) is an ordinary launch as far as the launcher goes."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--same-device", "--dry-run-ranks"], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2


def test_traffic_stamp_is_a_source_fingerprint():
    """profiles/pmc_traffic.meta.json (bench.py --stamp-traffic): the fingerprint of the kernel sources the committed
    counter passes were taken on, in the form bench.py compares its own tree with."""
    import json
    import re
    import bench
    here = bench.csrc_sha16()
    assert re.fullmatch(r"[0-9a-f]{16}", here)
    meta = json.load(open(os.path.join(bench.ROOT, "profiles", "pmc_traffic.meta.json")))
    assert re.fullmatch(r"[0-9a-f]{16}", meta["csrc_sha16"])
