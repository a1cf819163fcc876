"""The REAL multi-GPU branch of bench.py, executed once on an MI355X before an 8-GPU node ever sees it (VERDICT r5 item
5): `--force-collectives` takes every `world > 1` path with a world of ONE rank under the driver's own launcher -
dist.init_process_group("nccl", device_id=...) loads RCCL, the barriers around the timed regions, the MAX all_reduce of
the elapsed time, the SUM all_reduce of the decisions, the all_gather of the shards' load counters (device tensors), and
the strong-scaling leg (config #4's one space of 1 M groups x 5 hashed over the ranks) all really run.  SURVEY 8(e):
the collective carries counters only; there is none on the decide path."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_nccl_branch_with_a_world_of_one(hip_lib):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1",
                        "--force-collectives", "--steps", "4", "--warmup", "2", "--profile-steps", "1", "--groups", "200000",
                        "--no-cpu-baseline", "--no-end-to-end"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["collectives"]["backend"] == "nccl" and line["collectives"]["forced"]
    # the all_gather of the counters (votes, outputs, dropped) brought back this rank's own
    ctr = line["collectives"]["shard_counters"]
    assert len(ctr) == 1 and ctr[0][0] > 0 and ctr[0][2] == 0
    # the second timed leg: config #4's fixed space on this one rank
    assert line["strong"] and line["strong"]["groups_total"] == 1_000_000 and line["strong"]["replicas"] == 5
    assert line["strong"]["value"] > 0 and line["value"] > 0
    assert line["ms_per_step_spread"]["timed_regions"] == 3 and len(line["ms_per_step_spread"]["all"]) == 3
