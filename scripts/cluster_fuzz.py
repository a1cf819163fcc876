#!/usr/bin/env python
"""Random flag combinations for the C++ host layer's cluster example on the ORACLE build (CPU): node
counts, bursts, a node dying, lossy ACCEPTs / commits, logging delay, small device tables, stops.
Every run must end with `ok` (survivors identical, nothing lost that the scenario does not lose).
    python scripts/cluster_fuzz.py [seed] [runs]
Found in round 1: prepare replies have to include the executed accepts of the decision log
(getLoggedAccepts) or a new coordinator refills a slot another replica has already executed."""
import json, random, subprocess, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.host_cluster_common import build_oracle_cluster  # noqa: E402
exe = os.environ.get("GPX_CLUSTER_EXE") or build_oracle_cluster()  # e.g. a sanitizer build
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
t0 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    nodes = random.choice([3, 3, 5, 7])
    G = random.choice([1, 7, 60, 300, 1500])
    R = random.choice([3, 6, 10])
    args = ["--nodes", str(nodes), "--groups", str(G), "--rounds", str(R), "--seed", str(random.randint(1, 10 ** 6))]
    if random.random() < 0.4: args += ["--burst", str(random.choice([2, 3, 7]))]
    if random.random() < 0.3: args += ["--kill-round", str(random.randrange(R)), "--kill-node", str(random.randrange(nodes))]
    if random.random() < 0.3: args += ["--drop-commits", str(random.choice([50, 200]))]
    if random.random() < 0.3: args += ["--drop-accepts", str(random.choice([50, 150]))]
    if random.random() < 0.3: args += ["--log-delay", str(random.choice([0, 2, 6]))]
    if random.random() < 0.2 and G >= 60 and "--kill-round" not in args:
        args += ["--capacity", str(max(24, G // 3)), "--active", str(max(4, G // 12))]
    if random.random() < 0.15: args += ["--no-batching"]
    if random.random() < 0.15 and not ({"--burst", "--kill-round", "--active", "--drop-commits", "--drop-accepts"} & set(args)): args += ["--stop-last"]
    if random.random() < 0.2: args += ["--value-bytes", str(random.choice([0, 1, 900]))]
    try:
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        out = json.loads(p.stdout.strip().splitlines()[-1]) if p.stdout.strip() else None
        ok = p.returncode == 0 and out and out["ok"] and "runtime error" not in p.stderr and "Sanitizer" not in p.stderr
    except subprocess.TimeoutExpired:
        ok, out, p = False, None, None
    if not ok:
        bad += 1
        print("FAIL", " ".join(args), "rc", p.returncode if p else "timeout", (p.stderr[-200:] if p else ""), flush=True)
print("runs done, failures:", bad, "in %.0f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
