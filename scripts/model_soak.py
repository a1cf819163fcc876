#!/usr/bin/env python3
"""Fresh-seed soak of the Java readings against the CPU oracle (no GPU): every iteration draws a whole-round scenario
(group size 1-16, loss 0-40 %, rival ballots, stops, in-memory accepts, view change + rounds after, pauses, pokes, repeated
PREPARE_REPLYs, instances across Integer.MAX_VALUE), a set of vote streams and a set of acceptor op sequences.
Usage: PYTHONPATH=. python scripts/model_soak.py SEED SECONDS"""
import sys, time, traceback
import numpy as np
import tests.oracle_binding as ob
from tests.round_model import run_rounds
from tests.pcs_enum_common import run_streams
import tests.acc_enum_common as A
lib = ob.load_oracle()
rng = np.random.default_rng(int(sys.argv[1]))
t0 = time.time(); n = 0; bad = 0
while time.time() - t0 < float(sys.argv[2]):
    seed = int(rng.integers(1, 10**6))
    K = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 5, 7, 9, 16]))
    failover = bool(rng.random() < 0.4) and K >= 2
    base = 0 if failover or rng.random() < 0.6 else int(rng.choice([2**31 - int(rng.integers(2, 30)), -2**31 + int(rng.integers(0, 5))]))
    kw = dict(p_drop=float(rng.choice([0.0, 0.05, 0.12, 0.25, 0.4])), K=K, p_rival=float(rng.choice([0.0, 0.0, 0.03, 0.08])) if K >= 2 else 0.0,
              p_stop=float(rng.choice([0.0, 0.0, 0.02, 0.06])), from_disk=bool(rng.random() < 0.6), failover=failover,
              rounds_after=int(rng.integers(2, 9)) if failover else 0, p_pause=float(rng.choice([0.0, 0.1, 0.3])), pokes=True,
              p_dup_reply=float(rng.choice([0.0, 0.3])), p_double=float(rng.choice([0.1, 0.3, 0.6])), base=base)
    G, rounds = int(rng.integers(200, 1500)), int(rng.integers(6, 26))
    try:
        run_rounds(lib, G, rounds, seed, **kw)
        run_streams(lib, int(rng.choice([3, 4, 5, 7])), int(rng.integers(1, 7)), 2000, int(rng.integers(8, 60)), seed=seed,
                    p_stranger=float(rng.choice([0.0, 0.05])), p_extreme=float(rng.choice([0.0, 0.1])), base=int(rng.choice([0, 2**31 - 3, 2**31 - 1])))
        idx = rng.integers(0, len(A.WIDE), (3000, int(rng.integers(3, 14))))
        A.run_sequences(lib, [tuple(A.WIDE[i] for i in row) for row in idx.tolist()], init=str(rng.choice(["create", "initial"])),
                        order=str(rng.choice(["interleaved", "grouped"])), from_disk=bool(rng.random() < 0.7),
                        base=int(rng.choice([0, 0, 2**31 - 2, 2**31 - 1, -2**31 + 2])))
        n += 1
    except Exception as ex:
        bad += 1
        print("FAIL", seed, G, rounds, kw, str(ex)[:300], flush=True)
        traceback.print_exc()
        if bad >= 3: break
print(f"# {n} fresh-seed iterations clean, {bad} failed, {time.time()-t0:.0f} s", flush=True)
