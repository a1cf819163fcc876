import sys, numpy as np
sys.path.insert(0, '/root/repo')
from gigapaxos_amd import load_hip, Engine
from tests.oracle_binding import load_oracle
from tests import parity_common as pc

hip, orc = load_hip(), load_oracle()
rng = np.random.default_rng(1)
G = 64
eh, eo = pc.make_pair(hip, orc, 100, G, 3, 64)
NODES = [100, 101, 102, 103, 104, 105, 106, 107]
pc.create_mixed_groups(eh, eo, G, 3, NODES, rng)

# monkeypatch engines to log per-call args and compare dumps after each call
calls = []
def wrap(e_h, e_o, name):
    fh, fo = getattr(e_h, name), getattr(e_o, name)
    def both(eng, *a):
        return (fh if eng is e_h else fo)(*a)
    return fh, fo
step = [0]
orig = {}
for name in ("propose", "accept", "accept_reply", "commit"):
    orig[name] = (getattr(eh, name), getattr(eo, name))
def mk(name):
    def h(*a):
        r = orig[name][0](*a)
        calls.append((name, a))
        return r
    def o(*a):
        r = orig[name][1](*a)
        # after oracle applies the same op, compare all dumps
        for g in range(G):
            da, db = eh.dump(g), eo.dump(g)
            if da.tolist() != db.tolist():
                print("DIVERGE after call", len(calls), name, "group", g)
                a0 = [np.asarray(x) if x is not None else None for x in a]
                sel = np.nonzero(a0[0] == g)[0]
                print("records for group:", sel.tolist())
                for x in a0:
                    print("   ", None if x is None else x[sel].tolist())
                print("hip:", da.tolist())
                print("orc:", db.tolist())
                sys.exit(1)
        return r
    return h, o
for name in orig:
    h, o = mk(name)
    setattr(eh, name, h)
    setattr(eo, name, o)
pc.fuzz(eh, eo, G, NODES, rng, steps=250, batch=300)
print("no divergence")
