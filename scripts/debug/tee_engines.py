#!/usr/bin/env python
"""Debug aid (GPU box): every Engine call of a test harness goes to the HIP engine AND to a shadow oracle engine with the
same inputs; the first call whose results differ is reported with the records of the group concerned, every earlier call's
records of that group, and both engines' gpx_group_dump of it.  Not a test: tests compare the engine with the Python
readings; this finds WHERE the engine leaves the oracle when such a comparison has failed.

  python scripts/debug/tee_engines.py wrap 0        # tests/test_acc_enum_gpu.py::test_whole_round_across_the_int_wrap case 0
"""
import dataclasses
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gigapaxos_amd._abi as abi  # noqa: E402
from gigapaxos_amd import load_hip  # noqa: E402
from tests.oracle_binding import load_oracle  # noqa: E402

RealEngine = abi.Engine


def flat(x):
    if dataclasses.is_dataclass(x):
        return [v for f in dataclasses.fields(x) for v in flat(getattr(x, f.name))]
    if isinstance(x, (tuple, list)):
        return [v for y in x for v in flat(y)]
    return [x]


class Mismatch(AssertionError):
    pass


class TeeEngine:
    log = []      # (engine id, method, args) of every mirrored call
    watch = None  # (node id, group): both engines' dumps of the group around every call that names it

    def __init__(self, lib, *a, **kw):
        self.hip = RealEngine(load_hip(), *a, **kw)
        self.orc = RealEngine(load_oracle(), *a, **kw)
        self.lib, self.h, self.kmax, self.my_id, self.cfg = self.hip.lib, self.hip.h, self.hip.kmax, self.hip.my_id, self.hip.cfg

    def close(self):
        self.hip.close(), self.orc.close()

    def __getattr__(self, name):
        fh, fo = getattr(self.hip, name), getattr(self.orc, name)

        def call(*a, **kw):
            a2 = [np.array(x, copy=True) if isinstance(x, np.ndarray) else x for x in a]
            TeeEngine.log.append((self.my_id, name, a2, kw))
            w = TeeEngine.watch
            named = (w is not None and w[0] == self.my_id and a2 and isinstance(a2[0], np.ndarray) and a2[0].ndim == 1
                     and a2[0].dtype.kind in "iu" and (a2[0] == w[1]).any())
            if named:
                print(f"-- call {len(TeeEngine.log)} node {self.my_id} {name}: records of group {w[1]}:")
                for j in np.nonzero(a2[0] == w[1])[0].tolist():
                    print("     ", j, [int(c[j]) for c in a2 if isinstance(c, np.ndarray) and c.shape[:1] == a2[0].shape and c.ndim == 1 and c.dtype.kind in "iu"])
                print("   before hip   :", self.hip.dump(w[1]).tolist())
                print("   before oracle:", self.orc.dump(w[1]).tolist())
            rh, ro = fh(*a, **kw), fo(*a2, **kw)
            if named:
                dh, do = self.hip.dump(w[1]).tolist(), self.orc.dump(w[1]).tolist()
                print("   after  hip   :", dh)
                print("   after  oracle:", do, "" if dh == do else "   <<<< STATES DIFFER")
            for k, (x, y) in enumerate(zip(flat(rh), flat(ro))):
                x, y = np.asarray(x), np.asarray(y)
                if x.shape != y.shape or not (x == y).all():
                    self.report(name, a2, k, x, y)
            return rh
        return call

    def report(self, name, args, k, x, y):
        print(f"\n==== node {self.my_id}: {name}: output #{k} differs (shapes {x.shape} / {y.shape})")
        if x.shape == y.shape:
            bad = np.nonzero(x != y)[0] if x.ndim == 1 else np.nonzero((x != y).any(axis=tuple(range(1, x.ndim))))[0]
            i = int(bad[0])
            print(f"first at index {i} of {x.shape[0]} ({bad.shape[0]} differ): hip {x[i]}  oracle {y[i]}")
        else:
            i = None
        g = None
        if i is not None and isinstance(args[0], np.ndarray) and args[0].shape[0] == x.shape[0]:
            g = int(args[0][i])
        elif i is not None and isinstance(args[0], np.ndarray):
            print("(output is compacted: group taken from output #0)")
        if g is None:
            raise Mismatch(name)
        self.found = TeeEngine.found = (self.my_id, g)
        print(f"group {g}; this call's records of it (index: columns):")
        for j in np.nonzero(args[0] == g)[0].tolist():
            print("   ", j, [int(c[j]) for c in args if isinstance(c, np.ndarray) and c.shape[:1] == args[0].shape and c.ndim == 1 and c.dtype.kind in 'iu'])
        print("dump hip   :", self.hip.dump(g).tolist())
        print("dump oracle:", self.orc.dump(g).tolist())
        print("earlier calls naming the group (newest last):")
        for (nid, nm, a, kw) in TeeEngine.log[:-1]:
            if nid != self.my_id or not a or not isinstance(a[0], np.ndarray) or a[0].ndim != 1:
                continue
            for j in np.nonzero(a[0] == g)[0].tolist():
                print("   ", nm, j, [int(c[j]) for c in a if isinstance(c, np.ndarray) and c.shape[:1] == a[0].shape and c.ndim == 1 and c.dtype.kind in 'iu'])
        raise Mismatch(name)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "wrap"
    case = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import tests.round_model as RM
    RM.Engine = TeeEngine
    if what == "wrap":
        base, K, kw = [(2**31 - 6, 3, dict()), (2**31 - 20, 3, dict(p_rival=0.03)), (2**31 - 10, 3, dict(p_stop=0.02, from_disk=False)),
                       (2**31 - 12, 5, dict(p_pause=0.15, pokes=True))][case]
        kw = dict(kw)
        kw.setdefault("from_disk", True)
        try:
            RM.run_rounds(load_hip(), 10_000, 16, 7, p_drop=0.12, K=K, base=base, **kw)
            print("no difference between the HIP engine and the oracle; the harness passed")
        except Mismatch as e:
            print("mismatch in", e)
            if TeeEngine.watch is None:   # the same run again, the group's state printed around every call that names it
                TeeEngine.watch, TeeEngine.log = TeeEngine.found, []
                print("\n######## again, watching node %d group %d" % TeeEngine.watch)
                try:
                    RM.run_rounds(load_hip(), 10_000, 16, 7, p_drop=0.12, K=K, base=base, **kw)
                except Mismatch:
                    pass
        except AssertionError as e:
            print("the harness failed without an engine / oracle difference:", str(e)[:600])


if __name__ == "__main__":
    main()
