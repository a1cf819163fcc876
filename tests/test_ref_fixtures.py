"""Fixtures produced by the REFERENCE itself (scripts/make_ref_fixtures.sh: the Java classes of this path
replaying recorded coordinator streams).  Present only once someone has run that script on a box with a
JDK; without them the tests skip and docs/HISTORY.md keeps saying "parity unpinned by reference fixtures"."""
import glob
import os

import numpy as np
import pytest

from gigapaxos_amd import Engine, hri_create, S_OK

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "ref_*.npz")))


def _replay(lib, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_stream", os.path.join(os.path.dirname(GOLD), "..", "scripts", "ref_fixtures", "make_stream.py"))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    z = np.load(path)
    G, K, me, members, rounds = ms.rounds_of(str(z["case"]))
    e = Engine(lib, me, G, kmax=K, window=8, max_batch=1 << 18)
    mem = np.tile(np.array(members, np.int32), (G, 1))
    assert (e.create_groups(np.arange(G), mem, K, hri_create(G, K, me)) == S_OK).all()
    for r, (pg, cols) in enumerate(rounds):
        sl, bn, bc, md, st = e.propose(pg)
        ref = z["prop%d" % r]
        ok = ref[:, 4] == 1
        assert ((st == S_OK) == ok).all()
        assert (np.stack([sl, bn, bc, md], 1)[ok] == ref[ok, :4]).all(), f"round {r}: ACCEPTs differ from the Java"
        d = e.accept_reply(*cols)
        rd = z["dec%d" % r]
        rd = rd[np.argsort(rd[:, 1], kind="stable")]  # arrival order -> grouped by gidx (the output contract)
        assert d.as_tuple_array().shape == rd[:, 1:].shape, f"round {r}: decision count differs from the Java"
        assert (d.as_tuple_array() == rd[:, 1:]).all(), f"round {r}: decided stream differs from the Java"


@pytest.mark.skipif(not FIXTURES, reason="no reference-generated fixtures (needs a JDK: scripts/make_ref_fixtures.sh)")
@pytest.mark.parametrize("path", FIXTURES or ["-"])
def test_oracle_matches_reference_fixture(oracle_lib, path):
    _replay(oracle_lib, path)


@pytest.mark.gpu
@pytest.mark.skipif(not FIXTURES, reason="no reference-generated fixtures (needs a JDK: scripts/make_ref_fixtures.sh)")
@pytest.mark.parametrize("path", FIXTURES or ["-"])
def test_engine_matches_reference_fixture(hip_lib, path):
    _replay(hip_lib, path)
