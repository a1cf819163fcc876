"""The acceptor side of the HIP engine against an independent Python reading of the Java
(tests/acc_enum_common.py: PaxosAcceptor.java:302-385, 462-506; PaxosInstanceStateMachine.java:1080-1166,
1432-1528, 1619-1701), bounded exhaustive: every sequence of ACCEPT / DECISION / batched-commit ops of
length <= 2 over 300 ops, every length-3 sequence over 90 ops, every length-4 sequence over 22 ops,
seeded random sequences of length 5 and 6 - one group per sequence, a group's consecutive same-call ops in
one batch, batches both interleaved (partition path) and grouped by group (direct path)."""
import pytest

pytestmark = pytest.mark.gpu


def test_acceptor_side_enumerated_against_java_reading_gpu(hip_lib):
    import tests.acc_enum_common as A
    for k in A.COVERAGE:
        A.COVERAGE[k] = 0
    # (round 5: the two exhaustive plans - 729,000 and 456,976 sequences - see one batch order each instead of both, and
    # the second pass leaves them out, the random plans run at a quarter of their size: 77 -> about 30 s of the GPU
    # suite, every coverage counter still hit)
    n = A.run_plan(hip_lib, scale=0.25, single_order_above=100_000)
    n += A.run_plan(hip_lib, scale=0.05, from_disk=(False,), skip=("len3-medium", "len4-small"))
    assert n > 2_000_000
    assert all(v > 0 for v in A.COVERAGE.values()), A.COVERAGE


def test_acceptor_side_long_random_sequences_gpu(hip_lib):
    """Seeded random sequences of 8 and 12 ops per group, 70,000 groups per length, both batch orders."""
    import tests.acc_enum_common as A
    assert A.run_long_random(hip_lib, 70_000) > 2_500_000


def test_acceptor_side_enumerated_under_the_ordered_promise(hip_lib):
    """The same readings with gpx_engine_set_ordered_batches(ACCEPT | COMMIT): grouped batches keep the
    promise, so only the direct kernels run (no partition path launched behind them)."""
    import tests.acc_enum_common as A
    n = A.run_plan(hip_lib, scale=0.1, orders=("grouped",), promise=True)
    assert n > 400_000


@pytest.mark.parametrize("K,nprop,init,sample", [(3, 2, [1, 0, 2], None), (4, 1, [2, 0, 1, 0], None),
                                                 (5, 1, [0, 2, 1, 0, 3], None), (4, 2, [0, 1, 0, 2], 200_000),
                                                 (3, 3, [2, 1, 0], 200_000), (5, 2, [3, 0, 1, 2, 0], 200_000)])
def test_pcs_accept_reply_tail_with_checkpoint_slots_enumerated_on_engine(hip_lib, K, nprop, init, sample):
    """PaxosCoordinatorState.main's accept-reply tail (PCS:1173-1213) with the coin AND every vote's
    maxCheckpointedSlot in {-1, 0, slot - 1, slot} enumerated, nodeSlotNumbers starting non-zero: recordSlotNumber
    (PCS:809-825, plain <) and getMedianMinus (PCS:859-875) against tests/pcs_enum_common.py's reading of the
    Java - decided stream, final nodeSlotNumbers and coordinator flag of every pattern."""
    from tests.pcs_enum_common import run_maxcp
    n = run_maxcp(hip_lib, K, nprop, init, sample=sample, seed=K * 10 + nprop)
    assert n == (sample if sample else 8 ** (K * nprop))


@pytest.mark.parametrize("hint", [False, True], ids=["partition path", "sorted-runs hint"])
@pytest.mark.parametrize("K,nprop,G,nv", [(3, 3, 200_000, 24), (5, 4, 100_000, 40), (4, 2, 100_000, 16), (3, 6, 60_000, 60),
                                          (5, 3, 150_000, 9)])
def test_pcs_accept_replies_in_any_order_on_engine(hip_lib, monkeypatch, K, nprop, G, nv, hint):
    """Random accept-reply streams per group (any member, any slot, duplicates, lower / own / higher ballots,
    checkpoint slots) against tests/pcs_enum_common.model_stream - the reading of
    PaxosCoordinator.handleAcceptReply (:210-250) and PCS:597-683, 809-825 - in one batch of up to 4 M votes.
    The batch holds the groups' v-th votes one after the other, i.e. it is nv ascending runs: with the hint
    (GPX_TRY_RUNS=1) the cases of at most 16 votes per group go through k_ar_runs' general replay."""
    from tests.pcs_enum_common import run_streams
    monkeypatch.setenv("GPX_TRY_RUNS", "1" if hint else "0")
    assert run_streams(hip_lib, K, nprop, G, nv, seed=K * 100 + nprop) == G


# (round 5: four of round 4's nine cases - the GPU suite has a wall-clock limit; the dropped ones were further seeds of
# shapes that are still here: K = 3 heavy loss, K = 4, K = 5 with rivals, K = 3 with STOP requests; tests/test_oracle_kat.py
# runs all of them against the oracle on the CPU)
@pytest.mark.parametrize("G,rounds,seed,p_drop,K,p_rival", [(8_000, 30, 13, 0.35, 3, 0.0),
                                                            (6_000, 16, 16, 0.1, 4, 0.0),
                                                            (8_000, 16, 32, 0.2, 5, 0.05), (8_000, 20, 51, 0.1, 3, -0.02)])
def test_whole_round_against_the_two_java_readings_together_on_engine(hip_lib, G, rounds, seed, p_drop, K, p_rival):
    """tests/round_model.py on three HIP engines: the whole round with lost and retransmitted messages against
    the coordinator reading and the acceptor reading of the Java composed."""
    from tests.round_model import run_rounds
    p_stop = 0.02 if seed > 50 else 0.0   # the last two cases: STOP requests among the proposals
    checked, executed = run_rounds(hip_lib, G, rounds, seed, p_drop=p_drop, K=K, p_rival=max(p_rival, 0.0), p_stop=p_stop,
                                   from_disk=seed % 2 == 0)   # odd seeds: GET_ACCEPTED_PVALUES_FROM_DISK = false
    assert checked > G * rounds * 3 and executed > G * rounds // 5
    assert (run_rounds.resigned > G // 10) == (p_rival > 0.0)
    assert (run_rounds.stopped > G // 4 and run_rounds.refused > 0 and run_rounds.stopped_props > G) == (p_stop > 0.0)


@pytest.mark.parametrize("G,rounds,seed,p_drop,K,p_rival,p_stop", [(6000, 12, 81, 0.1, 3, 0.0, 0.0), (4000, 16, 82, 0.15, 5, 0.03, 0.02)])
def test_view_change_after_lossy_rounds_against_java_reading_on_engine(hip_lib, G, rounds, seed, p_drop, K, p_rival, p_stop):
    """tests/round_model.py with failover=True on HIP engines: node 0 dead, replica 1 elected in every group it can
    win - election_begin, prepare, prepare_reply, the view change's ACCEPTs - against the Candidate reading."""
    from tests.round_model import run_rounds
    run_rounds(hip_lib, G, rounds, seed, p_drop=p_drop, K=K, p_rival=p_rival, p_stop=p_stop, from_disk=seed % 2 == 0,
               failover=True, rounds_after=6)     # ... and six more rounds under the new coordinators
    assert run_rounds.after > G * 6
    elected, accepts, carried, noops = run_rounds.failover
    assert elected > G // 5 and carried > G // 8 and accepts == (carried + noops) * (K - 1)


# ---- round 4: the legs staged at the end of round 3 (tests/test_pending_gpu.py then), first run on a GPU in round 4.
# They found two places where the engine left the Java at the int wrap (gap scan's lastKey(), garbageCollectDecisions
# under a median half the int range behind: gpx_kernels.hip.h k_gap_scan / acc_gc) - fixed, now regular cases.

@pytest.mark.parametrize("G,rounds,seed,p_drop,K,p_rival,p_stop,failover", [
    (8_000, 14, 82, 0.15, 3, 0.03, 0.0, False), (8_000, 12, 84, 0.1, 4, 0.0, 0.0, True)])
def test_pause_and_hot_restore_between_rounds(hip_lib, G, rounds, seed, p_drop, K, p_rival, p_stop, failover):
    """tests/test_oracle_kat.py::test_pause_and_hot_restore_between_rounds_against_java_reading on the engine
    (PaxosInstanceStateMachine.java:677-690, 2004-2035; HotRestoreInfo.java:145-157; pokes, repeated PREPARE_REPLYs)"""
    from tests.round_model import run_rounds
    run_rounds(hip_lib, G, rounds, seed, p_drop=p_drop, K=K, p_rival=p_rival, p_stop=p_stop, from_disk=seed % 2 == 0,
               failover=failover, rounds_after=6 if failover else 0, p_pause=0.15, pokes=True, p_dup_reply=0.3)
    assert run_rounds.busy > G and run_rounds.paused > (G if seed % 2 == 0 else 0)


@pytest.mark.parametrize("K,kw", [(1, dict()), (2, dict(p_rival=0.03)), (2, dict(failover=True, rounds_after=6)), (16, dict(p_rival=0.02))])
def test_whole_round_with_unusual_group_sizes(hip_lib, K, kw):
    """tests/test_oracle_kat.py::test_whole_round_with_unusual_group_sizes on the engine: K = 1, 2 and
    PaxosConfig.java:532's MAX_GROUP_SIZE = 16 (WaitforUtility.java:64-68)"""
    from tests.round_model import run_rounds
    G = 3000 if K == 16 else 6000   # (sixteen replicas: the Python reading is what takes the time)
    checked, executed = run_rounds(hip_lib, G, 14, 90 + K, p_drop=0.12, K=K, from_disk=True, p_pause=0.1, pokes=True, **kw)
    assert checked > 60 * G


def test_accept_replies_with_checkpoint_slots_half_the_int_range_apart(hip_lib):
    """recordSlotNumber's plain < (PCS:809-825) under checkpoint slots near INT_MIN / INT_MAX, any vote order"""
    from tests.pcs_enum_common import run_streams
    for K, nprop, G, nv in ((3, 3, 50_000, 24), (5, 4, 25_000, 40), (4, 2, 25_000, 16)):
        assert run_streams(hip_lib, K, nprop, G, nv, seed=K * 100 + nprop + 9, p_extreme=0.1) == G
        for base in (2**31 - 3, 2**31 - 1):      # coordinators whose proposals cross Integer.MAX_VALUE
            assert run_streams(hip_lib, K, nprop, G // 2, nv, seed=K * 100 + nprop + 11, p_extreme=0.05, base=base) == G // 2


@pytest.mark.parametrize("base", [2**31 - 3, -2**31 + 1])
def test_acceptor_side_at_the_int_wrap(hip_lib, base):
    """tests/test_oracle_kat.py::test_acceptor_side_at_the_int_wrap_against_java_reading on the engine, more sequences
    (PaxosAcceptor.java:315, 341, 415-416, 481-489)"""
    import numpy as np
    import tests.acc_enum_common as A
    rng = np.random.default_rng(base % 1000)
    for L, count in ((2, None), (4, 15_000), (8, 10_000)):
        seqs = ([(a, b) for a in A.WIDE for b in A.WIDE[::2]] if count is None else
                [tuple(A.WIDE[i] for i in row) for row in rng.integers(0, len(A.WIDE), (count, L)).tolist()])
        for order, init in (("interleaved", "create"), ("grouped", "initial")):
            A.run_sequences(hip_lib, seqs, init=init, order=order, base=base)


@pytest.mark.parametrize("base,K,kw", [(2**31 - 6, 3, dict()), (2**31 - 20, 3, dict(p_rival=0.03)), (2**31 - 10, 3, dict(p_stop=0.02, from_disk=False)),
                                       (2**31 - 12, 5, dict(p_pause=0.15, pokes=True))])
def test_whole_round_across_the_int_wrap(hip_lib, base, K, kw):
    """tests/test_oracle_kat.py::test_whole_round_across_the_int_wrap on the engine.  Case 0 is the one that found
    garbageCollectDecisions: the first slot's ACCEPT (median 0) retransmitted after the slots wrapped drops the
    committed slot -2^31 + 1 in the Java (0 - key > 0), and so must the engine."""
    from tests.round_model import run_rounds
    kw = dict(kw)
    kw.setdefault("from_disk", True)
    G = 3_000 if K <= 3 else 2_500
    checked, executed = run_rounds(hip_lib, G, 16, 7, p_drop=0.12, K=K, base=base, **kw)
    assert checked > 90 * G
