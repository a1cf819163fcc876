"""The GPU suite at the sizes it had before rounds 4 and 5 trimmed it for wall clock (815 s then; the regular `-m gpu`
suite is the trimmed one) - ADVICE r5: "keep the full matrix under a slow / nightly marker instead of deleting cases".
Skipped unless GPX_FULL_MATRIX=1:
    GPX_FULL_MATRIX=1 python -m pytest tests/test_full_matrix_gpu.py -m gpu -q      (about seven minutes on an MI355X)
Every case is the regular test's body with round 4's parameters: more sequences, both batch orders for the exhaustive
acceptor plans, all nine whole-round cases, three pause cases, the third int-wrap base, larger ordered fuzzes, the 10 M
and 5 M churn at three rounds."""
import os

import pytest

import tests.test_acc_enum_gpu as E
import tests.test_fullsize_gpu as F
import tests.test_one_gpu as O

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(os.environ.get("GPX_FULL_MATRIX") != "1", reason="the untrimmed matrix: GPX_FULL_MATRIX=1")]


def test_acceptor_side_enumerated_both_orders_full_scale(hip_lib):
    import tests.acc_enum_common as A
    for k in A.COVERAGE:
        A.COVERAGE[k] = 0
    n = A.run_plan(hip_lib, scale=1.0)
    n += A.run_plan(hip_lib, scale=0.05, from_disk=(False,))
    assert n > 8_000_000
    assert all(v > 0 for v in A.COVERAGE.values()), A.COVERAGE


def test_acceptor_side_long_random_sequences_120k(hip_lib):
    import tests.acc_enum_common as A
    assert A.run_long_random(hip_lib, 120_000) > 4_000_000


@pytest.mark.parametrize("G,rounds,seed,p_drop,K,p_rival", [(12_000, 20, 12, 0.15, 3, 0.0), (8_000, 30, 13, 0.35, 3, 0.0),
                                                            (20_000, 10, 14, 0.0, 3, 0.0), (8_000, 16, 15, 0.2, 5, 0.0),
                                                            (6_000, 16, 16, 0.1, 4, 0.0), (15_000, 20, 31, 0.1, 3, 0.03),
                                                            (8_000, 16, 32, 0.2, 5, 0.05), (12_000, 24, 51, 0.1, 3, -0.02),
                                                            (8_000, 20, 52, 0.15, 5, 0.03)])
def test_whole_round_nine_cases(hip_lib, G, rounds, seed, p_drop, K, p_rival):
    E.test_whole_round_against_the_two_java_readings_together_on_engine(hip_lib, G, rounds, seed, p_drop, K, p_rival)


@pytest.mark.parametrize("G,rounds,seed,p_drop,K,p_rival,p_stop,failover", [
    (20_000, 14, 82, 0.15, 3, 0.03, 0.0, False), (12_000, 12, 84, 0.1, 4, 0.0, 0.0, True), (12_000, 12, 81, 0.1, 3, 0.0, 0.0, False)])
def test_pause_three_cases(hip_lib, G, rounds, seed, p_drop, K, p_rival, p_stop, failover):
    E.test_pause_and_hot_restore_between_rounds(hip_lib, G, rounds, seed, p_drop, K, p_rival, p_stop, failover)


def test_acceptor_side_at_the_third_int_wrap_base(hip_lib):
    E.test_acceptor_side_at_the_int_wrap(hip_lib, 2**31 - 1)


@pytest.mark.parametrize("seed,G,batch,steps", [(41, 12_000, 100_000, 6), (42, 20_000, 200_000, 4)])
def test_large_ordered_batches_round4_sizes(hip_lib, oracle_lib, seed, G, batch, steps):
    O.test_large_ordered_batches_under_the_promise(hip_lib, oracle_lib, seed, G, batch, steps)


def test_config5_churn_10m_three_rounds(hip_lib, oracle_lib):
    F._churn_across_ranges(hip_lib, oracle_lib, 10_000_000, 3, R=3, seed=11)


def test_config5_churn_5m_k5_three_rounds(hip_lib, oracle_lib):
    F._churn_across_ranges(hip_lib, oracle_lib, 5_000_000, 5, R=3, seed=12)
