import os
import subprocess
import sys

# Pageable host <-> device copies of this process go through the runtime's own pinned staging buffer, never by pinning the
# caller's pages for the length of the copy (the HIP runtime's choice above ~1 MB: scripts/probe_copy_path.py,
# profiles/r06_copy_path_probe.txt).  Every GPU page fault that ever killed a run of this suite (one run in each of rounds
# 3 and 5, two in round 6) found the main thread inside exactly that transient pinning
# (hsaCopyStagedOrPinned -> addPinnedMem), three times under one of torch's copies and once under gpx_group_create's.  The
# flag is read when the runtime initialises, so it is set before anything imports torch; the results of no test depend on it.
os.environ.setdefault("GPU_PINNED_MIN_XFER_SIZE", "1048576")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_fast: the subset of the GPU suite to run after every kernel change (about a minute "
                                       "and a half: one case per kernel family at full size, one fuzz seed each, the "
                                       "enumeration samples); the whole -m gpu suite takes a quarter of an hour")


# node-id fragments of the gpu_fast subset (every GPU test file contributes; chosen by measured duration, round 4:
# profiles/r04_gpu_fast_durations.txt)
GPU_FAST = [
    "test_fullsize_gpu.py::test_config3_config4_streams_1m_groups_vs_oracle[3-True]",      # partition pipeline, 1 M groups, mix
    "test_fullsize_gpu.py::test_config3_config4_streams_1m_groups_vs_oracle[5-True]",      # ... K = 5
    "test_runs_gpu.py::test_reply_runs_1m_groups_vs_oracle[3-8]",                          # sorted runs, promise
    "test_runs_gpu.py::test_reply_runs_fuzz[5-5000-42",                                    # ... hint, fuzz
    "test_one_gpu.py::test_lazy_outputs_on_the_device_path",                               # k_ac_one + lazy compaction
    "test_one_gpu.py::test_lazy_reply_runs[3-150000-True]", "test_one_gpu.py::test_lazy_reply_runs[5-9000-True]",
    "test_one_gpu.py::test_lazy_reply_runs[3-20000-False]", "test_one_gpu.py::test_lazy_outputs_on_the_device_path[20000-False]",
    "test_runs_gpu.py::test_reply_runs_fuzz[3-700-41",
    "test_one_gpu.py::test_broken_promise_refuses_from_the_first_violation[descent]",
    "test_one_gpu.py::test_broken_promise_refuses_from_the_first_violation[repeated group]",
    "test_parity_gpu.py::test_fuzz_mixed_ops[partition path-3-1]", "test_parity_gpu.py::test_fuzz_mixed_ops[sorted-runs hint-16-4]",
    "test_parity_gpu.py::test_fuzz_ordered_batches[partition path-5-700-22]",
    "test_parity_gpu.py::test_ordered_batches_promise", "test_parity_gpu.py::test_fuzz_wraparound[partition path]",
    "test_parity_gpu.py::test_steady_state_votes_fast_path[partition path-5-53]", "test_parity_gpu.py::test_hot_group_long_segments[partition path-4096]",
    "test_parity_gpu.py::test_config2_10k_groups_full_pipeline", "test_parity_gpu.py::test_propose_batch_orders[partition path",
    "test_edges_gpu.py::test_empty_batches_and_capacity", "test_edges_gpu.py::test_tile_boundary_batch_sizes[4097]",
    "test_edges_gpu.py::test_tile_boundary_batch_sizes[12288]", "test_edges_gpu.py::test_unaligned_device_columns",
    "test_wire_gpu.py::test_decode_fuzz[512-frame tiles-2-0.3]", "test_wire_gpu.py::test_wire_codec_against_java_reading[256-frame tiles]",
    "test_acc_enum_gpu.py::test_acceptor_side_enumerated_under_the_ordered_promise",
    "test_acc_enum_gpu.py::test_pcs_accept_replies_in_any_order_on_engine[4-2-100000-16-False]",
    "test_acc_enum_gpu.py::test_whole_round_against_the_two_java_readings_together_on_engine[6000-16-16-0.1-4-0.0]",
    "test_acc_enum_gpu.py::test_whole_round_with_unusual_group_sizes[1-kw0]",
    "test_election_gpu.py::test_election_fuzz_parity[1-", "test_election_gpu.py::test_failover_end_to_end_parity[200-0-8]",
    "test_host_rows_gpu.py::test_gap_detection_matches_oracle[0]", "test_host_rows_gpu.py::test_request_batcher_matches_oracle[1-True]",
    "test_route_gpu.py::test_route_matches_shard_map[8-300000-100000]",
    "test_async_gpu.py::test_async_rounds_match_oracle[300000-3]",
    "test_host_cluster_gpu.py::test_cluster_matches_oracle_build[0]",
    "test_small_ar_gpu.py::test_small_calls_vs_oracle[30000-3-1024-True]",                 # k_ar_tiny
    "test_small_ar_gpu.py::test_small_calls_vs_oracle[1000000-5-512-True]", "test_small_ar_gpu.py::test_small_calls_vs_oracle[7-3-1024-True]",
    "test_small_ar_gpu.py::test_skewed_small_calls[1024-",                                 # ... hot groups: a lane's own replay
    "test_small_ar_gpu.py::test_small_call_fuzz[5-3000-2]", "test_small_ar_gpu.py::test_runs_hint_and_small_calls",
]


def pytest_collection_modifyitems(config, items):
    for it in items:
        if any(frag in it.nodeid for frag in GPU_FAST):
            it.add_marker(pytest.mark.gpu_fast)


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (checker only), built on demand from oracle/gpx_oracle.cpp."""
    from tests.oracle_binding import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def hip_lib():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.zeros(1, device="cuda:0")  # wake the device before the HIP library's own runtime looks for it
    torch.cuda.synchronize()
    from gigapaxos_amd import load_hip

    return load_hip()
