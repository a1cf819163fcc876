"""Builds and runs the C++ host layer's in-process cluster (gigapaxos_amd/host) - against
libgpx_hip.so (the product build, made by __graft_entry__.build) or, for the CPU checks of the host
logic, against the oracle (symbols renamed by tests/host_oracle_prefix.h).  Test infrastructure."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "gigapaxos_amd", "host")
HIP_BIN = os.path.join(HOST, "gpx_loopback_cluster")
ORC_BIN = os.path.join(ROOT, "oracle", "_host_cluster_oracle")

CASES = [
    # config #1's shape: 3 replicas, 1 group, 10,000 requests, any entry replica
    ["--groups", "1", "--rounds", "10000"],
    # config #2's shape: 10 k groups, full pipeline over frames
    ["--groups", "10000", "--rounds", "4", "--seed", "7"],
    ["--groups", "500", "--rounds", "12", "--nodes", "5", "--seed", "3"],
    ["--groups", "300", "--rounds", "6", "--stop-last", "--entry", "coordinator"],
    ["--groups", "64", "--rounds", "9", "--value-bytes", "1500", "--seed", "11"],
    # coordinator failover: a node dies with ACCEPTs in flight, the next in line is elected everywhere
    ["--groups", "300", "--rounds", "8", "--kill-round", "3"],
    ["--groups", "2000", "--rounds", "6", "--kill-round", "2", "--kill-node", "1", "--seed", "5"],
    ["--groups", "400", "--rounds", "7", "--nodes", "5", "--kill-round", "4", "--kill-node", "4", "--seed", "9"],
    # RequestBatcher: bursts of requests per group, latched into one proposal where they meet
    ["--groups", "100", "--rounds", "4", "--burst", "6"],
    ["--groups", "50", "--rounds", "3", "--burst", "40", "--nodes", "5", "--seed", "4"],
    ["--groups", "200", "--rounds", "6", "--burst", "3", "--kill-round", "2"],
    ["--groups", "100", "--rounds", "3", "--burst", "4", "--no-batching"],
    # device tables smaller than the number of groups: pause / unpause by HotRestoreInfo on demand
    ["--groups", "600", "--rounds", "30", "--capacity", "128", "--active", "40"],
    ["--groups", "500", "--rounds", "25", "--capacity", "96", "--active", "24", "--nodes", "5", "--burst", "2",
     "--seed", "6"],
    # a lossy network for BATCHED_COMMITs: gap detection (gpx_gap_scan) and decision sync
    ["--groups", "200", "--rounds", "12", "--drop-commits", "100"],
    ["--groups", "1000", "--rounds", "10", "--drop-commits", "250", "--nodes", "5", "--seed", "8"],
    # lost ACCEPTs / accept replies: the retransmission timers (gpx_poke_scan) and the forced sync
    ["--groups", "200", "--rounds", "10", "--drop-accepts", "150"],
    ["--groups", "300", "--rounds", "8", "--drop-accepts", "100", "--drop-commits", "100", "--nodes", "5", "--seed", "2"],
    # a burst longer than the engine's proposal window, unbatched, on a network that loses ACCEPTs and replies:
    # window-refused requests are retried, and a retry that is refused again must not count as progress - the
    # retransmission timers (which only fire when there is no work) are what frees the window
    ["--groups", "20", "--rounds", "3", "--burst", "24", "--no-batching", "--drop-accepts", "200", "--seed", "13"],
    # logging on: accept replies wait for their batch's log write (durable three polls later)
    ["--groups", "200", "--rounds", "6", "--log-delay", "3"],
    ["--groups", "150", "--rounds", "6", "--log-delay", "5", "--kill-round", "3", "--burst", "2", "--seed", "12"],
    # a node dies on a lossy network: the candidate must hear of slots a survivor has already executed
    ["--groups", "7", "--rounds", "6", "--seed", "577593", "--kill-round", "1", "--kill-node", "0", "--drop-commits", "200",
     "--drop-accepts", "150"],
    ["--groups", "60", "--rounds", "6", "--seed", "2504", "--kill-round", "1", "--kill-node", "1", "--drop-commits", "200",
     "--drop-accepts", "150", "--no-batching"],
]


# on the GPU every engine call of the host-pointer API is a synchronous round trip: fewer rounds
CASES_GPU = [["--groups", "1", "--rounds", "1000"]] + CASES[1:]


def build_oracle_cluster():
    from tests.oracle_binding import build_oracle

    build_oracle()
    srcs = [os.path.join(HOST, f) for f in ("gpx_host.cpp", "loopback_cluster.cpp")]
    deps = srcs + [os.path.join(HOST, "gpx_host.hpp"), os.path.join(ROOT, "tests", "host_oracle_prefix.h"),
                   os.path.join(ROOT, "oracle", "libgpx_oracle.so")]
    if not os.path.exists(ORC_BIN) or os.path.getmtime(ORC_BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                               "-include", os.path.join(ROOT, "tests", "host_oracle_prefix.h"), "-o", ORC_BIN]
                              + srcs + ["-L", os.path.join(ROOT, "oracle"), "-lgpx_oracle", "-Wl,-rpath,$ORIGIN"])
    return ORC_BIN


def build_hip_cluster():
    """The product build of the cluster example (normally made by __graft_entry__.build()); rebuilt here
    if it is missing or older than its sources.  Needs libgpx_hip.so (hipcc's output) to be there."""
    srcs = [os.path.join(HOST, f) for f in ("gpx_host.cpp", "loopback_cluster.cpp")]
    lib = os.path.join(ROOT, "gigapaxos_amd", "csrc", "libgpx_hip.so")
    deps = srcs + [os.path.join(HOST, "gpx_host.hpp"), lib]
    if not os.path.exists(HIP_BIN) or os.path.getmtime(HIP_BIN) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-I", os.path.join(ROOT, "include"),
                               "-o", HIP_BIN] + srcs +
                              ["-L", os.path.dirname(lib), "-lgpx_hip", "-Wl,-rpath,$ORIGIN/../csrc",
                               "-Wl,-rpath-link,/opt/rocm/lib"])
    return HIP_BIN


def run_cluster(binary, args, timeout=300):
    p = subprocess.run([binary] + list(args), capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, (p.returncode, p.stdout[-400:], p.stderr[-400:])
    return json.loads(p.stdout.strip().splitlines()[-1])
