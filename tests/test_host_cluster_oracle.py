"""The C++ host layer (gpx::PaxosManager over the C-ABI) on the CPU: its in-process cluster built
against the oracle.  Checks the host logic - frame building, forwarding to the coordinator,
loopback short circuits, value bookkeeping, in-order upcalls: every replica executes every request
exactly once, slot == sequence number (TESTPaxosApp's invariant), identical hash chains."""
import pytest

from tests.host_cluster_common import CASES, build_oracle_cluster, run_cluster


@pytest.mark.parametrize("case", range(len(CASES)))
def test_cluster_invariants(case):
    out = run_cluster(build_oracle_cluster(), CASES[case])
    assert out["ok"] is True
    if "--kill-round" in CASES[case]:
        # the survivors agree (checked by the program: equal hash chains, slot == sequence number); the
        # next in line ran in every group the dead node coordinated, won, carried accepted values
        # over and had taken client requests while not yet active
        alive = [n for n in out["per_node"] if n["alive"]]
        assert len(alive) == out["nodes"] - 1
        assert len({n["executed"] for n in alive}) == 1
        assert sum(n["elections_started"] for n in alive) == sum(n["elections_won"] for n in alive) > 0
        assert sum(n["carried_over"] for n in alive) > 0 and sum(n["preactive"] for n in alive) > 0
        assert out["executed_per_node"] >= out["groups"] * (out["rounds"] - 1)
        return
    args = CASES[case]
    if "--capacity" in args:
        cap = int(args[args.index("--capacity") + 1])
        for n in out["per_node"]:
            assert n["pauses"] >= out["groups"] - cap and n["unpauses"] > 0
            assert n["paused_now"] >= out["groups"] - cap
    if "--log-delay" in args:
        for n in out["per_node"]:
            if n["alive"]:
                assert n["logged_accepts"] > 0 and n["held_replies"] > 0 and n["log_batches"] > 0
    if "--drop-accepts" in args:
        assert out["frames_lost"] > 0 and sum(n["accepts_resent"] for n in out["per_node"]) > 0
    if "--drop-commits" in args:
        assert out["frames_lost"] > 0
        assert sum(n["sync_requests"] for n in out["per_node"]) > 0
        assert sum(n["sync_decisions_applied"] for n in out["per_node"]) > 0
    burst = int(args[args.index("--burst") + 1]) if "--burst" in args else 1
    per_round = int(args[args.index("--active") + 1]) if "--active" in args else out["groups"]
    # bursts longer than the engine's window on a lossy network end with one catch-up request per group
    # (loopback_cluster.cpp: a lagging replica can lose both the ACCEPT and the commit of a group's last slot)
    extra = out["groups"] if burst > 8 and ("--drop-accepts" in args or "--drop-commits" in args) else 0
    assert out["executed_per_node"] == out["requests"] == per_round * out["rounds"] * burst + extra
    assert out["client_acks"] == out["requests"]  # every entry replica answered its clients
    assert all(n["checkpoints"] > 0 for n in out["per_node"]) or out["rounds"] * burst < 4 or "--active" in args
    for n in out["per_node"]:
        assert n["executed"] == out["requests"] and n["dropped_frames"] == 0 and n["refused"] == 0
    proposed = sum(n["proposed"] for n in out["per_node"])
    assert sum(n["decisions"] for n in out["per_node"]) == proposed
    if burst > 1 and "--no-batching" not in args:
        # several requests per proposal: fewer proposals than requests, none lost
        assert proposed < out["requests"] and sum(n["batched_requests"] for n in out["per_node"]) > 0
    else:
        assert proposed == out["requests"]
    if "--entry" not in CASES[case]:
        assert sum(n["forwarded"] for n in out["per_node"]) > 0 or out["groups"] == 1


def test_cluster_under_sanitizers(tmp_path):
    """The host layer and the oracle compiled together with AddressSanitizer + UBSan: a failover run
    and a batching run must finish clean (no report on stderr)."""
    import os
    import subprocess

    from tests.host_cluster_common import HOST, ROOT

    exe = str(tmp_path / "cluster_san")
    subprocess.check_call(["g++", "-O0", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-std=c++17",
                           "-I", os.path.join(ROOT, "include"), "-include",
                           os.path.join(ROOT, "tests", "host_oracle_prefix.h"), "-o", exe,
                           os.path.join(HOST, "gpx_host.cpp"), os.path.join(HOST, "loopback_cluster.cpp"),
                           os.path.join(ROOT, "oracle", "gpx_oracle.cpp")])
    for args in (["--groups", "300", "--rounds", "8", "--kill-round", "3"],
                 ["--groups", "50", "--rounds", "3", "--burst", "20", "--nodes", "5"],
                 ["--groups", "100", "--rounds", "4", "--log-file", str(tmp_path / "log")]):
        p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stderr[-600:]
        assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-600:]


def test_cpp_frame_builders_match_the_python_restatement():
    """gpx::makeRequestFrame / makeAcceptFrame / latchToBatch against gigapaxos_amd.wire's builders of
    RequestPacket.toBytes / AcceptPacket.toBytes (written independently), String.hashCode and
    roundRobinCoordinator against the oracle's."""
    import json
    import subprocess

    from gigapaxos_amd import wire as W
    from tests.oracle_binding import load_oracle

    out = json.loads(subprocess.run([build_oracle_cluster(), "--dump-frames"], capture_output=True, text=True,
                                    check=True).stdout)
    rq = W.request(b"TESTPaxosApp7", 0, 0x1122334455667788, b"hello-value", entry_replica=101)
    st = W.request(b"g", 3, -5, b"", stop=True, entry_replica=100)
    assert out["request"] == rq.hex() and out["stop"] == st.hex()
    import struct
    acc = W.request(b"TESTPaxosApp7", 0, 0x1122334455667788, b"hello-value", entry_replica=101, ptype=W.WT_ACCEPT)
    acc += struct.pack(">iiibibi", 42, 2, 101, 0, 40, 0, 101)  # AcceptPacket.toBytes's tail
    assert out["accept"] == acc.hex()
    assert out["batched"] == W.request(b"TESTPaxosApp7", 0, 0x1122334455667788, b"hello-value", entry_replica=101,
                                       batched=[st, rq]).hex()
    assert out["batch_size"] == 2 and out["parsed"] == 3
    # RequestPacket.main: "asd999" + 25 latched stop requests survive bytes -> packets; the device-side
    # walk of the same bytes (oracle restatement here) sees ONE proposal that is a stop
    assert out["rp_main_roundtrip"] is True
    subs = [W.request(b"pid", 0, 1000 + i, b"asd%d" % i, stop=True, entry_replica=100) for i in range(25)]
    assert out["rp_main"] == W.request(b"pid", 0, 999, b"asd999", stop=True, entry_replica=100, batched=subs).hex()
    from gigapaxos_amd import Engine, hri_create, S_OK
    e = Engine(load_oracle(), 100, 4, kmax=3, window=8, max_batch=64)
    we = W.WireEngine(e)
    import numpy as np
    assert (e.create_groups(np.arange(1), np.array([[100, 101, 102]], np.int32), 3, hri_create(1, 3, 100)) == S_OK).all()
    assert (we.bind([b"pid"], [0]) == S_OK).all()
    d = we.decode([bytes.fromhex(out["rp_main"])])
    assert d.f_status.tolist() == [W.W_OK] and d.requests["is_stop"].tolist() == [1] and d.requests["req_id"].tolist() == [999]
    e.close()
    assert out["hash"] == 99162322  # "hello".hashCode()
    import numpy as np
    import ctypes as C
    mem = np.array([100, 101, 102], np.int32)
    assert out["coordinator"] == load_oracle().lib.orc_round_robin_coordinator(b"TESTPaxosApp7", mem.ctypes.data_as(C.c_void_p), 3, 0)


def test_file_logger_same_outcome_and_durable_log(tmp_path):
    """Logging to fdatasync'ed files written by their own threads: the replicas end in the same state
    as with logging off (only the timing of replies changes), and every node's log holds the ACCEPTs
    it had to make durable, length-prefixed."""
    import struct

    exe = build_oracle_cluster()
    base = ["--groups", "120", "--rounds", "5", "--seed", "4"]
    plain = run_cluster(exe, base)
    logged = run_cluster(exe, base + ["--log-file", str(tmp_path / "wal")])
    assert logged["ok"] is True and logged["state_digest"] == plain["state_digest"]
    for n in logged["per_node"]:
        data = (tmp_path / ("wal.%d" % n["id"])).read_bytes()
        cnt, p = 0, 0
        while p < len(data):
            (ln,) = struct.unpack_from(">I", data, p)
            assert data[p + 4:p + 12] == struct.pack(">ii", 90, 3)  # a PAXOS_PACKET of type ACCEPT
            p += 4 + ln
            cnt += 1
        assert p == len(data) and cnt == n["logged_accepts"] == n["accepts"]


def test_cluster_flag_fuzz():
    """scripts/cluster_fuzz.py with a fixed seed: 150 random scenario mixes, all must converge."""
    import os
    import subprocess
    import sys

    from tests.host_cluster_common import ROOT

    p = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "cluster_fuzz.py"), "11", "150"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-800:]
