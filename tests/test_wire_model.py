"""The wire codec's oracle against tests/wire_model.py - the reference's byte constructors, demux, batcher and toBytes
methods read from the Java a second time, independently of oracle/gpx_wire_oracle.inc.  CPU only.  (Kept out of
tests/test_wire_oracle.py, whose scenarios tests/test_wire_gpu.py replays on the HIP library one to one: these
take parameters, and their GPU leg waits for a GPU visit that can confirm it.)"""
import struct

import numpy as np
import pytest

from gigapaxos_amd import D_DECISION, D_PREEMPTED
from gigapaxos_amd import wire as W
from tests.test_wire_oracle import _decisions


@pytest.mark.parametrize("G,n,seed,damage", [(300, 20_000, 11, 0.25), (2000, 30_000, 12, 0.5), (50, 20_000, 13, 0.9)])
def test_decode_of_damaged_bursts_against_java_reading(oracle_lib, G, n, seed, damage):
    """Bursts of all four byteified packet types, a quarter to nine tenths of them damaged (cut short, length
    fields poisoned, type ints replaced, slot lists out of order, ghosts, stale versions, flipped bits, trailing
    bytes): every frame's status / row / type and every decoded record against tests/wire_model.py - the byte
    constructors and the demux read from the Java on their own, not from the oracle."""
    from tests.wire_common import make_wire_pair, random_frames
    from tests.wire_model import check_decode
    rng = np.random.default_rng(seed)
    (pair, names) = make_wire_pair(oracle_lib, oracle_lib, G, 3, rng)
    (e, we), (e2, _) = pair
    e2.close()
    # the instances that exist (make_wire_pair: the last tenth of the rows is named but not created, every 13th
    # created row has no name; row g has version g % 3)
    instances = {names[g]: (g, g % 3) for g in range(G - G // 10) if g % 13 != 7}
    hist = [0] * 5
    for burst in range(4):
        frames = random_frames(names, n // 4, rng, damage=damage)
        for i in np.nonzero(rng.random(len(frames)) < 0.1)[0].tolist():
            # aimed at the length fields the constructors allocate from: the paxosID length byte (0: null paxosID,
            # >= 128: a negative byte), and whatever int sits where a well-formed frame of this name has its first
            # length / count field
            f = bytearray(frames[i])
            if len(f) > 13:
                if rng.random() < 0.25:
                    # every int PaxosPacketType knows (only four of them have a byte constructor) and their neighbours
                    f[4:8] = struct.pack(">i", int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 13, 21, 23, 32, 33, 34, 35, 36, 37, 90, 9999,
                                                                0, 10, 12, 14, 20, 22, 24, 31, 38, 89, 91, 9998, -1])))
                elif rng.random() < 0.5:
                    f[12] = int(rng.choice([0, 1, 127, 128, 200, 255, max(f[12] - 1, 0), min(f[12] + 1, 255)]))
                else:
                    p0 = 13 + f[12] + int(rng.choice([12, 16, 28, 33, 44, 48]))
                    if p0 + 4 <= len(f):
                        f[p0:p0 + 4] = struct.pack(">i", int(rng.choice([-1, 0, 1, 2, 1025, 2**31 - 1, -2**31])))
                frames[i] = bytes(f)
        for m in (1000, 1024, 1025, 1500):       # long slot lists: ascending, and not (the engine's own limit)
            asc = list(range(5, 5 + m))
            mixed = list(rng.permutation(asc + asc[:3]))[:m]
            for sl in (asc, mixed):
                frames.append(W.batched_commit(names[1], 1, 0, 100, 3, sl, [100, 101, 102]))
                frames.append(W.batched_accept_reply(names[2], 2, 101, 0, 100, 4, sl, req_ids=[7] * len(sl)))
        h = check_decode(we, frames, instances, f"burst {burst}")
        hist = [a + b for a, b in zip(hist, h)]
    e.close()
    assert min(hist) > n // 400, hist        # every status occurs, many times over


@pytest.mark.parametrize("G,seed", [(400, 21), (3000, 22)])
def test_pack_of_random_batches_against_java_reading(oracle_lib, G, seed):
    """gpx_wire_pack_commits / gpx_wire_pack_accept_replies over random batches - several ballots per group,
    repeated slots, medians around the int wrap, PREEMPTED rows, NACKs, dropped and refused ACCEPTs, rows without a
    name or an instance - byte for byte against the batcher and the toBytes methods as tests/wire_model.py reads
    them from the Java."""
    from tests.wire_common import make_wire_pair
    from tests import wire_model as M
    rng = np.random.default_rng(seed)
    k, my_id = 3, 101
    (pair, names) = make_wire_pair(oracle_lib, oracle_lib, G, k, rng, my_id=my_id)
    (e, we), (e2, _) = pair
    e2.close()
    info = {g: (names[g], g % 3, [100, 101, 102]) for g in range(G - G // 10) if g % 13 != 7}
    n_frames = 0
    for rnd in range(6):
        # decisions: group-major (as gpx_accept_reply_batch emits them), a group's rows in any slot / ballot order
        rows = []
        for g in sorted(rng.choice(G, G // 2, replace=False).tolist()):
            base = int(rng.choice([0, 5, 2**31 - 3, -2**31 + 2, -4]))
            for _ in range(int(rng.choice([1, 1, 2, 3, 8, 40]))):
                rows.append((g, int(rng.integers(-3, 30)), int(rng.integers(0, 3)), int(rng.choice([100, 101, 102])),
                             M._i32(base + int(rng.integers(0, 7))), D_DECISION if rng.random() < 0.85 else D_PREEMPTED))
        frames, fg, nbytes = we.pack_commits(_decisions(rows))
        want, want_g = M.pack_commits(rows, info, my_id, D_DECISION)
        assert fg.tolist() == want_g and frames == want, f"round {rnd}: BATCHED_COMMIT frames"
        assert nbytes == sum((len(f) + 3) // 4 * 4 for f in frames)
        n_frames += len(frames)
        # accept replies: in the order of the ACCEPT batch (any), status 0 = accepted or NACKed, 1 / 2 = no reply
        rows = []
        for g in rng.choice(G, G // 2, replace=False).tolist():
            for _ in range(int(rng.choice([1, 1, 2, 3, 8, 40]))):
                sender = int(rng.choice([100, 102]))
                bcoord = sender if rng.random() < 0.85 else int(rng.choice([100, 101, 102]))
                rows.append((g if rng.random() < 0.97 else -1, int(rng.integers(-3, 30)), int(rng.integers(0, 2)), bcoord,
                             int(rng.integers(-1, 9)), int(rng.choice([0] * 8 + [1, 2])), sender, int(rng.integers(-2**62, 2**62))))
        rows = [rows[i] for i in rng.permutation(len(rows))]
        a = np.array(rows, np.int64)
        frames, fg, fd, ub, nbytes = we.pack_accept_replies(a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5].astype(np.uint8),
                                                            sender=a[:, 6], req_id=a[:, 7])
        want, want_g, want_d, want_ub = M.pack_accept_replies(rows, info, my_id)
        assert fg.tolist() == want_g and fd.tolist() == want_d and frames == want, f"round {rnd}: BATCHED_ACCEPT_REPLY frames"
        assert ub.tolist() == want_ub, f"round {rnd}: replies that leave unbatched"
        n_frames += len(frames)
    e.close()
    assert n_frames > 3 * G


def plan_send_reading(est, key, max_payload, min_batch, across):
    """PaxosPacketBatcher.dequeueImpl (PaxosPacketBatcher.java:182-209: four `while (lengthEstimate < MAX && next !=
    null)` loops over the accept replies, commits, accepts and requests share one running estimate = one loop over the
    frames in that order, the test before the add) and process() -> batch() (:268-303: `tasks.length >
    MIN_PP_BATCH_SIZE` tasks are regrouped in a LinkedHashMap keyed by the recipient set)"""
    n = len(est)
    burst, env, pos = [0] * n, [-1] * n, [0] * n
    i = b = 0
    while i < n:
        length, tasks = 0, []
        while i < n and length < max_payload:
            tasks.append(i)
            length += int(est[i])
            i += 1
        if across and len(tasks) > min_batch:
            grouped = {}
            for f in tasks:
                grouped.setdefault(int(key[f]), []).append(f)
            for e_idx, fs in enumerate(grouped.values()):
                for p_idx, f in enumerate(fs):
                    env[f], pos[f] = e_idx, p_idx
        for f in tasks:
            burst[f] = b
        b += 1
    return burst, env, pos, b


def test_plan_send_against_java_reading(oracle_lib):
    """gpx_wire_plan_send of the oracle AND of the engine library (pure host code: it runs here) against the reading"""
    import __graft_entry__ as ge
    from gigapaxos_amd._abi import GpxLib
    ge.build()
    rng = np.random.default_rng(5)
    for lib in (oracle_lib, GpxLib(ge.HIP_SO, "gpx_", device_api=True)):
        for _ in range(300):
            n = int(rng.integers(0, 600))
            est = rng.integers(1, int(rng.choice([3, 300, 30_000])), n)
            key = rng.integers(0, int(rng.choice([1, 3, 50])), n)
            mp, mb, across = int(rng.choice([1, 50, 5000, 4 << 20])), int(rng.integers(0, 6)), bool(rng.random() < 0.8)
            burst, env, pos, nb = W.plan_send(lib, est, key, mp, mb, across)
            want = plan_send_reading(est, key, mp, mb, across)
            assert (burst.tolist(), env.tolist(), pos.tolist(), nb) == want, (n, mp, mb, across)
