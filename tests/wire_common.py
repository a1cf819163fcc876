"""Seeded generators of wire frames (well-formed and damaged) and the differential comparison of
two libraries behind include/gpx_wire.h (the HIP engine and the CPU oracle)."""
import struct

import numpy as np

from gigapaxos_amd import Engine, hri_create, S_OK
from gigapaxos_amd import wire as W


def group_names(G, rng):
    """paxosIDs of mixed length (1..127 bytes), pairwise distinct, ISO-8859-1 incl. bytes >= 0x80."""
    names = []
    for g in range(G):
        base = b"svc%d" % g
        if g % 7 == 3:
            base += bytes(rng.integers(0x80, 0x100, int(rng.integers(1, 100))).astype(np.uint8))
        elif g % 11 == 5:
            base = (base * 30)[:127]
        names.append(base[:127])
    assert len(set(names)) == G
    return names


def make_wire_pair(lib_a, lib_b, G, k, rng, my_id=100, max_batch=1 << 16):
    """Two engines with the same named groups; row g has version g % 3; the last G // 10 rows are
    named but their groups are not created, a few created groups stay unnamed."""
    names = group_names(G, rng)
    members = np.tile(np.arange(100, 100 + k, dtype=np.int32), (G, 1))
    rows = hri_create(G, k, my_id)
    rows["version"] = np.arange(G) % 3
    created = np.arange(G - G // 10, dtype=np.int32)
    named = np.array([g for g in range(G) if g % 13 != 7], np.int32)
    out = []
    for lib in (lib_a, lib_b):
        e = Engine(lib, my_id, G, kmax=k, window=8, max_batch=max_batch)
        we = W.WireEngine(e)
        assert (e.create_groups(created, members[created], k, rows[created]) == S_OK).all()
        st = we.bind([names[g] for g in named], named)
        assert (st == S_OK).all()
        out.append((e, we))
    return out, names


def random_frames(names, n, rng, damage=0.25):
    """A burst of n frames of all four byteified types; `damage` of them altered the way a broken
    or hostile sender could (cut short, length fields poisoned, type ints replaced, slot lists out
    of order, unknown names, stale versions, bytes flipped)."""
    G = len(names)
    frames = []
    for _ in range(n):
        g = int(rng.integers(0, G))
        name = names[g]
        ver = g % 3
        r = rng.random()
        if r < 0.05:
            name = b"ghost%d" % g
        elif r < 0.10:
            ver += 1
        t = rng.choice([W.WT_BATCHED_ACCEPT_REPLY] * 4 + [W.WT_BATCHED_COMMIT] * 3 + [W.WT_ACCEPT] * 2 + [W.WT_REQUEST])
        i32 = lambda: int(rng.integers(-2**31, 2**31))  # noqa: E731
        small = lambda: int(rng.integers(-3, 50))  # noqa: E731
        nsl = int(rng.choice([0, 1, 1, 1, 2, 3, 8, 40]))
        base = i32() if rng.random() < 0.1 else small()
        slots = [(base + j) for j in range(nsl)]
        slots = [((s + 2**31) % 2**32) - 2**31 for s in slots]
        slots = sorted(set(slots))
        if rng.random() < 0.15 and nsl > 1:
            slots = list(rng.permutation(slots + slots[:2]))  # out of order + duplicates
        if t == W.WT_BATCHED_ACCEPT_REPLY:
            f = W.batched_accept_reply(name, ver, int(rng.integers(100, 105)), small() % 3, int(rng.integers(100, 105)),
                                       small(), slots, req_ids=[i32() for _ in slots])
        elif t == W.WT_BATCHED_COMMIT:
            f = W.batched_commit(name, ver, small() % 3, int(rng.integers(100, 105)), small(), slots,
                                 [100, 101, 102][: int(rng.integers(0, 4))])
        else:
            nb = int(rng.choice([0, 0, 0, 1, 3]))
            sub = []
            for j in range(nb):
                inner = []
                if rng.random() < 0.2:
                    inner = [W.request(name, ver, i32(), b"z" * int(rng.integers(0, 9)), stop=rng.random() < 0.3)]
                sub.append(W.request(name, ver, i32(), b"y" * int(rng.integers(0, 20)),
                                     stop=rng.random() < 0.1, batched=inner))
            val = bytes(rng.integers(0, 256, int(rng.integers(0, 80))).astype(np.uint8))
            rid = int(rng.integers(-2**62, 2**62))
            if t == W.WT_ACCEPT:
                f = W.accept(name, ver, rid, small(), small() % 3, int(rng.integers(100, 105)), small(),
                             int(rng.integers(100, 105)), val, stop=rng.random() < 0.05, batched=sub)
            else:
                f = W.request(name, ver, rid, val, stop=rng.random() < 0.05, batched=sub,
                              digest=b"d" * int(rng.choice([0, 0, 20])))
        if rng.random() < damage:
            f = bytearray(f)
            k = rng.integers(0, 6)
            if k == 0 and len(f) > 1:
                f = f[: int(rng.integers(0, len(f)))]
            elif k == 1:
                f[4:8] = struct.pack(">i", int(rng.choice([0, 2, 6, 8, 12, 34, 35, 36, 90, 9999, 77])))
            elif k == 2:
                f[0:4] = struct.pack(">i", int(rng.choice([89, 91, 0])))
            elif k == 3 and len(f) > 20:
                p = int(rng.integers(12, len(f) - 4))
                f[p:p + 4] = struct.pack(">i", int(rng.choice([-1, -2**31, 2**31 - 1, 2**20, 0])))
            elif k == 4 and len(f) > 0:
                p = int(rng.integers(0, len(f)))
                f[p] ^= 1 << int(rng.integers(0, 8))
            else:
                f += bytes(rng.integers(0, 256, int(rng.integers(1, 9))).astype(np.uint8))
            f = bytes(f)
        frames.append(f)
    return frames


def assert_same_decode(da, db, tag=""):
    assert da.counts == db.counts, f"{tag} counts {da.counts} {db.counts}"
    assert da.f_status.tolist() == db.f_status.tolist(), f"{tag} f_status"
    assert da.f_gidx.tolist() == db.f_gidx.tolist(), f"{tag} f_gidx"
    assert da.f_type.tolist() == db.f_type.tolist(), f"{tag} f_type"
    for cls in ("votes", "commits", "accepts", "requests"):
        ca, cb = getattr(da, cls), getattr(db, cls)
        for k in ca:
            assert ca[k].shape == cb[k].shape and (ca[k] == cb[k]).all(), f"{tag} {cls}.{k}"
