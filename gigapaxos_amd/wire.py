"""Host side of the wire codec (include/gpx_wire.h): ctypes binding of the names / decode /
pack entry points and byte-level builders of the four byteified packet types.

The builders restate the reference's ``toBytes()`` methods (big-endian java.nio.ByteBuffer
layouts) so that tests, the loopback cluster and bench.py can produce the frames a gigapaxos
node would put on the wire:

* PaxosPacket header            paxospackets/PaxosPacket.java:461-476
* RequestPacket.toBytes         paxospackets/RequestPacket.java:819-948
* AcceptPacket.toBytes          paxospackets/AcceptPacket.java:95-135
* BatchedAcceptReply.toBytes    paxospackets/BatchedAcceptReply.java:119-173
* BatchedCommit.toBytes         paxospackets/BatchedCommit.java:184-215

The compute path (frames -> SoA, SoA -> frames) is the HIP library; nothing here parses a frame.
"""
from __future__ import annotations

import ctypes as C
import struct
from dataclasses import dataclass

import numpy as np

from ._abi import Engine, GpxLib, _p, _i32

WT_PAXOS_PACKET, WT_REQUEST, WT_ACCEPT, WT_BATCHED_ACCEPT_REPLY, WT_BATCHED_COMMIT = 90, 1, 3, 34, 35
W_OK, W_NOGROUP, W_VERSION, W_MALFORMED, W_UNSUPPORTED, W_CAPACITY = range(6)
W_MAX_UNSORTED = 1024

_VP = C.c_void_p


class WireVotes(C.Structure):
    _fields_ = [("cap", C.c_int32)] + [(n, _VP) for n in
                                       ("gidx", "bnum", "bcoord", "slot", "acceptor", "max_cp", "frame")]


class WireCommits(C.Structure):
    _fields_ = [("cap", C.c_int32)] + [(n, _VP) for n in
                                       ("gidx", "bnum", "bcoord", "slot", "median_cp", "kind", "frame")]


class WireAccepts(C.Structure):
    _fields_ = [("cap", C.c_int32)] + [(n, _VP) for n in
                                       ("gidx", "bnum", "bcoord", "slot", "median_cp", "flags", "sender",
                                        "req_id", "frame")]


class WireRequests(C.Structure):
    _fields_ = [("cap", C.c_int32)] + [(n, _VP) for n in ("gidx", "is_stop", "req_id", "frame")]


class WireCounts(C.Structure):
    _fields_ = [("n_votes", C.c_int32), ("n_commits", C.c_int32), ("n_accepts", C.c_int32),
                ("n_requests", C.c_int32), ("n_bad_frames", C.c_int32), ("reserved", C.c_int32 * 3)]


_WIRE_SIGS = {
    "names_bind": [C.c_int32, _VP, _VP, _VP, _VP],
    "names_unbind": [C.c_int32, _VP, _VP],
    "names_lookup": [C.c_int32, _VP, _VP, _VP],
    "rows_alloc": [C.c_int32, _VP],
    "rows_free": [C.c_int32, _VP],
    "wire_decode": [C.c_int32, _VP, _VP, _VP, _VP, _VP, C.POINTER(WireVotes), C.POINTER(WireCommits),
                    C.POINTER(WireAccepts), C.POINTER(WireRequests), _VP],
    "wire_pack_commits": [C.c_int32] + [_VP] * 7 + [C.c_int64] + [_VP] * 5,
    "wire_pack_accept_replies": [C.c_int32] + [_VP] * 10 + [C.c_int64] + [_VP] * 6,
}
_WIRE_DEV_SIGS = {
    "wire_decode_dev": _WIRE_SIGS["wire_decode"],
    "wire_pack_commits_dev": [C.c_int32, _VP] + [_VP] * 7 + [C.c_int64] + [_VP] * 5,
    "wire_pack_accept_replies_dev": [C.c_int32] + [_VP] * 10 + [C.c_int64] + [_VP] * 6,
}
WIRE_EXPORTED_SYMBOLS = list(_WIRE_SIGS) + list(_WIRE_DEV_SIGS) + ["wire_plan_send"]


def plan_send(lib: GpxLib, est, dest_key, max_payload=4 * 1024 * 1024, min_batch=3, batch_across_groups=True):
    """gpx_wire_plan_send: PaxosPacketBatcher.dequeueImpl's payload bound + process() -> batch() over a
    list of outgoing frames; returns (burst, envelope, position, n_bursts)."""
    est = np.ascontiguousarray(est, np.int64)
    dest_key = np.ascontiguousarray(dest_key, np.int64)
    n = est.shape[0]
    assert dest_key.shape[0] == n
    burst, env, pos = (np.zeros(max(n, 1), np.int32) for _ in range(3))
    nb = np.zeros(1, np.int32)
    f = getattr(lib.lib, lib.prefix + "wire_plan_send")
    f.restype = C.c_int
    f.argtypes = [C.c_int32, _VP, _VP, C.c_int64, C.c_int32, C.c_int32, _VP, _VP, _VP, _VP]
    lib.check(f(n, est.ctypes.data, dest_key.ctypes.data, int(max_payload), int(min_batch),
                int(bool(batch_across_groups)), burst.ctypes.data, env.ctypes.data, pos.ctypes.data,
                nb.ctypes.data), "wire_plan_send")
    return burst[:n], env[:n], pos[:n], int(nb[0])


def bind_wire(lib: GpxLib):
    """Adds the gpx_wire.h entry points to a loaded library (idempotent)."""
    if "wire_decode" in lib.fn:
        return lib
    sigs = dict(_WIRE_SIGS)
    if lib.device_api:
        sigs.update(_WIRE_DEV_SIGS)
    for name, args in sigs.items():
        f = getattr(lib.lib, lib.prefix + name)
        f.argtypes = [_VP] + args
        f.restype = C.c_int
        lib.fn[name] = f
    return lib


# ---- frame builders (what a reference node sends) ----------------------------------------


def _hdr(ptype: int, version: int, paxos_id: bytes) -> bytes:
    assert len(paxos_id) <= 127
    return struct.pack(">iiib", WT_PAXOS_PACKET, ptype, version, len(paxos_id)) + paxos_id


def batched_accept_reply(paxos_id: bytes, version: int, acceptor: int, bnum: int, bcoord: int,
                         max_cp: int, slots, req_ids=None, first_slot=None) -> bytes:
    """BatchedAcceptReply.toBytes: header + AcceptReplyPacket fields (29 B) + n + n x (slot, reqID)."""
    slots = list(slots)
    req_ids = [0] * len(slots) if req_ids is None else list(req_ids)
    first = slots[0] if (first_slot is None and slots) else (first_slot or 0)
    out = _hdr(WT_BATCHED_ACCEPT_REPLY, version, paxos_id)
    out += struct.pack(">iiiiiqb", acceptor, bnum, bcoord, first, max_cp, req_ids[0] if req_ids else 0, 0)
    out += struct.pack(">i", len(slots))
    for s, q in zip(slots, req_ids):
        out += struct.pack(">iq", s, q)
    return out


def batched_commit(paxos_id: bytes, version: int, bnum: int, bcoord: int, median_cp: int, slots,
                   group) -> bytes:
    """BatchedCommit.toBytes: header + ballot + medianCheckpointedSlot + n + slots + g + members."""
    slots, group = list(slots), list(group)
    out = _hdr(WT_BATCHED_COMMIT, version, paxos_id)
    out += struct.pack(">iiii", bnum, bcoord, median_cp, len(slots))
    out += b"".join(struct.pack(">i", s) for s in slots)
    out += struct.pack(">i", len(group)) + b"".join(struct.pack(">i", m) for m in group)
    return out


def request(paxos_id: bytes, version: int, req_id: int, value: bytes = b"", stop: bool = False,
            batched=(), ptype: int = WT_REQUEST, entry_replica: int = -1, digest: bytes = b"",
            response: bytes = b"") -> bytes:
    """RequestPacket.toBytes(): header, requestID, stop, addresses, entry info, digest, value,
    response, batched sub-requests (each a complete RequestPacket byte array)."""
    out = _hdr(ptype, version, paxos_id)
    out += struct.pack(">qb", req_id, 1 if stop else 0)
    out += b"\x00" * 4 + struct.pack(">h", 0) + b"\x00" * 4 + struct.pack(">h", 0)
    out += struct.pack(">iqbi", entry_replica, 0, 0, 0)
    out += struct.pack(">b", 0)  # broadcasted
    out += struct.pack(">i", len(digest)) + digest
    out += struct.pack(">i", len(value)) + value
    out += struct.pack(">i", len(response)) + response
    out += struct.pack(">i", len(batched))
    for el in batched:
        out += struct.pack(">i", len(el)) + el
    return out


def accept(paxos_id: bytes, version: int, req_id: int, slot: int, bnum: int, bcoord: int,
           median_cp: int, sender: int, value: bytes = b"", stop: bool = False, batched=(),
           recovery: bool = False) -> bytes:
    """AcceptPacket.toBytes: the request part (packet type ACCEPT) + slot + ballot + recovery +
    medianCheckpointedSlot + noCoalesce + sender."""
    out = request(paxos_id, version, req_id, value, stop, batched, ptype=WT_ACCEPT)
    out += struct.pack(">iiibibi", slot, bnum, bcoord, 1 if recovery else 0, median_cp, 0, sender)
    return out


def concat_frames(frames):
    """One byte buffer + int64 offsets (n + 1), the form gpx_wire_decode takes."""
    off = np.zeros(len(frames) + 1, np.int64)
    np.cumsum([len(f) for f in frames], out=off[1:])
    buf = np.frombuffer(b"".join(frames), dtype=np.uint8).copy() if frames else np.zeros(0, np.uint8)
    return buf, off


def concat_names(names):
    off = np.zeros(len(names) + 1, np.int32)
    np.cumsum([len(x) for x in names], out=off[1:])
    buf = np.frombuffer(b"".join(names), dtype=np.uint8).copy() if names else np.zeros(0, np.uint8)
    return buf, off


def java_string_hash(b: bytes) -> int:
    """java.lang.String.hashCode of the ISO-8859-1 string with these bytes."""
    h = 0
    for c in b:
        h = (31 * h + c) & 0xFFFFFFFF
    return h - (1 << 32) if h >= (1 << 31) else h


# ---- engine-level API ---------------------------------------------------------------------


@dataclass
class Decoded:
    f_status: np.ndarray
    f_gidx: np.ndarray
    f_type: np.ndarray
    votes: dict
    commits: dict
    accepts: dict
    requests: dict
    counts: dict


class WireEngine:
    """The wire entry points over one Engine handle."""

    def __init__(self, engine: Engine):
        self.e = engine
        self.lib = bind_wire(engine.lib)

    # names --------------------------------------------------------------------------------
    def bind(self, names, gidx) -> np.ndarray:
        buf, off = concat_names(names)
        gidx = _i32(gidx, len(names))
        status = np.zeros(len(names), np.uint8)
        self.lib.check(self.lib.fn["names_bind"](self.e.h, len(names), _p(buf), _p(off), _p(gidx),
                                                 _p(status)), "names_bind")
        return status

    def unbind(self, gidx) -> np.ndarray:
        gidx = _i32(gidx)
        status = np.zeros(gidx.shape[0], np.uint8)
        self.lib.check(self.lib.fn["names_unbind"](self.e.h, gidx.shape[0], _p(gidx), _p(status)),
                       "names_unbind")
        return status

    def lookup(self, names) -> np.ndarray:
        buf, off = concat_names(names)
        out = np.zeros(len(names), np.int32)
        self.lib.check(self.lib.fn["names_lookup"](self.e.h, len(names), _p(buf), _p(off), _p(out)),
                       "names_lookup")
        return out

    def rows_alloc(self, n: int) -> np.ndarray:
        out = np.zeros(n, np.int32)
        self.lib.check(self.lib.fn["rows_alloc"](self.e.h, n, _p(out)), "rows_alloc")
        return out

    def rows_free(self, gidx):
        gidx = _i32(gidx)
        self.lib.check(self.lib.fn["rows_free"](self.e.h, gidx.shape[0], _p(gidx)), "rows_free")

    # decode -------------------------------------------------------------------------------
    def decode(self, frames, cap_votes=None, cap_commits=None, cap_accepts=None, cap_requests=None,
               buf_off=None) -> Decoded:
        """frames: list of bytes (or pass buf_off=(uint8 buffer, int64 offsets))."""
        buf, off = concat_frames(frames) if buf_off is None else buf_off
        nf = off.shape[0] - 1
        total = int(off[-1])
        # a frame of L bytes holds at most L / 4 slot entries: safe default capacities
        dv = total // 12 + 1 if cap_votes is None else cap_votes
        dc = total // 4 + 1 if cap_commits is None else cap_commits
        da = nf if cap_accepts is None else cap_accepts
        dr = nf if cap_requests is None else cap_requests
        i32 = lambda n: np.zeros(max(n, 1), np.int32)  # noqa: E731
        u8 = lambda n: np.zeros(max(n, 1), np.uint8)  # noqa: E731
        i64 = lambda n: np.zeros(max(n, 1), np.int64)  # noqa: E731
        vcols = {k: i32(dv) for k in ("gidx", "bnum", "bcoord", "slot", "acceptor", "max_cp", "frame")}
        ccols = {k: i32(dc) for k in ("gidx", "bnum", "bcoord", "slot", "median_cp", "frame")}
        ccols["kind"] = u8(dc)
        acols = {k: i32(da) for k in ("gidx", "bnum", "bcoord", "slot", "median_cp", "sender", "frame")}
        acols["flags"], acols["req_id"] = u8(da), i64(da)
        rcols = {"gidx": i32(dr), "is_stop": u8(dr), "req_id": i64(dr), "frame": i32(dr)}
        V = WireVotes(dv, *[_p(vcols[k]) for k in ("gidx", "bnum", "bcoord", "slot", "acceptor", "max_cp", "frame")])
        Cc = WireCommits(dc, *[_p(ccols[k]) for k in ("gidx", "bnum", "bcoord", "slot", "median_cp", "kind", "frame")])
        A = WireAccepts(da, *[_p(acols[k]) for k in ("gidx", "bnum", "bcoord", "slot", "median_cp", "flags",
                                                     "sender", "req_id", "frame")])
        R = WireRequests(dr, *[_p(rcols[k]) for k in ("gidx", "is_stop", "req_id", "frame")])
        f_status, f_gidx, f_type = u8(nf), i32(nf), i32(nf)
        cn = WireCounts()
        self.lib.check(self.lib.fn["wire_decode"](self.e.h, nf, _p(buf), _p(off), _p(f_status), _p(f_gidx),
                                                  _p(f_type), C.byref(V), C.byref(Cc), C.byref(A),
                                                  C.byref(R), C.addressof(cn)), "wire_decode")
        counts = {k: int(getattr(cn, k)) for k in ("n_votes", "n_commits", "n_accepts", "n_requests",
                                                   "n_bad_frames")}
        cut = lambda cols, n, cap: {k: v[:min(n, cap)].copy() for k, v in cols.items()}  # noqa: E731
        return Decoded(f_status[:nf], f_gidx[:nf], f_type[:nf],
                       cut(vcols, counts["n_votes"], dv), cut(ccols, counts["n_commits"], dc),
                       cut(acols, counts["n_accepts"], da), cut(rcols, counts["n_requests"], dr), counts)

    # encode -------------------------------------------------------------------------------
    def pack_commits(self, decisions, cap_bytes=None):
        """decisions: a gigapaxos_amd.Decisions (group-major, as accept_reply returns them).
        Returns (list of frame bytes, f_gidx array)."""
        n = int(decisions.gidx.shape[0])
        cap = (256 + 8 * 4) * max(n, 1) if cap_bytes is None else cap_bytes
        out = np.zeros(cap, np.uint8)
        foff, flen, fg = np.zeros(max(n, 1), np.int64), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        nf, nb = np.zeros(1, np.int32), np.zeros(1, np.int64)
        cols = [_i32(getattr(decisions, k), n) for k in ("gidx", "slot", "bnum", "bcoord", "median_cp")]
        kind = np.ascontiguousarray(decisions.kind, dtype=np.uint8)
        self.lib.check(self.lib.fn["wire_pack_commits"](self.e.h, n, *[_p(c) for c in cols], _p(kind), _p(out),
                                                        cap, _p(foff), _p(flen), _p(fg), _p(nf), _p(nb)),
                       "wire_pack_commits")
        m = int(nf[0])
        frames = [out[int(foff[i]):int(foff[i]) + int(flen[i])].tobytes() for i in range(m)]
        return frames, fg[:m].copy(), int(nb[0])

    def pack_accept_replies(self, gidx, slot, r_bnum, r_bcoord, r_maxcp, status, sender=None, req_id=None,
                            cap_bytes=None):
        """The replies of one Engine.accept call -> BATCHED_ACCEPT_REPLY frames.
        Returns (frames, f_gidx, f_dest, unbatched flags, bytes used)."""
        gidx = _i32(gidx)
        n = int(gidx.shape[0])
        cols = [_i32(x, n) for x in (slot, r_bnum, r_bcoord, r_maxcp)]
        status = np.ascontiguousarray(status, dtype=np.uint8)
        sender = None if sender is None else _i32(sender, n)
        req_id = None if req_id is None else np.ascontiguousarray(req_id, dtype=np.int64)
        cap = 188 * max(n, 1) if cap_bytes is None else cap_bytes
        out = np.zeros(cap, np.uint8)
        m = max(n, 1)
        foff, flen, fg, fd = np.zeros(m, np.int64), np.zeros(m, np.int32), np.zeros(m, np.int32), np.zeros(m, np.int32)
        ub = np.zeros(m, np.uint8)
        nf, nb = np.zeros(1, np.int32), np.zeros(1, np.int64)
        self.lib.check(self.lib.fn["wire_pack_accept_replies"](
            self.e.h, n, _p(gidx), _p(cols[0]), _p(sender), _p(req_id), _p(cols[1]), _p(cols[2]), _p(cols[3]),
            _p(status), _p(ub), _p(out), cap, _p(foff), _p(flen), _p(fg), _p(fd), _p(nf), _p(nb)),
            "wire_pack_accept_replies")
        k = int(nf[0])
        frames = [out[int(foff[i]):int(foff[i]) + int(flen[i])].tobytes() for i in range(k)]
        return frames, fg[:k].copy(), fd[:k].copy(), ub[:n].copy(), int(nb[0])


# ---- vectorised builders for large synthetic bursts (bench / scale tests) -----------------


def fixed_names(gidx, width=9):
    """paxosIDs of equal length for many groups: b"g" + zero-padded decimal, as an [n, width] uint8 array."""
    gidx = np.asarray(gidx, np.int64)
    out = np.empty((gidx.shape[0], width), np.uint8)
    out[:, 0] = ord("g")
    v = gidx.copy()
    for c in range(width - 1, 0, -1):
        out[:, c] = ord("0") + v % 10
        v //= 10
    return out


def _be32_cols(x):
    return np.ascontiguousarray(np.asarray(x, np.int64).astype(">i4")).view(np.uint8).reshape(-1, 4)


def bar_frames_single_slot(name_rows, version, acceptor, bnum, bcoord, max_cp, slot):
    """n BATCHED_ACCEPT_REPLY frames of ONE slot each (the common case at 1 M groups: every remote
    acceptor answers one ACCEPT per group and round), all the same length: returns the byte buffer
    and the int64 offsets gpx_wire_decode takes.  Same layout as batched_accept_reply()."""
    name_rows = np.asarray(name_rows, np.uint8)
    n, L = name_rows.shape
    flen = 13 + L + 29 + 4 + 12
    f = np.zeros((n, flen), np.uint8)
    bc = lambda v: _be32_cols(np.broadcast_to(np.asarray(v, np.int64), (n,)))  # noqa: E731
    f[:, 0:4], f[:, 4:8], f[:, 8:12] = bc(WT_PAXOS_PACKET), bc(WT_BATCHED_ACCEPT_REPLY), bc(version)
    f[:, 12] = L
    f[:, 13:13 + L] = name_rows
    o = 13 + L
    f[:, o:o + 4], f[:, o + 4:o + 8], f[:, o + 8:o + 12] = bc(acceptor), bc(bnum), bc(bcoord)
    f[:, o + 12:o + 16], f[:, o + 16:o + 20] = bc(slot), bc(max_cp)
    # requestID (8) = 0, undigestRequest (1) = 0
    f[:, o + 29:o + 33] = bc(1)
    f[:, o + 33:o + 37] = bc(slot)
    off = np.arange(n + 1, dtype=np.int64) * flen
    return f.reshape(-1), off


def accept_frames_fixed(name_rows, version, slot, bnum, bcoord, median_cp, sender, value_len=64):
    """n ACCEPT frames (AcceptPacket.toBytes) with a `value_len`-byte request value each, all the same
    length; requestID = the frame's index.  Same layout as accept()."""
    name_rows = np.asarray(name_rows, np.uint8)
    n, L = name_rows.shape
    tmpl = np.frombuffer(accept(b"x" * L, version, 0, 0, bnum, bcoord, median_cp, sender, b"v" * value_len), np.uint8)
    f = np.tile(tmpl, (n, 1))
    f[:, 13:13 + L] = name_rows
    o = 13 + L
    rid = np.arange(n, dtype=np.int64)
    f[:, o:o + 4], f[:, o + 4:o + 8] = _be32_cols(rid >> 32), _be32_cols(rid & 0xFFFFFFFF)
    t = tmpl.shape[0] - 22  # the slot / ballot / ... tail
    f[:, t:t + 4] = _be32_cols(np.broadcast_to(np.asarray(slot, np.int64), (n,)))
    off = np.arange(n + 1, dtype=np.int64) * tmpl.shape[0]
    return f.reshape(-1), off


def _dev_struct(cls, cap, ptrs):
    return cls(cap, *[_VP(int(p)) if p else None for p in ptrs])


def decode_dev(we: "WireEngine", n_frames, frames_ptr, off_ptr, f_status_ptr, f_gidx_ptr, f_type_ptr,
               votes=None, commits=None, accepts=None, requests=None, counts_ptr=0):
    """gpx_wire_decode_dev with integer device addresses.  votes / commits / accepts / requests:
    (cap, [column addresses in struct order]) or None."""
    mk = lambda cls, spec: None if spec is None else _dev_struct(cls, spec[0], spec[1])  # noqa: E731
    V, Cc, A, R = mk(WireVotes, votes), mk(WireCommits, commits), mk(WireAccepts, accepts), mk(WireRequests, requests)
    ref = lambda s: None if s is None else C.byref(s)  # noqa: E731
    we.lib.check(we.lib.fn["wire_decode_dev"](we.e.h, int(n_frames), _VP(int(frames_ptr)), _VP(int(off_ptr)),
                                              _VP(int(f_status_ptr)), _VP(int(f_gidx_ptr) or None),
                                              _VP(int(f_type_ptr) or None), ref(V), ref(Cc), ref(A), ref(R),
                                              _VP(int(counts_ptr))), "wire_decode_dev")


def pack_commits_dev(we: "WireEngine", n, n_dev_ptr, dec_ptrs, out_ptr, cap_bytes, frame_off_ptr, frame_len_ptr,
                     f_gidx_ptr, n_frames_ptr, n_bytes_ptr):
    """gpx_wire_pack_commits_dev with integer device addresses; dec_ptrs = (gidx, slot, bnum, bcoord,
    median_cp, kind)."""
    a = [_VP(int(p)) for p in dec_ptrs]
    we.lib.check(we.lib.fn["wire_pack_commits_dev"](we.e.h, int(n), _VP(int(n_dev_ptr) or None), *a,
                                                    _VP(int(out_ptr)), int(cap_bytes), _VP(int(frame_off_ptr)),
                                                    _VP(int(frame_len_ptr)), _VP(int(f_gidx_ptr)),
                                                    _VP(int(n_frames_ptr)), _VP(int(n_bytes_ptr))),
                 "wire_pack_commits_dev")


_EXTRA_SIGS = {
    "names_coordinator": [C.c_int32, _VP, C.c_int32, _VP],
    "request_batch": [C.c_int32, _VP, _VP, _VP, _VP, C.c_int32, C.c_int32] + [_VP] * 9,
    "gap_scan": [C.c_int32, _VP, C.c_int32, C.c_int32, C.c_int32] + [_VP] * 5,
    "election_scan": [C.c_int32, _VP, _VP, C.c_int32, _VP, C.c_int32, C.c_int32] + [_VP] * 4,
}
_EXTRA_DEV_SIGS = {"request_batch_dev": _EXTRA_SIGS["request_batch"]}
WIRE_EXPORTED_SYMBOLS += list(_EXTRA_SIGS) + list(_EXTRA_DEV_SIGS)
_WIRE_SIGS.update(_EXTRA_SIGS)
_WIRE_DEV_SIGS.update(_EXTRA_DEV_SIGS)

SYNC_DEFAULT, SYNC_TO_PAUSE, SYNC_FORCE = 0, 1, 2


def names_coordinator(we: WireEngine, gidx, ballotnum=0) -> np.ndarray:
    """PISM.roundRobinCoordinator for bound rows (INT32_MIN where the Java would throw / no name)."""
    gidx = _i32(gidx)
    out = np.zeros(gidx.shape[0], np.int32)
    we.lib.check(we.lib.fn["names_coordinator"](we.e.h, gidx.shape[0], _p(gidx), int(ballotnum), _p(out)),
                 "names_coordinator")
    return out


def request_batch(we: WireEngine, gidx, est_bytes, weight=None, is_stop=None, max_bytes=1 << 20, max_size=2000):
    """RequestBatcher for a burst: returns (leader, status, batches dict of arrays)."""
    gidx = _i32(gidx)
    n = gidx.shape[0]
    est = _i32(est_bytes, n)
    weight = None if weight is None else _i32(weight, n)
    is_stop = None if is_stop is None else np.ascontiguousarray(is_stop, dtype=np.uint8)
    m = max(n, 1)
    leader, status = np.zeros(m, np.int32), np.zeros(m, np.uint8)
    cols = [np.zeros(m, np.int32) for _ in range(5)]
    bstop, nb = np.zeros(m, np.uint8), np.zeros(1, np.int32)
    we.lib.check(we.lib.fn["request_batch"](we.e.h, n, _p(gidx), _p(est), _p(weight), _p(is_stop), int(max_bytes),
                                            int(max_size), _p(leader), _p(status), *[_p(c) for c in cols],
                                            _p(bstop), _p(nb)), "request_batch")
    k = int(nb[0])
    names = ("gidx", "leader", "count", "bytes", "size")
    b = {nm: c[:k].copy() for nm, c in zip(names, cols)}
    b["stop"] = bstop[:k].copy()
    return leader[:n].copy(), status[:n].copy(), b


def gap_scan(we: WireEngine, gidx, threshold, sync_mode=SYNC_DEFAULT, size_limit=64):
    """PaxosAcceptor.getMissingCommittedSlots / getMaxCommittedSlot + PISM.shouldSync per group."""
    gidx = _i32(gidx)
    n = gidx.shape[0]
    first, maxc = np.zeros(n, np.int32), np.zeros(n, np.int32)
    missing, sync, st = np.zeros(n, np.uint64), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    we.lib.check(we.lib.fn["gap_scan"](we.e.h, n, _p(gidx), int(threshold), int(sync_mode), int(size_limit),
                                       _p(first), _p(maxc), _p(missing), _p(sync), _p(st)), "gap_scan")
    return first, maxc, missing, sync, st


def election_scan(we: WireEngine, gidx, down_nodes=(), long_dead_nodes=(), force=False):
    """PISM.checkRunForCoordinator's decision per group: (run reason, PREPARE ballot number,
    firstUndecidedSlot, status).  gidx None = all groups 0 .. max_groups-1."""
    if gidx is None:
        n, g = int(we.e.cfg.max_groups), None
    else:
        g = _i32(gidx)
        n = g.shape[0]
    dn = np.ascontiguousarray(list(down_nodes), dtype=np.int32)
    ld = np.ascontiguousarray(list(long_dead_nodes), dtype=np.int32)
    run, st = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    pb, pf = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    we.lib.check(we.lib.fn["election_scan"](we.e.h, n, _p(g), _p(dn) if dn.size else None, int(dn.size),
                                            _p(ld) if ld.size else None, int(ld.size), int(bool(force)),
                                            _p(run), _p(pb), _p(pf), _p(st)), "election_scan")
    return run[:n], pb[:n], pf[:n], st[:n]
