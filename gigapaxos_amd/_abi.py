"""ctypes binding of the C-ABI declared in include/gpx.h.

`GpxLib(path, prefix)` binds one shared library that exports the ABI under a symbol
prefix.  The product library is ``gigapaxos_amd/csrc/libgpx_hip.so`` (prefix ``gpx_``,
hand-written HIP for gfx950); there is NO CPU fallback: if that library is missing
or fails to load, `load_hip()` raises.  (tests/ bind the CPU oracle through the same
class with prefix ``orc_`` — as the checker only.)

`Engine` is the thin host-side handle: numpy arrays in, numpy arrays out, one method
per reference handler it replaces (PaxosInstanceStateMachine.handlePaxosMessage's
switch, PaxosInstanceStateMachine.java:423-583).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass

import numpy as np

KMAX_LIMIT = 16

# status / kind constants (include/gpx.h)
S_OK, S_NOGROUP, S_STOPPED, S_WINDOW, S_FORWARD, S_REFUSED, S_EXISTS, S_BUSY = range(8)
S_UNORDERED = 9
ORDERED_PROPOSE, ORDERED_ACCEPT, ORDERED_COMMIT = 1, 2, 4
ORDERED_REPLY_RUNS, TRY_REPLY_RUNS = 8, 16
LAZY_OUTPUTS = 32
D_DECISION, D_PREEMPTED = 1, 2
R_TOLOG, R_STORED = 1, 2
A_STOP = 1
C_HASVALUE, C_STOP = 1, 2
F_ACCEPTS_FROM_DISK = 1
RETIRE_PAUSE, RETIRE_KILL = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_HERE, "csrc", "libgpx_hip.so")


class GpxConfig(C.Structure):
    _fields_ = [
        ("my_id", C.c_int32),
        ("max_groups", C.c_int32),
        ("kmax", C.c_int32),
        ("window", C.c_int32),
        ("max_batch", C.c_int32),
        ("device", C.c_int32),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class GpxKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


# numpy mirror of struct gpx_hri (HotRestoreInfo.java:35-84 minus the name)
HRI_DTYPE = np.dtype(
    [
        ("version", "<i4"),
        ("acc_slot", "<i4"),
        ("acc_bnum", "<i4"),
        ("acc_bcoord", "<i4"),
        ("acc_gc_slot", "<i4"),
        ("has_coord", "<i4"),
        ("coord_bnum", "<i4"),
        ("coord_bcoord", "<i4"),
        ("next_proposal_slot", "<i4"),
        ("node_slots", "<i4", (KMAX_LIMIT,)),
    ]
)
assert HRI_DTYPE.itemsize == 4 * (9 + KMAX_LIMIT)

_I32P = C.POINTER(C.c_int32)
_U8P = C.POINTER(C.c_uint8)
_VP = C.c_void_p

# name -> argtypes (after the engine handle); every function returns int
_SIGS = {
    "engine_destroy": [],
    "engine_sync": [],
    "engine_counters": [C.POINTER(C.c_uint64)],
    "engine_set_ordered_batches": [C.c_int32],
    "host_register": [_VP, C.c_size_t],
    "host_unregister": [_VP],
    "host_alloc": [C.c_size_t, C.POINTER(_VP)],
    "host_free": [_VP],
    "group_create": [C.c_int32, _VP, _VP, _VP, _VP, _VP],
    "group_retire": [C.c_int32, _VP, C.c_int32, _VP, _VP],
    "group_snapshot": [C.c_int32, _VP, _VP, _VP],
    "group_dump": [C.c_int32, _VP, C.c_int32],
    "propose_batch": [C.c_int32] + [_VP] * 7,
    "accept_batch": [C.c_int32] + [_VP] * 15,
    "accept_reply_batch": [C.c_int32] + [_VP] * 14,
    "commit_batch": [C.c_int32] + [_VP] * 11,
    "prepare_batch": [C.c_int32] + [_VP] * 13,
    "election_begin": [C.c_int32] + [_VP] * 3,
    "prepare_reply_batch": [C.c_int32] + [_VP] * 19,
    "propose_batch_h": [C.c_int32] + [_VP] * 8,
    "poke_scan": [C.c_int32] + [_VP] * 9,
}
_DEV_SIGS = {
    "engine_path_counters": [C.POINTER(C.c_uint64)],
    "engine_set_stream": [_VP],
    "propose_batch_dev": [C.c_int32] + [_VP] * 7,
    "accept_batch_dev": [C.c_int32] + [_VP] * 15,
    "accept_reply_batch_dev": [C.c_int32] + [_VP] * 14,
    "commit_batch_dev": [C.c_int32] + [_VP] * 11,
    "prepare_batch_dev": [C.c_int32] + [_VP] * 13,
    "election_begin_dev": [C.c_int32] + [_VP] * 3,
    "propose_batch_h_dev": [C.c_int32] + [_VP] * 8,
    "prepare_reply_batch_dev": [C.c_int32] + [_VP] * 6 + [C.c_int32] + [_VP] * 13,
    "route_batch_dev": [C.c_int32, C.c_int32, _VP, _VP, C.c_int32, C.c_int32, _VP, _VP],
    # asynchronous host-pointer calls (HIP library only: the oracle has nothing to overlap)
    "propose_batch_async": [C.c_int32] + [_VP] * 7 + [C.POINTER(C.c_uint64)],
    "accept_batch_async": [C.c_int32] + [_VP] * 15 + [C.POINTER(C.c_uint64)],
    "accept_reply_batch_async": [C.c_int32] + [_VP] * 3 + [C.c_int32, C.c_int32] + [_VP] * 11 + [C.POINTER(C.c_uint64)],
    "commit_batch_async": [C.c_int32] + [_VP] * 11 + [C.POINTER(C.c_uint64)],
    "engine_wait": [C.c_uint64],
    "compact_last_dev": [],
    "profile_enable": [C.c_int32],
    "profile_read": [C.POINTER(GpxKernelStat), C.c_int32],
}

EXPORTED_SYMBOLS = (
    ["abi_version", "last_error", "engine_create"] + list(_SIGS) + list(_DEV_SIGS)
)


class GpxError(RuntimeError):
    pass


class GpxLib:
    """One loaded shared library exporting the gpx ABI under `prefix`."""

    def __init__(self, path: str, prefix: str = "gpx_", device_api: bool = True):
        if not os.path.exists(path):
            raise GpxError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            )
        self.path = path
        self.prefix = prefix
        if prefix == "gpx_":
            # ONE HIP runtime per process, and it is torch's: torch ships its own libamdhip64.so.7 and the HIP library
            # here needs the same soname.  Loaded after torch, the library binds to the copy torch already brought; loaded
            # BEFORE torch it pulls /opt/rocm's copy first, and a process that then also uses torch ends with the engine
            # reporting "no ROCm-capable device" (seen with build() and smoke() in one process, round 6).
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self.lib = C.CDLL(path)
        self.fn = {}
        f = getattr(self.lib, prefix + "abi_version")
        f.restype = C.c_int
        self.abi_version = f()
        f = getattr(self.lib, prefix + "last_error")
        f.restype = C.c_char_p
        self.fn["last_error"] = f
        f = getattr(self.lib, prefix + "engine_create")
        f.argtypes = [C.POINTER(GpxConfig), C.POINTER(_VP)]
        f.restype = C.c_int
        self.fn["engine_create"] = f
        sigs = dict(_SIGS)
        if device_api:
            sigs.update(_DEV_SIGS)
        for name, args in sigs.items():
            f = getattr(self.lib, prefix + name)
            f.argtypes = [_VP] + args
            f.restype = C.c_int
            self.fn[name] = f
        self.device_api = device_api

    def check(self, rc: int, what: str) -> int:
        if rc < 0:
            msg = self.fn["last_error"]()
            raise GpxError(f"{self.prefix}{what} failed rc={rc} {msg.decode() if msg else ''}")
        return rc


_hip_lib = None


def load_hip() -> GpxLib:
    """The product library.  Fails loudly when the HIP extension is missing."""
    global _hip_lib
    if _hip_lib is None:
        # GPX_HIP_LIB: tuning aid, points at an alternative build of the same HIP library
        _hip_lib = GpxLib(os.environ.get("GPX_HIP_LIB", HIP_LIB_PATH), "gpx_", device_api=True)
    return _hip_lib


def _p(a):
    return None if a is None else a.ctypes.data_as(_VP)


def _i32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if n is not None and a.shape[0] != n:
        raise ValueError("column length mismatch")
    return a


def _u8(a, n=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if n is not None and a.shape[0] != n:
        raise ValueError("column length mismatch")
    return a


@dataclass
class Decisions:
    gidx: np.ndarray
    slot: np.ndarray
    bnum: np.ndarray
    bcoord: np.ndarray
    median_cp: np.ndarray
    kind: np.ndarray
    status: np.ndarray  # per input vote

    def as_tuple_array(self):
        return np.stack(
            [self.gidx, self.slot, self.bnum, self.bcoord, self.median_cp, self.kind.astype(np.int32)],
            axis=1,
        )


@dataclass
class ExecRuns:
    gidx: np.ndarray
    first: np.ndarray
    count: np.ndarray

    def as_tuple_array(self):
        return np.stack([self.gidx, self.first, self.count], axis=1)


def make_hri(n: int) -> np.ndarray:
    return np.zeros(n, dtype=HRI_DTYPE)


def hri_create(n: int, k: int, coordinator) -> np.ndarray:
    """Rows as HotRestoreInfo.createHRI gives them (HotRestoreInfo.java:145-157):
    accSlot=1, accBallot=coordBallot=(0,coordinator), accGCSlot=-1, nextProposalSlot=1,
    nodeSlots = zeros."""
    rows = make_hri(n)
    rows["acc_slot"] = 1
    rows["acc_bcoord"] = coordinator
    rows["acc_gc_slot"] = -1
    rows["has_coord"] = 1
    rows["coord_bcoord"] = coordinator
    rows["next_proposal_slot"] = 1
    return rows


def hri_initial(n: int, k: int, coordinator) -> np.ndarray:
    """Rows equivalent to regular creation with an initial-state checkpoint
    (PaxosInstanceStateMachine.java:612-618, 656-668, 692-699): acceptor slot 1,
    gcSlot 0, ballot (0,coordinator); coordinator nextProposalSlot 1 and
    nodeSlotNumbers = -1 (PaxosCoordinatorState.java:173-175)."""
    rows = hri_create(n, k, coordinator)
    rows["acc_gc_slot"] = 0
    rows["node_slots"][:, :k] = -1
    return rows


def hri_to_string(paxos_id: str, members, row) -> str:
    """HotRestoreInfo.toString (paxosutil/HotRestoreInfo.java:102-122): the pipe-separated form the
    reference keeps in its pause table - paxosID|version|[members]|accSlot|accBallot|accGCSlot|
    coordBallot or null|nextProposalSlot|[nodeSlots] or null (Util.arrayOfIntToString, Ballot.toString)."""
    members = [int(m) for m in members]
    k = len(members)
    arr = lambda a: "[" + ",".join(str(int(x)) for x in a) + "]"  # noqa: E731
    coord = bool(row["has_coord"])
    return "|".join([paxos_id, str(int(row["version"])), arr(members), str(int(row["acc_slot"])),
                     "%d:%d" % (int(row["acc_bnum"]), int(row["acc_bcoord"])), str(int(row["acc_gc_slot"])),
                     "%d:%d" % (int(row["coord_bnum"]), int(row["coord_bcoord"])) if coord else "null",
                     str(int(row["next_proposal_slot"])),
                     arr(row["node_slots"][:k]) if coord else "null"])


def hri_from_string(s: str):
    """HotRestoreInfo(String) (HotRestoreInfo.java:86-98): -> (paxosID, members, row)."""
    t = s.split("|")
    ints = lambda x: [int(v) for v in x.replace("[", "").replace("]", "").replace(" ", "").split(",")]  # noqa: E731
    row = make_hri(1)
    members = ints(t[2])
    row["version"], row["acc_slot"], row["acc_gc_slot"] = int(t[1]), int(t[3]), int(t[5])
    row["acc_bnum"], row["acc_bcoord"] = (int(v) for v in t[4].split(":"))
    if t[6] != "null":
        row["has_coord"] = 1
        row["coord_bnum"], row["coord_bcoord"] = (int(v) for v in t[6].split(":"))
    row["next_proposal_slot"] = int(t[7])
    if t[8] != "null":
        ns = ints(t[8])
        row["node_slots"][0, :len(ns)] = ns
    return t[0], members, row


class Engine:
    """Host handle over one engine (`gpx_engine*`): the device-resident replacement of
    PaxosManager's name->PaxosInstanceStateMachine table for one node id."""

    def __init__(self, lib: GpxLib, my_id: int, max_groups: int, kmax: int = 3, window: int = 8,
                 max_batch: int = 1 << 20, device: int = -1, flags: int = F_ACCEPTS_FROM_DISK):
        self.lib = lib
        self.cfg = GpxConfig(my_id, max_groups, kmax, window, max_batch, device, flags, 0)
        h = _VP()
        lib.check(lib.fn["engine_create"](C.byref(self.cfg), C.byref(h)), "engine_create")
        self.h = h
        self.kmax = kmax
        self.my_id = my_id
        self._host_views = {}  # address -> weakref of the numpy view handed out by host_alloc

    def close(self, force: bool = False):
        """gpx_engine_destroy.  Memory from host_alloc goes with it: while arrays over it are still referenced the call
        raises (a later access would read freed pages) - drop them first, or force=True (the destructor's choice)."""
        if self.h:
            views = getattr(self, "_host_views", None)
            if views and not force:
                live = [r() for r in views.values() if r() is not None]
                if live:
                    raise GpxError(f"{len(live)} host_alloc array(s) still referenced at close(): host_free them or drop "
                                   "the references first (their memory is freed by gpx_engine_destroy)")
            self.lib.fn["engine_destroy"](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close(force=True)
        except Exception:
            pass

    # -- lifecycle (PaxosManager.createPaxosInstance / kill / pause) ---------------
    def create_groups(self, gidx, members, k, rows) -> np.ndarray:
        gidx = _i32(gidx)
        n = gidx.shape[0]
        members = np.ascontiguousarray(members, dtype=np.int32).reshape(n, self.kmax)
        k = _u8(np.broadcast_to(np.asarray(k, dtype=np.uint8), (n,)), n)
        rows = np.ascontiguousarray(rows, dtype=HRI_DTYPE)
        assert rows.shape[0] == n
        status = np.zeros(n, np.uint8)
        self.lib.check(
            self.lib.fn["group_create"](self.h, n, _p(gidx), _p(members), _p(k), _p(rows), _p(status)),
            "group_create",
        )
        return status

    def retire_groups(self, gidx, mode=RETIRE_PAUSE):
        gidx = _i32(gidx)
        n = gidx.shape[0]
        rows = make_hri(n)
        status = np.zeros(n, np.uint8)
        self.lib.check(
            self.lib.fn["group_retire"](self.h, n, _p(gidx), mode, _p(rows), _p(status)), "group_retire"
        )
        return rows, status

    def snapshot(self, gidx):
        gidx = _i32(gidx)
        n = gidx.shape[0]
        rows = make_hri(n)
        status = np.zeros(n, np.uint8)
        self.lib.check(
            self.lib.fn["group_snapshot"](self.h, n, _p(gidx), _p(rows), _p(status)), "group_snapshot"
        )
        return rows, status

    def dump(self, g: int) -> np.ndarray:
        cap = 16 + 3 * KMAX_LIMIT + 16 * 64 * 6
        buf = np.zeros(cap, np.int32)
        nw = self.lib.check(self.lib.fn["group_dump"](self.h, int(g), _p(buf), cap), "group_dump")
        return buf[:nw].copy()

    def counters(self):
        out = (C.c_uint64 * 3)()
        self.lib.check(self.lib.fn["engine_counters"](self.h, out), "engine_counters")
        return tuple(int(x) for x in out)

    def path_counters(self):
        """(accept-reply calls whose outputs were written in place, calls compacted from the staging) - gpx_engine_path_counters"""
        out = (C.c_uint64 * 2)()
        self.lib.check(self.lib.fn["engine_path_counters"](self.h, out), "engine_path_counters")
        return int(out[0]), int(out[1])

    def host_register(self, *arrays):
        """Pins numpy arrays for DMA (gpx_host_register); returns them.  Unpin with host_unregister."""
        for a in arrays:
            self.lib.check(self.lib.fn["host_register"](self.h, _p(a), a.nbytes), "host_register")
        return arrays

    def host_unregister(self, *arrays):
        for a in arrays:
            self.lib.check(self.lib.fn["host_unregister"](self.h, _p(a)), "host_unregister")

    def host_alloc(self, n: int, dtype=np.int32) -> np.ndarray:
        """A numpy array of n elements over memory from gpx_host_alloc (hipHostMalloc): the DMA engines reach it
        at the link's full rate.  Give it back with host_free(array) - or leave it to close().  The array is a VIEW of
        that block: it must not be touched after host_free / close() (numpy cannot be told; close() refuses to go on
        while such arrays are still referenced elsewhere, unless force=True)."""
        dt = np.dtype(dtype)
        nbytes = max(int(n), 1) * dt.itemsize
        p = _VP()
        self.lib.check(self.lib.fn["host_alloc"](self.h, nbytes, C.byref(p)), "host_alloc")
        buf = (C.c_char * nbytes).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt, count=int(n))
        a[...] = 0
        self._host_views[p.value] = weakref.ref(a)
        return a

    def host_free(self, *arrays):
        for a in arrays:
            self._host_views.pop(a.ctypes.data, None)
            self.lib.check(self.lib.fn["host_free"](self.h, _p(a)), "host_free")

    def live_host_views(self):
        """Arrays from host_alloc that somebody still holds (their memory goes away with close())."""
        return [r() for r in self._host_views.values() if r() is not None]

    def sync(self):
        self.lib.check(self.lib.fn["engine_sync"](self.h), "engine_sync")

    def set_ordered_batches(self, mask: int):
        """Promise (verified on the device) that later propose / accept / commit batches come grouped
        by group, groups ascending: only the direct path is launched (gpx_engine_set_ordered_batches).
        ORDERED_REPLY_RUNS: accept-reply batches are a few ascending runs (the acceptors' replies
        concatenated); TRY_REPLY_RUNS: a hint, checked on the device, any other batch is partitioned."""
        self.lib.check(self.lib.fn["engine_set_ordered_batches"](self.h, int(mask)), "engine_set_ordered_batches")

    # -- device-pointer path (batches already resident in HBM) ----------------------
    def set_stream(self, hip_stream_handle: int):
        """Run the *_dev calls on this hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        self.lib.check(self.lib.fn["engine_set_stream"](self.h, _VP(hip_stream_handle or None)),
                       "engine_set_stream")

    def compact_last_dev(self):
        """GPX_LAZY_OUTPUTS: make the most recent *_dev call's parked outputs dense (gpx_compact_last_dev)."""
        self.lib.check(self.lib.fn["compact_last_dev"](self.h), "compact_last_dev")

    def call_dev(self, name: str, n: int, *ptrs):
        """Raw asynchronous call of gpx_<name>_dev with integer device addresses (0 = NULL)."""
        args = [_VP(int(p)) if p else None for p in ptrs]
        self.lib.check(self.lib.fn[name + "_dev"](self.h, int(n), *args), name + "_dev")

    def route_dev(self, n: int, col_ptrs, g2l_ptr: int, n_groups_global: int, n_shards: int, out_ptrs,
                  shard_off_ptr: int):
        """gpx_route_batch_dev: stable partition of device columns by fmix32(gidx) % n_shards (integer
        device addresses; col_ptrs[0] = the global group index column)."""
        k = len(col_ptrs)
        ins = (C.c_void_p * k)(*[int(p) for p in col_ptrs])
        outs = (C.c_void_p * k)(*[int(p) for p in out_ptrs])
        self.lib.check(self.lib.fn["route_batch_dev"](self.h, int(n), k, ins, _VP(int(g2l_ptr) or None),
                                                      int(n_groups_global), int(n_shards), outs,
                                                      _VP(int(shard_off_ptr))), "route_batch_dev")

    def profile(self, enable: int):
        """0 = off, 1 = on, 2 = on + reset accumulated stats."""
        self.lib.check(self.lib.fn["profile_enable"](self.h, int(enable)), "profile_enable")

    def profile_read(self):
        buf = (GpxKernelStat * 32)()
        nk = self.lib.check(self.lib.fn["profile_read"](self.h, buf, 32), "profile_read")
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(min(nk, 32))}

    # -- asynchronous host-pointer path (gpx_*_batch_async / gpx_engine_wait) -------------------------
    class Pending:
        """A submitted call: keeps every buffer alive until wait()."""

        def __init__(self, eng, ticket, bufs, finish):
            self.eng, self.ticket, self.bufs, self.finish = eng, ticket, bufs, finish

        def wait(self):
            self.eng.lib.check(self.eng.lib.fn["engine_wait"](self.eng.h, C.c_uint64(self.ticket)), "engine_wait")
            return self.finish()

    @staticmethod
    def page_array(n: int, dtype=np.int32) -> np.ndarray:
        """A zeroed numpy array over its OWN anonymous pages (mmap: page-aligned, whole pages, nothing else of the heap
        on them) - what gpx_host_register should be given (include/gpx.h: a few page-aligned blocks, never sub-page
        ranges of the malloc heap).  Unmapped when the last reference goes."""
        import mmap
        dt = np.dtype(dtype)
        nbytes = max(int(n), 1) * dt.itemsize
        mm = mmap.mmap(-1, (nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE)
        return np.frombuffer(mm, dtype=dt, count=int(n))

    def propose_async(self, gidx, is_stop=None, pin_outputs=False):
        gidx = _i32(gidx)
        n = gidx.shape[0]
        is_stop = _u8(is_stop, n)
        mk = Engine.page_array if pin_outputs else np.zeros   # registered outputs: whole pages of their own
        slot, bnum, bcoord, median = (mk(n, np.int32) for _ in range(4))
        status = mk(n, np.uint8)
        outs = (slot, bnum, bcoord, median, status)
        if pin_outputs:
            self.host_register(*outs)
        t = C.c_uint64(0)
        self.lib.check(self.lib.fn["propose_batch_async"](self.h, n, _p(gidx), _p(is_stop), _p(slot), _p(bnum), _p(bcoord),
                                                          _p(median), _p(status), C.byref(t)), "propose_batch_async")
        def finish():
            if pin_outputs:
                self.host_unregister(*outs)
            return slot, bnum, bcoord, median, status
        return Engine.Pending(self, t.value, (gidx, is_stop), finish)

    def accept_reply_async(self, gidx, bnum, bcoord, slot, acceptor, max_cp, common_ballot=None, pin_outputs=False):
        """bnum / bcoord None + common_ballot = (bnum, bcoord): every vote carries that ballot.
        pin_outputs: the output arrays are registered (gpx_host_register) for the call - the engine then writes
        the compacted outputs there itself, without a host round trip for the count."""
        gidx = _i32(gidx)
        n = gidx.shape[0]
        slot, acceptor, max_cp = (_i32(x, n) for x in (slot, acceptor, max_cp))
        if bnum is not None:
            bnum, bcoord = _i32(bnum, n), _i32(bcoord, n)
        cb = common_ballot or (0, 0)
        cap = max(n, 1)
        mk = Engine.page_array if pin_outputs else np.zeros   # registered outputs: whole pages of their own
        dg, ds, db, dc, dm = (mk(cap, np.int32) for _ in range(5))
        dk = mk(cap, np.uint8)
        status = mk(n, np.uint8)
        no = mk(1, np.int32)
        outs = (dg, ds, db, dc, dm, dk, no, status)
        if pin_outputs:
            self.host_register(*outs)
        t = C.c_uint64(0)
        self.lib.check(self.lib.fn["accept_reply_batch_async"](
            self.h, n, _p(gidx), _p(bnum), _p(bcoord), int(cb[0]), int(cb[1]), _p(slot), _p(acceptor), _p(max_cp),
            _p(dg), _p(ds), _p(db), _p(dc), _p(dm), _p(dk), _p(no), _p(status), C.byref(t)), "accept_reply_batch_async")

        def finish():
            if pin_outputs:
                self.host_unregister(*outs)
            m = int(no[0])
            return Decisions(dg[:m], ds[:m], db[:m], dc[:m], dm[:m], dk[:m], status)
        return Engine.Pending(self, t.value, (gidx, bnum, bcoord, slot, acceptor, max_cp), finish)

    def accept_async(self, gidx, bnum, bcoord, slot, median_cp, a_flags=None):
        gidx = _i32(gidx)
        n = gidx.shape[0]
        bnum, bcoord, slot, median_cp = (_i32(x, n) for x in (bnum, bcoord, slot, median_cp))
        a_flags = _u8(a_flags, n)
        r_bnum, r_bcoord, r_maxcp = (np.zeros(n, np.int32) for _ in range(3))
        r_flags, status = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        xg, xf, xc = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        nr = np.zeros(1, np.int32)
        t = C.c_uint64(0)
        self.lib.check(self.lib.fn["accept_batch_async"](
            self.h, n, _p(gidx), _p(bnum), _p(bcoord), _p(slot), _p(median_cp), _p(a_flags), _p(r_bnum), _p(r_bcoord),
            _p(r_maxcp), _p(r_flags), _p(status), _p(xg), _p(xf), _p(xc), _p(nr), C.byref(t)), "accept_batch_async")

        def finish():
            m = int(nr[0])
            return (r_bnum, r_bcoord, r_maxcp, r_flags, status), ExecRuns(xg[:m], xf[:m], xc[:m])
        return Engine.Pending(self, t.value, (gidx, bnum, bcoord, slot, median_cp, a_flags), finish)

    def commit_async(self, gidx, bnum, bcoord, slot, median_cp, c_kind=None):
        gidx = _i32(gidx)
        n = gidx.shape[0]
        bnum, bcoord, slot, median_cp = (_i32(x, n) for x in (bnum, bcoord, slot, median_cp))
        c_kind = _u8(c_kind, n)
        status = np.zeros(n, np.uint8)
        xg, xf, xc = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        nr = np.zeros(1, np.int32)
        t = C.c_uint64(0)
        self.lib.check(self.lib.fn["commit_batch_async"](
            self.h, n, _p(gidx), _p(bnum), _p(bcoord), _p(slot), _p(median_cp), _p(c_kind), _p(status), _p(xg), _p(xf),
            _p(xc), _p(nr), C.byref(t)), "commit_batch_async")

        def finish():
            m = int(nr[0])
            return status, ExecRuns(xg[:m], xf[:m], xc[:m])
        return Engine.Pending(self, t.value, (gidx, bnum, bcoord, slot, median_cp, c_kind), finish)

    # -- data path ---------------------------------------------------------------
    def propose(self, gidx, is_stop=None, handle=None):
        """PISM.handleRequest/handleProposal for a batch of (already batched) requests.
        With `handle` (int64 per request) the call goes through gpx_propose_batch_h."""
        gidx = _i32(gidx)
        n = gidx.shape[0]
        is_stop = _u8(is_stop, n)
        slot, bnum, bcoord, median = (np.zeros(n, np.int32) for _ in range(4))
        status = np.zeros(n, np.uint8)
        if handle is None:
            self.lib.check(
                self.lib.fn["propose_batch"](self.h, n, _p(gidx), _p(is_stop), _p(slot), _p(bnum),
                                             _p(bcoord), _p(median), _p(status)),
                "propose_batch",
            )
        else:
            handle = np.ascontiguousarray(handle, np.int64)
            assert handle.shape[0] == n
            self.lib.check(
                self.lib.fn["propose_batch_h"](self.h, n, _p(gidx), _p(is_stop), _p(handle), _p(slot),
                                               _p(bnum), _p(bcoord), _p(median), _p(status)),
                "propose_batch_h",
            )
        return slot, bnum, bcoord, median, status

    def election_begin(self, gidx, bnum):
        """PISM.tryMakeCoordinator -> PaxosCoordinator.makeCoordinator for a batch of groups."""
        gidx = _i32(gidx)
        n = gidx.shape[0]
        bnum = _i32(bnum, n)
        st = np.zeros(max(n, 1), np.uint8)
        self.lib.check(self.lib.fn["election_begin"](self.h, n, _p(gidx), _p(bnum), _p(st)), "election_begin")
        return st[:n]

    def poke_scan(self, gidx=None):
        """What is waiting for replies (pokeLocalCoordinator / PREPARE resend minus the clocks):
        (poke kind, slot, bnum, bcoord, median_cp, flags, heard mask, status) per group; gidx None =
        all groups."""
        if gidx is None:
            n, g = int(self.cfg.max_groups), None
        else:
            g = _i32(gidx)
            n = g.shape[0]
        m = max(n, 1)
        pk, fl, st = (np.zeros(m, np.uint8) for _ in range(3))
        sl, bn, bc, md = (np.zeros(m, np.int32) for _ in range(4))
        hd = np.zeros(m, np.uint32)
        self.lib.check(self.lib.fn["poke_scan"](self.h, n, _p(g) if g is not None else None, _p(pk), _p(sl), _p(bn),
                                                _p(bc), _p(md), _p(fl), _p(hd), _p(st)), "poke_scan")
        return pk[:n], sl[:n], bn[:n], bc[:n], md[:n], fl[:n], hd[:n], st[:n]

    def prepare_reply(self, gidx, acceptor, r_bnum, r_bcoord, first_slot, pvalues=None):
        """PISM.handlePrepareReply for a batch of PREPARE_REPLYs.  `pvalues[i]` = the reply's
        accepted pvalues as (slot, bnum, bcoord, handle, flags) tuples.  Returns (v_kind, e_median,
        status) and per record the list of (slot, kind, handle, flags) entries."""
        gidx = _i32(gidx)
        n = gidx.shape[0]
        acceptor, r_bnum, r_bcoord, first_slot = (_i32(x, n) for x in (acceptor, r_bnum, r_bcoord, first_slot))
        pvalues = pvalues if pvalues is not None else [[] for _ in range(n)]
        off = np.zeros(n + 1, np.int32)
        off[1:] = np.cumsum([len(p) for p in pvalues])
        flat = [t for p in pvalues for t in p]
        m = max(len(flat), 1)
        ps, pb, pc = (np.zeros(m, np.int32) for _ in range(3))
        ph, pf = np.zeros(m, np.int64), np.zeros(m, np.uint8)
        for j, t in enumerate(flat):
            ps[j], pb[j], pc[j], ph[j], pf[j] = t
        W = int(self.cfg.window)
        q = max(n, 1)
        vk, st = np.zeros(q, np.uint8), np.zeros(q, np.uint8)
        ec, em = np.zeros(q, np.int32), np.zeros(q, np.int32)
        es = np.zeros(q * W, np.int32)
        ek, ef = np.zeros(q * W, np.uint8), np.zeros(q * W, np.uint8)
        eh = np.zeros(q * W, np.int64)
        self.lib.check(
            self.lib.fn["prepare_reply_batch"](self.h, n, _p(gidx), _p(acceptor), _p(r_bnum), _p(r_bcoord),
                                               _p(first_slot), _p(off), _p(ps), _p(pb), _p(pc), _p(ph), _p(pf),
                                               _p(vk), _p(ec), _p(em), _p(es), _p(ek), _p(eh), _p(ef), _p(st)),
            "prepare_reply_batch",
        )
        lists = [[(int(es[j * n + i]), int(ek[j * n + i]), int(eh[j * n + i]), int(ef[j * n + i]))
                  for j in range(int(ec[i]))] for i in range(n)]
        return (vk[:n], em[:n], st[:n]), lists

    def accept(self, gidx, bnum, bcoord, slot, median_cp, a_flags=None):
        """PISM.handleAccept for a batch of ACCEPTs."""
        gidx = _i32(gidx)
        n = gidx.shape[0]
        bnum, bcoord, slot, median_cp = (_i32(x, n) for x in (bnum, bcoord, slot, median_cp))
        a_flags = _u8(a_flags, n)
        r_bnum, r_bcoord, r_maxcp = (np.zeros(n, np.int32) for _ in range(3))
        r_flags = np.zeros(n, np.uint8)
        status = np.zeros(n, np.uint8)
        xg, xf, xc = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        nr = np.zeros(1, np.int32)
        self.lib.check(
            self.lib.fn["accept_batch"](self.h, n, _p(gidx), _p(bnum), _p(bcoord), _p(slot),
                                        _p(median_cp), _p(a_flags), _p(r_bnum), _p(r_bcoord),
                                        _p(r_maxcp), _p(r_flags), _p(status), _p(xg), _p(xf), _p(xc),
                                        _p(nr)),
            "accept_batch",
        )
        m = int(nr[0])
        return (r_bnum, r_bcoord, r_maxcp, r_flags, status), ExecRuns(xg[:m], xf[:m], xc[:m])

    def accept_reply(self, gidx, bnum, bcoord, slot, acceptor, max_cp) -> Decisions:
        """PISM.handleBatchedAcceptReply/handleAcceptReply for a batch of votes."""
        gidx = _i32(gidx)
        n = gidx.shape[0]
        bnum, bcoord, slot, acceptor, max_cp = (_i32(x, n) for x in (bnum, bcoord, slot, acceptor, max_cp))
        cap = max(n, 1)
        dg, ds, db, dc, dm = (np.zeros(cap, np.int32) for _ in range(5))
        dk = np.zeros(cap, np.uint8)
        status = np.zeros(n, np.uint8)
        no = np.zeros(1, np.int32)
        self.lib.check(
            self.lib.fn["accept_reply_batch"](self.h, n, _p(gidx), _p(bnum), _p(bcoord), _p(slot),
                                              _p(acceptor), _p(max_cp), _p(dg), _p(ds), _p(db), _p(dc),
                                              _p(dm), _p(dk), _p(no), _p(status)),
            "accept_reply_batch",
        )
        m = int(no[0])
        return Decisions(dg[:m], ds[:m], db[:m], dc[:m], dm[:m], dk[:m], status)

    def prepare(self, gidx, bnum, bcoord, first_slot):
        """PISM.handlePrepare for a batch of PREPAREs (acceptor side of a view change).
        Returns (r_bnum, r_bcoord, r_gc, r_flags, status) and the accepted pvalues as a list of
        (record index, slot, bnum, bcoord) rows sorted by (record, slot plane)."""
        gidx = _i32(gidx)
        n = gidx.shape[0]
        bnum, bcoord, first_slot = (_i32(x, n) for x in (bnum, bcoord, first_slot))
        W = int(self.cfg.window)
        m = max(n, 1)
        rb, rc, rg = (np.zeros(m, np.int32) for _ in range(3))
        rf, st = np.zeros(m, np.uint8), np.zeros(m, np.uint8)
        mask = np.zeros(m, np.uint64)
        ps, pb, pc = (np.zeros(m * W, np.int32) for _ in range(3))
        self.lib.check(
            self.lib.fn["prepare_batch"](self.h, n, _p(gidx), _p(bnum), _p(bcoord), _p(first_slot), _p(rb),
                                         _p(rc), _p(rg), _p(rf), _p(mask), _p(ps), _p(pb), _p(pc), _p(st)),
            "prepare_batch",
        )
        rows = []
        for w in range(W):
            sel = np.nonzero((mask[:n] >> np.uint64(w)) & np.uint64(1))[0]
            for i in sel:
                rows.append((int(i), int(ps[w * n + i]), int(pb[w * n + i]), int(pc[w * n + i])))
        rows.sort()
        return (rb[:n], rc[:n], rg[:n], rf[:n], st[:n]), rows

    def commit(self, gidx, bnum, bcoord, slot, median_cp, c_kind=None):
        """PISM.handleBatchedCommit/handleCommittedRequest for a batch of committed slots."""
        gidx = _i32(gidx)
        n = gidx.shape[0]
        bnum, bcoord, slot, median_cp = (_i32(x, n) for x in (bnum, bcoord, slot, median_cp))
        c_kind = _u8(c_kind, n)
        status = np.zeros(n, np.uint8)
        xg, xf, xc = (np.zeros(max(n, 1), np.int32) for _ in range(3))
        nr = np.zeros(1, np.int32)
        self.lib.check(
            self.lib.fn["commit_batch"](self.h, n, _p(gidx), _p(bnum), _p(bcoord), _p(slot),
                                        _p(median_cp), _p(c_kind), _p(status), _p(xg), _p(xf), _p(xc),
                                        _p(nr)),
            "commit_batch",
        )
        m = int(nr[0])
        return status, ExecRuns(xg[:m], xf[:m], xc[:m])
