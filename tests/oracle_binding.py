"""Binds oracle/libgpx_oracle.so (the CPU checker) through the same ctypes binder
the product uses, with the `orc_` symbol prefix.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

from gigapaxos_amd._abi import GpxLib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libgpx_oracle.so")

_lib = None


def build_oracle():
    deps = [os.path.join(ORACLE_DIR, "gpx_oracle.cpp"), os.path.join(ORACLE_DIR, "gpx_wire_oracle.inc"),
            os.path.join(ROOT, "include", "gpx.h"), os.path.join(ROOT, "include", "gpx_wire.h")]
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(d) for d in deps)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_SO


def load_oracle() -> GpxLib:
    global _lib
    if _lib is None:
        so = os.environ.get("GPX_ORACLE_SO")   # scripts/oracle_mutants.py: a deliberately broken oracle, to see the tests fail
        if not so:
            so = build_oracle()
        _lib = GpxLib(so, "orc_", device_api=False)
        L = _lib.lib
        L.orc_ballot_compare.argtypes = [C.c_int32] * 4
        L.orc_ballot_compare.restype = C.c_int
        L.orc_waitfor_trace.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
        L.orc_waitfor_trace.restype = C.c_int
        L.orc_median_minus.argtypes = [C.c_void_p, C.c_int32]
        L.orc_median_minus.restype = C.c_int32
        L.orc_round_robin_coordinator.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_int32]
        L.orc_round_robin_coordinator.restype = C.c_int32
    return _lib
