"""The C-ABI shared library builds for gfx950, loads without a GPU and exports every entry point
include/gpx.h declares (no compute calls here)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for hdr in ("gpx.h", "gpx_wire.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(gpx_[a-z_]+)\s*\(", src))
    return sorted(names)


def test_header_declares_expected_entry_points():
    names = _declared()
    for must in ("gpx_engine_create", "gpx_group_create", "gpx_group_retire", "gpx_propose_batch",
                 "gpx_accept_batch", "gpx_accept_reply_batch", "gpx_commit_batch",
                 "gpx_accept_reply_batch_dev", "gpx_names_bind", "gpx_wire_decode",
                 "gpx_wire_decode_dev", "gpx_wire_pack_commits", "gpx_wire_pack_commits_dev"):
        assert must in names


def test_hip_library_exports_every_declared_symbol():
    import ctypes

    import __graft_entry__ as ge

    ge.build()
    lib = ctypes.CDLL(ge.HIP_SO)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    lib.gpx_abi_version.restype = ctypes.c_int
    assert lib.gpx_abi_version() == 1


def test_one_hip_runtime_per_process_whatever_the_import_order():
    """build() (which loads the HIP library to check its symbols) followed by torch in the same process: the library must
    have bound to torch's libamdhip64, not pulled /opt/rocm's copy first - two HIP runtimes in one process ended with the
    engine reporting 'no ROCm-capable device' on the GPU box (round 6; gigapaxos_amd/_abi.py GpxLib)."""
    import subprocess
    import sys

    code = ("import os, re, __graft_entry__ as g; g.build(); import torch; "
            "m = open('/proc/self/maps').read(); "
            "print(len({os.path.realpath(p) for p in re.findall(r'/\\S*libamdhip64\\S*', m)}))")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "1", out.stdout


def test_oracle_exports_matching_symbols(oracle_lib):
    for name in _declared():
        if name.endswith(("_dev", "_async")) or name in ("gpx_engine_set_stream", "gpx_profile_enable", "gpx_profile_read",
                                                          "gpx_engine_wait", "gpx_engine_path_counters"):
            continue
        assert hasattr(oracle_lib.lib, "orc_" + name[4:]), name


def test_engine_create_rejects_bad_config_without_gpu():
    """Argument validation happens before any device call."""
    import ctypes

    import __graft_entry__ as ge
    from gigapaxos_amd._abi import GpxConfig

    ge.build()
    lib = ctypes.CDLL(ge.HIP_SO)
    h = ctypes.c_void_p()
    for bad in (GpxConfig(100, 0, 3, 8, 16, -1, 1, 0), GpxConfig(100, 8, 17, 8, 16, -1, 1, 0),
                GpxConfig(100, 8, 3, 6, 16, -1, 1, 0), GpxConfig(100, 8, 3, 128, 16, -1, 1, 0)):
        assert lib.gpx_engine_create(ctypes.byref(bad), ctypes.byref(h)) == -1
    assert lib.gpx_engine_create(None, ctypes.byref(h)) == -1


def test_jni_shim_compiles_against_the_headers(tmp_path):
    """gigapaxos_amd/jni/gpx_jni.c with GPX_HAVE_JNI, against a minimal stand-in of jni.h (the image has no
    JDK): every call in the shim matches its prototype in include/*.h (arity and pointer types; -Werror)."""
    import subprocess

    shim = os.path.join(ROOT, "gigapaxos_amd", "jni", "gpx_jni.c")
    subprocess.check_call(["gcc", "-fsyntax-only", "-Wall", "-Werror", "-DGPX_HAVE_JNI", "-I",
                           os.path.join(ROOT, "tests", "jni_stub"), "-I", os.path.join(ROOT, "include"), shim])
