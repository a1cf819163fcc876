#!/usr/bin/env python
"""Does the HIP runtime report PAGEABLE host memory as pinned after it has copied from / into it itself?  (gpx_engine.hip's
mapped_host used to accept any memory for which hipPointerGetAttributes says "host": round 6's reading of the aborts.)
For a fresh numpy array, before and after a large pageable torch copy, and after freeing torch's tensor:
hipPointerGetAttributes' answer and whether hipHostGetDevicePointer hands out a device address for it."""
import ctypes as C

import numpy as np
import torch

hip = C.CDLL("libamdhip64.so")


class Attr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p),
                ("isManaged", C.c_int), ("allocationFlags", C.c_uint), ("pad", C.c_char * 64)]


def look(tag, a):
    at = Attr()
    rc = hip.hipPointerGetAttributes(C.byref(at), C.c_void_p(a.ctypes.data))
    d = C.c_void_p(0)
    rc2 = hip.hipHostGetDevicePointer(C.byref(d), C.c_void_p(a.ctypes.data), 0)
    hip.hipGetLastError()
    print(f"  {tag:44s} hipPointerGetAttributes rc={rc} type={at.type if rc == 0 else '-'}   hipHostGetDevicePointer rc={rc2} "
          f"dev={'%#x' % d.value if d.value else None}")


dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
for mb in (0.25, 4, 32):
    a = np.arange(int(mb * (1 << 20)) // 4, dtype=np.int32)
    print(f"{mb} MB numpy array at {a.ctypes.data:#x}")
    look("fresh", a)
    t = torch.from_numpy(a).to(dev)
    look("after a pageable host -> device copy", a)
    b = np.empty_like(a)
    tb = torch.from_numpy(b)
    tb.copy_(t)
    look("destination of a pageable device -> host copy", b)
    torch.cuda.synchronize()
    look("... after torch.cuda.synchronize()", b)
    del t
    for _ in range(4):
        torch.from_numpy(np.arange(3_000_000, dtype=np.int32)).to(dev)
    look("source, four other large copies later", a)
