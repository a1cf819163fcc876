#!/usr/bin/env python3
"""Static instruction mix of named kernels in a device assembly file (hipcc -S --cuda-device-only):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -w -o /tmp/e.s gigapaxos_amd/csrc/gpx_engine.hip
    python scripts/ubench/isa_mix.py /tmp/e.s k_bucket_ar16_tiles_k4 k_scatter_tiles
Counts what the SGPR spills cost (v_readlane / v_writelane), the 64-bit address arithmetic, scalar loads, s_nop."""
import re
import sys
from collections import Counter


def main():
    text = open(sys.argv[1]).read()
    for want in sys.argv[2:]:
        for m in re.finditer(r"^(_Z\S*" + re.escape(want) + r"\S*): +; @.*?\n(.*?)\n\ts_endpgm", text, re.S | re.M):
            ins = [ln.split()[0] for ln in m.group(2).split("\n") if ln.startswith("\t") and not ln.lstrip().startswith((".", ";"))]
            c = Counter(ins)
            valu = sum(v for k, v in c.items() if k.startswith("v_"))
            salu = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith(("s_load", "s_waitcnt", "s_nop", "s_barrier", "s_cbranch", "s_branch")))
            print(f"{m.group(1)[:70]:70s} total {len(ins):5d}  valu {valu:5d}  salu {salu:5d}  readlane {c['v_readlane_b32']:4d}  "
                  f"writelane {c['v_writelane_b32']:4d}  lshl_add_u64 {c['v_lshl_add_u64']:4d}  ashr {c['v_ashrrev_i32']:4d}  "
                  f"s_load {sum(v for k, v in c.items() if k.startswith('s_load')):3d}  s_nop {c['s_nop']:4d}  "
                  f"global_load {sum(v for k, v in c.items() if k.startswith('global_load')):3d}  global_store {sum(v for k, v in c.items() if k.startswith('global_store')):3d}  "
                  f"ds {sum(v for k, v in c.items() if k.startswith('ds_')):4d}")


if __name__ == "__main__":
    main()
