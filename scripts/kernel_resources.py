#!/usr/bin/env python3
"""Static resource table of every kernel of the product library: compiles gigapaxos_amd/csrc/gpx_engine.hip with the
product's flags plus -Rpass-analysis=kernel-resource-usage (no GPU needed) and prints, per kernel: VGPRs, AGPRs, SGPRs,
scratch bytes per lane, spills, LDS bytes per block, waves per SIMD.  A kernel with scratch or spills is a finding."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gigapaxos_amd", "csrc")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
               "-Rpass-analysis=kernel-resource-usage", "-o", os.path.join(tmp, "x.so"), "gpx_engine.hip"] + sys.argv[1:]
        err = subprocess.run(cmd, cwd=CSRC, stderr=subprocess.PIPE, text=True, check=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]*\])?):\s*(\S+)", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            name = subprocess.run(["c++filt", "-p", val], stdout=subprocess.PIPE, text=True).stdout.strip() or val
            cur = {"name": name}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    cols = [("VGPRs", "VGPR"), ("AGPRs", "AGPR"), ("TotalSGPRs", "SGPR"), ("ScratchSize [bytes/lane]", "scratch"),
            ("VGPRs Spill", "vspill"), ("SGPRs Spill", "sspill"), ("LDS Size [bytes/block]", "LDS"), ("Occupancy [waves/SIMD]", "waves")]
    print(f"{'kernel':58s}" + "".join(f"{h:>8s}" for _, h in cols))
    bad = 0
    for r in sorted(rows, key=lambda r: r["name"]):
        print(f"{r['name'][:58]:58s}" + "".join(f"{r.get(k, '?'):>8s}" for k, _ in cols))
        bad += any(r.get(k, "0") != "0" for k in ("ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill"))
    print(f"# {len(rows)} kernels, {bad} with scratch or spills")


if __name__ == "__main__":
    main()
