#!/bin/bash
# How hipcc compiles the three ways to write wire_stage's chunk loop (no GPU needed): scratch instructions and
# vmcnt waits per variant.  0 = bounds-tested assignment only (as shipped), 1 = zeroed first (INFLIGHT),
# 2 = unconditional loads from a clamped index.  See profiles/r03_wire_stage_isa.txt.
set -e
cd "$(dirname "$0")"
out=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -S --cuda-device-only -o "$out/t.s" stage_codegen.hip
for v in 0 1 2; do
  awk -v v="$v" '$0 ~ "^_Z1kILi" v {p=1} p && /scratch_/ {s++} p && /s_waitcnt vmcnt/ {w++} p && /global_load|flat_load/ {l++}
       p && /s_endpgm/ {printf "variant %s: %d loads, %d vmcnt waits, %d scratch instructions\n", v, l, w, s; exit}' "$out/t.s"
done
rm -rf "$out"
