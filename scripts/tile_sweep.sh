#!/bin/bash
# Tiled front end (gpx_tiles.hip.h): votes / threads per scatter workgroup and the bucket -> XCD mapping, per call shape.
#   bash scripts/tile_sweep.sh ['bench args of one shape' ...]   (on the GPU box; one condensed bench line per variant)
line() { # $1 = tag, rest = bench args; environment of the caller
  local tag=$1
  shift
  timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-end-to-end "$@" 2>/dev/null | python scripts/bench_line.py "[$tag $*]"
}
if [ $# -gt 0 ]; then SHAPES=("$@"); else SHAPES=("" "--k 5" "--sorted" "--mix" "--groups 125000 --k 5" "--groups 500000"); fi
for shape in "${SHAPES[@]}"; do
  line "auto" $shape
  GPX_AR_TILES=0 line "partition" $shape
  for t in "16384 1024" "12288 1024" "8192 1024" "8192 512" "4096 512"; do
    set -- $t
    GPX_TILE_T=$1 GPX_TILE_NT=$2 line "T=$1 NT=$2" $shape
  done
  GPX_TILE_XCD_ROWS=0 line "auto, buckets in dispatch order" $shape
done
