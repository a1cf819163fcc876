#!/bin/bash
# round 6, last session: ordered proposals, records per lane of k_propose_one (1 = GPX_PROPOSE_TWO_MIN out of reach)
line() { python bench.py --no-cpu-baseline --no-end-to-end --steps 20 "$@" 2>/dev/null | python scripts/bench_line.py "[$TAG $*]"; }
for rep in 1 2 3; do
  TAG="R=1" GPX_PROPOSE_TWO_MIN=2000000000 line
  TAG="R=2" line
  TAG="R=4" GPX_PROPOSE_R=4 line
done
for rep in 1 2; do
  TAG="R=1" GPX_PROPOSE_TWO_MIN=2000000000 line --k 5
  TAG="R=2" line --k 5
  TAG="R=1" GPX_PROPOSE_TWO_MIN=2000000000 line --groups 500000
  TAG="R=2" line --groups 500000
  TAG="R=4" GPX_PROPOSE_R=4 line --groups 500000
  TAG="R=1" GPX_PROPOSE_TWO_MIN=2000000000 line --groups 125000 --k 5
  TAG="R=2" GPX_PROPOSE_TWO_MIN=1 line --groups 125000 --k 5
done
