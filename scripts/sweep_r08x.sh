#!/bin/bash
# round 6, last session: the scratch descriptor read from the kernarg segment (GPX_LAZY_X) in the secondary kernels
echo "#### --runs (sorted-runs path under the promise)"
SWEEP_REPS=2 bash scripts/sweep_variants.sh --runs
echo "#### partition front end (GPX_AR_TILES=0)"
SWEEP_ENV="GPX_AR_TILES=0" SWEEP_REPS=2 bash scripts/sweep_variants.sh
for rep in 1 2; do
for f in scratch/variants/*.so; do
  echo "== full round 1M  $f"; GPX_HIP_LIB=$PWD/$f python scripts/bench_full_round.py 2>&1 | tail -4 | cut -c1-400
  echo "== full round 10k $f"; GPX_HIP_LIB=$PWD/$f python scripts/bench_full_round.py --groups 10000 --rounds 101 2>&1 | tail -4 | cut -c1-400
done
done
