#!/bin/bash
# round 6, last session: the lazy-kernarg builds of the per-bucket and scatter kernels, interleaved
SWEEP_REPS=2 bash scripts/sweep_variants.sh
echo "#### K = 5"
bash scripts/sweep_variants.sh --k 5
echo "#### 125k x 5"
bash scripts/sweep_variants.sh --groups 125000 --k 5
