#!/bin/bash
# PMC passes (one counter group per run; --pmc is never combined with other trace domains).
#   gpurun --timeout 900 -- 'bash scripts/gpu_pmc.sh TAG'
TAG=${1:-pmc}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# GPX_PMC_CMD: another command to count (e.g. "python $REPO/scripts/bench_wire.py --rounds 3")
BENCH=${GPX_PMC_CMD:-"python $REPO/bench.py --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-end-to-end"}
cd /tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i + 1))
  timeout 300 rocprofv3 --kernel-trace --pmc $group -d "$OUT/p$i" -o p$i -- $BENCH >"$OUT/p$i.log" 2>&1
  echo "pass $i ($group) exit $?"
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM
TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum TCC_WRITE_sum
TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum
GRBM_GUI_ACTIVE GRBM_TA_BUSY
GROUPS
cd "$REPO"
python scripts/pmc_table.py "$OUT/pmc_table.txt" $(find "$OUT" -name '*_results.db' | sort) | head -150
find "$OUT" \( -name '*.db' -o -name '*.csv' -size +256k \) -delete
