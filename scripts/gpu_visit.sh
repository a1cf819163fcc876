#!/bin/bash
# One GPU-box visit, assembled from named steps (replaces the one-off gpu_r2*.sh / gpu_r3*.sh family).
#   gpurun --timeout 1500 -- 'bash scripts/gpu_visit.sh TAG step [step ...]'
# Everything lands under gpurun_out/TAG/ (merged back by gpurun); summaries worth judging are copied
# into profiles/ by hand afterwards.  Steps (each independent; a failing step does not stop the rest):
#   probe                 JDK / ant / nproc / memory of the box
#   tests[:EXPR]          pytest -m gpu (whole suite, or -k EXPR; '+' stands for a space in EXPR)
#   testfile:PATH[:EXPR]  pytest PATH -m gpu [-k EXPR]
#   bench[:ARGS]          bench.py ARGS -> TAG/bench[_N].json  ('+' = space)
#   lines:N[:ARGS]        N condensed bench lines (kernel times only, no CPU / end-to-end legs)
#   kt[:ARGS]             rocprofv3 --kernel-trace --stats of bench.py ARGS
#   traffic[:ARGS]        two PMC passes (FETCH_SIZE, WRITE_SIZE) of bench.py ARGS + kernel trace -> summary
#   py:SCRIPT[:ARGS]      python scripts/SCRIPT ARGS -> TAG/SCRIPT[_N].out
#   pmc:SCRIPT[:ARGS[:LABEL]]  FETCH_SIZE / WRITE_SIZE / SQ / TCC passes of python scripts/SCRIPT ARGS -> pmc table
#   pmclite:SCRIPT[:ARGS] FETCH_SIZE / WRITE_SIZE passes only
#   ktpy:SCRIPT[:ARGS[:LABEL]]  rocprofv3 --kernel-trace --stats of python scripts/SCRIPT ARGS -> per-kernel summary
#   sh:SCRIPT[:ARGS]      bash scripts/SCRIPT ARGS
#   reffix                scripts/make_ref_fixtures.sh when a JDK exists (row c of SURVEY 8)
#   fast                  pytest -m gpu_fast (about a minute: after every kernel change)
#   smoke                 __graft_entry__.smoke()
#   dmesg[:N]             the kernel log's last N lines (is it readable on the box at all?)
# Round 4's visits were mostly of this shape (1-2 GPU-minutes each):
#   bash scripts/gpu_visit.sh r04 testfile:tests/test_small_ar_gpu.py sh:ubench/sar_trace.sh:run \
#        py:bench_batch_sweep.py:--min-log2+7+--max-log2+11 py:bench_full_round.py:--groups+10000+--rounds+101 fast
# (GPX_SAR_MAX_N=0 in front of it for the partition pipeline's side of a comparison)
TAG=${1:-visit}
shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
sp() { echo "${1//+/ }"; }
uniq_name() { # first free NAME[_N].EXT under $OUT
  local base=$1 ext=$2 n=0 f="$OUT/$1.$2"
  while [ -e "$f" ]; do n=$((n + 1)); f="$OUT/${base}_$n.$ext"; done
  echo "$f"
}
pmc_passes() { # $1 = label, $2 = command, $3... = counter groups
  local label=$1 cmd=$2 i=0
  shift 2
  cd /tmp
  for group in "$@"; do
    i=$((i + 1))
    timeout 400 rocprofv3 --kernel-trace --pmc $group -d "$OUT/${label}_p$i" -o p$i -- $cmd >"$OUT/${label}_p$i.log" 2>&1
    echo "  pmc pass $i ($group) exit $?"
  done
  cd "$REPO"
  python scripts/pmc_table.py "$OUT/${label}_table.txt" $(find "$OUT" -path "*${label}_p*" -name '*_results.db' | sort) | head -${PMC_HEAD:-120}
}
for step in "$@"; do
  IFS=: read -r kind a b c <<<"$step"
  echo "=== $step"
  case $kind in
  fast)
    f=$(uniq_name pytest_fast log)
    timeout 400 python -m pytest tests -m gpu_fast -q -x >"$f" 2>&1
    echo "pytest exit $?" >>"$f"
    tail -4 "$f" | cut -c1-240
    ;;
  smoke)
    timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
    ;;
  probe)
    (java -version; javac -version; ant -version) >"$OUT/jdk_probe.txt" 2>&1
    (nproc; free -g | head -2; rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -2) >"$OUT/box.txt" 2>&1
    cat "$OUT/jdk_probe.txt" "$OUT/box.txt" | head -12
    ;;
  tests)
    f=$(uniq_name pytest_gpu log)
    if [ -n "$a" ]; then
      timeout 1500 python -m pytest tests -m gpu -q --durations=8 -k "$(sp "$a")" >"$f" 2>&1
    else
      timeout 1500 python -m pytest tests -m gpu -q --durations=8 >"$f" 2>&1
    fi
    rc=$?
    echo "pytest exit $rc" >>"$f"
    # a GPU page fault kills the process (ROCr's VMFaultHandler aborts): the kernel log names the faulting client and address
    [ $rc -ge 128 ] && (dmesg 2>&1 | tail -120 >"$OUT/dmesg_after_abort.txt"; grep -i -E "fault|amdgpu" "$OUT/dmesg_after_abort.txt" | tail -20)
    tail -14 "$f" | cut -c1-240
    ;;
  dmesg)
    dmesg 2>&1 | tail -${a:-60} >"$OUT/dmesg.txt"
    tail -12 "$OUT/dmesg.txt" | cut -c1-200
    ;;
  testfile)
    f=$(uniq_name pytest_file log)
    if [ -n "$b" ]; then
      timeout 1500 python -m pytest "$a" -m gpu -q --durations=8 -k "$(sp "$b")" >"$f" 2>&1
    else
      timeout 1500 python -m pytest "$a" -m gpu -q --durations=8 >"$f" 2>&1
    fi
    echo "pytest exit $?" >>"$f"
    tail -14 "$f" | cut -c1-240
    ;;
  bench)
    f=$(uniq_name bench json)
    timeout 600 python bench.py $(sp "$a") >"$f" 2>"${f%.json}.err"
    echo "bench exit $?"
    cat "$f"
    tail -3 "${f%.json}.err"
    ;;
  lines)
    for i in $(seq 1 "${a:-1}"); do
      timeout 300 python bench.py $(sp "$b") --no-cpu-baseline --no-end-to-end 2>/dev/null | python scripts/bench_line.py "[$(sp "$b")]"
    done | tee -a "$OUT/lines.txt"
    ;;
  kt)
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_kt" -o kt -- python $REPO/bench.py --steps 10 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-end-to-end $(sp "$a") >"$OUT/prof_kt.log" 2>&1
    echo "kt exit $?"
    cd "$REPO"
    KT=$(find "$OUT/prof_kt" -name '*_results.db' | head -1)
    [ -n "$KT" ] && python scripts/rocprof_summary.py "$KT" "$KT" "$KT" "$OUT/rocprof_summary.txt" /dev/null | head -30
    ;;
  traffic)
    BENCH="python $REPO/bench.py --steps 10 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-end-to-end $(sp "$a")"
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_kt" -o kt -- $BENCH >"$OUT/prof_kt.log" 2>&1
    echo "kt exit $?"
    timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- $BENCH >"$OUT/prof_fetch.log" 2>&1
    echo "fetch exit $?"
    timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- $BENCH >"$OUT/prof_write.log" 2>&1
    echo "write exit $?"
    cd "$REPO"
    KT=$(find "$OUT/prof_kt" -name '*_results.db' | head -1)
    FE=$(find "$OUT/prof_fetch" -name '*_results.db' | head -1)
    WR=$(find "$OUT/prof_write" -name '*_results.db' | head -1)
    [ -n "$KT" ] && [ -n "$FE" ] && [ -n "$WR" ] &&
      python scripts/rocprof_summary.py "$KT" "$FE" "$WR" "$OUT/rocprof_summary.txt" "$OUT/pmc_traffic.json" | head -60
    ;;
  py)
    f=$(uniq_name "${a%.py}" out)
    timeout 900 python "scripts/$a" $(sp "$b") >"$f" 2>"${f%.out}.err"
    echo "$a exit $?"
    tail -c 3000 "$f"
    tail -3 "${f%.out}.err"
    ;;
  pmc)
    pmc_passes "pmc_${c:-${a%.py}}" "python $REPO/scripts/$a $(sp "$b")" FETCH_SIZE WRITE_SIZE \
      "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
      "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" \
      "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
      "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
    ;;
  pmclite)
    pmc_passes "pmc_${c:-${a%.py}}" "python $REPO/scripts/$a $(sp "$b")" FETCH_SIZE WRITE_SIZE
    ;;
  ktpy)
    L=${c:-${a%.py}}
    cd /tmp
    timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/kt_$L" -o kt -- python $REPO/scripts/$a $(sp "$b") >"$OUT/kt_$L.log" 2>&1
    echo "kt exit $?"
    cd "$REPO"
    KT=$(find "$OUT/kt_$L" -name '*_results.db' | head -1)
    [ -n "$KT" ] && python scripts/rocprof_summary.py "$KT" "$KT" "$KT" "$OUT/kt_${L}_summary.txt" /dev/null | head -40
    ;;
  sh)
    f=$(uniq_name "$(basename "${a%.sh}")" out)
    timeout 1200 bash "scripts/$a" $(sp "$b") >"$f" 2>&1
    echo "$a exit $?"
    tail -25 "$f" | cut -c1-300
    ;;
  reffix)
    if command -v javac >/dev/null 2>&1; then
      bash scripts/make_ref_fixtures.sh >"$OUT/reffix.log" 2>&1
      echo "make_ref_fixtures exit $?"
      mkdir -p "$OUT/golden" && cp tests/golden/ref_*.npz "$OUT/golden/" 2>/dev/null
    else
      echo "no javac on this box: reference fixtures not generated"
    fi
    ;;
  *) echo "unknown step $step" ;;
  esac
done
# keep the merge-back small: drop the raw databases, keep csv/txt/json
find "$OUT" \( -name '*.db' -o -name '*.csv' -size +256k \) -delete
du -sh "$OUT"
