#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel trace and the two PMC passes.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh TAG [notest] [nopmc]'
# Everything lands under gpurun_out/TAG/ (merged back by gpurun); summaries are copied into
# profiles/ by hand afterwards (scripts/rocprof_summary.py output is already in the right form).
TAG=${1:-run}
shift
NOTEST=0
NOPMC=0
for a in "$@"; do
  [ "$a" = notest ] && NOTEST=1
  [ "$a" = nopmc ] && NOPMC=1
done
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
(java -version; javac -version; ant -version) >"$OUT/jdk_probe.txt" 2>&1
nproc >"$OUT/nproc.txt"
if [ $NOTEST = 0 ]; then
  timeout 900 python -m pytest tests -m gpu -x -q >"$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >>"$OUT/pytest_gpu.log"
  tail -5 "$OUT/pytest_gpu.log"
fi
timeout 600 python bench.py >"$OUT/bench.json" 2>"$OUT/bench.err"
echo "bench exit $?"
cat "$OUT/bench.json"
tail -3 "$OUT/bench.err"
# per-kernel traces are taken with --serial (no cross-call overlap: each kernel alone on the GPU, the
# regime bench.py's own hipEvent pass measures); the pipelined default is traced once as well
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-end-to-end"
BENCH_PIPE="python $REPO/bench.py --steps 10 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-end-to-end --pipelined"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_kt" -o kt -- $BENCH >"$OUT/prof_kt.log" 2>&1
echo "kt exit $?"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_kt_pipe" -o ktp -- $BENCH_PIPE >"$OUT/prof_kt_pipe.log" 2>&1
echo "kt pipelined exit $?"
if [ $NOPMC = 0 ]; then
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/prof_fetch" -o fetch -- $BENCH >"$OUT/prof_fetch.log" 2>&1
  echo "fetch exit $?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/prof_write" -o write -- $BENCH >"$OUT/prof_write.log" 2>&1
  echo "write exit $?"
fi
cd "$REPO"
KT=$(find "$OUT/prof_kt" -name '*_results.db' | head -1)
FE=$(find "$OUT/prof_fetch" -name '*_results.db' 2>/dev/null | head -1)
WR=$(find "$OUT/prof_write" -name '*_results.db' 2>/dev/null | head -1)
if [ -n "$KT" ] && [ -n "$FE" ] && [ -n "$WR" ]; then
  python scripts/rocprof_summary.py "$KT" "$FE" "$WR" "$OUT/rocprof_summary.txt" "$OUT/pmc_traffic.json" | head -60
elif [ -n "$KT" ]; then
  python scripts/rocprof_summary.py "$KT" "$KT" "$KT" "$OUT/rocprof_summary.txt" "$OUT/pmc_traffic.json" | head -30
fi
KTP=$(find "$OUT/prof_kt_pipe" -name '*_results.db' | head -1)
[ -n "$KTP" ] && python scripts/rocprof_summary.py "$KTP" "$KTP" "$KTP" "$OUT/rocprof_summary_pipelined.txt" /dev/null | head -12
# keep the merge-back small: drop the raw databases, keep csv/txt/json
find "$OUT" \( -name '*.db' -o -name '*.csv' -size +256k \) -delete
du -sh "$OUT"
