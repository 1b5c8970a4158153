#!/bin/bash
# rocprofv3 evidence for bench.py's numbers.  Run on the GPU box:  bash tools/profile.sh r01
# Each pass is its own short run; counters never share a run with API tracing.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { echo "trace pass failed"; tail -5 "$OUT/trace.err"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 $ARGS > /dev/null 2> "$OUT/pmc_sq.err" || { echo "sq pass failed"; tail -5 "$OUT/pmc_sq.err"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > /dev/null 2> "$OUT/pmc_fetch.err" || { echo "fetch pass failed"; tail -5 "$OUT/pmc_fetch.err"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > /dev/null 2> "$OUT/pmc_write.err" || { echo "write pass failed"; tail -5 "$OUT/pmc_write.err"; exit 1; }
find "$OUT" -name "*.csv" | head -30
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md"
cat "$OUT/summary.md"
