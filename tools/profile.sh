#!/bin/bash
# rocprofv3 evidence for bench.py's numbers.  Run on the GPU box:  bash tools/profile.sh r01
# Each pass is its own short run of the SAME command; counters never share a run with API tracing.
# Writes gpurun_out/prof_<tag>/{summary.md,pmc.json}; copy both into profiles/ to have them judged.
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-two-slots --no-high-error --no-files"
pass() {  # name, rocprofv3 options...
    local name=$1; shift
    timeout -s KILL 300 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" \
        || { echo "$name pass failed"; tail -5 "$OUT/$name.err"; exit 1; }
}
pass trace --kernel-trace --stats
pass pmc_sq --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS
pass pmc_fetch --pmc FETCH_SIZE GRBM_GUI_ACTIVE
pass pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# lanes: thread-cycles of VALU work over wave-level VALU cycles x 64 = the share of the issued lane slots that held an
# ACTIVE lane (rocprofv3's derived VALUUtilization is this ratio); SQ_INSTS_VALU alone counts masked-off lanes as work
pass pmc_lanes --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.md"
cat "$OUT/summary.md"
