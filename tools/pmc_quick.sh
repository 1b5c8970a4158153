#!/bin/bash
# One SQ counter pass over tools/quick_perf.py; per-kernel means printed.  bash tools/pmc_quick.sh [rows] [cells]
set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmcq
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/sq" -- python3 tools/quick_perf.py ${1:-1000000} ${2:-5000} > "$OUT/run.txt" 2> "$OUT/err.txt" || { echo "pmc pass failed"; tail -5 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:24s} n={len(v):3d} mean={sum(v)/len(v):.4g} min={min(v):.4g} max={max(v):.4g}")
PY
