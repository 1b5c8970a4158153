#!/bin/bash
# One counter pass over tools/quick_perf.py (product + all-rows legs), per-kernel means printed.
#   bash tools/pmc_pass.sh "SQ_WAVES SQ_INSTS_VALU ..." [kernel-name filter]
set -u
export TMPDIR=/tmp
OUT=gpurun_out/pmcp
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -s KILL 200 rocprofv3 --pmc $1 --output-format csv -d "$OUT/p" -- python3 tools/quick_perf.py 1000000 5000 > "$OUT/run.txt" 2> "$OUT/err.txt" || { echo "pmc pass failed"; tail -5 "$OUT/err.txt"; exit 1; }
python3 - "$OUT" "${2:-}" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        acc[r["Kernel_Name"].replace("void ", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if sys.argv[2] and sys.argv[2] not in k:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:30s} n={len(v):3d} mean={sum(v)/len(v):.4g} min={min(v):.4g} max={max(v):.4g}")
PY
