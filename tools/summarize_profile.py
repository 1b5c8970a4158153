"""Condense rocprofv3 CSV output (kernel stats + three PMC passes of `bench.py --steps 5`) into
summary.md (stdout) and pmc.json (per kernel and per dispatch class), the file bench.py reads its
`traffic` and `issued` figures from.

Dispatch classes of a scan kernel: "product" = the timed steps (exact-match shortcut on, ~29 % of
the rows are scored), "all_rows" = the roofline leg (shortcut off, every non-blank row scored).
They are told apart by SQ_INSTS_VALU / duration: the all-rows dispatches are ~3x larger."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for fn in sorted(glob.glob(os.path.join(out, pattern), recursive=True)):
        with open(fn, newline="") as f:
            yield from csv.DictReader(f)


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


print(f"# rocprofv3 summary ({out})\n")
print("Command of every pass: `python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-two-slots --no-high-error --no-files` (1M rows x 5,000 cells).\n")
print("## kernel stats (--kernel-trace --stats)\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
stats = {}
for r in rows("trace/**/*kernel_stats.csv"):
    name = short(r.get("Name", ""))
    stats[name] = dict(calls=int(r.get("Calls", 0)), avg_us=float(r.get("AverageNs", 0)) / 1e3,
                       min_us=float(r.get("MinNs", 0)) / 1e3, max_us=float(r.get("MaxNs", 0)) / 1e3)
    print(f"| {name} | {r.get('Calls')} | {float(r.get('TotalDurationNs', 0)) / 1e6:.3f} | "
          f"{float(r.get('AverageNs', 0)) / 1e3:.2f} | {float(r.get('MinNs', 0)) / 1e3:.2f} | "
          f"{float(r.get('MaxNs', 0)) / 1e3:.2f} | {r.get('Percentage')} |")

# per-dispatch durations from the kernel trace, in dispatch order per kernel
durs = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    durs[short(r.get("Kernel_Name", ""))].append((int(r.get("Start_Timestamp", 0)),
                                                  (int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0))) / 1e3))
for k in durs:
    durs[k] = [d for _, d in sorted(durs[k])]

# counters: kernel -> [dispatch -> {counter: value}] in dispatch order
pmc = defaultdict(lambda: defaultdict(dict))
meta = {}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_lanes"):
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        k = short(r.get("Kernel_Name", ""))
        pmc[k][(sub, int(r.get("Dispatch_Id")))][r.get("Counter_Name")] = float(r.get("Counter_Value", 0))
        meta[k] = dict(vgpr=r.get("VGPR_Count"), sgpr=r.get("SGPR_Count"), lds=r.get("LDS_Block_Size"),
                       grid=r.get("Grid_Size"), workgroup=r.get("Workgroup_Size"))

import hashlib
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "blaze_amd", "lib", "libblaze_hip.so")
with open(LIB, "rb") as f:
    lib_sha = hashlib.sha256(f.read()).hexdigest()
# the library these counters were taken on: bench.py joins them with LIVE timings only while the hash still matches
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blaze_amd import _lib as _bz_lib  # noqa: E402
src_sha = _bz_lib.source_sha256()
result = {"source": out, "library_sha256": lib_sha, "source_sha256": src_sha, "kernels": {}}
try:                                                 # the workload the passes ran (bench.py's own line of the trace pass)
    with open(os.path.join(out, "trace.json")) as f:
        line = [ln for ln in f if ln.startswith("{")][-1]
    cfg = json.loads(line)["config"]
    result["workload"] = {"rows": cfg["rows_per_gpu"], "cells": cfg["cells"], "max_ed": int(cfg["workload"].split("max_ed ")[1].split(",")[0])}
except Exception as e:                               # noqa: BLE001
    print(f"(no workload recorded: {e})\n")
print("\n## counters per kernel (mean per dispatch; scan kernels split into product / all_rows dispatches)\n")
for k in sorted(pmc):
    per_sub = defaultdict(list)                      # sub -> [counters dict] in dispatch order
    for (sub, disp) in sorted(pmc[k]):
        per_sub[sub].append(pmc[k][(sub, disp)])
    n_disp = max(len(v) for v in per_sub.values())
    # classes by position: the same command runs in every pass, so dispatch i of each pass is the same launch
    d = durs.get(k, [])
    classes = {"all": list(range(n_disp))}
    if ("scan" in k or "k_pair" in k) and len(d) == n_disp and n_disp >= 4:
        big = max(d)
        classes = {"product": [i for i in range(n_disp) if d[i] < 0.6 * big],
                   "all_rows": [i for i in range(n_disp) if d[i] >= 0.6 * big]}
        if not classes["product"] or not classes["all_rows"]:
            classes = {"all": list(range(n_disp))}
    entry = {"meta": meta.get(k), "classes": {}}
    print(f"### {k}\n")
    m = meta.get(k, {})
    print(f"VGPR {m.get('vgpr')}, SGPR {m.get('sgpr')}, LDS {m.get('lds')}, grid {m.get('grid')}, workgroup {m.get('workgroup')}\n")
    for cname, idx in classes.items():
        c = {"dispatches": len(idx)}
        if d and len(d) == n_disp:
            c["avg_us"] = sum(d[i] for i in idx) / len(idx)
        for sub, lst in per_sub.items():
            if len(lst) != n_disp:
                continue
            names = sorted({n for x in lst for n in x})
            for n in names:
                vals = [lst[i].get(n) for i in idx if lst[i].get(n) is not None]
                if vals:
                    c[n if sub != "pmc_lanes" else "LANES_" + n] = sum(vals) / len(vals)
        if c.get("LANES_SQ_ACTIVE_INST_VALU"):
            # = rocprofv3's VALUUtilization / 100: active lanes per issued VALU lane slot
            c["lane_utilisation"] = c["LANES_SQ_THREAD_CYCLES_VALU"] / (c["LANES_SQ_ACTIVE_INST_VALU"] * 64.0)
        entry["classes"][cname] = c
        print(f"**{cname}** ({len(idx)} dispatches" + (f", avg {c['avg_us']:.1f} us" if "avg_us" in c else "") + ")\n")
        print("| counter | mean per dispatch |")
        print("|---|---|")
        for n, v in sorted(c.items()):
            if n not in ("dispatches", "avg_us"):
                print(f"| {n} | {v:.4g} |")
        print()
    result["kernels"][k] = entry

with open(os.path.join(out, "pmc.json"), "w") as f:
    json.dump(result, f, indent=1, sort_keys=True)
print(f"Library profiled: sources sha256 {src_sha} (blaze_amd/csrc + include/blaze_hip.h), binary sha256 {lib_sha}.  lane_utilisation = SQ_THREAD_CYCLES_VALU / "
      "(SQ_ACTIVE_INST_VALU x 64) of the pmc_lanes pass (rocprofv3's VALUUtilization): the share of issued VALU lane slots with an active lane.\n")
print("Units: FETCH_SIZE / WRITE_SIZE are KiB as rocprofv3 reports them (gfx950: FETCH_SIZE counts a wide "
      "coalesced read at half its bytes, MI355X_MICROARCH.md 'HBM'); SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* are "
      "quad-cycles summed over waves; SQ_INSTS_* are wave-instructions; GRBM_GUI_ACTIVE sums the 8 XCDs.")
