"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a small markdown summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for fn in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(fn, newline="") as f:
            yield from csv.DictReader(f)


print(f"# rocprofv3 summary ({out})\n")
print("## kernel stats (--kernel-trace --stats)\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for r in rows("trace/**/*kernel_stats.csv"):
    name = r.get("Name", "")[:90]
    print(f"| {name} | {r.get('Calls')} | {float(r.get('TotalDurationNs', 0)) / 1e6:.3f} | "
          f"{float(r.get('AverageNs', 0)) / 1e3:.2f} | {float(r.get('MinNs', 0)) / 1e3:.2f} | "
          f"{float(r.get('MaxNs', 0)) / 1e3:.2f} | {r.get('Percentage')} |")

print("\n## counters (per-dispatch mean over dispatches of each kernel)\n")
acc = defaultdict(lambda: defaultdict(list))
extra = {}
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        k = r.get("Kernel_Name", "")[:70]
        acc[k][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
        extra[k] = (r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
per_dispatch = defaultdict(lambda: defaultdict(dict))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        if "k_scan_bitparallel" in r.get("Kernel_Name", ""):
            per_dispatch[sub][r.get("Dispatch_Id")][r.get("Counter_Name")] = float(r.get("Counter_Value", 0))
for k, cs in acc.items():
    print(f"### {k}\n")
    print(f"VGPR {extra[k][0]}, SGPR {extra[k][1]}, LDS {extra[k][2]}, grid {extra[k][3]}, workgroup {extra[k][4]}\n")
    print("| counter | mean per dispatch | dispatches |")
    print("|---|---|---|")
    for c, v in sorted(cs.items()):
        print(f"| {c} | {sum(v) / len(v):.4g} | {len(v)} |")
    print()

print("## k_scan_bitparallel, every dispatch (shortcut-on steps score ~29 % of the rows, the roofline leg scores every non-blank row)\n")
for sub, d in per_dispatch.items():
    names = sorted({c for v in d.values() for c in v})
    print(f"### {sub}\n")
    print("| dispatch | " + " | ".join(names) + " |")
    print("|---|" + "---|" * len(names))
    for disp in sorted(d, key=lambda x: int(x)):
        print(f"| {disp} | " + " | ".join(f"{d[disp].get(c, float('nan')):.4g}" for c in names) + " |")
    print()
