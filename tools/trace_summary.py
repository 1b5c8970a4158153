"""Per-dispatch kernel durations from a rocprofv3 --kernel-trace CSV directory: python tools/trace_summary.py DIR [first] [count]"""
import csv, glob, sys
fn = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = sorted(csv.DictReader(open(fn)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'probe' not in r['Kernel_Name']]
t0 = None
for r in rows[first:first + count]:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (st - t0) / 1e3 if t0 else 0
    t0 = en
    print(f"{n:32s} {(en - st) / 1e3:8.1f} us  gap {gap:7.1f}  grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>5s} vgpr {r['VGPR_Count']:>4s} lds {r['LDS_Block_Size']}")
