"""Read the per-phase cycle counters of an INSTRUMENTED build of k_pair_candidates (ab/instr.so, built ad hoc; not the product)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blaze_amd import _lib, synth
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from blaze_amd.engine import Engine
n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
full = synth.full_whitelist_codes(400_000, seed=1)
cells = synth.cell_whitelist_codes(5000, full, seed=2)
ext, el, bo, bl = synth.pack_ext(synth.make_rows(n_rows, cells, full, seed=3))
eng = Engine(0)
eng.set_whitelist_codes16(synth.codes_to_bytes(cells))
eng.stage(0, ext, el, bo, bl)
for _ in range(3): eng.run(0, 2)
eng.sync(0)
L = _lib.load()
out = (ctypes.c_ulonglong * 8)()
L.bz_debug_prof(out, 1)
eng.run(0, 2); eng.sync(0)
L.bz_debug_prof(out, 0)
v = list(out)
names = ["top clear+barrier", "probes", "pop+gather wait", "consume", "end barriers", "trip total", "waves"]
tot = v[5]
for n, x in zip(names, v):
    print(f"{n:20s} {x:14d}  {100.0 * x / max(tot, 1):6.1f} % of trip")
print("cycles per wave (all trips):", tot / max(v[6], 1))
