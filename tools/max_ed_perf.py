"""Step time of the product path per --max-edit-distance (1M rows x 5,000 cells, rows resident, one slot)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from blaze_amd import synth
from blaze_amd.engine import Engine
full = synth.full_whitelist_codes(400_000, seed=1)
cells = synth.cell_whitelist_codes(5000, full, seed=2)
ext, el, bo, bl = synth.pack_ext(synth.make_rows(1_000_000, cells, full, seed=3))
eng = Engine(0); eng.set_whitelist_codes16(synth.codes_to_bytes(cells)); eng.stage(0, ext, el, bo, bl)
for k in (0, 1, 2, 3, 4, 5, 8):
    for _ in range(3 if k < 4 else 1): eng.run(0, k)
    eng.sync(0)
    n = 50 if k < 3 else 3
    t0 = time.perf_counter()
    for _ in range(n): eng.run(0, k)
    eng.sync(0)
    print("max_ed", k, "ms/step", (time.perf_counter() - t0) / n * 1e3, "assigned", int((eng.fetch(0)[0] >= 0).sum()))
