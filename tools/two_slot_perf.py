"""Ad-hoc probe: K steps on one slot vs the same K steps alternating over both slots (two streams,
the engine's double buffering) with the rows resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blaze_amd import synth
from blaze_amd.engine import Engine

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_wl = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
full = synth.full_whitelist_codes(400_000, seed=1)
cells = synth.cell_whitelist_codes(n_wl, full, seed=2)
rows = synth.make_rows(n_rows, cells, full, seed=3)
ext, el, bo, bl = synth.pack_ext(rows)
eng = Engine(0)
eng.set_whitelist_codes16(synth.codes_to_bytes(cells))
eng.stage(0, ext, el, bo, bl)
eng.stage(1, ext, el, bo, bl)
for s in (0, 1):
    eng.run(s, 2); eng.sync(s)
K = 40
for label, slots in (("one slot", (0,)), ("two slots", (0, 1)), ("one slot", (0,)), ("two slots", (0, 1))):
    t0 = time.perf_counter()
    for k in range(K):
        eng.run(slots[k % len(slots)], 2)
    eng.sync(0); eng.sync(1)
    dt = (time.perf_counter() - t0) / K
    print(f"{label}: {dt * 1e3:.4f} ms per step, {n_rows / dt / 1e9:.2f} G reads/s")
