"""Step time when a few rows are irregular (N / lower case / short barcode: the byte-wise scan's rows), 1M rows x 5,000 cells."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from blaze_amd import synth, _lib
from blaze_amd.engine import Engine
full = synth.full_whitelist_codes(400_000, seed=1)
cells = synth.cell_whitelist_codes(5000, full, seed=2)
ext0, el0, bo0, bl0 = synth.pack_ext(synth.make_rows(1_000_000, cells, full, seed=3), stride=24)
eng = Engine(0)
eng.set_whitelist_codes16(synth.codes_to_bytes(cells))
cand = np.flatnonzero(bl0 == 16)
for n_irr in (0, 10, 100, 1000, 10000):
    ext, el, bo, bl = ext0.copy(), el0.copy(), bo0.copy(), bl0.copy()
    rows = cand[:: max(1, cand.size // max(n_irr, 1))][:n_irr]
    for k, r in enumerate(rows):
        if k % 3 == 0: ext[r, 5] = ord("N")
        elif k % 3 == 1: ext[r, 9] |= 0x20
        else: bl[r] = 15; el[r] = 19
    eng.stage(0, ext, el, bo, bl)
    for _ in range(2): eng.run(0, 2)
    eng.sync(0)
    reps = 20 if n_irr <= 100 else 5
    t0 = time.perf_counter()
    for _ in range(reps): eng.run(0, 2)
    eng.sync(0)
    dt = (time.perf_counter() - t0) / reps
    eng.set_option(_lib.BZ_OPT_PROFILE, 1); eng.run(0, 2); t = eng.timing(0); eng.set_option(_lib.BZ_OPT_PROFILE, 0)
    print(f"irregular rows {n_irr:6d}: {dt * 1e3:9.3f} ms/step (generic rows {t['rows_generic']}, generic stage {t['generic_ms']:.3f} ms)")
