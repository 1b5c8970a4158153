"""BASELINE configs[3] on one GPU: rows vs the full-size 10x list stand-in (6,794,880 barcodes), pair-seeded vs brute force."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blaze_amd import synth, _lib
from blaze_amd.engine import Engine

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
full = synth.full_whitelist_codes(synth.FULL_WHITELIST_SIZE, seed=1)
cells = synth.cell_whitelist_codes(5000, full, seed=2)
rows = synth.make_rows(n_rows, cells, full, seed=3)
ext, el, bo, bl = synth.pack_ext(rows)
eng = Engine(0)
t0 = time.time(); eng.set_whitelist_codes16(synth.codes_to_bytes(full)); print(f"set_whitelist (host index build + upload): {time.time() - t0:.2f} s")
eng.set_option(_lib.BZ_OPT_PROFILE, 1)
ref = None
for mode, name in ((0, "pair-seeded (one kernel)"), (1, "brute force")):
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, mode)
    eng.stage(0, ext, el, bo, bl)
    for rep in range(2):
        eng.run(0, 2); eng.sync(0)
        t = eng.timing(0)
    out = eng.fetch(0)
    if ref is None: ref = out
    else: assert all(np.array_equal(a, b) for a, b in zip(ref, out)), "modes disagree"
    pairs = t["rows_bitparallel"] * full.size
    print("rows_redo", t["rows_redo"]); print(f"{name}: total {t['total_ms']:.2f} ms scan {t['scan_ms']:.2f} ms, {t['rows_bitparallel']} rows scored of {n_rows} -> "
          f"{n_rows / (t['total_ms'] * 1e-3) / 1e6:.2f} M reads/s, {pairs / (t['scan_ms'] * 1e-3) / 1e9:.0f} G pairs/s decided")
print("assigned", int((ref[0] >= 0).sum()), "of", n_rows)
