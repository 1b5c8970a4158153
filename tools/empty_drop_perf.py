"""SURVEY 8f N4 measurement: bz_far_from_all (candidates x cells global edit distance, cut-off 5) vs the oracle's C loop."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blaze_amd import synth, whitelisting
from blaze_amd.engine import Engine
from oracle import oracle as o

n_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_cells = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
full = synth.full_whitelist_codes(n_cand + n_cells, seed=1)
s = lambda m: [w.tobytes().decode() for w in synth.codes_to_bytes(m)]
cells, cand = s(full[:n_cells]), s(full[n_cells:])
eng = Engine(0)
whitelisting.far_from_all(eng, cand[:1000], cells)
t0 = time.time(); far = whitelisting.far_from_all(eng, cand, cells); dt = time.time() - t0
print(f"device: {n_cand} candidates x {n_cells} cells in {dt * 1e3:.1f} ms (host packing + copies included) = {n_cand * n_cells / dt / 1e9:.1f} G pairs/s; far: {int(far.sum())}")
m = 300
t0 = time.time(); ref = o.far_from_all(cand[:m], cells); dc = time.time() - t0
print(f"oracle (1 thread, C): {m} candidates in {dc:.2f} s = {m * n_cells / dc / 1e6:.1f} M pairs/s; agrees: {bool((ref == far[:m]).all())}")
t0 = time.time(); lst = whitelisting.empty_drop_list(cand, cells, eng); dt = time.time() - t0
print(f"empty_drop_list (stop after 2001): {dt * 1e3:.1f} ms, {len(lst)} barcodes")
