"""Step time with a MIXED whitelist (5,000 regular barcodes + a few irregular entries) against the all-regular list:
what k_mixed_fix + the re-scored rows cost (1M rows, rows resident, one slot)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
from blaze_amd import synth, _lib
from blaze_amd.engine import Engine
full = synth.full_whitelist_codes(400_000, seed=1)
cells = synth.cell_whitelist_codes(5000, full, seed=2)
ext, el, bo, bl = synth.pack_ext(synth.make_rows(1_000_000, cells, full, seed=3), stride=24)
wl = [w.tobytes() for w in synth.codes_to_bytes(cells)]
eng = Engine(0)
eng.set_option(_lib.BZ_OPT_PROFILE, 0)
for n_irr in (0, 1, 5, 50, 250, 2500):
    lst = list(wl)
    for k in range(n_irr):
        w = lst[(17 * k + 3) % len(lst)]
        lst.insert((19 * k) % len(lst), w[:15] if k % 3 == 0 else (w[:7] + b"N" + w[8:] if k % 3 == 1 else w + b"AC"))
    eng.set_whitelist(lst)
    eng.stage(0, ext, el, bo, bl)
    for _ in range(3): eng.run(0, 2)
    eng.sync(0)
    t0 = time.perf_counter()
    reps = 30 if n_irr <= 250 else 3
    for _ in range(reps): eng.run(0, 2)
    eng.sync(0)
    dt = (time.perf_counter() - t0) / reps
    eng.set_option(_lib.BZ_OPT_PROFILE, 1); eng.run(0, 2); t = eng.timing(0); eng.set_option(_lib.BZ_OPT_PROFILE, 0)
    print(f"irregular entries {n_irr:3d}: {dt * 1e3:.3f} ms/step, rows scored {t['rows_bitparallel']}, re-scored over the whole list {t['rows_mixed_fix']}, generic {t['rows_generic']}")
# a blank line of the list: '' is at distance 0 from every text, so EVERY scored row is re-scored by the workgroup-per-row kernel
lst = list(wl) + [b""]
eng.set_whitelist(lst)
eng.stage(0, ext, el, bo, bl)
eng.run(0, 2)
eng.sync(0)
t0 = time.perf_counter()
for _ in range(3): eng.run(0, 2)
eng.sync(0)
dt = (time.perf_counter() - t0) / 3
eng.set_option(_lib.BZ_OPT_PROFILE, 1); eng.run(0, 2); t = eng.timing(0); eng.set_option(_lib.BZ_OPT_PROFILE, 0)
print(f"a blank line ('' entry): {dt * 1e3:.3f} ms/step, rows scored {t['rows_bitparallel']}, re-scored over the whole list {t['rows_mixed_fix']}")
# a list without one regular entry (every barcode cut to 15 nt): every non-blank row through the workgroup-per-row scan
lst = [w[:15] for w in wl]
eng.set_whitelist(lst)
eng.stage(0, ext, el, bo, bl)
eng.run(0, 2)
eng.sync(0)
t0 = time.perf_counter()
for _ in range(3): eng.run(0, 2)
eng.sync(0)
dt = (time.perf_counter() - t0) / 3
eng.set_option(_lib.BZ_OPT_PROFILE, 1); eng.run(0, 2); t = eng.timing(0); eng.set_option(_lib.BZ_OPT_PROFILE, 0)
print(f"no regular entry (5,000 15-mers): {dt * 1e3:.3f} ms/step, rows through the workgroup-per-row scan {t['rows_generic']}")
