"""Ad-hoc perf probe (not part of bench.py): scan kernel timing at a few sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from blaze_amd import synth, _lib
from blaze_amd.engine import Engine, device_info

n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_wl = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
print(device_info(0))
full = synth.full_whitelist_codes(200_000, seed=1)
cells = synth.cell_whitelist_codes(n_wl, full, seed=2)
rows = synth.make_rows(n_rows, cells, full, seed=3)
ext, el, bo, bl = synth.pack_ext(rows)
eng = Engine(0)
ops, ms = eng.valu_peak_probe(20000)
print(f"valu probe: {ops/1e12:.2f} T lane-ops/s ({ms:.2f} ms)")
eng.set_whitelist_codes16(synth.codes_to_bytes(cells))
import os as _os
eng.set_option(_lib.BZ_OPT_PROFILE, int(_os.environ.get('BZ_PROFILE', '1')))
for shortcut, mode in ((1, 0), (0, 0), (1, 3), (0, 3)):
    eng.set_option(_lib.BZ_OPT_SHORTCUT, shortcut)
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, mode)
    eng.stage(0, ext, el, bo, bl)
    eng.run(0, 2); eng.sync(0)
    t0 = time.time()
    for rep in range(20):
        eng.run(0, 2)
    eng.sync(0)
    print(f"mode={mode} shortcut={shortcut}: 20 back-to-back runs {(time.time() - t0) / 20 * 1e3:.3f} ms each")
    for rep in range(2):
        t0 = time.time(); eng.run(0, 2); eng.sync(0); wall = time.time() - t0
        t = eng.timing(0)
        pairs = t["rows_bitparallel"] * n_wl
        print(f"mode={mode} shortcut={shortcut} wall {wall*1e3:.2f} ms total {t['total_ms']:.2f} classify {t['classify_ms']:.3f} scan {t['scan_ms']:.3f} (cand {t['candidates_ms']:.4f} verify {t['verify_ms']:.4f}) "
              f"gen {t['generic_ms']:.3f} hist {t['hist_ms']:.3f} scored {t['rows_bitparallel']} redo {t['rows_redo']} "
              f"-> {n_rows/ (t['total_ms']*1e-3)/1e6:.2f} M reads/s, {pairs/(t['scan_ms']*1e-3)/1e9:.1f} G pairs/s, "
              f"{pairs*340/(t['scan_ms']*1e-3)/1e12:.2f} T alg-ops/s")
