"""A/B step time of two builds of libblaze_hip.so on the SAME box (boxes differ by ~15 % in clock):
alternating child processes, each times the product step (rows resident, one slot) and its kernels.

usage: python tools/ab_perf.py libA.so libB.so [n_rows] [n_cells] [alternations]   (or libA.so,libB.so,libC.so,... as ONE argument)
child: python tools/ab_perf.py --child lib.so n_rows n_cells
"""
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(lib, n_rows, n_cells):
    import numpy as np  # noqa: F401
    from blaze_amd import _lib, synth
    _lib.LIB_PATH = os.path.abspath(lib)
    from blaze_amd.engine import Engine
    full = synth.full_whitelist_codes(400_000, seed=1)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=2)
    ext, el, bo, bl = synth.pack_ext(synth.make_rows(n_rows, cells, full, seed=3))
    eng = Engine(0)
    eng.set_whitelist_codes16(synth.codes_to_bytes(cells))
    eng.stage(0, ext, el, bo, bl)
    for _ in range(20):
        eng.run(0, 2)
    eng.sync(0)
    best = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200):
            eng.run(0, 2)
        eng.sync(0)
        best.append((time.perf_counter() - t0) / 200 * 1e3)
    eng.set_option(_lib.BZ_OPT_PROFILE, 1)
    ks = []
    for _ in range(10):
        eng.run(0, 2)
        eng.sync(0)
        t = eng.timing(0)
        ks.append((t["classify_ms"], t["candidates_ms"], t["verify_ms"], t["hist_ms"]))
    med = [statistics.median(k[i] for k in ks) * 1e3 for i in range(4)]
    print(json.dumps({"step_ms_min": min(best), "step_ms_med": statistics.median(best),
                      "classify_us": med[0], "candidates_us": med[1], "verify_us": med[2], "hist_us": med[3]}))


def main():
    if sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    if "," in sys.argv[1]:                                       # several builds: libA.so,libB.so,libC.so [n_rows] ...
        libs, rest = sys.argv[1].split(","), sys.argv[2:]
    else:
        libs, rest = sys.argv[1:3], sys.argv[3:]
    n_rows = rest[0] if len(rest) > 0 else "1000000"
    n_cells = rest[1] if len(rest) > 1 else "5000"
    alts = int(rest[2]) if len(rest) > 2 else 3
    res = {lib: [] for lib in libs}
    for _ in range(alts):
        for lib in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, n_rows, n_cells], capture_output=True,
                                 text=True, timeout=300)
            if out.returncode != 0:
                print(lib, "FAILED", out.stderr[-2000:])
                sys.exit(1)
            r = json.loads(out.stdout.strip().splitlines()[-1])
            res[lib].append(r)
            print(lib, json.dumps(r), flush=True)
    for lib in libs:
        keys = res[lib][0].keys()
        print(lib, {k: round(min(r[k] for r in res[lib]), 4) for k in keys})


if __name__ == "__main__":
    main()
