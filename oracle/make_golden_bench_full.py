#!/usr/bin/env python3
"""EVERY row of bench.py's own workload (BASELINE configs[1]: 1,000,000 synthetic 10x 3v3 rows against the 5,000-cell
whitelist, the seeds of bench.py rank 0) through the REAL `_match_bc_row` (/root/reference/blaze/read_assignment.py:35-99).
TEST INFRASTRUCTURE ONLY, build container only (oracle/shim stands in for the absent fast-edit-distance wheel).  The
whitelist is handed over as the ordered list bench.py uploads, so no set() order is involved.

One million "[bc, umi, strand]" triples are too much text for a fixture, so the file keeps, per 1000-row frame (the
reference's batch size), the sha256 of the frame's triples and how many of its rows got a barcode, plus the sha256 over all
frames: tests/golden/bench_full_<tie>.npz (data only, ~70 KB).  oracle/make_golden_bench_sample.py holds the first 50,000
rows in full (indices + UMIs), which is where a mismatch would be looked at.
Run:  python oracle/make_golden_bench_full.py [leftmost|rightmost]      (~20 minutes per tie rule on 8 cores)
"""
import hashlib
import os
import subprocess
import sys
from io import StringIO

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
N_ROWS, N_CELLS, FRAME = 1_000_000, 5000, 1000
N_DO = int(os.environ.get("BENCH_FULL_ROWS", N_ROWS))              # (a smaller number only to try the script out: nothing is written then)

if "ORACLE_TIE_RULE" not in os.environ:
    want = sys.argv[1:] or ["leftmost", "rightmost"]
    for tag in want:
        env = dict(os.environ, PYTHONHASHSEED="0", ORACLE_TIE_RULE={"leftmost": "0", "rightmost": "1"}[tag], MPLBACKEND="Agg",
                   PYTHONDONTWRITEBYTECODE="1")
        subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)

sys.path[:0] = [os.path.join(HERE, "shim"), REF, ROOT]
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import blaze.read_assignment as ra  # noqa: E402  (the real reference)
from blaze_amd import synth  # noqa: E402

_state = {}


def _rows_and_list():
    full = synth.full_whitelist_codes(400_000, seed=1)        # exactly bench.py's workload (rank 0)
    cells = synth.cell_whitelist_codes(N_CELLS, full, seed=2)
    rows = synth.make_rows(N_ROWS, cells, full, umi_len=12, seed=3)
    wl = [w.tobytes().decode() for w in synth.codes_to_bytes(cells)]
    return rows, wl


def _work(lo):
    if not _state:
        _state["rows"], _state["wl"] = _rows_and_list()
    rows, wl = _state["rows"], _state["wl"]
    hi = min(N_ROWS, lo + FRAME)
    body = synth.csv_text(rows, ["x"] * rows["n"], lo, hi)           # the reference's table text for these rows
    df = pd.read_csv(StringIO(body)).fillna("")
    res = [ra._match_bc_row(row, wl, 2, 0) for row in df.itertuples()]
    text = "".join(f"{r[0]},{r[1]},{r[2]}\n" for r in res)
    return hashlib.sha256(text.encode()).hexdigest(), sum(1 for r in res if r[0] and r[0] != "ambiguous")


def main():
    from multiprocessing import Pool
    tie = int(os.environ["ORACLE_TIE_RULE"])
    tag = "leftmost" if tie == 0 else "rightmost"
    with Pool(8) as pool:
        parts = pool.map(_work, range(0, N_DO, FRAME), chunksize=5)
    frame_sha = np.array([p[0] for p in parts], dtype="S64")
    frame_assigned = np.array([p[1] for p in parts], dtype=np.int32)
    sha = hashlib.sha256(b"".join(frame_sha)).hexdigest()
    if N_DO != N_ROWS:
        print(tag, "tried", N_DO, "rows, assigned", int(frame_assigned.sum()), frame_sha[0][:16])
        return
    out = os.path.join(ROOT, "tests", "golden", f"bench_full_{tag}.npz")
    np.savez_compressed(out, frame_sha=frame_sha, frame_assigned=frame_assigned, sha256=np.array(sha), n_rows=np.array(N_ROWS),
                        n_cells=np.array(N_CELLS), frame=np.array(FRAME))
    print(tag, "rows", N_ROWS, "assigned", int(frame_assigned.sum()), "sha256", sha[:16], os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
