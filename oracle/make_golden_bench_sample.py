#!/usr/bin/env python3
"""The first 50,000 rows of bench.py's own workload (BASELINE configs[1]: synthetic 10x 3v3 rows against the 5,000-cell
whitelist, the seeds of bench.py rank 0) through the REAL `_match_bc_row` (/root/reference/blaze/read_assignment.py:35-99),
one file per end-index tie rule.  TEST INFRASTRUCTURE ONLY, build container only (oracle/shim stands in for the absent
fast-edit-distance wheel).  The whitelist is handed over as the ordered list bench.py uploads, so no set() order is
involved.  Written: tests/golden/bench_sample_<tie>.npz = per row the assigned barcode's index in that list (-1: none)
and the length-prefixed UMI the reference returned, + a sha256 over all "[bc, umi, strand]" triples (data only).
Run:  python oracle/make_golden_bench_sample.py      (~1 minute on 8 cores)
"""
import hashlib
import os
import subprocess
import sys
from io import StringIO

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
N_ROWS, N_CELLS = 50_000, 5000

if "ORACLE_TIE_RULE" not in os.environ:
    for tie in ("0", "1"):
        env = dict(os.environ, PYTHONHASHSEED="0", ORACLE_TIE_RULE=tie, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
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
    rows = synth.make_rows(1_000_000, cells, full, umi_len=12, seed=3)
    wl = [w.tobytes().decode() for w in synth.codes_to_bytes(cells)]
    return rows, wl


def _work(span):
    lo, hi = span[:2]
    max_ed = span[2] if len(span) > 2 else 2
    if not _state:
        _state["rows"], _state["wl"] = _rows_and_list()
    rows, wl = _state["rows"], _state["wl"]
    ids = [f"r{i}" for i in range(lo, hi)]
    body = synth.csv_text(rows, ["x"] * rows["n"], lo, hi)           # the reference's table text for these rows
    df = pd.read_csv(StringIO(body)).fillna("")
    return [ra._match_bc_row(row, wl, max_ed, 0) for row in df.itertuples()]


def main():
    from multiprocessing import Pool
    tie = int(os.environ["ORACLE_TIE_RULE"])
    tag = "leftmost" if tie == 0 else "rightmost"
    spans = [(lo, min(N_ROWS, lo + 1000)) for lo in range(0, N_ROWS, 1000)]   # 1000-row frames, as the reference batches them
    with Pool(8) as pool:
        parts = pool.map(_work, spans)
    res = [r for p in parts for r in p]
    _, wl = _rows_and_list()
    pos = {w: k for k, w in enumerate(wl)}
    hit = np.array([pos.get(r[0], -1) if r[0] else -1 for r in res], dtype=np.int32)
    umi = np.array([r[1] for r in res], dtype="S16")
    strand = np.array([r[2] for r in res], dtype="S1")
    sha = hashlib.sha256("".join(f"{r[0]},{r[1]},{r[2]}\n" for r in res).encode()).hexdigest()
    out = os.path.join(ROOT, "tests", "golden", f"bench_sample_{tag}.npz")
    np.savez_compressed(out, hit=hit, umi=umi, strand=strand, sha256=np.array(sha), n_rows=np.array(N_ROWS), n_cells=np.array(N_CELLS))
    print(tag, "rows", len(res), "assigned", int((hit >= 0).sum()), "sha256", sha[:16], os.path.getsize(out) // 1024, "KiB")
    if tie == 0:
        # the same workload at the other --max-edit-distance settings (parser.py:100), first 20,000 rows each
        arrays = {}
        n_other = 20_000
        for max_ed in (0, 1, 3, 4, 6):
            with Pool(8) as pool:
                parts = pool.map(_work, [(lo, min(n_other, lo + 1000), max_ed) for lo in range(0, n_other, 1000)])
            res = [r for p in parts for r in p]
            arrays[f"ed{max_ed}_hit"] = np.array([pos.get(r[0], -1) if r[0] else -1 for r in res], dtype=np.int32)
            arrays[f"ed{max_ed}_sha256"] = np.array(hashlib.sha256("".join(f"{r[0]},{r[1]},{r[2]}\n" for r in res).encode()).hexdigest())
            print("max_ed", max_ed, "rows", len(res), "assigned", int((arrays[f"ed{max_ed}_hit"] >= 0).sum()), flush=True)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "bench_sample_max_ed_leftmost.npz"), n_rows=np.array(n_other), **arrays)


if __name__ == "__main__":
    main()
