#!/usr/bin/env python3
"""BASELINE.md slot B1: the REAL reference's step 3 timed in the build container.

TEST / MEASUREMENT INFRASTRUCTURE ONLY; needs /root/reference, so it never runs on the
GPU box.  It imports blaze.read_assignment from /root/reference with oracle/shim on
sys.path (the shim stands in for the absent fast-edit-distance wheel and routes every
call to oracle_ed.c through ctypes) and times

  assign_read   /root/reference/blaze/read_assignment.py:253-330

over synthetic 10x 3v3 files (20k reads x 5,000-cell whitelist, max_ed 2,
batch size 1000) at n_process = 1 (serial loop, :293-305) and n_process = 8 (ordered
pool of 7 workers, helper.py:137).  The figure is interpreter-bound by construction
(one Python -> native call per (read, barcode) pair, :71-72); the ctypes hop of the shim
costs more per call than the wheel's Cython entry would, so read it as a lower bound of
the reference's speed on these cores, within a small factor.

Run:  python oracle/time_reference.py [n_reads] [n_cells]   -> one JSON line
"""
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"

if os.environ.get("PYTHONHASHSEED") != "0":
    env = dict(os.environ, PYTHONHASHSEED="0", ORACLE_TIE_RULE="0", MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

sys.path[:0] = [os.path.join(HERE, "shim"), REF, ROOT]

import blaze.read_assignment as ra  # noqa: E402  (the real reference)
from blaze_amd import synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
    n_cells = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000
    full = synth.full_whitelist_codes(400_000, seed=1)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=2)
    rows = synth.make_rows(n, cells, full, umi_len=12, seed=3)
    ids = synth.read_ids(n, seed=4)
    recs = synth.fastq_records(n, ids, seed=5)
    out = {"reads": n, "cells": n_cells, "max_ed": 2, "batch_size": 1000, "host_cpus": os.cpu_count(),
           "rows_blank": int(rows["blank"].sum()), "what": "real blaze.read_assignment.assign_read, "
           "sub_edit_distance = oracle_ed.c behind a ctypes shim"}
    with tempfile.TemporaryDirectory() as d:
        fq = os.path.join(d, "reads.fastq")
        with open(fq, "w") as f:
            f.write(synth.fastq_text(recs))
        pc = os.path.join(d, "putative_bc.csv")
        with open(pc, "w") as f:
            f.write(synth.csv_text(rows, ids))
        wl = os.path.join(d, "whitelist.csv")
        with open(wl, "w") as f:
            for w in synth.codes_to_bytes(cells):
                f.write(w.tobytes().decode() + "-1\n")
        for threads in (1, 8):
            t0 = time.time()
            demul, total = ra.assign_read([fq], os.path.join(d, f"out{threads}.fastq"), pc, wl, 2, threads, 1000)
            dt = time.time() - t0
            out[f"threads_{threads}"] = {"s": round(dt, 2), "reads_per_s": round(total / dt, 1),
                                         "assigned": int(demul), "total": int(total),
                                         "workers": 1 if threads == 1 else min(threads - 1, os.cpu_count() - 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
