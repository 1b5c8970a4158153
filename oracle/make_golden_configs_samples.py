#!/usr/bin/env python3
"""Samples of the workloads of BASELINE configs[2], [3], [4] (as tests/test_gpu_full_size.py builds them) through the REAL
reference, leftmost tie rule.  TEST INFRASTRUCTURE ONLY, build container only (oracle/shim stands in for the absent
fast-edit-distance wheel).

  configs2   first 30,000 rows of the 10M-row 3v2 workload (10-nt UMI) x 20,000 cells      real _match_bc_row
  configs4   first 30,000 rows of the 25M-row 5v2 shard (10-nt UMI) x 10,000 cells          real _match_bc_row
  configs3   first 600 rows of the stress workload x the full 6,794,880-barcode list        real _match_bc_row (~14 s per scored row)
  configs4_files   30,000 reads in 4 FASTQ files, 10,000 cells, trim + restrand             real assign_read (n_process 8), plain and .gz

Written: tests/golden/configs_samples_leftmost.npz (per sample: assigned index per row, returned UMI, strand, sha256 of the
triples) and tests/golden/configs4_files_leftmost.json (sha256 / size of the output text, counts, the set() order).
The whitelists are handed to _match_bc_row as the ordered lists the device gets.  Data only.
Run:  python oracle/make_golden_configs_samples.py [configs2 configs4 configs3 configs4_files]
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from io import StringIO

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"

if "ORACLE_TIE_RULE" not in os.environ:
    env = dict(os.environ, PYTHONHASHSEED="0", ORACLE_TIE_RULE="0", MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
    sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))

sys.path[:0] = [os.path.join(HERE, "shim"), REF, ROOT]
import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import blaze.read_assignment as ra  # noqa: E402  (the real reference)
from blaze_amd import synth  # noqa: E402

SAMPLES = {   # name: (rows of the sample, cells, umi_len, seed of chunk 0, rows of chunk 0, size of the full list, whitelist = full list?)
    "configs2": (30_000, 20_000, 10, 3, 500_000, 400_000, False),
    "configs4": (30_000, 10_000, 10, 11, 1_000_000, 400_000, False),
    "configs3": (600, 5_000, 12, 3, 500_000, synth.FULL_WHITELIST_SIZE, True),
}
_state = {}


def _load(name):
    if name not in _state:
        n, n_cells, umi_len, seed, chunk, full_size, use_full = SAMPLES[name]
        full = synth.full_whitelist_codes(full_size, seed=1)
        cells = synth.cell_whitelist_codes(n_cells, full, seed=2)
        rows = synth.make_rows(chunk, cells, full, umi_len=umi_len, seed=seed)
        wl = [w.tobytes().decode() for w in synth.codes_to_bytes(full if use_full else cells)]
        _state[name] = (rows, wl)
    return _state[name]


def _work(job):
    name, lo, hi = job
    rows, wl = _load(name)
    df = pd.read_csv(StringIO(synth.csv_text(rows, ["x"] * rows["n"], lo, hi))).fillna("")
    return [ra._match_bc_row(row, wl, 2, 0) for row in df.itertuples()]


def row_sample(name, pool):
    n = SAMPLES[name][0]
    step = 1000 if n >= 1000 else 5
    parts = pool.map(_work, [(name, lo, min(n, lo + step)) for lo in range(0, n, step)])
    res = [r for p in parts for r in p]
    _, wl = _load(name)
    pos = {}
    for k, w in enumerate(wl):
        pos.setdefault(w, k)
    hit = np.array([pos.get(r[0], -1) if r[0] else -1 for r in res], dtype=np.int32)
    sha = hashlib.sha256("".join(f"{r[0]},{r[1]},{r[2]}\n" for r in res).encode()).hexdigest()
    print(name, "rows", len(res), "assigned", int((hit >= 0).sum()), "sha256", sha[:16], flush=True)
    return {f"{name}_hit": hit, f"{name}_umi": np.array([r[1] for r in res], dtype="S16"),
            f"{name}_strand": np.array([r[2] for r in res], dtype="S1"), f"{name}_sha256": np.array(sha)}


def files_sample():
    job = dict(n=30_000, n_cells=10_000, umi_len=10, n_files=4, seed=3, median_len=400)
    out = dict(job=job, outputs={})
    with tempfile.TemporaryDirectory() as d:
        fns, pc, wl, _ = synth.write_file_job(d, job["n"], job["n_cells"], umi_len=job["umi_len"], n_files=job["n_files"], seed=job["seed"],
                                              median_len=job["median_len"])
        whitelist = [line.split("-")[0].strip() for line in open(wl)]
        out["iteration_order"] = list(set(whitelist))           # what every pool worker iterates (PYTHONHASHSEED=0 is inherited)
        for name, restrand in (("out.fastq", True), ("out.fastq.gz", False)):
            po = os.path.join(d, name)
            demul, total = ra.assign_read(fastq_fns=fns, fastq_out=po, putative_bc_csv=pc, whitelsit_csv=wl, max_ed=2, n_process=8,
                                          batchsize=1000, restrand=restrand)
            raw = open(po, "rb").read()
            if name.endswith(".gz"):
                raw = gzip.decompress(raw)
            out["outputs"][name] = dict(restrand=restrand, demul=int(demul), total=int(total), bytes=len(raw),
                                        sha256=hashlib.sha256(raw).hexdigest())
            print("configs4_files", name, out["outputs"][name], flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "configs4_files_leftmost.json"), "w") as f:
        json.dump(out, f)


def main():
    from multiprocessing import Pool
    which = sys.argv[1:] or ["configs2", "configs4", "configs3", "configs4_files"]
    path = os.path.join(ROOT, "tests", "golden", "configs_samples_leftmost.npz")
    arrays = dict(np.load(path)) if os.path.exists(path) else {}
    for name in which:
        if name == "configs4_files":
            files_sample()
            continue
        with Pool(8) as pool:                               # a fresh pool per sample: each worker holds one workload
            arrays.update(row_sample(name, pool))
        np.savez_compressed(path, **arrays)
    print("done")


if __name__ == "__main__":
    main()
