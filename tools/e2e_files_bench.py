"""End-to-end step 3 over FILES (FASTQ + putative_bc.csv -> demultiplexed FASTQ) on the GPU box's host cores:
native host pipeline (bz_assign_files) vs the pandas host that mirrors the reference's loop.  Not bench.py's
metric (that one is HBM-resident); this is the PCIe + file-system + host-CPU inclusive figure of SURVEY 8f N1/N2
and BASELINE configs[4]'s shape (10-nt UMI, 10k cells, restrand + trim).

usage: python tools/e2e_files_bench.py [n_reads] [n_cells] [threads] [--gz] [--gz-in] [--n-files=N] [--python-too]
"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from blaze_amd import synth  # noqa: E402
from blaze_amd import read_assignment as ra  # noqa: E402


write_inputs = synth.write_file_job


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if len(args) > 0 else 500_000
    n_cells = int(args[1]) if len(args) > 1 else 10_000
    threads = int(args[2]) if len(args) > 2 else 16
    gz = "--gz" in sys.argv
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        t0 = time.time()
        n_files = 64 if "--gz-in" in sys.argv else 4
        for a in sys.argv:
            if a.startswith("--n-files="):
                n_files = int(a.split("=")[1])
        fns, pc, wl, fq_bytes = write_inputs(d, n, n_cells, n_files=n_files)
        if "--gz-in" in sys.argv:                       # a nanopore run's shape: many .fastq.gz files of a few thousand reads
            import subprocess
            for lo in range(0, len(fns), 256):
                subprocess.check_call(["gzip", "-1"] + fns[lo:lo + 256])
            fns = [f + ".gz" for f in fns]
        gen_s = time.time() - t0
        out = {"reads": n, "cells": n_cells, "threads": threads, "gz": gz, "gz_in": "--gz-in" in sys.argv, "n_files": n_files, "fastq_bytes": fq_bytes, "generate_s": gen_s,
               "gz_prefetch": os.environ.get("BLAZE_AMD_GZ_PREFETCH", "1")}
        suffix = ".fastq.gz" if gz else ".fastq"
        kw = dict(fastq_fns=fns, putative_bc_csv=pc, whitelsit_csv=wl, max_ed=2, batchsize=1000, restrand=True)
        ra.assign_read(fastq_out=os.path.join(d, "warm" + suffix), n_process=threads, host="native", **kw)   # page cache + engine warm
        for rep in range(2):
            t0 = time.time()
            got = ra.assign_read(fastq_out=os.path.join(d, "native" + suffix), n_process=threads, host="native", **kw)
            dt = time.time() - t0
            out[f"native_run{rep}"] = {"s": dt, "reads_per_s": n / dt, "input_GB_per_s": fq_bytes / dt / 1e9,
                                       "assigned": int(got[0]), "total": int(got[1])}
        if "--python-too" in sys.argv:
            m = min(n, 50_000)
            kw2 = dict(kw)
            t0 = time.time()
            sub = os.path.join(d, "sub.fastq")
            with open(fns[0], "rb") as f, open(sub, "wb") as g:
                for _ in range(4 * m):
                    g.write(f.readline())
            kw2["fastq_fns"] = [sub]
            t0 = time.time()
            got = ra.assign_read(fastq_out=os.path.join(d, "pandas" + suffix), n_process=1, host="python", **kw2)
            dt = time.time() - t0
            out["pandas_host"] = {"s": dt, "reads": int(got[1]), "reads_per_s": got[1] / dt}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
