"""End-to-end BLAZE pipeline over FILES on the GPU box: FASTQ -> putative_bc.csv (stage 1) -> whitelist.csv +
emtpy_bc_list.csv (stage 2) -> demultiplexed FASTQ (stage 3), through blaze_amd.main.main with the native hosts.
Not bench.py's metric (HBM-resident step 3): this is the file-system + host-CPU + PCIe inclusive figure.

usage: python tools/e2e_pipeline_bench.py [n_reads] [n_cells] [threads] [--python-stage1-too]
"""
import json
import os
import sys
import tempfile
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from blaze_amd import main as bmain  # noqa: E402
from blaze_amd import synth  # noqa: E402

ADPT_FULL = b"CTACACGACGCTCTTCCGATCT"
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def write_inputs(d, n, n_cells, umi_len=12, n_files=4, median_len=900):
    full = synth.codes_to_bytes(synth.full_whitelist_codes(400_000, seed=1))
    cells_codes = synth.cell_whitelist_codes(n_cells, synth.full_whitelist_codes(400_000, seed=1), seed=2)
    cells = [w.tobytes() for w in synth.codes_to_bytes(cells_codes)]
    full_l = [w.tobytes() for w in full]
    rng = np.random.Generator(np.random.PCG64(7))
    lens = np.maximum(200, rng.lognormal(np.log(median_len), 0.6, size=n).astype(np.int64))
    pool = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(lens.max()) + (1 << 20))].tobytes()
    starts = rng.integers(0, 1 << 20, size=n)
    which = rng.random(n)
    pick = rng.integers(0, 1 << 30, size=n)
    tlen = rng.integers(12, 35, size=n)
    mut = rng.random(n)
    mpos = rng.integers(0, 16, size=n)
    fns, per, total = [], -(-n // n_files), 0
    for k in range(n_files):
        fn = os.path.join(d, f"reads_{k}.fastq")
        with open(fn, "wb") as f:
            for i in range(k * per, min(n, (k + 1) * per)):
                bc = cells[pick[i] % n_cells] if which[i] < 0.85 else full_l[pick[i] % len(full_l)]
                if mut[i] < 0.2:                                   # one substitution in a fifth of the barcodes
                    p = int(mpos[i])
                    bc = bc[:p] + (b"A" if bc[p:p + 1] != b"A" else b"C") + bc[p + 1:]
                s, L = int(starts[i]), int(lens[i])
                body = pool[s:s + 20] + ADPT_FULL + bc + pool[s + 30:s + 30 + umi_len] + b"T" * int(tlen[i]) + pool[s + 50:s + 50 + L]
                if which[i] > 0.97:
                    body = pool[s:s + L]                           # no adaptor at all
                if i & 1:
                    body = body.translate(COMP)[::-1]
                rec = b"@read_%d runid=synthetic\n" % i + body + b"\n+\n" + b"I" * len(body) + b"\n"
                f.write(rec)
                total += len(rec)
        fns.append(fn)
    full_fn = os.path.join(d, "full_whitelist.txt")
    with open(full_fn, "wb") as f:
        f.write(b"\n".join(full_l) + b"\n")
    return fns, full_fn, total


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(argv[0]) if len(argv) > 0 else 500_000
    n_cells = int(argv[1]) if len(argv) > 1 else 5_000
    threads = int(argv[2]) if len(argv) > 2 else 16
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        t0 = time.time()
        fns, full_fn, fq_bytes = write_inputs(d, n, n_cells)
        out = {"reads": n, "cells": n_cells, "threads": threads, "fastq_bytes": fq_bytes, "generate_s": time.time() - t0}

        def ns(prefix):
            p = os.path.join(d, prefix)
            return SimpleNamespace(fastq_fns=fns, output_fastq=p + "matched_reads.fastq", out_raw_bc_fn=p + "putative_bc.csv",
                                   out_whitelist_fn=p + "whitelist.csv", out_emptydrop_fn=p + "emtpy_bc_list.csv",
                                   out_plot_fn=p + "knee_plot.png", summary_fn=p + "summary.txt", threads=threads, batch_size=1000,
                                   minimal_stdout=True, expect_cells=n_cells, count_threshold=None, force_cells=None, minQ=15,
                                   max_edit_distance=2, kit_version="3v3", umi_len=12, full_bc_whitelist=full_fn,
                                   high_sensitivity_mode=False, do_empty_drops_output=True, known_bc_list=None, restrand=True,
                                   empty_max_count=float("inf"), do_bc_search=True, do_whitelisting=True, do_demultiplexing=True)
        for rep in ("warm", "run"):
            a = ns(rep + "_")
            a.do_whitelisting = a.do_demultiplexing = False
            t0 = time.time(); bmain.main(a); t1 = time.time()
            a.do_bc_search, a.do_whitelisting = False, True
            if rep == "run" and "--profile-stage2" in sys.argv:
                import cProfile, pstats
                pr = cProfile.Profile(); pr.enable(); bmain.main(a); pr.disable()
                pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(22)
            else:
                bmain.main(a)
            t2 = time.time()
            a.do_whitelisting, a.do_demultiplexing = False, True
            bmain.main(a); t3 = time.time()
            out[rep] = {"stage1_s": t1 - t0, "stage2_s": t2 - t1, "stage3_s": t3 - t2, "total_s": t3 - t0,
                        "stage1_reads_per_s": n / (t1 - t0), "stage3_reads_per_s": n / (t3 - t2), "pipeline_reads_per_s": n / (t3 - t0),
                        "stage1_input_GB_per_s": fq_bytes / (t1 - t0) / 1e9,
                        "cells_called": sum(1 for _ in open(a.out_whitelist_fn)), "summary_tail": open(a.summary_fn).read()[-90:]}
        a = ns("all_")                                  # the three stages in ONE call: the knee plot of stage 2 is drawn while stage 3 runs
        t0 = time.time()
        bmain.main(a)
        out["one_call"] = {"total_s": time.time() - t0, "pipeline_reads_per_s": n / (time.time() - t0)}
        if "--python-stage1-too" in sys.argv:
            m = min(n, 20_000)
            sub = os.path.join(d, "sub.fastq")
            with open(fns[0], "rb") as f, open(sub, "wb") as g:
                for _ in range(4 * m):
                    g.write(f.readline())
            a = ns("py_")
            a.fastq_fns = [sub]
            t0 = time.time()
            bmain.search_putative_bc(a, host="python")
            out["stage1_pandas_host"] = {"reads": m, "reads_per_s": m / (time.time() - t0)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
