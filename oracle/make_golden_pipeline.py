#!/usr/bin/env python3
"""Generate tests/golden/pipeline_leftmost.json by running the REAL reference end to end:
/root/reference/blaze/main.py:332-416 `main(args_string)` (argument parsing, stage 1 over FASTQ files,
whitelisting + empty-drop list, demultiplexing, summary) -- TEST INFRASTRUCTURE ONLY, build container only.
The reference is imported with oracle/shim on sys.path (stand-in for the absent fast-edit-distance wheel).
Only data is written: the input files' text, the parsed argument namespace and the text of every output file
(the FASTQ output decompressed, F9; the knee plot is not kept).  Run:  python oracle/make_golden_pipeline.py
"""
import gzip
import json
import math
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"

if os.environ.get("PYTHONHASHSEED") != "0" or "ORACLE_TIE_RULE" not in os.environ:
    env = dict(os.environ, PYTHONHASHSEED="0", ORACLE_TIE_RULE="0", MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)

sys.path[:0] = [os.path.join(HERE, "shim"), REF, ROOT]
import numpy as np  # noqa: E402
import blaze.main as rmain  # noqa: E402  (the real reference)
import blaze.helper as rhelper  # noqa: E402

ACGT = "ACGT"
ADPT_FULL = "CTACACGACGCTCTTCCGATCT"
TSO = "TTTCTTATATGGG"


def rnd(rng, n):
    return "".join(ACGT[i] for i in rng.integers(0, 4, n))


def noisy(rng, s, sub, ins, dele):
    out = []
    for c in s:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            c = ACGT[int(rng.integers(0, 4))]
        out.append(c)
        if rng.random() < ins:
            out.append(ACGT[int(rng.integers(0, 4))])
    return "".join(out)


def make_fastq(rng, kit, n_reads, cells, full, tag):
    """10x-shaped reads: 80 % carry a cell barcode, 12 % some other barcode of the full list, 8 % junk."""
    umi_len = 10 if kit in ("3v2", "5v2") else 12
    recs = []
    for k in range(n_reads):
        u = rng.random()
        bc = cells[int(rng.integers(0, len(cells)))] if u < 0.80 else full[int(rng.integers(0, len(full)))]
        lead, cdna = rnd(rng, int(rng.integers(0, 30))), rnd(rng, int(rng.integers(40, 160)))
        tail = "T" * int(rng.integers(8, 35)) if kit.startswith("3") else TSO
        body = lead + ADPT_FULL + bc + rnd(rng, umi_len) + tail + cdna
        if u >= 0.92:
            body = rnd(rng, int(rng.integers(20, 250)))
        err = [(0, 0, 0), (0.01, 0.005, 0.005), (0.03, 0.01, 0.02)][k % 3]
        seq = noisy(rng, body, *err) or "A"
        if k % 2:
            seq = rhelper.reverse_complement(seq)
        lo_q = 3 if k % 5 == 0 else 16                    # a fifth of the reads may fall below minQ
        qual = "".join(chr(33 + int(q)) for q in rng.integers(lo_q, 45, len(seq)))
        title = f"@{tag}_{k}" + (f" runid=x{k} ch=7" if k % 4 == 0 else "")
        recs.append(f"{title}\n{seq}\n+\n{qual}\n")
    return "".join(recs)


def jsonable(v):
    if isinstance(v, float) and math.isinf(v):
        return "inf"
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, (str, int, float, bool)) or v is None:
        return v
    return str(v)


def run_case(name, kit, flags, n_files, reads_per_file, seed):
    rng = np.random.default_rng(seed)
    full = sorted({rnd(rng, 16) for _ in range(3000)})
    rng.shuffle(full)
    cells = full[:12]
    with tempfile.TemporaryDirectory() as tmp:
        fq_dir = os.path.join(tmp, "fq")
        os.makedirs(fq_dir)
        inputs = {}
        for f in range(n_files):
            text = make_fastq(rng, kit, reads_per_file + 7 * f, cells, full, f"r{f}")
            fn = f"part_{f}.fastq" + (".gz" if f == 1 else "")
            inputs[fn] = text
            if fn.endswith(".gz"):
                with gzip.open(os.path.join(fq_dir, fn), "wt") as h:
                    h.write(text)
            else:
                with open(os.path.join(fq_dir, fn), "w") as h:
                    h.write(text)
        full_fn = os.path.join(tmp, "full_whitelist.txt")
        with open(full_fn, "w") as h:
            h.write("".join(b + "\n" for b in full))
        prefix = os.path.join(tmp, "out", "t_")
        arg_string = f"--kit-version {kit} --full-bc-whitelist {full_fn} --output-prefix {prefix} --threads 1 " \
                     f"--minimal_stdout {flags} {fq_dir}"
        args = rmain.parse_arg(arg_string)                 # recorded so that the test needs no CLI parser
        ns = {k: jsonable(v) for k, v in vars(args).items()}
        rmain.main(arg_string)
        outs = {}
        for key in ("out_raw_bc_fn", "out_whitelist_fn", "out_emptydrop_fn", "summary_fn"):
            with open(getattr(args, key)) as h:
                outs[key] = h.read()
        opener = gzip.open if args.output_fastq.endswith(".gz") else open
        with opener(args.output_fastq, "rt") as h:
            outs["output_fastq"] = h.read()
        print(name, "reads", sum(t.count("\n") // 4 for t in inputs.values()), "cells", outs["out_whitelist_fn"].count("\n"),
              "empty", outs["out_emptydrop_fn"].count("\n"), "demultiplexed", outs["output_fastq"].count("\n") // 4)
        print(outs["summary_fn"][-160:])
        return dict(name=name, kit=kit, arg_string=arg_string.replace(tmp, "<TMP>"),
                    args={k: (v.replace(tmp, "<TMP>") if isinstance(v, str) else
                              [x.replace(tmp, "<TMP>") for x in v] if isinstance(v, list) else v) for k, v in ns.items()},
                    input_fastq=inputs, full_whitelist=full, expect=outs)


def main():
    cases = [
        run_case("3v3_expect_cells_gz", "3v3", "--expect-cells 12 --batch-size 100", 3, 230, 501),
        run_case("5v2_count_threshold_plain_norestrand", "5v2",
                 "--count-threshold 4 --output-fastq m.fastq --no-restrand --batch-size 64 --minQ 10 --max-edit-distance 1", 2, 260, 502),
        run_case("3v2_force_cells_hs", "3v2", "--force-cells 9 --high-sensitivity-mode --batch-size 1000", 2, 200, 503),
    ]
    out = os.path.join(ROOT, "tests", "golden", "pipeline_leftmost.json")
    with open(out, "w") as f:
        json.dump(dict(generator="oracle/make_golden_pipeline.py (real blaze.main.main, PYTHONHASHSEED=0, leftmost tie rule)",
                       cases=cases), f)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
