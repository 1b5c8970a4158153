#!/usr/bin/env python3
"""BASELINE configs[0] at its stated size through the REAL reference: 29 FASTQ files x 4,000 reads, 3v3 kit,
`--expect-cells 500`, /root/reference/blaze/main.py:332-416 `main(args_string)` end to end.
TEST INFRASTRUCTURE ONLY, build container only (the reference is imported with oracle/shim on sys.path, stand-in for the
absent fast-edit-distance wheel; leftmost tie rule).  The bundled FASTQ files are missing (.MISSING_LARGE_BLOBS:5-33), so
the inputs are the deterministic stand-ins of blaze_amd.synth.write_pipeline_job, which the test regenerates; the
fixture holds the argument namespace, the sha256 of the inputs and, per output file, its size, line count and sha256
(the FASTQ output decompressed, F9) -- 116,000 reads of text do not belong in a repository.
Run:  python oracle/make_golden_configs0.py        (a few minutes: the reference's stage 1 and 3 are Python loops)
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"

if os.environ.get("PYTHONHASHSEED") != "0" or "ORACLE_TIE_RULE" not in os.environ:
    env = dict(os.environ, PYTHONHASHSEED="0", ORACLE_TIE_RULE="0", MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)

sys.path[:0] = [os.path.join(HERE, "shim"), REF, ROOT]
import blaze.main as rmain  # noqa: E402  (the real reference)
from blaze_amd import synth  # noqa: E402
from oracle.make_golden_pipeline import jsonable  # noqa: E402


def digest(path, gz=False):
    raw = open(path, "rb").read()
    if gz:
        raw = gzip.decompress(raw)
    return dict(bytes=len(raw), lines=raw.count(b"\n"), sha256=hashlib.sha256(raw).hexdigest())


def main():
    job = dict(n_files=29, reads_per_file=4000, n_cells=500, n_full=20000, seed=116)
    with tempfile.TemporaryDirectory() as tmp:
        fq_dir, full_fn, in_sha = synth.write_pipeline_job(tmp, **job)
        prefix = os.path.join(tmp, "out", "t_")
        arg_string = f"--kit-version 3v3 --expect-cells 500 --full-bc-whitelist {full_fn} --output-prefix {prefix} --threads 8 " \
                     f"--minimal_stdout {fq_dir}"
        args = rmain.parse_arg(arg_string)
        ns = {k: jsonable(v) for k, v in vars(args).items()}
        t0 = time.time()
        rmain.main(arg_string)
        took = time.time() - t0
        outs = {key: digest(getattr(args, key)) for key in ("out_raw_bc_fn", "out_whitelist_fn", "out_emptydrop_fn", "summary_fn")}
        outs["output_fastq"] = digest(args.output_fastq, gz=args.output_fastq.endswith(".gz"))
        with open(args.summary_fn) as h:
            summary = h.read()
        print(summary)
        print({k: (v["bytes"], v["lines"]) for k, v in outs.items()}, f"reference took {took:.0f} s")
        case = dict(name="configs0_29x4000_3v3_expect_cells_500", job=job, inputs_sha256=in_sha,
                    arg_string=arg_string.replace(tmp, "<TMP>"),
                    args={k: (v.replace(tmp, "<TMP>") if isinstance(v, str) else
                              [x.replace(tmp, "<TMP>") for x in v] if isinstance(v, list) else v) for k, v in ns.items()},
                    expect=outs, summary=summary, reference_seconds=took)
    out = os.path.join(ROOT, "tests", "golden", "pipeline_configs0.json")
    with open(out, "w") as f:
        json.dump(dict(generator="oracle/make_golden_configs0.py (real blaze.main.main, PYTHONHASHSEED=0, leftmost tie rule, --threads 8)",
                       case=case), f, indent=0)
    print("wrote", out)


if __name__ == "__main__":
    main()
