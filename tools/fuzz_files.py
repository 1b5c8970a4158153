"""Randomised run of step 3 over FILES on the GPU box: the native host (bz_assign_files) against the pandas host that
mirrors the reference's loop, on freshly drawn inputs (tests/test_gpu_native_host.py's generator with its csv / FASTQ
quirks, in three cases of ten with barcode cells wider than the packed row, in three of ten with a whitelist file that
holds short, lower-case, blank, repeated, long and N-containing lines) and drawn --batch-size, device batch, threads, file count, CRLF, gz, restrand, max_ed; three cases of five run a second time sharded by file over 2-4 ranks
(dist.assign_read_sharded, ranks as threads).  Same bytes (decompressed
for .gz), same counts, same gzip member count, or exit code 1.

usage: python tools/fuzz_files.py [seconds] [first_seed]
"""
import json
import os
import pathlib
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_gpu_native_host as tg  # noqa: E402
from blaze_amd import read_assignment as ra  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t_end = time.time() + seconds
    n_cases = n_fail = n_reads = n_sharded = 0
    last = time.time()
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        p = dict(n=int(rng.choice([2, 50, 999, 1000, 1001, 3000, 6000])), cells=int(rng.choice([1, 40, 300, 1500])),
                 crlf=bool(rng.random() < 0.25), gz=bool(rng.random() < 0.5), restrand=bool(rng.random() < 0.7),
                 batchsize=int(rng.choice([1, 7, 64, 333, 1000, 5000])), device_batch_rows=int(rng.choice([1, 100, 700, 1 << 16])),
                 threads=int(rng.choice([1, 3, 8, 16])), n_files=int(rng.integers(1, 7)), err=float(rng.choice([0.5, 2.0, 5.0])),
                 umi_len=int(rng.choice([10, 12])), max_ed=int(rng.choice([0, 1, 2, 2, 3])), float_polyT=bool(rng.random() < 0.5),
                 wide=bool(rng.random() < 0.3), world=int(rng.choice([1, 1, 2, 3, 4])), locate=bool(rng.random() < 0.5),
                 odd_wl=bool(rng.random() < 0.3))
        if p["batchsize"] == 1 and p["n"] > 1001:
            p["n"] = 999                      # pandas per one-row batch: keep the mirror's run short
        p["n_files"] = min(p["n_files"], p["n"] - 1) or 1
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
            tmp = pathlib.Path(d)
            fns, pc, pw, bounds = tg._make_files(tmp, p["n"], p["cells"], seed=seed, crlf=p["crlf"], float_polyT=p["float_polyT"],
                                                 n_files=p["n_files"], err=p["err"], umi_len=p["umi_len"], wide=p["wide"], odd_wl=p["odd_wl"])
            suffix = ".fastq.gz" if p["gz"] else ".fastq"
            out_n, out_p = str(tmp / ("native" + suffix)), str(tmp / ("pandas" + suffix))
            kw = dict(fastq_fns=fns, putative_bc_csv=pc, whitelsit_csv=pw, max_ed=p["max_ed"], batchsize=p["batchsize"],
                      restrand=p["restrand"], device_batch_rows=p["device_batch_rows"])
            got_p = ra.assign_read(fastq_out=out_p, n_process=1, host="python", **kw)
            got_n = ra.assign_read(fastq_out=out_n, n_process=p["threads"], host="native", **kw)
            ok = tuple(map(int, got_n)) == tuple(map(int, got_p)) and tg._text(out_n) == tg._text(out_p)
            if ok and p["gz"]:
                ok = tg._gz_members(open(out_n, "rb").read()) == tg._gz_members(open(out_p, "rb").read())
            if ok and p["world"] > 1:       # the same job sharded by file over ranks (threads here): the single run's file
                out_s = str(tmp / ("sharded" + suffix))
                res = tg._run_sharded(fns, out_s, {k: kw[k] for k in ("putative_bc_csv", "whitelsit_csv", "max_ed", "batchsize", "restrand")},
                                      p["world"], p["locate"], threads=p["threads"])
                ok = all(tuple(map(int, r)) == tuple(map(int, got_n)) for r in res.values()) and tg._text(out_s) == tg._text(out_n)
                if ok and not p["gz"]:
                    ok = open(out_s, "rb").read() == open(out_n, "rb").read()
                if ok and p["gz"]:
                    ok = tg._gz_members(open(out_s, "rb").read()) == tg._gz_members(open(out_n, "rb").read())
                ok = ok and not [f for f in os.listdir(tmp) if ".part" in f]
                n_sharded += 1
        n_cases += 1
        n_reads += p["n"]
        if not ok:
            n_fail += 1
            print(f"FAIL seed {seed}: {json.dumps(p)} native {got_n} pandas {got_p}", flush=True)
            if n_fail >= 5:
                break
        seed += 1
        if time.time() - last > 30:
            last = time.time()
            print(f"... {n_cases} cases, {n_reads} reads, {n_fail} failures", flush=True)
    print(json.dumps({"cases": n_cases, "cases_also_run_sharded": n_sharded, "reads": n_reads, "failures": n_fail, "first_seed": seed0, "next_seed": seed, "seconds": seconds}), flush=True)
    sys.exit(1 if n_fail else 0)


if __name__ == "__main__":
    main()
