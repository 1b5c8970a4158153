"""Randomised run of the THREE stages over files on the GPU box: blaze_amd.main.main with the native host (C++ readers,
device search / scoring, pandas-exact writers) against the same call with the pandas host that mirrors the reference's
loops, on freshly drawn FASTQ files (structured reads carrying barcodes of a drawn cell population, unstructured reads,
read ids with commas and quotes, plain and gzipped files) and drawn settings (kit, UMI length, minQ, --batch-size,
threads, max_ed, restrand, plain / .gz output, the three cell-calling modes, high-sensitivity mode).  putative_bc.csv,
whitelist.csv, emtpy_bc_list.csv, summary.txt and the demultiplexed FASTQ (decompressed for .gz) must be the same bytes,
or both calls must raise the same exception type; two jobs of five run a third time as one process per GPU would
(main(args, comm=...), 2-4 ranks as threads) and must write the same files again.  Exit code 1 otherwise.

usage: python tools/fuzz_pipeline.py [seconds] [first_seed]
"""
import gzip
import json
import os
import sys
import tempfile
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import fuzz_stage1 as fs  # noqa: E402
from blaze_amd import main as bmain  # noqa: E402

ID_ODD = ["a,b", 'q"uote', "x,y\"z", "plain", "in@side"]


def draw_job(rng, d):
    kit = str(rng.choice(["3v3", "3v2", "5v2", "5v3"]))
    kit5 = kit.startswith("5")
    umi_len = int(rng.choice([10, 12]))
    p = float(rng.choice([0.0, 0.01, 0.03]))
    n_cells = int(rng.choice([5, 40, 150]))
    n = int(rng.choice([60, 400, 1500, 3000]))
    cells = list(dict.fromkeys(fs.rand_seq(rng, 16) for _ in range(n_cells)))
    extra = [fs.rand_seq(rng, 16) for _ in range(3 * n_cells)]
    weights = rng.zipf(1.6, size=len(cells)).astype(float)
    weights /= weights.sum()
    recs = []
    bad_id_at = int(rng.integers(0, n)) if rng.random() < 0.04 else -1
    for i in range(n):
        if rng.random() < 0.2:
            s = fs.draw_read(rng, kit5, umi_len, p)
        else:
            bc = cells[int(rng.choice(len(cells), p=weights))] if rng.random() < 0.9 else extra[int(rng.integers(0, len(extra)))]
            head = fs.rand_seq(rng, int(rng.integers(0, 90))) + "CTACACGACGCTCTTCCGATCT"
            tail = ("TTTCTTATATGGG" if kit5 else "T" * int(rng.integers(8, 40))) + fs.rand_seq(rng, int(rng.integers(20, 400)))
            s = fs.noisy(rng, head + bc + fs.rand_seq(rng, umi_len) + tail, p)
            if rng.random() < 0.5:
                s = s.translate(fs.COMP)[::-1]
        s = s or "A"
        q = "".join(chr(33 + int(x)) for x in rng.integers(5, 45, len(s)))
        rid = f"r{i}_{ID_ODD[int(rng.integers(0, len(ID_ODD)))]}" if rng.random() < 0.1 else f"read{i:06d}"
        if bad_id_at == i:
            rid += "@"             # stage 3 strips it (read_assignment.py:31), stage 1 does not: the reference's "Different order" exit
        recs.append(f"@{rid} ch={i % 7}\n{s}\n+\n{q}\n")
    n_files = int(rng.integers(1, 5))
    cuts = sorted(rng.choice(np.arange(1, n), size=min(n_files - 1, n - 1), replace=False).tolist()) if n_files > 1 else []
    bounds = [0] + cuts + [n]
    fns = []
    for k in range(len(bounds) - 1):
        body = "".join(recs[bounds[k]:bounds[k + 1]]).encode()
        if rng.random() < 0.4:
            fn = os.path.join(d, f"in_{k}.fastq.gz")
            with gzip.open(fn, "wb", compresslevel=1) as f:
                f.write(body)
        else:
            fn = os.path.join(d, f"in_{k}.fastq")
            with open(fn, "wb") as f:
                f.write(body)
        fns.append(fn)
    full = os.path.join(d, "full.txt")
    with open(full, "w") as f:
        f.write("".join(b + "\n" for b in rng.permutation(cells + extra)))
    mode = int(rng.integers(0, 3))
    common = dict(fastq_fns=fns, threads=int(rng.choice([1, 3, 8, 16])), batch_size=int(rng.choice([3, 50, 1000])),
                  minimal_stdout=True, expect_cells=len(cells) if mode == 0 else None, count_threshold=int(rng.integers(1, 4)) if mode == 1 else None,
                  force_cells=max(1, len(cells) // 2) if mode == 2 else None, minQ=int(rng.choice([0, 10, 15])),
                  max_edit_distance=int(rng.choice([0, 1, 2, 2, 3])), kit_version=kit, umi_len=umi_len, full_bc_whitelist=full,
                  high_sensitivity_mode=bool(rng.random() < 0.3), do_empty_drops_output=bool(rng.random() < 0.8), known_bc_list=None,
                  restrand=bool(rng.random() < 0.7), empty_max_count=float("inf"), do_bc_search=True, do_whitelisting=True,
                  do_demultiplexing=True)
    gz = bool(rng.random() < 0.5)

    def ns(prefix):
        p_ = os.path.join(d, prefix)
        return SimpleNamespace(output_fastq=p_ + ("matched_reads.fastq.gz" if gz else "matched_reads.fastq"), out_raw_bc_fn=p_ + "putative_bc.csv",
                               out_whitelist_fn=p_ + "whitelist.csv", out_emptydrop_fn=p_ + "emtpy_bc_list.csv", out_plot_fn=p_ + "knee_plot.png",
                               summary_fn=p_ + "summary.txt", **common)
    shown = {k: v for k, v in common.items() if k not in ("fastq_fns", "full_bc_whitelist")}
    shown.update(reads=n, cells=len(cells), files=len(fns), gz=gz, err=p)
    return ns, shown, n


def outputs(a):
    out = {}
    for key in ("out_raw_bc_fn", "out_whitelist_fn", "out_emptydrop_fn", "summary_fn", "output_fastq"):
        fn = getattr(a, key)
        if not os.path.exists(fn):
            out[key] = None
            continue
        with (gzip.open(fn, "rb") if fn.endswith(".gz") else open(fn, "rb")) as f:
            out[key] = f.read()
    return out


def run(a, host):
    try:
        bmain.main(a, host=host)
        return None
    except SystemExit:
        return "SystemExit"
    except Exception as e:      # noqa: BLE001  (compared by type between the two hosts)
        return type(e).__name__


def run_ranks(a, world):
    """main(args, comm=...) with `world` ranks as threads of this process (tests/test_gpu_pipeline.py does the same): the
    native file loops and the whitelist upload are serialised on the one GPU, everything around them runs concurrently."""
    import threading
    from test_gpu_native_host import _ThreadComm
    shared = {"bar": threading.Barrier(world), "lock": threading.Lock()}
    errs = []

    def rank_main(rank):
        try:
            bmain.main(a, comm=_ThreadComm(rank, world, shared))
        except BaseException as e:      # noqa: BLE001
            errs.append(type(e).__name__)
            shared["bar"].abort()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    real = [e for e in errs if e != "BrokenBarrierError"]
    return (real or errs or [None])[0]


def serialise_device_calls():
    import threading
    from blaze_amd import putative_bc, read_assignment
    lock = threading.Lock()

    def locked(fn):
        def call(*a, **k):
            with lock:
                return fn(*a, **k)
        return call
    putative_bc.putative_bc_files = locked(putative_bc.putative_bc_files)
    read_assignment._assign_files_native = locked(read_assignment._assign_files_native)
    read_assignment._ensure_whitelist = locked(read_assignment._ensure_whitelist)
    read_assignment.get_engine(read_assignment._device_ids(None)[0])


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    serialise_device_calls()
    seed = seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t_end = time.time() + seconds
    n_cases = n_fail = n_reads = n_raised = n_ranks = 0
    last = time.time()
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
            ns, shown, n = draw_job(rng, d)
            a_n, a_p = ns("native_"), ns("pandas_")
            e_n, e_p = run(a_n, "native"), run(a_p, "python")
            o_n, o_p = outputs(a_n), outputs(a_p)
            diff = [k for k in o_n if o_n[k] != o_p[k]]
            n_raised += bool(e_n or e_p)
            if e_n != e_p:
                diff.append("exception")
            elif e_n:               # both stopped: how far the output had got by then is each host's own business
                diff = [k for k in diff if k != "output_fastq"]
            world = int(rng.choice([1, 1, 2, 3, 4]))
            if world > 1 and not diff and not e_n:      # the same job as one process per GPU (ranks as threads here)
                a_r = ns("ranks_")
                e_r = run_ranks(a_r, world)
                o_r = outputs(a_r)
                diff = [f"{world} ranks: {k}" for k in o_n if o_n[k] != o_r[k]] + ([f"{world} ranks raised {e_r}"] if e_r else [])
                diff += [f for f in os.listdir(d) if ".part" in f]
                n_ranks += 1
        n_cases += 1
        n_reads += n
        if diff:
            n_fail += 1
            print(f"FAIL seed {seed}: {json.dumps(shown)} differ: {diff} raised: {e_n} / {e_p}", flush=True)
            if n_fail >= 5:
                break
        seed += 1
        if time.time() - last > 30:
            last = time.time()
            print(f"... {n_cases} cases, {n_reads} reads, {n_fail} failures", flush=True)
    print(json.dumps({"cases": n_cases, "reads": n_reads, "cases_that_raised_on_both_hosts": n_raised, "cases_also_run_with_2_to_4_ranks": n_ranks, "failures": n_fail, "first_seed": seed0,
                      "next_seed": seed, "seconds": seconds}), flush=True)
    sys.exit(1 if n_fail else 0)


if __name__ == "__main__":
    main()
