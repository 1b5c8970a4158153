#!/usr/bin/env python3
"""bench.py -- step-3 read assignment throughput on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path (classify + exact shortcut + bit-parallel whitelist scan +
histogram) over one batch of synthetic 10x-shaped rows that is already resident in HBM.
N=1 workload = BASELINE.json configs[1]: 1M nanopore-error 3v3 reads vs a 5,000-cell whitelist.
N>1: every rank gets its own 1M-row shard (weak scaling, whitelist replicated, no data-path
collective); RCCL is used for the barrier, the max-over-ranks time and the final per-barcode
count all-reduce only.

Prints ONE JSON line on rank 0.  No PyTorch: the engine is the C-ABI library (ctypes).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from blaze_amd import _lib, dist, synth  # noqa: E402
from blaze_amd.engine import Engine, device_info  # noqa: E402

ALG_OPS_PER_PAIR = 340            # 20 text columns x 17 int32 bit-vector ops (Myers 1999)
PEAK_INT32_TOPS = 256 * 4 * 32 * 2.4e9 / 1e12   # 256 CU x 4 SIMD-32 x 2.4 GHz (MI355X_MICROARCH.md)
BRUTE_INSTR_PER_PAIR = 124.5      # k_scan_bitparallel<2>: 498 VALU instructions per four barcodes


def make_engine(device):
    """The engine the bench measures: the C-ABI library on `device`.  BLAZE_AMD_BENCH_ENGINE=module:Class swaps in a
    stand-in for the CPU test of the launch logic (tests/stub_engine.py); its line is labelled as such."""
    spec = os.environ.get("BLAZE_AMD_BENCH_ENGINE")
    if spec:
        import importlib
        mod, cls = spec.split(":")
        return getattr(importlib.import_module(mod), cls)(device), True
    return Engine(device), False


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start N fresh rank processes (one per GPU) BEFORE this
    process touches a GPU, hand them RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, pass rank 0's JSON line through, and
    fail if any rank fails.  (Under torchrun the ranks exist already and this is not used.)"""
    import socket
    import subprocess
    import uuid
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    run_id = uuid.uuid4().hex[:12]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), BLAZE_AMD_RUN_ID=run_id)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out0 = procs[0].communicate()[0]
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    if any(codes):
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        return 1
    return 0


class Plumbing:
    """barrier / max / sum over ranks on the job's communicator (blaze_amd.dist.make_comm): RCCL through the C ABI, or the
    host-side socket hub when RCCL cannot be initialised on every rank -- `kind` says which and why."""

    def __init__(self, eng, rank, world, try_rccl=True):
        self.rank, self.world, self.eng = rank, world, eng
        self.comm = dist.make_comm(eng, try_rccl=try_rccl)
        self.rccl = self.comm if self.comm.kind == "rccl" else None
        self.hung = bool(getattr(self.comm, "hung", False))
        if world == 1:
            self.kind = "local"
        elif self.rccl:
            self.kind = "rccl"
        elif not try_rccl:
            self.kind = "socket-hub (stand-in engine: no RCCL)"
        else:
            self.kind = f"socket-hub (RCCL unavailable: {self.comm.why})"[:240]

    def barrier(self):
        self.comm.barrier()

    def max(self, x):
        return self.comm.allreduce_max_f64(x)

    def sum_counts(self, local):
        return self.comm.allreduce_sum_i64(np.asarray(local, dtype=np.int64))


def cpu_baseline(ext, el, bo, bl, wl_bytes, max_ed, target_s=15.0):
    """The oracle's scalar per-pair DP + fold (oracle/oracle_ed.c, a port of the reference's
    per-call structure without the interpreter) on the host cores, bounded sample."""
    from oracle import oracle as o
    threads = min(16, os.cpu_count() or 1)
    n0 = 2000
    t0 = time.time()
    o.match_rows(ext[:n0], el[:n0], bo[:n0], bl[:n0], wl_bytes, max_ed, threads=threads)
    dt = max(time.time() - t0, 1e-3)
    n = int(min(ext.shape[0], max(n0, n0 * target_s / dt)))
    t0 = time.time()
    hit, _, _, flag = o.match_rows(ext[:n], el[:n], bo[:n], bl[:n], wl_bytes, max_ed, threads=threads)
    dt = time.time() - t0
    return dict(value=n / dt, unit="reads/s", cores=threads, kind="port",
                sample=f"first {n} rows of the same 1M-row workload (shortcut on), {dt:.1f} s, "
                       f"{int(((flag & 3) == 0).sum())} rows scored x {len(wl_bytes[1]) - 1} barcodes"), (hit, n)


def newest_profile(rows, cells, max_ed):
    """The newest committed profiles/*_pmc.json taken on THIS workload (tools/profile.sh writes the workload into it;
    files from round 1 carry none and were all taken on the default one)."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc.json")), reverse=True):
        with open(fn) as f:
            prof = json.load(f)
        w = prof.get("workload") or {"rows": 1_000_000, "cells": 5000, "max_ed": 2}
        if (w.get("rows"), w.get("cells"), w.get("max_ed")) == (rows, cells, max_ed):
            return fn, prof
    return None, None


def profile_counters(prof, kernels, cls):
    """Sum over `kernels` of the per-launch PMC means of dispatch class `cls`: executed VALU wave-instructions, HBM bytes
    (gfx950 corrections per MI355X_MICROARCH.md 'HBM': FETCH_SIZE doubled, WRITE_SIZE exact, both KiB) and the
    instruction-weighted lane utilisation (active lanes per issued VALU lane slot, None when the profile has no lanes pass)."""
    insts, traffic, us, lanes, per_kernel = 0.0, 0.0, 0.0, 0.0, {}
    for k in kernels:
        ent = prof["kernels"].get("bz::" + k, {}).get("classes", {})
        c = ent.get(cls) or ent.get("all")
        if not c or "SQ_INSTS_VALU" not in c or "avg_us" not in c:
            return None
        insts += c["SQ_INSTS_VALU"]
        us += c["avg_us"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            traffic += (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        if lanes is not None and "lane_utilisation" in c:
            lanes += c["lane_utilisation"] * c["SQ_INSTS_VALU"]
            per_kernel[k] = c["lane_utilisation"]
        else:
            lanes = None
    return insts, traffic or None, us, (lanes / insts if lanes and insts else None), per_kernel


def files_leg(a, threads=16):
    """Step 3 over FILES on this host's cores (what a user of the drop-in sees): FASTQ + putative_bc.csv ->
    demultiplexed FASTQ through assign_read's native host, plain and `.gz` (the reference's default output name,
    config.py:45).  PCIe-, file-system- and host-CPU-inclusive; reported beside, never as, `value`."""
    import tempfile
    from blaze_amd import read_assignment as ra
    n = a.file_reads
    out = {"reads": n, "cells": 10_000, "umi_len": 10, "host_threads": threads,
           "shape": "BASELINE configs[4]: 5v2-shaped rows (10-nt UMI) vs 10,000 cells, trim + restrand, 4 FASTQ files, page cache warm"}
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        fns, pc, wl, fq_bytes = synth.write_file_job(d, n, 10_000)
        out["fastq_bytes_in"] = fq_bytes
        kw = dict(fastq_fns=fns, putative_bc_csv=pc, whitelsit_csv=wl, max_ed=2, batchsize=1000, restrand=True, n_process=threads,
                  host="native")
        ra.assign_read(fastq_out=os.path.join(d, "warm.fastq"), **kw)          # page cache + engine warm
        for name, suffix, reps, level in (("plain", ".fastq", 3, None), ("gz", ".fastq.gz", 3, None), ("gz_level9", ".fastq.gz", 1, 9)):
            best = None
            for _ in range(reps):
                fo = os.path.join(d, "out" + suffix)
                t0 = time.perf_counter()
                got = ra.assign_read(fastq_out=fo, gz_level=level, **kw)
                dt = time.perf_counter() - t0
                if best is None or dt < best[0]:
                    best = (dt, got, os.path.getsize(fo))
            dt, got, size = best
            out[name] = {"reads_per_s": n / dt, "s": dt, "fastq_in_GB_per_s": fq_bytes / dt / 1e9, "bytes_out": size,
                         "assigned": int(got[0]), "total": int(got[1])}
        # a nanopore run's shape: many gzipped input files (inflated ahead of the reader by a thread pool), `.gz` output
        import subprocess
        from concurrent.futures import ThreadPoolExecutor
        gdir = os.path.join(d, "gzin")
        os.makedirs(gdir)
        gfns, gpc, gwl, gbytes = synth.write_file_job(gdir, n, 10_000, n_files=32)
        with ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda f: subprocess.check_call(["gzip", "-1", f]), gfns))
        gkw = dict(kw, fastq_fns=[f + ".gz" for f in gfns], putative_bc_csv=gpc, whitelsit_csv=gwl)
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            got = ra.assign_read(fastq_out=os.path.join(d, "gzin_out.fastq.gz"), **gkw)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out["gz_in_gz_out"] = {"reads_per_s": n / best, "s": best, "input_files": 32, "assigned": int(got[0]), "total": int(got[1]),
                               "note": "32 gzipped FASTQ files in (gzip -1), .gz out: the shape of a real run"}
        out["gz"]["encoder"] = "FASTQ-aware deflate blocks (fastq_deflate.h), the default" if "BLAZE_AMD_GZ_LEVEL" not in os.environ else f"zlib level {os.environ['BLAZE_AMD_GZ_LEVEL']} (env)"
        out["gz_level9"]["zlib_level"] = int(os.environ.get("BLAZE_AMD_GZ_LEVEL", "9"))
        out["gz_level9"]["note"] = "the reference's gzip.compress setting (read_assignment.py:244-245)"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--cells", type=int, default=5000)
    ap.add_argument("--max-ed", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-two-slots", action="store_true",
                    help="skip the overlapped two-slot leg (tools/profile.sh: keeps every traced launch solo)")
    ap.add_argument("--no-high-error", action="store_true", help="skip the leg with SURVEY 8d's high-error reads")
    ap.add_argument("--no-files", action="store_true", help="skip the step-3-over-files leg (N=1 only)")
    ap.add_argument("--file-reads", type=int, default=500_000)
    ap.add_argument("--allow-shared-gpu", action="store_true",
                    help="rehearsal only: let several ranks share one physical GPU (the line then says n_gpus = distinct GPUs)")
    a = ap.parse_args()
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")

    launched = "WORLD_SIZE" in os.environ
    if not launched and a.gpus > 1:
        # no launcher around us: become one, before anything here touches a GPU
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    rank, world, local_rank = dist.env_rank_world()
    if world != a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to print a line "
                         f"that would be read as {a.gpus} GPUs\n")
        sys.exit(2)

    from blaze_amd.engine import device_count, device_identity
    stub = bool(os.environ.get("BLAZE_AMD_BENCH_ENGINE"))
    n_dev = 1 if stub else device_count()     # a launcher may narrow each rank's view to its own GPU
    device = local_rank % max(n_dev, 1)
    eng, stub = make_engine(device)
    info = {"stand-in": True} if stub else device_info(device)
    # Which PHYSICAL GPU does every rank hold?  A local device index says nothing (a launcher that narrows visibility gives
    # every rank "device 0"): all-gather the PCI address + UUID and count the distinct ones.  Fewer GPUs than --gpus would
    # make the line read as a scaling point it is not: refuse, on every rank, unless this is a declared rehearsal.
    ident = eng.device_identity() if stub else device_identity(device)
    idents = dist.get_hub().allgather(ident) if world > 1 else [ident]
    n_distinct = len(set(idents))
    if n_distinct < a.gpus and not a.allow_shared_gpu:
        if rank == 0:
            sys.stderr.write(f"bench.py: --gpus {a.gpus} but the {world} ranks hold only {n_distinct} distinct GPU(s) {sorted(set(idents))}: "
                             f"refusing to print a line that would be read as {a.gpus} GPUs (--allow-shared-gpu for a rehearsal)\n")
        sys.exit(2)
    # librccl prints a version banner on stdout when it initialises: keep stdout for the one JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        plumb = Plumbing(eng, rank, world, try_rccl=not stub)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    # ---- synthetic workload, a different read shard per rank, same whitelist
    full = synth.full_whitelist_codes(400_000, seed=1)       # stand-in pool for the 6.8M 10x list
    cells = synth.cell_whitelist_codes(a.cells, full, seed=2)
    rows = synth.make_rows(a.rows, cells, full, umi_len=12, seed=3 + 1000 * rank)
    ext, el, bo, bl = synth.pack_ext(rows, stride=24)        # 24 B/row: the tightest packed layout for 20-nt extensions
    wl_ascii = synth.codes_to_bytes(cells)
    eng.set_whitelist_codes16(wl_ascii)
    eng.stage(0, ext, el, bo, bl)
    eng.sync(0)

    def timed_steps(steps, slots=(0,)):
        """barrier + sync | K steps | sync + barrier; the clock stops at this rank's own sync (the
        max over ranks is taken by the caller), so the cost of the closing barrier is not billed to the steps."""
        plumb.barrier()
        for s in slots:
            eng.sync(s)
        t0 = time.perf_counter()
        for k in range(steps):
            eng.run(slots[k % len(slots)], a.max_ed)
        for s in slots:
            eng.sync(s)
        dt = time.perf_counter() - t0
        plumb.barrier()
        return dt

    def timed_blocks(steps, slots=(0,), reset_counts=False):
        """The timed region as the contract states it (K steps between barriers + syncs, max over ranks).  A block shorter
        than 50 ms is at the mercy of one scheduling hiccup, so it is then run five times in all and the MEDIAN block is
        the one reported (min and max beside it); every block is exactly K steps, so steps x ms_per_step still is the
        region's length.  The decision is taken on the max over ranks: the same on every rank."""
        blocks = []
        for b in range(5):
            if reset_counts:
                eng.counts_reset()                       # the counts all-reduced below are those of ONE block of K steps
            blocks.append(plumb.max(timed_steps(steps, slots)))
            if blocks[0] >= 0.050:
                break
        return sorted(blocks)[len(blocks) // 2], blocks

    def kernel_avg(reps):
        """Average HIP-event durations of `reps` further launches (events on the slot's own stream)."""
        acc = {}
        eng.set_option(_lib.BZ_OPT_PROFILE, 1)
        for _ in range(reps):
            eng.run(0, a.max_ed)
            t = eng.timing(0)
            for k, v in t.items():
                acc[k] = acc.get(k, 0.0) + float(v)
        eng.set_option(_lib.BZ_OPT_PROFILE, 0)
        return {k: v / reps for k, v in acc.items()}

    # ---- product path: exact-match shortcut on (reference semantics, read_assignment.py:61-62),
    #      seeded scan for the rows that need scoring
    eng.set_option(_lib.BZ_OPT_SHORTCUT, 1)
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
    for _ in range(a.warmup):
        eng.run(0, a.max_ed)
    eng.sync(0)
    dt, dt_blocks = timed_blocks(a.steps, reset_counts=True)
    ms_per_step = dt / a.steps * 1e3
    value = a.rows * world / (dt / a.steps)

    # ---- the one collective: per-barcode counts + [assigned, total]
    counts = plumb.rccl.allreduce_counts_on_device() if plumb.rccl else plumb.sum_counts(eng.counts())
    assigned_total, rows_total = int(counts[-2]), int(counts[-1])

    # ---- the engine's double buffering with resident rows: the same K steps alternating over both slots
    #      (two streams; kernels of consecutive steps overlap).  Reported beside `value`, which stays one slot so
    #      that the per-kernel durations below and in profiles/ describe the launches of the timed region.
    two_slots = None
    if not a.no_two_slots:
        eng.stage(1, ext, el, bo, bl)
        eng.run(1, a.max_ed)
        eng.sync(1)
        dt2, dt2_blocks = timed_blocks(a.steps, slots=(0, 1))
        two_slots = {"reads_per_s": a.rows * world / (dt2 / a.steps), "ms_per_step": dt2 / a.steps * 1e3, "timed_blocks": len(dt2_blocks),
                     "reads_per_s_min": a.rows * world / (max(dt2_blocks) / a.steps), "reads_per_s_max": a.rows * world / (min(dt2_blocks) / a.steps),
                     "note": "same K steps, alternating over the engine's two slots (streams): what bz_assign_files does"}

    # ---- SURVEY 8d's second error setting (sub 3 %, ins 1 %, del 2 %: ~37 % of the barcode rows stay exact instead of
    #      ~72 %, so about twice the rows are scored), same K steps on slot 0; reported beside `value`
    high_error = None
    if not a.no_high_error:
        rows_he = synth.make_rows(a.rows, cells, full, umi_len=12, sub=0.03, ins=0.01, dele=0.02, seed=5 + 1000 * rank)
        eng.stage(0, *synth.pack_ext(rows_he, stride=24))
        for _ in range(max(2, a.warmup)):
            eng.run(0, a.max_ed)
        eng.sync(0)
        dt_he, _ = timed_blocks(a.steps)
        t_he = eng.timing(0)
        high_error = {"reads_per_s": a.rows * world / (dt_he / a.steps), "ms_per_step": dt_he / a.steps * 1e3,
                      "rows_scored_per_step": int(t_he["rows_bitparallel"]), "errors": rows_he["params"]}
        del rows_he
        eng.stage(0, ext, el, bo, bl)
        eng.run(0, a.max_ed)
        eng.sync(0)

    reps = max(3, min(a.steps, 10))
    t_short = kernel_avg(reps)                            # same launches as the timed region, timed per kernel

    # ---- every non-blank row scored (shortcut off): seeded, then brute force (the all-pairs roofline leg)
    eng.set_option(_lib.BZ_OPT_SHORTCUT, 0)
    eng.run(0, a.max_ed)
    t_all_seeded = kernel_avg(reps)
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, 1)
    eng.run(0, a.max_ed)
    t_all_brute = kernel_avg(reps)
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
    eng.set_option(_lib.BZ_OPT_SHORTCUT, 1)
    probe_ops, _ = eng.valu_peak_probe(20000)

    # ---- the boundary as a caller with host arrays uses it: rows in host memory -> bz_assign_submit (H2D, kernels,
    #      D2H) -> bz_assign_wait -> results in host memory, two slots in flight.  PCIe-inclusive: `value_boundary`,
    #      reported beside, never as, `value`.  Twice: pageable numpy arrays (one extra host copy into the engine's
    #      pinned staging) and arrays in pinned memory from bz_host_alloc (DMA straight from / to the caller's buffers).
    def boundary_rate(submit, wait):
        n_e2e = max(6, min(a.steps, 20))
        for k in range(2):
            submit(k)
        wait(0), wait(1)
        t0 = time.perf_counter()
        for k in range(n_e2e):
            if k >= 2:
                wait(k % 2)
            submit(k % 2)
        for k in range(2):
            wait((n_e2e + k) % 2)
        return a.rows * n_e2e / (time.perf_counter() - t0)

    pageable_rate = boundary_rate(lambda k: eng.submit(k, ext, el, bo, bl, a.max_ed), lambda k: eng.wait(k))
    boundary = {"pageable_reads_per_s": pageable_rate}
    pinned = getattr(eng, "pinned_rows", None)
    if pinned is not None:
        bufs = [pinned(ext, el, bo, bl) for _ in range(2)]
        pinned_rate = boundary_rate(lambda k: eng.submit_pinned(k, bufs[k], a.max_ed), lambda k: eng.wait_pinned(k, bufs[k]))
        for b in bufs:
            b.free()
        boundary["pinned_reads_per_s"] = pinned_rate
    eng.stage(0, ext, el, bo, bl)
    eng.sync(0)

    if rank != 0:
        plumb.barrier()
        if plumb.hung:
            os._exit(0)                     # a thread is still inside RCCL's init: do not wait for it at exit
        return

    # ---- roofline objects.  The path is integer-VALU work: the roof is the int32 issue peak, and the
    #      fraction is EXECUTED lane-ops per second over it -- never the pairs the seeds let the scan skip.
    row_bytes = ext.shape[1] + 3 + 7
    if a.max_ed == 2 and t_short["candidates_ms"] > 0:
        scan_kernels, scan_ms = ["k_pair_candidates", f"k_pair_verify<{a.max_ed}>"], t_short["candidates_ms"] + t_short["verify_ms"]
    else:
        scan_kernels, scan_ms = [f"k_scan_seeded<{a.max_ed}>"], t_short["scan_ms"]
    prof_fn, prof = newest_profile(a.rows, a.cells, a.max_ed)
    pairs_decided = t_short["rows_bitparallel"] * a.cells
    roof = {"bound": "valu-int32", "achieved": None, "peak": PEAK_INT32_TOPS, "unit": "T lane-op/s", "frac": None, "traffic": None,
            "kernel": " + ".join(scan_kernels), "kernels": scan_kernels, "launch_ms": scan_ms,
            "rows_scored_per_launch": t_short["rows_bitparallel"],
            "note": "dominant kernels of the timed region (candidate generation + verify).  achieved = executed VALU lane-ops per "
                    "launch (rocprofv3 SQ_INSTS_VALU x 64 of the same launches on the same seeded workload, from `profile`) / "
                    "the launch duration measured live here with HIP events; frac = achieved / the int32 issue peak.  What "
                    "bounds these kernels instead of the VALU: instruction issue shared with scalar + LDS traffic and L2 "
                    "gathers (DESIGN.md 4.2)"}
    # the committed counters describe ONE build of the library: joined with live timings only while its hash matches
    src_sha = _lib.source_sha256()
    prof_ok = bool(prof) and not stub and prof.get("source_sha256") == src_sha
    roof["source_sha256"] = src_sha
    roof["profile_matches_library"] = prof_ok
    if prof and not prof_ok:
        roof["profile"] = os.path.relpath(prof_fn, ROOT)
        roof["note"] += ("  THE COMMITTED PROFILE WAS TAKEN ON ANOTHER BUILD OF THE LIBRARY (source_sha256 of blaze_amd/csrc differs): its instruction "
                         "counts are not joined with this run's timings, frac is null; run tools/profile.sh again.")
    pc = profile_counters(prof, scan_kernels, "product") if prof_ok else None
    if pc:
        insts, traffic, prof_us, lane_util, lane_per_kernel = pc
        lane_ops = insts * 64.0
        frac = lane_ops / (scan_ms * 1e-3) / 1e12 / PEAK_INT32_TOPS
        roof.update(achieved=lane_ops / (scan_ms * 1e-3) / 1e12, frac=frac,
                    lane_utilisation=lane_util, lane_utilisation_per_kernel=lane_per_kernel,
                    frac_useful=frac * lane_util if lane_util else None,
                    frac_note="frac counts every ISSUED lane slot (wave-instructions x 64); lane_utilisation = active lanes per issued slot "
                              "(SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64), rocprofv3's VALUUtilization); frac_useful = frac x lane_utilisation",
                    traffic=traffic, profile=os.path.relpath(prof_fn, ROOT), valu_instructions_per_launch=insts,
                    launch_us_in_profile=prof_us,
                    algorithmic_speedup={"pairs_decided_per_launch": pairs_decided, "ops_per_pair_survey_8d": ALG_OPS_PER_PAIR,
                                         "value": pairs_decided * ALG_OPS_PER_PAIR / lane_ops,
                                         "note": "all-pairs work at SURVEY 8d's 340 ops per pair / lane-ops executed: what the pair "
                                                 "seeds save, NOT a roofline fraction"})
    brute_pairs = t_all_brute["rows_bitparallel"] * a.cells
    brute_s = t_all_brute["scan_ms"] * 1e-3
    brute_ach = brute_pairs * BRUTE_INSTR_PER_PAIR / brute_s / 1e12 if brute_s > 0 else None
    roof_brute = {"bound": "valu-int32", "achieved": brute_ach, "peak": PEAK_INT32_TOPS, "unit": "T lane-op/s",
                  "frac": brute_ach / PEAK_INT32_TOPS if brute_ach else None, "traffic": None,
                  "kernel": f"k_scan_bitparallel<{a.max_ed}>", "launch_ms": t_all_brute["scan_ms"],
                  "pairs_per_launch": brute_pairs, "pairs_per_s": brute_pairs / brute_s if brute_s > 0 else None,
                  "instructions_per_pair": BRUTE_INSTR_PER_PAIR,
                  "note": "all-pairs leg (BZ_OPT_SCAN_MODE 1, shortcut off): every (row, barcode) pair runs the packed automaton; "
                          "priced at the 124.5 VALU instructions per pair the kernel's algorithm needs (498 per four barcodes, "
                          "DESIGN.md 4.5), not at SURVEY 8d's 340-op Myers count, which is not a lower bound"}
    pcb = profile_counters(prof, [f"k_scan_bitparallel<{a.max_ed}>"], "all_rows") if prof_ok else None
    roof_brute["profile_matches_library"] = prof_ok
    if pcb:
        roof_brute["traffic"] = pcb[1]
        roof_brute["issued_frac_in_profile"] = pcb[0] * 64.0 / (pcb[2] * 1e-6) / 1e12 / PEAK_INT32_TOPS
        roof_brute["lane_utilisation"] = pcb[3]
        roof_brute["frac_useful"] = roof_brute["frac"] * pcb[3] if (pcb[3] and roof_brute["frac"]) else None
    roof["measured_valu_ceiling_Tops"] = roof_brute["measured_valu_ceiling_Tops"] = probe_ops / 1e12
    step_bytes = a.rows * row_bytes + a.cells * 20
    roof_hbm = {"bound": "hbm", "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / 8000.0, "traffic": None,
                "kernel": "whole step (classify + scan + histogram)", "algorithmic_bytes_per_row": row_bytes,
                "note": "algorithmic HBM bytes of one step (rows in + results out + the whitelist once) over the timed step: the "
                        "step is nowhere near the HBM roof (DESIGN.md 4.2)"}

    best_boundary = max(boundary.values())
    out = {
        "metric": "reads assigned/sec (whole node) vs 5k-cell whitelist; % int roofline",
        "value": value, "unit": "reads/s", "n_gpus": n_distinct, "n_ranks": world, "n_gpus_requested": a.gpus, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "timed_blocks": len(dt_blocks), "value_min": a.rows * world / (max(dt_blocks) / a.steps), "value_max": a.rows * world / (min(dt_blocks) / a.steps),
        "timed_blocks_note": "a timed region of K steps shorter than 50 ms is repeated (5 blocks of exactly K steps each, barriers and syncs "
                             "around every block) and `value` / `ms_per_step` are the MEDIAN block's; counts are those of one block",
        "dtype": "u32", "dtype_note": "32-bit integer bit-vectors: two 16-bit automata per lane register",
        "data": "synthetic" if not stub else "STAND-IN ENGINE (launch-logic test, not a measurement)",
        "config": {"workload": f"configs[1]: synthetic 10x 3v3, {a.rows} nanopore-error reads per GPU vs "
                               f"{a.cells}-cell whitelist, max_ed {a.max_ed}, exact-match shortcut on",
                   "rows_per_gpu": a.rows, "cells": a.cells, "umi_len": 12, "errors": rows["params"],
                   "parallelism": f"read-sharded x{world}, whitelist replicated", "comm": plumb.kind,
                   "launcher": "torchrun / env" if launched and "BLAZE_AMD_RUN_ID" not in os.environ else ("bench.py --gpus" if world > 1 else "single process"),
                   "devices_visible": n_dev, "distinct_gpus": n_distinct, "ranks_per_gpu": max(idents.count(x) for x in set(idents)),
                   "gpu_of_rank": idents, "shared_gpu_rehearsal": bool(n_distinct < world), "device": info},
        "assigned": assigned_total, "rows_counted": rows_total,
        "rows_scored_per_step": t_short["rows_bitparallel"], "rows_generic_per_step": t_short["rows_generic"],
        "rows_redone_per_step": t_short["rows_redo"],
        "kernel_ms": {k: t_short[k] for k in ("total_ms", "classify_ms", "scan_ms", "candidates_ms", "verify_ms", "generic_ms", "hist_ms")},
        "kernel_ms_note": "HIP events on the slot's stream in separate profiled launches (BZ_OPT_PROFILE): each event adds ~6 us of idle GPU, so total_ms > ms_per_step",
        "all_rows_scored": {"seeded": {"reads_per_s": t_all_seeded["rows"] / (t_all_seeded["total_ms"] * 1e-3),
                                       "scan_ms": t_all_seeded["scan_ms"], "rows_scored": t_all_seeded["rows_bitparallel"]},
                            "bruteforce": {"reads_per_s": t_all_brute["rows"] / (t_all_brute["total_ms"] * 1e-3),
                                           "scan_ms": t_all_brute["scan_ms"], "rows_scored": t_all_brute["rows_bitparallel"]}},
        "two_slots_in_flight": two_slots, "high_error_setting": high_error,
        "value_boundary": {"reads_per_s": best_boundary * world, "bytes_per_row": row_bytes,
                           "GB_per_s_over_pcie": best_boundary * row_bytes / 1e9, **boundary,
                           "note": "rows in host memory -> bz_assign_submit / bz_assign_wait -> results in host memory, two "
                                   "slots in flight, per GPU x ranks; PCIe-inclusive (SURVEY 8d 'Timing'), never `value`"},
        "pcie_inclusive_reads_per_s": best_boundary,
        "roofline": roof, "roofline_bruteforce": roof_brute, "roofline_hbm": roof_hbm,
    }
    if not a.no_files and world == 1 and not stub:
        out["value_files"] = files_leg(a)
    if not a.no_cpu_baseline and world == 1 and not stub:
        eng.run(0, a.max_ed)                       # shortcut on again: the rows the oracle sample is checked against
        wl_packed = (np.ascontiguousarray(wl_ascii).reshape(-1), np.arange(a.cells + 1, dtype=np.int64) * 16)
        base, (o_hit, n_chk) = cpu_baseline(ext, el, bo, bl, wl_packed, a.max_ed)
        eng.sync(0)
        hit, *_ = eng.fetch(0)
        base["gpu_matches_oracle_on_sample"] = bool(np.array_equal(hit[:n_chk], o_hit))
        # and against the REAL reference: the first 50,000 rows of this very workload went through the real _match_bc_row in
        # the build container (oracle/make_golden_bench_sample.py -> tests/golden/bench_sample_leftmost.npz, data only)
        ref_npz = os.path.join(ROOT, "tests", "golden", "bench_sample_leftmost.npz")
        if os.path.exists(ref_npz) and (a.rows, a.cells, a.max_ed) == (1_000_000, 5000, 2):
            g = np.load(ref_npz)
            n_ref = int(g["n_rows"])
            base["gpu_matches_real_reference_on_sample"] = {"rows": n_ref, "equal": bool(np.array_equal(hit[:n_ref], g["hit"])),
                                                            "assigned": int((g["hit"] >= 0).sum()), "fixture": "tests/golden/bench_sample_leftmost.npz"}
        # the REAL reference (blaze.read_assignment.assign_read) cannot travel to the GPU box; its figure, measured in the
        # build container by oracle/time_reference.py on 20k reads of this workload's shape, rides along as a quotation
        ref_fn = os.path.join(ROOT, "profiles", "r01f_reference_cpu_B1.json")
        if os.path.exists(ref_fn):
            with open(ref_fn) as f:
                b1 = json.load(f)
            base["reference_quoted"] = {"reads_per_s_1_thread": b1["threads_1"]["reads_per_s"],
                                        "reads_per_s_threads_8": b1["threads_8"]["reads_per_s"], "host_cpus": b1["host_cpus"],
                                        "where": "build container, not this host; " + b1["what"], "file": "profiles/r01f_reference_cpu_B1.json"}
        out["cpu_baseline"] = base
    print(json.dumps(out), flush=True)
    plumb.barrier()
    if plumb.hung:
        os._exit(0)


if __name__ == "__main__":
    main()
