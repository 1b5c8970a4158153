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


def file_allgather(tag, rank, world, value, timeout_s=120.0):
    """Fallback plumbing when RCCL cannot be initialised: exchange small JSON values through /tmp."""
    base = dist._rendezvous_path() + f".{tag}"
    with open(f"{base}.{rank}.tmp", "w") as f:
        json.dump(value, f)
    os.replace(f"{base}.{rank}.tmp", f"{base}.{rank}")
    out, t0 = [], time.time()
    for r in range(world):
        while not os.path.exists(f"{base}.{r}"):
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rank {r} never wrote {tag}")
            time.sleep(0.01)
        for _ in range(100):
            try:
                with open(f"{base}.{r}") as f:
                    out.append(json.load(f))
                break
            except (json.JSONDecodeError, OSError):
                time.sleep(0.01)
    return out


class Plumbing:
    """barrier / max / sum over ranks: RCCL through the C ABI, /tmp files if that fails."""

    def __init__(self, eng, rank, world):
        self.rank, self.world, self.kind, self.eng, self.seq = rank, world, "local", eng, 0
        self.rccl = None
        self.hung = False
        if world > 1:
            # RCCL's init is collective and cannot be cancelled: run it on a daemon thread with a deadline, so that a
            # bootstrap that never completes costs 90 s and a file-fallback line instead of the whole measurement
            import threading
            box = {}

            def init():
                try:
                    box["comm"] = dist.RcclComm(eng, rank, world)
                except Exception as e:      # noqa: BLE001  (keep the bench alive, say what happened)
                    box["err"] = e
            t = threading.Thread(target=init, daemon=True)
            t.start()
            t.join(float(os.environ.get("BLAZE_AMD_RCCL_INIT_TIMEOUT", "90")))
            if "comm" in box:
                self.rccl, self.kind = box["comm"], "rccl"
            elif "err" in box:
                self.kind = f"file-fallback ({type(box['err']).__name__}: {box['err']})"[:200]
            else:
                self.hung = True
                self.kind = "file-fallback (RCCL init did not finish in time)"

    def gather(self, value):
        self.seq += 1
        if self.world == 1:
            return [value]
        return file_allgather(f"g{self.seq}", self.rank, self.world, value)

    def barrier(self):
        if self.rccl:
            self.rccl.barrier()
        elif self.world > 1:
            self.gather(0)

    def max(self, x):
        if self.rccl:
            return self.rccl.allreduce_max_f64(x)
        return max(self.gather(float(x)))


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


def attach_profile(roof, roof_brute):
    """`traffic` (HBM bytes per launch) and `issued` (executed VALU lane-ops/s) come from the PMC
    counters, which cannot be read inside this process: tools/profile.sh collects them with rocprofv3
    (separate --pmc passes of this very command) and the newest committed profiles/*_pmc.json is read
    back here.  gfx950 corrections per MI355X_MICROARCH.md 'HBM': FETCH_SIZE is doubled, WRITE_SIZE is
    exact, both are KiB.  A roofline object that spans several kernels sums them."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc.json")))
    if not files:
        return
    with open(files[-1]) as f:
        prof = json.load(f)
    for roof_obj, cls in ((roof, "product"), (roof_brute, "all_rows")):
        traffic, insts, us, found = 0.0, 0.0, 0.0, 0
        for kname in roof_obj["kernels"]:
            ent = prof["kernels"].get("bz::" + kname, {}).get("classes", {})
            c = ent.get(cls) or ent.get("all")
            if not c or "avg_us" not in c:
                continue
            found += 1
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                traffic += (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
            insts += c.get("SQ_INSTS_VALU", 0.0)
            us += c["avg_us"]
        if found != len(roof_obj["kernels"]):
            continue
        roof_obj["profile"] = os.path.relpath(files[-1], ROOT)
        if traffic:
            roof_obj["traffic"] = traffic
        if insts and us:
            lane_ops = insts * 64.0 / (us * 1e-6) / 1e12
            roof_obj["issued"] = {"valu_lane_Tops": lane_ops, "frac_of_peak": lane_ops / PEAK_INT32_TOPS,
                                  "valu_instructions_per_launch": insts, "launch_us_in_profile": us}


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
    a = ap.parse_args()

    rank, world, local_rank = dist.env_rank_world()
    from blaze_amd.engine import device_count
    n_dev = device_count()                    # a launcher may narrow each rank's view to its own GPU
    device = local_rank % max(n_dev, 1)
    eng = Engine(device)
    info = device_info(device)
    # librccl prints a version banner on stdout when it initialises: keep stdout for the one JSON line
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        plumb = Plumbing(eng, rank, world)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    # ---- synthetic workload, a different read shard per rank, same whitelist
    full = synth.full_whitelist_codes(400_000, seed=1)       # stand-in pool for the 6.8M 10x list
    cells = synth.cell_whitelist_codes(a.cells, full, seed=2)
    rows = synth.make_rows(a.rows, cells, full, umi_len=12, seed=3 + 1000 * rank)
    ext, el, bo, bl = synth.pack_ext(rows)
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
    eng.counts_reset()
    dt = plumb.max(timed_steps(a.steps))
    ms_per_step = dt / a.steps * 1e3
    value = a.rows * world / (dt / a.steps)

    # ---- the one collective: per-barcode counts + [assigned, total]
    if plumb.rccl:
        counts = plumb.rccl.allreduce_counts_on_device()
    else:
        local = eng.counts()
        counts = np.sum(plumb.gather(local.tolist()), axis=0) if world > 1 else local
    assigned_total, rows_total = int(counts[-2]), int(counts[-1])

    # ---- the engine's double buffering with resident rows: the same K steps alternating over both slots
    #      (two streams; kernels of consecutive steps overlap).  Reported beside `value`, which stays one slot so
    #      that the per-kernel durations below and in profiles/ describe the launches of the timed region.
    two_slots = None
    if not a.no_two_slots:
        eng.stage(1, ext, el, bo, bl)
        eng.run(1, a.max_ed)
        eng.sync(1)
        dt2 = plumb.max(timed_steps(a.steps, slots=(0, 1)))
        two_slots = {"reads_per_s": a.rows * world / (dt2 / a.steps), "ms_per_step": dt2 / a.steps * 1e3,
                     "note": "same K steps, alternating over the engine's two slots (streams): what bz_assign_files does"}

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

    # ---- boundary as the drop-in uses it: host arrays in, host arrays out, two slots in flight
    #      (PCIe-inclusive; reported beside, never as, `value`)
    n_e2e = max(4, min(a.steps, 10))
    for k in range(2):
        eng.submit(k, ext, el, bo, bl, a.max_ed)
    eng.wait(0), eng.wait(1)
    t0 = time.perf_counter()
    for k in range(n_e2e):
        if k >= 2:
            eng.wait(k % 2)
        eng.submit(k % 2, ext, el, bo, bl, a.max_ed)
    for k in range(min(2, n_e2e)):
        eng.wait((n_e2e + k) % 2)
    pcie_rate = a.rows * n_e2e / (time.perf_counter() - t0)
    eng.stage(0, ext, el, bo, bl)
    eng.sync(0)

    if rank != 0:
        plumb.barrier()
        if plumb.hung:
            os._exit(0)                     # a thread is still inside RCCL's init: do not wait for it at exit
        return

    def roofline(kernels, ms, t, note):
        pairs = t["rows_bitparallel"] * a.cells
        sec = ms * 1e-3
        ach = pairs * ALG_OPS_PER_PAIR / sec / 1e12
        return {"bound": "valu-int32", "achieved": ach, "peak": PEAK_INT32_TOPS, "unit": "Top/s",
                "frac": ach / PEAK_INT32_TOPS, "traffic": None, "kernel": " + ".join(kernels), "kernels": kernels,
                "launch_ms": ms, "algorithmic_ops_per_pair": ALG_OPS_PER_PAIR, "pairs_per_launch": pairs,
                "rows_scored_per_launch": t["rows_bitparallel"], "pairs_per_s": pairs / sec, "note": note}

    row_bytes = ext.shape[1] + 3 + 7
    if a.max_ed == 2 and t_short["candidates_ms"] > 0:
        scan_kernels, scan_ms = ["k_pair_candidates", f"k_pair_verify<{a.max_ed}>"], t_short["candidates_ms"] + t_short["verify_ms"]
    else:
        scan_kernels, scan_ms = [f"k_scan_seeded<{a.max_ed}>"], t_short["scan_ms"]
    roof = roofline(scan_kernels, scan_ms, t_short,
                    "dominant kernels of the timed region (the scan = candidate generation + verify).  `achieved` prices every "
                    "(row, barcode) pair the scan DECIDES at SURVEY 8d's 340 ops; the pair seeds prove ~99.8 % of them cannot "
                    "match without running the automaton, so frac > 1 is algorithmic gain, not an issue rate: the "
                    "executed-instruction rate is `issued` (rocprof SQ_INSTS_VALU), the all-pairs kernel is priced in "
                    "roofline_bruteforce and the whole step against HBM in roofline_hbm")
    roof_brute = roofline([f"k_scan_bitparallel<{a.max_ed}>"], t_all_brute["scan_ms"], t_all_brute,
                          "all-pairs leg (BZ_OPT_SCAN_MODE 1, shortcut off): every pair runs the packed automaton, "
                          "124.5 VALU instructions per pair against SURVEY 8d's 340-op Myers count")
    roof["measured_valu_ceiling_Tops"] = roof_brute["measured_valu_ceiling_Tops"] = probe_ops / 1e12
    attach_profile(roof, roof_brute)
    step_bytes = a.rows * row_bytes + a.cells * 20
    roof_hbm = {"bound": "hbm", "achieved": step_bytes / (ms_per_step * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                "frac": step_bytes / (ms_per_step * 1e-3) / 1e9 / 8000.0, "traffic": None,
                "kernel": "whole step (classify + scan + histogram)", "algorithmic_bytes_per_row": row_bytes,
                "note": "algorithmic HBM bytes of one step (rows in + results out + the whitelist once) over the timed step: the "
                        "step is nowhere near the HBM roof; it is bound by instruction issue, LDS and L2 gathers in the "
                        "candidate kernel and by VALU issue in the verify kernel (DESIGN.md 4.2)"}

    out = {
        "metric": "reads assigned/sec (whole node) vs 5k-cell whitelist; % int roofline",
        "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32", "dtype_note": "32-bit integer bit-vectors: two 16-bit automata per lane register", "data": "synthetic",
        "config": {"workload": f"configs[1]: synthetic 10x 3v3, {a.rows} nanopore-error reads per GPU vs "
                               f"{a.cells}-cell whitelist, max_ed {a.max_ed}, exact-match shortcut on",
                   "rows_per_gpu": a.rows, "cells": a.cells, "umi_len": 12, "errors": rows["params"],
                   "parallelism": f"read-sharded x{world}, whitelist replicated", "comm": plumb.kind,
                   "device": info},
        "assigned": assigned_total, "rows_counted": rows_total,
        "rows_scored_per_step": t_short["rows_bitparallel"], "rows_generic_per_step": t_short["rows_generic"],
        "rows_redone_per_step": t_short["rows_redo"],
        "kernel_ms": {k: t_short[k] for k in ("total_ms", "classify_ms", "scan_ms", "candidates_ms", "verify_ms", "generic_ms", "hist_ms")},
        "kernel_ms_note": "HIP events on the slot's stream in separate profiled launches (BZ_OPT_PROFILE): each event adds ~6 us of idle GPU, so total_ms > ms_per_step",
        "all_rows_scored": {"seeded": {"reads_per_s": t_all_seeded["rows"] / (t_all_seeded["total_ms"] * 1e-3),
                                       "scan_ms": t_all_seeded["scan_ms"], "rows_scored": t_all_seeded["rows_bitparallel"]},
                            "bruteforce": {"reads_per_s": t_all_brute["rows"] / (t_all_brute["total_ms"] * 1e-3),
                                           "scan_ms": t_all_brute["scan_ms"], "rows_scored": t_all_brute["rows_bitparallel"]}},
        "two_slots_in_flight": two_slots,
        "pcie_inclusive_reads_per_s": pcie_rate,
        "roofline": roof, "roofline_bruteforce": roof_brute, "roofline_hbm": roof_hbm,
    }
    if not a.no_cpu_baseline and world == 1:
        eng.run(0, a.max_ed)                       # shortcut on again: the rows the oracle sample is checked against
        wl_packed = (np.ascontiguousarray(wl_ascii).reshape(-1), np.arange(a.cells + 1, dtype=np.int64) * 16)
        base, (o_hit, n_chk) = cpu_baseline(ext, el, bo, bl, wl_packed, a.max_ed)
        eng.sync(0)
        hit, *_ = eng.fetch(0)
        base["gpu_matches_oracle_on_sample"] = bool(np.array_equal(hit[:n_chk], o_hit))
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
