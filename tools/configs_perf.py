"""Every BASELINE.json config's per-GPU shard through the product path, rows resident in HBM (the same timed region
as bench.py: K back-to-back bz_assign_run on one slot, clock stopped at the stream's sync).  Not bench.py's line
(that is configs[1] only); this fills the per-config table of DESIGN.md section 6 / BASELINE.md.

  configs[1]  1M x 5,000 cells, 12-nt UMI                       one GPU
  configs[2]  10M x 20,000 cells, 10-nt UMI                     shards of 10M / 5M / 2.5M / 1.25M rows (1 / 2 / 4 / 8 GPUs)
  configs[3]  10M x 6,794,880 barcodes (full 10x list stand-in)  shards of 10M and 1.25M rows
  configs[4]  200M x 10,000 cells, 10-nt UMI                    shard of 25M rows (8 GPUs), kernels only (files: tools/e2e_*)

usage: python tools/configs_perf.py [out.json] [--skip-full]
"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from blaze_amd import synth  # noqa: E402
from blaze_amd.engine import Engine  # noqa: E402


def workload(n_rows, cells, full, umi_len, seed=3, chunk=500_000):
    def part(k):
        lo = k * chunk
        return synth.pack_ext(synth.make_rows(min(chunk, n_rows - lo), cells, full, umi_len=umi_len, seed=seed + 1000 * k))
    with ThreadPoolExecutor(8) as pool:
        parts = list(pool.map(part, range((n_rows + chunk - 1) // chunk)))
    return tuple(np.concatenate([p[i] for p in parts]) for i in range(4))


def timed(eng, arrays, n, steps):
    ext, el, bo, bl = (a[:n] for a in arrays)
    eng.stage(0, ext, el, bo, bl)
    eng.run(0, 2)
    eng.sync(0)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run(0, 2)
    eng.sync(0)
    dt = (time.perf_counter() - t0) / steps
    hit = eng.fetch(0)[0]
    t = eng.timing(0)
    return {"rows": n, "ms_per_step": dt * 1e3, "reads_per_s": n / dt, "rows_scored": int(t["rows_bitparallel"]),
            "rows_handed_over": int(t["rows_redo"]), "assigned": int((hit >= 0).sum())}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_fn = args[0] if args else None
    res = {"note": "rows resident in HBM, one slot, product path (max_ed 2, shortcut on); per-GPU shard sizes of each config"}
    eng = Engine(0)
    small_full = synth.full_whitelist_codes(400_000, seed=1)

    def leg(name, n_cells, umi_len, shards, steps, full_list=False):
        t0 = time.time()
        if full_list:
            full = synth.full_whitelist_codes(synth.FULL_WHITELIST_SIZE, seed=1)
            cells = synth.cell_whitelist_codes(5000, full, seed=2)
            wl = full
        else:
            full = small_full
            cells = synth.cell_whitelist_codes(n_cells, full, seed=2)
            wl = cells
        arrays = workload(max(shards), cells, full, umi_len)
        gen_s = time.time() - t0
        t0 = time.time()
        eng.set_whitelist_codes16(synth.codes_to_bytes(wl))
        wl_s = time.time() - t0
        res[name] = {"barcodes": int(wl.size), "umi_len": umi_len, "generate_s": round(gen_s, 1), "set_whitelist_s": round(wl_s, 2),
                     "shards": [timed(eng, arrays, n, steps) for n in shards]}
        print(name, json.dumps(res[name]), flush=True)

    leg("configs[1] 1M x 5k", 5_000, 12, [1_000_000], 100)
    leg("configs[2] 10M x 20k", 20_000, 10, [10_000_000, 5_000_000, 2_500_000, 1_250_000], 10)
    leg("configs[4] 200M x 10k (kernels only)", 10_000, 10, [25_000_000], 5)
    if "--skip-full" not in sys.argv:
        leg("configs[3] 10M x 6.8M", 0, 12, [10_000_000, 1_250_000], 2, full_list=True)
    if out_fn:
        with open(out_fn, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
