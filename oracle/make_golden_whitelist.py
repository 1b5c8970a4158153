#!/usr/bin/env python3
"""Generate tests/golden/empty_drops.json by running the REAL reference's get_bc_whitelist
(/root/reference/blaze/main.py:151-284) -- TEST INFRASTRUCTURE ONLY, build container only.

The reference module is imported from /root/reference with oracle/shim on sys.path (stand-in for the absent
fast-edit-distance wheel, routed to oracle_ed.c).  Pinned by these vectors: the candidate order, thresholds,
`>=` / `>` rules and the stop rule (2001 barcodes) AROUND edit_distance; not pinned: edit_distance itself
(third-party; any implementation that is exact below its cut-off gives the same `>= 5` decisions).
Only data is written.  Run:  python oracle/make_golden_whitelist.py
"""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "empty_drops.json")

if os.environ.get("MPLBACKEND") != "Agg":
    env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1", ORACLE_TIE_RULE="0")
    subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)

sys.path[:0] = [os.path.join(HERE, "shim"), REF, ROOT]
import numpy as np  # noqa: E402
import blaze.main as rmain  # noqa: E402  (the real reference)

ACGT = "ACGT"


def rand_bc(rng):
    return "".join(ACGT[i] for i in rng.integers(0, 4, 16))


def mutate(rng, s, n_edits):
    s = list(s)
    for _ in range(n_edits):
        op, pos = int(rng.integers(0, 3)), int(rng.integers(0, len(s)))
        if op == 0:
            s[pos] = ACGT[int(rng.integers(0, 4))]
        elif op == 1 and len(s) > 1:
            del s[pos]
        else:
            s.insert(pos, ACGT[int(rng.integers(0, 4))])
    s = "".join(s)
    return (s + "ACGTACGTACGTACGT")[:16]          # back to 16 nt (adds an edit at the end at most)


def main():
    rng = np.random.default_rng(20240611)
    cells = list(dict.fromkeys(rand_bc(rng) for _ in range(260)))
    background = list(dict.fromkeys(rand_bc(rng) for _ in range(7000)))
    near = list(dict.fromkeys(mutate(rng, cells[int(rng.integers(0, len(cells)))], int(rng.integers(1, 6))) for _ in range(1500)))
    full = list(dict.fromkeys(cells + background + near))
    rng.shuffle(full)
    outside = [rand_bc(rng) for _ in range(300)]                      # not in the 10x list: dropped at :226
    counts = {}
    order = cells + background[:6000] + near + outside
    order = [order[i] for i in rng.permutation(len(order))]
    cell_set, near_set = set(cells), set(near)
    for bc in order:
        if bc in cell_set:
            counts[bc] = int(rng.integers(150, 3000))
        elif bc in near_set:
            counts[bc] = int(rng.integers(1, 12))
        else:
            counts[bc] = int(rng.integers(1, 6))
    cases = []
    with tempfile.TemporaryDirectory() as d:
        wl_fn = os.path.join(d, "full_whitelist.txt")
        with open(wl_fn, "w") as f:
            f.write("".join(b + "\n" for b in full))
        variants = [
            dict(name="expect_cells", kw=dict(exp_cells=250)),
            dict(name="high_sensitivity", kw=dict(exp_cells=250, high_sensitivity_mode=True)),
            dict(name="count_threshold", kw=dict(count_t=150)),
            dict(name="force_cells", kw=dict(force_cell_n=200)),
            dict(name="force_cells_too_many", kw=dict(force_cell_n=100000)),
            dict(name="empty_max_count_2", kw=dict(exp_cells=250, empty_max_count=2)),
            dict(name="no_empty_output", kw=dict(exp_cells=250, output_empty=False)),
            dict(name="few_candidates_no_stop", kw=dict(exp_cells=250), subset=1500),
        ]
        for v in variants:
            items = list(counts.items())
            if "subset" in v:
                items = items[:v["subset"]]
            raw = dict(items)
            try:
                cells_bc, ept = rmain.get_bc_whitelist(raw, full_bc_whitelist=wl_fn, out_plot_fn=os.path.join(d, "knee.png"), **v["kw"])
                exp = dict(cells_bc=[[k, int(c)] for k, c in cells_bc.items()], ept_bc=list(ept))
            except Exception as e:                                     # e.g. force_cells_too_many: min() of ... still fine
                exp = dict(error=type(e).__name__)
            cases.append(dict(name=v["name"], kw=v["kw"], n_items=len(items), expect=exp))
            print(v["name"], {k: (len(x) if isinstance(x, list) else x) for k, x in exp.items()})
    with open(OUT, "w") as f:
        json.dump(dict(generator="oracle/make_golden_whitelist.py (real blaze.main.get_bc_whitelist)", full_whitelist=full,
                       raw_bc_count=[[k, int(c)] for k, c in counts.items()], cases=cases), f)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
