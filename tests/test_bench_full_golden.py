"""bench.py's own workload (BASELINE configs[1]) against the REAL reference, EVERY row: all 1,000,000 rows of the seeded
workload x 5,000 cells went through the real `_match_bc_row` (oracle/make_golden_bench_full.py, both end-index tie rules);
the fixture keeps the sha256 of the "[bc, umi, strand]" triples of every 1000-row frame and the number of rows that got a
barcode.  The HIP path (GPU test, rows packed exactly as bench.py packs them, one 1M-row batch = one bench step) must
reproduce every frame; the oracle's C driver (CPU test) is checked on every 20th frame (50,000 rows: the whole million
takes it minutes).  tests/test_bench_sample_golden.py holds the first 50,000 rows in full."""
import hashlib
import os

import numpy as np
import pytest

from blaze_amd import _lib
from blaze_amd import read_assignment as ra

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TIES = {"leftmost": 0, "rightmost": 1}
from test_bench_sample_golden import _workload as _sample_workload   # the same seeded 1M-row workload, generated once per session


def _workload():
    w = _sample_workload()
    if "wl" not in w:
        w["wl"] = [x.tobytes().decode() for x in w["wl_ascii"]]
    return w


def _frame_check(g, lo, hi, hit, end, flags, base):
    """sha256 + assigned count of rows [lo, hi) from results hit/end/flags[lo - base : hi - base]"""
    w = _workload()
    rows, wl = w["rows"], w["wl"]
    el = w["packed"][1]
    s = lambda m, i: m[i].tobytes().decode() if not rows["blank"][i] else ""
    idx = range(lo, hi)
    bcs, umis, posts = [s(rows["bc"], i) for i in idx], [s(rows["umi"], i) for i in idx], [s(rows["post"], i) for i in idx]
    polyT = [int(rows["polyT_end"][i]) if not rows["blank"][i] else "" for i in idx]
    got = ra._decode_rows(hit[lo - base:hi - base], end[lo - base:hi - base], flags[lo - base:hi - base], el[lo:hi], bcs, umis, posts, polyT, wl)
    text = "".join(f"{b},{u},{st}\n" for b, u, st in got)
    k = lo // int(g["frame"])
    assert int((hit[lo - base:hi - base] >= 0).sum()) == int(g["frame_assigned"][k]), (k, "assigned rows")
    assert hashlib.sha256(text.encode()).hexdigest().encode() == bytes(g["frame_sha"][k]), (k, "triples")


def _gold(tag):
    path = os.path.join(GOLD, f"bench_full_{tag}.npz")
    g = np.load(path)
    assert int(g["n_rows"]) == 1_000_000 and int(g["frame"]) == 1000 and len(g["frame_sha"]) == 1000
    assert hashlib.sha256(b"".join(bytes(x) for x in g["frame_sha"])).hexdigest() == str(g["sha256"])
    return g


@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
def test_oracle_on_every_20th_frame_of_the_bench_workload(oracle, tag):
    g = _gold(tag)
    w = _workload()
    for k in range(7, 1000, 20):
        lo, hi = 1000 * k, 1000 * (k + 1)
        ext, el, bo, bl = (a[lo:hi] for a in w["packed"])
        hit, _, end, flags = oracle.match_rows(ext, el, bo, bl, w["wl"], 2, TIES[tag], threads=8)
        _frame_check(g, lo, hi, hit, end, flags, base=lo)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
def test_hip_on_the_whole_bench_workload(engine, tag):
    g = _gold(tag)
    w = _workload()
    engine.set_option(_lib.BZ_OPT_TIE_RULE, TIES[tag])
    try:
        engine.set_whitelist_codes16(w["wl_ascii"])
        hit, _, end, flags = engine.assign(*w["packed"], 2)          # one 1M-row batch: exactly a bench step
    finally:
        engine.set_option(_lib.BZ_OPT_TIE_RULE, 0)
    assert int((hit >= 0).sum()) == int(g["frame_assigned"].sum())
    for k in range(1000):
        _frame_check(g, 1000 * k, 1000 * (k + 1), hit, end, flags, base=0)
