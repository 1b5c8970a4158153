"""bench.py's own workload (BASELINE configs[1]) against the REAL reference: the first 50,000 rows of the seeded 1M-row x
5,000-cell workload went through the real `_match_bc_row` (oracle/make_golden_bench_sample.py, both end-index tie rules);
here the oracle's C driver (CPU) and the HIP path (GPU, rows packed exactly as bench.py packs them) must give the same
barcode index, UMI and strand for every one of them."""
import hashlib
import os

import numpy as np
import pytest

from blaze_amd import _lib, synth
from blaze_amd import read_assignment as ra

GOLD = os.path.join(os.path.dirname(__file__), "golden")
TIES = {"leftmost": 0, "rightmost": 1}
_w = {}


def _workload():
    if not _w:
        full = synth.full_whitelist_codes(400_000, seed=1)           # bench.py rank 0
        cells = synth.cell_whitelist_codes(5000, full, seed=2)
        rows = synth.make_rows(1_000_000, cells, full, umi_len=12, seed=3)
        _w["rows"], _w["wl_ascii"] = rows, synth.codes_to_bytes(cells)
        _w["packed"] = synth.pack_ext(rows, stride=24)
    return _w


def _check(tag, assign):
    g = np.load(os.path.join(GOLD, f"bench_sample_{tag}.npz"))
    n = int(g["n_rows"])
    w = _workload()
    rows = w["rows"]
    ext, el, bo, bl = (a[:n] for a in w["packed"])
    wl = [x.tobytes().decode() for x in w["wl_ascii"]]
    hit, _, end, flags = assign(ext, el, bo, bl, wl)
    bad = np.flatnonzero(hit != g["hit"])
    assert bad.size == 0, (bad[:5], hit[bad[:5]], g["hit"][bad[:5]])
    s = lambda m, i: m[i].tobytes().decode() if not rows["blank"][i] else ""
    cols = ([s(rows["bc"], i) for i in range(n)], [s(rows["umi"], i) for i in range(n)], [s(rows["post"], i) for i in range(n)])
    polyT = [int(rows["polyT_end"][i]) if not rows["blank"][i] else "" for i in range(n)]
    got = ra._decode_rows(hit, end, flags, el, cols[0], cols[1], cols[2], polyT, wl)
    text = "".join(f"{b},{u},{st}\n" for b, u, st in got)
    assert hashlib.sha256(text.encode()).hexdigest() == str(g["sha256"])
    assert [x[1].encode() for x in got] == list(g["umi"]) and int((hit >= 0).sum()) > 30_000


@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
def test_oracle_on_the_bench_workload_sample(oracle, tag):
    _check(tag, lambda ext, el, bo, bl, wl: oracle.match_rows(ext, el, bo, bl, wl, 2, TIES[tag], threads=8))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
def test_hip_on_the_bench_workload_sample(engine, tag):
    def assign(ext, el, bo, bl, wl):
        engine.set_option(_lib.BZ_OPT_TIE_RULE, TIES[tag])
        try:
            engine.set_whitelist_codes16(_workload()["wl_ascii"])
            return engine.assign(ext, el, bo, bl, 2)
        finally:
            engine.set_option(_lib.BZ_OPT_TIE_RULE, 0)
    _check(tag, assign)


def _check_max_ed(max_ed, assign):
    g = np.load(os.path.join(GOLD, "bench_sample_max_ed_leftmost.npz"))
    n = int(g["n_rows"])
    w = _workload()
    rows = w["rows"]
    ext, el, bo, bl = (a[:n] for a in w["packed"])
    wl = [x.tobytes().decode() for x in w["wl_ascii"]]
    hit, _, end, flags = assign(ext, el, bo, bl, wl, max_ed)
    bad = np.flatnonzero(hit != g[f"ed{max_ed}_hit"])
    assert bad.size == 0, (max_ed, bad[:5], hit[bad[:5]], g[f"ed{max_ed}_hit"][bad[:5]])
    s = lambda m, i: m[i].tobytes().decode() if not rows["blank"][i] else ""
    cols = ([s(rows["bc"], i) for i in range(n)], [s(rows["umi"], i) for i in range(n)], [s(rows["post"], i) for i in range(n)])
    polyT = [int(rows["polyT_end"][i]) if not rows["blank"][i] else "" for i in range(n)]
    got = ra._decode_rows(hit, end, flags, el, cols[0], cols[1], cols[2], polyT, wl)
    text = "".join(f"{b},{u},{st}\n" for b, u, st in got)
    assert hashlib.sha256(text.encode()).hexdigest() == str(g[f"ed{max_ed}_sha256"]), max_ed


@pytest.mark.parametrize("max_ed", [0, 1, 3, 4, 6])
def test_oracle_on_the_bench_workload_other_max_ed(oracle, max_ed):
    """--max-edit-distance 0 / 1 / 3 / 4 / 6 (parser.py:100) on the first 20,000 rows, against the real _match_bc_row."""
    _check_max_ed(max_ed, lambda ext, el, bo, bl, wl, k: oracle.match_rows(ext, el, bo, bl, wl, k, 0, threads=8))


@pytest.mark.gpu
@pytest.mark.parametrize("max_ed", [0, 1, 3, 4, 6])
def test_hip_on_the_bench_workload_other_max_ed(engine, max_ed):
    def assign(ext, el, bo, bl, wl, k):
        engine.set_whitelist_codes16(_workload()["wl_ascii"])
        return engine.assign(ext, el, bo, bl, k)
    _check_max_ed(max_ed, assign)
