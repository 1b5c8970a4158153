"""GPU parity of the empty-drop list: blaze_amd.whitelisting.get_bc_whitelist against vectors recorded
from the REAL reference's get_bc_whitelist, and bz_far_from_all against the oracle.  Bit-exact: same barcodes, same order."""
import json
import os

import numpy as np
import pytest

from blaze_amd import _lib, whitelisting

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "empty_drops.json")


def test_get_bc_whitelist_golden(tmp_path, engine):
    with open(GOLD) as f:
        g = json.load(f)
    wl = tmp_path / "full_whitelist.txt"
    wl.write_text("".join(b + "\n" for b in g["full_whitelist"]))
    for c in g["cases"]:
        raw = {k: v for k, v in g["raw_bc_count"][:c["n_items"]]}
        exp = c["expect"]
        cells_bc, ept = whitelisting.get_bc_whitelist(raw, full_bc_whitelist=str(wl), engine=engine, out_plot_fn=str(tmp_path / "knee.png"), **c["kw"])
        assert [[k, int(v)] for k, v in cells_bc.items()] == exp["cells_bc"], c["name"]
        assert list(ept) == exp["ept_bc"], c["name"]


def test_far_from_all_vs_oracle(engine, oracle):
    rng = np.random.default_rng(8)
    acgt = "ACGTN"
    cells = ["".join(acgt[i] for i in rng.integers(0, 4, rng.integers(12, 20))) for _ in range(257)]
    cand = []
    for _ in range(3000):
        s = list(cells[rng.integers(0, len(cells))])
        for _ in range(int(rng.integers(0, 10))):
            op, pos = rng.integers(0, 3), int(rng.integers(0, len(s)))
            if op == 0:
                s[pos] = acgt[rng.integers(0, 5)]
            elif op == 1 and len(s) > 1:
                del s[pos]
            else:
                s.insert(pos, acgt[rng.integers(0, 5)])
        cand.append("".join(s))
    cand += ["", "A", "ACGT" * 10]
    for min_ed in (1, 3, 5, 8):
        got = whitelisting.far_from_all(engine, cand, cells, min_ed)
        assert np.array_equal(got, oracle.far_from_all(cand, cells, min_ed)), min_ed
    assert whitelisting.far_from_all(engine, [], cells).size == 0
    with pytest.raises(_lib.BlazeHipError):
        whitelisting.far_from_all(engine, ["A" * 65], cells)


def test_empty_drop_list_stop_rule(engine, oracle):
    rng = np.random.default_rng(9)
    acgt = "ACGT"
    cells = ["".join(acgt[i] for i in rng.integers(0, 4, 16)) for _ in range(50)]
    cand = ["".join(acgt[i] for i in rng.integers(0, 4, 16)) for _ in range(9000)]
    got = whitelisting.empty_drop_list(cand, cells, engine)
    assert len(got) == 2001 and got == oracle.empty_drop_list(cand, cells)
    assert whitelisting.empty_drop_list(cand[:100], cells, engine, max_n=10) == oracle.empty_drop_list(cand[:100], cells, max_n=10)


def test_far_from_all_16mers_bitparallel_vs_oracle(engine, oracle):
    """16-nt ACGT strings on both sides take the bit-parallel kernel: candidates a few edit operations away from a cell
    barcode (cut or padded back to 16 nt), every threshold around the true distances."""
    rng = np.random.default_rng(12)
    acgt = "ACGT"
    cells = list(dict.fromkeys("".join(acgt[i] for i in rng.integers(0, 4, 16)) for _ in range(700)))
    cand = []
    for _ in range(5000):
        s = list(cells[rng.integers(0, len(cells))])
        for _ in range(int(rng.integers(0, 9))):
            op, pos = rng.integers(0, 3), int(rng.integers(0, len(s)))
            if op == 0:
                s[pos] = acgt[rng.integers(0, 4)]
            elif op == 1 and len(s) > 1:
                del s[pos]
            else:
                s.insert(pos, acgt[rng.integers(0, 4)])
        cand.append(("".join(s) + "ACGTACGTACGTACGT")[:16])
    for min_ed in (1, 2, 4, 5, 6, 9):
        got = whitelisting.far_from_all(engine, cand, cells, min_ed)
        ref = oracle.far_from_all(cand, cells, min_ed)
        assert np.array_equal(got, ref), (min_ed, int((got != ref).sum()))
        assert 0 < got.sum() < len(cand) or min_ed in (1, 9)
