"""Host logic of blaze_amd.main that needs no GPU, against the files the REAL reference's main.main wrote
(tests/golden/pipeline_leftmost.json): the summary text, the high-quality barcode counts and the cell calling."""
import json
import os
import re
from collections import defaultdict
from types import SimpleNamespace

import pytest

from blaze_amd import main as bmain
from blaze_amd import whitelisting

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pipeline_leftmost.json")


def _cases():
    with open(GOLD) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_summary_text(case):
    text = case["expect"]["summary_fn"]
    head = text[:text.index("\nIdentified # of cells")]
    nums = [int(x.replace(",", "")) for x in re.findall(r"^\s+([\d,]+)(?: \(| in which|$)", head, flags=re.M)]
    total, found, hq, none, both, multi = nums
    counts = defaultdict(int, {0: hq, 100: found - hq, 1: none, 2: both, 10: multi})
    assert sum(counts.values()) == total
    assert bmain.bc_search_qc_report(counts, SimpleNamespace(minQ=case["args"]["minQ"])) == head


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_counts_and_cell_calling(tmp_path, case):
    a = case["args"]
    csv, full = tmp_path / "putative_bc.csv", tmp_path / "full.txt"
    csv.write_text(case["expect"]["out_raw_bc_fn"])
    full.write_text("".join(b + "\n" for b in case["full_whitelist"]))
    args = SimpleNamespace(out_raw_bc_fn=str(csv), minQ=a["minQ"], full_bc_whitelist=str(full), expect_cells=a["expect_cells"],
                           count_threshold=a["count_threshold"], high_sensitivity_mode=a["high_sensitivity_mode"],
                           out_plot_fn="unused.png", empty_max_count=float("inf"), force_cells=a["force_cells"],
                           do_empty_drops_output=False)       # the empty-drop test needs the device: covered in -m gpu
    counts = bmain.count_high_quality_bc(args)
    cells, ept = whitelisting.get_bc_whitelist(counts, args=args)
    assert ept == []
    assert "".join(k + "\n" for k in cells) == case["expect"]["out_whitelist_fn"]
