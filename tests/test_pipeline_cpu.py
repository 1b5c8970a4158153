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
                           out_plot_fn=str(tmp_path / "knee_plot.png"), empty_max_count=float("inf"), force_cells=a["force_cells"],
                           do_empty_drops_output=False)       # the empty-drop test needs the device: covered in -m gpu
    counts = bmain.count_high_quality_bc(args)
    cells, ept = whitelisting.get_bc_whitelist(counts, args=args)
    assert ept == []
    assert "".join(k + "\n" for k in cells) == case["expect"]["out_whitelist_fn"]
    png = (tmp_path / "knee_plot.png").read_bytes()            # main.py:286-302: an 8 x 8 inch figure at the default 100 dpi
    assert png[:8] == b"\x89PNG\r\n\x1a\n" and int.from_bytes(png[16:20], "big") == 800 == int.from_bytes(png[20:24], "big")


def _reference_count(fn, min_q, chunk):
    """The reference's statement, literally (/root/reference/blaze/main.py:378-382), with its chunk size as a parameter."""
    from collections import Counter
    import pandas as pd
    c = Counter()
    for df in pd.read_csv(fn, chunksize=chunk):
        c += Counter(df[df.putative_bc_min_q >= min_q].putative_bc.value_counts().to_dict())
    return c


@pytest.mark.parametrize("arrow", [True, False], ids=["arrow_parser", "pandas_chunks"])
@pytest.mark.parametrize("chunk", [7, 1000, 1_000_000])
def test_count_reads_two_columns_like_the_reference_reads_seven(tmp_path, monkeypatch, chunk, arrow):
    """count_high_quality_bc parses only the two columns it needs (Arrow parser or pandas chunks): same counts in the
    same insertion order (which decides the order of whitelist.csv), for any chunking, with blank rows, an all-blank
    chunk, quoted ids and ties in the counts."""
    import numpy as np
    from blaze_amd import synth
    full = synth.full_whitelist_codes(3000, seed=1)
    cells = synth.cell_whitelist_codes(40, full, seed=2)
    rows = synth.make_rows(5000, cells, full, seed=11, blank_frac=0.2)
    rows["blank"][100:121] = True                                   # three whole chunks of 7 are blank
    ids = synth.read_ids(5000, seed=4)
    ids[5] = '"id,with,commas"'
    fn = tmp_path / "putative_bc.csv"
    fn.write_text(synth.csv_text(rows, ids))
    monkeypatch.setattr(bmain, "_COUNT_CHUNK_ROWS", chunk)
    monkeypatch.setattr(bmain, "_COUNT_ARROW_MAX_BYTES", (1 << 30) if arrow else 0)
    for min_q in (0, 15, 41):
        got = bmain.count_high_quality_bc(SimpleNamespace(out_raw_bc_fn=str(fn), minQ=min_q))
        want = _reference_count(str(fn), min_q, chunk)
        assert list(got.items()) == list(want.items())
    empty = tmp_path / "empty.csv"
    empty.write_text(synth.csv_text(rows, ids, 0, 0))
    assert bmain.count_high_quality_bc(SimpleNamespace(out_raw_bc_fn=str(empty), minQ=15)) == _reference_count(str(empty), 15, chunk)
    blank = tmp_path / "blank.csv"
    blank.write_text(synth.csv_text(rows, ids, 100, 121))
    assert bmain.count_high_quality_bc(SimpleNamespace(out_raw_bc_fn=str(blank), minQ=15)) == _reference_count(str(blank), 15, chunk)
