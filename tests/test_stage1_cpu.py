"""CPU: the oracle's restatement of stage 1 (oracle/stage1.py) against vectors recorded from the REAL reference's
get_raw_bc_from_reads (oracle/make_golden_stage1.py), both end-index tie rules."""
import json
import os
from io import StringIO

import pandas as pd
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
COLS = ['read_id', 'putative_bc', 'putative_bc_min_q', 'putative_umi', 'polyT_end', 'pre_bc_flanking', 'post_umi_flanking']


def rows_to_csv(rows):
    return pd.DataFrame(rows, columns=COLS).to_csv(index=False)


@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
def test_oracle_stage1_matches_reference(oracle, tag):
    from oracle import stage1
    with open(os.path.join(GOLD, f"stage1_{tag}.json")) as f:
        g = json.load(f)
    n_ok = 0
    for c in g["cases"]:
        counter, rows = stage1.get_raw_bc_from_reads([tuple(r) for r in c["reads"]], c["min_q"], c["kit"], c["umi_len"], g["tie_rule"])
        assert rows_to_csv(rows) == c["expect_csv"], c["kit"]
        assert {str(k): v for k, v in counter.items()} == c["expect_pass"], c["kit"]
        n_ok += c["expect_pass"].get("0", 0)
        for rid, seq, qual, msg in c["reads_the_reference_raises_on"]:
            with pytest.raises(ValueError):
                stage1.read_stage1(seq, qual, c["kit"], c["umi_len"], g["tie_rule"])
    assert n_ok > 100
