"""CPU: pin the oracle.  (a) the DP against an independent brute-force definition, (b) the
restated fold / UMI correction / trimming / record format against golden vectors produced by the
REAL reference functions (oracle/make_golden.py), (c) the C batch driver against the Python one."""
import gzip
import json
import os
import random
from io import StringIO

import numpy as np
import pandas as pd
import pytest
from hypothesis import given, settings, strategies as st

GOLD = os.path.join(os.path.dirname(__file__), "golden")
HEADER = "read_id,putative_bc,putative_bc_min_q,putative_umi,polyT_end,pre_bc_flanking,post_umi_flanking\n"
TIES = {"leftmost": 0, "rightmost": 1}


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _frame(csv_body):
    return pd.read_csv(StringIO(HEADER + csv_body)).fillna("")


# ---- (a) sub_edit_distance: definition, tie rules, cut-off, role swap ------------------------
@settings(max_examples=300, deadline=None)
@given(st.text("ACGT", max_size=8), st.text("ACGTN", max_size=12), st.integers(0, 6))
def test_sub_edit_distance_matches_bruteforce(oracle, a, b, max_ed):
    best, ends = oracle.sub_edit_distance_bruteforce(a, b)
    for tie, pick in ((0, min), (1, max)):
        ed, end = oracle.sub_edit_distance(a, b, max_ed, tie)
        if best > max_ed:
            assert (ed, end) == (max_ed + 1, -1)
        else:
            # leftmost: the empty substring (end -1) wins a tie with every real end; rightmost: the largest end
            assert ed == best and end == pick(ends)
    # argument order is free: the shorter one is the pattern (polyT_adaptor_finder.py:53 vs read_assignment.py:72)
    if len(a) != len(b):
        assert oracle.sub_edit_distance(a, b, max_ed) == oracle.sub_edit_distance(b, a, max_ed)


def test_sub_edit_distance_known_answers(oracle):
    w = "ACGTACGTTGCATGCA"
    ext = "CT" + w + "GG"
    assert oracle.sub_edit_distance(w, ext, 2) == (0, 17)          # indel-free hit ends at len(ext)-1-2 (:93)
    assert oracle.sub_edit_distance(w, "CT" + w[:5] + w[6:] + "GGG", 2) == (1, 16)      # one deletion
    assert oracle.sub_edit_distance(w, "CT" + w[:5] + "T" + w[5:] + "G", 2) == (1, 18)  # one insertion
    assert oracle.sub_edit_distance(w, "T" * 20, 2) == (3, -1)     # miss: > cut-off, end -1
    assert oracle.sub_edit_distance("CTTCCGATCT", "AAAACTTCCGATCTAAAA", 2) == (0, 13)   # stage-1 shape
    assert oracle.edit_distance("kitten", "sitting", 5) == 3
    assert oracle.edit_distance("AAAAAAAA", "TTTTTTTT", 5) == 6    # above the cut-off -> max_ed+1


# ---- (b) restated logic vs the real reference ------------------------------------------------
@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
def test_match_rows_golden(oracle, tag):
    gold = _load(f"match_rows_{tag}.json")
    assert "PARITY UNPINNED" in gold["meta"]["sub_edit_distance"]
    n = 0
    for case in gold["cases"]:
        df = _frame(case["csv"])
        for row, expect in zip(df.itertuples(), case["expect"]):
            got = oracle.match_bc_row(row.putative_bc, row.putative_umi, row.polyT_end, row.pre_bc_flanking,
                                      row.post_umi_flanking, case["whitelist"], case["max_ed"], TIES[tag])
            assert got == expect, (case["name"], row, got, expect)
            n += 1
    assert n > 600


def test_fold_is_order_dependent_in_the_golden():
    cases = {c["name"]: c for c in _load("match_rows_leftmost.json")["cases"]}
    assert cases["fold_2_2_1"]["expect"][0][0] == "" and cases["fold_1_2_2"]["expect"][0][0] != ""
    assert cases["fold_1_1_0"]["expect"][0][0] == "" and cases["fold_2_2_0"]["expect"][0][0] != ""
    assert sorted(cases["fold_2_2_1"]["whitelist"]) == sorted(cases["fold_1_2_2"]["whitelist"])


@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
def test_batches_golden(oracle, tag):
    for case in _load(f"batches_{tag}.json")["cases"]:
        df = _frame(case["csv"])
        rows = df.to_dict("records")
        reads = [tuple(r) for r in case["reads"]]
        assigned, data, demul, total = oracle.assign_batch(reads, rows, case["iteration_order"], case["max_ed"],
                                                           case["gz"], case["restrand"], TIES[tag])
        text = gzip.decompress(data).decode() if case["gz"] else data.decode()
        assert assigned == case["expect_cols"]
        assert text == case["expect_fastq"]
        assert (demul, total) == (case["expect_demul"], case["expect_total"])
        assert set(case["iteration_order"]) == set(case["whitelist"]) and len(case["iteration_order"]) < len(case["whitelist"])


def test_revcomp_golden(oracle):
    for c in _load("revcomp.json")["cases"]:
        assert oracle.reverse_complement(c["seq"]) == c["expect"]


# ---- (c) C batch driver == Python restatement ---------------------------------------------------
def test_c_batch_driver_matches_python(oracle):
    from blaze_amd import synth
    full = synth.full_whitelist_codes(20_000, seed=5)
    cells = synth.cell_whitelist_codes(120, full, seed=6)
    rows = synth.make_rows(700, cells, full, sub=0.03, ins=0.015, dele=0.015, seed=7)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl = [w.tobytes().decode() for w in synth.codes_to_bytes(cells)]
    for tie in (0, 1):
        for max_ed in (0, 2, 4):
            hit, ed, end, flag = oracle.match_rows(ext, el, bo, bl, wl, max_ed, tie, threads=4)
            for r in range(rows["n"]):
                s = lambda m: m[r].tobytes().decode()
                if rows["blank"][r]:
                    assert hit[r] == -1 and flag[r] == 2
                    continue
                got = oracle.match_bc_row(s(rows["bc"]), s(rows["umi"]), int(rows["polyT_end"][r]), s(rows["pre"]),
                                          s(rows["post"]), wl, max_ed, tie)
                assert got[0] == (wl[hit[r]] if hit[r] >= 0 else ""), (r, got, hit[r])
                if hit[r] >= 0 and not flag[r] & 1:
                    adj = int(end[r]) - (int(el[r]) - 1 - 2)
                    umi, bc, post = s(rows["umi"]), s(rows["bc"]), s(rows["post"])
                    exp = umi[adj:] + post[:adj] if adj > 0 else (bc[adj:] + umi[:adj] if adj < 0 else umi)
                    assert got[1] == exp


def test_synthetic_generator_shape():
    from blaze_amd import synth
    full = synth.full_whitelist_codes(30_000, seed=1)
    assert np.unique(full).size == full.size
    cells = synth.cell_whitelist_codes(500, full)
    rows = synth.make_rows(20_000, cells, full)
    assert 0.12 < rows["blank"].mean() < 0.16                     # 14 % blank (test_summary.txt:8)
    code = np.zeros(256, np.uint64); code[np.frombuffer(b"ACGT", np.uint8)] = np.arange(4, dtype=np.uint64)
    bc_codes = (code[rows["bc"]] << (2 * (15 - np.arange(16, dtype=np.uint64)))).sum(axis=1).astype(np.uint32)
    exact = np.isin(bc_codes, cells) & ~rows["blank"]
    assert 0.60 < exact.sum() / (~rows["blank"]).sum() < 0.80     # ~0.98^16 * 0.9 of barcodes survive intact
