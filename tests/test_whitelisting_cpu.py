"""CPU: the oracle's restatement of the empty-drop loop (main.py:244-256) against vectors recorded from the REAL
reference's get_bc_whitelist (oracle/make_golden_whitelist.py)."""
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "empty_drops.json")


def load_cases():
    with open(GOLD) as f:
        g = json.load(f)
    for c in g["cases"]:
        c["raw"] = {k: v for k, v in g["raw_bc_count"][:c["n_items"]]}
    return g


def test_oracle_empty_drop_list_matches_reference(oracle):
    g = load_cases()
    n = 0
    for c in g["cases"]:
        exp = c["expect"]
        if "error" in exp or not exp["ept_bc"]:
            continue
        cells = [k for k, _ in exp["cells_bc"]]
        full = set(g["full_whitelist"])
        raw = {k: v for k, v in c["raw"].items() if k in full}
        limit = min(min(v for _, v in exp["cells_bc"]), c["kw"].get("empty_max_count", np.inf))
        cand = [k for k, v in raw.items() if v < limit]
        assert oracle.empty_drop_list(cand, cells) == exp["ept_bc"], c["name"]
        n += 1
    assert n >= 5


def test_oracle_far_from_all_is_the_pairwise_minimum(oracle):
    rng = np.random.default_rng(4)
    acgt = "ACGT"
    cells = ["".join(acgt[i] for i in rng.integers(0, 4, 16)) for _ in range(40)]
    cand = []
    for _ in range(300):
        s = list(cells[rng.integers(0, len(cells))])
        for _ in range(int(rng.integers(0, 9))):
            s[rng.integers(0, 16)] = acgt[rng.integers(0, 4)]
        cand.append("".join(s))
    far = oracle.far_from_all(cand, cells, 5)
    for k, f in zip(cand, far):
        assert f == (min(oracle.edit_distance(k, x, 5) for x in cells) >= 5)
    assert 0 < far.sum() < len(cand)
