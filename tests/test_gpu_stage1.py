"""GPU parity of stage 1: blaze_amd.putative_bc.get_raw_bc_from_reads against vectors recorded from the
REAL reference's get_raw_bc_from_reads (both tie rules), and against the oracle restatement on random reads."""
import json
import os
from collections import namedtuple

import numpy as np
import pytest

from blaze_amd import _lib, putative_bc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
read_tuple = namedtuple('read_tuple', ['id', 'seq', 'q_letter'])


@pytest.fixture
def tie_engine(engine):
    yield engine
    engine.set_option(_lib.BZ_OPT_TIE_RULE, 0)


@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
def test_get_raw_bc_from_reads_golden(tie_engine, tag):
    with open(os.path.join(GOLD, f"stage1_{tag}.json")) as f:
        g = json.load(f)
    tie_engine.set_option(_lib.BZ_OPT_TIE_RULE, g["tie_rule"])
    for c in g["cases"]:
        reads = [read_tuple(*r) for r in c["reads"]]
        counter, df = putative_bc.get_raw_bc_from_reads(reads, min_q=c["min_q"], kit=c["kit"], umi_len=c["umi_len"], engine=tie_engine)
        assert df.to_csv(index=False) == c["expect_csv"], c["kit"]
        assert {str(k): int(v) for k, v in counter.items()} == c["expect_pass"], c["kit"]
        for rid, seq, qual, msg in c["reads_the_reference_raises_on"]:
            with pytest.raises(ValueError):
                putative_bc.get_raw_bc_from_reads([read_tuple(rid, seq, qual)], min_q=0, kit=c["kit"], umi_len=c["umi_len"], engine=tie_engine)


@pytest.mark.parametrize("kit", ["3v3", "5v2"])
def test_stage1_random_reads_vs_oracle(tie_engine, oracle, kit):
    """Unstructured reads (poly-T runs, adaptor fragments, TSO fragments at random places, very short reads)."""
    from oracle import stage1
    rng = np.random.default_rng(3 if kit == "3v3" else 4)
    acgt = "ACGT"
    parts = ["CTTCCGATCT", "CTTCCGTCT", "TTTTTTTTTTTT", "TTTT", "TTTCTTATATGGG", "TTTCTATATGG", "AGATCGGAAG", "AAAAAAAAAA", "CCCATATAAGAAA"]
    reads = []
    for k in range(1500):
        s = ""
        for _ in range(int(rng.integers(1, 9))):
            s += parts[rng.integers(0, len(parts))] if rng.random() < 0.5 else "".join(acgt[i] for i in rng.integers(0, 4, rng.integers(1, 60)))
        s = s[:int(rng.integers(1, 500))]
        reads.append((f"r{k}", s, "".join(chr(33 + int(x)) for x in rng.integers(1, 40, len(s)))))
    umi_len = 10 if kit == "5v2" else 12
    for tie in (0, 1):
        tie_engine.set_option(_lib.BZ_OPT_TIE_RULE, tie)
        code, strand, start, minq, polyt = putative_bc.search(tie_engine, [r[1] for r in reads], [r[2] for r in reads], kit, umi_len)
        n_ok = 0
        for i, (rid, s, q) in enumerate(reads):
            try:
                o = stage1.read_stage1(s, q, kit, umi_len, tie)
            except ValueError:
                assert code[i] == -1, (rid, s)
                continue
            assert code[i] == o["pass_code"], (rid, s, code[i], o)
            if o["pass_code"] == 0:
                n_ok += 1
                assert (chr(strand[i]), start[i], minq[i], polyt[i]) == (o["strand"], o["raw_bc_start"], o["raw_bc_min_q"], o["polyT_end"]), (rid, s)
        assert n_ok > 50


def test_stage1_empty_batch_and_bad_input(tie_engine):
    counter, df = putative_bc.get_raw_bc_from_reads([], min_q=15, kit="3v3", umi_len=12, engine=tie_engine)
    assert len(df) == 0 and not counter
    with pytest.raises(ValueError):
        putative_bc.search(tie_engine, ["ACGT"], ["II"], "3v3", 12)
