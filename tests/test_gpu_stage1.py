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


@pytest.mark.parametrize("kit", ["3v3", "5v2"])
def test_stage1_empty_and_tiny_reads(tie_engine, oracle, tmp_path, kit):
    """Reads of 0, 1, 3 ... nt (an empty sequence line is legal FASTQ): pass code 1, as the oracle's restatement, through the
    batch call and through the file loop, mixed with a read that does carry a barcode."""
    from oracle import stage1
    good = "ACGTTGCA" * 3 + "CTACACGACGCTCTTCCGATCT" + "ACGTACGTACGTACGT" + "ACGTACGTACGT" + ("TTTCTTATATGGG" if kit == "5v2" else "T" * 25) + "GATTACA" * 20
    seqs = ["", "A", "TTT", "T" * 30, good, "", "ACGTACGTACGTACGTACGT"]
    quals = ["I" * len(s) for s in seqs]
    code, strand, start, minq, polyt = putative_bc.search(tie_engine, seqs, quals, kit, 12)
    for i, (s, q) in enumerate(zip(seqs, quals)):
        o = stage1.read_stage1(s, q, kit, 12, 0)
        assert code[i] == o["pass_code"], (i, s)
        if o["pass_code"] == 0:
            assert (chr(strand[i]), start[i], minq[i], polyt[i]) == (o["strand"], o["raw_bc_start"], o["raw_bc_min_q"], o["polyT_end"])
    assert list(code) == [1, 1, 1, 1, 0, 1, 1]
    fq = tmp_path / "tiny.fastq"
    fq.write_text("".join(f"@t{i} x\n{s}\n+\n{q}\n" for i, (s, q) in enumerate(zip(seqs, quals))))
    out = tmp_path / "tiny.csv"
    counts = putative_bc.putative_bc_files([tie_engine], [str(fq)], str(out), kit=kit, umi_len=12, batch_size=3)
    assert counts == {0: 1, 1: 6}
    rows = out.read_text().splitlines()
    assert len(rows) == 8 and rows[1] == "t0,,,,,," and rows[5].startswith("t4,ACGTACGTACGTACGT,")


def _write_fastq(path, reads, gz=False):
    import gzip
    text = "".join(f"@{rid} runid=abc\n{seq}\n+\n{q}\n" for rid, seq, q in reads)
    if gz:
        with gzip.open(path, "wt") as h:
            h.write(text)
    else:
        with open(path, "w") as h:
            h.write(text)


@pytest.mark.parametrize("tag", ["leftmost", "rightmost"])
@pytest.mark.parametrize("batch_size", [3, 50, 1000])
def test_putative_bc_files_native_golden(tie_engine, tmp_path, tag, batch_size):
    """bz_putative_bc_files (C++ reader / packer / csv writer) on the recorded reads, cut into files and batches:
    the bytes must be what per-batch `DataFrame.to_csv` calls over the REAL reference's frames give (int / float /
    blank number cells depend on the batch: small batches exercise all three forms)."""
    import pandas as pd
    from io import StringIO
    with open(os.path.join(GOLD, f"stage1_{tag}.json")) as f:
        g = json.load(f)
    tie_engine.set_option(_lib.BZ_OPT_TIE_RULE, g["tie_rule"])
    for c in g["cases"]:
        reads = c["reads"]
        cut = len(reads) // 3
        fa, fb = str(tmp_path / f"{c['kit']}_a.fastq"), str(tmp_path / f"{c['kit']}_b.fastq.gz")
        _write_fastq(fa, reads[:cut])
        _write_fastq(fb, reads[cut:], gz=True)
        out = str(tmp_path / f"{c['kit']}_{batch_size}.csv")
        counts = putative_bc.putative_bc_files([tie_engine], [fa, fb], out, kit=c["kit"], umi_len=c["umi_len"], min_q=c["min_q"],
                                               batch_size=batch_size, n_threads=4, device_batch_reads=97)
        assert {str(k): v for k, v in counts.items()} == c["expect_pass"], c["kit"]
        # expected text: the recorded frame re-cut into the reference's batches (per file), each written as main.py:361-364 does
        whole = pd.read_csv(StringIO(c["expect_csv"]), dtype=str, keep_default_na=False)
        num = ["putative_bc_min_q", "polyT_end"]
        expect, first = "", True
        for lo_f, hi_f in ((0, cut), (cut, len(reads))):
            for lo in range(lo_f, hi_f, batch_size):
                part = whole.iloc[lo:min(lo + batch_size, hi_f)].copy()
                blank = part["putative_bc_min_q"] == ""
                for col in num:                      # the recorded text came from ONE frame (float cells): redo the batch's dtype
                    vals = [None if b else int(float(v)) for v, b in zip(part[col], blank)]
                    part[col] = pd.Series(vals, index=part.index, dtype=object if blank.all() else None)
                for col in ("putative_bc", "putative_umi", "pre_bc_flanking", "post_umi_flanking"):
                    part[col] = [None if b else v for v, b in zip(part[col], blank)]
                expect += part.to_csv(index=False, header=first)
                first = False
        assert open(out).read() == expect, (c["kit"], batch_size)


def test_putative_bc_files_quoting_ids_and_python_host(tie_engine, tmp_path):
    """Read ids with csv-special characters are quoted as csv.QUOTE_MINIMAL does; native and pandas hosts agree."""
    from types import SimpleNamespace
    from blaze_amd import main as bmain
    with open(os.path.join(GOLD, "stage1_leftmost.json")) as f:
        c = json.load(f)["cases"][0]
    reads = [list(r) for r in c["reads"][:120]]
    reads[0][0] = 'a,b'
    reads[1][0] = 'say"hi"'
    reads[2][0] = '"q'
    fa = str(tmp_path / "x.fastq")
    _write_fastq(fa, reads)
    outs = {}
    for host in ("python", "native"):
        args = SimpleNamespace(fastq_fns=[fa], out_raw_bc_fn=str(tmp_path / f"{host}.csv"), kit_version=c["kit"], umi_len=c["umi_len"],
                               minQ=c["min_q"], batch_size=7, threads=4)
        counts = dict(bmain.search_putative_bc(args, host=host))
        outs[host] = (open(args.out_raw_bc_fn).read(), {k: v for k, v in counts.items() if v})
    assert outs["native"] == outs["python"]
    assert '"a,b",' in outs["native"][0] and '"say""hi""",' in outs["native"][0]


def test_putative_bc_files_errors(tie_engine, tmp_path):
    fa = str(tmp_path / "bad.fastq")
    with open(fa, "w") as h:
        h.write("@r1\nACGT\n+\nII\n")
    with pytest.raises(_lib.BlazeHipError):
        putative_bc.putative_bc_files([tie_engine], [fa], str(tmp_path / "o.csv"), kit="3v3", umi_len=12)
    with pytest.raises(_lib.BlazeHipError):
        putative_bc.putative_bc_files([tie_engine], [str(tmp_path / "missing.fastq")], str(tmp_path / "o.csv"), kit="3v3", umi_len=12)
    empty = str(tmp_path / "empty.fastq")
    open(empty, "w").close()
    assert putative_bc.putative_bc_files([tie_engine], [empty], str(tmp_path / "none.csv"), kit="3v3", umi_len=12) == {}
    assert not os.path.exists(str(tmp_path / "none.csv"))
