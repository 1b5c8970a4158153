#!/usr/bin/env python3
"""Generate tests/golden/stage1_{leftmost,rightmost}.json by running the REAL reference's stage 1
(/root/reference/blaze/main.py:43-106 get_raw_bc_from_reads over /root/reference/blaze/polyT_adaptor_finder.py)
-- TEST INFRASTRUCTURE ONLY, build container only.  The reference is imported with oracle/shim on sys.path.
Only data is written (reads in, CSV text + pass-code counts out).  Run:  python oracle/make_golden_stage1.py
"""
import json
import os
import subprocess
import sys
from collections import namedtuple

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"

if "ORACLE_TIE_RULE" not in os.environ:
    for tie in ("0", "1"):
        env = dict(os.environ, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1", ORACLE_TIE_RULE=tie)
        subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)

TIE = int(os.environ["ORACLE_TIE_RULE"])
sys.path[:0] = [os.path.join(HERE, "shim"), REF, ROOT]
import numpy as np  # noqa: E402
import blaze.main as rmain  # noqa: E402  (the real reference)
import blaze.helper as rhelper  # noqa: E402

read_tuple = namedtuple('read_tuple', ['id', 'seq', 'q_letter'])
ACGT = "ACGT"
ADPT_FULL = "CTACACGACGCTCTTCCGATCT"
TSO = "TTTCTTATATGGG"


def rnd(rng, n):
    return "".join(ACGT[i] for i in rng.integers(0, 4, n))


def noisy(rng, s, sub, ins, dele):
    out = []
    for c in s:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            c = ACGT[int(rng.integers(0, 4))]
        out.append(c)
        if rng.random() < ins:
            out.append(ACGT[int(rng.integers(0, 4))])
    return "".join(out)


def make_read(rng, kit, k):
    umi_len = 10 if kit in ("3v2", "5v2") else 12
    bc, umi, cdna = rnd(rng, 16), rnd(rng, umi_len), rnd(rng, int(rng.integers(30, 300)))
    err = [(0, 0, 0), (0.01, 0.005, 0.005), (0.04, 0.02, 0.02), (0.1, 0.04, 0.04)][k % 4]
    lead = rnd(rng, int(rng.integers(0, 40)))
    if kit.startswith("3"):
        body = lead + ADPT_FULL + bc + umi + "T" * int(rng.integers(4, 40)) + cdna
    else:
        body = lead + ADPT_FULL + bc + umi + TSO + cdna
    kind = k % 11
    if kind == 7:                         # no adaptor at all
        body = rnd(rng, int(rng.integers(5, 400)))
    elif kind == 8:                       # structure at both ends (chimeric): found on both strands
        body = body + rhelper.reverse_complement(lead + ADPT_FULL + rnd(rng, 16) + rnd(rng, umi_len) + ("T" * 25 if kit.startswith("3") else TSO))
    elif kind == 9:                       # two adaptors upstream of one poly-T / TSO
        body = lead + "CTTCCGATCT" + rnd(rng, 12) + ADPT_FULL + bc + umi + ("T" * 30 if kit.startswith("3") else TSO) + cdna
    elif kind == 10:                      # very short read
        body = body[:int(rng.integers(1, 80))]
    seq = noisy(rng, body, *err)
    if not seq:
        seq = "A"
    if k % 2:
        seq = rhelper.reverse_complement(seq)
    if k % 13 == 0:
        seq = seq.replace("A", "N", 1).lower() if k % 26 == 0 else seq.replace("C", "N", 2)
    lo_q = 0 if k % 17 == 0 else (3 if k % 3 == 0 else 16)          # a third of the reads can fall below minQ
    qual = "".join(chr(33 + int(q)) for q in rng.integers(lo_q, 45, len(seq)))
    return read_tuple(f"read_{kit}_{k}", seq, qual)


def main():
    rng = np.random.default_rng(77 + TIE)
    cases = []
    for kit, n, min_q in (("3v3", 400, 15), ("3v2", 200, 0), ("5v2", 300, 15), ("5v3", 200, 7), ("3v4", 100, 15)):
        umi_len = 10 if kit in ("3v2", "5v2") else 12
        reads, crashing = [], []
        for k in range(n):
            r = make_read(rng, kit, k)
            try:                                       # the reference itself raises on some (very short) reads:
                rmain.get_raw_bc_from_reads([r], min_q=min_q, kit=kit, umi_len=umi_len)
                reads.append(r)
            except ValueError as e:                    # `min() arg is an empty sequence`, polyT_adaptor_finder.py:278
                crashing.append([r.id, r.seq, r.q_letter, str(e)])
        counter, df = rmain.get_raw_bc_from_reads(reads, min_q=min_q, kit=kit, umi_len=umi_len)
        cases.append(dict(kit=kit, umi_len=umi_len, min_q=min_q, reads=[[r.id, r.seq, r.q_letter] for r in reads],
                          expect_csv=df.to_csv(index=False), expect_pass={str(k): int(v) for k, v in counter.items()},
                          reads_the_reference_raises_on=crashing))
        print(kit, dict(counter), 'raises on', len(crashing))
    out = os.path.join(ROOT, "tests", "golden", f"stage1_{'rightmost' if TIE else 'leftmost'}.json")
    with open(out, "w") as f:
        json.dump(dict(generator="oracle/make_golden_stage1.py (real blaze.main.get_raw_bc_from_reads)", tie_rule=TIE, cases=cases), f)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
