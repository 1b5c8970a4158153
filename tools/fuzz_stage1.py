"""Randomised parity run of stage 1 on the GPU box: k_stage1 (through blaze_amd.putative_bc.search) against the oracle's
per-read restatement (oracle/stage1.py) on freshly drawn reads -- structured ones (adaptor + barcode + UMI + poly-T / TSO
+ cDNA on either strand, with nanopore-style errors at drawn rates, truncated at either end) and unstructured ones
(fragments of the searched patterns at random places), 3' and 5' kits, both tie rules.

usage: python tools/fuzz_stage1.py [seconds] [first_seed]      (test infrastructure: uses oracle/ as the checker)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from blaze_amd import _lib, putative_bc  # noqa: E402
from blaze_amd.engine import Engine  # noqa: E402
from oracle import oracle, stage1  # noqa: E402

ACGT = "ACGT"
COMP = str.maketrans("ACGT", "TGCA")
PARTS = ["CTTCCGATCT", "CTTCCGTCT", "TTTTTTTTTTTT", "TTTT", "TTTCTTATATGGG", "TTTCTATATGG", "AGATCGGAAG", "AAAAAAAAAA", "CCCATATAAGAAA"]


def rand_seq(rng, n):
    return "".join(ACGT[i] for i in rng.integers(0, 4, n))


def noisy(rng, s, p):
    out = []
    for c in s:
        v = rng.random()
        if v < p:
            out.append(ACGT[rng.integers(0, 4)])
        elif v < 1.5 * p:
            out.append(c); out.append(ACGT[rng.integers(0, 4)])
        elif v < 2.0 * p:
            continue
        else:
            out.append(c)
    return "".join(out)


def draw_read(rng, kit5, umi_len, p):
    if rng.random() < 0.3:                                     # unstructured
        s = ""
        for _ in range(int(rng.integers(1, 9))):
            s += PARTS[rng.integers(0, len(PARTS))] if rng.random() < 0.5 else rand_seq(rng, int(rng.integers(1, 60)))
        return s[:int(rng.integers(1, 500))]
    head = rand_seq(rng, int(rng.integers(0, 90))) + "CTACACGACGCTCTTCCGATCT"
    body = rand_seq(rng, 16) + rand_seq(rng, umi_len)
    tail = ("TTTCTTATATGGG" if kit5 else "T" * int(rng.integers(3, 40))) + rand_seq(rng, int(rng.integers(0, 400)))
    s = noisy(rng, head + body + tail, p)
    if rng.random() < 0.5:
        s = s.translate(COMP)[::-1]
    if rng.random() < 0.15:
        s = s[int(rng.integers(0, 60)):]
    if rng.random() < 0.15:
        s = s[:max(1, len(s) - int(rng.integers(0, 200)))]
    if rng.random() < 0.03:
        k = int(rng.integers(0, len(s)))
        s = s[:k] + "N" + s[k + 1:]
    return s or "A"


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle.build()
    eng = Engine(0)
    t_end = time.time() + seconds
    n_cases = n_reads = n_fail = n_found = 0
    last = time.time()
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        kit = str(rng.choice(["3v3", "3v2", "5v2", "5v3"]))
        kit5 = kit.startswith("5")
        umi_len = int(rng.choice([10, 12]))
        tie = int(rng.integers(0, 2))
        p = float(rng.choice([0.0, 0.01, 0.03, 0.08]))
        n = int(rng.choice([1, 64, 300, 700]))
        seqs = [draw_read(rng, kit5, umi_len, p) for _ in range(n)]
        quals = ["".join(chr(33 + int(x)) for x in rng.integers(1, 40, len(s))) for s in seqs]
        eng.set_option(_lib.BZ_OPT_TIE_RULE, tie)
        code, strand, start, minq, polyt = putative_bc.search(eng, seqs, quals, kit, umi_len)
        bad = None
        for i, (s, q) in enumerate(zip(seqs, quals)):
            try:
                o = stage1.read_stage1(s, q, kit, umi_len, tie)
            except ValueError:
                if code[i] != -1:
                    bad = (i, s, int(code[i]), "oracle raises")
                continue
            if code[i] != o["pass_code"]:
                bad = (i, s, int(code[i]), o)
            elif o["pass_code"] == 0:
                n_found += 1
                if (chr(strand[i]), start[i], minq[i], polyt[i]) != (o["strand"], o["raw_bc_start"], o["raw_bc_min_q"], o["polyT_end"]):
                    bad = (i, s, (chr(strand[i]), int(start[i]), int(minq[i]), int(polyt[i])), o)
            if bad:
                break
        n_cases += 1
        n_reads += n
        if bad:
            n_fail += 1
            print(f"FAIL seed {seed} kit {kit} umi {umi_len} tie {tie} p {p}: {bad}", flush=True)
            if n_fail >= 5:
                break
        seed += 1
        if time.time() - last > 30:
            last = time.time()
            print(f"... {n_cases} cases, {n_reads} reads ({n_found} with a barcode), {n_fail} failures", flush=True)
    print(json.dumps({"cases": n_cases, "reads": n_reads, "reads_with_barcode": n_found, "failures": n_fail, "first_seed": seed0,
                      "next_seed": seed, "seconds": seconds}), flush=True)
    sys.exit(1 if n_fail else 0)


if __name__ == "__main__":
    main()
