"""TEST INFRASTRUCTURE ONLY -- CPU restatement of BLAZE's stage 1 ("get putative barcodes") per read.

Follows /root/reference/blaze/polyT_adaptor_finder.py (class Read) and /root/reference/blaze/main.py:43-106
(get_raw_bc_from_reads).  Pure Python + the oracle's C sub_edit_distance; pinned by tests/golden/stage1_*.json, which
oracle/make_golden_stage1.py records from the REAL reference.  Parity unpinned: sub_edit_distance's end index on ties
(third-party, see oracle_ed.c) -- it decides `raw_bc_start` here, so goldens exist for both tie rules.
"""
from __future__ import annotations

from collections import Counter

from . import oracle as _o

ADPT_SEQ = 'CTTCCGATCT'            # config.py:12
ADPT_WIN = 200                     # config.py:13
ADPT_MAX_ED = 2                    # config.py:14
AFT_ADPT = (20, 50)                # config.py:19
PLY_T_LEN = 4                      # config.py:22
TSO_SEQ = 'TTTCTTATATGGG'          # config.py:24
BC_SIZE = 16                       # config.py:34
FLANK = 5                          # config.py:46
KITS_3P = ('3v4', '3v3', '3v2')    # polyT_adaptor_finder.py:190


def _adaptor_ends_3p(seq, tie):
    """polyT_adaptor_finder.py:118-144 on the poly-T strand view `seq` (first 200 nt)."""
    n = len(seq)
    # :111-113  T_prop[i] = mean(code[i+1 .. i+4]); indices with 4 T's, then >= d1 (:123)
    d1, d2 = AFT_ADPT
    idx = [i for i in range(max(n - PLY_T_LEN, 0)) if seq[i + 1:i + 1 + PLY_T_LEN] == 'T' * PLY_T_LEN and i >= d1]
    ends = []
    if idx:
        wins = []
        ws, we = idx[0] - d2, idx[0] - d1
        for i in idx:
            if i - d2 > we:
                wins.append((ws, we))
                ws, we = i - d2, i - d1
            else:
                we = i - d1
        wins.append((ws, we))
        for s, e in wins:
            _, end = _o.sub_edit_distance(seq[max(s, 0):e], ADPT_SEQ, ADPT_MAX_ED, tie)
            if end > 0:
                ends.append(max(s, 0) + end + 1)
    return sorted(set(ends))


def _adaptor_ends_5p(seq, tie):
    """polyT_adaptor_finder.py:50-63 on the strand view `seq` (first 200 nt)."""
    d1, d2 = AFT_ADPT
    _, tso_end = _o.sub_edit_distance(seq, TSO_SEQ, ADPT_MAX_ED, tie)
    if tso_end > 0:
        tso_start = max(0, tso_end - len(TSO_SEQ))
        lo, hi = max(0, tso_start - d2), max(0, tso_start - d1)
        _, end = _o.sub_edit_distance(seq[lo:hi], ADPT_SEQ, ADPT_MAX_ED, tie)
        return [lo + end + 1] if end > 0 else []
    return []


def read_stage1(seq, qual, kit, umi_len, tie=0):
    """One read -> dict(pass_code, strand, raw_bc_start, raw_bc, raw_bc_min_q, putative_umi, polyT_end, pre, post),
    None where the reference has None (polyT_adaptor_finder.py:198-379)."""
    finder = _adaptor_ends_3p if kit in KITS_3P else _adaptor_ends_5p
    rc = _o.reverse_complement(seq)
    found = {}
    fwd = finder(seq[:ADPT_WIN], tie)                       # strand '-': the read as it is
    if fwd:
        found['-'] = fwd
    rev = finder(_o.reverse_complement(seq[-ADPT_WIN:]), tie)   # strand '+' (:148-153; seq[-200:] of a shorter read = all of it)
    if rev:
        found['+'] = rev
    code = 0
    if len(found) == 0:
        code = 1
    elif len(found) == 2:
        code = 2
    elif len(next(iter(found.values()))) > 1:
        code = 10
    out = dict(pass_code=code, strand=None, raw_bc_start=None, raw_bc=None, raw_bc_min_q=None, putative_umi=None,
               polyT_end=None, pre=None, post=None)
    if code:
        return out
    strand, (start,) = next(iter(found.items()))
    view, q = (rc, qual[::-1]) if strand == '+' else (seq, qual)
    out.update(strand=strand, raw_bc_start=start, raw_bc=view[start:start + BC_SIZE],
               raw_bc_min_q=min(ord(x) for x in q[start:start + BC_SIZE]) - 33,
               putative_umi=view[start + BC_SIZE:start + BC_SIZE + umi_len],
               pre=view[start - FLANK:start],                # Python slice semantics on purpose (negative start, :321-323)
               post=view[start + BC_SIZE + umi_len:start + BC_SIZE + umi_len + FLANK])
    # :345-379
    umi_end = start + BC_SIZE + umi_len
    p = view.find('TTTT', umi_end)
    if p == -1:
        out['polyT_end'] = umi_end            # :368-369 returns abs(umi_end_idx) WITHOUT the sign, also on the '+' strand
        return out
    else:
        code_t = [int(x == 'T') for x in view]
        for nt in code_t[p:]:
            if nt == 1:
                p += 1
            elif sum(code_t[p:p + 10]) >= 7:
                p += 1
            else:
                break
    out['polyT_end'] = -p if strand == '+' else p
    return out


def get_raw_bc_from_reads(reads, min_q=0, kit=None, umi_len=12, tie=0):
    """main.py:43-106: (Counter of pass codes, rows) for reads = [(id, seq, qual)]; rows = list of the 7 CSV fields."""
    rows, passes = [], []
    for rid, seq, qual in reads:
        r = read_stage1(seq, qual, kit, umi_len, tie)
        rows.append([rid, r['raw_bc'], r['raw_bc_min_q'], r['putative_umi'], r['polyT_end'], r['pre'], r['post']])
        mq = r['raw_bc_min_q']
        passes.append(100 if (mq and mq < min_q) else r['pass_code'])
    return Counter(passes), rows
