"""oracle.py -- TEST INFRASTRUCTURE ONLY (never imported by blaze_amd/).

CPU restatement of BLAZE step 3 ("assign reads to the barcodes").  Only tests/,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker / the timed CPU baseline.

Pinned part (goldens in tests/golden/ come from the real functions imported
from /root/reference by oracle/make_golden.py):
  * strand rule, blank / exact-match shortcut, 2-nt extension, the
    order-dependent whitelist fold, UMI start correction
    (/root/reference/blaze/read_assignment.py:35-99),
  * trim / restrand / record format / per-batch gzip member
    (/root/reference/blaze/read_assignment.py:184-249),
  * reverse_complement (/root/reference/blaze/helper.py:14-25).

PARITY UNPINNED part: ``sub_edit_distance`` itself lives in the third-party
wheel fast-edit-distance==1.2.2 (/root/reference/pyproject.toml:22), absent from
/root/reference and from this image; see oracle_ed.c's header for the three
policy points (end-index tie rule, equal-length roles, over-cut-off value).
"""
from __future__ import annotations

import ctypes
import gzip
import os
import subprocess
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "oracle_ed.c")
_SO = os.path.join(_HERE, "liboracle_ed.so")

ED_FLANKING = 2          # config.py:72-75 (DEFAULT_ED_FLANKING = DEFAULT_ASSIGNMENT_ED = 2)
TIE_LEFTMOST = 0
TIE_RIGHTMOST = 1

_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile oracle_ed.c with gcc (plain C, -O2)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, _SRC])
    return _SO


def lib():
    global _lib
    with _lock:
        if _lib is None:
            build()
            L = ctypes.CDLL(_SO)
            c_int, c_i64, c_p = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p
            L.orc_sub_edit_distance.argtypes = [ctypes.c_char_p, c_int, ctypes.c_char_p, c_int, c_int, c_int,
                                                ctypes.POINTER(c_int)]
            L.orc_sub_edit_distance.restype = c_int
            L.orc_edit_distance.argtypes = [ctypes.c_char_p, c_int, ctypes.c_char_p, c_int, c_int]
            L.orc_edit_distance.restype = c_int
            L.orc_match_rows.argtypes = [c_p, c_int, c_p, c_p, c_p, c_i64, c_i64, c_p, c_p, c_i64,
                                         c_int, c_int, c_int, c_p, c_p, c_p, c_p]
            L.orc_match_rows.restype = None
            L.orc_far_from_all.argtypes = [c_p, c_p, c_int, c_i64, c_p, c_p, c_int, c_i64, c_int, c_p]
            L.orc_far_from_all.restype = None
            _lib = L
    return _lib


def _b(s) -> bytes:
    return s if isinstance(s, bytes) else s.encode("latin-1")


def sub_edit_distance(a, b, max_ed, tie_rule: int = TIE_LEFTMOST):
    """(ed, end_idx) with fast_edit_distance.sub_edit_distance's call-site contract."""
    a, b = _b(a), _b(b)
    end = ctypes.c_int(-1)
    ed = lib().orc_sub_edit_distance(a, len(a), b, len(b), int(max_ed), int(tie_rule), ctypes.byref(end))
    return ed, end.value


def edit_distance(a, b, max_ed):
    a, b = _b(a), _b(b)
    return lib().orc_edit_distance(a, len(a), b, len(b), int(max_ed))


def sub_edit_distance_bruteforce(a: str, b: str):
    """Independent definition used to pin the DP: min over ALL substrings of the
    longer string of the plain Levenshtein distance to the shorter one, plus the
    set of inclusive end indices attaining it (-1 = empty substring)."""
    p, t = (a, b) if len(a) <= len(b) else (b, a)

    def lev(x, y):
        prev = list(range(len(y) + 1))
        for i in range(1, len(x) + 1):
            cur = [i] + [0] * len(y)
            for j in range(1, len(y) + 1):
                cur[j] = min(prev[j - 1] + (x[i - 1] != y[j - 1]), prev[j] + 1, cur[j - 1] + 1)
            prev = cur
        return prev[len(y)]

    best, ends = len(p), {-1}
    for e in range(len(t)):            # inclusive end; s = e+1 is the empty substring "ending" at e
        for s in range(e + 2):
            d = lev(p, t[s:e + 1])
            if d < best:
                best, ends = d, {e}
            elif d == best:
                ends.add(e)
    return best, ends


def reverse_complement(seq: str) -> str:
    """helper.py:14-25 -- complement ACGT/acgt, leave every other character, reverse."""
    table = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a"}
    return "".join(table.get(ch, ch) for ch in reversed(seq))


def strand_of(polyT_end) -> str:
    """read_assignment.py:51-56."""
    if not polyT_end:
        return ""
    return "-" if polyT_end > 0 else "+"


def match_bc_row(putative_bc: str, putative_umi: str, polyT_end, pre_bc_flanking: str,
                 post_umi_flanking: str, ordered_whitelist, max_ed: int,
                 tie_rule: int = TIE_LEFTMOST, wl_set=None):
    """Pure-Python restatement of _match_bc_row (read_assignment.py:35-99) on an
    explicit ORDERED whitelist.  Returns [BC_corrected, putative_umi, strand]."""
    strand = strand_of(polyT_end)
    if wl_set is None:
        wl_set = set(ordered_whitelist)
    if not putative_bc or putative_bc in wl_set:
        return [putative_bc, putative_umi, strand]
    ext = pre_bc_flanking[-ED_FLANKING:] + putative_bc + putative_umi[:ED_FLANKING]
    best, hit, hit_end = max_ed, "", -1
    for w in ordered_whitelist:
        ed, end = sub_edit_distance(w, ext, best, tie_rule)
        if ed < best:
            best, hit, hit_end = ed, w, end
        elif ed == best:
            if not hit:
                hit, hit_end = w, end
            else:
                hit = "ambiguous"
                best -= 1
                if best < 0:
                    return ["", putative_umi, strand]
    if hit in ("", "ambiguous"):
        return ["", putative_umi, strand]
    adj = hit_end - (len(ext) - 1 - ED_FLANKING)
    umi = putative_umi
    if adj > 0:
        umi = putative_umi[adj:] + post_umi_flanking[:adj]
    elif adj < 0:
        umi = putative_bc[adj:] + putative_umi[:adj]
    return [hit, umi, strand]


def format_batch(reads, rows, assigned, restrand: bool) -> str:
    """FASTQ text of one batch (read_assignment.py:218-242).
    reads: [(id, seq, qual)], rows: [(read_id, polyT_end)], assigned: [[bc, umi, strand]]."""
    out = []
    for (rid, seq, qual), (row_id, pe), (bc, umi, strand) in zip(reads, rows, assigned):
        if row_id != rid:
            raise SystemExit("Different order in putative bc file and input fastq!")
        if not bc:
            continue
        cut = int(pe)
        if pe < 0:
            seq, qual = seq[:cut], qual[:cut]
        else:
            seq, qual = seq[cut:], qual[cut:]
        if restrand and strand == "-":
            seq, qual = reverse_complement(seq), qual[::-1]
        out.append(f"@{bc}_{umi}#{row_id}_{strand}\tCB:Z:{bc}\tUB:Z:{umi}\n{seq}\n+\n{qual}\n")
    return "".join(out)


def assign_batch(reads, rows, ordered_whitelist, max_ed, gz, restrand, tie_rule=TIE_LEFTMOST):
    """Restatement of _assign_read_batches (read_assignment.py:184-249) on plain
    Python data.  rows: dicts with the putative_bc.csv columns, blanks as ''.
    Returns (assigned [[bc, umi, strand]], bytes, demul_count, n_reads)."""
    wl_set = set(ordered_whitelist)
    assigned = [match_bc_row(r["putative_bc"], r["putative_umi"], r["polyT_end"], r["pre_bc_flanking"],
                             r["post_umi_flanking"], ordered_whitelist, max_ed, tie_rule, wl_set)
                for r in rows]
    text = format_batch(reads, [(r["read_id"], r["polyT_end"]) for r in rows], assigned, restrand)
    data = text.encode("utf-8")
    if gz:
        data = gzip.compress(data)
    return assigned, data, sum(1 for a in assigned if a[0] != ""), len(reads)


# ---------------------------------------------------------------- packed batch
def pack_whitelist(whitelist):
    enc = [_b(w) for w in whitelist]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        off[1:] = np.cumsum([len(w) for w in enc])
    return np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy(), off


def match_rows(ext, ext_len, bc_off, bc_len, whitelist, max_ed, tie_rule=TIE_LEFTMOST,
               shortcut=True, threads: int = 1):
    """C batch driver (orc_match_rows) over packed rows; same array contract as
    the product's bz_assign_* (include/blaze_hip.h).  Returns (hit, ed, end, flag)."""
    ext = np.ascontiguousarray(ext, dtype=np.uint8)
    n, stride = ext.shape
    ext_len = np.ascontiguousarray(ext_len, dtype=np.uint8)
    bc_off = np.ascontiguousarray(bc_off, dtype=np.uint8)
    bc_len = np.ascontiguousarray(bc_len, dtype=np.uint8)
    if isinstance(whitelist, tuple):
        wl, off = whitelist
    else:
        wl, off = pack_whitelist(whitelist)
    hit = np.full(n, -1, dtype=np.int32)
    ed = np.zeros(n, dtype=np.int8)
    end = np.zeros(n, dtype=np.int8)
    flag = np.zeros(n, dtype=np.uint8)
    L = lib()

    def run(lo, hi):
        L.orc_match_rows(ext.ctypes.data, stride, ext_len.ctypes.data, bc_off.ctypes.data, bc_len.ctypes.data,
                         lo, hi, wl.ctypes.data, off.ctypes.data, len(off) - 1, int(max_ed), int(tie_rule),
                         int(bool(shortcut)), hit.ctypes.data, ed.ctypes.data, end.ctypes.data, flag.ctypes.data)

    threads = max(1, int(threads))
    if threads == 1 or n < 2 * threads:
        run(0, n)
    else:
        chunk = max(1, (n + threads * 8 - 1) // (threads * 8))
        spans = [(lo, min(n, lo + chunk)) for lo in range(0, n, chunk)]
        with ThreadPoolExecutor(threads) as ex:     # ctypes drops the GIL during the call
            list(ex.map(lambda s: run(*s), spans))
    return hit, ed, end, flag


def _records(strings):
    enc = [_b(x) for x in strings]
    stride = max([len(x) for x in enc] + [1])
    buf = np.zeros((len(enc), stride), dtype=np.uint8)
    for k, x in enumerate(enc):
        buf[k, :len(x)] = np.frombuffer(x, dtype=np.uint8)
    return buf, np.array([len(x) for x in enc], dtype=np.int32), stride


def far_from_all(candidates, cells, min_ed=5):
    """main.py:249-250: [min(edit_distance(k, x, min_ed) for x in cells) >= min_ed for k in candidates]."""
    candidates, cells = list(candidates), list(cells)
    far = np.zeros(len(candidates), dtype=np.uint8)
    if candidates:
        a, al, ast = _records(candidates)
        b, bl, bst = _records(cells)
        lib().orc_far_from_all(a.ctypes.data, al.ctypes.data, ast, len(candidates), b.ctypes.data, bl.ctypes.data, bst,
                               len(cells), int(min_ed), far.ctypes.data)
    return far.astype(bool)


def empty_drop_list(candidates, cells, min_ed=5, max_n=2000):
    """main.py:248-255, literally: keep far candidates in order, stop once MORE than max_n are kept."""
    cells = list(cells)
    ept_bc = []
    for k in candidates:
        if far_from_all([k], cells, min_ed)[0]:
            ept_bc.append(k)
        if len(ept_bc) > max_n:
            break
    return ept_bc
