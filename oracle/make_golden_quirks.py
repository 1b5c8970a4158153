#!/usr/bin/env python3
"""Golden vectors for the ODD inputs of step 3, recorded from the REAL reference functions.

TEST INFRASTRUCTURE ONLY; runs in the build container (needs /root/reference).  Like
oracle/make_golden.py it imports blaze.read_assignment from /root/reference with oracle/shim on
sys.path (stand-in for the absent fast-edit-distance wheel, routed to oracle_ed.c) and records
inputs -> outputs of

  _match_bc_row   /root/reference/blaze/read_assignment.py:35-99
  assign_read     /root/reference/blaze/read_assignment.py:253-330   (n_process=1: the serial loop :293-305)

on the inputs the regular goldens do not hold (round-2 verdict, item 1):

  match_rows_quirks_<tie>.json
    * whitelists with '' (a blank line of --known-bc-list) at the front / middle / end / twice and rows that tie
      with it at distance 0 (the `if not bc_hit` rule of :78 and the `bc_hit == ''` test of :87),
    * 15-mers, 18-mers, entries with N and lower case beside regular barcodes, lists without a regular entry,
    * entries of 31-127 characters (the row's text is then the shorter operand of :72),
    * max_ed 4-8,
    * putative barcodes of 1-28 nt and of 17-60 nt (extensions of 21-64 columns), short flanks and UMIs;
  e2e_quirks_<tie>.json
    * assign_read over files whose putative_bc.csv holds pandas NA tokens, quoted cells, CRLF line ends, float /
      blank / zero polyT_end, barcode cells wider than 28 nt, batches whose columns get different dtypes, several
      FASTQ files (one gzipped), plain and .gz output, restrand on and off, odd whitelist lines.

What the vectors pin: everything AROUND the edit-distance arithmetic.  sub_edit_distance itself stays PARITY
UNPINNED (third-party, absent); one file per end-index tie rule.  The order in which the reference iterated
set(whitelist) (PYTHONHASHSEED=0) is stored with every file case, because the fold depends on it.
Only data (inputs + expected outputs) is written; no reference source text.
Run:  python oracle/make_golden_quirks.py
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile
from io import StringIO

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

if os.environ.get("PYTHONHASHSEED") != "0" or "ORACLE_TIE_RULE" not in os.environ:
    for tie in ("0", "1"):
        env = dict(os.environ, PYTHONHASHSEED="0", ORACLE_TIE_RULE=tie, MPLBACKEND="Agg", PYTHONDONTWRITEBYTECODE="1")
        subprocess.check_call([sys.executable, os.path.abspath(__file__)], env=env)
    sys.exit(0)

TIE = int(os.environ["ORACLE_TIE_RULE"])
sys.path[:0] = [os.path.join(HERE, "shim"), REF, ROOT]

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import blaze.read_assignment as ra  # noqa: E402  (the real reference)

HEADER = "read_id,putative_bc,putative_bc_min_q,putative_umi,polyT_end,pre_bc_flanking,post_umi_flanking\n"
ACGT = "ACGT"


def run_rows(csv_body, ordered_wl, max_ed):
    df = pd.read_csv(StringIO(HEADER + csv_body)).fillna("")     # :160-164, :205
    return [ra._match_bc_row(row, list(ordered_wl), max_ed, 0) for row in df.itertuples()]


def rand_seq(rng, n):
    return "".join(ACGT[c] for c in rng.integers(0, 4, size=n))


def mutate(rng, s, n_edits):
    for _ in range(n_edits):
        kind = int(rng.integers(0, 3))
        if kind == 0 and s:
            p = int(rng.integers(0, len(s)))
            s = s[:p] + ACGT[(ACGT.find(s[p].upper()) + 1 + int(rng.integers(0, 3))) % 4] + s[p + 1:]
        elif kind == 1 and len(s) > 1:
            p = int(rng.integers(0, len(s)))
            s = s[:p] + s[p + 1:]
        else:
            p = int(rng.integers(0, len(s) + 1))
            s = s[:p] + ACGT[int(rng.integers(0, 4))] + s[p:]
    return s


def row_text(rid, read, a, bc_len, umi_len=12, pe=None, pre_len=5, post_len=5):
    """Cut a synthetic read as stage 1 would: barcode = read[a : a + bc_len], flanks and UMI around it."""
    pre = read[max(0, a - pre_len):a]
    bc = read[a:a + bc_len]
    umi = read[a + bc_len:a + bc_len + umi_len]
    post = read[a + bc_len + umi_len:a + bc_len + umi_len + post_len]
    pe = (a + bc_len + umi_len + 20) if pe is None else pe
    return f"{rid},{bc},20,{umi},{pe},{pre},{post}\n"


def rows_around(rng, wl, n, *, edits=(0, 1, 2, 3), bc_len=(16, 16), pre_len=(5, 5), umi_len=(12, 12), tag="r"):
    """n rows whose barcode region is an entry of `wl` (any length, any alphabet) with a few random edits, embedded
    in random sequence; the cut takes `bc_len` characters from where the entry starts."""
    out = []
    pool = [w for w in wl if w]
    for k in range(n):
        w = pool[int(rng.integers(0, len(pool)))] if rng.random() < 0.9 else rand_seq(rng, 16)
        body = mutate(rng, w, int(edits[int(rng.integers(0, len(edits)))]))
        up = rand_seq(rng, 8)
        read = up + body + rand_seq(rng, 90)
        a = len(up) + int(rng.integers(-1, 2)) * int(rng.random() < 0.15)    # a cut one base off now and then
        bl = int(rng.integers(bc_len[0], bc_len[1] + 1))
        pl = int(rng.integers(pre_len[0], pre_len[1] + 1))
        ul = int(rng.integers(umi_len[0], umi_len[1] + 1))
        pe = (1 if k % 2 else -1) * (a + bl + ul + 15)
        out.append(row_text(f"{tag}{k}", read, a, bl, ul, pe, pl))
    return "".join(out)


def match_row_cases():
    cases = []

    def add(name, body, wl, max_ed):
        cases.append(dict(name=name, csv=body, whitelist=list(wl), max_ed=max_ed, expect=run_rows(body, wl, max_ed)))

    rng = np.random.default_rng(2024)
    base = [rand_seq(rng, 16) for _ in range(40)]

    # ---- '' in the whitelist: front / middle / end / twice; rows that hold a barcode verbatim but shifted (distance 0
    # without the shortcut), rows one and two edits away, rows near nothing
    def blank_rows(tag):
        out = []
        for k in range(60):
            w = base[k % len(base)]
            up = rand_seq(rng, 8)
            kind = k % 6
            if kind == 0:      # verbatim, cut one base late: distance 0 through the extension, no shortcut
                read, a = up + w + rand_seq(rng, 60), len(up) + 1
            elif kind == 1:    # verbatim, cut one base early
                read, a = up + w + rand_seq(rng, 60), len(up) - 1
            elif kind == 2:    # exact: the shortcut (:61-62) wins before any ''
                read, a = up + w + rand_seq(rng, 60), len(up)
            elif kind == 3:
                read, a = up + mutate(rng, w, 1) + rand_seq(rng, 60), len(up)
            elif kind == 4:
                read, a = up + mutate(rng, w, 2) + rand_seq(rng, 60), len(up)
            else:
                read, a = rand_seq(rng, 100), 8
            out.append(row_text(f"{tag}{k}", read, a, 16, 12, (1 if k % 2 else -1) * (a + 40)))
        return "".join(out)

    for max_ed in (0, 1, 2, 3, 5):
        for where, wl in (("front", [""] + base), ("middle", base[:17] + [""] + base[17:]), ("end", base + [""]),
                          ("twice", base[:5] + [""] + base[5:30] + [""] + base[30:]),
                          ("front_and_twin", ["", base[3]] + base)):
            add(f"blank_entry_{where}_ed{max_ed}", blank_rows(f"b{where}{max_ed}_"), wl, max_ed)
    # two barcodes verbatim inside one extension, '' before / between / behind them: every order of the falsy rule
    twin_a = rand_seq(rng, 16)
    twin_b = twin_a[1:] + "G"                       # twin_b = twin_a shifted by one: both sit in ext = x + twin_a + G...
    read = "ACGTACGT" + twin_a + "G" + rand_seq(rng, 50)
    # the putative barcode must not equal an entry (no shortcut): cut one base early
    body = row_text("twins", read, 7, 16, 12, 60) + row_text("twins2", read, 9, 16, 12, -60)
    for name, wl in (("e_a_b", ["", twin_a, twin_b]), ("a_e_b", [twin_a, "", twin_b]), ("a_b_e", [twin_a, twin_b, ""]),
                     ("e_a", ["", twin_a]), ("a_e", [twin_a, ""]), ("e_e_a", ["", "", twin_a]), ("e_a_e", ["", twin_a, ""]),
                     ("e_a_e_b", ["", twin_a, "", twin_b]), ("only_e", [""]), ("e_e", ["", ""])):
        for max_ed in (0, 2):
            add(f"blank_entry_twins_{name}_ed{max_ed}", body, wl + base[:6], max_ed)
            add(f"blank_entry_twins_{name}_alone_ed{max_ed}", body, wl, max_ed)

    # ---- irregular entries beside regular ones
    odd = [base[0][:15], base[1] + "AC", base[2][:7] + "N" + base[2][8:], base[3].lower(), base[4][:8].lower() + base[4][8:],
           base[5][:12], "N" * 16, base[6][:3], "A", base[7] + base[8][:4]]
    mixed = list(base)
    for k, e in enumerate(odd):
        mixed.insert((k * 7) % len(mixed), e)
    for max_ed in (0, 1, 2, 3):
        add(f"mixed_list_ed{max_ed}", rows_around(rng, mixed, 160, tag=f"m{max_ed}_"), mixed, max_ed)
        add(f"mixed_list_odd_cuts_ed{max_ed}",
            rows_around(rng, mixed, 120, bc_len=(12, 20), pre_len=(0, 5), umi_len=(0, 12), tag=f"mo{max_ed}_"), mixed, max_ed)
    # rows whose barcode cell IS an odd entry (shortcut on an irregular entry) or has N / lower case itself
    body = "".join(row_text(f"s{k}", "ACGTACGT" + e + rand_seq(rng, 60), 8, len(e), 12, 50 + k) for k, e in enumerate(odd) if e)
    body += "".join(row_text(f"n{k}", "ACGTACGT" + (base[k][:k] + "N" + base[k][k + 1:]) + rand_seq(rng, 60), 8, 16, 12, -50)
                    for k in range(10))
    body += "".join(row_text(f"l{k}", "ACGTACGT" + (base[k][:k] + base[k][k:].lower()) + rand_seq(rng, 60), 8, 16, 12, 50)
                    for k in range(10))
    add("mixed_list_shortcut_on_odd_entries", body, mixed, 2)
    # lists WITHOUT a regular entry
    for name, wl in (("all_15mers", [w[:15] for w in base]), ("all_lower", [w.lower() for w in base]),
                     ("all_18mers", [w + "GT" for w in base]), ("all_with_N", [w[:5] + "N" + w[6:] for w in base])):
        add(f"no_regular_entry_{name}", rows_around(rng, wl, 80, tag=name[:5]), wl, 2)
        add(f"no_regular_entry_{name}_ed1", rows_around(rng, wl, 40, tag=name[:5]), wl, 1)

    # ---- entries longer than the row's text (31-127 characters): the text becomes the pattern of :72
    longs = []
    for L in (31, 32, 33, 38, 47, 64, 65, 80, 100, 126, 127):
        longs.append(rand_seq(rng, L))
    wl_long = list(base[:20])
    for k, e in enumerate(longs):
        wl_long.insert((k * 5) % len(wl_long), e)

    def long_rows(tag, n, bc_len=(16, 16)):
        out = []
        for k in range(n):
            e = longs[k % len(longs)] if k % 4 else base[k % 20]
            # a window of the long entry (anywhere, also its very end: end index > 30) with a few edits
            bl = int(rng.integers(bc_len[0], bc_len[1] + 1))
            span = bl + 4
            s = int(rng.integers(0, max(1, len(e) - span + 1)))
            if k % 3 == 0 and len(e) > span:
                s = len(e) - span
            window = mutate(rng, e[s:s + span], int(rng.integers(0, 3))) if len(e) >= span else mutate(rng, e, int(rng.integers(0, 3)))
            read = rand_seq(rng, 6) + window + rand_seq(rng, 70)
            out.append(row_text(f"{tag}{k}", read, 8, bl, 12, (1 if k % 2 else -1) * 70))
        return "".join(out)

    for max_ed in (0, 1, 2, 3, 5):
        add(f"long_entries_ed{max_ed}", long_rows(f"L{max_ed}_", 110), wl_long, max_ed)
    add("long_entries_wide_rows_ed2", long_rows("LW_", 90, bc_len=(17, 60)), wl_long, 2)
    add("long_entries_only_ed2", long_rows("LO_", 60), longs, 2)

    # ---- max_ed 4..8 (regular lists and a mixed one)
    wl200 = [rand_seq(rng, 16) for _ in range(200)]
    for max_ed in (4, 5, 6, 7, 8):
        add(f"max_ed_{max_ed}", rows_around(rng, wl200, 100, edits=(0, 1, 2, 3, 4, 5, 6, 7, 8), tag=f"e{max_ed}_"), wl200, max_ed)
        add(f"max_ed_{max_ed}_mixed", rows_around(rng, mixed, 60, edits=(1, 2, 3, 4, 5, 6), tag=f"em{max_ed}_"), mixed, max_ed)
        add(f"max_ed_{max_ed}_blank_entry", rows_around(rng, base, 30, edits=(0, 2, 4, 6), tag=f"eb{max_ed}_"),
            base[:9] + [""] + base[9:], max_ed)

    # ---- putative barcodes of 1..28 nt and 17..60 nt (extensions of up to 64 columns), flanks and UMIs of every length
    for max_ed in (0, 1, 2, 3, 6):
        add(f"short_and_long_barcode_cells_ed{max_ed}",
            rows_around(rng, base, 140, bc_len=(1, 28), pre_len=(0, 5), umi_len=(0, 12), tag=f"c{max_ed}_"), base, max_ed)
        add(f"wide_barcode_cells_ed{max_ed}",
            rows_around(rng, base, 120, bc_len=(17, 60), pre_len=(0, 5), umi_len=(0, 12), tag=f"w{max_ed}_"), base, max_ed)
    add("wide_barcode_cells_mixed_list", rows_around(rng, mixed, 120, bc_len=(17, 60), pre_len=(0, 5), tag="wm_"), mixed, 2)
    add("short_barcode_cells_mixed_list", rows_around(rng, mixed, 120, bc_len=(1, 15), pre_len=(0, 5), umi_len=(0, 12), tag="sm_"),
        mixed, 2)
    # the widest row the device takes: 60-nt cell + 2 + 2
    add("widest_rows", rows_around(rng, base, 40, bc_len=(60, 60), tag="x_"), base, 2)
    return cases


# --------------------------------------------------------------------------------------------- files
NA_TOKENS = ["NA", "null", "NaN", "N/A", "NULL", "nan", "None", "n/a", "<NA>", "#N/A", "-NaN", "#NA", "-nan", "1.#QNAN"]


def file_case(name, seed, *, n=420, n_cells=60, crlf=False, batchsize=64, max_ed=2, odd_wl=False, wide=False,
              gz_table=False):
    """Step-3 inputs with every quirk the native parser had to restate, run through the real assign_read."""
    rng = np.random.default_rng(seed)
    cells = [rand_seq(rng, 16) for _ in range(n_cells)]
    wl_lines = [c + ("-1" if k % 2 else "") for k, c in enumerate(cells)]
    if odd_wl:
        for k, e in enumerate(["", cells[0][:15], cells[1].lower(), cells[2], "  ", cells[3] + "ACGTACGTACGTACGTACGT",
                               cells[4][:6] + "N" + cells[4][7:], cells[5] + "-1-2", "\t" + cells[6] + " "]):
            wl_lines.insert(int(rng.integers(0, len(wl_lines) + 1)), e)
    wl_text = "\n".join(wl_lines) + ("\n" if seed % 2 else "")       # last line with and without a newline
    parsed = [line.split("-")[0].strip() for line in wl_text.splitlines()]

    ids, rows, recs = [], [], []
    for i in range(n):
        rid = "%08x-%04x-read%d" % (int(rng.integers(0, 1 << 32)), int(rng.integers(0, 1 << 16)), i)
        L = int(rng.integers(40, 400))
        seq = rand_seq(rng, L)
        if i % 3 == 0:      # reverse_complement leaves anything but ACGT/acgt alone (helper.py:14-25)
            b = list(seq)
            for p in rng.integers(0, L, size=max(1, L // 9)):
                b[int(p)] = "NnacgtRYKM-.*"[int(rng.integers(0, 13))]
            seq = "".join(b)
        qual = "".join(chr(int(c)) for c in rng.integers(33, 75, size=L))
        ids.append(rid)
        recs.append((rid, seq, qual))
    # batch b (rows b*batchsize ...) gets its own flavour so that pandas infers different dtypes per batch
    for i in range(n):
        b = i // batchsize
        rid = ids[i]
        L = len(recs[i][1])
        c = cells[int(rng.integers(0, n_cells))]
        body = mutate(rng, c, int(rng.choice([0, 0, 1, 1, 2, 3])))
        read = rand_seq(rng, 8) + body + rand_seq(rng, 60)
        a = 8
        pre, bc, umi, post = read[a - 5:a], read[a:a + 16], read[a + 16:a + 28], read[a + 28:a + 33]
        pe = int(rng.integers(1, L + 30)) * (1 if rng.random() < 0.5 else -1)       # also beyond the read: empty slices
        pe_s, q_s = str(pe), str(int(rng.integers(5, 40)))
        blank = rng.random() < 0.14
        flavour = b % 6
        if flavour == 1:                                   # a batch with blanks: polyT_end / min_q become float64
            pe_s, q_s = f"{pe}.0", f"{q_s}.0"
        elif flavour == 2:                                 # all-integer batch: no blank row at all
            blank = False
        elif flavour == 3:                                 # a batch of failed reads only: every column all-NaN
            blank = True
        elif flavour == 4:
            pe_s = f"{pe}.7" if i % 2 else f"{pe}.0"      # int() truncates toward zero (:228-232)
        k = int(rng.integers(0, 36))
        if blank:
            rows.append(f"{rid},,,,,,\n")
            continue
        if k == 0:
            pre = NA_TOKENS[i % len(NA_TOKENS)]            # a flank pandas reads as NaN -> ''
        elif k == 1:
            post = NA_TOKENS[i % len(NA_TOKENS)]
        elif k == 2:
            bc = NA_TOKENS[i % len(NA_TOKENS)]             # the barcode cell itself: the row turns blank
        elif k == 3:
            umi = NA_TOKENS[i % len(NA_TOKENS)]
        elif k == 4:
            bc = f'"{bc}"'                                 # quoted cells
        elif k == 5:
            umi, pre = f'"{umi}"', f'"{pre}"'
        elif k == 6:
            pre = ""
        elif k == 7:
            pre = pre[-1:]
        elif k == 8:
            umi, post = umi[:1], ""
        elif k == 9:
            umi, post = "", ""
        elif k == 10:
            pe_s = "0"                                     # falsy: strand '' and nothing trimmed
        elif k == 11:
            pe_s = "0.0"
        elif k == 12:
            bc = bc[:7] + "N" + bc[8:]
        elif k == 13:
            bc = bc[:9] + bc[9:].lower()
        elif k == 14:
            bc = bc[:int(rng.integers(1, 16))]             # truncated barcode at the end of a read
            umi, post = "", ""
        elif k == 15 and wide:
            m = int(rng.integers(29, 61))                  # a hand-made cell wider than the packed row
            bc = (bc + rand_seq(rng, m))[:m] if i % 2 else (rand_seq(rng, m) + bc)[-m:]
        elif k == 16 and wide:
            m = int(rng.integers(17, 29))
            bc = (bc + rand_seq(rng, m))[:m]
        elif k == 17:
            post = post[:2]
        elif k == 18:
            q_s = ""                                       # a blank number cell in a non-blank row
        elif k == 19 and odd_wl:
            e = [x for x in parsed if x and (len(x) != 16 or x != x.upper() or "N" in x)]
            bc = e[i % len(e)][:60]
            if i % 2 and len(bc) > 3:
                bc = bc[:2] + ACGT[int(rng.integers(0, 4))] + bc[3:]
        elif k == 20:
            pe_s = f"{float(pe):.1e}" if abs(pe) % 10 == 0 else pe_s   # exponent form of a whole number
        rows.append(f"{rid},{bc},{q_s},{umi},{pe_s},{pre},{post}\n")
    # an unassignable row may carry a blank polyT_end (an assigned one would raise in the reference, :227)
    far = rand_seq(rng, 16)
    rows[5] = f"{ids[5]},{far},20,{rand_seq(rng, 12)},,{rand_seq(rng, 5)},{rand_seq(rng, 5)}\n"
    # an id with a comma, quoted in the table
    ids[7] = "comma,in,id-7"
    recs[7] = (ids[7], recs[7][1], recs[7][2])
    rows[7] = '"comma,in,id-7",' + rows[7].split(",", 1)[1]

    nl = "\r\n" if crlf else "\n"
    csv = (HEADER + "".join(rows)).replace("\n", nl)
    cuts = [0, n // 3 + 5, 2 * n // 3 - 11, n]
    fq = []
    for f in range(3):
        body = "".join(f"@{rid} runid=abc ch={j}\n{seq}\n+{rid if j % 5 == 0 else ''}\n{q}\n"
                       for j, (rid, seq, q) in enumerate(recs[cuts[f]:cuts[f + 1]]))
        fq.append(body.replace("\n", nl))

    res = {}
    with tempfile.TemporaryDirectory() as td:
        fns = []
        for f, body in enumerate(fq):
            if f == 1:
                fn = os.path.join(td, "reads_1.fastq.gz")
                with gzip.open(fn, "wb") as h:
                    h.write(body.encode())
            else:
                fn = os.path.join(td, f"reads_{f}.fastq")
                with open(fn, "wb") as h:
                    h.write(body.encode())
            fns.append(fn)
        pc, pw = os.path.join(td, "putative_bc.csv"), os.path.join(td, "whitelist.csv")
        with open(pc, "wb") as h:
            h.write(csv.encode())
        with open(pw, "wb") as h:
            h.write(wl_text.replace("\n", nl).encode())
        for out_name, restrand in (("out.fastq", True), ("out.fastq.gz", False)):
            po = os.path.join(td, out_name)
            demul, total = ra.assign_read(fastq_fns=fns, fastq_out=po, putative_bc_csv=pc, whitelsit_csv=pw, max_ed=max_ed,
                                          n_process=1, batchsize=batchsize, restrand=restrand)
            raw = open(po, "rb").read()
            text = gzip.decompress(raw).decode() if out_name.endswith(".gz") else raw.decode()
            res[out_name] = dict(restrand=restrand, fastq=text, demul=int(demul), total=int(total))
    return dict(name=name, fastq=fq, gz_file=1, csv=csv, whitelist_text=wl_text.replace("\n", nl), max_ed=max_ed,
                batchsize=batchsize, iteration_order=list(set(parsed)), outputs=res)


def file_cases():
    out = [file_case("plain_lf", 11, batchsize=64),
           file_case("crlf_wide_cells", 12, crlf=True, batchsize=50, wide=True),
           file_case("odd_whitelist", 13, batchsize=37, odd_wl=True, wide=True),
           file_case("odd_whitelist_ed1_crlf", 14, batchsize=100, odd_wl=True, crlf=True, max_ed=1, n=300),
           file_case("max_ed_3_one_batch", 15, batchsize=1000, max_ed=3, n=260)]
    for c in out:
        assert 0 < c["outputs"]["out.fastq"]["demul"] < c["outputs"]["out.fastq"]["total"], c["name"]
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    tag = "leftmost" if TIE == 0 else "rightmost"
    meta = dict(generator="oracle/make_golden_quirks.py", reference="/root/reference @ 2024-12-23",
                functions="blaze.read_assignment._match_bc_row / assign_read (real, imported)",
                sub_edit_distance="oracle/oracle_ed.c via oracle/shim (third-party original absent: PARITY UNPINNED)",
                tie_rule=tag, pandas=pd.__version__, numpy=np.__version__, python=sys.version.split()[0],
                hash_seed=os.environ["PYTHONHASHSEED"])
    cases = match_row_cases()
    with open(os.path.join(OUT, f"match_rows_quirks_{tag}.json"), "w") as f:
        json.dump(dict(meta=meta, cases=cases), f, indent=0)
    files = file_cases() if TIE == 0 else file_cases()[2:4]
    with open(os.path.join(OUT, f"e2e_quirks_{tag}.json"), "w") as f:
        json.dump(dict(meta=meta, cases=files), f, indent=0)
    print("quirk vectors written for tie rule", tag, ":", len(cases), "row cases /", sum(len(c["expect"]) for c in cases),
          "rows,", len(files), "file cases")


if __name__ == "__main__":
    main()
