"""Synthetic 10x-shaped inputs for tests and bench.py.

The reference's own test inputs (test/data/*.fastq, test_putative_bc.csv) and all
four 10x whitelists are missing from /root/reference (.MISSING_LARGE_BLOBS), so
every measured workload is synthesised here from fixed seeds:

  full whitelist : 6,794,880 distinct uniform 16-mers, PCG64(1)
  cell list      : first N of a PCG64(2) permutation of `full`
  reads          : PCG64(3); true barcode from the cell list w.p. 0.90 else from
                   `full`; template = adaptor + BC16 + UMI + T*30 with i.i.d.
                   per-base nanopore-style errors; cut as BLAZE's stage 1 would
                   with a perfectly located adaptor (pre 5 / bc 16 / umi U / post 5,
                   /root/reference/blaze/polyT_adaptor_finder.py:263-342);
                   14 % of rows blank (/root/reference/test/expect_output/test_summary.txt:8).

Everything is vectorised numpy so 1e6-1e7 rows are generated in seconds.
"""
from __future__ import annotations

import numpy as np

FULL_WHITELIST_SIZE = 6_794_880
ADAPTOR = b"CTTCCGATCT"          # config.py:12
BC_LEN = 16
FLANK = 5                        # config.py:46
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def codes_to_bytes(codes: np.ndarray, length: int = BC_LEN) -> np.ndarray:
    """uint32/uint64 2-bit codes (base 0 in the top bits) -> (n, length) ASCII."""
    codes = codes.astype(np.uint64)
    shifts = (2 * (length - 1 - np.arange(length))).astype(np.uint64)
    return _ACGT[((codes[:, None] >> shifts[None, :]) & np.uint64(3)).astype(np.intp)]


def full_whitelist_codes(size: int = FULL_WHITELIST_SIZE, seed: int = 1) -> np.ndarray:
    """`size` distinct uniform-random 16-mers as uint32 codes, in draw order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    draw = rng.integers(0, 1 << 32, size=int(size * 1.01) + 1024, dtype=np.uint64).astype(np.uint32)
    _, first = np.unique(draw, return_index=True)
    first.sort()
    assert first.size >= size
    return draw[first[:size]]


def cell_whitelist_codes(n_cells: int, full: np.ndarray | None = None, seed: int = 2) -> np.ndarray:
    if full is None:
        full = full_whitelist_codes()
    rng = np.random.Generator(np.random.PCG64(seed))
    return full[rng.permutation(full.size)[:n_cells]]


def make_rows(n: int, cell_codes: np.ndarray, full_codes: np.ndarray | None = None, *, umi_len: int = 12,
              sub: float = 0.01, ins: float = 0.005, dele: float = 0.005, blank_frac: float = 0.14,
              p_cell: float = 0.90, seed: int = 3) -> dict:
    """Putative-barcode rows as fixed-width ASCII matrices (blank rows have len 0).

    Returns dict with uint8 matrices `pre`(n,5) `bc`(n,16) `umi`(n,U) `post`(n,5),
    per-field lengths, `polyT_end` (int64, 0 for blank rows), `min_q`, `blank` mask
    and `true_idx` (index into the cell list or -1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if full_codes is None:
        full_codes = cell_codes
    from_cell = rng.random(n) < p_cell
    pick_cell = rng.integers(0, cell_codes.size, size=n)
    pick_full = rng.integers(0, full_codes.size, size=n)
    codes = np.where(from_cell, cell_codes[pick_cell], full_codes[pick_full])
    true_idx = np.where(from_cell, pick_cell, -1)

    up = rng.integers(0, 4, size=(n, 6))                       # a little read upstream of the adaptor
    tmpl = np.concatenate([
        _ACGT[up],
        np.broadcast_to(np.frombuffer(ADAPTOR, dtype=np.uint8), (n, len(ADAPTOR))),
        codes_to_bytes(codes),
        _ACGT[rng.integers(0, 4, size=(n, umi_len))],
        np.full((n, 30), ord("T"), dtype=np.uint8),
    ], axis=1)
    T = tmpl.shape[1]
    bc_start_t = 6 + len(ADAPTOR)

    is_del = rng.random((n, T)) < dele
    is_sub = (rng.random((n, T)) < sub) & ~is_del
    is_ins = rng.random((n, T)) < ins
    shift = rng.integers(1, 4, size=(n, T))
    code_of = np.zeros(256, dtype=np.int64)
    code_of[_ACGT] = np.arange(4)
    mutated = np.where(is_sub, _ACGT[(code_of[tmpl] + shift) & 3], tmpl)
    ins_chars = _ACGT[rng.integers(0, 4, size=(n, T))]

    contrib = (~is_del).astype(np.int64) + is_ins.astype(np.int64)
    start = np.cumsum(contrib, axis=1) - contrib               # output offset of each template base
    W = T + 24
    out = np.full((n, W), ord("T"), dtype=np.uint8)
    rows = np.broadcast_to(np.arange(n)[:, None], (n, T))
    keep = ~is_del
    out[rows[keep], start[keep]] = mutated[keep]
    ins_pos = start + keep.astype(np.int64)
    ok = is_ins & (ins_pos < W)
    out[rows[ok], ins_pos[ok]] = ins_chars[ok]

    a_end = start[:, bc_start_t]                               # perfectly located adaptor end
    a_end = np.maximum(a_end, FLANK)
    ar = np.arange(n)[:, None]
    pre = out[ar, a_end[:, None] - FLANK + np.arange(FLANK)[None, :]]
    bc = out[ar, a_end[:, None] + np.arange(BC_LEN)[None, :]]
    umi = out[ar, a_end[:, None] + BC_LEN + np.arange(umi_len)[None, :]]
    post = out[ar, a_end[:, None] + BC_LEN + umi_len + np.arange(FLANK)[None, :]]

    blank = rng.random(n) < blank_frac
    minus = rng.random(n) < 0.5
    pe_mag = a_end + BC_LEN + umi_len + rng.integers(8, 40, size=n)
    polyT_end = np.where(minus, pe_mag, -pe_mag).astype(np.int64)
    polyT_end[blank] = 0
    min_q = rng.integers(5, 41, size=n)

    def lens(width):
        v = np.full(n, width, dtype=np.uint8)
        v[blank] = 0
        return v

    return dict(n=n, umi_len=umi_len, pre=pre, bc=bc, umi=umi, post=post,
                pre_len=lens(FLANK), bc_len=lens(BC_LEN), umi_len_arr=lens(umi_len), post_len=lens(FLANK),
                polyT_end=polyT_end, min_q=min_q, blank=blank, true_idx=true_idx,
                params=dict(sub=sub, ins=ins, dele=dele, blank_frac=blank_frac, p_cell=p_cell, seed=seed,
                            umi_len=umi_len))


def pack_ext(rows: dict, stride: int = 32):
    """ext = pre[-2:] + bc + umi[:2] (read_assignment.py:70) as the C-ABI's packed
    arrays: (ext (n,stride) uint8, ext_len, bc_off, bc_len)."""
    n = rows["n"]
    ext = np.zeros((n, stride), dtype=np.uint8)
    ext[:, 0:2] = rows["pre"][:, -2:]
    ext[:, 2:2 + BC_LEN] = rows["bc"]
    ext[:, 2 + BC_LEN:4 + BC_LEN] = rows["umi"][:, :2]
    blank = rows["blank"]
    ext[blank] = 0
    ext_len = np.where(blank, 0, BC_LEN + 4).astype(np.uint8)
    bc_off = np.where(blank, 0, 2).astype(np.uint8)
    bc_len = np.where(blank, 0, BC_LEN).astype(np.uint8)
    return ext, ext_len, bc_off, bc_len


def read_ids(n: int, seed: int = 4):
    rng = np.random.Generator(np.random.PCG64(seed))
    h = rng.integers(0, 1 << 62, size=(n, 2))
    return ["%08x-%04x-%04x-%04x-%012x" % (a & 0xFFFFFFFF, (a >> 32) & 0xFFFF, (a >> 48) & 0x0FFF | 0x4000,
                                           b & 0x3FFF | 0x8000, (b >> 14) & 0xFFFFFFFFFFFF)
            for a, b in h.tolist()]


def csv_text(rows: dict, ids, lo: int = 0, hi: int | None = None) -> str:
    """putative_bc.csv text with the reference's header and column order
    (/root/reference/blaze/main.py:96-105); blank rows have empty cells."""
    hi = rows["n"] if hi is None else hi
    lines = ["read_id,putative_bc,putative_bc_min_q,putative_umi,polyT_end,pre_bc_flanking,post_umi_flanking\n"]
    s = lambda m, i: m[i].tobytes().decode("ascii")
    for i in range(lo, hi):
        if rows["blank"][i]:
            lines.append(f"{ids[i]},,,,,,\n")
        else:
            lines.append(f"{ids[i]},{s(rows['bc'], i)},{int(rows['min_q'][i])},{s(rows['umi'], i)},"
                         f"{int(rows['polyT_end'][i])},{s(rows['pre'], i)},{s(rows['post'], i)}\n")
    return "".join(lines)


def fastq_records(n: int, ids, seed: int = 5, median_len: int = 900, sigma: float = 0.6, min_len: int = 150):
    """[(id, seq, qual)] with uniform ACGT bodies, log-normal lengths, constant Q."""
    rng = np.random.Generator(np.random.PCG64(seed))
    lens = np.maximum(min_len, rng.lognormal(np.log(median_len), sigma, size=n).astype(np.int64))
    recs = []
    for i in range(n):
        seq = _ACGT[rng.integers(0, 4, size=int(lens[i]))].tobytes().decode("ascii")
        recs.append((ids[i], seq, "I" * int(lens[i])))
    return recs


def fastq_text(recs) -> str:
    return "".join(f"@{rid} runid=synthetic\n{seq}\n+\n{q}\n" for rid, seq, q in recs)


def write_file_job(d, n, n_cells, umi_len=10, n_files=4, seed=3, median_len=900):
    """Step-3 inputs on disk in BASELINE configs[4]'s shape: `n_files` FASTQ files of log-normal read
    lengths, their putative_bc.csv and a `-1`-suffixed whitelist.  Returns (fastq_fns, csv, whitelist, fastq_bytes)."""
    import os
    full = full_whitelist_codes(400_000, seed=1)
    cells = cell_whitelist_codes(n_cells, full, seed=2)
    rows = make_rows(n, cells, full, umi_len=umi_len, seed=seed)
    ids = read_ids(n, seed=4)
    rng = np.random.Generator(np.random.PCG64(5))
    lens = np.maximum(150, rng.lognormal(np.log(median_len), 0.6, size=n).astype(np.int64))
    pool = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(lens.max()) + (1 << 20))].tobytes()
    starts = rng.integers(0, 1 << 20, size=n)
    with open(os.path.join(d, "putative_bc.csv"), "w") as f:
        f.write(csv_text(rows, ids))
    fns, per = [], -(-n // n_files)
    total_bytes = 0
    for k in range(n_files):
        fn = os.path.join(d, f"reads_{k}.fastq")
        with open(fn, "wb") as f:
            for i in range(k * per, min(n, (k + 1) * per)):
                L = int(lens[i])
                s = int(starts[i])
                rec = b"@" + ids[i].encode() + b" runid=synthetic\n" + pool[s:s + L] + b"\n+\n" + b"I" * L + b"\n"
                f.write(rec)
                total_bytes += len(rec)
        fns.append(fn)
    wl = os.path.join(d, "whitelist.csv")
    with open(wl, "w") as f:
        for w in codes_to_bytes(cells):
            f.write(w.tobytes().decode() + "-1\n")
    return fns, os.path.join(d, "putative_bc.csv"), wl, total_bytes


# ---------------------------------------------------------------------------------------------- whole reads (BASELINE configs[0])
ADAPTOR_FULL = b"CTACACGACGCTCTTCCGATCT"     # 10x read-1 adaptor, ends with config.py:12's CTTCCGATCT


def _with_errors(rng, arr, sub, ins, dele):
    """i.i.d. per-base substitutions / insertions / deletions on a uint8 base array (vectorised)."""
    n = arr.size
    if n == 0 or (sub == 0 and ins == 0 and dele == 0):
        return arr
    r = rng.random((3, n))
    arr = np.where(r[0] < sub, _ACGT[rng.integers(0, 4, size=n)], arr)
    keep = r[1] >= dele
    extra = keep & (r[2] < ins)
    reps = keep.astype(np.int64) + extra
    out = np.repeat(arr, reps)
    if extra.any():
        ends = np.cumsum(reps)[extra] - 1              # the second copy of a base that got an insertion behind it
        out[ends] = _ACGT[rng.integers(0, 4, size=ends.size)]
    return out


_COMP = np.arange(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTacgt", b"TGCAtgca"):
    _COMP[_a] = _b


def write_pipeline_job(d, n_files=29, reads_per_file=4000, n_cells=500, n_full=20000, seed=116):
    """FASTQ files in the shape of the reference's bundled test run (BASELINE configs[0]: 29 files x 4,000 reads, 3v3 kit;
    the real files are missing from /root/reference, .MISSING_LARGE_BLOBS:5-33) + a stand-in full whitelist: 80 % of the
    reads carry one of `n_cells` cell barcodes, 12 % another barcode of the full list, 8 % are junk; nanopore-style errors
    at three levels; every second read reverse-complemented; noisy qualities.  Deterministic in its arguments.
    Returns (fastq directory, full whitelist path, sha256 of everything written)."""
    import hashlib
    import os
    rng = np.random.Generator(np.random.PCG64(seed))
    full = codes_to_bytes(full_whitelist_codes(n_full, seed=seed + 1))
    cells = full[rng.permutation(n_full)[:n_cells]]
    weight = rng.gamma(2.0, 1.0, size=n_cells)
    weight /= weight.sum()                               # cells of different sizes, as a knee plot wants them
    fq_dir = os.path.join(d, "fastq")
    os.makedirs(fq_dir, exist_ok=True)
    digest = hashlib.sha256()
    errs = [(0.0, 0.0, 0.0), (0.01, 0.005, 0.005), (0.03, 0.01, 0.02)]
    for f in range(n_files):
        parts = []
        for k in range(reads_per_file):
            u = rng.random()
            bc = cells[rng.choice(n_cells, p=weight)] if u < 0.80 else full[int(rng.integers(0, n_full))]
            if u >= 0.92:
                body = _ACGT[rng.integers(0, 4, size=int(rng.integers(20, 250)))]
            else:
                body = np.concatenate([
                    _ACGT[rng.integers(0, 4, size=int(rng.integers(0, 30)))], np.frombuffer(ADAPTOR_FULL, dtype=np.uint8), bc,
                    _ACGT[rng.integers(0, 4, size=12)], np.full(int(rng.integers(8, 35)), ord("T"), dtype=np.uint8),
                    _ACGT[rng.integers(0, 4, size=int(rng.integers(40, 160)))]])
            seq = _with_errors(rng, body, *errs[k % 3])
            if seq.size == 0:
                seq = np.frombuffer(b"A", dtype=np.uint8)
            if k % 2:
                seq = _COMP[seq[::-1]]
            qual = (33 + rng.integers(3 if k % 5 == 0 else 16, 45, size=seq.size)).astype(np.uint8)
            title = f"@f{f}_r{k}" + (f" runid=x{k} ch=7" if k % 4 == 0 else "")
            parts.append(title.encode() + b"\n" + seq.tobytes() + b"\n+\n" + qual.tobytes() + b"\n")
        text = b"".join(parts)
        digest.update(text)
        with open(os.path.join(fq_dir, f"reads_{f:02d}.fastq"), "wb") as h:
            h.write(text)
    full_fn = os.path.join(d, "full_whitelist.txt")
    with open(full_fn, "wb") as h:
        text = b"".join(w.tobytes() + b"\n" for w in full)
        digest.update(text)
        h.write(text)
    return fq_dir, full_fn, digest.hexdigest()
