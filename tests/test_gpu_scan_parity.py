"""GPU parity: the HIP scan (through the C ABI) against the CPU oracle, bit-exact.

Integer / index work: the bar is equality of hit index, final best_ed, end index
and the blank / exact / ambiguous flags on every row.
"""
import numpy as np
import pytest

from blaze_amd import _lib, synth

pytestmark = pytest.mark.gpu

FLAG_MASK = _lib.BZ_F_EXACT | _lib.BZ_F_BLANK | _lib.BZ_F_AMBIG
FUSED = 4          # scan mode 0 with BZ_OPT_PAIR_FUSED: the one-kernel form of the pair-seeded scan
MODES = dict(argvalues=[0, 1, 2, 3, FUSED], ids=["seeded", "bruteforce", "filtered", "seeded3", "fused"])


def _compare(engine, oracle, ext, el, bo, bl, wl, max_ed, tie=0, shortcut=True, force_generic=False, threads=8,
             scan_mode=0):
    engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0 if scan_mode == FUSED else scan_mode)
    engine.set_option(_lib.BZ_OPT_PAIR_FUSED, int(scan_mode == FUSED))
    engine.set_option(_lib.BZ_OPT_TIE_RULE, tie)
    engine.set_option(_lib.BZ_OPT_SHORTCUT, int(shortcut))
    engine.set_option(_lib.BZ_OPT_FORCE_GENERIC, int(force_generic))
    engine.set_whitelist(wl)
    hit, ed, end, flags = engine.assign(ext, el, bo, bl, max_ed)
    o_hit, o_ed, o_end, o_flag = oracle.match_rows(ext, el, bo, bl, wl, max_ed, tie, shortcut, threads=threads)
    engine.set_option(_lib.BZ_OPT_TIE_RULE, 0)
    engine.set_option(_lib.BZ_OPT_SHORTCUT, 1)
    engine.set_option(_lib.BZ_OPT_FORCE_GENERIC, 0)
    engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
    engine.set_option(_lib.BZ_OPT_PAIR_FUSED, 0)
    bad = np.flatnonzero((hit != o_hit) | (ed != o_ed) | (end != o_end) | ((flags & FLAG_MASK) != o_flag))
    assert bad.size == 0, (
        f"{bad.size} rows differ, first {bad[0]}: ext={bytes(ext[bad[0], :el[bad[0]]])!r} "
        f"gpu=({hit[bad[0]]},{ed[bad[0]]},{end[bad[0]]},{flags[bad[0]]}) "
        f"oracle=({o_hit[bad[0]]},{o_ed[bad[0]]},{o_end[bad[0]]},{o_flag[bad[0]]})")
    return hit, ed, end, flags


def _synthetic(n_rows, n_cells, err=1.0, seed=3, umi_len=12):
    full = synth.full_whitelist_codes(50_000, seed=21)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=22)
    rows = synth.make_rows(n_rows, cells, full, sub=0.01 * err, ins=0.005 * err, dele=0.005 * err, seed=seed,
                           umi_len=umi_len)
    return rows, synth.codes_to_bytes(cells)


@pytest.mark.parametrize("scan_mode", **MODES)
@pytest.mark.parametrize("max_ed", [0, 1, 2, 3])
@pytest.mark.parametrize("tie", [0, 1])
def test_bitparallel_vs_oracle(engine, oracle, max_ed, tie, scan_mode):
    rows, wl = _synthetic(6000, 600, err=3.0, seed=100 + max_ed)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl_list = [w.tobytes() for w in wl]
    hit, ed, end, flags = _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, tie, scan_mode=scan_mode)
    t = engine.timing(0)
    assert t["rows_generic"] == 0 and t["rows_bitparallel"] > 0
    assert (hit >= 0).sum() > 1000


@pytest.mark.parametrize("scan_mode", **MODES)
def test_all_rows_scored_mode(engine, oracle, scan_mode):
    rows, wl = _synthetic(4000, 333, err=2.0, seed=7)        # odd whitelist size: exercises the padding
    ext, el, bo, bl = synth.pack_ext(rows)
    _compare(engine, oracle, ext, el, bo, bl, [w.tobytes() for w in wl], 2, shortcut=False, scan_mode=scan_mode)
    t = engine.timing(0)
    assert t["rows_bitparallel"] == int((bl > 0).sum())


@pytest.mark.parametrize("max_ed", [2, 4, 6])
def test_generic_kernel_vs_oracle(engine, oracle, max_ed):
    rows, wl = _synthetic(1500, 150, err=3.0, seed=11)
    ext, el, bo, bl = synth.pack_ext(rows)
    _compare(engine, oracle, ext, el, bo, bl, [w.tobytes() for w in wl], max_ed, force_generic=True)
    t = engine.timing(0)
    assert t["rows_bitparallel"] == 0 and t["rows_generic"] == int((bl > 0).sum())


def test_generic_equals_bitparallel(engine):
    rows, wl = _synthetic(3000, 257, err=4.0, seed=12)
    ext, el, bo, bl = synth.pack_ext(rows)
    engine.set_whitelist([w.tobytes() for w in wl])
    a = engine.assign(ext, el, bo, bl, 2)
    engine.set_option(_lib.BZ_OPT_FORCE_GENERIC, 1)
    b = engine.assign(ext, el, bo, bl, 2)
    engine.set_option(_lib.BZ_OPT_FORCE_GENERIC, 0)
    for x, y in zip(a[:3], b[:3]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[3] & FLAG_MASK, b[3] & FLAG_MASK)


@pytest.mark.parametrize("stride", [20, 21, 24, 32, 40, 64])
def test_row_strides(engine, oracle, stride):
    """Row layouts: multiples of 8 from 24 up take the packed kernels (aligned 8-byte loads), everything else the
    byte-wise loaders; same results, bytes behind a row's text ignored."""
    rows, wl = _synthetic(5000, 700, err=3.0, seed=23)
    ext, el, bo, bl = synth.pack_ext(rows, stride=stride)
    if stride > 20:
        ext[:, 20:] = np.frombuffer(b"ACGTN\0", dtype=np.uint8)[np.random.default_rng(3).integers(0, 6, size=(ext.shape[0], stride - 20))]
    _compare(engine, oracle, ext, el, bo, bl, [w.tobytes() for w in wl], 2)
    t = engine.timing(0)
    assert t["rows_generic"] == 0 and t["rows_bitparallel"] > 1000


def test_irregular_rows(engine, oracle):
    """Short flanks / short UMI (ext 16..19), N and lower case, short and long putative_bc."""
    rows, wl = _synthetic(3000, 200, err=2.0, seed=13)
    ext, el, bo, bl = synth.pack_ext(rows)
    rng = np.random.default_rng(5)
    n = ext.shape[0]
    for r in range(n):
        if bl[r] == 0:
            continue
        k = r % 8
        if k == 1:                                  # pre flank of 1: drop the first byte
            ext[r, :19] = ext[r, 1:20]; el[r] = 19; bo[r] = 1
        elif k == 2:                                # no pre flank, no umi
            ext[r, :16] = ext[r, 2:18]; el[r] = 16; bo[r] = 0
        elif k == 3:                                # one umi base only
            el[r] = 19
        elif k == 4:                                # N somewhere
            ext[r, rng.integers(0, 20)] = ord("N")
        elif k == 5:                                # lower case base
            j = rng.integers(0, 20); ext[r, j] = ext[r, j] | 0x20
        elif k == 6:                                # 14-nt putative bc
            ext[r, 16:18] = ext[r, 18:20]; el[r] = 18; bl[r] = 14
    ext[np.arange(ext.shape[1])[None, :] >= el[:, None]] = 0
    hit, ed, end, flags = _compare(engine, oracle, ext, el, bo, bl, [w.tobytes() for w in wl], 2)
    t = engine.timing(0)
    assert t["rows_generic"] > 0 and t["rows_bitparallel"] > 0


@pytest.mark.parametrize("mixed", [False, True], ids=["regular_list", "mixed_list"])
@pytest.mark.parametrize("tie", [0, 1])
@pytest.mark.parametrize("max_ed", [0, 1, 2, 3])
def test_irregular_rows_every_kind(engine, oracle, max_ed, tie, mixed):
    """The workgroup-per-row scan of irregular rows: N / lower case / other bytes anywhere, putative barcodes of 1..28 nt,
    extensions of 1..32 columns (shorter than the barcodes: the text becomes the pattern; shorter than max_ed: within
    reach of everything), barcode at odd offsets, exact copies of whitelist entries inside irregular rows (the shortcut),
    against an all-regular and a mixed whitelist."""
    rows, wl = _synthetic(1600, 250, err=3.0, seed=91)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl_list = [w.tobytes() for w in wl]
    if mixed:
        wl_list.insert(7, wl_list[3][:15])
        wl_list.insert(100, wl_list[50] + b"AC")
        wl_list.append(wl_list[9][:6] + b"N" + wl_list[9][7:])
    rng = np.random.default_rng(6)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    n = ext.shape[0]
    for r in range(n):
        if bl[r] == 0:
            continue
        k = r % 12
        if k == 0:
            ext[r, rng.integers(0, 20)] = ord("N")
        elif k == 1:
            ext[r, rng.integers(0, 20)] |= 0x20
        elif k == 2:                                # long extension: 5-nt flanks kept by some other producer
            L = int(rng.integers(21, 33)); ext[r, 20:L] = acgt[rng.integers(0, 4, L - 20)]; el[r] = L
        elif k == 3:                                # barcode of 10 nt at the start of a 10-column text: the text is the pattern
            el[r] = 10; bo[r] = 0; bl[r] = 10
        elif k == 4:                                # two columns: shorter than max_ed 2 and 3
            el[r] = 2; bo[r] = 0; bl[r] = 2
        elif k == 5:                                # one column
            el[r] = 1; bo[r] = 0; bl[r] = 1
        elif k == 6:                                # 28-nt putative barcode
            ext[r, 2:30] = np.tile(ext[r, 2:18], 2)[:28]; el[r] = 32; bl[r] = 28
        elif k == 7:                                # 15 columns, an N inside
            el[r] = 15; bo[r] = 0; bl[r] = 15; ext[r, 4] = ord("N")
        elif k == 8:                                # an exact copy of an entry behind an N flank: the shortcut inside an irregular row
            w = np.frombuffer(wl_list[int(rng.integers(0, len(wl_list)))], dtype=np.uint8)
            ext[r, :] = 0; ext[r, 0] = ord("N"); ext[r, 1] = ord("A"); ext[r, 2:2 + w.size] = w
            ext[r, 2 + w.size:4 + w.size] = acgt[rng.integers(0, 4, 2)]; el[r] = 4 + w.size; bo[r] = 2; bl[r] = w.size
        elif k == 9:                                # 16-column text, barcode 14
            el[r] = 16; bo[r] = 1; bl[r] = 14
    ext[np.arange(ext.shape[1])[None, :] >= el[:, None]] = 0
    hit, ed, end, flags = _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, tie)
    t = engine.timing(0)
    assert t["rows_generic"] > 700 and t["rows_bitparallel"] > 50
    assert ((flags & _lib.BZ_F_EXACT) != 0)[np.arange(n) % 12 == 8].sum() > 50


@pytest.mark.parametrize("max_ed", [4, 5, 6, 7, 8, 9])
def test_max_ed_above_the_seeded_kernels(engine, oracle, max_ed):
    """--max-edit-distance above 3: no seed plan is exact any more; regular rows take the packed automaton over every pair
    up to 8 (irregular rows the workgroup-per-row scan), 9 and above the one-lane byte-wise scan.  Near matches are
    everywhere at these cut-offs, so the fold's tie cascade is exercised."""
    rows, wl = _synthetic(700, 120, err=4.0, seed=95)
    ext, el, bo, bl = synth.pack_ext(rows)
    ext[5, 3] = ord("N")
    for tie in (0, 1):
        hit, ed, end, flags = _compare(engine, oracle, ext, el, bo, bl, [w.tobytes() for w in wl], max_ed, tie)
    t = engine.timing(0)
    if max_ed <= 8:
        assert t["rows_bitparallel"] > 400 and 1 <= t["rows_generic"] < 20
    else:
        assert t["rows_bitparallel"] == 0 and t["rows_generic"] > 400
    assert ((hit >= 0) & ((flags & _lib.BZ_F_EXACT) == 0)).sum() > 50
    # a mixed whitelist at these cut-offs: the irregular entries are within reach of most rows, every such row is re-scored
    wl_list = [w.tobytes() for w in wl]
    wl_list.insert(7, wl_list[3][:15])
    wl_list.append(wl_list[9][:6] + b"N" + wl_list[9][7:])
    _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, 0)


@pytest.mark.parametrize("scan_mode", **MODES)
def test_bytes_past_the_text_are_ignored(engine, oracle, scan_mode):
    """The C ABI gives a row ext_len bytes; whatever the caller's buffer holds behind them (here: random ACGT, N and
    NUL) must not change a result.  Texts of 16..20 columns, barcode at offset 0..2."""
    rows, wl = _synthetic(4000, 300, err=2.0, seed=17)
    ext, el, bo, bl = synth.pack_ext(rows)
    for r in range(ext.shape[0]):
        if bl[r] == 0:
            continue
        k = r % 5
        if k == 1:
            ext[r, :19] = ext[r, 1:20]; el[r] = 19; bo[r] = 1
        elif k == 2:
            ext[r, :16] = ext[r, 2:18]; el[r] = 16; bo[r] = 0
        elif k == 3:
            el[r] = 18
        elif k == 4:
            ext[r, :19] = ext[r, 1:20]; el[r] = 17; bo[r] = 1
    clean = ext.copy()
    clean[np.arange(ext.shape[1])[None, :] >= el[:, None]] = 0
    wl_list = [w.tobytes() for w in wl]
    want = _compare(engine, oracle, clean, el, bo, bl, wl_list, 2, scan_mode=scan_mode)
    rng = np.random.default_rng(9)
    dirty = clean.copy()
    junk = np.frombuffer(b"ACGTN\0acgt", dtype=np.uint8)[rng.integers(0, 10, size=ext.shape)]
    past = np.arange(ext.shape[1])[None, :] >= el[:, None]
    dirty[past] = junk[past]
    engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0 if scan_mode == FUSED else scan_mode)
    engine.set_option(_lib.BZ_OPT_PAIR_FUSED, int(scan_mode == FUSED))
    try:
        got = engine.assign(dirty, el, bo, bl, 2)
    finally:
        engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
        engine.set_option(_lib.BZ_OPT_PAIR_FUSED, 0)
    for a, b, what in zip(want, got, ("hit", "ed", "end", "flags")):
        if what == "flags":
            a, b = a & FLAG_MASK, b & FLAG_MASK
        bad = np.flatnonzero(a != b)
        assert bad.size == 0, f"{what}: {bad.size} rows change with bytes behind the text, first row {bad[0]} (len {el[bad[0]]})"


def test_irregular_whitelist(engine, oracle):
    """Whitelist with an N barcode, a 15-mer, an 18-mer and an empty string.  The empty string is within any max_ed of
    every text, so the mixed path re-scores every scored row over the whole list (workgroup per row, falsy-hit rule)."""
    rows, wl = _synthetic(800, 60, err=2.0, seed=14)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl_list = [w.tobytes() for w in wl]
    wl_list[3] = wl_list[3][:7] + b"N" + wl_list[3][8:]
    wl_list[10] = wl_list[10][:15]
    wl_list[20] = wl_list[20] + b"AC"
    wl_list.insert(30, b"")
    _compare(engine, oracle, ext, el, bo, bl, wl_list, 2)
    _compare(engine, oracle, ext, el, bo, bl, wl_list, 1, tie=1)


@pytest.mark.parametrize("where", ["front", "middle", "end", "twice"])
@pytest.mark.parametrize("max_ed", [0, 1, 2, 3, 5])
def test_blank_line_in_the_whitelist(engine, oracle, max_ed, where):
    """A blank line of --known-bc-list becomes the entry '' (read_assignment.py:286-289): at distance 0 from every text, and
    as `bc_hit` falsy -- the next tie replaces it (:76) and :87 reports no hit.  The list keeps its fast-path structures
    (the empty entry is one of the irregular ones), every scored row is re-scored by the workgroup-per-row kernel with the
    falsy rule, and rows that hold a barcode verbatim one column off (distance 0 without the shortcut) tie with it on
    either side.  Equal to the oracle on every row, both tie rules, irregular rows included."""
    rows, wl = _synthetic(3000, 400, err=3.0, seed=57)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl_list = [w.tobytes() for w in wl]
    rng = np.random.default_rng(8)
    scored = np.flatnonzero((bl == 16) & (el == 20))
    for r in scored[::7]:                               # a barcode verbatim at columns 3..18, the row says 2..17
        w = np.frombuffer(wl_list[int(rng.integers(0, len(wl_list)))], dtype=np.uint8)
        ext[r, 3:19] = w
    ext[scored[3], 5] = ord("N")                        # and two irregular rows
    el[scored[10]] = 18
    ext[scored[10], 18:] = 0
    pos = {"front": [0], "middle": [200], "end": [len(wl_list)], "twice": [150, 320]}[where]
    for p in reversed(pos):
        wl_list.insert(p, b"")
    for tie in (0, 1):
        hit, ed, end, flags = _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, tie)
    t = engine.timing(0)
    assert t["rows_bitparallel"] > 500 and t["rows_mixed_fix"] == t["rows_bitparallel"]      # the packed kernels ran, every scored row was re-scored
    n_scored_hits = int(((hit >= 0) & ((flags & _lib.BZ_F_EXACT) == 0)).sum())
    if where == "front":
        assert n_scored_hits > 20        # '' first: falsy, a distance-0 barcode behind it replaces it and is assigned
    if where == "end":
        assert n_scored_hits == 0        # '' last: it ties with (or undercuts) whatever was found before it


@pytest.mark.parametrize("scan_mode", **MODES)
@pytest.mark.parametrize("max_ed", [2, 1])
def test_mixed_whitelist_keeps_the_fast_path(engine, oracle, max_ed, scan_mode):
    """5,000 regular barcodes + a handful of irregular entries (15-mer, 18-mer, N, lower case -- all variants of REAL
    cell barcodes, so rows do come close to them, and at the front / middle / end of the list, so the order-dependent
    fold sees them on both sides of the regular near matches): the regular rows stay on the seeded kernels, only rows
    within max_ed of an irregular entry are re-scored over the whole list, and every row equals the oracle's."""
    rows, wl = _synthetic(10_000, 5000, err=3.0, seed=71)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl_list = [w.tobytes() for w in wl]
    w0, w1, w2, w3, w4 = (wl_list[k] for k in (5, 1200, 2500, 3700, 4999))
    some = np.flatnonzero((bl == 16) & (el == 20))[:40]              # reads of those five cells, whatever the draw gave them
    for j, r in enumerate(some):
        ext[r, 2:18] = np.frombuffer((w0, w1, w2, w3, w4)[j % 5], dtype=np.uint8)
        if j >= 20:
            ext[r, 9] = ord("ACGT"[(("ACGT".index(chr(ext[r, 9]))) + 1) % 4])        # ... half of them with one substitution
    wl_list.insert(0, w0[:15])                      # a 15-mer in front of everything
    wl_list.insert(1500, w1[:7] + b"N" + w1[8:])
    wl_list[wl_list.index(w2)] = w2 + b"AC"         # an 18-mer INSTEAD of its cell's barcode: that cell's rows can only go to it
    wl_list.insert(4000, w3.lower())
    wl_list.append(w4[1:])                          # a 15-mer suffix at the very end
    wl_list.append(w0[:15])                         # and a duplicate of the first irregular entry
    hit, ed, end, flags = _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, scan_mode=scan_mode)
    t = engine.timing(0)
    assert t["rows_generic"] == 0 and t["rows_bitparallel"] > 2000
    assert 0 < t["rows_mixed_fix"] < 100, t                          # the rows of five cells out of 5,000, no more
    assert (hit == wl_list.index(w2 + b"AC")).sum() > 0              # some rows ARE assigned to an irregular entry
    counts = None
    engine.counts_reset()
    hit2, *_ = engine.assign(ext, el, bo, bl, max_ed)
    counts = engine.counts()
    assert np.array_equal(hit2, hit)
    assert np.array_equal(counts[:len(wl_list)], np.bincount(hit[hit >= 0], minlength=len(wl_list)))   # caller's indices


@pytest.mark.parametrize("tie", [0, 1])
@pytest.mark.parametrize("max_ed", [0, 1, 2, 3])
def test_mixed_whitelist_odd_entries(engine, oracle, max_ed, tie):
    """Irregular entries of every kind the mixed path scores differently: shorter than max_ed (within reach of EVERY text:
    every scored row is re-scored by k_fix_rows), 1-, 3-, 17-, 20-character entries (bit-parallel), 21-, 28- and 30-character
    ones (longer than the text: the text becomes the pattern, byte-wise; longer ones in
    test_whitelist_entries_longer_than_the_row), lower case and N inside; texts of 16..20 columns."""
    rows, wl = _synthetic(2500, 300, err=3.0, seed=81)
    ext, el, bo, bl = synth.pack_ext(rows)
    for r in range(ext.shape[0]):
        if bl[r] and r % 4 == 1:
            ext[r, :19] = ext[r, 1:20]; el[r] = 19; bo[r] = 1
        elif bl[r] and r % 4 == 2:
            ext[r, :16] = ext[r, 2:18]; el[r] = 16; bo[r] = 0
    ext[np.arange(ext.shape[1])[None, :] >= el[:, None]] = 0
    wl_list = [w.tobytes() for w in wl]
    a, b, c = wl_list[10], wl_list[100], wl_list[200]
    odd = [b"A", b"ACG", a + b"T", b"GG" + b + b"CA", (c + a)[:30], c + b"ACGTA", (a + b)[:28], a.lower(), b[:5] + b"N" + b[6:], c[:15] + b"a"]
    if max_ed >= 2:
        odd.append(b"TC")
    rng = np.random.default_rng(4)
    for k, w in enumerate(odd):
        wl_list.insert(int(rng.integers(0, len(wl_list) + 1)), w)
    _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, tie)
    t = engine.timing(0)
    assert t["rows_generic"] == 0 and t["rows_bitparallel"] > 500 and t["rows_mixed_fix"] > 0
    if max_ed >= 1:
        assert t["rows_mixed_fix"] == t["rows_bitparallel"]          # "A" is within one edit of any text


def test_list_without_a_regular_entry(engine, oracle):
    """No 16-nt ACGT entry at all (here: every barcode cut to 15 nt): no seed index can be built, every row takes the
    workgroup-per-row scan (entries scored bit-parallel through their match masks on plain-ACGT rows, byte-wise on the
    others).  Three quarters irregular, on the other hand, keeps the seeded kernels for the regular quarter."""
    rows, wl = _synthetic(600, 400, err=2.0, seed=72)
    ext, el, bo, bl = synth.pack_ext(rows)
    ext[7, 4] = ord("N"); ext[11, 9] |= 0x20                  # two rows that are not plain ACGT: byte-wise inside the same kernel
    for cut in (lambda w: w[:15], lambda w: w + b"AC", lambda w: w.lower(), lambda w: w[:3] + b"N" + w[4:]):
        wl_list = [cut(w.tobytes()) for w in wl]
        for max_ed, tie in ((2, 0), (1, 1), (4, 0), (9, 0)):             # 9: above the kernel's range, the one-lane scan
            _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, tie)
        t = engine.timing(0)
        assert t["rows_bitparallel"] == 0 and t["rows_generic"] > 300
    wl_list = [w.tobytes() if k % 4 == 3 else w.tobytes()[:15] for k, w in enumerate(wl)]      # 300 irregular entries of 400
    _compare(engine, oracle, ext, el, bo, bl, wl_list, 2)
    t = engine.timing(0)
    assert t["rows_bitparallel"] > 300 and t["rows_generic"] == 2          # the two rows that are not plain ACGT


def test_near_duplicate_whitelist_order_dependence(engine, oracle):
    """Dense ties: the fold is order dependent; shuffled orders must each match the oracle."""
    base = np.frombuffer(b"ACGTACGTTGCATGCA", dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rng = np.random.default_rng(9)
    wl = [base.tobytes()]
    for p in range(16):
        for c in acgt:
            if c != base[p]:
                w = base.copy(); w[p] = c; wl.append(w.tobytes())
    for _ in range(200):
        w = base.copy()
        for p in rng.choice(16, size=2, replace=False):
            w[p] = acgt[(np.flatnonzero(acgt == w[p])[0] + rng.integers(1, 4)) % 4]
        wl.append(w.tobytes())
    wl = list(dict.fromkeys(wl))
    n = 2000
    ext = np.zeros((n, 32), dtype=np.uint8)
    for r in range(n):
        w = base.copy()
        for p in rng.choice(16, size=rng.integers(0, 4), replace=False):
            w[p] = acgt[rng.integers(0, 4)]
        ext[r, :20] = np.concatenate([acgt[rng.integers(0, 4, 2)], w, acgt[rng.integers(0, 4, 2)]])
    el = np.full(n, 20, np.uint8); bo = np.full(n, 2, np.uint8); bl = np.full(n, 16, np.uint8)
    outs = []
    for seed in range(3):
        order = list(wl)
        np.random.default_rng(seed).shuffle(order)
        for tie in (0, 1):
            hit, *_ = _compare(engine, oracle, ext, el, bo, bl, order, 2, tie=tie)
            for mode in (1, 2, 3, FUSED):
                hit_b, *_ = _compare(engine, oracle, ext, el, bo, bl, order, 2, tie=tie, scan_mode=mode)
                assert np.array_equal(hit, hit_b)
        assert engine.timing(0)["rows_redo"] >= 0
        outs.append(np.array([order[h] if h >= 0 else b"" for h in hit], dtype=object))
    assert any((outs[0] != outs[k]).any() for k in (1, 2)), "expected at least one order-dependent row"


def test_low_complexity_whitelist_fills_the_candidate_fifo(engine, oracle):
    """Homopolymer-rich barcodes have tiny 3-gram sets, so the filter passes almost every pair of a
    low-complexity read: every lane's candidate FIFO overflows many times and drains mid-scan."""
    rng = np.random.default_rng(31)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    wl = set()
    while len(wl) < 700:
        a, b = acgt[rng.integers(0, 4, 2)]
        w = np.where(rng.random(16) < 0.85, a, b).astype(np.uint8)
        for p in rng.choice(16, size=rng.integers(0, 2), replace=False):
            w[p] = acgt[rng.integers(0, 4)]
        wl.add(w.tobytes())
    wl = list(wl)
    n = 3000
    ext = np.zeros((n, 32), dtype=np.uint8)
    for r in range(n):
        w = np.frombuffer(wl[rng.integers(0, len(wl))], dtype=np.uint8).copy()
        for p in rng.choice(16, size=rng.integers(0, 4), replace=False):
            w[p] = acgt[rng.integers(0, 4)]
        ext[r, :20] = np.concatenate([acgt[rng.integers(0, 4, 2)], w, acgt[rng.integers(0, 4, 2)]])
    el = np.full(n, 20, np.uint8); bo = np.full(n, 2, np.uint8); bl = np.full(n, 16, np.uint8)
    redo = 0
    for max_ed in (1, 2):
        for shortcut in (True, False):
            for mode in (0, 2, 3, FUSED):
                _compare(engine, oracle, ext, el, bo, bl, wl, max_ed, shortcut=shortcut, scan_mode=mode)
                if mode in (0, 3, FUSED):
                    redo += engine.timing(0)["rows_redo"]
    assert redo > 0, "expected rows whose candidate lists overflow (the pair-seeded scan hands them to the ordered filtered scan)"


def test_seeded_filtered_bruteforce_agree_large(engine):
    """BASELINE configs[2] shape (3v2 kit, 10-nt UMI, 20,000 cells) at a size the oracle would need
    minutes for: the two device paths must agree on every row (and each is oracle-checked above)."""
    rows, wl = _synthetic(120_000, 20_000, err=2.0, seed=41, umi_len=10)
    ext, el, bo, bl = synth.pack_ext(rows)
    engine.set_whitelist_codes16(wl)
    a = engine.assign(ext, el, bo, bl, 2)
    for mode in (1, 2, 3, FUSED):
        engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0 if mode == FUSED else mode)
        engine.set_option(_lib.BZ_OPT_PAIR_FUSED, int(mode == FUSED))
        b = engine.assign(ext, el, bo, bl, 2)
        engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
        engine.set_option(_lib.BZ_OPT_PAIR_FUSED, 0)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    assert (a[0] >= 0).sum() > 60_000


def test_single_pair_entry_points(engine, oracle):
    rng = np.random.default_rng(3)
    alpha = "ACGTN"
    for _ in range(300):
        a = "".join(alpha[i] for i in rng.integers(0, 5, rng.integers(0, 20)))
        b = "".join(alpha[i] for i in rng.integers(0, 5, rng.integers(0, 40)))
        k = int(rng.integers(0, 6))
        assert engine.sub_edit_distance(a, b, k) == oracle.sub_edit_distance(a, b, k), (a, b, k)
        assert engine.edit_distance(a, b, k) == oracle.edit_distance(a, b, k), (a, b, k)
    # the stage-1 call shape: long text first, short pattern second (polyT_adaptor_finder.py:53,139)
    assert engine.sub_edit_distance("TTTTCTTCCGATCTACGT" * 3, "CTTCCGATCT", 2) == \
        oracle.sub_edit_distance("TTTTCTTCCGATCTACGT" * 3, "CTTCCGATCT", 2)


def test_empty_and_tiny_batches(engine, oracle):
    rows, wl = _synthetic(70, 50, seed=15)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl_list = [w.tobytes() for w in wl]
    engine.set_whitelist(wl_list)
    hit, ed, end, flags = engine.assign(ext[:0], el[:0], bo[:0], bl[:0], 2)
    assert hit.size == 0
    for n in (1, 2, 63, 64, 65):
        _compare(engine, oracle, ext[:n], el[:n], bo[:n], bl[:n], wl_list, 2)
    _compare(engine, oracle, ext, el, bo, bl, wl_list[:1], 2)
    _compare(engine, oracle, ext, el, bo, bl, wl_list[:3], 2)
    for max_ed in (0, 2, 5, 9):                   # an empty whitelist (no cell called): the reference's loop has nothing to walk
        _compare(engine, oracle, ext, el, bo, bl, [], max_ed)
    assert (engine.assign(ext, el, bo, bl, 2)[0] == -1).all()


def test_counts_histogram(engine, oracle):
    rows, wl = _synthetic(5000, 100, seed=16)
    ext, el, bo, bl = synth.pack_ext(rows)
    engine.set_whitelist([w.tobytes() for w in wl])
    engine.counts_reset()
    hit, *_ = engine.assign(ext, el, bo, bl, 2)
    hit2, *_ = engine.assign(ext[:1000], el[:1000], bo[:1000], bl[:1000], 2, slot=1)
    c = engine.counts()
    expect = np.bincount(hit[hit >= 0], minlength=100) + np.bincount(hit2[hit2 >= 0], minlength=100)
    assert np.array_equal(c[:100], expect)
    assert c[100] == (hit >= 0).sum() + (hit2 >= 0).sum() and c[101] == 6000


def test_counts_histogram_with_both_slots_in_flight(engine):
    """Double buffering: the two slots' runs overlap on their own streams and add to the same
    per-barcode counts.  Every run must be counted exactly once (each slot has its own histogram
    partials, the reduce adds with atomics)."""
    rows, wl = _synthetic(300_000, 3000, seed=18)
    ext, el, bo, bl = synth.pack_ext(rows)
    half = ext.shape[0] // 2
    engine.set_whitelist_codes16(wl)
    engine.stage(0, ext[:half], el[:half], bo[:half], bl[:half])
    engine.stage(1, ext[half:], el[half:], bo[half:], bl[half:])
    engine.run(0, 2), engine.sync(0)
    hit_a = engine.fetch(0)[0]
    engine.run(1, 2), engine.sync(1)
    hit_b = engine.fetch(1)[0]
    assert (hit_a >= 0).sum() > 50_000 and (hit_b >= 0).sum() > 50_000
    engine.counts_reset()
    reps = 25
    for _ in range(reps):                       # back to back on both streams, no sync in between
        engine.run(0, 2)
        engine.run(1, 2)
    engine.sync(0), engine.sync(1)
    c = engine.counts()
    expect = reps * (np.bincount(hit_a[hit_a >= 0], minlength=3000) + np.bincount(hit_b[hit_b >= 0], minlength=3000))
    assert np.array_equal(c[:3000], expect)
    assert c[3000] == expect.sum() and c[3001] == reps * ext.shape[0]
    for slot, ref in ((0, hit_a), (1, hit_b)):  # and the rows themselves are untouched by the overlap
        assert np.array_equal(engine.fetch(slot)[0], ref)


@pytest.mark.parametrize("scan_mode", **MODES)
def test_barcode_anywhere_in_the_text(engine, oracle, scan_mode):
    """The seeds' window ranges at their limits: a whitelist barcode with 0-2 random edit operations placed at EVERY
    offset of a 16-20 nt text (the reference's scan is semi-global: `sub_edit_distance` finds it wherever it sits),
    with flanks drawn from the barcode's own bases so that shifted alignments tie."""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    wl = [acgt[rng.integers(0, 4, 16)].tobytes() for _ in range(400)]
    wl = list(dict.fromkeys(wl))
    n = 12000
    ext = np.zeros((n, 32), dtype=np.uint8)
    el = np.zeros(n, np.uint8)
    for r in range(n):
        w = list(np.frombuffer(wl[rng.integers(0, len(wl))], dtype=np.uint8))
        for _ in range(int(rng.integers(0, 3))):                  # 0-2 edit operations
            op, pos = rng.integers(0, 3), int(rng.integers(0, len(w)))
            if op == 0:
                w[pos] = acgt[rng.integers(0, 4)]
            elif op == 1:
                del w[pos]
            else:
                w.insert(pos, acgt[rng.integers(0, 4)])
        length = int(rng.integers(max(16, len(w)), 21))
        start = int(rng.integers(0, length - len(w) + 1))
        text = acgt[rng.integers(0, 4, length)].copy()
        if r % 3 == 0:                                            # low-complexity flanks: many tied alignments
            text[:] = w[0]
        text[start:start + len(w)] = w
        ext[r, :length] = text
        el[r] = length
    bo = np.minimum(2, el - 16).astype(np.uint8)
    bl = np.full(n, 16, np.uint8)
    for max_ed in ((2,) if scan_mode in (3, FUSED) else (0, 1, 2, 3)):
        for tie in (0, 1):
            hit, ed, *_ = _compare(engine, oracle, ext, el, bo, bl, wl, max_ed, tie=tie, shortcut=False, scan_mode=scan_mode)
    assert (hit >= 0).sum() > n // 2


def test_full_10x_whitelist_stress(engine, oracle):
    """BASELINE configs[3] shape: rows against the full-size 10x list stand-in (6,794,880 distinct 16-mers).
    The oracle would need hours, so the bar here is agreement of three independent device paths on every row
    (pair-seeded one-kernel form, ordered 3-gram-filtered scan, brute force -- each oracle-checked at small sizes above)."""
    full = synth.full_whitelist_codes(synth.FULL_WHITELIST_SIZE, seed=1)
    assert full.size == synth.FULL_WHITELIST_SIZE
    cells = synth.cell_whitelist_codes(5000, full, seed=2)
    rows = synth.make_rows(6000, cells, full, sub=0.02, ins=0.01, dele=0.01, seed=5)
    ext, el, bo, bl = synth.pack_ext(rows)
    engine.set_whitelist_codes16(synth.codes_to_bytes(full))
    outs = {}
    for mode in (0, 2, 1):
        engine.set_option(_lib.BZ_OPT_SCAN_MODE, mode)
        outs[mode] = engine.assign(ext, el, bo, bl, 2)
    engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
    for mode in (2, 1):
        for x, y in zip(outs[0], outs[mode]):
            assert np.array_equal(x, y), f"mode 0 vs mode {mode}"
    hit, ed, end, flags = outs[0]
    assert (hit >= 0).sum() > 2000
    # with 6.8M barcodes many rows have several near matches: ambiguity must show up
    assert ((flags & _lib.BZ_F_AMBIG) != 0).sum() > 0


def test_rccl_single_rank_collective(engine):
    """The one collective of the path (per-barcode counts + [assigned, total]) through RCCL itself, world size 1:
    librccl is dlopen'ed, the id comes from ncclGetUniqueId, the all-reduce runs on the device."""
    from blaze_amd import dist
    rows, wl = _synthetic(3000, 120, seed=17)
    ext, el, bo, bl = synth.pack_ext(rows)
    engine.set_whitelist([w.tobytes() for w in wl])
    engine.counts_reset()
    hit, *_ = engine.assign(ext, el, bo, bl, 2)
    comm = dist.RcclComm(engine, 0, 1)
    try:
        counts = comm.allreduce_counts_on_device()
        assert np.array_equal(counts[:120], np.bincount(hit[hit >= 0], minlength=120))
        assert counts[120] == (hit >= 0).sum() and counts[121] == 3000
        assert comm.allreduce_max_f64(2.5) == 2.5
        assert np.array_equal(comm.allreduce_sum_i64(np.arange(100)), np.arange(100))
        comm.barrier()
    finally:
        engine.comm_destroy()


def test_submit_pinned_equals_submit(engine):
    """Rows and results in pinned host memory (bz_host_alloc), DMA'd straight from / into the caller's arrays, on both
    slots: the same results as the staged submit; pageable arrays are refused, bad rows are named."""
    rows, wl = _synthetic(50_000, 2000, err=3.0, seed=61)
    ext, el, bo, bl = synth.pack_ext(rows)
    engine.set_whitelist_codes16(wl)
    ref = engine.assign(ext, el, bo, bl, 2)
    bufs = [engine.pinned_rows(ext, el, bo, bl), engine.pinned_rows(ext[::-1], el[::-1], bo[::-1], bl[::-1])]
    try:
        engine.submit_pinned(0, bufs[0], 2)
        engine.submit_pinned(1, bufs[1], 2)
        for slot, sl in ((0, slice(None)), (1, slice(None, None, -1))):
            got = engine.wait_pinned(slot, bufs[slot])
            for g, r in zip(got, ref):
                assert np.array_equal(g, r[sl])
        L = _lib.load()
        hit = np.empty(ext.shape[0], np.int32)
        rc = L.bz_assign_submit_pinned(engine._h, 0, ext.ctypes.data, 32, el.ctypes.data, bo.ctypes.data, bl.ctypes.data, ext.shape[0], 2,
                                       hit.ctypes.data, hit.ctypes.data, hit.ctypes.data, hit.ctypes.data)
        assert rc == -1 and b"pinned" in L.bz_last_error()
        bufs[0].ext_len[777] = 99
        with pytest.raises(_lib.BlazeHipError, match="row 777"):
            engine.submit_pinned(0, bufs[0], 2)
    finally:
        for b in bufs:
            b.free()


@pytest.mark.parametrize("max_ed", [1, 2, 4])
def test_whitelist_with_hundreds_of_irregular_entries(engine, oracle, max_ed):
    """A list where 40 % of the entries are outside the packed shape (15-mers, 17-mers, an N, lower case -- variants of the
    regular ones, so most rows come near some of them): the regular entries keep their seeded kernels, every row near an
    irregular entry is re-scored over the whole list; equal to the oracle on every row."""
    rows, wl = _synthetic(2500, 600, err=3.0, seed=61)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl_list = [w.tobytes() for w in wl]
    rng = np.random.default_rng(13)
    for k in range(400):
        w = wl_list[int(rng.integers(0, len(wl_list)))]
        if len(w) != 16 or not w.isupper() or b"N" in w:
            continue
        new = [w[:15], w + b"A", w[:5] + b"N" + w[6:], w.lower()][k % 4]
        wl_list.insert(int(rng.integers(0, len(wl_list) + 1)), new)
    n_irr = sum(1 for w in wl_list if len(w) != 16 or not set(w) <= set(b"ACGT"))
    assert n_irr > 300
    for tie in (0, 1):
        _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, tie)
    t = engine.timing(0)
    assert t["rows_bitparallel"] > 500 and t["rows_generic"] == 0 and t["rows_mixed_fix"] > 100


@pytest.mark.parametrize("stride", [64, 61])
@pytest.mark.parametrize("mixed", [False, True], ids=["regular_list", "mixed_list"])
@pytest.mark.parametrize("max_ed", [0, 2, 3, 5])
def test_rows_up_to_the_device_limit(engine, oracle, max_ed, mixed, stride):
    """Extensions of 33..64 bytes (BZ_EXT_MAX; what the file hosts send for barcode cells of up to 60 nt): texts longer
    than the 32 columns the bit-parallel form of the workgroup scan holds, with a cell barcode verbatim or a few edits
    away somewhere inside, N / lower case in places, barcode offsets 0..2 -- against the oracle, packed (64) and byte-wise
    (61) row layout."""
    rng = np.random.default_rng(1000 + max_ed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    wl_list = list(dict.fromkeys(bytes(acgt[rng.integers(0, 4, 16)]) for _ in range(300)))
    if mixed:
        wl_list.insert(5, wl_list[2][:15])
        wl_list.insert(40, wl_list[20] + b"ACG")
        wl_list.append(wl_list[9].lower())
        wl_list.insert(60, b"")
        wl_list.insert(80, wl_list[70] + wl_list[71] + b"ACGTACGT")             # 40 characters: longer than the match masks hold
        wl_list.insert(90, (wl_list[10] + wl_list[11]) * 3)                      # 96: longer than any row (the row becomes the pattern)
        wl_list.insert(95, wl_list[12] + wl_list[13] + wl_list[14] + wl_list[15])   # 64
    n = 1500
    ext = np.zeros((n, stride), np.uint8); el = np.zeros(n, np.uint8); bo = np.zeros(n, np.uint8); bl = np.zeros(n, np.uint8)
    for r in range(n):
        L = int(rng.integers(33, stride + 1)) if r % 7 else int(rng.integers(20, 33))
        row = acgt[rng.integers(0, 4, L)].copy()
        k = r % 5
        if k != 4:                                   # plant a cell barcode (or an irregular entry), k edits away
            w = np.frombuffer(wl_list[int(rng.integers(0, len(wl_list)))] or b"A", dtype=np.uint8)[:L].copy()   # (a longer entry: its head)
            for _ in range(k if k < 3 else 0):
                w[rng.integers(0, w.size)] = acgt[rng.integers(0, 4)]
            at = int(rng.integers(0, L - w.size + 1))
            row[at:at + w.size] = w
        if r % 11 == 0:
            row[rng.integers(0, L)] = ord("N")
        if r % 13 == 0:
            row[rng.integers(0, L)] |= 0x20
        off = int(rng.integers(0, 3))
        umi = int(rng.integers(0, 3))
        ext[r, :L] = row; el[r] = L; bo[r] = off; bl[r] = L - off - umi
    bl[::50] = 0                                     # blank rows in between
    hit, ed, end, flags = _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, tie=max_ed & 1)
    assert (hit >= 0).sum() > 200 and engine.timing(0)["rows_generic"] > 1000


@pytest.mark.parametrize("longest,fast", [(25, True), (30, True), (31, True), (80, True), (127, True)])
@pytest.mark.parametrize("max_ed", [1, 2, 5])
def test_whitelist_entries_longer_than_the_row(engine, oracle, max_ed, longest, fast):
    """A list line longer than the 20-column row: the ROW is then the pattern of the replaced call and the end index a
    position inside the entry (k_fix_rows' 64-bit event words hold it; its earlier 5-bit field broke on the entries of 38
    and 80 characters the randomised runs drew).  Such entries ride beside the regular ones (k_mixed_fix / k_fix_rows);
    beyond 127 characters the end index no longer fits the int8 result and bz_set_whitelist refuses.  Rows: ordinary ones, and ones cut out of the long entry (distance 0..2 inside it)."""
    rows, wl = _synthetic(1200, 150, err=2.0, seed=400 + longest)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl_list = [w.tobytes() for w in wl]
    rng = np.random.default_rng(longest)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    long_entry = bytes(acgt[rng.integers(0, 4, longest)])
    wl_list.insert(40, long_entry)
    wl_list.insert(90, wl_list[3] + wl_list[4][:longest - 16])            # a second one that starts with a regular entry
    for r in range(0, ext.shape[0], 6):                                    # 20 columns out of the long entry, a few edits
        if bl[r] == 0:
            continue
        at = int(rng.integers(0, longest - 20 + 1))
        row = np.frombuffer(long_entry[at:at + 20], dtype=np.uint8).copy()
        for _ in range(int(rng.integers(0, 3))):
            row[rng.integers(0, 20)] = acgt[rng.integers(0, 4)]
        ext[r, :20] = row
    hit, ed, end, flags = _compare(engine, oracle, ext, el, bo, bl, wl_list, max_ed, tie=longest & 1)
    assert (hit == 40).sum() > 50 and end[hit == 40].max() > 19
    t = engine.timing(0)
    assert (t["rows_generic"] < 50) == fast, t
    if longest == 127:
        with pytest.raises(_lib.BlazeHipError, match="at most 127"):
            engine.set_whitelist(wl_list + [long_entry + b"A"])
        engine.set_whitelist(wl_list)
