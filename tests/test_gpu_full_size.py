"""Parity at BASELINE.json's FULL sizes, where the oracle would need hours: size-independent properties.

  configs[1]  1M nanopore-error 3v3 reads x 5,000 cells
  configs[2]  10M 3v2 reads (10-nt UMI) x 20,000 cells
  configs[3]  10M reads x the full 10x list (6,794,880 barcodes): the 1.25M-row shard one of 8 GPUs takes
  configs[4]  200M 5v2 reads (10-nt UMI) x 10,000 cells: the 25M-row shard one of 8 GPUs takes, and the file legs
              (trim + restrand, native host vs the pandas mirror) at 200K reads

Properties: (1) the independent device paths -- pair-seeded two-kernel scan (product), pair-seeded stream scan,
single-piece seeded scan, all-pairs bit-parallel scan -- agree on every row; (2) rows are independent: a permuted
batch gives the permuted results; (3) the oracle agrees on a sample of the same rows; (4) the per-barcode counts
are the histogram of the hits.  Bit-exact bar (integer / index work)."""
import numpy as np
import pytest

from blaze_amd import _lib, synth

pytestmark = pytest.mark.gpu
FLAG_MASK = _lib.BZ_F_EXACT | _lib.BZ_F_BLANK | _lib.BZ_F_AMBIG


def _workload(n_rows, n_cells, umi_len, seed, chunk=500_000, full_size=400_000):
    from concurrent.futures import ThreadPoolExecutor
    full = synth.full_whitelist_codes(full_size, seed=1)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=2)

    def part(k):                                  # slices with their own seeds: bounded host memory, numpy drops the GIL
        lo = k * chunk
        return synth.pack_ext(synth.make_rows(min(chunk, n_rows - lo), cells, full, umi_len=umi_len, seed=seed + 1000 * k))
    with ThreadPoolExecutor(12) as pool:
        parts = list(pool.map(part, range((n_rows + chunk - 1) // chunk)))
    ext, el, bo, bl = (np.concatenate([p[i] for p in parts]) for i in range(4))
    return ext, el, bo, bl, synth.codes_to_bytes(cells)


def _run(engine, ext, el, bo, bl, mode, fused=0):
    engine.set_option(_lib.BZ_OPT_SCAN_MODE, mode)
    engine.set_option(_lib.BZ_OPT_PAIR_FUSED, fused)
    try:
        return engine.assign(ext, el, bo, bl, 2)
    finally:
        engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
        engine.set_option(_lib.BZ_OPT_PAIR_FUSED, 0)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_rows,n_cells,umi_len,sample", [(1_000_000, 5_000, 12, 20_000), (10_000_000, 20_000, 10, 6_000)],
                         ids=["configs1_1Mx5k", "configs2_10Mx20k"])
def test_full_size_properties(engine, oracle, n_rows, n_cells, umi_len, sample):
    ext, el, bo, bl, wl = _workload(n_rows, n_cells, umi_len, seed=3)
    engine.set_whitelist_codes16(wl)
    engine.counts_reset()
    ref = _run(engine, ext, el, bo, bl, 0)                                  # product path
    counts = engine.counts()
    hit = ref[0]
    assert (hit >= 0).mean() > 0.7
    # (4) counts = histogram of the hits
    assert np.array_equal(counts[:n_cells], np.bincount(hit[hit >= 0], minlength=n_cells))
    assert counts[n_cells] == (hit >= 0).sum() and counts[n_cells + 1] == n_rows
    # (1) independent device paths agree on every row
    for name, mode, fused in (("all pairs", 1, 0), ("pair stream", 0, 1), ("single-piece seeds", 3, 0)):
        got = _run(engine, ext, el, bo, bl, mode, fused)
        for a, b, what in zip(ref, got, ("hit", "ed", "end", "flags")):
            if what == "flags":
                a, b = a & FLAG_MASK, b & FLAG_MASK
            bad = np.flatnonzero(a != b)
            assert bad.size == 0, f"{name}: {bad.size} rows differ in {what}, first row {bad[0]}"
    # (2) permutation equivariance
    perm = np.random.default_rng(5).permutation(n_rows)
    got = _run(engine, ext[perm], el[perm], bo[perm], bl[perm], 0)
    for a, b in zip(ref, got):
        assert np.array_equal(a[perm], b)
    # (3) the oracle on a strided sample of the same rows
    idx = np.arange(0, n_rows, n_rows // sample)[:sample]
    wl_list = [w.tobytes() for w in wl]
    o_hit, o_ed, o_end, o_flag = oracle.match_rows(ext[idx], el[idx], bo[idx], bl[idx], wl_list, 2, threads=16)
    assert np.array_equal(ref[0][idx], o_hit) and np.array_equal(ref[1][idx], o_ed) and np.array_equal(ref[2][idx], o_end)
    assert np.array_equal(ref[3][idx] & FLAG_MASK, o_flag)


def _agree(ref, got, name):
    for a, b, what in zip(ref, got, ("hit", "ed", "end", "flags")):
        if what == "flags":
            a, b = a & FLAG_MASK, b & FLAG_MASK
        bad = np.flatnonzero(a != b)
        assert bad.size == 0, f"{name}: {bad.size} rows differ in {what}, first row {bad[0]}"


@pytest.mark.timeout(900)
def test_configs3_shard_1p25M_rows_vs_full_10x_list(engine, oracle):
    """BASELINE configs[3] at the size one of 8 GPUs sees: 1.25M rows against all 6,794,880 barcodes.  The product path
    here is k_pair_stream (candidate lists cannot be written down for this list); every row must agree with the
    all-pairs scan (8.5e12 pairs, no seeds, no filter), counts = histogram of the hits, and the oracle agrees on a
    strided sample (the oracle scores ~40 M pairs/s on the box's 16 threads: a sample of scored rows, not thousands)."""
    n_rows = 1_250_000
    ext, el, bo, bl, _ = _workload(n_rows, 5_000, 12, seed=3, full_size=synth.FULL_WHITELIST_SIZE)
    full = synth.full_whitelist_codes(synth.FULL_WHITELIST_SIZE, seed=1)
    wl = synth.codes_to_bytes(full)
    engine.set_whitelist_codes16(wl)
    engine.counts_reset()
    ref = _run(engine, ext, el, bo, bl, 0)
    t = engine.timing(0)
    assert t["rows_generic"] == 0 and t["rows_bitparallel"] > 200_000
    counts = engine.counts()
    hit = ref[0]
    assert np.array_equal(counts[:full.size], np.bincount(hit[hit >= 0], minlength=full.size))
    assert counts[full.size] == (hit >= 0).sum() and counts[full.size + 1] == n_rows
    assert ((ref[3] & _lib.BZ_F_AMBIG) != 0).sum() > 1000          # 6.8M barcodes: near matches abound
    _agree(ref, _run(engine, ext, el, bo, bl, 1), "all pairs")
    scored = np.flatnonzero((ref[3] & (_lib.BZ_F_EXACT | _lib.BZ_F_BLANK)) == 0)
    idx = np.concatenate([scored[:: scored.size // 100][:100], np.arange(0, n_rows, n_rows // 300)[:300]])
    o = oracle.match_rows(ext[idx], el[idx], bo[idx], bl[idx], (np.ascontiguousarray(wl).reshape(-1), np.arange(full.size + 1, dtype=np.int64) * 16),
                          2, threads=16)
    assert np.array_equal(ref[0][idx], o[0]) and np.array_equal(ref[1][idx], o[1]) and np.array_equal(ref[2][idx], o[2])
    assert np.array_equal(ref[3][idx] & FLAG_MASK, o[3])


@pytest.mark.timeout(900)
def test_configs4_shard_25M_rows_10nt_umi(engine, oracle):
    """BASELINE configs[4] at the size one of 8 GPUs sees: 25M 5v2-shaped rows (10-nt UMI) against 10,000 cells in ONE
    submit.  Product path vs the all-pairs scan on every row, counts, and the oracle on a strided sample."""
    n_rows, n_cells = 25_000_000, 10_000
    ext, el, bo, bl, wl = _workload(n_rows, n_cells, 10, seed=11, chunk=1_000_000)
    engine.set_whitelist_codes16(wl)
    engine.counts_reset()
    ref = _run(engine, ext, el, bo, bl, 0)
    counts = engine.counts()
    hit = ref[0]
    assert np.array_equal(counts[:n_cells], np.bincount(hit[hit >= 0], minlength=n_cells))
    assert counts[n_cells] == (hit >= 0).sum() and counts[n_cells + 1] == n_rows and (hit >= 0).mean() > 0.7
    _agree(ref, _run(engine, ext, el, bo, bl, 1), "all pairs")
    idx = np.arange(0, n_rows, n_rows // 6000)[:6000]
    o = oracle.match_rows(ext[idx], el[idx], bo[idx], bl[idx], [w.tobytes() for w in wl], 2, threads=16)
    assert np.array_equal(ref[0][idx], o[0]) and np.array_equal(ref[1][idx], o[1]) and np.array_equal(ref[2][idx], o[2])
    assert np.array_equal(ref[3][idx] & FLAG_MASK, o[3])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("restrand", [True, False], ids=["restrand", "no_restrand"])
def test_configs4_files_200k_reads_native_equals_pandas(tmp_path, restrand):
    """configs[4]'s file legs (5v2 shape: 10-nt UMI, 10,000 cells, trim, restrand on / --no-restrand) at 200K reads of
    ~900 nt: the native host pipeline writes the bytes the pandas mirror of the reference's loop writes (that mirror is
    pinned by the goldens recorded from the real reference), plain and `.gz` (decompressed stream, F9)."""
    import gzip
    from blaze_amd import read_assignment as ra
    fns, pc, wl, _ = synth.write_file_job(str(tmp_path), 200_000, 10_000, umi_len=10, n_files=4)
    kw = dict(fastq_fns=fns, putative_bc_csv=pc, whitelsit_csv=wl, max_ed=2, batchsize=1000, restrand=restrand)
    exp = ra.assign_read(fastq_out=str(tmp_path / "pandas.fastq"), n_process=1, host="python", **kw)
    want = open(tmp_path / "pandas.fastq", "rb").read()
    assert exp[1] == 200_000 and exp[0] > 140_000
    got = ra.assign_read(fastq_out=str(tmp_path / "native.fastq"), n_process=16, host="native", **kw)
    assert tuple(got) == tuple(exp)
    assert open(tmp_path / "native.fastq", "rb").read() == want
    if restrand:
        got = ra.assign_read(fastq_out=str(tmp_path / "native.fastq.gz"), n_process=16, host="native", **kw)
        assert tuple(got) == tuple(exp)
        assert gzip.decompress(open(tmp_path / "native.fastq.gz", "rb").read()) == want


def _run_k(engine, ext, el, bo, bl, max_ed, mode):
    engine.set_option(_lib.BZ_OPT_SCAN_MODE, mode)
    try:
        return engine.assign(ext, el, bo, bl, max_ed)
    finally:
        engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("max_ed", [0, 1, 3])
def test_configs1_other_max_edit_distances(engine, oracle, max_ed):
    """--max-edit-distance is a user flag (parser.py:100): configs[1]'s 1M rows x 5,000 cells at 0, 1 and 3 -- the seeded
    product scans (8|8-nt pieces; four 4-nt pieces at 3) against the all-pairs scan on every row and the oracle on a
    strided sample, and the high-error setting of SURVEY 8d (sub 3 %, ins 1 %, del 2 %) so that distance 3 is populated."""
    from concurrent.futures import ThreadPoolExecutor
    n_rows, n_cells, chunk = 1_000_000, 5_000, 250_000
    full = synth.full_whitelist_codes(400_000, seed=1)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=2)
    with ThreadPoolExecutor(8) as pool:
        parts = list(pool.map(lambda k: synth.pack_ext(synth.make_rows(chunk, cells, full, sub=0.03, ins=0.01, dele=0.02, seed=40 + k)),
                              range(n_rows // chunk)))
    ext, el, bo, bl = (np.concatenate([p[i] for p in parts]) for i in range(4))
    wl = synth.codes_to_bytes(cells)
    engine.set_whitelist_codes16(wl)
    ref = _run_k(engine, ext, el, bo, bl, max_ed, 0)
    assert engine.timing(0)["rows_generic"] == 0
    _agree(ref, _run_k(engine, ext, el, bo, bl, max_ed, 1), f"all pairs, max_ed {max_ed}")
    if max_ed:
        assert (ref[1][ref[0] >= 0] == max_ed).sum() > 1000                      # assignments at the cut-off itself
    idx = np.arange(0, n_rows, n_rows // 8000)[:8000]
    o = oracle.match_rows(ext[idx], el[idx], bo[idx], bl[idx], [w.tobytes() for w in wl], max_ed, threads=16)
    assert np.array_equal(ref[0][idx], o[0]) and np.array_equal(ref[1][idx], o[1]) and np.array_equal(ref[2][idx], o[2])
    assert np.array_equal(ref[3][idx] & FLAG_MASK, o[3])
