"""Parity at BASELINE.json's FULL sizes, where the oracle would need hours: size-independent properties.

  configs[1]  1M nanopore-error 3v3 reads x 5,000 cells
  configs[2]  10M 3v2 reads (10-nt UMI) x 20,000 cells

Properties: (1) the independent device paths -- pair-seeded two-kernel scan (product), pair-seeded stream scan,
single-piece seeded scan, all-pairs bit-parallel scan -- agree on every row; (2) rows are independent: a permuted
batch gives the permuted results; (3) the oracle agrees on a sample of the same rows; (4) the per-barcode counts
are the histogram of the hits.  Bit-exact bar (integer / index work)."""
import numpy as np
import pytest

from blaze_amd import _lib, synth

pytestmark = pytest.mark.gpu
FLAG_MASK = _lib.BZ_F_EXACT | _lib.BZ_F_BLANK | _lib.BZ_F_AMBIG


def _workload(n_rows, n_cells, umi_len, seed, chunk=500_000):
    from concurrent.futures import ThreadPoolExecutor
    full = synth.full_whitelist_codes(400_000, seed=1)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=2)

    def part(k):                                  # slices with their own seeds: bounded host memory, numpy drops the GIL
        lo = k * chunk
        return synth.pack_ext(synth.make_rows(min(chunk, n_rows - lo), cells, full, umi_len=umi_len, seed=seed + 1000 * k))
    with ThreadPoolExecutor(8) as pool:
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
