"""The deferred histogram (blaze_hip.hip: Slot::pend, scan_kernels.h: kPiggyBins).

A run's per-barcode counts ride along with the slot's NEXT run (workgroups in front of k_classify_packed's and
k_pair_candidates' / k_scan_seeded's grids) or are flushed by whoever reads counts / list counters / timings first.  Whatever happens between
two runs -- another max_ed (no carrier kernel), profiling on, a new whitelist, a reset, a bigger batch that replaces the slot's
buffers, a fetch, the other slot, the entry point that allocates nothing -- every run is counted exactly once and the list
counters bz_slot_timing reports are those of the LAST run.  Expectations are bincounts of the fetched hits (parity of the
hits themselves is test_gpu_scan_parity.py's job).
"""
import numpy as np
import pytest

from blaze_amd import _lib, synth

pytestmark = pytest.mark.gpu

N_WL = 3000


def _workload(n_rows, seed, n_cells=N_WL):
    full = synth.full_whitelist_codes(50_000, seed=21)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=22)
    rows = synth.make_rows(n_rows, cells, full, seed=seed)
    return synth.pack_ext(rows, stride=24), synth.codes_to_bytes(cells)


def _bins(hit, n_wl=N_WL):
    return np.bincount(hit[hit >= 0], minlength=n_wl).astype(np.int64)


@pytest.fixture()
def eng(engine):
    for opt, v in ((_lib.BZ_OPT_SCAN_MODE, 0), (_lib.BZ_OPT_PAIR_FUSED, 0), (_lib.BZ_OPT_TIE_RULE, 0), (_lib.BZ_OPT_SHORTCUT, 1),
                   (_lib.BZ_OPT_FORCE_GENERIC, 0), (_lib.BZ_OPT_PROFILE, 0)):
        engine.set_option(opt, v)
    yield engine
    engine.set_option(_lib.BZ_OPT_PROFILE, 0)
    engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0)


def test_every_run_counted_once_whatever_comes_between(eng):
    (a, wl) = _workload(70_000, seed=5)
    (b, _) = _workload(41_000, seed=6)
    eng.set_whitelist_codes16(wl)
    hits = {}
    for name, rows in (("a", a), ("b", b)):
        for max_ed in (0, 1, 2, 3):
            hits[name, max_ed] = eng.assign(*rows, max_ed)[0].copy()
    eng.counts_reset()
    expect = np.zeros(N_WL, np.int64)
    n_rows = 0

    def run(slot, name, max_ed):
        nonlocal expect, n_rows
        eng.run(slot, max_ed)
        expect = expect + _bins(hits[name, max_ed])
        n_rows += (a if name == "a" else b)[0].shape[0]

    eng.stage(0, *a)
    eng.stage(1, *b)
    run(0, "a", 2); run(0, "a", 2); run(0, "a", 2)            # carried along twice, one left pending
    run(0, "a", 1)                                           # no carrier kernel on this path: flushed in front of it
    run(0, "a", 2); run(1, "b", 2); run(0, "a", 2); run(1, "b", 2)   # both slots, each with its own pending run
    assert np.array_equal(eng.fetch(0)[0], hits["a", 2])     # a fetch in between reads the CURRENT hit buffer
    run(0, "a", 3); run(0, "a", 0); run(1, "b", 2)
    run(0, "a", 1); run(0, "a", 1); run(0, "a", 1)            # the single-piece seeded scan carries too (max_ed 0, 1, 3) ...
    run(1, "b", 0); run(1, "b", 3); run(1, "b", 3); run(1, "b", 2); run(1, "b", 1)   # ... and hands over to the pair kernels and back
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, 3)                 # max_ed 2 through the single-piece seeds
    run(0, "a", 2); run(0, "a", 2)
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
    eng.set_option(_lib.BZ_OPT_PROFILE, 1)                   # profiling runs take their histogram at once
    run(0, "a", 2); run(1, "b", 2)
    t = eng.timing(0)
    assert t["rows"] == a[0].shape[0] and t["hist_ms"] > 0
    eng.set_option(_lib.BZ_OPT_PROFILE, 0)
    run(0, "a", 2); run(0, "a", 2)
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, 1)                 # brute force: another path with a pending run in front of it
    run(0, "a", 2)
    eng.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
    run(0, "a", 2)
    c = eng.counts()                                         # flushes what is still pending on both slots
    assert np.array_equal(c[:N_WL], expect)
    assert c[N_WL] == expect.sum() and c[N_WL + 1] == n_rows
    c2 = eng.counts()                                        # ... once
    assert np.array_equal(c, c2)
    for slot, name in ((0, "a"), (1, "b")):
        assert np.array_equal(eng.fetch(slot)[0], hits[name, 2])


def test_list_counters_are_those_of_the_last_run(eng):
    (a, wl) = _workload(50_000, seed=7)
    eng.set_whitelist_codes16(wl)
    eng.set_option(_lib.BZ_OPT_SHORTCUT, 0)
    eng.stage(0, *a)
    eng.run(0, 2)
    scored_all = eng.timing(0)["rows_bitparallel"]
    eng.set_option(_lib.BZ_OPT_SHORTCUT, 1)
    for _ in range(3):
        eng.run(0, 2)
    scored_shortcut = eng.timing(0)["rows_bitparallel"]      # published by the flush of the LAST run, after the carried ones
    assert 0 < scored_shortcut < scored_all
    eng.set_option(_lib.BZ_OPT_SHORTCUT, 0)
    eng.run(0, 2)
    assert eng.timing(0)["rows_bitparallel"] == scored_all   # the counter sets come back clean (zeroed by the histogram that read them)
    eng.set_option(_lib.BZ_OPT_SHORTCUT, 1)
    eng.run(0, 2); eng.run(0, 2); eng.run(0, 2); eng.run(0, 2)
    assert eng.timing(0)["rows_bitparallel"] == scored_shortcut


def test_reset_new_whitelist_and_bigger_batch_with_a_run_pending(eng):
    (a, wl) = _workload(30_000, seed=8)
    (big, _) = _workload(90_000, seed=9)
    (c5, wl5) = _workload(30_000, seed=10, n_cells=500)
    eng.set_whitelist_codes16(wl)
    hit_a = eng.assign(*a, 2)[0].copy()
    hit_big = eng.assign(*big, 2)[0].copy()
    # reset with a run pending: what was submitted before the reset is not counted after it
    eng.counts_reset()
    eng.stage(0, *a)
    eng.run(0, 2); eng.run(0, 2)
    eng.counts_reset()
    eng.run(0, 2)
    c = eng.counts()
    assert np.array_equal(c[:N_WL], _bins(hit_a)) and c[N_WL + 1] == a[0].shape[0]
    # a batch that outgrows the slot's buffers while the previous run is still pending
    small = tuple(x[:5_000] for x in a)
    hit_small = eng.assign(*small, 2)[0].copy()
    eng2_expect = _bins(hit_small) + _bins(hit_big)
    eng.counts_reset()
    eng.stage(1, *small)
    eng.run(1, 2)                                            # pending on slot 1 ...
    eng.stage(1, *big)                                       # ... whose buffers are replaced here (first use of slot 1 at this size, or not: both fine)
    eng.run(1, 2)
    c = eng.counts()
    assert np.array_equal(c[:N_WL], eng2_expect) and c[N_WL + 1] == 5_000 + big[0].shape[0]
    # a new whitelist with a run pending: its counts go with the old list, the list counters start clean
    eng.stage(0, *a)
    eng.run(0, 2); eng.run(0, 2)
    eng.set_whitelist_codes16(wl5)
    hit5 = eng.assign(*c5, 2)[0].copy()
    eng.counts_reset()
    eng.stage(0, *c5)
    eng.run(0, 2); eng.run(0, 2); eng.run(0, 2)
    c = eng.counts()
    assert np.array_equal(c[:500], 3 * _bins(hit5, 500)) and c[501] == 3 * c5[0].shape[0]
    t = eng.timing(0)
    assert t["rows"] == c5[0].shape[0] and 0 < t["rows_bitparallel"] < c5[0].shape[0]
