"""GPU parity on randomly drawn cases (tools/fuzz_parity.py's generator, fixed seeds): whitelists with families of
near-identical barcodes, irregular rows and entries, every max_ed 0..5, both tie rules, shortcut on / off, every scan
mode, three row strides -- bit-exact against the oracle.  The open-ended run of the same generator is
`python tools/fuzz_parity.py <seconds>` (profiles/r02u_fuzz_parity.txt: 69,676 cases, 74.2 M rows; the one difference it found -- whitelist lines longer than the row, a match ending past their 30th character -- is fixed and in tests/test_gpu_scan_parity.py)."""
import importlib.util
import os

import numpy as np
import pytest

from blaze_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAG_MASK = _lib.BZ_F_EXACT | _lib.BZ_F_BLANK | _lib.BZ_F_AMBIG


def _generator():
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("first_seed", [100_000, 100_100, 100_200])
def test_random_cases(engine, oracle, first_seed):
    fz = _generator()
    seen = set()
    try:
        for seed in range(first_seed, first_seed + 100):
            p, wl, ext, el, bo, bl = fz.draw_case(np.random.default_rng(seed))
            ext_s = np.ascontiguousarray(ext[:, :p["stride"]]) if p["stride"] >= int(el.max(initial=0)) else ext
            engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0 if p["scan_mode"] == 4 else p["scan_mode"])
            engine.set_option(_lib.BZ_OPT_PAIR_FUSED, int(p["scan_mode"] == 4))
            engine.set_option(_lib.BZ_OPT_TIE_RULE, p["tie"])
            engine.set_option(_lib.BZ_OPT_SHORTCUT, int(p["shortcut"]))
            engine.set_whitelist(wl)
            hit, ed, end, flags = engine.assign(ext_s, el, bo, bl, p["max_ed"])
            o_hit, o_ed, o_end, o_flag = oracle.match_rows(ext, el, bo, bl, wl, p["max_ed"], p["tie"], p["shortcut"], threads=8)
            bad = np.flatnonzero((hit != o_hit) | (ed != o_ed) | (end != o_end) | ((flags & FLAG_MASK) != o_flag))
            assert bad.size == 0, (seed, p, int(bad[0]), bytes(ext[bad[0], :el[bad[0]]]),
                                   (hit[bad[0]], ed[bad[0]], end[bad[0]], flags[bad[0]]),
                                   (o_hit[bad[0]], o_ed[bad[0]], o_end[bad[0]], o_flag[bad[0]]))
            seen.add((p["max_ed"], p["scan_mode"], p["mixed"], p["irregular_rows"]))
    finally:
        engine.set_option(_lib.BZ_OPT_SCAN_MODE, 0)
        engine.set_option(_lib.BZ_OPT_PAIR_FUSED, 0)
        engine.set_option(_lib.BZ_OPT_TIE_RULE, 0)
        engine.set_option(_lib.BZ_OPT_SHORTCUT, 1)
    assert len(seen) > 30                                  # the draw does spread over the options


def test_random_pipeline_jobs(tmp_path):
    """The three stages over drawn FASTQ files and settings (tools/fuzz_pipeline.py's generator, fixed seeds): the native
    host and the pandas host of blaze_amd.main.main write the same five files, or stop with the same exception."""
    spec = importlib.util.spec_from_file_location("fuzz_pipeline", os.path.join(ROOT, "tools", "fuzz_pipeline.py"))
    fp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fp)
    modes = set()
    for seed in range(7000, 7016):
        d = tmp_path / str(seed)
        d.mkdir()
        ns, shown, _ = fp.draw_job(np.random.default_rng(seed), str(d))
        a_n, a_p = ns("native_"), ns("pandas_")
        e_n, e_p = fp.run(a_n, "native"), fp.run(a_p, "python")
        assert e_n == e_p, (seed, shown)
        o_n, o_p = fp.outputs(a_n), fp.outputs(a_p)
        for key in o_n:
            if e_n and key == "output_fastq":
                continue
            assert o_n[key] == o_p[key], (seed, shown, key)
        modes.add((shown["kit_version"], shown["gz"], shown["batch_size"]))
    assert len(modes) > 8
