"""Stage 2's empty-droplet barcode list on MI355X: drop-in for
/root/reference/blaze/main.py:151-284 `get_bc_whitelist`, same arguments and return values.

What changed underneath: the loop of :248-255 / :273-280,

    for k in ept_bc_candidate:
        if min([edit_distance(k, x, max_ed=5) for x in cells_bc.keys()]) >= 5: ept_bc.append(k)
        if len(ept_bc) > 2000: break

(candidates x cells third-party `edit_distance` calls from Python) is one device launch per block of
candidates (bz_far_from_all, blaze_amd/csrc/scan_kernels.h:k_far_from_all); the order of the
candidates, the `>` in the stop rule (2001 barcodes come out, as in test/expect_output/test_emtpy_bc_list.csv)
and everything around the loop are kept, the knee plot (:286-302) included where matplotlib is installed (a host
without it gets a warning instead of the picture).  No CPU scoring path: without libblaze_hip.so and a GPU the
empty-drop list raises.
"""
from __future__ import annotations

import io
import logging
import zipfile

import numpy as np

from . import _lib, helper
from .engine import pack_whitelist

logger = logging.getLogger(__name__)

DEFAULT_EMPTY_DROP_MIN_ED = 5      # config.py:64
DEFAULT_EMPTY_DROP_NUM = 2000      # config.py:65
_BLOCK = 4096                      # candidates per device launch


def default_count_threshold_calculation(count_array, exp_cells):          # config.py:52-54
    top_count = np.sort(count_array)[::-1][:exp_cells]
    return np.quantile(top_count, 0.95) / 20


def high_sensitivity_threshold_calculation(count_array, exp_cells):       # config.py:56-58
    top_count = np.sort(count_array)[::-1][:exp_cells]
    return np.quantile(top_count, 0.95) / 200


def far_from_all(engine, candidates, cells, min_ed=DEFAULT_EMPTY_DROP_MIN_ED):
    """bool array: candidates[i] is at global edit distance >= min_ed from every string of `cells`."""
    out = np.zeros(len(candidates), dtype=np.uint8)
    if not len(candidates):
        return out.astype(bool)
    a, ao = pack_whitelist(candidates)
    b, bo = pack_whitelist(cells)
    _lib.check(_lib.load().bz_far_from_all(engine._h, a.ctypes.data, ao.ctypes.data, len(candidates), b.ctypes.data,
                                           bo.ctypes.data, len(cells), int(min_ed), out.ctypes.data))
    return out.astype(bool)


def empty_drop_list(candidates, cells, engine=None, min_ed=DEFAULT_EMPTY_DROP_MIN_ED, max_n=DEFAULT_EMPTY_DROP_NUM):
    """main.py:248-255: candidates in order, keep those far from every cell barcode, stop once MORE than max_n are kept."""
    if engine is None:
        from .read_assignment import get_engine
        engine = get_engine()
    candidates, cells = list(candidates), list(cells)
    if candidates and not cells:
        raise ValueError("min() arg is an empty sequence")       # what the reference's min([]) raises
    ept_bc = []
    for lo in range(0, len(candidates), _BLOCK):
        block = candidates[lo:lo + _BLOCK]
        for k, far in zip(block, far_from_all(engine, block, cells, min_ed)):
            if far:
                ept_bc.append(k)
            if len(ept_bc) > max_n:
                return ept_bc
    return ept_bc


def knee_plot(counts, threshold=None, out_fn='knee_plot.png'):
    """Barcode rank plot of the high-quality putative barcode counts with the cell-calling threshold (main.py:286-302):
    counts in descending order on log-log axes, the threshold as a dashed red line, 8 x 8 inches.  Drawn through
    matplotlib's object interface on an Agg canvas (no pyplot state), so it may run on a thread of its own."""
    try:
        from matplotlib.backends.backend_agg import FigureCanvasAgg
        from matplotlib.figure import Figure
    except ImportError:
        logger.warning(helper.warning_msg(f"matplotlib is not installed: the knee plot `{out_fn}` is not drawn."))
        return
    ranked = sorted(counts, reverse=True)
    fig = Figure(figsize=(8, 8))
    FigureCanvasAgg(fig)
    ax = fig.add_subplot()
    ax.set_title('Barcode rank plot (from high-quality putative BC)')
    ax.loglog(ranked, marker='o', linestyle="", alpha=1, markersize=6)
    ax.set_xlabel('Barcodes')
    ax.set_ylabel('Read counts')
    ax.axhline(y=threshold, color='r', linestyle='--', label='cell calling threshold')
    ax.legend()
    fig.savefig(out_fn)


def _start_knee_plot(counts, threshold, out_fn, plot_threads):
    """The plot now, or -- when the caller collects threads -- on a thread the caller joins (main.main: the picture is
    not an input of stage 3, and rendering 10^5 markers takes as long as the rest of a small job)."""
    if plot_threads is None:
        knee_plot(counts, threshold, out_fn)
        return
    import threading

    def run():
        try:
            knee_plot(counts, threshold, out_fn)
        except Exception as e:        # noqa: BLE001  (a thread cannot raise into its caller: say it)
            logger.exception(e)
    t = threading.Thread(target=run, name="knee_plot")
    t.start()
    plot_threads.append(t)


def get_bc_whitelist(raw_bc_count, full_bc_whitelist=None, exp_cells=None, count_t=None, force_cell_n=None,
                     high_sensitivity_mode=False, output_empty=True, empty_max_count=np.inf,
                     out_plot_fn='knee_plot.png', args=None, engine=None, plot_threads=None):
    """Cell barcodes {barcode: count} and the empty-droplet barcode list, as the reference's function of the same name.
    `plot_threads`: a list -> the knee plot is drawn on a thread appended to it (join it before reading the picture)."""
    if args:
        full_bc_whitelist = args.full_bc_whitelist
        exp_cells = args.expect_cells
        count_t = args.count_threshold
        high_sensitivity_mode = args.high_sensitivity_mode
        out_plot_fn = args.out_plot_fn
        empty_max_count = args.empty_max_count
        force_cell_n = args.force_cells
        output_empty = args.do_empty_drops_output
    percentile_count_thres = high_sensitivity_threshold_calculation if high_sensitivity_mode \
        else default_count_threshold_calculation

    # main.py:207-225: every line stripped, and the counted barcodes that are among them kept.  The reference builds a set
    # of all 6.8M lines of a 10x list first (2.1 s, 600 MB); only membership of the counted barcodes is ever asked, so
    # the lines are streamed against the (small) set of counted barcodes instead (0.66 s, strip in C)
    counted = set(raw_bc_count)
    if full_bc_whitelist.endswith('.zip'):
        with zipfile.ZipFile(full_bc_whitelist) as zf:
            assert len(zf.namelist()) == 1
            with io.TextIOWrapper(zf.open(zf.namelist()[0]), encoding="utf-8") as f:
                listed = counted.intersection(map(str.strip, f))
    else:
        with open(full_bc_whitelist, 'r') as f:
            listed = counted.intersection(map(str.strip, f))
    raw_bc_count = {k: v for k, v in raw_bc_count.items() if k in listed}

    if force_cell_n:
        if force_cell_n > len(raw_bc_count):
            logger.warning(helper.warning_msg(
                f"force_cells ({force_cell_n}) is larger than the number of unique barcodes found in the data "
                f"({len(raw_bc_count)})."))
            count_t = 0
        else:
            count_t = sorted(list(raw_bc_count.values()))[-force_cell_n]
    if count_t is not None:
        _start_knee_plot(list(raw_bc_count.values()), count_t, out_plot_fn, plot_threads)
        cells_bc = {k: v for k, v in raw_bc_count.items() if v >= count_t}
    elif exp_cells:
        t = percentile_count_thres(list(raw_bc_count.values()), exp_cells)
        _start_knee_plot(list(raw_bc_count.values()), t, out_plot_fn, plot_threads)
        cells_bc = {k: v for k, v in raw_bc_count.items() if v > t}
    else:
        raise ValueError('Invalid value of count_t and/or exp_cells.')
    if not output_empty:
        return cells_bc, []
    logger.info("Creating emtpy droplets barcode list...")
    ept_bc_max_count = min(min(cells_bc.values()), empty_max_count)
    ept_bc_candidate = [k for k, v in raw_bc_count.items() if v < ept_bc_max_count]
    return cells_bc, empty_drop_list(ept_bc_candidate, cells_bc.keys(), engine)
