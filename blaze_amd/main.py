"""The three BLAZE stages rebatched around the device: drop-in for the body of
/root/reference/blaze/main.py:332-416 `main`, taking the namespace the reference's own
`blaze.parser.parse_arg` returns (argument parsing / CLI stay the reference's: out of scope here).

    stage 1  main.py:344-371   FASTQ files -> putative_bc.csv + pass-code counts -> summary.txt
             host="native": bz_putative_bc_files (C++ reader / device search / pandas-exact CSV writer),
             host="python": read_batch_generator + putative_bc.get_raw_bc_from_reads + DataFrame.to_csv
    stage 2  main.py:376-404   high-quality barcode counts (pandas, as the reference) -> whitelisting.get_bc_whitelist
             (empty-drop test on the device) -> whitelist.csv, emtpy_bc_list.csv
    stage 3  main.py:410-416   read_assignment.assign_read -> demultiplexed FASTQ, two summary lines

Files written, their names and their bytes are the reference's (the .gz output up to the gzip
container, F9), the knee plot of stage 2 included (whitelisting.knee_plot).  Every stage needs the
GPU library: there is no CPU search / scoring path behind these functions.
"""
from __future__ import annotations

import gzip
import logging
import os
from collections import Counter, defaultdict

import pandas as pd

from . import helper
from . import putative_bc, read_assignment, whitelisting

logger = logging.getLogger(__name__)
logger.setLevel(logging.DEBUG)


def read_batch_generator(fastq_fns, batch_size):
    """Batches of reads, file by file (a batch never spans two files): main.py:304-323."""
    for fn in fastq_fns:
        opener = gzip.open(fn, "rt") if str(fn).endswith('.gz') else open(fn, "r")
        with opener as handle:
            yield from helper.batch_iterator(helper.fastq_parser(handle), batch_size=batch_size)


def add_summary(content, args, write_mode='w'):
    with open(args.summary_fn, write_mode) as f:
        f.write(content)
    if not args.minimal_stdout:
        print(content)


def bc_search_qc_report(pass_count, args):
    """Text of main.py:114-149 (the reference's dedent leaves its 8-space indentation in place)."""
    total = sum(pass_count.values())
    found = pass_count[0] + pass_count[100]
    pad = ' ' * 8

    def pct(x):
        return f"{x / total * 100:.2f}% of all reads"
    lines = [
        '', '',
        '----------------------stats of the putative barcodes--------------------------', '',
        pad + 'Total number of reads: ',
        pad + f'    {total:,}',
        pad + 'Reads with unambiguous polyT and adapter positions found:            ',
        pad + f'    {found:,} ({pct(found)})',
        pad + f'    {pass_count[0]:,} in which all bases in the putative BC have Q>={args.minQ}',
        pad + 'Failed Reads: ',
        pad + '    no polyT and adapter positions found: ',
        pad + f'        {pass_count[1]:,} ({pct(pass_count[1])})',
        pad + '    polyT and adapter positions found in both end (fail to determine strand): ',
        pad + f'        {pass_count[2]:,} ({pct(pass_count[2])})',
        pad + '    multiple polyT and adapter found in one end',
        pad + f'        {pass_count[10]:,} ({pct(pass_count[10])})',
        pad + '-------------------------------------------------------------------------------', '', '',
    ]
    return '\n'.join(lines)


def _engines(devices):
    return [read_assignment.get_engine(d) for d in read_assignment._device_ids(devices)]


def search_putative_bc(args, devices=None, host="native", comm=None):
    """Stage 1 over files: writes args.out_raw_bc_fn, returns {pass code: reads} (main.py:344-364).
    With a multi-rank `comm` the files are sharded over the ranks (dist.putative_bc_files_sharded)."""
    counts = defaultdict(int)
    if comm is not None and comm.world > 1:
        from . import dist
        counts.update(dist.putative_bc_files_sharded(comm, args.fastq_fns, args.out_raw_bc_fn, args.kit_version, args.umi_len,
                                                     args.minQ, args.batch_size, args.threads,     # the host's budget: split per rank there
                                                     device=read_assignment._device_ids(devices)[0]))
        return counts
    if host == "native":
        threads = max(1, min(int(args.threads or 1), os.cpu_count() or 1, 64))
        for k, v in putative_bc.putative_bc_files(_engines(devices), args.fastq_fns, args.out_raw_bc_fn, kit=args.kit_version,
                                                  umi_len=args.umi_len, min_q=args.minQ, batch_size=args.batch_size,
                                                  n_threads=threads).items():
            counts[k] += v
        return counts
    eng = _engines(devices)[0]
    for idx, batch in enumerate(read_batch_generator(args.fastq_fns, batch_size=args.batch_size)):
        count_pass, rst_df = putative_bc.get_raw_bc_from_reads(batch, min_q=args.minQ, kit=args.kit_version,
                                                               engine=eng, umi_len=args.umi_len)
        for k, v in count_pass.items():
            counts[k] += v
        if idx == 0:
            rst_df.to_csv(args.out_raw_bc_fn, index=False)
        else:
            rst_df.to_csv(args.out_raw_bc_fn, mode='a', index=False, header=False)
    return counts


def count_high_quality_bc(args):
    """main.py:378-382: counts of putative barcodes whose minimum base quality reaches minQ, in the
    reference's own insertion order (it decides the order of whitelist.csv and of the empty-drop candidates)."""
    raw_bc_count = Counter()
    for df in _count_chunks(args.out_raw_bc_fn):
        # the reference's `raw_bc_count += Counter(d)` (main.py:381-382): counts added, new keys appended in d's order, and
        # nothing to drop (value_counts gives counts >= 1) -- which is Counter.update(d), a C-level dict.update for the
        # first chunk instead of a Python loop over every barcode and a second pass that looks for non-positive counts
        raw_bc_count.update(df[df.putative_bc_min_q >= args.minQ].putative_bc.value_counts().to_dict())
    return raw_bc_count


_COUNT_CHUNK_ROWS = 1_000_000          # main.py:380 (the chunk boundaries shape the Counter's insertion order)
_COUNT_ARROW_MAX_BYTES = 1 << 30       # above this the table is parsed chunk by chunk by pandas' own reader


def _count_chunks(fn):
    """The reference's `pd.read_csv(fn, chunksize=1_000_000)` (main.py:380) restricted to the two columns the count
    reads.  Same frames for that purpose (same values in the same row order, so `value_counts` breaks its ties the same
    way), but the 5 unused string columns are never turned into Python objects: 2.4x faster with pandas' C parser, 9x
    when the table fits the multi-threaded Arrow parser (then the 1M-row chunks are slices of one frame)."""
    cols = ['putative_bc', 'putative_bc_min_q']
    try:
        small = os.path.getsize(fn) <= _COUNT_ARROW_MAX_BYTES
        if small:
            import pyarrow  # noqa: F401
    except (OSError, ImportError):
        small = False
    if small:
        try:
            df = pd.read_csv(fn, usecols=cols, engine='pyarrow')
        except Exception:                       # noqa: BLE001  (an empty or odd table: let pandas' own parser decide)
            df = None
        if df is not None:
            if df['putative_bc_min_q'].dtype == object:          # an all-blank column comes back as None objects; pandas' parser gives NaN
                df['putative_bc_min_q'] = pd.to_numeric(df['putative_bc_min_q'], errors='coerce')
            for lo in range(0, max(len(df), 1), _COUNT_CHUNK_ROWS):
                yield df.iloc[lo:lo + _COUNT_CHUNK_ROWS]
            return
    yield from pd.read_csv(fn, usecols=cols, chunksize=_COUNT_CHUNK_ROWS)


def main(args, devices=None, host="native", comm=None):
    """Run the stages `args.do_bc_search / do_whitelisting / do_demultiplexing` select.  `args` is the namespace
    of blaze.parser.parse_arg (fastq_fns, out_raw_bc_fn, out_whitelist_fn, out_emptydrop_fn, summary_fn, output_fastq,
    kit_version, umi_len, minQ, batch_size, threads, expect_cells / count_threshold / force_cells, ...).
    One process per GPU (`comm` from dist.make_comm(), world > 1): stages 1 and 3 shard the FASTQ files over the
    ranks, stage 2 and the summary are rank 0's; the files on disk are those of a single-process run."""
    if args is None or isinstance(args, str):
        raise TypeError("blaze_amd.main.main takes the namespace returned by blaze.parser.parse_arg(arg_string)")
    multi = comm is not None and comm.world > 1
    if multi and comm.rank != 0:
        return _main_worker_rank(args, devices, comm)
    plot_threads = []                                 # the knee plot is drawn while stage 3 runs
    try:
        _main_rank0(args, devices, host, comm, multi, plot_threads)
    finally:
        for t in plot_threads:
            t.join()


def _agree_to_go_on(comm, failed: bool) -> bool:
    """The collective between stage 2 (rank 0's) and stage 3 (everybody's): a sum of 'I failed' flags instead of a bare
    barrier, so that a rank that raised still reaches it and the others leave instead of waiting for ever."""
    import numpy as np
    return int(comm.allreduce_sum_i64(np.array([1 if failed else 0], dtype=np.int64))[0]) == 0


def _main_rank0(args, devices, host, comm, multi, plot_threads):
    _search(args, devices, host, comm)               # stage 1: with several ranks a collective of its own (sharded files)
    if not multi:
        _whitelist(args, devices, plot_threads)
        return _demultiplex(args, devices, host, comm, multi)
    try:
        _whitelist(args, devices, plot_threads)      # stage 2 and the summary are rank 0's alone
    except BaseException:
        _agree_to_go_on(comm, True)                  # the other ranks are waiting in this collective: tell them, then raise
        raise
    if not _agree_to_go_on(comm, False):             # whitelist.csv is on disk for the other ranks
        raise RuntimeError("another rank failed before read assignment")
    _demultiplex(args, devices, host, comm, multi)


def _search(args, devices, host, comm):
    if args.do_bc_search:
        logger.info(f'Getting putative barcodes from {len(args.fastq_fns)} FASTQ files...')
        raw_bc_pass_count = search_putative_bc(args, devices, host, comm)
        helper.green_msg(f'Putative barcode table saved in {args.out_raw_bc_fn}', printit=True)
        add_summary(bc_search_qc_report(raw_bc_pass_count, args), args=args, write_mode='w')


def _whitelist(args, devices, plot_threads):
    if args.do_whitelisting:
        raw_bc_count = count_high_quality_bc(args)
        try:
            bc_whitelist, ept_bc = whitelisting.get_bc_whitelist(raw_bc_count, args=args, engine=_engines(devices)[0],
                                                                  plot_threads=plot_threads)
            with open(args.out_whitelist_fn, 'w') as f:
                for k in bc_whitelist.keys():
                    f.write(k + '\n')
            with open(args.out_emptydrop_fn, 'w') as f:
                for k in ept_bc:
                    f.write(k + '\n')
                helper.green_msg(f'Empty droplet barcode list saved as `{args.out_emptydrop_fn}`.', printit=True)
            add_summary(f'\nIdentified # of cells: {len(bc_whitelist)}\n', args, write_mode='a')
        except Exception as e:            # noqa: BLE001  (the reference logs and carries on, main.py:398-402)
            logger.exception(e)
            helper.err_msg("Error: Failed to get whitelist. Please check the input files and settings.", printit=True)
        else:
            helper.green_msg(f'Whitelist saved as `{args.out_whitelist_fn}`!', printit=True)


def _demultiplex(args, devices, host, comm, multi):
    if args.do_demultiplexing:
        logger.info("Assigning reads to whitelist.\n")
        if multi:
            demul_count_tot, count_tot = _assign_sharded(args, devices, comm)
        else:
            demul_count_tot, count_tot = read_assignment.assign_read(args=args, devices=devices, host=host)
        add_summary(f'\nTotal reads: {count_tot}'
                    f'\nTotal reads in cells: {demul_count_tot} ({demul_count_tot / count_tot * 100:.2f}%)', args, write_mode='a')


def _assign_sharded(args, devices, comm):
    from . import dist
    return dist.assign_read_sharded(comm, args.fastq_fns, args.output_fastq, args.out_raw_bc_fn, args.out_whitelist_fn,
                                    args.max_edit_distance, args.threads, args.batch_size, restrand=args.restrand,
                                    device=read_assignment._device_ids(devices)[0])


def _main_worker_rank(args, devices, comm):
    """Ranks > 0 of a multi-GPU run: their share of stages 1 and 3, in step with rank 0's collectives."""
    if args.do_bc_search:
        search_putative_bc(args, devices, "native", comm)
    if not _agree_to_go_on(comm, False):              # rank 0 has written whitelist.csv (or skipped whitelisting) -- or failed
        raise SystemExit("blaze_amd: rank 0 failed before read assignment (its message is above); nothing assigned")
    if args.do_demultiplexing:
        _assign_sharded(args, devices, comm)
