"""Step 3 of BLAZE ("assign reads to the barcodes") on MI355X: drop-in for
/root/reference/blaze/read_assignment.py with the same entry points and file contract

    assign_read(...)                               reference :253-330
    _assign_read_batches(r_batch, whitelist, ...)  reference :184-249
    _match_bc_row(row, whitelist, max_ed, minQ)    reference :35-99
    _read_and_bc_batch_generator_with_idx(...)     reference :138-181
    read_fastq                                     reference :27-33

What changed underneath: the per-read Python loop over the whitelist with one
third-party `sub_edit_distance` call per barcode (:71-72) is one HIP kernel launch
per device batch (blaze_amd/csrc/scan_kernels.h) behind the C ABI of
include/blaze_hip.h; the process pool (:309-317, helper.py:111-205) becomes
double-buffered device batches that are consumed in submit order.  There is no CPU
scoring path here: without libblaze_hip.so and a GPU these functions raise.
"""
from __future__ import annotations

import gzip
import logging
import os
import sys
from collections import deque
from io import StringIO

import numpy as np
import pandas as pd
from tqdm import tqdm

from . import _lib, helper
from .config import DEFAULT_ED_FLANKING, DEVICE_BATCH_ROWS
from .engine import Engine, TIE_LEFTMOST, TIE_RIGHTMOST

LOG_FORMAT = '(%(asctime)s) %(message)s'
DATE_FORMATE = '%d/%m/%Y %H:%M:%S'
logging.basicConfig(format=LOG_FORMAT, datefmt=DATE_FORMATE)
logger = logging.getLogger(__name__)
logger.setLevel(logging.DEBUG)

_EXT_STRIDE = 32
# The only unpinned policy of the replaced FFI (end index on ties): env override for maintainers
# who find the third-party library behaves the other way.
TIE_RULE = TIE_RIGHTMOST if os.environ.get("BLAZE_AMD_TIE_RULE", "leftmost").lower().startswith("r") else TIE_LEFTMOST


class read_fastq:
    """Minimal FASTQ record (id = first token of the title without '@'), as the reference's."""

    def __init__(self, title, sequence, qscore, quality_map=False):
        self.id = title.split()[0].strip("@")
        self.seq = sequence
        self.qscore = qscore


# ------------------------------------------------------------------------------ engines
_engines = {}
_uploaded = {}        # device -> ordered whitelist currently resident on that GPU


def get_engine(device: int = 0) -> Engine:
    eng = _engines.get(device)
    if eng is None:
        eng = Engine(device, TIE_RULE)
        _engines[device] = eng
    return eng


def close_engines():
    for eng in _engines.values():
        eng.close()
    _engines.clear()
    _uploaded.clear()


def _ensure_whitelist(eng: Engine, ordered):
    """Upload `ordered` (a list, order is semantic) unless that list is already resident: the same object (the usual case, one
    pointer compare), else a list of the same length whose elements compare equal (a per-row caller that rebuilds its list)."""
    cur = _uploaded.get(eng.device)
    if cur is not None and (cur is ordered or (len(cur) == len(ordered) and cur == ordered)):
        return
    eng.set_whitelist(ordered)
    _uploaded[eng.device] = ordered


_ordered_of = {}      # id(whitelist container) -> (len, first element, the list made of it): one list per set, not one per row


def _ordered_list(whitelist):
    """The iteration order of `whitelist` as a list.  The reference hands `_match_bc_row` the SAME set for every row of a
    batch (read_assignment.py:206-211): the list made of it is kept (keyed on the container's identity, its length and its
    first element) instead of being rebuilt -- and compared with the resident one, 6.8M strings for the full 10x list -- per row."""
    if isinstance(whitelist, list):
        return whitelist
    key = id(whitelist)
    hit = _ordered_of.get(key)
    n = len(whitelist)
    first = next(iter(whitelist)) if n else None
    if hit is not None and hit[3] is whitelist and hit[0] == n and hit[1] == first:
        return hit[2]
    ordered = list(whitelist)
    _ordered_of.clear()                                  # one container at a time: nothing accumulates
    _ordered_of[key] = (n, first, ordered, whitelist)     # (holding the container keeps its id from being reused)
    return ordered


# ------------------------------------------------------------------------------ row packing
def _strand(polyT_end):
    if not polyT_end:
        return ''
    return '-' if polyT_end > 0 else '+'


def _pack_rows(bcs, umis, pres):
    """ext = pre_bc_flanking[-2:] + putative_bc + putative_umi[:2] (reference :70) -> C-ABI arrays."""
    n = len(bcs)
    exts, bc_off, bc_len = [], np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    stride = _EXT_STRIDE
    for i in range(n):
        bc = bcs[i]
        if not bc:
            exts.append(b'')
            continue
        pre = pres[i][-DEFAULT_ED_FLANKING:]
        e = (pre + bc + umis[i][:DEFAULT_ED_FLANKING]).encode('latin-1', 'replace')
        if len(e) > 64 or len(bc) > 64:
            raise ValueError(f"putative_bc row {i}: extension of {len(e)} nt exceeds the device limit of 64")
        stride = max(stride, len(e))
        exts.append(e)
        bc_off[i] = len(pre)
        bc_len[i] = len(bc)
    ext_len = np.fromiter((len(e) for e in exts), dtype=np.uint8, count=n)
    ext = np.frombuffer(b''.join(e.ljust(stride, b'\0') for e in exts), dtype=np.uint8).reshape(n, stride) \
        if n else np.zeros((0, stride), np.uint8)
    return ext, ext_len, bc_off, bc_len


def _columns(df):
    return (df['putative_bc'].tolist(), df['putative_umi'].tolist(), df['pre_bc_flanking'].tolist(),
            df['post_umi_flanking'].tolist(), df['polyT_end'].tolist())


def _decode_rows(hit, end, flags, ext_len, bcs, umis, posts, polyT, ordered):
    """Device results -> [BC_corrected, putative_umi, strand] per row (reference :61-62, :87-99)."""
    out = []
    for i in range(len(bcs)):
        strand = _strand(polyT[i])
        bc, umi = bcs[i], umis[i]
        h = hit[i]
        if not bc:                                # blank row: returned untouched
            out.append([bc, umi, strand])
        elif h < 0:                               # no hit / ambiguous
            out.append(['', umi, strand])
        elif flags[i] & 1:                        # exact-match shortcut: nothing adjusted
            out.append([bc, umi, strand])
        else:
            adj = int(end[i]) - (int(ext_len[i]) - 1 - DEFAULT_ED_FLANKING)
            if adj > 0:
                umi = umi[adj:] + posts[i][:adj]
            elif adj < 0:
                umi = bc[adj:] + umi[:adj]
            out.append([ordered[h], umi, strand])
    return out


def _format_batch(read_batch, df, restrand):
    """FASTQ text of the assigned reads of one batch (reference :218-242)."""
    parts = []
    for r, bc in zip(read_batch, df.itertuples()):
        if bc.read_id != r.id:
            helper.err_msg("Different order in putative bc file and input fastq!", printit=True)
            sys.exit()
        if not bc.BC_corrected:
            continue
        cut = int(bc.polyT_end)
        if bc.polyT_end < 0:
            seq, qscore = r.seq[:cut], r.qscore[:cut]
        else:
            seq, qscore = r.seq[cut:], r.qscore[cut:]
        if restrand and bc.strand == '-':
            seq = helper.reverse_complement(seq)
            qscore = qscore[::-1]
        parts.append(f"@{bc.BC_corrected}_{bc.putative_umi}#{bc.read_id}_{bc.strand}"
                     f"\tCB:Z:{bc.BC_corrected}\tUB:Z:{bc.putative_umi}\n{seq}\n+\n{qscore}\n")
    return ''.join(parts)


def _finish_batch(read_batch, df, new_cols, gz, restrand):
    df[['BC_corrected', 'putative_umi', 'strand']] = new_cols
    demul_read_count = sum(df.BC_corrected != '')
    text = _format_batch(read_batch, df, restrand)
    data = text.encode('utf-8')
    if gz:
        data = gzip.compress(data)
    return df, data, demul_read_count, len(read_batch)


# ------------------------------------------------------------------------------ reference entry points
def _match_bc_row(row, whitelist, max_ed, minQ, _engine=None):
    """One row against the whitelist, iterated in the order `whitelist` yields (a set in the
    reference, :206).  Returns [assigned barcode or '', adjusted putative_umi, strand]."""
    strand = _strand(row.polyT_end)
    if minQ and row.putative_bc_qscore < minQ:
        return ['', '', '']
    if not row.putative_bc or row.putative_bc in whitelist:
        return [row.putative_bc, row.putative_umi, strand]
    eng = _engine or get_engine()
    ordered = _ordered_list(whitelist)
    _ensure_whitelist(eng, ordered)
    ext, ext_len, bc_off, bc_len = _pack_rows([row.putative_bc], [row.putative_umi], [row.pre_bc_flanking])
    hit, _, end, flags = eng.assign(ext, ext_len, bc_off, bc_len, max_ed)
    return _decode_rows(hit, end, flags, ext_len, [row.putative_bc], [row.putative_umi], [row.post_umi_flanking],
                        [row.polyT_end], ordered)[0]


def _assign_ordered(r_batch, ordered, max_ed, gz, restrand, engine=None):
    """_assign_read_batches on an explicit iteration order (tests pin the order with this)."""
    read_batch, start_df_idx, df = r_batch
    df = df.fillna('')
    eng = engine or get_engine()
    _ensure_whitelist(eng, ordered)
    bcs, umis, pres, posts, polyT = _columns(df)
    ext, ext_len, bc_off, bc_len = _pack_rows(bcs, umis, pres)
    hit, _, end, flags = eng.assign(ext, ext_len, bc_off, bc_len, max_ed)
    new_cols = _decode_rows(hit, end, flags, ext_len, bcs, umis, posts, polyT, ordered)
    return _finish_batch(read_batch, df, new_cols, gz, restrand)


def _assign_read_batches(r_batch, whitelist, max_ed, gz, restrand, minQ=0):
    """Assign one (read_batch, start_idx, batch_bc_df) batch.

    Returns (df + BC_corrected/putative_umi/strand, FASTQ bytes, n assigned, n reads) like the
    reference (:249).  The whitelist is de-duplicated and iterated as `set(whitelist)` (:206);
    minQ is accepted and unused, as in the reference (assign_read never forwards it)."""
    return _assign_ordered(r_batch, list(set(whitelist)), max_ed, gz, restrand)


def _read_and_bc_batch_generator_with_idx(fastq_fns, putative_bc_csv, batch_size):
    """Lock-step batches (reads, index of the first read, putative-bc frame of the same rows)."""
    read_idx = 0
    with open(putative_bc_csv, 'r') as putative_bc_f:
        header = next(putative_bc_f)
        for fn in fastq_fns:
            opener = gzip.open(fn, "rt") if str(fn).endswith('.gz') else open(fn)
            with opener as handle:
                fastq = (read_fastq(title, sequence, qscore) for title, sequence, qscore in helper.fastq_parser(handle))
                for batch in helper.batch_iterator(fastq, batch_size=batch_size):
                    n = len(batch)
                    frame = pd.read_csv(StringIO(header + ''.join([next(putative_bc_f) for _ in range(n)])))
                    yield batch, read_idx, frame
                    read_idx += n


def _device_ids(devices):
    if devices is None:
        env = os.environ.get("BLAZE_AMD_DEVICES")
        if env:
            devices = [int(x) for x in env.split(',') if x.strip() != '']
        elif "LOCAL_RANK" in os.environ:
            devices = [int(os.environ["LOCAL_RANK"])]
        else:
            devices = [0]
    return list(devices)


class OrderMismatch(Exception):
    """The table's read ids and the FASTQ reads disagree (raised instead of the reference's message + sys.exit() only where
    the caller has another way to find the rows: dist.assign_read_sharded's located-row route)."""


def _assign_files_native(engines, fastq_fns, putative_bc_csv, fastq_out, gz, restrand, max_ed, batchsize,
                         device_batch_rows, n_threads, csv_skip_rows=0, csv_byte_offset=None, on_mismatch="exit"):
    """bz_assign_files(_from / _at): the C++ host pipeline (blaze_amd/csrc/host_io.h) around the same device scan.
    `csv_skip_rows`: rows of the table that belong to FASTQ files in front of `fastq_fns` (another rank's), or
    `csv_byte_offset`: where this rank's first row starts in the table (dist.locate_first_row)."""
    import ctypes
    from . import _lib
    L = _lib.load()
    handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    fns = (ctypes.c_char_p * len(fastq_fns))(*[os.fsencode(str(f)) for f in fastq_fns])
    counts = (ctypes.c_int64 * 4)()
    entry, where = (L.bz_assign_files_from, csv_skip_rows) if csv_byte_offset is None else (L.bz_assign_files_at, csv_byte_offset)
    rc = entry(handles, len(engines), fns, len(fastq_fns), os.fsencode(str(putative_bc_csv)),
               int(where), os.fsencode(str(fastq_out)), int(bool(gz)), int(bool(restrand)), int(max_ed),
               int(batchsize), int(device_batch_rows), int(n_threads), counts)
    if rc == _lib.BZ_E_MISMATCH and on_mismatch == "raise":
        raise OrderMismatch(str(fastq_fns[:1]))
    if rc == _lib.BZ_E_MISMATCH:                    # the reference's reaction, read_assignment.py:219-223
        helper.err_msg("Different order in putative bc file and input fastq!", printit=True)
        sys.exit()
    _lib.check(rc)
    return int(counts[0]), int(counts[1])


def assign_read(fastq_fns=None, fastq_out=None, putative_bc_csv=None,
                whitelsit_csv=None, max_ed=None, n_process=None,
                batchsize=None, minQ=0, restrand=True, args=None, devices=None,
                device_batch_rows=DEVICE_BATCH_ROWS, host="native", gz_level=None):
    """Demultiplex fastq files using the putative barcode csv and the whitelist csv.

    Same arguments as the reference (note its `whitelsit_csv` spelling) or an `args`
    namespace with fastq_fns / output_fastq / out_raw_bc_fn / out_whitelist_fn /
    max_edit_distance / threads / batch_size / restrand.  `n_process` is accepted for
    compatibility (it sizes the host thread pool of the native pipeline); device parallelism comes
    from the GPUs in `devices` (default: $BLAZE_AMD_DEVICES, else $LOCAL_RANK, else GPU 0).
    host="native" (default) runs the batch loop in C++ (bz_assign_files); host="python" runs the
    pandas mirror of the reference's loop below.  Both give the same bytes (of a `.gz` output: the same DECOMPRESSED
    bytes; the native host writes its members with a FASTQ-aware deflate encoder, `gz_level` 1..9 selects zlib at that
    level instead -- the reference's level 9 makes files 1.8 % smaller at ~80x the CPU time, see INTEGRATION.md section 5;
    $BLAZE_AMD_GZ_LEVEL overrides both).
    Returns (reads assigned, reads seen)."""
    if args:
        fastq_fns = args.fastq_fns
        fastq_out = args.output_fastq
        putative_bc_csv = args.out_raw_bc_fn
        whitelsit_csv = args.out_whitelist_fn
        max_ed = args.max_edit_distance
        n_process = args.threads
        batchsize = args.batch_size
        restrand = args.restrand

    gz = fastq_out.endswith('.gz')
    r_batches = _read_and_bc_batch_generator_with_idx(fastq_fns, putative_bc_csv, batchsize)

    whitelist = []
    with open(whitelsit_csv, 'r') as f:
        for line in f:
            whitelist.append(line.split('-')[0].strip())
    ordered = list(set(whitelist))                 # the order the reference's workers iterate (:206)

    engines = [get_engine(d) for d in _device_ids(devices)]
    for eng in engines:
        _ensure_whitelist(eng, ordered)
    if host == "native":
        threads = max(1, min(int(n_process or 1), os.cpu_count() or 1, 256))
        engines[0].set_option(_lib.BZ_OPT_GZ_LEVEL, int(gz_level) if gz_level else 0)
        out = _assign_files_native(engines, fastq_fns, putative_bc_csv, fastq_out, gz, restrand, max_ed, batchsize,
                                   device_batch_rows, threads)
        logger.info(helper.green_msg(f"Reads assignment completed. Demultiplexed read saved in {fastq_out}!", printit=False))
        return out
    n_slots = 2 * len(engines)
    pending = deque()
    demul_count_tot = 0
    count_tot = 0

    def drain(output_handle, pbar):
        nonlocal demul_count_tot, count_tot
        eng, slot, group, ext_len = pending.popleft()
        hit, _, end, flags = eng.wait(slot)
        lo = 0
        for read_batch, df, cols in group:
            n = len(df)
            bcs, umis, posts, polyT = cols
            new_cols = _decode_rows(hit[lo:lo + n], end[lo:lo + n], flags[lo:lo + n], ext_len[lo:lo + n],
                                    bcs, umis, posts, polyT, ordered)
            lo += n
            _, data, demul, total = _finish_batch(read_batch, df, new_cols, gz, restrand)
            demul_count_tot += demul
            count_tot += total
            output_handle.write(data)
            pbar.update(total)

    def submit(k, group_raw):
        eng = engines[k % len(engines)]
        slot = (k // len(engines)) % 2
        group, all_bc, all_umi, all_pre = [], [], [], []
        for read_batch, _, df in group_raw:
            df = df.fillna('')
            bcs, umis, pres, posts, polyT = _columns(df)
            all_bc += bcs
            all_umi += umis
            all_pre += pres
            group.append((read_batch, df, (bcs, umis, posts, polyT)))
        ext, ext_len, bc_off, bc_len = _pack_rows(all_bc, all_umi, all_pre)
        eng.submit(slot, ext, ext_len, bc_off, bc_len, max_ed)
        pending.append((eng, slot, group, ext_len))

    with open(fastq_out, 'wb') as output_handle:
        pbar = tqdm(unit="Reads", desc='Processed')
        k, group_raw, rows = 0, [], 0
        for r_batch in r_batches:
            group_raw.append(r_batch)
            rows += len(r_batch[0])
            if rows >= device_batch_rows:
                if len(pending) == n_slots:
                    drain(output_handle, pbar)
                submit(k, group_raw)
                k, group_raw, rows = k + 1, [], 0
        if group_raw:
            if len(pending) == n_slots:
                drain(output_handle, pbar)
            submit(k, group_raw)
        while pending:
            drain(output_handle, pbar)
        pbar.close()

    logger.info(helper.green_msg(f"Reads assignment completed. Demultiplexed read saved in {fastq_out}!", printit=False))
    return demul_count_tot, count_tot
