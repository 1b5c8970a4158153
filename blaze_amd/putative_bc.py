"""Stage 1 of BLAZE ("get putative barcodes") on MI355X: drop-in for
/root/reference/blaze/main.py:43-106 `get_raw_bc_from_reads`, same arguments and return values
(Counter of adaptor_polyT_pass codes, DataFrame with the seven putative_bc.csv columns).

What changed underneath: the per-read `polyT_adaptor_finder.Read` object (poly-T / TSO search, adaptor
sub_edit_distance in the search windows, strand and ambiguity rules, barcode quality, poly-T end:
/root/reference/blaze/polyT_adaptor_finder.py:43-379) is one device launch per batch of reads
(bz_putative_bc, blaze_amd/csrc/stage1_kernels.h) that returns five numbers per read; the string fields are slices
of the read at those numbers, taken here with Python's own slice semantics (the reference's negative-start
flanks, :321-323, included).  No CPU search path: without libblaze_hip.so and a GPU this raises.
"""
from __future__ import annotations

from collections import Counter

import numpy as np
import pandas as pd

from . import _lib
from .helper import reverse_complement

DEFAULT_BC_SIZE = 16               # config.py:34
DEFAULT_GRB_FLANKING_SIZE = 5      # config.py:46
KITS_3P = ('3v4', '3v3', '3v2')    # polyT_adaptor_finder.py:190


def _rc_slice(seq, a, b):
    """reverse_complement(seq)[a:b] without reversing the whole read (same result for every a, b Python accepts)."""
    n = len(seq)
    s, e, _ = slice(a, b).indices(n)
    return reverse_complement(seq[n - e:n - s]) if e > s else ''


def search(engine, seqs, quals, kit, umi_len):
    """Device search of a batch: (pass_code, strand, raw_bc_start, min_q, polyT_end) arrays, one entry per read."""
    n = len(seqs)
    enc = [s.encode('latin-1', 'replace') for s in seqs]
    qenc = [q.encode('latin-1', 'replace') for q in quals]
    if any(len(a) != len(b) for a, b in zip(enc, qenc)):
        raise ValueError("sequence and quality strings differ in length")
    off = np.zeros(n + 1, dtype=np.int64)
    if n:
        np.cumsum([len(s) for s in enc], out=off[1:])
    sb = np.frombuffer(b''.join(enc) or b'\0', dtype=np.uint8)
    qb = np.frombuffer(b''.join(qenc) or b'\0', dtype=np.uint8)
    code = np.zeros(n, np.int32); strand = np.zeros(n, np.int8); start = np.zeros(n, np.int32)
    minq = np.zeros(n, np.int32); polyt = np.zeros(n, np.int32)
    _lib.check(_lib.load().bz_putative_bc(engine._h, sb.ctypes.data, qb.ctypes.data, off.ctypes.data, n,
                                          0 if kit in KITS_3P else 1, int(umi_len), code.ctypes.data, strand.ctypes.data,
                                          start.ctypes.data, minq.ctypes.data, polyt.ctypes.data))
    return code, strand, start, minq, polyt


def get_raw_bc_from_reads(reads, min_q=0, kit=None, engine=None, **kwargs):
    """reads: objects with .id, .seq, .q_letter (helper.fastq_parser's tuples).  kwargs carries umi_len, as main.py:352."""
    if engine is None:
        from .read_assignment import get_engine
        engine = get_engine()
    umi_len = kwargs['umi_len']                     # AttributeError on self.umi_len in the reference when it is missing
    seqs = [str(r.seq) for r in reads]
    code, strand, start, minq, polyt = search(engine, seqs, [r.q_letter for r in reads], kit, umi_len)
    read_ids, bcs, min_qs, umis, trims, pres, posts, raw_bc_pass = [], [], [], [], [], [], [], []
    for i, r in enumerate(reads):
        read_ids.append(r.id)
        c = int(code[i])
        if c < 0:
            raise ValueError("min() arg is an empty sequence")      # polyT_adaptor_finder.py:278 on this read
        if c != 0:
            bcs.append(None); min_qs.append(None); umis.append(None); trims.append(None); pres.append(None); posts.append(None)
            raw_bc_pass.append(c)
            continue
        s, seq = int(start[i]), seqs[i]
        cut = (lambda a, b: _rc_slice(seq, a, b)) if strand[i] == ord('+') else (lambda a, b: seq[a:b])
        mq = int(minq[i])
        bcs.append(cut(s, s + DEFAULT_BC_SIZE))
        min_qs.append(mq)
        umis.append(cut(s + DEFAULT_BC_SIZE, s + DEFAULT_BC_SIZE + umi_len))
        trims.append(int(polyt[i]))
        pres.append(cut(s - DEFAULT_GRB_FLANKING_SIZE, s))
        posts.append(cut(s + DEFAULT_BC_SIZE + umi_len, s + DEFAULT_BC_SIZE + umi_len + DEFAULT_GRB_FLANKING_SIZE))
        raw_bc_pass.append(100 if (mq and mq < min_q) else c)      # :89-92 (a minimum quality of 0 is falsy there)
    rst_df = pd.DataFrame({'read_id': read_ids, 'putative_bc': bcs, 'putative_bc_min_q': min_qs, 'putative_umi': umis,
                           'polyT_end': trims, 'pre_bc_flanking': pres, 'post_umi_flanking': posts})
    return Counter(raw_bc_pass), rst_df


STAGE1_DEVICE_BATCH_READS = 1 << 15       # reads per device submit of the native file pipeline


def putative_bc_files(engines, fastq_fns, out_csv, kit, umi_len, min_q=15, batch_size=1000, n_threads=8,
                      device_batch_reads=STAGE1_DEVICE_BATCH_READS):
    """Stage 1 over files in C++ (bz_putative_bc_files, blaze_amd/csrc/host_io.h): the loop of main.py:344-364.
    Writes `out_csv` with the bytes the reference's per-batch `DataFrame.to_csv` calls produce and returns
    {pass code: reads} (raw_bc_pass_count, main.py:355-360; only codes that occurred)."""
    import ctypes
    import os
    L = _lib.load()
    handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    fns = (ctypes.c_char_p * len(fastq_fns))(*[os.fsencode(str(f)) for f in fastq_fns])
    counts = (ctypes.c_int64 * 8)()
    rc = L.bz_putative_bc_files(handles, len(engines), fns, len(fastq_fns), os.fsencode(str(out_csv)),
                                0 if kit in KITS_3P else 1, int(umi_len), int(min_q), int(batch_size), int(device_batch_reads),
                                int(n_threads), counts)
    if rc < 0:
        msg = (L.bz_last_error() or b'').decode(errors='replace')
        if msg.startswith("min() arg is an empty sequence"):        # what the reference raises on such a read (:278)
            raise ValueError(msg)
    _lib.check(rc)
    return {code: int(counts[1 + k]) for k, code in enumerate((0, 1, 2, 10, 100)) if counts[1 + k]}
