"""GPU parity of the native host pipeline (bz_assign_files, blaze_amd/csrc/host_io.h) = SURVEY 8f rows N1 + N2.

Two anchors, both bit-exact on the decompressed FASTQ bytes and the two counts:
  * the golden end-to-end case recorded from the REAL reference (tests/golden/e2e_leftmost.json);
  * the pandas host (`assign_read(host="python")`), which mirrors the reference's batch loop line by line and
    is itself pinned by the goldens, on files that stress what the native parser had to restate: pandas' NA
    tokens, float-typed polyT_end ("123.0", the form DataFrame.to_csv gives a column with blanks), quoted cells,
    CRLF line ends, several input files (one gzipped), ragged last batches, title lines with extra tokens.
"""
import gzip
import json
import os

import numpy as np
import pytest

from blaze_amd import _lib, synth
from blaze_amd import read_assignment as ra

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
HEADER = "read_id,putative_bc,putative_bc_min_q,putative_umi,polyT_end,pre_bc_flanking,post_umi_flanking\n"


def _load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def _text(path):
    raw = open(path, "rb").read()
    if str(path).endswith(".gz"):
        return gzip.decompress(raw)      # multi-member: gzip.decompress concatenates the members
    return raw


def _gz_members(raw):
    """Number of gzip members in a byte string."""
    import zlib
    n = 0
    while raw:
        d = zlib.decompressobj(31)
        d.decompress(raw)
        raw = d.unused_data
        n += 1
    return n


@pytest.mark.parametrize("threads", [1, 8])
@pytest.mark.parametrize("device_batch_rows", [1 << 18, 100])
def test_native_end_to_end_golden(tmp_path, device_batch_rows, threads):
    case = _load("e2e_leftmost.json")["case"]
    pa, pb = tmp_path / "a.fastq", tmp_path / "b.fastq.gz"
    pa.write_text(case["fastq_a"])
    with gzip.open(pb, "wt") as f:
        f.write(case["fastq_b_gz_text"])
    pc, pw = tmp_path / "putative_bc.csv", tmp_path / "whitelist.csv"
    pc.write_text(case["csv"])
    pw.write_text(case["whitelist_text"])
    for out_name, exp in case["outputs"].items():
        po = tmp_path / out_name
        demul, total = ra.assign_read(fastq_fns=[str(pa), str(pb)], fastq_out=str(po), putative_bc_csv=str(pc),
                                      whitelsit_csv=str(pw), max_ed=case["max_ed"], n_process=threads,
                                      batchsize=case["batchsize"], restrand=exp["restrand"],
                                      device_batch_rows=device_batch_rows, host="native")
        assert _text(po).decode() == exp["fastq"]
        assert (int(demul), int(total)) == (exp["demul"], exp["total"])


def _make_files(tmp_path, n, n_cells, seed, *, crlf=False, float_polyT=True, quirks=True, n_files=3, err=2.0, umi_len=12, wide=False, odd_wl=False):
    """Synthetic step-3 inputs: FASTQ files (second one gzipped), putative_bc.csv, whitelist."""
    rng = np.random.default_rng(seed)
    full = synth.full_whitelist_codes(20_000, seed=seed + 1)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=seed + 2)
    rows = synth.make_rows(n, cells, full, sub=0.01 * err, ins=0.005 * err, dele=0.005 * err, seed=seed + 3, umi_len=umi_len)
    ids = synth.read_ids(n, seed=seed + 4)
    recs = synth.fastq_records(n, ids, seed=seed + 5, median_len=300, min_len=60)
    if quirks:      # restrand must reverse real qualities and complement ACGT/acgt only (helper.py:14-25): N, IUPAC, '-', lower case
        odd = np.frombuffer(b"NnacgtRYKM-.*", dtype=np.uint8)
        for j in range(n):
            rid, seq, q = recs[j]
            q = bytes(rng.integers(33, 75, size=len(q), dtype=np.uint8)).decode("ascii")
            if j % 3 == 0:
                b = bytearray(seq.encode())
                at = rng.integers(0, len(b), size=max(1, len(b) // 9))
                b_arr = np.frombuffer(bytes(b), dtype=np.uint8).copy()
                b_arr[at] = odd[rng.integers(0, odd.size, size=at.size)]
                seq = b_arr.tobytes().decode("ascii")
            recs[j] = (rid, seq, q)
    s = lambda m, i: m[i].tobytes().decode("ascii")
    wl = ["".join(map(chr, w)) for w in synth.codes_to_bytes(cells)]
    odd = []
    if odd_wl:      # lines a hand-made --known-bc-list may hold (read_assignment.py:286-289 takes any): short, lower case, blank, repeated, long, N
        odd = [wl[0][:15], wl[1 % len(wl)].lower(), "", wl[2 % len(wl)], wl[3 % len(wl)] + "ACGTACGTA", wl[4 % len(wl)][:6] + "N" + wl[4 % len(wl)][7:]]
        for e in odd:
            wl.insert(int(rng.integers(0, len(wl) + 1)), e)
    lines = [HEADER]
    for i in range(n):
        if rows["blank"][i]:
            lines.append(f"{ids[i]},,,,,,\n")
            continue
        bc, umi, pre, post = s(rows["bc"], i), s(rows["umi"], i), s(rows["pre"], i), s(rows["post"], i)
        pe = int(rows["polyT_end"][i])
        if abs(pe) > len(recs[i][1]) and i % 3:
            pe = int(np.sign(pe)) * int(rng.integers(1, len(recs[i][1])))
        pe_s = f"{pe}.0" if float_polyT else str(pe)
        q = int(rows["min_q"][i])
        if quirks:
            k = int(rng.integers(0, 40))
            if k == 0:
                pre = "NA"                      # a 2-nt flank that pandas reads as NaN -> ''
            elif k == 1:
                post = "NA"
            elif k == 2:
                pre = ""                        # BC at the very start of the read (polyT_adaptor_finder.py:321)
            elif k == 3:
                post = post[:2]
            elif k == 4:
                bc = f'"{bc}"'                  # quoted cell
            elif k == 5:
                pre = pre[:1]
            elif k == 6:
                bc = bc[:7] + "N" + bc[8:]
            elif k == 7:
                pe_s = "0"                      # falsy polyT_end: strand '' and nothing trimmed
            elif k == 8:
                umi = umi[:1]                   # truncated UMI at the end of a read
            elif k == 9:
                pe_s = f"{pe}.7"                # int() truncates toward zero
            elif k == 10:
                post = "null"
            elif k == 12 and odd:               # the barcode cell IS one of the odd lines (the exact-match shortcut on it), or one edit away
                bc = odd[int(rng.integers(0, len(odd)))] or bc
                if rng.integers(0, 2) and len(bc) > 3:
                    bc = bc[:2] + "ACGT"[int(rng.integers(0, 4))] + bc[3:]
            elif k == 11 and wide:              # a hand-made barcode cell wider than the 32-byte row (up to the device's 64)
                m = int(rng.integers(17, 61))
                bc = (bc + "".join("ACGT"[c] for c in rng.integers(0, 4, size=m)))[:m] if rng.integers(0, 2) else \
                    ("".join("ACGT"[c] for c in rng.integers(0, 4, size=m - 16)) + bc)[-m:]
        lines.append(f"{ids[i]},{bc},{q},{umi},{pe_s},{pre},{post}\n")
    nl = "\r\n" if crlf else "\n"
    csv = "".join(lines).replace("\n", nl)
    pc = tmp_path / "putative_bc.csv"
    pc.write_bytes(csv.encode())
    cuts = sorted(rng.choice(np.arange(1, n), size=n_files - 1, replace=False).tolist()) if n_files > 1 else []
    bounds = [0] + cuts + [n]
    fns = []
    for k in range(n_files):
        body = "".join(f"@{rid} runid=abc ch={j}\n{seq}\n+{rid if j % 5 == 0 else ''}\n{q}\n"
                       for j, (rid, seq, q) in enumerate(recs[bounds[k]:bounds[k + 1]])).replace("\n", nl)
        if k == 1:
            fn = tmp_path / f"reads_{k}.fastq.gz"
            with gzip.open(fn, "wb") as f:
                f.write(body.encode())
        else:
            fn = tmp_path / f"reads_{k}.fastq"
            fn.write_bytes(body.encode())
        fns.append(str(fn))
    pw = tmp_path / "whitelist.csv"
    pw.write_text("".join(f"{w}-1\n" for w in wl))
    return fns, str(pc), str(pw), bounds


@pytest.mark.parametrize("crlf,gz,restrand,batchsize,device_batch_rows,threads", [
    (False, False, True, 1000, 1 << 16, 8),
    (False, True, True, 333, 700, 3),
    (True, False, False, 64, 100, 8),
    (False, True, False, 5000, 1 << 16, 1),
])
def test_native_equals_pandas_host(tmp_path, crlf, gz, restrand, batchsize, device_batch_rows, threads):
    n = 6000
    fns, pc, pw, bounds = _make_files(tmp_path, n, 500, seed=71 + batchsize, crlf=crlf)
    suffix = ".fastq.gz" if gz else ".fastq"
    out_n, out_p = str(tmp_path / ("native" + suffix)), str(tmp_path / ("pandas" + suffix))
    kw = dict(fastq_fns=fns, putative_bc_csv=pc, whitelsit_csv=pw, max_ed=2, batchsize=batchsize, restrand=restrand,
              device_batch_rows=device_batch_rows)
    got_p = ra.assign_read(fastq_out=out_p, n_process=1, host="python", **kw)
    got_n = ra.assign_read(fastq_out=out_n, n_process=threads, host="native", **kw)
    assert tuple(map(int, got_n)) == tuple(map(int, got_p))
    assert got_n[1] == n and 0.3 * n < got_n[0] < n
    a, b = _text(out_n), _text(out_p)
    assert a == b
    if gz:      # member boundaries are the reference's: one per --batch-size reads per input file (:244-245)
        expect = sum(-(-(hi - lo) // batchsize) for lo, hi in zip(bounds[:-1], bounds[1:]))
        assert _gz_members(open(out_n, "rb").read()) == expect == _gz_members(open(out_p, "rb").read())


def test_native_host_errors(tmp_path):
    fns, pc, pw, _ = _make_files(tmp_path, 400, 50, seed=5, quirks=False, n_files=1)
    out = str(tmp_path / "o.fastq")
    kw = dict(fastq_fns=fns, fastq_out=out, whitelsit_csv=pw, max_ed=2, n_process=4, batchsize=100, host="native")
    lines = open(pc).read().splitlines(keepends=True)
    # swapped rows -> the reference's message + sys.exit() (read_assignment.py:219-223)
    bad = tmp_path / "swapped.csv"
    bad.write_text("".join([lines[0], lines[2], lines[1]] + lines[3:]))
    with pytest.raises(SystemExit):
        ra.assign_read(putative_bc_csv=str(bad), **kw)
    # fewer csv rows than reads -> StopIteration in the reference, an error here
    short = tmp_path / "short.csv"
    short.write_text("".join(lines[:200]))
    with pytest.raises(_lib.BlazeHipError):
        ra.assign_read(putative_bc_csv=str(short), **kw)
    # missing FASTQ / table / output directory: a BlazeHipError that is also the FileNotFoundError the reference's open() raises
    for bad in ({"fastq_fns": [str(tmp_path / "nope.fastq")]}, {"putative_bc_csv": str(tmp_path / "nope.csv")},
                {"fastq_out": str(tmp_path / "no_such_dir" / "o.fastq")}):
        with pytest.raises(_lib.BlazeHipError) as ei:
            ra.assign_read(**{**dict(putative_bc_csv=pc), **kw, **bad})
        assert isinstance(ei.value, FileNotFoundError), bad
    # and the engine still works afterwards
    assert ra.assign_read(putative_bc_csv=pc, **kw)[1] == 400


def test_native_empty_inputs(tmp_path):
    pw = tmp_path / "wl.csv"
    pw.write_text("ACGTACGTACGTACGT\nTTTTACGTACGTACGT\n")
    pc = tmp_path / "p.csv"
    pc.write_text(HEADER)
    pf = tmp_path / "empty.fastq"
    pf.write_text("")
    for name in ("o.fastq", "o.fastq.gz"):
        out = tmp_path / name
        got = ra.assign_read(fastq_fns=[str(pf)], fastq_out=str(out), putative_bc_csv=str(pc), whitelsit_csv=str(pw),
                             max_ed=2, n_process=2, batchsize=10, host="native")
        assert tuple(got) == (0, 0) and out.read_bytes() == b""


def test_empty_whitelist_file(tmp_path):
    """whitelist.csv without a line (no cell called): every read is seen, none assigned, the output exists and is empty
    (plain) or holds one empty member per batch (.gz), on both hosts."""
    fns, pc, _, bounds = _make_files(tmp_path, 700, 30, seed=3, n_files=2)
    pw = tmp_path / "none.csv"
    pw.write_text("")
    for suffix in (".fastq", ".fastq.gz"):
        outs = {}
        for host in ("native", "python"):
            out = tmp_path / (host + suffix)
            got = ra.assign_read(fastq_fns=fns, fastq_out=str(out), putative_bc_csv=pc, whitelsit_csv=str(pw), max_ed=2,
                                 n_process=4, batchsize=100, host=host)
            assert tuple(map(int, got)) == (0, 700)
            outs[host] = out.read_bytes()
            assert _text(str(out)) == b""
        if suffix.endswith(".gz"):
            assert _gz_members(outs["native"]) == _gz_members(outs["python"]) == sum(-(-(hi - lo) // 100) for lo, hi in zip(bounds[:-1], bounds[1:]))
        else:
            assert outs["native"] == outs["python"] == b""


def test_native_configs3_shape_10nt_umi(tmp_path):
    """BASELINE configs[2]/[4] row shape: 10-nt UMI kits (3v2 / 5v2)."""
    fns, pc, pw, _ = _make_files(tmp_path, 3000, 800, seed=9, umi_len=10, n_files=2)
    kw = dict(fastq_fns=fns, putative_bc_csv=pc, whitelsit_csv=pw, max_ed=2, batchsize=500, restrand=True)
    a = ra.assign_read(fastq_out=str(tmp_path / "n.fastq"), n_process=8, host="native", **kw)
    b = ra.assign_read(fastq_out=str(tmp_path / "p.fastq"), n_process=1, host="python", **kw)
    assert tuple(map(int, a)) == tuple(map(int, b))
    assert (tmp_path / "n.fastq").read_bytes() == (tmp_path / "p.fastq").read_bytes()


class _ThreadComm:
    """The ranks of a pretend world as THREADS of this process on the box's one GPU: real barriers and real sums, so the
    whole of dist.assign_read_sharded runs (read counts -> csv rows to skip -> device pipeline -> every rank placing its
    own part); only the device section is serialised (one engine, calls on a handle are not re-entrant)."""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self._s = rank, world, shared

    def barrier(self):
        self._s["bar"].wait()

    def allreduce_sum_i64(self, arr):
        a = np.asarray(arr, dtype=np.int64)
        with self._s["lock"]:
            self._s["acc"] = self._s.get("acc", 0) + a
        self._s["bar"].wait()
        out = self._s["acc"].copy()
        self._s["bar"].wait()
        if self.rank == 0:
            self._s.pop("acc")
        self._s["bar"].wait()
        return out


def _run_sharded(fns, out, kw, world, locate, threads=4):
    """dist.assign_read_sharded with `world` ranks as threads of this process (one GPU: the device section is serialised).
    kw: putative_bc_csv, whitelsit_csv, max_ed, batchsize, restrand.  Returns {rank: (assigned, total)}."""
    import threading
    from blaze_amd import dist
    shared = {"bar": threading.Barrier(world), "lock": threading.Lock()}
    device_lock = threading.Lock()
    native = dist.native_run_rank(kw["putative_bc_csv"], kw["whitelsit_csv"], kw["max_ed"], threads, kw["batchsize"], kw["restrand"], 0)

    def run_rank(files, part, skip, **how):
        with device_lock:
            return native(files, part, skip, **how)
    run_rank.takes_byte_offset = locate            # True: every rank finds its first row in the table; False: reads are counted
    run_rank.can_raise_mismatch = True
    res, errs = {}, []

    def rank_main(rank):
        try:
            res[rank] = dist.assign_read_sharded(_ThreadComm(rank, world, shared), fns, out, n_process=threads, device=0, run_rank=run_rank, **kw)
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            shared["bar"].abort()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t_ in ts:
        t_.start()
    for t_ in ts:
        t_.join(300)
    assert not errs, errs
    return res


@pytest.mark.parametrize("locate", [True, False], ids=["located_row", "counted_reads"])
@pytest.mark.parametrize("gz", [False, True])
def test_file_shards_concatenate_to_the_single_run(tmp_path, gz, locate):
    """Reads sharded by FILE over ranks (dist.assign_read_sharded -> bz_assign_files_from with the csv rows of the
    earlier files skipped, every rank placing its own part at its offset): the output is the single-process output."""
    import threading
    from blaze_amd import dist
    n = 5000
    fns, pc, pw, bounds = _make_files(tmp_path, n, 400, seed=91, n_files=4)
    suffix = ".fastq.gz" if gz else ".fastq"
    whole = str(tmp_path / ("whole" + suffix))
    kw = dict(putative_bc_csv=pc, whitelsit_csv=pw, max_ed=2, batchsize=300, restrand=True)
    demul, total = ra.assign_read(fastq_fns=fns, fastq_out=whole, n_process=4, host="native", **kw)
    per_file = [dist.count_fastq_reads(f) for f in fns]
    assert per_file == [hi - lo for lo, hi in zip(bounds[:-1], bounds[1:])]          # incl. the gzipped one
    world = 3
    out = str(tmp_path / ("sharded" + suffix))
    res = _run_sharded(fns, out, kw, world, locate)
    assert all(res[r] == (int(demul), int(total)) for r in range(world)) and int(total) == n
    d, t = res[0]
    assert (d, t) == (int(demul), int(total)) and t == n
    assert _text(out) == _text(whole)
    if not gz:
        assert open(out, "rb").read() == open(whole, "rb").read()
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]


def test_many_gzipped_inputs_prefetch_pool(tmp_path, monkeypatch):
    """A nanopore run's shape: many small `.fastq.gz` files (some empty, one plain).  The pool that inflates the next
    files ahead of the reader (host_io.h: FastqFiles) gives the bytes of the streaming reader and of the pandas host,
    with a roomy and with a one-megabyte budget, through libdeflate and through zlib, for step 3 and for the stage-1 table."""
    from blaze_amd import putative_bc
    n = 6000
    fns, pc, pw, bounds = _make_files(tmp_path, n, 300, seed=17, n_files=12, quirks=False)
    gz_fns = []
    for k, fn in enumerate(fns):
        if k == 5 or fn.endswith(".gz"):
            gz_fns.append(fn)
            continue
        raw = open(fn, "rb").read()
        if k == 2:                                   # two members in one file (cat a.gz b.gz), cut in the middle of a line
            blob = gzip.compress(raw[:len(raw) // 3], 1) + gzip.compress(raw[len(raw) // 3:], 9)
        elif k == 7:                                 # zero padding after the member (tape blocks): gzip and zlib read past it
            blob = gzip.compress(raw, 1) + b"\0" * 64
        else:
            blob = gzip.compress(raw, 1)
        with open(fn + ".gz", "wb") as f:
            f.write(blob)
        gz_fns.append(fn + ".gz")
    empty = str(tmp_path / "none.fastq.gz")
    with gzip.open(empty, "wb") as f:
        pass
    gz_fns.insert(3, empty)
    kw = dict(fastq_fns=gz_fns, putative_bc_csv=pc, whitelsit_csv=pw, max_ed=2, batchsize=250, restrand=True)
    want = ra.assign_read(fastq_out=str(tmp_path / "pandas.fastq"), n_process=1, host="python", **kw)
    ref = (tmp_path / "pandas.fastq").read_bytes()
    eng = ra.get_engine(0)
    for name, env in (("pool", {}), ("tight", {"BLAZE_AMD_GZ_PREFETCH_MB": "1"}), ("stream", {"BLAZE_AMD_GZ_PREFETCH": "0"}),
                      ("zlib", {"BLAZE_AMD_LIBDEFLATE": "0"})):
        for k in ("BLAZE_AMD_GZ_PREFETCH_MB", "BLAZE_AMD_GZ_PREFETCH", "BLAZE_AMD_LIBDEFLATE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        got = ra.assign_read(fastq_out=str(tmp_path / f"{name}.fastq"), n_process=8, host="native", **kw)
        assert tuple(map(int, got)) == tuple(map(int, want)), name
        assert (tmp_path / f"{name}.fastq").read_bytes() == ref, name
        counts = putative_bc.putative_bc_files([eng], gz_fns, str(tmp_path / f"{name}.csv"), kit="3v3", umi_len=12, min_q=15,
                                               batch_size=250, n_threads=8)
        assert sum(counts.values()) == n
    assert (tmp_path / "pool.csv").read_bytes() == (tmp_path / "stream.csv").read_bytes() == (tmp_path / "tight.csv").read_bytes()


@pytest.mark.parametrize("gz", [False, True])
def test_two_engines_in_one_process_alternate_chunks(tmp_path, gz):
    """assign_read(devices=[0..7]) in ONE process hands the device batches to the engines round-robin, two slots each
    (bz_assign_files with n_engines > 1).  The box has one GPU: two engine handles on it stand for two devices; the bytes
    and counts are those of the one-engine run, whatever the chunk size."""
    from blaze_amd.engine import Engine
    n = 5000
    fns, pc, pw, _ = _make_files(tmp_path, n, 300, seed=131, n_files=4)
    suffix = ".fastq.gz" if gz else ".fastq"
    one = str(tmp_path / ("one" + suffix))
    want = ra.assign_read(fastq_fns=fns, fastq_out=one, putative_bc_csv=pc, whitelsit_csv=pw, max_ed=2, n_process=8, batchsize=200,
                          restrand=True, host="native")
    e0 = ra.get_engine(0)
    ordered = ra._uploaded[0]                      # the order the one-engine run iterated (list(set(...)) of this process)
    e1 = Engine(0, ra.TIE_RULE)
    try:
        e1.set_whitelist(ordered)
        for rows_per_chunk in (400, 1 << 16):
            two = str(tmp_path / (f"two_{rows_per_chunk}" + suffix))
            got = ra._assign_files_native([e0, e1], fns, pc, two, gz, True, 2, 200, rows_per_chunk, 8)
            assert tuple(map(int, got)) == tuple(map(int, want))
            assert _text(two) == _text(one)
    finally:
        e1.close()


def test_rows_wider_than_the_packed_row(tmp_path):
    """A hand-made table may hold putative barcodes of any length (stage 1 only writes 16 nt).  The native host packs
    rows of up to 32 bytes; a chunk with a longer one is packed again at 64, the device's and the pandas host's limit,
    and gives the pandas host's bytes; beyond 64 both hosts refuse."""
    n = 3000
    fns, pc, pw, _ = _make_files(tmp_path, n, 200, seed=909, quirks=False, n_files=2)
    wl = [w.split('-')[0].strip() for w in open(pw)]
    rng = np.random.default_rng(4)
    lines = open(pc).read().splitlines(keepends=True)
    rnd = lambda k: "".join("ACGT"[c] for c in rng.integers(0, 4, size=k))
    n_wide = 0
    for i in range(1, len(lines)):
        cells = lines[i].rstrip("\n").split(",")
        if not cells[1] or rng.integers(0, 6):
            continue
        k = int(rng.integers(0, 4))
        if k == 0:
            cells[1] = rnd(int(rng.integers(1, 20))) + wl[int(rng.integers(0, len(wl)))] + rnd(int(rng.integers(1, 20)))   # a cell inside a long cell
        elif k == 1:
            cells[1] = rnd(int(rng.integers(29, 61)))
        elif k == 2:
            cells[1] = rnd(60)                                   # 2 + 60 + 2 = 64: the limit itself
        else:
            cells[1] = rnd(29)                                   # 33 bytes: the first width past the packed row
        n_wide += len(cells[1]) > 28
        lines[i] = ",".join(cells) + "\n"
    assert n_wide > 100
    wide = tmp_path / "wide.csv"
    wide.write_text("".join(lines))
    kw = dict(fastq_fns=fns, putative_bc_csv=str(wide), whitelsit_csv=pw, max_ed=2, batchsize=200, restrand=True)
    for dbr in (1 << 16, 400):                                   # all wide rows in one chunk / chunks with and without them
        out_n, out_p = str(tmp_path / f"n{dbr}.fastq"), str(tmp_path / f"p{dbr}.fastq")
        got_p = ra.assign_read(fastq_out=out_p, n_process=1, host="python", device_batch_rows=dbr, **kw)
        got_n = ra.assign_read(fastq_out=out_n, n_process=4, host="native", device_batch_rows=dbr, **kw)
        assert tuple(map(int, got_n)) == tuple(map(int, got_p))
        assert open(out_n, "rb").read() == open(out_p, "rb").read()
    # 65 bytes: refused by both hosts
    cells = lines[5].rstrip("\n").split(",")
    cells[1], cells[3], cells[5] = rnd(61), "ACGTACGTACGT", "ACGTA"
    lines[5] = ",".join(cells) + "\n"
    too = tmp_path / "too_wide.csv"
    too.write_text("".join(lines))
    with pytest.raises(_lib.BlazeHipError, match="exceeds the device limit of 64"):
        ra.assign_read(fastq_out=str(tmp_path / "x.fastq"), n_process=4, host="native", **{**kw, "putative_bc_csv": str(too)})
    with pytest.raises(ValueError, match="exceeds the device limit of 64"):
        ra.assign_read(fastq_out=str(tmp_path / "y.fastq"), n_process=1, host="python", **{**kw, "putative_bc_csv": str(too)})


def test_sharded_run_with_a_read_id_that_occurs_twice(tmp_path, monkeypatch):
    """A read id that occurs twice in the job can send a group's row search to the earlier of its rows; the id check then
    fails at once and the rank counts the reads in front of the group instead: same bytes as the single-process run."""
    monkeypatch.setenv("BLAZE_AMD_GROUP_MIN_BYTES", "1")
    n = 3000
    fns, pc, pw, bounds = _make_files(tmp_path, n, 200, seed=17, n_files=4, quirks=False)
    # the first read of the third file gets a twin: read 5 of the first file takes its id, in the FASTQ file and in the table
    lines = open(pc).read().splitlines(keepends=True)
    twin = lines[1 + bounds[2]].split(",")[0]
    old = lines[1 + 5].split(",")[0]
    lines[1 + 5] = twin + lines[1 + 5][len(old):]
    open(pc, "w").write("".join(lines))
    fq = open(fns[0]).read()
    assert fq.count("@" + old + " ") == 1
    open(fns[0], "w").write(fq.replace("@" + old + " ", "@" + twin + " ").replace("+" + old + "\n", "+\n"))
    kw = dict(putative_bc_csv=pc, whitelsit_csv=pw, max_ed=2, batchsize=250, restrand=True)
    whole = str(tmp_path / "whole.fastq")
    demul, total = ra.assign_read(fastq_fns=fns, fastq_out=whole, n_process=4, host="native", **kw)
    out = str(tmp_path / "sharded.fastq")
    res = _run_sharded(fns, out, kw, 3, True)
    assert all(res[r] == (int(demul), int(total)) for r in range(3)) and int(total) == n
    assert open(out, "rb").read() == open(whole, "rb").read()
