"""CPU: host-side logic that needs no GPU, and the C-ABI library's surface."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from blaze_amd import _lib
    header = open(os.path.join(ROOT, "include", "blaze_hip.h")).read()
    declared = set(re.findall(r"\b(bz_[a-z0-9_]+)\s*\(", header))
    declared.discard("bz_engine")
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    L = _lib.load()                      # dlopen works without a GPU; prototypes resolve every symbol
    for name in declared:
        assert hasattr(L, name)
    assert L.bz_abi_version() == 1


def test_no_gpu_means_loud_failure_not_fallback():
    from blaze_amd import _lib
    L = _lib.load()
    n = ctypes.c_int(-1)
    rc = L.bz_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible here")
    from blaze_amd.engine import Engine
    with pytest.raises(_lib.BlazeHipError):
        Engine(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "blaze_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle" not in text.lower(), f"{fn} mentions the oracle"


def test_helper_mirrors():
    from blaze_amd import helper
    assert helper.reverse_complement("AAGGCCTTNNxyz*-") == "-*zyxNNAAGGCCTT"
    assert helper.reverse_complement("acgtn") == "nacgt"
    assert list(helper.batch_iterator(iter(range(7)), 3)) == [[0, 1, 2], [3, 4, 5], [6]]
    assert list(helper.batch_iterator(iter([]), 3)) == []
    recs = list(helper.fastq_parser(iter(["@id1 extra words\n", "ACGT\n", "+\n", "IIII\n", "@id2\n", "GG\n", "+id2\n", "##\n"])))
    assert recs == [("id1", "ACGT", "IIII"), ("id2", "GG", "##")]


def test_row_packing_and_decoding():
    from blaze_amd import read_assignment as ra
    ext, el, bo, bl = ra._pack_rows(["ACGTACGTACGTACGT", "", "ACGTACGTACGTAC"], ["GGGGGGGGGGGG", "", "T"], ["GATCT", "", "A"])
    assert bytes(ext[0, :20]) == b"CTACGTACGTACGTACGTGG" and (el[0], bo[0], bl[0]) == (20, 2, 16)
    assert (el[1], bo[1], bl[1]) == (0, 0, 0)
    assert bytes(ext[2, :16]) == b"AACGTACGTACGTACT" and (el[2], bo[2], bl[2]) == (16, 1, 14)
    wl = ["ACGTACGTACGTACGA", "TTTTTTTTTTTTTTTT"]
    hit = np.array([0, -1, 1]); end = np.array([18, -1, 11]); flags = np.array([0, 2, 0])
    out = ra._decode_rows(hit, end, flags, el, ["ACGTACGTACGTACGT", "", "ACGTACGTACGTAC"], ["GGGGGGGGGGGG", "", "T"],
                          ["CCCCC", "", "AAAAA"], [33.0, "", -20], wl)
    assert out[0] == ["ACGTACGTACGTACGA", "GGGGGGGGGGGC", "-"]      # adj +1: umi[1:] + post[:1]
    assert out[1] == ["", "", ""]
    assert out[2] == ["TTTTTTTTTTTTTTTT", "AC" + "", "+"]            # adj -2: bc[-2:] + umi[:-2]
    assert ra._strand(0) == "" and ra._strand("") == "" and ra._strand(5.0) == "-" and ra._strand(-1) == "+"


def test_shard_ranges_cover_and_preserve_order():
    from blaze_amd import dist
    for n in (0, 1, 7, 8, 1000, 1001):
        for world in (1, 2, 3, 8):
            spans = [dist.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- the FASTQ-aware gzip member writer (blaze_amd/csrc/fastq_deflate.h) through the C ABI: host only, no GPU ----------
def _fastq_text(rng, n_reads, max_len=1500, odd=False):
    import numpy as np
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for k in range(n_reads):
        L = int(rng.integers(0 if odd else 60, max_len))
        seq = acgt[rng.integers(0, 4, L)].tobytes()
        if odd and k % 3 == 0:
            seq = seq.replace(b"A", b"N", 2).lower()
        q = bytes(rng.integers(33, 90, L, dtype=np.uint8))
        out.append(b"@ACGTACGTACGTACGT_TTTTGGGGAAAA#%08x-read_+\tCB:Z:ACGTACGTACGTACGT\tUB:Z:TTTTGGGGAAAA\n" % k + seq + b"\n+\n" + q + b"\n")
    return b"".join(out)


def test_gzip_fastq_member_round_trips():
    """Python's gzip reads back exactly the text the native member writer was given: FASTQ-shaped members, records with
    empty / very long / non-ACGT sequences, texts that are not FASTQ at all, and the empty member of an empty batch."""
    import gzip
    import numpy as np
    from blaze_amd import helper
    rng = np.random.default_rng(7)
    texts = [b"", b"\n", b"A", b"@r\n\n+\n\n", _fastq_text(rng, 1), _fastq_text(rng, 300), _fastq_text(rng, 50, odd=True),
             _fastq_text(rng, 3, max_len=200_000), bytes(rng.integers(0, 256, 100_000, dtype=np.uint8)), b"A" * 300_000,
             b"@x\n" + b"ACGT" * 5000, b"\n" * 1000, bytes(range(256)) * 40]
    sizes = []
    for t in texts:
        z = helper.gzip_fastq_member(t)
        assert gzip.decompress(z) == t
        sizes.append((len(t), len(z)))
    # several members back to back are one gzip file (what assign_read writes: one member per batch)
    z = b"".join(helper.gzip_fastq_member(t) for t in texts[4:7])
    assert gzip.decompress(z) == b"".join(texts[4:7])
    # and it is a compressor: FASTQ text with uniform qualities over 57 levels shrinks to about (2.25 + 5.9) / 16 of its size
    n, zn = sizes[5]
    assert zn < 0.56 * n, (n, zn)


def test_gzip_fastq_member_hypothesis():
    import gzip
    from hypothesis import given, settings, strategies as st
    from blaze_amd import helper

    line = st.one_of(st.binary(max_size=200), st.text(alphabet="ACGTN", max_size=400).map(str.encode),
                     st.text(alphabet="ACGT", min_size=48, max_size=300).map(str.encode))

    @settings(max_examples=150, deadline=None)
    @given(st.lists(line, max_size=40), st.booleans())
    def check(lines, trailing_newline):
        t = b"\n".join(lines) + (b"\n" if trailing_newline else b"")
        assert gzip.decompress(helper.gzip_fastq_member(t)) == t
    check()


def test_gzip_member_crc_matches_zlib_on_edge_lengths():
    """The carry-less-multiplication CRC-32 takes whole 16-byte chunks of texts of at least 64 bytes, zlib the rest:
    lengths around every one of those edges, against zlib.crc32 (the trailer's CRC field) and against gzip itself."""
    import gzip
    import zlib
    import numpy as np
    from blaze_amd import helper
    rng = np.random.default_rng(1)
    for n in [0, 1, 15, 16, 63, 64, 65, 79, 80, 127, 128, 129, 143, 1000, 4095, 4096, 4097, 100003]:
        t = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        z = helper.gzip_fastq_member(t)
        assert gzip.decompress(z) == t
        assert int.from_bytes(z[-8:-4], "little") == zlib.crc32(t), n
        assert int.from_bytes(z[-4:], "little") == n


def test_gzip_member_sequence_line_trees():
    """Sequence lines of plain ACGT take one of four fixed trees, sixteen bases per step: every length around the step
    sizes, each base as the rarest one, ties, lopsided lines (their own tree), one- / two- / three-letter lines, CRLF,
    a last line without a newline, lines with one foreign byte at every position class -- gzip reads the text back, and
    balanced lines cost the 2.25 bits per base of the balanced tree."""
    import gzip
    import numpy as np
    from blaze_amd import helper
    rng = np.random.default_rng(11)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rec(seq, nl=b"\n"):
        return b"@r" + nl + seq + nl + b"+" + nl + b"I" * len(seq) + nl
    for L in list(range(40, 120)) + [255, 256, 257, 1000, 1023, 1024, 1025, 4099]:
        seq = acgt[rng.integers(0, 4, L)].tobytes()
        t = rec(seq) * 3
        assert gzip.decompress(helper.gzip_fastq_member(t)) == t, L
    for rare in range(4):                                   # which base is the rare one (and exact ties between the others)
        p = np.full(4, 0.3); p[rare] = 0.1
        seq = acgt[rng.choice(4, 900, p=p)].tobytes()
        t = rec(seq) + rec(bytes(acgt[[rare, (rare + 1) % 4, (rare + 2) % 4, (rare + 3) % 4] * 50]))
        assert gzip.decompress(helper.gzip_fastq_member(t)) == t
    odd = [b"A" * 500, b"AC" * 300, b"ACG" * 200, b"A" * 400 + b"CGT" * 20, b"ACGT" * 100 + b"N", b"N" + b"ACGT" * 100,
           b"ACGT" * 50 + b"a" + b"ACGT" * 50, b"ACGT" * 20 + b"\0" + b"ACGT" * 20, bytes(acgt[rng.integers(0, 4, 64)]) + b"\r"]
    for seq in odd:
        for nl in (b"\n", b"\r\n"):
            t = rec(seq, nl) + rec(b"ACGT" * 30, nl)
            assert gzip.decompress(helper.gzip_fastq_member(t)) == t, seq[:20]
            assert gzip.decompress(helper.gzip_fastq_member(t[:-len(nl)])) == t[:-len(nl)]
    t = b"@r\n" + acgt[rng.integers(0, 4, 5000)].tobytes()   # the sequence line is the last line and has no newline
    assert gzip.decompress(helper.gzip_fastq_member(t)) == t
    seq = acgt[rng.integers(0, 4, 100_000)].tobytes()
    z = helper.gzip_fastq_member(b"@r\n" + seq + b"\n+\n")
    assert len(z) * 8 < 2.26 * len(seq) + 1000


def test_host_threading_under_thread_sanitizer(tmp_path):
    """tools/host_tsan.cpp (the job-long thread pool, the member encoder's per-thread buffers, the ordered
    reader / submit / finish / writer frame of the file jobs with a pretend device, the pool that inflates gzipped inputs
    ahead of the reader) built with -fsanitize=thread: no report.  Host code only."""
    import shutil
    import subprocess
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_tsan")
    if shutil.which("g++") is None or shutil.which("gzip") is None:
        pytest.skip("needs g++ and gzip")
    b = subprocess.run(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17", "-pthread", "-w", os.path.join(root, "tools", "host_tsan.cpp"),
                        "-lz", "-ldl", "-o", exe], capture_output=True, text=True)
    if b.returncode != 0 and "tsan" in b.stderr.lower():
        pytest.skip("no ThreadSanitizer runtime on this host")
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe, str(tmp_path / "fq")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert "lines 65000" in r.stdout and "caught 50" in r.stdout
    assert r.stdout.count("as expected") == 5 and "WRONG" not in r.stdout      # the reader -> submit -> finish -> writer frame, every way of failing


def test_host_parse_and_format_under_address_sanitizer(tmp_path):
    """tools/host_asan.cpp: the lock-step reader, parse_batch, format_batch and the stage-1 host side on quirky inputs
    with made-up device results, built with -fsanitize=address,undefined and every copied line in a heap block of its
    own (BZ_ARENA_EXACT): no report, with the streaming gz reader and with the prefetch pool.  Host code only."""
    import shutil
    import subprocess
    import pytest
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_asan")
    if shutil.which("g++") is None or shutil.which("gzip") is None:
        pytest.skip("needs g++ and gzip")
    b = subprocess.run(["g++", "-O1", "-g", "-DBZ_ARENA_EXACT", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17",
                        "-pthread", "-w", os.path.join(root, "tools", "host_asan.cpp"), "-lz", "-ldl", "-o", exe], capture_output=True, text=True)
    if b.returncode != 0 and "asan" in b.stderr.lower():
        pytest.skip("no AddressSanitizer runtime on this host")
    assert b.returncode == 0, b.stderr[-2000:]
    for prefetch in ("0", "1"):
        env = dict(os.environ, BLAZE_AMD_GZ_PREFETCH=prefetch)
        r = subprocess.run([exe, str(tmp_path / f"d{prefetch}")], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        assert "bytes formatted" in r.stdout


def test_file_errors_are_also_file_not_found():
    """An input / output path the native loops cannot open: BlazeHipError for callers of this package, FileNotFoundError for
    callers written against the reference's open() calls (read_assignment.py:149-153, :279)."""
    from blaze_amd import _lib
    assert issubclass(_lib.BlazeFileNotFoundError, _lib.BlazeHipError) and issubclass(_lib.BlazeFileNotFoundError, FileNotFoundError)
    L = _lib.load()
    n = __import__("ctypes").c_int64(0)
    rc = L.bz_fastq_count_reads(b"/no/such/dir/reads.fastq", __import__("ctypes").byref(n))       # host-only entry point
    assert rc < 0
    try:
        _lib.check(rc)
    except FileNotFoundError as e:
        assert isinstance(e, RuntimeError) and "cannot open" in str(e)
    else:
        raise AssertionError("no exception")


def test_native_open_errors_carry_errno(tmp_path):
    """The native open / map failures raise what Python's own open() raises for the same path (ENOENT, EISDIR, EACCES),
    each also a BlazeHipError, with the errno set."""
    import ctypes
    import errno
    from blaze_amd import _lib
    L = _lib.load()
    n = ctypes.c_int64(0)

    def err(path):
        rc = L.bz_fastq_count_reads(os.fsencode(str(path)), ctypes.byref(n))
        assert rc < 0
        with pytest.raises(_lib.BlazeHipError) as ei:
            _lib.check(rc)
        return ei.value
    e = err(tmp_path / "nope.fastq")
    assert isinstance(e, FileNotFoundError) and e.errno == errno.ENOENT and "cannot open" in str(e)
    e = err(tmp_path)
    assert isinstance(e, IsADirectoryError) and e.errno == errno.EISDIR
    e = err(tmp_path / "nope.fastq.gz")
    assert isinstance(e, FileNotFoundError)
    if os.geteuid() != 0:                                  # root opens anything
        p = tmp_path / "secret.fastq"
        p.write_text("@a\nA\n+\nI\n")
        p.chmod(0)
        e = err(p)
        assert isinstance(e, PermissionError) and e.errno == errno.EACCES
