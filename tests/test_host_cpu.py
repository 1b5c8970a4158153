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
