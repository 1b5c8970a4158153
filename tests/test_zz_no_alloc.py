"""bz_assign_device_noalloc -- the entry point that allocates nothing -- against bz_assign_submit/wait.

Called through the C ABI with rows the test itself puts into device memory (ctypes on the HIP runtime: the
product never sees host arrays here).  Bar: bit-exact hit / ed / end / flags and the same per-barcode counts,
on two calls in a row, the second with a smaller input (the scratch array keeps the first call's contents).
Collected last on purpose (file name): the round's end runs `pytest -x`.
"""
import ctypes

import numpy as np
import pytest

from blaze_amd import _lib, synth

pytestmark = pytest.mark.gpu

_hip = None


def hip():
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
        _hip.hipFree.argtypes = [ctypes.c_void_p]
        _hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        _hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
    return _hip


class DevBuf:
    """A device allocation owned by the TEST (the library under test allocates nothing in this call)."""

    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        self.nbytes = max(int(nbytes), 16)
        assert hip().hipMalloc(ctypes.byref(self.ptr), self.nbytes) == 0

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        if a.nbytes:
            assert hip().hipMemcpy(self.ptr, a.ctypes.data, a.nbytes, 1) == 0      # hipMemcpyHostToDevice
        return self

    def download(self, dtype, n):
        out = np.empty(n, dtype=dtype)
        if out.nbytes:
            assert hip().hipMemcpy(out.ctypes.data, self.ptr, out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    def fill(self, byte):
        assert hip().hipMemset(self.ptr, byte, self.nbytes) == 0

    def free(self):
        if self.ptr:
            hip().hipFree(self.ptr)
            self.ptr = ctypes.c_void_p()


def run_noalloc(engine, bufs, ext, el, bo, bl, max_ed):
    n, stride = ext.shape
    d_ext, d_len, d_off, d_bcl, d_hit, d_ed, d_end, d_flags = bufs
    d_ext.upload(ext), d_len.upload(el), d_off.upload(bo), d_bcl.upload(bl)
    for b in (d_hit, d_ed, d_end, d_flags):
        b.fill(0x5A)                                 # stale bytes: every row's result must be written by the call
    # uploads and fills went to the null stream; the engine's streams are non-blocking (no implicit order with it) and a
    # copy from pageable memory may still be in flight when hipMemcpy returns: the caller orders its inputs itself
    assert hip().hipDeviceSynchronize() == 0
    L = _lib.load()
    _lib.check(L.bz_assign_device_noalloc(engine._h, d_ext.ptr, stride, d_len.ptr, d_off.ptr, d_bcl.ptr, n, max_ed,
                                          d_hit.ptr, d_ed.ptr, d_end.ptr, d_flags.ptr, None))
    engine.sync(0)                                   # stream NULL = the engine's slot-0 stream
    return (d_hit.download(np.int32, n), d_ed.download(np.int8, n), d_end.download(np.int8, n),
            d_flags.download(np.uint8, n))


def _workload(n_rows, n_cells, seed):
    full = synth.full_whitelist_codes(60_000, seed=31)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=32)
    rows = synth.make_rows(n_rows, cells, full, sub=0.03, ins=0.01, dele=0.02, seed=seed)
    ext, el, bo, bl = synth.pack_ext(rows)
    # a few irregular rows (N, lower case, short barcode) so the byte-wise scan runs too
    ext = ext.copy()
    full_rows = np.flatnonzero((bl == 16) & (el == 20))
    ext[full_rows[5], 7] = ord('N')
    ext[full_rows[11], 3] = ord('a')
    return ext, el, bo, bl, synth.codes_to_bytes(cells)


@pytest.mark.parametrize("max_ed", [2, 1, 3])
def test_noalloc_equals_submit_wait_twice(engine, max_ed):
    n1, n2 = 200_000, 37_777
    ext, el, bo, bl, wl = _workload(n1, 3000, seed=77 + max_ed)
    engine.set_whitelist_codes16(wl)
    for opt, v in ((_lib.BZ_OPT_SCAN_MODE, 0), (_lib.BZ_OPT_PAIR_FUSED, 0), (_lib.BZ_OPT_TIE_RULE, 0),
                   (_lib.BZ_OPT_SHORTCUT, 1), (_lib.BZ_OPT_FORCE_GENERIC, 0)):
        engine.set_option(opt, v)
    stride = ext.shape[1]
    bufs = [DevBuf(n1 * stride), DevBuf(n1), DevBuf(n1), DevBuf(n1), DevBuf(4 * n1), DevBuf(n1), DevBuf(n1), DevBuf(n1)]
    try:
        for n, lo in ((n1, 0), (n2, 1234)):          # the second call is smaller and starts elsewhere
            e, l, o, b = ext[lo:lo + n], el[lo:lo + n], bo[lo:lo + n], bl[lo:lo + n]
            engine.counts_reset()
            ref = engine.assign(e, l, o, b, max_ed)
            ref_counts = engine.counts()
            engine.counts_reset()
            got = run_noalloc(engine, bufs, e, l, o, b, max_ed)
            got_counts = engine.counts()
            for name, g, r in zip(("hit", "ed", "end", "flags"), got, ref):
                bad = np.flatnonzero(g != r)
                assert bad.size == 0, f"n={n} max_ed={max_ed}: {name} differs on {bad.size} rows, first {bad[0]}: {g[bad[0]]} vs {r[bad[0]]}"
            assert np.array_equal(got_counts, ref_counts)
            assert (got[0] >= 0).sum() > n // 4
            assert lo or (got[3] & _lib.BZ_F_GENERIC).any()       # the injected irregular rows sit in the first call's range
    finally:
        for b in bufs:
            b.free()


def test_noalloc_limits_and_invalid_rows(engine):
    ext, el, bo, bl, wl = _workload(4096, 500, seed=5)
    engine.set_whitelist_codes16(wl)
    L = _lib.load()
    one = DevBuf(64)
    try:
        rc = L.bz_assign_device_noalloc(engine._h, one.ptr, 32, one.ptr, one.ptr, one.ptr, _lib.BZ_NOALLOC_MAX_ROWS + 1, 2,
                                        one.ptr, one.ptr, one.ptr, one.ptr, None)
        assert rc == -5 and b"at most" in L.bz_last_error()           # BZ_E_LIMIT, nothing launched
        assert L.bz_assign_device_noalloc(engine._h, one.ptr, 32, one.ptr, one.ptr, one.ptr, 0, 2,
                                          one.ptr, one.ptr, one.ptr, one.ptr, None) == 0
    finally:
        one.free()
    # rows whose lengths do not fit: the host entry point rejects the batch, the device entry point marks the rows
    n = 4096
    el2, bo2, bl2 = el.copy(), bo.copy(), bl.copy()
    r0, r1, r2, r3 = (int(x) for x in np.flatnonzero((bl == 16) & (el == 20))[[10, 20, 30, 40]])
    el2[r0] = 200                                    # ext_len > stride
    bo2[r1] = 10                                     # barcode runs past the extension
    with pytest.raises(_lib.BlazeHipError):
        engine.assign(ext, el2, bo2, bl2, 2)
    el2[r2], bl2[r3] = 17, 19                        # one column short of the barcode's end; barcode one longer than fits
    bad = [r0, r1, r2, r3]
    stride = ext.shape[1]
    bufs = [DevBuf(n * stride), DevBuf(n), DevBuf(n), DevBuf(n), DevBuf(4 * n), DevBuf(n), DevBuf(n), DevBuf(n)]
    try:
        for width in (stride, 24, 20):               # 32, 24: the packed row layouts; 20: the general kernels
            e = np.ascontiguousarray(ext[:, :width])
            got = run_noalloc(engine, bufs, e, el2, bo2, bl2, 2)
            ref = engine.assign(e, el, bo, bl, 2)
            keep = np.ones(n, dtype=bool)
            keep[bad] = False
            for g, r in zip(got, ref):
                assert np.array_equal(g[keep], r[keep])
            for r in bad:
                assert got[0][r] == -1 and got[3][r] == (_lib.BZ_F_BLANK | _lib.BZ_F_INVALID), (width, r, [int(g[r]) for g in got])
    finally:
        for b in bufs:
            b.free()


def test_noalloc_under_stream_capture(engine):
    """The call recorded into a HIP graph by stream capture in GLOBAL mode (an allocating or synchronising call from the
    capturing thread would fail the capture) and the graph replayed on two different batches: what the entry point is for."""
    H = hip()
    for name in ("hipStreamCreateWithFlags", "hipStreamBeginCapture", "hipStreamEndCapture", "hipGraphInstantiate", "hipGraphLaunch",
                 "hipStreamSynchronize", "hipGraphExecDestroy", "hipGraphDestroy", "hipStreamDestroy"):
        getattr(H, name).restype = ctypes.c_int
    H.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
    H.hipStreamBeginCapture.argtypes = [ctypes.c_void_p, ctypes.c_int]
    H.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
    H.hipGraphInstantiate.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    H.hipGraphLaunch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    H.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
    H.hipGraphExecDestroy.argtypes = [ctypes.c_void_p]
    H.hipGraphDestroy.argtypes = [ctypes.c_void_p]
    H.hipStreamDestroy.argtypes = [ctypes.c_void_p]
    n = 100_000
    ext, el, bo, bl, wl = _workload(2 * n, 2000, seed=123)
    engine.set_whitelist_codes16(wl)
    for opt, v in ((_lib.BZ_OPT_SCAN_MODE, 0), (_lib.BZ_OPT_PAIR_FUSED, 0), (_lib.BZ_OPT_TIE_RULE, 0),
                   (_lib.BZ_OPT_SHORTCUT, 1), (_lib.BZ_OPT_FORCE_GENERIC, 0)):
        engine.set_option(opt, v)
    stride = ext.shape[1]
    bufs = [DevBuf(n * stride), DevBuf(n), DevBuf(n), DevBuf(n), DevBuf(4 * n), DevBuf(n), DevBuf(n), DevBuf(n)]
    d_ext, d_len, d_off, d_bcl, d_hit, d_ed, d_end, d_flags = bufs
    stream, graph, gexec = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    L = _lib.load()
    try:
        assert H.hipStreamCreateWithFlags(ctypes.byref(stream), 1) == 0           # hipStreamNonBlocking
        assert H.hipStreamBeginCapture(stream, 0) == 0                            # hipStreamCaptureModeGlobal
        rc = L.bz_assign_device_noalloc(engine._h, d_ext.ptr, stride, d_len.ptr, d_off.ptr, d_bcl.ptr, n, 2,
                                        d_hit.ptr, d_ed.ptr, d_end.ptr, d_flags.ptr, stream)
        end_rc = H.hipStreamEndCapture(stream, ctypes.byref(graph))
        assert rc == 0, L.bz_last_error()
        assert end_rc == 0 and graph.value
        assert H.hipGraphInstantiate(ctypes.byref(gexec), graph, None, None, 0) == 0
        for lo in (0, n):                                                         # the same graph on two different batches
            e, l, o, b = ext[lo:lo + n], el[lo:lo + n], bo[lo:lo + n], bl[lo:lo + n]
            engine.counts_reset()
            ref = engine.assign(e, l, o, b, 2)
            ref_counts = engine.counts()
            engine.counts_reset()
            d_ext.upload(e), d_len.upload(l), d_off.upload(o), d_bcl.upload(b)
            for buf in (d_hit, d_ed, d_end, d_flags):
                buf.fill(0x5A)
            assert H.hipDeviceSynchronize() == 0
            assert H.hipGraphLaunch(gexec, stream) == 0
            assert H.hipStreamSynchronize(stream) == 0
            got = (d_hit.download(np.int32, n), d_ed.download(np.int8, n), d_end.download(np.int8, n), d_flags.download(np.uint8, n))
            for name, g, r in zip(("hit", "ed", "end", "flags"), got, ref):
                assert np.array_equal(g, r), (lo, name)
            assert np.array_equal(engine.counts(), ref_counts)
    finally:
        if gexec.value:
            H.hipGraphExecDestroy(gexec)
        if graph.value:
            H.hipGraphDestroy(graph)
        if stream.value:
            H.hipStreamDestroy(stream)
        for buf in bufs:
            buf.free()
