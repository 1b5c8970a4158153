"""Engine: one GPU's whitelist-assignment engine, a thin numpy front for the C ABI.

Replaces, for step 3, both the per-pair FFI (fast_edit_distance.sub_edit_distance,
/root/reference/blaze/read_assignment.py:72) and the worker pool that fanned batches
out (/root/reference/blaze/helper.py:111-205): batches become device batches on two
double-buffered slots and results come back in submit order.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import BlazeHipError, check

TIE_LEFTMOST, TIE_RIGHTMOST = 0, 1


def _ptr(a):
    return None if a is None else a.ctypes.data


def pack_whitelist(whitelist):
    """list[str|bytes] -> (uint8 bytes, int64 offsets) in the caller's order."""
    enc = [w if isinstance(w, bytes) else w.encode("latin-1") for w in whitelist]
    off = np.zeros(len(enc) + 1, dtype=np.int64)
    if enc:
        np.cumsum([len(w) for w in enc], out=off[1:])
    data = np.frombuffer(b"".join(enc) or b"\0", dtype=np.uint8).copy()
    return data, off


class PinnedRows:
    """One batch's input and result arrays as numpy views of ONE pinned host allocation (bz_host_alloc): what
    bz_assign_submit_pinned DMAs from and into.  Fill `ext / ext_len / bc_off / bc_len`, read `hit / ed / end / flags`."""

    def __init__(self, n: int, stride: int):
        self._L = _lib.load()
        self.n, self.stride = int(n), int(stride)
        sizes = [n * stride, n, n, n, 4 * n, n, n, n]
        offs = np.concatenate([[0], np.cumsum([(b + 255) // 256 * 256 for b in sizes])])
        self._ptr = ctypes.c_void_p()
        check(self._L.bz_host_alloc(ctypes.byref(self._ptr), int(offs[-1])))
        raw = (ctypes.c_uint8 * int(offs[-1])).from_address(self._ptr.value)
        buf = np.frombuffer(raw, dtype=np.uint8)
        v = [buf[offs[k]:offs[k] + sizes[k]] for k in range(8)]
        self.ext, self.ext_len, self.bc_off, self.bc_len = v[0].reshape(n, stride), v[1], v[2], v[3]
        self.hit, self.ed, self.end, self.flags = v[4].view(np.int32), v[5].view(np.int8), v[6].view(np.int8), v[7]

    def fill(self, ext, ext_len, bc_off, bc_len):
        self.ext[...], self.ext_len[...], self.bc_off[...], self.bc_len[...] = ext, ext_len, bc_off, bc_len
        return self

    _pending = None          # (engine handle, slot) while a submit_pinned on these arrays has not been waited for

    def free(self):
        """Give the pinned allocation back.  If a submit on these arrays is still in flight its slot is synchronised first:
        the DMA into hit / ed / end / flags must not land in freed memory."""
        if self._ptr:
            if self._pending is not None:
                h, slot = self._pending
                self._L.bz_slot_sync(h, slot)
                self._pending = None
            self.ext = self.ext_len = self.bc_off = self.bc_len = self.hit = self.ed = self.end = self.flags = None
            self._L.bz_host_free(self._ptr)
            self._ptr = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:      # noqa: BLE001  (interpreter shutdown: the library may be gone)
            pass


class Engine:
    def __init__(self, device: int = 0, tie_rule: int = TIE_LEFTMOST):
        self._L = _lib.load()
        h = ctypes.c_void_p()
        check(self._L.bz_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = int(device)
        self.n_whitelist = -1
        self._pending = [None, None]
        self.set_option(_lib.BZ_OPT_TIE_RULE, tie_rule)

    # -- lifetime -----------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.bz_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- configuration ------------------------------------------------------------------
    def set_option(self, option: int, value: int):
        check(self._L.bz_set_option(self._h, int(option), int(value)))

    def set_whitelist(self, whitelist):
        """Upload an ORDERED whitelist (list of str/bytes, or (uint8 bytes, int64 offsets))."""
        if isinstance(whitelist, tuple):
            data, off = whitelist
            data = np.ascontiguousarray(data, dtype=np.uint8)
            off = np.ascontiguousarray(off, dtype=np.int64)
        else:
            data, off = pack_whitelist(whitelist)
        check(self._L.bz_set_whitelist(self._h, data.ctypes.data, off.ctypes.data, len(off) - 1))
        self.n_whitelist = len(off) - 1

    def set_whitelist_codes16(self, ascii_matrix: np.ndarray):
        """(n,16) uint8 ASCII matrix -> upload without building Python strings."""
        m = np.ascontiguousarray(ascii_matrix, dtype=np.uint8)
        off = np.arange(m.shape[0] + 1, dtype=np.int64) * m.shape[1]
        self.set_whitelist((m.reshape(-1), off))

    # -- rows ---------------------------------------------------------------------------
    @staticmethod
    def _rows(ext, ext_len, bc_off, bc_len):
        ext = np.ascontiguousarray(ext, dtype=np.uint8)
        if ext.ndim != 2:
            raise ValueError("ext must be (n, stride) uint8")
        n = ext.shape[0]
        arrs = [np.ascontiguousarray(a, dtype=np.uint8) for a in (ext_len, bc_off, bc_len)]
        if any(a.shape != (n,) for a in arrs):
            raise ValueError("ext_len / bc_off / bc_len must have one entry per row")
        return ext, arrs[0], arrs[1], arrs[2]

    def submit(self, slot, ext, ext_len, bc_off, bc_len, max_ed: int):
        ext, el, bo, bl = self._rows(ext, ext_len, bc_off, bc_len)
        n, stride = ext.shape
        check(self._L.bz_assign_submit(self._h, slot, _ptr(ext), max(stride, 1), _ptr(el), _ptr(bo), _ptr(bl), n,
                                       int(max_ed)))
        self._pending[slot] = n

    def wait(self, slot):
        n = self._pending[slot]
        if n is None:
            raise BlazeHipError(f"slot {slot} has nothing submitted")
        hit = np.empty(n, dtype=np.int32)
        ed = np.empty(n, dtype=np.int8)
        end = np.empty(n, dtype=np.int8)
        flags = np.empty(n, dtype=np.uint8)
        check(self._L.bz_assign_wait(self._h, slot, _ptr(hit), _ptr(ed), _ptr(end), _ptr(flags)))
        self._pending[slot] = None
        return hit, ed, end, flags

    def assign(self, ext, ext_len, bc_off, bc_len, max_ed: int, slot: int = 0):
        """submit + wait on one slot: (hit, ed, end_idx, flags) in row order."""
        self.submit(slot, ext, ext_len, bc_off, bc_len, max_ed)
        return self.wait(slot)

    # rows and results in pinned host memory: no staging copy on either side of the PCIe transfers
    def pinned_rows(self, ext, ext_len, bc_off, bc_len):
        ext, el, bo, bl = self._rows(ext, ext_len, bc_off, bc_len)
        return PinnedRows(ext.shape[0], max(ext.shape[1], 1)).fill(ext, el, bo, bl)

    def submit_pinned(self, slot, rows: "PinnedRows", max_ed: int):
        check(self._L.bz_assign_submit_pinned(self._h, slot, _ptr(rows.ext), rows.stride, _ptr(rows.ext_len), _ptr(rows.bc_off),
                                              _ptr(rows.bc_len), rows.n, int(max_ed), _ptr(rows.hit), _ptr(rows.ed), _ptr(rows.end),
                                              _ptr(rows.flags)))
        self._pending[slot] = rows.n
        rows._pending = (self._h, slot)

    def wait_pinned(self, slot, rows: "PinnedRows"):
        check(self._L.bz_assign_wait(self._h, slot, None, None, None, None))
        self._pending[slot] = None
        rows._pending = None
        return rows.hit, rows.ed, rows.end, rows.flags

    # stage / run / fetch, for callers that keep rows resident (bench.py)
    def stage(self, slot, ext, ext_len, bc_off, bc_len):
        ext, el, bo, bl = self._rows(ext, ext_len, bc_off, bc_len)
        n, stride = ext.shape
        check(self._L.bz_assign_stage(self._h, slot, _ptr(ext), max(stride, 1), _ptr(el), _ptr(bo), _ptr(bl), n))
        self._pending[slot] = n

    def run(self, slot, max_ed: int):
        check(self._L.bz_assign_run(self._h, slot, int(max_ed)))

    def sync(self, slot):
        check(self._L.bz_slot_sync(self._h, slot))

    def fetch(self, slot):
        n = self._pending[slot]
        hit = np.empty(n, dtype=np.int32)
        ed = np.empty(n, dtype=np.int8)
        end = np.empty(n, dtype=np.int8)
        flags = np.empty(n, dtype=np.uint8)
        check(self._L.bz_assign_fetch(self._h, slot, _ptr(hit), _ptr(ed), _ptr(end), _ptr(flags)))
        return hit, ed, end, flags

    def timing(self, slot):
        ms = (ctypes.c_float * 8)()
        cnt = (ctypes.c_int64 * 8)()
        check(self._L.bz_slot_timing(self._h, slot, ms, cnt))
        return dict(total_ms=ms[0], classify_ms=ms[1], scan_ms=ms[2], generic_ms=ms[3], hist_ms=ms[4],
                    candidates_ms=ms[5], verify_ms=ms[6],
                    rows=cnt[0], rows_bitparallel=cnt[1], rows_generic=cnt[2], n_whitelist=cnt[3], rows_redo=cnt[4], rows_mixed_fix=cnt[5])

    # -- single pair (the FFI the reference binds) ---------------------------------------
    def sub_edit_distance(self, a, b, max_ed: int):
        a = a if isinstance(a, bytes) else a.encode("latin-1")
        b = b if isinstance(b, bytes) else b.encode("latin-1")
        end = ctypes.c_int(-1)
        ed = check(self._L.bz_sub_edit_distance(self._h, a, len(a), b, len(b), int(max_ed), ctypes.byref(end)))
        return ed, end.value

    def edit_distance(self, a, b, max_ed: int):
        a = a if isinstance(a, bytes) else a.encode("latin-1")
        b = b if isinstance(b, bytes) else b.encode("latin-1")
        return check(self._L.bz_edit_distance(self._h, a, len(a), b, len(b), int(max_ed)))

    # -- counts + collective ---------------------------------------------------------------
    def counts_reset(self):
        check(self._L.bz_counts_reset(self._h))

    def counts(self):
        out = np.zeros(self.n_whitelist + 2, dtype=np.int64)
        check(self._L.bz_counts_get(self._h, out.ctypes.data, out.size))
        return out

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (ctypes.c_uint8 * 128)()
        check(_lib.load().bz_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, uid: bytes, n_ranks: int, rank: int):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(uid)
        check(self._L.bz_comm_init(self._h, buf, int(n_ranks), int(rank)))

    def comm_allreduce_counts(self):
        check(self._L.bz_comm_allreduce_counts(self._h))

    def comm_allreduce_f64(self, values, op: str = "sum"):
        arr = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(self._L.bz_comm_allreduce_f64(self._h, arr.ctypes.data, arr.size, 1 if op == "max" else 0))
        return arr

    def comm_allreduce_i64(self, values):
        arr = np.ascontiguousarray(values, dtype=np.int64).copy()
        if arr.size:
            check(self._L.bz_comm_allreduce_i64(self._h, arr.ctypes.data, arr.size))
        return arr

    def comm_destroy(self):
        check(self._L.bz_comm_destroy(self._h))

    # -- measurement -----------------------------------------------------------------------
    def valu_peak_probe(self, iters: int = 20000):
        ops = ctypes.c_double()
        ms = ctypes.c_float()
        check(self._L.bz_valu_peak_probe(self._h, int(iters), ctypes.byref(ops), ctypes.byref(ms)))
        return ops.value, ms.value


def device_count() -> int:
    n = ctypes.c_int(0)
    check(_lib.load().bz_device_count(ctypes.byref(n)))
    return n.value


def device_identity(device: int = 0) -> str:
    """'<PCI bus id>/<uuid>' of a visible device: equal strings on two ranks = one physical GPU shared."""
    buf = ctypes.create_string_buffer(128)
    check(_lib.load().bz_device_identity(int(device), buf, 128))
    return buf.value.decode()


def device_info(device: int = 0) -> dict:
    info = (ctypes.c_int64 * 8)()
    arch = ctypes.create_string_buffer(64)
    check(_lib.load().bz_device_info(int(device), info, arch, 64))
    return dict(compute_units=info[0], clock_khz=info[1], wavefront=info[2], hbm_mib=info[3], lds_per_cu=info[4],
                arch=arch.value.decode())
