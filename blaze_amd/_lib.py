"""ctypes binding of libblaze_hip.so (include/blaze_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There
is no fallback of any kind: a missing library raises at first use, and a library
without a usable GPU raises ``BlazeHipError`` from ``bz_create``.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libblaze_hip.so")

BZ_OPT_TIE_RULE, BZ_OPT_SHORTCUT, BZ_OPT_FORCE_GENERIC, BZ_OPT_SCAN_MODE, BZ_OPT_PAIR_FUSED, BZ_OPT_PROFILE, BZ_OPT_GZ_LEVEL = 1, 2, 3, 4, 5, 6, 7
BZ_F_EXACT, BZ_F_BLANK, BZ_F_AMBIG, BZ_F_GENERIC, BZ_F_INVALID = 1, 2, 4, 8, 16
BZ_NOALLOC_MAX_ROWS = 262144
BZ_EXT_MAX = 64
BZ_E_MISMATCH = -6

EXPORTS = [
    "bz_abi_version", "bz_last_error", "bz_device_count", "bz_device_info", "bz_create", "bz_destroy",
    "bz_set_option", "bz_set_whitelist", "bz_assign_submit", "bz_assign_wait", "bz_assign_stage",
    "bz_assign_run", "bz_assign_fetch", "bz_slot_sync", "bz_slot_timing", "bz_sub_edit_distance",
    "bz_edit_distance", "bz_counts_reset", "bz_counts_get", "bz_comm_unique_id", "bz_comm_init",
    "bz_comm_allreduce_counts", "bz_comm_allreduce_f64", "bz_comm_destroy", "bz_valu_peak_probe", "bz_assign_files", "bz_far_from_all", "bz_putative_bc",
    "bz_putative_bc_files", "bz_assign_files_from", "bz_fastq_count_reads", "bz_assign_device_noalloc", "bz_comm_allreduce_i64", "bz_host_alloc", "bz_host_free", "bz_assign_submit_pinned", "bz_gzip_fastq_bound", "bz_gzip_fastq_member", "bz_fastq_first_read_id", "bz_csv_locate_read", "bz_assign_files_at", "bz_device_identity",
]


def source_sha256() -> str:
    """Hash of the sources libblaze_hip.so is built from (blaze_amd/csrc/*, include/blaze_hip.h): what a committed
    rocprof counter file is tied to -- bench.py joins its instruction counts with live timings only while this matches."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*"))) + [os.path.join(os.path.dirname(_HERE), "include", "blaze_hip.h")]
    for fn in files:
        if os.path.isfile(fn):
            h.update(os.path.basename(fn).encode() + b"\0")
            with open(fn, "rb") as f:
                h.update(f.read())
    return h.hexdigest()


class BlazeHipError(RuntimeError):
    """A negative status from the C ABI (map to RuntimeError)."""


class BlazeFileNotFoundError(BlazeHipError, FileNotFoundError):
    """The native file loops could not open an input or the output: also a FileNotFoundError, which is what the
    reference's own `open()` calls raise for the same paths (read_assignment.py:149-153, :279; main.py:311-315)."""


class BlazePermissionError(BlazeHipError, PermissionError):
    """EACCES / EPERM from the native open(): the PermissionError the reference's `open()` raises."""


class BlazeIsADirectoryError(BlazeHipError, IsADirectoryError):
    """A directory where a file was expected: the IsADirectoryError the reference's `open()` raises."""


class BlazeOSError(BlazeHipError, OSError):
    """Any other errno of a native open() / mmap()."""


def _os_error(rc: int, text: str):
    """The native open / map failures carry '[errno N: ...]' (host_io.h: open_error): raise what Python's own open() would."""
    import errno as _errno
    import re
    m = re.search(r"\[errno (\d+): ", text)
    if not m:
        return None
    code = int(m.group(1))
    cls = {_errno.ENOENT: BlazeFileNotFoundError, _errno.ENOTDIR: BlazeFileNotFoundError, _errno.EACCES: BlazePermissionError,
           _errno.EPERM: BlazePermissionError, _errno.EISDIR: BlazeIsADirectoryError}.get(code, BlazeOSError)
    e = cls(f"libblaze_hip status {rc}: {text}")
    e.errno, e.strerror = code, f"libblaze_hip status {rc}: {text}"          # (RuntimeError comes first in the MRO: OSError's own constructor does not fill these)
    return e


_lib = None


def load():
    """dlopen the in-tree library and declare every prototype.  Raises loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BlazeHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "blaze_amd has no CPU path.")
    # One HSA knob, the same on every launch route (torchrun, `bench.py --gpus N`, a plain import): this pool's host
    # driver shares device buffers between processes through dmabuf only, and RCCL's intra-node P2P transport needs
    # that sharing (without it: `hipIpcGetMemHandle: invalid argument` at the first collective).  Set before the HIP
    # runtime initialises (it reads the variable once); a value the user exported wins.  INTEGRATION.md section 4.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    L = ctypes.CDLL(LIB_PATH)
    i, i64, p, cp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_char_p
    pi = ctypes.POINTER(ctypes.c_int)
    sig = {
        "bz_abi_version": ([], i),
        "bz_last_error": ([], cp),
        "bz_device_count": ([pi], i),
        "bz_device_info": ([i, p, p, i], i),
        "bz_device_identity": ([i, p, i], i),
        "bz_create": ([i, ctypes.POINTER(p)], i),
        "bz_destroy": ([p], i),
        "bz_set_option": ([p, i, i64], i),
        "bz_set_whitelist": ([p, p, p, i64], i),
        "bz_assign_submit": ([p, i, p, i, p, p, p, i64, i], i),
        "bz_assign_wait": ([p, i, p, p, p, p], i),
        "bz_assign_stage": ([p, i, p, i, p, p, p, i64], i),
        "bz_assign_run": ([p, i, i], i),
        "bz_assign_fetch": ([p, i, p, p, p, p], i),
        "bz_slot_sync": ([p, i], i),
        "bz_slot_timing": ([p, i, p, p], i),
        "bz_sub_edit_distance": ([p, cp, i, cp, i, i, pi], i),
        "bz_edit_distance": ([p, cp, i, cp, i, i], i),
        "bz_counts_reset": ([p], i),
        "bz_counts_get": ([p, p, i64], i),
        "bz_comm_unique_id": ([p], i),
        "bz_comm_init": ([p, p, i, i], i),
        "bz_comm_allreduce_counts": ([p], i),
        "bz_comm_allreduce_f64": ([p, p, i, i], i),
        "bz_comm_allreduce_i64": ([p, p, i64], i),
        "bz_host_alloc": ([ctypes.POINTER(p), i64], i),
        "bz_host_free": ([p], i),
        "bz_gzip_fastq_bound": ([i64], i64),
        "bz_gzip_fastq_member": ([cp, i64, p, i64, ctypes.POINTER(i64)], i),
        "bz_assign_submit_pinned": ([p, i, p, i, p, p, p, i64, i, p, p, p, p], i),
        "bz_comm_destroy": ([p], i),
        "bz_valu_peak_probe": ([p, i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_float)], i),
        "bz_putative_bc": ([p, p, p, p, i64, i, i, p, p, p, p, p], i),
        "bz_far_from_all": ([p, p, p, i64, p, p, i64, i, p], i),
        "bz_putative_bc_files": ([ctypes.POINTER(p), i, ctypes.POINTER(cp), i, cp, i, i, i, i, i64, i, ctypes.POINTER(i64)], i),
        "bz_assign_files_from": ([ctypes.POINTER(p), i, ctypes.POINTER(cp), i, cp, i64, cp, i, i, i, i, i64, i, ctypes.POINTER(i64)], i),
        "bz_fastq_count_reads": ([cp, ctypes.POINTER(i64)], i),
        "bz_fastq_first_read_id": ([cp, p, i, pi], i),
        "bz_csv_locate_read": ([cp, cp, i, ctypes.c_double, ctypes.POINTER(i64)], i),
        "bz_assign_files_at": ([ctypes.POINTER(p), i, ctypes.POINTER(cp), i, cp, i64, cp, i, i, i, i, i64, i, ctypes.POINTER(i64)], i),
        "bz_assign_device_noalloc": ([p, p, i, p, p, p, i64, i, p, p, p, p, p], i),
        "bz_assign_files": ([ctypes.POINTER(p), i, ctypes.POINTER(cp), i, cp, cp, i, i, i, i, i64, i, ctypes.POINTER(i64)], i),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)       # AttributeError if the .so lacks a declared symbol
        fn.argtypes, fn.restype = args, res
    _lib = L
    return L


def check(rc: int) -> int:
    if rc < 0:
        msg = load().bz_last_error()
        text = msg.decode(errors='replace') if msg else ''
        err = _os_error(rc, text)
        if err is not None:
            raise err
        raise BlazeHipError(f"libblaze_hip status {rc}: {text}")
    return rc
