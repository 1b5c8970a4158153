"""Read sharding across GPUs (one process per GPU) and the single collective of step 3.

The reference shards work across a process pool and sums two scalars in the parent
(/root/reference/blaze/read_assignment.py:299-301,324-326; helper.py:111-205).  Here reads
are split in CONTIGUOUS ranges per rank (so concatenating rank outputs in rank order keeps the
input order, the guarantee of helper.py:191-205), the whitelist is replicated, the scan uses no
collective at all, and the per-barcode histogram + [assigned, total] are summed once at the end:
RCCL over xGMI through the C ABI on GPUs, torch.distributed (gloo) in the CPU tests.
"""
from __future__ import annotations

import os
import time

import numpy as np


def shard_range(n: int, rank: int, world: int):
    """Contiguous [lo, hi) of `n` rows for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


class LocalComm:
    """world_size 1."""
    rank, world = 0, 1
    kind = "local"

    def allreduce_sum_i64(self, arr):
        return np.asarray(arr, dtype=np.int64).copy()

    def allreduce_max_f64(self, x):
        return float(x)

    def barrier(self):
        pass


class TorchComm:
    """torch.distributed process group (gloo on CPU); plumbing for tests, never the data path."""
    kind = "torch"

    def __init__(self):
        import torch.distributed as td
        self._td = td
        self.rank, self.world = td.get_rank(), td.get_world_size()

    def allreduce_sum_i64(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64).copy())
        self._td.all_reduce(t, op=self._td.ReduceOp.SUM)
        return t.numpy()

    def allreduce_max_f64(self, x):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64)
        self._td.all_reduce(t, op=self._td.ReduceOp.MAX)
        return float(t[0])

    def barrier(self):
        self._td.barrier()


def _rendezvous_path():
    key = f"{os.environ.get('MASTER_ADDR', 'local')}_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}"
    return os.path.join(os.environ.get("BLAZE_AMD_RDZV_DIR", "/tmp"), f"blaze_amd_rccl_{key}.id")


def exchange_unique_id(rank: int, make_id, timeout_s: float = 120.0) -> bytes:
    """Single-node rendezvous for the 128-byte RCCL id: rank 0 writes it to a file named after
    (MASTER_ADDR, MASTER_PORT, parent pid = the launcher), the other ranks poll for a fresh one."""
    path = _rendezvous_path()
    if rank == 0:
        uid = make_id()
        tmp = path + f".tmp{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)
        return uid
    start = time.time()
    while True:
        try:
            st = os.stat(path)
            if st.st_size == 128 and st.st_mtime >= start - 600:
                with open(path, "rb") as f:
                    return f.read()
        except FileNotFoundError:
            pass
        if time.time() - start > timeout_s:
            raise TimeoutError(f"no RCCL id at {path} after {timeout_s}s")
        time.sleep(0.02)


class RcclComm:
    """RCCL communicator owned by the engine (C ABI bz_comm_*): one process per GPU."""
    kind = "rccl"

    def __init__(self, engine, rank: int, world: int):
        from .engine import Engine
        self.engine, self.rank, self.world = engine, rank, world
        uid = exchange_unique_id(rank, Engine.comm_unique_id)
        engine.comm_init(uid, world, rank)
        self.barrier()
        if rank == 0:
            try:
                os.remove(_rendezvous_path())
            except OSError:
                pass

    def allreduce_counts_on_device(self):
        """In place on the GPU: per-barcode histogram + [assigned, total]."""
        self.engine.comm_allreduce_counts()
        return self.engine.counts()

    def allreduce_sum_i64(self, arr):
        a = np.asarray(arr, dtype=np.int64)
        out = np.empty_like(a)
        flat, res = a.reshape(-1), out.reshape(-1)
        for lo in range(0, flat.size, 64):          # small host vectors only; exact below 2^53
            res[lo:lo + 64] = self.engine.comm_allreduce_f64(flat[lo:lo + 64].astype(np.float64), "sum").astype(np.int64)
        return out

    def allreduce_max_f64(self, x):
        return float(self.engine.comm_allreduce_f64([float(x)], "max")[0])

    def barrier(self):
        self.engine.comm_allreduce_f64([0.0], "sum")


def assign_sharded(ext, ext_len, bc_off, bc_len, max_ed, assign_fn, comm, n_whitelist):
    """Score this rank's contiguous shard with `assign_fn(ext, ext_len, bc_off, bc_len, max_ed) ->
    (hit, ed, end, flags)` and sum the per-barcode counts + [assigned, total] over ranks.
    Returns (lo, hi, local results, global counts[n_whitelist + 2])."""
    n = ext.shape[0]
    lo, hi = shard_range(n, comm.rank, comm.world)
    res = assign_fn(ext[lo:hi], ext_len[lo:hi], bc_off[lo:hi], bc_len[lo:hi], max_ed)
    hit = res[0]
    counts = np.zeros(n_whitelist + 2, dtype=np.int64)
    ok = hit >= 0
    counts[:n_whitelist] = np.bincount(hit[ok], minlength=n_whitelist)
    counts[n_whitelist] = int(ok.sum())
    counts[n_whitelist + 1] = hi - lo
    return lo, hi, res, comm.allreduce_sum_i64(counts)
