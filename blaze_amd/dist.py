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


def local_world(comm=None) -> int:
    """Ranks of this job on THIS node: $LOCAL_WORLD_SIZE (torchrun and `bench.py --gpus` set it), else the
    communicator's size (ranks run as threads of one process), else $WORLD_SIZE (single-node launchers), else 1."""
    env = os.environ.get("LOCAL_WORLD_SIZE")
    if env:
        return max(1, int(env))
    if comm is not None:
        return max(1, int(comm.world))
    return max(1, int(os.environ.get("WORLD_SIZE", "1")))


def host_threads_per_rank(n_process, comm=None) -> int:
    """The job's --threads (reference default cpu_count - 1, parser.py:64) is a budget for the HOST, as the reference's one
    pool is (helper.py:137: min(n_process - 1, cpu_count - 1) workers): with one process per GPU every rank of the node
    takes its share of it, not all of it -- 8 ranks x (cpu - 1) threads would oversubscribe the node 8 times."""
    budget = max(1, min(int(n_process or 1), os.cpu_count() or 1))
    return max(1, min(budget // local_world(comm), 256))


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


def _rendezvous_key():
    """What the ranks of one job share whatever launched them: (MASTER_ADDR, MASTER_PORT) and, when the launcher
    gives one, its run id.  No pids: ranks need not have a common parent."""
    run = os.environ.get("BLAZE_AMD_RUN_ID") or os.environ.get("TORCHELASTIC_RUN_ID") or ""
    return f"{os.environ.get('MASTER_ADDR', 'local')}_{os.environ.get('MASTER_PORT', '0')}" + (f"_{run}" if run else "")


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous socket")
        buf += chunk
    return bytes(buf)


class Hub:
    """Host-side plumbing between the ranks of one node: rank 0 listens on MASTER_ADDR:(MASTER_PORT + 1)
    (BLAZE_AMD_RDZV_PORT overrides; MASTER_PORT itself belongs to the launcher's store), the others connect.
    If that port is taken rank 0 moves on to the next of PORT_TRIES consecutive ports; a client tells the hub from
    whatever else may listen there by a handshake that carries the job's key (MASTER_ADDR, MASTER_PORT, run id).
    It carries the 128-byte RCCL id, and -- when RCCL cannot be initialised -- the barrier / max / sum of a few scalars.
    Never the data path.  Sockets die with their processes: nothing stale survives a crashed job."""
    PORT_TRIES = 16
    MAGIC = b"BZHUB1"

    def __init__(self, rank, world, addr=None, port=None, timeout_s=120.0):
        import hashlib
        import socket
        import struct
        self.rank, self.world, self._struct = rank, world, struct
        self.peers, self.endpoint = {}, "local"
        if world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("BLAZE_AMD_RDZV_PORT") or int(os.environ.get("MASTER_PORT", "29500")) + 1)
        hello = self.MAGIC + hashlib.sha1(f"{_rendezvous_key()}|{world}".encode()).digest()[:10]
        io_timeout = float(os.environ.get("BLAZE_AMD_HUB_TIMEOUT", "1800"))
        if rank == 0:
            srv = None
            for k in range(self.PORT_TRIES):
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                try:
                    try:
                        srv.bind((addr, port + k))
                    except socket.gaierror:
                        srv.bind(("", port + k))      # MASTER_ADDR is a name that does not resolve here
                    port += k
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError(f"rendezvous hub: none of the ports {port}..{port + self.PORT_TRIES - 1} can be bound")
            self.endpoint = f"{addr}:{port}"
            srv.listen(max(world, 8))
            deadline = time.time() + timeout_s
            while len(self.peers) < world - 1:
                srv.settimeout(max(0.1, deadline - time.time()))
                c, _ = srv.accept()
                try:
                    c.settimeout(5.0)
                    if _recv_exact(c, len(hello)) != hello:
                        raise ConnectionError("not one of this job's ranks")
                    r = struct.unpack("<i", _recv_exact(c, 4))[0]
                    if not 0 < r < world or r in self.peers:
                        raise ConnectionError(f"unexpected rank {r}")
                    c.sendall(b"BZOK")
                except (OSError, ConnectionError):
                    c.close()
                    continue
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(io_timeout)
                self.peers[r] = c
            srv.close()
        else:
            deadline = time.time() + timeout_s
            k = 0
            while True:
                c = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                c.settimeout(5.0)
                try:
                    c.connect((addr, port + k))
                    c.sendall(hello + struct.pack("<i", rank))
                    if _recv_exact(c, 4) != b"BZOK":
                        raise ConnectionError("no hub behind this port")
                    port += k
                    break
                except (OSError, ConnectionError):
                    c.close()
                    if time.time() > deadline:
                        raise TimeoutError(f"rank {rank}: no rendezvous hub at {addr}:{port}..{port + self.PORT_TRIES - 1} after {timeout_s}s")
                    k = (k + 1) % self.PORT_TRIES
                    if k == 0:
                        time.sleep(0.05)
            self.endpoint = f"{addr}:{port}"
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(io_timeout)
            self.peers[0] = c

    def _send(self, sock, payload):
        sock.sendall(self._struct.pack("<q", len(payload)) + payload)

    def _recv(self, sock):
        return _recv_exact(sock, self._struct.unpack("<q", _recv_exact(sock, 8))[0])

    def allgather_bytes(self, payload: bytes):
        """Every rank's payload, in rank order, on every rank."""
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [self._recv(self.peers[r]) for r in range(1, self.world)]
            for r in range(1, self.world):
                for part in parts:
                    self._send(self.peers[r], part)
            return parts
        self._send(self.peers[0], payload)
        return [self._recv(self.peers[0]) for _ in range(self.world)]

    def allgather(self, value):
        import json
        return [json.loads(b.decode()) for b in self.allgather_bytes(json.dumps(value).encode())]

    def close(self):
        for c in self.peers.values():
            try:
                c.close()
            except OSError:
                pass
        self.peers = {}


_hub = None


def get_hub():
    """This process's hub (created on first use from RANK / WORLD_SIZE / MASTER_*)."""
    global _hub
    if _hub is None:
        rank, world, _ = env_rank_world()
        _hub = Hub(rank, world)
    return _hub


def exchange_unique_id(rank: int, make_id, timeout_s: float = 120.0) -> bytes:
    """Single-node rendezvous for the 128-byte RCCL id: through the hub (rank 0 makes it, everyone gets it)."""
    hub = get_hub()
    return hub.allgather_bytes(make_id() if rank == 0 else b"")[0]


class RcclComm:
    """RCCL communicator owned by the engine (C ABI bz_comm_*): one process per GPU."""
    kind = "rccl"

    def __init__(self, engine, rank: int, world: int):
        from .engine import Engine
        self.engine, self.rank, self.world = engine, rank, world
        uid = exchange_unique_id(rank, Engine.comm_unique_id)
        engine.comm_init(uid, world, rank)
        self.barrier()

    def allreduce_counts_on_device(self):
        """In place on the GPU: per-barcode histogram + [assigned, total]."""
        self.engine.comm_allreduce_counts()
        return self.engine.counts()

    def allreduce_sum_i64(self, arr):
        """Exact int64 sum of a host vector of any length (ncclInt64 on the device, one call)."""
        return self.engine.comm_allreduce_i64(arr)

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


# ---------------------------------------------------------------------------------------------- step 3 over files
def partition_files(sizes, world: int):
    """Contiguous runs of the job's FASTQ files, one per rank, balanced by bytes: [(lo, hi)] * world.
    Contiguous so that concatenating the ranks' outputs in rank order keeps the input order (helper.py:191-205's
    guarantee); a rank may get no file when there are fewer files than ranks."""
    n = len(sizes)
    total = float(sum(sizes)) or 1.0
    bounds, acc, k = [0], 0.0, 0
    for r in range(1, world):
        target = total * r / world
        while k < n and acc + sizes[k] / 2.0 <= target:
            acc += sizes[k]
            k += 1
        bounds.append(k)
    bounds.append(n)
    return [(bounds[r], max(bounds[r], bounds[r + 1])) for r in range(world)]


def count_fastq_reads(path) -> int:
    """Reads in a FASTQ(.gz) file as the reference's parser yields them (native line count / 4)."""
    import ctypes
    from . import _lib
    n = ctypes.c_int64(0)
    _lib.check(_lib.load().bz_fastq_count_reads(os.fsencode(str(path)), ctypes.byref(n)))
    return n.value


def part_path(fastq_out, rank: int) -> str:
    """Per-rank piece of the output; keeps the suffix that selects gzip (read_assignment.py:279)."""
    gz = str(fastq_out).endswith('.gz')
    stem = str(fastq_out)[:-3] if gz else str(fastq_out)
    return f"{stem}.part{rank:04d}" + ('.gz' if gz else '')


def first_read_id(path) -> str:
    """Id of the first read of a FASTQ(.gz) file as the reference's parser takes it ('' when the file holds no read)."""
    import ctypes
    from . import _lib
    buf, n = ctypes.create_string_buffer(4096), ctypes.c_int(0)
    _lib.check(_lib.load().bz_fastq_first_read_id(os.fsencode(str(path)), buf, 4096, ctypes.byref(n)))
    return buf.raw[:n.value].decode("latin-1")


def locate_first_row(putative_bc_csv, fastq_fns, lo, hi):
    """Where the rows of the files fastq_fns[lo:hi] start in the (plain) table, WITHOUT counting the reads of the files in
    front of them: the row whose read_id is the id of the first read of the first non-empty file (searched outward from
    where it would be if reads were spread evenly over the FASTQ bytes).  -> byte offset; 0 when those files hold no read
    at all (nothing will be asked of the table); None when the row cannot be found (gzipped table, quoted ids).
    A read id that occurs twice in the job can send a rank to the earlier of its rows; the id check of the file loop
    (read_assignment.py:219-223) then stops the job instead of writing wrong reads, and BLAZE_AMD_CSV_LOCATE=0 selects the
    counting route, which does not depend on ids."""
    import ctypes
    from . import _lib
    rid, first = "", lo
    while first < hi and not rid:
        rid = first_read_id(fastq_fns[first])
        first += 1
    if not rid:
        return 0
    sizes = [os.path.getsize(f) for f in fastq_fns]
    hint = sum(sizes[:first - 1]) / float(sum(sizes) or 1)
    off = ctypes.c_int64(-1)
    enc = rid.encode("latin-1")
    _lib.check(_lib.load().bz_csv_locate_read(os.fsencode(str(putative_bc_csv)), enc, len(enc), hint, ctypes.byref(off)))
    return None if off.value < 0 else int(off.value)


def partition_groups(sizes, world: int, per_rank: int = 4, min_bytes: int | None = None):
    """Contiguous runs of the job's FASTQ files that are dealt to the ranks ROUND ROBIN (group g -> rank g % world):
    up to `per_rank` groups per rank, none smaller than `min_bytes` unless that leaves a rank without one, balanced by
    bytes.  Small groups are what lets a rank put a finished group into the output while the next one is computed
    (Ledger): the offset of a group only needs the groups in front of it, which the other ranks finish at about the same time."""
    n = len(sizes)
    if n == 0:
        return []
    if min_bytes is None:
        min_bytes = int(os.environ.get("BLAZE_AMD_GROUP_MIN_BYTES", str(64 << 20)))
    total = sum(sizes)
    want = max(world, min(world * per_rank, int(total // max(1, min_bytes))))
    return [(lo, hi) for lo, hi in partition_files(sizes, min(n, want)) if hi > lo]


class Ledger:
    """Where the groups of a file-sharded job go in the ONE output file: a small file beside it, mapped by every rank of the
    node, end[g] = offset behind group g (-1 = not known yet).  The rank that finishes group g waits until end[g - 1] is
    known, publishes end[g] = end[g - 1] + its size, and then writes its bytes at end[g - 1] -- the wait is short (the groups
    in front are being finished by the other ranks at the same time) and the bytes of the output are written ONCE, by all
    ranks side by side, while later groups are still being computed.  Single node (a shared file mapping); a rank that fails
    sets the abort flag so that nobody waits for its groups."""

    def __init__(self, path, n_groups, comm):
        import mmap
        self.path, self.n, self.comm = path, n_groups, comm
        size = 8 * (n_groups + 2)
        if comm.rank == 0:
            with open(path, "wb") as f:
                f.write(np.concatenate([[0], np.full(n_groups, -1), [0]]).astype(np.int64).tobytes())
        comm.barrier()
        self._f = open(path, "r+b")
        self._m = mmap.mmap(self._f.fileno(), size)
        self.v = np.frombuffer(self._m, dtype=np.int64)          # v[0] = 0, v[g + 1] = end of group g, v[-1] = abort flag

    def reserve(self, g, size_at, timeout_s=None):
        """Offset of group g once every group in front of it has reserved; publishes this group's end = offset +
        size_at(offset) (the size may depend on the offset: only the first bytes of a table keep their header line)."""
        deadline = time.time() + float(timeout_s if timeout_s is not None else os.environ.get("BLAZE_AMD_HUB_TIMEOUT", "1800"))
        pause = 0.0002
        while self.v[g] < 0:
            if self.v[-1]:
                raise RuntimeError("another rank of the sharded job failed; its groups will never be placed")
            if time.time() > deadline:
                raise TimeoutError(f"group {g}: the groups in front of it were not placed in time")
            time.sleep(pause)
            pause = min(pause * 2, 0.005)
        off = int(self.v[g])
        size = int(size_at(off))
        self.v[g + 1] = off + size
        return off, size

    def abort(self):
        self.v[-1] = 1

    def close(self, remove=False):
        self.v = None
        try:
            self._m.close()
        except BufferError:
            pass
        self._f.close()
        if remove and os.path.exists(self.path):
            os.remove(self.path)


def _place_file(src_path, dst_fd, dst_off, skip=0):
    """All bytes of src_path behind its first `skip` to dst_fd at dst_off, through a large user buffer (pread / pwrite of
    8 MiB pieces moved 12-15 GB/s into one file from 2-8 processes on the box's ext4, copy_file_range 5-7 GB/s:
    profiles/r03b_place_parts.txt), then the part is removed."""
    src = os.open(src_path, os.O_RDONLY)
    try:
        off = skip
        while True:
            buf = os.pread(src, 8 << 20, off)
            if not buf:
                break
            done = 0
            while done < len(buf):
                done += os.pwrite(dst_fd, buf[done:] if done else buf, dst_off + (off - skip) + done)
            off += len(buf)
    finally:
        os.close(src)
    os.remove(src_path)


def run_groups(comm, out_path, groups, run_group, n_sums, drop_header=False, create_empty=True):
    """Group g of files -> rank g % world -> `run_group(g, lo, hi, part)` writes the group's part and returns n_sums
    counts -> the part goes into `out_path` at the offset the Ledger gives it, placed by a helper thread while the rank
    computes its next group.  drop_header: every part starts with the same header line and only the bytes at offset 0 keep
    it (the stage-1 table).  create_empty: the output exists even when no group wrote a byte (step 3: the reference opens
    it before the first batch, read_assignment.py:296,322; the stage-1 table is only made with the first batch,
    main.py:361).  Returns the counts summed over all groups and ranks."""
    import queue
    import threading
    if comm.rank == 0:
        import glob
        gz = str(out_path).endswith('.gz')
        stem = str(out_path)[:-3] if gz else str(out_path)
        for stale in glob.glob(glob.escape(stem) + ".part[0-9][0-9][0-9][0-9]*" + ('.gz' if gz else '')):
            os.remove(stale)                         # parts of an earlier run that died: not this run's bytes
        if os.path.exists(out_path):
            os.remove(out_path)                      # an earlier run's: the placers below create, they never truncate
        if create_empty:
            open(out_path, "wb").close()
    ledger = Ledger(str(out_path) + ".ledger", len(groups), comm)      # (its barrier is also the one behind the lines above)
    todo, errs = queue.Queue(), []

    def placer():
        fd = -1
        try:
            while True:
                item = todo.get()
                if item is None:
                    return
                g, part = item
                if errs:
                    continue
                try:
                    size = os.path.getsize(part) if os.path.exists(part) else 0
                    head = 0
                    if drop_header and size:
                        with open(part, "rb") as f:
                            head = len(f.readline())
                    off, eff = ledger.reserve(g, lambda o: size - (head if o > 0 else 0))
                    if eff > 0:
                        if fd < 0:
                            fd = os.open(out_path, os.O_WRONLY | os.O_CREAT, 0o666)
                        _place_file(part, fd, off, skip=size - eff)
                    elif os.path.exists(part):
                        os.remove(part)
                except BaseException as e:      # noqa: BLE001  (raised by the rank's main thread below)
                    errs.append(e)
                    ledger.abort()
        finally:
            if fd >= 0:
                os.close(fd)
    th = threading.Thread(target=placer, daemon=True)
    th.start()
    sums = np.zeros(n_sums, dtype=np.int64)
    try:
        for g in range(comm.rank, len(groups), comm.world):
            if errs:
                break
            part = part_path(out_path, g)
            if os.path.exists(part):
                os.remove(part)                      # a dead run's: a group without reads may write no part at all
            sums += np.asarray(run_group(g, groups[g][0], groups[g][1], part), dtype=np.int64)
            todo.put((g, part))
    except BaseException:
        ledger.abort()
        todo.put(None)
        th.join()
        ledger.close()
        raise
    todo.put(None)
    th.join()
    if errs:
        ledger.close()
        raise errs[0]
    sums = comm.allreduce_sum_i64(sums)              # also the barrier behind every rank's writes
    ledger.close(remove=comm.rank == 0)
    return sums


def native_run_rank(putative_bc_csv, whitelsit_csv, max_ed, n_process, batchsize, restrand=True, device=None, comm=None):
    """What one rank does with its files in assign_read_sharded: the native device pipeline (bz_assign_files_from) on this
    process's GPU, `csv_skip_rows` rows of the table skipped, on this rank's share of the host threads
    (host_threads_per_rank).  -> run_rank(fns, out, csv_skip_rows) = (assigned, seen)."""
    from . import read_assignment as ra

    def run_rank(fns, out, csv_skip_rows, csv_byte_offset=None, on_mismatch="exit"):
        whitelist = []
        with open(whitelsit_csv, 'r') as f:
            for line in f:
                whitelist.append(line.split('-')[0].strip())
        ordered = list(set(whitelist))
        dev = device if device is not None else ra._device_ids(None)[0]
        eng = ra.get_engine(dev)
        ra._ensure_whitelist(eng, ordered)
        threads = host_threads_per_rank(n_process, comm)
        run_rank.host_threads = threads
        return ra._assign_files_native([eng], fns, putative_bc_csv, out, str(out).endswith('.gz'), restrand, max_ed, batchsize,
                                       ra.DEVICE_BATCH_ROWS, threads, csv_skip_rows=csv_skip_rows or 0, csv_byte_offset=csv_byte_offset,
                                       on_mismatch=on_mismatch)
    run_rank.takes_byte_offset = True
    run_rank.can_raise_mismatch = True
    return run_rank


def assign_read_sharded(comm, fastq_fns, fastq_out, putative_bc_csv, whitelsit_csv, max_ed, n_process, batchsize,
                        restrand=True, device=None, count_reads=count_fastq_reads, run_rank=None):
    """assign_read (read_assignment.py:253-330) with the FASTQ files sharded over the ranks of `comm`, one process
    per GPU.  First choice (plain table): the files are cut into groups dealt round robin (partition_groups), a rank finds
    where a group's rows of putative_bc.csv start from the id of the group's first read (locate_first_row),
    demultiplexes the group into a part, and a helper thread puts the part into `fastq_out` at the offset the Ledger
    gives it while the rank computes its next group: the output's bytes are written once, by all ranks side by side
    (run_groups).  Otherwise (gzipped table, quoted ids): rank r takes ONE contiguous run of files, counts the reads of its
    own files, learns from the other ranks' counts how many rows precede them, writes `part_path(fastq_out, r)`, and after
    one all-reduce of the part sizes every rank copies its part to its offset (place_parts).  Either way gzip members and
    plain text concatenate to the bytes of a single-process run (reference batches never span files,
    read_assignment.py:149-158), the scan uses no collective and the two totals are summed once.
    Returns the GLOBAL (reads assigned, reads seen)."""
    from . import read_assignment as ra
    fastq_fns = [str(f) for f in fastq_fns]
    sizes = [os.path.getsize(f) for f in fastq_fns]
    spans = partition_files(sizes, comm.world)
    lo, hi = spans[comm.rank]
    part = part_path(fastq_out, comm.rank)
    if hi <= lo and os.path.exists(part):
        os.remove(part)                              # left behind by an earlier run that died: not this run's bytes
    if run_rank is None:
        run_rank = native_run_rank(putative_bc_csv, whitelsit_csv, max_ed, n_process, batchsize, restrand, device, comm)
    # Where do this rank's rows of the table start?  First choice: find the row of its first read in the table itself (no
    # FASTQ byte is read twice); every rank must succeed, else all fall back to counting the reads of their own files and
    # one all-reduce of the counts.
    if comm.world == 1 and count_reads is count_fastq_reads:
        return tuple(int(x) for x in run_rank(fastq_fns, fastq_out, 0))       # nothing to shard: straight into the output
    if getattr(run_rank, "takes_byte_offset", False) and count_reads is count_fastq_reads and os.environ.get("BLAZE_AMD_CSV_LOCATE", "1") != "0":
        groups = partition_groups(sizes, comm.world)
        offsets, ok = {}, 1
        for g in range(comm.rank, len(groups), comm.world):
            offsets[g] = locate_first_row(putative_bc_csv, fastq_fns, groups[g][0], groups[g][1])
            if offsets[g] is None:
                ok = 0
                break
        found = comm.allreduce_sum_i64(np.array([ok], dtype=np.int64))
        if int(found[0]) == comm.world:
            def run_group(g, lo, hi, part):
                if not getattr(run_rank, "can_raise_mismatch", False):
                    return run_rank(fastq_fns[lo:hi], part, None, csv_byte_offset=offsets[g])
                try:
                    return run_rank(fastq_fns[lo:hi], part, None, csv_byte_offset=offsets[g], on_mismatch="raise")
                except ra.OrderMismatch:
                    # a read id that occurs twice in the job sent the search to the wrong row: this rank counts the reads in
                    # front of the group instead (its own cost, no collective) -- and if the rows still disagree with the
                    # reads that is the reference's error (message + sys.exit(), read_assignment.py:219-223)
                    skip = sum(count_fastq_reads(f) for f in fastq_fns[:lo])
                    return run_rank(fastq_fns[lo:hi], part, skip)
            sums = run_groups(comm, fastq_out, groups, run_group, 2)
            return int(sums[0]), int(sums[1])
    mine = np.zeros(comm.world, dtype=np.int64)
    # counting gzipped files means inflating them: the files of this rank side by side (the native counter drops the GIL)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max(1, min(host_threads_per_rank(n_process, comm), 16, hi - lo or 1))) as pool:
        mine[comm.rank] = sum(pool.map(count_reads, fastq_fns[lo:hi]))
    per_rank = comm.allreduce_sum_i64(mine)                     # every rank's read count
    skip = int(per_rank[:comm.rank].sum())
    demul, total = run_rank(fastq_fns[lo:hi], part, skip) if hi > lo else (0, 0)
    sums = place_parts(comm, fastq_out, part, extra_sums=(demul, total))
    return int(sums[0]), int(sums[1])


class HubComm:
    """The same four operations over the host-side socket hub: what the ranks fall back to when RCCL cannot be initialised
    on every one of them (`why` says what RCCL reported).  A few scalars and short vectors per job, never the data path."""
    kind = "socket-hub"

    def __init__(self, hub, why=""):
        self.hub, self.rank, self.world, self.why = hub, hub.rank, hub.world, why

    def allreduce_sum_i64(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.int64)
        parts = self.hub.allgather_bytes(a.tobytes())
        return np.sum([np.frombuffer(p, dtype=np.int64) for p in parts], axis=0).reshape(a.shape)

    def allreduce_max_f64(self, x):
        return max(self.hub.allgather(float(x)))

    def barrier(self):
        self.hub.allgather(0)


def make_comm(engine=None, try_rccl=True, init_timeout_s=None):
    """The communicator of this process: local for one rank; RCCL through the engine when launched with WORLD_SIZE > 1
    (one process per GPU); the socket hub when RCCL cannot be initialised on EVERY rank -- all ranks take the same route,
    `comm.kind` / `comm.why` say which and why.  RCCL's init is collective and cannot be cancelled, so it runs on a daemon
    thread with a deadline (BLAZE_AMD_RCCL_INIT_TIMEOUT, 90 s): a bootstrap that never completes costs the deadline, not
    the job (`comm.hung` is then set: exit with os._exit, the thread is still inside RCCL)."""
    rank, world, _ = env_rank_world()
    if world == 1:
        return LocalComm()
    hub = get_hub()
    if not try_rccl:
        comm = HubComm(hub, "RCCL not attempted")
        comm.hung = False
        return comm
    import threading
    box = {}

    def init():
        try:
            eng = engine
            if eng is None:                         # this process's engine: $BLAZE_AMD_DEVICES, else $LOCAL_RANK, else GPU 0
                from . import read_assignment as ra
                eng = ra.get_engine(ra._device_ids(None)[0])
            box["comm"] = RcclComm(eng, rank, world)
        except Exception as e:      # noqa: BLE001  (every rank must learn about it: reported through the hub below)
            box["err"] = e
    t = threading.Thread(target=init, daemon=True)
    t.start()
    t.join(float(init_timeout_s if init_timeout_s is not None else os.environ.get("BLAZE_AMD_RCCL_INIT_TIMEOUT", "90")))
    mine = "rccl" if "comm" in box else (f"{type(box['err']).__name__}: {box['err']}" if "err" in box else "RCCL init did not finish in time")
    states = hub.allgather(mine)
    if all(x == "rccl" for x in states):
        comm = box["comm"]
    else:
        comm = HubComm(hub, next(x for x in states if x != "rccl"))
    comm.hung = "comm" not in box and "err" not in box
    return comm


# ---------------------------------------------------------------------------------------------- stage 1 over files
def _copy_range(src_fd, dst_fd, count, src_off, dst_off):
    """count bytes from src_fd@src_off to dst_fd@dst_off inside the kernel (copy_file_range: no pass through user
    space, reflink where the file system has it); pread / pwrite where the call is not available."""
    # copy_file_range pays off where the file system shares extents (XFS / btrfs reflinks: no byte moves); on ext4 it moved
    # 5-7 GB/s into one file against 12-15 GB/s for large pread / pwrite pieces (profiles/r03b_place_parts.txt): opt in
    use_cfr = hasattr(os, "copy_file_range") and os.environ.get("BLAZE_AMD_COPY_FILE_RANGE", "0") == "1"
    while count > 0:
        if use_cfr:
            try:
                n = os.copy_file_range(src_fd, dst_fd, min(count, 1 << 30), src_off, dst_off)
            except OSError:
                use_cfr = False
                continue
        else:
            buf = os.pread(src_fd, min(count, 8 << 20), src_off)
            n = len(buf)
            done = 0
            while done < n:
                done += os.pwrite(dst_fd, buf[done:], dst_off + done)
        if n <= 0:
            raise OSError("short copy while placing a rank's part")
        count -= n
        src_off += n
        dst_off += n


def place_parts(comm, out_path, my_part, extra_sums=(), drop_header=False, create_empty=True):
    """Every rank wrote `my_part` (or nothing).  Put the parts, in rank order, into `out_path` WITHOUT a second serial
    pass on rank 0: one all-reduce tells every rank all part sizes (it also sums `extra_sums` and is the barrier behind
    the writes), the first non-empty part is renamed into place (it starts at offset 0), and every other rank copies its
    own bytes to its own offset, all ranks at once (copy_file_range).  drop_header: every part starts with the same
    header line and only the first non-empty part keeps it (the stage-1 table).  Returns the summed `extra_sums`."""
    world, rank = comm.world, comm.rank
    size = os.path.getsize(my_part) if os.path.exists(my_part) else 0
    head = 0
    if drop_header and size:
        with open(my_part, 'rb') as f:
            head = len(f.readline())
    vec = np.zeros(2 * world + len(extra_sums), dtype=np.int64)
    vec[rank], vec[world + rank] = size, head
    vec[2 * world:] = extra_sums
    vec = comm.allreduce_sum_i64(vec)
    sizes, heads = vec[:world], vec[world:2 * world]
    nonempty = np.flatnonzero(sizes > 0)
    first = int(nonempty[0]) if nonempty.size else -1
    eff = np.array([sizes[r] - (heads[r] if drop_header and r != first else 0) for r in range(world)], dtype=np.int64)
    if first < 0:
        if rank == 0 and create_empty:
            open(out_path, 'wb').close()
        if os.path.exists(my_part):
            os.remove(my_part)
        comm.barrier()
        return vec[2 * world:]
    if rank == first:
        os.replace(my_part, out_path)
    comm.barrier()                                   # the file exists under its final name from here on
    if rank != first:
        if eff[rank] > 0:
            src, dst = os.open(my_part, os.O_RDONLY), os.open(out_path, os.O_WRONLY)
            try:
                _copy_range(src, dst, int(eff[rank]), int(sizes[rank] - eff[rank]), int(eff[:rank].sum()))
            finally:
                os.close(src)
                os.close(dst)
        if os.path.exists(my_part):
            os.remove(my_part)
    comm.barrier()
    return vec[2 * world:]


def putative_bc_files_sharded(comm, fastq_fns, out_csv, kit, umi_len, min_q, batch_size, n_threads, device=None, run_rank=None):
    """The stage-1 loop (main.py:344-364) with the FASTQ files sharded over the ranks of `comm`, one process per GPU:
    rank r searches its contiguous run of files into `part_path(out_csv, r)`, the pass-code counts are summed once, and
    every rank places its own part (one header kept) at its offset of `out_csv` (place_parts): the bytes of a single-process run, because
    reference batches never span files (main.py:311-323).  `n_threads` is the HOST's budget (--threads): every rank takes its
    share (host_threads_per_rank).  Returns the GLOBAL {pass code: reads}."""
    from . import putative_bc, read_assignment as ra
    fastq_fns = [str(f) for f in fastq_fns]
    if run_rank is None:
        def run_rank(fns, out):
            eng = ra.get_engine(device if device is not None else ra._device_ids(None)[0])
            return putative_bc.putative_bc_files([eng], fns, out, kit=kit, umi_len=umi_len, min_q=min_q, batch_size=batch_size,
                                                 n_threads=host_threads_per_rank(n_threads, comm))
    codes = (0, 1, 2, 10, 100)
    groups = partition_groups([os.path.getsize(f) for f in fastq_fns], comm.world)

    def run_group(g, lo, hi, part):
        counts = run_rank(fastq_fns[lo:hi], part)
        return [counts.get(c, 0) for c in codes]
    # no reads at all: the reference never creates the table (main.py:361)
    sums = run_groups(comm, out_csv, groups, run_group, len(codes), drop_header=True, create_empty=False)
    return {c: int(v) for c, v in zip(codes, sums) if v}
