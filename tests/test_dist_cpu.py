"""CPU, world_size 2 over gloo: the read-sharded N>1 path.  The scan itself cannot run without a
GPU, so the oracle stands in for the engine here (tests only); what is under test is the
sharding, the order-preserving concatenation and the counts all-reduce."""
import os

import pytest
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as td
    from blaze_amd import dist, synth
    from oracle import oracle as o
    td.init_process_group("gloo", init_method="env://")
    comm = dist.TorchComm()
    full = synth.full_whitelist_codes(20_000, seed=1)
    cells = synth.cell_whitelist_codes(97, full, seed=2)
    rows = synth.make_rows(1001, cells, full, sub=0.02, ins=0.01, dele=0.01, seed=3)
    ext, el, bo, bl = synth.pack_ext(rows)
    wl = [w.tobytes() for w in synth.codes_to_bytes(cells)]
    fn = lambda *a: o.match_rows(a[0], a[1], a[2], a[3], wl, a[4])
    lo, hi, res, counts = dist.assign_sharded(ext, el, bo, bl, 2, fn, comm, len(wl))
    whole = o.match_rows(ext, el, bo, bl, wl, 2)
    assert (lo, hi) == dist.shard_range(1001, comm.rank, comm.world)
    for a, b in zip(res, whole):
        assert np.array_equal(a, b[lo:hi])                      # rank order == input order
    ok = whole[0] >= 0
    assert np.array_equal(counts[:97], np.bincount(whole[0][ok], minlength=97))
    assert counts[97] == ok.sum() and counts[98] == 1001
    t = comm.allreduce_max_f64(1.0 + comm.rank)
    assert t == float(comm.world)
    comm.barrier()
    print("rank", comm.rank, "ok")
""") % ROOT


FILE_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as td
    from blaze_amd import dist
    td.init_process_group("gloo", init_method="env://")
    comm = dist.TorchComm()
    d = sys.argv[1]
    fns = [os.path.join(d, "f%%d.fastq" %% k) for k in range(5)]
    seen = {}
    def count(path):                               # stands in for the native line counter (no library call needed, but it works on CPU too)
        return dist.count_fastq_reads(path)
    def run_rank(files, out, csv_skip_rows):       # stands in for the device pipeline: records what the rank was asked to do
        seen["files"], seen["skip"] = [os.path.basename(f) for f in files], csv_skip_rows
        n = 0
        with open(out, "w") as o:
            for f in files:
                for line in open(f):
                    if line.startswith("@"):
                        o.write(line); n += 1
        return n // 2, n
    demul, total = dist.assign_read_sharded(comm, fns, os.path.join(d, "out.fastq"), "unused.csv", "unused.txt", 2, 1, 1000,
                                            count_reads=count, run_rank=run_rank)
    # stage 1 the same way: every part starts with the header, the merged table has one
    def run_s1(files, out):
        n = 0
        with open(out, "w") as o:
            o.write("read_id,putative_bc\\n")
            for f in files:
                for line in open(f):
                    if line.startswith("@"):
                        o.write(line[1:].strip() + ",\\n"); n += 1
        return {0: n - 1, 1: 1}
    got = dist.putative_bc_files_sharded(comm, fns, os.path.join(d, "table.csv"), "3v3", 12, 15, 1000, 1, run_rank=run_s1)
    reads = [3, 1000, 7, 2000, 11]                 # per file (f1 and f3 are the big ones)
    assert got == {0: sum(reads) - 2, 1: 2}, got
    if comm.rank == 0:
        rows = open(os.path.join(d, "table.csv")).read().split()
        assert rows == ["read_id,putative_bc"] + ["f%%d_%%d," %% (k, i) for k in range(5) for i in range(reads[k])]
    assert total == sum(reads), total
    if comm.rank == 0:
        assert seen["files"] == ["f0.fastq", "f1.fastq", "f2.fastq"] and seen["skip"] == 0, seen
        titles = open(os.path.join(d, "out.fastq")).read().split()
        assert titles == ["@f%%d_%%d" %% (k, i) for k in range(5) for i in range(reads[k])]      # input order kept
        assert not [f for f in os.listdir(d) if ".part" in f]
    else:
        assert seen["files"] == ["f3.fastq", "f4.fastq"] and seen["skip"] == 1010, seen
    print("rank", comm.rank, "ok")
""") % ROOT


def _launch_two(script, *argv):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), *argv], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out


def test_files_sharded_over_two_ranks_gloo(tmp_path):
    """assign_read_sharded: contiguous file runs per rank, csv rows to skip from the other rank's read count,
    parts merged in rank order, totals summed (the device pipeline itself is stubbed: no GPU here)."""
    reads = [3, 1000, 7, 2000, 11]
    for k, n in enumerate(reads):
        with open(tmp_path / f"f{k}.fastq", "w") as f:
            for i in range(n):
                f.write(f"@f{k}_{i}\nACGT\n+\nIIII\n")
    script = tmp_path / "file_worker.py"
    script.write_text(FILE_WORKER)
    _launch_two(script, str(tmp_path))


def test_partition_files():
    from blaze_amd import dist
    for sizes, world in (([10, 10, 10, 10], 2), ([1, 100, 1, 100, 1], 2), ([5], 4), ([], 3), ([3, 3, 3, 3, 3, 3, 3, 3, 3], 8),
                         ([0, 0, 0], 2), ([1000, 1, 1, 1], 3)):
        spans = dist.partition_files(sizes, world)
        assert len(spans) == world and spans[0][0] == 0 and spans[-1][1] == len(sizes)
        for (a, b), (c, d) in zip(spans, spans[1:]):
            assert a <= b == c <= d                                # contiguous, ordered, nothing dropped
    assert dist.partition_files([10, 10, 10, 10], 2) == [(0, 2), (2, 4)]
    assert dist.part_path("x/out.fastq.gz", 3) == "x/out.fastq.part0003.gz" and dist.part_path("o.fq", 0) == "o.fq.part0000"


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out


def test_rendezvous_key_has_no_pids(monkeypatch):
    from blaze_amd import dist
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "12345")
    key = dist._rendezvous_key()
    assert "12345" in key and str(os.getppid()) not in key and str(os.getpid()) not in key      # ranks need not share a parent


def test_hub_allgather_three_ranks():
    """The socket hub that carries the RCCL id (and the scalars when RCCL is unavailable): rank order, bytes and JSON."""
    import threading
    from blaze_amd import dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    got = {}

    def rank_main(r):
        hub = dist.Hub(r, 3, addr="127.0.0.1", port=port, timeout_s=30)
        a = hub.allgather_bytes(bytes([r]) * (r + 1))
        b = hub.allgather({"rank": r, "t": 1.5 * r})
        c = hub.allgather_bytes(bytes(range(128)) if r == 0 else b"")[0]        # the id exchange
        got[r] = (a, b, c)
        hub.close()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in (2, 0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    for r in range(3):
        a, b, c = got[r]
        assert a == [b"\x00", b"\x01\x01", b"\x02\x02\x02"]
        assert b == [{"rank": k, "t": 1.5 * k} for k in range(3)]
        assert c == bytes(range(128))


def test_hub_moves_past_a_taken_port_and_ignores_strangers(monkeypatch):
    """MASTER_PORT + 1 already belongs to something else: rank 0 binds the next port, the clients find it by the
    handshake (the foreign listener never answers it), and a stranger that connects to the hub is dropped."""
    import threading
    from blaze_amd import dist
    monkeypatch.setenv("MASTER_PORT", "0")
    foreign = socket.socket()
    foreign.bind(("127.0.0.1", 0))
    foreign.listen(4)
    port = foreign.getsockname()[1]
    got = {}

    def rank_main(r):
        hub = dist.Hub(r, 2, addr="127.0.0.1", port=port, timeout_s=40)
        got[r] = (hub.endpoint, hub.allgather(r * 10))
        hub.close()
    t0 = threading.Thread(target=rank_main, args=(0,))
    t0.start()
    import time
    time.sleep(0.3)
    stranger = socket.socket()
    try:
        stranger.connect(("127.0.0.1", port + 1))
        stranger.sendall(b"GET / HTTP/1.0\r\n\r\n and some more bytes")
    except OSError:
        pass
    t1 = threading.Thread(target=rank_main, args=(1,))
    t1.start()
    t0.join(60), t1.join(60)
    stranger.close(), foreign.close()
    assert got[0][1] == [0, 10] and got[1][1] == [0, 10]
    assert got[0][0] == got[1][0] and not got[0][0].endswith(f":{port}")


def _run_bench(args, env_extra, timeout=300):
    env = dict(os.environ, BLAZE_AMD_BENCH_ENGINE="tests.stub_engine:StubEngine", PYTHONPATH=ROOT, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher: two rank processes, one JSON line, whole-job value, comm named.
    (Stand-in engine: this checks the launch / rendezvous / reduction logic, which needs no GPU.)"""
    import json
    p = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rows", "4096", "--cells", "64"], {})
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_gpus_requested"] == 2 and out["steps"] == 3
    assert out["rows_counted"] == 2 * 3 * 4096                  # both ranks' steps were counted: the sum went over the ranks
    assert "socket-hub" in out["config"]["comm"] and out["config"]["launcher"] == "bench.py --gpus"
    assert abs(out["value"] - 2 * 4096 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert "STAND-IN" in out["data"]


def test_bench_refuses_world_size_mismatch():
    p = _run_bench(["--gpus", "8", "--steps", "1", "--warmup", "0", "--rows", "1024", "--cells", "16"],
                   dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29511"))
    assert p.returncode == 2 and "WORLD_SIZE=1" in p.stderr and not p.stdout.strip()


class _ThreadComm:
    """world ranks as threads of this process: enough to drive dist.place_parts without a launcher."""

    def __init__(self, rank, world, shared):
        self.rank, self.world, self._s = rank, world, shared

    def barrier(self):
        self._s["bar"].wait()

    def allreduce_sum_i64(self, arr):
        import numpy as np
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


def _place(tmp_path, parts, **kw):
    import threading
    from blaze_amd import dist
    world = len(parts)
    shared = {"bar": threading.Barrier(world), "lock": threading.Lock()}
    out = str(tmp_path / "final.out")
    res = {}

    def rank_main(r):
        part = dist.part_path(out, r)
        if parts[r] is not None:
            with open(part, "wb") as f:
                f.write(parts[r])
        res[r] = dist.place_parts(_ThreadComm(r, world, shared), out, part, extra_sums=(r, 10), **kw)
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    assert all(list(res[r]) == [sum(range(world)), 10 * world] for r in range(world))
    assert not [f for f in os.listdir(tmp_path) if ".part" in f]
    return out


def test_place_parts_offsets_headers_and_empty_ranks(tmp_path):
    big = bytes(range(256)) * 5000                               # 1.28 MB: more than one copy call's worth of nothing special
    out = _place(tmp_path, [b"aaa\n", big, b"", None, b"zz"])
    assert open(out, "rb").read() == b"aaa\n" + big + b"zz"
    # stage-1 table: every part starts with the header, the first NON-EMPTY part keeps it (rank 0 has no reads here)
    h = b"read_id,putative_bc\n"
    out = _place(tmp_path, [None, h + b"r1,\nr2,\n", None, h + b"r3,\n", h], drop_header=True, create_empty=False)
    assert open(out, "rb").read() == h + b"r1,\nr2,\nr3,\n"
    os.remove(out)
    # nothing anywhere: step 3 leaves an empty file (the reference opens its output first), stage 1 leaves none
    out = _place(tmp_path, [None, b"", None])
    assert os.path.getsize(out) == 0
    os.remove(out)
    out = _place(tmp_path, [None, None], drop_header=True, create_empty=False)
    assert not os.path.exists(out)


def test_locate_first_row_of_a_ranks_files(tmp_path):
    """A rank finds where its rows of putative_bc.csv start from the id of its first read (host only): every partition
    of the files, empty files in front, the first and the last row, a gzipped table / an unknown id -> None."""
    import gzip
    import numpy as np
    from blaze_amd import dist
    rng = np.random.default_rng(3)
    reads = [0, 700, 0, 1500, 1, 0, 3000, 40]
    fns, rows, starts, k = [], ["read_id,putative_bc,putative_bc_min_q,putative_umi,polyT_end,pre_bc_flanking,post_umi_flanking\n"], [], 0
    for f, n in enumerate(reads):
        fn = tmp_path / (f"f{f}.fastq" + (".gz" if f == 3 else ""))
        body = "".join(f"@read-{f}-{i}-{int(rng.integers(1 << 30)):x} runid=x\nACGT\n+\nIIII\n" for i in range(n))
        if f == 3:
            with gzip.open(fn, "wt") as g:
                g.write(body)
        else:
            fn.write_text(body)
        fns.append(str(fn))
        starts.append(len("".join(rows)))
        rows += [line[1:].split()[0] + ",ACGTACGTACGTACGT,15,ACGTACGTACGT,300,ACGTA,TTTTT\n" for line in body.splitlines()[::4]]
    csv = tmp_path / "putative_bc.csv"
    csv.write_text("".join(rows))
    assert dist.first_read_id(fns[0]) == "" and dist.first_read_id(fns[1]).startswith("read-1-0-")
    assert dist.first_read_id(fns[3]).startswith("read-3-0-")                      # gzipped file
    for world in (1, 2, 3, 5, 8):
        for lo, hi in dist.partition_files([os.path.getsize(f) for f in fns], world):
            off = dist.locate_first_row(str(csv), fns, lo, hi)
            first = next((f for f in range(lo, hi) if reads[f]), None)
            assert off == (0 if first is None else starts[first]), (world, lo, hi)
    # the hint may be far off: files whose bytes say nothing about their reads
    assert dist.locate_first_row(str(csv), fns, 7, 8) == starts[7] and dist.locate_first_row(str(csv), fns, 0, 2) == starts[1]
    with gzip.open(tmp_path / "t.csv.gz", "wt") as g:
        g.write("".join(rows))
    assert dist.locate_first_row(str(tmp_path / "t.csv.gz"), fns, 1, 2) is None   # not seekable: the caller counts instead
    (tmp_path / "other.fastq").write_text("@nobody\nA\n+\nI\n")
    assert dist.locate_first_row(str(csv), [str(tmp_path / "other.fastq")], 0, 1) is None


def test_sharded_runs_ignore_parts_left_by_a_dead_run(tmp_path):
    """A rank with no file of its own must not contribute a `.partNNNN` file that an earlier, crashed run left behind
    (step 3 and the stage-1 table)."""
    import threading
    from blaze_amd import dist
    fq = tmp_path / "only.fastq"
    fq.write_text("@r0\nACGT\n+\nIIII\n")
    world = 3
    for out_name, sharded in (("out.fastq", "step3"), ("putative_bc.csv", "stage1")):
        out = str(tmp_path / out_name)
        for r in range(world):
            with open(dist.part_path(out, r), "wb") as f:
                f.write(b"STALE BYTES OF A DEAD RUN\n")
        shared = {"bar": threading.Barrier(world), "lock": threading.Lock()}
        res = {}

        def rank_main(r):
            comm = _ThreadComm(r, world, shared)
            if sharded == "step3":
                def run_rank(fns, part, skip):
                    with open(part, "wb") as f:
                        f.write(b"fresh\n")
                    return 1, 1
                res[r] = dist.assign_read_sharded(comm, [str(fq)], out, "unused.csv", "unused", 2, 1, 1000,
                                                  count_reads=lambda p: 1, run_rank=run_rank)
            else:
                def run_rank(fns, part):
                    with open(part, "wb") as f:
                        f.write(b"header\nfresh\n")
                    return {0: 1}
                res[r] = dist.putative_bc_files_sharded(comm, [str(fq)], out, "3v3", 12, 15, 1000, 1, run_rank=run_rank)
        ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(60)
        assert len(res) == world
        assert open(out, "rb").read() == (b"fresh\n" if sharded == "step3" else b"header\nfresh\n")
        assert not [f for f in os.listdir(tmp_path) if ".part" in f]


COMM_WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np
    from blaze_amd import dist

    class NoRccl:                                   # an engine whose RCCL refuses (two ranks on one device, no library, ...)
        @staticmethod
        def comm_unique_id():
            return bytes(128)
        def comm_init(self, uid, world, rank):
            if sys.argv[1] == "all" or rank == int(sys.argv[1]):
                raise RuntimeError("ncclCommInitRank: invalid usage")
        def comm_allreduce_f64(self, v, op):        # (RcclComm's own barrier right after a successful init)
            return v

    dist.RcclComm.__init__.__globals__["Engine"] = NoRccl          # RcclComm asks the engine class for the id
    import blaze_amd.engine as eng_mod
    eng_mod.Engine = NoRccl
    comm = dist.make_comm(NoRccl(), init_timeout_s=20)
    assert comm.kind == "socket-hub" and "invalid usage" in comm.why and not comm.hung, (comm.kind, getattr(comm, "why", ""))
    v = comm.allreduce_sum_i64(np.arange(5, dtype=np.int64) * (comm.rank + 1) + (1 << 60))
    assert v.tolist() == [int(k * 3 + (1 << 61)) for k in range(5)], v            # exact above 2^53
    assert comm.allreduce_max_f64(1.5 + comm.rank) == 2.5
    comm.barrier()
    print("rank", comm.rank, "ok")
""") % ROOT


@pytest.mark.parametrize("who", ["1", "all"])
def test_make_comm_falls_back_to_the_hub_on_every_rank(tmp_path, who):
    """dist.make_comm: when RCCL cannot be initialised on one rank (or on all), EVERY rank takes the socket hub -- same
    sums (exact int64), max and barrier -- and says why."""
    script = tmp_path / "comm_worker.py"
    script.write_text(COMM_WORKER)
    _launch_two(script, who)


def test_bench_counts_distinct_gpus_not_ranks():
    """Two ranks that hold the SAME physical GPU (identities all-gathered over the hub) are not a 2-GPU point: refused
    with status 2 on every rank, or -- declared a rehearsal -- printed with n_gpus = 1."""
    import json
    args = ["--gpus", "2", "--steps", "2", "--warmup", "0", "--rows", "2048", "--cells", "32"]
    p = _run_bench(args, dict(STUB_GPU_OF_RANK="0000:05:00.0/aa,0000:05:00.0/aa"))
    assert p.returncode != 0 and "1 distinct GPU" in p.stderr and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    p = _run_bench(args + ["--allow-shared-gpu"], dict(STUB_GPU_OF_RANK="0000:05:00.0/aa,0000:05:00.0/aa"))
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["n_ranks"] == 2 and out["n_gpus_requested"] == 2
    assert out["config"]["shared_gpu_rehearsal"] is True and out["config"]["ranks_per_gpu"] == 2
    # two different GPUs behind the same local index (a launcher that narrows every rank's view): a real 2-GPU point
    p = _run_bench(args, dict(STUB_GPU_OF_RANK="0000:05:00.0/aa,0000:26:00.0/bb"))
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["shared_gpu_rehearsal"] is False and len(out["config"]["gpu_of_rank"]) == 2


def test_bench_under_torch_distributed_run_two_ranks():
    """The launch form the driver uses for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`), two ranks, stand-in engine: ONE JSON line from rank 0,
    whole-job value, the launcher named, the hub beside the launcher's own store on MASTER_PORT."""
    import json
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BLAZE_AMD_BENCH_ENGINE="tests.stub_engine:StubEngine", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--rows", "4096", "--cells", "64"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks"] == 2 and out["n_gpus_requested"] == 2 and out["scaling"] == "weak"
    assert out["rows_counted"] == 2 * 3 * 4096 and out["config"]["launcher"].startswith("torchrun")
    assert abs(out["value"] - 2 * 4096 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]


def test_host_threads_are_split_over_the_ranks_of_a_node(monkeypatch):
    from blaze_amd import dist
    cpus = os.cpu_count() or 1

    class C:
        world = 4
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert dist.host_threads_per_rank(cpus) == cpus                         # one process: the whole budget
    assert dist.host_threads_per_rank(cpus, C()) == max(1, cpus // 4)        # ranks as threads of one process
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert dist.host_threads_per_rank(10_000) == max(1, cpus // 8)          # the reference's default cpu-1, 8 ranks per node
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")                             # 2 nodes x ... : only this node's ranks share its cores
    assert dist.host_threads_per_rank(10_000, C()) == max(1, cpus // 2)
    assert dist.host_threads_per_rank(1) == 1 and dist.host_threads_per_rank(None) == 1
    # and that is what the sharded file jobs hand to the native pipelines
    seen = {}
    from blaze_amd import putative_bc, read_assignment as ra
    monkeypatch.setattr(putative_bc, "putative_bc_files", lambda engines, fns, out, **kw: seen.update(kw) or {})
    monkeypatch.setattr(ra, "get_engine", lambda d: None)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    f = os.path.join(ROOT, "bench.py")
    dist.putative_bc_files_sharded(dist.LocalComm(), [f], "/tmp/never_written.csv", "3v3", 12, 15, 1000, 16, device=0)
    assert seen["n_threads"] == max(1, min(16, cpus) // 4)
    monkeypatch.setattr(ra, "_ensure_whitelist", lambda e, o: None)
    monkeypatch.setattr(ra, "_assign_files_native", lambda engines, fns, csv, out, gz, restrand, max_ed, bs, dbr, threads, **kw: (threads, 0))
    wl = os.path.join(ROOT, "BASELINE.md")
    run = dist.native_run_rank("x.csv", wl, 2, 16, 1000, device=0, comm=None)
    assert run([f], "/tmp/never_written.fastq", 0)[0] == max(1, min(16, cpus) // 4)


def test_rank0_failure_in_stage_2_releases_the_other_ranks(tmp_path):
    """Rank 0 alone runs stage 2; when it raises there (the reference's own FileNotFoundError on a job without reads) the
    other ranks must leave the collective they wait in, not hang."""
    import threading
    import types
    from blaze_amd import main as bmain
    shared = {"bar": threading.Barrier(2, timeout=60), "lock": threading.Lock()}
    args = types.SimpleNamespace(do_bc_search=False, do_whitelisting=True, do_demultiplexing=True, minQ=15,
                                 out_raw_bc_fn=str(tmp_path / "no_such_table.csv"), fastq_fns=[])
    res = {}

    def rank_main(r):
        try:
            bmain.main(args, devices=[0], comm=_ThreadComm(r, 2, shared))
            res[r] = "returned"
        except BaseException as e:      # noqa: BLE001
            res[r] = type(e).__name__
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(90)
    assert not any(t.is_alive() for t in ts)
    assert res[0] == "FileNotFoundError" and res[1] == "SystemExit", res


def _grouped_job(tmp_path, reads):
    """FASTQ files with `reads[k]` reads each and their table; -> (fns, csv path, byte offset of every file's first row)."""
    fns, rows, starts = [], ["read_id,putative_bc,putative_bc_min_q,putative_umi,polyT_end,pre_bc_flanking,post_umi_flanking\n"], []
    for f, n in enumerate(reads):
        fn = tmp_path / f"g{f}.fastq"
        fn.write_text("".join(f"@read-{f}-{i} runid=x\nACGT\n+\nIIII\n" for i in range(n)))
        fns.append(str(fn))
        starts.append(len("".join(rows)))
        rows += [f"read-{f}-{i},ACGTACGTACGTACGT,15,ACGTACGTACGT,300,ACGTA,TTTTT\n" for i in range(n)]
    csv = tmp_path / "putative_bc.csv"
    csv.write_text("".join(rows))
    return fns, str(csv), starts


def _ranks_as_threads(world, body):
    import threading
    shared = {"bar": threading.Barrier(world, timeout=60), "lock": threading.Lock()}
    res, errs = {}, {}

    def rank_main(r):
        try:
            res[r] = body(_ThreadComm(r, world, shared))
        except BaseException as e:      # noqa: BLE001
            errs[r] = e
            shared["bar"].abort()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not any(t.is_alive() for t in ts)
    return res, errs


@pytest.mark.parametrize("world", [2, 3, 5])
def test_groups_round_robin_are_placed_in_file_order(tmp_path, monkeypatch, world):
    """The located-row route: groups of files dealt round robin, every group's part placed at the offset the ledger gives
    it by a helper thread -- the output is the files' outputs in file order, whatever order the groups finish in."""
    import time
    from blaze_amd import dist
    monkeypatch.setenv("BLAZE_AMD_GROUP_MIN_BYTES", "1")            # tiny files: still several groups per rank
    reads = [5, 0, 700, 30, 0, 0, 1500, 1, 260, 40, 900, 3, 77]
    fns, csv, starts = _grouped_job(tmp_path, reads)
    out = str(tmp_path / "out.fastq")
    seen = []

    def body(comm):
        def run_rank(files, part, skip, csv_byte_offset=None):
            first = next((f for f in files if os.path.getsize(f)), None)
            assert skip is None and csv_byte_offset == (starts[fns.index(first)] if first else 0)
            time.sleep(0.002 * ((len(files) * 7 + comm.rank) % 5))     # groups finish out of order
            n = 0
            with open(part, "wb") as o:
                for f in files:
                    for line in open(f, "rb"):
                        if line.startswith(b"@"):
                            o.write(line)
                            n += 1
            seen.append((comm.rank, [os.path.basename(f) for f in files]))
            return n // 2, n
        run_rank.takes_byte_offset = True
        return dist.assign_read_sharded(comm, fns, out, csv, "unused", 2, 4, 1000, run_rank=run_rank)
    res, errs = _ranks_as_threads(world, body)
    assert not errs, errs
    expect = b"".join(f"@read-{f}-{i} runid=x\n".encode() for f, n in enumerate(reads) for i in range(n))
    assert open(out, "rb").read() == expect
    total = sum(reads)
    assert all(res[r] == res[0] and res[r][1] == total for r in range(world))
    groups = dist.partition_groups([os.path.getsize(f) for f in fns], world)
    assert world < len(groups) <= 4 * world                                                         # several groups per rank
    assert sorted(x for _, fl in seen for x in fl) == sorted(os.path.basename(f) for f in fns)      # every file once
    assert {r for r, _ in seen} == set(range(min(world, len(groups))))                              # dealt over the ranks
    assert not [f for f in os.listdir(tmp_path) if ".part" in f or f.endswith(".ledger")]


def test_a_failing_rank_releases_the_ranks_that_wait_for_its_groups(tmp_path, monkeypatch):
    from blaze_amd import dist
    monkeypatch.setenv("BLAZE_AMD_GROUP_MIN_BYTES", "1")
    fns, csv, _ = _grouped_job(tmp_path, [50, 60, 70, 80, 90, 100])
    out = str(tmp_path / "out.fastq")

    def body(comm):
        def run_rank(files, part, skip, csv_byte_offset=None):
            if comm.rank == 0:
                raise ValueError("rank 0 cannot read its file")
            with open(part, "wb") as o:
                o.write(b"x" * 10)
            return 1, 1
        run_rank.takes_byte_offset = True
        return dist.assign_read_sharded(comm, fns, out, csv, "unused", 2, 4, 1000, run_rank=run_rank)
    res, errs = _ranks_as_threads(2, body)
    assert isinstance(errs[0], ValueError) and 1 in errs and not res, (res, errs)
