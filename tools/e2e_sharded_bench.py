"""Step 3 over files with the FASTQ files SHARDED over rank processes (blaze_amd.dist.assign_read_sharded: one process per
GPU on a node; here every rank uses the box's one GPU, which RCCL refuses, so the ranks talk over the socket hub).
Same total of host threads for every world size: what the part files + the parallel placement cost, and that the bytes
are those of the one-process run.

usage: python tools/e2e_sharded_bench.py [n_reads] [n_cells] [total_threads] [--gz]
"""
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(d, gz, threads):
    from blaze_amd import dist
    comm = dist.make_comm(init_timeout_s=60)
    fns = sorted((os.path.join(d, f) for f in os.listdir(d) if f.startswith("reads_") and f.endswith(".fastq")),
                 key=lambda f: int(os.path.basename(f)[6:-6]))         # the order the table was written in
    out = os.path.join(d, f"sharded_w{comm.world}.fastq" + (".gz" if gz else ""))
    best = None
    for rep in range(3):
        if comm.rank == 0 and os.path.exists(out):
            os.remove(out)                          # dropping the last repetition's 1.7 GB is not part of a run
        comm.barrier()
        t0 = time.perf_counter()
        got = dist.assign_read_sharded(comm, fns, out, os.path.join(d, "putative_bc.csv"), os.path.join(d, "whitelist.csv"), 2, threads, 1000)
        dt = comm.allreduce_max_f64(time.perf_counter() - t0)
        best = dt if best is None else min(best, dt)
    if comm.rank == 0:
        import gzip
        raw = open(out, "rb").read()
        text = gzip.decompress(raw) if gz else raw
        print(json.dumps({"world": comm.world, "threads_per_rank": dist.host_threads_per_rank(threads, comm), "comm": comm.kind, "why": getattr(comm, "why", ""), "s": best,
                          "assigned": int(got[0]), "total": int(got[1]), "sha1_of_text": hashlib.sha1(text).hexdigest(), "bytes_out": len(raw)}), flush=True)
    comm.barrier()
    if getattr(comm, "hung", False):
        os._exit(0)


def main():
    if "--worker" in sys.argv:
        k = sys.argv.index("--worker")
        return worker(sys.argv[k + 1], sys.argv[k + 2] == "1", int(sys.argv[k + 3]))
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if len(args) > 0 else 1_000_000
    n_cells = int(args[1]) if len(args) > 1 else 10_000
    total_threads = int(args[2]) if len(args) > 2 else 16
    gz = "--gz" in sys.argv
    from blaze_amd import synth
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        synth.write_file_job(d, n, n_cells, n_files=int(os.environ.get('N_FILES', '32')))
        out = {"reads": n, "cells": n_cells, "gz": gz, "runs": []}
        for world in [int(w) for w in os.environ.get('WORLDS', '1,2,4,8').split(',')]:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            procs = []
            for rank in range(world):
                env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                           PYTHONHASHSEED="0")       # the same set() order of the whitelist in every process: comparable bytes
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", d, "1" if gz else "0",
                                               str(total_threads)], env=env     # the HOST's budget: the library gives every rank its share
                                              , stdout=subprocess.PIPE, text=True))
            lines = [p.communicate(timeout=600)[0] for p in procs]
            if any(p.returncode for p in procs):
                raise SystemExit(f"world {world}: rank exit codes {[p.returncode for p in procs]}")
            out["runs"].append(json.loads(lines[0].strip().splitlines()[-1]))
        out["same_text_for_every_world"] = len({r["sha1_of_text"] for r in out["runs"]}) == 1
        print(json.dumps(out))


if __name__ == "__main__":
    main()
