#!/usr/bin/env python3
"""How fast do the ranks of a file-sharded job put their parts into ONE output file?  (No GPU needed.)

Measures, for world sizes 2 / 4 / 8 and a fixed number of output bytes in total, with real processes:
  place_parts   blaze_amd.dist.place_parts as the sharded jobs call it: the first part renamed, every other rank copies its
                own part to its own offset (copy_file_range), all ranks at once
  direct_pwrite every rank writes its bytes from memory straight to its offset of the final file (what a job whose ranks
                knew their offsets before writing would do: the bytes are written once)
  one_stream    one process writes all the bytes to one file (what a single-process job does)
  parts_only    every rank writes its own part file (distinct inodes: the first half of place_parts' cost)
Usage: python tools/place_parts_bench.py [--total-gib 8] [--dir DIR] [--worlds 2,4,8]
"""
import argparse
import json
import multiprocessing as mp
import os
import shutil
import socket
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BLOCK = 8 << 20


def _fill(path, n):
    buf = os.urandom(1 << 20) * 8
    with open(path, "wb", buffering=0) as f:
        left = n
        while left > 0:
            f.write(buf[:min(BLOCK, left)])
            left -= BLOCK


def _rank(mode, rank, world, port, d, part_bytes, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from blaze_amd import dist
    comm = dist.HubComm(dist.get_hub(), "bench")
    out = os.path.join(d, "final.out")
    part = dist.part_path(out, rank)
    buf = os.urandom(1 << 20) * 8
    if mode == "place_parts":
        _fill(part, part_bytes)
    elif mode == "direct_pwrite" and rank == 0:
        open(out, "wb").close()
    comm.barrier()
    t0 = time.perf_counter()
    if mode == "place_parts":
        dist.place_parts(comm, out, part)
    elif mode == "parts_only":
        _fill(part, part_bytes)
        comm.barrier()
    elif mode == "direct_pwrite":
        fd = os.open(out, os.O_WRONLY)
        off, left = rank * part_bytes, part_bytes
        while left > 0:
            n = os.pwrite(fd, buf[:min(BLOCK, left)], off)
            off += n
            left -= n
        os.close(fd)
        comm.barrier()
    dt = time.perf_counter() - t0
    dt = comm.allreduce_max_f64(dt)
    if rank == 0:
        size = os.path.getsize(out) if os.path.exists(out) else 0
        q.put((dt, size))
    comm.barrier()


def run(mode, world, total, d):
    work = tempfile.mkdtemp(dir=d)
    try:
        if mode == "one_stream":
            t0 = time.perf_counter()
            _fill(os.path.join(work, "final.out"), total)
            return time.perf_counter() - t0, total
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = mp.Queue()
        ps = [mp.Process(target=_rank, args=(mode, r, world, port, work, total // world, q)) for r in range(world)]
        for p in ps:
            p.start()
        dt, size = q.get(timeout=1200)
        for p in ps:
            p.join(60)
        return dt, size
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--total-gib", type=float, default=8.0)
    ap.add_argument("--dir", default=os.environ.get("TMPDIR", "/tmp"))
    ap.add_argument("--worlds", default="2,4,8")
    a = ap.parse_args()
    total = int(a.total_gib * (1 << 30))
    free = shutil.disk_usage(a.dir).free
    if free < 2.5 * total:
        total = int(free / 2.5) // (8 << 20) * (8 << 20)
    fs = os.popen(f"stat -f -c %T {a.dir}").read().strip()
    res = {"dir": a.dir, "fs": fs, "total_bytes": total, "kernel": os.uname().release, "cpus": os.cpu_count(), "runs": []}
    dt, _ = run("one_stream", 1, total, a.dir)
    res["runs"].append({"mode": "one_stream", "world": 1, "s": dt, "GB_per_s": total / dt / 1e9})
    print(res["runs"][-1], flush=True)
    for world in [int(x) for x in a.worlds.split(",")]:
        for mode in ("parts_only", "place_parts", "direct_pwrite"):
            dt, size = run(mode, world, total, a.dir)
            moved = total if mode != "place_parts" else total * (world - 1) // world
            res["runs"].append({"mode": mode, "world": world, "s": dt, "GB_per_s": moved / dt / 1e9, "bytes_moved": moved,
                                "final_size_ok": (size == total // world * world) if mode != "parts_only" else None})
            print(res["runs"][-1], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    main()
