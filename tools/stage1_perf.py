"""SURVEY 8f N3 measurement: bz_putative_bc over synthetic 10x 3v3 reads vs the oracle's per-read Python restatement."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from blaze_amd import putative_bc
from blaze_amd.engine import Engine
from oracle import stage1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", np.uint8)
comp = bytes.maketrans(b"ACGT", b"TGCA")
lens = np.maximum(200, rng.lognormal(np.log(900), 0.6, n).astype(np.int64))
pool = acgt[rng.integers(0, 4, int(lens.max()) + (1 << 20))].tobytes()
starts = rng.integers(0, 1 << 20, n)
seqs = []
for i in range(n):
    body = pool[starts[i]:starts[i] + int(lens[i])]
    head = b"GCAATG" + b"CTACACGACGCTCTTCCGATCT" + pool[starts[i] + 7:starts[i] + 7 + 28] + b"T" * 25
    s = head + body
    if i % 2:
        s = s.translate(comp)[::-1]
    seqs.append(s.decode())
quals = ["I" * len(s) for s in seqs]
total = sum(len(s) for s in seqs)
eng = Engine(0)
putative_bc.search(eng, seqs[:1000], quals[:1000], "3v3", 12)
t0 = time.time(); code, strand, start, minq, polyt = putative_bc.search(eng, seqs, quals, "3v3", 12); dt = time.time() - t0
print(f"device (python packing + H2D + kernel + D2H): {n} reads, {total / 1e6:.0f} MB in {dt:.3f} s = {n / dt / 1e3:.0f} K reads/s; found {int((code == 0).sum())}")
from blaze_amd import _lib
enc = [s.encode() for s in seqs]; off = np.zeros(n + 1, np.int64); np.cumsum([len(s) for s in enc], out=off[1:])
sb = np.frombuffer(b"".join(enc), np.uint8); qb = np.full(sb.size, ord("I"), np.uint8)
outs = [np.zeros(n, np.int32) for _ in range(4)]; st = np.zeros(n, np.int8)
L = _lib.load()
for rep in range(2):
    t0 = time.time()
    _lib.check(L.bz_putative_bc(eng._h, sb.ctypes.data, qb.ctypes.data, off.ctypes.data, n, 0, 12, outs[0].ctypes.data, st.ctypes.data, outs[1].ctypes.data, outs[2].ctypes.data, outs[3].ctypes.data))
    dt = time.time() - t0
print(f"C ABI call on packed arrays (H2D + kernel + D2H): {dt * 1e3:.1f} ms = {n / dt / 1e6:.2f} M reads/s, {total / dt / 1e9:.2f} GB/s of read bytes")
m = 2000
t0 = time.time()
for i in range(m):
    stage1.read_stage1(seqs[i], quals[i], "3v3", 12)
dc = time.time() - t0
print(f"oracle restatement (1 thread, Python + C DP): {m / dc:.0f} reads/s")
