"""Samples of BASELINE configs[2], [3], [4] against the REAL reference (oracle/make_golden_configs_samples.py): the first
rows of the very workloads tests/test_gpu_full_size.py builds went through the real `_match_bc_row`, and a 30,000-read
file job in configs[4]'s shape (10-nt UMI, 10,000 cells, trim + restrand) through the real `assign_read` with its process
pool.  CPU half: the oracle's C driver on configs[2] / [4].  GPU half: the HIP path on all three row samples (configs[3]:
the full 6,794,880-barcode list, k_pair_stream) and the native host on the file job (sha256 of the output text)."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from blaze_amd import _lib, synth
from blaze_amd import read_assignment as ra

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SAMPLES = {   # as oracle/make_golden_configs_samples.py: rows, cells, umi_len, seed, rows of chunk 0, full list size, whitelist = full list
    "configs2": (30_000, 20_000, 10, 3, 500_000, 400_000, False),
    "configs4": (30_000, 10_000, 10, 11, 1_000_000, 400_000, False),
    "configs3": (600, 5_000, 12, 3, 500_000, synth.FULL_WHITELIST_SIZE, True),
}


def _check(name, assign):
    g = np.load(os.path.join(GOLD, "configs_samples_leftmost.npz"))
    if f"{name}_hit" not in g:
        pytest.skip(f"{name}: sample not recorded")
    n, n_cells, umi_len, seed, chunk, full_size, use_full = SAMPLES[name]
    full = synth.full_whitelist_codes(full_size, seed=1)
    cells = synth.cell_whitelist_codes(n_cells, full, seed=2)
    rows = synth.make_rows(chunk, cells, full, umi_len=umi_len, seed=seed)
    ext, el, bo, bl = (a[:n] for a in synth.pack_ext(rows))
    wl_ascii = synth.codes_to_bytes(full if use_full else cells)
    hit, _, end, flags = assign(ext, el, bo, bl, wl_ascii)
    bad = np.flatnonzero(hit != g[f"{name}_hit"])
    assert bad.size == 0, (name, bad[:5], hit[bad[:5]], g[f"{name}_hit"][bad[:5]])
    s = lambda m, i: m[i].tobytes().decode() if not rows["blank"][i] else ""
    bcs, umis, posts = ([s(rows[k], i) for i in range(n)] for k in ("bc", "umi", "post"))
    polyT = [int(rows["polyT_end"][i]) if not rows["blank"][i] else "" for i in range(n)]

    class Names:                                   # _decode_rows only indexes the list: no 6.8M Python strings
        def __getitem__(self, k):
            return wl_ascii[k].tobytes().decode()
    got = ra._decode_rows(hit, end, flags, el, bcs, umis, posts, polyT, Names())
    text = "".join(f"{b},{u},{st}\n" for b, u, st in got)
    assert hashlib.sha256(text.encode()).hexdigest() == str(g[f"{name}_sha256"]), name
    return int((hit >= 0).sum())


@pytest.mark.parametrize("name", ["configs2", "configs4"])
def test_oracle_on_config_samples(oracle, name):
    def assign(ext, el, bo, bl, wl_ascii):
        return oracle.match_rows(ext, el, bo, bl, (np.ascontiguousarray(wl_ascii).reshape(-1), np.arange(wl_ascii.shape[0] + 1, dtype=np.int64) * 16),
                                 2, threads=8)
    assert _check(name, assign) > 20_000


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", ["configs2", "configs4", "configs3"])
def test_hip_on_config_samples(engine, name):
    def assign(ext, el, bo, bl, wl_ascii):
        engine.set_whitelist_codes16(wl_ascii)
        return engine.assign(ext, el, bo, bl, 2)
    assert _check(name, assign) > (300 if name == "configs3" else 20_000)


@pytest.mark.gpu
@pytest.mark.parametrize("threads,device_batch_rows", [(8, 1 << 16), (3, 5000)])
def test_native_host_on_the_configs4_file_job(tmp_path, monkeypatch, threads, device_batch_rows):
    with open(os.path.join(GOLD, "configs4_files_leftmost.json")) as f:
        case = json.load(f)
    job = case["job"]
    fns, pc, wl, _ = synth.write_file_job(str(tmp_path), job["n"], job["n_cells"], umi_len=job["umi_len"], n_files=job["n_files"],
                                          seed=job["seed"], median_len=job["median_len"])
    order = list(case["iteration_order"])           # the reference's workers iterated set(whitelist) under PYTHONHASHSEED=0

    def recorded(items):
        assert set(items) == set(order)
        return order
    monkeypatch.setattr(ra, "set", recorded, raising=False)
    for name, exp in case["outputs"].items():
        po = tmp_path / name
        demul, total = ra.assign_read(fastq_fns=fns, fastq_out=str(po), putative_bc_csv=pc, whitelsit_csv=wl, max_ed=2, n_process=threads,
                                      batchsize=1000, restrand=exp["restrand"], host="native", device_batch_rows=device_batch_rows)
        raw = po.read_bytes()
        if name.endswith(".gz"):
            raw = gzip.decompress(raw)
        assert (int(demul), int(total), len(raw)) == (exp["demul"], exp["total"], exp["bytes"]), name
        assert hashlib.sha256(raw).hexdigest() == exp["sha256"], name
