"""CPU, world_size 2 over gloo: the read-sharded N>1 path.  The scan itself cannot run without a
GPU, so the oracle stands in for the engine here (tests only); what is under test is the
sharding, the order-preserving concatenation and the counts all-reduce."""
import os
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


def test_file_rendezvous_roundtrip(tmp_path, monkeypatch):
    from blaze_amd import dist
    monkeypatch.setenv("BLAZE_AMD_RDZV_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "12345")
    uid = bytes(range(128))
    assert dist.exchange_unique_id(0, lambda: uid) == uid
    assert dist.exchange_unique_id(1, None, timeout_s=5) == uid
