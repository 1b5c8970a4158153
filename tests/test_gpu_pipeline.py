"""GPU parity of the whole three-stage pipeline (blaze_amd.main.main) against files written by the REAL
reference's blaze.main.main on the same FASTQ files and arguments (oracle/make_golden_pipeline.py ->
tests/golden/pipeline_leftmost.json).  Bar: every output file byte for byte (the .gz output decompressed, F9)."""
import gzip
import json
import os
from types import SimpleNamespace

import pytest

from blaze_amd import main as bmain

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "pipeline_leftmost.json")


def _cases():
    with open(GOLD) as f:
        return json.load(f)["cases"]


def _materialise(case, tmp):
    """Write the recorded inputs under tmp and rebuild the recorded parse_arg namespace with tmp's paths."""
    def sub(v):
        if isinstance(v, str):
            return float("inf") if v == "inf" else v.replace("<TMP>", tmp)
        if isinstance(v, list):
            return [sub(x) for x in v]
        return v
    args = SimpleNamespace(**{k: sub(v) for k, v in case["args"].items()})
    os.makedirs(os.path.join(tmp, "fq"))
    os.makedirs(os.path.join(tmp, "out"))
    for fn, text in case["input_fastq"].items():
        path = os.path.join(tmp, "fq", fn)
        if fn.endswith(".gz"):
            with gzip.open(path, "wt") as h:
                h.write(text)
        else:
            with open(path, "w") as h:
                h.write(text)
    with open(args.full_bc_whitelist, "w") as h:
        h.write("".join(b + "\n" for b in case["full_whitelist"]))
    assert sorted(os.path.basename(f) for f in args.fastq_fns) == sorted(case["input_fastq"])
    return args


def _read_outputs(args):
    outs = {}
    for key in ("out_raw_bc_fn", "out_whitelist_fn", "out_emptydrop_fn", "summary_fn"):
        with open(getattr(args, key)) as h:
            outs[key] = h.read()
    opener = gzip.open if args.output_fastq.endswith(".gz") else open
    with opener(args.output_fastq, "rt") as h:
        outs["output_fastq"] = h.read()
    return outs


@pytest.mark.parametrize("host", ["python", "native"])
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_pipeline_files_match_the_reference(tmp_path, case, host):
    args = _materialise(case, str(tmp_path))
    bmain.main(args, host=host)
    got = _read_outputs(args)
    for key, expect in case["expect"].items():
        assert got[key] == expect, f"{case['name']} / {host}: {key} differs"
    with open(args.out_plot_fn, "rb") as h:                     # the knee plot of stage 2 (main.py:286-302)
        assert h.read(8) == b"\x89PNG\r\n\x1a\n"


def test_stage_gating_follows_the_namespace(tmp_path):
    """do_bc_search / do_whitelisting / do_demultiplexing select the stages (parser.py:219-340 sets them)."""
    case = _cases()[0]
    args = _materialise(case, str(tmp_path))
    args.do_whitelisting = args.do_demultiplexing = False
    bmain.main(args)
    assert os.path.exists(args.out_raw_bc_fn) and not os.path.exists(args.out_whitelist_fn)
    args.do_bc_search, args.do_whitelisting = False, True
    bmain.main(args)
    assert open(args.out_whitelist_fn).read() == case["expect"]["out_whitelist_fn"]
    assert not os.path.exists(args.output_fastq)
    with pytest.raises(TypeError):
        bmain.main("--expect-cells 10 reads/")


TWO_RANK_WORKER = '''
import json, os, sys
sys.path.insert(0, %r)
sys.path.insert(0, os.path.join(%r, "tests"))
import torch.distributed as td
from blaze_amd import dist, main as bmain
import test_gpu_pipeline as T
td.init_process_group("gloo", init_method="env://")
comm = dist.TorchComm()                      # host-side sums only; the data path has no collective
case = [c for c in T._cases() if c["name"] == sys.argv[2]][0]
args = T._namespace(case, sys.argv[1])
bmain.main(args, devices=[0], comm=comm)     # both ranks share the box's one GPU
comm.barrier()
print("rank", comm.rank, "ok")
'''


def _namespace(case, tmp):
    def sub(v):
        if isinstance(v, str):
            return float("inf") if v == "inf" else v.replace("<TMP>", tmp)
        if isinstance(v, list):
            return [sub(x) for x in v]
        return v
    return SimpleNamespace(**{k: sub(v) for k, v in case["args"].items()})


@pytest.mark.parametrize("name", ["3v3_expect_cells_gz", "5v2_count_threshold_plain_norestrand"])
def test_pipeline_two_processes_shard_the_files(tmp_path, name):
    """One process per rank (two ranks on the box's one GPU, gloo for the host-side sums): stages 1 and 3 shard the
    FASTQ files, stage 2 is rank 0's; every file on disk equals what the REAL reference wrote in one process."""
    import socket
    import subprocess
    import sys
    case = [c for c in _cases() if c["name"] == name][0]
    args = _materialise(case, str(tmp_path))
    script = tmp_path / "two_rank_worker.py"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script.write_text(TWO_RANK_WORKER % (root, root))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path), name], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert f"rank {rank} ok" in out
    got = _read_outputs(args)
    for key, expect in case["expect"].items():
        assert got[key] == expect, f"{name}: {key} differs"
    assert not [f for f in os.listdir(tmp_path / "out") if ".part" in f]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_multi_rank_main_writes_the_reference_files(tmp_path, monkeypatch, case, world):
    """main(args, comm=...) with one process per GPU: stages 1 and 3 sharded by FASTQ file, stage 2 on rank 0.  Ranks are
    threads of this process here (one GPU: the two native file loops are serialised, everything around them -- partition,
    parts, the all-reduces, every rank placing its part, the barriers around stage 2 -- runs as it would in processes);
    the five files equal what the REAL reference wrote."""
    import threading
    from test_gpu_native_host import _ThreadComm
    from blaze_amd import putative_bc, read_assignment
    args = _materialise(case, str(tmp_path))
    device_lock = threading.Lock()
    real_s1, real_s3 = putative_bc.putative_bc_files, read_assignment._assign_files_native

    def locked(fn):
        def call(*a, **k):
            with device_lock:
                return fn(*a, **k)
        return call
    monkeypatch.setattr(putative_bc, "putative_bc_files", locked(real_s1))
    monkeypatch.setattr(read_assignment, "_assign_files_native", locked(real_s3))
    monkeypatch.setattr(read_assignment, "_ensure_whitelist", locked(read_assignment._ensure_whitelist))
    read_assignment.get_engine(read_assignment._device_ids(None)[0])          # made before the threads start
    shared = {"bar": threading.Barrier(world), "lock": threading.Lock()}
    errs = []

    def rank_main(rank):
        try:
            bmain.main(args, comm=_ThreadComm(rank, world, shared))
        except BaseException as e:      # noqa: BLE001
            errs.append(e)
            shared["bar"].abort()
    ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errs, errs
    got = _read_outputs(args)
    for key, expect in case["expect"].items():
        assert got[key] == expect, f"{case['name']} / {world} ranks: {key} differs"
    assert not [f for f in os.listdir(os.path.dirname(args.output_fastq)) if ".part" in f]


def test_configs0_at_its_stated_size(tmp_path):
    """BASELINE configs[0] at the size it names: 29 FASTQ files x 4,000 reads, 3v3 kit, --expect-cells 500, all three stages.
    The fixture (oracle/make_golden_configs0.py) holds what the REAL reference's main() wrote for the same inputs as size,
    line count and sha256 per output file (the FASTQ output decompressed); the inputs are the deterministic stand-ins of
    synth.write_pipeline_job (the bundled files are missing from the reference), regenerated here and checked by hash."""
    import hashlib
    from blaze_amd import synth
    with open(os.path.join(os.path.dirname(GOLD), "pipeline_configs0.json")) as f:
        case = json.load(f)["case"]
    tmp = str(tmp_path)
    fq_dir, full_fn, in_sha = synth.write_pipeline_job(tmp, **case["job"])
    assert in_sha == case["inputs_sha256"], "the generator no longer writes the inputs the reference was run on"

    def sub(v):
        if isinstance(v, str):
            return float("inf") if v == "inf" else v.replace("<TMP>", tmp)
        return [sub(x) for x in v] if isinstance(v, list) else v
    args = SimpleNamespace(**{k: sub(v) for k, v in case["args"].items()})
    assert args.full_bc_whitelist == full_fn and len(args.fastq_fns) == 29 and os.path.dirname(args.fastq_fns[0]) == fq_dir
    os.makedirs(os.path.join(tmp, "out"))
    bmain.main(args, host="native")
    for key, exp in case["expect"].items():
        raw = open(getattr(args, key), "rb").read()
        if key == "output_fastq" and args.output_fastq.endswith(".gz"):
            raw = gzip.decompress(raw)
        got = dict(bytes=len(raw), lines=raw.count(b"\n"), sha256=hashlib.sha256(raw).hexdigest())
        assert got == exp, f"configs[0]: {key} differs from the reference's file: {got} != {exp}"
    assert "Total reads: 116000" in open(args.summary_fn).read()
