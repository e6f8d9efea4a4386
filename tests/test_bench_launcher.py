"""bench.py's N > 1 contract on a box without GPUs: ``--gpus N`` must end in ONE JSON line printed by
N ranks (``n_gpus == N``, parity block present) or fail loudly -- it must never print ``n_gpus: 1``.
The tile kernels are the NumPy stand-in of tests/np_tile_ops.py (``--tile-ops numpy``), the transport
is gloo; what is exercised is the launcher path itself: re-exec under torch.distributed.run, the
world-size assertions, the parity block and the C3 leg of the N > 1 line."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + argv, cwd=ROOT, env=e, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)


def _json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                out.append(json.loads(ln))
            except ValueError:
                pass
    return out


def test_gpus_2_launches_two_ranks_and_prints_one_line(tmp_path):
    detail = str(tmp_path / "detail.json")
    r = _run(["--gpus", "2", "--backend", "gloo", "--tile-ops", "numpy", "--n", "600", "--nb", "128",
              "--steps", "1", "--warmup", "0", "--no-cpu", "--detail", detail])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [d for d in _json_lines(r.stdout) if "metric" in d]
    assert len(lines) == 1
    assert r.stdout.strip().splitlines()[-1].startswith("{")     # THE line is the last thing on stdout
    assert len(r.stdout.strip().splitlines()[-1]) < 6000          # ... and fits the driver's 8 kB stdout tail
    d = lines[0]
    assert d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["rccl"]["ranks_seen"] == 2
    assert d["config"]["grid"] == "2x1" and d["config"]["parallelism"] == "block-cyclic-2"
    assert d["scaling"] == "strong" and d["steps"] == 1 and d["warmup"] == 0
    assert "NOT a measurement" in d["data"]
    p = d["parity"]
    assert p["ok"] and p["rel"]["headline"] <= 1e-9 and p["rel"]["ranks_agree"] == 0.0
    assert "C3_matern32" in d["also"]                        # configs[2]'s kernel on the same workspace
    # one run adjudicates the grid: the other grid timed on the same ranks, the links probed before the timed steps
    assert d["also"]["other_grid"]["grid"] == "1x2" and d["also"]["other_grid"]["seconds_per_step"] > 0
    assert p["rel"]["other_grid"] <= 1e-9
    lk = d["links"]
    assert lk["one_link_GBs"]["min"] > 0 and lk["all_links_busy_GBs_per_link"] > 0 and lk["allgather_busbw_GBs"] > 0
    full = json.load(open(detail))                           # the full record behind the line
    assert len(full["rccl"]["members"]) == 2 and full["parity"]["ranks_agree"]["spread"] == 0.0
    assert "also_C3_matern32" in full["config"]


def test_matern32_kernel_flag():
    r = _run(["--gpus", "2", "--backend", "gloo", "--tile-ops", "numpy", "--size", "500", "--nb", "128",
              "--steps", "1", "--warmup", "0", "--no-cpu", "--kernel", "matern32"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = [x for x in _json_lines(r.stdout) if "metric" in x][0]
    assert d["n_gpus"] == 2 and d["config"]["kernel"] == "Matern32" and d["parity"]["ok"]
    assert "C3_matern32" not in d.get("also", {})


@pytest.mark.skipif(torch.cuda.is_available() and torch.cuda.device_count() >= 2, reason="box has the GPUs: would really launch")
def test_gpus_2_without_two_gpus_fails_loudly():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"])
    assert r.returncode != 0
    assert not [d for d in _json_lines(r.stdout) if "metric" in d]         # in particular no "n_gpus": 1 line
    assert "refusing to run" in r.stderr


def test_world_size_mismatch_fails_loudly():
    r = _run(["--gpus", "2", "--tile-ops", "numpy", "--steps", "1", "--warmup", "0", "--no-cpu"],
             env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2
    assert not [d for d in _json_lines(r.stdout) if "metric" in d]
    assert "WORLD_SIZE=1" in r.stderr
