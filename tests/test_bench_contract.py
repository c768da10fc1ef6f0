"""bench.py's output contract and its multi-rank control flow, exercised WITHOUT a GPU through the script's
self-test mode (CPU tensors, gloo, kernel emulation, toy model — the numbers are meaningless).  What is checked is
what the driver depends on: one JSON line from rank 0 with the agreed keys, at N = 1 and under
`python -m torch.distributed.run --nproc-per-node 2` (all-reduce of the factors, max-over-ranks timing, the
eigendecomposition sharded over the ranks)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(cmd):
    env = dict(os.environ, LK_BENCH_SELFTEST="1", PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected exactly one JSON line, got {len(lines)}: {out.stdout[-2000:]}"
    return json.loads(lines[0])


def _check(d, n, steps, warmup):
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["n_gpus"] == n and d["steps"] == steps and d["warmup"] == warmup
    assert d["metric"] == "KFAC-GGN fit samples/sec, ResNet-18" and d["unit"] == "samples/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and "workload" in d["config"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert d["eigh_converged"] is True and d["eigh_ms"] > 0


def test_single_process_line():
    d = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1"])
    _check(d, 1, 2, 1)
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}


def test_two_ranks_line():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1"]
    d = _run(cmd)
    _check(d, 2, 2, 1)
    assert d["eigh_sharded_over_gpus"] == 2 and d["config"]["parallelism"] == "dp2"
