"""GPU: bench.py contract -- one JSON line with the required keys at N = 1, and the N > 1 control flow
(torch.distributed.run, barrier + max over ranks, rank-0-only output, whole-job aggregate) with two ranks
sharing the box's single GPU over gloo (DWS_BENCH_SHARE_GPU=1; real runs use one GPU per rank over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_and_cpu_baseline(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "wnet_h128_d30_T200", "--steps", "4",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["config"]["workload"] == "wnet_h128_d30_T200" and "model" not in d["config"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert abs(d["value"] - 16 * 16000 / (200 * d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_two_ranks_aggregate(gpu):
    env = dict(os.environ, DWS_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "wnet_h128_d30_T200",
           "--batch", "4", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only
    # whole-job aggregate: 2 ranks x (4 clips x 16000 samples / 200 steps) per step time
    assert abs(d["value"] - 2 * 4 * 16000 / (200 * d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
