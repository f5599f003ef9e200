"""GPU: the RCCL ("nccl" backend) code paths execute on the GPU box -- a 1-rank group on the single GPU: process-group
initialisation pinned to the device, the flattened weight broadcast, the bucketed asynchronous gradient all-reduce from
the post-accumulate hooks, `reduce_tensor`, the device-side barrier / max / gather of `bench.py`'s timing -- and
`python bench.py --gpus N` as a PLAIN command (it spawns its own ranks like `generate.py:220-227`)."""
import json
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch, torch.nn as nn, torch.distributed as dist
from tests import cases
from diffwave_sashimi_amd import dist as ddist
from diffwave_sashimi_amd.distributed_util import apply_gradient_allreduce, init_distributed, reduce_tensor
from diffwave_sashimi_amd.training import training_loss
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams

def run(dp):
    cfg, L = cases.ss_cfg(d_model=32, n_layers=1, L=512, diffusion_step_embed_dim_mid=64), 512
    net = cases.build_ours(cfg, 300).cuda().train()
    nb = 0
    if dp:
        net = apply_gradient_allreduce(net, bucket_bytes=64 * 1024)    # several buckets, complex S4 params as real views
        nb = len(net._dws_grad_reducer.buckets)
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    data = torch.randn(5, 2, 1, L, generator=torch.Generator().manual_seed(7)) * 0.3
    losses = []
    global per_step
    per_step = []
    for step in range(5):
        opt.zero_grad()
        loss = training_loss(net, nn.MSELoss(), data[step].cuda(), dh, generator=torch.Generator().manual_seed(1000 + step))
        losses.append(float(reduce_tensor(loss.detach(), 1)) if dp else float(loss))
        loss.backward()
        opt.step()
        if dp:
            red = net._dws_grad_reducer
            per_step.append({"buckets": len(red.buckets), "stats": red.last_stats, "exposed_ms": red.exposed_ms(),
                             "allreduce_ms": red.allreduce_ms(), "ready": red.bucket_ready_points()})
    torch.cuda.synchronize()
    global slot_stats
    slot_stats = net._dws_grad_reducer.last_stats if dp else None
    return losses, [p.detach().clone() for p in net.parameters()], nb

plain_losses, plain_params, _ = run(False)
init_distributed(0, 1, "g", "nccl", "tcp://127.0.0.1:" + os.environ["MASTER_PORT"])
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dp_losses, dp_params, nb = run(True)
ddist.barrier()
dev = torch.device("cuda", 0)
mx = ddist.max_over_ranks(1.25, dev)
ga = ddist.gather_over_ranks(2.5, dev)
same = all(torch.equal(a, b) for a, b in zip(plain_params, dp_params))
print(json.dumps({"backend": dist.get_backend(), "world": dist.get_world_size(), "buckets": nb, "same": same, "slots": slot_stats, "per_step": per_step,
                  "losses": [plain_losses, dp_losses], "max": mx, "gather": ga}))
dist.destroy_process_group()
'''


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-3000:]
    return json.loads(lines[0])


def test_one_rank_rccl_group_runs_the_dp_exchange(gpu):
    env = dict(os.environ, DWS_ROOT=ROOT, MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    d = _last_json(r.stdout)
    assert d["backend"] == "nccl" and d["world"] == 1 and d["buckets"] >= 3
    assert d["same"], "averaging over one rank must leave the training trajectory bit-identical"
    # zero-copy exchange: the engine wrote every gradient straight into the flat all-reduce buckets (no copy in / back)
    assert d["slots"]["copied"] == 0 and d["slots"]["in_place"] > 100, d["slots"]
    assert d["losses"][0] == d["losses"][1] and d["max"] == 1.25 and d["gather"] == [2.5]
    # STAGED hand-over (dws_model_set_grad_sinks): the engine's backward delivers the buckets as their last gradient is produced
    # and every bucket's all-reduce is launched behind ITS event.  Step 1 records the order and re-cuts the buckets by
    # readiness, step 2 records again for the new buckets, from step 3 on buckets leave in the middle of backward: most of them
    # are ready before the last flush point (only the stacked fc_t / embedding gradients come at the very end).
    ps = d["per_step"]
    assert all(s_["stats"]["copied"] == 0 for s_ in ps) and all(s_["exposed_ms"] is not None and s_["exposed_ms"] >= 0 for s_ in ps)
    last = ps[-1]
    assert last["stats"]["overlapped_buckets"] == last["buckets"] >= 3
    pts, end = last["ready"]["bucket_ready_point"], last["ready"]["last_point"]
    assert sorted(pts) == pts and pts[-1] == end and sum(p_ < end for p_ in pts) >= len(pts) // 2, last["ready"]


def test_bench_timing_collectives_over_rccl(gpu):
    """N = 1 with a forced 1-rank RCCL group: the barrier / max / gather bracket of bench.py runs on the device."""
    env = dict(os.environ, DWS_BENCH_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "wnet_h128_d30_T200", "--batch", "2",
                        "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-roofline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    d = _last_json(r.stdout)
    assert d["process_group"] == {"backend": "nccl", "world_size": 1} and d["n_gpus"] == 1
    assert len(d["per_rank_ms_per_step"]) == 1 and abs(d["per_rank_ms_per_step"][0] - d["ms_per_step"]) < 1e-9


@pytest.mark.parametrize("mode", ["sample", "train"])
def test_plain_bench_command_spawns_its_ranks(gpu, mode):
    """`python bench.py --gpus 2` without torch.distributed.run: two ranks (sharing this box's one GPU over gloo,
    DWS_BENCH_SHARE_GPU=1; real runs: one GPU per rank over RCCL), one JSON line, whole-job aggregate."""
    env = dict(os.environ, DWS_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "wnet_h128_d30_T200", "--batch", "2",
           "--steps", "2", "--warmup", "1", "--mode", mode]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-4000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["process_group"]["world_size"] == 2 and len(d["per_rank_ms_per_step"]) == 2
    assert max(d["per_rank_ms_per_step"]) == pytest.approx(d["ms_per_step"], rel=1e-9)
    per_rank_units = 2 * 16000 / (200 if mode == "sample" else 1)
    assert d["value"] == pytest.approx(2 * per_rank_units / (d["ms_per_step"] * 1e-3), rel=1e-6)
    assert "cpu_baseline" not in d
