"""N > 1 on real hardware -- self-proving wherever at least two GPUs are visible (skipped on the one-GPU box).

(a) the N-rank RCCL gradient exchange of `apply_gradient_allreduce` (`distributed_util.py:97-149`): N ranks, one per
    GPU, the HIP SaShiMi engine under the bucketed asynchronous all-reduce, 2 optimizer steps on per-rank shards ==
    ONE process stepping on the concatenated global batch (the assertion of tests/test_training_dp_gloo.py, on `nccl`),
    every gradient slot written in place in the bucket arena (`last_stats["copied"] == 0`), the exchange timed;
(b) `python bench.py --gpus N` as the plain command (it spawns one rank per GPU like `generate.py:217-227`): the line
    carries world_size == N over nccl, N distinct per-rank Philox seeds and state digests, and per-rank step times
    within 10 % of each other.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _ngpu():
    try:
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:   # noqa: BLE001
        return 0


needs_two = pytest.mark.skipif(_ngpu() < 2, reason="needs at least two GPUs (N-rank RCCL over xGMI)")

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch, torch.nn as nn, torch.distributed as dist
from tests import cases
from diffwave_sashimi_amd.distributed_util import apply_gradient_allreduce, init_distributed, reduce_tensor
from diffwave_sashimi_amd.training import training_loss
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
init_distributed(rank, world, "g", "nccl", "tcp://127.0.0.1:" + os.environ["MASTER_PORT"])
assert dist.get_backend() == "nccl" and dist.get_world_size() == world
PB, L, STEPS = 2, 512, 4      # step 1 records the gradient order and re-cuts the buckets, step 2 records again, steps 3-4 overlap
cfg = cases.ss_cfg(d_model=32, n_layers=1, L=L, diffusion_step_embed_dim_mid=64)
net = cases.build_ours(cfg, 300 + rank).cuda().train()          # ranks start with DIFFERENT weights ...
net = apply_gradient_allreduce(net, bucket_bytes=64 * 1024)      # ... rank 0's are broadcast; several buckets
red = net._dws_grad_reducer
dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
opt = torch.optim.SGD(net.parameters(), lr=0.05)
data = torch.randn(STEPS, PB * world, 1, L, generator=torch.Generator().manual_seed(7)) * 0.3
losses, ar_ms, ex_ms = [], [], []
for step in range(STEPS):
    opt.zero_grad(set_to_none=True)
    shard = data[step, PB * rank: PB * rank + PB].cuda()
    loss = training_loss(net, nn.MSELoss(), shard, dh, generator=torch.Generator().manual_seed(1000 + 10 * step + rank))
    losses.append(float(reduce_tensor(loss.detach(), world)))
    loss.backward()
    opt.step()
    ar_ms.append(red.allreduce_ms())
    ex_ms.append(red.exposed_ms())
torch.cuda.synchronize()
print(json.dumps({"rank": rank, "world": dist.get_world_size(), "backend": dist.get_backend(), "losses": losses,
                  "slots": red.last_stats, "buckets": len(red.buckets), "allreduce_ms": ar_ms, "exposed_ms": ex_ms, "ready": red.bucket_ready_points(),
                  "device": torch.cuda.current_device(),
                  "digest": [float(p.detach().double().sum()) for p in net.parameters()]}))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single_process_reference(world):
    """The same steps in ONE process on the concatenated global batch, rank 0's initial weights, every shard's own
    (t, z) draw -- through the same HIP engine."""
    import torch.nn as nn
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import q_sample
    from tests import cases
    PB, L, STEPS = 2, 512, 4
    cfg = cases.ss_cfg(d_model=32, n_layers=1, L=L, diffusion_step_embed_dim_mid=64)
    net = cases.build_ours(cfg, 300).cuda().train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    data = torch.randn(STEPS, PB * world, 1, L, generator=torch.Generator().manual_seed(7)) * 0.3
    losses = []
    for step in range(STEPS):
        opt.zero_grad(set_to_none=True)
        total = 0.0
        for rank in range(world):   # mean over ranks of per-shard MSE means == what the averaged gradients descend on
            shard = data[step, PB * rank: PB * rank + PB].cuda()
            loss = cases_training_loss(net, shard, dh, torch.Generator().manual_seed(1000 + 10 * step + rank)) / world
            loss.backward()
            total += float(loss)
        losses.append(total)
        opt.step()
    torch.cuda.synchronize()
    return losses, [float(p.detach().double().sum()) for p in net.parameters()]


def cases_training_loss(net, shard, dh, gen):
    import torch.nn as nn
    from diffwave_sashimi_amd.training import training_loss
    return training_loss(net, nn.MSELoss(), shard, dh, generator=gen)


@needs_two
def test_n_rank_rccl_allreduce_equals_the_single_process_global_batch(gpu):
    world = min(_ngpu(), 8)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(rank), MASTER_PORT=str(port), MASTER_ADDR="127.0.0.1",
                   DWS_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True, cwd=ROOT))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.strip().splitlines() if l.startswith("{")][-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [o["rank"] for o in outs] == list(range(world))
    assert sorted(o["device"] for o in outs) == list(range(world))            # one rank per GPU
    assert all(o["backend"] == "nccl" and o["world"] == world for o in outs)
    for o in outs:
        assert o["slots"]["copied"] == 0 and o["slots"]["in_place"] > 0, o["slots"]    # zero-copy arena on every rank
        assert o["buckets"] > 1 and all(ms is not None and ms >= 0 for ms in o["allreduce_ms"])
        # staged hand-over: every bucket's all-reduce was launched behind its own event, most of them before backward ended
        assert o["slots"]["overlapped_buckets"] == o["buckets"] and all(ms is not None and ms >= 0 for ms in o["exposed_ms"])
        pts, end = o["ready"]["bucket_ready_point"], o["ready"]["last_point"]
        assert pts == outs[0]["ready"]["bucket_ready_point"] and sum(p_ < end for p_ in pts) >= len(pts) // 2      # same buckets on every rank
        assert o["losses"] == outs[0]["losses"]
        assert o["digest"] == pytest.approx(outs[0]["digest"], rel=0, abs=1e-9)          # identical weights everywhere
    ref_losses, ref_digest = _single_process_reference(world)
    assert outs[0]["losses"] == pytest.approx(ref_losses, rel=5e-5)
    assert outs[0]["digest"] == pytest.approx(ref_digest, rel=2e-5, abs=2e-5)
    print(f"world {world}: allreduce_ms per step on rank 0 {outs[0]['allreduce_ms']}, exposed_ms {outs[0]['exposed_ms']}, "
          f"buckets {outs[0]['buckets']} ready at {outs[0]['ready']}")


@needs_two
@pytest.mark.parametrize("mode", ["sample", "train"])
def test_bench_line_at_n_gpus(gpu, mode):
    world = min(_ngpu(), 8)
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "1",
            "--no-cpu-baseline", "--no-roofline", "--no-extra", "--no-full-loop"]
    if mode == "train":
        args += ["--mode", "train", "--config", "unet_d128_n6_T200"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run(args, capture_output=True, text=True, timeout=1800, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["process_group"]["world_size"] == world
    assert line["process_group"]["backend"] == "nccl"
    ms = line["per_rank_ms_per_step"]
    assert len(ms) == world and max(ms) <= 1.10 * min(ms), ms
    if mode == "sample":
        assert len(set(line["per_rank_seed"])) == world                      # every rank its own Philox stream ...
        assert len(set(line["per_rank_state_digest"])) == world              # ... and its own clips
    else:
        dig = line["per_rank_param_digest"]
        assert max(dig) - min(dig) <= 1e-6 * abs(dig[0])                      # the averaged steps keep the replicas identical
        assert len(line["dp"]["allreduce_ms_per_rank"]) == world and min(line["dp"]["allreduce_ms_per_rank"]) > 0
        assert line["dp"]["gradient_slots"]["copied"] == 0
