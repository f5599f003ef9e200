"""CPU, world_size 2 and 8 over gloo: the data-parallel gradient exchange and the
training-loss caller (row a19) around a differentiable net of the reference surface."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch

from tests.conftest import ROOT, load_golden

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch, torch.nn as nn
import torch.distributed as dist
from diffwave_sashimi_amd.distributed_util import apply_gradient_allreduce, init_distributed, reduce_tensor
from diffwave_sashimi_amd.training import training_loss
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
init_distributed(rank, world, "g", "gloo", "tcp://127.0.0.1:" + os.environ["MASTER_PORT"])

class Net(nn.Module):                       # reference surface: net((x, t), mel_spec=None)
    def __init__(self):
        super().__init__()
        self.a = nn.Conv1d(1, 8, 3, padding=1)
        self.b = nn.Conv1d(8, 1, 1)
        self.unused = nn.Linear(4, 4)       # never touched by forward: gets no gradient
        self.cplx = nn.Parameter(torch.view_as_real(torch.randn(3, dtype=torch.cfloat)))  # S4-style real view
        self.emb = nn.Linear(1, 8)
    def forward(self, inp, mel_spec=None):
        x, t = inp
        h = torch.tanh(self.a(x) + self.emb(t.float().view(-1, 1)).unsqueeze(-1))
        return self.b(h) * (1 + self.cplx.sum())

torch.manual_seed(100 + rank)               # ranks start with DIFFERENT weights ...
net = Net()
before = [p.detach().clone() for p in net.parameters()]
net = apply_gradient_allreduce(net, bucket_bytes=64)     # ... tiny buckets: several all-reduces in flight
after = [p.detach().clone() for p in net.parameters()]
# rank 0's weights everywhere after the flattened broadcast
ref0 = [torch.zeros_like(p) for p in after]
for r, p in zip(ref0, after):
    r.copy_(p); dist.broadcast(r, 0)
same_after_bcast = all(torch.equal(r, p) for r, p in zip(ref0, after))

dh = calc_diffusion_hyperparams(20, 1e-4, 0.05)
opt = torch.optim.SGD(net.parameters(), lr=0.05)
gall = torch.Generator().manual_seed(7)
PB = int(os.environ["DWS_PER_RANK_BATCH"])
data = torch.randn(3, PB * world, 1, 32, generator=gall)  # 3 steps x global batch PB * world
losses = []
for step in range(3):
    shard = data[step, PB * rank: PB * rank + PB]
    g = torch.Generator().manual_seed(1000 + 10 * step + rank)
    opt.zero_grad()
    loss = training_loss(net, nn.MSELoss(), shard, dh, generator=g)
    losses.append(float(reduce_tensor(loss.detach(), world)))
    loss.backward()
    opt.step()
print(json.dumps({"rank": rank, "same_after_bcast": same_after_bcast, "losses": losses,
                  "unused_grad_none": net.unused.weight.grad is None,
                  "params": [p.detach().reshape(-1).tolist() for p in net.parameters()]}))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest


@pytest.mark.parametrize("world,PB", [(2, 2), (8, 1)])
def test_dp_matches_single_process_full_batch(tmp_path, world, PB):
    """world 2 (2 clips per rank) and world 8 (the node size of BASELINE config 5; 1 clip per rank): the DP result --
    broadcast, bucketed all-reduce of the averaged gradients, 3 optimizer steps -- equals ONE process stepping on the
    concatenated global batch."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, WORLD_SIZE=str(world), RANK=str(rank), MASTER_PORT=str(port), DWS_ROOT=ROOT,
                   OMP_NUM_THREADS="1", DWS_PER_RANK_BATCH=str(PB))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert [o["rank"] for o in outs] == list(range(world))
    assert all(o["same_after_bcast"] and o["unused_grad_none"] for o in outs)
    # every rank holds identical parameters after 3 averaged steps, and the logged loss is the mean
    for o in outs[1:]:
        for a, b in zip(outs[0]["params"], o["params"]):
            assert np.allclose(a, b, rtol=0, atol=1e-7)
        assert outs[0]["losses"] == o["losses"]

    # single-process reference: the same 3 steps on the concatenated (global) batch
    import torch.nn as nn
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import q_sample

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Conv1d(1, 8, 3, padding=1)
            self.b = nn.Conv1d(8, 1, 1)
            self.unused = nn.Linear(4, 4)
            self.cplx = nn.Parameter(torch.view_as_real(torch.randn(3, dtype=torch.cfloat)))
            self.emb = nn.Linear(1, 8)

        def forward(self, inp, mel_spec=None):
            x, t = inp
            h = torch.tanh(self.a(x) + self.emb(t.float().view(-1, 1)).unsqueeze(-1))
            return self.b(h) * (1 + self.cplx.sum())

    torch.manual_seed(100)                  # rank 0's initial weights
    net = Net()
    dh = calc_diffusion_hyperparams(20, 1e-4, 0.05)
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    gall = torch.Generator().manual_seed(7)
    data = torch.randn(3, PB * world, 1, 32, generator=gall)
    for step in range(3):
        xs, ts, zs = [], [], []
        for rank in range(world):           # the shards draw (t, z) from their own seeded streams
            g = torch.Generator().manual_seed(1000 + 10 * step + rank)
            shard = data[step, PB * rank: PB * rank + PB]
            t = torch.randint(20, size=(PB, 1, 1), generator=g)
            z = torch.normal(0, 1, size=shard.shape, generator=g)
            xs.append(q_sample(shard, t, dh["Alpha_bar"], z)); ts.append(t); zs.append(z)
        x, t, z = torch.cat(xs), torch.cat(ts), torch.cat(zs)
        opt.zero_grad()
        loss = nn.MSELoss()(net((x, t.view(PB * world, 1))), z)
        assert abs(float(loss.detach()) - outs[0]["losses"][step]) < 1e-6
        loss.backward()
        opt.step()
    for a, p in zip(outs[0]["params"], net.parameters()):
        assert np.allclose(a, p.detach().reshape(-1).numpy(), rtol=0, atol=2e-6)


def test_training_loss_consumes_rng_like_the_reference():
    """`train.py:218-219`: randint for t, then normal for z, both on the CPU generator."""
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    audio = torch.randn(3, 1, 16)
    seen = {}

    def net(inp, mel_spec=None):
        seen["x"], seen["t"] = inp
        return torch.zeros_like(inp[0])

    g = torch.Generator().manual_seed(5)
    loss = training_loss(net, torch.nn.MSELoss(), audio, dh, generator=g)
    g2 = torch.Generator().manual_seed(5)
    t = torch.randint(50, size=(3, 1, 1), generator=g2)
    z = torch.normal(0, 1, size=audio.shape, generator=g2)
    assert torch.equal(seen["t"], t.view(3, 1))
    ab = dh["Alpha_bar"][t]
    assert torch.allclose(seen["x"], torch.sqrt(ab) * audio + torch.sqrt(1 - ab) * z)
    assert abs(float(loss) - float((z ** 2).mean())) < 1e-6
