"""GPU: data-parallel training of both backbones through the HIP engine -- two ranks (one GPU shared, gloo for
the exchange since one device cannot host two RCCL ranks) against one process on the global batch."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from tests import cases
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

DP_CASES = {
    "wavenet": (cases.wn_cfg(res_channels=64, skip_channels=64, num_res_layers=2, dilation_cycle=2), 192),
    # SaShiMi: complex S4 parameters travel as real views; _setup_C runs after the broadcast on identical weights
    "sashimi": (cases.ss_cfg(d_model=32, n_layers=1, L=512, diffusion_step_embed_dim_mid=64), 512),
}
STEPS, LR = 2, 0.05

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch, torch.nn as nn, torch.distributed as dist
from tests import cases
from tests.test_training_dp_gpu import DP_CASES, STEPS, LR
CFG, L = DP_CASES[os.environ["DWS_DP_CASE"]]
from diffwave_sashimi_amd.distributed_util import apply_gradient_allreduce, init_distributed, reduce_tensor
from diffwave_sashimi_amd.training import training_loss
from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
init_distributed(rank, world, "g", "gloo", "tcp://127.0.0.1:" + os.environ["MASTER_PORT"])
net = cases.build_ours(CFG, 300 + rank).cuda().train()        # different weights per rank before the broadcast
net = apply_gradient_allreduce(net, bucket_bytes=64 * 1024)    # several buckets
dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
opt = torch.optim.SGD(net.parameters(), lr=LR)
data = torch.randn(STEPS, 4, 1, L, generator=torch.Generator().manual_seed(7)) * 0.3
losses = []
for step in range(STEPS):
    shard = data[step, 2 * rank: 2 * rank + 2].cuda()
    opt.zero_grad()
    loss = training_loss(net, nn.MSELoss(), shard, dh, generator=torch.Generator().manual_seed(1000 + 10 * step + rank))
    losses.append(float(reduce_tensor(loss.detach(), world)))
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print(json.dumps({"rank": rank, "losses": losses,
                  "digest": [float(p.detach().double().sum()) for p in net.parameters()],
                  "abs": [float(p.detach().double().abs().sum()) for p in net.parameters()]}))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("case", list(DP_CASES))
def test_two_rank_engine_training_matches_global_batch(tmp_path, gpu, case):
    CFG, L = DP_CASES[case]
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import q_sample
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), MASTER_PORT=str(port), DWS_ROOT=ROOT, OMP_NUM_THREADS="1", DWS_DP_CASE=case,
                   PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert np.allclose(outs[0]["digest"], outs[1]["digest"], rtol=0, atol=1e-9)   # replicas stay identical
    assert outs[0]["losses"] == outs[1]["losses"]

    # one process, global batch of 4, same per-shard (t, z) draws
    net = cases.build_ours(CFG, 300).to(gpu).train()
    dh = calc_diffusion_hyperparams(50, 1e-4, 0.05)
    opt = torch.optim.SGD(net.parameters(), lr=LR)
    data = torch.randn(STEPS, 4, 1, L, generator=torch.Generator().manual_seed(7)) * 0.3
    for step in range(STEPS):
        xs, ts, zs = [], [], []
        for rank in range(2):
            g = torch.Generator().manual_seed(1000 + 10 * step + rank)
            shard = data[step, 2 * rank: 2 * rank + 2]
            t = torch.randint(50, size=(2, 1, 1), generator=g)
            z = torch.normal(0, 1, size=shard.shape, generator=g)
            xs.append(q_sample(shard, t, dh["Alpha_bar"], z)); ts.append(t); zs.append(z)
        x, t, z = torch.cat(xs).to(gpu), torch.cat(ts).to(gpu), torch.cat(zs).to(gpu)
        opt.zero_grad()
        loss = nn.MSELoss()(net((x, t.view(4, 1))), z)
        assert abs(float(loss.detach()) - outs[0]["losses"][step]) < 1e-5
        loss.backward()
        opt.step()
    # SaShiMi's scalar LayerNorm parameters (initialised 0 / 1) move by a sum over B*H*L signed terms that cancels
    # by ~3 orders of magnitude, so fp32 rounding differences between the two batch splits show up at 1e-4..1e-3
    # relative in those scalars (seen: 5e-4); everything else agrees to 1e-5
    tol = 1e-5 if case == "wavenet" else 2e-3
    for d, a, p in zip(outs[0]["digest"], outs[0]["abs"], net.parameters()):
        assert abs(d - float(p.detach().double().sum())) < tol * max(a, 1e-3)
