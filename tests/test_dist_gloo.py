"""CPU, world_size 2 over gloo: the N>1 plumbing bench.py / multi-GPU sampling relies on
(sharding by independent clips, rank seeds, barrier + max-over-ranks timing)."""
import os
import socket
import subprocess
import sys

from tests.conftest import ROOT

WORKER = r'''
import json, os, sys, time
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch
from diffwave_sashimi_amd import dist as ddist
world, rank, local_rank = ddist.init("gloo")
assert world == 2
ddist.barrier()
t0 = time.perf_counter()
time.sleep(0.05 * (rank + 1))          # rank 1 is the slow shard
ddist.barrier()
el = time.perf_counter() - t0
mx = ddist.max_over_ranks(0.1 * (rank + 1))
seed = ddist.rank_seed(1234, rank)
# each rank "generates" its own clips from its own stream; no collective on the data path
g = torch.Generator().manual_seed(seed)
x = torch.randn(4, generator=g)
gathered = [torch.zeros(4), torch.zeros(4)]
torch.distributed.all_gather(gathered, x)   # test-only: compare the two streams
print(json.dumps({"rank": rank, "max": mx, "seed": seed, "clips": ddist.clip_indices(3, rank),
                  "same_stream": bool(torch.equal(gathered[0], gathered[1])),
                  "agg": ddist.aggregate_throughput(16 * 16000, world, mx), "elapsed": el}))
ddist.shutdown()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_sharding_and_timing(tmp_path):
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DWS_ROOT=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=120)
        assert p.returncode == 0, e[-2000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    assert outs[0]["max"] == outs[1]["max"] == 0.2                 # max over ranks, identical on both
    assert outs[0]["seed"] == 1234 and outs[1]["seed"] == 1235
    assert not outs[0]["same_stream"]                                # independent RNG streams
    assert outs[0]["clips"] == [0, 1, 2] and outs[1]["clips"] == [3, 4, 5]   # generate.py:189
    assert abs(outs[0]["agg"] - 2 * 16 * 16000 / 0.2) < 1e-6
    assert min(o["elapsed"] for o in outs) >= 0.09                   # barrier waited for the slow rank


def test_single_process_is_a_noop():
    from diffwave_sashimi_amd import dist as ddist
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK") if k in os.environ}
    try:
        assert ddist.init() == (1, 0, 0)
        assert ddist.max_over_ranks(0.5) == 0.5
        ddist.barrier()
    finally:
        os.environ.update(env)


ARENA_WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch, torch.nn as nn, torch.distributed as dist
from diffwave_sashimi_amd.distributed_util import apply_gradient_allreduce

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)


class ArenaFn(torch.autograd.Function):
    """What models/engine._EngineTrainFn.backward does with the reducer's views: the gradients are WRITTEN into the views
    handed out for this backward and those very tensors are returned."""
    @staticmethod
    def forward(ctx, module, x, *params):
        ctx.module, ctx.x = module, x
        return sum((p * p).sum() for p in params) * x.sum()

    @staticmethod
    def backward(ctx, dout):
        m = ctx.module
        views = m._dws_grad_reducer.grad_views()
        outs = []
        for p in m.parameters():
            v = views[id(p)] if p.grad is None else torch.empty_like(p)
            v.copy_(2 * p.detach() * ctx.x.sum() * dout)
            outs.append(v)
        del views
        return (None, None, *outs)


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.a = nn.Parameter(torch.randn(7, 3, generator=g))
        self.b = nn.Parameter(torch.randn(11, generator=g))
        self.c = nn.Parameter(torch.randn(2, 2, 2, generator=g))

    def forward(self, x):
        return ArenaFn.apply(self, x, *self.parameters())


net = apply_gradient_allreduce(Net(), bucket_bytes=64)        # tiny buckets: several of them
red = net._dws_grad_reducer
x = torch.full((4,), float(rank + 1))
net(x).backward()
stats1 = dict(red.last_stats)
flat_ptrs = [(b.flat.data_ptr(), b.flat.data_ptr() + b.flat.numel() * 4) for b in red.buckets]
inside = all(any(lo <= p.grad.data_ptr() < hi for lo, hi in flat_ptrs) for p in net.parameters())
# mean over ranks of 2 p sum(x): sum(x) = 4 (rank + 1)
want = [2 * p.detach() * 4 * (sum(r + 1 for r in range(world)) / world) for p in net.parameters()]
ok1 = all(torch.allclose(p.grad, w) for p, w in zip(net.parameters(), want))
# second backward WITHOUT zeroing: the new gradient must not alias the arena slot it is accumulated into
net(x).backward()
stats2 = dict(red.last_stats)
ok2 = all(torch.allclose(p.grad, 2 * w) for p, w in zip(net.parameters(), want))
# after zero_grad(set_to_none) the arena is used again
for p in net.parameters():
    p.grad = None
net(x).backward()
stats3 = dict(red.last_stats)
ok3 = all(torch.allclose(p.grad, w) for p, w in zip(net.parameters(), want))
print(json.dumps({"rank": rank, "buckets": len(red.buckets), "stats": [stats1, stats2, stats3], "inside": inside,
                  "ok": [ok1, ok2, ok3]}))
dist.destroy_process_group()
'''


def test_gradients_written_into_the_bucket_views_are_adopted_without_a_copy(tmp_path):
    """The zero-copy exchange of `distributed_util.GradientAllReducer` rests on autograd ADOPTING a returned bucket view as
    `p.grad` (no clone) and on the hook recognising it: a stand-in for the engine's backward writes its gradients into
    `grad_views()` and returns them (world 2, gloo).  In place: every slot, `p.grad` lives inside a bucket buffer and holds
    the cross-rank mean.  A second backward WITHOUT zeroing gets its gradients in fresh tensors (never in the slot `p.grad`
    still occupies), autograd accumulates them into the view in place, and the exchange of "mean so far + local" yields the
    sum of the two means -- still without a copy."""
    import json
    script = tmp_path / "arena_worker.py"
    script.write_text(ARENA_WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DWS_ROOT=ROOT,
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    for p in procs:
        o, e = p.communicate(timeout=180)
        assert p.returncode == 0, e[-3000:]
        d = json.loads(o.strip().splitlines()[-1])
        assert d["buckets"] >= 2 and d["inside"] and d["ok"] == [True, True, True], d
        # ("kept": retained gradients of non-leaf tensors stashed over the exchange -- none in this worker)
        assert all(st == {"in_place": 3, "copied": 0, "kept": 0, "overlapped_buckets": 0} for st in d["stats"]), d
