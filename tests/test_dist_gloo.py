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
