"""Process-group plumbing for multi-GPU sampling (one process per GPU).

Sampling shards by independent clips (`generate.py:217-227`): every rank runs its
own batches with its own RNG stream and writes clip indices
``n_samples * rank + i`` (`generate.py:189`); there is no data-path collective.
The process group is only used to bracket timed regions (barrier + max over
ranks).  Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests."""
import os

import torch
import torch.distributed as dist


def env_world():
    return (int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend=None, force=False):
    """Initialise from the torchrun environment; no-op for a single process unless ``force`` (a 1-rank
    group: the same RCCL barrier / reductions as an N-rank job, runnable on a one-GPU box)."""
    world, rank, local_rank = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kw)
    return world, rank, local_rank


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def barrier():
    if dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def max_over_ranks(seconds, device=None):
    """Slowest rank's time: the job is only as fast as its slowest shard."""
    if not dist.is_initialized():
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(seconds, device=None):
    """Every rank's time, in rank order (reported beside the max so a slow shard is visible)."""
    if not dist.is_initialized():
        return [float(seconds)]
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def group_info():
    """Backend and size of the live process group ("nccl" is RCCL on ROCm)."""
    if not dist.is_initialized():
        return {"backend": None, "world_size": 1}
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size()}


def rank_seed(base_seed, rank):
    """Independent Philox sub-stream per rank (seed = 1234 + rank)."""
    return (int(base_seed) + int(rank)) & 0xFFFFFFFFFFFFFFFF


def clip_indices(n_samples_per_rank, rank):
    """Global indices of the clips a rank generates (`generate.py:189`)."""
    return [n_samples_per_rank * rank + i for i in range(n_samples_per_rank)]


def aggregate_throughput(units_per_rank, world, seconds):
    """Whole-job rate: every rank processed `units_per_rank` units in `seconds` (max over ranks)."""
    return world * units_per_rank / seconds


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
