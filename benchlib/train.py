"""`bench.py --mode train`: one data-parallel training step through the HIP engine."""
import json
import os
import time

import torch

from .configs import DTYPE_NAMES, PEAK_F32_MFMA_TFLOPS, build_model
from .cpu_baseline import cpu_train_baseline
from .work import forward_gemm_flops, layer_algorithmic_work, wino_dgrad_executed_work, wino_executed_work, wino_wgrad_executed_work


def train_bench(args, cfg, world, rank, dev, ddist, red_dev=None, emit=True):
    red_dev = red_dev or dev
    """One data-parallel training step (`train.py:118-143`): q-sample + forward_train + MSE + backward through the
    HIP engine, bucketed asynchronous RCCL all-reduce of the gradients, Adam.  Synthetic audio
    U(-0.3, 0.3) (SURVEY.md 8d).  Not the headline metric; reported as training audio samples/s."""
    import torch.nn as nn
    from diffwave_sashimi_amd.distributed_util import apply_gradient_allreduce
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    from diffwave_sashimi_amd.training import training_loss
    # per-GPU batch: `configs/config.yaml:12` batch_size_per_gpu = 4 for WaveNet; BASELINE.json configs[4]
    # (SaShiMi unet_d128_n6) is quoted at 256 global on 8 GPUs = 32 per GPU
    B, L = (args.batch or (4 if cfg["model"]["_name_"] == "wavenet" else 32)), cfg["L"]
    net = build_model(cfg, dev).train()
    tprec = getattr(args, "precision", "f32")
    if tprec != "f32":       # SaShiMi: bf16x6 = the pointwise GEMMs and weight gradients of the step on the bf16 matrix cores
        net.set_option("precision", tprec)
    if world > 1:
        net = apply_gradient_allreduce(net)
    opt = torch.optim.Adam(net.parameters(), lr=2e-4)     # `train.py:91`
    dh = calc_diffusion_hyperparams(**cfg["diffusion"])
    g = torch.Generator().manual_seed(99 + rank)
    audio = ((torch.rand(B, 1, L, generator=g) * 2 - 1) * 0.3).to(dev)
    loss_fn = nn.MSELoss()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = training_loss(net, loss_fn, audio, dh, generator=g)
        loss.backward()
        opt.step()
        return loss

    for _ in range(max(args.warmup, 1)):
        step()
    ddist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    ddist.barrier()
    mine = time.perf_counter() - t0
    elapsed = ddist.max_over_ranks(mine, red_dev)
    per_rank_ms = [t / args.steps * 1e3 for t in ddist.gather_over_ranks(mine, red_dev)]
    per_rank_loss = ddist.gather_over_ranks(float(loss.detach()), red_dev)      # each rank's own shard (seed 99 + rank)
    with torch.no_grad():   # after the averaged steps every rank holds the same weights: digest of all parameters, per rank
        digest = float(sum(p.detach().double().abs().sum() for p in net.parameters()))
    per_rank_param_digest = ddist.gather_over_ranks(digest, red_dev)
    ms = elapsed / args.steps * 1e3
    dp_overhead = None
    if world > 1:      # the exchange as the last timed step saw it, per rank: first bucket launch -> last wait()
        red = net._dws_grad_reducer
        dp_overhead = {"allreduce_ms_per_rank": ddist.gather_over_ranks(float(red.allreduce_ms() or 0.0), red_dev),
                       "exposed_ms_per_rank": ddist.gather_over_ranks(float(red.exposed_ms() or 0.0), red_dev),
                       "buckets": len(red.buckets), "bucket_mbytes": [b.flat.numel() * 4 / 2 ** 20 for b in red.buckets],
                       "bucket_ready": red.bucket_ready_points(), "gradient_slots": red.last_stats}
    if world == 1 and not ddist.dist.is_initialized() and os.environ.get("DWS_BENCH_NO_DP_OVERHEAD") is None:
        # What data parallelism adds to ONE rank's step besides the wire time: the same steps inside a 1-rank RCCL group
        # (apply_gradient_allreduce: gradients written into the flat buckets, hooks, bucketed asynchronous all-reduces,
        # division) minus the plain steps above.  Measurable on a one-GPU box; the N-rank exchange itself is the driver's
        # scaling run.
        try:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ["MASTER_PORT"] = str(ddist._free_port())
            ddist.dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            apply_gradient_allreduce(net)
            for _ in range(max(args.warmup, 1)):
                step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            ms_pg = (time.perf_counter() - t1) / args.steps * 1e3
            red = net._dws_grad_reducer
            dp_overhead = {"dp_overhead_ms": ms_pg - ms, "ms_per_step_in_1rank_rccl_group": ms_pg, "ms_per_step_plain": ms,
                           "buckets": len(red.buckets), "bucket_mbytes": [b.flat.numel() * 4 / 2 ** 20 for b in red.buckets],
                           "gradient_slots": red.last_stats,
                           # first bucket launch -> last wait() of the last step (HIP events on the gradients' stream)
                           "allreduce_ms": red.allreduce_ms(),
                           # end of backward (its last kernel) -> last wait(): what the step could not hide behind backward
                           "exposed_ms": red.exposed_ms(), "bucket_ready": red.bucket_ready_points()}
            red.remove()
            del net._dws_grad_reducer
        except Exception as e:      # noqa: BLE001 -- reported in the line
            dp_overhead = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            if ddist.dist.is_initialized():
                ddist.dist.destroy_process_group()
    roofline = None
    if world == 1 and not args.no_roofline:     # (an extra step on one rank only would hang the other ranks' all-reduce)
        # the MFMA GEMM kernels of one step (forward layer / 1x1 GEMMs, data gradients, weight gradients), timed with
        # HIP events on their launch stream; algorithmic flops = 3 x the forward's dense contractions
        import ctypes
        from diffwave_sashimi_amd import _lib
        lib = _lib.load()
        _lib.check(lib.dws_profile_enable(b"mfma"))
        step()
        torch.cuda.synchronize()
        n_launch, tot_ms = ctypes.c_int64(), ctypes.c_double()
        _lib.check(lib.dws_profile_query(ctypes.byref(n_launch), ctypes.byref(tot_ms)))
        lib.dws_profile_disable()
        flops = 3 * forward_gemm_flops(cfg, B)
        # the WaveNet forward layer runs the Winograd form (8 C^2 instead of 12 C^2 flop per position for the conv): `frac`
        # is priced on the flops the kernels EXECUTE, the algorithmic figure is reported beside it
        executed = flops
        if cfg["model"]["_name_"] == "wavenet" and os.environ.get("DWS_WN_DIRECT") is None:
            m = cfg["model"]
            executed = flops - m["num_res_layers"] * (layer_algorithmic_work(dict(cfg, B=B))[0] - wino_executed_work(dict(cfg, B=B)))
            if m["res_channels"] % 128 == 0 and os.environ.get("DWS_TAPCONV_DIRECT") is None:   # the data gradient too
                C = m["res_channels"]
                executed -= m["num_res_layers"] * (B * cfg["L"] * 12 * C * C - wino_dgrad_executed_work(dict(cfg, B=B)))
            if os.environ.get("DWS_WGRAD_DIRECT") is None:   # and the weight gradient (any channel count)
                C = m["res_channels"]
                executed -= m["num_res_layers"] * (B * cfg["L"] * 12 * C * C - wino_wgrad_executed_work(dict(cfg, B=B)))
        if n_launch.value > 0:
            ach = executed / (tot_ms.value * 1e-3) / 1e12
            roofline = {"kernel": "all MFMA GEMM launches of one training step (tapconv_mfma / wgrad_mfma / forward layer): "
                                  "%d launches" % n_launch.value,
                        "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None, "ms_per_step_in_kernels": tot_ms.value,
                        "algorithmic_flops_per_step": flops, "executed_flops_per_step": executed,
                        "whole_step_frac": executed / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
    line = None
    if rank == 0:
        line = ({
            **({"roofline": roofline} if roofline else {}),
            "metric": "training audio samples/sec (train.py-style DP step: fwd + bwd + grad all-reduce + Adam)",
            "value": ddist.aggregate_throughput(B * L, world, ms * 1e-3), "unit": "audio samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAMES[tprec], "data": "synthetic U(-0.3,0.3) audio",
            "config": {"workload": args.config + " training", "batch_per_gpu": B, "L": L,
                       "parallelism": "dp%d, bucketed asynchronous RCCL all-reduce of the gradients" % world},
            "per_rank_ms_per_step": per_rank_ms, "process_group": ddist.group_info(),
            "per_rank_final_loss": per_rank_loss, "per_rank_param_digest": per_rank_param_digest,
            "final_loss": float(loss.detach()), **({"dp": dp_overhead} if dp_overhead else {})})
    del net, opt
    torch.cuda.empty_cache()
    if not emit:
        return line
    if rank == 0 and world == 1 and getattr(args, "cpu_train_baseline", False):
        line["cpu_baseline"] = cpu_train_baseline(cfg)
        line["gpu_over_cpu"] = line["value"] / line["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(line))
    ddist.shutdown()
    return line


