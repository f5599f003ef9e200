#!/usr/bin/env python
"""Headline benchmark: audio samples/sec of `generate.py`-style reverse-diffusion
sampling (BASELINE.json `metric`), on BASELINE config 2:

    SC09 unconditional, WaveNet wnet_h256_d36_T200, B=16 per GPU, L=16000

A "step" is one reverse-diffusion step (network forward + x update + noise) over
the whole batch, replayed from the captured hipGraph; `value` is
B*L*n_gpus / (T * seconds_per_step), i.e. finished audio samples per second of
the full T=200 loop, with x, weights and schedule resident in HBM.

    python bench.py --gpus 1 --steps 200 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: sampling shards by independent clips -- every rank runs its own batch
with its own Philox stream and there is NO data-path collective ("weak" scaling;
`generate.py:217-227`).  RCCL is used only for the timing barrier / max.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchlib.configs import (CONFIGS, DTYPE_NAMES, PEAK_BF16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS,  # noqa: E402,F401
                              build_model)
from benchlib.cpu_baseline import _cpu_model, cpu_baseline, cpu_train_baseline  # noqa: E402,F401
from benchlib.sample import sample_bench  # noqa: E402
from benchlib.train import train_bench  # noqa: E402
from benchlib.work import layer_algorithmic_work, wino_executed_work  # noqa: E402,F401


def spawn_ranks(n):
    """Re-execute this command line once per rank (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* in the environment, rendezvous
    on 127.0.0.1); rank 0 inherits stdout, so exactly one JSON line is printed.  Any rank failing fails the job."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        # poll ALL ranks: a rank that dies leaves the others inside a collective, so they are stopped as soon as any
        # exits non-zero (waiting for them in order would sit out the collective timeout on rank 0 first)
        live = list(procs)
        while live and not rc:
            for p in list(live):
                r = p.poll()
                if r is not None:
                    live.remove(p)
                    rc = r or rc
            if live and not rc:
                time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:
                p.kill()
    if rc:
        sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="wnet_h256_d36_T200", choices=list(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--precision", default="bf16x6", choices=["f32", "bf16x3", "bf16x6", "f16x3"],
                    help="matrix arithmetic of the GEMMs: bf16x6 (default) = fp32-EQUIVALENT 3-term bf16 split of every operand, six "
                         "products on the bf16 matrix cores, fp32 accumulate (error against float64 = the exact-f32 path's: "
                         "tests/test_bf16x6_gpu.py, test_full_size_gpu.py, test_split_trajectory_gpu.py); f32 = exact-f32 MFMA "
                         "(timed beside the headline as extra_f32_exact); f16x3 / bf16x3 = narrower experiments, never the headline")
    ap.add_argument("--mode", default="sample", choices=["sample", "train"],
                    help="sample: the headline reverse-diffusion step; train: one DP training step "
                         "(forward_train + backward + RCCL gradient all-reduce + Adam)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-train-baseline", action="store_true",
                    help="also time one training step of the CPU oracle at B=1 beside the training leg / --mode train (~100 s)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs legs of the default headline run")
    ap.add_argument("--no-full-loop", action="store_true",
                    help="skip the complete T-step dws_sampler_run beside the timed steps (profiler passes: counters serialise kernels)")
    args = ap.parse_args()

    cfg = dict(CONFIGS[args.config])
    if args.batch:
        cfg["B"] = args.batch
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves, one process per GPU, as the reference
        # does (`generate.py:220-227`, `train.py:244-251`); under torch.distributed.run the env is already set
        return spawn_ranks(args.gpus)
    from diffwave_sashimi_amd import dist as ddist
    # DWS_BENCH_SHARE_GPU=1 (tests only): all ranks on cuda:0 with gloo, to exercise the N > 1 control flow on a
    # one-GPU box (one device cannot host two RCCL ranks).  Real runs: one rank per GPU over RCCL.
    share = os.environ.get("DWS_BENCH_SHARE_GPU") == "1"
    # DWS_BENCH_FORCE_PG=1: a 1-rank RCCL group at N = 1, so the timing barrier / reductions run over RCCL on a one-GPU box
    force_pg = os.environ.get("DWS_BENCH_FORCE_PG") == "1"
    world, rank, local_rank = ddist.init(("gloo" if share else "nccl")
                                         if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or force_pg) else None, force=force_pg)
    if world == 1 or share:
        torch.cuda.set_device(0)
        local_rank = 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    red_dev = torch.device("cpu") if share else dev     # gloo reduces the timing scalar on the host

    from diffwave_sashimi_amd import _lib
    _lib.load()  # no fallback: fails loudly without the HIP engine
    if args.mode == "train":
        if args.precision not in ("f32", "bf16x6") or cfg["model"]["_name_"] == "wavenet":
            args.precision = "f32"       # WaveNet training and the narrower splits have no split backward
        return train_bench(args, cfg, world, rank, dev, ddist, red_dev)
    result = sample_bench(args, cfg, world, rank, dev, ddist, red_dev)
    default_run = (args.config == "wnet_h256_d36_T200" and not args.batch and args.precision == "bf16x6")
    if rank == 0 and world == 1 and not args.no_extra and default_run:
        # the other BASELINE configs, short legs in the same process (NOT `value`): C3, C4 sampling, C5's per-GPU training step.
        # A failing leg must not take the headline line with it: it is reported as an error string instead.
        result["extra_configs"] = extra_legs(args, world, rank, dev, ddist, red_dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, config_name=None if args.batch else args.config)
        # against the better of the single process and the quota-sized whole-host leg (context only: a GPU/CPU ratio says
        # nothing about kernel quality, the roofline fraction does)
        result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["best"]["value"]
    if rank == 0 and default_run:
        result["summary"] = summary(result)          # LAST key: every config's number survives a tail of the line
    if rank == 0:
        print(json.dumps(result))
    ddist.barrier()   # rank 0's roofline / extra legs are done: every rank leaves the process group together
    ddist.shutdown()


def summary(result):
    """Compact digest (<= 1500 characters) of every BASELINE config's timed number in this run, as the LAST key of the line:
    ms per step at bf16x6 (the headline arithmetic) and at exact f32, each with its dominant kernel's roofline fraction."""
    def r3(v):
        return None if v is None else float("%.4g" % v)

    def leg(d, other):
        o = d.get(other) or {}
        return {"ms": r3(d.get("ms_per_step")), "frac": r3((d.get("roofline") or {}).get("frac")),
                "f32_ms": r3(o.get("ms_per_step")), "f32_frac": r3((o.get("roofline") or {}).get("frac")),
                "finite": d.get("state_finite")}
    out = {"keys": "ms = ms/step at bf16x6 (fp32-equivalent split), frac = roofline.frac of the dominant kernel family; f32_* = exact-f32 leg",
           "C2 wnet_h256_d36 B16": leg(result, "extra_f32_exact")}
    ex = result.get("extra_configs") or {}
    for key, name in (("C3 unet_d64 B16", "unet_d64_n6_T200"), ("C4 unet_d32 cond B32", "unet_d32_n6_T50_cond")):
        if name in ex:
            out[key] = {"error": ex[name]["error"][:80]} if "error" in ex[name] else leg(ex[name], "extra_f32_exact")
    tr = ex.get("unet_d128_n6_T200 --mode train")
    if tr:
        if "error" in tr:
            out["C5 unet_d128 train B32/GPU"] = {"error": tr["error"][:80]}
        else:
            o = tr.get("extra_bf16x6") or {}
            out["C5 unet_d128 train B32/GPU"] = {"ms": r3(o.get("ms_per_step")), "f32_ms": r3(tr.get("ms_per_step")),
                                                 "f32_gemm_frac": r3((tr.get("roofline") or {}).get("frac")),
                                                 "f32_whole_step_frac": r3((tr.get("roofline") or {}).get("whole_step_frac")),
                                                 "exposed_allreduce_ms": r3((tr.get("dp") or {}).get("exposed_ms")),
                                                 "f32_gpu_over_cpu": r3(tr.get("gpu_over_cpu"))}
    cb = result.get("cpu_baseline") or {}
    if cb:
        out["cpu"] = {"samples_per_s": r3((cb.get("best") or {}).get("value")), "cores": cb.get("cores"),
                      "quota_cpus": cb.get("cpu_quota_cpus"), "gpu_over_cpu": r3(result.get("gpu_over_cpu"))}
    return out


def extra_legs(args, world, rank, dev, ddist, red_dev):
    """BASELINE configs 3, 4 (sampling) and 5 (one GPU's share of the DP training step), a few seconds each, so that
    the driver's default run evidences all of them: ms/step, throughput and the per-kernel roofline of each, under the
    headline arithmetic (bf16x6) with the exact-f32 leg beside it."""
    import copy
    out = {}
    for name, steps, warmup in (("unet_d64_n6_T200", 40, 3), ("unet_d32_n6_T50_cond", 50, 3)):
        a = copy.copy(args)
        a.config, a.steps, a.warmup = name, steps, warmup
        t0 = time.perf_counter()
        try:
            r = sample_bench(a, dict(CONFIGS[name]), world, rank, dev, ddist, red_dev, extras=False)
            out[name] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline",
                                           "full_loop", "state_finite", "hbm_bytes_in_use", "conditioner_ms_per_batch",
                                           "conditioner_ms_all_calls", "end_to_end_samples_per_s_incl_conditioner") if k in r}
            if not args.no_cpu_baseline:     # the oracle on this box's host cores beside the leg (B = 1, bounded)
                out[name]["cpu_baseline"] = cpu_baseline(dict(CONFIGS[name]), seconds_budget=10.0, light=True)
                out[name]["gpu_over_cpu"] = out[name]["value"] / out[name]["cpu_baseline"]["value"]
            # the same leg with every tail GEMM in exact-f32 MFMA arithmetic, with its own roofline
            a32 = copy.copy(a)
            a32.precision, a32.steps = "f32", max(steps // 2, 10)
            r32 = sample_bench(a32, dict(CONFIGS[name]), world, rank, dev, ddist, red_dev, extras=False, full=False)
            out[name]["extra_f32_exact"] = {k: r32[k] for k in ("ms_per_step", "value", "unit", "dtype", "roofline", "state_finite")
                                            if k in r32}
            out[name]["extra_f32_exact"]["note"] = "the same steps under precision=f32; not the leg's value"
        except Exception as e:      # noqa: BLE001 -- reported, not swallowed
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        out[name]["leg_seconds"] = time.perf_counter() - t0
    # the reference's own documented operating points (`README.md:215`: unet_d128 sampled at B = 128 per GPU, "the largest
    # batch that fits on an A100"; `README.md:228`: unet_d64 at B = 256): short legs, with the HBM actually in use
    for name, bsz, steps in (("unet_d128_n6_T200", 128, 6), ("unet_d64_n6_T200", 256, 8)):
        a = copy.copy(args)
        a.config, a.steps, a.warmup, a.no_roofline = name, steps, 2, True
        key = "%s B=%d (README operating point)" % (name, bsz)
        t0 = time.perf_counter()
        try:
            r = sample_bench(a, dict(CONFIGS[name], B=bsz), world, rank, dev, ddist, red_dev, extras=False, full=False)
            out[key] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config",
                                          "state_finite", "hbm_bytes_in_use") if k in r}
        except Exception as e:      # noqa: BLE001
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        out[key]["leg_seconds"] = time.perf_counter() - t0
    a = copy.copy(args)
    a.config, a.steps, a.warmup, a.mode, a.batch, a.precision = "unet_d128_n6_T200", 4, 2, "train", None, "f32"
    t0 = time.perf_counter()
    key = "unet_d128_n6_T200 --mode train"
    try:
        r = train_bench(a, dict(CONFIGS[a.config]), world, rank, dev, ddist, red_dev, emit=False)
        out[key] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline",
                                      "final_loss", "dp") if k in r}
        # the same step with its pointwise GEMMs and weight gradients on the bf16 matrix cores (exact 3-term split, six
        # products: tapconv_mfma_kernel<.., SPLIT>, wgrad_dma4_kernel<1>; gradients against float64 within 2x the f32 path's,
        # tests/test_sashimi_training_gpu.py)
        a6 = copy.copy(a)
        a6.precision = "bf16x6"
        prev = os.environ.get("DWS_BENCH_NO_DP_OVERHEAD")
        os.environ["DWS_BENCH_NO_DP_OVERHEAD"] = "1"          # (the DP-overhead measurement belongs to the f32 leg above)
        try:
            r6 = train_bench(a6, dict(CONFIGS[a.config]), world, rank, dev, ddist, red_dev, emit=False)
        finally:
            if prev is None:
                del os.environ["DWS_BENCH_NO_DP_OVERHEAD"]
            else:
                os.environ["DWS_BENCH_NO_DP_OVERHEAD"] = prev
        if r6:
            out[key]["extra_bf16x6"] = {"ms_per_step": r6["ms_per_step"], "value": r6["value"], "unit": r6["unit"],
                                        "dtype": r6["dtype"], "final_loss": r6["final_loss"],
                                        "note": "precision=bf16x6 (fp32-equivalent split); the leg's value is the exact-f32 step"}
        if not args.no_cpu_baseline:
            if args.cpu_train_baseline:
                out[key]["cpu_baseline"] = cpu_train_baseline(dict(CONFIGS[a.config]))
                out[key]["gpu_over_cpu"] = out[key]["value"] / out[key]["cpu_baseline"]["value"]
            else:   # one oracle training step of unet_d128 at B = 1, L = 16000 is ~100 s of host time: the default run times a
                    # BOUNDED sample instead -- one step on a 4000-sample clip through the same network (~25 s)
                out[key]["cpu_baseline"] = cpu_train_baseline(dict(CONFIGS[a.config]), sample_L=4000)
                out[key]["gpu_over_cpu"] = out[key]["value"] / out[key]["cpu_baseline"]["value"]
    except Exception as e:      # noqa: BLE001
        out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    out[key]["leg_seconds"] = time.perf_counter() - t0
    return out


if __name__ == "__main__":
    main()

