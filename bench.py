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

CONFIGS = {
    # BASELINE.json configs[1]
    "wnet_h256_d36_T200": dict(
        model=dict(_name_="wavenet", unconditional=True, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, res_channels=256, skip_channels=256,
                   num_res_layers=36, dilation_cycle=12),
        diffusion=dict(T=200, beta_0=1e-4, beta_T=0.02), B=16, L=16000),
    # BASELINE.json configs[0] (the reference's CPU-runnable case)
    "wnet_h128_d30_T200": dict(
        model=dict(_name_="wavenet", unconditional=True, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, res_channels=128, skip_channels=256,
                   num_res_layers=30, dilation_cycle=10),
        diffusion=dict(T=200, beta_0=1e-4, beta_T=0.02), B=16, L=16000),
    # BASELINE.json configs[2]
    "unet_d64_n6_T200": dict(
        model=dict(_name_="sashimi", unconditional=True, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, unet=True, d_model=64, n_layers=6, pool=[4, 4],
                   expand=2, ff=2, L=16000),
        diffusion=dict(T=200, beta_0=1e-4, beta_T=0.02), B=16, L=16000),
    # sampling with the architecture of BASELINE.json configs[4] (unet_d128_n6; README.md:215 samples it at B=128/GPU)
    "unet_d128_n6_T200": dict(
        model=dict(_name_="sashimi", unconditional=True, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, unet=True, d_model=128, n_layers=6, pool=[4, 4],
                   expand=2, ff=2, L=16000),
        diffusion=dict(T=200, beta_0=1e-4, beta_T=0.02), B=16, L=16000),
    # BASELINE.json configs[3] (mel conditioner installed once per utterance)
    "unet_d32_n6_T50_cond": dict(
        model=dict(_name_="sashimi", unconditional=False, in_channels=1, out_channels=1,
                   diffusion_step_embed_dim_in=128, diffusion_step_embed_dim_mid=512,
                   diffusion_step_embed_dim_out=512, unet=True, d_model=32, n_layers=6, pool=[4, 4],
                   expand=2, ff=2, L=16000, mel_upsample=[16, 16]),
        diffusion=dict(T=50, beta_0=1e-4, beta_T=0.05), B=32, L=16000, Tmel=63),
}

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA (three bf16 MFMAs per fp32-equivalent product)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


def build_model(cfg, device):
    """Random-init weights of the named architecture (no checkpoints exist offline):
    reference initialisers under manual_seed(0), final zero-conv re-initialised
    N(0, 0.1^2) so the network output is not identically zero."""
    from diffwave_sashimi_amd.models import construct_model
    torch.manual_seed(0)
    net = construct_model(dict(cfg["model"]))
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        sd = net.state_dict()
        for k in ("final_conv.2.conv.weight", "final_conv.2.conv.bias"):
            sd[k].copy_(torch.randn(sd[k].shape, generator=g) * 0.1)
    return net.to(device).eval()


def layer_algorithmic_work(cfg):
    """Per launch of the fused residual-layer kernel:
    flops = B*L*(14 C^2 + 2 C S); compulsory HBM bytes = B*L*4*(2C + 2S)."""
    m = cfg["model"]
    C, S, B, L = m["res_channels"], m["skip_channels"], cfg["B"], cfg["L"]
    return B * L * (14 * C * C + 2 * C * S), B * L * 4 * (2 * C + 2 * S)


def wino_executed_work(cfg):
    """MFMA flops the Winograd layer kernel (csrc/wavenet_wino.hip) really executes per launch, averaged over the
    dilations of the stack: workgroups(d) = B * ceil(ceil(L / 2d) * d / 32) tiles of 32 position pairs, C/32 waves each,
    per wave C/2 k-steps x 8 MFMAs (4 Winograd products x the tanh and the sigmoid row tile) + 8 (step-embedding / bias
    k-step) for the convolution and (C/2 + 1) k-steps x 2 column tiles x (1 + S/C) row tiles for [res; skip];
    4096 flop per v_mfma_f32_32x32x2_f32.  (Direct-conv algorithmic flops: layer_algorithmic_work.)"""
    m = cfg["model"]
    C, S, B, L = m["res_channels"], m["skip_channels"], cfg["B"], cfg["L"]
    per_wave = (C // 2) * 8 + 8 + (C // 2 + 1) * 2 * (1 + S // C)
    tot = 0
    dil = [1 << (n % m["dilation_cycle"]) for n in range(m["num_res_layers"])]
    for d in dil:
        nblk = -(-L // (2 * d))
        tot += B * (-(-(nblk * d) // 32)) * (C // 32) * per_wave * 4096
    return tot / len(dil)


def wino_dgrad_executed_work(cfg):
    """MFMA flops the Winograd data-gradient kernel of the dilated conv (csrc/wavenet_backward_wino.hip) executes per
    launch, averaged over the dilations: workgroups(d) = B * ceil(ceil(L / 2d) * d / 64) tiles of 64 position pairs x
    C / (128 MT) row blocks, 8 waves x C k-steps (K = 2C) x 4 Winograd products x MT row tiles each = 1024 C^2 flop per
    tile column block.  (The direct form: 12 C^2 flop per position.)"""
    m = cfg["model"]
    C, B, L = m["res_channels"], cfg["B"], cfg["L"]
    tot = 0
    dil = [1 << (n % m["dilation_cycle"]) for n in range(m["num_res_layers"])]
    for d in dil:
        nblk = -(-L // (2 * d))
        tot += B * (-(-(nblk * d) // 64)) * 1024 * C * C
    return tot / len(dil)


def wino_wgrad_executed_work(cfg):
    """MFMA flops of the dilated conv's weight gradient in the Winograd pairing (wgrad_wino_kernel): four [2C x C] GEMMs
    over the pair columns, chunks of 64, instead of three over all positions (12 C^2 flop per position)."""
    m = cfg["model"]
    C, B, L = m["res_channels"], cfg["B"], cfg["L"]
    tot = 0
    dil = [1 << (n % m["dilation_cycle"]) for n in range(m["num_res_layers"])]
    for d in dil:
        nblk = -(-L // (2 * d))
        tot += B * (-(-(nblk * d) // 64)) * 64 * 16 * C * C
    return tot / len(dil)


def sashimi_tail_work(cfg):
    """All S4-tail launches of one step: per block 12 H^2 flops and 12 H bytes per position
    (read g and x, write out; the three GEMMs Wo, W1, W2), summed over the U-Net's blocks."""
    m = cfg["model"]
    H, L, B = m["d_model"], cfg["L"], cfg["B"]
    flops = bytes_ = 0
    n_down = []
    for p in m["pool"]:
        n_down.append((H, L))
        L //= p
        H *= m["expand"]
    blocks = [(H, L)] * m["n_layers"]
    for (h, l) in n_down:
        blocks += [(h, l)] * (m["n_layers"] * (2 if m.get("unet", True) else 1))
    for h, l in blocks:
        flops += 12 * h * h * l * B
        bytes_ += 12 * h * l * B
    return flops, bytes_, len(blocks)


def forward_gemm_flops(cfg, B):
    """Algorithmic flops of the dense contractions of ONE forward over a batch of B: WaveNet
    B L [n (14 C^2 + 2 C S) + 2 S^2]; SaShiMi: 12 H^2 per position per block + 4 H_in H_out-style pool GEMMs + the final
    D x D conv.  A training step (forward, data gradients, weight gradients) is 3x this."""
    m, L = cfg["model"], cfg["L"]
    if m["_name_"] == "wavenet":
        C, S, n = m["res_channels"], m["skip_channels"], m["num_res_layers"]
        return B * L * (n * (14 * C * C + 2 * C * S) + 2 * S * S)
    flops, _, _ = sashimi_tail_work(dict(cfg, B=B))
    H, Ls = m["d_model"], L
    for p in m["pool"]:      # DownPool (H p -> H e) at L/p positions, UpPool (H e -> H p) at L/p positions
        flops += 2 * 2 * (H * p) * (H * m["expand"]) * (Ls // p) * B
        H, Ls = H * m["expand"], Ls // p
    return flops + 2 * m["d_model"] ** 2 * L * B


def cpu_baseline(cfg, seconds_budget=25.0, light=False, config_name=None):
    """The oracle (reference-equivalent PyTorch-CPU graph: conv1d per layer, weight-norm
    per call, no hoisting) timed on this box's host cores at B=1; bounded sample.
    MKL-DNN convolutions of this size get *slower* with hundreds of threads, so a few
    thread counts are probed first and the best one is used (`cores` = threads used).
    light: the short form beside an extra_configs leg (B=1 only, at least two timed steps, no single-thread leg)."""
    from oracle import sashimi as osa
    from oracle import wavenet as own
    from diffwave_sashimi_amd.models import construct_model
    ncpu = os.cpu_count() or 1
    torch.manual_seed(0)
    net = construct_model(dict(cfg["model"]))
    sd = {k: v.detach() for k, v in net.state_dict().items()}
    L, T = cfg["L"], cfg["diffusion"]["T"]
    audio = torch.randn(1, 1, L)
    steps = torch.full((1, 1), float(T - 1))
    mel = None
    if "Tmel" in cfg:
        mel = torch.rand(1, 80, cfg["Tmel"]) * 13.5 - 11.5
    fwd = own.wavenet_forward if cfg["model"]["_name_"] == "wavenet" else osa.sashimi_forward

    def one():
        t0 = time.perf_counter()
        with torch.no_grad():
            fwd(sd, cfg["model"], audio, steps, mel_spec=mel)
        return time.perf_counter() - t0

    t_begin = time.perf_counter()
    best, best_t = None, float("inf")
    if light:
        # beside an extra leg: one thread count, and a first step that already takes > 6 s IS the sample (SaShiMi regenerates
        # its S4 kernels in every call, 88 % of a step: there is nothing to warm up)
        best = min(16, ncpu)
        torch.set_num_threads(best)
        times = [one()]
        if times[0] <= 6.0:
            times = [one(), one()]
        per_step = sum(times) / len(times)
        return {"value": L / (T * per_step), "unit": "audio samples/s", "cores": best, "host_cpus": ncpu, "kind": "port",
                "sample": f"{len(times)} forward step(s) at B=1, L={L} with {best} threads, extrapolated to the T={T} loop",
                "ms_per_step_b1": per_step * 1e3, "cpu_model": _cpu_model()}
    for th in [c for c in (8, 16, 32, 64) if c <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        one()                      # warm-up at this thread count
        t = one()
        if t < best_t:
            best, best_t = th, t
        if time.perf_counter() - t_begin > seconds_budget * 0.6 or t > 2.5 * best_t:
            break
    torch.set_num_threads(best)
    one()
    times = []
    while len(times) < 3 or (time.perf_counter() - t_begin < seconds_budget and len(times) < 10):
        times.append(one())
        if time.perf_counter() - t_begin > 2 * seconds_budget:
            break
    per_step = sum(times) / len(times)
    out = {"value": L / (T * per_step), "unit": "audio samples/s", "cores": best, "host_cpus": ncpu, "kind": "port",
           "sample": f"{len(times)} forward steps at B=1, L={L} with {best} threads (best of a probe over 8..64), "
                     f"extrapolated to the T={T} loop",
           "ms_per_step_b1": per_step * 1e3, "cpu_model": _cpu_model()}
    # the config's own batch (SURVEY.md 8d asks for B=1 and the config's B): one warm-up + up to 2 timed steps, bounded
    Bc = cfg["B"]
    if Bc > 1 and per_step * Bc < 40.0:
        audio_b, steps_b = torch.randn(Bc, 1, L), torch.full((Bc, 1), float(T - 1))
        mel_b = None if mel is None else mel.expand(Bc, -1, -1).contiguous()

        def one_b():
            t0 = time.perf_counter()
            with torch.no_grad():
                fwd(sd, cfg["model"], audio_b, steps_b, mel_spec=mel_b)
            return time.perf_counter() - t0

        first = one_b()
        # MKL-DNN's B > 1 convolutions can be far slower per clip than B = 1: if the first (warm-up) step already took
        # > 12 s it IS the sample; otherwise one or two more steps are timed
        tbs = [first] if first > 12.0 else [one_b()]
        if tbs[0] < 6.0:
            tbs.append(one_b())
        tb = sum(tbs) / len(tbs)
        out["at_config_batch"] = {"B": Bc, "value": Bc * L / (T * tb), "ms_per_step": tb * 1e3, "steps_timed": len(tbs),
                                  "warm": first <= 12.0}
    if per_step * best < 20.0:     # single-thread figure when one step is predicted to fit in ~20 s
        torch.set_num_threads(1)
        t1 = one()
        out["single_thread_value"] = L / (T * t1)
        torch.set_num_threads(best)
    if config_name is not None and per_step < 8.0:
        out["whole_host"] = cpu_whole_host(config_name, best)
    return out


_WHOLE_HOST_WORKER = r'''
import json, os, sys, time
cpus = [int(c) for c in os.environ["DWS_CPUSET"].split(",")]
os.sched_setaffinity(0, cpus)
os.environ["OMP_NUM_THREADS"] = str(len(cpus))
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch
torch.set_num_threads(len(cpus))
import bench
from oracle import sashimi as osa, wavenet as own
from diffwave_sashimi_amd.models import construct_model
cfg = bench.CONFIGS[os.environ["DWS_CONFIG"]]
torch.manual_seed(0)
sd = {k: v.detach() for k, v in construct_model(dict(cfg["model"])).state_dict().items()}
L, T = cfg["L"], cfg["diffusion"]["T"]
audio, steps = torch.randn(1, 1, L), torch.full((1, 1), float(T - 1))
mel = torch.rand(1, 80, cfg["Tmel"]) * 13.5 - 11.5 if "Tmel" in cfg else None
fwd = own.wavenet_forward if cfg["model"]["_name_"] == "wavenet" else osa.sashimi_forward
def one():
    t0 = time.perf_counter()
    with torch.no_grad():
        fwd(sd, cfg["model"], audio, steps, mel_spec=mel)
    return time.perf_counter() - t0
one()
open(os.environ["DWS_READY"], "w").close()                    # warmed up: wait for the common start
while not os.path.exists(os.environ["DWS_GO"]):
    time.sleep(0.01)
ts = [one() for _ in range(int(os.environ["DWS_NSTEPS"]))]
print(json.dumps({"steps_s": ts}))
'''


def _physical_cores():
    """One logical CPU per physical core (SMT siblings dropped), from /proc/cpuinfo; all logical CPUs if that fails."""
    try:
        seen, cur = {}, {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                seen.setdefault((cur.get("physical id"), cur.get("core id")), int(cur["processor"]))
                cur = {}
        if cur:
            seen.setdefault((cur.get("physical id"), cur.get("core id")), int(cur["processor"]))
        allowed = os.sched_getaffinity(0)
        cores = sorted(c for c in seen.values() if c in allowed)
        return cores or sorted(allowed)
    except Exception:   # noqa: BLE001
        return sorted(os.sched_getaffinity(0))


def cpu_whole_host(config_name, threads, nsteps=2):
    """BASELINE.md section 2's CPU figure: the WHOLE host, as N concurrent B = 1 workers of `threads` threads each, every
    worker pinned to its own physical cores (one process of hundreds of threads is slower than 16: MKL-DNN's convolutions
    of this size do not scale).  All workers warm up, start their timed steps together, and the aggregate rate is
    N x L / (T x the slowest worker's mean step)."""
    import subprocess
    import tempfile
    cores = _physical_cores()
    n = max(1, len(cores) // threads)
    cfg = CONFIGS[config_name]
    L, T = cfg["L"], cfg["diffusion"]["T"]
    tmp = tempfile.mkdtemp(prefix="dws_whole_host_")
    go = os.path.join(tmp, "go")
    procs = []
    for w in range(n):
        cs = cores[w * threads:(w + 1) * threads]
        env = dict(os.environ, DWS_CPUSET=",".join(map(str, cs)), DWS_ROOT=ROOT, DWS_CONFIG=config_name,
                   DWS_READY=os.path.join(tmp, "ready%d" % w), DWS_GO=go, DWS_NSTEPS=str(nsteps))
        procs.append(subprocess.Popen([sys.executable, "-c", _WHOLE_HOST_WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    t0 = time.perf_counter()
    try:
        while not all(os.path.exists(os.path.join(tmp, "ready%d" % w)) for w in range(n)):
            if time.perf_counter() - t0 > 240 or any(p.poll() not in (None, 0) for p in procs):
                raise RuntimeError("a whole-host worker did not come up: " + "; ".join((p.stderr.read() or "")[-300:]
                                                                                       for p in procs if p.poll() not in (None, 0)))
            time.sleep(0.05)
        open(go, "w").close()
        means = []
        for p in procs:
            o, e = p.communicate(timeout=600)
            if p.returncode != 0:
                raise RuntimeError(e[-500:])
            ts = json.loads(o.strip().splitlines()[-1])["steps_s"]
            means.append(sum(ts) / len(ts))
    except Exception as e:   # noqa: BLE001 -- reported in the line, the headline survives
        for p in procs:
            if p.poll() is None:
                p.kill()
        return {"error": "%s: %s" % (type(e).__name__, e)}
    slow = max(means)
    return {"value": n * L / (T * slow), "unit": "audio samples/s", "workers": n, "threads_per_worker": threads,
            "cores": n * threads, "physical_cores": len(cores), "steps_per_worker": nsteps,
            "ms_per_step_b1_slowest_worker": slow * 1e3, "ms_per_step_b1_fastest_worker": min(means) * 1e3,
            "sample": "%d concurrent B=1 workers x %d threads, each pinned to its own physical cores; %d forward steps each "
                      "after a warm-up, common start; rate = workers x L / (T x slowest worker's mean step)" % (n, threads, nsteps)}


def cpu_train_baseline(cfg, seconds_budget=30.0):
    """One `train.py:118-143`-style step of the oracle on the host cores at B=1: q-sample, forward, MSE against the noise,
    backward through torch autograd of the reference-equivalent CPU graph (no optimizer: its cost is negligible beside
    the backward).  Bounded sample: a first step that already takes > 8 s IS the sample (it includes the one-time
    allocator warm-up), otherwise a second step is timed."""
    from oracle import sashimi as osa
    from oracle import wavenet as own
    from diffwave_sashimi_amd.models import construct_model
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    ncpu = os.cpu_count() or 1
    th = min(32, ncpu)
    torch.set_num_threads(th)
    torch.manual_seed(0)
    net = construct_model(dict(cfg["model"]))
    leaf = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v.clone())
            for k, v in net.state_dict().items()}
    fwd = own.wavenet_forward if cfg["model"]["_name_"] == "wavenet" else osa.sashimi_forward
    L, T = cfg["L"], cfg["diffusion"]["T"]
    dh = calc_diffusion_hyperparams(**cfg["diffusion"])
    g = torch.Generator().manual_seed(7)
    audio = (torch.rand(1, 1, L, generator=g) * 2 - 1) * 0.3

    def one():
        t0 = time.perf_counter()
        for v in leaf.values():
            if v.is_floating_point():
                v.grad = None
        ts = torch.randint(T, (1, 1, 1), generator=g)
        z = torch.randn(audio.shape, generator=g)
        ab = dh["Alpha_bar"][ts]
        xt = torch.sqrt(ab) * audio + torch.sqrt(1 - ab) * z          # `train.py:221`
        eps = fwd(leaf, cfg["model"], xt, ts.view(1, 1))
        loss = torch.nn.functional.mse_loss(eps, z)
        loss.backward()
        return time.perf_counter() - t0

    times = [one()]
    if times[0] < 8.0 or times[0] * 2 < seconds_budget:
        times.append(one())
    t = times[-1]
    return {"value": L / t, "unit": "training audio samples/s", "cores": th, "host_cpus": ncpu, "kind": "port",
            "sample": "%d training step(s) (q-sample + forward + MSE + autograd backward of the oracle) at B=1, L=%d with %d "
                      "threads; the last one is reported" % (len(times), L, th),
            "ms_per_step_b1": t * 1e3, "steps_ms": [x * 1e3 for x in times], "cpu_model": _cpu_model()}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
                       "buckets": len(red.buckets), "bucket_mbytes": [b.flat.numel() * 4 / 2 ** 20 for b in red.buckets],
                       "gradient_slots": red.last_stats}
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
                           "allreduce_ms": red.allreduce_ms()}
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
            "final_loss": float(loss), **({"dp": dp_overhead} if dp_overhead else {})})
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


DTYPE_NAMES = {
    "f32": "f32",
    "bf16x3": "bf16x3 split (hi/lo bf16 MFMA inputs, fp32 accumulate; ~1e-5 rel)",
    "bf16x6": "f32-equivalent (3-term bf16 split, 6 products, fp32 accumulate)",
    "f16x3": "f32-class (2-term fp16 split of power-of-two scaled operands, 3 products, fp32 accumulate; 22 bits per operand)",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="wnet_h256_d36_T200", choices=list(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the config's)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16x6", "f16x3"],
                    help="WaveNet matrix arithmetic: exact-f32 MFMA (default); bf16x6 = fp32-equivalent 3-term bf16 split "
                         "(six products, Winograd form); f16x3 = 2-term fp16 split of scaled operands (three products, same kernel and "
                         "same float64 acceptance); bf16x3 = 2-term bf16 split (~1e-5, narrower than fp32)")
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
        return train_bench(args, cfg, world, rank, dev, ddist, red_dev)
    result = sample_bench(args, cfg, world, rank, dev, ddist, red_dev)
    if (rank == 0 and world == 1 and not args.no_extra and args.config == "wnet_h256_d36_T200" and not args.batch
            and args.precision == "f32"):
        # the other BASELINE configs, short legs in the same process (NOT `value`): C3, C4 sampling, C5's per-GPU training step.
        # A failing leg must not take the headline line with it: it is reported as an error string instead.
        result["extra_configs"] = extra_legs(args, world, rank, dev, ddist, red_dev)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(cfg, config_name=None if args.batch else args.config)
        result["gpu_over_cpu"] = result["value"] / result["cpu_baseline"]["value"]
        wh = result["cpu_baseline"].get("whole_host") or {}
        if "value" in wh:      # against ALL physical cores of the host, not the best single process
            result["gpu_over_cpu_whole_host"] = result["value"] / wh["value"]
    if rank == 0:
        print(json.dumps(result))
    ddist.barrier()   # rank 0's roofline / extra legs are done: every rank leaves the process group together
    ddist.shutdown()


def extra_legs(args, world, rank, dev, ddist, red_dev):
    """BASELINE configs 3, 4 (sampling) and 5 (one GPU's share of the DP training step), a few seconds each, so that
    the driver's default run evidences all of them: ms/step, throughput and the per-kernel roofline of each."""
    import copy
    out = {}
    for name, steps, warmup in (("unet_d64_n6_T200", 40, 3), ("unet_d32_n6_T50_cond", 50, 3)):
        a = copy.copy(args)
        a.config, a.steps, a.warmup = name, steps, warmup
        t0 = time.perf_counter()
        try:
            r = sample_bench(a, dict(CONFIGS[name]), world, rank, dev, ddist, red_dev, extras=False)
            out[name] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline",
                                           "full_loop", "hbm_bytes_in_use", "conditioner_ms_per_batch", "conditioner_ms_all_calls",
                                           "end_to_end_samples_per_s_incl_conditioner") if k in r}
            if not args.no_cpu_baseline:     # the oracle on this box's host cores beside the leg (B = 1, bounded)
                out[name]["cpu_baseline"] = cpu_baseline(dict(CONFIGS[name]), seconds_budget=10.0, light=True)
                out[name]["gpu_over_cpu"] = out[name]["value"] / out[name]["cpu_baseline"]["value"]
            # the same leg with the H <= 64 tails on the bf16 matrix cores at fp32-equivalent accuracy (3-term split, six
            # products: csrc/sashimi_chain6.hip; error vs float64 = the f32 path's, tests/test_sashimi_bf16x6_gpu.py)
            a6 = copy.copy(a)
            a6.precision, a6.no_roofline, a6.steps = "bf16x6", True, max(steps // 2, 10)
            r6 = sample_bench(a6, dict(CONFIGS[name]), world, rank, dev, ddist, red_dev, extras=False, full=False)
            out[name]["extra_bf16x6"] = {"ms_per_step": r6["ms_per_step"], "value": r6["value"], "unit": r6["unit"],
                                         "dtype": "f32, register-chained tails (H <= 128) f32-equivalent on bf16 MFMA "
                                                  "(3-term split, 6 products, fp32 accumulate)",
                                         "note": "opt-in precision=bf16x6; not the leg's value"}
            # and with the 2-term fp16 split of scaled operands (three products; the same kernels and float64 criterion)
            a3 = copy.copy(a6)
            a3.precision = "f16x3"
            r3 = sample_bench(a3, dict(CONFIGS[name]), world, rank, dev, ddist, red_dev, extras=False, full=False)
            out[name]["extra_f16x3"] = {"ms_per_step": r3["ms_per_step"], "value": r3["value"], "unit": r3["unit"],
                                        "dtype": "f32, register-chained tails (H <= 128) on fp16 MFMA (2-term split of "
                                                 "power-of-two scaled operands, 3 products, fp32 accumulate)",
                                        "note": "opt-in precision=f16x3; not the leg's value"}
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
                                          "hbm_bytes_in_use") if k in r}
        except Exception as e:      # noqa: BLE001
            out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
        out[key]["leg_seconds"] = time.perf_counter() - t0
    a = copy.copy(args)
    a.config, a.steps, a.warmup, a.mode, a.batch = "unet_d128_n6_T200", 4, 2, "train", None
    t0 = time.perf_counter()
    key = "unet_d128_n6_T200 --mode train"
    try:
        r = train_bench(a, dict(CONFIGS[a.config]), world, rank, dev, ddist, red_dev, emit=False)
        out[key] = {k: r[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline",
                                      "final_loss", "dp") if k in r}
        # the same step with its pointwise GEMMs and weight gradients on the bf16 matrix cores (exact 3-term split, six
        # products: tapconv_mfma_kernel<.., SPLIT>, wgrad_dma4_kernel<1>; gradients against float64 within 2x the f32 path's,
        # tests/test_sashimi_training_gpu.py).  Opt-in, not the leg's value.
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
                                        "note": "opt-in precision=bf16x6; not the leg's value"}
        if not args.no_cpu_baseline:
            if args.cpu_train_baseline:
                out[key]["cpu_baseline"] = cpu_train_baseline(dict(CONFIGS[a.config]))
                out[key]["gpu_over_cpu"] = out[key]["value"] / out[key]["cpu_baseline"]["value"]
            else:   # one oracle training step of unet_d128 at B = 1 is ~100 s of host time: beyond a default run's budget
                out[key]["cpu_baseline"] = {"skipped": "one oracle training step (forward + autograd backward) at B=1 takes "
                                                       "~100 s on the host: run with --cpu-train-baseline "
                                                       "(measured once per round: profiles/r04_bench_c5train_cpu.json)"}
    except Exception as e:      # noqa: BLE001
        out[key] = {"error": "%s: %s" % (type(e).__name__, e)}
        torch.cuda.empty_cache()
    out[key]["leg_seconds"] = time.perf_counter() - t0
    return out


def wavenet_traffic(precision, kname, executed):
    """PMC-derived HBM bytes per launch of the WaveNet layer kernel (tools/r05_traffic.sh: rocprofv3 in separate --pmc
    passes on the bench command, corrected as the guide prescribes).  The newest profiles/r*_wavenet_traffic_<precision>.json
    is used only if it was measured on THIS kernel: same name, and SQ_INSTS_MFMA x (flops per instruction) within 1 % of
    the executed flops (bf16x6: 32768 flops per bf16 MFMA, six per fp32-equivalent term; its correction and bias k-blocks
    add 1.6 % to the count, so the window is 3 % there) -- a file left over from another kernel version is refused, not
    silently reported.  Returns (bytes or None, file name or the reason of the refusal)."""
    import glob
    # (both split precisions run the same kernel template: the MFMA count per launch tells a bf16x6 file from an f16x3 one)
    per_inst, tol = {"f32": (4096.0, 0.01), "bf16x6": (32768.0 / 6.0, 0.03), "f16x3": (32768.0 / 3.0, 0.03)}[precision]
    tfiles = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_wavenet_traffic_%s.json" % precision)), reverse=True)
    if precision == "f32":
        tfiles += sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_wavenet_traffic.json")), reverse=True)
    for tfile in tfiles:
        tj = json.load(open(tfile))
        cnt = tj.get("sq_insts_mfma_per_launch")
        if not tj.get("kernel", "").startswith(kname):
            return None, "%s refused: measured on %s" % (os.path.basename(tfile), tj.get("kernel"))
        if cnt is None or abs(cnt * per_inst / executed - 1) > tol:
            return None, "%s refused: SQ_INSTS_MFMA x %.0f = %s vs executed flops %.4g" % (
                os.path.basename(tfile), per_inst, cnt and "%.4g" % (cnt * per_inst), executed)
        return tj["hbm_bytes_per_launch"], os.path.basename(tfile)
    return None, None


def profiled_step_ms(lib, run_eager_steps, name, nprof, per_step):
    """ms one step spends in the kernels whose name contains `name`: `nprof` eager steps are timed launch by launch
    (HIP events on the launch stream) and every launch position of the step takes its MEDIAN over the repeats -- the
    first eager step after graph replays carries cold caches and lazily created events.  None if the launch count is
    not nprof * per_step."""
    import ctypes
    from diffwave_sashimi_amd import _lib
    _lib.check(lib.dws_profile_enable(name))
    run_eager_steps(nprof)
    torch.cuda.synchronize()
    n = ctypes.c_int64()
    buf = (ctypes.c_double * (nprof * per_step))()
    _lib.check(lib.dws_profile_query_each(buf, nprof * per_step, ctypes.byref(n)))
    lib.dws_profile_disable()
    if n.value != nprof * per_step:
        return None
    tot = 0.0
    for i in range(per_step):
        v = sorted(buf[r * per_step + i] for r in range(nprof))
        tot += v[len(v) // 2]
    return tot


def sample_bench(args, cfg, world, rank, dev, ddist, red_dev, extras=True, full=True):
    """One sampling measurement (the headline, or an extra leg): returns the result line as a dict (rank 0 prints it)."""
    import ctypes
    import numpy as np
    from diffwave_sashimi_amd import _lib
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams

    lib = _lib.load()
    B, L = cfg["B"], cfg["L"]
    dcfg = cfg["diffusion"]
    T = dcfg["T"]
    net = build_model(cfg, dev)
    if args.precision != "f32":
        net.set_option("precision", args.precision)
    dh = calc_diffusion_hyperparams(**dcfg)
    tabs = [np.ascontiguousarray(dh[k].numpy()) for k in ("Alpha", "Alpha_bar", "Sigma")]
    ptabs = [t.ctypes.data_as(ctypes.POINTER(ctypes.c_float)) for t in tabs]
    net._sync_params()
    net._prepare(B, L)
    if "Tmel" in cfg:  # vocoder config: mel [B, 80, 63] ~ U(-11.5, 2), installed once (hoisted conditioner)
        gm = torch.Generator().manual_seed(4321 + rank)
        mel = (torch.rand(B, 80, cfg["Tmel"], generator=gm) * 13.5 - 11.5).to(dev)
        net._set_condition(mel)
        # the conditioner runs once per batch of utterances, not per reverse step: timed on its own (outside `value`,
        # whose unit is the per-step rate) so that the end-to-end rate of a whole T-step run can be stated beside it
        cond_all = []
        for _ in range(6):
            m2 = mel + 0.0                  # a new tensor each time: the module caches on the mel it was given
            torch.cuda.synchronize()
            tc = time.perf_counter()
            net._set_condition(m2)
            torch.cuda.synchronize()
            cond_all.append((time.perf_counter() - tc) * 1e3)
        cond_ms = sorted(cond_all[1:])[len(cond_all[1:]) // 2]      # median of five after one more untimed call
    x = torch.randn(B, 1, L, device=dev, generator=torch.Generator(device=dev).manual_seed(1234 + rank))
    stream = _lib.current_stream()
    seed = ddist.rank_seed(1234, rank)

    def run(n_steps):
        """n_steps reverse steps of the T-loop (wrapping to t=T-1 when the loop ends)."""
        done = 0
        while done < n_steps:
            k = min(T, n_steps - done)
            _lib.check(lib.dws_sampler_steps(net._handle, x.data_ptr(), *ptabs, T, T - 1, k, seed, 1, stream))
            done += k

    barrier = ddist.barrier

    run(max(args.warmup, 1))  # >= 1: captures the graph outside the timed region
    barrier()
    t0 = time.perf_counter()
    run(args.steps)
    barrier()
    mine = time.perf_counter() - t0
    elapsed = ddist.max_over_ranks(mine, red_dev)
    per_rank_ms = [t / args.steps * 1e3 for t in ddist.gather_over_ranks(mine, red_dev)]
    ms_per_step = elapsed / args.steps * 1e3
    value = ddist.aggregate_throughput(B * L / T, world, ms_per_step * 1e-3)

    result = {
        "metric": "audio samples/sec (generate.py-style reverse-diffusion sampling, T=%d)" % T,
        "value": value, "unit": "audio samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_NAMES[args.precision], "data": "synthetic (seeded reference initialisers, x_T ~ N(0,1))",
        "config": {"workload": args.config, "backbone": cfg["model"]["_name_"], "batch_per_gpu": B, "L": L, "T": T,
                   "parallelism": "independent clips per GPU, no collective",
                   "sampler": "hipGraph replay, on-device Philox noise"},
        "per_rank_ms_per_step": per_rank_ms, "process_group": ddist.group_info(),
        # every rank samples its own clips from its own Philox stream (`generate.py:217-227`): seeds and a digest of
        # each rank's state after the timed steps, in rank order
        "per_rank_seed": [int(v) for v in ddist.gather_over_ranks(float(seed), red_dev)],
        "per_rank_state_digest": ddist.gather_over_ranks(float(x.double().abs().sum()), red_dev),
    }

    free_b, total_b = torch.cuda.mem_get_info()
    result["hbm_bytes_in_use"] = int(total_b - free_b)     # device-wide (engine workspaces are hipMalloc'ed, not torch's)

    if rank == 0 and full and not getattr(args, "no_full_loop", False):
        # the metric as `generate.py:49-54` defines it: ONE complete T-step loop -- Philox draw of x_T, then T replays of the
        # captured step -- wall-clocked end to end on this rank (launch to synchronize), beside the per-step rate above
        torch.cuda.synchronize()
        tl = time.perf_counter()
        _lib.check(lib.dws_sampler_run(net._handle, x.data_ptr(), *ptabs, T, None, seed, 1, 1, stream))
        torch.cuda.synchronize()
        loop_ms = (time.perf_counter() - tl) * 1e3
        result["full_loop"] = {
            "what": "one complete dws_sampler_run: on-device Philox x_T + T=%d graph replays, host wall clock" % T,
            "ms": loop_ms, "T": T, "ms_per_step": loop_ms / T, "ratio_to_timed_ms_per_step": loop_ms / T / ms_per_step,
            "samples_per_s_this_rank": B * L / (loop_ms * 1e-3), "finite": bool(torch.isfinite(x).all())}

    if "Tmel" in cfg:
        result["conditioner_ms_per_batch"] = cond_ms
        result["conditioner_ms_all_calls"] = cond_all
        result["end_to_end_samples_per_s_incl_conditioner"] = ddist.aggregate_throughput(
            B * L, world, T * ms_per_step * 1e-3 + cond_ms * 1e-3)

    if rank == 0 and not args.no_roofline and cfg["model"]["_name_"] == "wavenet":
        # dominant kernel: the fused residual layer.  Timed with HIP events on its own
        # launch stream inside the engine (eager launches, outside any capture).
        flops, bytes_ = layer_algorithmic_work(cfg)
        peak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x3": PEAK_BF16_MFMA_TFLOPS / 3.0, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6.0,
                "f16x3": PEAK_BF16_MFMA_TFLOPS / 3.0}[args.precision]      # fp16 and bf16 MFMA run at the same dense rate
        # every launch position of a step takes its MEDIAN over five eager steps (the first eager step after graph replays
        # runs with cold caches and lazily created events: averaged in, it put this leg 1.5 % above the launch durations
        # rocprof sees inside the timed replays)
        NLAY = cfg["model"]["num_res_layers"]
        nprof = 5
        eager = lambda k: _lib.check(lib.dws_sampler_steps(net._handle, x.data_ptr(), *ptabs, T, T - 1, k, seed, 0, stream))
        step_ms = profiled_step_ms(lib, eager, b"wn_layer", nprof, NLAY)
        n_launch = ctypes.c_int64(nprof * NLAY)
        if step_ms is None:          # launch count differs from n_layers per step: fall back to the plain average
            _lib.check(lib.dws_profile_enable(b"wn_layer"))
            eager(nprof)
            torch.cuda.synchronize()
            tot_ms = ctypes.c_double()
            _lib.check(lib.dws_profile_query(ctypes.byref(n_launch), ctypes.byref(tot_ms)))
            lib.dws_profile_disable()
            step_ms = tot_ms.value / max(n_launch.value, 1) * NLAY
        avg_ms = step_ms / NLAY
        wino = (args.precision == "f32" and os.environ.get("DWS_WN_DIRECT") is None) or args.precision in ("bf16x6", "f16x3")
        # `achieved` / `frac` are priced on the flops the kernel EXECUTES (frac <= 1 by construction): the Winograd
        # F(2,3) form does 8 C^2 instead of 12 C^2 flop per position for the convolution.  The direct-convolution
        # algorithmic flops of SURVEY.md 8(d) over the same time are reported beside it as `effective_*`.
        executed = wino_executed_work(cfg) if wino else flops
        ach = executed / (avg_ms * 1e-3) / 1e12
        eff = flops / (avg_ms * 1e-3) / 1e12
        kname = {"f32": "wn_layer_wino_kernel" if wino else "wn_layer_mfma_kernel", "bf16x3": "wn_layer_bf16x3_kernel",
                 "bf16x6": "wn_layer_bx6_kernel", "f16x3": "wn_layer_bx6_kernel"}[args.precision]
        traffic, traffic_note = None, None
        if args.config == "wnet_h256_d36_T200" and cfg["B"] == 16 and args.precision in ("f32", "bf16x6", "f16x3"):
            traffic, traffic_note = wavenet_traffic(args.precision, kname, executed)
        result["roofline"] = {
            "kernel": "%s<%s%d,%d>" % (kname, {"bf16x6": "SplitBf16x3,", "f16x3": "SplitF16x2,"}.get(args.precision, ""),
                                       cfg["model"]["res_channels"], cfg["model"]["skip_channels"]),
            "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
            "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_note,
            "executed_flops_per_launch": executed,
            "effective_TFLOPs_on_direct_conv_flops": eff, "effective_frac": eff / peak,
            "algorithm": ("Winograd F(2,3) along the dilation stride (4 K=C products per position pair instead of 6)"
                          if wino else "direct 3-tap convolution"),
            "avg_launch_ms": avg_ms, "launches_timed": n_launch.value,
            "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": bytes_,
            "hbm_achieved_GBs": bytes_ / (avg_ms * 1e-3) / 1e9,
            "hbm_frac": bytes_ / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
        }
    if rank == 0 and not args.no_roofline and cfg["model"]["_name_"] == "sashimi":
        # dominant kernel family: the fused S4 tail (three GEMMs + GLU + LN + GELU per block), all stages together
        flops, bytes_, nblocks = sashimi_tail_work(cfg)
        nprof = 5
        eager = lambda k: _lib.check(lib.dws_sampler_steps(net._handle, x.data_ptr(), *ptabs, T, T - 1, k, seed, 0, stream))
        step_ms = profiled_step_ms(lib, eager, b"s4_tail", nprof, nblocks)
        if step_ms is not None:
            ach = flops / (step_ms * 1e-3) / 1e12
            # counter-derived HBM bytes per step of the two families (tools/r05_traffic_sashimi.sh): the newest
            # profiles/r*_sashimi_traffic_<config>.json, used only if it was measured on THESE kernels -- same config, f32
            # tails (SQ_INSTS_MFMA x 4096 within 5 % of the tail flops computed above), whole steps (dispatches % blocks == 0)
            traffic, traffic_fc, traffic_note = None, None, None
            if args.precision == "f32" and not args.batch:
                import glob
                for tfile in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sashimi_traffic_%s.json" % args.config)), reverse=True):
                    tj = json.load(open(tfile))
                    ft, ff_ = tj["families"].get("s4_tail", {}), tj["families"].get("fftconv", {})
                    cnt = ft.get("sq_insts_mfma_per_launch")
                    if tj.get("config") != args.config or not ft.get("dispatches") or ft["dispatches"] % nblocks:
                        traffic_note = "%s refused: config / dispatch count" % os.path.basename(tfile)
                    elif cnt is None or abs(cnt * 4096 * nblocks / flops - 1) > 0.05:
                        traffic_note = "%s refused: SQ_INSTS_MFMA x 4096 x %d = %.4g vs tail flops %.4g" % (
                            os.path.basename(tfile), nblocks, (cnt or 0) * 4096 * nblocks, flops)
                    else:
                        traffic = ft["hbm_bytes_per_launch"] * nblocks
                        traffic_fc = ff_.get("hbm_bytes_per_launch", 0) * nblocks if ff_.get("dispatches") else None
                        traffic_note = os.path.basename(tfile)
                    break
            # the split precisions run every tail GEMM (H = 32 ... 512) on the 16-bit matrix cores: priced against that rate
            # over the products per fp32-equivalent multiply-add (six / three), like the WaveNet split legs
            tpeak = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x6": PEAK_BF16_MFMA_TFLOPS / 6.0, "f16x3": PEAK_BF16_MFMA_TFLOPS / 3.0}[args.precision]
            result["roofline"] = {
                "kernel": "s4_tail kernels (all %d block launches of a step; precision %s)" % (nblocks, args.precision), "bound": "mfma",
                "achieved": ach, "peak": tpeak, "unit": "TFLOP/s", "frac": ach / tpeak,
                "traffic": traffic, "traffic_source": traffic_note, "ms_per_step_in_kernel": step_ms, "launches_timed": nprof * nblocks,
                "algorithmic_flops_per_step": flops, "algorithmic_bytes_per_step": bytes_,
                "hbm_achieved_GBs": bytes_ / (step_ms * 1e-3) / 1e9,
                "note": "fp32 MFMA and VALU do not co-issue on gfx950 (DESIGN.md 6): the GELU/GLU/LN VALU work of the "
                        "tail adds to the MFMA time"}
        # second kernel family of the step: the fused FFT long convolution, HBM-bound by design (8 H L bytes per block:
        # the row is read once and written once), in fact limited by its LDS passes and butterflies
        fc_ms = profiled_step_ms(lib, eager, b"fftconv", nprof, nblocks)
        if fc_ms is not None and "roofline" in result:
            fc_bytes = bytes_ * 8 // 12          # 8 H L per block against the tail's 12 H L
            result["roofline"]["fftconv"] = {
                "kernel": "fftconv_kernel<log2 M, M/16> (all %d block launches of a step)" % nblocks, "bound": "hbm",
                "achieved": fc_bytes / (fc_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": fc_bytes / (fc_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "ms_per_step_in_kernel": fc_ms,
                "algorithmic_bytes_per_step": fc_bytes, "traffic": traffic_fc}
    if (rank == 0 and world == 1 and args.precision == "f32" and cfg["model"]["_name_"] == "wavenet"
            and not args.no_roofline and extras):
        # Additional, clearly separate measurements (NOT `value`): the two bf16-split arithmetics of the WaveNet layer.
        #   bf16x6: fp32-EQUIVALENT (exact 3-term split of every operand, six products, fp32 accumulate, Winograd form);
        #           its error against float64 is measured beside the f32 path's in tests/test_bf16x6_gpu.py
        #   bf16x3: 2-term split, ~1e-5 relative: narrower than fp32, reported for comparison only
        #   f16x3:  2-term fp16 split of power-of-two scaled operands, three products, the same kernel and the same float64
        #           acceptance (tests/test_f16x3_gpu.py): 22 bits per operand, so fp32-CLASS rather than fp32-faithful
        for prec in ("bf16x6", "f16x3", "bf16x3"):
            net.set_option("precision", prec)
            run(max(args.warmup, 1))
            barrier()
            t0 = time.perf_counter()
            run(args.steps)
            barrier()
            ms3 = (time.perf_counter() - t0) / args.steps * 1e3
            leg = {"ms_per_step": ms3, "value": B * L / (T * ms3 * 1e-3), "unit": "audio samples/s", "dtype": DTYPE_NAMES[prec]}
            if prec in ("bf16x6", "f16x3"):
                NLAY = cfg["model"]["num_res_layers"]
                nprod = 6 if prec == "bf16x6" else 3
                kn = "wn_layer_bx6_kernel<%s," % ("SplitBf16x3" if prec == "bf16x6" else "SplitF16x2")
                eager = lambda k: _lib.check(lib.dws_sampler_steps(net._handle, x.data_ptr(), *ptabs, T, T - 1, k, seed, 0, stream))
                st = profiled_step_ms(lib, eager, b"wn_layer", 5, NLAY)
                if st is not None:
                    flops, bytes_ = layer_algorithmic_work(cfg)
                    executed = wino_executed_work(cfg)      # fp32-equivalent GEMM flops; the MFMA pipe executes nprod x that
                    avg = st / NLAY
                    peakn = PEAK_BF16_MFMA_TFLOPS / nprod
                    leg["roofline"] = {
                        "kernel": "%s%d,%d>" % (kn, cfg["model"]["res_channels"], cfg["model"]["skip_channels"]),
                        "bound": "mfma", "achieved": executed / (avg * 1e-3) / 1e12, "peak": peakn, "unit": "TFLOP/s",
                        "frac": executed / (avg * 1e-3) / 1e12 / peakn, "avg_launch_ms": avg,
                        **dict(zip(("traffic", "traffic_source"), wavenet_traffic(prec, "wn_layer_bx6_kernel", executed)
                                   if args.config == "wnet_h256_d36_T200" and cfg["B"] == 16 else (None, None))),
                        "peak_note": "2.5 PFLOP/s dense 16-bit MFMA / %d products per fp32-equivalent multiply-add" % nprod,
                        "executed_flops_per_launch": executed, "mfma_flops_per_launch": nprod * executed,
                        "algorithmic_bytes_per_launch": bytes_,
                        "hbm_achieved_GBs": bytes_ / (avg * 1e-3) / 1e9, "hbm_frac": bytes_ / (avg * 1e-3) / 1e9 / PEAK_HBM_GBS}
                if prec == "bf16x6":
                    leg["note"] = ("opt-in precision=bf16x6: fp32-equivalent accuracy (error vs a float64 evaluation <= 2x the "
                                   "exact-f32 MFMA path's, tests/test_bf16x6_gpu.py); not the headline value")
                else:
                    leg["note"] = ("opt-in precision=f16x3: 22-bit operands, fp32 accumulate; accepted by the same float64 "
                                   "criterion as bf16x6 (tests/test_f16x3_gpu.py); not the headline value")
            else:
                leg["note"] = ("opt-in precision=bf16x3 (hi/lo bf16 MFMA inputs, fp32 accumulate); max rel err vs reference "
                               "1e-5: narrower than fp32; not the headline value")
            result["extra_" + prec] = leg
        net.set_option("precision", "f32")
    del net
    torch.cuda.empty_cache()
    return result


if __name__ == "__main__":
    main()
