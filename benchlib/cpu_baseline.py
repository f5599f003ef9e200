"""`cpu_baseline`: the oracle (reference-equivalent PyTorch-CPU graph) timed on this box's host cores.  This module and
tests/ + smoke() are the only importers of oracle/: it is the checker and the reported baseline, never the product path."""
import json
import os
import sys
import time

import torch

from .configs import CONFIGS, ROOT


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
    q = cpu_quota()
    out.update(cpu_quota_cpus=q["cpu_quota_cpus"], cpuset_cpus=q["cpuset_cpus"], cpu_quota_source=q["source"])
    if config_name is not None and per_step < 8.0:
        wh = cpu_whole_host(config_name, best)
        out["whole_host"] = wh
        if wh.get("workers", 0) <= 1 and "value" in wh:
            wh["note"] = "the container's CPU allowance covers one %d-thread worker: this IS the single-process figure re-measured" % best
    # the figure GPU/CPU ratios are computed from: the better of the single process and the whole-host (quota-sized) leg
    single = out["value"]
    whv = (out.get("whole_host") or {}).get("value")
    out["single_process_value"] = single
    out["best"] = {"value": max(single, whv or 0.0), "which": "whole_host" if (whv or 0.0) > single else "single_process"}
    return out


def cpu_quota():
    """What this container may use of the host: the cgroup CPU bandwidth quota (`cpu.max`: quota / period, v2; the v1 files as
    a fallback) and the effective cpuset.  Returns {"cpu_quota_cpus": float or None (unlimited), "cpuset_cpus": int, "source": ...}.
    A whole-host figure measured with more runnable threads than the quota allows is throttled, not parallel: the whole-host leg
    is sized to min(physical cores, quota)."""
    out = {"cpu_quota_cpus": None, "cpuset_cpus": len(os.sched_getaffinity(0)), "source": None}
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        out["source"] = "/sys/fs/cgroup/cpu.max = %s %s" % (q, per)
        if q != "max":
            out["cpu_quota_cpus"] = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            out["source"] = "cpu.cfs_quota_us / cpu.cfs_period_us = %d / %d" % (q, per)
            if q > 0:
                out["cpu_quota_cpus"] = q / per
        except (OSError, ValueError):
            out["source"] = "no cgroup cpu controller files readable"
    try:
        out["cpuset_effective"] = open("/sys/fs/cgroup/cpuset.cpus.effective").read().strip()
    except OSError:
        pass
    return out


_WHOLE_HOST_WORKER = r'''
import json, os, sys, time
cpus = [int(c) for c in os.environ["DWS_CPUSET"].split(",")]
os.sched_setaffinity(0, cpus)
os.environ["OMP_NUM_THREADS"] = str(len(cpus))
sys.path.insert(0, os.environ["DWS_ROOT"])
import torch
torch.set_num_threads(len(cpus))
from benchlib.configs import CONFIGS
from oracle import sashimi as osa, wavenet as own
from diffwave_sashimi_amd.models import construct_model
cfg = CONFIGS[os.environ["DWS_CONFIG"]]
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
    quota = cpu_quota()["cpu_quota_cpus"]
    usable = len(cores) if quota is None else max(threads, min(len(cores), int(quota)))
    n = max(1, usable // threads)
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
            "cores": n * threads, "physical_cores": len(cores), "cpu_quota_cpus": quota, "steps_per_worker": nsteps,
            "ms_per_step_b1_slowest_worker": slow * 1e3, "ms_per_step_b1_fastest_worker": min(means) * 1e3,
            "sample": "%d concurrent B=1 workers x %d threads, each pinned to its own physical cores; %d forward steps each "
                      "after a warm-up, common start; rate = workers x L / (T x slowest worker's mean step)" % (n, threads, nsteps)}


def cpu_train_baseline(cfg, seconds_budget=30.0, sample_L=None):
    """One `train.py:118-143`-style step of the oracle on the host cores at B=1: q-sample, forward, MSE against the noise,
    backward through torch autograd of the reference-equivalent CPU graph (no optimizer: its cost is negligible beside
    the backward).  Bounded sample: a first step that already takes > 8 s IS the sample (it includes the one-time
    allocator warm-up), otherwise a second step is timed.
    sample_L: the bounded form for the default run -- ONE step on a clip of sample_L samples through the same network built
    for that length (the oracle regenerates every S4 kernel in every call, 88 % of its step and proportional to the clip
    length: a full 16000-sample step is ~100 s of host time, a 4000-sample one ~25 s); the rate is samples / second of that
    step, so it scales to the metric's unit directly."""
    from oracle import sashimi as osa
    from oracle import wavenet as own
    from diffwave_sashimi_amd.models import construct_model
    from diffwave_sashimi_amd.sampling import calc_diffusion_hyperparams
    ncpu = os.cpu_count() or 1
    quota = cpu_quota()["cpu_quota_cpus"]
    th = min(32, ncpu) if quota is None else max(1, min(32, ncpu, int(quota)))     # more runnable threads than the quota = throttled
    torch.set_num_threads(th)
    torch.manual_seed(0)
    mcfg = dict(cfg["model"])
    L, T = cfg["L"], cfg["diffusion"]["T"]
    if sample_L and mcfg["_name_"] == "sashimi":
        mcfg["L"] = L = int(sample_L)
    net = construct_model(mcfg)
    leaf = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() else v.clone())
            for k, v in net.state_dict().items()}
    fwd = own.wavenet_forward if cfg["model"]["_name_"] == "wavenet" else osa.sashimi_forward
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
        eps = fwd(leaf, mcfg, xt, ts.view(1, 1))
        loss = torch.nn.functional.mse_loss(eps, z)
        loss.backward()
        return time.perf_counter() - t0

    times = [one()]
    if not sample_L and (times[0] < 8.0 or times[0] * 2 < seconds_budget):
        times.append(one())
    t = times[-1]
    return {"value": L / t, "unit": "training audio samples/s", "cores": th, "host_cpus": ncpu, "kind": "port",
            "sample": "%d training step(s) (q-sample + forward + MSE + autograd backward of the oracle) at B=1, L=%d with %d "
                      "threads; the last one is reported%s" % (len(times), L, th, (
                          " -- bounded sample: a %d-sample clip through the same network built for that length (config: L=%d; "
                          "full length measured once per round, profiles/r04_bench_c5train_cpu.json)" % (L, cfg["L"])) if sample_L else ""),
            "ms_per_step_b1": t * 1e3, "steps_ms": [x * 1e3 for x in times], "cpu_model": _cpu_model(), "cpu_quota_cpus": quota}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


