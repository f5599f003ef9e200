"""GPU: bench.py contract -- one JSON line with the required keys at N = 1, and the N > 1 control flow
(torch.distributed.run, barrier + max over ranks, rank-0-only output, whole-job aggregate) with two ranks
sharing the box's single GPU over gloo (DWS_BENCH_SHARE_GPU=1; real runs use one GPU per rank over RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_and_cpu_baseline(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "wnet_h128_d30_T200", "--steps", "4",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1
    assert d["config"]["workload"] == "wnet_h128_d30_T200" and "model" not in d["config"]
    rf, cb = d["roofline"], d["cpu_baseline"]
    assert rf["bound"] == "mfma" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    assert abs(d["value"] - 16 * 16000 / (200 * d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # the metric as generate.py defines it: one complete T-step loop (x_T draw + T replays), wall-clocked
    fl = d["full_loop"]
    assert fl["T"] == 200 and fl["finite"] and abs(fl["ms_per_step"] - fl["ms"] / 200) < 1e-9
    assert 0.9 < fl["ratio_to_timed_ms_per_step"] < 1.15, fl
    # the headline arithmetic is the fp32-equivalent split (precision=bf16x6): its own dtype string, roofline against
    # 2.5 PFLOP/s / 6; the exact-f32 leg rides beside it with its own roofline against the fp32 MFMA rate, never as `value`
    assert d["dtype"].startswith("f32-equivalent") and d["state_finite"]
    assert abs(rf["peak"] - 2500.0 / 6) < 1e-6 and rf["kernel"].startswith("wn_layer_bx6_kernel<SplitBf16x3")
    x32 = d["extra_f32_exact"]
    assert x32["dtype"] == "f32" and x32["state_finite"] and abs(x32["roofline"]["peak"] - 157.3) < 1e-6
    assert 0 < x32["roofline"]["frac"] < 1 and x32["roofline"]["kernel"].startswith("wn_layer_wino_kernel")
    assert d["ms_per_step"] < x32["ms_per_step"]
    assert "extra_f16x3" not in d and "extra_bf16x3" not in d          # the narrower splits are not advertised in the line
    # the whole host beside the best single process: N pinned B = 1 workers, sized to the container's
    # CPU quota; GPU/CPU is computed from the better of the two
    wh = cb["whole_host"]
    assert "error" not in wh and wh["workers"] >= 1 and wh["cores"] == wh["workers"] * wh["threads_per_worker"] and wh["value"] > 0
    assert "cpu_quota_cpus" in cb and cb["cpu_quota_source"]
    if cb["cpu_quota_cpus"] is not None:
        assert wh["cores"] <= max(cb["cpu_quota_cpus"], wh["threads_per_worker"])
    assert cb["best"]["value"] == max(cb["single_process_value"], wh["value"])
    assert abs(d["gpu_over_cpu"] - d["value"] / cb["best"]["value"]) < 1e-6 * d["gpu_over_cpu"]
    assert "gpu_over_cpu_whole_host" not in d


def test_two_ranks_aggregate(gpu):
    env = dict(os.environ, DWS_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "wnet_h128_d30_T200",
           "--batch", "4", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert "cpu_baseline" not in d                       # rank 0 at N = 1 only
    # whole-job aggregate: 2 ranks x (4 clips x 16000 samples / 200 steps) per step time
    assert abs(d["value"] - 2 * 4 * 16000 / (200 * d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


def test_default_headline_line_carries_the_other_baseline_configs(gpu):
    """The default command (what the driver runs) evidences BASELINE configs 3, 4 and 5 too: `extra_configs` holds a
    short sampling leg for unet_d64 and the conditional unet_d32, and one GPU's training step of unet_d128, each with
    its own ms/step and roofline; the WaveNet roofline prices the Winograd kernel on the flops it executes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d) and d["config"]["workload"] == "wnet_h256_d36_T200" and d["dtype"].startswith("f32-equivalent")
    rf = d["roofline"]
    assert 0 < rf["frac"] < 1 and rf["executed_flops_per_launch"] < rf["algorithmic_flops_per_launch"]
    assert abs(rf["achieved"] - rf["executed_flops_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * rf["achieved"]
    assert rf["effective_TFLOPs_on_direct_conv_flops"] > rf["achieved"]
    ex = d["extra_configs"]
    ops = {"unet_d128_n6_T200 B=128 (README operating point)", "unet_d64_n6_T200 B=256 (README operating point)"}
    assert set(ex) == {"unet_d64_n6_T200", "unet_d32_n6_T50_cond", "unet_d128_n6_T200 --mode train"} | ops
    for name, leg in ex.items():
        want = "f32" if name.endswith("--mode train") else "f32-equivalent"      # (the training leg's value is the exact-f32 step)
        assert leg["ms_per_step"] > 0 and leg["value"] > 0 and leg["dtype"].startswith(want), (name, leg)
        if name not in ops:
            assert leg["roofline"]["bound"] == "mfma" and 0 < leg["roofline"]["frac"] < 1, name
    for name in ops:     # the reference's documented batch sizes: 8x / 16x the tested batch through the same 32-bit offsets
        assert ex[name]["hbm_bytes_in_use"] > 2 ** 30 and ex[name]["config"]["batch_per_gpu"] in (128, 256)
    assert d["full_loop"]["T"] == 200 and d["full_loop"]["finite"]
    for name in ("unet_d64_n6_T200", "unet_d32_n6_T50_cond"):
        fc = ex[name]["roofline"]["fftconv"]
        assert fc["bound"] == "hbm" and 0 < fc["frac"] < 1
        fl = ex[name]["full_loop"]
        assert fl["finite"] and 0.9 < fl["ratio_to_timed_ms_per_step"] < 1.2, (name, fl)
    c3 = ex["unet_d64_n6_T200"]
    assert abs(c3["value"] - 16 * 16000 / (200 * c3["ms_per_step"] * 1e-3)) < 1e-6 * c3["value"]
    tr = ex["unet_d128_n6_T200 --mode train"]
    assert tr["config"]["batch_per_gpu"] == 32 and "whole_step_frac" in tr["roofline"]
    # the exact-f32 legs ride beside the split ones, each with its own roofline
    for name in ("unet_d64_n6_T200", "unet_d32_n6_T50_cond"):
        x32 = ex[name]["extra_f32_exact"]
        assert x32["ms_per_step"] > 0 and x32["dtype"] == "f32" and x32["state_finite"] and ex[name]["state_finite"]
        assert abs(x32["roofline"]["peak"] - 157.3) < 1e-6 and abs(ex[name]["roofline"]["peak"] - 2500.0 / 6) < 1e-6
        # tail bytes: four tensor transits per block (g, x, out, ynext), the convolution in front of it two
        rf3 = ex[name]["roofline"]
        assert rf3["fftconv"]["algorithmic_bytes_per_step"] * 2 == rf3["algorithmic_bytes_per_step"]
    t6 = tr["extra_bf16x6"]
    assert t6["dtype"].startswith("f32-equivalent") and t6["ms_per_step"] > 0 and abs(t6["final_loss"] - tr["final_loss"]) < 1e-3
    assert d["extra_f32_exact"]["ms_per_step"] > d["ms_per_step"] and d["extra_f32_exact"]["state_finite"]
    # the LAST key of the line is a compact digest of every config's number (the driver keeps only a tail of the line)
    assert list(d)[-1] == "summary" and len(json.dumps(d["summary"])) <= 1500, len(json.dumps(d["summary"]))
    sm = d["summary"]
    assert abs(sm["C2 wnet_h256_d36 B16"]["ms"] - d["ms_per_step"]) < 1e-3 * d["ms_per_step"]
    assert sm["C3 unet_d64 B16"]["f32_ms"] > 0 and sm["C4 unet_d32 cond B32"]["ms"] > 0 and sm["C5 unet_d128 train B32/GPU"]["f32_ms"] > 0


@pytest.mark.parametrize("mode", ["sample", "train"])
def test_eight_ranks_from_the_plain_command(gpu, mode):
    """`python bench.py --gpus 8` (no launcher: bench.py spawns its ranks) with all eight ranks on this box's one GPU over
    gloo (DWS_BENCH_SHARE_GPU=1; the driver's node has one GPU per rank over RCCL, same code path otherwise): eight
    per-rank timings, eight DISTINCT Philox seeds / states (sampling) or shard losses (training), identical weights on
    every rank after the averaged steps, whole-job aggregate = 8 x per-rank units / slowest rank's time."""
    env = dict(os.environ, DWS_BENCH_SHARE_GPU="1", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "wnet_h128_d30_T200", "--batch", "1",
           "--steps", "2", "--warmup", "1", "--mode", mode]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 8 and d["scaling"] == "weak" and d["process_group"]["world_size"] == 8
    assert len(d["per_rank_ms_per_step"]) == 8 and abs(max(d["per_rank_ms_per_step"]) - d["ms_per_step"]) < 1e-6 * d["ms_per_step"] + 1e-9
    assert "cpu_baseline" not in d and "extra_configs" not in d
    if mode == "sample":
        assert d["per_rank_seed"] == [1234 + r_ for r_ in range(8)]
        assert len(set(d["per_rank_state_digest"])) == 8            # eight different sets of clips
        assert abs(d["value"] - 8 * 1 * 16000 / (200 * d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    else:
        assert len(set(d["per_rank_final_loss"])) == 8              # eight different shards
        assert len(set(d["per_rank_param_digest"])) == 1            # one model: the gradients were averaged
        assert abs(d["value"] - 8 * 1 * 16000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
