"""The sampling measurement: T-step reverse diffusion replayed from the captured hipGraph, plus the rooflines of its
dominant kernels."""
import json
import os
import time

import torch

from .configs import (DTYPE_NAMES, PEAK_BF16_MFMA_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS, ROOT, build_model)
from .work import (layer_algorithmic_work, profiled_step_ms, sashimi_tail_work, wavenet_traffic, wino_executed_work)


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
        # the sampler state after the timed steps, every leg and every precision: a split arithmetic that overflowed shows here
        "state_finite": bool(torch.isfinite(x).all()),
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

    def eager(k):
        _lib.check(lib.dws_sampler_steps(net._handle, x.data_ptr(), *ptabs, T, T - 1, k, seed, 0, stream))

    def wavenet_roofline(prec):
        """Roofline of the dominant kernel -- the fused residual layer -- under the arithmetic the net is set to.  Timed with
        HIP events on its own launch stream inside the engine (eager launches, outside any capture): every launch position
        of a step takes its MEDIAN over five eager steps (the first eager step after graph replays runs with cold caches and
        lazily created events).  `achieved` / `frac` are priced on the flops the kernel EXECUTES (frac <= 1 by construction):
        the Winograd F(2,3) form does 8 C^2 instead of 12 C^2 flop per position for the convolution; the direct-convolution
        algorithmic flops of SURVEY.md 8(d) over the same time are reported beside it as `effective_*`.  The split
        precisions are priced against the dense 16-bit MFMA rate over the products per fp32-equivalent multiply-add."""
        flops, bytes_ = layer_algorithmic_work(cfg)
        nprod = {"f32": 1, "bf16x3": 3, "bf16x6": 6, "f16x3": 3}[prec]
        peak = PEAK_F32_MFMA_TFLOPS if prec == "f32" else PEAK_BF16_MFMA_TFLOPS / nprod
        NLAY = cfg["model"]["num_res_layers"]
        nprof = 5
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
        wino = (prec == "f32" and os.environ.get("DWS_WN_DIRECT") is None) or prec in ("bf16x6", "f16x3")
        executed = wino_executed_work(cfg) if wino else flops
        ach = executed / (avg_ms * 1e-3) / 1e12
        eff = flops / (avg_ms * 1e-3) / 1e12
        kname = {"f32": "wn_layer_wino_kernel" if wino else "wn_layer_mfma_kernel", "bf16x3": "wn_layer_bf16x3_kernel",
                 "bf16x6": "wn_layer_bx6_kernel", "f16x3": "wn_layer_bx6_kernel"}[prec]
        traffic, traffic_note = None, None
        if args.config == "wnet_h256_d36_T200" and cfg["B"] == 16 and prec in ("f32", "bf16x6", "f16x3"):
            traffic, traffic_note = wavenet_traffic(prec, kname, executed)
        roof = {
            "kernel": "%s<%s%d,%d>" % (kname, {"bf16x6": "SplitBf16x3,", "f16x3": "SplitF16x2,"}.get(prec, ""),
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
        if prec != "f32":
            roof["peak_note"] = "2.5 PFLOP/s dense 16-bit MFMA / %d products per fp32-equivalent multiply-add" % nprod
            roof["mfma_flops_per_launch"] = nprod * executed
        return roof

    if rank == 0 and not args.no_roofline and cfg["model"]["_name_"] == "wavenet":
        result["roofline"] = wavenet_roofline(args.precision)
    if rank == 0 and not args.no_roofline and cfg["model"]["_name_"] == "sashimi":
        # dominant kernel family: the fused S4 tail (three GEMMs + GLU + LN + GELU per block), all stages together
        flops, bytes_, nblocks = sashimi_tail_work(cfg)
        nprof = 5
        step_ms = profiled_step_ms(lib, eager, b"s4_tail", nprof, nblocks)
        if step_ms is not None:
            ach = flops / (step_ms * 1e-3) / 1e12
            # counter-derived HBM bytes per step of the two families (tools/r05_traffic_sashimi.sh): the newest
            # profiles/r*_sashimi_traffic_<config>.json, used only if it was measured on THESE kernels -- same config, f32
            # tails (SQ_INSTS_MFMA x 4096 within 5 % of the tail flops computed above), whole steps (dispatches % blocks == 0)
            traffic, traffic_fc, traffic_note = None, None, None
            if args.precision in ("f32", "bf16x6") and not args.batch:
                import glob
                # fp32-equivalent flops per MFMA instruction: v_mfma_f32_32x32x2_f32 = 4096; six v_mfma_f32_32x32x16_bf16 (32768
                # each) per fp32-equivalent product term under the 3-term split
                per_inst = 4096.0 if args.precision == "f32" else 32768.0 / 6.0
                pats = ["r*_sashimi_traffic_%s_%s.json" % (args.config, args.precision)]
                if args.precision == "f32":
                    pats.append("r*_sashimi_traffic_%s.json" % args.config)          # (rounds 4-5: f32 files carry no suffix)
                tfiles = sorted((f for pat in pats for f in glob.glob(os.path.join(ROOT, "profiles", pat))),
                                key=lambda f: os.path.basename(f)[:3], reverse=True)
                for tfile in tfiles:
                    tj = json.load(open(tfile))
                    ft, ff_ = tj["families"].get("s4_tail", {}), tj["families"].get("fftconv", {})
                    cnt = ft.get("sq_insts_mfma_per_launch")
                    if tj.get("config") != args.config or not ft.get("dispatches") or ft["dispatches"] % nblocks:
                        traffic_note = "%s refused: config / dispatch count" % os.path.basename(tfile)
                    elif tj.get("precision", "f32") != args.precision:
                        traffic_note = "%s refused: measured under precision=%s" % (os.path.basename(tfile), tj.get("precision"))
                    elif cnt is None or abs(cnt * per_inst * nblocks / flops - 1) > 0.08:
                        # (the split kernels' bias / correction k-blocks add a few per cent of MFMA instructions to the count)
                        traffic_note = "%s refused: SQ_INSTS_MFMA x %.0f x %d = %.4g vs tail flops %.4g" % (
                            os.path.basename(tfile), per_inst, nblocks, (cnt or 0) * per_inst * nblocks, flops)
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
            fc_bytes = bytes_ // 2               # 8 H L per block (row read once, written once) against the tail's 16 H L
            result["roofline"]["fftconv"] = {
                "kernel": "fftconv_kernel<log2 M, M/16> (all %d block launches of a step)" % nblocks, "bound": "hbm",
                "achieved": fc_bytes / (fc_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": fc_bytes / (fc_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "ms_per_step_in_kernel": fc_ms,
                "algorithmic_bytes_per_step": fc_bytes, "traffic": traffic_fc}
    if rank == 0 and world == 1 and cfg["model"]["_name_"] == "wavenet" and not args.no_roofline and extras:
        # Additional, clearly separate measurements (NOT `value`) of the same network under the other arithmetic:
        #   headline bf16x6 (fp32-equivalent: exact 3-term split of every operand, six products, fp32 accumulate; its error
        #   against float64 is measured beside the f32 path's in tests/test_bf16x6_gpu.py, at this very size in
        #   tests/test_full_size_gpu.py, over T = 200 steps in tests/test_split_trajectory_gpu.py) -> `extra_f32_exact`;
        #   headline f32 -> `extra_bf16x6`.  Each with its own roofline.
        for prec in [p for p in ("f32", "bf16x6") if p != args.precision]:
            net.set_option("precision", prec)
            run(max(args.warmup, 1))
            barrier()
            t0 = time.perf_counter()
            run(args.steps)
            barrier()
            ms3 = (time.perf_counter() - t0) / args.steps * 1e3
            leg = {"ms_per_step": ms3, "value": B * L / (T * ms3 * 1e-3), "unit": "audio samples/s", "dtype": DTYPE_NAMES[prec],
                   "state_finite": bool(torch.isfinite(x).all()), "roofline": wavenet_roofline(prec),
                   "note": "the same steps under precision=%s; not the headline value" % prec}
            result["extra_f32_exact" if prec == "f32" else "extra_" + prec] = leg
        net.set_option("precision", args.precision)
    del net
    torch.cuda.empty_cache()
    return result


