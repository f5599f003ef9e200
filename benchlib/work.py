"""Work accounting of the measured kernels: algorithmic and executed flops / bytes per launch, the lookup of
counter-derived HBM traffic under a refuse-if-stale rule, and per-launch timing through the engine's HIP-event profiler."""
import json
import os

import torch

from .configs import ROOT


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
    """All S4-tail launches of one step: per block 12 H^2 flops (the three GEMMs Wo, W1, W2) and FOUR
    tensor transits = 16 H bytes per position (read the convolution output g and the block input x, write the block
    output and the next block's `LN1(out) + fc_t(e)`: `ynext`), summed over the U-Net's blocks.  The fused FFT convolution
    in front of every tail is two more transits (8 H bytes per position: `fftconv_bytes = tail_bytes / 2`)."""
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
        bytes_ += 16 * h * l * B
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


