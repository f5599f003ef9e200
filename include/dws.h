/*
 * dws.h -- flat C ABI of libdws.so, the MI355X (gfx950) DiffWave denoising-loop engine.
 *
 * This is the drop-in boundary.  Everything below is
 * `extern "C"`, plain pointers + explicit sizes + a stream handle; there are no
 * torch types in any signature.  All data pointers are DEVICE pointers unless a
 * parameter is documented as host memory.  Outputs are allocated by the caller
 * (the reference allocates them through torch, `cauchy_cuda.cu:355,462-463`);
 * inputs are borrowed for the duration of the call and never written.
 *
 * Threading / streams: every entry point enqueues on the `stream` it is given
 * (a `hipStream_t` passed as `void*`; NULL = the default stream) and returns
 * without synchronising, exactly like the reference kernels which launch on
 * `at::cuda::getCurrentCUDAStream()` (`cauchy_cuda.cu:124,222,356,464`).
 * One process <-> one device (`generate.py:86`, `distributed_util.py:55`).
 *
 * Errors: every function returns an `int` status: 0 = ok, negative = error
 * class (below).  `dws_last_error()` returns a thread-local message.  The
 * Python host side turns DWS_ERR_UNSUPPORTED into `NotImplementedError`
 * (`extensions/cauchy/cauchy.py:72-77,95-101`) and everything else into
 * `RuntimeError` (the reference's `TORCH_CHECK`s, `cauchy.cpp:6-7,58-64`).
 * Nothing fails silently: an unsupported N raises instead of falling through
 * the `switch` as the reference does (`cauchy_cuda.cu:366-372`).
 *
 * Reference citations are `path:line` relative to albertfgu/diffwave-sashimi.
 */
#ifndef DWS_H_
#define DWS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DWS_OK               0
#define DWS_ERR_INVALID     -1   /* bad argument / shape mismatch (reference: TORCH_CHECK -> RuntimeError) */
#define DWS_ERR_UNSUPPORTED -2   /* size the kernels do not cover (reference: NotImplementedError)         */
#define DWS_ERR_HIP         -3   /* a HIP runtime / hipFFT call failed                                      */
#define DWS_ERR_STATE       -4   /* call order violated (e.g. forward before prepare)                       */

/* Thread-local description of the last non-zero status returned on this thread. */
const char* dws_last_error(void);
/* ABI version of the library (bumped when a signature changes). */
int dws_abi_version(void);
/* Name of the GPU architecture the kernels were compiled for ("gfx950"). */
const char* dws_arch(void);

/* ------------------------------------------------------------------------
 * Cauchy multiply -- replaces the pybind module `cauchy_mult`
 * (`extensions/cauchy/cauchy.cpp:86-95`).  Complex64 tensors are passed as
 * interleaved (re,im) float pairs, i.e. the memory of a contiguous
 * torch.cfloat tensor.
 *
 *   v, w : [B, N]   z : [L]   out, dout : [B, L]   dv, dw : [B, N]
 * ------------------------------------------------------------------------ */

/* cauchy_mult_sym_fwd (`cauchy.cpp:55-66`, kernel `cauchy_cuda.cu:242-375`):
 *   out[b,l] = sum_{n<N} v[b,n]/(z[l]-w[b,n]) + conj(v[b,n])/(z[l]-conj(w[b,n]))
 * N is the HALF state size.  Any 1 <= N <= 1024 is accepted (the reference
 * accepts powers of two 2..1024 only, `cauchy.py:95-98`). */
int dws_cauchy_sym_fwd(const float* v, const float* z, const float* w, float* out,
                       int64_t B, int64_t N, int64_t L, void* stream);

/* cauchy_mult_sym_bwd (`cauchy.cpp:68-82`, kernel `cauchy_cuda.cu:377-487`):
 *   dv[b,n] = sum_l dout/(conj z - conj w) + conj(dout)/(z - conj w)
 *   dw[b,n] = conj(v) * sum_l dout/(conj z - conj w)^2 + conj(dout)/(z - conj w)^2 */
int dws_cauchy_sym_bwd(const float* v, const float* z, const float* w, const float* dout,
                       float* dv, float* dw, int64_t B, int64_t N, int64_t L, void* stream);

/* cauchy_mult_fwd (`cauchy.cpp:25-36`, kernel `cauchy_cuda.cu:44-139`), non-symmetric:
 *   out[b,l] = sum_{n<N} v[b,n]/(z[l]-w[b,n]) */
int dws_cauchy_fwd(const float* v, const float* z, const float* w, float* out,
                   int64_t B, int64_t N, int64_t L, void* stream);

/* cauchy_mult_bwd (`cauchy.cpp:38-53`, kernel `cauchy_cuda.cu:141-240`):
 *   dv[b,n] = sum_l dout/conj(z-w),  dw[b,n] = conj(v) * sum_l dout/conj(z-w)^2 */
int dws_cauchy_bwd(const float* v, const float* z, const float* w, const float* dout,
                   float* dv, float* dw, int64_t B, int64_t N, int64_t L, void* stream);

/* ------------------------------------------------------------------------
 * Model -- replaces `models.construct_model(cfg)` + `net((audio, t), mel)`
 * (`models/__init__.py:4-12`, `models/wavenet.py:202-210`,
 * `models/sashimi.py:277-313`).
 * ------------------------------------------------------------------------ */

#define DWS_KIND_WAVENET 1   /* model._name_ == "wavenet" */
#define DWS_KIND_SASHIMI 2   /* model._name_ == "sashimi" */
#define DWS_MAX_POOL 8

/* Field names follow the YAML keys of configs/model/{wavenet,sashimi}.yaml. */
typedef struct dws_model_desc {
    int32_t kind;
    int32_t in_channels, out_channels;
    int32_t diffusion_step_embed_dim_in, diffusion_step_embed_dim_mid, diffusion_step_embed_dim_out;
    int32_t unconditional;            /* 1: no mel path */
    int32_t mel_upsample[2];          /* conditional only; default {16,16} (`wavenet.py:50`) */
    int32_t mel_bands;                /* 80 (`wavenet.py:70`) */
    /* wavenet */
    int32_t res_channels, skip_channels, num_res_layers, dilation_cycle;
    /* sashimi */
    int32_t d_model, n_layers, n_pool, pool[DWS_MAX_POOL], expand, ff, unet, L;
} dws_model_desc;

typedef struct dws_model dws_model;

int dws_model_create(const dws_model_desc* desc, dws_model** out);
int dws_model_destroy(dws_model* m);

/* Number of state-dict entries the model expects and their names/shapes, in
 * the reference's state_dict key layout.  `shape` must
 * hold 8 entries; returns ndim through *ndim.  dtype: 0 = float32, 1 = int64. */
int dws_model_num_params(const dws_model* m);
int dws_model_param_info(const dws_model* m, int index, const char** name,
                         int64_t* shape, int* ndim, int* dtype);

/* Hand one RAW state-dict tensor (weight_g / weight_v / bias / S4 parameters as
 * real (...,2) views, `s4.py:631-638`) to the model.  `data` may be a device or
 * host pointer; it is copied, not retained.  Weight-norm folding, MFMA operand
 * packing and S4 kernel generation happen inside dws_model_commit(). */
int dws_model_set_param(dws_model* m, const char* name, const void* data,
                        const int64_t* shape, int ndim, int dtype, void* stream);

/* Refresh `count` float32 parameters that were set before from device tensors of the same shape, in one launch
 * (a training loop re-sends every parameter after each optimizer step).  Marks the model dirty like set_param. */
int dws_model_update_params(dws_model* m, int32_t count, const char* const* names, const float* const* srcs, void* stream);

/* String options (unknown key/value -> DWS_ERR_INVALID):
 *   "precision" = "f32"    (default) exact-f32 MFMA (v_mfma_f32_32x32x2_f32): bitwise an fmaf chain
 *               = "bf16x3" WaveNet residual layers on the bf16 matrix cores with a 3-term hi/lo split
 *                          (W_hi x_hi + W_hi x_lo + W_lo x_hi, fp32 accumulate): ~1e-5 relative, 5.3x the
 *                          matrix rate.  Not part of the reference surface.
 *               = "bf16x6" WaveNet residual layers on the bf16 matrix cores at fp32-EQUIVALENT accuracy: every GEMM
 *                          operand as an exact 3-term bf16 split (24 significand bits), the six partial products above
 *                          2^-26 accumulated in fp32, Winograd F(2,3) form (the f32 path's algorithm and roundings).
 *                          WaveNet: inference only.  SaShiMi: the S4 block tails
 *                          (all H) in sampling; in training the pointwise GEMMs and weight gradients of the step.
 *                          What "fp32-equivalent" was MEASURED to mean (tests/test_bf16x6_gpu.py, test_full_size_gpu.py,
 *                          test_split_trajectory_gpu.py): operands are carried exactly, products are exact, the fp32
 *                          accumulation is the matrix core's own adder, which is NOT an fmaf chain -- fp32-CLASS, not
 *                          bit-compatible.  GEMM level, error relative to sum |a||b|: 2^-24.1 (K = 16) .. 2^-22.4 (K = 256);
 *                          on operands spread over 2^(+-12) up to 7.5e-7, i.e. 2 - 3 x a sequential fp32 sum in the same
 *                          k-block order (torch's own fp32 matmul: 8.7e-7 on the same operands); bound asserted:
 *                          min(2^-20, 4 x sequential fp32).  Network level against float64: 0.73 .. 1.07 x the exact-f32
 *                          path's error (WaveNet and SaShiMi, B = 1 .. 32, L = 16000), T = 200 trajectories within 3e-7
 *                          of the f32 path's.  Results do not depend on the batch position of a clip (bitwise).
 *                          Where no split instance exists (SaShiMi stages whose length is not a multiple of 4, channel
 *                          counts the MFMA tiling does not cover, the pooling GEMMs, 3-tap training GEMMs) the f32
 *                          kernels run: the tap "split_launches" reports how many GEMM launches of the last forward ran
 *                          split and how many fell back.
 *               = "f16x3"  the same kernels with a 2-term fp16 split of power-of-two scaled operands (22 significand
 *                          bits per operand, three products: half the matrix work of bf16x6).  Inference only.  Accepted
 *                          by the same float64 criterion; activations beyond 2^11 overflow fp16 and yield NaN, and so
 *                          does a WaveNet step-embedding row fc_t(e) beyond ~40 at C = 256 (it rides in an fp16 k-block
 *                          times the weight scale).  An experiment: narrower than the reference's arithmetic, never a default.
 *   "conv_algo" = "winograd" (default) WaveNet residual layers (precision f32) with the dilated 3-tap convolution in
 *                          Winograd F(2,3) form along the dilation stride: 8 C^2 instead of 12 C^2 flop per position,
 *                          one extra fp32 rounding in the weights and in the inputs (same 1e-6 class error); the
 *                          training step's data and weight gradients of that convolution run the same pairing
 *               = "direct" the direct three-tap form, forward and both adjoints (A/B runs). */
int dws_model_set_option(dws_model* m, const char* key, const char* value);

/* Fold / pack everything that depends only on the weights.  Called implicitly
 * by forward when parameters changed since the last commit. */
int dws_model_commit(dws_model* m, void* stream);

/* Size the workspace for inputs of shape audio[B, in_channels, L]. */
int dws_model_prepare(dws_model* m, int64_t B, int64_t L);

/* Install (or with mel == NULL remove) the mel-spectrogram condition
 * mel[Bm, mel_bands, Tmel], Bm in {1, B} (`generate.py:140,155`).  The
 * upsample + 1x1 terms of every block (`wavenet.py:98-111`,
 * `sashimi.py:160-175`) are evaluated once here, not per step. */
int dws_model_set_condition(dws_model* m, const float* mel, int64_t Bm, int64_t Tmel, void* stream);

/* eps[B, out_channels, L] = net((audio[B, in_channels, L], steps[B]))  (fp32).
 * `steps` holds the diffusion step of every batch element as float32
 * (`generate.py:50`; the int64 steps of `train.py:218` are converted by the host side). */
int dws_model_forward(dws_model* m, const float* audio, const float* steps, float* out, void* stream);

/* Training path (`train.py:198-222`): both backbones, unconditional and mel-conditional, fp32 precision
 * (precision=bf16x3 returns DWS_ERR_UNSUPPORTED; SaShiMi channel counts that are not multiples of 32 train on a
 * plain-FMA GEMM instead of the MFMA adjoints).
 * forward_train == forward but keeps the activations backward needs inside the model -- of ONE forward:
 * every forward_train must be followed by its backward before the next forward_train.  backward takes dLoss/d(eps)[B, out_channels, L] and produces the gradient of every RAW
 * state-dict tensor (weight_g / weight_v / bias ...), fetched with get_grad (device copy).  The
 * data-parallel exchange of those gradients is the host side's job (RCCL all-reduce,
 * `distributed_util.py:97-149`). */
int dws_model_forward_train(dws_model* m, const float* audio, const float* steps, float* out, void* stream);
int dws_model_backward(dws_model* m, const float* dout, void* stream);
int dws_model_get_grad(dws_model* m, const char* name, float* dst, int64_t numel, void* stream);
/* The same for `count` parameters in one launch (the autograd wrapper fetches every gradient after backward). */
int dws_model_get_grads(dws_model* m, int32_t count, const char* const* names, float* const* dsts, const int64_t* numels,
                        void* stream);

/* Staged hand-over of the gradients for the data-parallel exchange (`distributed_util.py:112-142` flattens and all-reduces
 * AFTER backward; here the exchange of a bucket starts while backward still runs).  set_grad_sinks names, for `count`
 * parameters, a device destination and a GROUP (the host's all-reduce bucket) each; it stays in force until it is called
 * again (count == 0 removes it).  With sinks installed, dws_model_backward itself delivers the gradients: as soon as the last
 * gradient of a group has been produced it copies the group to its destinations (one launch) and records the group's
 * event on `stream`; groups left over are delivered at the end, so after backward every destination is written in stream
 * order (no get_grads call is needed for them).  grad_group_wait makes `waiting_stream` wait for a group's event (call it
 * after backward has returned: a collective launched on that stream then depends on the group's gradients only, not on
 * the rest of backward).  When a gradient is final is learnt from the first backward after set_grad_sinks (which delivers
 * everything at its end); grad_ready_seq reports, per parameter, the flush point (block / layer number in backward order,
 * -1 = never written) after which its gradient was final in the last backward -- the order a host should bucket in. */
int dws_model_set_grad_sinks(dws_model* m, int32_t count, const char* const* names, float* const* dsts, const int64_t* numels,
                             const int32_t* groups, int32_t ngroups);
int dws_model_grad_group_wait(dws_model* m, int32_t group, void* waiting_stream);
int dws_model_grad_ready_seq(dws_model* m, int32_t count, const char* const* names, int32_t* seq_out);

/* Debug/parity tap: copy an internal activation into `dst` (device pointer,
 * `capacity` floats).  WaveNet: "pre_final" = ReLU(final_conv[0](skip)) [B,S,L],
 * "skip" [B,S,L], "x" (last residual output) [B,C,L]; the step-only terms: "part_t"
 * [B, n_layers*C] / "abt" (per-clip rows of the last forward) and "tab_part_t"
 * [T, n_layers*C] / "tab_abt" (the sampler's step table).  Function-level taps of the last per-clip forward, both
 * models: "emb" [B, embed_dim_in] (`models/utils.py:20-27`), "emb_mlp" [B, embed_dim_out] (the two swish layers), "part_t"
 * (every block's fc_t row); SaShiMi also "nfin" [B, d_model, L] = the final TransposedLayerNorm, "out:<layer prefix>" the
 * output of a layer and "k:<block prefix>" its S4 kernel.  Both models:
 * "sampler_eps" = the network output of the sampler's last reverse step [B,Cout,L]. */
int dws_model_read_tap(dws_model* m, const char* tap, float* dst, int64_t capacity, void* stream);

/* ------------------------------------------------------------------------
 * Reverse-diffusion sampler -- replaces `generate.sampling`
 * (`generate.py:23-55`).  One reverse step is captured as a hipGraph on first
 * use and replayed T times; step index, schedule coefficients and the RNG
 * counter live in device memory.  Everything of the network that depends on the
 * diffusion step only (embedding, its MLP, every layer's fc_t projection) is
 * evaluated once per (weights, T) for t = 0..T-1 and indexed by the device step
 * counter inside the replays; the update x <- (x - c1 eps) / c2 (+ sigma z) rounds
 * every product, difference, quotient and sum once, as the reference's op-by-op
 * float32 evaluation does.
 *
 *   x          [B, C, L]  in: x_T (or anything when seed-driven, see below); out: x_0
 *   alpha, alpha_bar, sigma  HOST float[T] tables from calc_diffusion_hyperparams
 *                            (`utils.py:121-151`)
 *   noise      optional DEVICE [T, B, C, L]: noise[t] is added after step t
 *              (t > 0).  NULL -> on-device Philox4x32-10 + Box-Muller keyed by
 *              (seed, t, element).
 *   init_from_seed  non-zero: also draw x_T ~ N(0, I) from the Philox stream.
 * ------------------------------------------------------------------------ */
int dws_sampler_run(dws_model* m, float* x, const float* alpha, const float* alpha_bar,
                    const float* sigma, int32_t T, const float* noise, uint64_t seed,
                    int32_t init_from_seed, int32_t use_graph, void* stream);

/* Run `n_steps` reverse steps starting at step index t_start (for benchmarking
 * a bounded number of steps of the T-step loop with the same graph). */
int dws_sampler_steps(dws_model* m, float* x, const float* alpha, const float* alpha_bar,
                      const float* sigma, int32_t T, int32_t t_start, int32_t n_steps,
                      uint64_t seed, int32_t use_graph, void* stream);

/* Mel-spectrogram front-end of the vocoding path: TacotronSTFT.mel_spectrogram
 * (`dataloaders/stft.py:196-244`) as called by Mel2Samp.get_mel (`dataloaders/mel2samp.py:76-82`) and
 * generate.py:147-153.  audio [B][T] in [-1, 1]; window [n_fft] (the Hann window, centre-padded to
 * filter_length, `stft.py:122-129`); mel_basis [n_mels][n_fft/2+1] (`stft.py:203-210`); out
 * [B][n_mels][T/hop + 1] = log(max(mel_basis . |STFT|, clip)).  All pointers are device pointers. */
int dws_mel_spectrogram(const float* audio, int64_t B, int64_t T, const float* window, const float* mel_basis,
                        int32_t n_fft, int32_t hop, int32_t n_mels, float clip, float* out, void* stream);

/* The arithmetic of precision="bf16x6" alone (accuracy tests; not a tuned GEMM): C[M][N] = A[M][K] . B[K][N], row-major
 * fp32 device tensors, every operand split into three bf16 terms in registers, six bf16 MFMA products per term pair
 * accumulated in fp32 -- exactly what the bf16x6 layer kernels execute per k-block.  M, N multiples of 32, K of 16. */
int dws_gemm_bf16x6(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, void* stream);

/* The arithmetic of precision="f16x3" alone, same shapes: every operand is multiplied by its scale (scale_a / scale_b,
 * powers of two: the layer kernels use 2^4 for activations, 2^12 for the gate and a per-matrix power of two that brings
 * the largest weight into (1, 2]), split into two fp16 terms (22 significand bits), three fp16 MFMA products per term
 * pair accumulated in fp32, the result multiplied by 1 / (scale_a scale_b).  Scaled operands beyond 65504 overflow. */
int dws_gemm_f16x3(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, float scale_a, float scale_b,
                   void* stream);

/* Timing of the dominant kernel, measured with HIP events on the stream the
 * kernel was launched on (bench.py roofline leg).  Enables per-launch event
 * recording for kernels whose name contains `substr`; query returns the number
 * of launches and their total milliseconds since enable. */
int dws_profile_enable(const char* substr);
int dws_profile_query(int64_t* launches, double* total_ms);
/* the same launches one by one, in launch order: ms[i] for i < min(*launches, capacity) */
int dws_profile_query_each(double* ms, int64_t capacity, int64_t* launches);
int dws_profile_disable(void);

#ifdef __cplusplus
}
#endif
#endif /* DWS_H_ */
