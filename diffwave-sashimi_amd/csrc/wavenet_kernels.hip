// WaveNet backbone kernels for gfx950 (MI355X).
//
// Hot path: one fused kernel per residual layer,
//   h = x + fc_t(emb)            (zero outside [0,L): padding applies to h)
//   H = Wd (*) h  (k=3, dilation d)      -> exact-f32 MFMA, K = 3C
//   g = tanh(H[:C]) * sigmoid(H[C:])     -> registers
//   [res; skip] = [Wr; Ws] g             -> exact-f32 MFMA, K = C
//   x' = (x + res) * sqrt(.5) ; skip_acc += skip
// following `models/wavenet.py:82-121,149-165`.
//
// Data layout: activations [B, C, L] fp32, L contiguous (the reference's
// layout): the MFMA B operand of v_mfma_f32_32x32x2_f32 wants, per k, 32
// consecutive positions -> a coalesced 128-byte row segment.  Weights are
// folded (weight-norm) and pre-packed ONCE into MFMA A-fragment order so every
// wave streams them with 1 KiB dwordx4 loads from L2.
#include "dws_common.h"
#include "model.h"
#include "wavenet.h"

namespace dws {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------
// Weight preparation (runs once per weight load, not per step)
// ---------------------------------------------------------------------------

// out[o,:] = g[o] * v[o,:] / ||v[o,:]||_2   (torch weight_norm, dim=0; `wavenet.py:21`); one 256-thread block per row o
__device__ __forceinline__ void fold_weight_norm_row(const float* __restrict__ v, const float* __restrict__ g,
                                                     float* __restrict__ out, int inner, int o, float* red) {
    const float* vr = v + (size_t)o * inner;
    float s = 0.f;
    for (int i = threadIdx.x; i < inner; i += 256) s += vr[i] * vr[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    const float scale = g[o] / sqrtf(red[0]);
    for (int i = threadIdx.x; i < inner; i += 256) out[(size_t)o * inner + i] = vr[i] * scale;
}

__global__ __launch_bounds__(256) void fold_weight_norm_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                               float* __restrict__ out, int inner) {
    __shared__ float red[256];
    fold_weight_norm_row(v, g, out, inner, blockIdx.x, red);
}

int launch_fold_weight_norm(const float* v, const float* g, float* out, int O, int inner, hipStream_t s) {
    hipLaunchKernelGGL(fold_weight_norm_kernel, dim3(O), dim3(256), 0, s, v, g, out, inner);
    return DWS_OK;
}

// Dilated-conv weight [2C][C][3] -> row-major [2C][3C] with the K order the
// fused kernel walks: k = ((cb*3 + tap)*KC + cc), c = cb*KC + cc.
__global__ void permute_dconv_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int KC, int M) {
    const int K = 3 * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * K) return;
    int o = (int)(i / K), k = (int)(i % K);
    int cb = k / (3 * KC), rem = k % (3 * KC), tap = rem / KC, cc = rem % KC;
    out[i] = w[((size_t)o * C + cb * KC + cc) * 3 + tap];
}

int launch_permute_dconv(const float* w, float* out, int C, int KC, hipStream_t s) {
    size_t n = (size_t)2 * C * 3 * C;
    hipLaunchKernelGGL(permute_dconv_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, out, C, KC, 2 * C);
    return DWS_OK;
}

// Row-major W[M][K] -> A-fragment order of v_mfma_f32_32x32x2_f32:
//   pack[mt][kg][lane][j] = W[mt*32 + (lane&31)][(kg*4 + j)*2 + (lane>>5)]
// so one dwordx4 load per lane yields the A operands of 4 consecutive k-steps.
__device__ __forceinline__ void pack_a_frag_at(const float* __restrict__ w, float* __restrict__ out, int M, int K, size_t i) {
    if (i >= (size_t)M * K) return;
    int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    size_t r = i >> 8;  // mt * (K/8) + kg
    int kg = (int)(r % (K / 8)), mt = (int)(r / (K / 8));
    int row = mt * 32 + (lane & 31), col = (kg * 4 + j) * 2 + (lane >> 5);
    out[i] = w[(size_t)row * K + col];
}
__global__ void pack_a_frag_kernel(const float* __restrict__ w, float* __restrict__ out, int M, int K) {
    pack_a_frag_at(w, out, M, K, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// The same fragment order for the TRANSPOSE of a row-major W[O][K] (the adjoint GEMM's operand W^T [K][O]) straight from
// W: one launch instead of transpose-to-scratch + pack (and no shared scratch buffer between consecutive weights).
__device__ __forceinline__ void pack_a_frag_t_at(const float* __restrict__ w, float* __restrict__ out, int O, int K, size_t i) {
    if (i >= (size_t)O * K) return;
    int j = (int)(i & 3), lane = (int)((i >> 2) & 63);
    size_t r = i >> 8;  // mt * (O/8) + kg   (rows of W^T = K, contraction = O)
    int kg = (int)(r % (O / 8)), mt = (int)(r / (O / 8));
    int row = mt * 32 + (lane & 31), col = (kg * 4 + j) * 2 + (lane >> 5);
    out[i] = w[(size_t)col * K + row];
}
__global__ void pack_a_frag_t_kernel(const float* __restrict__ w, float* __restrict__ out, int O, int K) {
    pack_a_frag_t_at(w, out, O, K, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

int launch_pack_a_frag_t(const float* w, float* out, int O, int K, hipStream_t s) {
    size_t n = (size_t)O * K;
    hipLaunchKernelGGL(pack_a_frag_t_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, out, O, K);
    return DWS_OK;
}

int launch_pack_a_frag(const float* w, float* out, int M, int K, hipStream_t s) {
    size_t n = (size_t)M * K;
    hipLaunchKernelGGL(pack_a_frag_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, out, M, K);
    return DWS_OK;
}

// Batched weight preparation (model.h: PrepBatch): block -> job by bisection of the jobs' first_block, then the job's own
// kernel body on the block's index inside the job.  PREP_ROW_SUM: four rows per block, one per wave (sashimi_mfma.hip:
// row_sum_kernel's lane-strided sum and butterfly, the same order).
int prep_job_blocks(int kind, int n0, int n1) {
    switch (kind) {
        case PREP_FOLD: return n0;
        case PREP_ROW_SUM: return ceil_div(n0, 4);
        default: return (int)ceil_div((size_t)n0 * n1, 256);
    }
}

__global__ __launch_bounds__(256) void weight_prep_kernel(const PrepJob* __restrict__ jobs, int njobs) {
    __shared__ float red[256];
    int lo = 0, hi = njobs - 1;
    const int blk = blockIdx.x;
    while (lo < hi) {       // the last job whose first_block <= blk (uniform: scalar loads)
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= blk) lo = mid; else hi = mid - 1;
    }
    const PrepJob j = jobs[lo];
    const int b = blk - j.first_block;
    switch (j.kind) {
        case PREP_FOLD: fold_weight_norm_row(j.a, j.b, j.out, j.n1, b, red); break;
        case PREP_PACK: pack_a_frag_at(j.a, j.out, j.n0, j.n1, (size_t)b * 256 + threadIdx.x); break;
        case PREP_PACK_T: pack_a_frag_t_at(j.a, j.out, j.n0, j.n1, (size_t)b * 256 + threadIdx.x); break;
        default: {
            const int o = b * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
            if (o < j.n0) {
                float s = 0.f;
                for (int k = lane; k < j.n1; k += 64) s += j.a[(size_t)o * j.n1 + k];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
                if (lane == 0) j.out[o] = s;
            }
        }
    }
}

int launch_weight_prep(const PrepJob* table_dev, int njobs, int nblocks, hipStream_t s) {
    if (njobs <= 0 || nblocks <= 0) return DWS_OK;
    hipLaunchKernelGGL(weight_prep_kernel, dim3(nblocks), dim3(256), 0, s, table_dev, njobs);
    DWS_HIP(hipGetLastError());
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// Diffusion-step embedding (`models/utils.py:20-27`) and small dense layers
// ---------------------------------------------------------------------------

// emb[b, i] = sin(t_b * f_i), emb[b, half+i] = cos(t_b * f_i)
__global__ void step_embed_kernel(const float* __restrict__ steps, const float* __restrict__ freq,
                                  float* __restrict__ emb, int half) {
    const int b = blockIdx.x;
    const float t = steps[b];
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float a = t * freq[i];
        emb[(size_t)b * 2 * half + i] = sinf(a);
        emb[(size_t)b * 2 * half + half + i] = cosf(a);
    }
}

__global__ void iota_f32_kernel(float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)i;
}
// out[i] = float(i): the step values t = 0..T-1 of a sampler's step table, as `generate.py:50` feeds them to the network
int launch_iota_f32(float* out, int n, hipStream_t s) {
    hipLaunchKernelGGL(iota_f32_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, out, n);
    return DWS_OK;
}

int launch_step_embed(const float* steps, const float* freq, float* emb, int B, int half, hipStream_t s) {
    hipLaunchKernelGGL(step_embed_kernel, dim3(B), dim3(64), 0, s, steps, freq, emb, half);
    return DWS_OK;
}

// out[b, o] = act(bias[o] + W[o,:] . in[b,:]); one wave per output row, the
// row is held in registers and reused for every batch element.
// act: 0 = identity, 1 = swish (`wavenet.py:10-11`).
template <int ACT>
__global__ void linear_rows_kernel(const float* __restrict__ in, const float* __restrict__ W,
                                   const float* __restrict__ bias, float* __restrict__ out, int B, int K, int O,
                                   float* __restrict__ pre_out) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (o >= O) return;
    constexpr int MAXR = 16;  // K <= 1024
    float w[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        int k = lane + 64 * i;
        w[i] = (k < K) ? W[(size_t)o * K + k] : 0.f;
    }
    const float bo = bias[o];
    for (int b = 0; b < B; ++b) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            int k = lane + 64 * i;
            if (k < K) acc = fmaf(w[i], in[(size_t)b * K + k], acc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if (lane == 0) {
            float y = acc + bo;
            if (pre_out) pre_out[(size_t)b * O + o] = y;   // training: pre-activation for the swish adjoint
            if (ACT == 1) y = y / (1.f + expf(-y));
            out[(size_t)b * O + o] = y;
        }
    }
}

int launch_linear_rows(const float* in, const float* W, const float* bias, float* out, int B, int K, int O,
                       int act, hipStream_t s, float* pre_out) {
    DWS_CHECK(K <= 1024, DWS_ERR_UNSUPPORTED, "linear_rows: K=%d > 1024 not supported", K);
    dim3 grid(ceil_div(O, 4)), block(256);
    if (act == 1)
        hipLaunchKernelGGL(linear_rows_kernel<1>, grid, block, 0, s, in, W, bias, out, B, K, O, pre_out);
    else
        hipLaunchKernelGGL(linear_rows_kernel<0>, grid, block, 0, s, in, W, bias, out, B, K, O, pre_out);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// init_conv: x[b,c,l] = relu(b_i[c] + sum_ci W_i[c,ci] * audio[b,ci,l])   (`wavenet.py:184,206`)
// ---------------------------------------------------------------------------
__global__ void init_conv_kernel(const float* __restrict__ audio, const float* __restrict__ W,
                                 const float* __restrict__ bias, float* __restrict__ x, int Cin, int C, int L) {
    const int b = blockIdx.z, c = blockIdx.y;
    const float bc = bias[c];
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        float acc = bc;
        for (int ci = 0; ci < Cin; ++ci) acc = fmaf(W[c * Cin + ci], audio[((size_t)b * Cin + ci) * L + l], acc);
        x[((size_t)b * C + c) * L + l] = dws_relu(acc);
    }
}

int launch_init_conv(const float* audio, const float* W, const float* bias, float* x, int B, int Cin, int C, int L,
                     hipStream_t s) {
    ProfileScope ps("init_conv", s);
    dim3 grid(min(ceil_div(L, 256), 64), C, B);
    hipLaunchKernelGGL(init_conv_kernel, grid, dim3(256), 0, s, audio, W, bias, x, Cin, C, L);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// Fused residual layer, exact-f32 MFMA path
// ---------------------------------------------------------------------------
template <int C, int S>
struct WnTile {
    static constexpr int P = 64;                            // positions per workgroup
    static constexpr int WAVES = 4;
    static constexpr int WM = (C / 32 >= 4) ? 4 : C / 32;   // waves along M
    static constexpr int WN = WAVES / WM;                   // waves along N
    static constexpr int NT = (P / 32) / WN;                // N tiles per wave
    static constexpr int MP = C / 32 / WM;                  // (tanh, sigmoid) tile pairs per wave
    static constexpr int MR = C / 32 / WM;                  // res tiles per wave
    static constexpr int MS = S / 32 / WM;                  // skip tiles per wave
    static constexpr int KC = WN_LAYER_KC;                  // channels per staged chunk
    static constexpr int NCB = C / KC;
    static constexpr int XS_FLOATS = 2 * 3 * KC * P;        // double-buffered x window
    static constexpr int G_FLOATS = C * P;                  // gate tile
    static constexpr int IND_FLOATS = 8 * P;                // tap-validity indicator rows (extra K rows)
    static constexpr int LDS_FLOATS = (XS_FLOATS + IND_FLOATS) > G_FLOATS ? (XS_FLOATS + IND_FLOATS) : G_FLOATS;
    static_assert(WN * NT * 32 == P, "N split");
    static_assert(C % (32 * WM) == 0 && S % (32 * WM) == 0 && C % KC == 0, "channel counts");
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// gate nonlinearities of the fused kernel: v_exp_f32 / v_rcp_f32 based (abs. error ~1e-7, far below
// the 1e-3 parity bound); tanh(x) = 1 - 2/(exp(2x)+1) saturates cleanly for |x| large
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
// tanh(t) * sigmoid(s) as ONE quotient: (e^{2t} - 1) / ((e^{2t} + 1)(1 + e^{-s})) -- three quarter-rate transcendentals
// (two v_exp_f32, one v_rcp_f32) instead of four.  t is clamped to +-30 (tanh(30) = 1 in fp32) so e^{2t} stays finite;
// e^{-s} may overflow to +inf, which gives the correct limit 0.
__device__ __forceinline__ float fast_gate(float t, float s) {
    const float tc = __builtin_amdgcn_fmed3f(t, -30.f, 30.f);
    const float e2 = __builtin_amdgcn_exp2f(tc * 2.8853900817779268f);      // e^{2t}
    const float en = __builtin_amdgcn_exp2f(s * -1.4426950408889634f);      // e^{-s}
    return (e2 - 1.f) * __builtin_amdgcn_rcpf((e2 + 1.f) * (1.f + en));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer load (SGPR descriptor + wave-uniform byte offset + per-lane offset): no 64-bit
// per-lane address registers, which is what lets the A-fragment double buffer stay in registers.
__device__ __forceinline__ f32x4 buf_load_f4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// EXTRA = false is the unconditional sampling instance: the conditioner add and the training save of the gate
// pre-activations (128 never-taken branches per tile otherwise) are compiled out.
template <int C, int S, bool EXTRA>
__global__ __launch_bounds__(256, 2) void wn_layer_mfma_kernel(WnLayerArgs a) {
    using T = WnTile<C, S>;
    constexpr int P = T::P, KC = T::KC, NT = T::NT, MP = T::MP, MR = T::MR, MS = T::MS;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % T::WM, wn = wave / T::WM;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int ntl = (a.L + P - 1) / P;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    // readfirstlane: the integer divisions run on the VALU, and without it hipcc treats everything derived from b
    // (every buffer descriptor below) as possibly divergent and wraps each access in a waterfall loop
    const int b = __builtin_amdgcn_readfirstlane(tile / ntl);
    const int l0 = __builtin_amdgcn_readfirstlane((tile % ntl) * P);
    const int L = a.L, dil = a.dilation;

    const float* __restrict__ xb = a.x_in + (size_t)b * C * L;

    // ---- staging: chunk cb = channels [cb*KC, cb*KC+KC) x 3 taps x P positions of RAW x, by LDS-DMA.
    // One descriptor per (channel) row with num_records = L*4 bytes: a tap position outside [0, L)
    // is out of range for the row and the hardware returns 0 -- the conv's zero padding for free,
    // no VGPRs, no address VALU, fully asynchronous.  The step-embedding term h = x + fc_t(e) is NOT
    // added here; it enters GEMM1 as three extra K rows (see `ind` below).
    constexpr int ROWS = 3 * KC;            // rows per chunk
    constexpr int RPW = ROWS / T::WAVES;    // rows per wave
    auto stage_dma = [&](int cb, int buf) {
        float* xs = lds + buf * (3 * KC * P);
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + T::WAVES * i;
            const int tap = row / KC, cc = row % KC;
            const int c = cb * KC + cc;
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)c * L), 0, L * 4, 0x00020000);
            const int voff = (l0 + lane + (tap - 1) * dil) * 4;  // negative -> huge unsigned -> out of range -> 0
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, xs + row * P, 4, voff, 0, 0, 0);
        }
    };

    // indicator rows: ind[t][col] = 1 if tap t of column col lies inside [0, L), rows 3..7 = 0.
    // GEMM1 gets one extra k-group  Abt[2C x 8] . ind[8 x P]  with Abt[o][t] = sum_c Wd[o,c,t] fc_t(e)[c]
    // (wn_bias_tap_kernel), which adds exactly Wd (*) (fc_t(e) on the in-range taps).
    float* ind = lds + T::XS_FLOATS;
    for (int i = tid; i < 8 * P; i += 256) {
        const int t = i / P, col = i % P;
        const int pos = l0 + col + (t - 1) * dil;
        ind[i] = (t < 3 && (unsigned)pos < (unsigned)L) ? 1.f : 0.f;
    }

    // ---- GEMM1: H[2C x P] = Wd[2C x 3C] . Xs[3C x P]
    f32x16 acc[2 * MP][NT];
#pragma unroll
    for (int m = 0; m < 2 * MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    constexpr int NKG1 = 3 * C / 8;  // k-groups (of 4 k-steps) per M tile
    __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A1, 0, 2 * C * 3 * C * 4, 0x00020000);
    const int lane16 = lane * 16;
    // tile ids of this wave: tanh tiles wm*MP+i, sigmoid tiles C/32 + wm*MP+i
    int mt1[2 * MP];
#pragma unroll
    for (int m = 0; m < 2 * MP; ++m) mt1[m] = (m < MP) ? (wm * MP + m) : (C / 32 + wm * MP + (m - MP));

    stage_dma(0, 0);
    f32x4 a_cur[2 * MP], a_nxt[2 * MP];
#pragma unroll
    for (int m = 0; m < 2 * MP; ++m) a_cur[m] = buf_load_f4(rA1, lane16, (mt1[m] * NKG1) * 1024);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a barrier does not wait for this wave's LDS-DMA by itself
    __syncthreads();

    for (int cb = 0; cb < T::NCB; ++cb) {
        if (cb + 1 < T::NCB) stage_dma(cb + 1, (cb + 1) & 1);
        const float* xs = lds + (cb & 1) * (3 * KC * P);
#pragma unroll
        for (int it = 0; it < 3 * KC / 8; ++it) {  // (tap, kg) flattened: 8 consecutive k per iteration
            const int kg = cb * (3 * KC / 8) + it;
            const int kgn = (kg + 1 < NKG1) ? kg + 1 : kg;
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m) a_nxt[m] = buf_load_f4(rA1, lane16, (mt1[m] * NKG1 + kgn) * 1024);
            // keep the prefetch a full k-group (32 MFMAs) ahead of its use: without this fence hipcc sinks
            // the loads below the MFMAs into the registers of a_cur and waits vmcnt(0) right after issuing them
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int krow = it * 8 + j * 2 + lhi;  // row inside the chunk (tap*KC + cc)
                float bf[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) bf[n] = xs[krow * P + (wn * NT + n) * 32 + l31];
#pragma unroll
                for (int m = 0; m < 2 * MP; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m][j], bf[n], acc[m][n], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m) a_cur[m] = a_nxt[m];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a barrier does not wait for this wave's LDS-DMA by itself
        __syncthreads();  // LDS-DMA of chunk cb+1 has landed (the only younger loads are the next k-group's A fragments)
    }
    // extra k-group: step-embedding correction rows
    {
        const f32x4* Abt = reinterpret_cast<const f32x4*>(a.Abt + (size_t)b * a.abt_bstride + step_row_off(a.step_idx, a.abt_tstride));
        f32x4 ab[2 * MP];
#pragma unroll
        for (int m = 0; m < 2 * MP; ++m) ab[m] = Abt[mt1[m] * 64 + lane];
#pragma unroll
        for (int j = 0; j < 2; ++j) {  // k = 0..3 (k = 3 is a zero row); k-steps 2,3 are all-zero and skipped
            const int krow = j * 2 + lhi;
            float bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = ind[krow * P + (wn * NT + n) * 32 + l31];
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[m][j], bf[n], acc[m][n], 0, 0, 0);
        }
    }
    __syncthreads();  // gate tile below aliases the staging + indicator regions

    // ---- gate: g = tanh(H_t + b_t (+mel_t)) * sigmoid(H_s + b_s (+mel_s)) -> LDS [C][P]
    float* gt = lds;
    const float* melb = a.melc ? a.melc + (size_t)(a.mel_bstride ? b : 0) * 2 * C * L : nullptr;
    // biases by buffer load: lane part 16*lhi, row part in the scalar offset (no per-lane 64-bit address math)
    __amdgpu_buffer_rsrc_t rB1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias1, 0, 2 * C * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias2, 0, (C + S) * 4, 0x00020000);
    const int vb = 16 * lhi;
#pragma unroll
    for (int m = 0; m < MP; ++m) {
        float bt_[16], bs_[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * MP + m) * 32 + (r & 3) + 8 * (r >> 2);
            bt_[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB1, vb, row * 4, 0));
            bs_[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB1, vb, (C + row) * 4, 0));
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = (wn * NT + n) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (wm * MP + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float ht = acc[m][n][r] + bt_[r];
                float hs = acc[MP + m][n][r] + bs_[r];
                if (EXTRA && melb) {
                    const int pos = l0 + col;
                    if (pos < L) {
                        ht += melb[ch * L + pos];
                        hs += melb[(C + ch) * L + pos];
                    }
                }
                if (EXTRA && a.hsave) {  // training: keep the pre-activations for the gate adjoint
                    const int pos = l0 + col;
                    if (pos < L) {
                        float* __restrict__ hb = a.hsave + (size_t)b * 2 * C * L;
                        hb[ch * L + pos] = ht;
                        hb[(C + ch) * L + pos] = hs;
                    }
                }
                gt[ch * P + col] = fast_gate(ht, hs);   // (same box: 71.6 vs 71.9 ms/step for tanh * sigmoid apart)
            }
        }
    }
    __syncthreads();

    // ---- GEMM2: [res; skip][(C+S) x P] = [Wr; Ws][(C+S) x C] . g[C x P]
    f32x16 acc2[MR + MS][NT];
#pragma unroll
    for (int m = 0; m < MR + MS; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][n][r] = 0.f;

    constexpr int NKG2 = C / 8;
    __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A2, 0, (C + S) * C * 4, 0x00020000);
    int mt2[MR + MS];
#pragma unroll
    for (int m = 0; m < MR + MS; ++m) mt2[m] = (m < MR) ? (wm * MR + m) : (C / 32 + wm * MS + (m - MR));
    f32x4 c_cur[MR + MS], c_nxt[MR + MS];
#pragma unroll
    for (int m = 0; m < MR + MS; ++m) c_cur[m] = buf_load_f4(rA2, lane16, (mt2[m] * NKG2) * 1024);
#pragma unroll 2
    for (int kg = 0; kg < NKG2; ++kg) {
        const int kgn = (kg + 1 < NKG2) ? kg + 1 : kg;
#pragma unroll
        for (int m = 0; m < MR + MS; ++m) c_nxt[m] = buf_load_f4(rA2, lane16, (mt2[m] * NKG2 + kgn) * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int krow = kg * 8 + j * 2 + lhi;
            float bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = gt[krow * P + (wn * NT + n) * 32 + l31];
#pragma unroll
            for (int m = 0; m < MR + MS; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc2[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(c_cur[m][j], bf[n], acc2[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MR + MS; ++m) c_cur[m] = c_nxt[m];
    }

    // ---- epilogue: x' = (x + res + b_r) * sqrt(.5);  skip_acc (+)= skip + b_s
    // Buffer loads / stores: the lane part of the address is ONE 32-bit offset per column group, the row part is
    // wave-uniform and rides in the scalar offset, so the 128 accesses of a lane cost no address VALU at all
    // (fp32 MFMA and VALU do not overlap on a SIMD: every VALU instruction here is MFMA time lost).  A lane whose
    // position is past L gets an out-of-range offset: its loads return 0 and its stores are dropped.
    const float rs = 0.70710678118654752440f;
    const bool first = a.first_layer, last = a.last_layer;
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, C * L * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rXo = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x_out + (size_t)b * C * L), 0, C * L * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rSk = __builtin_amdgcn_make_buffer_rsrc((void*)(a.skip + (size_t)b * S * L), 0, S * L * 4, 0x00020000);
    const int L4 = L * 4;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int pos = l0 + (wn * NT + n) * 32 + l31;
        const int voff = (pos < L) ? (4 * lhi * L + pos) * 4 : 0x7ffffff0;
        if (!last) {
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                const int s0 = ((wm * MR + m) * 32) * L4;
                float xr[16], br[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    xr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, voff, s0 + ((r & 3) + 8 * (r >> 2)) * L4, 0));
                    br[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB2, vb, ((wm * MR + m) * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0));
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (xr[r] + (acc2[m][n][r] + br[r])) * rs;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rXo, voff, s0 + ((r & 3) + 8 * (r >> 2)) * L4, 0);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            const int s0 = ((wm * MS + m) * 32) * L4;
            float sr[16], bq[16];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                bq[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB2, vb, (C + (wm * MS + m) * 32 + (r & 3) + 8 * (r >> 2)) * 4, 0));
            if (!first) {  // wave-uniform: all 16 loads issue back to back
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rSk, voff, s0 + ((r & 3) + 8 * (r >> 2)) * L4, 0));
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sr[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = sr[r] + (acc2[MR + m][n][r] + bq[r]);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rSk, voff, s0 + ((r & 3) + 8 * (r >> 2)) * L4, 0);
            }
        }
    }
}

// Abt[n][b][mt][lane][j]: A fragments (mfma 32x32x2 order, one k-group of 8) of the step-embedding
// correction  Abt[o][t] = sum_c Wd_n[o,c,t] * fc_t_n(e_b)[c],  t = 0..2 (k = t), k = 3..7 zero.
// One wave per (layer n, output row o); the folded weight row [C][3] stays in registers for all b.
__global__ void wn_bias_tap_kernel(const float* __restrict__ Wd_all, const float* __restrict__ part_t,
                                   float* __restrict__ Abt, int NL, int B, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // n * 2C + o
    if (row >= NL * 2 * C) return;
    const int n = row / (2 * C), o = row % (2 * C);
    const float* w = Wd_all + (size_t)row * C * 3;
    constexpr int MAXR = 8;  // C <= 512
    float w0[MAXR], w1[MAXR], w2[MAXR];
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int c = lane + 64 * i;
        const bool ok = c < C;
        w0[i] = ok ? w[c * 3 + 0] : 0.f;
        w1[i] = ok ? w[c * 3 + 1] : 0.f;
        w2[i] = ok ? w[c * 3 + 2] : 0.f;
    }
    for (int b = 0; b < B; ++b) {
        const float* pt = part_t + ((size_t)b * NL + n) * C;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int c = lane + 64 * i;
            const float p = (c < C) ? pt[c] : 0.f;
            s0 = fmaf(w0[i], p, s0); s1 = fmaf(w1[i], p, s1); s2 = fmaf(w2[i], p, s2);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off);
        }
        if (lane == 0) {
            float* dst = Abt + (((size_t)n * B + b) * (2 * C / 32) + o / 32) * 256;
            dst[(o % 32) * 4 + 0] = s0;          // k = 0: lane_hi 0, j 0
            dst[(32 + o % 32) * 4 + 0] = s1;     // k = 1: lane_hi 1, j 0
            dst[(o % 32) * 4 + 1] = s2;          // k = 2: lane_hi 0, j 1
        }
    }
}

int launch_wn_bias_tap(const float* Wd_all, const float* part_t, float* Abt, int NL, int B, int C, hipStream_t s) {
    DWS_CHECK(C <= 512, DWS_ERR_UNSUPPORTED, "wn_bias_tap: C=%d > 512", C);
    hipLaunchKernelGGL(wn_bias_tap_kernel, dim3(ceil_div((int64_t)NL * 2 * C, 4)), dim3(256), 0, s, Wd_all, part_t, Abt, NL,
                       B, C);
    return DWS_OK;
}

template <int C, int S>
static int launch_layer_t(const WnLayerArgs& a, hipStream_t s) {
    // (Start skews were measured and dropped: delaying the round-0 workgroups per CU, or the second workgroup of each
    // CU by up to half a tile time so the pair runs in anti-phase, left the step at 72.0-72.1 ms or made it slower.)
    // (Also measured and dropped in round 2, same box each: a 32-position tile with three or four workgroups per CU --
    // 71.3-71.9 ms on one box, 72.9-73.1 on another, against 71.5-72.1 for this shape; and B fragments read one k-step
    // ahead of their MFMAs instead of directly in front of them -- 71.62 against 71.46 ms.)
    ProfileScope ps("wn_layer_mfma", s);
    const int ntl = ceil_div(a.L, WnTile<C, S>::P);
    if (a.melc || a.hsave) hipLaunchKernelGGL((wn_layer_mfma_kernel<C, S, true>), dim3(a.B * ntl), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wn_layer_mfma_kernel<C, S, false>), dim3(a.B * ntl), dim3(256), 0, s, a);
    return DWS_OK;
}

bool wn_layer_mfma_supported(int C, int S) {
    return (C == 64 && S == 64) || (C == 128 && S == 128) || (C == 128 && S == 256) || (C == 256 && S == 256);
}

int launch_wn_layer_mfma(int C, int S, const WnLayerArgs& a, hipStream_t s) {
    if (C == 64 && S == 64) return launch_layer_t<64, 64>(a, s);
    if (C == 128 && S == 128) return launch_layer_t<128, 128>(a, s);
    if (C == 128 && S == 256) return launch_layer_t<128, 256>(a, s);
    if (C == 256 && S == 256) return launch_layer_t<256, 256>(a, s);
    return set_error(DWS_ERR_UNSUPPORTED, "wn_layer_mfma: (C=%d,S=%d) not instantiated", C, S);
}

// ---------------------------------------------------------------------------
// Generic residual layer (any C, S): plain FMA kernels used for channel counts
// the MFMA tiling does not cover (tiny test models).  Same math, same order of
// stages; two kernels because the gate needs all of H.
// ---------------------------------------------------------------------------
__global__ void wn_gate_generic_kernel(WnLayerArgs a, int C) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int L = a.L, d = a.dilation;
    const float* xb = a.x_in + (size_t)b * C * L;
    const float* pt = a.part_t + (size_t)b * a.part_t_bstride + step_row_off(a.step_idx, a.part_t_tstride);
    const float* wt = a.Wd + (size_t)c * C * 3;        // folded [2C][C][3]
    const float* ws = a.Wd + (size_t)(C + c) * C * 3;
    const float* melb = a.melc ? a.melc + (size_t)(a.mel_bstride ? b : 0) * 2 * C * L : nullptr;
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        float ht = a.bias1[c], hs = a.bias1[C + c];
        for (int ci = 0; ci < C; ++ci) {
            const float p = pt[ci];
#pragma unroll
            for (int tap = 0; tap < 3; ++tap) {
                const int pos = l + (tap - 1) * d;
                if (pos >= 0 && pos < L) {
                    const float hv = xb[(size_t)ci * L + pos] + p;
                    ht = fmaf(wt[ci * 3 + tap], hv, ht);
                    hs = fmaf(ws[ci * 3 + tap], hv, hs);
                }
            }
        }
        if (melb) {
            ht += melb[(size_t)c * L + l];
            hs += melb[(size_t)(C + c) * L + l];
        }
        if (a.hsave) {
            a.hsave[((size_t)b * 2 * C + c) * L + l] = ht;
            a.hsave[((size_t)b * 2 * C + C + c) * L + l] = hs;
        }
        a.gate_ws[((size_t)b * C + c) * L + l] = tanhf(ht) * sigmoidf_(hs);
    }
}

__global__ void wn_resskip_generic_kernel(WnLayerArgs a, int C, int S) {
    const int b = blockIdx.z, o = blockIdx.y;  // o < C: res row, else skip row o-C
    const int L = a.L;
    const float* g = a.gate_ws + (size_t)b * C * L;
    const float* w = (o < C) ? a.Wr + (size_t)o * C : a.Ws + (size_t)(o - C) * C;
    const float bias = a.bias2[o];
    if (o < C && a.last_layer) return;
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int ci = 0; ci < C; ++ci) acc = fmaf(w[ci], g[(size_t)ci * L + l], acc);
        acc += bias;
        if (o < C) {
            const size_t idx = ((size_t)b * C + o) * L + l;
            a.x_out[idx] = (a.x_in[idx] + acc) * 0.70710678118654752440f;
        } else {
            const size_t idx = ((size_t)b * S + (o - C)) * L + l;
            a.skip[idx] = a.first_layer ? acc : a.skip[idx] + acc;
        }
    }
}

int launch_wn_layer_generic(int C, int S, const WnLayerArgs& a, hipStream_t s) {
    ProfileScope ps("wn_layer_generic", s);
    const int gx = min(ceil_div(a.L, 256), 256);
    hipLaunchKernelGGL(wn_gate_generic_kernel, dim3(gx, C, a.B), dim3(256), 0, s, a, C);
    hipLaunchKernelGGL(wn_resskip_generic_kernel, dim3(gx, C + S, a.B), dim3(256), 0, s, a, C, S);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// final_conv: out = Wz . relu(Wf . (skip * scale) + bf) + bz   (`wavenet.py:165,198-200,208`)
// ---------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256, 2) void wn_final_mfma_kernel(WnFinalArgs a) {
    constexpr int P = (S >= 64) ? 64 : 128, WAVES = 4;   // S = 32: one M tile, the four waves split 128 positions
    constexpr int WM = (S / 32 >= 4) ? 4 : S / 32, WN = WAVES / WM, NT = (P / 32) / WN, MT = S / 32 / WM;
    static_assert(NT >= 1 && WN * NT * 32 == P, "N split");
    __shared__ __attribute__((aligned(16))) float lds[S * P];
    __shared__ float red[WAVES][P];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L;
    const int ntl = (L + P - 1) / P;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int b = tile / ntl, l0 = (tile % ntl) * P;

    const float* sk = a.skip + (size_t)b * S * L;
    // the skip tile [S][P]: 16-byte LDS-DMA when the rows allow it (one instruction = 256 consecutive floats of the tile =
    // 256 / P whole rows; columns past L get an out-of-range offset and read 0) -- no VALU, no VGPRs, a quarter of the
    // instructions; the 1/sqrt(n_layers) factor then sits in the packed weights (af_scaled).  Otherwise a dword loop.
    // (one 32-bit buffer descriptor spans a clip's [S][L] tensor: clips of 2 GiB and more keep the size_t dword loop)
    if ((L & 3) == 0 && (((size_t)sk & 15) == 0) && (a.af_scaled || a.scale == 1.f) && (long long)S * L * 4 < (1ll << 31)) {
        constexpr int F4_ROW = P / 4, RPI = 256 / P, NI = S / RPI / WAVES;
        static_assert(S % (RPI * WAVES) == 0, "DMA split");
        __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)sk, 0, S * L * 4, 0x00020000);
        float* const tile = lds;   // (the builtin takes a pointer variable, not the array expression: with `lds + ...` hipcc's
                                   // host pass silently drops the kernel's stub)
        const int pos = l0 + 4 * (lane % F4_ROW);
        const int voff = pos < L ? ((lane / F4_ROW) * L + pos) * 4 : 0x7ffffff0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row0 = (wave + WAVES * i) * RPI;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, tile + row0 * P, 16, voff, row0 * L * 4, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a barrier does not wait for this wave's LDS-DMA by itself
    } else {
        const float sc = a.af_scaled ? 1.f : a.scale;
        for (int i = tid; i < S * P; i += 256) {
            const int row = i / P, col = i % P;
            const int pos = l0 + col;
            lds[i] = (pos < L) ? sk[(size_t)row * L + pos] * sc : 0.f;
        }
    }
    __syncthreads();

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    constexpr int NKG = S / 8;
    const float4* A = reinterpret_cast<const float4*>(a.Af);
    // A fragments one k-group ahead of their MFMAs (a k-group feeds only 4 MT NT of them, less than an L2 round trip);
    // the sched_barrier pins the prefetch where it is written
    float4 av4[MT], avn[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) av4[m] = A[((size_t)(wm * MT + m) * NKG) * 64 + lane];
#pragma unroll 2
    for (int kg = 0; kg < NKG; ++kg) {
        const int kn = (kg + 1 < NKG) ? kg + 1 : kg;
#pragma unroll
        for (int m = 0; m < MT; ++m) avn[m] = A[((size_t)(wm * MT + m) * NKG + kn) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int krow = kg * 8 + j * 2 + lhi;
            float bf[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) bf[n] = lds[krow * P + (wn * NT + n) * 32 + l31];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const float av = (j == 0) ? av4[m].x : (j == 1) ? av4[m].y : (j == 2) ? av4[m].z : av4[m].w;
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bf[n], acc[m][n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) av4[m] = avn[m];
    }
    // y = relu(acc + bf[row]); optional tap; out[oc] = bz[oc] + sum_row Wz[oc,row] * y
    for (int oc = 0; oc < a.Cout; ++oc) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float part = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    const float y = dws_relu(acc[m][n][r] + a.bf[row]);
                    part = fmaf(a.Wz[oc * S + row], y, part);
                    if (oc == 0 && a.tap) {
                        const int pos = l0 + (wn * NT + n) * 32 + l31;
                        if (pos < L) a.tap[((size_t)b * S + row) * L + pos] = y;
                    }
                }
            part += __shfl_xor(part, 32);
            if (lhi == 0) red[wave][(wn * NT + n) * 32 + l31] = part;
        }
        __syncthreads();
        if (tid < P) {
            // waves with the same wn cover the same columns; sum over wm
            const int col = tid;
            const int wn_of_col = (col / 32) / NT;
            float sacc = a.bz[oc];
            for (int w = 0; w < WM; ++w) sacc += red[wn_of_col * WM + w][col];
            const int pos = l0 + col;
            if (pos < L) a.out[((size_t)b * a.Cout + oc) * L + pos] = sacc;
        }
        __syncthreads();
    }
}

__global__ void wn_final_generic_kernel(WnFinalArgs a, int S) {
    // one thread per position; y rows computed on the fly (tiny models only)
    const int b = blockIdx.y, L = a.L;
    const float* sk = a.skip + (size_t)b * S * L;
    for (int l = blockIdx.x * blockDim.x + threadIdx.x; l < L; l += gridDim.x * blockDim.x) {
        for (int oc = 0; oc < a.Cout; ++oc) {
            float o = a.bz[oc];
            for (int r = 0; r < S; ++r) {
                float acc = 0.f;
                for (int k = 0; k < S; ++k) acc = fmaf(a.Wf[(size_t)r * S + k], sk[(size_t)k * L + l] * a.scale, acc);
                const float y = dws_relu(acc + a.bf[r]);
                if (oc == 0 && a.tap) a.tap[((size_t)b * S + r) * L + l] = y;
                o = fmaf(a.Wz[oc * S + r], y, o);
            }
            a.out[((size_t)b * a.Cout + oc) * L + l] = o;
        }
    }
}

bool wn_final_mfma_supported(int S) { return S == 32 || S == 64 || S == 128 || S == 256; }

int launch_wn_final(int S, const WnFinalArgs& a, hipStream_t s) {
    ProfileScope ps("wn_final", s);
    const int ntl = ceil_div(a.L, S >= 64 ? 64 : 128);
    if (S == 32)
        hipLaunchKernelGGL((wn_final_mfma_kernel<32>), dim3(a.B * ntl), dim3(256), 0, s, a);
    else if (S == 64)
        hipLaunchKernelGGL((wn_final_mfma_kernel<64>), dim3(a.B * ntl), dim3(256), 0, s, a);
    else if (S == 128)
        hipLaunchKernelGGL((wn_final_mfma_kernel<128>), dim3(a.B * ntl), dim3(256), 0, s, a);
    else if (S == 256)
        hipLaunchKernelGGL((wn_final_mfma_kernel<256>), dim3(a.B * ntl), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(wn_final_generic_kernel, dim3(min(ceil_div(a.L, 128), 1024), a.B), dim3(128), 0, s, a, S);
    return DWS_OK;
}

}  // namespace dws
