// MFMA (exact-f32) versions of the two heavy adjoints of the WaveNet residual layer.
//
// tapconv_mfma_kernel: position-tile GEMM with shifted taps, the data-gradient workhorse
//     out[m, l] = sum_{k < K} sum_{t < T} A[m, (k, t)] * src[k, l + sign * (t - T/2) * dil]      (0 outside [0, L))
//   * adjoint of the dilated conv w.r.t. its input (`wavenet.py:95`): M = C, K = 2C, T = 3, sign = -1,
//     A[c, (o, t)] = Wd[o, c, t]; epilogue adds dx' * sqrt(.5) (the residual path, `wavenet.py:121`)
//   * adjoint of the res/skip 1x1 convs (`wavenet.py:117-119`): M = C, K = S (+ C), T = 1 over the
//     concatenated inputs [dskip; dx'] with A = [Ws^T, sqrt(.5) Wr^T]; epilogue = gate adjoint
//     (dHt, dHs, g from the saved pre-activations, `wavenet.py:114`)
//   Same structure as the forward kernel: LDS-DMA staging through per-row descriptors (hardware zero
//   padding), A fragments streamed from L2 with a pinned one-group-ahead prefetch.
//
// wgrad_mfma_kernel: weight gradients, a GEMM whose contraction runs over POSITIONS
//     dW[o, c, t] = sum_{b, l} dY[b, o, l] * Xh[b, c, l + (t - T/2) * dil],   Xh = X (+ addc[b, c]) in range
//   128 x 128 output tile per block and tap, the (b, l) range split over blocks; partial tiles go to a
//   scratch buffer and are summed in a fixed order by wgrad_reduce_kernel (deterministic, no atomics).
#include <cstdlib>

#include "bf16_split.h"
#include "wavenet_backward.h"

// (Nontemporal cache policy on the streamed operands -- tapconv input DMA, its output stores, the wgrad DMA -- was
// measured on the config-5 step: 158-162 ms with and without, no effect.)

namespace dws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// 16-byte buffer store + the wait state hipcc leaves out (gfx950: a VALU write to the data registers of a 16-byte
// buffer_store that carries an SGPR offset, in the issue slot right behind it, corrupts the last dword of the last lanes of each
// 16-lane group -- DESIGN.md 10, found in wavenet_bx6.hip; the LayerNorm epilogue's store loop reproduced it: the next
// row's normalised values are computed into the registers the store still reads).
__device__ __forceinline__ void store4_hz(u32x4_t v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int aux) {
    __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, 0);
    (void)aux;
    asm volatile("s_nop 1" ::: "memory");
}
__device__ __forceinline__ float sigm_b(float x) { return dws_sigmoid(x); }

__device__ __forceinline__ float gelu_b(float x) { return dws_gelu(x); }
__device__ __forceinline__ float gelu_grad_b(float x) { return dws_gelu_grad(x); }  // Phi(x) + x phi(x)

// EPI (all rows m of an M-block, positions of the tile):
//   0  out = acc (+ addin * addscale)                       1  gate adjoint (WaveNet)
//   2  out = acc + bias[m]                                   3  out = acc + bias (pre-activation), out2 = gelu(out)
//   4  out = acc + bias + res (+ addend)                     5  out = acc * gelu'(aux)
//   6  GLU + residual (`s4.py:1435`, `sashimi.py:177`), M = 2H, MT = 2: a wave owns the 32-row tiles q and q + H/32
//      (the two GLU halves of the same channels): out = o = acc + bias [B,2H,L] and
//      out2 = x1 = res + o_a sigmoid(o_b) (+ aux)  [B,H,L]
//
// SPLIT (precision = "bf16x6" in training; T = 1, 16-byte staging only): the same GEMM on the bf16 matrix cores with every
// operand as an exact 3-term bf16 split and six partial products (bf16_split.h).  The staged fp32 chunk is split ONCE by
// the workgroup (one 16-byte B item of eight k values per thread and chunk, three terms) into a second LDS buffer while
// the previous chunk's MFMAs run; the A fragments stay the fp32 ones of pack_a_frag (the weights change every step) and
// are split by the wave that owns the rows.  Slot e = 4 g + j of a 16-wide k-block is k = 16 kb + 8 g + 2 j + lhi -- the
// order the fp32 fragments of k-groups 2 kb, 2 kb + 1 already hold per lane; the B items are built in the same order.
template <int MT, int T, int EPI, int SPLIT = 0>
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_waves_per_eu(2, 2))) void tapconv_mfma_kernel(TapConvArgs a) {
    constexpr int P = 64, NT = 2, KC = 32;
    constexpr int ROWS = T * KC, RPW = ROWS / 4;
    static_assert(!SPLIT || (T == 1 && EPI != 1), "split instances: pointwise GEMMs");
    constexpr int BOP_FLOATS = SPLIT ? (KC / 16) * 2 * 3 * P * 4 : 0;     // one split chunk: [k-block][k half][term][column] 16-byte items
    // T = 3: measured faster with 2 workgroups per CU than with 3 (456 -> 384 us on the C = 256 adjoint), so the
    // allocation is padded past a third of the 160 KB LDS
    constexpr int LDS_PAD = (T == 3) ? 2304 : 0;
    // one 32-row transposition tile per wave (float4 epilogue) + the LayerNorm epilogue's column partials [2][4 waves][64]
    constexpr int EPI_FLOATS = (EPI == 1) ? 0 : 4 * 32 * P + ((EPI == 4 || EPI == 6) ? 2 * 4 * P : 0);
    constexpr int MAIN_FLOATS = 2 * ROWS * P + LDS_PAD + 2 * BOP_FLOATS;
    constexpr int LDS_FLOATS = MAIN_FLOATS > EPI_FLOATS ? MAIN_FLOATS : EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L;
    const int ntl = (L + P - 1) / P;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int b = tile / ntl, l0 = (tile % ntl) * P;
    const int mt0 = blockIdx.y * (4 * MT);   // first 32-row tile of this M-block
    const int K = a.K0 + a.K1;
    const int ncb = K / KC;

    // T = 1 with L % 4 == 0: the chunk moves as 16-byte LDS-DMA, four whole rows (4 x 64 positions) per instruction -- two
    // instructions per wave and chunk instead of eight (each costs ~100 cycles of issue beside the chunk's 64 MFMAs per
    // wave); positions past L get an offset beyond the buffer (read 0).  Otherwise one dword per lane through a per-row
    // descriptor whose bounds check zero-fills the shifted taps.
    const bool x4 = (T == 1) && (L % 4 == 0) && ((((size_t)a.src0) | ((size_t)a.src1)) % 16 == 0);
    const int voff4 = (l0 + 4 * (lane & 15) < L) ? ((lane >> 4) * L + l0 + 4 * (lane & 15)) * 4 : 0x7ffffff0;
    auto stage_dma = [&](int cb, int buf) {
        float* xs = lds + buf * (ROWS * P);
        const int k0 = cb * KC;
        const float* base = (k0 < a.K0) ? a.src0 + ((size_t)b * a.K0 + k0) * L
                                        : a.src1 + ((size_t)b * a.K1 + (k0 - a.K0)) * L;
        if (T == 1 && x4) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, KC * L * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < KC / 16; ++i) {
                const int row = 4 * (wave + 4 * i);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, xs + row * P, 16, voff4, row * L * 4, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + 4 * i;
            const int tap = row / KC, cc = row % KC;
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)cc * L), 0, L * 4, 0x00020000);
            const int voff = (l0 + lane + a.sign * (tap - T / 2) * a.dil) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, xs + row * P, 4, voff, 0, 0, 0);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.M * a.nkg_total * 8 * 4, 0x00020000);
    const int lane16 = lane * 16;
    int mt[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) mt[m] = (EPI == 6) ? blockIdx.y * 4 + wave + m * (a.M / 64) : mt0 + wave * MT + m;
    // rows beyond M (M not a multiple of the M-block): the wave only helps staging; A loads are OOB -> 0
    const bool wave_live = mt[0] * 32 < a.M;

    // (Requesting ALL of the next chunk's A fragments and its LDS-DMA together at the top of a chunk and not waiting on
    // VMEM until the end-of-chunk barrier -- VMEM returns in order, so with the fragments fetched one k-group ahead the DMA
    // has to land within a k-group -- was built for T = 1 and measured: -13 % on the K = 2H GEMMs (8+ chunks), but +5..8 %
    // on the K = H ones (4 chunks: the longer prologue shows); choosing per launch by the chunk count kept both gains apart
    // but the larger kernel slowed its other path by 4 %: a wash over a training step either way.  Not kept.)
    if constexpr (SPLIT) {
        char* const bop = reinterpret_cast<char*>(lds + 2 * ROWS * P);
        // this thread's B item of a chunk: k-block wave >> 1, k half wave & 1 (wave-uniform), column lane
        auto transform = [&](int cb) {
            const float* xs = lds + (cb & 1) * (ROWS * P) + ((wave >> 1) * 16 + (wave & 1)) * P + lane;
            bx_bf16x8 it[3];
#pragma unroll
            for (int e = 0; e < 8; ++e) SplitBf16x3::split1(xs[(8 * (e >> 2) + 2 * (e & 3)) * P], it, e);
            char* dst = bop + (cb & 1) * (BOP_FLOATS * 4) + ((wave * 3) * P + lane) * 16;
#pragma unroll
            for (int t = 0; t < 3; ++t) *reinterpret_cast<bx_bf16x8*>(dst + t * (P * 16)) = it[t];
        };
        constexpr int NAF = MT * 2;                    // fp32 A fragments (k-groups) per k-block of 16
        f32x4 a_cur[MT][2], a_nxt[MT][2];
        const int nkb = ncb * (KC / 16);
        auto load_a = [&](f32x4 (&dst)[MT][2], int kb) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int g = 0; g < 2; ++g) dst[m][g] = buf_load4(rA, lane16, (mt[m] * a.nkg_total + 2 * kb + g) * 1024);
        };
        stage_dma(0, 0);
        if (ncb > 1) stage_dma(1, 1);
        load_a(a_cur, 0);
        // chunk 0 has landed: all but the loads issued after it (chunk 1: KC / 16 per wave, and the A fragments)
        if (ncb > 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (KC / 16 + NAF));
        else __builtin_amdgcn_s_waitcnt(0x0F70 | NAF);
        __syncthreads();
        transform(0);
        __builtin_amdgcn_s_waitcnt(0x0F70 | NAF);      // chunk 1 too
        __syncthreads();
        for (int cb = 0; cb < ncb; ++cb) {
            if (cb + 2 < ncb) stage_dma(cb + 2, cb & 1);   // raw buffer of chunk cb: split an iteration ago
            const char* bb = bop + (cb & 1) * (BOP_FLOATS * 4) + (lhi * 3 * P + l31) * 16;
#pragma unroll
            for (int kl = 0; kl < KC / 16; ++kl) {
                const int kb = cb * (KC / 16) + kl;
                load_a(a_nxt, kb + 1 < nkb ? kb + 1 : kb);
                __builtin_amdgcn_sched_barrier(0);
                bx_bf16x8 bq[NT][3];
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        bq[n][t] = *reinterpret_cast<const bx_bf16x8*>(bb + ((kl * 2 * 3 + t) * P + n * 32) * 16);
                bx_bf16x8 af[MT][3];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int e = 0; e < 8; ++e) SplitBf16x3::split1(a_cur[m][e >> 2][e & 3], af[m], e);
                if (kl == 0 && cb + 1 < ncb) transform(cb + 1);   // rides in this k-block's MFMA stream
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[m][BX6_IA[t]], bq[n][BX6_IB[t]], acc[m][n], 0, 0, 0);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int g = 0; g < 2; ++g) a_cur[m][g] = a_nxt[m][g];
            }
            // chunk cb + 2 has landed (younger: only the NAF fragments of the next k-block); split chunk cb + 1 visible
            __builtin_amdgcn_s_waitcnt(0x0F70 | NAF);
            __syncthreads();
        }
    } else {
    stage_dma(0, 0);
    f32x4 a_cur[MT], a_nxt[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = buf_load4(rA, lane16, (mt[m] * a.nkg_total) * 1024);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): hipcc does not make a barrier wait for LDS-DMA
    __syncthreads();
    const int nkg = ncb * (ROWS / 8);
    for (int cb = 0; cb < ncb; ++cb) {
        if (cb + 1 < ncb) stage_dma(cb + 1, (cb + 1) & 1);
        const float* xs = lds + (cb & 1) * (ROWS * P);
#pragma unroll
        for (int it = 0; it < ROWS / 8; ++it) {
            const int kg = cb * (ROWS / 8) + it;
            const int kgn = (kg + 1 < nkg) ? kg + 1 : kg;
#pragma unroll
            for (int m = 0; m < MT; ++m) a_nxt[m] = buf_load4(rA, lane16, (mt[m] * a.nkg_total + kgn) * 1024);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int krow = it * 8 + j * 2 + lhi;
                float bf[NT];
#pragma unroll
                for (int n = 0; n < NT; ++n) bf[n] = xs[krow * P + n * 32 + l31];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m][j], bf[n], acc[m][n], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // the LDS-DMA of chunk cb+1 (the only younger loads: the next k-group's A fragments)
        __syncthreads();
    }
    }


    const int M = a.M;
    if (!wave_live) return;
    if (EPI != 1 && (L & 3) == 0) {
        // float4 epilogue: the accumulators of this wave's MT tiles go through a private LDS tile [rows][64 positions]
        // and come back with 16 lanes per row, so every global access is a 16-byte buffer instruction covering 256
        // contiguous bytes of a row (the per-lane dword form issued 4x the VMEM instructions and ran the HBM-bound
        // 1x1 GEMMs at 3.2 TB/s).  (The last main-loop barrier has passed: the staging buffers are free.)
        float* wl = lds + wave * (32 * P);   // one tile at a time: LDS operations of a wave execute in order
        const size_t boff = (size_t)b * M * L;
        const int ML4 = M * L * 4, L4 = L * 4;
        constexpr int OOB = 0x7ffffff0;
        __amdgpu_buffer_rsrc_t rOut = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + boff), 0, ML4, 0x00020000);
        __amdgpu_buffer_rsrc_t rAux = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((EPI == 0 ? (a.addin ? a.addin : a.out) : EPI == 4 ? a.res : EPI == 5 ? a.aux : a.out) + boff), 0, ML4, 0x00020000);
        __amdgpu_buffer_rsrc_t rAdd = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == 4 && a.addend ? a.addend : a.out) + boff), 0, ML4, 0x00020000);
        __amdgpu_buffer_rsrc_t rOut2 = __builtin_amdgcn_make_buffer_rsrc((void*)((EPI == 3 ? a.out2 : a.out) + boff), 0, ML4, 0x00020000);
        __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)(a.bias ? a.bias : a.out), 0, M * 4, 0x00020000);
        const int lrow = lane >> 4, p4 = (lane & 15) * 4;
        const int voff = (l0 + p4 < L) ? (lrow * L + l0 + p4) * 4 : OOB;
        const bool has_bias = a.bias != nullptr, has_addin = a.addin != nullptr, has_addend = a.addend != nullptr;
        // LayerNorm epilogue (EPI 4 / 6, a.ln_out): the NV f32x4 values a lane keeps are rows (.. + lrow) x positions p4..p4+3
        // of the normalised tensor; the column statistics meet in LDS across the four row groups of a wave (lanes 16 apart)
        // and the four waves (the workgroup holds ALL channels: tapconv_ln_supported).  Two passes -- mean, then the centred
        // sum of squares -- as `torch.std_mean` does; every wave of the workgroup passes both barriers.
        auto ln_finish = [&](auto& keep, auto row_of, int nch) {
            constexpr int NV = sizeof(keep) / sizeof(f32x4);
            float* red = lds + 4 * 32 * P;
            auto column_total = [&](f32x4 part, int which) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    part[j] += __shfl_xor(part[j], 16, 64);
                    part[j] += __shfl_xor(part[j], 32, 64);
                }
                if (lrow == 0) *reinterpret_cast<f32x4*>(red + (which * 4 + wave) * P + p4) = part;
                __syncthreads();
                f32x4 t = *reinterpret_cast<const f32x4*>(red + (which * 4) * P + p4);
#pragma unroll
                for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + (which * 4 + w) * P + p4);
                return t;
            };
            f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NV; ++i) sum += keep[i];
            const f32x4 mean = column_total(sum, 0) * (1.f / (float)nch);
            f32x4 sq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const f32x4 dlt = keep[i] - mean;
                sq += dlt * dlt;
            }
            const f32x4 var = column_total(sq, 1) * (1.f / (float)nch);
            const float ls = a.ln_s[0], lm = a.ln_m[0];
            f32x4 scale, shift;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                scale[j] = ls / sqrtf(var[j]);
                shift[j] = lm - mean[j];
            }
            __amdgpu_buffer_rsrc_t rLn = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ln_out + (size_t)b * nch * L), 0, nch * L * 4, 0x00020000);
            const float* pt = a.ln_pt ? a.ln_pt + (size_t)b * a.ln_pt_bstride : nullptr;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int row0 = row_of(i);
                f32x4 y = scale * (keep[i] + shift);
                if (pt) y += pt[row0 + lrow];
                store4_hz(__builtin_bit_cast(u32x4_t, y), rLn, voff, row0 * L4, 0);
            }
        };
        if constexpr (EPI == 6) {
            static_assert(MT == 2, "the GLU epilogue pairs two M-tiles per wave");
            const int Hh = M / 2;
            const size_t boff2 = (size_t)b * Hh * L;
            const int HL4 = Hh * L * 4;
            __amdgpu_buffer_rsrc_t rX1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out2 + boff2), 0, HL4, 0x00020000);
            __amdgpu_buffer_rsrc_t rRes = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res + boff2), 0, HL4, 0x00020000);
            __amdgpu_buffer_rsrc_t rMel = __builtin_amdgcn_make_buffer_rsrc((void*)((a.aux ? a.aux : a.res) + boff2), 0, HL4, 0x00020000);
            const bool has_mel = a.aux != nullptr;
            f32x4 va[8], x1k[8];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * lhi) * P + n * 32 + l31] = acc[m][n][r];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row0 = mt[m] * 32 + it * 4;
                    f32x4 v = *reinterpret_cast<const f32x4*>(wl + (it * 4 + lrow) * P + p4);
                    if (has_bias) v += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBias, lrow * 4, row0 * 4, 0));
                    store4_hz(__builtin_bit_cast(u32x4_t, v), rOut, voff, row0 * L4, 0);
                    if (m == 0) {
                        va[it] = v;
                    } else {
                        const int soff2 = (mt[0] * 32 + it * 4) * L4;      // the channel rows in the [B,H,L] tensors
                        f32x4 x1 = buf_load4(rRes, voff, soff2);
                        if (has_mel) x1 += buf_load4(rMel, voff, soff2);
#pragma unroll
                        for (int j = 0; j < 4; ++j) x1[j] += va[it][j] * sigm_b(v[j]);
                        store4_hz(__builtin_bit_cast(u32x4_t, x1), rX1, voff, soff2, 0);
                        x1k[it] = x1;
                    }
                }
            }
            if (a.ln_out) ln_finish(x1k, [&](int i) { return mt[0] * 32 + i * 4; }, Hh);
            return;
        }
        f32x4 vk[EPI == 4 ? MT * 8 : 1];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) wl[((r & 3) + 8 * (r >> 2) + 4 * lhi) * P + n * 32 + l31] = acc[m][n][r];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row0 = (mt[m] * 32 + it * 4);
            const int soff = row0 * L4;
            f32x4 v = *reinterpret_cast<const f32x4*>(wl + (it * 4 + lrow) * P + p4);
            if (EPI == 0) {
                if (has_addin) {
                    const f32x4 ad = buf_load4(rAux, voff, soff);
                    v += ad * a.addscale;
                }
            } else if (EPI == 5) {
                const f32x4 ax = buf_load4(rAux, voff, soff);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= gelu_grad_b(ax[j]);
            } else {
                if (has_bias) {
                    const float bv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rBias, lrow * 4, row0 * 4, 0));
                    v += bv;
                }
                if (EPI == 4) {
                    v += buf_load4(rAux, voff, soff);
                    if (has_addend) v += buf_load4(rAdd, voff, soff);
                }
            }
            store4_hz(__builtin_bit_cast(u32x4_t, v), rOut, voff, soff, 0);
            if constexpr (EPI == 4) vk[m * 8 + it] = v;
            if (EPI == 3) {
                f32x4 gq;
#pragma unroll
                for (int j = 0; j < 4; ++j) gq[j] = gelu_b(v[j]);
                store4_hz(__builtin_bit_cast(u32x4_t, gq), rOut2, voff, soff, 0);
            }
        }
        }
        if constexpr (EPI == 4) {
            if (a.ln_out) ln_finish(vk, [&](int i) { return mt[i >> 3] * 32 + (i & 7) * 4; }, M);
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int pos = l0 + n * 32 + l31;
        const bool ok = pos < L;
        const int posc = ok ? pos : 0;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (EPI == 1) {  // gate adjoint: M == C, H / dH are [B, 2C, L]
                const float* __restrict__ Hb = a.H + (size_t)b * 2 * M * L;
                float* __restrict__ dHb = a.dH + (size_t)b * 2 * M * L;
                float* __restrict__ gb = a.g + (size_t)b * M * L;
                const int ML = M * L;
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 4) {   // four rows at a time keeps the register footprint down
                    float ht[4], hs[4];
                    const int i0 = (mt[m] * 32 + 8 * (r0 >> 2) + 4 * lhi) * L + posc;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ht[r] = Hb[i0 + r * L];
                        hs[r] = Hb[i0 + r * L + ML];
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float th = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + dws_exp(2.f * ht[r])), sg = sigm_b(hs[r]), d = acc[m][n][r0 + r];
                        if (ok) {
                            gb[i0 + r * L] = th * sg;
                            dHb[i0 + r * L] = d * sg * (1.f - th * th);
                            dHb[i0 + r * L + ML] = d * th * sg * (1.f - sg);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // do not hoist the next group's loads (they would spill)
                }
            } else {
                // 32-bit indices off per-batch base pointers: 64-bit address math per element costs ~2x the registers
                const size_t boff = (size_t)b * M * L;
                float* __restrict__ ob = a.out + boff;
                float ad[16];
                const int i0 = (mt[m] * 32 + 4 * lhi) * L + posc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = (r & 3) + 8 * (r >> 2);
                    const int row = mt[m] * 32 + rr + 4 * lhi;
                    const int idx = i0 + rr * L;
                    if (EPI == 0) ad[r] = a.addin ? (a.addin + boff)[idx] * a.addscale : 0.f;
                    else if (EPI == 2 || EPI == 3) ad[r] = a.bias ? a.bias[row] : 0.f;
                    else if (EPI == 4) ad[r] = a.bias[row] + (a.res + boff)[idx] + (a.addend ? (a.addend + boff)[idx] : 0.f);
                    else ad[r] = (a.aux + boff)[idx];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int idx = i0 + ((r & 3) + 8 * (r >> 2)) * L;
                    if (!ok) continue;
                    if (EPI == 5) {
                        ob[idx] = acc[m][n][r] * gelu_grad_b(ad[r]);
                    } else {
                        const float v = acc[m][n][r] + ad[r];
                        ob[idx] = v;
                        if (EPI == 3) (a.out2 + boff)[idx] = gelu_b(v);
                    }
                }
            }
        }
    }
}

bool tapconv_mfma_supported(int M, int K0, int K1, int T) {
    return M % 32 == 0 && M > 0 && K0 % 32 == 0 && K1 % 32 == 0 && (K0 + K1) > 0 && (T == 1 || T == 3);
}

// epilogue 6 exists for MT = 2 only (M % 256 == 0) and needs the float4 epilogue (L % 4 == 0)
bool tapconv_glu_supported(int M, int K, int L) {
    return tapconv_mfma_supported(M, K, 0, 1) && M % 256 == 0 && (L & 3) == 0;
}

// The LayerNorm epilogue needs every channel of a column in one workgroup (128 MT rows, one M-block) and 16-byte rows.
// epi 4 normalises its M output rows (M = 128 -> MT = 1, M = 256 -> MT = 2); epi 6 its M/2 rows of x1 (M = 256).
bool tapconv_ln_supported(int epi, int M, int L) {
    if ((L & 3) != 0) return false;
    if (epi == 4) return M == 128 || M == 256;
    if (epi == 6) return M == 256;
    return false;
}

// split instances: T = 1, not the gate adjoint, and the 16-byte staging form (the kernel's x4 condition)
static bool tapconv_split_ok(const TapConvArgs& a) {
    return a.split == 1 && a.T == 1 && a.epi != 1 && a.L % 4 == 0 && ((((size_t)a.src0) | ((size_t)a.src1)) % 16 == 0);
}

template <int T, int EPI>
static int launch_tc(const TapConvArgs& a, hipStream_t s) {
    const int nt = a.B * ceil_div(a.L, 64);
    if constexpr (T == 1 && EPI != 1) {
        if (tapconv_split_ok(a)) {
            if constexpr (EPI == 6) {
                hipLaunchKernelGGL((tapconv_mfma_kernel<2, T, EPI, 1>), dim3(nt, a.M / 256), dim3(256), 0, s, a);
            } else {
                if (a.M % 256 == 0)
                    hipLaunchKernelGGL((tapconv_mfma_kernel<2, T, EPI, 1>), dim3(nt, ceil_div(a.M, 256)), dim3(256), 0, s, a);
                else
                    hipLaunchKernelGGL((tapconv_mfma_kernel<1, T, EPI, 1>), dim3(nt, ceil_div(a.M, 128)), dim3(256), 0, s, a);
            }
            return DWS_OK;
        }
    }
    if constexpr (EPI == 6) {
        hipLaunchKernelGGL((tapconv_mfma_kernel<2, T, EPI>), dim3(nt, a.M / 256), dim3(256), 0, s, a);
    } else {
        if (a.M % 256 == 0)   // M-blocks are always full with MT = 2; with MT = 1 a partial block idles whole waves
            hipLaunchKernelGGL((tapconv_mfma_kernel<2, T, EPI>), dim3(nt, ceil_div(a.M, 256)), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((tapconv_mfma_kernel<1, T, EPI>), dim3(nt, ceil_div(a.M, 128)), dim3(256), 0, s, a);
    }
    return DWS_OK;
}

int launch_tapconv_mfma(const TapConvArgs& a, hipStream_t s) {
    ProfileScope ps(tapconv_split_ok(a) ? "tapconv_bx6" : "tapconv_mfma", s);
    DWS_CHECK(tapconv_mfma_supported(a.M, a.K0, a.K1, a.T), DWS_ERR_UNSUPPORTED, "tapconv_mfma: M=%d K=%d+%d T=%d", a.M,
              a.K0, a.K1, a.T);
    DWS_CHECK(!a.ln_out || (a.T == 1 && tapconv_ln_supported(a.epi, a.M, a.L)), DWS_ERR_UNSUPPORTED,
              "tapconv_mfma: no LayerNorm epilogue for epi %d, M=%d, L=%d", a.epi, a.M, a.L);
    if (a.T == 3) {
        DWS_CHECK(a.epi == 0, DWS_ERR_UNSUPPORTED, "tapconv_mfma: T=3 has epilogue 0 only");
        return launch_tc<3, 0>(a, s);
    }
    switch (a.epi) {
        case 0: return launch_tc<1, 0>(a, s);
        case 1: return launch_tc<1, 1>(a, s);
        case 2: return launch_tc<1, 2>(a, s);
        case 3: return launch_tc<1, 3>(a, s);
        case 4: return launch_tc<1, 4>(a, s);
        case 5: return launch_tc<1, 5>(a, s);
        case 6:
            DWS_CHECK(tapconv_glu_supported(a.M, a.K0, a.L) && a.K1 == 0 && a.out2 && a.res, DWS_ERR_UNSUPPORTED,
                      "tapconv_mfma: GLU epilogue needs M %% 256 == 0, L %% 4 == 0 (M=%d L=%d)", a.M, a.L);
            return launch_tc<1, 6>(a, s);
    }
    return set_error(DWS_ERR_INVALID, "tapconv_mfma: epilogue %d", a.epi);
}

// Row-major block of A in the kernel's K order k' = ((cb*T + tap)*KC + cc), o = cb*KC + cc, from a conv
// weight W[o][c][t] (o is the contraction index of the adjoint):
//     out[m = c][coff + k'] = scale * W[o][c][t]          (row stride ldo; pack_a_frag makes the fragments)
__global__ void tapconv_pack_transposed_kernel(const float* __restrict__ W, float* __restrict__ out, int O, int C,
                                               int T, int KC, int ldo, int coff, float scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * O * T) return;
    const int m = (int)(i / ((size_t)O * T)), kp = (int)(i % ((size_t)O * T));
    const int cb = kp / (T * KC), rem = kp % (T * KC), tap = rem / KC, cc = rem % KC;
    const int o = cb * KC + cc;
    out[(size_t)m * ldo + coff + kp] = W[((size_t)o * C + m) * T + tap] * scale;
}

int launch_tapconv_pack_transposed(const float* W, float* out, int O, int C, int T, int ldo, int coff, float scale,
                                   hipStream_t s) {
    const size_t n = (size_t)C * O * T;
    hipLaunchKernelGGL(tapconv_pack_transposed_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, W, out, O, C, T, 32, ldo,
                       coff, scale);
    return DWS_OK;
}

// ---------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------
// Chunk i+1 is fetched into registers while the MFMAs of chunk i run from LDS (software pipeline; the
// activation / addc / range masks are applied when the registers are written to LDS).  Blocks of the
// first c-tile and tap also accumulate sum_pos dY[o] (the conv's bias gradient) from the staged tile.
template <int T>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradArgs a) {
    constexpr int PC = 64, LD = PC + 2;   // +2: lanes walk channels (stride LD) and the two half-waves read adjacent positions: 66 = 2 mod 64 puts all 64 lanes on distinct banks (65 left a 2-way conflict between the halves)
    constexpr int RPW = 32;               // rows of each operand a wave stages per chunk
    __shared__ float sdy[128 * LD];
    __shared__ float sx[128 * LD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wo = wave & 1, wc = wave >> 1;                 // 2 x 2 waves over the 128 x 128 tile
    const int o0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
    const int tap = blockIdx.z % T, split = blockIdx.z / T;
    const int L = a.L, shift = (tap - T / 2) * a.dil;
    const int chunks_per_b = (L + PC - 1) / PC;
    const int total_chunks = a.B * chunks_per_b;
    const int per = (total_chunks + a.nsplit - 1) / a.nsplit;
    const int ch_begin = split * per, ch_end = min(total_chunks, ch_begin + per);
    const bool do_bias = a.bias_part != nullptr && blockIdx.y == 0 && tap == 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum = 0.f;

    // one descriptor per operand; the wave-uniform row offset rides in the scalar offset of the load
    __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, (int)((size_t)a.B * a.O * L * 4), 0x00020000);
    const int xL = a.xL ? a.xL : L;   // X rows may be longer than L (the conditioner's un-truncated upsampled mel)
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)((size_t)a.B * a.C * xL * 4), 0x00020000);
    float rdy[RPW], rx[RPW];
    float mx = 0.f;              // 1 where this lane's (shifted) position of the fetched chunk is in range
    int fb = 0, sad_b = -1;      // batch index of the fetched chunk / of the addc row staged in LDS
    __shared__ float sad[128];
    constexpr int OOB = 0x7ffffff0;   // lane offset past the descriptor: the load returns 0 (no mask multiply later)
    auto fetch = [&](int ch) {
        const int b = ch / chunks_per_b, l0 = (ch % chunks_per_b) * PC;
        const int pos = l0 + lane, ps = pos + shift;
        const bool pin = pos < L, sin = (unsigned)ps < (unsigned)L && pin;
        mx = sin ? 1.f : 0.f;
        fb = b;
        const int vy = pin ? pos * 4 : OOB, vx = sin ? ps * 4 : OOB;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + 4 * i;
            // rows past O / C are clamped, not masked: they only feed output elements that are never stored
            const int o = min(o0 + row, a.O - 1), c = min(c0 + row, a.C - 1);
            rdy[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rY, vy, (b * a.O + o) * L * 4, 0));
            rx[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, vx, (b * a.C + c) * xL * 4, 0));
        }
    };
    auto commit = [&]() {
        if (a.addc && sad_b != fb) {   // block-uniform: the per-(b, c) constants of this batch element -> LDS
            if (tid < 128) sad[tid] = a.addc[(size_t)fb * a.addc_bstride + min(c0 + tid, a.C - 1)];
            sad_b = fb;
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + 4 * i;
            sdy[row * LD + lane] = rdy[i];
            float xv = a.xact ? gelu_b(rx[i]) : rx[i];       // gelu(0) = 0: masked lanes stay 0
            if (a.addc) xv = fmaf(sad[row], mx, xv);
            sx[row * LD + lane] = xv;
        }
    };

    if (ch_begin < ch_end) fetch(ch_begin);
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        __syncthreads();           // the previous chunk's MFMAs are done with LDS
        commit();
        __syncthreads();
        if (ch + 1 < ch_end) fetch(ch + 1);
        __builtin_amdgcn_sched_barrier(0);   // keep the global loads ahead of the MFMA loop
        if (do_bias && tid < 128) {
#pragma unroll 8
            for (int p = 0; p < PC; ++p) bsum += sdy[tid * LD + p];
        }
#pragma unroll 8
        for (int ks = 0; ks < PC / 2; ++ks) {
            const int pp = ks * 2 + lhi;
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = sdy[(wo * 64 + i * 32 + l31) * LD + pp];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = sx[(wc * 64 + j * 32 + l31) * LD + pp];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // partial[split][o][c][t]
    float* part = a.partial + (size_t)split * a.O * a.C * T;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wo * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int c = c0 + wc * 64 + j * 32 + l31;
                if (o < a.O && c < a.C) part[((size_t)o * a.C + c) * T + tap] = acc[i][j][r];
            }
    if (do_bias && tid < 128 && o0 + tid < a.O) a.bias_part[(size_t)split * a.O + o0 + tid] = bsum;
}

// The same weight-gradient GEMM with NO register staging: both operand tiles of a chunk go global -> LDS by LDS-DMA
// (out-of-range positions zero-filled by the descriptor), double-buffered, so chunk i+1 lands while chunk i's MFMAs
// run and there is no commit phase (the ablation of wgrad_mfma_kernel showed that phase and its barriers cost ~25 %).
// For the plain case only: single tap, X used as stored (no activation, no per-(b,c) constant).  512 threads:
// 8 waves = 2 (o) x 4 (c), a wave owns a 64 x 32 sub-tile; 135 KB of LDS -> one workgroup per CU.
__global__ __launch_bounds__(512, 1) void wgrad_dma_kernel(WgradArgs a) {
    constexpr int PC = 64, LD = PC + 2, TILE = 128 * LD;
    extern __shared__ __attribute__((aligned(16))) float wlds[];   // [2 buffers][dY tile | X tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wo = wave & 1, wc = wave >> 1;
    const int o0 = blockIdx.x * 128, c0 = blockIdx.y * 128, split = blockIdx.z;
    const int L = a.L, xL = a.xL ? a.xL : L;
    const int chunks_per_b = (L + PC - 1) / PC;
    const int total_chunks = a.B * chunks_per_b;
    const int per = (total_chunks + a.nsplit - 1) / a.nsplit;
    const int ch_begin = split * per, ch_end = min(total_chunks, ch_begin + per);
    const bool do_bias = a.bias_part != nullptr && blockIdx.y == 0;
    constexpr int OOB = 0x7ffffff0;
    __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, (int)((size_t)a.B * a.O * L * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)((size_t)a.B * a.C * xL * 4), 0x00020000);

    auto stage = [&](int ch, int buf) {
        const int b = ch / chunks_per_b, l0 = (ch % chunks_per_b) * PC;
        const int pos = l0 + lane;
        const int voff = pos < L ? pos * 4 : OOB;
        float* sdy = wlds + buf * 2 * TILE;
        float* sx = sdy + TILE;
#pragma unroll
        for (int i = 0; i < 16; ++i) {           // 8 waves x 16 rows of each operand
            const int row = wave + 8 * i;
            const int o = min(o0 + row, a.O - 1), c = min(c0 + row, a.C - 1);   // rows past O / C feed unstored outputs only
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rY, sdy + row * LD, 4, voff, (b * a.O + o) * L * 4, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, sx + row * LD, 4, voff, (b * a.C + c) * xL * 4, 0, 0);
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float bsum = 0.f;

    if (ch_begin < ch_end) stage(ch_begin, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a barrier does not wait for this wave's LDS-DMA by itself
    __syncthreads();
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int buf = (ch - ch_begin) & 1;
        if (ch + 1 < ch_end) stage(ch + 1, buf ^ 1);
        const float* sdy = wlds + buf * 2 * TILE;
        const float* sx = sdy + TILE;
        if (do_bias && tid < 128) {
#pragma unroll 8
            for (int p = 0; p < PC; ++p) bsum += sdy[tid * LD + p];
        }
        // (Operands fetched one group of two k-steps ahead of their MFMAs, and the B fragments of the tapconv loop one
        // k-step ahead: 160-162 ms per config-5 step either way on the same box.  Not kept.)
#pragma unroll 8
        for (int ks = 0; ks < PC / 2; ++ks) {
            const int pp = ks * 2 + lhi;
            const float bv = sx[(wc * 32 + l31) * LD + pp];
            const float a0 = sdy[(wo * 64 + l31) * LD + pp], a1 = sdy[(wo * 64 + 32 + l31) * LD + pp];
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a barrier does not wait for this wave's LDS-DMA by itself
        __syncthreads();   // chunk ch+1 has landed and buffer `buf` is free again
    }
    float* part = a.partial + (size_t)split * a.O * a.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wo * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int c = c0 + wc * 32 + l31;
            if (o < a.O && c < a.C) part[(size_t)o * a.C + c] = acc[i][r];
        }
    if (do_bias && tid < 128 && o0 + tid < a.O) a.bias_part[(size_t)split * a.O + o0 + tid] = bsum;
}


// 16-byte form of the kernel above (L and the X row stride multiples of 4).  What the dword form pays for, per chunk of
// 64 positions and wave: 32 LDS-DMA instructions (~100 cycles of issue each beside MFMAs) and three ds_read_b32 per two
// MFMAs.  Here
//  * one LDS-DMA instruction moves 16 bytes per lane = four whole rows of a 128 x 64 operand tile (8 instructions per
//    wave and chunk).  The rows then sit 64 floats apart, which would put every row of a column in one bank; so the
//    16-byte quads of row r are stored rotated by r (slot (q + r) mod 16) -- the rotation happens on the GLOBAL side of
//    the DMA (a lane's source address), the LDS side stays "lane l -> bytes [16 l, 16 l + 16)";
//  * a lane reads one ds_read_b128 per operand and FOUR k-steps: the contraction index of MFMA e of block j is position
//    8 j + 4 lhi + e for both operands (any pairing of positions with k is as good as any other).  The sixteen lanes a
//    ds_read_b128 services together sit in sixteen different rows, i.e. sixteen different rotations: conflict-free.
// SPLIT (precision = "bf16x6"): both operands as exact 3-term bf16 splits, six products per 16 positions on
// v_mfma_f32_32x32x16_bf16; a lane supplies eight consecutive positions of its row (two rotated quads) and splits them in
// registers.  (The waves of a workgroup that share a row tile each split it again: ~165 VALU instructions per 12 MFMAs and
// wave.  Splitting block j + 1 under block j's MFMAs with sched_group_barrier was built and measured: 270 us against 254 us
// per launch -- the kernel now moves 1.07 GB per launch at 4.2 TB/s (the X tile is read by both 128-row halves of a 2H
// output): it is bound by those bytes, no longer by an execution pipe.)
template <int SPLIT>
__global__ __launch_bounds__(512, 1) void wgrad_dma4_kernel(WgradArgs a) {
    constexpr int PC = 64, TILE = 128 * PC;
    extern __shared__ __attribute__((aligned(16))) float wlds[];   // [2 buffers][dY tile | X tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wo = wave & 1, wc = wave >> 1;
    // The (o, c) tiles of one position range read the same dY / X rows (a 2H x H gradient: both 128-row halves read the
    // whole X tile).  Workgroups are dealt round-robin to the eight XCDs, each with its own L2: renumber so that the tiles of
    // a range run back to back on ONE XCD and the second reader finds the rows in that L2.  (Measured on the config-5 step:
    // 129.1 against 129.8 ms with the split kernels, 139.5 against 139.9 in f32 -- the second read was mostly served by
    // the 256 MB last-level cache already.)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#ifndef WGRAD_NO_XCD_GROUP
    {
        const int group = gridDim.x * gridDim.y;
        if (group > 1 && gridDim.z % 8 == 0) {
            const int n = bx + gridDim.x * (by + gridDim.y * bz);
            const int xcd = n & 7, slot = n >> 3;
            const int t = slot % group;
            bz = (slot / group) * 8 + xcd;
            bx = t % gridDim.x;
            by = t / gridDim.x;
        }
    }
#endif
    const int o0 = bx * 128, c0 = by * 128, split = bz;
    const int L = a.L, xL = a.xL ? a.xL : L;
    const int chunks_per_b = (L + PC - 1) / PC;
    const int total_chunks = a.B * chunks_per_b;
    const int per = (total_chunks + a.nsplit - 1) / a.nsplit;
    const int ch_begin = split * per, ch_end = min(total_chunks, ch_begin + per);
    const bool do_bias = a.bias_part != nullptr && by == 0;
    constexpr int OOB = 0x7ffffff0;
    __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, (int)((size_t)a.B * a.O * L * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)((size_t)a.B * a.C * xL * 4), 0x00020000);

    // staging: instruction i of a wave fills rows 4 g .. 4 g + 3 (g = wave + 8 i) of a tile; lane = (row in group, slot);
    // slot s of row r holds source quad (s - r) mod 16, and (s - r) mod 16 does not depend on i (32 i = 0 mod 16)
    const int rsub = lane >> 4, srcq = ((lane & 15) - rsub - 4 * wave) & 15;
    auto stage = [&](int ch, int buf) {
        const int b = ch / chunks_per_b, l0 = (ch % chunks_per_b) * PC;
        const int pos = l0 + 4 * srcq;
        const int voffY = pos < L ? (rsub * L + pos) * 4 : OOB;
        const int voffX = pos < L ? (rsub * xL + pos) * 4 : OOB;
        float* sdy = wlds + buf * 2 * TILE;
        float* sx = sdy + TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * (wave + 8 * i);      // rows past O / C feed outputs that are never stored
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rY, sdy + row * PC, 16, voffY, (b * a.O + o0 + row) * L * 4, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, sx + row * PC, 16, voffX, (b * a.C + c0 + row) * xL * 4, 0, 0);
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float bsum = 0.f;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int rowA = wo * 64 + l31, rowB = wc * 32 + l31;     // (rowA + 32 has the same rotation: 32 = 0 mod 16)

    if (ch_begin < ch_end) stage(ch_begin, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (int ch = ch_begin; ch < ch_end; ++ch) {
        const int buf = (ch - ch_begin) & 1;
        if (ch + 1 < ch_end) stage(ch + 1, buf ^ 1);
        const float* sdy = wlds + buf * 2 * TILE;
        const float* sx = sdy + TILE;
        if (do_bias) {     // row sums of dY: ALL 512 threads take a quarter row each (the row sum does not care about the rotation).
                           // Done by the first 128 threads alone (a whole row each) the two waves that own them reached every
                           // chunk barrier ~600 cycles after the other six.
            const int brow = tid >> 2, bq = (tid & 3) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(sdy + brow * PC + 4 * ((bq + q + brow) & 15));
                bsum += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
        // operands of block j + 1 are read while block j's eight MFMAs run (the sched_barriers pin that order): left to
        // itself hipcc reads one operand, waits for it (lgkmcnt(0)) and issues four MFMAs on ONE accumulator, eight registers
        // of operands in all -- every group of four MFMAs then starts with an exposed LDS round trip
        if constexpr (SPLIT) {
            f32x4 q0[2], q1[2], qb[2], n0[2], n1[2], nb[2];
            auto rd16 = [&](int jb, f32x4 (&x0)[2], f32x4 (&x1)[2], f32x4 (&y)[2]) {     // positions 16 jb + 8 lhi + 0..7
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int qa = (4 * jb + 2 * lhi + h + rowA) & 15, qq = (4 * jb + 2 * lhi + h + rowB) & 15;
                    x0[h] = *reinterpret_cast<const f32x4*>(sdy + rowA * PC + 4 * qa);
                    x1[h] = *reinterpret_cast<const f32x4*>(sdy + (rowA + 32) * PC + 4 * qa);
                    y[h] = *reinterpret_cast<const f32x4*>(sx + rowB * PC + 4 * qq);
                }
            };
            rd16(0, q0, q1, qb);
#pragma unroll
            for (int jb = 0; jb < PC / 16; ++jb) {
                if (jb + 1 < PC / 16) rd16(jb + 1, n0, n1, nb);
                __builtin_amdgcn_sched_barrier(0);
                bx_bf16x8 f0[3], f1[3], fb[3];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    SplitBf16x3::split1(q0[e >> 2][e & 3], f0, e);
                    SplitBf16x3::split1(q1[e >> 2][e & 3], f1, e);
                    SplitBf16x3::split1(qb[e >> 2][e & 3], fb, e);
                }
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0[BX6_IA[t]], fb[BX6_IB[t]], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1[BX6_IA[t]], fb[BX6_IB[t]], acc[1], 0, 0, 0);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) { q0[h] = n0[h]; q1[h] = n1[h]; qb[h] = nb[h]; }
            }
        } else {
        f32x4 a0, a1, bv, a0n, a1n, bvn;
        auto rd = [&](int j, f32x4& x0, f32x4& x1, f32x4& y) {
            const int qa = (2 * j + lhi + rowA) & 15, qb = (2 * j + lhi + rowB) & 15;
            x0 = *reinterpret_cast<const f32x4*>(sdy + rowA * PC + 4 * qa);
            x1 = *reinterpret_cast<const f32x4*>(sdy + (rowA + 32) * PC + 4 * qa);
            y = *reinterpret_cast<const f32x4*>(sx + rowB * PC + 4 * qb);
        };
        rd(0, a0, a1, bv);
#pragma unroll
        for (int j = 0; j < PC / 8; ++j) {
            if (j + 1 < PC / 8) rd(j + 1, a0n, a1n, bvn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bv[e], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bv[e], acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a0 = a0n; a1 = a1n; bv = bvn;
        }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // chunk ch+1 has landed (hipcc does not make a barrier wait for LDS-DMA)
        __syncthreads();                      // ... for every wave, and buffer `buf` is free again
    }
    float* part = a.partial + (size_t)split * a.O * a.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wo * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int c = c0 + wc * 32 + l31;
            if (o < a.O && c < a.C) part[(size_t)o * a.C + c] = acc[i][r];
        }
    if (do_bias) {     // the four quarter-row partials of a row sit in four adjacent lanes
        bsum += __shfl_xor(bsum, 1);
        bsum += __shfl_xor(bsum, 2);
        if ((tid & 3) == 0 && o0 + (tid >> 2) < a.O) a.bias_part[(size_t)split * a.O + o0 + (tid >> 2)] = bsum;
    }
}


// (A 256 x 128 / 128 x 256 output tile per workgroup -- no operand fetched twice by the two halves of a 2H dimension,
// 0.79 GB instead of 1.05 GB per launch at H = 128, 32-position chunks with a swizzle on the global side -- was built and
// measured SLOWER: 365 us against 316 us.  The kernel is not bound by those bytes.)

// out[i] = scale * sum_k partial[k][i], fixed order.  Four elements per thread (float4) and the k loop unrolled by
// four keeps 16 independent loads in flight per thread: the first version (one dependent load chain per thread)
// ran at 0.75 TB/s.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                           size_t n, int nsplit, float scale,
                                                           const float* __restrict__ partial2, float* __restrict__ dW2,
                                                           size_t n2, float scale2, int blocks1) {
    // blocks [0, blocks1): the weight partials; blocks [blocks1, ...): the bias partials of the same GEMM (one launch)
    int blk = blockIdx.x;
    if (blk >= blocks1) {
        blk -= blocks1;
        partial = partial2; dW = dW2; n = n2; scale = scale2;
    }
    const size_t i4 = ((size_t)blk * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 4 <= n && (n & 3) == 0) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {
            const float4 a = *reinterpret_cast<const float4*>(partial + (size_t)k * n + i4);
            const float4 b = *reinterpret_cast<const float4*>(partial + (size_t)(k + 1) * n + i4);
            const float4 c = *reinterpret_cast<const float4*>(partial + (size_t)(k + 2) * n + i4);
            const float4 d = *reinterpret_cast<const float4*>(partial + (size_t)(k + 3) * n + i4);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
            s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        for (; k < nsplit; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(partial + (size_t)k * n + i4);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
        float4 r;
        r.x = ((s0.x + s1.x) + (s2.x + s3.x)) * scale;
        r.y = ((s0.y + s1.y) + (s2.y + s3.y)) * scale;
        r.z = ((s0.z + s1.z) + (s2.z + s3.z)) * scale;
        r.w = ((s0.w + s1.w) + (s2.w + s3.w)) * scale;
        *reinterpret_cast<float4*>(dW + i4) = r;
    } else {
        for (size_t i = i4; i < n && i < i4 + 4; ++i) {
            float s = 0.f;
            for (int k = 0; k < nsplit; ++k) s += partial[(size_t)k * n + i];
            dW[i] = s * scale;
        }
    }
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {     // (wavenet_backward.hip's, same order)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// The reduce with the weight-norm adjoint on top (WgradArgs::wn_*): block o sums row o of the partials (the element order of
// wgrad_reduce_kernel), keeps it in LDS, and writes  dg[o] = <dW, v> / ||v||,  dv = (g / ||v||) (dW - dg v / ||v||)  -- the
// arithmetic of weight_norm_bwd_kernel (wavenet_backward.hip) on the row.  Blocks [O, ...): the bias partials.
__global__ __launch_bounds__(256) void wgrad_reduce_wn_kernel(const float* __restrict__ partial, size_t n, int nsplit, float scale,
                                                              const float* __restrict__ v, const float* __restrict__ g,
                                                              float* __restrict__ dv, float* __restrict__ dg, int O, int inner,
                                                              const float* __restrict__ partial2, float* __restrict__ dW2,
                                                              float scale2) {
    __shared__ float row[WGRAD_WN_MAX_INNER];
    __shared__ float red[4];
    const int o = blockIdx.x;
    if (o >= O) {        // bias partials: one element per thread, fixed order
        const int i = (o - O) * 256 + threadIdx.x;
        if (i < O) {
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int k = 0;
            for (; k + 4 <= nsplit; k += 4) {
                s0 += partial2[(size_t)k * O + i]; s1 += partial2[(size_t)(k + 1) * O + i];
                s2 += partial2[(size_t)(k + 2) * O + i]; s3 += partial2[(size_t)(k + 3) * O + i];
            }
            for (; k < nsplit; ++k) s0 += partial2[(size_t)k * O + i];
            dW2[i] = ((s0 + s1) + (s2 + s3)) * scale2;
        }
        return;
    }
    const float* vr = v + (size_t)o * inner;
    const float* pr = partial + (size_t)o * inner;
    float nn = 0.f, dot = 0.f;
    for (int i = threadIdx.x; i < inner; i += 256) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int k = 0;
        for (; k + 4 <= nsplit; k += 4) {
            s0 += pr[(size_t)k * n + i]; s1 += pr[(size_t)(k + 1) * n + i];
            s2 += pr[(size_t)(k + 2) * n + i]; s3 += pr[(size_t)(k + 3) * n + i];
        }
        for (; k < nsplit; ++k) s0 += pr[(size_t)k * n + i];
        const float d = ((s0 + s1) + (s2 + s3)) * scale;
        row[i] = d;
        nn = fmaf(vr[i], vr[i], nn);
        dot = fmaf(d, vr[i], dot);
    }
    nn = block_sum_256(nn, red);
    dot = block_sum_256(dot, red);
    const float norm = sqrtf(nn), dgo = dot / norm, sc = g[o] / norm;
    if (threadIdx.x == 0) dg[o] = dgo;
    for (int i = threadIdx.x; i < inner; i += 256) dv[(size_t)o * inner + i] = sc * (row[i] - dgo * vr[i] / norm);
}

int wgrad_mfma_nsplit(int B, int O, int C, int L, int T) {
    const int tiles = ceil_div(O, 128) * ceil_div(C, 128) * T;
    const int chunks = B * ceil_div(L, 64);
    // workgroups to aim for: the single-tap DMA kernel holds one workgroup per CU (135 KB of LDS), so 256 of them do the
    // work in one round with half the partial sums of 512 to write and reduce (same box, config-5 step: 161-163 ms
    // against 165 ms); the T = 3 kernel fits two per CU
    static const int t1_target = getenv("DWS_WGRAD_T1_TARGET") ? atoi(getenv("DWS_WGRAD_T1_TARGET")) : 256;
    const int target = (T == 1) ? t1_target : 512;
    int ns = std::max(1, target / tiles);
    return std::min(ns, chunks);
}

int launch_wgrad_mfma(const WgradArgs& a_in, int T, float scale, float* dW, hipStream_t s) {
    ProfileScope ps(a_in.split == 1 ? "wgrad_bx6" : "wgrad_mfma", s);
    DWS_CHECK((size_t)a_in.B * std::max(a_in.O, a_in.C) * std::max(a_in.L, a_in.xL) * 4 < ((size_t)1 << 31), DWS_ERR_UNSUPPORTED,
              "wgrad_mfma: operand larger than 2 GiB (B=%d rows=%d L=%d)", a_in.B, std::max(a_in.O, a_in.C), a_in.L);
    WgradArgs a = a_in;
    const dim3 grid(ceil_div(a.O, 128), ceil_div(a.C, 128), T * a.nsplit);
    static const bool no_dma = getenv("DWS_WGRAD_NO_DMA") != nullptr;
    static const bool no_dma4 = getenv("DWS_WGRAD_NO_DMA4") != nullptr;
    if (T == 1 && !a.xact && !a.addc && !no_dma && !no_dma4 && a.L % 4 == 0 && (a.xL ? a.xL : a.L) % 4 == 0 &&
        ((size_t)a.dY | (size_t)a.X) % 16 == 0) {
        constexpr int lds = 2 * 2 * 128 * 64 * 4;
        static bool attr4_dev[DWS_MAX_DEVICES] = {};
    bool& attr4 = attr4_dev[current_device_slot()];
        if (!attr4) {
            DWS_HIP(hipFuncSetAttribute((const void*)wgrad_dma4_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            DWS_HIP(hipFuncSetAttribute((const void*)wgrad_dma4_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr4 = true;
        }
        if (a.split == 1) hipLaunchKernelGGL(wgrad_dma4_kernel<1>, grid, dim3(512), lds, s, a);
        else hipLaunchKernelGGL(wgrad_dma4_kernel<0>, grid, dim3(512), lds, s, a);
    } else if (T == 1 && !a.xact && !a.addc && !no_dma) {
        constexpr int lds = 2 * 2 * 128 * 66 * 4;
        static bool attr_dev[DWS_MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device_slot()];
        if (!attr) {
            DWS_HIP(hipFuncSetAttribute((const void*)wgrad_dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr = true;
        }
        hipLaunchKernelGGL(wgrad_dma_kernel, grid, dim3(512), lds, s, a);
    } else if (T == 3) {
        hipLaunchKernelGGL(wgrad_mfma_kernel<3>, grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL(wgrad_mfma_kernel<1>, grid, dim3(256), 0, s, a);
    }
    const size_t n = (size_t)a.O * a.C * T;
    if (a.wn_v) {
        DWS_CHECK(T == 1 && a.C <= WGRAD_WN_MAX_INNER && a.wn_g && a.wn_dv && a.wn_dg, DWS_ERR_INVALID,
                  "wgrad: the fused weight-norm adjoint needs T = 1 and at most %d columns (T=%d C=%d)", WGRAD_WN_MAX_INNER, T, a.C);
        const int bias_blocks = a.bias_part ? ceil_div(a.O, 256) : 0;
        hipLaunchKernelGGL(wgrad_reduce_wn_kernel, dim3(a.O + bias_blocks), dim3(256), 0, s, a.partial, n, a.nsplit, scale, a.wn_v,
                           a.wn_g, a.wn_dv, a.wn_dg, a.O, a.C, (const float*)a.bias_part, a.dbias, a.bias_scale);
        return DWS_OK;
    }
    const int blocks1 = (int)ceil_div(n, 1024), blocks2 = a.bias_part ? (int)ceil_div(a.O, 1024) : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks1 + blocks2), dim3(256), 0, s, a.partial, dW, n, a.nsplit, scale,
                       (const float*)a.bias_part, a.dbias, (size_t)a.O, a.bias_scale, blocks1);
    return DWS_OK;
}

}  // namespace dws
