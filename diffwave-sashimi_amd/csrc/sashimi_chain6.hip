// Register-chained S4 tail (sashimi_chain.hip: everything of DiffWaveBlock.forward after the S4 convolution,
// `sashimi.py:177-184`, `s4.py:1435`, for H = 32, 64) with its three GEMMs on the bf16 matrix cores at fp32-equivalent
// accuracy: every operand as an exact 3-term bf16 split, six partial products per term pair, fp32 accumulate
// (bf16_split.h).  precision = "bf16x6".
//
// Why here: the f32 chain kernel is arithmetic-bound on ONE pipe -- v_mfma_f32_32x32x2_f32 runs at the VALU rate and
// shares the VALU's issue, so the tile's GELU / GLU / LayerNorm work adds to its MFMA time (phase trace
// profiles/r05_chain_phase_trace.txt: 2 x (25.2 k MFMA + ~7 k VALU) cycles of a 69 k-cycle tile pair per SIMD).  Six bf16
// MFMAs cost 6/16 of the f32 MFMA they replace and run beside the VALU.
//
// The chain carries over: the accumulator layout of a 32-row tile (register r of tile t = row 32 t + (r & 3) + 8 (r >> 2) +
// 4 lhi) is also a legal B-operand layout of v_mfma_f32_32x32x16_bf16 for the NEXT GEMM -- registers 8 hb .. 8 hb + 7 of
// tile t are the eight k values a lane supplies to k-block 2 t + hb (its k half = lhi) -- once the weight columns are
// packed in that order (chain16_permute_cols).  A lane splits its own eight fp32 values into three bf16x8 fragments in
// registers: no LDS round trip, no shuffle, no barrier between the stages, as before.
//
// Weights: 6 H^2 x 6 bytes (three bf16 terms) in LDS in A-fragment order -- 36 KB at H = 32, 144 KB at H = 64.
//
// precision = "f16x3": the same kernels instantiated with the 2-term fp16 split (bf16_split.h: SplitF16x2; three products,
// 4 bytes per weight): every GEMM's B operand (g, y, u: the register-resident activations) is multiplied by 2^4 before its
// split, every weight matrix by its own power of two (S4TailArgs::wscale_c6, weight_scale_kernel), the bias k-block carries
// both scales, and the accumulators are multiplied back by the exact inverse before GLU / GELU / LayerNorm see them.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "bf16_split.h"
#include "sashimi.h"
#include "sashimi_mfma.h"
#include "wavenet.h"

namespace dws {

typedef float c6_f32x4 __attribute__((ext_vector_type(4)));

// Column order of the 16-wide k-blocks: out[row][kappa] = W[row][32 t + 16 hb + (i & 3) + 8 (i >> 2) + 4 h],
// kappa = 32 t + 16 hb + 8 h + i  (t = source row tile, hb = its register half, h = the lane half that supplies it)
__global__ void chain16_permute_cols_kernel(const float* __restrict__ w, float* __restrict__ out, int M, int K) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)M * K) return;
    const int row = (int)(idx / K), kp = (int)(idx % K);
    const int t = kp >> 5, hb = (kp >> 4) & 1, h = (kp >> 3) & 1, i = kp & 7;
    out[idx] = w[(size_t)row * K + 32 * t + 16 * hb + (i & 3) + 8 * (i >> 2) + 4 * h];
}

int launch_chain16_permute_cols(const float* w, float* out, int M, int K, hipStream_t s) {
    DWS_CHECK(K % 32 == 0, DWS_ERR_INVALID, "chain16_permute_cols: K=%d", K);
    hipLaunchKernelGGL(chain16_permute_cols_kernel, dim3(ceil_div((int64_t)M * K, 256)), dim3(256), 0, s, w, out, M, K);
    return DWS_OK;
}

template <int H, int FFE, int NT>
struct Chain6Cfg {
    static constexpr int TH = H / 32, TO = 2 * H / 32, TF = FFE * H / 32;
    static constexpr int WAVES = (H >= 64) ? 8 : 4;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int KBH = H / 16, KBF = FFE * H / 16;                 // k-blocks of an H / ff H contraction
    static constexpr int WO_BYTES = 2 * H * H * 2 * NT, W1_BYTES = FFE * H * H * 2 * NT, W2_BYTES = FFE * H * H * 2 * NT;
    static constexpr int W_BYTES = WO_BYTES + W1_BYTES + W2_BYTES;
    static constexpr int B_FLOATS = 2 * H + FFE * H + H;                    // bo | b1 | b2
    static constexpr int LDS_BYTES = W_BYTES + B_FLOATS * 4;
    static_assert(H % 32 == 0 && LDS_BYTES <= 163840, "shape");
};

__device__ __forceinline__ float c6_xhalf_sum(float v) { return v + __shfl_xor(v, 32); }

// eight fp32 values of an accumulator-layout tile (register half hb) -> the NT B fragments of one k-block
template <typename P>
__device__ __forceinline__ void c6_bfrag(const bx_f32x16& v, int hb, typename P::v8 (&out)[P::NT]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) P::split1(P::SCALED ? v[8 * hb + i] * P::SX : v[8 * hb + i], out, i);
}

// bias k-block of a GEMM: acc[m] = (bias column x weight scale, NT terms) . (a row of ones x operand scale)
template <typename P, int MT, typename ACC>
__device__ __forceinline__ void c6_bias_block(ACC& acc, const float* __restrict__ bias, float ws, int l31, int lhi) {
    using v8 = typename P::v8;
    const v8 bf = P::bvals(lhi ? 0.f : P::SX, 0.f);
    bx_f32x16 zero;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        v8 af[P::NT];
        P::rank2(P::SCALED ? bias[m * 32 + l31] * ws : bias[m * 32 + l31], 0.f, lhi == 0, af);
        acc[m] = zero;
#pragma unroll
        for (int t = P::NT - 1; t >= 0; --t) acc[m] = P::mfma(af[t], bf, acc[m]);
    }
}

// acc[m] = W[m] . B + bias for all MT row tiles and NKB k-blocks; fragments of W at w + ((m NKB + kb) NT + term) 1024 + lane 16;
// src[t] are the accumulator-layout tiles whose registers are the B operand (k-block kb <- tile kb >> 1, half kb & 1).
// bias: acc starts from a rank-1 k-block (A = bias column in NT terms, B = a row of ones).  ws: the power of two the weights
// were packed with (scaled splits); the accumulators leave multiplied by 1 / (ws SX).
template <typename P, int MT, int NKB, int NS>
__device__ __forceinline__ void c6_gemm(bx_f32x16 (&acc)[MT], const char* __restrict__ w, const float* __restrict__ bias, float ws,
                                        const bx_f32x16 (&src)[NS], int lane, int l31, int lhi) {
    using v8 = typename P::v8;
    constexpr int NT = P::NT, NPR = P::NP;
    static_assert(NKB == 2 * NS, "two k-blocks per source tile");
    constexpr int MU = (MT % 2 == 0) ? 2 : 1, NU = MT / MU;      // row tiles per unit: two accumulators alternate in the MFMA stream
    const char* wl = w + lane * 16;
    c6_bias_block<P, MT>(acc, bias, ws, l31, lhi);
    v8 a_cur[MU][NT], a_nxt[MU][NT];
    auto load_a = [&](v8 (&dst)[MU][NT], int kb, int u) {
#pragma unroll
        for (int mm = 0; mm < MU; ++mm)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                dst[mm][t] = *reinterpret_cast<const v8*>(wl + (((u * MU + mm) * NKB + kb) * NT + t) * 1024);
    };
    load_a(a_cur, 0, 0);
    v8 bq[NT], bn[NT];
    c6_bfrag<P>(src[0], 0, bq);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const bool last_u = (u + 1 == NU);
            const int kbn = last_u ? kb + 1 : kb, un = last_u ? 0 : u + 1;
            if (kbn < NKB) load_a(a_nxt, kbn, un);
            __builtin_amdgcn_sched_barrier(0);   // the fragment reads stay a whole unit ahead of their MFMAs
            // the split of the NEXT k-block's operand rides in this unit's MFMA stream (the MFMAs do not depend on it)
            if (last_u && kb + 1 < NKB) c6_bfrag<P>(src[(kb + 1) >> 1], (kb + 1) & 1, bn);
#pragma unroll
            for (int t = 0; t < NPR; ++t)
#pragma unroll
                for (int mm = 0; mm < MU; ++mm)
                    acc[u * MU + mm] = P::mfma(a_cur[mm][P::ia(t)], bq[P::ib(t)], acc[u * MU + mm]);
#ifndef C6_NO_INTERLEAVE
            if (last_u && kb + 1 < NKB) {
#pragma unroll
                for (int i = 0; i < NPR * MU; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                 // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 48 / (NPR * MU), 0);   // a share of the ~44 split instructions
                }
            }
#endif
#pragma unroll
            for (int mm = 0; mm < MU; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t) a_cur[mm][t] = a_nxt[mm][t];
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) bq[t] = bn[t];
    }
    if (P::SCALED) {
        const float inv = 1.f / (ws * P::SX);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] *= inv;
    }
}

template <typename P, int H, int FFE, bool YNEXT>
__global__ __launch_bounds__((H >= 64 ? 512 : 256), 2) void s4_tail_chain6_kernel(S4TailArgs a) {
    using T = Chain6Cfg<H, FFE, P::NT>;
    constexpr int TH = T::TH, TO = T::TO, TF = T::TF;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char lds6[];
    char* const wo = lds6;                                   // [TO][KBH][NT][64] 16-byte fragments
    char* const w1 = wo + T::WO_BYTES;                       // [TF][KBH][NT][64]
    char* const w2 = w1 + T::W1_BYTES;                       // [TH][KBF][NT][64]
    float* const bo = reinterpret_cast<float*>(w2 + T::W2_BYTES);   // [2H]
    float* const b1 = bo + 2 * H;                            // [FFE*H]
    float* const b2 = b1 + FFE * H;                          // [H]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L, L4 = L * 4;

    {   // weights and biases -> LDS, once per workgroup (the only barrier of the kernel)
        const c6_f32x4* so = reinterpret_cast<const c6_f32x4*>(a.Ao_c6);
        const c6_f32x4* s1 = reinterpret_cast<const c6_f32x4*>(a.A1_c6);
        const c6_f32x4* s2 = reinterpret_cast<const c6_f32x4*>(a.A2_c6);
        c6_f32x4* d = reinterpret_cast<c6_f32x4*>(lds6);
        for (int i = tid; i < T::WO_BYTES / 16; i += T::THREADS) d[i] = so[i];
        for (int i = tid; i < T::W1_BYTES / 16; i += T::THREADS) d[T::WO_BYTES / 16 + i] = s1[i];
        for (int i = tid; i < T::W2_BYTES / 16; i += T::THREADS) d[(T::WO_BYTES + T::W1_BYTES) / 16 + i] = s2[i];
        for (int i = tid; i < 2 * H; i += T::THREADS) bo[i] = a.bo[i];
        for (int i = tid; i < FFE * H; i += T::THREADS) b1[i] = a.b1[i];
        for (int i = tid; i < H; i += T::THREADS) b2[i] = a.b2[i];
    }
    __syncthreads();

    const float ln_m = a.ln_m[0], ln_s = a.ln_s[0];
    const float n1_m = YNEXT ? a.n1_m[0] : 0.f, n1_s = YNEXT ? a.n1_s[0] : 0.f;
    const bool has_mel = a.mel != nullptr, has_add = a.addend != nullptr;
    const float one = lhi ? 0.f : 1.f;
    const float wso = P::SCALED ? a.wscale_c6[0] : 1.f, ws1 = P::SCALED ? a.wscale_c6[1] : 1.f, ws2 = P::SCALED ? a.wscale_c6[2] : 1.f;
    const int ntl = (L + 31) / 32, ntiles = a.B * ntl;
    const float invH = 1.f / (float)H;

#define C6_SOFF(t, r) ((32 * (t) + ((r) & 3) + 8 * ((r) >> 2)) * L4)
    unsigned long long* __restrict__ trc = a.trace ? a.trace + ((size_t)blockIdx.x * T::WAVES + wave) * 16 : nullptr;
    int tile_no = 0;
#define C6_STAMP(i)                                                        \
    if (trc && tile_no == 1) {                                             \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
        if (lane == 0) trc[i] = t_;                                        \
    }
    for (int tile = blockIdx.x * T::WAVES + wave; tile < ntiles; tile += gridDim.x * T::WAVES) {
        C6_STAMP(0)
        const int b = __builtin_amdgcn_readfirstlane(tile / ntl);
        const int l0 = __builtin_amdgcn_readfirstlane((tile % ntl) * 32);
        const int pos = l0 + l31;
        const int voff = pos < L ? (4 * lhi * L + pos) * 4 : OOB;
        __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * H * L), 0, H * L4, 0x00020000);
        bx_f32x16 g[TH], x1[TH];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rG, voff, C6_SOFF(t, r), 0));
                x1[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, voff, C6_SOFF(t, r), 0));
            }
        if (has_mel) {
            __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.mel + (size_t)(a.mel_bstride ? b : 0) * H * L), 0, H * L4, 0x00020000);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    x1[t][r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, voff, C6_SOFF(t, r), 0));
        }
        C6_STAMP(1)
        // ---- GEMM-o: o[2H x 32] = Wo g + bo
        bx_f32x16 ao[TO];
        c6_gemm<P, TO, T::KBH, TH>(ao, wo, bo, wso, g, lane, l31, lhi);
        C6_STAMP(2)
        // ---- GLU + residual: x1 = x (+ mel) + o_a * sigmoid(o_b); LN2 down the channel column
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x1[t][r] = fmaf(ao[t][r], dws_sigmoid(ao[TH + t][r]), x1[t][r]);
                s1 += x1[t][r];
            }
        const float mean = c6_xhalf_sum(s1) * invH;
        float sv = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x1[t][r] -= mean;                       // x1 holds the centred value from here on
                sv = fmaf(x1[t][r], x1[t][r], sv);
            }
        const float alpha = ln_s / sqrtf(c6_xhalf_sum(sv) * invH);
        bx_f32x16 y[TH];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[t][r] = alpha * (x1[t][r] + ln_m);
        C6_STAMP(3)
        // ---- GEMM-1: u[ff H x 32] = GELU(W1 y + b1)
        bx_f32x16 u[TF];
        c6_gemm<P, TF, T::KBH, TH>(u, w1, b1, ws1, y, lane, l31, lhi);
        C6_STAMP(4)
#ifndef C6_AD_LATE
        // the U-Net skip of this tile: requested now, needed after GEMM-2
        bx_f32x16 ad[TH];
        if (has_add) {
            __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.addend + (size_t)b * H * L), 0, H * L4, 0x00020000);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ad[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, voff, C6_SOFF(t, r), 0));
        }
#endif
#pragma unroll
        for (int m = 0; m < TF; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) u[m][r] = dws_gelu(u[m][r]);
        C6_STAMP(5)
        // ---- GEMM-2: f[H x 32] = W2 u + b2;  out = x1 + f (+ skip)
        bx_f32x16 f[TH];
        c6_gemm<P, TH, T::KBF, TF>(f, w2, b2, ws2, u, lane, l31, lhi);
        C6_STAMP(6)
#ifdef C6_AD_LATE
        // the U-Net skip of this tile
        bx_f32x16 ad[TH];
        if (has_add) {
            __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.addend + (size_t)b * H * L), 0, H * L4, 0x00020000);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ad[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, voff, C6_SOFF(t, r), 0));
        }
#endif
        float so = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (x1[t][r] + mean) + f[t][r];
                if (has_add) v += ad[t][r];
                f[t][r] = v;
                so += v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, voff, C6_SOFF(t, r), 0);
            }
        C6_STAMP(7)
        if constexpr (YNEXT) {
            // ---- the next block's S4 input: LN1_next down the columns of the output + its step-embedding projection, which
            // enters as an exact rank-1 product on the f32 matrix instruction (A = e column, B = row of ones)
            __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ynext + (size_t)b * H * L), 0, H * L4, 0x00020000);
            const float* eb = a.e_next + (size_t)b * a.e_stride + step_row_off(a.e_step, a.e_tstride);
            const float m2 = c6_xhalf_sum(so) * invH;
            float sv2 = 0.f;
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    f[t][r] -= m2;
                    sv2 = fmaf(f[t][r], f[t][r], sv2);
                }
            const float al2 = n1_s / sqrtf(c6_xhalf_sum(sv2) * invH);
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                const float ev = lhi ? 0.f : eb[t * 32 + l31];
                bx_f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                const bx_f32x16 et = __builtin_amdgcn_mfma_f32_32x32x2f32(ev, one, z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float yv = fmaf(al2, f[t][r] + n1_m, et[r]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), rY, voff, C6_SOFF(t, r), 0);
                }
            }
        }
        C6_STAMP(8)
        ++tile_no;
    }
#undef C6_SOFF
#undef C6_STAMP
}

bool s4_tail_chain6_supported(int H, int ff) { return ff == 2 && (H == 32 || H == 64); }

// =====================================================================================================================
// H = 128: the same chain, one wave per SIMD.  A wave still owns 32 positions and ALL channels of them (g, x1, the GLU
// pre-activations, u: up to 128 + 128 + 64 accumulator-layout registers at a time -- 512 registers per lane, so four waves
// per CU), but the weights (6 H^2 x 6 bytes = 576 KB) no longer fit in LDS: they STREAM through a ring of 24 KB chunks
// (one k-block of all row tiles of a GEMM; 24 chunks per tile, the same sequence for every tile) that the four waves fill
// by LDS-DMA several chunks ahead and all read -- every byte of weights crosses L2 -> CU once per 128 positions.
// One workgroup barrier per chunk (48 MFMAs per wave) both publishes the chunk that has landed and frees the slot that
// was consumed before it; the stream does not stop at tile boundaries (the next tile's first chunks arrive under GEMM-2).
// Fragment order in memory: k-block major, [k-block][row tile][term][lane] (pack_a_bx6_kmajor).
template <typename P>
__global__ void pack_a_bx6_kmajor_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int M, int K, const float* __restrict__ scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * K) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const size_t r = i >> 9;                       // kb * MT + mt
    const int MT = M / 32;
    const int mt = (int)(r % MT), kb = (int)(r / MT);
    float v = w[(size_t)(mt * 32 + (lane & 31)) * K + kb * 16 + 8 * (lane >> 5) + e];
    if (P::SCALED) v *= *scale;
    unsigned short b[P::NT];
    P::bits(v, b);
    const size_t base = (r * P::NT) * 512 + (size_t)lane * 8 + e;
#pragma unroll
    for (int t = 0; t < P::NT; ++t) out[base + t * 512] = b[t];
}

int launch_pack_a_bx6_kmajor(const float* w, void* out, int M, int K, int split, const float* scale, hipStream_t s) {
    DWS_CHECK(M % 32 == 0 && K % 16 == 0, DWS_ERR_UNSUPPORTED, "pack_a_bx6_kmajor: M=%d K=%d", M, K);
    if (split == WN_SPLIT_F16X3) {
        DWS_CHECK(scale != nullptr, DWS_ERR_INVALID, "pack_a_kmajor: the fp16 split needs the matrix scale");
        hipLaunchKernelGGL(pack_a_bx6_kmajor_kernel<SplitF16x2>, dim3(ceil_div((int64_t)M * K, 256)), dim3(256), 0, s, w, (unsigned short*)out, M, K, scale);
    } else {
        hipLaunchKernelGGL(pack_a_bx6_kmajor_kernel<SplitBf16x3>, dim3(ceil_div((int64_t)M * K, 256)), dim3(256), 0, s, w, (unsigned short*)out, M, K, scale);
    }
    return DWS_OK;
}

template <int H, int FFE, int NT>
struct Wide6Cfg {
    static constexpr int TH = H / 32, TO = 2 * H / 32, TF = FFE * H / 32;
    static constexpr int WAVES = 4, THREADS = 256;
    static constexpr int KBH = H / 16, KBF = FFE * H / 16;
    static constexpr int CHUNK = TO * NT * 1024;                           // bytes: one k-block of the 2H-row GEMMs
    static constexpr int KPC2 = TO / TH;                                   // k-blocks of GEMM-2 per chunk
    static constexpr int NCH = KBH + KBH + KBF / KPC2;                     // chunks per tile
    static constexpr int NSLOT = (NT == 3) ? 5 : 7, AHEAD = NSLOT - 1;     // ring slots; chunks requested ahead of use
    static constexpr int DPW = CHUNK / 1024 / WAVES;                       // LDS-DMA instructions per wave and chunk
    static constexpr int B_FLOATS = 2 * H + FFE * H + H;
    static constexpr int LDS_BYTES = NSLOT * CHUNK + B_FLOATS * 4;
    static_assert(TO == TF && TO % TH == 0 && KBF % KPC2 == 0 && (CHUNK / 1024) % WAVES == 0 && LDS_BYTES <= 163840, "shape");
};

template <typename P, int H, int FFE, bool YNEXT>
__global__ __launch_bounds__(256, 1) void s4_tail_wide6_kernel(S4TailArgs a) {
    using T = Wide6Cfg<H, FFE, P::NT>;
    using v8 = typename P::v8;
    constexpr int NT = P::NT, NPR = P::NP;
    constexpr int TH = T::TH, TO = T::TO, TF = T::TF;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) char ldsw[];
    float* const bo = reinterpret_cast<float*>(ldsw + T::NSLOT * T::CHUNK);
    float* const b1 = bo + 2 * H;
    float* const b2 = b1 + FFE * H;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L, L4 = L * 4;
    for (int i = tid; i < 2 * H; i += T::THREADS) bo[i] = a.bo[i];
    for (int i = tid; i < FFE * H; i += T::THREADS) b1[i] = a.b1[i];
    for (int i = tid; i < H; i += T::THREADS) b2[i] = a.b2[i];

    // ---- the weight stream: chunk number gc (counted over all tiles of this workgroup) lives in slot gc % NSLOT
    __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)a.Ao_c6, 0, T::NCH * T::CHUNK, 0x00020000);
    int gc = 0;                                                            // next chunk to consume (uniform)
    auto request = [&](int c) {                                            // this wave's share of chunk c
        const int src = (c % T::NCH) * T::CHUNK, slot = (c % T::NSLOT) * T::CHUNK;
#pragma unroll
        for (int i = 0; i < T::DPW; ++i) {
            const int piece = (wave + T::WAVES * i) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, reinterpret_cast<float*>(ldsw + slot + piece), 16, lane * 16, src + piece, 0, 0);
        }
    };
    const int ntl = (L + 127) / 128, ntiles = a.B * ntl;                   // workgroup tiles of 128 positions
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int total_chunks = my_tiles * T::NCH;
#pragma unroll
    for (int c = 0; c < T::AHEAD; ++c) request(c);
    // chunk gc has landed everywhere and the slot consumed before it is free again: one barrier for both, then the request
    // for the chunk AHEAD of this one goes into the freed slot
    auto next_chunk = [&]() -> const char* {
        // this wave's part of chunk gc has landed: everything but its requests for the AHEAD - 1 chunks behind it (other
        // VMEM issued since only makes the wait stricter); at the end of the stream fewer requests follow: wait for all
        if (gc + T::AHEAD <= total_chunks)
            __builtin_amdgcn_s_waitcnt(0x0F70 | (((T::AHEAD - 1) * T::DPW) & 15) | ((((T::AHEAD - 1) * T::DPW) >> 4) << 14));
        else
            __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (gc + T::AHEAD < total_chunks) request(gc + T::AHEAD);
        const char* p = ldsw + (gc % T::NSLOT) * T::CHUNK + lane * 16;
        ++gc;
        return p;
    };
    // acc[m] = W[m] . B + bias over NKB k-blocks whose fragments arrive KPC k-blocks per chunk, [k-block][row tile][term]
    auto gemm = [&](auto& acc, auto MT_, auto NKB_, auto KPC_, const float* bias, float ws, const auto& src) {
        constexpr int MT = decltype(MT_)::value, NKB = decltype(NKB_)::value, KPC = decltype(KPC_)::value;
        constexpr int NU = MT / 2;
        c6_bias_block<P, MT>(acc, bias, ws, l31, lhi);
        v8 bq[NT], bn[NT];
        c6_bfrag<P>(src[0], 0, bq);
#pragma unroll
        for (int ch = 0; ch < NKB / KPC; ++ch) {
            const char* wl = next_chunk();
            v8 a_cur[2][NT], a_nxt[2][NT];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int t = 0; t < NT; ++t) a_cur[mm][t] = *reinterpret_cast<const v8*>(wl + (mm * NT + t) * 1024);
#pragma unroll
            for (int kl = 0; kl < KPC; ++kl) {
                const int kb = ch * KPC + kl;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const bool last_u = (u + 1 == NU);
                    const int kln = last_u ? kl + 1 : kl, un = last_u ? 0 : u + 1;
                    if (kln < KPC) {
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                a_nxt[mm][t] = *reinterpret_cast<const v8*>(wl + (((kln * MT) + un * 2 + mm) * NT + t) * 1024);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (last_u && kb + 1 < NKB) c6_bfrag<P>(src[(kb + 1) >> 1], (kb + 1) & 1, bn);
#pragma unroll
                    for (int t = 0; t < NPR; ++t)
#pragma unroll
                        for (int mm = 0; mm < 2; ++mm)
                            acc[u * 2 + mm] = P::mfma(a_cur[mm][P::ia(t)], bq[P::ib(t)], acc[u * 2 + mm]);
                    if (last_u && kb + 1 < NKB) {
#pragma unroll
                        for (int i = 0; i < 2 * NPR; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 48 / (2 * NPR), 0);
                        }
                    }
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                        for (int t = 0; t < NT; ++t) a_cur[mm][t] = a_nxt[mm][t];
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) bq[t] = bn[t];
            }
        }
        if (P::SCALED) {
            const float inv = 1.f / (ws * P::SX);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] *= inv;
        }
    };

    const float ln_m = a.ln_m[0], ln_s = a.ln_s[0];
    const float n1_m = YNEXT ? a.n1_m[0] : 0.f, n1_s = YNEXT ? a.n1_s[0] : 0.f;
    const bool has_mel = a.mel != nullptr, has_add = a.addend != nullptr;
    const float one = lhi ? 0.f : 1.f;
    const float invH = 1.f / (float)H;
    const float wso = P::SCALED ? a.wscale_c6[0] : 1.f, ws1 = P::SCALED ? a.wscale_c6[1] : 1.f, ws2 = P::SCALED ? a.wscale_c6[2] : 1.f;
#define W6_SOFF(t, r) ((32 * (t) + ((r) & 3) + 8 * ((r) >> 2)) * L4)
    for (int wt = blockIdx.x; wt < ntiles; wt += gridDim.x) {
        const int b = __builtin_amdgcn_readfirstlane(wt / ntl);
        const int l0 = __builtin_amdgcn_readfirstlane((wt % ntl) * 128 + wave * 32);
        const int pos = l0 + l31;
        const int voff = pos < L ? (4 * lhi * L + pos) * 4 : OOB;
        __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * H * L), 0, H * L4, 0x00020000);
        bx_f32x16 g[TH], x1[TH];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rG, voff, W6_SOFF(t, r), 0));
                x1[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, voff, W6_SOFF(t, r), 0));
            }
        if (has_mel) {
            __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.mel + (size_t)(a.mel_bstride ? b : 0) * H * L), 0, H * L4, 0x00020000);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    x1[t][r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, voff, W6_SOFF(t, r), 0));
        }
        // ---- GEMM-o, GLU + residual, LN2
        bx_f32x16 ao[TO];
        gemm(ao, std::integral_constant<int, TO>{}, std::integral_constant<int, T::KBH>{}, std::integral_constant<int, 1>{}, bo, wso, g);
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x1[t][r] = fmaf(ao[t][r], dws_sigmoid(ao[TH + t][r]), x1[t][r]);
                s1 += x1[t][r];
            }
        const float mean = c6_xhalf_sum(s1) * invH;
        float sv = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x1[t][r] -= mean;
                sv = fmaf(x1[t][r], x1[t][r], sv);
            }
        const float alpha = ln_s / sqrtf(c6_xhalf_sum(sv) * invH);
        bx_f32x16 y[TH];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[t][r] = alpha * (x1[t][r] + ln_m);
        // ---- GEMM-1, GELU
        bx_f32x16 u[TF];
        gemm(u, std::integral_constant<int, TF>{}, std::integral_constant<int, T::KBH>{}, std::integral_constant<int, 1>{}, b1, ws1, y);
        bx_f32x16 ad[TH];
        if (has_add) {
            __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.addend + (size_t)b * H * L), 0, H * L4, 0x00020000);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ad[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, voff, W6_SOFF(t, r), 0));
        }
#pragma unroll
        for (int m = 0; m < TF; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) u[m][r] = dws_gelu(u[m][r]);
        // ---- GEMM-2, output
        bx_f32x16 f[TH];
        gemm(f, std::integral_constant<int, TH>{}, std::integral_constant<int, T::KBF>{}, std::integral_constant<int, T::KPC2>{}, b2, ws2, u);
        float so = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (x1[t][r] + mean) + f[t][r];
                if (has_add) v += ad[t][r];
                f[t][r] = v;
                so += v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, voff, W6_SOFF(t, r), 0);
            }
        if constexpr (YNEXT) {
            __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ynext + (size_t)b * H * L), 0, H * L4, 0x00020000);
            const float* eb = a.e_next + (size_t)b * a.e_stride + step_row_off(a.e_step, a.e_tstride);
            const float m2 = c6_xhalf_sum(so) * invH;
            float sv2 = 0.f;
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    f[t][r] -= m2;
                    sv2 = fmaf(f[t][r], f[t][r], sv2);
                }
            const float al2 = n1_s / sqrtf(c6_xhalf_sum(sv2) * invH);
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                const float ev = lhi ? 0.f : eb[t * 32 + l31];
                bx_f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                const bx_f32x16 et = __builtin_amdgcn_mfma_f32_32x32x2f32(ev, one, z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float yv = fmaf(al2, f[t][r] + n1_m, et[r]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), rY, voff, W6_SOFF(t, r), 0);
                }
            }
        }
    }
#undef W6_SOFF
}

bool s4_tail_wide6_supported(int H, int ff) { return ff == 2 && H == 128; }

template <typename P>
static int launch_wide6_p(const S4TailArgs& a, hipStream_t s) {
    using T = Wide6Cfg<128, 2, P::NT>;
    ProfileScope ps(P::NT == 3 ? "s4_tail_mfma_wide6" : "s4_tail_mfma_wide_f16x3", s);
    static int ncu_dev[DWS_MAX_DEVICES] = {};
    int& ncu = ncu_dev[current_device_slot()];
    if (ncu == 0) {
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_wide6_kernel<P, 128, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_wide6_kernel<P, 128, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        int dev = 0;
        DWS_HIP(hipGetDevice(&dev));
        DWS_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const int ntiles = a.B * ceil_div(a.L, 128);
    const int grid = std::min(ncu, ntiles);
    if (a.ynext) hipLaunchKernelGGL((s4_tail_wide6_kernel<P, 128, 2, true>), dim3(grid), dim3(T::THREADS), T::LDS_BYTES, s, a);
    else hipLaunchKernelGGL((s4_tail_wide6_kernel<P, 128, 2, false>), dim3(grid), dim3(T::THREADS), T::LDS_BYTES, s, a);
    return DWS_OK;
}

int launch_s4_tail_wide6(int H, const S4TailArgs& a, hipStream_t s) {
    DWS_CHECK(H == 128 && a.Ao_c6, DWS_ERR_STATE, "s4_tail_wide6: H=%d / weights not packed", H);
    if (a.split_c6 == WN_SPLIT_F16X3) {
        DWS_CHECK(a.wscale_c6 != nullptr, DWS_ERR_STATE, "s4_tail_wide6: no weight scales for the fp16 split");
        return launch_wide6_p<SplitF16x2>(a, s);
    }
    return launch_wide6_p<SplitBf16x3>(a, s);
}

template <typename F>
static void chain6_trace_launch(int H, int nwg, int waves, S4TailArgs a, hipStream_t s, F launch) {
    static const char* names[9] = {"", "issue g,x loads", "GEMM-o (incl. load wait, split of g)", "GLU+res+LN2", "GEMM-1 (incl. split of y)",
                                   "GELU (+skip request)", "GEMM-2 (incl. split of u)", "out stores", "next LN1 + stores"};
    unsigned long long* d = nullptr;
    const size_t n = (size_t)nwg * waves * 16;
    if (hipMalloc(&d, n * 8) != hipSuccess) return;
    (void)hipMemsetAsync(d, 0, n * 8, s);
    a.trace = d;
    launch(a);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    double ph[9] = {0}, life = 0;
    size_t cnt = 0;
    for (size_t w = 0; w < (size_t)nwg * waves; ++w) {
        const unsigned long long* t = &h[w * 16];
        if (!t[0] || !t[8]) continue;
        for (int i = 1; i < 9; ++i) ph[i] += (double)(t[i] - t[i - 1]);
        life += (double)(t[8] - t[0]);
        ++cnt;
    }
    if (!cnt) return;
    fprintf(stderr, "[chain6 trace] H=%d L=%d B=%d wgs=%d waves/wg=%d ynext=%d traced waves %zu; mean ticks per phase of a wave's 2nd tile:",
            H, a.L, a.B, nwg, waves, a.ynext ? 1 : 0, cnt);
    for (int i = 1; i < 9; ++i) fprintf(stderr, " %s %.0f |", names[i], ph[i] / cnt);
    fprintf(stderr, " tile %.0f\n", life / cnt);
}

template <typename P, int H>
static int launch_chain6_t(const S4TailArgs& a, hipStream_t s) {
    using T = Chain6Cfg<H, 2, P::NT>;
    ProfileScope ps(P::NT == 3 ? "s4_tail_mfma_chain6" : "s4_tail_mfma_chain_f16x3", s);
    const size_t lds = (size_t)T::LDS_BYTES;
    static int slots_dev[DWS_MAX_DEVICES] = {};
    int& slots = slots_dev[current_device_slot()];
    if (slots == 0) {
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_chain6_kernel<P, H, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_chain6_kernel<P, H, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0, ncu = 0, per_cu = 0;
        DWS_HIP(hipGetDevice(&dev));
        DWS_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        DWS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, s4_tail_chain6_kernel<P, H, 2, true>, T::THREADS, lds));
        DWS_CHECK(ncu > 0 && per_cu > 0, DWS_ERR_HIP, "s4_tail_chain6: occupancy query returned %d x %d", ncu, per_cu);
        slots = ncu * per_cu;
    }
    const int ntiles = a.B * ceil_div(a.L, 32);
    const int grid = std::min(slots, ceil_div(ntiles, T::WAVES));
    static const bool trace = std::getenv("DWS_CHAIN_TRACE") != nullptr;
    if (trace && a.ynext) {
        chain6_trace_launch(H, grid, T::WAVES, a, s, [&](const S4TailArgs& at) {
            hipLaunchKernelGGL((s4_tail_chain6_kernel<P, H, 2, true>), dim3(grid), dim3(T::THREADS), lds, s, at);
        });
        return DWS_OK;
    }
    if (a.ynext) hipLaunchKernelGGL((s4_tail_chain6_kernel<P, H, 2, true>), dim3(grid), dim3(T::THREADS), lds, s, a);
    else hipLaunchKernelGGL((s4_tail_chain6_kernel<P, H, 2, false>), dim3(grid), dim3(T::THREADS), lds, s, a);
    return DWS_OK;
}

int launch_s4_tail_chain6(int H, const S4TailArgs& a, hipStream_t s) {
    DWS_CHECK(a.Ao_c6 && a.A1_c6 && a.A2_c6, DWS_ERR_STATE, "s4_tail_chain6: the split chain-ordered weights were not packed");
    if (a.split_c6 == WN_SPLIT_F16X3) {
        DWS_CHECK(a.wscale_c6 != nullptr, DWS_ERR_STATE, "s4_tail_chain6: no weight scales for the fp16 split");
        if (H == 32) return launch_chain6_t<SplitF16x2, 32>(a, s);
        if (H == 64) return launch_chain6_t<SplitF16x2, 64>(a, s);
    } else {
        if (H == 32) return launch_chain6_t<SplitBf16x3, 32>(a, s);
        if (H == 64) return launch_chain6_t<SplitBf16x3, 64>(a, s);
    }
    return set_error(DWS_ERR_UNSUPPORTED, "s4_tail_chain6: H=%d not instantiated", H);
}

}  // namespace dws
