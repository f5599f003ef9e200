// Fused WaveNet residual layer (`models/wavenet.py:82-121`) on the bf16 matrix cores at fp32-equivalent accuracy:
// the Winograd F(2,3) form of wavenet_wino.hip (four K = C products per position PAIR (p, p+d) instead of six) with
// every GEMM operand carried as a 3-term bf16 split and six of the nine partial products accumulated in fp32
// (bf16_split.h: what is dropped is below a quarter of an fp32 ulp of each product).  precision = "bf16x6".
//
// Per tile of 32 pairs (64 positions) and all channels, C/32 waves, a wave owns one (tanh, sigmoid) row-tile pair:
//   * raw x at the four shifts -d, 0, +d, +2d staged by LDS-DMA exactly as in wavenet_wino.hip (16-byte pieces,
//     out-of-range shifts read 0 = the conv's zero padding; one contiguous row piece for d <= 16);
//   * transform pass, all waves together once per chunk of KC channels: t0..t3 (fp32, the same roundings as the
//     f32 Winograd path), each split into three bf16 terms and stored in the B-fragment order of
//     v_mfma_f32_32x32x16_bf16: 16-byte items [k-octet][product][term][column], one conflict-free ds_read_b128 per
//     fragment;
//   * GEMM1: m_j = G_j t_j for the wave's two row tiles; the A fragments of G0..G3 (three bf16 terms each, packed at
//     commit: 6 bytes per weight) stream from L2 one (k-block, product) step ahead; 12 MFMAs per step;
//     step embedding + conv bias enter as one extra k-block per product (A = the fp32 correction row of the f32
//     Winograd path split in registers, B = the Winograd transform of the in-range indicator: exact in bf16);
//   * gate in registers, split, -> LDS gate tile [k-octet][term][64 columns];
//   * GEMM2 [res; skip] = [Wr; Ws] g with the biases as an extra k-block;
//   * epilogue: each wave transposes its row tiles through a private 8 KB LDS slot and moves them as row-major
//     16-byte accesses: x' = (x + res) sqrt(.5) with x re-read (L2-hot: this tile staged it), skip += as load-add-store
//     (one workgroup owns an element per layer, layers are stream-ordered).  The x / skip tiles are requested before
//     the gate stage.  Clips whose rows are not 16-byte aligned take a per-lane dword path.
//
// Work per launch at C = S = 256, B = 16, L = 16000: 204.5 GFLOP of fp32-equivalent GEMM (the f32 Winograd kernel's
// executed flops) = 1.227 PFLOP of bf16 MFMA = 0.49 ms at 2.5 PFLOP/s, against 1.30 ms at the fp32 matrix rate.
#include <cstdlib>
#include <type_traits>

#include "bf16_split.h"
#include "wavenet.h"
#include "wn_trace.h"

namespace dws {

#ifndef BX6_PF3
#define BX6_PF3 1   // A-fragment prefetch distance in (k-block, product) steps: 3-term split (12 MFMAs per step)
#endif
#ifndef BX6_PF2
#define BX6_PF2 2   // 2-term split (6 MFMAs per step)
#endif
#define BX6_PF(NT) ((NT) == 2 ? BX6_PF2 : BX6_PF3)
#ifndef BX6_T_PER_MFMA
#define BX6_T_PER_MFMA 7   // transform instructions dealt out behind each MFMA of a chunk's first step
#endif

typedef float bx6_f32x4 __attribute__((ext_vector_type(4)));

// The activation streams (x in, x' out, running skip in / out: each element touched once per launch, 1 GB in all) carry the
// nontemporal policy so that they do not push the weight fragments -- 3.75 MB that every tile re-reads -- out of the
// 4 MB L2 of an XCD.
#ifdef BX6_ABL_NO_NT
#define BX6_NT 0
#else
#define BX6_NT 2
#endif
__device__ __forceinline__ bx6_f32x4 bx6_load_f4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bx6_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, BX6_NT));
}
__device__ __forceinline__ void bx6_store_f4(bx6_f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, voff, soff, BX6_NT);
}

__device__ __forceinline__ float bx6_gate(float t, float s) {   // tanh(t) * sigmoid(s), see fast_gate (wavenet_kernels.hip)
    const float tc = __builtin_amdgcn_fmed3f(t, -30.f, 30.f);
    const float e2 = __builtin_amdgcn_exp2f(tc * 2.8853900817779268f);
    const float en = __builtin_amdgcn_exp2f(s * -1.4426950408889634f);
    return (e2 - 1.f) * __builtin_amdgcn_rcpf((e2 + 1.f) * (1.f + en));
}

// ---- weight packing (commit time) ----------------------------------------------------------------------------------
// Folded dilated-conv weight [2C][C][3] -> Winograd matrices G0..G3 (the f32 path's formulas and roundings,
// wavenet_wino.hip: wino_dconv_kernel) -> the split's NT terms in A-fragment order:
//   out[((((mt * NKB + kb) * 4 + j) * NT + term) * 64 + lane) * 8 + e] = term of G_j[mt*32 + (lane & 31)][kb*16 + 8*(lane >> 5) + e]
// Scaled splits (fp16 terms) multiply by *scale first: a power of two chosen per matrix by weight_scale_kernel.
template <typename P>
__global__ void pack_a1_bx6_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int C, const float* __restrict__ scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)2 * C * C * 4) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63), j = (int)((i >> 9) & 3);
    const size_t r = i >> 11;                     // mt * NKB + kb
    const int nkb = C / 16;
    const int kb = (int)(r % nkb), mt = (int)(r / nkb);
    const int o = mt * 32 + (lane & 31), c = kb * 16 + 8 * (lane >> 5) + e;
    const float* wr = w + ((size_t)o * C + c) * 3;
    const float w0 = wr[0], w1 = wr[1], w2 = wr[2];
    float g;
    if (j == 0) g = w0;
    else if (j == 1) g = 0.5f * ((w0 + w2) + w1);
    else if (j == 2) g = 0.5f * ((w0 + w2) - w1);
    else g = w2;
    if (P::SCALED) g *= *scale;
    unsigned short b[P::NT];
    P::bits(g, b);
    const size_t base = ((r * 4 + j) * P::NT) * 512 + (size_t)lane * 8 + e;
#pragma unroll
    for (int t = 0; t < P::NT; ++t) out[base + t * 512] = b[t];
}

// Row-major fp32 W[M][K] -> out[(((mt * NKB + kb) * NT + term) * 64 + lane) * 8 + e]
template <typename P>
__global__ void pack_a_bx6_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int M, int K, const float* __restrict__ scale) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * K) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const size_t r = i >> 9;
    const int nkb = K / 16;
    const int kb = (int)(r % nkb), mt = (int)(r / nkb);
    float v = w[(size_t)(mt * 32 + (lane & 31)) * K + kb * 16 + 8 * (lane >> 5) + e];
    if (P::SCALED) v *= *scale;
    unsigned short b[P::NT];
    P::bits(v, b);
    const size_t base = (r * P::NT) * 512 + (size_t)lane * 8 + e;
#pragma unroll
    for (int t = 0; t < P::NT; ++t) out[base + t * 512] = b[t];
}

// *out = the power of two that brings m = max(max|w|, max|bias| / 1024) into [1, 2)  (one workgroup; commit time).  The
// Winograd combinations G1, G2 of three taps then stay below 3, every fp16 high term is normal down to 2^-14 and the low terms
// resolve 2^-25 absolute = 2^-26 of the matrix maximum or better.  The bias rows ride in the same accumulators (correction
// k-block), so they enter the maximum far enough down that a bias 1000 x the weights neither overflows fp16 (< 2^11 scaled)
// nor costs the weights precision in the usual case.
__global__ __launch_bounds__(1024) void weight_scale_kernel(const float* __restrict__ w, size_t n, const float* __restrict__ bias, int nb,
                                                            const float* __restrict__ bias_b, int nb_b, float* __restrict__ out) {
    __shared__ float red[16];
    float m = 0.f;
    for (size_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(w[i]));
    if (bias)
        for (int i = threadIdx.x; i < nb; i += 1024) m = fmaxf(m, fabsf(bias[i]) * (1.f / 1024.f));
    if (bias_b)
        for (int i = threadIdx.x; i < nb_b; i += 1024) m = fmaxf(m, fabsf(bias_b[i]) * (1.f / 1024.f));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
        int ex = 0;
        if (m > 0.f && m < 3.0e38f) {
            frexpf(m, &ex);                         // m = f * 2^ex, f in [0.5, 1)  ->  m * 2^(1 - ex) in [1, 2)
            if (ex > 100) ex = 100;
            if (ex < -100) ex = -100;
        }
        *out = ldexpf(1.f, 1 - ex);
    }
}

int launch_weight_scale(const float* w, size_t n, const float* bias, int nb, const float* bias_b, int nb_b, float* out, hipStream_t s) {
    hipLaunchKernelGGL(weight_scale_kernel, dim3(1), dim3(1024), 0, s, w, n, bias, nb, bias_b, nb_b, out);
    return DWS_OK;
}

int launch_pack_a1_bx6(const float* w, void* out, int C, int split, const float* scale, hipStream_t s) {
    const size_t n = (size_t)2 * C * C * 4;
    if (split == WN_SPLIT_F16X3) {
        DWS_CHECK(scale != nullptr, DWS_ERR_INVALID, "pack_a1: the fp16 split needs the matrix scale");
        hipLaunchKernelGGL(pack_a1_bx6_kernel<SplitF16x2>, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, (unsigned short*)out, C, scale);
    } else {
        hipLaunchKernelGGL(pack_a1_bx6_kernel<SplitBf16x3>, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, (unsigned short*)out, C, scale);
    }
    return DWS_OK;
}

int launch_pack_a_bx6(const float* w, void* out, int M, int K, int split, const float* scale, hipStream_t s) {
    DWS_CHECK(M % 32 == 0 && K % 16 == 0, DWS_ERR_UNSUPPORTED, "pack_a_bx6: M=%d K=%d", M, K);
    const size_t n = (size_t)M * K;
    if (split == WN_SPLIT_F16X3) {
        DWS_CHECK(scale != nullptr, DWS_ERR_INVALID, "pack_a: the fp16 split needs the matrix scale");
        hipLaunchKernelGGL(pack_a_bx6_kernel<SplitF16x2>, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, (unsigned short*)out, M, K, scale);
    } else {
        hipLaunchKernelGGL(pack_a_bx6_kernel<SplitBf16x3>, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, (unsigned short*)out, M, K, scale);
    }
    return DWS_OK;
}

// ---- the layer kernel ----------------------------------------------------------------------------------------------
template <int C, int S, int NT>
struct Bx6Tile {
    static constexpr int WAVES = C / 32;          // one (tanh, sigmoid) tile pair per wave
    static constexpr int NTH = WAVES * 64;
    static constexpr int NP = 32;                 // position pairs per workgroup (64 positions)
    static constexpr int KC = (C >= 256) ? 32 : 16;   // channels per staged chunk (between two workgroup barriers)
    static constexpr int NCB = C / KC;
    static constexpr int NKB = C / 16;            // k-blocks of 16 channels
    static constexpr int MS = S / C;              // skip tiles per wave
    static constexpr int RAW_FLOATS = KC * 4 * NP;          // one raw chunk: [row][shift][32] (or [row][128] contiguous)
    static constexpr int BOP_BYTES = (KC / 8) * 4 * NT * NP * 16;   // one transformed chunk: [octet][product][term][column] items
    static constexpr int STAGE_FLOATS = 2 * RAW_FLOATS + 2 * BOP_BYTES / 4;
    static constexpr int GATE_BYTES = (C / 8) * NT * 64 * 16;       // gate tile: [octet][term][column] items
    static constexpr int MAIN_FLOATS = STAGE_FLOATS > GATE_BYTES / 4 ? STAGE_FLOATS : GATE_BYTES / 4;
    static constexpr int TR_FLOATS = 32 * 64;     // a wave's transpose slot: one row tile x 64 columns
    static constexpr int LDS_FLOATS = MAIN_FLOATS + WAVES * TR_FLOATS;
    static constexpr int TITEMS = (KC / 4) * NP * 2;        // transform items: (4-channel group, column, product pair)
    static_assert(C % 32 == 0 && S % C == 0 && C % KC == 0 && KC % 16 == 0 && KC % (2 * WAVES) == 0 &&
                  (TITEMS % NTH == 0) && LDS_FLOATS * 4 <= 163840, "channel counts");
};

template <typename P, int C, int S, bool EXTRA>
__global__ __launch_bounds__(C * 2, (C >= 128 ? 2 : 1)) void wn_layer_bx6_kernel(WnLayerArgs a, int log2d) {
    using T = Bx6Tile<C, S, P::NT>;
    using v8 = typename P::v8;
    using v4 = typename P::v4;
    constexpr int NT = P::NT, NPR = P::NP, NA = 2 * P::NT;   // terms, products per term pair, A fragments per GEMM1 step
    constexpr int KC = T::KC, MS = T::MS, WAVES = T::WAVES, NCB = T::NCB, NTH = T::NTH, NKB = T::NKB;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    float* const Xraw = lds;                                            // 2 raw chunks (LDS-DMA target)
    char* const Bop = reinterpret_cast<char*>(lds + 2 * T::RAW_FLOATS); // 2 transformed chunks (B operands)
    char* const gt = reinterpret_cast<char*>(lds);                      // gate tile (aliases the staging buffers)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L, dil = 1 << log2d;
    float* const trw = lds + T::MAIN_FLOATS + wave * T::TR_FLOATS;

    const int nblk = (L + 2 * dil - 1) >> (log2d + 1);     // blocks of 2d
    const int ntl = ((nblk << log2d) + 31) >> 5;           // tiles of 32 pairs per batch element
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int b = __builtin_amdgcn_readfirstlane(tile / ntl);
    // Order of the tiles inside a clip.  A tile reads its own 2 x 32 positions AND two halos of 32 (shifts -d and +2d): the
    // second half of the previous block of 2d and the first half of the next one -- the main columns of the tiles d/32 tile
    // numbers away.  d >= 64: the BLOCK index runs fastest (tile t -> block t % nblk, 32-pair slice t / nblk), so that the tiles
    // that share columns are neighbours in launch order = run at the same time on CUs of one XCD and meet in its L2 (4 MB:
    // one round of 32 tiles already moves 6 MB through it); linear order left every x element to be fetched twice.
    const int tic = tile % ntl;                             // tile number inside the clip
#ifdef WN_TILE_ORDER_LINEAR
    const int q0 = __builtin_amdgcn_readfirstlane(tic * 32);
#else
    const int q0 = __builtin_amdgcn_readfirstlane(log2d >= 6 ? ((tic % nblk) << log2d) + (tic / nblk) * 32 : tic * 32);
#endif
    const int q = q0 + l31;                                // first-half position number of this lane's column
    const int p = ((q >> log2d) << (log2d + 1)) + (q & (dil - 1));

    const float* __restrict__ xb = a.x_in + (size_t)b * C * L;
    unsigned long long* __restrict__ trc = a.trace ? a.trace + ((size_t)blockIdx.x * WAVES + wave) * 32 : nullptr;
    auto stamp = [&](int i) {
        if (trc) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0) trc[i] = t;
        }
    };
    stamp(0);

    // ---- operands of the correction k-blocks, fetched first (their latency hides behind the first staging wait)
    const int mt1[2] = {wave, C / 32 + wave};
    int mt2[1 + MS];
    mt2[0] = wave;
#pragma unroll
    for (int m = 0; m < MS; ++m) mt2[1 + m] = C / 32 + wave * MS + m;
    float av1[2][4], ab1[2], av2[1 + MS];
    {
        const float* Abt = a.Abt + (size_t)b * a.abt_bstride + step_row_off(a.step_idx, a.abt_tstride);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = mt1[m] * 32 + l31;
#pragma unroll
            for (int j = 0; j < 4; ++j) av1[m][j] = Abt[j * 2 * C + row];
            ab1[m] = a.bias1[row];
        }
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m) av2[m] = a.bias2[mt2[m] * 32 + l31];
    }
    // scaled splits (fp16 terms): this layer's two weight matrices were packed times a power of two each (wscale[0], [1]);
    // inv1 / inv2 undo weight x operand scale on the fp32 accumulators
    const float ws1 = P::SCALED ? a.wscale[0] : 1.f, ws2 = P::SCALED ? a.wscale[1] : 1.f;
    const float inv1 = 1.f / (ws1 * P::SX), inv2 = 1.f / (ws2 * P::SG);

    // ---- staging of raw x (wavenet_wino.hip: same three forms).  16-byte accesses need every row 16-byte aligned.
    const bool al16 = (L % 4 == 0) && ((((size_t)a.x_in | (size_t)a.x_out | (size_t)a.skip) & 15) == 0);
    const bool contig = al16 && log2d <= 4;
    const bool x4 = contig || (al16 && log2d >= 2);
    constexpr int RPW = KC / WAVES;                // channel rows per wave and chunk (as pairs of adjacent rows)
    __amdgpu_buffer_rsrc_t rXall = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, C * L * 4, 0x00020000);
    const int pbase = (q0 >> log2d) << (log2d + 1);          // first position of the tile's block (d < 32: a multiple of 64)
    const int ob = contig ? 32 + (p - pbase) - dil : l31;    // shift s of this lane's column sits at ob + s * os in a staged row
    const int os = contig ? dil : 32;
    // columns of the [res; skip] GEMM: tile n = 0 the first halves, n = 1 the partners (d >= 32: two runs of 32
    // consecutive positions); contiguous staging (d <= 16): positions [pbase, pbase+32) and [pbase+32, pbase+64), the
    // gate tile is written in that order
    const int x0l = p - pbase, x1l = x0l + dil;
    const int gcol0 = contig ? x0l : l31, gcol1 = contig ? x1l : 32 + l31;
    int voffA, voffB;
    if (contig) {
        const int pp = pbase - 32 + 4 * (lane & 31);
        voffA = ((unsigned)pp < (unsigned)L) ? (lhi * L + pp) * 4 : 0x7ffffff0;
        voffB = 0;
    } else if (x4) {
        const int s4 = (lane >> 3) & 3, qq = q0 + 4 * (lane & 7);
        const int pp = ((qq >> log2d) << (log2d + 1)) + (qq & (dil - 1)) + (s4 - 1) * dil;
        voffA = ((unsigned)pp < (unsigned)L) ? (lhi * L + pp) * 4 : 0x7ffffff0;
        voffB = 0;
    } else {
        voffA = (p + (lhi - 1) * dil) * 4;
        voffB = (p + (lhi + 1) * dil) * 4;
    }
    // The K loop visits the channel chunks in an order ROTATED by the tile number: workgroups that run in lockstep on the
    // CUs of an XCD then ask the L2 for different parts of the weight matrices at any moment (every tile streams all of
    // A1 / A2, 3.75 MB at C = 256; worth 3 k of 77 k GEMM1 cycles.  The stream is NOT what bounds GEMM1: with the fragments
    // served from L1 (BX6_ABL_A_HOT) GEMM1 moves by 2-3 %, profiles/r05_split_layer_ablations.txt)
#ifdef BX6_ABL_NO_ROT
    const int rot = 0;
#else
    // Keyed by the tile's number INSIDE its clip (consecutive workgroups of an XCD run consecutive tiles: xcd_remap), never by
    // the batch index: the summation order of a position must not depend on where its clip sits in the batch
    // (tests/test_full_size_gpu.py: equal clips give equal bits, a clip alone == the clip inside a batch of 16).
    const int rot = __builtin_amdgcn_readfirstlane((int)((unsigned)tic % NCB));
#endif
    auto chunk_of = [&](int cb) { const int c = cb + rot; return c >= NCB ? c - NCB : c; };
    constexpr int NPIECE = RPW / 2;                // staging requests (row pairs) per wave and chunk
    auto stage_piece = [&](int cbi, int i) {
        float* xs = Xraw + (cbi & 1) * T::RAW_FLOATS;
        const int cb = chunk_of(cbi);
        {
            const int cc = 2 * (wave + WAVES * i);
            if (x4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rXall, xs + cc * 128, 16, voffA, (cb * KC + cc) * L * 4, 0, BX6_NT);
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)(cb * KC + cc + h) * L), 0, L * 4, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, xs + (cc + h) * 128, 4, voffA, 0, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, xs + (cc + h) * 128 + 64, 4, voffB, 0, 0, 0);
                }
            }
        }
    };
    auto stage_dma = [&](int cbi) {
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) stage_piece(cbi, i);
    };

    // ---- transform pass: raw chunk c1 -> t_j = Winograd input transform, three bf16 terms each, in B-fragment order.
    // Item = (column, group of 4 channels, product pair): 12 LDS reads, 8 transforms, 8 splits, six 8-byte writes.
    auto transform = [&](int c1) {
        const float* xs = Xraw + (c1 & 1) * T::RAW_FLOATS;
        char* bo = Bop + (c1 & 1) * T::BOP_BYTES;
#pragma unroll
        for (int k = 0; k < T::TITEMS / NTH; ++k) {
            const int i = tid + NTH * k;
            const int gi = __builtin_amdgcn_readfirstlane(i >> 6);      // wave-uniform part of the item number
            const int g4 = (gi * 2 + lhi) % (KC / 4);
            const int jp = __builtin_amdgcn_readfirstlane((gi * 2) / (KC / 4));   // KC/4 even: both halves of a wave share jp
            float ta[4], tb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* xr = xs + (g4 * 4 + e) * 128 + ob;
                const float d1 = xr[os], d2 = xr[2 * os];
                if (jp == 0) {
                    const float d0 = xr[0];
                    ta[e] = d0 - d2; tb[e] = d1 + d2;
                } else {
                    const float d3 = xr[3 * os];
                    ta[e] = d2 - d1; tb[e] = d1 - d3;
                }
            }
            if (P::SCALED) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { ta[e] *= P::SX; tb[e] *= P::SX; }
            }
            v4 pa[NT], pb[NT];
            P::split4(ta, pa);
            P::split4(tb, pb);
            char* dst = bo + ((((g4 >> 1) * 4 + 2 * jp) * NT) * 32 + l31) * 16 + (g4 & 1) * 8;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                *reinterpret_cast<v4*>(dst + t * 512) = pa[t];
                *reinterpret_cast<v4*>(dst + (NT + t) * 512) = pb[t];
            }
        }
        // The residual x of this wave's 32 output rows passes through the staging buffers exactly once: the wave keeps a
        // copy in its (until the epilogue unused) transpose slot, row-major [row][64 columns] as the epilogue reads it --
        // shifts 0 and +d of a staged row sit at floats 32..95 in both staging forms.  No second read of x from memory.
        if (al16) {
            const int ch0 = chunk_of(c1) * KC;
            if (ch0 / 32 == wave) {
                const int r0 = ch0 % 32;
#pragma unroll
                for (int i = 0; i < KC / 4; ++i) {
                    const int rr = (lane >> 4) + 4 * i;
                    const bx6_f32x4 v = *reinterpret_cast<const bx6_f32x4*>(xs + rr * 128 + 32 + 4 * (lane & 15));
                    *reinterpret_cast<bx6_f32x4*>(trw + (r0 + rr) * 64 + 4 * (lane & 15)) = v;
                }
            }
        }
    };

    // ---- GEMM1: m_j[2C x 32] = G_j[2C x C] . t_j[C x 32], j = 0..3; this wave: row tiles `wave` and C/32 + wave
    bx_f32x16 acc[2][4];
    __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A1, 0, 2 * C * 4 * C * 2 * NT, 0x00020000);
    const int lane16 = lane * 16;
    auto load_a1 = [&](v8 (&dst)[2][NT], int kb, int j) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#ifdef BX6_ABL_A_HOT     // timing ablation only (wrong results): every step re-reads the same fragments -> they come from the CU's L1
                dst[m][t] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(rA1, lane16, ((mt1[m] * NKB * 4) * NT + t) * 1024 + 0 * (kb + j), 0));
#else
                dst[m][t] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(rA1, lane16, (((mt1[m] * NKB + kb) * 4 + j) * NT + t) * 1024, 0));
#endif
    };

    stage_dma(0);
    if (NCB > 1) stage_dma(1);
    // A fragments run PF (k-block, product) steps ahead of their use, in a ring of four register sets indexed by the
    // step number (compile time: the steps of a chunk are a multiple of four).  One step is 2 NPR MFMAs = 32 NPR matrix-pipe
    // cycles per wave: six products cover an L2 round trip with one step, three products need two.
    constexpr int PF = BX6_PF(NT);
    v8 a_ring[4][2][NT];
#pragma unroll
    for (int s0 = 0; s0 < PF; ++s0) load_a1(a_ring[s0], chunk_of(0) * (KC / 16) + (s0 >> 2), s0 & 3);
    constexpr int NAF = PF * NA;              // A fragment loads in flight behind anything older
    // chunk 0 has landed (this wave's part; the barrier makes it everyone's): all but the loads issued after it -- chunk 1
    // and the NAF A fragments.  hipcc does not make a barrier wait for LDS-DMA.
    if (NCB > 1) {
        if (x4) __builtin_amdgcn_s_waitcnt(0x0F70 | ((RPW / 2 + NAF) & 15) | (((RPW / 2 + NAF) >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * RPW + NAF) & 15) | (((2 * RPW + NAF) >> 4) << 14));
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70 | NAF);
    }
    __syncthreads();
    stamp(1);
    {   // correction k-block: k = 0 the step-embedding row (x the Winograd transform of the in-range indicator of the four
        // shifts), k = 1 of product 1 the conv bias (m1 enters both outputs of a pair with +1)
        const float v0 = ((unsigned)(p - dil) < (unsigned)L) ? 1.f : 0.f;
        const float v1 = ((unsigned)p < (unsigned)L) ? 1.f : 0.f;
        const float v2 = ((unsigned)(p + dil) < (unsigned)L) ? 1.f : 0.f;
        const float v3 = ((unsigned)(p + 2 * dil) < (unsigned)L) ? 1.f : 0.f;
        const float bi[4] = {v0 - v2, v1 + v2, v2 - v1, v1 - v3};
        bx_f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // (scaled splits: the indicator operand carries the activation scale, the A rows the weight scale: exact)
            const v8 bf = P::bvals(lhi ? 0.f : bi[j] * P::SX, (lhi || j != 1) ? 0.f : P::SX);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                v8 af[NT];
                P::rank2(av1[m][j] * ws1, j == 1 ? ab1[m] * ws1 : 0.f, lhi == 0, af);
                acc[m][j] = zero;
#pragma unroll
                for (int t = NT - 1; t >= 0; --t) acc[m][j] = P::mfma(af[t], bf, acc[m][j]);
            }
        }
    }
    transform(0);
    __builtin_amdgcn_s_waitcnt(0x0F70 | NAF);  // chunk 1 has landed too (younger: only the NAF A fragments)
    __syncthreads();                          // transformed chunk 0 visible, raw chunk 1 complete

    constexpr int SPC = (KC / 16) * 4;        // (k-block, product) steps per chunk
    // One (k-block, product) step: prefetch of the next step's A fragments, three B fragments from LDS, 12 MFMAs.
    // WITH_T: the transform pass of the NEXT chunk rides in this step's MFMA stream -- its ~80 VALU / LDS instructions are
    // dealt out between the 12 MFMAs (an MFMA holds the matrix pipe for 32 cycles; the wave issues other work meanwhile)
    // instead of standing in front of them with the matrix pipe idle (both waves of a SIMD reach it at the same time).
    auto do_step = [&](int cb, auto ST, auto WITH_T) {
        constexpr int st = decltype(ST)::value;
        constexpr bool with_t = decltype(WITH_T)::value;
        constexpr int it = st >> 2, j = st & 3;
        constexpr int sn = (st + PF) % SPC, itn = sn >> 2, jn = sn & 3;
        const int cbn = cb + (st + PF) / SPC;
        const char* tb = Bop + (cb & 1) * T::BOP_BYTES + lhi * (4 * NT * 512) + l31 * 16;
#if !defined(BX6_ABL_NO_STAGE) && defined(BX6_STAGE_SPREAD)
        // (measured and NOT the default: the staging requests of chunk cb + 2 dealt out over the first SPC - PF steps instead
        // of standing together behind the chunk barrier -- GEMM1 50.4 k -> 54.5 k cycles (f16x3), 76.0 k -> 83.0 k (bf16x6):
        // an LDS-DMA request in the MFMA stream holds the wave's issue longer than it costs next to the barrier)
        if (cb + 2 < NCB) {
            constexpr int NS = (SPC - PF) < 1 ? 1 : (SPC - PF);
#pragma unroll
            for (int i = 0; i < NPIECE; ++i)
                if (i * NS / NPIECE == st) stage_piece(cb + 2, i);
        }
#endif
        if (cbn < NCB) load_a1(a_ring[(st + PF) & 3], chunk_of(cbn) * (KC / 16) + itn, jn);   // (no load left in flight behind the last step)
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch PF whole steps ahead of its use
        v8 bq[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bq[t] = *reinterpret_cast<const v8*>(tb + ((2 * it * 4 + j) * NT + t) * 512);
#ifndef BX6_ABL_NO_TRANSFORM
        if (with_t) transform(cb + 1);
#endif
#pragma unroll
        for (int t = 0; t < NPR; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
                acc[m][j] = P::mfma(a_ring[st & 3][m][P::ia(t)], bq[P::ib(t)], acc[m][j]);
#ifndef BX6_ABL_NO_INTERLEAVE
        if (with_t) {
#pragma unroll
            for (int i = 0; i < 2 * NPR; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x302, BX6_T_PER_MFMA * 6 / NPR, 0);   // then VALU / DS read / DS write of the transform
            }
        }
#endif
    };
    auto do_steps_from1 = [&](int cb) {
        if constexpr (SPC == 4) {
            do_step(cb, std::integral_constant<int, 1>{}, std::false_type{});
            do_step(cb, std::integral_constant<int, 2>{}, std::false_type{});
            do_step(cb, std::integral_constant<int, 3>{}, std::false_type{});
        } else {
            do_step(cb, std::integral_constant<int, 1>{}, std::false_type{});
            do_step(cb, std::integral_constant<int, 2>{}, std::false_type{});
            do_step(cb, std::integral_constant<int, 3>{}, std::false_type{});
            do_step(cb, std::integral_constant<int, 4>{}, std::false_type{});
            do_step(cb, std::integral_constant<int, 5>{}, std::false_type{});
            do_step(cb, std::integral_constant<int, 6>{}, std::false_type{});
            do_step(cb, std::integral_constant<int, 7>{}, std::false_type{});
        }
    };
    static_assert((SPC == 4 || SPC == 8) && PF >= 1 && PF <= 3, "steps per chunk, prefetch distance");
    for (int cb = 0; cb < NCB; ++cb) {
#if !defined(BX6_ABL_NO_STAGE) && !defined(BX6_STAGE_SPREAD)
        if (cb + 2 < NCB) stage_dma(cb + 2);       // into the raw buffer chunk cb occupied (transformed an iteration ago)
#endif
        if (cb + 1 < NCB) do_step(cb, std::integral_constant<int, 0>{}, std::true_type{});
        else do_step(cb, std::integral_constant<int, 0>{}, std::false_type{});
        do_steps_from1(cb);
        stamp(8 + 2 * cb);
        // transformed chunk cb+1 visible after the barrier; the LDS-DMA of chunk cb+2 must have landed too (hipcc does not
        // count LDS-DMA among the accesses a barrier waits for): the only younger loads are the NAF A fragments of the next steps
#ifndef BX6_ABL_NO_DMA_WAIT     // (timing ablations, wrong results: no wait for the staged chunk / no chunk barrier / no staging)
        __builtin_amdgcn_s_waitcnt(0x0F70 | NAF);
#endif
#ifndef BX6_ABL_NO_BARRIER
        __syncthreads();
#endif
        stamp(9 + 2 * cb);
    }
    stamp(2);

    // ---- x and running-skip tiles of this wave's output rows, requested before the gate stage (row-major 16-byte pieces:
    // lane = (row lane>>4 of a group of four rows, column quad lane&15; quads 0..7 = column tile 0, 8..15 = tile 1)
    const int L4 = L * 4;
    const bool first = a.first_layer, last = a.last_layer;
    __amdgpu_buffer_rsrc_t rXo = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x_out + (size_t)b * C * L), 0, C * L * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rSk = __builtin_amdgcn_make_buffer_rsrc((void*)(a.skip + (size_t)b * S * L), 0, S * L * 4, 0x00020000);
    const int p0 = ((q0 >> log2d) << (log2d + 1)) + (q0 & (dil - 1));   // position of column 0 of tile 0 (d >= 32: q0 % 32 == 0 keeps it a multiple of 32)
    int voff4;
    {
        const int n4 = (lane & 15) >> 3;
        const int pos4 = (contig ? pbase + 32 * n4 : p0 + n4 * dil) + 4 * (lane & 7);
        voff4 = (pos4 < L) ? ((lane >> 4) * L + pos4) * 4 : 0x7ffffff0;
    }
    // (S = 2C: three row tiles per wave would take 96 registers here -- those instances fetch each tile in the epilogue;
    // they run two workgroups per CU, which cover each other's waits)
    constexpr bool PRELOAD = MS == 1;
    // a tile that is not read (the first layer's skip, the last layer's x) asks for an offset past the buffer: reads 0
    const int voff4s = first ? 0x7ffffff0 : voff4;
    auto load_pre = [&](int m, bx6_f32x4 (&dst)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (m == 0) dst[i] = *reinterpret_cast<const bx6_f32x4*>(trw + ((lane >> 4) + 4 * i) * 64 + 4 * (lane & 15));
            else dst[i] = bx6_load_f4(rSk, voff4s, ((wave * MS + (m - 1)) * 32 + 4 * i) * L4);
        }
    };
    bx6_f32x4 pre[PRELOAD ? 1 + MS : 1][8];
    if (PRELOAD && al16) {
#pragma unroll
        for (int m = 1; m < 1 + MS; ++m) load_pre(m, pre[PRELOAD ? m : 0]);
    }

    // ---- gate: g = tanh(H_t (+mel_t)) * sigmoid(H_s (+mel_s)) for both outputs of every pair, split -> LDS gate tile
    const float* melb = (EXTRA && a.melc) ? a.melc + (size_t)(a.mel_bstride ? b : 0) * 2 * C * L : nullptr;
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
        float g0[4], g1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int r = qd * 4 + e;
            const int ch = wave * 32 + e + 8 * qd + 4 * lhi;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int pos = p + n * dil;
                float ht, hs;
                if (n == 0) {
                    ht = (acc[0][0][r] + acc[0][1][r]) + acc[0][2][r];
                    hs = (acc[1][0][r] + acc[1][1][r]) + acc[1][2][r];
                } else {
                    ht = (acc[0][1][r] - acc[0][2][r]) - acc[0][3][r];
                    hs = (acc[1][1][r] - acc[1][2][r]) - acc[1][3][r];
                }
                if (P::SCALED) { ht *= inv1; hs *= inv1; }       // undo the operand scales (powers of two: exact)
                if (EXTRA && melb && pos < L) {
                    ht += melb[(size_t)ch * L + pos];
                    hs += melb[(size_t)(C + ch) * L + pos];
                }
                const float g = bx6_gate(ht, hs) * P::SG;
                if (n == 0) g0[e] = g; else g1[e] = g;
            }
        }
        v4 s0[NT], s1[NT];
        P::split4(g0, s0);
        P::split4(g1, s1);
        char* dst = gt + ((wave * 4 + qd) * NT * 64) * 16 + lhi * 8;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            *reinterpret_cast<v4*>(dst + (t * 64 + gcol0) * 16) = s0[t];
            *reinterpret_cast<v4*>(dst + (t * 64 + gcol1) * 16) = s1[t];
        }
    }
    stamp(3);

    // ---- GEMM2: [res; skip][(C+S) x 64] = [Wr; Ws][(C+S) x C] . g[C x 64] (+ bias k-block); this wave: res tile `wave`
    // and its skip tiles, both column tiles
    __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A2, 0, (C + S) * C * 2 * NT, 0x00020000);
    v8 c_cur[1 + MS][NT], c_nxt[1 + MS][NT];
    auto load_a2 = [&](v8 (&dst)[1 + MS][NT], int kb) {
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                dst[m][t] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(rA2, lane16, ((mt2[m] * NKB + kb) * NT + t) * 1024, 0));
    };
    const int rot2 = rot * (KC / 16);          // the same rotation of the k-block order in GEMM2
    auto kblock_of = [&](int kb) { const int k = kb + rot2; return k >= NKB ? k - NKB : k; };
    load_a2(c_cur, kblock_of(0));
    bx_f32x16 acc2[1 + MS][2];
    {
        const v8 bf = P::bvals(lhi ? 0.f : P::SG, 0.f);      // a row of ones, at the gate operand's scale
        bx_f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m) {
            v8 af[NT];
            P::rank2(av2[m] * ws2, 0.f, lhi == 0, af);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc2[m][n] = zero;
#pragma unroll
                for (int t = NT - 1; t >= 0; --t) acc2[m][n] = P::mfma(af[t], bf, acc2[m][n]);
            }
        }
    }
    __syncthreads();   // gate tile complete
    stamp(4);
    const char* gb = gt + lhi * (NT * 64 * 16) + l31 * 16;
#pragma unroll 2
    for (int kb = 0; kb < NKB; ++kb) {
        const int kbn = (kb + 1 < NKB) ? kb + 1 : kb;
        load_a2(c_nxt, kblock_of(kbn));
        __builtin_amdgcn_sched_barrier(0);
        const char* gk = gb + kblock_of(kb) * (2 * NT * 64 * 16);
        v8 bq[2][NT];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int t = 0; t < NT; ++t) bq[n][t] = *reinterpret_cast<const v8*>(gk + (t * 64 + n * 32) * 16);
#pragma unroll
        for (int t = 0; t < NPR; ++t)
#pragma unroll
            for (int m = 0; m < 1 + MS; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc2[m][n] = P::mfma(c_cur[m][P::ia(t)], bq[n][P::ib(t)], acc2[m][n]);
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m)
#pragma unroll
            for (int t = 0; t < NT; ++t) c_cur[m][t] = c_nxt[m][t];
    }
    stamp(5);

    // ---- epilogue
    const float rs = 0.70710678118654752440f;
    if (al16) {
        // row tile by row tile through this wave's own LDS slot (no other wave touches it: no barrier), out as 16-byte rows
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m) {
            if (m == 0 && last) continue;   // the last layer's residual output feeds nothing (`wavenet.py:165`)
            if (!PRELOAD || m == 0) load_pre(m, pre[0]);      // (m = 0: the x rows out of the wave's own slot, before it is overwritten)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) trw[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 64 + n * 32 + l31] = P::SCALED ? acc2[m][n][r] * inv2 : acc2[m][n][r];
            // one wave, LDS operations of a wave execute in order: only the compiler has to be kept from moving the reads
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int row0 = (m == 0) ? wave * 32 : (wave * MS + (m - 1)) * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bx6_f32x4 v = *reinterpret_cast<const bx6_f32x4*>(trw + ((lane >> 4) + 4 * i) * 64 + 4 * (lane & 15));
                bx6_f32x4 o = pre[PRELOAD ? m : 0][i] + v;
                if (m == 0) o = o * rs;
                bx6_store_f4(o, m == 0 ? rXo : rSk, voff4, (row0 + 4 * i) * L4);
                // gfx950 hazard, found by the bit-for-bit determinism test: a VALU write (here the next v_pk_add / v_pk_mul) to
                // the data registers of a 16-byte buffer store in the slot right behind it changes what the store writes
                // for the last lanes of each 16-lane group (their last dword) -- hipcc's hazard recognizer leaves the wait
                // state out when the store carries an SGPR offset.  One wait state removes it (measured: 0 -> wrong in
                // 2 of 3 runs, 1 / 2 / 4 / 16 -> never); two are kept.
                asm volatile("s_nop 1" ::: "memory");
            }
            asm volatile("" ::: "memory");   // the next row tile overwrites the slot
        }
    } else {
        // rows not 16-byte aligned: per-lane dwords straight from the accumulator layout
        int voffn[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int pos = contig ? pbase + 32 * n + l31 : p + n * dil;
            voffn[n] = (pos < L) ? (4 * lhi * L + pos) * 4 : 0x7ffffff0;
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (!last) {
                const int s0 = (wave * 32) * L4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int so = s0 + ((r & 3) + 8 * (r >> 2)) * L4;
                    const float x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rXall, voffn[n], so, 0));
                    const float v = (x + (P::SCALED ? acc2[0][n][r] * inv2 : acc2[0][n][r])) * rs;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rXo, voffn[n], so, 0);
                }
            }
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                const int s0 = ((wave * MS + m) * 32) * L4;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int so = s0 + ((r & 3) + 8 * (r >> 2)) * L4;
                    const float old = first ? 0.f : __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rSk, voffn[n], so, 0));
                    const float v = old + (P::SCALED ? acc2[1 + m][n][r] * inv2 : acc2[1 + m][n][r]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rSk, voffn[n], so, 0);
                }
            }
        }
    }
    stamp(6);
    if (trc) {
        __builtin_amdgcn_s_waitcnt(0);   // everything (stores included) retired
        stamp(7);
    }
}

template <typename P, int C, int S>
static int launch_bx6_t(const WnLayerArgs& a, int log2d, hipStream_t s) {
    ProfileScope ps(P::NT == 3 ? "wn_layer_bx6" : "wn_layer_f16x3", s);
    using T = Bx6Tile<C, S, P::NT>;
    const int dil = 1 << log2d;
    const int nblk = (a.L + 2 * dil - 1) / (2 * dil);
    const int ntl = (nblk * dil + 31) / 32;
    const int ntiles = a.B * ntl;
    static const bool trace = std::getenv("DWS_BX6_TRACE") != nullptr;
    if (trace && !a.melc) {
        wino_trace_launch(ntiles, T::WAVES, a, s, [&](const WnLayerArgs& at) {
            hipLaunchKernelGGL((wn_layer_bx6_kernel<P, C, S, false>), dim3(ntiles), dim3(T::NTH), 0, s, at, log2d);
        }, P::NT == 3 ? "bx6" : "f16x3");
        return DWS_OK;
    }
    if (a.melc) hipLaunchKernelGGL((wn_layer_bx6_kernel<P, C, S, true>), dim3(ntiles), dim3(T::NTH), 0, s, a, log2d);
    else hipLaunchKernelGGL((wn_layer_bx6_kernel<P, C, S, false>), dim3(ntiles), dim3(T::NTH), 0, s, a, log2d);
    return DWS_OK;
}

bool wn_layer_bx6_supported(int C, int S) {
    return (C == 64 && S == 64) || (C == 128 && S == 128) || (C == 128 && S == 256) || (C == 256 && S == 256);
}

template <typename P>
static int launch_bx6_p(int C, int S, const WnLayerArgs& a, int log2d, hipStream_t s) {
    if (C == 64 && S == 64) return launch_bx6_t<P, 64, 64>(a, log2d, s);
    if (C == 128 && S == 128) return launch_bx6_t<P, 128, 128>(a, log2d, s);
    if (C == 128 && S == 256) return launch_bx6_t<P, 128, 256>(a, log2d, s);
    if (C == 256 && S == 256) return launch_bx6_t<P, 256, 256>(a, log2d, s);
    return set_error(DWS_ERR_UNSUPPORTED, "wn_layer_bx6: (C=%d,S=%d) not instantiated", C, S);
}

int launch_wn_layer_bx6(int C, int S, const WnLayerArgs& a, int split, hipStream_t s) {
    int log2d = 0;
    while ((1 << log2d) < a.dilation) ++log2d;
    DWS_CHECK((1 << log2d) == a.dilation, DWS_ERR_UNSUPPORTED, "wn_layer_bx6: dilation %d is not a power of two", a.dilation);
    DWS_CHECK((int64_t)a.L + 4 * (int64_t)a.dilation < ((int64_t)1 << 28), DWS_ERR_UNSUPPORTED, "wn_layer_bx6: L too large");
    DWS_CHECK((int64_t)(C > S ? C : S) * a.L * 4 < ((int64_t)1 << 31), DWS_ERR_UNSUPPORTED,
              "wn_layer_bx6: %d channels x L=%d exceed a 2 GiB tensor per clip", C > S ? C : S, a.L);
    DWS_CHECK(a.hsave == nullptr, DWS_ERR_UNSUPPORTED, "wn_layer_bx6: the training forward runs with precision=f32");
    if (split == WN_SPLIT_F16X3) {
        DWS_CHECK(a.wscale != nullptr, DWS_ERR_INVALID, "wn_layer_f16x3: no weight scales");
        return launch_bx6_p<SplitF16x2>(C, S, a, log2d, s);
    }
    return launch_bx6_p<SplitBf16x3>(C, S, a, log2d, s);
}

// ---- the arithmetic alone, for the GEMM-level accuracy test: C[M][N] = A[M][K] . B[K][N], fp32 in and out, every
// operand split in registers, one wave per 32 x 32 output tile (tests/test_bf16x6_gpu.py; not a fast GEMM).
// Scaled splits: A enters times sa, B times sb (powers of two from the caller), the result leaves times 1/(sa sb).
template <typename P>
__global__ __launch_bounds__(64) void gemm_bx6_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ Cm,
                                                      int M, int N, int K, float sa, float sb) {
    using v8 = typename P::v8;
    const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
    const int mt = blockIdx.y, nt = blockIdx.x;
    bx_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int kb = 0; kb < K / 16; ++kb) {
        v8 af[P::NT], bf[P::NT];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = kb * 16 + 8 * lhi + e;
            P::split1(A[(size_t)(mt * 32 + l31) * K + k] * sa, af, e);
            P::split1(B[(size_t)k * N + nt * 32 + l31] * sb, bf, e);
        }
#pragma unroll
        for (int t = 0; t < P::NP; ++t) acc = P::mfma(af[P::ia(t)], bf[P::ib(t)], acc);
    }
    const float inv = 1.f / (sa * sb);
#pragma unroll
    for (int r = 0; r < 16; ++r)
        Cm[(size_t)(mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * N + nt * 32 + l31] = acc[r] * inv;
}

int launch_gemm_bx6(const float* A, const float* B, float* Cm, int M, int N, int K, int split, float sa, float sb, hipStream_t s) {
    DWS_CHECK(M > 0 && N > 0 && K > 0 && M % 32 == 0 && N % 32 == 0 && K % 16 == 0, DWS_ERR_UNSUPPORTED,
              "gemm_bf16x6 / gemm_f16x3: M=%d N=%d must be multiples of 32, K=%d of 16", M, N, K);
    if (split == WN_SPLIT_F16X3) {
        int ea = 0, eb = 0;
        DWS_CHECK(sa > 0.f && sb > 0.f && frexpf(sa, &ea) == 0.5f && frexpf(sb, &eb) == 0.5f, DWS_ERR_INVALID,
                  "gemm_f16x3: the operand scales must be powers of two (%g, %g)", (double)sa, (double)sb);
        hipLaunchKernelGGL(gemm_bx6_kernel<SplitF16x2>, dim3(N / 32, M / 32), dim3(64), 0, s, A, B, Cm, M, N, K, sa, sb);
    } else {
        hipLaunchKernelGGL(gemm_bx6_kernel<SplitBf16x3>, dim3(N / 32, M / 32), dim3(64), 0, s, A, B, Cm, M, N, K, 1.f, 1.f);
    }
    return DWS_OK;
}

}  // namespace dws

extern "C" int dws_gemm_bf16x6(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, void* stream) {
    DWS_CHECK(A && B && C, DWS_ERR_INVALID, "dws_gemm_bf16x6: null argument");
    DWS_CHECK(M <= (1 << 20) && N <= (1 << 20) && K <= (1 << 20), DWS_ERR_UNSUPPORTED, "dws_gemm_bf16x6: dimension too large");
    return dws::launch_gemm_bx6(A, B, C, (int)M, (int)N, (int)K, dws::WN_SPLIT_BF16X6, 1.f, 1.f, (hipStream_t)stream);
}

extern "C" int dws_gemm_f16x3(const float* A, const float* B, float* C, int64_t M, int64_t N, int64_t K, float scale_a, float scale_b,
                              void* stream) {
    DWS_CHECK(A && B && C, DWS_ERR_INVALID, "dws_gemm_f16x3: null argument");
    DWS_CHECK(M <= (1 << 20) && N <= (1 << 20) && K <= (1 << 20), DWS_ERR_UNSUPPORTED, "dws_gemm_f16x3: dimension too large");
    return dws::launch_gemm_bx6(A, B, C, (int)M, (int)N, (int)K, dws::WN_SPLIT_F16X3, scale_a, scale_b, (hipStream_t)stream);
}
