// WaveNet fused residual layer on the bf16 matrix cores with a 3-term split ("bf16x3"):
//   x = x_hi + x_lo, W = W_hi + W_lo (each part bf16),  W x ~= W_hi x_hi + W_hi x_lo + W_lo x_hi
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped W_lo x_lo term and the bf16
// rounding of the lo parts are each <= 2^-18 relative, i.e. ~1e-5 worst case per product --
// two orders of magnitude inside the 1e-3 parity bound (measured: see tests/test_wavenet_gpu.py),
// at 16/3 = 5.3x the matrix rate of the exact-f32 MFMA.  Opt-in: precision = "bf16x3".
//
// Same structure as wn_layer_mfma_kernel (wavenet_kernels.hip) -- LDS-DMA staging of the raw x
// window (16-channel chunks: a k-block of 16 is one tap), the step embedding AND the conv bias as extra K rows,
// gate in registers, [res; skip] GEMMs from the LDS gate tile -- but with a 128-position tile and 8 waves so every A
// fragment (streamed from L2) feeds 4 position tiles.  The fp32 window is split into (hi, lo) bf16 one chunk AHEAD of
// the MFMAs, a 4-channel half item per thread and k-block, and stored in MFMA B-fragment order ([k-octet][position]
// 16-byte items), so a B fragment is a single conflict-free ds_read_b128.  Epilogue: the running skip tile is
// requested before the gate stage, the skip rows are multiplied and stored first (transposed piecewise through the LDS
// beside the gate tile), the res rows last; all tile streams carry the nontemporal hint.
#include <cstdlib>

#include "wavenet.h"

namespace dws {

// The epilogue's tile-sized streams (running skip in/out, residual x in, x' out: each touched once per layer and far larger
// than the caches together) go with the nontemporal hint: 719 vs 737 us per launch on the same box.
typedef float bx3_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void bx3_nt_store(float4* p, const float4& v) {
    bx3_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<bx3_f4*>(p));
}
__device__ __forceinline__ float4 bx3_nt_load(const float4* p) {
    const bx3_f4 t = __builtin_nontemporal_load(reinterpret_cast<const bx3_f4*>(p));
    return make_float4(t.x, t.y, t.z, t.w);
}
#define BX3_STORE(p, v) bx3_nt_store(p, v)
#define BX3_LOAD(p) bx3_nt_load(p)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 buf_load_bf8(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// tanh(t) * sigmoid(s) with a single quotient (same as wavenet_kernels.hip fast_gate)
__device__ __forceinline__ float fast_gate3(float t, float s) {
    const float tc = __builtin_amdgcn_fmed3f(t, -30.f, 30.f);
    const float e2 = __builtin_amdgcn_exp2f(tc * 2.8853900817779268f);      // e^{2t}
    const float en = __builtin_amdgcn_exp2f(s * -1.4426950408889634f);      // e^{-s}
    return (e2 - 1.f) * __builtin_amdgcn_rcpf((e2 + 1.f) * (1.f + en));
}

template <int C, int S, int P_, int WAVES_>
struct Bx3Tile {
    static constexpr int P = P_;
    static constexpr int WAVES = WAVES_;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int WM = (C / 32 >= WAVES) ? WAVES : C / 32;
    static constexpr int WN = WAVES / WM;
    static constexpr int NT = (P / 32) / WN;
    static constexpr int MP = C / 32 / WM;   // (tanh, sigmoid) tile pairs per wave
    static constexpr int MR = C / 32 / WM;
    static constexpr int MS = S / 32 / WM;
    static constexpr int KC = WN_BX3_KC;
    static constexpr int NCB = C / KC;
    static constexpr int NKBC = 3 * KC / 16;             // k-blocks per chunk
    static constexpr int OCT = 3 * KC / 8;               // k-octets per chunk
    // LDS map (float units): F = 2 fp32 DMA buffers; X = 2 buffers of (hi, lo) bf16x8 items [octet][pos]; IND = indicator
    // items (2 octets).  The gate tile (hi, lo items [C/8][pos]) aliases everything from offset 0 once GEMM1 is done.
    static constexpr int FB_FLOATS = 3 * KC * P;
    static constexpr int XB_FLOATS = 2 * OCT * P * 4;
    static constexpr int F_FLOATS = 2 * FB_FLOATS;
    static constexpr int X_FLOATS = 2 * XB_FLOATS;
    static constexpr int IND_FLOATS = 2 * P * 4;
    static constexpr int G_FLOATS = 2 * (C / 8) * P * 4;
    static constexpr int GEMM1_FLOATS = F_FLOATS + X_FLOATS + IND_FLOATS;
    // epilogue: the skip rows are transposed through the LDS beside the gate tile in pieces of PIECE_ROWS rows
    // (largest power of two that fits in 160 KB), the res rows through [C][P] floats from offset 0 (gate tile dead)
    static constexpr int FREE_ROWS = (40960 - G_FLOATS) / P;
    static constexpr int PIECE_ROWS0 = FREE_ROWS >= 256 ? 256 : FREE_ROWS >= 128 ? 128 : FREE_ROWS >= 64 ? 64 : 32;
    static constexpr int PIECE_ROWS = PIECE_ROWS0 < S ? PIECE_ROWS0 : S;
    static constexpr int EPI_FLOATS = (G_FLOATS + PIECE_ROWS * P) > C * P ? (G_FLOATS + PIECE_ROWS * P) : C * P;
    static constexpr int LDS_FLOATS0 = GEMM1_FLOATS > G_FLOATS ? GEMM1_FLOATS : G_FLOATS;
    static constexpr int LDS_FLOATS = LDS_FLOATS0 > EPI_FLOATS ? LDS_FLOATS0 : EPI_FLOATS;
    static_assert(LDS_FLOATS <= 40960 && FREE_ROWS >= 32 && S % PIECE_ROWS == 0 && PIECE_ROWS % 32 == 0, "LDS budget");
    // one convert slice per k-block: 4-channel half items, one per thread
    static_assert(2 * OCT * P == NKBC * THREADS, "convert slices");
    static_assert(WN * NT * 32 == P && C % (32 * WM) == 0 && S % (32 * WM) == 0 && C % KC == 0, "tiling");
};

// acc[m][n] += (A_hi + A_lo)[m] . (x_hi + x_lo)[n] minus the lo*lo term, for one k-block of 16.
// B items: bf16x8 at item index (octet*P + col), octet = 2*kb + (lane>>5).
template <int MT, int NT, int P>
__device__ __forceinline__ void kblock(f32x16 (&acc)[MT][NT], const bf16x8 (&ahi)[MT], const bf16x8 (&alo)[MT],
                                       const u32x4* __restrict__ xhi, const u32x4* __restrict__ xlo, int item0) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const bf16x8 bhi = __builtin_bit_cast(bf16x8, xhi[item0 + n * 32]);
        const bf16x8 blo = __builtin_bit_cast(bf16x8, xlo[item0 + n * 32]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[m], bhi, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[m], blo, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[m], bhi, acc[m][n], 0, 0, 0);
        }
    }
}

template <int C, int S, int PP, int WV, bool VEC>
__global__ __launch_bounds__(64 * WV) void wn_layer_bf16x3_kernel(WnLayerArgs a) {
    using T = Bx3Tile<C, S, PP, WV>;
    constexpr int THREADS = T::THREADS;
    constexpr int P = T::P, KC = T::KC, NT = T::NT, MP = T::MP, MR = T::MR, MS = T::MS, OCT = T::OCT;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % T::WM, wn = wave / T::WM;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int ntl = (a.L + P - 1) / P;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int b = tile / ntl;
    const int l0 = (tile % ntl) * P;
    const int L = a.L, dil = a.dilation;
    const float* __restrict__ xb = a.x_in + (size_t)b * C * L;

    u32x4* indi = reinterpret_cast<u32x4*>(lds + T::F_FLOATS + T::X_FLOATS);

    // ---- staging by LDS-DMA: 3*KC rows x P floats per chunk, P/64 256-byte pieces per row, all through ONE
    // descriptor for this batch element (row = wave-uniform soffset; per-row descriptors as in
    // wn_layer_mfma_kernel cost 4 SGPRs each and spill here).  A tap position outside [0, L) then reads a
    // neighbouring row (or 0 beyond the tensor) -- it is zeroed by the mask of the convert pass below.
    constexpr int PIECES = 3 * KC * (P / 64);
    constexpr int PPW = PIECES / T::WAVES;
    static_assert(PIECES % T::WAVES == 0, "DMA pieces per wave");
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, C * L * 4, 0x00020000);
    auto stage_dma = [&](int cb, int buf) {
        float* xs = lds + buf * T::FB_FLOATS;
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + T::WAVES * i;
            const int row = piece / (P / 64), half = piece % (P / 64);
            const int tap = row / KC, cc = row % KC;
            const int c = cb * KC + cc;
            const int voff = (l0 + half * 64 + lane + (tap - 1) * dil) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rX, xs + row * P + half * 64, 4, voff, c * L * 4, 0, 0);
        }
    };
    // fp32 -> (hi, lo) bf16 split, laid out as the MFMA B fragment wants it: 8 consecutive k per 16-byte item, items of
    // one octet contiguous over positions -> conflict-free ds_read_b128.  The split runs ONE CHUNK AHEAD of the MFMAs
    // and in slices, one per k-block of the chunk being multiplied (slice `it` = tap `it`: KC = 16 makes a k-block one
    // tap), so its VALU/LDS work issues between the MFMAs instead of in a phase of its own.  A thread converts a
    // 4-channel half item per slice.  The conv's zero padding is applied here.
    const int cv_pos = tid % P, cv_oh = tid / P;
    static_assert(KC == 16 && THREADS / P == 4, "slice == tap");
    auto convert_slice = [&](int fbuf, int xbuf, int it) {
        const float* xs = lds + fbuf * T::FB_FLOATS;
        unsigned long long* xh = reinterpret_cast<unsigned long long*>(lds + T::F_FLOATS + xbuf * T::XB_FLOATS);
        unsigned long long* xl = xh + OCT * P * 2;
        const int oct = (cv_oh >> 1) + 2 * it, h = cv_oh & 1;
        const float mask = ((unsigned)(l0 + cv_pos + (it - 1) * dil) < (unsigned)L) ? 1.f : 0.f;
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 h4, l4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x = xs[(oct * 8 + 4 * h + e) * P + cv_pos] * mask;
            const __bf16 hh = (__bf16)x;
            h4[e] = hh;
            l4[e] = (__bf16)(x - (float)hh);
        }
        xh[(oct * P + cv_pos) * 2 + h] = __builtin_bit_cast(unsigned long long, h4);
        xl[(oct * P + cv_pos) * 2 + h] = __builtin_bit_cast(unsigned long long, l4);
    };

    // indicator items: octet 0 = {tap0, tap1, tap2 in range, 1 (bias row), 0...}, octet 1 = 0 (exact in bf16)
    for (int i = tid; i < 2 * P; i += THREADS) {
        const int oct = i / P, col = i % P;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pos = l0 + col + (e - 1) * dil;
            v[e] = (__bf16)((oct == 0 && (e == 3 || (e < 3 && (unsigned)pos < (unsigned)L))) ? 1.f : 0.f);
        }
        indi[i] = __builtin_bit_cast(u32x4, v);
    }

    f32x16 acc[2 * MP][NT];
#pragma unroll
    for (int m = 0; m < 2 * MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    constexpr int NKB1 = 3 * C / 16;  // k-blocks of GEMM1
    __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A1, 0, 2 * C * 3 * C * 4, 0x00020000);
    const int lane16 = lane * 16;
    int mt1[2 * MP];
#pragma unroll
    for (int m = 0; m < 2 * MP; ++m) mt1[m] = (m < MP) ? (wm * MP + m) : (C / 32 + wm * MP + (m - MP));
    const int col0 = wn * NT * 32 + l31;

    stage_dma(0, 0);
    if (T::NCB > 1) stage_dma(1, 1);
    bf16x8 ahi[2 * MP], alo[2 * MP], nhi[2 * MP], nlo[2 * MP];
#pragma unroll
    for (int m = 0; m < 2 * MP; ++m) {
        ahi[m] = buf_load_bf8(rA1, lane16, (mt1[m] * NKB1) * 2048);
        alo[m] = buf_load_bf8(rA1, lane16, (mt1[m] * NKB1) * 2048 + 1024);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a barrier does not wait for this wave's LDS-DMA by itself
    __syncthreads();                           // DMA(0), DMA(1) landed
#pragma unroll
    for (int it = 0; it < T::NKBC; ++it) convert_slice(0, 0, it);

    for (int cb = 0; cb < T::NCB; ++cb) {
        // X(cb) complete, DMA(cb+1) landed; every wave is done with X(cb-1) and with the fp32 buffer of chunk cb
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): a barrier does not wait for this wave's LDS-DMA by itself
        __syncthreads();
        if (cb + 2 < T::NCB) stage_dma(cb + 2, cb & 1);
        const u32x4* xhi = reinterpret_cast<const u32x4*>(lds + T::F_FLOATS + (cb & 1) * T::XB_FLOATS);
        const u32x4* xlo = xhi + OCT * P;
#pragma unroll
        for (int it = 0; it < T::NKBC; ++it) {
            const int kb = cb * T::NKBC + it;
            const int kbn = (kb + 1 < NKB1) ? kb + 1 : kb;
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m) {
                nhi[m] = buf_load_bf8(rA1, lane16, (mt1[m] * NKB1 + kbn) * 2048);
                nlo[m] = buf_load_bf8(rA1, lane16, (mt1[m] * NKB1 + kbn) * 2048 + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the A prefetch one k-block ahead (see wavenet_kernels.hip)
            if (cb + 1 < T::NCB) convert_slice((cb + 1) & 1, (cb + 1) & 1, it);
            kblock<2 * MP, NT, P>(acc, ahi, alo, xhi, xlo, (it * 2 + lhi) * P + col0);
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m) { ahi[m] = nhi[m]; alo[m] = nlo[m]; }
        }
    }
    // step-embedding correction rows (bf16 hi/lo fragments from wn_bias_tap_bf16_kernel); the
    // indicator operand is exact in bf16, so only (hi + lo) x ind is needed
    {
        const u32x4* Abt = reinterpret_cast<const u32x4*>(a.Abt + (size_t)b * a.abt_bstride + step_row_off(a.step_idx, a.abt_tstride));
#pragma unroll
        for (int m = 0; m < 2 * MP; ++m) {
            ahi[m] = __builtin_bit_cast(bf16x8, Abt[(mt1[m] * 2 + 0) * 64 + lane]);
            alo[m] = __builtin_bit_cast(bf16x8, Abt[(mt1[m] * 2 + 1) * 64 + lane]);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const bf16x8 bi = __builtin_bit_cast(bf16x8, indi[lhi * P + col0 + n * 32]);
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[m], bi, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[m], bi, acc[m][n], 0, 0, 0);
            }
        }
    }
    // row-major float4 view of the tile for the vector epilogue; the running skip tile is fetched here so that its HBM
    // latency passes under the gate stage (which issues no global loads: the conv bias came in with the correction rows)
    constexpr int F4_ROW = P / 4;                 // float4 per tile row
    constexpr int ROWS_PASS = THREADS / F4_ROW;   // rows covered by one pass of the workgroup
    const int f4 = tid % F4_ROW, rsub = tid / F4_ROW;
    const int pos4 = l0 + f4 * 4;
    const bool ok4 = pos4 < L;                    // VEC: L % 4 == 0, a float4 is entirely inside or outside
    const int pos4c = ok4 ? pos4 : 0;
    float4 v4s[VEC ? S / ROWS_PASS : 1];
    if constexpr (VEC) {
#pragma unroll
        for (int i = 0; i < S / ROWS_PASS; ++i)
            v4s[i] = a.first_layer ? make_float4(0.f, 0.f, 0.f, 0.f)
                                   : BX3_LOAD(reinterpret_cast<const float4*>(a.skip + ((size_t)b * S + i * ROWS_PASS + rsub) * L + pos4c));
    }
    __syncthreads();

    // ---- gate -> (hi, lo) bf16 items [C/8][P] in LDS (aliases the GEMM1 buffers).  A lane's four
    // consecutive accumulator registers are four consecutive channels: half of one 16-byte item.
    unsigned long long* ghi = reinterpret_cast<unsigned long long*>(lds);
    unsigned long long* glo = ghi + (C / 8) * P * 2;
    const float* melb = a.melc ? a.melc + (size_t)(a.mel_bstride ? b : 0) * 2 * C * L : nullptr;
#pragma unroll
    for (int m = 0; m < MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = col0 + n * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                bf16x4 h4, l4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = q * 4 + e;
                    const int ch = (wm * MP + m) * 32 + 8 * q + 4 * lhi + e;
                    float ht = acc[m][n][r];      // bias1 came in through the correction rows
                    float hs = acc[MP + m][n][r];
                    if (melb) {
                        const int pos = l0 + col;
                        if (pos < L) {
                            ht += melb[(size_t)ch * L + pos];
                            hs += melb[(size_t)(C + ch) * L + pos];
                        }
                    }
                    const float g = fast_gate3(ht, hs);
                    const __bf16 hh = (__bf16)g;
                    h4[e] = hh;
                    l4[e] = (__bf16)(g - (float)hh);
                }
                const int oct = (wm * MP + m) * 4 + q;
                ghi[(oct * P + col) * 2 + lhi] = __builtin_bit_cast(unsigned long long, h4);
                glo[(oct * P + col) * 2 + lhi] = __builtin_bit_cast(unsigned long long, l4);
            }
        }
    __syncthreads();

    // ---- GEMM2 + epilogue.
    constexpr int NKB2 = C / 16;
    __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A2, 0, (C + S) * C * 4, 0x00020000);
    const u32x4* g_hi = reinterpret_cast<const u32x4*>(lds);
    const u32x4* g_lo = g_hi + (C / 8) * P;
    const float rs = 0.70710678118654752440f;
    float* __restrict__ xo = a.x_out + (size_t)b * C * L;
    float* __restrict__ sk = a.skip + (size_t)b * S * L;
    const bool first = a.first_layer, last = a.last_layer;

    auto gemm2 = [&](auto& acc2, const auto& mt2) {
        constexpr int MT = sizeof(mt2) / sizeof(int);
        bf16x8 chi[MT], clo[MT], dhi[MT], dlo[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            chi[m] = buf_load_bf8(rA2, lane16, (mt2[m] * NKB2) * 2048);
            clo[m] = buf_load_bf8(rA2, lane16, (mt2[m] * NKB2) * 2048 + 1024);
        }
        for (int kb = 0; kb < NKB2; ++kb) {
            const int kbn = (kb + 1 < NKB2) ? kb + 1 : kb;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                dhi[m] = buf_load_bf8(rA2, lane16, (mt2[m] * NKB2 + kbn) * 2048);
                dlo[m] = buf_load_bf8(rA2, lane16, (mt2[m] * NKB2 + kbn) * 2048 + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            kblock<MT, NT, P>(acc2, chi, clo, g_hi, g_lo, (kb * 2 + lhi) * P + col0);
#pragma unroll
            for (int m = 0; m < MT; ++m) { chi[m] = dhi[m]; clo[m] = dlo[m]; }
        }
    };

    if constexpr (VEC) {
        // Vector path (L % 4 == 0, chosen at launch).  Per-lane dword loads/stores in the accumulator layout are
        // store-ISSUE bound, so outputs are transposed through LDS and moved as row-major dwordx4; and a CU sustains only
        // ~8 B/clk to/from HBM (131 KB each of x, x', skip in, skip out per tile), so that traffic is spread under the
        // MFMA phases instead of following them:
        //   skip tile loaded BEFORE the gate stage (v4s: 16 float4 per thread) -> lands under it;
        //   GEMM2 for the skip rows -> transposed through the LDS left beside the gate tile, piece by piece ->
        //   skip stores drain under GEMM2 for the res rows (whose residual x tile is loaded just before it);
        //   only the x' stores are left at the end of the tile.
        f32x16 accS[MS][NT];
        int mtS[MS];
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            mtS[m] = C / 32 + wm * MS + m;
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) accS[m][n][r] = 0.f;
        }
        gemm2(accS, mtS);
        constexpr int PR = T::PIECE_ROWS, NPIECE = S / PR;
        float* otp = lds + T::G_FLOATS;
#pragma unroll
        for (int p = 0; p < NPIECE; ++p) {
#pragma unroll
            for (int m = 0; m < MS; ++m) {
                if (((wm * MS + m) * 32) / PR == p) {
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int sc = (wm * MS + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                            otp[(sc - p * PR) * P + col0 + n * 32] = accS[m][n][r] + a.bias2[C + sc];
                        }
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < PR / ROWS_PASS; ++i) {
                const int rl = i * ROWS_PASS + rsub;
                const float4 v = *reinterpret_cast<const float4*>(otp + rl * P + f4 * 4);
                const float4 w = v4s[p * (PR / ROWS_PASS) + i];
                const float4 o = make_float4(w.x + v.x, w.y + v.y, w.z + v.z, w.w + v.w);
                if (ok4) BX3_STORE(reinterpret_cast<float4*>(sk + (size_t)(p * PR + rl) * L + pos4), o);
            }
            if (p + 1 < NPIECE) __syncthreads();
        }
        if (!last) {
            float4 x4[C / ROWS_PASS];
#pragma unroll
            for (int i = 0; i < C / ROWS_PASS; ++i)
                x4[i] = BX3_LOAD(reinterpret_cast<const float4*>(xb + (size_t)(i * ROWS_PASS + rsub) * L + pos4c));
            f32x16 accR[MR][NT];
            int mtR[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                mtR[m] = wm * MR + m;
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) accR[m][n][r] = 0.f;
            }
            gemm2(accR, mtR);
            __syncthreads();  // every wave is done with the gate tile
            float* ot = lds;  // [C][P] fp32 transpose buffer
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = (wm * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        ot[ch * P + col0 + n * 32] = accR[m][n][r] + a.bias2[ch];
                    }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < C / ROWS_PASS; ++i) {
                const int row = i * ROWS_PASS + rsub;
                const float4 v = *reinterpret_cast<const float4*>(ot + row * P + f4 * 4);
                const float4 o = make_float4((x4[i].x + v.x) * rs, (x4[i].y + v.y) * rs, (x4[i].z + v.z) * rs,
                                             (x4[i].w + v.w) * rs);
                if (ok4) BX3_STORE(reinterpret_cast<float4*>(xo + (size_t)row * L + pos4), o);
            }
        }
    } else {
    // Scalar path (L not a multiple of 4: rows are not 16-byte aligned): loads of the residual x /
    // running skip are issued before the GEMM that produces their partner, stores per lane.
    float xr[MR][NT][16];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int pos = l0 + col0 + n * 32;
        const int posc = pos < L ? pos : 0;
#pragma unroll
        for (int m = 0; m < MR; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (wm * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                xr[m][n][r] = xb[(size_t)ch * L + posc];
            }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!last) {
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            f32x16 acc2[1][NT];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[0][n][r] = 0.f;
            const int mt2[1] = {wm * MR + m};
            gemm2(acc2, mt2);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int pos = l0 + col0 + n * 32;
                const bool ok = pos < L;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = (wm * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (ok) xo[(size_t)ch * L + pos] = (xr[m][n][r] + (acc2[0][n][r] + a.bias2[ch])) * rs;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MS; ++m) {
        float sr[NT][16];
        if (!first) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int pos = l0 + col0 + n * 32;
                const int posc = pos < L ? pos : 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sc = (wm * MS + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    sr[n][r] = sk[(size_t)sc * L + posc];
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) sr[n][r] = 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc2[1][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[0][n][r] = 0.f;
        const int mt2[1] = {C / 32 + wm * MS + m};
        gemm2(acc2, mt2);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int pos = l0 + col0 + n * 32;
            const bool ok = pos < L;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sc = (wm * MS + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (ok) sk[(size_t)sc * L + pos] = sr[n][r] + (acc2[0][n][r] + a.bias2[C + sc]);
            }
        }
    }
    }
}

// Row-major fp32 W[M][K] -> (hi, lo) bf16 A fragments of v_mfma_f32_32x32x16_bf16:
//   out[((mt*NKB + kb)*2 + part)*64 + lane] = 8 bf16: W[mt*32 + (lane&31)][kb*16 + 8*(lane>>5) + i]
__global__ void pack_a_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int M, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one (mt, kb, lane, elem)
    if (i >= (size_t)M * K) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const size_t r = i >> 9;  // mt * NKB + kb
    const int nkb = K / 16;
    const int kb = (int)(r % nkb), mt = (int)(r / nkb);
    const float v = w[(size_t)(mt * 32 + (lane & 31)) * K + kb * 16 + 8 * (lane >> 5) + e];
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    out[((r * 2 + 0) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, h);
    out[((r * 2 + 1) * 64 + lane) * 8 + e] = __builtin_bit_cast(unsigned short, l);
}

int launch_pack_a_bf16x3(const float* w, void* out, int M, int K, hipStream_t s) {
    const size_t n = (size_t)M * K;
    hipLaunchKernelGGL(pack_a_bf16x3_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, (unsigned short*)out, M, K);
    return DWS_OK;
}

// bf16x3 variant of wn_bias_tap_kernel: Abt[n][b][mt][part][lane][8] with k = t in lanes 0..31; k = 3 carries the
// dilated conv's bias (its indicator row is 1 at every position), so the gate stage has no bias loads
__global__ void wn_bias_tap_bf16_kernel(const float* __restrict__ Wd_all, const float* __restrict__ part_t,
                                        const float* __restrict__ bias1_all, unsigned short* __restrict__ Abt, int NL, int B,
                                        int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= NL * 2 * C) return;
    const int n = row / (2 * C), o = row % (2 * C);
    const float* w = Wd_all + (size_t)row * C * 3;
    constexpr int MAXR = 8;
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
        float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int c = lane + 64 * i;
            const float p = (c < C) ? pt[c] : 0.f;
            s[0] = fmaf(w0[i], p, s[0]); s[1] = fmaf(w1[i], p, s[1]); s[2] = fmaf(w2[i], p, s[2]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            s[0] += __shfl_xor(s[0], off); s[1] += __shfl_xor(s[1], off); s[2] += __shfl_xor(s[2], off);
        }
        if (lane == 0) {
            unsigned short* dst = Abt + ((((size_t)n * B + b) * (2 * C / 32) + o / 32) * 2) * 512;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float v = (t < 3) ? s[t] : bias1_all[row];
                const __bf16 h = (__bf16)v;
                const __bf16 l = (__bf16)(v - (float)h);
                dst[(o % 32) * 8 + t] = __builtin_bit_cast(unsigned short, h);
                dst[512 + (o % 32) * 8 + t] = __builtin_bit_cast(unsigned short, l);
            }
        }
    }
}

int launch_wn_bias_tap_bf16(const float* Wd_all, const float* part_t, const float* bias1_all, void* Abt, int NL, int B, int C,
                            hipStream_t s) {
    DWS_CHECK(C <= 512, DWS_ERR_UNSUPPORTED, "wn_bias_tap: C=%d > 512", C);
    hipLaunchKernelGGL(wn_bias_tap_bf16_kernel, dim3(ceil_div((int64_t)NL * 2 * C, 4)), dim3(256), 0, s, Wd_all, part_t,
                       bias1_all, (unsigned short*)Abt, NL, B, C);
    return DWS_OK;
}

template <int C, int S, int PP, int WV>
static int launch_bx3_t(const WnLayerArgs& a, hipStream_t s) {
    using T = Bx3Tile<C, S, PP, WV>;
    ProfileScope ps("wn_layer_bf16x3", s);
    const size_t lds = (size_t)T::LDS_FLOATS * 4;
    static bool attr_dev[DWS_MAX_DEVICES] = {};
    bool& attr = attr_dev[current_device_slot()];
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)wn_layer_bf16x3_kernel<C, S, PP, WV, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DWS_HIP(hipFuncSetAttribute((const void*)wn_layer_bf16x3_kernel<C, S, PP, WV, false>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const dim3 grid(a.B * ceil_div(a.L, T::P));
    if ((a.L & 3) == 0)
        hipLaunchKernelGGL((wn_layer_bf16x3_kernel<C, S, PP, WV, true>), grid, dim3(T::THREADS), lds, s, a);
    else
        hipLaunchKernelGGL((wn_layer_bf16x3_kernel<C, S, PP, WV, false>), grid, dim3(T::THREADS), lds, s, a);
    return DWS_OK;
}

bool wn_layer_bf16x3_supported(int C, int S) {
    return (C == 64 && S == 64) || (C == 128 && S == 128) || (C == 128 && S == 256) || (C == 256 && S == 256);
}

int launch_wn_layer_bf16x3(int C, int S, const WnLayerArgs& a, hipStream_t s) {
    // tile shape: 128 positions x 8 waves, one workgroup per CU.  (64 positions x 4 waves with two
    // workgroups per CU measured 1.02 ms vs 0.79 ms per launch at C = S = 256: twice the weight traffic.)
    if (C == 64 && S == 64) return launch_bx3_t<64, 64, 128, 8>(a, s);
    if (C == 128 && S == 128) return launch_bx3_t<128, 128, 128, 8>(a, s);
    if (C == 128 && S == 256) return launch_bx3_t<128, 256, 128, 8>(a, s);
    if (C == 256 && S == 256) return launch_bx3_t<256, 256, 128, 8>(a, s);
    return set_error(DWS_ERR_UNSUPPORTED, "wn_layer_bf16x3: (C=%d,S=%d) not instantiated", C, S);
}

}  // namespace dws
