// WaveNet fused residual layer on the bf16 matrix cores with a 3-term split ("bf16x3"):
//   x = x_hi + x_lo, W = W_hi + W_lo (each part bf16),  W x ~= W_hi x_hi + W_hi x_lo + W_lo x_hi
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped W_lo x_lo term and the bf16
// rounding of the lo parts are each <= 2^-18 relative, i.e. ~1e-5 worst case per product --
// two orders of magnitude inside the 1e-3 parity bound (measured: see tests/test_wavenet_gpu.py),
// at 16/3 = 5.3x the matrix rate of the exact-f32 MFMA.  Opt-in: precision = "bf16x3".
//
// Same structure as wn_layer_mfma_kernel (wavenet_kernels.hip) -- LDS-DMA staging of the raw x
// window with hardware zero padding, the step embedding as extra K rows, gate in registers,
// [res; skip] GEMM from the LDS gate tile -- but with a 128-position tile and 8 waves so every A
// fragment (streamed from L2) feeds 4 position tiles, and B fragments built on the fly from the
// fp32 LDS window: 8 ds_read_b32 down the k axis -> split into (hi, lo) bf16x8.
#include "wavenet.h"

namespace dws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 buf_load_bf8(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)x[i];
        hi[i] = h;
        lo[i] = (__bf16)(x[i] - (float)h);
    }
}

__device__ __forceinline__ float fast_sigmoid3(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh3(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

template <int C, int S>
struct Bx3Tile {
    static constexpr int P = 128;
    static constexpr int WAVES = 8;
    static constexpr int WM = (C / 32 >= 8) ? 8 : C / 32;
    static constexpr int WN = WAVES / WM;
    static constexpr int NT = (P / 32) / WN;
    static constexpr int MP = C / 32 / WM;   // (tanh, sigmoid) tile pairs per wave
    static constexpr int MR = C / 32 / WM;
    static constexpr int MS = S / 32 / WM;
    static constexpr int KC = WN_LAYER_KC;
    static constexpr int NCB = C / KC;
    static constexpr int XS_FLOATS = 2 * 3 * KC * P;
    static constexpr int IND_FLOATS = 16 * P;
    static constexpr int G_FLOATS = C * P;
    static constexpr int LDS_FLOATS = (XS_FLOATS + IND_FLOATS) > G_FLOATS ? (XS_FLOATS + IND_FLOATS) : G_FLOATS;
    static_assert(WN * NT * 32 == P && C % (32 * WM) == 0 && S % (32 * WM) == 0 && C % KC == 0, "tiling");
};

// acc[m][n] += (Ahi + Alo)[m] . (x_hi + x_lo)[n]  minus the lo*lo term, for one k-block of 16
template <int MT, int NT, int P>
__device__ __forceinline__ void kblock(f32x16 (&acc)[MT][NT], const bf16x8 (&ahi)[MT], const bf16x8 (&alo)[MT],
                                       const float* __restrict__ bt, int krow0, int col0) {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = bt[(krow0 + i) * P + col0 + n * 32];
        bf16x8 bhi, blo;
        split8(x, bhi, blo);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[m], bhi, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[m], blo, acc[m][n], 0, 0, 0);
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[m], bhi, acc[m][n], 0, 0, 0);
        }
    }
}

template <int C, int S>
__global__ __launch_bounds__(512) void wn_layer_bf16x3_kernel(WnLayerArgs a) {
    using T = Bx3Tile<C, S>;
    constexpr int P = T::P, KC = T::KC, NT = T::NT, MP = T::MP, MR = T::MR, MS = T::MS;
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

    // ---- staging by LDS-DMA (see wn_layer_mfma_kernel): 3*KC rows x P floats per chunk, two 256-byte
    // pieces per row; positions outside [0, L) are out of range for the row descriptor -> 0.
    constexpr int PIECES = 3 * KC * (P / 64);
    constexpr int PPW = PIECES / T::WAVES;
    auto stage_dma = [&](int cb, int buf) {
        float* xs = lds + buf * (3 * KC * P);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int piece = wave + T::WAVES * i;
            const int row = piece / (P / 64), half = piece % (P / 64);
            const int tap = row / KC, cc = row % KC;
            const int c = cb * KC + cc;
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)c * L), 0, L * 4, 0x00020000);
            const int voff = (l0 + half * 64 + lane + (tap - 1) * dil) * 4;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, xs + row * P + half * 64, 4, voff, 0, 0, 0);
        }
    };

    // indicator rows (one k-block of 16): rows 0..2 = tap in range, rows 3..15 = 0
    float* ind = lds + T::XS_FLOATS;
    for (int i = tid; i < 16 * P; i += 512) {
        const int t = i / P, col = i % P;
        const int pos = l0 + col + (t - 1) * dil;
        ind[i] = (t < 3 && (unsigned)pos < (unsigned)L) ? 1.f : 0.f;
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
    bf16x8 ahi[2 * MP], alo[2 * MP], nhi[2 * MP], nlo[2 * MP];
#pragma unroll
    for (int m = 0; m < 2 * MP; ++m) {
        ahi[m] = buf_load_bf8(rA1, lane16, (mt1[m] * NKB1) * 2048);
        alo[m] = buf_load_bf8(rA1, lane16, (mt1[m] * NKB1) * 2048 + 1024);
    }
    __syncthreads();

    for (int cb = 0; cb < T::NCB; ++cb) {
        if (cb + 1 < T::NCB) stage_dma(cb + 1, (cb + 1) & 1);
        const float* xs = lds + (cb & 1) * (3 * KC * P);
#pragma unroll
        for (int it = 0; it < 3 * KC / 16; ++it) {
            const int kb = cb * (3 * KC / 16) + it;
            const int kbn = (kb + 1 < NKB1) ? kb + 1 : kb;
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m) {
                nhi[m] = buf_load_bf8(rA1, lane16, (mt1[m] * NKB1 + kbn) * 2048);
                nlo[m] = buf_load_bf8(rA1, lane16, (mt1[m] * NKB1 + kbn) * 2048 + 1024);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the A prefetch one k-block ahead (see wavenet_kernels.hip)
            kblock<2 * MP, NT, P>(acc, ahi, alo, xs, it * 16 + 8 * lhi, col0);
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m) { ahi[m] = nhi[m]; alo[m] = nlo[m]; }
        }
        __syncthreads();
    }
    // step-embedding correction rows (bf16 hi/lo fragments from wn_bias_tap_bf16_kernel); the
    // indicator operand is exact in bf16, so only (hi + lo) x ind is needed
    {
        const u32x4* Abt = reinterpret_cast<const u32x4*>(a.Abt) + (size_t)b * (2 * C / 32) * 2 * 64;
#pragma unroll
        for (int m = 0; m < 2 * MP; ++m) {
            ahi[m] = __builtin_bit_cast(bf16x8, Abt[(mt1[m] * 2 + 0) * 64 + lane]);
            alo[m] = __builtin_bit_cast(bf16x8, Abt[(mt1[m] * 2 + 1) * 64 + lane]);
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float x[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = ind[(8 * lhi + i) * P + col0 + n * 32];
            bf16x8 bi;
#pragma unroll
            for (int i = 0; i < 8; ++i) bi[i] = (__bf16)x[i];
#pragma unroll
            for (int m = 0; m < 2 * MP; ++m) {
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[m], bi, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[m], bi, acc[m][n], 0, 0, 0);
            }
        }
    }
    __syncthreads();

    // ---- gate -> fp32 tile [C][P] in LDS (aliases the staging buffers)
    float* gt = lds;
    const float* melb = a.melc ? a.melc + (size_t)(a.mel_bstride ? b : 0) * 2 * C * L : nullptr;
#pragma unroll
    for (int m = 0; m < MP; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int col = col0 + n * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (wm * MP + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float ht = acc[m][n][r] + a.bias1[ch];
                float hs = acc[MP + m][n][r] + a.bias1[C + ch];
                if (melb) {
                    const int pos = l0 + col;
                    if (pos < L) {
                        ht += melb[(size_t)ch * L + pos];
                        hs += melb[(size_t)(C + ch) * L + pos];
                    }
                }
                gt[ch * P + col] = fast_tanh3(ht) * fast_sigmoid3(hs);
            }
        }
    __syncthreads();

    // ---- GEMM2: [res; skip] = [Wr; Ws] g
    f32x16 acc2[MR + MS][NT];
#pragma unroll
    for (int m = 0; m < MR + MS; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][n][r] = 0.f;
    constexpr int NKB2 = C / 16;
    __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A2, 0, (C + S) * C * 4, 0x00020000);
    int mt2[MR + MS];
#pragma unroll
    for (int m = 0; m < MR + MS; ++m) mt2[m] = (m < MR) ? (wm * MR + m) : (C / 32 + wm * MS + (m - MR));
    bf16x8 chi[MR + MS], clo[MR + MS], dhi[MR + MS], dlo[MR + MS];
#pragma unroll
    for (int m = 0; m < MR + MS; ++m) {
        chi[m] = buf_load_bf8(rA2, lane16, (mt2[m] * NKB2) * 2048);
        clo[m] = buf_load_bf8(rA2, lane16, (mt2[m] * NKB2) * 2048 + 1024);
    }
    for (int kb = 0; kb < NKB2; ++kb) {
        const int kbn = (kb + 1 < NKB2) ? kb + 1 : kb;
#pragma unroll
        for (int m = 0; m < MR + MS; ++m) {
            dhi[m] = buf_load_bf8(rA2, lane16, (mt2[m] * NKB2 + kbn) * 2048);
            dlo[m] = buf_load_bf8(rA2, lane16, (mt2[m] * NKB2 + kbn) * 2048 + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        kblock<MR + MS, NT, P>(acc2, chi, clo, gt, kb * 16 + 8 * lhi, col0);
#pragma unroll
        for (int m = 0; m < MR + MS; ++m) { chi[m] = dhi[m]; clo[m] = dlo[m]; }
    }

    // ---- epilogue: x' = (x + res + b_r) * sqrt(.5);  skip_acc (+)= skip + b_s
    const float rs = 0.70710678118654752440f;
    float* __restrict__ xo = a.x_out + (size_t)b * C * L;
    float* __restrict__ sk = a.skip + (size_t)b * S * L;
    const bool first = a.first_layer, last = a.last_layer;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int pos = l0 + col0 + n * 32;
        const bool ok = pos < L;
        const int posc = ok ? pos : 0;
        if (!last) {
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                float xr[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = (wm * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    xr[r] = xb[(size_t)ch * L + posc];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = (wm * MR + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (ok) xo[(size_t)ch * L + pos] = (xr[r] + (acc2[m][n][r] + a.bias2[ch])) * rs;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MS; ++m) {
            float sr[16];
            if (!first) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int sc = (wm * MS + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    sr[r] = sk[(size_t)sc * L + posc];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sr[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sc = (wm * MS + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (ok) sk[(size_t)sc * L + pos] = sr[r] + (acc2[MR + m][n][r] + a.bias2[C + sc]);
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

// bf16x3 variant of wn_bias_tap_kernel: Abt[n][b][mt][part][lane][8] with k = t in lanes 0..31
__global__ void wn_bias_tap_bf16_kernel(const float* __restrict__ Wd_all, const float* __restrict__ part_t,
                                        unsigned short* __restrict__ Abt, int NL, int B, int C) {
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
            for (int t = 0; t < 3; ++t) {
                const __bf16 h = (__bf16)s[t];
                const __bf16 l = (__bf16)(s[t] - (float)h);
                dst[(o % 32) * 8 + t] = __builtin_bit_cast(unsigned short, h);
                dst[512 + (o % 32) * 8 + t] = __builtin_bit_cast(unsigned short, l);
            }
        }
    }
}

int launch_wn_bias_tap_bf16(const float* Wd_all, const float* part_t, void* Abt, int NL, int B, int C, hipStream_t s) {
    DWS_CHECK(C <= 512, DWS_ERR_UNSUPPORTED, "wn_bias_tap: C=%d > 512", C);
    hipLaunchKernelGGL(wn_bias_tap_bf16_kernel, dim3(ceil_div((int64_t)NL * 2 * C, 4)), dim3(256), 0, s, Wd_all, part_t,
                       (unsigned short*)Abt, NL, B, C);
    return DWS_OK;
}

template <int C, int S>
static int launch_bx3_t(const WnLayerArgs& a, hipStream_t s) {
    using T = Bx3Tile<C, S>;
    ProfileScope ps("wn_layer_bf16x3", s);
    auto kern = wn_layer_bf16x3_kernel<C, S>;
    const size_t lds = (size_t)T::LDS_FLOATS * 4;
    static bool attr = false;
    if (!attr) {
        DWS_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL(kern, dim3(a.B * ceil_div(a.L, T::P)), dim3(512), lds, s, a);
    return DWS_OK;
}

bool wn_layer_bf16x3_supported(int C, int S) {
    return (C == 64 && S == 64) || (C == 128 && S == 128) || (C == 128 && S == 256) || (C == 256 && S == 256);
}

int launch_wn_layer_bf16x3(int C, int S, const WnLayerArgs& a, hipStream_t s) {
    if (C == 64 && S == 64) return launch_bx3_t<64, 64>(a, s);
    if (C == 128 && S == 128) return launch_bx3_t<128, 128>(a, s);
    if (C == 128 && S == 256) return launch_bx3_t<128, 256>(a, s);
    if (C == 256 && S == 256) return launch_bx3_t<256, 256>(a, s);
    return set_error(DWS_ERR_UNSUPPORTED, "wn_layer_bf16x3: (C=%d,S=%d) not instantiated", C, S);
}

}  // namespace dws
