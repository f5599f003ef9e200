// Data gradient of the dilated 3-tap convolution (`wavenet.py:95`, adjoint w.r.t. its input) in Winograd F(2,3) form along
// the dilation stride -- the training-side twin of wavenet_wino.hip:
//     y[l] = sum_k g0[m,k] x[k, l-d] + g1[m,k] x[k, l] + g2[m,k] x[k, l+d]
// Outputs l and l+d share inputs: with  t0 = x[l-d]-x[l+d], t1 = x[l]+x[l+d], t2 = x[l+d]-x[l], t3 = x[l]-x[l+2d]  and
// G0 = g0, G1 = (g0+g1+g2)/2, G2 = (g0-g1+g2)/2, G3 = g2 (folded when the adjoint weights are packed),
//     y[l] = G0 t0 + G1 t1 + G2 t2,   y[l+d] = G1 t1 - G2 t2 - G3 t3:
// four K-deep products per position PAIR instead of six.  A workgroup owns 64 pair columns q (positions
// p(q) = (q/d) 2d + q%d and p+d) x one 256- (MT = 2) or 128-row (MT = 1) block of M: 8 waves = 4 (rows) x 2 (columns),
// every wave keeps the four products of its MT x one 32-column tiles in 4 MT accumulators (128 registers at MT = 2, hence
// one workgroup per CU) and combines them in the epilogue.  The raw rows x[k, p + (t-1) d], t = 0..3, are staged by
// LDS-DMA exactly like four taps of the direct kernel (per-row descriptors: positions outside [0, L) read 0); the
// transform is four VALU operations per k-step beside 4 MT MFMAs.
#include <cstdlib>

#include "wavenet_backward.h"

namespace dws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Row-major [M = C][4 K] block of the transformed adjoint weights in the kernel's K order
// k' = (cb * 4 + j) * 32 + cc, o = cb * 32 + cc, from the folded conv weight W[o][c][t]:  tap t of the forward reads
// x[l + (t-1) d], so the adjoint's g0 (on dH[l-d]) is W[.,.,2], g1 = W[.,.,1], g2 = W[.,.,0].
__global__ void tapwino_pack_transposed_kernel(const float* __restrict__ W, float* __restrict__ out, int O, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * O) return;
    const int m = (int)(i / O), o = (int)(i % O);
    const float* w = W + ((size_t)o * C + m) * 3;
    const float g0 = w[2], g1 = w[1], g2 = w[0];
    const int cb = o / 32, cc = o % 32;
    float* dst = out + (size_t)m * 4 * O + (size_t)cb * 128 + cc;
    dst[0] = g0;
    dst[32] = 0.5f * (g0 + g1 + g2);
    dst[64] = 0.5f * (g0 - g1 + g2);
    dst[96] = g2;
}

int launch_tapwino_pack_transposed(const float* W, float* out, int O, int C, hipStream_t s) {
    const size_t n = (size_t)C * O;
    hipLaunchKernelGGL(tapwino_pack_transposed_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, W, out, O, C);
    return DWS_OK;
}

template <int MT>
__global__ __launch_bounds__(512, 1) void tapwino_mfma_kernel(TapConvArgs a, int log2d) {
    constexpr int P = 64, KC = 32, ROWS = 4 * KC;
    __shared__ __attribute__((aligned(16))) float lds[2 * ROWS * P];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 3, wn = wave >> 2;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L, d = 1 << log2d;
    const int nq = ((L + 2 * d - 1) >> (log2d + 1)) << log2d;      // pair columns of a row: whole 2d-blocks
    const int ntq = (nq + P - 1) / P;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int b = tile / ntq, q0 = (tile % ntq) * P;
    auto pos_of = [&](int q) { return ((q >> log2d) << (log2d + 1)) + (q & (d - 1)); };
    if (pos_of(q0) >= L) return;                                    // d > L: the columns past the row (uniform)
    const int K = a.K0, ncb = K / KC;

    // staging: row (j, cc) of a chunk = x[k0 + cc, p(q0 + lane) + (j - 1) d]; a wave moves 16 of the 128 rows
    const int pv = pos_of(q0 + lane) * 4;
    auto stage_dma = [&](int cb, int buf) {
        float* xs = lds + buf * (ROWS * P);
        const float* base = a.src0 + ((size_t)b * K + (size_t)cb * KC) * L;
#pragma unroll
        for (int i = 0; i < ROWS / 8; ++i) {
            const int row = wave + 8 * i;
            const int j = row / KC, cc = row % KC;
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)cc * L), 0, L * 4, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, xs + row * P, 4, pv + (j - 1) * d * 4, 0, 0, 0);
        }
    };

    f32x16 acc[MT][4];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.M * a.nkg_total * 8 * 4, 0x00020000);
    const int lane16 = lane * 16;
    int mt[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) mt[m] = blockIdx.y * (4 * MT) + wm * MT + m;
    auto load_a = [&](f32x4 (&dst)[4][MT], int cb, int it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int kg = cb * (ROWS / 8) + j * (KC / 8) + it;
                dst[j][m] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rA, lane16, (mt[m] * a.nkg_total + kg) * 1024, 0));
            }
    };

    stage_dma(0, 0);
    f32x4 a_cur[4][MT], a_nxt[4][MT];
    load_a(a_cur, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): hipcc does not make a barrier wait for LDS-DMA
    __syncthreads();
    for (int cb = 0; cb < ncb; ++cb) {
        if (cb + 1 < ncb) stage_dma(cb + 1, (cb + 1) & 1);
        const float* xs = lds + (cb & 1) * (ROWS * P) + wn * 32 + l31;
#pragma unroll
        for (int it = 0; it < KC / 8; ++it) {
            const bool last = (it + 1 == KC / 8);
            if (!(last && cb + 1 == ncb)) load_a(a_nxt, last ? cb + 1 : cb, last ? 0 : it + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int js = 0; js < 4; ++js) {
                const int krow = it * 8 + js * 2 + lhi;
                const float x0 = xs[(0 * KC + krow) * P], x1 = xs[(1 * KC + krow) * P];
                const float x2 = xs[(2 * KC + krow) * P], x3 = xs[(3 * KC + krow) * P];
                const float t[4] = {x0 - x2, x1 + x2, x2 - x1, x1 - x3};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[j][m][js], t[j], acc[m][j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m) a_cur[j][m] = a_nxt[j][m];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // the LDS-DMA of chunk cb+1 (the A fragments of its first k-group ride along)
        __syncthreads();
    }

    // y[p] = m0 + m1 + m2, y[p+d] = m1 - m2 - m3  (+ addin * addscale)
    const int p = pos_of(q0 + wn * 32 + l31);
    const bool ok0 = p < L, ok1 = p + d < L;
    const size_t boff = (size_t)b * a.M * L;
    float* __restrict__ ob = a.out + boff;
    const float* __restrict__ ab = a.addin ? a.addin + boff : nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mt[m] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int i0 = row * L + p;
            const float m0 = acc[m][0][r], m1 = acc[m][1][r], m2 = acc[m][2][r], m3 = acc[m][3][r];
            float y0 = m0 + m1 + m2, y1 = m1 - m2 - m3;
            if (ab) {
                if (ok0) y0 = fmaf(ab[i0], a.addscale, y0);
                if (ok1) y1 = fmaf(ab[i0 + d], a.addscale, y1);
            }
            if (ok0) ob[i0] = y0;
            if (ok1) ob[i0 + d] = y1;
        }
    }
}

bool tapwino_mfma_supported(int M, int K, int dil) {
    static const bool off = getenv("DWS_TAPCONV_DIRECT") != nullptr;
    return !off && M % 128 == 0 && K % 32 == 0 && K > 0 && dil > 0 && (dil & (dil - 1)) == 0;
}

int launch_tapwino_mfma(const TapConvArgs& a, hipStream_t s) {
    ProfileScope ps("tapconv_mfma", s);
    DWS_CHECK(tapwino_mfma_supported(a.M, a.K0, a.dil) && a.K1 == 0 && a.T == 3 && a.epi == 0 && a.sign == -1, DWS_ERR_UNSUPPORTED,
              "tapwino_mfma: M=%d K=%d+%d T=%d dil=%d epi=%d sign=%d", a.M, a.K0, a.K1, a.T, a.dil, a.epi, a.sign);
    DWS_CHECK((long long)a.M * a.L < (1ll << 31) / 4, DWS_ERR_UNSUPPORTED, "tapwino_mfma: M * L too large for 32-bit offsets");
    int log2d = 0;
    while ((1 << log2d) < a.dil) ++log2d;
    const int nq = (int)((((long long)a.L + 2 * a.dil - 1) >> (log2d + 1)) << log2d);
    const int nt = a.B * ceil_div(nq, 64);
    if (a.M % 256 == 0)
        hipLaunchKernelGGL((tapwino_mfma_kernel<2>), dim3(nt, a.M / 256), dim3(512), 0, s, a, log2d);
    else
        hipLaunchKernelGGL((tapwino_mfma_kernel<1>), dim3(nt, a.M / 128), dim3(512), 0, s, a, log2d);
    DWS_HIP(hipGetLastError());
    return DWS_OK;
}

}  // namespace dws
