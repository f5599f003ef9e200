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
#include <algorithm>
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

    // staging: row (j, cc) of a chunk = x[k0 + cc, p(q0 + lane) + (j - 1) d]; a wave moves 16 of the 128 rows.
    // L % 4 == 0 and d >= 4: four consecutive pair columns are four consecutive, 16-byte aligned positions at every shift,
    // so the chunk moves as 16-byte LDS-DMA -- four whole rows (4 x 64 columns) per instruction, 4 instructions per wave and
    // chunk instead of 16; a quad outside [0, L) gets an offset past the chunk's descriptor (reads 0).
    const int pv = pos_of(q0 + lane) * 4;
    const bool x4 = (L % 4 == 0) && log2d >= 2 && ((size_t)a.src0 % 16 == 0);
    const int p4 = pos_of(q0 + 4 * (lane & 15));
    auto stage_dma = [&](int cb, int buf) {
        float* xs = lds + buf * (ROWS * P);
        const float* base = a.src0 + ((size_t)b * K + (size_t)cb * KC) * L;
        if (x4) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, KC * L * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ii = wave + 8 * i, j = ii >> 3, r4 = ii & 7;
                const int pos = p4 + (j - 1) * d;
                const int voff = ((unsigned)pos < (unsigned)L) ? ((4 * r4 + (lane >> 4)) * L + pos) * 4 : 0x7ffffff0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, xs + (j * KC + 4 * r4) * P, 16, voff, 0, 0, 0);
            }
            return;
        }
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

// ---------------------------------------------------------------------------
// Weight gradient of the same convolution in the same pairing:  with u0 = dy[p], u1 = dy[p] + dy[p+d], u2 = dy[p] - dy[p+d],
// u3 = -dy[p+d] and the input transforms t0..t3 above,   dG_j[o, c] = sum_{b, q} u_j[o, q] t_j[c, q]   (four GEMMs over the
// PAIR columns q instead of three over all positions: 2 L instead of 3 L multiply-adds per weight), and
//     dW[.,.,0] = dG0 + (dG1 + dG2)/2,   dW[.,.,1] = (dG1 - dG2)/2,   dW[.,.,2] = dG3 + (dG1 + dG2)/2.
// Same structure as wgrad_mfma_kernel<3> (128 x 128 output tile per block and product j, the column range split over
// blocks, register-staged chunks: both operands of a chunk are sums / differences of two loads each, formed when the
// registers go to LDS; Xh = X + addc[b, c] inside [0, L), 0 outside, so the per-(b, c) constant enters with the factor
// in(pa) +- in(pb)).  The bias gradient rides on product 1, whose dY operand sums every position exactly once.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void wgrad_wino_kernel(WgradArgs a, int log2d) {
    constexpr int PC = 64, LD = PC + 2, RPW = 32;
    __shared__ float sdy[128 * LD];
    __shared__ float sx[128 * LD];
    __shared__ float sad[128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wo = wave & 1, wc = wave >> 1;                 // 2 x 2 waves over the 128 x 128 tile
    const int o0 = blockIdx.x * 128, c0 = blockIdx.y * 128;
    const int tap = blockIdx.z & 3, split = blockIdx.z >> 2;
    const int L = a.L, d = 1 << log2d;
    const int nq = ((L + 2 * d - 1) >> (log2d + 1)) << log2d;
    const int chunks_per_b = (nq + PC - 1) / PC;
    const int total_chunks = a.B * chunks_per_b;
    const int per = (total_chunks + a.nsplit - 1) / a.nsplit;
    const int ch_begin = split * per, ch_end = min(total_chunks, ch_begin + per);
    const bool do_bias = a.bias_part != nullptr && blockIdx.y == 0 && tap == 1;
    // product j reads Xh at p + xa and p + xb (t_j = Xh[p + xa] + xsign Xh[p + xb]) and dY at p and / or p + d
    const int xa = (tap == 0) ? -d : (tap == 2) ? d : 0;
    const int xb = (tap == 2) ? 0 : (tap == 3) ? 2 * d : d;
    const float xsign = (tap == 1) ? 1.f : -1.f, ysign = (tap == 1) ? 1.f : -1.f;
    const bool y0_used = tap != 3, y1_used = tap != 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum = 0.f;

    __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)a.dY, 0, (int)((size_t)a.B * a.O * L * 4), 0x00020000);
    __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)a.X, 0, (int)((size_t)a.B * a.C * L * 4), 0x00020000);
    float rdy0[RPW], rdy1[RPW], rx0[RPW], rx1[RPW];
    float mfac = 0.f;            // in(pa) + xsign in(pb): the factor of the per-(b, c) constant in this lane's column
    int fb = 0, sad_b = -1;
    constexpr int OOB = 0x7ffffff0;   // lane offset past the descriptor: the load returns 0
    auto fetch = [&](int ch) {
        const int b = ch / chunks_per_b, q = (ch % chunks_per_b) * PC + lane;
        const int p = ((q >> log2d) << (log2d + 1)) + (q & (d - 1));
        const int pa = p + xa, pb = p + xb;
        const bool ina = (unsigned)pa < (unsigned)L, inb = (unsigned)pb < (unsigned)L;
        mfac = (ina ? 1.f : 0.f) + (inb ? xsign : 0.f);
        fb = b;
        const int vy0 = (y0_used && p < L) ? p * 4 : OOB, vy1 = (y1_used && p + d < L) ? (p + d) * 4 : OOB;
        const int vxa = ina ? pa * 4 : OOB, vxb = inb ? pb * 4 : OOB;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + 4 * i;
            // rows past O / C are clamped, not masked: they only feed output elements that are never stored
            const int o = min(o0 + row, a.O - 1), c = min(c0 + row, a.C - 1);
            rdy0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rY, vy0, (b * a.O + o) * L * 4, 0));
            rdy1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rY, vy1, (b * a.O + o) * L * 4, 0));
            rx0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, vxa, (b * a.C + c) * L * 4, 0));
            rx1[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, vxb, (b * a.C + c) * L * 4, 0));
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
            sdy[row * LD + lane] = fmaf(ysign, rdy1[i], rdy0[i]);
            float xv = fmaf(xsign, rx1[i], rx0[i]);
            if (a.addc) xv = fmaf(sad[row], mfac, xv);
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
    // partial[split][o][c][j]
    float* part = a.partial + (size_t)split * a.O * a.C * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + wo * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int c = c0 + wc * 64 + j * 32 + l31;
                if (o < a.O && c < a.C) part[((size_t)o * a.C + c) * 4 + tap] = acc[i][j][r];
            }
    if (do_bias && tid < 128 && o0 + tid < a.O) a.bias_part[(size_t)split * a.O + o0 + tid] = bsum;
}

// dW[o][c][0..2] from the four product partials, summed over the splits in a fixed order (deterministic); blocks past
// blocks1: the bias partials of the same GEMM.
__global__ __launch_bounds__(256) void wgrad_wino_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW,
                                                                size_t noc, int nsplit, float scale,
                                                                const float* __restrict__ bias_part, float* __restrict__ dbias,
                                                                int O, float bias_scale, int blocks1) {
    if ((int)blockIdx.x >= blocks1) {
        const int o = ((int)blockIdx.x - blocks1) * 256 + threadIdx.x;
        if (o >= O) return;
        float sacc = 0.f;
        for (int k = 0; k < nsplit; ++k) sacc += bias_part[(size_t)k * O + o];
        dbias[o] = sacc * bias_scale;
        return;
    }
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= noc) return;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nsplit; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(partial + ((size_t)k * noc + i) * 4);
        g.x += v.x; g.y += v.y; g.z += v.z; g.w += v.w;
    }
    const float hs = 0.5f * (g.y + g.z);
    dW[i * 3 + 0] = (g.x + hs) * scale;
    dW[i * 3 + 1] = 0.5f * (g.y - g.z) * scale;
    dW[i * 3 + 2] = (g.w + hs) * scale;
}

bool wgrad_wino_supported(const WgradArgs& a) {
    static const bool off = getenv("DWS_WGRAD_DIRECT") != nullptr;
    return !off && !a.xact && (a.xL == 0 || a.xL == a.L) && a.dil > 0 && (a.dil & (a.dil - 1)) == 0 &&
           (size_t)a.B * std::max(a.O, a.C) * a.L * 4 < ((size_t)1 << 31);
}

static int log2_of(int d) {
    int l = 0;
    while ((1 << l) < d) ++l;
    return l;
}

int wgrad_wino_nsplit(int B, int O, int C, int L, int dil) {
    const int log2d = log2_of(dil);
    const int nq = (int)((((long long)L + 2 * dil - 1) >> (log2d + 1)) << log2d);
    const int tiles = ceil_div(O, 128) * ceil_div(C, 128) * 4;
    const int chunks = B * ceil_div(nq, 64);
    // workgroups to aim for (experiments: DWS_WGRAD_WINO_TARGET).  Same box, WaveNet training step: 512 -> 59.4 ms,
    // 768 -> 61.7, 1024 -> 60.0, 256 -> 63.6
    static const int target = getenv("DWS_WGRAD_WINO_TARGET") ? atoi(getenv("DWS_WGRAD_WINO_TARGET")) : 512;
    return std::min(std::max(1, target / tiles), chunks);  // two workgroups per CU, as the three-tap kernel
}

// a.partial: [nsplit][O][C][4] floats, a.nsplit from wgrad_wino_nsplit
int launch_wgrad_wino(const WgradArgs& a, float scale, float* dW, hipStream_t s) {
    ProfileScope ps("wgrad_mfma", s);
    DWS_CHECK(wgrad_wino_supported(a), DWS_ERR_UNSUPPORTED, "wgrad_wino: dil=%d xact=%d xL=%d", a.dil, a.xact, a.xL);
    const dim3 grid(ceil_div(a.O, 128), ceil_div(a.C, 128), 4 * a.nsplit);
    hipLaunchKernelGGL(wgrad_wino_kernel, grid, dim3(256), 0, s, a, log2_of(a.dil));
    const size_t noc = (size_t)a.O * a.C;
    const int blocks1 = (int)ceil_div(noc, 256), blocks2 = a.bias_part ? ceil_div(a.O, 256) : 0;
    hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(blocks1 + blocks2), dim3(256), 0, s, (const float*)a.partial, dW, noc,
                       a.nsplit, scale, (const float*)a.bias_part, a.dbias, a.O, a.bias_scale, blocks1);
    DWS_HIP(hipGetLastError());
    return DWS_OK;
}

}  // namespace dws
