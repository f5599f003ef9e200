// Fused WaveNet residual layer with the dilated 3-tap convolution in Winograd F(2,3) form ALONG THE DILATION STRIDE
// (`models/wavenet.py:82-121`; the conv is `wavenet.py:19-20,95`).
//
//   H[o,l] = sum_c W0[o,c] h[c,l-d] + W1[o,c] h[c,l] + W2[o,c] h[c,l+d]          (h = x + fc_t(e), 0 outside [0,L))
//
// Outputs l and l+d share three of their inputs.  With d0..d3 = h[l-d], h[l], h[l+d], h[l+2d] and
//   t0 = d0-d2   t1 = d1+d2   t2 = d2-d1   t3 = d1-d3           (one VALU op each, per B fragment)
//   G0 = W0      G1 = (W0+W1+W2)/2   G2 = (W0-W1+W2)/2   G3 = W2  (folded once at commit)
//   m_j = G_j t_j                                                (four K=C GEMMs per position PAIR instead of six)
//   H[l] = m0+m1+m2      H[l+d] = m1-m2-m3
// the convolution costs 8 C^2 flop per position instead of 12 C^2; the layer 12 C^2 + 2CS... (-25 % at C=S).
//
// Pairing: the positions are cut into blocks of 2d; position q of the "first halves" (q = 0,1,2,...) is
//   p(q) = (q / d) * 2d + q % d,  its partner p(q) + d.
// A workgroup owns 32 consecutive q (for d >= 32 that is 32 contiguous positions and their 32 partners d further on;
// for d < 32 a contiguous block of 64 positions, pairs interleaved).  L is zero-extended to a multiple of 2d by the
// buffer descriptors' bounds check (a row descriptor covers exactly [0, L): anything outside reads 0 and its stores
// are dropped), at most 2.4 % extra pairs at L = 16000, d = 2048.
//
// The price of Winograd is accumulators: four products per pair = 2x the registers per output.  A wave therefore owns
// ONE (tanh, sigmoid) row-tile pair (64 of the 2C conv outputs) x 32 pairs x 4 products = 128 accumulator registers,
// and a workgroup has C/32 waves (8 at C = 256: one workgroup per CU, two waves per SIMD).
//
// Everything else follows the direct kernel (wavenet_kernels.hip): raw x staged by LDS-DMA through per-row
// descriptors (zero padding for free), A fragments streamed from L2 one k-group ahead, the step embedding as one
// extra k-step per product (exact at the padded edges: the B row is the Winograd transform of the in-range
// indicator), the conv bias riding in the second k of that step, gate in registers, [res; skip] GEMM from the gate
// tile in LDS, epilogue through buffer instructions.
#include "dws_common.h"
#include "wavenet.h"
#include "wn_trace.h"

#include <cstdlib>
#include <vector>

namespace dws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_ __attribute__((ext_vector_type(4)));

// Folded dilated-conv weight [2C][C][3] -> row-major [2C][4C] Winograd matrices in the K order the kernel walks:
// column (kg*4 + j)*8 + cc  holds  G_j[o][kg*8 + cc]  (so pack_a_frag's k-group kg*4+j is product j of channel group kg)
__global__ void wino_dconv_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int M) {
    const int K = 4 * C;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * K) return;
    const int o = (int)(i / K), k = (int)(i % K);
    const int kg = k / 32, j = (k % 32) / 8, cc = k % 8;
    const float* wr = w + ((size_t)o * C + kg * 8 + cc) * 3;
    const float w0 = wr[0], w1 = wr[1], w2 = wr[2];
    float g;
    if (j == 0) g = w0;
    else if (j == 1) g = 0.5f * ((w0 + w2) + w1);
    else if (j == 2) g = 0.5f * ((w0 + w2) - w1);
    else g = w2;
    out[i] = g;
}

int launch_wino_dconv(const float* w, float* out, int C, hipStream_t s) {
    const size_t n = (size_t)2 * C * 4 * C;
    hipLaunchKernelGGL(wino_dconv_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, s, w, out, C, 2 * C);
    return DWS_OK;
}

// Step-embedding correction of the Winograd products: Abt[n][b][j][o] = sum_c G_j[o,c] fc_t_n(e_b)[c].
// One wave per (layer n, output row o); the folded weight row [C][3] stays in registers for all b.
__global__ void wn_wino_bias_kernel(const float* __restrict__ Wd_all, const float* __restrict__ part_t,
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
            float* dst = Abt + ((size_t)n * B + b) * 4 * 2 * C;
            dst[0 * 2 * C + o] = s0;
            dst[1 * 2 * C + o] = 0.5f * ((s0 + s2) + s1);
            dst[2 * 2 * C + o] = 0.5f * ((s0 + s2) - s1);
            dst[3 * 2 * C + o] = s2;
        }
    }
}

int launch_wn_wino_bias(const float* Wd_all, const float* part_t, float* Abt, int NL, int B, int C, hipStream_t s) {
    DWS_CHECK(C <= 512, DWS_ERR_UNSUPPORTED, "wn_wino_bias: C=%d > 512", C);
    hipLaunchKernelGGL(wn_wino_bias_kernel, dim3(ceil_div((int64_t)NL * 2 * C, 4)), dim3(256), 0, s, Wd_all, part_t, Abt,
                       NL, B, C);
    return DWS_OK;
}

__device__ __forceinline__ float wino_gate(float t, float s) {   // tanh(t) * sigmoid(s), see fast_gate (wavenet_kernels.hip)
    const float tc = __builtin_amdgcn_fmed3f(t, -30.f, 30.f);
    const float e2 = __builtin_amdgcn_exp2f(tc * 2.8853900817779268f);
    const float en = __builtin_amdgcn_exp2f(s * -1.4426950408889634f);
    return (e2 - 1.f) * __builtin_amdgcn_rcpf((e2 + 1.f) * (1.f + en));
}

__device__ __forceinline__ f32x4 wino_load_f4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int C, int S>
struct WinoTile {
    static constexpr int WAVES = C / 32;          // one (tanh, sigmoid) tile pair per wave
    static constexpr int NTH = WAVES * 64;
    static constexpr int NP = 32;                 // position pairs per workgroup (64 positions)
    // channels per staged chunk = channels between two workgroup barriers.  C = 256: 64 (four barriers per tile; same box,
    // 32 -> 64: 59.99 -> 58.42 ms per C2 step -- the two waves of a SIMD pay their per-chunk staging / transform /
    // restart overhead at the same time, right after each barrier, so it does not hide behind the partner's MFMAs);
    // C <= 128: 16, so that two workgroups fit a CU's LDS
    static constexpr int KC = (C >= 256) ? 64 : 16;
    static constexpr int NCB = C / KC;
    static constexpr int MS = S / C;              // skip tiles per wave
    static constexpr int XS = KC * 4 * NP;        // floats of one chunk: raw [cc][s][j] / transformed [cc][j][4]
    static constexpr int G_FLOATS = C * 2 * NP;   // gate tile [C][32][2]: (first half, partner) of a pair adjacent
    // the gate tile lives in the two raw buffers when it fits there (their last use is the transform of the last chunk,
    // a chunk barrier before the gate is written)
    static constexpr bool G_ALIAS = G_FLOATS <= 2 * XS;
    static constexpr int LDS_FLOATS = 4 * XS + (G_ALIAS ? 0 : G_FLOATS);
    static constexpr int ITEMS = KC * NP / NTH;   // (channel, column) items per thread of the transform pass
    static_assert(C % 32 == 0 && S % C == 0 && C % KC == 0 && KC % (2 * WAVES) == 0 && (KC * NP) % NTH == 0 &&
                  (32 % KC == 0 || KC % 32 == 0) && LDS_FLOATS * 4 <= 163840, "channel counts");
};

template <int C, int S, bool EXTRA>
__global__ __launch_bounds__(C * 2, 2) void wn_layer_wino_kernel(WnLayerArgs a, int log2d) {
    using T = WinoTile<C, S>;
    constexpr int KC = T::KC, XS = T::XS, MS = T::MS, WAVES = T::WAVES, NCB = T::NCB, NTH = T::NTH;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    float* const Xraw = lds;                 // 2 chunks of raw x (LDS-DMA target)
    float* const Tt = lds + 2 * XS;          // 2 chunks of Winograd-transformed x (B operands)
    float* const gt = T::G_ALIAS ? lds : lds + 4 * XS;   // gate tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L, dil = 1 << log2d;

    const int nblk = (L + 2 * dil - 1) >> (log2d + 1);     // blocks of 2d
    const int ntl = ((nblk << log2d) + 31) >> 5;           // tiles of 32 pairs per batch element
    // (One tile per workgroup.  Persistent workgroups walking tiles bid, bid + G, ... -- with the epilogue of the waves that
    // finish GEMM2 last overlapping the next tile's staging -- were built and measured on the same box: 62.1 against 58.9 ms
    // per C2 step; the loop costs 26 spilled registers and the extra barrier exposes the skew between the two waves of a SIMD.)
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
    // first-half position of this lane's column
    const int q = q0 + l31;
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

    // ---- operands of the two extra k-steps, fetched first (their latency hides behind the first staging wait).
    // GEMM1: k = 0 carries the step-embedding correction (A = G_j fc_t(e), B = Winograd transform of the in-range indicator
    // of the four shifts), k = 1 of product 1 the conv bias (A = bias, B = 1: m1 enters both outputs of a pair with +1).
    // GEMM2: A = [b_r; b_s] (k = 0), B = 1.
    const int mt1[2] = {wave, C / 32 + wave};
    int mt2[1 + MS];
    mt2[0] = wave;
#pragma unroll
    for (int m = 0; m < MS; ++m) mt2[1 + m] = C / 32 + wave * MS + m;
    float av1[2][4], av2[1 + MS];
    {
        const float* Abt = a.Abt + (size_t)b * a.abt_bstride + step_row_off(a.step_idx, a.abt_tstride);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int row = mt1[m] * 32 + l31;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* src = (j == 1 && lhi) ? a.bias1 + row : Abt + j * 2 * C + row;
                av1[m][j] = *src;
            }
        }
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m) av2[m] = a.bias2[mt2[m] * 32 + l31];
    }

    // ---- staging: chunk cb = channels [cb*KC, cb*KC+KC) x 4 shifts (-d, 0, +d, +2d) x 32 columns of RAW x by LDS-DMA,
    // layout [row][shift][32].  A shifted position outside [0, L) must read 0 (the conv's zero padding).
    //  * L % 4 == 0 and d >= 4: 16 bytes per lane, one instruction = two channel rows x four shifts (1 KiB); lanes whose
    //    shifted position is out of range get an offset beyond the buffer (reads 0).  A quarter of the instructions of the
    //    dword form -- each LDS-DMA costs ~100 cycles of issue beside MFMAs (nodma ablation: 4 k cycles per tile);
    //  * otherwise: one dword per lane through one descriptor per channel row (num_records = L*4 bounds the row), one
    //    instruction = two shifts of one row.
    //  * d <= 16 (and L % 4 == 0): the four shifted copies of a tile overlap almost entirely -- the tile is one contiguous
    //    block of 64 positions -- so ONE contiguous row piece [base - 32, base + 96) is staged per channel (16-byte LDS-DMA,
    //    two rows per instruction, 512 contiguous bytes per row) and the transform picks its four inputs out of it
    //    (per-dilation times before: d = 1, 2 on the dword path 1.71 ms, d = 4, 8 1.66 ms, d >= 16 1.60 ms).
    // 16-byte DMA also needs the clip's base 16-byte aligned (true for every engine buffer when L % 4 == 0; checked so that
    // an offset pointer handed in through the C ABI falls back to the dword form instead of misaligned b128 accesses)
    const bool al16 = (L % 4 == 0) && (((size_t)a.x_in & 15) == 0);
    const bool contig = al16 && log2d <= 4;
    const bool x4 = contig || (al16 && log2d >= 2);
    constexpr int RPW = KC / WAVES;                // channel rows per wave and chunk (as pairs of adjacent rows)
    __amdgpu_buffer_rsrc_t rXall = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, C * L * 4, 0x00020000);
    // element offsets of this lane's column inside a staged row: shift s (0..3 = -d, 0, +d, +2d) sits at ob + s * os
    const int pbase = (q0 >> log2d) << (log2d + 1);          // first position of the tile's block (d < 32: a multiple of 64)
    const int ob = contig ? 32 + (p - pbase) - dil : l31;
    const int os = contig ? dil : 32;
    // Output columns of the [res; skip] GEMM.  Its two 32-column tiles are the first halves p(q) and the partners p(q) + d
    // of the tile's 32 pairs -- for d >= 32 two runs of 32 consecutive positions.  For d < 32 (contiguous staging: the tile
    // is one block of 64 positions with the pairs interleaved) that would scatter every store / atomic instruction over
    // twice the cache lines (d = 1: even positions in one instruction, odd ones in the other; the d <= 8 layers ran 1.68-1.70
    // ms against 1.60, `profiles/r03_wino_layer_times.txt`), so there the tiles are positions [base, base + 32) and
    // [base + 32, base + 64): the gate tile is WRITTEN in that column order (two dword writes per element instead of one
    // 8-byte write), the B-fragment reads of the GEMM and everything after it stay as they are.
    //   xo0 / xo1: where this lane's residual x values sit in a staged raw row;  go0 / go1: float offsets of this lane's
    //   (first half, partner) gate values inside a channel's 64 floats [column][tile].
    const int x0l = p - pbase, x1l = x0l + dil;                 // local positions of the pair (contig only)
    const int xo0 = contig ? 32 + l31 : ob + os, xo1 = contig ? 32 : os;
    const int go0 = contig ? (x0l & 31) * 2 + (x0l >> 5) : l31 * 2, go1 = contig ? (x1l & 31) * 2 + (x1l >> 5) : l31 * 2 + 1;
    int voffA, voffB;
    if (contig) {   // lane = (row parity, 4-float piece of [pbase - 32, pbase + 96))
        const int pp = pbase - 32 + 4 * (lane & 31);
        voffA = ((unsigned)pp < (unsigned)L) ? (lhi * L + pp) * 4 : 0x7ffffff0;
        voffB = 0;
    } else if (x4) {   // lane = (row parity, shift, column quad)
        const int s4 = (lane >> 3) & 3, qq = q0 + 4 * (lane & 7);
        const int pp = ((qq >> log2d) << (log2d + 1)) + (qq & (dil - 1)) + (s4 - 1) * dil;
        voffA = ((unsigned)pp < (unsigned)L) ? (lhi * L + pp) * 4 : 0x7ffffff0;
        voffB = 0;
    } else {
        voffA = (p + (lhi - 1) * dil) * 4;   // shifts -d (lhi 0), 0 (lhi 1); negative -> huge unsigned -> reads 0
        voffB = (p + (lhi + 1) * dil) * 4;   // shifts +d, +2d
    }
    auto stage_dma = [&](int cb) {
        float* xs = Xraw + (cb & 1) * XS;
#pragma unroll
        for (int i = 0; i < RPW / 2; ++i) {
            const int cc = 2 * (wave + WAVES * i);
            if (x4) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rXall, xs + cc * 128, 16, voffA, (cb * KC + cc) * L * 4, 0, 0);
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

    // [res; skip] accumulators: the res tile starts from x itself (picked out of the staged window below), so the
    // residual add costs neither a reload of x nor an add
    f32x16 acc2[1 + MS][2];
#pragma unroll
    for (int m = 0; m < 1 + MS; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[m][n][r] = 0.f;

    // ---- transform pass (all waves together, once per chunk instead of once per wave): raw chunk c1 -> t0..t3 in
    // B-fragment order [cc][j][4] (one ds_read_b128 per k-step and lane in the MFMA loop, address = lane*16 + immediate);
    // the wave whose res rows live in this chunk copies x[., p] and x[., p+d] (shifts 0 and +d) into its accumulators.
    auto transform = [&](int c1) {
        const float* xs = Xraw + (c1 & 1) * XS;
        f32x4* tt = reinterpret_cast<f32x4*>(Tt + (c1 & 1) * XS);
#pragma unroll
        for (int k = 0; k < T::ITEMS; ++k) {
            const int i = tid + NTH * k;
            const int cc = i >> 5, j = i & 31;
            const float* xr = xs + cc * 128 + ob;      // (j = tid & 31 = this lane's column in every item)
            const float d0 = xr[0], d1 = xr[os], d2 = xr[2 * os], d3 = xr[3 * os];
            f32x4 t;
            t[0] = d0 - d2; t[1] = d1 + d2; t[2] = d2 - d1; t[3] = d1 - d3;
            tt[i] = t;
        }
        if (KC >= 32) {
            if ((wave * 32) / KC == c1) {         // the chunk holds all 32 res rows of this wave
                const float* xw = xs + ((wave * 32) % KC) * 128 + xo0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* xr = xw + ((r & 3) + 8 * (r >> 2) + 4 * lhi) * 128;
                    acc2[0][0][r] = xr[0];
                    acc2[0][1][r] = xr[xo1];
                }
            }
        } else if ((c1 * KC) / 32 == wave) {      // KC < 32: the wave's rows span 32 / KC chunks
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (r & 3) + 8 * (r >> 2);            // + 4*lhi: never crosses a multiple of 8
                if ((ch % 32) / KC == c1 % ((32 / KC) > 0 ? (32 / KC) : 1)) {   // compile-time per r once c1's parity is known
                    const float* xr = xs + ((ch % KC) + 4 * lhi) * 128 + xo0;
                    acc2[0][0][r] = xr[0];
                    acc2[0][1][r] = xr[xo1];
                }
            }
        }
    };

    // ---- GEMM1: m_j[2C x 32] = G_j[2C x C] . t_j[C x 32],  j = 0..3;  this wave: rows of tiles `wave` and C/32 + wave
    f32x16 acc[2][4];
    constexpr int NKG = C / 8;   // channel groups of 8 (4 k-steps)
    __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A1, 0, 2 * C * 4 * C * 4, 0x00020000);
    const int lane16 = lane * 16;

    stage_dma(0);
    if (NCB > 1) stage_dma(1);
    f32x4 a_cur[2][4], a_nxt[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) a_cur[m][j] = wino_load_f4(rA1, lane16, ((mt1[m] * NKG) * 4 + j) * 1024);
    // chunk 0 has landed (this wave's part; the barrier makes it everyone's): all but the loads issued after it -- chunk 1
    // (RPW/2 or 2 RPW LDS-DMA instructions) and the 8 A fragments.  hipcc does not make a barrier wait for LDS-DMA.
    if (NCB > 1) {
        if (x4) __builtin_amdgcn_s_waitcnt(0x0F70 | ((RPW / 2 + 8) & 15) | (((RPW / 2 + 8) >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * RPW + 8) & 15) | (((2 * RPW + 8) >> 4) << 14));
    } else {
        __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
    }
    __syncthreads();
    stamp(1);
    {
        const float v0 = ((unsigned)(p - dil) < (unsigned)L) ? 1.f : 0.f;
        const float v1 = ((unsigned)p < (unsigned)L) ? 1.f : 0.f;
        const float v2 = ((unsigned)(p + dil) < (unsigned)L) ? 1.f : 0.f;
        const float v3 = ((unsigned)(p + 2 * dil) < (unsigned)L) ? 1.f : 0.f;
        float bi[4];
        bi[0] = lhi ? 0.f : v0 - v2;
        bi[1] = lhi ? 1.f : v1 + v2;
        bi[2] = lhi ? 0.f : v2 - v1;
        bi[3] = lhi ? 0.f : v1 - v3;
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float av = (lhi && j != 1) ? 0.f : av1[m][j];
                acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bi[j], zero, 0, 0, 0);
            }
    }
    transform(0);
    __builtin_amdgcn_s_waitcnt(0x0F70 | 8);   // chunk 1 has landed too (younger: only the 8 A fragments)
    __syncthreads();                          // transformed chunk 0 visible, raw chunk 1 complete

    // (Measured and dropped, same box: doing the staging / transform of the two waves of a SIMD at different k-groups, so
    // that one's non-MFMA work sits beside the other's MFMAs: 61.3 against 58.4 ms per step.)
    for (int cb = 0; cb < NCB; ++cb) {
        if (cb + 2 < NCB) stage_dma(cb + 2);       // into the raw buffer chunk cb occupied (transformed an iteration ago)
        if (cb + 1 < NCB) transform(cb + 1);
        const char* tb = reinterpret_cast<const char*>(Tt + (cb & 1) * XS) + lane16;
#pragma unroll
        for (int it = 0; it < KC / 8; ++it) {
            const int kg = cb * (KC / 8) + it;
            const int kgn = (kg + 1 < NKG) ? kg + 1 : kg;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) a_nxt[m][j] = wino_load_f4(rA1, lane16, ((mt1[m] * NKG + kgn) * 4 + j) * 1024);
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch a whole k-group (32 MFMAs) ahead of its use
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(tb + (it * 8 + ks * 2) * 512);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[m][j][ks], t[j], acc[m][j], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) a_cur[m][j] = a_nxt[m][j];
        }
        stamp(8 + 2 * cb);
        // transformed chunk cb+1 visible after the barrier; the LDS-DMA of chunk cb+2 must have landed too, and hipcc does
        // not count LDS-DMA among the accesses a barrier has to wait for: explicit vmcnt(0) (the only younger loads are
        // the A fragments of the next k-group, fetched a whole k-group ago)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        stamp(9 + 2 * cb);
    }
    stamp(2);

    // ---- gate: g = tanh(H_t (+mel_t)) * sigmoid(H_s (+mel_s)) for both outputs of every pair -> LDS [C][32][2]
    const float* melb = (EXTRA && a.melc) ? a.melc + (size_t)(a.mel_bstride ? b : 0) * 2 * C * L : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        float g2[2];
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
            if (EXTRA && melb && pos < L) {
                ht += melb[(size_t)ch * L + pos];
                hs += melb[(size_t)(C + ch) * L + pos];
            }
            if (EXTRA && a.hsave && pos < L) {   // training: keep the pre-activations for the gate adjoint
                float* __restrict__ hb = a.hsave + (size_t)b * 2 * C * L;
                hb[(size_t)ch * L + pos] = ht;
                hb[(size_t)(C + ch) * L + pos] = hs;
            }
            g2[n] = wino_gate(ht, hs);
        }
        if (contig) {
            gt[ch * 64 + go0] = g2[0];
            gt[ch * 64 + go1] = g2[1];
        } else {
            *reinterpret_cast<float2*>(gt + (ch * 32 + l31) * 2) = make_float2(g2[0], g2[1]);
        }
    }
    stamp(3);
    __syncthreads();
    stamp(4);

    // ---- GEMM2: [res; skip][(C+S) x 64] = [Wr; Ws][(C+S) x C] . g[C x 64] (+ bias k-step);  this wave: res tile `wave`
    // (accumulating onto x) and its skip tiles
    __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.A2, 0, (C + S) * C * 4, 0x00020000);
    const float rs = 0.70710678118654752440f;
    const bool first = a.first_layer, last = a.last_layer;
    __amdgpu_buffer_rsrc_t rXo = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x_out + (size_t)b * C * L), 0, C * L * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rSk = __builtin_amdgcn_make_buffer_rsrc((void*)(a.skip + (size_t)b * S * L), 0, S * L * 4, 0x00020000);
    const int L4 = L * 4;
    const char* gb = reinterpret_cast<const char*>(gt) + lane * 8;
    int voffn[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int pos = contig ? pbase + 32 * n + l31 : p + n * dil;
        voffn[n] = (pos < L) ? (4 * lhi * L + pos) * 4 : 0x7ffffff0;
    }
    constexpr bool TR_OK = 2 * XS >= WAVES * 2048;
#ifdef DWS_WINO_DWORD_EPI
    const bool vec_epi = false;
#else
    const bool vec_epi = TR_OK && al16 && ((((size_t)a.x_out | (size_t)a.skip) & 15) == 0);
#endif
    typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
    float* const trw = Tt + wave * 2048;
    const int p0 = ((q0 >> log2d) << (log2d + 1)) + (q0 & (dil - 1));
    const int n4 = (lane & 15) >> 3;
    const int pos4 = (contig ? pbase + 32 * n4 : p0 + n4 * dil) + 4 * (lane & 7);
    const int voff4 = (pos4 < L) ? ((lane >> 4) * L + pos4) * 4 : 0x7ffffff0;
    const int voff4s = first ? 0x7ffffff0 : voff4;      // the first layer starts the running sum: reads 0
    f32x4 sk[TR_OK ? MS : 1][TR_OK ? 8 : 1];
    if constexpr (TR_OK) {
        if (vec_epi) {   // requested here: their HBM round trip passes under GEMM2
#pragma unroll
            for (int m = 0; m < MS; ++m)
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    sk[m][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rSk, voff4s, ((wave * MS + m) * 32 + 4 * i) * L4, 2));
        }
    }
    // (Measured and dropped, same box: one row tile of [res; skip] at a time, so that the x' stores go out under the skip
    // tile's MFMAs: 58.1 against 58.8 / 59.5 ms per step in back-to-back runs -- inside the run-to-run spread.)
    f32x4 c_cur[1 + MS], c_nxt[1 + MS];
#pragma unroll
    for (int m = 0; m < 1 + MS; ++m) c_cur[m] = wino_load_f4(rA2, lane16, (mt2[m] * NKG) * 1024);
    {
        const float one = lhi ? 0.f : 1.f;
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc2[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(lhi ? 0.f : av2[m], one, acc2[m][n], 0, 0, 0);
    }
#pragma unroll 2
    for (int kg = 0; kg < NKG; ++kg) {
        const int kgn = (kg + 1 < NKG) ? kg + 1 : kg;
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m) c_nxt[m] = wino_load_f4(rA2, lane16, (mt2[m] * NKG + kgn) * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float2 bf = *reinterpret_cast<const float2*>(gb + (kg * 8 + ks * 2) * 256);
#pragma unroll
            for (int m = 0; m < 1 + MS; ++m) {
                acc2[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c_cur[m][ks], bf.x, acc2[m][0], 0, 0, 0);
                acc2[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c_cur[m][ks], bf.y, acc2[m][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m) c_cur[m] = c_nxt[m];
    }
    stamp(5);

    // ---- epilogue: x' = (x + res + b_r) * sqrt(.5) is the res accumulator scaled; skip_acc += skip + b_s goes out as a
    // no-return float atomic (one add per element and layer, layers are stream-ordered: the same bits as load-add-store,
    // without the load).  Buffer instructions: the row rides in the scalar offset, the lane part is one 32-bit offset per
    // column; a position past L gets an out-of-range offset and is dropped.
    // Round 5: where the rows are 16-byte aligned and the two transformed-chunk buffers (dead since GEMM1) give every wave an
    // 8 KB slot, the tiles leave as row-major 16-byte rows -- each row tile transposed through the wave's own slot (no other
    // wave touches it: no barrier), skip as load-add-store (one workgroup owns an element per layer): 16 + 16 MS wide VMEM
    // instructions per lane instead of 32 + 32 MS dword stores / atomics, whose ISSUE was the epilogue's time (12-18 k cycles
    // per tile, `profiles/r04_wino_phase_trace.txt`).  DWS_WINO_DWORD_EPI (build flag) keeps the dword form for A/B runs.
    if (TR_OK && vec_epi) {
      if constexpr (TR_OK) {
#pragma unroll
        for (int m = 0; m < 1 + MS; ++m) {
            if (m == 0 && last) continue;
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    trw[((r & 3) + 8 * (r >> 2) + 4 * lhi) * 64 + n * 32 + l31] = (m == 0) ? acc2[0][n][r] * rs : acc2[m][n][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave, its LDS operations execute in order
            const int row0 = (m == 0) ? wave * 32 : (wave * MS + (m - 1)) * 32;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                f32x4 v = *reinterpret_cast<const f32x4*>(trw + ((lane >> 4) + 4 * i) * 64 + 4 * (lane & 15));
                if (m > 0) v += sk[m > 0 ? m - 1 : 0][i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4e, v), m == 0 ? rXo : rSk, voff4, (row0 + 4 * i) * L4, 2);
                asm volatile("s_nop 1" ::: "memory");   // gfx950: no VALU write to a 16-byte store's data in the next slot (wavenet_bx6.hip)
            }
            asm volatile("" ::: "memory");
        }
      }
    } else {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int voff = voffn[n];
            if (!last) {
                const int s0 = (wave * 32) * L4;
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc2[0][n][r] * rs;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rXo, voff, s0 + ((r & 3) + 8 * (r >> 2)) * L4, 0);
                }
            }
    #pragma unroll
            for (int m = 0; m < MS; ++m) {
                const int s0 = ((wave * MS + m) * 32) * L4;
                if (first) {
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc2[1 + m][n][r];   // (bit_cast straight from a vector element picks element 0)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rSk, voff,
                                                              s0 + ((r & 3) + 8 * (r >> 2)) * L4, 0);
                    }
                } else {
    #pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(acc2[1 + m][n][r], rSk, voff,
                                                                        s0 + ((r & 3) + 8 * (r >> 2)) * L4, 0);
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

template <int C, int S>
static int launch_wino_t(const WnLayerArgs& a, int log2d, hipStream_t s) {
    ProfileScope ps("wn_layer_wino_mfma", s);   // ("mfma": counted among the MFMA GEMM launches of the training roofline)
    using T = WinoTile<C, S>;
    const int dil = 1 << log2d;
    const int nblk = (a.L + 2 * dil - 1) / (2 * dil);
    const int ntl = (nblk * dil + 31) / 32;
    const int ntiles = a.B * ntl;
    static const bool trace = std::getenv("DWS_WINO_TRACE") != nullptr;
    if (trace && !(a.melc || a.hsave)) {
        wino_trace_launch(ntiles, T::WAVES, a, s, [&](const WnLayerArgs& at) {
            hipLaunchKernelGGL((wn_layer_wino_kernel<C, S, false>), dim3(ntiles), dim3(T::NTH), 0, s, at, log2d);
        });
        return DWS_OK;
    }
    if (a.melc || a.hsave) hipLaunchKernelGGL((wn_layer_wino_kernel<C, S, true>), dim3(ntiles), dim3(T::NTH), 0, s, a, log2d);
    else hipLaunchKernelGGL((wn_layer_wino_kernel<C, S, false>), dim3(ntiles), dim3(T::NTH), 0, s, a, log2d);
    return DWS_OK;
}

bool wn_layer_wino_supported(int C, int S) {
    return (C == 64 && S == 64) || (C == 128 && S == 128) || (C == 128 && S == 256) || (C == 256 && S == 256);
}

int launch_wn_layer_wino(int C, int S, const WnLayerArgs& a, hipStream_t s) {
    int log2d = 0;
    while ((1 << log2d) < a.dilation) ++log2d;
    DWS_CHECK((1 << log2d) == a.dilation, DWS_ERR_UNSUPPORTED, "wn_layer_wino: dilation %d is not a power of two", a.dilation);
    DWS_CHECK((int64_t)a.L + 4 * (int64_t)a.dilation < ((int64_t)1 << 28), DWS_ERR_UNSUPPORTED, "wn_layer_wino: L too large");
    // one buffer descriptor spans a batch element's [C][L] (or [S][L]) tensor: 32-bit byte offsets
    DWS_CHECK((int64_t)(C > S ? C : S) * a.L * 4 < ((int64_t)1 << 31), DWS_ERR_UNSUPPORTED,
              "wn_layer_wino: %d channels x L=%d exceed a 2 GiB tensor per clip", C > S ? C : S, a.L);
    if (C == 64 && S == 64) return launch_wino_t<64, 64>(a, log2d, s);
    if (C == 128 && S == 128) return launch_wino_t<128, 128>(a, log2d, s);
    if (C == 128 && S == 256) return launch_wino_t<128, 256>(a, log2d, s);
    if (C == 256 && S == 256) return launch_wino_t<256, 256>(a, log2d, s);
    return set_error(DWS_ERR_UNSUPPORTED, "wn_layer_wino: (C=%d,S=%d) not instantiated", C, S);
}

}  // namespace dws
