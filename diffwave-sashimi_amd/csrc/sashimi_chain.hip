// Register-chained S4 tail for small channel counts (H = 32, 64): everything of DiffWaveBlock.forward after the S4
// convolution (`sashimi.py:177-184`, `s4.py:1435`) -- the same chain as s4_tail_mfma_kernel (sashimi_mfma.hip) --
// with NO LDS round trip and NO barrier between its stages:
//
//   o  = Wo g + bo            (H -> 2H)      x1 = x + o[:H] * sigmoid(o[H:]) (+ mel)
//   y  = LN2(x1)              (down the channel column, population std, no eps; `sashimi.py:17-20`)
//   u  = GELU(W1 y + b1)      (H -> ff H)    out = x1 + W2 u + b2 (+ U-Net skip)
//   (optional) ynext = LN1_next(out) + fc_t_next(e)      (`sashimi.py:148-152`)
//
// A WAVE owns 32 positions and all channels of them, and walks tiles on its own.  The accumulator layout of
// v_mfma_f32_32x32x2_f32 (lane = (column l31, half lhi); register r of tile t = row 32 t + (r & 3) + 8 (r >> 2) + 4 lhi) IS a
// legal B-operand layout of the next GEMM: as the B operand, register (t, r) supplies k = 0 from the lhi = 0 lanes and k = 1
// from the lhi = 1 lanes, i.e. channels c and c + 4 with c = 32 t + (r & 3) + 8 (r >> 2) -- so each GEMM consumes the previous
// stage's registers directly when its weight columns are packed in that order (chain_permute_cols at commit).  The
// channel-column reductions of the LayerNorms are sums over a lane's own registers plus ONE cross-half shuffle.
//
// Why: at H <= 64 the LDS kernel spends its time between GEMMs -- staging, two LayerNorm reductions with three barriers
// each, GELU / GLU passes over the tile in LDS, ~12 barrier-separated phases per 64-position tile for 96 (H = 32) or
// 192 (H = 64) MFMAs per wave: 32 % / 56 % of the MFMA roof.  Here a tile is one straight line of code per wave.
// The weights (6 H^2 floats: 24 KB at H = 32, 96 KB at H = 64) sit in LDS in A-fragment order, loaded once per
// workgroup; every wave reads each fragment once per tile (16 B/clk per CU at full MFMA rate).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sashimi.h"
#include "sashimi_mfma.h"

namespace dws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Column permutation of a row-major W[M][K] into chain order: out[row][2 kappa + h] = W[row][32 t + (r & 3) + 8 (r >> 2) + 4 h],
// kappa = 16 t + r.  pack_a_frag of the result gives the A fragments whose k-step kappa multiplies register (t, r).
__global__ void chain_permute_cols_kernel(const float* __restrict__ w, float* __restrict__ out, int M, int K) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * K) return;
    const int row = (int)(i / K), kp = (int)(i % K);
    const int kappa = kp >> 1, h = kp & 1, t = kappa >> 4, r = kappa & 15;
    out[i] = w[(size_t)row * K + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h];
}

int launch_chain_permute_cols(const float* w, float* out, int M, int K, hipStream_t s) {
    DWS_CHECK(K % 32 == 0, DWS_ERR_INVALID, "chain_permute_cols: K=%d", K);
    hipLaunchKernelGGL(chain_permute_cols_kernel, dim3(ceil_div((int64_t)M * K, 256)), dim3(256), 0, s, w, out, M, K);
    return DWS_OK;
}

template <int H, int FFE>
struct ChainCfg {
    static constexpr int TH = H / 32;            // row tiles of an H-channel tensor
    static constexpr int TO = 2 * H / 32;        // output_linear rows (GLU halves)
    static constexpr int TF = FFE * H / 32;      // feed-forward rows
    static constexpr int WAVES = (H >= 64) ? 8 : 4;
    static constexpr int THREADS = 64 * WAVES;
    static constexpr int W_FLOATS = 2 * H * H + 2 * FFE * H * H;        // Ao | A1 | A2, A-fragment order
    static constexpr int B_FLOATS = 2 * H + FFE * H + H;                // bo | b1 | b2
    static constexpr int LDS_FLOATS = W_FLOATS + B_FLOATS;
    static_assert(H % 32 == 0 && LDS_FLOATS * 4 <= 163840, "shape");
};

__device__ __forceinline__ float xhalf_sum(float v) { return v + __shfl_xor(v, 32); }

template <int H, int FFE, bool YNEXT>
__global__ __launch_bounds__((H >= 64 ? 512 : 256), 2) void s4_tail_chain_kernel(S4TailArgs a) {
    using T = ChainCfg<H, FFE>;
    constexpr int TH = T::TH, TO = T::TO, TF = T::TF;
    constexpr int OOB = 0x7ffffff0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const wo = lds;                         // [TO][H/8][64][4]
    float* const w1 = wo + 2 * H * H;              // [TF][H/8][64][4]
    float* const w2 = w1 + FFE * H * H;            // [TH][FFE*H/8][64][4]
    float* const bo = w2 + FFE * H * H;            // [2H]
    float* const b1 = bo + 2 * H;                  // [FFE*H]
    float* const b2 = b1 + FFE * H;                // [H]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int L = a.L, L4 = L * 4;

    // weights and biases -> LDS, once per workgroup (the only barrier of the kernel)
    {
        const f32x4* so = reinterpret_cast<const f32x4*>(a.Ao_c);
        const f32x4* s1 = reinterpret_cast<const f32x4*>(a.A1_c);
        const f32x4* s2 = reinterpret_cast<const f32x4*>(a.A2_c);
        f32x4* d = reinterpret_cast<f32x4*>(lds);
        for (int i = tid; i < 2 * H * H / 4; i += T::THREADS) d[i] = so[i];
        for (int i = tid; i < FFE * H * H / 4; i += T::THREADS) d[2 * H * H / 4 + i] = s1[i];
        for (int i = tid; i < FFE * H * H / 4; i += T::THREADS) d[(2 + FFE) * H * H / 4 + i] = s2[i];
        for (int i = tid; i < 2 * H; i += T::THREADS) bo[i] = a.bo[i];
        for (int i = tid; i < FFE * H; i += T::THREADS) b1[i] = a.b1[i];
        for (int i = tid; i < H; i += T::THREADS) b2[i] = a.b2[i];
    }
    __syncthreads();

    const float ln_m = a.ln_m[0], ln_s = a.ln_s[0];
    const float n1_m = YNEXT ? a.n1_m[0] : 0.f, n1_s = YNEXT ? a.n1_s[0] : 0.f;
    const bool has_mel = a.mel != nullptr, has_add = a.addend != nullptr;
    const float one = lhi ? 0.f : 1.f;             // B operand of the bias k-steps (k = 0 row of ones)
    const int ntl = (L + 31) / 32, ntiles = a.B * ntl;
    const float invH = 1.f / (float)H;

#define CH_SOFF(t, r) ((32 * (t) + ((r) & 3) + 8 * ((r) >> 2)) * L4)
    // DWS_CHAIN_TRACE=1 (tools only): every wave stamps the phases of its SECOND tile (steady state: weights in LDS, the
    // partner waves of its SIMD in their own tiles) into trace[workgroup][wave][16]
    unsigned long long* __restrict__ trc = a.trace ? a.trace + ((size_t)blockIdx.x * T::WAVES + wave) * 16 : nullptr;
    int tile_no = 0;
#define CH_STAMP(i)                                                        \
    if (trc && tile_no == 1) {                                             \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();        \
        if (lane == 0) trc[i] = t_;                                        \
    }
    // (Requesting g and x -- or g alone -- of the NEXT tile under the current tile's GEMM-2 was measured: 141.7 / 138.5 vs
    // 135.7 / 139.4 us at H = 64, 90.0 vs 88.9 at H = 32: no gain.  Counters (profiles/r03_c*_chain*_pmc.txt): MFMA busy 66 % /
    // 52 % of the cycles at H = 64 / 32, the non-MFMA VALU work -- GELU is half of it -- another 18 % / 25 %: the kernel is
    // arithmetic-bound (fp32 MFMA and VALU share the pipe; SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU = 4.4 cycles per VALU
    // instruction), not latency-bound.  Also measured: all tile I/O as 16-byte accesses through a per-wave 32 x 32 LDS
    // scratch tile (24 instead of 96-112 VMEM instructions per tile at H = 32): 90.5 vs 88.9 us -- the address path was not
    // the limit either.)
    for (int tile = blockIdx.x * T::WAVES + wave; tile < ntiles; tile += gridDim.x * T::WAVES) {
        CH_STAMP(0)
        const int b = __builtin_amdgcn_readfirstlane(tile / ntl);
        const int l0 = __builtin_amdgcn_readfirstlane((tile % ntl) * 32);
        const int pos = l0 + l31;
        const int voff = pos < L ? (4 * lhi * L + pos) * 4 : OOB;    // register (t, r): scalar offset (32 t + (r&3) + 8 (r>>2)) * L4
        __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(a.g + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x + (size_t)b * H * L), 0, H * L4, 0x00020000);
        __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)(a.out + (size_t)b * H * L), 0, H * L4, 0x00020000);
        f32x16 g[TH], x1[TH];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                g[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rG, voff, CH_SOFF(t, r), 0));
                x1[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, voff, CH_SOFF(t, r), 0));
            }
        if (has_mel) {
            __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(a.mel + (size_t)(a.mel_bstride ? b : 0) * H * L), 0, H * L4, 0x00020000);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    x1[t][r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rM, voff, CH_SOFF(t, r), 0));
        }

        CH_STAMP(1)
        // ---- GEMM-o: o[2H x 32] = Wo g + bo
        f32x16 ao[TO];
#pragma unroll
        for (int m = 0; m < TO; ++m) {
            const float bv = lhi ? 0.f : bo[m * 32 + l31];
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            ao[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, one, z, 0, 0, 0);
        }
        {   // A fragments of k-group kg+1 are read from LDS before the MFMAs of k-group kg (a wave has one or three partners
            // on its SIMD: the ~120-cycle LDS round trip per k-group showed); the fences keep hipcc from hoisting ALL reads
            f32x4 af[2][TO];
#pragma unroll
            for (int m = 0; m < TO; ++m) af[0][m] = *reinterpret_cast<const f32x4*>(wo + ((m * (H / 8)) * 64 + lane) * 4);
#pragma unroll
            for (int kg = 0; kg < H / 8; ++kg) {
                if (kg + 1 < H / 8) {
#pragma unroll
                    for (int m = 0; m < TO; ++m)
                        af[(kg + 1) & 1][m] = *reinterpret_cast<const f32x4*>(wo + ((m * (H / 8) + kg + 1) * 64 + lane) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kappa = kg * 4 + j;
#pragma unroll
                    for (int m = 0; m < TO; ++m)
                        ao[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg & 1][m][j], g[kappa >> 4][kappa & 15], ao[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        CH_STAMP(2)
        // ---- GLU + residual: x1 = x (+ mel) + o_a * sigmoid(o_b); LN2 down the channel column
        float s1 = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x1[t][r] = fmaf(ao[t][r], dws_sigmoid(ao[TH + t][r]), x1[t][r]);
                s1 += x1[t][r];
            }
        const float mean = xhalf_sum(s1) * invH;
        float sv = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                x1[t][r] -= mean;                       // x1 holds the centred value from here on
                sv = fmaf(x1[t][r], x1[t][r], sv);
            }
        const float alpha = ln_s / sqrtf(xhalf_sum(sv) * invH);
        f32x16 y[TH];
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[t][r] = alpha * (x1[t][r] + ln_m);

        CH_STAMP(3)
        // ---- GEMM-1: u[ff H x 32] = GELU(W1 y + b1)
        f32x16 u[TF];
#pragma unroll
        for (int m = 0; m < TF; ++m) {
            const float bv = lhi ? 0.f : b1[m * 32 + l31];
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            u[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, one, z, 0, 0, 0);
        }
        {
            f32x4 af[2][TF];
#pragma unroll
            for (int m = 0; m < TF; ++m) af[0][m] = *reinterpret_cast<const f32x4*>(w1 + ((m * (H / 8)) * 64 + lane) * 4);
#pragma unroll
            for (int kg = 0; kg < H / 8; ++kg) {
                if (kg + 1 < H / 8) {
#pragma unroll
                    for (int m = 0; m < TF; ++m)
                        af[(kg + 1) & 1][m] = *reinterpret_cast<const f32x4*>(w1 + ((m * (H / 8) + kg + 1) * 64 + lane) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kappa = kg * 4 + j;
#pragma unroll
                    for (int m = 0; m < TF; ++m)
                        u[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg & 1][m][j], y[kappa >> 4][kappa & 15], u[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        CH_STAMP(4)
        // the U-Net skip of this tile: requested now, needed after GEMM-2
        f32x16 ad[TH];
        if (has_add) {
            __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(a.addend + (size_t)b * H * L), 0, H * L4, 0x00020000);
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ad[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, voff, CH_SOFF(t, r), 0));
        }
#pragma unroll
        for (int m = 0; m < TF; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) u[m][r] = dws_gelu(u[m][r]);

        __builtin_amdgcn_sched_barrier(0);
        CH_STAMP(5)
        // ---- GEMM-2: f[H x 32] = W2 u + b2;  out = x1 + f (+ skip)
        f32x16 f[TH];
#pragma unroll
        for (int m = 0; m < TH; ++m) {
            const float bv = lhi ? 0.f : b2[m * 32 + l31];
            f32x16 z;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = 0.f;
            f[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, one, z, 0, 0, 0);
        }
        {
            constexpr int NKG2 = FFE * H / 8;
            f32x4 af[2][TH];
#pragma unroll
            for (int m = 0; m < TH; ++m) af[0][m] = *reinterpret_cast<const f32x4*>(w2 + ((m * NKG2) * 64 + lane) * 4);
#pragma unroll
            for (int kg = 0; kg < NKG2; ++kg) {
                if (kg + 1 < NKG2) {
#pragma unroll
                    for (int m = 0; m < TH; ++m)
                        af[(kg + 1) & 1][m] = *reinterpret_cast<const f32x4*>(w2 + ((m * NKG2 + kg + 1) * 64 + lane) * 4);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kappa = kg * 4 + j;
#pragma unroll
                    for (int m = 0; m < TH; ++m)
                        f[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kg & 1][m][j], u[kappa >> 4][kappa & 15], f[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        CH_STAMP(6)
        float so = 0.f;
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = (x1[t][r] + mean) + f[t][r];
                if (has_add) v += ad[t][r];
                f[t][r] = v;
                so += v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rO, voff, CH_SOFF(t, r), 0);
            }
        CH_STAMP(7)
        if constexpr (YNEXT) {
            // ---- the next block's S4 input: LN1_next down the columns of the output + its step-embedding projection, which
            // enters as a rank-1 product (A = e column, B = row of ones): one MFMA per row tile puts e[row] into every column
            __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ynext + (size_t)b * H * L), 0, H * L4, 0x00020000);
            const float* eb = a.e_next + (size_t)b * a.e_stride + step_row_off(a.e_step, a.e_tstride);
            const float m2 = xhalf_sum(so) * invH;
            float sv2 = 0.f;
#pragma unroll
            for (int t = 0; t < TH; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    f[t][r] -= m2;
                    sv2 = fmaf(f[t][r], f[t][r], sv2);
                }
            const float al2 = n1_s / sqrtf(xhalf_sum(sv2) * invH);
#pragma unroll
            for (int t = 0; t < TH; ++t) {
                const float ev = lhi ? 0.f : eb[t * 32 + l31];
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                const f32x16 et = __builtin_amdgcn_mfma_f32_32x32x2f32(ev, one, z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float yv = fmaf(al2, f[t][r] + n1_m, et[r]);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), rY, voff, CH_SOFF(t, r), 0);
                }
            }
        }
        CH_STAMP(8)
        ++tile_no;
    }
#undef CH_SOFF
#undef CH_STAMP
}

// DWS_CHAIN_TRACE=1 (tools only): mean shader-clock ticks per phase of the traced tile of every wave, on stderr
template <typename F>
static void chain_trace_launch(int H, int nwg, int waves, S4TailArgs a, hipStream_t s, F launch) {
    static const char* names[9] = {"", "issue g,x loads", "GEMM-o (incl. load wait)", "GLU+res+LN2", "GEMM-1", "GELU (+skip request)",
                                   "GEMM-2", "out stores", "next LN1 + stores"};
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
        if (!t[0] || !t[8]) continue;            // the wave had fewer than two tiles
        for (int i = 1; i < 9; ++i) ph[i] += (double)(t[i] - t[i - 1]);
        life += (double)(t[8] - t[0]);
        ++cnt;
    }
    if (!cnt) return;
    fprintf(stderr, "[chain trace] H=%d L=%d B=%d wgs=%d waves/wg=%d ynext=%d traced waves %zu; mean ticks per phase of a wave's 2nd tile:",
            H, a.L, a.B, nwg, waves, a.ynext ? 1 : 0, cnt);
    for (int i = 1; i < 9; ++i) fprintf(stderr, " %s %.0f |", names[i], ph[i] / cnt);
    fprintf(stderr, " tile %.0f\n", life / cnt);
}

bool s4_tail_chain_supported(int H, int ff) { return ff == 2 && (H == 32 || H == 64); }

template <int H>
static int launch_chain_t(const S4TailArgs& a, hipStream_t s) {
    using T = ChainCfg<H, 2>;
    ProfileScope ps("s4_tail_mfma_chain", s);
    const size_t lds = (size_t)T::LDS_FLOATS * 4;
    static int slots_dev[DWS_MAX_DEVICES] = {};
    int& slots = slots_dev[current_device_slot()];
    if (slots == 0) {
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_chain_kernel<H, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        DWS_HIP(hipFuncSetAttribute((const void*)s4_tail_chain_kernel<H, 2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0, ncu = 0, per_cu = 0;
        DWS_HIP(hipGetDevice(&dev));
        DWS_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        DWS_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, s4_tail_chain_kernel<H, 2, true>, T::THREADS, lds));
        DWS_CHECK(ncu > 0 && per_cu > 0, DWS_ERR_HIP, "s4_tail_chain: occupancy query returned %d x %d", ncu, per_cu);
        slots = ncu * per_cu;
    }
    const int ntiles = a.B * ceil_div(a.L, 32);
    const int grid = std::min(slots, ceil_div(ntiles, T::WAVES));
    static const bool trace = std::getenv("DWS_CHAIN_TRACE") != nullptr;
    if (trace && a.ynext) {
        chain_trace_launch(H, grid, T::WAVES, a, s, [&](const S4TailArgs& at) {
            hipLaunchKernelGGL((s4_tail_chain_kernel<H, 2, true>), dim3(grid), dim3(T::THREADS), lds, s, at);
        });
        return DWS_OK;
    }
    if (a.ynext) hipLaunchKernelGGL((s4_tail_chain_kernel<H, 2, true>), dim3(grid), dim3(T::THREADS), lds, s, a);
    else hipLaunchKernelGGL((s4_tail_chain_kernel<H, 2, false>), dim3(grid), dim3(T::THREADS), lds, s, a);
    return DWS_OK;
}

int launch_s4_tail_chain(int H, const S4TailArgs& a, hipStream_t s) {
    DWS_CHECK(a.Ao_c && a.A1_c && a.A2_c, DWS_ERR_STATE, "s4_tail_chain: chain-ordered weights were not packed");
    if (H == 32) return launch_chain_t<32>(a, s);
    if (H == 64) return launch_chain_t<64>(a, s);
    return set_error(DWS_ERR_UNSUPPORTED, "s4_tail_chain: H=%d not instantiated", H);
}

}  // namespace dws
