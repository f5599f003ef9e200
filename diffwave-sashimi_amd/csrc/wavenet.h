// Launch interface of wavenet_kernels.hip (internal to libdws.so).
#pragma once
#include "dws_common.h"

namespace dws {

constexpr int WN_LAYER_KC = 32;  // channels per staged K chunk of the fused layer kernel
constexpr int WN_BX3_KC = 16;    // same for the bf16x3 kernel (one k-block of 16 per tap)

struct WnLayerArgs {
    const float* x_in;     // [B, C, L]
    float* x_out;          // [B, C, L]   (ping-pong partner of x_in)
    float* skip;           // [B, S, L]   running skip sum
    const float* part_t;   // fc_t(emb) of this layer, element (b, c) at part_t[b*part_t_bstride + c]
    int part_t_bstride;
    const float* A1;       // packed dilated-conv weights (MFMA path)
    const float* A2;       // packed [res; skip] weights   (MFMA path)
    const float* Abt;      // MFMA path: step-embedding correction fragments of this layer [B][2C/32][64][4]
    int abt_bstride;       // floats between two clips' Abt rows (0 in step-table mode: every clip shares the row of step t)
    // step-table mode (the sampler: every clip is at the same step t, `generate.py:50`): Abt / part_t point at row 0 of this
    // layer's [T][...] table built once per sampler run; the kernels add *step_idx * tstride (a scalar load)
    const int* step_idx;   // nullable: device-resident step counter
    int abt_tstride, part_t_tstride;
    const float* Wd;       // folded [2C][C][3]            (generic path)
    const float* Wr;       // folded [C][C]
    const float* Ws;       // folded [S][C]
    const float* bias1;    // [2C]
    const float* bias2;    // [C + S]  (res bias, then skip bias)
    const float* melc;     // nullable: conditioner term [Bm, 2C, L] of this layer
    int mel_bstride;       // 0: broadcast batch-1 mel, 1: per-batch
    float* gate_ws;        // generic path scratch [B, C, L]
    float* hsave;          // nullable (training): pre-gate activations H of this layer [B, 2C, L]
    int B, L, dilation, first_layer, last_layer;
    const float* wscale;   // fp16-split path only: the two power-of-two scales this layer's A1 / A2 were packed with
    unsigned long long* trace;   // nullable (tools only): s_memtime stamps [tile][wave][8] of the Winograd kernel's phases
};

struct WnFinalArgs {
    const float* skip;  // [B, S, L]
    const float* Af;    // packed final_conv[0] weight (MFMA path)
    const float* Wf;    // folded [S][S]              (generic path)
    const float* bf;    // [S]
    const float* Wz;    // [Cout][S]
    const float* bz;    // [Cout]
    float* out;         // [B, Cout, L]
    float* tap;         // nullable: relu(final_conv[0]) [B, S, L]
    float scale;        // sqrt(1 / num_res_layers)
    int B, L, Cout;
    int af_scaled;      // Af already holds Wf * scale (the MFMA kernel then stages the skip tile by LDS-DMA, unscaled)
};

int launch_fold_weight_norm(const float* v, const float* g, float* out, int O, int inner, hipStream_t s);
int launch_permute_dconv(const float* w, float* out, int C, int KC, hipStream_t s);
int launch_pack_a_frag(const float* w, float* out, int M, int K, hipStream_t s);
int launch_pack_a_frag_t(const float* w, float* out, int O, int K, hipStream_t s);   // fragments of W^T from row-major W[O][K]
int launch_step_embed(const float* steps, const float* freq, float* emb, int B, int half, hipStream_t s);
int launch_iota_f32(float* out, int n, hipStream_t s);
int launch_linear_rows(const float* in, const float* W, const float* bias, float* out, int B, int K, int O,
                       int act, hipStream_t s, float* pre_out = nullptr);
int launch_init_conv(const float* audio, const float* W, const float* bias, float* x, int B, int Cin, int C, int L,
                     hipStream_t s);
int launch_wn_bias_tap(const float* Wd_all, const float* part_t, float* Abt, int NL, int B, int C, hipStream_t s);
bool wn_layer_mfma_supported(int C, int S);
int launch_wn_layer_mfma(int C, int S, const WnLayerArgs& a, hipStream_t s);
int launch_wn_layer_generic(int C, int S, const WnLayerArgs& a, hipStream_t s);
bool wn_final_mfma_supported(int S);
// Winograd F(2,3) form of the fused layer (wavenet_wino.hip)
bool wn_layer_wino_supported(int C, int S);
int launch_wino_dconv(const float* w, float* out, int C, hipStream_t s);   // folded [2C][C][3] -> [2C][4C] (G0..G3)
int launch_wn_wino_bias(const float* Wd_all, const float* part_t, float* Abt, int NL, int B, int C, hipStream_t s);
int launch_wn_layer_wino(int C, int S, const WnLayerArgs& a, hipStream_t s);
// bf16x3 path (wavenet_bf16x3.hip)
bool wn_layer_bf16x3_supported(int C, int S);
int launch_wn_layer_bf16x3(int C, int S, const WnLayerArgs& a, hipStream_t s);
int launch_pack_a_bf16x3(const float* w, void* out, int M, int K, hipStream_t s);
int launch_wn_bias_tap_bf16(const float* Wd_all, const float* part_t, const float* bias1_all, void* Abt, int NL, int B, int C,
                            hipStream_t s);
// split-precision path (wavenet_bx6.hip): Winograd F(2,3) layer on the bf16 / fp16 matrix cores.
//   WN_SPLIT_BF16X6: 3 bf16 terms per operand, six products (exact split: fp32-faithful)
//   WN_SPLIT_F16X3:  2 fp16 terms per operand, three products, power-of-two operand scaling (22 bits per operand)
enum { WN_SPLIT_BF16X6 = 0, WN_SPLIT_F16X3 = 1 };
inline int wn_split_terms(int split) { return split == WN_SPLIT_F16X3 ? 2 : 3; }
bool wn_layer_bx6_supported(int C, int S);
int launch_wn_layer_bx6(int C, int S, const WnLayerArgs& a, int split, hipStream_t s);
int launch_weight_scale(const float* w, size_t n, const float* bias, int nb, const float* bias_b, int nb_b, float* out, hipStream_t s);   // power of two bringing max(|w|, |bias|/1024) into [1, 2)
int launch_pack_a1_bx6(const float* w, void* out, int C, int split, const float* scale, hipStream_t s);        // folded [2C][C][3] -> G0..G3 fragments
int launch_pack_a_bx6(const float* w, void* out, int M, int K, int split, const float* scale, hipStream_t s);  // row-major [M][K] -> fragments
int launch_gemm_bx6(const float* A, const float* B, float* C, int M, int N, int K, int split, float sa, float sb, hipStream_t s);
int launch_wn_final(int S, const WnFinalArgs& a, hipStream_t s);

}  // namespace dws
