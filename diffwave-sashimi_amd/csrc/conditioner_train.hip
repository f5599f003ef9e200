// Adjoint of the mel conditioner of one layer (`wavenet.py:98-111`, `sashimi.py:160-175`), training path:
//   melc = Wc . lrelu(up1(lrelu(up0(mel))))[:, :, :L] + bc
// given d melc [B][O][L].  The upsampled mels are recomputed here (they are tiny next to the activations);
// the 1x1 conv's two GEMMs run on the MFMA adjoint kernels (80 mel bands padded to 96 rows for the data gradient).
#include "conditioner.h"
#include "wavenet.h"
#include "sashimi_train.h"
#include "wavenet_backward.h"

namespace dws {

int conditioner_backward(CondTrainWs& ws, const float* mel, int B, int MB, int Tmel, int s0, int s1, const float* W0f,
                         const float* b0, const float* W1f, const float* b1, const float* Wcf, int O, int L,
                         const float* dmelc, float* gW0f, float* gb0, float* gW1f, float* gb1, float* gWcf, hipStream_t s) {
    const int T0 = mel_upsampled_len(Tmel, s0), T1 = mel_upsampled_len(T0, s1);
    const int MP = ceil_div(MB, 32) * 32;   // mel bands padded to whole MFMA tiles
    DWS_TRY(ws.u0.ensure((size_t)B * MB * T0 * 4));
    DWS_TRY(ws.u1.ensure((size_t)B * MB * T1 * 4));
    DWS_TRY(ws.du0.ensure((size_t)B * MB * T0 * 4));
    DWS_TRY(ws.du1.ensure((size_t)B * MP * L * 4));
    DWS_TRY(ws.tmp.ensure((size_t)MP * O * 4));
    DWS_TRY(ws.AT.ensure((size_t)MP * O * 4));
    DWS_TRY(launch_mel_upsample(mel, W0f, b0, ws.u0.f(), B, MB, Tmel, T0, s0, 0.4f, s));
    DWS_TRY(launch_mel_upsample(ws.u0.f(), W1f, b1, ws.u1.f(), B, MB, T0, T1, s1, 0.4f, s));
    // d Wc[o, k] = sum_{b, l < L} dmelc[b, o, l] * u1[b, k, l]
    {
        WgradArgs w{};
        w.dY = dmelc; w.X = ws.u1.f(); w.xL = T1; w.B = B; w.O = O; w.C = MB; w.L = L; w.dil = 1;
        w.nsplit = wgrad_mfma_nsplit(B, O, MB, L, 1);
        DWS_TRY(ws.wpart.ensure((size_t)w.nsplit * O * MB * 4));
        w.partial = ws.wpart.f();
        DWS_TRY(launch_wgrad_mfma(w, 1, 1.f, gWcf, s));
    }
    // d u1[b, k, l] = sum_o Wc[o, k] * dmelc[b, o, l]   (rows k >= MB of the padded result are zero)
    {
        DWS_HIP(hipMemsetAsync(ws.tmp.p, 0, (size_t)MP * O * 4, s));
        DWS_TRY(launch_tapconv_pack_transposed(Wcf, ws.tmp.f(), O, MB, 1, O, 0, 1.f, s));
        if (O % 32 != 0) {   // test-sized models: plain-FMA GEMM on the row-major transpose [MP][O]
            GemmRowsArgs g{};
            g.W = ws.tmp.f(); g.src = dmelc; g.out = ws.du1.f(); g.B = B; g.M = MP; g.K = O; g.L = L; g.epi = 2;
            DWS_TRY(launch_gemm_rows_generic(g, s));
        } else {
        DWS_TRY(launch_pack_a_frag(ws.tmp.f(), ws.AT.f(), MP, O, s));
        TapConvArgs q{};
        q.src0 = dmelc; q.K0 = O; q.A = ws.AT.f(); q.nkg_total = O / 8; q.M = MP; q.T = 1; q.dil = 1; q.sign = 1; q.epi = 2;
        q.out = ws.du1.f(); q.B = B; q.L = L;
        DWS_TRY(launch_tapconv_mfma(q, s));
        }
    }
    DWS_TRY(launch_mel_upsample_bwd(ws.u0.f(), ws.u1.f(), ws.du1.f(), W1f, ws.du0.f(), gW1f, gb1, B, MB, T0, T1, s1, L, MP * L,
                                    L, 0.4f, s));
    DWS_TRY(launch_mel_upsample_bwd(mel, ws.u0.f(), ws.du0.f(), W0f, nullptr, gW0f, gb0, B, MB, Tmel, T0, s0, T0, MB * T0, T0,
                                    0.4f, s));
    return DWS_OK;
}

}  // namespace dws
