// WaveNet backbone behind the C ABI: parameter ingest in the reference's
// state_dict layout, one-time weight folding/packing, per-step forward.
// Mirrors `models/wavenet.py:168-210` (WaveNet), `:124-165` (Residual_group).
#include <cmath>

#include "conditioner.h"
#include "model.h"
#include "wavenet.h"

namespace dws {

struct WaveNetModel : dws_model {
    int Cin, Cout, C, S, NL, cycle, Ein, Emid, Eout, MB;
    bool cond, mfma_layer, mfma_final;
    bool bf16x3 = false;             // precision option (see include/dws.h)

    // folded / packed weights
    DevBuf Wi;                       // init conv [C][Cin]
    std::vector<DevBuf> A1, Wrs, A2, bias2;       // per layer
    DevBuf Wd_all;                   // folded dilated-conv weights of all layers [NL][2C][C][3]
    DevBuf Abt;                      // per-step embedding correction fragments [NL][B][2C/32][64][4]
    DevBuf Wt_all, bt_all;           // stacked fc_t [NL*C][Eout], [NL*C]
    DevBuf Wf, Af;                   // final_conv[0]
    DevBuf freq;                     // embedding frequencies [Ein/2]
    DevBuf tmp_pack;                 // scratch for permute -> pack
    // conditioner
    std::vector<DevBuf> melW0, melW1, melWc;  // folded upsampler kernels + mel_conv weight per layer
    DevBuf melc;                     // [NL][Bm][2C][L]
    DevBuf mel_u0, mel_u1;           // upsample scratch
    int64_t melBm = 0;               // 0 = no condition installed
    // workspace
    DevBuf x0, x1, skip, gate, emb, h1, h2, part_t, scratch_out;

    explicit WaveNetModel(const dws_model_desc& dd) {
        d = dd;
        Cin = d.in_channels; Cout = d.out_channels; C = d.res_channels; S = d.skip_channels;
        NL = d.num_res_layers; cycle = d.dilation_cycle;
        Ein = d.diffusion_step_embed_dim_in; Emid = d.diffusion_step_embed_dim_mid; Eout = d.diffusion_step_embed_dim_out;
        MB = d.mel_bands;
        cond = !d.unconditional;
        mfma_layer = wn_layer_mfma_supported(C, S);
        mfma_final = wn_final_mfma_supported(S);
        auto wn = [&](const std::string& p, std::vector<int64_t> vshape) {
            std::vector<int64_t> g(vshape.size(), 1);
            g[0] = vshape[0];
            add_param(p + ".bias", {vshape[0]});
            add_param(p + ".weight_g", g);
            add_param(p + ".weight_v", vshape);
        };
        wn("init_conv.0.conv", {C, Cin, 1});
        add_param("residual_layer.fc_t1.weight", {Emid, Ein});
        add_param("residual_layer.fc_t1.bias", {Emid});
        add_param("residual_layer.fc_t2.weight", {Eout, Emid});
        add_param("residual_layer.fc_t2.bias", {Eout});
        for (int n = 0; n < NL; ++n) {
            const std::string p = "residual_layer.residual_blocks." + std::to_string(n);
            add_param(p + ".fc_t.weight", {C, Eout});
            add_param(p + ".fc_t.bias", {C});
            wn(p + ".dilated_conv_layer.conv", {2 * C, C, 3});
            if (cond) {
                for (int i = 0; i < 2; ++i) {
                    const std::string u = p + ".upsample_conv2d." + std::to_string(i);
                    const int s = d.mel_upsample[i];
                    // ConvTranspose2d weight is [in=1, out=1, 3, 2s]; bias [1]
                    add_param(u + ".bias", {1});
                    add_param(u + ".weight_g", {1, 1, 1, 1});
                    add_param(u + ".weight_v", {1, 1, 3, 2 * s});
                }
                wn(p + ".mel_conv.conv", {2 * C, MB, 1});
            }
            wn(p + ".res_conv", {C, C, 1});
            wn(p + ".skip_conv", {S, C, 1});
        }
        wn("final_conv.0.conv", {S, S, 1});
        add_param("final_conv.2.conv.weight", {Cout, S, 1});
        add_param("final_conv.2.conv.bias", {Cout});
        A1.resize(NL); Wrs.resize(NL); A2.resize(NL); bias2.resize(NL);
        if (cond) { melW0.resize(NL); melW1.resize(NL); melWc.resize(NL); }
    }

    int set_option(const std::string& key, const std::string& value) override {
        if (key == "precision") {
            if (value == "f32") { bf16x3 = false; dirty = true; return DWS_OK; }
            if (value == "bf16x3") {
                DWS_CHECK(wn_layer_bf16x3_supported(C, S), DWS_ERR_UNSUPPORTED,
                          "precision=bf16x3 is not built for (res_channels=%d, skip_channels=%d)", C, S);
                bf16x3 = true; dirty = true;
                return DWS_OK;
            }
        }
        return dws_model::set_option(key, value);
    }

    float* Wd(int n) const { return Wd_all.f() + (size_t)n * 2 * C * C * 3; }

    int fold(const std::string& p, float* out, int O, int inner, hipStream_t s) {
        return launch_fold_weight_norm(P(p + ".weight_v"), P(p + ".weight_g"), out, O, inner, s);
    }

    int commit(hipStream_t s) override {
        DWS_TRY(Wi.ensure((size_t)C * Cin * 4));
        DWS_TRY(fold("init_conv.0.conv", Wi.f(), C, Cin, s));
        DWS_TRY(Wt_all.ensure((size_t)NL * C * Eout * 4));
        DWS_TRY(bt_all.ensure((size_t)NL * C * 4));
        DWS_TRY(Wd_all.ensure((size_t)NL * 2 * C * C * 3 * 4));
        if (mfma_layer) DWS_TRY(tmp_pack.ensure((size_t)2 * C * 3 * C * 4));
        for (int n = 0; n < NL; ++n) {
            const std::string p = "residual_layer.residual_blocks." + std::to_string(n);
            DWS_HIP(hipMemcpyAsync(Wt_all.f() + (size_t)n * C * Eout, P(p + ".fc_t.weight"), (size_t)C * Eout * 4,
                                   hipMemcpyDeviceToDevice, s));
            DWS_HIP(hipMemcpyAsync(bt_all.f() + (size_t)n * C, P(p + ".fc_t.bias"), (size_t)C * 4,
                                   hipMemcpyDeviceToDevice, s));
            DWS_TRY(fold(p + ".dilated_conv_layer.conv", Wd(n), 2 * C, C * 3, s));
            DWS_TRY(Wrs[n].ensure((size_t)(C + S) * C * 4));
            DWS_TRY(fold(p + ".res_conv", Wrs[n].f(), C, C, s));
            DWS_TRY(fold(p + ".skip_conv", Wrs[n].f() + (size_t)C * C, S, C, s));
            DWS_TRY(bias2[n].ensure((size_t)(C + S) * 4));
            DWS_HIP(hipMemcpyAsync(bias2[n].f(), P(p + ".res_conv.bias"), (size_t)C * 4, hipMemcpyDeviceToDevice, s));
            DWS_HIP(hipMemcpyAsync(bias2[n].f() + C, P(p + ".skip_conv.bias"), (size_t)S * 4, hipMemcpyDeviceToDevice, s));
            if (mfma_layer) {
                DWS_TRY(A1[n].ensure((size_t)2 * C * 3 * C * 4));
                DWS_TRY(launch_permute_dconv(Wd(n), tmp_pack.f(), C, WN_LAYER_KC, s));
                DWS_TRY(A2[n].ensure((size_t)(C + S) * C * 4));
                if (bf16x3) {
                    DWS_TRY(launch_pack_a_bf16x3(tmp_pack.f(), A1[n].p, 2 * C, 3 * C, s));
                    DWS_TRY(launch_pack_a_bf16x3(Wrs[n].f(), A2[n].p, C + S, C, s));
                } else {
                    DWS_TRY(launch_pack_a_frag(tmp_pack.f(), A1[n].f(), 2 * C, 3 * C, s));
                    DWS_TRY(launch_pack_a_frag(Wrs[n].f(), A2[n].f(), C + S, C, s));
                }
            }
            if (cond) {
                for (int i = 0; i < 2; ++i) {
                    const int sc = d.mel_upsample[i];
                    DevBuf& w = (i == 0) ? melW0[n] : melW1[n];
                    DWS_TRY(w.ensure((size_t)3 * 2 * sc * 4));
                    DWS_TRY(fold(p + ".upsample_conv2d." + std::to_string(i), w.f(), 1, 3 * 2 * sc, s));
                }
                DWS_TRY(melWc[n].ensure((size_t)2 * C * MB * 4));
                DWS_TRY(fold(p + ".mel_conv.conv", melWc[n].f(), 2 * C, MB, s));
            }
        }
        DWS_TRY(Wf.ensure((size_t)S * S * 4));
        DWS_TRY(fold("final_conv.0.conv", Wf.f(), S, S, s));
        if (mfma_final) {
            DWS_TRY(Af.ensure((size_t)S * S * 4));
            DWS_TRY(launch_pack_a_frag(Wf.f(), Af.f(), S, S, s));
        }
        // embedding frequencies: exp(float(i) * float(-ln(1e4)/(half-1)))  (`models/utils.py:22-23`)
        {
            const int half = Ein / 2;
            std::vector<float> f(half);
            const float e = (float)(-(std::log(10000.0) / (half - 1)));
            for (int i = 0; i < half; ++i) f[i] = (float)std::exp((double)((float)i * e));
            DWS_TRY(freq.ensure((size_t)half * 4));
            DWS_HIP(hipMemcpyAsync(freq.p, f.data(), (size_t)half * 4, hipMemcpyHostToDevice, s));
            DWS_HIP(hipStreamSynchronize(s));  // f goes out of scope
        }
        if (Abt.p) DWS_HIP(hipMemsetAsync(Abt.p, 0, Abt.bytes, s));  // correction layout depends on the precision
        dirty = false;
        melBm = 0;  // conditioner terms depend on the weights: must be re-installed
        return DWS_OK;
    }

    int prepare(int64_t nB, int64_t nL) override {
        DWS_CHECK(nB > 0 && nL > 0, DWS_ERR_INVALID, "prepare: B=%lld L=%lld", (long long)nB, (long long)nL);
        DWS_CHECK(nB * nL * (int64_t)std::max(2 * C, S) < (int64_t)1 << 40, DWS_ERR_UNSUPPORTED, "workspace too large");
        if (nB != B || nL != L) { drop_graph(); melBm = 0; }
        B = nB; L = nL;
        const size_t act = (size_t)B * C * L * 4;
        DWS_TRY(x0.ensure(act));
        DWS_TRY(x1.ensure(act));
        DWS_TRY(skip.ensure((size_t)B * S * L * 4));
        if (!mfma_layer) DWS_TRY(gate.ensure(act));
        DWS_TRY(emb.ensure((size_t)B * Ein * 4));
        DWS_TRY(h1.ensure((size_t)B * Emid * 4));
        DWS_TRY(h2.ensure((size_t)B * Eout * 4));
        DWS_TRY(part_t.ensure((size_t)B * NL * C * 4));
        if (mfma_layer) {
            const size_t n = (size_t)NL * B * (2 * C / 32) * 2048;  // sized for the larger (bf16 hi+lo, k-block 16) form
            DWS_TRY(Abt.ensure(n));
            DWS_HIP(hipMemset(Abt.p, 0, n));  // the unused k entries of the correction k-group stay zero
        }
        return DWS_OK;
    }

    int set_condition(const float* mel, int64_t Bm, int64_t Tmel, hipStream_t s) override {
        if (mel == nullptr) { melBm = 0; return DWS_OK; }
        DWS_CHECK(cond, DWS_ERR_INVALID, "set_condition on an unconditional model (`wavenet.py:99`)");
        DWS_CHECK(B > 0, DWS_ERR_STATE, "set_condition before prepare");
        DWS_CHECK(Bm == 1 || Bm == B, DWS_ERR_INVALID, "mel batch %lld must be 1 or B=%lld", (long long)Bm, (long long)B);
        if (dirty) DWS_TRY(commit(s));
        const int s0 = d.mel_upsample[0], s1 = d.mel_upsample[1];
        const int T0 = mel_upsampled_len((int)Tmel, s0), T1 = mel_upsampled_len(T0, s1);
        DWS_CHECK(T1 >= L, DWS_ERR_INVALID, "upsampled mel length %d < L=%lld (`wavenet.py:105`)", T1, (long long)L);
        DWS_TRY(mel_u0.ensure((size_t)Bm * MB * T0 * 4));
        DWS_TRY(mel_u1.ensure((size_t)Bm * MB * T1 * 4));
        DWS_TRY(melc.ensure((size_t)NL * Bm * 2 * C * L * 4));
        for (int n = 0; n < NL; ++n) {
            const std::string p = "residual_layer.residual_blocks." + std::to_string(n);
            DWS_TRY(launch_mel_upsample(mel, melW0[n].f(), P(p + ".upsample_conv2d.0.bias"), mel_u0.f(), (int)Bm, MB,
                                        (int)Tmel, T0, s0, 0.4f, s));
            DWS_TRY(launch_mel_upsample(mel_u0.f(), melW1[n].f(), P(p + ".upsample_conv2d.1.bias"), mel_u1.f(), (int)Bm,
                                        MB, T0, T1, s1, 0.4f, s));
            DWS_TRY(launch_conv1x1_trunc(mel_u1.f(), melWc[n].f(), P(p + ".mel_conv.conv.bias"),
                                         melc.f() + (size_t)n * Bm * 2 * C * L, (int)Bm, MB, 2 * C, T1, (int)L, s));
        }
        melBm = Bm;
        return DWS_OK;
    }

    int final_stage(float* out, float* tap, hipStream_t s) {
        WnFinalArgs f{};
        f.skip = skip.f(); f.Af = Af.f(); f.Wf = Wf.f(); f.bf = P("final_conv.0.conv.bias");
        f.Wz = P("final_conv.2.conv.weight"); f.bz = P("final_conv.2.conv.bias");
        f.out = out; f.tap = tap; f.scale = (float)std::sqrt(1.0 / NL);
        f.B = (int)B; f.L = (int)L; f.Cout = Cout;
        return launch_wn_final(S, f, s);
    }

    int forward(const float* audio, const float* steps, float* out, hipStream_t s) override {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "forward before prepare");
        if (dirty) DWS_TRY(commit(s));
        DWS_TRY(launch_init_conv(audio, Wi.f(), P("init_conv.0.conv.bias"), x0.f(), (int)B, Cin, C, (int)L, s));
        DWS_TRY(launch_step_embed(steps, freq.f(), emb.f(), (int)B, Ein / 2, s));
        DWS_TRY(launch_linear_rows(emb.f(), P("residual_layer.fc_t1.weight"), P("residual_layer.fc_t1.bias"), h1.f(),
                                   (int)B, Ein, Emid, 1, s));
        DWS_TRY(launch_linear_rows(h1.f(), P("residual_layer.fc_t2.weight"), P("residual_layer.fc_t2.bias"), h2.f(),
                                   (int)B, Emid, Eout, 1, s));
        DWS_TRY(launch_linear_rows(h2.f(), Wt_all.f(), bt_all.f(), part_t.f(), (int)B, Eout, NL * C, 0, s));
        if (mfma_layer && bf16x3) DWS_TRY(launch_wn_bias_tap_bf16(Wd_all.f(), part_t.f(), Abt.p, NL, (int)B, C, s));
        else if (mfma_layer) DWS_TRY(launch_wn_bias_tap(Wd_all.f(), part_t.f(), Abt.f(), NL, (int)B, C, s));
        for (int n = 0; n < NL; ++n) {
            const std::string p = "residual_layer.residual_blocks." + std::to_string(n);
            WnLayerArgs a{};
            a.x_in = (n & 1) ? x1.f() : x0.f();
            a.x_out = (n & 1) ? x0.f() : x1.f();
            a.skip = skip.f();
            a.part_t = part_t.f() + (size_t)n * C;
            a.part_t_bstride = NL * C;
            a.A1 = A1[n].f(); a.A2 = A2[n].f();
            a.Abt = mfma_layer ? Abt.f() + (size_t)n * B * (2 * C / 32) * (bf16x3 ? 512 : 256) : nullptr;
            a.Wd = Wd(n); a.Wr = Wrs[n].f(); a.Ws = Wrs[n].f() + (size_t)C * C;
            a.bias1 = P(p + ".dilated_conv_layer.conv.bias");
            a.bias2 = bias2[n].f();
            a.melc = melBm ? melc.f() + (size_t)n * melBm * 2 * C * L : nullptr;
            a.mel_bstride = (melBm > 1) ? 1 : 0;
            a.gate_ws = gate.f();
            a.B = (int)B; a.L = (int)L;
            a.dilation = 1 << (n % cycle);
            a.first_layer = (n == 0); a.last_layer = (n == NL - 1);
            if (mfma_layer && bf16x3) DWS_TRY(launch_wn_layer_bf16x3(C, S, a, s));
            else if (mfma_layer) DWS_TRY(launch_wn_layer_mfma(C, S, a, s));
            else DWS_TRY(launch_wn_layer_generic(C, S, a, s));
        }
        DWS_TRY(final_stage(out, nullptr, s));
        DWS_HIP(hipGetLastError());
        return DWS_OK;
    }

    int read_tap(const char* tap, float* dst, int64_t capacity, hipStream_t s) override {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "read_tap before prepare/forward");
        const std::string t(tap);
        if (t == "skip") {
            DWS_CHECK(capacity >= B * S * L, DWS_ERR_INVALID, "tap buffer too small");
            DWS_HIP(hipMemcpyAsync(dst, skip.p, (size_t)B * S * L * 4, hipMemcpyDeviceToDevice, s));
            return DWS_OK;
        }
        if (t == "x") {  // output of the second-to-last layer == input of the last one
            DWS_CHECK(capacity >= B * C * L, DWS_ERR_INVALID, "tap buffer too small");
            const float* src = ((NL - 1) & 1) ? x1.f() : x0.f();
            DWS_HIP(hipMemcpyAsync(dst, src, (size_t)B * C * L * 4, hipMemcpyDeviceToDevice, s));
            return DWS_OK;
        }
        if (t == "pre_final") {
            DWS_CHECK(capacity >= B * S * L, DWS_ERR_INVALID, "tap buffer too small");
            DWS_TRY(scratch_out.ensure((size_t)B * Cout * L * 4));
            return final_stage(scratch_out.f(), dst, s);
        }
        return set_error(DWS_ERR_INVALID, "unknown tap '%s'", tap);
    }
};

dws_model* make_wavenet(const dws_model_desc& d) { return new WaveNetModel(d); }

}  // namespace dws
