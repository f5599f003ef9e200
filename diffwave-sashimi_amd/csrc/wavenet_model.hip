// WaveNet backbone behind the C ABI: parameter ingest in the reference's
// state_dict layout, one-time weight folding/packing, per-step forward.
// Mirrors `models/wavenet.py:168-210` (WaveNet), `:124-165` (Residual_group).
#include <cmath>

#include "conditioner.h"
#include "model.h"
#include "wavenet.h"
#include "wavenet_backward.h"

namespace dws {

struct WaveNetModel : dws_model {
    int Cin, Cout, C, S, NL, cycle, Ein, Emid, Eout, MB;
    bool cond, mfma_layer, mfma_final;
    bool bf16x3 = false;             // precision option (see include/dws.h)
    bool bf16x6 = false;             // precision option: 3-term split, six products, Winograd form (wavenet_bx6.hip)
    bool f16x3 = false;              // precision option: 2-term fp16 split, three products, scaled operands (same kernel)
    DevBuf wscale;                   // f16x3: [NL][2] power-of-two scales of the packed A1 / A2
    bool wino_opt = true;            // conv_algo option: Winograd F(2,3) along the dilation stride (f32 path) or direct
    // the Winograd launcher addresses a clip's [C][L] tensor through one 32-bit buffer descriptor: over-long clips
    // (L >= ~2.1 M samples at C = 256) stay on the direct kernel, which has no such bound.  prepare() marks the model
    // dirty when the answer flips (the A1 / Abt layouts follow the choice).
    bool wino_fits(int64_t nL) const {
        return (int64_t)std::max(C, S) * nL * 4 < ((int64_t)1 << 31) && nL + 4 * ((int64_t)1 << (cycle - 1)) < ((int64_t)1 << 28);
    }
    bool wino() const { return wino_opt && mfma_layer && !bf16x3 && !bf16x6 && !f16x3 && wn_layer_wino_supported(C, S) && (L == 0 || wino_fits(L)); }
    // the step-embedding correction rows in the Winograd layout [4][2C] (both Winograd layer kernels read it)
    bool wino_rows() const { return wino() || bf16x6 || f16x3; }

    // folded / packed weights
    DevBuf Wi;                       // init conv [C][Cin]
    std::vector<DevBuf> A1, Wrs, A2, bias2;       // per layer
    DevBuf Wd_all;                   // folded dilated-conv weights of all layers [NL][2C][C][3]
    DevBuf Abt;                      // per-step embedding correction fragments [NL][B][2C/32][64][4]
    DevBuf Wt_all, bt_all;           // stacked fc_t [NL*C][Eout], [NL*C]
    DevBuf b1_all;                   // stacked dilated-conv biases [NL][2C] (bf16x3: folded into the correction rows)
    DevBuf Wf, Af;                   // final_conv[0]
    DevBuf freq;                     // embedding frequencies [Ein/2]
    CopyBatch stack_params, unstack_fc_t;   // per-layer tensors <-> their stacked buffers, one launch each
    bool freq_ready = false;
    DevBuf tmp_pack;                 // scratch for permute -> pack
    // conditioner
    std::vector<DevBuf> melW0, melW1, melWc;  // folded upsampler kernels + mel_conv weight per layer
    DevBuf melc;                     // [NL][Bm][2C][L]
    DevBuf mel_u0, mel_u1;           // upsample scratch
    int64_t melBm = 0;               // 0 = no condition installed
    DevBuf mel_in;                   // copy of the installed mel [Bm][MB][Tmel] (the conditioner adjoint needs it)
    int mel_T = 0;
    CondTrainWs cws;
    DevBuf gW0f, gW1f, gWcf;         // folded-weight gradients of one layer's conditioner
    // workspace
    DevBuf x0, x1, skip, gate, emb, h1, h2, part_t, scratch_out;
    // training workspace: per-layer inputs and pre-gate activations, saved pre-activations of the
    // embedding MLP, gradient scratch
    std::vector<DevBuf> tx, tH;
    std::vector<DevBuf> ATd, ATg;    // transposed-weight A fragments of the adjoint GEMMs (training only)
    std::vector<DevBuf> ATw;         // the dilated conv's adjoint in Winograd F(2,3) form (wavenet_backward_wino.hip)
    DevBuf ATf, wpart, bpart;               // same for final_conv[0]; split-N partials of the weight gradients
    bool mfma_bwd = false;
    uint64_t commit_version = 0, bwd_pack_version = ~0ull;
    DevBuf ty, ta1, ta2, dxa, dxb, dskip, dgb, dHb, dresb, dyb, dWfold, dpt, dh2, dh1, demb, dWt_all, dbt_all;
    bool trained_fwd = false;

    explicit WaveNetModel(const dws_model_desc& dd) {
        d = dd;
        Cin = d.in_channels; Cout = d.out_channels; C = d.res_channels; S = d.skip_channels;
        NL = d.num_res_layers; cycle = d.dilation_cycle;
        Ein = d.diffusion_step_embed_dim_in; Emid = d.diffusion_step_embed_dim_mid; Eout = d.diffusion_step_embed_dim_out;
        MB = d.mel_bands;
        cond = !d.unconditional;
        mfma_layer = wn_layer_mfma_supported(C, S);
        mfma_bwd = tapconv_mfma_supported(C, S, C, 1) && tapconv_mfma_supported(C, 2 * C, 0, 3) &&
                   tapconv_mfma_supported(S, S, 0, 1) && std::getenv("DWS_WAVENET_GENERIC_BWD") == nullptr;
        mfma_final = wn_final_mfma_supported(S);
        if (std::getenv("DWS_WN_DIRECT")) wino_opt = false;
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
            if (value == "f32") { bf16x3 = bf16x6 = f16x3 = false; dirty = true; trained_fwd = false; return DWS_OK; }
            if (value == "bf16x3") {
                DWS_CHECK(wn_layer_bf16x3_supported(C, S), DWS_ERR_UNSUPPORTED,
                          "precision=bf16x3 is not built for (res_channels=%d, skip_channels=%d)", C, S);
                bf16x3 = true; bf16x6 = f16x3 = false; dirty = true; trained_fwd = false;
                return DWS_OK;
            }
            if (value == "bf16x6") {
                DWS_CHECK(mfma_layer && wn_layer_bx6_supported(C, S), DWS_ERR_UNSUPPORTED,
                          "precision=bf16x6 is not built for (res_channels=%d, skip_channels=%d)", C, S);
                bf16x6 = true; bf16x3 = f16x3 = false; dirty = true; trained_fwd = false;
                return DWS_OK;
            }
            if (value == "f16x3") {
                DWS_CHECK(mfma_layer && wn_layer_bx6_supported(C, S), DWS_ERR_UNSUPPORTED,
                          "precision=f16x3 is not built for (res_channels=%d, skip_channels=%d)", C, S);
                f16x3 = true; bf16x3 = bf16x6 = false; dirty = true; trained_fwd = false;
                return DWS_OK;
            }
        }
        if (key == "conv_algo") {
            // a pending backward was packed (pack_bwd) and saved for the other algorithm: it may not run against this one
            if (value == "winograd") { wino_opt = true; dirty = true; trained_fwd = false; return DWS_OK; }
            if (value == "direct") { wino_opt = false; dirty = true; trained_fwd = false; return DWS_OK; }
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
        DWS_TRY(b1_all.ensure((size_t)NL * 2 * C * 4));
        DWS_TRY(Wd_all.ensure((size_t)NL * 2 * C * C * 3 * 4));
        if (mfma_layer) DWS_TRY(tmp_pack.ensure((size_t)2 * C * 4 * C * 4));
        stack_params.begin();
        for (int n = 0; n < NL; ++n) {
            const std::string p = "residual_layer.residual_blocks." + std::to_string(n);
            stack_params.add(P(p + ".fc_t.weight"), Wt_all.f() + (size_t)n * C * Eout, (size_t)C * Eout);
            stack_params.add(P(p + ".fc_t.bias"), bt_all.f() + (size_t)n * C, (size_t)C);
            stack_params.add(P(p + ".dilated_conv_layer.conv.bias"), b1_all.f() + (size_t)n * 2 * C, (size_t)2 * C);
            DWS_TRY(fold(p + ".dilated_conv_layer.conv", Wd(n), 2 * C, C * 3, s));
            DWS_TRY(Wrs[n].ensure((size_t)(C + S) * C * 4));
            DWS_TRY(fold(p + ".res_conv", Wrs[n].f(), C, C, s));
            DWS_TRY(fold(p + ".skip_conv", Wrs[n].f() + (size_t)C * C, S, C, s));
            DWS_TRY(bias2[n].ensure((size_t)(C + S) * 4));
            stack_params.add(P(p + ".res_conv.bias"), bias2[n].f(), (size_t)C);
            stack_params.add(P(p + ".skip_conv.bias"), bias2[n].f() + C, (size_t)S);
            if (mfma_layer && (bf16x6 || f16x3)) {   // 3 bf16 / 2 fp16 terms per weight, packed straight from the folded weights
                const int split = f16x3 ? WN_SPLIT_F16X3 : WN_SPLIT_BF16X6;
                const size_t tb = 2 * (size_t)wn_split_terms(split);
                float* sc = nullptr;
                if (f16x3) {
                    DWS_TRY(wscale.ensure((size_t)NL * 2 * 4));
                    sc = wscale.f() + 2 * n;
                    // (the raw parameter buffers: the stacked copies b1_all / bias2 are filled later in this commit)
                    DWS_TRY(launch_weight_scale(Wd(n), (size_t)2 * C * C * 3, P(p + ".dilated_conv_layer.conv.bias"), 2 * C, nullptr, 0, sc, s));
                    DWS_TRY(launch_weight_scale(Wrs[n].f(), (size_t)(C + S) * C, P(p + ".res_conv.bias"), C, P(p + ".skip_conv.bias"), S,
                                                sc + 1, s));
                }
                DWS_TRY(A1[n].ensure((size_t)2 * C * 4 * C * tb));
                DWS_TRY(A2[n].ensure((size_t)(C + S) * C * tb));
                DWS_TRY(launch_pack_a1_bx6(Wd(n), A1[n].p, C, split, sc, s));
                DWS_TRY(launch_pack_a_bx6(Wrs[n].f(), A2[n].p, C + S, C, split, f16x3 ? sc + 1 : nullptr, s));
            } else if (mfma_layer) {
                DWS_TRY(A1[n].ensure((size_t)2 * C * 4 * C * 4));
                if (wino()) DWS_TRY(launch_wino_dconv(Wd(n), tmp_pack.f(), C, s));
                else DWS_TRY(launch_permute_dconv(Wd(n), tmp_pack.f(), C, bf16x3 ? WN_BX3_KC : WN_LAYER_KC, s));
                DWS_TRY(A2[n].ensure((size_t)(C + S) * C * 4));
                if (wino()) {
                    DWS_TRY(launch_pack_a_frag(tmp_pack.f(), A1[n].f(), 2 * C, 4 * C, s));
                    DWS_TRY(launch_pack_a_frag(Wrs[n].f(), A2[n].f(), C + S, C, s));
                } else if (bf16x3) {
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
        if (mfma_final) {   // the packed copy carries the 1/sqrt(n_layers) of `wavenet.py:165` (the kernel's skip tile arrives by
                            // LDS-DMA, which cannot scale); Wf itself stays unscaled for the generic kernel and the adjoints
            DWS_TRY(Af.ensure((size_t)S * S * 4));
            DWS_TRY(tmp_pack.ensure((size_t)std::max(2 * C * 4 * C, S * S) * 4));
            DWS_TRY(launch_scale(Wf.f(), tmp_pack.f(), (float)std::sqrt(1.0 / NL), (size_t)S * S, s));
            DWS_TRY(launch_pack_a_frag(tmp_pack.f(), Af.f(), S, S, s));
        }
        // embedding frequencies: exp(float(i) * float(-ln(1e4)/(half-1)))  (`models/utils.py:22-23`)
        DWS_TRY(stack_params.run(s));
        if (!freq_ready) {   // depends on the embedding width only: uploaded (and waited for) once, not on every commit --
                             // a training step commits once, and a blocking wait there keeps the host from running ahead of the GPU
            const int half = Ein / 2;
            std::vector<float> f(half);
            const float e = (float)(-(std::log(10000.0) / (half - 1)));
            for (int i = 0; i < half; ++i) f[i] = (float)std::exp((double)((float)i * e));
            DWS_TRY(freq.ensure((size_t)half * 4));
            DWS_HIP(hipMemcpyAsync(freq.p, f.data(), (size_t)half * 4, hipMemcpyHostToDevice, s));
            DWS_HIP(hipStreamSynchronize(s));
            freq_ready = true;  // f goes out of scope
        }
        if (Abt.p) DWS_HIP(hipMemsetAsync(Abt.p, 0, Abt.bytes, s));  // correction layout depends on the precision
        dirty = false;
        ++commit_version;
        melBm = 0;  // conditioner terms depend on the weights: must be re-installed
        return DWS_OK;
    }

    // Transposed weights of the adjoint GEMMs in A-fragment order (rebuilt after every commit).
    int pack_bwd(hipStream_t s) {
        const float r2 = 0.70710678118654752440f;
        ATd.resize(NL); ATg.resize(NL); ATw.resize(NL);
        DWS_TRY(tmp_pack.ensure((size_t)std::max(std::max(8 * C * C, (S + C) * C), S * S) * 4));
        for (int n = 0; n < NL; ++n) {
            // the dilated conv's adjoint weights in the form the step will use (re-packed after every optimizer step):
            // Winograd [C][4 * 2C] or direct [C][3 * 2C]; a conv_algo change goes through commit and lands here again
            if (wino_opt && tapwino_mfma_supported(C, 2 * C, 1 << (n % cycle))) {
                DWS_TRY(ATw[n].ensure((size_t)8 * C * C * 4));
                DWS_TRY(launch_tapwino_pack_transposed(Wd(n), tmp_pack.f(), 2 * C, C, s));
                DWS_TRY(launch_pack_a_frag(tmp_pack.f(), ATw[n].f(), C, 8 * C, s));
            } else {
                ATw[n].release();
                DWS_TRY(ATd[n].ensure((size_t)6 * C * C * 4));
                DWS_TRY(launch_tapconv_pack_transposed(Wd(n), tmp_pack.f(), 2 * C, C, 3, 6 * C, 0, 1.f, s));
                DWS_TRY(launch_pack_a_frag(tmp_pack.f(), ATd[n].f(), C, 6 * C, s));
            }
            DWS_TRY(ATg[n].ensure((size_t)(S + C) * C * 4));
            DWS_TRY(launch_tapconv_pack_transposed(Wrs[n].f() + (size_t)C * C, tmp_pack.f(), S, C, 1, S + C, 0, 1.f, s));
            DWS_TRY(launch_tapconv_pack_transposed(Wrs[n].f(), tmp_pack.f(), C, C, 1, S + C, S, r2, s));
            DWS_TRY(launch_pack_a_frag(tmp_pack.f(), ATg[n].f(), C, S + C, s));
        }
        DWS_TRY(ATf.ensure((size_t)S * S * 4));
        DWS_TRY(launch_tapconv_pack_transposed(Wf.f(), tmp_pack.f(), S, S, 1, S, 0, (float)std::sqrt(1.0 / NL), s));
        DWS_TRY(launch_pack_a_frag(tmp_pack.f(), ATf.f(), S, S, s));
        bwd_pack_version = commit_version;
        return DWS_OK;
    }

    // weight gradient, and (db != null) the bias gradient db[o] = bscale * sum dY[b, o, l] in the same pass
    int wgrad(const float* dY, const float* X, const float* addc, int addc_bs, float* dW, int O, int Cc, int T, int dil,
              float scale, hipStream_t s, float* db = nullptr, float bscale = 1.f) {
        if (!mfma_bwd) {
            if (db) DWS_TRY(launch_rowsum(dY, db, (int)B, O, (int)L, bscale, 0, s));
            return launch_wgrad(dY, X, addc, addc_bs, dW, (int)B, O, Cc, (int)L, T, dil, scale, s);
        }
        WgradArgs w{};
        w.dY = dY; w.X = X; w.addc = addc; w.addc_bstride = addc_bs;
        w.B = (int)B; w.O = O; w.C = Cc; w.L = (int)L; w.dil = dil;
        const bool wino_w = T == 3 && wino_opt && wgrad_wino_supported(w);   // conv_algo covers this adjoint too
        w.nsplit = wino_w ? wgrad_wino_nsplit((int)B, O, Cc, (int)L, dil) : wgrad_mfma_nsplit((int)B, O, Cc, (int)L, T);
        DWS_TRY(wpart.ensure((size_t)w.nsplit * O * Cc * (wino_w ? 4 : T) * 4));
        w.partial = wpart.f();
        if (db) {
            DWS_TRY(bpart.ensure((size_t)w.nsplit * O * 4));
            w.bias_part = bpart.f(); w.dbias = db; w.bias_scale = bscale;
        }
        if (wino_w) return launch_wgrad_wino(w, scale, dW, s);
        return launch_wgrad_mfma(w, T, scale, dW, s);
    }

    int prepare(int64_t nB, int64_t nL) override {
        DWS_CHECK(nB > 0 && nL > 0, DWS_ERR_INVALID, "prepare: B=%lld L=%lld", (long long)nB, (long long)nL);
        DWS_CHECK(nB * nL * (int64_t)std::max(2 * C, S) < (int64_t)1 << 40, DWS_ERR_UNSUPPORTED, "workspace too large");
        if (nB != B || nL != L) { drop_graph(); melBm = 0; trained_fwd = false; }
        if ((L == 0 || wino_fits(L)) != wino_fits(nL)) dirty = true;   // the layer kernel (and with it the A1 / Abt layout) changes
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
        DWS_TRY(mel_in.ensure((size_t)Bm * MB * Tmel * 4));
        DWS_HIP(hipMemcpyAsync(mel_in.p, mel, (size_t)Bm * MB * Tmel * 4, hipMemcpyDeviceToDevice, s));
        mel_T = (int)Tmel;
        melBm = Bm;
        return DWS_OK;
    }

    int final_stage(float* out, float* tap, hipStream_t s) {
        WnFinalArgs f{};
        f.skip = skip.f(); f.Af = Af.f(); f.Wf = Wf.f(); f.bf = P("final_conv.0.conv.bias");
        f.Wz = P("final_conv.2.conv.weight"); f.bz = P("final_conv.2.conv.bias");
        f.out = out; f.tap = tap; f.scale = (float)std::sqrt(1.0 / NL); f.af_scaled = mfma_final ? 1 : 0;
        f.B = (int)B; f.L = (int)L; f.Cout = Cout;
        return launch_wn_final(S, f, s);
    }

    int forward(const float* audio, const float* steps, float* out, hipStream_t s) override {
        trained_fwd = false;
        return run_forward(audio, steps, out, false, s);
    }

    int forward_train(const float* audio, const float* steps, float* out, hipStream_t s) override {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "forward before prepare");
        DWS_CHECK(!bf16x3 && !bf16x6 && !f16x3, DWS_ERR_UNSUPPORTED, "training runs with precision=f32 (no split-precision backward is built)");
        DWS_CHECK(melBm == 0 || (melBm == B && mfma_bwd), DWS_ERR_UNSUPPORTED,
                  "mel-conditional training needs one mel per clip (got %lld for B=%lld) and the MFMA adjoints (channels %% 32 == 0)",
                  (long long)melBm, (long long)B);
        DWS_CHECK(!cond || melBm > 0, DWS_ERR_STATE, "conditional model: install the mel (set_condition) before forward_train");
        const size_t act = (size_t)B * C * L * 4;
        tx.resize(NL + 1); tH.resize(NL);
        for (int n = 0; n <= NL; ++n) DWS_TRY(tx[n].ensure(act));
        for (int n = 0; n < NL; ++n) DWS_TRY(tH[n].ensure(2 * act));
        DWS_TRY(ty.ensure((size_t)B * S * L * 4));
        DWS_TRY(ta1.ensure((size_t)B * Emid * 4));
        DWS_TRY(ta2.ensure((size_t)B * Eout * 4));
        DWS_TRY(run_forward(audio, steps, out, true, s));
        train_audio = audio;
        trained_fwd = true;
        return DWS_OK;
    }

    const float* train_audio = nullptr;

    // floats of one (layer, clip) row of the step-embedding correction fragments, by layer kernel
    int abt_row() const { return wino_rows() ? 4 * 2 * C : (2 * C / 32) * (bf16x3 ? 512 : 256); }

    // everything of the forward that depends on the diffusion step only (`wavenet.py:153-155,89`; a1, a2 of SURVEY 8):
    // embedding -> MLP -> every layer's fc_t (one stacked GEMV) -> the layer kernels' correction fragments, for `rows`
    // step values.  Row results do not depend on how many rows a launch carries (one wave per output row).
    int embed_rows(const float* steps, int rows, float* emb_, float* h1_, float* h2_, float* pt, void* abt, float* pre1,
                   float* pre2, hipStream_t s) {
        DWS_TRY(launch_step_embed(steps, freq.f(), emb_, rows, Ein / 2, s));
        DWS_TRY(launch_linear_rows(emb_, P("residual_layer.fc_t1.weight"), P("residual_layer.fc_t1.bias"), h1_, rows, Ein, Emid,
                                   1, s, pre1));
        DWS_TRY(launch_linear_rows(h1_, P("residual_layer.fc_t2.weight"), P("residual_layer.fc_t2.bias"), h2_, rows, Emid, Eout,
                                   1, s, pre2));
        DWS_TRY(launch_linear_rows(h2_, Wt_all.f(), bt_all.f(), pt, rows, Eout, NL * C, 0, s));
        if (mfma_layer && bf16x3) DWS_TRY(launch_wn_bias_tap_bf16(Wd_all.f(), pt, b1_all.f(), abt, NL, rows, C, s));
        else if (wino_rows()) DWS_TRY(launch_wn_wino_bias(Wd_all.f(), pt, (float*)abt, NL, rows, C, s));
        else if (mfma_layer) DWS_TRY(launch_wn_bias_tap(Wd_all.f(), pt, (float*)abt, NL, rows, C, s));
        return DWS_OK;
    }

    // Step table of a sampler run (sampler.hip): in sampling every clip is at the same step (`generate.py:50`), so the
    // step-only part of the forward is evaluated ONCE for t = 0..T-1 -- tab_pt [T][NL*C], tab_abt [NL][T][abt_row] -- and
    // the captured reverse step reads row *step_idx: no embedding kernels in a replay.
    DevBuf tab_steps, tab_emb, tab_h1, tab_h2, tab_pt, tab_abt;
    int tab_T = 0;
    uint64_t tab_version = ~0ull;
    int build_step_table(int T, hipStream_t s) override {
        if (dirty) DWS_TRY(commit(s));
        if (tab_T == T && tab_version == commit_version) return DWS_OK;
        drop_graph();   // a captured step holds pointers into the old table
        {   // the table grows with T x layers x channels (0.06 GB at T = 200, C = 256; 0.3-0.6 GB at T = 1000): bounded, so that
            // a wrong T fails with a message instead of an allocation of whatever size it implies
            const size_t bytes = (size_t)T * ((size_t)NL * C + (mfma_layer ? (size_t)NL * abt_row() : 0) + Ein + Emid + Eout + 1) * 4;
            DWS_CHECK(T > 0 && bytes <= ((size_t)4 << 30), DWS_ERR_UNSUPPORTED,
                      "sampler step table: T=%d needs %.2f GB (limit 4 GB): sample with fewer steps or per-step forwards", T, bytes / 1e9);
        }
        DWS_TRY(tab_steps.ensure((size_t)T * 4));
        DWS_TRY(tab_emb.ensure((size_t)T * Ein * 4));
        DWS_TRY(tab_h1.ensure((size_t)T * Emid * 4));
        DWS_TRY(tab_h2.ensure((size_t)T * Eout * 4));
        DWS_TRY(tab_pt.ensure((size_t)T * NL * C * 4));
        if (mfma_layer) {
            const size_t n = (size_t)NL * T * abt_row() * 4;
            DWS_TRY(tab_abt.ensure(n));
            DWS_HIP(hipMemsetAsync(tab_abt.p, 0, n, s));   // unused k entries of the correction k-group stay zero
        }
        DWS_TRY(launch_iota_f32(tab_steps.f(), T, s));   // steps[t] = float(t), as `generate.py:50` feeds them
        DWS_TRY(embed_rows(tab_steps.f(), T, tab_emb.f(), tab_h1.f(), tab_h2.f(), tab_pt.f(), tab_abt.p, nullptr, nullptr, s));
        tab_T = T;
        tab_version = commit_version;
        return DWS_OK;
    }

    int run_forward(const float* audio, const float* steps, float* out, bool train, hipStream_t s) {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "forward before prepare");
        if (dirty) DWS_TRY(commit(s));
        const bool tab = step_idx != nullptr && !train;
        DWS_CHECK(!tab || (tab_T > 0 && tab_version == commit_version), DWS_ERR_STATE, "step-table forward without a current table");
        DWS_CHECK(tab || steps, DWS_ERR_INVALID, "forward: steps == null");
        float* xfirst = train ? tx[0].f() : x0.f();
        DWS_TRY(launch_init_conv(audio, Wi.f(), P("init_conv.0.conv.bias"), xfirst, (int)B, Cin, C, (int)L, s));
        if (!tab)
            DWS_TRY(embed_rows(steps, (int)B, emb.f(), h1.f(), h2.f(), part_t.f(), Abt.p, train ? ta1.f() : nullptr,
                               train ? ta2.f() : nullptr, s));
        const int arow = abt_row();
        for (int n = 0; n < NL; ++n) {
            const std::string p = "residual_layer.residual_blocks." + std::to_string(n);
            WnLayerArgs a{};
            a.x_in = train ? tx[n].f() : ((n & 1) ? x1.f() : x0.f());
            a.x_out = train ? tx[n + 1].f() : ((n & 1) ? x0.f() : x1.f());
            a.hsave = train ? tH[n].f() : nullptr;
            a.skip = skip.f();
            a.A1 = A1[n].f(); a.A2 = A2[n].f();
            if (tab) {
                a.part_t = tab_pt.f() + (size_t)n * C; a.part_t_bstride = 0; a.part_t_tstride = NL * C;
                a.Abt = mfma_layer ? tab_abt.f() + (size_t)n * tab_T * arow : nullptr; a.abt_bstride = 0; a.abt_tstride = arow;
                a.step_idx = step_idx;
            } else {
                a.part_t = part_t.f() + (size_t)n * C; a.part_t_bstride = NL * C;
                a.Abt = mfma_layer ? Abt.f() + (size_t)n * B * arow : nullptr; a.abt_bstride = arow;
            }
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
            else if (bf16x6) DWS_TRY(launch_wn_layer_bx6(C, S, a, WN_SPLIT_BF16X6, s));
            else if (f16x3) { a.wscale = wscale.f() + 2 * n; DWS_TRY(launch_wn_layer_bx6(C, S, a, WN_SPLIT_F16X3, s)); }
            else if (wino()) DWS_TRY(launch_wn_layer_wino(C, S, a, s));
            else if (mfma_layer) DWS_TRY(launch_wn_layer_mfma(C, S, a, s));
            else DWS_TRY(launch_wn_layer_generic(C, S, a, s));
        }
        DWS_TRY(final_stage(out, train ? ty.f() : nullptr, s));
        DWS_HIP(hipGetLastError());
        return DWS_OK;
    }

    // weight-norm adjoint of one folded weight into the raw (weight_g, weight_v) gradients
    int wn_bwd(const std::string& p, const float* dWf, int O, int inner, hipStream_t s) {
        return launch_weight_norm_bwd(dWf, P(p + ".weight_v"), P(p + ".weight_g"), G(p + ".weight_v"), G(p + ".weight_g"), O,
                                      inner, s);
    }

    // Adjoint of run_forward (`models/wavenet.py:82-121,149-165,202-210`), layer by layer in reverse.
    int backward(const float* dout, hipStream_t s) override {
        DWS_CHECK(trained_fwd, DWS_ERR_STATE, "backward without a preceding forward_train");
        const int nB = (int)B, nL = (int)L;
        const size_t act = (size_t)B * C * L * 4, nact = (size_t)B * C * L;
        DWS_TRY(dxa.ensure(act)); DWS_TRY(dxb.ensure(act)); DWS_TRY(dgb.ensure(act)); DWS_TRY(dresb.ensure(act));
        DWS_TRY(gate.ensure(act));
        DWS_TRY(dHb.ensure(2 * act));
        DWS_TRY(dskip.ensure((size_t)B * S * L * 4)); DWS_TRY(dyb.ensure((size_t)B * S * L * 4));
        DWS_TRY(dWfold.ensure((size_t)std::max(2 * C * C * 3, std::max((C + S) * C, S * S)) * 4));
        DWS_TRY(dpt.ensure((size_t)B * NL * C * 4));
        DWS_TRY(dh2.ensure((size_t)B * Eout * 4)); DWS_TRY(dh1.ensure((size_t)B * Emid * 4));
        DWS_TRY(dWt_all.ensure((size_t)NL * C * Eout * 4)); DWS_TRY(dbt_all.ensure((size_t)NL * C * 4));
        const float scale = (float)std::sqrt(1.0 / NL);

        // ---- final_conv: out = Wz y + bz, y = relu(Wf (skip * scale) + bf)
        DWS_TRY(wgrad(dout, ty.f(), nullptr, 0, G("final_conv.2.conv.weight"), Cout, S, 1, 1, 1.f, s, G("final_conv.2.conv.bias")));
        DWS_TRY(launch_final_dy(dout, P("final_conv.2.conv.weight"), ty.f(), dyb.f(), nB, S, Cout, nL, s));
        DWS_TRY(wgrad(dyb.f(), skip.f(), nullptr, 0, dWfold.f(), S, S, 1, 1, scale, s, G("final_conv.0.conv.bias")));
        DWS_TRY(wn_bwd("final_conv.0.conv", dWfold.f(), S, S, s));
        if (mfma_bwd) {  // dskip = scale * Wf^T dy, the same for every layer
            if (bwd_pack_version != commit_version) DWS_TRY(pack_bwd(s));
            TapConvArgs f{};
            f.src0 = dyb.f(); f.K0 = S; f.A = ATf.f(); f.nkg_total = S / 8; f.M = S; f.T = 1; f.dil = 1; f.sign = 1;
            f.out = dskip.f(); f.B = nB; f.L = nL;
            DWS_TRY(launch_tapconv_mfma(f, s));
        } else {
            DWS_TRY(launch_conv_t(dyb.f(), Wf.f(), dskip.f(), nB, S, S, nL, 1, 1, scale, 0, s));
        }

        // ---- residual layers, last to first
        float* dx_out = nullptr;  // gradient w.r.t. the layer's x output (none for the last layer)
        for (int n = NL - 1; n >= 0; --n) {
            const std::string p = "residual_layer.residual_blocks." + std::to_string(n);
            const int dil = 1 << (n % cycle);
            const float* Wr = Wrs[n].f();
            const float* Ws = Wrs[n].f() + (size_t)C * C;
            const float r2 = 0.70710678118654752440f;
            float* dh = (dx_out == dxa.f()) ? dxb.f() : dxa.f();
            if (mfma_bwd) {
                // dg = Ws^T dskip + sqrt(.5) Wr^T dx', gate adjoint fused in the epilogue
                TapConvArgs q{};
                q.src0 = dskip.f(); q.K0 = S; q.src1 = dx_out; q.K1 = dx_out ? C : 0;
                q.A = ATg[n].f(); q.nkg_total = (S + C) / 8; q.M = C; q.T = 1; q.dil = 1; q.sign = 1; q.epi = 1;
                q.H = tH[n].f(); q.dH = dHb.f(); q.g = gate.f(); q.B = nB; q.L = nL;
                DWS_TRY(launch_tapconv_mfma(q, s));
            } else {
                if (dx_out) {
                    DWS_TRY(launch_scale(dx_out, dresb.f(), r2, nact, s));
                    DWS_TRY(launch_conv_t(dresb.f(), Wr, dgb.f(), nB, C, C, nL, 1, 1, 1.f, 0, s));
                    DWS_TRY(launch_conv_t(dskip.f(), Ws, dgb.f(), nB, S, C, nL, 1, 1, 1.f, 1, s));
                } else {
                    DWS_TRY(launch_conv_t(dskip.f(), Ws, dgb.f(), nB, S, C, nL, 1, 1, 1.f, 0, s));
                }
                DWS_TRY(launch_gate_bwd(dgb.f(), tH[n].f(), dHb.f(), gate.f(), nB, C, nL, s));
            }
            if (melBm) {  // conditioner of this layer: d melc = dH (`wavenet.py:98-111`)
                const int s0 = d.mel_upsample[0], s1 = d.mel_upsample[1];
                DWS_TRY(gW0f.ensure((size_t)3 * 2 * s0 * 4)); DWS_TRY(gW1f.ensure((size_t)3 * 2 * s1 * 4));
                DWS_TRY(gWcf.ensure((size_t)2 * C * MB * 4));
                DWS_TRY(conditioner_backward(cws, mel_in.f(), nB, MB, mel_T, s0, s1, melW0[n].f(), P(p + ".upsample_conv2d.0.bias"),
                                             melW1[n].f(), P(p + ".upsample_conv2d.1.bias"), melWc[n].f(), 2 * C, nL, dHb.f(),
                                             gW0f.f(), G(p + ".upsample_conv2d.0.bias"), gW1f.f(),
                                             G(p + ".upsample_conv2d.1.bias"), gWcf.f(), s));
                DWS_TRY(wn_bwd(p + ".upsample_conv2d.0", gW0f.f(), 1, 3 * 2 * s0, s));
                DWS_TRY(wn_bwd(p + ".upsample_conv2d.1", gW1f.f(), 1, 3 * 2 * s1, s));
                DWS_TRY(wn_bwd(p + ".mel_conv.conv", gWcf.f(), 2 * C, MB, s));
            }
            // res / skip 1x1 weights and biases (dres = dx' * sqrt(.5))
            if (dx_out) {
                DWS_TRY(wgrad(dx_out, gate.f(), nullptr, 0, dWfold.f(), C, C, 1, 1, r2, s, G(p + ".res_conv.bias"), r2));
                DWS_TRY(wn_bwd(p + ".res_conv", dWfold.f(), C, C, s));
            } else {  // the last layer's residual branch feeds nothing (`wavenet.py:165` uses only the skips)
                DWS_HIP(hipMemsetAsync(G(p + ".res_conv.weight_v"), 0, (size_t)C * C * 4, s));
                DWS_HIP(hipMemsetAsync(G(p + ".res_conv.weight_g"), 0, (size_t)C * 4, s));
                DWS_HIP(hipMemsetAsync(G(p + ".res_conv.bias"), 0, (size_t)C * 4, s));
            }
            DWS_TRY(wgrad(dskip.f(), gate.f(), nullptr, 0, dWfold.f(), S, C, 1, 1, 1.f, s, G(p + ".skip_conv.bias")));
            DWS_TRY(wn_bwd(p + ".skip_conv", dWfold.f(), S, C, s));
            // dilated conv: weights see h = x + pt (zero padded), input gets the transposed conv
            DWS_TRY(wgrad(dHb.f(), tx[n].f(), part_t.f() + (size_t)n * C, NL * C, dWfold.f(), 2 * C, C, 3, dil, 1.f, s,
                          G(p + ".dilated_conv_layer.conv.bias")));
            DWS_TRY(wn_bwd(p + ".dilated_conv_layer.conv", dWfold.f(), 2 * C, C * 3, s));
            if (melBm)    // the conditioner's 1x1 bias enters H next to the dilated conv's: same gradient
                DWS_HIP(hipMemcpyAsync(G(p + ".mel_conv.conv.bias"), G(p + ".dilated_conv_layer.conv.bias"), (size_t)2 * C * 4,
                                       hipMemcpyDeviceToDevice, s));
            if (mfma_bwd) {
                TapConvArgs q{};
                q.src0 = dHb.f(); q.K0 = 2 * C; q.A = ATd[n].f(); q.nkg_total = 6 * C / 8; q.M = C; q.T = 3; q.dil = dil;
                q.sign = -1; q.out = dh; q.B = nB; q.L = nL;
                if (wino_opt && tapwino_mfma_supported(C, 2 * C, dil) && ATw[n].p) {   // conv_algo covers the adjoint too
                    q.A = ATw[n].f(); q.nkg_total = C;
                    DWS_TRY(launch_tapwino_mfma(q, s));
                } else {
                    DWS_CHECK(ATd[n].p, DWS_ERR_STATE, "backward: layer %d's direct adjoint weights were not packed", n);
                    DWS_TRY(launch_tapconv_mfma(q, s));
                }
            } else {
                DWS_TRY(launch_conv_t(dHb.f(), Wd(n), dh, nB, 2 * C, C, nL, 3, dil, 1.f, 0, s));
            }
            DWS_TRY(launch_rowsum_bc(dh, dpt.f() + (size_t)n * C, NL * C, nB, C, nL, s));  // d fc_t(e)[b, n, c]
            if (dx_out) DWS_TRY(launch_dx_combine(dh, dx_out, nact, s));                   // + dx' * sqrt(.5)
            dx_out = dh;
            DWS_TRY(grad_point(s));   // staged hand-over: buckets whose last gradient this layer produced leave now
        }
        // ---- init_conv: x0 = relu(Wi audio + bi)
        DWS_TRY(launch_relu_bwd(dx_out, tx[0].f(), nact, s));
        DWS_TRY(wgrad(dx_out, train_audio, nullptr, 0, dWfold.f(), C, Cin, 1, 1, 1.f, s, G("init_conv.0.conv.bias")));
        DWS_TRY(wn_bwd("init_conv.0.conv", dWfold.f(), C, Cin, s));

        // ---- step embedding: per-layer fc_t (stacked), then the shared swish MLP
        DWS_TRY(launch_lin_bwd_w(dpt.f(), h2.f(), dWt_all.f(), dbt_all.f(), nB, Eout, NL * C, s));
        unstack_fc_t.begin();
        for (int n = 0; n < NL; ++n) {
            const std::string p = "residual_layer.residual_blocks." + std::to_string(n);
            unstack_fc_t.add(dWt_all.f() + (size_t)n * C * Eout, G(p + ".fc_t.weight"), (size_t)C * Eout);
            unstack_fc_t.add(dbt_all.f() + (size_t)n * C, G(p + ".fc_t.bias"), (size_t)C);
        }
        DWS_TRY(unstack_fc_t.run(s));
        DWS_TRY(launch_lin_bwd_x(dpt.f(), Wt_all.f(), ta2.f(), dh2.f(), nB, Eout, NL * C, lin_scratch, s));   // d(pre-activation 2)
        DWS_TRY(launch_lin_bwd_w(dh2.f(), h1.f(), G("residual_layer.fc_t2.weight"), G("residual_layer.fc_t2.bias"), nB, Emid,
                                 Eout, s));
        DWS_TRY(launch_lin_bwd_x(dh2.f(), P("residual_layer.fc_t2.weight"), ta1.f(), dh1.f(), nB, Emid, Eout, lin_scratch, s));
        DWS_TRY(launch_lin_bwd_w(dh1.f(), emb.f(), G("residual_layer.fc_t1.weight"), G("residual_layer.fc_t1.bias"), nB, Ein,
                                 Emid, s));
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
        if (t == "split_launches") {   // [residual layers per forward on a split instance, on exact-f32 kernels]: all or nothing here
            DWS_CHECK(capacity >= 2, DWS_ERR_INVALID, "tap buffer too small");   // (set_option refuses a split the channel counts do not cover)
            const bool sp = mfma_layer && (bf16x3 || bf16x6 || f16x3);
            const float v[2] = {sp ? (float)NL : 0.f, sp ? 0.f : (float)NL};
            DWS_HIP(hipMemcpyAsync(dst, v, 8, hipMemcpyHostToDevice, s));
            DWS_HIP(hipStreamSynchronize(s));
            return DWS_OK;
        }
        if (t == "pre_final") {
            DWS_CHECK(capacity >= B * S * L, DWS_ERR_INVALID, "tap buffer too small");
            DWS_TRY(scratch_out.ensure((size_t)B * Cout * L * 4));
            return final_stage(scratch_out.f(), dst, s);
        }
        // step-only terms: the per-clip rows of the last forward and the sampler's step table (parity / debugging)
        struct { const char* name; const DevBuf* buf; size_t n; } small[] = {
            {"emb", &emb, (size_t)B * Ein}, {"emb_mlp", &h2, (size_t)B * Eout},
            {"part_t", &part_t, (size_t)B * NL * C}, {"abt", &Abt, mfma_layer ? (size_t)NL * B * abt_row() : 0},
            {"tab_part_t", &tab_pt, (size_t)tab_T * NL * C}, {"tab_abt", &tab_abt, mfma_layer ? (size_t)NL * tab_T * abt_row() : 0}};
        for (auto& e : small)
            if (t == e.name) {
                DWS_CHECK(e.buf->p && e.n > 0 && capacity >= (int64_t)e.n, DWS_ERR_INVALID, "tap '%s': %zu floats, capacity %lld",
                          tap, e.n, (long long)capacity);
                DWS_HIP(hipMemcpyAsync(dst, e.buf->p, e.n * 4, hipMemcpyDeviceToDevice, s));
                return DWS_OK;
            }
        return set_error(DWS_ERR_INVALID, "unknown tap '%s'", tap);
    }
};

dws_model* make_wavenet(const dws_model_desc& d) { return new WaveNetModel(d); }

}  // namespace dws
