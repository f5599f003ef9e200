// SaShiMi backbone behind the C ABI (`models/sashimi.py:187-313`, `models/s4.py`).
//
// Everything that does not depend on x_t is evaluated once per weight load in
// commit(): weight-norm folding, the S4 convolution kernels (Cauchy -> Woodbury
// -> bilinear factor -> irfft -> two-sided kernel -> rfft = K_f per layer; the
// reference regenerates these in all 30 layers on every reverse step,
// `s4.py:1388`) and, in set_condition(), the mel conditioner terms.
// The per-step path is then LN+emb -> rocFFT r2c -> spectrum multiply -> c2r ->
// D-skip+GELU -> 1x1+GLU+residual -> LN -> FF -> residual per block.
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdlib>

#include "conditioner.h"
#include "fftconv.h"
#include "model.h"
#include "sashimi.h"
#include "sashimi_mfma.h"
#include "wavenet.h"

namespace dws {

#define DWS_FFT(expr)                                                                                  \
    do {                                                                                               \
        hipfftResult _r = (expr);                                                                      \
        if (_r != HIPFFT_SUCCESS)                                                                      \
            return set_error(DWS_ERR_HIP, "%s failed: hipfftResult %d (%s:%d)", #expr, (int)_r, __FILE__, __LINE__); \
    } while (0)

struct FftPlans {
    std::map<std::tuple<int, int, int>, hipfftHandle> plans;  // (type, n, batch)
    ~FftPlans() {
        for (auto& kv : plans) hipfftDestroy(kv.second);
    }
    // type 0: R2C rows of n reals (dist n) -> n/2+1 complex; type 1: C2R n/2+1 complex -> n reals (dist n)
    int get(int type, int n, int batch, hipfftHandle* out) {
        auto key = std::make_tuple(type, n, batch);
        auto it = plans.find(key);
        if (it == plans.end()) {
            hipfftHandle h;
            int nn[1] = {n};
            int nr[1] = {n}, nc[1] = {n / 2 + 1};
            if (type == 0)
                DWS_FFT(hipfftPlanMany(&h, 1, nn, nr, 1, n, nc, 1, n / 2 + 1, HIPFFT_R2C, batch));
            else
                DWS_FFT(hipfftPlanMany(&h, 1, nn, nc, 1, n / 2 + 1, nr, 1, n, HIPFFT_C2R, batch));
            it = plans.emplace(key, h).first;
        }
        *out = it->second;
        return DWS_OK;
    }
};

enum { L_BLOCK = 0, L_DOWN = 1, L_UP = 2 };

struct SLayer {
    int kind, H, L, p = 1, Hout = 0, Lout = 0;
    std::string prefix;
    int pt_off = 0;       // offset of this block's fc_t rows in the stacked projection
    int stage = 0;        // index into per-(H,L) workspaces
    DevBuf W1, W2, Wp;    // folded ff / pool weights
    DevBuf Ao, A1, A2, rs1, Ap;  // MFMA-packed copies (+ row sums of W1 for the folded LayerNorm)
    bool mfma = false;
    DevBuf Kf;            // rocFFT path: [H][L+1] complex spectrum of the two-sided kernel (n = 2L)
    DevBuf kfa, kfb, kfs; // fused path: pair-ordered spectrum at the power-of-two size (fftconv.h)
    int log2m = 0;        // > 0: fused LDS FFT convolution is used for this block
    DevBuf melW0, melW1, melWc, melc;
    DevBuf out;           // activation produced by this layer
};

struct Stage {
    int H, L;
    bool rocfft = false;  // some block of this stage needs the rocFFT path
    DevBuf U, Uf, Y, g, x1, n2, ffu, y;
};

struct FftTables {
    DevBuf tw, twn, twp;
};

struct SashimiModel : dws_model {
    int Cin, Cout, D, NL, E, FF, NS = 32, Ein, Emid, Eout, MB;
    bool cond, unet;
    std::vector<int> pool;
    std::vector<SLayer*> d_layers, c_layers, u_layers, all;
    std::vector<Stage*> stages;
    int pt_total = 0;
    FftPlans fft;
    std::map<int, FftTables*> tables;  // by log2(M)
    DevBuf Wi, Wt_all, bt_all, Wf, Af, freq;
    DevBuf x_init, emb, h1, h2, part_t, nfin, scratch_out;
    // commit scratch
    DevBuf cv, cwdt, cdt, cr, ckf, ck, cK, cKf;
    int64_t melBm = 0;

    ~SashimiModel() override {
        for (auto* l : all) delete l;
        for (auto* s : stages) delete s;
        for (auto& kv : tables) delete kv.second;
    }

    int get_tables(int log2m, FftTables** out, hipStream_t s) {
        auto it = tables.find(log2m);
        if (it == tables.end()) {
            auto* t = new FftTables();
            std::vector<float> tw, twn, twp;
            build_fft_tables(log2m, tw, twn, twp);
            DWS_TRY(t->tw.ensure(tw.size() * 4));
            DWS_TRY(t->twn.ensure(twn.size() * 4));
            DWS_TRY(t->twp.ensure(twp.size() * 4));
            DWS_HIP(hipMemcpyAsync(t->tw.p, tw.data(), tw.size() * 4, hipMemcpyHostToDevice, s));
            DWS_HIP(hipMemcpyAsync(t->twn.p, twn.data(), twn.size() * 4, hipMemcpyHostToDevice, s));
            DWS_HIP(hipMemcpyAsync(t->twp.p, twp.data(), twp.size() * 4, hipMemcpyHostToDevice, s));
            DWS_HIP(hipStreamSynchronize(s));
            it = tables.emplace(log2m, t).first;
        }
        *out = it->second;
        return DWS_OK;
    }

    void wn(const std::string& p, std::vector<int64_t> vshape) {
        std::vector<int64_t> g(vshape.size(), 1);
        g[0] = vshape[0];
        add_param(p + ".bias", {vshape[0]});
        add_param(p + ".weight_g", g);
        add_param(p + ".weight_v", vshape);
    }

    int stage_of(int H, int L) {
        for (size_t i = 0; i < stages.size(); ++i)
            if (stages[i]->H == H && stages[i]->L == L) return (int)i;
        auto* s = new Stage();
        s->H = H; s->L = L;
        stages.push_back(s);
        // FFT nodes for this length, computed by the host side with the reference's own
        // complex64 expression (s4.py:561-565) -- rounding-sensitive, SURVEY.md 7.
        add_param("__omega." + std::to_string(L), {L / 2 + 1, 2});
        add_param("__z." + std::to_string(L), {L / 2 + 1, 2});
        return (int)stages.size() - 1;
    }

    SLayer* mk_block(const std::string& prefix, int H, int L) {
        auto* l = new SLayer();
        l->kind = L_BLOCK; l->H = H; l->L = L; l->prefix = prefix;
        l->pt_off = pt_total; pt_total += H;
        l->stage = stage_of(H, L);
        add_param(prefix + ".fc_t.weight", {H, Eout});
        add_param(prefix + ".fc_t.bias", {H});
        add_param(prefix + ".layer.D", {1, H});
        const std::string k = prefix + ".layer.kernel.kernel";
        add_param(k + ".C", {2, H, NS, 2});
        add_param(k + ".log_dt", {H});
        add_param(k + ".B", {1, H, NS, 2});
        add_param(k + ".P", {1, H, NS, 2});
        add_param(k + ".inv_w_real", {H, NS});
        add_param(k + ".w_imag", {H, NS});
        add_param(k + ".L", {}, 1);
        add_param(prefix + ".layer.output_linear.0.weight", {2 * H, H, 1});
        add_param(prefix + ".layer.output_linear.0.bias", {2 * H});
        wn(prefix + ".ff.ff.0.conv", {FF * H, H, 1});
        wn(prefix + ".ff.ff.2.conv", {H, FF * H, 1});
        for (const char* nrm : {".norm1", ".norm2"}) {
            add_param(prefix + nrm + ".m", {1});
            add_param(prefix + nrm + ".s", {1});
        }
        if (cond) {
            for (int i = 0; i < 2; ++i) {
                const std::string u = prefix + ".upsample_conv2d." + std::to_string(i);
                add_param(u + ".bias", {1});
                add_param(u + ".weight_g", {1, 1, 1, 1});
                add_param(u + ".weight_v", {1, 1, 3, 2 * d.mel_upsample[i]});
            }
            wn(prefix + ".mel_conv.conv", {H, MB, 1});
        }
        all.push_back(l);
        return l;
    }

    SLayer* mk_pool(const std::string& prefix, int kind, int Hin, int Lin, int p) {
        auto* l = new SLayer();
        l->kind = kind; l->H = Hin; l->L = Lin; l->p = p; l->prefix = prefix;
        if (kind == L_DOWN) {
            l->Hout = Hin * E; l->Lout = Lin / p;
            wn(prefix + ".linear.conv", {l->Hout, Hin * p, 1});
        } else {
            l->Hout = Hin / E; l->Lout = Lin * p;
            wn(prefix + ".linear.conv", {l->Hout * p, Hin, 1});
        }
        all.push_back(l);
        return l;
    }

    explicit SashimiModel(const dws_model_desc& dd) {
        d = dd;
        Cin = d.in_channels; Cout = d.out_channels; D = d.d_model; NL = d.n_layers; E = d.expand; FF = d.ff;
        Ein = d.diffusion_step_embed_dim_in; Emid = d.diffusion_step_embed_dim_mid; Eout = d.diffusion_step_embed_dim_out;
        MB = d.mel_bands; cond = !d.unconditional; unet = d.unet != 0;
        for (int i = 0; i < d.n_pool; ++i) pool.push_back(d.pool[i]);
        wn("init_conv.0.conv", {D, Cin, 1});
        add_param("fc_t1.weight", {Emid, Ein});
        add_param("fc_t1.bias", {Emid});
        add_param("fc_t2.weight", {Eout, Emid});
        add_param("fc_t2.bias", {Eout});
        // layer plan of `Sashimi.__init__` (sashimi.py:236-268)
        int H = D, L = d.L, idx = 0;
        for (int p : pool) {
            if (unet)
                for (int i = 0; i < NL; ++i) d_layers.push_back(mk_block("d_layers." + std::to_string(idx++), H, L));
            d_layers.push_back(mk_pool("d_layers." + std::to_string(idx++), L_DOWN, H, L, p));
            L /= p; H *= E;
        }
        for (int i = 0; i < NL; ++i) c_layers.push_back(mk_block("c_layers." + std::to_string(i), H, L));
        idx = 0;
        for (auto it = pool.rbegin(); it != pool.rend(); ++it) {
            const int p = *it;
            u_layers.push_back(mk_pool("u_layers." + std::to_string(idx++), L_UP, H, L, p));
            H /= E; L *= p;
            for (int i = 0; i < NL; ++i) u_layers.push_back(mk_block("u_layers." + std::to_string(idx++), H, L));
        }
        add_param("norm.m", {1});
        add_param("norm.s", {1});
        wn("final_conv.0.conv", {D, D, 1});
        add_param("final_conv.2.conv.weight", {Cout, D, 1});
        add_param("final_conv.2.conv.bias", {Cout});
    }

    int fold(const std::string& p, float* out, int O, int inner, hipStream_t s) {
        return launch_fold_weight_norm(P(p + ".weight_v"), P(p + ".weight_g"), out, O, inner, s);
    }

    // S4 convolution kernel of one block: parameters -> K_f   (s4.py:704-807, 1391-1403)
    int build_kernel(SLayer* l, hipStream_t s) {
        const int H = l->H, L = l->L, Lh = L / 2 + 1, N = NS;
        const std::string k = l->prefix + ".layer.kernel.kernel";
        int64_t Lbuf = 0;
        DWS_HIP(hipMemcpyAsync(&Lbuf, P(k + ".L"), 8, hipMemcpyDeviceToHost, s));
        DWS_HIP(hipStreamSynchronize(s));
        DWS_CHECK(Lbuf == L, DWS_ERR_STATE,
                  "%s.L = %lld but the layer runs at length %d: C must have been through _setup_C(l_max) "
                  "(s4.py:524-551) before it is handed to the engine",
                  k.c_str(), (long long)Lbuf, L);
        DWS_TRY(cv.ensure((size_t)6 * H * N * 8));
        DWS_TRY(cwdt.ensure((size_t)H * N * 8));
        DWS_TRY(cdt.ensure((size_t)H * 4));
        DWS_TRY(cr.ensure((size_t)6 * H * Lh * 8));
        DWS_TRY(ckf.ensure((size_t)2 * H * Lh * 8));
        DWS_TRY(ck.ensure((size_t)2 * H * L * 4));
        DWS_TRY(launch_s4_prep(P(k + ".C"), P(k + ".B"), P(k + ".P"), P(k + ".inv_w_real"), P(k + ".w_imag"),
                               P(k + ".log_dt"), cv.f(), cwdt.f(), cdt.f(), H, N, s));
        DWS_TRY(launch_cauchy_sym_fwd_bcast(cv.f(), P("__z." + std::to_string(L)), cwdt.f(), cr.f(), 6 * H, N, Lh, H, s));
        DWS_TRY(launch_s4_woodbury(cr.f(), P("__omega." + std::to_string(L)), cdt.f(), ckf.f(), H, Lh, (L % 2) == 0, s));
        hipfftHandle plan;
        DWS_TRY(fft.get(1, L, 2 * H, &plan));
        DWS_FFT(hipfftSetStream(plan, s));
        DWS_FFT(hipfftExecC2R(plan, (hipfftComplex*)ckf.p, (hipfftReal*)ck.p));
        int lg = 0;
        if (fftconv_supported(L, &lg) && !getenv("DWS_SASHIMI_ROCFFT")) {
            // fused path: spectrum at the power-of-two size Nf = 2M with the anti-causal half re-placed,
            // produced by the same LDS FFT the per-step kernel uses, stored in its pair order
            const int M = 1 << lg, Nf = 2 * M;
            FftTables* t;
            DWS_TRY(get_tables(lg, &t, s));
            DWS_TRY(cK.ensure((size_t)H * Nf * 4));
            DWS_TRY(cKf.ensure((size_t)H * (M + 1) * 8));
            DWS_TRY(l->kfa.ensure((size_t)H * (M / 2) * 8));
            DWS_TRY(l->kfb.ensure((size_t)H * (M / 2) * 8));
            DWS_TRY(l->kfs.ensure((size_t)H * 3 * 8));
            DWS_TRY(launch_s4_twosided_pow2(ck.f(), cK.f(), H, L, Nf, s));
            DWS_TRY(launch_rfft_rows(lg, cK.f(), cKf.f(), t->tw.f(), t->twn.f(), H, s));
            DWS_TRY(launch_kf_permute(cKf.f(), l->kfa.f(), l->kfb.f(), l->kfs.f(), H, lg, s));
            l->log2m = lg;
        } else {
            DWS_TRY(cK.ensure((size_t)H * 2 * L * 4));
            DWS_TRY(l->Kf.ensure((size_t)H * (L + 1) * 8));
            DWS_TRY(launch_s4_twosided(ck.f(), cK.f(), H, L, s));
            DWS_TRY(fft.get(0, 2 * L, H, &plan));
            DWS_FFT(hipfftSetStream(plan, s));
            DWS_FFT(hipfftExecR2C(plan, (hipfftReal*)cK.p, (hipfftComplex*)l->Kf.p));
            l->log2m = 0;
            stages[l->stage]->rocfft = true;
        }
        return DWS_OK;
    }

    int commit(hipStream_t s) override {
        DWS_TRY(Wi.ensure((size_t)D * Cin * 4));
        DWS_TRY(fold("init_conv.0.conv", Wi.f(), D, Cin, s));
        DWS_TRY(Wt_all.ensure((size_t)pt_total * Eout * 4));
        DWS_TRY(bt_all.ensure((size_t)pt_total * 4));
        for (auto* l : all) {
            if (l->kind == L_BLOCK) {
                const int H = l->H;
                DWS_HIP(hipMemcpyAsync(Wt_all.f() + (size_t)l->pt_off * Eout, P(l->prefix + ".fc_t.weight"),
                                       (size_t)H * Eout * 4, hipMemcpyDeviceToDevice, s));
                DWS_HIP(hipMemcpyAsync(bt_all.f() + l->pt_off, P(l->prefix + ".fc_t.bias"), (size_t)H * 4,
                                       hipMemcpyDeviceToDevice, s));
                DWS_TRY(l->W1.ensure((size_t)FF * H * H * 4));
                DWS_TRY(l->W2.ensure((size_t)FF * H * H * 4));
                DWS_TRY(fold(l->prefix + ".ff.ff.0.conv", l->W1.f(), FF * H, H, s));
                DWS_TRY(fold(l->prefix + ".ff.ff.2.conv", l->W2.f(), H, FF * H, s));
                l->mfma = s4_tail_mfma_supported(H, FF) && !getenv("DWS_SASHIMI_GENERIC");
                if (l->mfma) {
                    DWS_TRY(l->Ao.ensure((size_t)2 * H * H * 4));
                    DWS_TRY(l->A1.ensure((size_t)FF * H * H * 4));
                    DWS_TRY(l->A2.ensure((size_t)FF * H * H * 4));
                    DWS_TRY(l->rs1.ensure((size_t)FF * H * 4));
                    DWS_TRY(launch_pack_a_frag(P(l->prefix + ".layer.output_linear.0.weight"), l->Ao.f(), 2 * H, H, s));
                    DWS_TRY(launch_pack_a_frag(l->W1.f(), l->A1.f(), FF * H, H, s));
                    DWS_TRY(launch_pack_a_frag(l->W2.f(), l->A2.f(), H, FF * H, s));
                    DWS_TRY(launch_row_sum(l->W1.f(), l->rs1.f(), FF * H, H, s));
                }
                DWS_TRY(build_kernel(l, s));
                if (cond) {
                    for (int i = 0; i < 2; ++i) {
                        const int sc = d.mel_upsample[i];
                        DevBuf& w = (i == 0) ? l->melW0 : l->melW1;
                        DWS_TRY(w.ensure((size_t)3 * 2 * sc * 4));
                        DWS_TRY(fold(l->prefix + ".upsample_conv2d." + std::to_string(i), w.f(), 1, 3 * 2 * sc, s));
                    }
                    DWS_TRY(l->melWc.ensure((size_t)H * MB * 4));
                    DWS_TRY(fold(l->prefix + ".mel_conv.conv", l->melWc.f(), H, MB, s));
                }
            } else {
                const int O = (l->kind == L_DOWN) ? l->Hout : l->Hout * l->p;
                const int K = (l->kind == L_DOWN) ? l->H * l->p : l->H;
                DWS_TRY(l->Wp.ensure((size_t)O * K * 4));
                DWS_TRY(fold(l->prefix + ".linear.conv", l->Wp.f(), O, K, s));
                l->mfma = pw_mfma_supported(l->kind == L_DOWN ? 0 : 1, K, O, l->p) && !getenv("DWS_SASHIMI_GENERIC");
                if (l->mfma) {
                    DWS_TRY(l->Ap.ensure((size_t)O * K * 4));
                    DWS_TRY(launch_pack_a_frag(l->Wp.f(), l->Ap.f(), O, K, s));
                }
            }
        }
        DWS_TRY(Wf.ensure((size_t)D * D * 4));
        DWS_TRY(fold("final_conv.0.conv", Wf.f(), D, D, s));
        if (wn_final_mfma_supported(D)) {
            DWS_TRY(Af.ensure((size_t)D * D * 4));
            DWS_TRY(launch_pack_a_frag(Wf.f(), Af.f(), D, D, s));
        }
        {
            const int half = Ein / 2;
            std::vector<float> f(half);
            const float e = (float)(-(std::log(10000.0) / (half - 1)));
            for (int i = 0; i < half; ++i) f[i] = (float)std::exp((double)((float)i * e));
            DWS_TRY(freq.ensure((size_t)half * 4));
            DWS_HIP(hipMemcpyAsync(freq.p, f.data(), (size_t)half * 4, hipMemcpyHostToDevice, s));
            DWS_HIP(hipStreamSynchronize(s));
        }
        dirty = false;
        melBm = 0;
        return DWS_OK;
    }

    int prepare(int64_t nB, int64_t nL) override {
        DWS_CHECK(nB > 0 && nL > 0, DWS_ERR_INVALID, "prepare: B=%lld L=%lld", (long long)nB, (long long)nL);
        DWS_CHECK(nL == d.L, DWS_ERR_UNSUPPORTED,
                  "sashimi: input length %lld != model L=%d (variable-length generation, s4.py:1387, is not built yet)",
                  (long long)nL, d.L);
        if (nB != B) { drop_graph(); melBm = 0; }
        B = nB; L = nL;
        for (auto* st : stages) {
            const size_t rows = (size_t)B * st->H, Ls = st->L;
            int lg = 0;
            const bool need_rocfft = !fftconv_supported((int)Ls, &lg) || getenv("DWS_SASHIMI_ROCFFT");
            if (need_rocfft) {
                const bool fresh = st->U.bytes < rows * 2 * Ls * 4;
                DWS_TRY(st->U.ensure(rows * 2 * Ls * 4));
                if (fresh) DWS_HIP(hipMemset(st->U.p, 0, rows * 2 * Ls * 4));  // zero padding of the FFT input rows
                DWS_TRY(st->Uf.ensure(rows * (Ls + 1) * 8));
                DWS_TRY(st->Y.ensure(rows * 2 * Ls * 4));
            } else {
                DWS_TRY(st->y.ensure(rows * Ls * 4));
            }
            DWS_TRY(st->g.ensure(rows * Ls * 4));
            DWS_TRY(st->x1.ensure(rows * Ls * 4));
            DWS_TRY(st->n2.ensure(rows * Ls * 4));
            DWS_TRY(st->ffu.ensure(rows * FF * Ls * 4));
        }
        for (auto* l : all) {
            const size_t n = (l->kind == L_BLOCK) ? (size_t)B * l->H * l->L : (size_t)B * l->Hout * l->Lout;
            DWS_TRY(l->out.ensure(n * 4));
        }
        // FFT plans allocate: create them here, never inside a stream capture
        for (auto* st : stages) {
            int lg = 0;
            if (fftconv_supported(st->L, &lg) && !getenv("DWS_SASHIMI_ROCFFT")) continue;
            hipfftHandle plan;
            DWS_TRY(fft.get(0, 2 * st->L, (int)B * st->H, &plan));
            DWS_TRY(fft.get(1, 2 * st->L, (int)B * st->H, &plan));
        }
        DWS_TRY(x_init.ensure((size_t)B * D * L * 4));
        DWS_TRY(nfin.ensure((size_t)B * D * L * 4));
        DWS_TRY(emb.ensure((size_t)B * Ein * 4));
        DWS_TRY(h1.ensure((size_t)B * Emid * 4));
        DWS_TRY(h2.ensure((size_t)B * Eout * 4));
        DWS_TRY(part_t.ensure((size_t)B * pt_total * 4));
        return DWS_OK;
    }

    int set_condition(const float* mel, int64_t Bm, int64_t Tmel, hipStream_t s) override {
        if (mel == nullptr) { melBm = 0; return DWS_OK; }
        DWS_CHECK(cond, DWS_ERR_INVALID, "set_condition on an unconditional model (`sashimi.py:161`)");
        DWS_CHECK(B > 0, DWS_ERR_STATE, "set_condition before prepare");
        DWS_CHECK(Bm == 1 || Bm == B, DWS_ERR_INVALID, "mel batch %lld must be 1 or B=%lld", (long long)Bm, (long long)B);
        if (dirty) DWS_TRY(commit(s));
        const int s0 = d.mel_upsample[0], s1 = d.mel_upsample[1];
        const int T0 = mel_upsampled_len((int)Tmel, s0), T1 = mel_upsampled_len(T0, s1);
        DevBuf u0, u1;
        DWS_TRY(u0.ensure((size_t)Bm * MB * T0 * 4));
        DWS_TRY(u1.ensure((size_t)Bm * MB * T1 * 4));
        for (auto* l : all) {
            if (l->kind != L_BLOCK) continue;
            // pooled stages take the FIRST L_stage upsampled frames (sashimi.py:170-172): a truncation
            DWS_CHECK(T1 >= l->L, DWS_ERR_INVALID, "upsampled mel length %d < L=%d (`sashimi.py:169`)", T1, l->L);
            DWS_TRY(l->melc.ensure((size_t)Bm * l->H * l->L * 4));
            DWS_TRY(launch_mel_upsample(mel, l->melW0.f(), P(l->prefix + ".upsample_conv2d.0.bias"), u0.f(), (int)Bm, MB,
                                        (int)Tmel, T0, s0, 0.4f, s));
            DWS_TRY(launch_mel_upsample(u0.f(), l->melW1.f(), P(l->prefix + ".upsample_conv2d.1.bias"), u1.f(), (int)Bm, MB,
                                        T0, T1, s1, 0.4f, s));
            DWS_TRY(launch_conv1x1_trunc(u1.f(), l->melWc.f(), P(l->prefix + ".mel_conv.conv.bias"), l->melc.f(), (int)Bm,
                                         MB, l->H, T1, l->L, s));
        }
        DWS_HIP(hipStreamSynchronize(s));  // u0/u1 are freed on return
        melBm = Bm;
        return DWS_OK;
    }

    // DiffWaveBlock.forward (sashimi.py:143-184)
    int run_block(SLayer* l, const float* x, const float* addend, hipStream_t s) {
        Stage* st = stages[l->stage];
        const int H = l->H, Ls = l->L, nB = (int)B;
        const std::string& p = l->prefix;
        if (l->log2m > 0) {
            DWS_TRY(launch_ln(x, P(p + ".norm1.m"), P(p + ".norm1.s"), part_t.f() + l->pt_off, pt_total, st->y.f(), nB, H,
                              Ls, (size_t)Ls, s));
            FftTables* t = tables[l->log2m];
            FftConvArgs fa{};
            fa.u = st->y.f(); fa.g = st->g.f(); fa.D = P(p + ".layer.D");
            fa.tw = (const float2*)t->tw.p; fa.twp = (const float2*)t->twp.p;
            fa.kfa = (const float2*)l->kfa.p; fa.kfb = (const float2*)l->kfb.p; fa.kfs = (const float2*)l->kfs.p;
            fa.B = nB; fa.H = H; fa.L = Ls;
            DWS_TRY(launch_fftconv(l->log2m, fa, s));
            return run_tail(l, st, x, addend, s);
        }
        DWS_TRY(launch_ln(x, P(p + ".norm1.m"), P(p + ".norm1.s"), part_t.f() + l->pt_off, pt_total, st->U.f(), nB, H, Ls,
                          (size_t)2 * Ls, s));
        hipfftHandle plan;
        DWS_TRY(fft.get(0, 2 * Ls, nB * H, &plan));
        DWS_FFT(hipfftSetStream(plan, s));
        {
            ProfileScope ps("rocfft_r2c", s);
            DWS_FFT(hipfftExecR2C(plan, (hipfftReal*)st->U.p, (hipfftComplex*)st->Uf.p));
        }
        DWS_TRY(launch_spec_mul(st->Uf.f(), l->Kf.f(), nB, H, Ls + 1, s));
        DWS_TRY(fft.get(1, 2 * Ls, nB * H, &plan));
        DWS_FFT(hipfftSetStream(plan, s));
        {
            ProfileScope ps("rocfft_c2r", s);
            DWS_FFT(hipfftExecC2R(plan, (hipfftComplex*)st->Uf.p, (hipfftReal*)st->Y.p));
        }
        DWS_TRY(launch_s4_post(st->Y.f(), st->U.f(), P(p + ".layer.D"), st->g.f(), nB, H, Ls, s));
        return run_tail(l, st, x, addend, s);
    }

    // everything of the block after the S4 convolution (sashimi.py:177-184, s4.py:1435)
    int run_tail(SLayer* l, Stage* st, const float* x, const float* addend, hipStream_t s) {
        const int H = l->H, Ls = l->L, nB = (int)B;
        const std::string& p = l->prefix;
        if (l->mfma) {
            S4TailArgs t{};
            t.g = st->g.f(); t.x = x; t.Ao = l->Ao.f(); t.bo = P(p + ".layer.output_linear.0.bias");
            t.mel = melBm ? l->melc.f() : nullptr; t.mel_bstride = melBm > 1 ? 1 : 0;
            t.ln_m = P(p + ".norm2.m"); t.ln_s = P(p + ".norm2.s");
            t.A1 = l->A1.f(); t.b1 = P(p + ".ff.ff.0.conv.bias"); t.rs1 = l->rs1.f();
            t.A2 = l->A2.f(); t.b2 = P(p + ".ff.ff.2.conv.bias");
            t.addend = addend; t.out = l->out.f(); t.B = nB; t.L = Ls;
            return launch_s4_tail_mfma(H, t, s);
        }
        DWS_TRY(launch_pw_glu_res(st->g.f(), P(p + ".layer.output_linear.0.weight"), P(p + ".layer.output_linear.0.bias"),
                                  x, melBm ? l->melc.f() : nullptr, melBm > 1 ? 1 : 0, st->x1.f(), nB, H, Ls, s));
        DWS_TRY(launch_ln(st->x1.f(), P(p + ".norm2.m"), P(p + ".norm2.s"), nullptr, 0, st->n2.f(), nB, H, Ls, (size_t)Ls, s));
        DWS_TRY(launch_pw_conv(st->n2.f(), l->W1.f(), P(p + ".ff.ff.0.conv.bias"), st->ffu.f(), nB, H, FF * H, Ls, 1, s));
        DWS_TRY(launch_pw_res(st->ffu.f(), l->W2.f(), P(p + ".ff.ff.2.conv.bias"), st->x1.f(), addend, l->out.f(), nB,
                              FF * H, H, Ls, s));
        return DWS_OK;
    }

    int run_layer(SLayer* l, const float* x, const float* addend, hipStream_t s) {
        if (l->kind == L_BLOCK) return run_block(l, x, addend, s);
        if (l->mfma) {
            PwMfmaArgs a{};
            a.in = x; a.A = l->Ap.f(); a.bias = P(l->prefix + ".linear.conv.bias"); a.out = l->out.f();
            a.B = (int)B; a.p = l->p;
            if (l->kind == L_DOWN) { a.K = l->H * l->p; a.M = l->Hout; a.L = l->Lout; a.addend = nullptr; }
            else { a.K = l->H; a.M = l->Hout * l->p; a.L = l->L; a.addend = addend; }
            return launch_pw_mfma(l->kind == L_DOWN ? 0 : 1, a, s);
        }
        if (l->kind == L_DOWN)
            return launch_pw_downpool(x, l->Wp.f(), P(l->prefix + ".linear.conv.bias"), l->out.f(), (int)B, l->H, l->p,
                                      l->Hout, l->Lout, s);
        return launch_pw_uppool(x, l->Wp.f(), P(l->prefix + ".linear.conv.bias"), addend, l->out.f(), (int)B, l->H, l->p,
                                l->Hout, l->L, s);
    }

    int final_stage(const float* xin, float* out, float* tap, hipStream_t s) {
        DWS_TRY(launch_ln(xin, P("norm.m"), P("norm.s"), nullptr, 0, nfin.f(), (int)B, D, (int)L, (size_t)L, s));
        WnFinalArgs f{};
        f.skip = nfin.f(); f.Af = Af.f(); f.Wf = Wf.f(); f.bf = P("final_conv.0.conv.bias");
        f.Wz = P("final_conv.2.conv.weight"); f.bz = P("final_conv.2.conv.bias");
        f.out = out; f.tap = tap; f.scale = 1.f;
        f.B = (int)B; f.L = (int)L; f.Cout = Cout;
        return launch_wn_final(D, f, s);
    }

    const float* last_x = nullptr;

    // Sashimi.forward (sashimi.py:277-313)
    int forward(const float* audio, const float* steps, float* out, hipStream_t s) override {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "forward before prepare");
        if (dirty) DWS_TRY(commit(s));
        DWS_TRY(launch_init_conv(audio, Wi.f(), P("init_conv.0.conv.bias"), x_init.f(), (int)B, Cin, D, (int)L, s));
        DWS_TRY(launch_step_embed(steps, freq.f(), emb.f(), (int)B, Ein / 2, s));
        DWS_TRY(launch_linear_rows(emb.f(), P("fc_t1.weight"), P("fc_t1.bias"), h1.f(), (int)B, Ein, Emid, 1, s));
        DWS_TRY(launch_linear_rows(h1.f(), P("fc_t2.weight"), P("fc_t2.bias"), h2.f(), (int)B, Emid, Eout, 1, s));
        DWS_TRY(launch_linear_rows(h2.f(), Wt_all.f(), bt_all.f(), part_t.f(), (int)B, Eout, pt_total, 0, s));
        std::vector<const float*> stack;  // LIFO skip stack (sashimi.py:293-307)
        const float* x = x_init.f();
        for (auto* l : d_layers) {
            stack.push_back(x);
            DWS_TRY(run_layer(l, x, nullptr, s));
            x = l->out.f();
        }
        stack.push_back(x);
        for (size_t i = 0; i < c_layers.size(); ++i) {
            const float* add = nullptr;
            if (i + 1 == c_layers.size()) { add = stack.back(); stack.pop_back(); }
            DWS_TRY(run_layer(c_layers[i], x, add, s));
            x = c_layers[i]->out.f();
        }
        for (auto* l : u_layers) {
            const float* add = nullptr;
            if (l->kind == L_UP || unet) { add = stack.back(); stack.pop_back(); }
            DWS_TRY(run_layer(l, x, add, s));
            x = l->out.f();
        }
        last_x = x;
        DWS_TRY(final_stage(x, out, nullptr, s));
        DWS_HIP(hipGetLastError());
        return DWS_OK;
    }

    int read_tap(const char* tap, float* dst, int64_t capacity, hipStream_t s) override {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "read_tap before prepare/forward");
        const std::string t(tap);
        if (t == "pre_final") {
            DWS_CHECK(last_x, DWS_ERR_STATE, "read_tap before forward");
            DWS_CHECK(capacity >= B * D * L, DWS_ERR_INVALID, "tap buffer too small");
            DWS_TRY(scratch_out.ensure((size_t)B * Cout * L * 4));
            return final_stage(last_x, scratch_out.f(), dst, s);
        }
        if (t.rfind("k:", 0) == 0) {  // S4 kernel of the block with this prefix: L * k, k = [2][H][L] (s4.py:796-805)
            if (dirty) DWS_TRY(commit(s));
            for (auto* l : all)
                if (l->kind == L_BLOCK && l->prefix == t.substr(2)) {
                    const size_t n = (size_t)2 * l->H * l->L;
                    DWS_CHECK((size_t)capacity >= n, DWS_ERR_INVALID, "tap buffer too small");
                    DWS_TRY(build_kernel(l, s));  // regenerates the (unnormalised) time-domain kernel into scratch
                    DWS_HIP(hipMemcpyAsync(dst, ck.p, n * 4, hipMemcpyDeviceToDevice, s));
                    return DWS_OK;
                }
        }
        if (t.rfind("out:", 0) == 0) {  // output activation of the layer with this prefix
            for (auto* l : all)
                if (l->prefix == t.substr(4)) {
                    const size_t n = (l->kind == L_BLOCK) ? (size_t)B * l->H * l->L : (size_t)B * l->Hout * l->Lout;
                    DWS_CHECK((size_t)capacity >= n, DWS_ERR_INVALID, "tap buffer too small");
                    DWS_HIP(hipMemcpyAsync(dst, l->out.p, n * 4, hipMemcpyDeviceToDevice, s));
                    return DWS_OK;
                }
        }
        return set_error(DWS_ERR_INVALID, "unknown tap '%s'", tap);
    }
};

dws_model* make_sashimi(const dws_model_desc& d) {
    if (d.d_model <= 0 || d.n_layers <= 0 || d.n_pool < 0 || d.n_pool > DWS_MAX_POOL || d.expand <= 0 || d.ff <= 0 ||
        d.L <= 0) {
        set_error(DWS_ERR_INVALID, "sashimi: bad d_model/n_layers/pool/expand/ff/L");
        return nullptr;
    }
    int L = d.L;
    for (int i = 0; i < d.n_pool; ++i) {
        if (d.pool[i] <= 0 || L % d.pool[i] != 0) {
            set_error(DWS_ERR_INVALID, "sashimi: L=%d is not divisible by the pooling factors", d.L);
            return nullptr;
        }
        L /= d.pool[i];
    }
    return new SashimiModel(d);
}

}  // namespace dws
