// SaShiMi backbone behind the C ABI (`models/sashimi.py:187-313`, `models/s4.py`).
//
// Everything that does not depend on x_t is evaluated once per weight load in
// commit(): weight-norm folding, the S4 convolution kernels (Cauchy -> Woodbury
// -> bilinear factor -> irfft -> two-sided kernel -> rfft = K_f per layer; the
// reference regenerates these in all 30 layers on every reverse step,
// `s4.py:1388`) and, in set_condition(), the mel conditioner terms.
// The per-step path is then LN+emb -> rocFFT r2c -> spectrum multiply -> c2r ->
// D-skip+GELU -> 1x1+GLU+residual -> LN -> FF -> residual per block.
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "conditioner.h"
#include "fftconv.h"
#include "model.h"
#include "sashimi.h"
#include "sashimi_mfma.h"
#include "sashimi_train.h"
#include "wavenet.h"
#include "wavenet_backward.h"

namespace dws {

#define DWS_FFT(expr)                                                                                  \
    do {                                                                                               \
        rocfft_status _r = (expr);                                                                     \
        if (_r != rocfft_status_success)                                                               \
            return set_error(DWS_ERR_HIP, "%s failed: rocfft_status %d (%s:%d)", #expr, (int)_r, __FILE__, __LINE__); \
    } while (0)

// rocFFT, through its native API: batched 1-D real transforms, out of place, unscaled, rows packed back to back (the
// defaults of a plan without a description: input distance n reals / n/2+1 complex, output the other way round).  Used
// where the fused LDS FFT does not apply: the irfft(n = L) of the kernel generation (`s4.py:796-805`) and its adjoint,
// and the R2C / C2R pair of stage lengths the fused kernels do not cover (odd lengths; more than 16384 taps).
// A plan owns its work buffer and execution info, so rocfft_execute never allocates (it may run inside a stream capture).
struct RocfftPlan {
    rocfft_plan plan = nullptr;
    rocfft_execution_info info = nullptr;
    DevBuf work;
};

// rocfft_setup() / rocfft_cleanup() act on process-global state (plan repository, RTC cache, logging): set up once per
// process; never torn down from a model's destructor -- another model in the process may still hold plans.
static int rocfft_setup_once() {
    static std::once_flag once;
    static rocfft_status st = rocfft_status_success;
    std::call_once(once, [] { st = rocfft_setup(); });
    if (st != rocfft_status_success) return set_error(DWS_ERR_HIP, "rocfft_setup failed: rocfft_status %d", (int)st);
    return DWS_OK;
}

struct FftPlans {
    std::map<std::tuple<int, int, int>, RocfftPlan*> plans;  // (type, n, batch)
    ~FftPlans() {
        for (auto& kv : plans) {
            if (kv.second->info) rocfft_execution_info_destroy(kv.second->info);
            if (kv.second->plan) rocfft_plan_destroy(kv.second->plan);
            delete kv.second;
        }
    }
    // type 0: R2C rows of n reals (dist n) -> n/2+1 complex; type 1: C2R n/2+1 complex -> n reals (dist n)
    int get(int type, int n, int batch, RocfftPlan** out) {
        auto key = std::make_tuple(type, n, batch);
        auto it = plans.find(key);
        if (it == plans.end()) {
            DWS_TRY(rocfft_setup_once());
            RocfftPlan* p = new RocfftPlan();
            it = plans.emplace(key, p).first;      // owned by the map from here on (freed with it, also after an error)
            const size_t len[1] = {(size_t)n};
            DWS_FFT(rocfft_plan_create(&p->plan, rocfft_placement_notinplace,
                                       type == 0 ? rocfft_transform_type_real_forward : rocfft_transform_type_real_inverse,
                                       rocfft_precision_single, 1, len, (size_t)batch, nullptr));
            size_t wbytes = 0;
            DWS_FFT(rocfft_plan_get_work_buffer_size(p->plan, &wbytes));
            DWS_FFT(rocfft_execution_info_create(&p->info));
            if (wbytes) {
                DWS_TRY(p->work.ensure(wbytes));
                DWS_FFT(rocfft_execution_info_set_work_buffer(p->info, p->work.p, wbytes));
            }
        }
        DWS_CHECK(it->second->plan && it->second->info, DWS_ERR_HIP, "rocFFT plan (type %d, n %d, batch %d) was not created", type, n, batch);
        *out = it->second;
        return DWS_OK;
    }
    // one batched transform on stream s (the real inverse may use its input as scratch, as rocFFT documents)
    int exec(int type, int n, int batch, void* in, void* out, hipStream_t s) {
        RocfftPlan* p = nullptr;
        DWS_TRY(get(type, n, batch, &p));
        DWS_FFT(rocfft_execution_info_set_stream(p->info, (void*)s));
        void* ib[1] = {in};
        void* ob[1] = {out};
        DWS_FFT(rocfft_execute(p->plan, ib, ob, p->info));
        return DWS_OK;
    }
};

enum { L_BLOCK = 0, L_DOWN = 1, L_UP = 2 };

struct SLayer {
    int kind, H, L, p = 1, Hout = 0, Lout = 0;   // L / Lout: lengths of the CURRENT run (prepare rescales them)
    int L0 = 0, Lout0 = 0;                       // as configured (model L)
    int Lk = 0;                                  // S4 kernel length = the block's `L` buffer
    uint64_t Lk_version = 0;                     // int_params_version at which Lk was read back
    std::string prefix;
    int pt_off = 0;       // offset of this block's fc_t rows in the stacked projection
    int stage = 0;        // index into per-(H,L) workspaces
    DevBuf W1, W2, Wp;    // folded ff / pool weights
    DevBuf Ao, A1, A2, rs1, Ap;  // MFMA-packed copies (+ row sums of W1 for the folded LayerNorm)
    DevBuf Ao_c, A1_c, A2_c;     // H <= 64: the same weights with chain-ordered columns (sashimi_chain.hip)
    DevBuf Ao_c6, A1_c6, A2_c6;  // precision = bf16x6 / f16x3: split fragments in the 16-wide chain order (sashimi_chain6.hip)
    DevBuf wscale_c6;            // f16x3: the three matrices' power-of-two scales
    bool mfma = false, mfma2 = false;
    DevBuf Kf;            // rocFFT path: [H][L+1] complex spectrum of the two-sided kernel (n = 2L)
    DevBuf kfa, kfb, kfs; // fused path: pair-ordered spectrum at the power-of-two size (fftconv.h)
    const void *kfa_v = nullptr, *kfb_v = nullptr, *kfs_v = nullptr;   // what the convolution reads: the buffers above, or this block's
                                                                       // slice of its group's stacked spectra (KGroup)
    int grp = -1, gidx = 0;   // training: stacked kernel generation -- group (blocks of one shape) and position in it
    int log2m = 0;        // > 0: fused LDS FFT convolution is used for this block
    bool seg = false;     // stage longer than the largest LDS transform: segmented fused path (fftconv_seg_kernel)
    // training: the Cauchy products of the kernel generation (v, w dt, dt, r: `s4.py:740-775`) are kept per block so the
    // backward does not regenerate them (one Cauchy forward per block and step less); valid for commit cache_version
    DevBuf t_cv, t_cwdt, t_cdt, t_cr;
    uint64_t cache_version = ~0ull;
    DevBuf kfa_c, kfb_c, kfs_c, kfa_a, kfb_a, kfs_a;   // seg: spectra of the causal / anti-causal kernel half, shifted
    DevBuf melW0, melW1, melWc, melc;
    DevBuf out;           // activation produced by this layer
    // training path: saved activations, gradient w.r.t. `out`, transposed weights in A-fragment order
    DevBuf t_u, t_a, t_g, t_o, t_x1, t_n2, t_f1, t_ge, t_xr, dbuf;   // t_g = gelu(t_a), t_ge = gelu(t_f1): kept so the
                                                                    // weight gradients need no GELU recompute
    DevBuf tAo, tA1, tA2, tAp, tAoT, tA1T, tA2T, tApT;
    const float *pAo = nullptr, *pA1 = nullptr, *pA2 = nullptr, *pAp = nullptr;   // forward operands of the training GEMMs: the
                                                                                 // commit's packs where they exist, else tA*
};

struct Exec {             // one step of Sashimi.forward: out_node = layer(in_node) (+ add_node)
    SLayer* l;
    int in_node, add_node;
};

struct Stage {
    int H, L, L0 = 0;
    bool rocfft = false;  // some block of this stage needs the rocFFT path
    bool seg = false;     // stage runs the segmented fused convolution (L > 16384)
    bool force_rocfft = false;   // resolve_segmented_stages(): a checkpoint's kernels carry more taps than one segment holds, so
                                 // this stage runs on rocFFT whatever prepare() would pick from the configured length; holds
                                 // until the run length changes (the decision depends on L and the `L` buffers only, not on B)
    DevBuf U, Uf, Y, g, x1, n2, ffu, y;
    DevBuf d2, dh, dx1, du;  // training path gradients: [B][max(2,FF) H][L], [B][H][L] x 3
};

struct FftTables {
    DevBuf tw, twn, twp;
};

// Training: the S4 blocks of one shape (H, L) -- 12 + 12 + 6 in BASELINE config 5 -- generate their kernels, and run the adjoint of
// that generation, STACKED along H: every launch of the chain (s4_prep, Cauchy, Woodbury, rocFFT, re-placement, spectrum, pair
// order; and backwards) is indexed by a row h < H and is launched once over n * H rows.  Per block the chain is ~12 launches of
// 5-80 us each way, latency-bound: 30 blocks x 24 launches = 7.6 ms of a 134 ms step before, three chains each way after.
struct KGroup {
    int H = 0, L = 0, log2m = 0;
    std::vector<SLayer*> layers;
    DevBuf C, Bp, P, iwr, wim, logdt;      // stacked parameters: [2][Ht][N] complex, [Ht][N] complex x 2, [Ht][N] x 2, [Ht]
    DevBuf v, wdt, dt, r, kf;              // s4_prep / Cauchy products (read again by the adjoint), spectrum [2][Ht][Lh]
    DevBuf kfa, kfb, kfs;                  // pair-ordered spectra [Ht][M/2] x 2, [Ht][3]: a block reads its H rows
    DevBuf dKf, gC, gB, gP, giwr, gwim, glogdt, gD;   // adjoint: stacked spectrum gradient, stacked parameter gradients
    int pending = 0;                       // blocks of this backward whose correlation has not run yet
    CopyBatch stack, unstack;
    int Ht() const { return H * (int)layers.size(); }
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
    CopyBatch stack_fc_t, unstack_fc_t;   // per-layer fc_t tensors <-> their stacked buffers, one launch each
    bool freq_ready = false;
    DevBuf x_init, emb, h1, h2, part_t, nfin, scratch_out;
    // commit scratch
    DevBuf cv, cwdt, cdt, cr, ckf, ck, cK, cKf;
    int64_t melBm = 0;
    // training path
    std::vector<Exec> plan;
    DevBuf dx_init, ty, ta1, ta2, tAfT, tmp_pack, wpart, dWfold, lnpart, pool_scr, chain_tmp, dpt, dh2, dh1, dWt_all, dbt_all;
    DevBuf bpart, fpart, dKf, dKt, dkt, dkf, cgr, cgv, cgw, cpdt, dyb, dnf;
    uint64_t commit_version = 0, train_pack_version = ~0ull;
    bool keep_cauchy = false;         // set by the first forward_train: build_kernel then fills the per-block caches
    std::vector<KGroup*> kgroups;     // training commits: blocks grouped by shape (build_kernels_stacked)
    bool kernels_stacked = false;     // the last commit generated the kernels group by group
    bool trained_fwd = false;
    const float* train_audio = nullptr;
    DevBuf mel_in;                    // copy of the installed mel [Bm][MB][Tmel] (conditioner adjoint)
    DevBuf mel_u0, mel_u1;            // upsampler scratch of set_condition (kept: no allocation / wait per utterance)
    int mel_T = 0;
    CondTrainWs cws;
    DevBuf gW0f, gW1f, gWcf;

    // precision option: the H <= 128 tails on the 16-bit matrix cores (sashimi_chain6.hip): bf16x6 = 3-term bf16 split, six
    // products; f16x3 = 2-term fp16 split of scaled operands, three products
    bool bf16x6 = false, f16x3 = false;
    bool split_tails() const { return bf16x6 || f16x3; }
    int set_option(const std::string& key, const std::string& value) override {
        if (key == "precision") {
            if (value == "f32" || value == "bf16x6" || value == "f16x3") {
                const bool b6 = value == "bf16x6", f3 = value == "f16x3";
                if (b6 != bf16x6 || f3 != f16x3) { bf16x6 = b6; f16x3 = f3; dirty = true; drop_graph(); trained_fwd = false; }
                return DWS_OK;
            }
            return set_error(DWS_ERR_UNSUPPORTED, "sashimi: precision=%s is not built (f32 | bf16x6 | f16x3)", value.c_str());
        }
        if (key == "train_ln_fusion") {     // 0: separate LayerNorm passes in forward_train (parity / A-B runs); default 1
            train_ln_fusion = value != "0";
            return DWS_OK;
        }
        return dws_model::set_option(key, value);
    }

    ~SashimiModel() override {
        for (auto* g : kgroups) delete g;
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
        for (auto* l : all) { l->L0 = l->L; l->Lout0 = l->Lout; }
        for (auto* st : stages) st->L0 = st->L;
        add_param("norm.m", {1});
        add_param("norm.s", {1});
        wn("final_conv.0.conv", {D, D, 1});
        add_param("final_conv.2.conv.weight", {Cout, D, 1});
        add_param("final_conv.2.conv.bias", {Cout});
    }

    // commit(): the fold joins the commit's batched weight preparation (PrepBatch: run once, before anything reads `out`)
    PrepBatch prep, prep_train;
    int fold(const std::string& p, float* out, int O, int inner, hipStream_t) {
        prep.fold(P(p + ".weight_v"), P(p + ".weight_g"), out, O, inner);
        return DWS_OK;
    }

    // length the kernel of a block was set up for (its `L` buffer): read back only when an int64 buffer was re-sent
    int kernel_len(SLayer* l, hipStream_t s, int64_t* out) {
        int64_t Lbuf = l->Lk;
        if (l->Lk_version != int_params_version) {
            DWS_HIP(hipMemcpyAsync(&Lbuf, P(l->prefix + ".layer.kernel.kernel.L"), 8, hipMemcpyDeviceToHost, s));
            DWS_HIP(hipStreamSynchronize(s));
            l->Lk_version = int_params_version;
            l->Lk = (int)Lbuf;
        }
        *out = Lbuf;
        return DWS_OK;
    }

    // rocFFT path of a stage: zero-padded input rows, spectrum, output rows, and the two plans (allocated here, never
    // inside a stream capture)
    int ensure_rocfft_stage(Stage* st) {
        const size_t rows = (size_t)B * st->H, Ls = st->L;
        const bool fresh = st->U.bytes < rows * 2 * Ls * 4;
        DWS_TRY(st->U.ensure(rows * 2 * Ls * 4));
        if (fresh) DWS_HIP(hipMemset(st->U.p, 0, rows * 2 * Ls * 4));  // zero padding of the FFT input rows
        DWS_TRY(st->Uf.ensure(rows * (Ls + 1) * 8));
        DWS_TRY(st->Y.ensure(rows * 2 * Ls * 4));
        RocfftPlan* plan = nullptr;
        DWS_TRY(fft.get(0, 2 * st->L, (int)B * st->H, &plan));
        DWS_TRY(fft.get(1, 2 * st->L, (int)B * st->H, &plan));
        return DWS_OK;
    }

    // prepare() picks the segmented convolution from the CONFIGURED stage length; a checkpoint whose kernels were set
    // up for a longer l_max (its `L` buffers) can carry more taps than one segment holds: such a stage runs on rocFFT
    int resolve_segmented_stages(hipStream_t s) {
        for (auto* st : stages) {
            if (!st->seg && !st->force_rocfft) continue;
            bool fits = true;
            for (auto* l : all) {
                if (l->kind != L_BLOCK || stages[l->stage] != st) continue;
                int64_t Lk = 0;
                DWS_TRY(kernel_len(l, s, &Lk));
                if (Lk > 0 && !fftconv_seg_supported(st->L, (int)std::min<int64_t>(st->L, Lk))) fits = false;
            }
            if (!fits && st->seg) {
                st->seg = false;
                st->force_rocfft = true;      // prepare() keeps this stage on rocFFT (also for another B) until L changes
                drop_graph();
                DWS_TRY(ensure_rocfft_stage(st));
            } else if (fits && st->force_rocfft) {   // the `L` buffers were re-sent with shorter kernels: back to the segmented path
                st->force_rocfft = false;
                st->rocfft = false;           // (build_kernel sets it again if a block of the stage still needs the fallback)
                st->seg = true;
                drop_graph();
                DWS_TRY(st->y.ensure((size_t)B * st->H * st->L * 4));
            }
        }
        return DWS_OK;
    }

    // S4 convolution kernel of one block: parameters -> K_f   (s4.py:704-807, 1391-1403)
    int build_kernel(SLayer* l, hipStream_t s) {
        const int H = l->H, L = l->L, N = NS;
        const std::string k = l->prefix + ".layer.kernel.kernel";
        int64_t Lbuf = 0;
        DWS_TRY(kernel_len(l, s, &Lbuf));
        // the kernel is generated at its own length l_max; a run uses its first min(L, l_max) taps per direction
        // (`L_kernel`, s4.py:1387,805): shorter inputs truncate it, longer inputs keep l_max taps
        DWS_CHECK(Lbuf > 0 && Lbuf < (1 << 28), DWS_ERR_STATE,
                  "%s.L = %lld: C must have been through _setup_C(l_max) (s4.py:524-551) before it is handed to the engine",
                  k.c_str(), (long long)Lbuf);
        const int Lk = (int)Lbuf, Lh = Lk / 2 + 1, Lt = std::min(L, Lk);
        l->Lk = Lk;
        const std::string zname = "__z." + std::to_string(Lk), oname = "__omega." + std::to_string(Lk);
        DWS_CHECK(P(zname) && P(oname), DWS_ERR_STATE, "FFT nodes %s / %s were not handed to the engine", zname.c_str(),
                  oname.c_str());
        // training keeps the Cauchy products per block (kernel_backward reads them); sampling shares one scratch
        DevBuf& bv = keep_cauchy ? l->t_cv : cv;
        DevBuf& bw = keep_cauchy ? l->t_cwdt : cwdt;
        DevBuf& bd = keep_cauchy ? l->t_cdt : cdt;
        DevBuf& br = keep_cauchy ? l->t_cr : cr;
        DWS_TRY(bv.ensure((size_t)6 * H * N * 8));
        DWS_TRY(bw.ensure((size_t)H * N * 8));
        DWS_TRY(bd.ensure((size_t)H * 4));
        DWS_TRY(br.ensure((size_t)6 * H * Lh * 8));
        DWS_TRY(ckf.ensure((size_t)2 * H * Lh * 8));
        DWS_TRY(ck.ensure((size_t)2 * H * Lk * 4));
        DWS_TRY(launch_s4_prep(P(k + ".C"), P(k + ".B"), P(k + ".P"), P(k + ".inv_w_real"), P(k + ".w_imag"),
                               P(k + ".log_dt"), bv.f(), bw.f(), bd.f(), H, N, s));
        DWS_TRY(launch_cauchy_sym_fwd_bcast(bv.f(), P(zname), bw.f(), br.f(), 6 * H, N, Lh, H, s));
        DWS_TRY(launch_s4_woodbury(br.f(), P(oname), bd.f(), ckf.f(), H, Lh, (Lk % 2) == 0, s));
        l->cache_version = keep_cauchy ? commit_version + 1 : ~0ull;   // commit() bumps commit_version when it is done
        DWS_TRY(fft.exec(1, Lk, 2 * H, ckf.p, ck.p, s));
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
            DWS_TRY(launch_s4_twosided_pow2(ck.f(), cK.f(), H, Lt, Nf, Lk, s));
            DWS_TRY(launch_rfft_rows(lg, cK.f(), cKf.f(), t->tw.f(), t->twn.f(), H, s));
            DWS_TRY(launch_kf_permute(cKf.f(), l->kfa.f(), l->kfb.f(), l->kfs.f(), H, lg, s));
            if (l->grp < 0) { l->kfa_v = l->kfa.p; l->kfb_v = l->kfb.p; l->kfs_v = l->kfs.p; }   // (a block of a stacked commit keeps reading its group's rows: the tap "k:" regenerates into the block's own buffers)
            l->log2m = lg;
            l->seg = false;
        } else if (stages[l->stage]->seg) {
            // vocoding lengths: three pair-ordered spectra at M = 16384 -- the two-sided kernel, and its causal and
            // anti-causal halves alone shifted by half the transform ((-1)^k), see FftConvSegArgs
            DWS_CHECK(fftconv_seg_supported(L, Lt), DWS_ERR_UNSUPPORTED,
                      "%s: stage of %d samples with %d kernel taps per direction: more taps than one 16384-sample segment",
                      k.c_str(), L, Lt);
            lg = FFTCONV_SEG_LOG2M;
            const int M = 1 << lg, Nf = 2 * M;
            FftTables* t;
            DWS_TRY(get_tables(lg, &t, s));
            DWS_TRY(cK.ensure((size_t)H * Nf * 4));
            DWS_TRY(cKf.ensure((size_t)H * (M + 1) * 8));
            DevBuf* dst[3][3] = {{&l->kfa, &l->kfb, &l->kfs}, {&l->kfa_c, &l->kfb_c, &l->kfs_c}, {&l->kfa_a, &l->kfb_a, &l->kfs_a}};
            for (int which = 0; which < 3; ++which) {
                DWS_TRY(dst[which][0]->ensure((size_t)H * (M / 2) * 8));
                DWS_TRY(dst[which][1]->ensure((size_t)H * (M / 2) * 8));
                DWS_TRY(dst[which][2]->ensure((size_t)H * 3 * 8));
                DWS_TRY(launch_s4_twosided_pow2_part(ck.f(), cK.f(), H, Lt, Nf, Lk, which, s));
                DWS_TRY(launch_rfft_rows(lg, cK.f(), cKf.f(), t->tw.f(), t->twn.f(), H, s));
                DWS_TRY(launch_kf_permute_signed(cKf.f(), dst[which][0]->f(), dst[which][1]->f(), dst[which][2]->f(), H, lg,
                                                 which != 0, s));
            }
            l->log2m = 0;
            l->seg = true;
        } else {
            DWS_TRY(cK.ensure((size_t)H * 2 * L * 4));
            DWS_TRY(l->Kf.ensure((size_t)H * (L + 1) * 8));
            DWS_TRY(launch_s4_twosided(ck.f(), cK.f(), H, L, Lk, Lt, s));
            DWS_TRY(fft.exec(0, 2 * L, H, cK.p, l->Kf.p, s));
            l->log2m = 0;
            l->seg = false;
            stages[l->stage]->rocfft = true;
        }
        return DWS_OK;
    }

    // Can this commit generate the kernels group by group?  Training mode, every block on the fused convolution at its kernel's own
    // length (what train_supported() asks for anyway).  DWS_S4_KERNELS_PER_BLOCK=1: the per-block chain (same-box A/B).
    bool stacked_kernels_possible(hipStream_t s) {
        static const bool off = getenv("DWS_S4_KERNELS_PER_BLOCK") != nullptr;
        if (off || !keep_cauchy || getenv("DWS_SASHIMI_ROCFFT")) return false;
        for (auto* l : all) {
            if (l->kind != L_BLOCK) continue;
            int lg = 0;
            int64_t Lk = 0;
            if (kernel_len(l, s, &Lk) != DWS_OK || Lk != l->L || !fftconv_supported(l->L, &lg) || stages[l->stage]->seg) return false;
            if (!P("__z." + std::to_string(l->L)) || !P("__omega." + std::to_string(l->L))) return false;
        }
        return true;
    }

    // parameters -> K_f of every block, one chain per group of same-shaped blocks (s4.py:704-807, 1391-1403; build_kernel's chain)
    int build_kernels_stacked(hipStream_t s) {
        const int N = NS;
        for (auto* g : kgroups) g->layers.clear();      // (the run length may have changed since the last commit)
        for (auto* l : all) {
            if (l->kind != L_BLOCK) continue;
            KGroup* g = nullptr;
            for (auto* c : kgroups)
                if (c->H == l->H && c->L == l->L) g = c;
            if (!g) { g = new KGroup(); g->H = l->H; g->L = l->L; kgroups.push_back(g); }
            l->grp = (int)(std::find(kgroups.begin(), kgroups.end(), g) - kgroups.begin());
            l->gidx = (int)g->layers.size();
            g->layers.push_back(l);
        }
        for (auto* g : kgroups) {
            if (g->layers.empty()) continue;
            const int H = g->H, L = g->L, Ht = g->Ht(), Lk = L, Lh = Lk / 2 + 1;
            int lg = 0;
            DWS_CHECK(fftconv_supported(L, &lg), DWS_ERR_STATE, "stacked kernel generation on a stage without the fused convolution");
            const int M = 1 << lg, Nf = 2 * M;
            g->log2m = lg;
            FftTables* t;
            DWS_TRY(get_tables(lg, &t, s));
            const size_t HN = (size_t)Ht * N;
            DWS_TRY(g->C.ensure(2 * HN * 8)); DWS_TRY(g->Bp.ensure(HN * 8)); DWS_TRY(g->P.ensure(HN * 8));
            DWS_TRY(g->iwr.ensure(HN * 4)); DWS_TRY(g->wim.ensure(HN * 4)); DWS_TRY(g->logdt.ensure((size_t)Ht * 4));
            DWS_TRY(g->v.ensure(6 * HN * 8)); DWS_TRY(g->wdt.ensure(HN * 8)); DWS_TRY(g->dt.ensure((size_t)Ht * 4));
            DWS_TRY(g->r.ensure((size_t)6 * Ht * Lh * 8)); DWS_TRY(g->kf.ensure((size_t)2 * Ht * Lh * 8));
            DWS_TRY(g->kfa.ensure((size_t)Ht * (M / 2) * 8)); DWS_TRY(g->kfb.ensure((size_t)Ht * (M / 2) * 8));
            DWS_TRY(g->kfs.ensure((size_t)Ht * 3 * 8));
            DWS_TRY(ck.ensure((size_t)2 * Ht * Lk * 4)); DWS_TRY(cK.ensure((size_t)Ht * Nf * 4)); DWS_TRY(cKf.ensure((size_t)Ht * (M + 1) * 8));
            // the blocks' parameters into the stacked layout (one launch; the C planes of a block go to their plane of the stack)
            g->stack.begin();
            const size_t hn = (size_t)H * N;
            for (size_t i = 0; i < g->layers.size(); ++i) {
                const std::string k = g->layers[i]->prefix + ".layer.kernel.kernel";
                for (int c = 0; c < 2; ++c) g->stack.add(P(k + ".C") + c * hn * 2, g->C.f() + (c * HN + i * hn) * 2, hn * 2);
                g->stack.add(P(k + ".B"), g->Bp.f() + i * hn * 2, hn * 2);
                g->stack.add(P(k + ".P"), g->P.f() + i * hn * 2, hn * 2);
                g->stack.add(P(k + ".inv_w_real"), g->iwr.f() + i * hn, hn);
                g->stack.add(P(k + ".w_imag"), g->wim.f() + i * hn, hn);
                g->stack.add(P(k + ".log_dt"), g->logdt.f() + i * H, (size_t)H);
            }
            DWS_TRY(g->stack.run(s));
            const float* z = P("__z." + std::to_string(Lk));
            DWS_TRY(launch_s4_prep(g->C.f(), g->Bp.f(), g->P.f(), g->iwr.f(), g->wim.f(), g->logdt.f(), g->v.f(), g->wdt.f(), g->dt.f(), Ht, N, s));
            DWS_TRY(launch_cauchy_sym_fwd_bcast(g->v.f(), z, g->wdt.f(), g->r.f(), 6 * Ht, N, Lh, Ht, s));
            DWS_TRY(launch_s4_woodbury(g->r.f(), P("__omega." + std::to_string(Lk)), g->dt.f(), g->kf.f(), Ht, Lh, (Lk % 2) == 0, s));
            DWS_TRY(fft.exec(1, Lk, 2 * Ht, g->kf.p, ck.p, s));
            DWS_TRY(launch_s4_twosided_pow2(ck.f(), cK.f(), Ht, L, Nf, Lk, s));
            DWS_TRY(launch_rfft_rows(lg, cK.f(), cKf.f(), t->tw.f(), t->twn.f(), Ht, s));
            DWS_TRY(launch_kf_permute(cKf.f(), g->kfa.f(), g->kfb.f(), g->kfs.f(), Ht, lg, s));
            for (size_t i = 0; i < g->layers.size(); ++i) {
                SLayer* l = g->layers[i];
                l->kfa_v = g->kfa.f() + i * (size_t)H * (M / 2) * 2;
                l->kfb_v = g->kfb.f() + i * (size_t)H * (M / 2) * 2;
                l->kfs_v = g->kfs.f() + i * (size_t)H * 3 * 2;
                l->Lk = Lk; l->log2m = lg; l->seg = false;
                l->cache_version = commit_version + 1;      // (commit() bumps commit_version when it is done)
            }
        }
        return DWS_OK;
    }

    int commit(hipStream_t s) override {
        if (B > 0) DWS_TRY(resolve_segmented_stages(s));
        // Pass 1: every fold, fragment pack and row sum of the model as ONE batched preparation (two launches: the folds, then
        // the packs that read them).  Pass 2: what reads the folded weights through its own launches (chain-ordered and
        // split fragments) and the S4 kernel generation.
        prep.begin();
        DWS_TRY(Wi.ensure((size_t)D * Cin * 4));
        DWS_TRY(fold("init_conv.0.conv", Wi.f(), D, Cin, s));
        DWS_TRY(Wt_all.ensure((size_t)pt_total * Eout * 4));
        DWS_TRY(bt_all.ensure((size_t)pt_total * 4));
        stack_fc_t.begin();
        for (auto* l : all) {
            if (l->kind == L_BLOCK) {
                const int H = l->H;
                stack_fc_t.add(P(l->prefix + ".fc_t.weight"), Wt_all.f() + (size_t)l->pt_off * Eout, (size_t)H * Eout);
                stack_fc_t.add(P(l->prefix + ".fc_t.bias"), bt_all.f() + l->pt_off, (size_t)H);
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
                    prep.pack(P(l->prefix + ".layer.output_linear.0.weight"), l->Ao.f(), 2 * H, H);
                    prep.pack(l->W1.f(), l->A1.f(), FF * H, H);
                    prep.pack(l->W2.f(), l->A2.f(), H, FF * H);
                    prep.row_sum(l->W1.f(), l->rs1.f(), FF * H, H);
                }
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
                // shapes the fused pooling kernel does not cover (e.g. 128 -> 64 channels of unet_d32) still run on MFMA:
                // explicit rearrangement + the position-tile GEMM of the training path
                l->mfma2 = !l->mfma && tapconv_mfma_supported(O, K, 0, 1) && !getenv("DWS_SASHIMI_GENERIC");
                if (l->mfma || l->mfma2) {
                    DWS_TRY(l->Ap.ensure((size_t)O * K * 4));
                    prep.pack(l->Wp.f(), l->Ap.f(), O, K);
                }
            }
        }
        DWS_TRY(Wf.ensure((size_t)D * D * 4));
        DWS_TRY(fold("final_conv.0.conv", Wf.f(), D, D, s));
        if (wn_final_mfma_supported(D)) {
            DWS_TRY(Af.ensure((size_t)D * D * 4));
            prep.pack(Wf.f(), Af.f(), D, D);
        }
        DWS_TRY(prep.run(s));
        kernels_stacked = stacked_kernels_possible(s);
        for (auto* l : all) {
            if (l->kind == L_BLOCK) {
                const int H = l->H;
                if (l->mfma) {
                    if (s4_tail_chain_supported(H, FF)) {   // chain-ordered columns for the register-chained tail kernel
                        DWS_TRY(chain_tmp.ensure((size_t)FF * H * H * 4));
                        DWS_TRY(l->Ao_c.ensure((size_t)2 * H * H * 4));
                        DWS_TRY(l->A1_c.ensure((size_t)FF * H * H * 4));
                        DWS_TRY(l->A2_c.ensure((size_t)FF * H * H * 4));
                        DWS_TRY(launch_chain_permute_cols(P(l->prefix + ".layer.output_linear.0.weight"), chain_tmp.f(), 2 * H, H, s));
                        DWS_TRY(launch_pack_a_frag(chain_tmp.f(), l->Ao_c.f(), 2 * H, H, s));
                        DWS_TRY(launch_chain_permute_cols(l->W1.f(), chain_tmp.f(), FF * H, H, s));
                        DWS_TRY(launch_pack_a_frag(chain_tmp.f(), l->A1_c.f(), FF * H, H, s));
                        DWS_TRY(launch_chain_permute_cols(l->W2.f(), chain_tmp.f(), H, FF * H, s));
                        DWS_TRY(launch_pack_a_frag(chain_tmp.f(), l->A2_c.f(), H, FF * H, s));
                    }
                    const int split = f16x3 ? WN_SPLIT_F16X3 : WN_SPLIT_BF16X6;
                    const size_t wb = 2 * (size_t)wn_split_terms(split);            // bytes per packed weight
                    float* sc = nullptr;
                    if (f16x3) {     // (H >= 256: the LDS-tile kernel's split instances scale the fp32 fragments in registers)
                        DWS_TRY(l->wscale_c6.ensure(3 * 4));
                        sc = l->wscale_c6.f();
                        DWS_TRY(launch_weight_scale(P(l->prefix + ".layer.output_linear.0.weight"), (size_t)2 * H * H,
                                                    P(l->prefix + ".layer.output_linear.0.bias"), 2 * H, nullptr, 0, sc, s));
                        DWS_TRY(launch_weight_scale(l->W1.f(), (size_t)FF * H * H, P(l->prefix + ".ff.ff.0.conv.bias"), FF * H, nullptr, 0, sc + 1, s));
                        DWS_TRY(launch_weight_scale(l->W2.f(), (size_t)FF * H * H, P(l->prefix + ".ff.ff.2.conv.bias"), H, nullptr, 0, sc + 2, s));
                    }
                    if (split_tails() && s4_tail_chain6_supported(H, FF)) {
                        DWS_TRY(chain_tmp.ensure((size_t)FF * H * H * 4));
                        DWS_TRY(l->Ao_c6.ensure((size_t)2 * H * H * wb));
                        DWS_TRY(l->A1_c6.ensure((size_t)FF * H * H * wb));
                        DWS_TRY(l->A2_c6.ensure((size_t)FF * H * H * wb));
                        DWS_TRY(launch_chain16_permute_cols(P(l->prefix + ".layer.output_linear.0.weight"), chain_tmp.f(), 2 * H, H, s));
                        DWS_TRY(launch_pack_a_bx6(chain_tmp.f(), l->Ao_c6.p, 2 * H, H, split, sc, s));
                        DWS_TRY(launch_chain16_permute_cols(l->W1.f(), chain_tmp.f(), FF * H, H, s));
                        DWS_TRY(launch_pack_a_bx6(chain_tmp.f(), l->A1_c6.p, FF * H, H, split, sc ? sc + 1 : nullptr, s));
                        DWS_TRY(launch_chain16_permute_cols(l->W2.f(), chain_tmp.f(), H, FF * H, s));
                        DWS_TRY(launch_pack_a_bx6(chain_tmp.f(), l->A2_c6.p, H, FF * H, split, sc ? sc + 2 : nullptr, s));
                    } else if (split_tails() && s4_tail_wide6_supported(H, FF)) {   // one blob [Wo | W1 | W2], k-block-major fragments
                        DWS_TRY(chain_tmp.ensure((size_t)FF * H * H * 4));
                        DWS_TRY(l->Ao_c6.ensure((size_t)(2 + 2 * FF) * H * H * wb));
                        l->A1_c6.release(); l->A2_c6.release();
                        char* blob = static_cast<char*>(l->Ao_c6.p);
                        DWS_TRY(launch_chain16_permute_cols(P(l->prefix + ".layer.output_linear.0.weight"), chain_tmp.f(), 2 * H, H, s));
                        DWS_TRY(launch_pack_a_bx6_kmajor(chain_tmp.f(), blob, 2 * H, H, split, sc, s));
                        DWS_TRY(launch_chain16_permute_cols(l->W1.f(), chain_tmp.f(), FF * H, H, s));
                        DWS_TRY(launch_pack_a_bx6_kmajor(chain_tmp.f(), blob + (size_t)2 * H * H * wb, FF * H, H, split, sc ? sc + 1 : nullptr, s));
                        DWS_TRY(launch_chain16_permute_cols(l->W2.f(), chain_tmp.f(), H, FF * H, s));
                        DWS_TRY(launch_pack_a_bx6_kmajor(chain_tmp.f(), blob + (size_t)(2 + FF) * H * H * wb, H, FF * H, split, sc ? sc + 2 : nullptr, s));
                    } else {
                        l->Ao_c6.release(); l->A1_c6.release(); l->A2_c6.release();
                    }
                }
                if (!kernels_stacked) { l->grp = -1; DWS_TRY(build_kernel(l, s)); }
            }
        }
        if (kernels_stacked) DWS_TRY(build_kernels_stacked(s));
        DWS_TRY(stack_fc_t.run(s));
        if (!freq_ready) {   // depends on the embedding width only: uploaded (and waited for) once, not on every commit --
                             // a training step commits once, and a blocking wait there keeps the host from running ahead of the GPU
            const int half = Ein / 2;
            std::vector<float> f(half);
            const float e = (float)(-(std::log(10000.0) / (half - 1)));
            for (int i = 0; i < half; ++i) f[i] = (float)std::exp((double)((float)i * e));
            DWS_TRY(freq.ensure((size_t)half * 4));
            DWS_HIP(hipMemcpyAsync(freq.p, f.data(), (size_t)half * 4, hipMemcpyHostToDevice, s));
            DWS_HIP(hipStreamSynchronize(s));
            freq_ready = true;
        }
        dirty = false;
        ++commit_version;
        melBm = 0;
        return DWS_OK;
    }

    int prepare(int64_t nB, int64_t nL) override {
        DWS_CHECK(nB > 0 && nL > 0, DWS_ERR_INVALID, "prepare: B=%lld L=%lld", (long long)nB, (long long)nL);
        int span = 1;
        for (int p : pool) span *= p;
        DWS_CHECK(nL % span == 0, DWS_ERR_INVALID, "sashimi: input length %lld is not divisible by the pooling factors (%d)",
                  (long long)nL, span);
        if (nB != B || nL != L) { drop_graph(); melBm = 0; trained_fwd = false; }
        if (nL != L) {
            // variable-length run (s4.py:1387): every stage length scales with the input; the S4 kernels keep their
            // own length (the `L` buffers) and contribute min(run, l_max) taps per direction
            for (auto* l : all) {
                l->L = (int)((int64_t)l->L0 * nL / d.L);
                l->Lout = (int)((int64_t)l->Lout0 * nL / d.L);
            }
            for (auto* st : stages) {
                st->L = (int)((int64_t)st->L0 * nL / d.L);
                st->rocfft = false; st->seg = false; st->force_rocfft = false;
            }
            dirty = true;   // K_f depends on the run length (two-sided assembly, transform size)
        }
        B = nB; L = nL;
        for (auto* st : stages) {
            const size_t rows = (size_t)B * st->H, Ls = st->L;
            int lg = 0;
            const bool no_fused = !fftconv_supported((int)Ls, &lg) || getenv("DWS_SASHIMI_ROCFFT");
            // beyond the largest LDS transform: segmented fused path when the kernels (at most the configured stage
            // length of taps per direction, s4.py:1387) fit one segment
            st->seg = no_fused && !st->force_rocfft && !getenv("DWS_SASHIMI_ROCFFT") &&
                      fftconv_seg_supported((int)Ls, std::min((int)Ls, st->L0));
            const bool need_rocfft = no_fused && !st->seg;
            if (need_rocfft) {
                B = nB;
                DWS_TRY(ensure_rocfft_stage(st));
            } else {
                DWS_TRY(st->y.ensure(rows * Ls * 4));
            }
            DWS_TRY(st->g.ensure(rows * Ls * 4));
            DWS_TRY(st->x1.ensure(rows * Ls * 4));
            DWS_TRY(st->n2.ensure(rows * Ls * 4));
            DWS_TRY(st->ffu.ensure(rows * FF * Ls * 4));
        }
        size_t pool_max = 4;
        for (auto* l : all) {
            const size_t n = (l->kind == L_BLOCK) ? (size_t)B * l->H * l->L : (size_t)B * l->Hout * l->Lout;
            DWS_TRY(l->out.ensure(n * 4));
            if (l->kind != L_BLOCK) pool_max = std::max(pool_max, std::max((size_t)B * l->H * l->L, n) * 4);
        }
        DWS_TRY(pool_scr.ensure(pool_max));
        DWS_TRY(x_init.ensure((size_t)B * D * L * 4));
        DWS_TRY(nfin.ensure((size_t)B * D * L * 4));
        DWS_TRY(emb.ensure((size_t)B * Ein * 4));
        DWS_TRY(h1.ensure((size_t)B * Emid * 4));
        DWS_TRY(h2.ensure((size_t)B * Eout * 4));
        DWS_TRY(part_t.ensure((size_t)B * pt_total * 4));
        if (!zero_row.p) {     // (allocated here, never inside a stream capture)
            DWS_TRY(zero_row.ensure((size_t)D * 4));
            DWS_HIP(hipMemset(zero_row.p, 0, (size_t)D * 4));
        }
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
        DevBuf& u0 = mel_u0;
        DevBuf& u1 = mel_u1;
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
        DWS_TRY(mel_in.ensure((size_t)Bm * MB * Tmel * 4));
        DWS_HIP(hipMemcpyAsync(mel_in.p, mel, (size_t)Bm * MB * Tmel * 4, hipMemcpyDeviceToDevice, s));
        mel_T = (int)Tmel;
        melBm = Bm;
        return DWS_OK;
    }

    // The tail of `l` can also produce the S4 input of the block that runs next (LN1 + step embedding fused into its
    // epilogue) when that block sits on the same stage and both run the fused paths.
    bool feeds_next(const SLayer* l, const SLayer* next) const {
        return next && l->kind == L_BLOCK && next->kind == L_BLOCK && next->stage == l->stage && l->mfma &&
               (next->log2m > 0 || next->seg) && !getenv("DWS_SASHIMI_NO_LN_FUSION");
    }

    // A pooling layer on the fused MFMA kernel whose whole output column sits in one workgroup can emit the LN1 + step
    // embedding of the block that follows it (the first block of the new stage), when that block reads the stage's y buffer.
    bool pool_feeds(const SLayer* l, const SLayer* next) const {
        if (!next || l->kind == L_BLOCK || next->kind != L_BLOCK || !l->mfma || getenv("DWS_SASHIMI_NO_LN_FUSION")) return false;
        const int M = (l->kind == L_DOWN) ? l->Hout : l->Hout * l->p;
        return pw_mfma_ln_supported(M) && (next->log2m > 0 || next->seg);
    }

    // DiffWaveBlock.forward (sashimi.py:143-184).  y_ready: the previous block's tail already wrote this block's S4 input
    // into the stage's y buffer; next: the block whose S4 input this block's tail should write (or null).
    // LayerNorm a tail kernel applies to ITS OUTPUT on the way out (the tile is still in LDS / registers): the next block's
    // LN1 + step-embedding projection into the stage's y buffer, or -- after the last block -- the network's final norm
    // (`sashimi.py:309`) into nfin
    struct OutLN {
        const float *m = nullptr, *s = nullptr, *e = nullptr;
        int e_stride = 0, e_tstride = 0;
        const int* e_step = nullptr;
        float* y = nullptr;
    };
    OutLN next_ln(Stage* st, const SLayer* next) {
        OutLN o;
        o.m = P(next->prefix + ".norm1.m"); o.s = P(next->prefix + ".norm1.s");
        o.e = pt_base() + next->pt_off; o.e_stride = pt_bstride(); o.e_step = step_idx; o.e_tstride = pt_total;
        o.y = st->y.f();
        return o;
    }
    DevBuf zero_row;     // [max H] zeros: the "step embedding" of the final norm
    bool final_ln_fused = false;   // set by forward(): the last block's tail wrote norm(x) into nfin

    int run_block(SLayer* l, const float* x, const float* addend, bool y_ready, const OutLN* next, hipStream_t s) {
        Stage* st = stages[l->stage];
        const int H = l->H, Ls = l->L, nB = (int)B;
        const std::string& p = l->prefix;
        if (l->log2m > 0) {
            if (!y_ready)
                DWS_TRY(launch_ln(x, P(p + ".norm1.m"), P(p + ".norm1.s"), pt_base() + l->pt_off, pt_bstride(), st->y.f(), nB,
                                  H, Ls, (size_t)Ls, s, step_idx, pt_total));
            FftTables* t = tables[l->log2m];
            FftConvArgs fa{};
            fa.u = st->y.f(); fa.g = st->g.f(); fa.D = P(p + ".layer.D");
            fa.tw = (const c2*)t->tw.p; fa.twp = (const c2*)t->twp.p;
            fa.kfa = (const c2*)l->kfa_v; fa.kfb = (const c2*)l->kfb_v; fa.kfs = (const c2*)l->kfs_v;
            fa.B = nB; fa.H = H; fa.L = Ls;
            DWS_TRY(launch_fftconv(l->log2m, fa, s));
            return run_tail(l, st, x, addend, next, s);
        }
        if (l->seg) {
            if (!y_ready)
                DWS_TRY(launch_ln(x, P(p + ".norm1.m"), P(p + ".norm1.s"), pt_base() + l->pt_off, pt_bstride(), st->y.f(), nB,
                                  H, Ls, (size_t)Ls, s, step_idx, pt_total));
            FftTables* t = tables[FFTCONV_SEG_LOG2M];
            FftConvSegArgs fa{};
            fa.u = st->y.f(); fa.g = st->g.f(); fa.D = P(p + ".layer.D");
            fa.tw = (const c2*)t->tw.p; fa.twp = (const c2*)t->twp.p;
            fa.kfa[0] = (const c2*)l->kfa.p; fa.kfb[0] = (const c2*)l->kfb.p; fa.kfs[0] = (const c2*)l->kfs.p;
            fa.kfa[1] = (const c2*)l->kfa_c.p; fa.kfb[1] = (const c2*)l->kfb_c.p; fa.kfs[1] = (const c2*)l->kfs_c.p;
            fa.kfa[2] = (const c2*)l->kfa_a.p; fa.kfb[2] = (const c2*)l->kfb_a.p; fa.kfs[2] = (const c2*)l->kfs_a.p;
            fa.B = nB; fa.H = H; fa.L = Ls;
            DWS_TRY(launch_fftconv_seg(fa, s));
            return run_tail(l, st, x, addend, next, s);
        }
        DWS_TRY(launch_ln(x, P(p + ".norm1.m"), P(p + ".norm1.s"), pt_base() + l->pt_off, pt_bstride(), st->U.f(), nB, H, Ls,
                          (size_t)2 * Ls, s, step_idx, pt_total));
        {
            ProfileScope ps("rocfft_r2c", s);
            DWS_TRY(fft.exec(0, 2 * Ls, nB * H, st->U.p, st->Uf.p, s));
        }
        DWS_TRY(launch_spec_mul(st->Uf.f(), l->Kf.f(), nB, H, Ls + 1, s));
        {
            ProfileScope ps("rocfft_c2r", s);
            DWS_TRY(fft.exec(1, 2 * Ls, nB * H, st->Uf.p, st->Y.p, s));
        }
        DWS_TRY(launch_s4_post(st->Y.f(), st->U.f(), P(p + ".layer.D"), st->g.f(), nB, H, Ls, s));
        return run_tail(l, st, x, addend, next, s);
    }

    // everything of the block after the S4 convolution (sashimi.py:177-184, s4.py:1435)
    int n_tail_split = 0, n_tail_f32 = 0;     // block-tail launches of the last forward, by arithmetic
    int run_tail(SLayer* l, Stage* st, const float* x, const float* addend, const OutLN* next, hipStream_t s) {
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
            t.Ao_c = l->Ao_c.f(); t.A1_c = l->A1_c.f(); t.A2_c = l->A2_c.f();
            if (split_tails()) {
                t.split_on = 1;
                t.Ao_c6 = l->Ao_c6.p; t.A1_c6 = l->A1_c6.p; t.A2_c6 = l->A2_c6.p;
                t.split_c6 = f16x3 ? WN_SPLIT_F16X3 : WN_SPLIT_BF16X6;
                t.wscale_c6 = f16x3 ? l->wscale_c6.f() : nullptr;
            }
            if (next) {     // next block: feeds_next(l, next) holds, the stage's y buffer is free once this block's convolution ran
                t.ynext = next->y;
                t.n1_m = next->m; t.n1_s = next->s;
                t.e_next = next->e; t.e_stride = next->e_stride;
                t.e_step = next->e_step; t.e_tstride = next->e_tstride;
            }
            bool ran_split = false;
            const int rc = launch_s4_tail_mfma(H, t, s, &ran_split);
            ++(ran_split ? n_tail_split : n_tail_f32);
            return rc;
        }
        ++n_tail_f32;          // plain-FMA block (channel counts the MFMA tiling does not cover): no split instance
        DWS_TRY(launch_pw_glu_res(st->g.f(), P(p + ".layer.output_linear.0.weight"), P(p + ".layer.output_linear.0.bias"),
                                  x, melBm ? l->melc.f() : nullptr, melBm > 1 ? 1 : 0, st->x1.f(), nB, H, Ls, s));
        DWS_TRY(launch_ln(st->x1.f(), P(p + ".norm2.m"), P(p + ".norm2.s"), nullptr, 0, st->n2.f(), nB, H, Ls, (size_t)Ls, s));
        DWS_TRY(launch_pw_conv(st->n2.f(), l->W1.f(), P(p + ".ff.ff.0.conv.bias"), st->ffu.f(), nB, H, FF * H, Ls, 1, s));
        DWS_TRY(launch_pw_res(st->ffu.f(), l->W2.f(), P(p + ".ff.ff.2.conv.bias"), st->x1.f(), addend, l->out.f(), nB,
                              FF * H, H, Ls, s));
        return DWS_OK;
    }

    int run_layer(SLayer* l, const float* x, const float* addend, hipStream_t s, bool y_ready = false, const OutLN* next = nullptr) {
        if (l->kind == L_BLOCK) return run_block(l, x, addend, y_ready, next, s);
        if (l->mfma) {
            PwMfmaArgs a{};
            a.in = x; a.A = l->Ap.f(); a.bias = P(l->prefix + ".linear.conv.bias"); a.out = l->out.f();
            a.B = (int)B; a.p = l->p;
            if (l->kind == L_DOWN) { a.K = l->H * l->p; a.M = l->Hout; a.L = l->Lout; a.addend = nullptr; }
            else { a.K = l->H; a.M = l->Hout * l->p; a.L = l->L; a.addend = addend; }
            if (next) {     // pool_feeds(l, next block) holds: the first block of the new stage gets its S4 input from here
                a.ln_y = next->y; a.ln_m = next->m; a.ln_s = next->s;
                a.ln_e = next->e; a.ln_e_stride = next->e_stride; a.ln_step = next->e_step; a.ln_e_tstride = next->e_tstride;
            }
            return launch_pw_mfma(l->kind == L_DOWN ? 0 : 1, a, s);
        }
        if (l->mfma2) {
            const float* bias = P(l->prefix + ".linear.conv.bias");
            if (l->kind == L_DOWN) {
                DWS_TRY(launch_pool_rearrange(x, pool_scr.f(), nullptr, 0, 0, (int)B, l->H, l->p, l->Lout, s));
                return gemm(l->Ap.f(), l->Hout, l->H * l->p, pool_scr.f(), l->out.f(), l->Lout, 2, bias, nullptr, nullptr, nullptr,
                            nullptr, s);
            }
            DWS_TRY(gemm(l->Ap.f(), l->Hout * l->p, l->H, x, pool_scr.f(), l->L, 2, bias, nullptr, nullptr, nullptr, nullptr, s));
            return launch_pool_rearrange(pool_scr.f(), l->out.f(), addend, 1, 0, (int)B, l->Hout, l->p, l->L, s);
        }
        if (l->kind == L_DOWN)
            return launch_pw_downpool(x, l->Wp.f(), P(l->prefix + ".linear.conv.bias"), l->out.f(), (int)B, l->H, l->p,
                                      l->Hout, l->Lout, s);
        return launch_pw_uppool(x, l->Wp.f(), P(l->prefix + ".linear.conv.bias"), addend, l->out.f(), (int)B, l->H, l->p,
                                l->Hout, l->L, s);
    }

    int final_stage(const float* xin, float* out, float* tap, hipStream_t s, bool ln_done = false) {
        if (!ln_done) DWS_TRY(launch_ln(xin, P("norm.m"), P("norm.s"), nullptr, 0, nfin.f(), (int)B, D, (int)L, (size_t)L, s));
        WnFinalArgs f{};
        f.skip = nfin.f(); f.Af = Af.f(); f.Wf = Wf.f(); f.bf = P("final_conv.0.conv.bias");
        f.Wz = P("final_conv.2.conv.weight"); f.bz = P("final_conv.2.conv.bias");
        f.out = out; f.tap = tap; f.scale = 1.f;
        f.B = (int)B; f.L = (int)L; f.Cout = Cout;
        return launch_wn_final(D, f, s);
    }

    const float* last_x = nullptr;

    // everything of the forward that depends on the diffusion step only (`sashimi.py:287-289,151`; a1, a2 of SURVEY 8):
    // embedding -> MLP -> every block's fc_t as one stacked GEMV, for `rows` step values (one wave per output row: a row's
    // result does not depend on how many rows the launch carries)
    int embed_rows(const float* steps, int rows, float* emb_, float* h1_, float* h2_, float* pt, hipStream_t s) {
        DWS_TRY(launch_step_embed(steps, freq.f(), emb_, rows, Ein / 2, s));
        DWS_TRY(launch_linear_rows(emb_, P("fc_t1.weight"), P("fc_t1.bias"), h1_, rows, Ein, Emid, 1, s));
        DWS_TRY(launch_linear_rows(h1_, P("fc_t2.weight"), P("fc_t2.bias"), h2_, rows, Emid, Eout, 1, s));
        DWS_TRY(launch_linear_rows(h2_, Wt_all.f(), bt_all.f(), pt, rows, Eout, pt_total, 0, s));
        return DWS_OK;
    }

    // Step table of a sampler run (sampler.hip): every clip of a reverse step is at the same t (`generate.py:50`), so the
    // projections are evaluated once for t = 0..T-1 -- tab_pt [T][pt_total] -- and the captured step's LayerNorm / tail
    // kernels read row *step_idx: no embedding kernels in a replay.
    DevBuf tab_steps, tab_emb, tab_h1, tab_h2, tab_pt;
    int tab_T = 0;
    uint64_t tab_version = ~0ull;
    const float* pt_base() const { return step_idx ? tab_pt.f() : part_t.f(); }
    int pt_bstride() const { return step_idx ? 0 : pt_total; }
    int build_step_table(int T, hipStream_t s) override {
        if (dirty) DWS_TRY(commit(s));
        if (tab_T == T && tab_version == commit_version) return DWS_OK;
        drop_graph();   // a captured step holds pointers into the old table
        DWS_TRY(tab_steps.ensure((size_t)T * 4));
        DWS_TRY(tab_emb.ensure((size_t)T * Ein * 4));
        DWS_TRY(tab_h1.ensure((size_t)T * Emid * 4));
        DWS_TRY(tab_h2.ensure((size_t)T * Eout * 4));
        DWS_TRY(tab_pt.ensure((size_t)T * pt_total * 4));
        DWS_TRY(launch_iota_f32(tab_steps.f(), T, s));   // steps[t] = float(t), as `generate.py:50` feeds them
        DWS_TRY(embed_rows(tab_steps.f(), T, tab_emb.f(), tab_h1.f(), tab_h2.f(), tab_pt.f(), s));
        tab_T = T;
        tab_version = commit_version;
        return DWS_OK;
    }

    // Sashimi.forward (sashimi.py:277-313)
    int forward(const float* audio, const float* steps, float* out, hipStream_t s) override {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "forward before prepare");
        trained_fwd = false;   // this forward (eval call, or a step of the sampler) overwrites the activations a pending backward needs
        if (dirty) DWS_TRY(commit(s));
        DWS_CHECK(!step_idx || (tab_T > 0 && tab_version == commit_version), DWS_ERR_STATE, "step-table forward without a current table");
        DWS_CHECK(step_idx || steps, DWS_ERR_INVALID, "forward: steps == null");
        n_tail_split = n_tail_f32 = 0;     // tap "split_launches": which arithmetic THIS forward's tails ran
        if (!step_idx) DWS_TRY(embed_rows(steps, (int)B, emb.f(), h1.f(), h2.f(), part_t.f(), s));
        std::vector<const float*> stack;  // LIFO skip stack (sashimi.py:293-307)
        const float* x = x_init.f();
        // execution order; a block whose successor is a fused block of the same stage also writes that block's S4 input
        std::vector<SLayer*> order(d_layers);
        order.insert(order.end(), c_layers.begin(), c_layers.end());
        order.insert(order.end(), u_layers.begin(), u_layers.end());
        const bool ln_fusion = getenv("DWS_SASHIMI_NO_LN_FUSION") == nullptr;   // (read per call: tests switch it inside one process)
        // the first block's LN1 + step embedding comes out of the init-conv pass when that block runs a fused convolution
        // (which reads the stage's y buffer); otherwise the block's own LayerNorm launch does it
        SLayer* first = order.empty() ? nullptr : order[0];
        bool y_ready = false;
        if (ln_fusion && first && first->kind == L_BLOCK && (first->log2m > 0 || first->seg) && init_conv_ln_supported(Cin)) {
            DWS_TRY(launch_init_conv_ln(audio, Wi.f(), P("init_conv.0.conv.bias"), P(first->prefix + ".norm1.m"),
                                        P(first->prefix + ".norm1.s"), pt_base() + first->pt_off, pt_bstride(), step_idx, pt_total,
                                        x_init.f(), stages[first->stage]->y.f(), (int)B, Cin, D, (int)L, s));
            y_ready = true;
        } else {
            DWS_TRY(launch_init_conv(audio, Wi.f(), P("init_conv.0.conv.bias"), x_init.f(), (int)B, Cin, D, (int)L, s));
        }
        // the network's final norm (`sashimi.py:309`) comes out of the LAST block's tail the same way (a zero "embedding")
        SLayer* last = order.empty() ? nullptr : order.back();
        final_ln_fused = ln_fusion && last && last->kind == L_BLOCK && last->mfma && last->H == D && zero_row.p;
        size_t oi = 0;
        auto run = [&](SLayer* l, const float* add) -> int {
            SLayer* nxt = oi + 1 < order.size() ? order[oi + 1] : nullptr;
            SLayer* nx = (nxt && (feeds_next(l, nxt) || pool_feeds(l, nxt))) ? nxt : nullptr;
            OutLN o;
            const OutLN* po = nullptr;
            if (nx) { o = next_ln(stages[nx->stage], nx); po = &o; }
            else if (final_ln_fused && l == last) {
                o.m = P("norm.m"); o.s = P("norm.s"); o.e = zero_row.f(); o.y = nfin.f();
                po = &o;
            }
            const int st_ = run_layer(l, x, add, s, y_ready, po);
            y_ready = nx != nullptr;
            ++oi;
            return st_;
        };
        for (auto* l : d_layers) {
            stack.push_back(x);
            DWS_TRY(run(l, nullptr));
            x = l->out.f();
        }
        stack.push_back(x);
        for (size_t i = 0; i < c_layers.size(); ++i) {
            const float* add = nullptr;
            if (i + 1 == c_layers.size()) { add = stack.back(); stack.pop_back(); }
            DWS_TRY(run(c_layers[i], add));
            x = c_layers[i]->out.f();
        }
        for (auto* l : u_layers) {
            const float* add = nullptr;
            if (l->kind == L_UP || unet) { add = stack.back(); stack.pop_back(); }
            DWS_TRY(run(l, add));
            x = l->out.f();
        }
        last_x = x;
        DWS_TRY(final_stage(x, out, nullptr, s, final_ln_fused));
        DWS_HIP(hipGetLastError());
        return DWS_OK;
    }


    // ------------------------------------------------------------------ training path
    // Execution plan of Sashimi.forward (sashimi.py:293-307) as nodes: node 0 = init-conv output,
    // node i+1 = output of plan[i].  The skip stack becomes add_node references.
    void build_plan() {
        if (!plan.empty()) return;
        std::vector<int> stack;
        int x = 0;
        auto push = [&](SLayer* l, int add) {
            plan.push_back({l, x, add});
            x = (int)plan.size();
        };
        for (auto* l : d_layers) { stack.push_back(x); push(l, -1); }
        stack.push_back(x);
        for (size_t i = 0; i < c_layers.size(); ++i) {
            int add = -1;
            if (i + 1 == c_layers.size()) { add = stack.back(); stack.pop_back(); }
            push(c_layers[i], add);
        }
        for (auto* l : u_layers) {
            int add = -1;
            if (l->kind == L_UP || unet) { add = stack.back(); stack.pop_back(); }
            push(l, add);
        }
    }
    float* node_act(int i) { return i == 0 ? x_init.f() : plan[i - 1].l->out.f(); }
    float* node_grad(int i) { return i == 0 ? dx_init.f() : plan[i - 1].l->dbuf.f(); }
    size_t node_numel(int i) {
        if (i == 0) return (size_t)B * D * L;
        SLayer* l = plan[i - 1].l;
        return (l->kind == L_BLOCK) ? (size_t)B * l->H * l->L : (size_t)B * l->Hout * l->Lout;
    }

    // Row-major weights behind the packed operands handed to gemm(): shapes the MFMA position-tile GEMM does not tile
    // (rows or contraction not a multiple of 32: test-sized models) run the plain-FMA GEMM on these instead.
    struct RowMajor { const float* w; int M, K; };
    std::map<const float*, RowMajor> row_major;
    DevBuf rAfT;
    std::vector<DevBuf*> row_major_bufs;   // owned transposes

    // A-fragment pack of W [O][K] and of its transpose (the adjoint GEMM).  `packed`: the commit's own pack of W when it
    // made one (same layout) -- reused instead of packing again; *pA receives the operand to hand to gemm().
    int pack_pair(const float* W, int O, int K, DevBuf& A, DevBuf& AT, const float* packed, const float** pA, hipStream_t s) {
        DWS_TRY(A.ensure((size_t)O * K * 4));
        DWS_TRY(AT.ensure((size_t)O * K * 4));
        if (tapconv_mfma_supported(O, K, 0, 1) && tapconv_mfma_supported(K, O, 0, 1)) {
            if (packed) {
                *pA = packed;
            } else {
                prep_train.pack(W, A.f(), O, K);
                *pA = A.f();
            }
            prep_train.pack_t(W, AT.f(), O, K);     // (pack_train runs the batch: one launch for every weight)
            return DWS_OK;
        }
        // generic: A / AT only serve as keys; AT's storage holds the row-major transpose itself
        DWS_TRY(launch_tapconv_pack_transposed(W, AT.f(), O, K, 1, O, 0, 1.f, s));
        row_major[A.f()] = RowMajor{W, O, K};
        row_major[AT.f()] = RowMajor{AT.f(), K, O};
        *pA = A.f();
        return DWS_OK;
    }

    int pack_train(hipStream_t s) {
        row_major.clear();
        prep_train.begin();
        for (auto* l : all) {
            if (l->kind == L_BLOCK) {
                const int H = l->H;
                const bool cp = l->mfma;   // the commit packed Ao / A1 / A2 for the sampling tail kernel
                DWS_TRY(pack_pair(P(l->prefix + ".layer.output_linear.0.weight"), 2 * H, H, l->tAo, l->tAoT, cp ? l->Ao.f() : nullptr,
                                  &l->pAo, s));
                DWS_TRY(pack_pair(l->W1.f(), FF * H, H, l->tA1, l->tA1T, cp ? l->A1.f() : nullptr, &l->pA1, s));
                DWS_TRY(pack_pair(l->W2.f(), H, FF * H, l->tA2, l->tA2T, cp ? l->A2.f() : nullptr, &l->pA2, s));
            } else {
                const int O = (l->kind == L_DOWN) ? l->Hout : l->Hout * l->p;
                const int K = (l->kind == L_DOWN) ? l->H * l->p : l->H;
                DWS_TRY(pack_pair(l->Wp.f(), O, K, l->tAp, l->tApT, (l->mfma || l->mfma2) ? l->Ap.f() : nullptr, &l->pAp, s));
            }
        }
        DWS_TRY(tmp_pack.ensure((size_t)D * D * 4));
        DWS_TRY(tAfT.ensure((size_t)D * D * 4));
        if (tapconv_mfma_supported(D, D, 0, 1)) {
            prep_train.pack_t(Wf.f(), tAfT.f(), D, D);
        } else {
            DWS_TRY(launch_tapconv_pack_transposed(Wf.f(), tAfT.f(), D, D, 1, D, 0, 1.f, s));
            row_major[tAfT.f()] = RowMajor{tAfT.f(), D, D};
        }
        DWS_TRY(prep_train.run(s));
        train_pack_version = commit_version;
        return DWS_OK;
    }

    // out[b, m, l] = epi(sum_k A[m, k] src[b, k, l])   (tapconv_mfma, T = 1)
    // LayerNorm fused into a training GEMM's epilogue (TapConvArgs::ln_*): destination, the norm's scalars, optional fc_t rows
    struct LnFuse { float* out; const float* m; const float* s; const float* pt; int pt_bstride; };
    bool train_ln_fusion = true;
    bool ln_fusable(const float* A, int epi, int M, int Lx) const {
        static const bool off = getenv("DWS_TRAIN_NO_LN_FUSION") != nullptr;     // same-box A/B switch
        return !off && train_ln_fusion && row_major.find(A) == row_major.end() && tapconv_ln_supported(epi, M, Lx);
    }
    int gemm(const float* A, int M, int K, const float* src, float* out, int Lx, int epi, const float* bias,
             const float* res, const float* addend, const float* aux, float* out2, hipStream_t s, const LnFuse* ln = nullptr) {
        auto it = row_major.find(A);
        if (it != row_major.end()) {
            DWS_CHECK(!ln, DWS_ERR_STATE, "gemm: LayerNorm epilogue requested on the plain-FMA path");
            DWS_CHECK(it->second.M == M && it->second.K == K, DWS_ERR_STATE, "gemm: operand registered as %d x %d, used as %d x %d",
                      it->second.M, it->second.K, M, K);
            GemmRowsArgs g{};
            g.W = it->second.w; g.src = src; g.out = out; g.bias = bias; g.res = res; g.addend = addend; g.aux = aux;
            g.addin = (epi == 0) ? aux : nullptr; g.out2 = out2; g.B = (int)B; g.M = M; g.K = K; g.L = Lx; g.epi = epi;
            return launch_gemm_rows_generic(g, s);
        }
        TapConvArgs q{};
        q.src0 = src; q.K0 = K; q.A = A; q.nkg_total = K / 8; q.M = M; q.T = 1; q.dil = 1; q.sign = 1; q.epi = epi;
        q.out = out; q.bias = bias; q.res = res; q.addend = addend; q.aux = aux; q.out2 = out2;
        q.addin = (epi == 0) ? aux : nullptr; q.addscale = 1.f;
        q.B = (int)B; q.L = Lx;
        q.split = bf16x6 ? 1 : 0;      // precision = bf16x6: the pointwise GEMMs of the training step on the bf16 matrix cores
        if (ln) { q.ln_out = ln->out; q.ln_m = ln->m; q.ln_s = ln->s; q.ln_pt = ln->pt; q.ln_pt_bstride = ln->pt_bstride; }
        return launch_tapconv_mfma(q, s);
    }

    // dW[o, c] = sum_{b, l} dY[b, o, l] * act(X[b, c, l]);  db[o] = sum_{b, l} dY[b, o, l] (optional, same pass)
    // wn != nullptr: the weight is the weight-normed tensor `*wn` -- its adjoint runs inside the split-K reduce (dv / dg written
    // directly; dW is not used)
    int wgrad(const float* dY, const float* X, int O, int Cc, int Lx, int xact, float* dW, float* db, hipStream_t s,
              const std::string* wn = nullptr) {
        static const bool wn_separate = getenv("DWS_WN_BWD_SEPARATE") != nullptr;     // same-box A/B switch
        if (wn && (wn_separate || Cc > WGRAD_WN_MAX_INNER)) {
            DWS_TRY(wgrad(dY, X, O, Cc, Lx, xact, dW, db, s));
            return wn_bwd(*wn, dW, O, Cc, s);
        }
        WgradArgs w{};
        if (wn) {
            w.wn_v = P(*wn + ".weight_v"); w.wn_g = P(*wn + ".weight_g");
            w.wn_dv = G(*wn + ".weight_v"); w.wn_dg = G(*wn + ".weight_g");
        }
        w.dY = dY; w.X = X; w.B = (int)B; w.O = O; w.C = Cc; w.L = Lx; w.dil = 1; w.xact = xact;
        w.nsplit = wgrad_mfma_nsplit((int)B, O, Cc, Lx, 1);
        DWS_TRY(wpart.ensure((size_t)w.nsplit * O * Cc * 4));
        w.partial = wpart.f();
        if (db) {
            DWS_TRY(bpart.ensure((size_t)w.nsplit * O * 4));
            w.bias_part = bpart.f(); w.dbias = db; w.bias_scale = 1.f;
        }
        w.split = bf16x6 ? 1 : 0;
        return launch_wgrad_mfma(w, 1, 1.f, dW, s);
    }

    int wn_bwd(const std::string& p, const float* dWf, int O, int inner, hipStream_t s) {
        return launch_weight_norm_bwd(dWf, P(p + ".weight_v"), P(p + ".weight_g"), G(p + ".weight_v"), G(p + ".weight_g"), O,
                                      inner, s);
    }

    int train_supported() {
        DWS_CHECK(melBm == 0 || melBm == B, DWS_ERR_UNSUPPORTED,
                  "mel-conditional training needs one mel per clip (got %lld for B=%lld)", (long long)melBm, (long long)B);
        DWS_CHECK(!cond || melBm > 0, DWS_ERR_STATE, "conditional model: install the mel (set_condition) before forward_train");
        for (auto* l : all) {
            if (l->kind == L_BLOCK) {
                DWS_CHECK(l->log2m > 0, DWS_ERR_UNSUPPORTED,
                          "sashimi training needs the fused FFT convolution (L even, <= 16384 per stage); stage L=%d", l->L);
                DWS_CHECK(l->Lk == l->L, DWS_ERR_UNSUPPORTED,
                          "sashimi training runs at the kernels' own length (stage runs at %d, kernel length %d)", l->L, l->Lk);
            }
        }
        return DWS_OK;   // channel counts that are not multiples of 32 train on the plain-FMA GEMM (row_major)
    }

    int ensure_train_buffers() {
        build_plan();
        for (auto* st : stages) {
            const size_t rows = (size_t)B * st->H * st->L * 4;
            DWS_TRY(st->d2.ensure(rows * std::max(2, FF)));
            DWS_TRY(st->dh.ensure(rows));
            DWS_TRY(st->dx1.ensure(rows));
            DWS_TRY(st->du.ensure(rows));
            DWS_TRY(st->y.ensure(rows));
        }
        size_t pool_max = 4;
        for (auto* l : all) {
            if (l->kind == L_BLOCK) {
                const size_t n = (size_t)B * l->H * l->L * 4;
                DWS_TRY(l->t_u.ensure(n)); DWS_TRY(l->t_a.ensure(n)); DWS_TRY(l->t_o.ensure(2 * n));
                DWS_TRY(l->t_g.ensure(n)); DWS_TRY(l->t_ge.ensure((size_t)FF * n));
                DWS_TRY(l->t_x1.ensure(n)); DWS_TRY(l->t_n2.ensure(n)); DWS_TRY(l->t_f1.ensure((size_t)FF * n));
                DWS_TRY(l->dbuf.ensure(n));
            } else {
                const size_t nin = (size_t)B * l->H * l->L * 4, nout = (size_t)B * l->Hout * l->Lout * 4;
                DWS_TRY(l->dbuf.ensure(nout));
                if (l->kind == L_DOWN) DWS_TRY(l->t_xr.ensure(nin));
                pool_max = std::max(pool_max, std::max(nin, nout));
            }
        }
        DWS_TRY(pool_scr.ensure(pool_max));
        DWS_TRY(dx_init.ensure((size_t)B * D * L * 4));
        DWS_TRY(ty.ensure((size_t)B * D * L * 4));
        DWS_TRY(dyb.ensure((size_t)B * D * L * 4));
        DWS_TRY(dnf.ensure((size_t)B * D * L * 4));
        DWS_TRY(ta1.ensure((size_t)B * Emid * 4));
        DWS_TRY(ta2.ensure((size_t)B * Eout * 4));
        ln_slot_floats = (size_t)B * ceil_div(L, 64) * 2;       // a slot per LayerNorm adjoint: two per block + the final norm
        ln_slots = 1;
        for (auto* l : all) ln_slots += (l->kind == L_BLOCK) ? 2 : 0;
        DWS_TRY(lnpart.ensure(ln_slot_floats * ln_slots * 4));
        return DWS_OK;
    }

    int forward_train(const float* audio, const float* steps, float* out, hipStream_t s) override {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "forward before prepare");
        DWS_CHECK(!f16x3, DWS_ERR_UNSUPPORTED,
                  "training runs with precision=f32 or bf16x6 (gradients span too many binades for the fixed scales of the fp16 split)");
        keep_cauchy = true;
        if (dirty) DWS_TRY(commit(s));
        DWS_TRY(train_supported());
        DWS_TRY(ensure_train_buffers());
        if (train_pack_version != commit_version) DWS_TRY(pack_train(s));
        const int nB = (int)B;
        DWS_TRY(launch_init_conv(audio, Wi.f(), P("init_conv.0.conv.bias"), x_init.f(), nB, Cin, D, (int)L, s));
        DWS_TRY(launch_step_embed(steps, freq.f(), emb.f(), nB, Ein / 2, s));
        DWS_TRY(launch_linear_rows(emb.f(), P("fc_t1.weight"), P("fc_t1.bias"), h1.f(), nB, Ein, Emid, 1, s, ta1.f()));
        DWS_TRY(launch_linear_rows(h1.f(), P("fc_t2.weight"), P("fc_t2.bias"), h2.f(), nB, Emid, Eout, 1, s, ta2.f()));
        DWS_TRY(launch_linear_rows(h2.f(), Wt_all.f(), bt_all.f(), part_t.f(), nB, Eout, pt_total, 0, s));
        bool ln1_done = false;      // this block's LN1 came out of the previous block's epilogue
        for (size_t i = 0; i < plan.size(); ++i) {
            const Exec& e = plan[i];
            SLayer* l = e.l;
            const float* x = node_act(e.in_node);
            const float* add = e.add_node >= 0 ? node_act(e.add_node) : nullptr;
            if (l->kind == L_BLOCK) {
                Stage* st = stages[l->stage];
                const int H = l->H, Ls = l->L;
                const std::string& p = l->prefix;
                // LN1(x) + fc_t(e): already written by the previous block's last GEMM when that one could fuse it (below)
                if (!ln1_done)
                    DWS_TRY(launch_ln(x, P(p + ".norm1.m"), P(p + ".norm1.s"), part_t.f() + l->pt_off, pt_total, l->t_u.f(), nB, H,
                                      Ls, (size_t)Ls, s));
                ln1_done = false;
                FftTables* t = tables[l->log2m];
                FftConvArgs fa{};
                fa.u = l->t_u.f(); fa.g = l->t_g.f(); fa.pre = l->t_a.f(); fa.D = P(p + ".layer.D");
                fa.tw = (const c2*)t->tw.p; fa.twp = (const c2*)t->twp.p;
                fa.kfa = (const c2*)l->kfa_v; fa.kfb = (const c2*)l->kfb_v; fa.kfs = (const c2*)l->kfs_v;
                fa.B = nB; fa.H = H; fa.L = Ls;
                DWS_TRY(launch_fftconv(l->log2m, fa, s));
                // LN2(x1) out of the epilogue that produces x1 when the tile holds every channel (H = 128): the separate
                // LayerNorm pass (read x1, write n2) becomes one extra store
                bool ln2_done = false;
                if (tapconv_glu_supported(2 * H, H, Ls)) {   // o and x1 = x + GLU(o) (+ mel) from one kernel
                    LnFuse f2{l->t_n2.f(), P(p + ".norm2.m"), P(p + ".norm2.s"), nullptr, 0};
                    ln2_done = ln_fusable(l->pAo, 6, 2 * H, Ls);
                    DWS_TRY(gemm(l->pAo, 2 * H, H, l->t_g.f(), l->t_o.f(), Ls, 6, P(p + ".layer.output_linear.0.bias"), x,
                                 nullptr, melBm ? l->melc.f() : nullptr, l->t_x1.f(), s, ln2_done ? &f2 : nullptr));
                } else {
                    DWS_TRY(gemm(l->pAo, 2 * H, H, l->t_g.f(), l->t_o.f(), Ls, 2, P(p + ".layer.output_linear.0.bias"),
                                 nullptr, nullptr, nullptr, nullptr, s));
                    DWS_TRY(launch_glu_res(l->t_o.f(), x, melBm ? l->melc.f() : nullptr, l->t_x1.f(), nB, H, Ls, s));
                }
                if (!ln2_done)
                    DWS_TRY(launch_ln(l->t_x1.f(), P(p + ".norm2.m"), P(p + ".norm2.s"), nullptr, 0, l->t_n2.f(), nB, H, Ls,
                                      (size_t)Ls, s));
                DWS_TRY(gemm(l->pA1, FF * H, H, l->t_n2.f(), l->t_f1.f(), Ls, 3, P(p + ".ff.ff.0.conv.bias"), nullptr,
                             nullptr, nullptr, l->t_ge.f(), s));
                // the NEXT block's LN1(out) + fc_t(e) out of this block's last epilogue (same stage, H = 128 / 256)
                SLayer* nx = (i + 1 < plan.size() && plan[i + 1].l->kind == L_BLOCK) ? plan[i + 1].l : nullptr;
                LnFuse f1{};
                if (nx && nx->H == H && nx->L == Ls && ln_fusable(l->pA2, 4, H, Ls)) {
                    f1 = LnFuse{nx->t_u.f(), P(nx->prefix + ".norm1.m"), P(nx->prefix + ".norm1.s"), part_t.f() + nx->pt_off, (int)pt_total};
                    ln1_done = true;
                }
                DWS_TRY(gemm(l->pA2, H, FF * H, l->t_ge.f(), l->out.f(), Ls, 4, P(p + ".ff.ff.2.conv.bias"), l->t_x1.f(),
                             add, nullptr, nullptr, s, ln1_done ? &f1 : nullptr));
            } else if (l->kind == L_DOWN) {
                DWS_TRY(launch_pool_rearrange(x, l->t_xr.f(), nullptr, 0, 0, nB, l->H, l->p, l->Lout, s));
                DWS_TRY(gemm(l->pAp, l->Hout, l->H * l->p, l->t_xr.f(), l->out.f(), l->Lout, 2,
                             P(l->prefix + ".linear.conv.bias"), nullptr, nullptr, nullptr, nullptr, s));
            } else {
                DWS_TRY(gemm(l->pAp, l->Hout * l->p, l->H, x, pool_scr.f(), l->L, 2, P(l->prefix + ".linear.conv.bias"),
                             nullptr, nullptr, nullptr, nullptr, s));
                DWS_TRY(launch_pool_rearrange(pool_scr.f(), l->out.f(), add, 1, 0, nB, l->Hout, l->p, l->L, s));
            }
        }
        last_x = node_act((int)plan.size());
        DWS_TRY(final_stage(last_x, out, ty.f(), s));
        DWS_HIP(hipGetLastError());
        train_audio = audio;
        trained_fwd = true;
        return DWS_OK;
    }

    // gradient of the S4 kernel parameters of one block from u and da (s4.py:704-807 backwards)
    // The adjoint of a group's stacked kernel generation, once the spectrum gradients of all its blocks are in g->dKf
    // (kernel_backward's chain with Ht = n H rows); the stacked parameter gradients leave for the blocks' tensors in one launch.
    int group_backward(KGroup* g, hipStream_t s) {
        const int H = g->H, Ht = g->Ht(), Ls = g->L, Lh = Ls / 2 + 1, N = NS;
        const int M = 1 << g->log2m, Nf = 2 * M;
        const size_t HN = (size_t)Ht * N, hn = (size_t)H * N;
        DWS_TRY(dKt.ensure((size_t)Ht * Nf * 4));
        DWS_TRY(dkt.ensure((size_t)2 * Ht * Ls * 4));
        DWS_TRY(dkf.ensure((size_t)2 * Ht * Lh * 8));
        DWS_TRY(cgr.ensure((size_t)6 * Ht * Lh * 8));
        DWS_TRY(cgv.ensure(6 * HN * 8));
        DWS_TRY(cgw.ensure(6 * HN * 8));
        const int nparts = ceil_div(Lh, 256);
        DWS_TRY(cpdt.ensure((size_t)Ht * nparts * 4));
        DWS_TRY(g->gC.ensure(2 * HN * 8)); DWS_TRY(g->gB.ensure(HN * 8)); DWS_TRY(g->gP.ensure(HN * 8));
        DWS_TRY(g->giwr.ensure(HN * 4)); DWS_TRY(g->gwim.ensure(HN * 4)); DWS_TRY(g->glogdt.ensure((size_t)Ht * 4));
        DWS_TRY(g->gD.ensure((size_t)Ht * 4));
        DWS_TRY(fft.exec(1, Nf, Ht, g->dKf.p, dKt.p, s));
        DWS_TRY(launch_s4_twosided_pow2_bwd(dKt.f(), dkt.f(), g->gD.f(), Ht, Ls, Nf, 1.f / ((float)Nf * (float)Ls), 1.f / (float)Nf, s));
        DWS_TRY(fft.exec(0, Ls, 2 * Ht, dkt.p, dkf.p, s));
        DWS_TRY(launch_s4_woodbury_bwd(g->r.f(), P("__omega." + std::to_string(Ls)), g->dt.f(), dkf.f(), cgr.f(), cpdt.f(), Ht, Lh,
                                       (Ls % 2) == 0, s));
        DWS_TRY(launch_cauchy_sym_bwd_bcast(g->v.f(), P("__z." + std::to_string(Ls)), g->wdt.f(), cgr.f(), cgv.f(), cgw.f(), 6 * Ht, N, Lh,
                                            Ht, s));
        DWS_TRY(launch_s4_prep_bwd(g->C.f(), g->Bp.f(), g->P.f(), g->iwr.f(), g->wim.f(), g->logdt.f(), cgv.f(), cgw.f(), cpdt.f(), nparts,
                                   g->gC.f(), g->gB.f(), g->gP.f(), g->giwr.f(), g->gwim.f(), g->glogdt.f(), Ht, N, s));
        g->unstack.begin();
        for (size_t i = 0; i < g->layers.size(); ++i) {
            const std::string p = g->layers[i]->prefix, k = p + ".layer.kernel.kernel";
            for (int c = 0; c < 2; ++c) g->unstack.add(g->gC.f() + (c * HN + i * hn) * 2, G(k + ".C") + c * hn * 2, hn * 2);
            g->unstack.add(g->gB.f() + i * hn * 2, G(k + ".B"), hn * 2);
            g->unstack.add(g->gP.f() + i * hn * 2, G(k + ".P"), hn * 2);
            g->unstack.add(g->giwr.f() + i * hn, G(k + ".inv_w_real"), hn);
            g->unstack.add(g->gwim.f() + i * hn, G(k + ".w_imag"), hn);
            g->unstack.add(g->glogdt.f() + i * H, G(k + ".log_dt"), (size_t)H);
            g->unstack.add(g->gD.f() + i * H, G(p + ".layer.D"), (size_t)H);
        }
        return g->unstack.run(s);
    }

    int kernel_backward(SLayer* l, const float* da, hipStream_t s) {
        const int H = l->H, Ls = l->L, Lh = Ls / 2 + 1, N = NS, nB = (int)B;
        const int M = 1 << l->log2m, Nf = 2 * M;
        if (kernels_stacked && l->grp >= 0) {
            // this block's spectrum gradient into its rows of the group's stack; the rest of the chain runs once per group
            KGroup* g = kgroups[l->grp];
            FftTables* t = tables[l->log2m];
            const int nbs = std::max(1, std::min(nB, ceil_div(512, H)));
            const int bchunk = ceil_div(nB, nbs);
            const int nchunks = ceil_div(nB, bchunk);
            DWS_TRY(fpart.ensure((size_t)nchunks * H * (M + 1) * 8));
            DWS_TRY(g->dKf.ensure((size_t)g->Ht() * (M + 1) * 8));
            FftCorrArgs c{};
            c.u = l->t_u.f(); c.da = da; c.part = (c2*)fpart.p; c.tw = (const c2*)t->tw.p; c.twp = (const c2*)t->twp.p;
            c.B = nB; c.H = H; c.L = Ls; c.bchunk = bchunk;
            DWS_TRY(launch_fftcorr(l->log2m, c, s));
            DWS_TRY(launch_sum_leading(fpart.f(), g->dKf.f() + (size_t)l->gidx * H * (M + 1) * 2, (size_t)H * (M + 1) * 2, nchunks, 1.f, s));
            if (--g->pending == 0) DWS_TRY(group_backward(g, s));
            return DWS_OK;
        }
        const std::string k = l->prefix + ".layer.kernel.kernel";
        FftTables* t = tables[l->log2m];
        const int nbs = std::max(1, std::min(nB, ceil_div(512, H)));
        const int bchunk = ceil_div(nB, nbs);
        const int nchunks = ceil_div(nB, bchunk);
        DWS_TRY(fpart.ensure((size_t)nchunks * H * (M + 1) * 8));
        DWS_TRY(dKf.ensure((size_t)H * (M + 1) * 8));
        DWS_TRY(dKt.ensure((size_t)H * Nf * 4));
        DWS_TRY(dkt.ensure((size_t)2 * H * Ls * 4));
        DWS_TRY(dkf.ensure((size_t)2 * H * Lh * 8));
        FftCorrArgs c{};
        c.u = l->t_u.f(); c.da = da; c.part = (c2*)fpart.p; c.tw = (const c2*)t->tw.p; c.twp = (const c2*)t->twp.p;
        c.B = nB; c.H = H; c.L = Ls; c.bchunk = bchunk;
        DWS_TRY(launch_fftcorr(l->log2m, c, s));
        DWS_TRY(launch_sum_leading(fpart.f(), dKf.f(), (size_t)H * (M + 1) * 2, nchunks, 1.f, s));
        DWS_TRY(fft.exec(1, Nf, H, dKf.p, dKt.p, s));
        // dK_t = C2R / Nf; k enters K as k / L (s4_twosided_pow2); dD[h] = sum u da = dK_t[h][0]
        DWS_TRY(launch_s4_twosided_pow2_bwd(dKt.f(), dkt.f(), G(l->prefix + ".layer.D"), H, Ls, Nf,
                                            1.f / ((float)Nf * (float)Ls), 1.f / (float)Nf, s));
        DWS_TRY(fft.exec(0, Ls, 2 * H, dkt.p, dkf.p, s));
        // v, w dt, dt, r of this block: kept by build_kernel when this commit already ran in training mode, else regenerated
        const bool cached = l->cache_version == commit_version && l->t_cr.p;
        DevBuf& bv = cached ? l->t_cv : cv;
        DevBuf& bw = cached ? l->t_cwdt : cwdt;
        DevBuf& bd = cached ? l->t_cdt : cdt;
        DevBuf& br = cached ? l->t_cr : cr;
        DWS_TRY(cgr.ensure((size_t)6 * H * Lh * 8));
        DWS_TRY(cgv.ensure((size_t)6 * H * N * 8));
        DWS_TRY(cgw.ensure((size_t)6 * H * N * 8));
        const int nparts = ceil_div(Lh, 256);
        DWS_TRY(cpdt.ensure((size_t)H * nparts * 4));
        const float* z = P("__z." + std::to_string(Ls));
        if (!cached) {
            DWS_TRY(bv.ensure((size_t)6 * H * N * 8));
            DWS_TRY(bw.ensure((size_t)H * N * 8));
            DWS_TRY(bd.ensure((size_t)H * 4));
            DWS_TRY(br.ensure((size_t)6 * H * Lh * 8));
            DWS_TRY(launch_s4_prep(P(k + ".C"), P(k + ".B"), P(k + ".P"), P(k + ".inv_w_real"), P(k + ".w_imag"),
                                   P(k + ".log_dt"), bv.f(), bw.f(), bd.f(), H, N, s));
            DWS_TRY(launch_cauchy_sym_fwd_bcast(bv.f(), z, bw.f(), br.f(), 6 * H, N, Lh, H, s));
        }
        DWS_TRY(launch_s4_woodbury_bwd(br.f(), P("__omega." + std::to_string(Ls)), bd.f(), dkf.f(), cgr.f(), cpdt.f(), H,
                                       Lh, (Ls % 2) == 0, s));
        DWS_TRY(launch_cauchy_sym_bwd_bcast(bv.f(), z, bw.f(), cgr.f(), cgv.f(), cgw.f(), 6 * H, N, Lh, H, s));
        DWS_TRY(launch_s4_prep_bwd(P(k + ".C"), P(k + ".B"), P(k + ".P"), P(k + ".inv_w_real"), P(k + ".w_imag"),
                                   P(k + ".log_dt"), cgv.f(), cgw.f(), cpdt.f(), nparts, G(k + ".C"), G(k + ".B"), G(k + ".P"),
                                   G(k + ".inv_w_real"), G(k + ".w_imag"), G(k + ".log_dt"), H, N, s));
        return DWS_OK;
    }

    // (dm, ds) of a LayerNorm from its adjoint's per-block partials [2][nblk].  Every adjoint of a backward writes its own slot of
    // lnpart and the sums of all of them run as ONE launch at the end of the backward (ln_scalars_flush) -- they were 61 launches
    // of 8 us per config-5 step.  The scalars' gradients are therefore final at the end of backward only (G() is called there:
    // the staged hand-over learns it); they are 122 floats.
    struct LnPending { float* part; std::string name; int nblk; };
    std::vector<LnPending> ln_pending;
    std::vector<SumPairJob> ln_jobs, ln_jobs_uploaded;
    DevBuf ln_table;
    size_t ln_slot_floats = 0;
    int ln_slots = 0;
    bool ln_deferred() const {
        static const bool off = getenv("DWS_LN_SCALARS_SEPARATE") != nullptr;     // same-box A/B switch
        return !off;
    }
    float* ln_slot() {      // partial buffer of the NEXT LayerNorm adjoint of this backward
        if (!ln_deferred() || (int)ln_pending.size() >= ln_slots) return lnpart.f();
        return lnpart.f() + ln_pending.size() * ln_slot_floats;
    }
    int ln_scalars(const std::string& p, int nblk, hipStream_t s) {
        if (!ln_deferred() || (int)ln_pending.size() >= ln_slots)
            return launch_sum_pair(ln_slot(), G(p + ".m"), G(p + ".s"), nblk, s);
        ln_pending.push_back({ln_slot(), p, nblk});
        return DWS_OK;
    }
    int ln_scalars_flush(hipStream_t s) {
        if (ln_pending.empty()) return DWS_OK;
        ln_jobs.clear();
        for (auto& e : ln_pending) ln_jobs.push_back({e.part, G(e.name + ".m"), G(e.name + ".s"), e.nblk, 0});
        ln_pending.clear();
        const bool same = ln_jobs.size() == ln_jobs_uploaded.size() &&
                          std::memcmp(ln_jobs.data(), ln_jobs_uploaded.data(), ln_jobs.size() * sizeof(SumPairJob)) == 0;
        if (!same) {
            DWS_HIP(hipStreamSynchronize(s));
            ln_jobs_uploaded = ln_jobs;
            DWS_TRY(ln_table.ensure(ln_jobs.size() * sizeof(SumPairJob)));
            DWS_HIP(hipMemcpy(ln_table.p, ln_jobs.data(), ln_jobs.size() * sizeof(SumPairJob), hipMemcpyHostToDevice));
        }
        return launch_sum_pair_multi((const SumPairJob*)ln_table.p, (int)ln_jobs.size(), s);
    }

    // Adjoint of forward_train, plan steps in reverse (sashimi.py:143-184,277-313).
    int backward(const float* dout, hipStream_t s) override {
        DWS_CHECK(trained_fwd, DWS_ERR_STATE, "backward without a preceding forward_train");
        const int nB = (int)B, nL = (int)L;
        const int nnodes = (int)plan.size() + 1;
        ln_pending.clear();
        for (auto* g : kgroups) g->pending = (int)g->layers.size();
        std::vector<char> written(nnodes, 0);
        std::vector<float*> gp(nnodes);       // where each node's gradient lives during THIS backward (a skip node may adopt a dy buffer)
        for (int n = 0; n < nnodes; ++n) gp[n] = node_grad(n);
        size_t wmax = (size_t)D * D;
        for (auto* l : all) {
            if (l->kind == L_BLOCK) wmax = std::max(wmax, (size_t)2 * l->H * l->H * std::max(1, FF));
            else wmax = std::max(wmax, (size_t)l->H * l->p * l->Hout * std::max(l->p, 1));
        }
        DWS_TRY(dWfold.ensure(wmax * 4));
        DWS_TRY(dpt.ensure((size_t)B * pt_total * 4));
        DWS_TRY(dh2.ensure((size_t)B * Eout * 4)); DWS_TRY(dh1.ensure((size_t)B * Emid * 4));
        DWS_TRY(dWt_all.ensure((size_t)pt_total * Eout * 4)); DWS_TRY(dbt_all.ensure((size_t)pt_total * 4));

        // ---- final stage: out = Wz y + bz, y = relu(Wf LN(x) + bf)
        // (1 x D and D x 1 weight gradients also go through the MFMA kernel: its tile is mostly padding there, but the
        // generic FMA kernel took 2.3 ms per call on [B, ., L] = 512000 positions)
        DWS_TRY(wgrad(dout, ty.f(), Cout, D, nL, 0, G("final_conv.2.conv.weight"), G("final_conv.2.conv.bias"), s));
        DWS_TRY(launch_final_dy(dout, P("final_conv.2.conv.weight"), ty.f(), dyb.f(), nB, D, Cout, nL, s));
        { const std::string wn_ = "final_conv.0.conv"; DWS_TRY(wgrad(dyb.f(), nfin.f(), D, D, nL, 0, dWfold.f(), G("final_conv.0.conv.bias"), s, &wn_)); }
        DWS_TRY(gemm(tAfT.f(), D, D, dyb.f(), dnf.f(), nL, 2, nullptr, nullptr, nullptr, nullptr, nullptr, s));
        {
            const int last = nnodes - 1;
            DWS_TRY(launch_ln_bwd(node_act(last), dnf.f(), P("norm.m"), P("norm.s"), nullptr, gp[last], 0, ln_slot(),
                                  nB, D, nL, s));
            DWS_TRY(ln_scalars("norm", nB * ceil_div(nL, 64), s));
            written[last] = 1;
        }

        for (int i = (int)plan.size() - 1; i >= 0; --i) {
            const Exec& e = plan[i];
            SLayer* l = e.l;
            const float* x = node_act(e.in_node);
            const float* dy = gp[i + 1];
            float* din = gp[e.in_node];
            DWS_CHECK(written[i + 1], DWS_ERR_STATE, "backward: no gradient reached node %d", i + 1);
            const std::string& p = l->prefix;
            if (l->kind == L_BLOCK) {
                Stage* st = stages[l->stage];
                const int H = l->H, Ls = l->L, nblk = nB * ceil_div(Ls, 64);
                // ff: out = x1 + W2 gelu(f1) + b2, f1 = W1 n2 + b1
                DWS_TRY(gemm(l->tA2T.f(), FF * H, H, dy, st->d2.f(), Ls, 5, nullptr, nullptr, nullptr, l->t_f1.f(), nullptr, s));
                { const std::string wn_ = p + ".ff.ff.2.conv"; DWS_TRY(wgrad(dy, l->t_ge.f(), H, FF * H, Ls, 0, dWfold.f(), G(p + ".ff.ff.2.conv.bias"), s, &wn_)); }
                DWS_TRY(gemm(l->tA1T.f(), H, FF * H, st->d2.f(), st->dh.f(), Ls, 2, nullptr, nullptr, nullptr, nullptr, nullptr, s));
                { const std::string wn_ = p + ".ff.ff.0.conv"; DWS_TRY(wgrad(st->d2.f(), l->t_n2.f(), FF * H, H, Ls, 0, dWfold.f(), G(p + ".ff.ff.0.conv.bias"), s, &wn_)); }
                // norm2: dx1 = dy + LN'(dn2)
                // (the GLU adjoint rides on this kernel when it can: d o is written from the d x1 values in registers)
                const bool glu_fused = ln_bwd_fuses_glu(H);
                DWS_TRY(launch_ln_bwd(l->t_x1.f(), st->dh.f(), P(p + ".norm2.m"), P(p + ".norm2.s"), dy, st->dx1.f(), 0,
                                      ln_slot(), nB, H, Ls, s, glu_fused ? l->t_o.f() : nullptr,
                                      glu_fused ? st->d2.f() : nullptr));
                DWS_TRY(ln_scalars(p + ".norm2", nblk, s));
                if (melBm) {  // x1 = ... + melc: the block's conditioner sees d x1 (`sashimi.py:160-175`)
                    const int s0 = d.mel_upsample[0], s1 = d.mel_upsample[1];
                    DWS_TRY(gW0f.ensure((size_t)3 * 2 * s0 * 4)); DWS_TRY(gW1f.ensure((size_t)3 * 2 * s1 * 4));
                    DWS_TRY(gWcf.ensure((size_t)H * MB * 4));
                    DWS_TRY(conditioner_backward(cws, mel_in.f(), nB, MB, mel_T, s0, s1, l->melW0.f(), P(p + ".upsample_conv2d.0.bias"),
                                                 l->melW1.f(), P(p + ".upsample_conv2d.1.bias"), l->melWc.f(), H, Ls, st->dx1.f(),
                                                 gW0f.f(), G(p + ".upsample_conv2d.0.bias"), gW1f.f(),
                                                 G(p + ".upsample_conv2d.1.bias"), gWcf.f(), s));
                    DWS_TRY(wn_bwd(p + ".upsample_conv2d.0", gW0f.f(), 1, 3 * 2 * s0, s));
                    DWS_TRY(wn_bwd(p + ".upsample_conv2d.1", gW1f.f(), 1, 3 * 2 * s1, s));
                    DWS_TRY(wn_bwd(p + ".mel_conv.conv", gWcf.f(), H, MB, s));
                    DWS_TRY(launch_rowsum(st->dx1.f(), G(p + ".mel_conv.conv.bias"), nB, H, Ls, 1.f, 0, s));
                }
                // x1 = x + glu(o), o = Wo gelu(a) + bo
                if (!glu_fused) DWS_TRY(launch_glu_bwd(st->dx1.f(), l->t_o.f(), st->d2.f(), nB, H, Ls, s));
                DWS_TRY(gemm(l->tAoT.f(), H, 2 * H, st->d2.f(), st->dh.f(), Ls, 5, nullptr, nullptr, nullptr, l->t_a.f(), nullptr, s));
                DWS_TRY(wgrad(st->d2.f(), l->t_g.f(), 2 * H, H, Ls, 0, G(p + ".layer.output_linear.0.weight"),
                              G(p + ".layer.output_linear.0.bias"), s));
                // a = conv(u, K) + D u: du = conv^T(da) + D da; kernel parameters from corr(u, da)
                FftTables* t = tables[l->log2m];
                FftConvArgs fa{};
                fa.u = st->dh.f(); fa.g = st->du.f(); fa.D = P(p + ".layer.D"); fa.conj_k = 1; fa.no_act = 1;
                fa.tw = (const c2*)t->tw.p; fa.twp = (const c2*)t->twp.p;
                fa.kfa = (const c2*)l->kfa_v; fa.kfb = (const c2*)l->kfb_v; fa.kfs = (const c2*)l->kfs_v;
                fa.B = nB; fa.H = H; fa.L = Ls;
                // d fc_t(e)[b, h] = sum_l du[b, h, l] leaves with the row (a workgroup owns it) where the plan allows
                const bool rs_fused = fftconv_rowsum_supported(l->log2m) && getenv("DWS_NO_ROWSUM_FUSION") == nullptr;
                if (rs_fused) { fa.rowsum = dpt.f() + l->pt_off; fa.rowsum_bstride = (int)pt_total; }
                DWS_TRY(launch_fftconv(l->log2m, fa, s));
                DWS_TRY(kernel_backward(l, st->dh.f(), s));
                // u = LN1(x) + fc_t(e): dx = dx1 + LN'(du)
                if (!rs_fused) DWS_TRY(launch_rowsum_bc(st->du.f(), dpt.f() + l->pt_off, pt_total, nB, H, Ls, s));
                DWS_TRY(launch_ln_bwd(x, st->du.f(), P(p + ".norm1.m"), P(p + ".norm1.s"), st->dx1.f(), din, written[e.in_node],
                                      ln_slot(), nB, H, Ls, s));
                DWS_TRY(ln_scalars(p + ".norm1", nblk, s));
                written[e.in_node] = 1;
            } else if (l->kind == L_DOWN) {
                const int K = l->H * l->p, O = l->Hout;
                { const std::string wn_ = p + ".linear.conv"; DWS_TRY(wgrad(dy, l->t_xr.f(), O, K, l->Lout, 0, dWfold.f(), G(p + ".linear.conv.bias"), s, &wn_)); }
                DWS_TRY(gemm(l->tApT.f(), K, O, dy, pool_scr.f(), l->Lout, 2, nullptr, nullptr, nullptr, nullptr, nullptr, s));
                DWS_TRY(launch_pool_rearrange(pool_scr.f(), din, nullptr, 1, written[e.in_node], nB, l->H, l->p, l->Lout, s));
                written[e.in_node] = 1;
            } else {
                const int K = l->H, O = l->Hout * l->p;   // xl = Wp x + b, [B][O][L_in]
                DWS_TRY(launch_pool_rearrange(dy, pool_scr.f(), nullptr, 0, 0, nB, l->Hout, l->p, l->L, s));
                { const std::string wn_ = p + ".linear.conv"; DWS_TRY(wgrad(pool_scr.f(), x, O, K, l->L, 0, dWfold.f(), G(p + ".linear.conv.bias"), s, &wn_)); }
                if (written[e.in_node])
                    DWS_TRY(gemm(l->tApT.f(), K, O, pool_scr.f(), din, l->L, 0, nullptr, nullptr, nullptr, din, nullptr, s));
                else
                    DWS_TRY(gemm(l->tApT.f(), K, O, pool_scr.f(), din, l->L, 2, nullptr, nullptr, nullptr, nullptr, nullptr, s));
                written[e.in_node] = 1;
            }
            if (e.add_node >= 0) {  // the skip connection receives the same gradient
                // The up path runs first in backward, so the skip node has no gradient yet: instead of COPYING dy into its buffer
                // (15 copies of up to 262 MB per C5 step, 1.5 ms) the node ADOPTS this layer's dy buffer -- everything of this
                // layer that reads dy is already enqueued, and the node's own consumer accumulates into it in place later.
                static const bool copy_skip = getenv("DWS_SKIP_GRAD_COPY") != nullptr;      // same-box A/B switch
                if (!written[e.add_node] && !copy_skip && node_numel(e.add_node) == node_numel(i + 1)) {
                    gp[e.add_node] = const_cast<float*>(dy);
                } else {
                    DWS_TRY(launch_add_into(dy, gp[e.add_node], written[e.add_node], node_numel(e.add_node), s));
                }
                written[e.add_node] = 1;
            }
            DWS_TRY(grad_point(s));   // staged hand-over: buckets whose last gradient this layer produced leave now
        }

        // ---- init_conv: x0 = relu(Wi audio + bi)
        const size_t nact = (size_t)B * D * L;
        DWS_CHECK(written[0], DWS_ERR_STATE, "backward: no gradient reached the init conv");
        DWS_TRY(launch_relu_bwd(gp[0], x_init.f(), nact, s));      // (node 0 is a skip node too: its gradient may live in an adopted buffer)
        { const std::string wn_ = "init_conv.0.conv"; DWS_TRY(wgrad(gp[0], train_audio, D, Cin, nL, 0, dWfold.f(), G("init_conv.0.conv.bias"), s, &wn_)); }

        // ---- step embedding: per-block fc_t (stacked), then the shared swish MLP
        DWS_TRY(launch_lin_bwd_w(dpt.f(), h2.f(), dWt_all.f(), dbt_all.f(), nB, Eout, pt_total, s));
        unstack_fc_t.begin();
        for (auto* l : all) {
            if (l->kind != L_BLOCK) continue;
            unstack_fc_t.add(dWt_all.f() + (size_t)l->pt_off * Eout, G(l->prefix + ".fc_t.weight"), (size_t)l->H * Eout);
            unstack_fc_t.add(dbt_all.f() + l->pt_off, G(l->prefix + ".fc_t.bias"), (size_t)l->H);
        }
        DWS_TRY(unstack_fc_t.run(s));
        DWS_TRY(launch_lin_bwd_x(dpt.f(), Wt_all.f(), ta2.f(), dh2.f(), nB, Eout, pt_total, lin_scratch, s));
        DWS_TRY(launch_lin_bwd_w(dh2.f(), h1.f(), G("fc_t2.weight"), G("fc_t2.bias"), nB, Emid, Eout, s));
        DWS_TRY(launch_lin_bwd_x(dh2.f(), P("fc_t2.weight"), ta1.f(), dh1.f(), nB, Emid, Eout, lin_scratch, s));
        DWS_TRY(launch_lin_bwd_w(dh1.f(), emb.f(), G("fc_t1.weight"), G("fc_t1.bias"), nB, Ein, Emid, s));
        DWS_TRY(ln_scalars_flush(s));
        DWS_HIP(hipGetLastError());
        return DWS_OK;
    }

    int read_tap(const char* tap, float* dst, int64_t capacity, hipStream_t s) override {
        DWS_CHECK(B > 0, DWS_ERR_STATE, "read_tap before prepare/forward");
        const std::string t(tap);
        if (t == "split_launches") {   // [block tails of the last forward that ran a split instance, those that ran exact-f32 kernels]
            DWS_CHECK(capacity >= 2, DWS_ERR_INVALID, "tap buffer too small");     // (the pooling GEMMs are exact f32 under every precision)
            const float v[2] = {(float)n_tail_split, (float)n_tail_f32};
            DWS_HIP(hipMemcpyAsync(dst, v, 8, hipMemcpyHostToDevice, s));
            DWS_HIP(hipStreamSynchronize(s));
            return DWS_OK;
        }
        if (t == "pre_final") {
            DWS_CHECK(last_x, DWS_ERR_STATE, "read_tap before forward");
            DWS_CHECK(capacity >= B * D * L, DWS_ERR_INVALID, "tap buffer too small");
            DWS_TRY(scratch_out.ensure((size_t)B * Cout * L * 4));
            return final_stage(last_x, scratch_out.f(), dst, s);
        }
        // function-level taps: the step embedding, the embedding MLP's output and the stacked
        // fc_t rows of the last per-clip forward; the network's final TransposedLayerNorm (`sashimi.py:309`)
        struct { const char* name; const DevBuf* buf; size_t n; } small[] = {
            {"emb", &emb, (size_t)B * Ein}, {"emb_mlp", &h2, (size_t)B * Eout}, {"part_t", &part_t, (size_t)B * pt_total},
            {"nfin", &nfin, last_x ? (size_t)B * D * L : 0}};
        for (auto& e : small)
            if (t == e.name) {
                DWS_CHECK(e.buf->p && e.n > 0 && capacity >= (int64_t)e.n, DWS_ERR_INVALID, "tap '%s': %zu floats, capacity %lld",
                          tap, e.n, (long long)capacity);
                DWS_HIP(hipMemcpyAsync(dst, e.buf->p, e.n * 4, hipMemcpyDeviceToDevice, s));
                return DWS_OK;
            }
        if (t.rfind("k:", 0) == 0) {  // S4 kernel of the block with this prefix: L * k, k = [2][H][L] (s4.py:796-805)
            if (dirty) DWS_TRY(commit(s));
            for (auto* l : all)
                if (l->kind == L_BLOCK && l->prefix == t.substr(2)) {
                    DWS_TRY(build_kernel(l, s));  // regenerates the (unnormalised) time-domain kernel into scratch
                    const size_t n = (size_t)2 * l->H * l->Lk;
                    DWS_CHECK((size_t)capacity >= n, DWS_ERR_INVALID, "tap buffer too small");
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
