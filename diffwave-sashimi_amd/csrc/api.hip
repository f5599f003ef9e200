// extern "C" surface of libdws.so for the model and sampler entry points
// (include/dws.h); the Cauchy entry points live in cauchy_kernels.hip.
#include <cstring>
#include "model.h"

namespace dws {
int sampler_run(dws_model* m, float* x, const float* alpha, const float* alpha_bar, const float* sigma, int T,
                const float* noise, uint64_t seed, int init_from_seed, int use_graph, hipStream_t s);
int sampler_steps(dws_model* m, float* x, const float* alpha, const float* alpha_bar, const float* sigma, int T,
                  int t_start, int n_steps, uint64_t seed, int use_graph, hipStream_t s);
}  // namespace dws

dws_model::~dws_model() {
    drop_graph();
    if (smp_ev_in) hipEventDestroy(smp_ev_in);
    if (smp_ev_out) hipEventDestroy(smp_ev_out);
    if (smp_stream) hipStreamDestroy(smp_stream);
    for (int i = 0; i < COPY_SLOTS; ++i) {
        if (copy_consumed[i]) (void)hipEventDestroy(copy_consumed[i]);
        if (copy_pinned[i]) (void)hipHostFree(copy_pinned[i]);
    }
    for (auto& g : grad_groups)
        if (g.ev) (void)hipEventDestroy(g.ev);
    for (auto* p : params) delete p;
}

void dws_model::drop_graph() {
    if (smp_graph) hipGraphExecDestroy(smp_graph);
    smp_graph = nullptr;
}

dws::ParamSpec* dws_model::add_param(const std::string& name, std::vector<int64_t> shape, int dtype) {
    auto* p = new dws::ParamSpec();
    p->name = name;
    p->shape = std::move(shape);
    p->dtype = dtype;
    index[name] = (int)params.size();
    params.push_back(p);
    return p;
}

int dws_model::forward_train(const float*, const float*, float*, hipStream_t) {
    return dws::set_error(DWS_ERR_UNSUPPORTED, "the training forward/backward of this backbone is not built yet");
}

int dws_model::backward(const float*, hipStream_t) {
    return dws::set_error(DWS_ERR_UNSUPPORTED, "the training forward/backward of this backbone is not built yet");
}

float* dws_model::G(const std::string& name) {
    auto it = index.find(name);
    if (it == index.end()) return nullptr;
    dws::ParamSpec* p = params[it->second];
    if (!p->grad.p) {
        if (p->grad.ensure(p->nbytes()) != DWS_OK) return nullptr;
        hipMemset(p->grad.p, 0, p->nbytes());
    }
    if (in_backward && !grad_touch.empty()) {      // staged hand-over: this gradient is (still) being produced at flush point grad_seq
        grad_touch[it->second] = grad_seq;
        const int g = grad_group_of[it->second];
        if (g >= 0 && grad_groups[g].flushed) grad_groups[g].touched_after_flush = true;
    }
    return p->grad.f();
}

int dws_model::set_grad_sinks(int32_t count, const char* const* names, float* const* dsts, const int64_t* numels,
                              const int32_t* groups, int32_t ngroups) {
    for (auto& g : grad_groups)
        if (g.ev) (void)hipEventDestroy(g.ev);
    grad_groups.clear();
    grad_group_of.assign(params.size(), -1);
    grad_sink.assign(params.size(), nullptr);
    grad_touch.assign(params.size(), -1);
    grad_touch_prev.clear();
    grad_order_known = false;
    if (count == 0) { grad_touch.clear(); return DWS_OK; }
    grad_groups.resize((size_t)ngroups);
    for (int i = 0; i < count; ++i) {
        auto it = index.find(names[i]);
        DWS_CHECK(it != index.end(), DWS_ERR_INVALID, "unknown parameter '%s'", names[i]);
        dws::ParamSpec* p = params[it->second];
        DWS_CHECK(p->dtype == 0 && (int64_t)p->numel() == numels[i] && dsts[i], DWS_ERR_INVALID,
                  "'%s': gradient sink of %lld elements for a tensor of %zu", names[i], (long long)numels[i], p->numel());
        DWS_CHECK(groups[i] >= 0 && groups[i] < ngroups, DWS_ERR_INVALID, "'%s': group %d of %d", names[i], groups[i], ngroups);
        grad_group_of[it->second] = groups[i];
        grad_sink[it->second] = dsts[i];
        grad_groups[groups[i]].params.push_back(it->second);
    }
    for (auto& g : grad_groups) DWS_HIP(hipEventCreateWithFlags(&g.ev, hipEventDisableTiming));
    return DWS_OK;
}

void dws_model::grad_begin() {
    in_backward = true;
    grad_seq = 0;
    std::fill(grad_touch.begin(), grad_touch.end(), -1);
    for (auto& g : grad_groups) g.flushed = g.touched_after_flush = false;
}

int dws_model::grad_flush(GradGroup& g, hipStream_t s) {
    g.copy.begin();
    for (int pi : g.params) {
        dws::ParamSpec* p = params[pi];
        const bool was = in_backward;
        in_backward = false;               // (fetching the buffer for the copy is not a production of the gradient)
        float* src = G(p->name);
        in_backward = was;
        DWS_CHECK(src, DWS_ERR_HIP, "could not allocate the gradient of '%s'", p->name.c_str());
        g.copy.add(src, grad_sink[pi], p->numel());
    }
    DWS_TRY(g.copy.run(s));
    DWS_HIP(hipEventRecord(g.ev, s));
    g.flushed = true;
    g.touched_after_flush = false;
    return DWS_OK;
}

int dws_model::grad_point(hipStream_t s) {
    if (!in_backward || grad_groups.empty()) return DWS_OK;
    if (grad_order_known)
        for (auto& g : grad_groups)
            if (!g.flushed && g.ready_seq <= grad_seq) DWS_TRY(grad_flush(g, s));
    ++grad_seq;
    return DWS_OK;
}

int dws_model::grad_end(hipStream_t s) {
    if (!in_backward) return DWS_OK;
    // whatever is left, and any group a late write invalidated (the learnt order was wrong for it: re-copied, its event
    // re-recorded -- the host waits on the events only after backward has returned, so it sees this record)
    bool order_held = true;
    for (auto& g : grad_groups) {
        if (g.touched_after_flush) order_held = false;
        if (!g.flushed || g.touched_after_flush) DWS_TRY(grad_flush(g, s));
    }
    in_backward = false;
    if (grad_groups.empty()) return DWS_OK;
    // learn: a group is ready after the flush point of its last-produced gradient.  The order of THIS backward is used by the
    // next one (the graph is static); a backward that disagrees with its predecessor -- new shapes, another config, a late
    // write caught above -- makes the next one a recording-only backward again.
    const bool same = grad_touch_prev.empty() || grad_touch_prev == grad_touch;
    for (auto& g : grad_groups) {
        int r = 0;
        for (int pi : g.params) r = std::max(r, grad_touch[pi] + 1);
        g.ready_seq = r;
    }
    grad_order_known = same && order_held;
    grad_touch_prev = grad_touch;
    return DWS_OK;
}

int dws_model::set_option(const std::string& key, const std::string& value) {
    return dws::set_error(DWS_ERR_INVALID, "unknown option %s=%s for this model", key.c_str(), value.c_str());
}

float* dws_model::P(const std::string& name) const {
    auto it = index.find(name);
    if (it == index.end()) return nullptr;
    return params[it->second]->buf.f();
}

int dws_model::alloc_params() {
    for (auto* p : params) {
        DWS_TRY(p->buf.ensure(p->nbytes()));
        DWS_HIP(hipMemset(p->buf.p, 0, p->nbytes()));
    }
    return DWS_OK;
}

extern "C" {

int dws_model_create(const dws_model_desc* desc, dws_model** out) {
    DWS_CHECK(desc && out, DWS_ERR_INVALID, "dws_model_create: null argument");
    const dws_model_desc& d = *desc;
    DWS_CHECK(d.in_channels > 0 && d.out_channels > 0, DWS_ERR_INVALID, "in/out channels must be positive");
    DWS_CHECK(d.diffusion_step_embed_dim_in > 0 && d.diffusion_step_embed_dim_in % 2 == 0, DWS_ERR_INVALID,
              "diffusion_step_embed_dim_in must be even (`models/utils.py:20`)");
    DWS_CHECK(d.diffusion_step_embed_dim_in <= 1024 && d.diffusion_step_embed_dim_mid <= 1024 &&
                  d.diffusion_step_embed_dim_out <= 1024,
              DWS_ERR_UNSUPPORTED, "embedding dims above 1024 are not supported");
    dws_model* m = nullptr;
    if (d.kind == DWS_KIND_WAVENET) {
        DWS_CHECK(d.res_channels > 0 && d.skip_channels > 0 && d.num_res_layers > 0 && d.dilation_cycle > 0,
                  DWS_ERR_INVALID, "wavenet: bad channel/layer counts");
        DWS_CHECK(d.dilation_cycle <= 24, DWS_ERR_UNSUPPORTED, "dilation_cycle > 24");
        m = dws::make_wavenet(d);
    } else if (d.kind == DWS_KIND_SASHIMI) {
        m = dws::make_sashimi(d);
        if (!m) return DWS_ERR_INVALID;  // message set by make_sashimi
    } else {
        return dws::set_error(DWS_ERR_INVALID, "unknown model kind %d (model._name_ must be wavenet|sashimi, "
                                               "`models/__init__.py:6-9`)", d.kind);
    }
    int st = m->alloc_params();
    if (st != DWS_OK) {
        delete m;
        return st;
    }
    *out = m;
    return DWS_OK;
}

int dws_model_destroy(dws_model* m) {
    delete m;
    return DWS_OK;
}

int dws_model_num_params(const dws_model* m) { return m ? (int)m->params.size() : 0; }

int dws_model_param_info(const dws_model* m, int i, const char** name, int64_t* shape, int* ndim, int* dtype) {
    DWS_CHECK(m && i >= 0 && i < (int)m->params.size(), DWS_ERR_INVALID, "param index %d out of range", i);
    const dws::ParamSpec* p = m->params[i];
    DWS_CHECK(p->shape.size() <= 8, DWS_ERR_INVALID, "rank > 8");
    if (name) *name = p->name.c_str();
    if (ndim) *ndim = (int)p->shape.size();
    if (dtype) *dtype = p->dtype;
    if (shape)
        for (size_t k = 0; k < p->shape.size(); ++k) shape[k] = p->shape[k];
    return DWS_OK;
}

int dws_model_set_param(dws_model* m, const char* name, const void* data, const int64_t* shape, int ndim, int dtype,
                        void* stream) {
    DWS_CHECK(m && name && data && shape, DWS_ERR_INVALID, "dws_model_set_param: null argument");
    auto it = m->index.find(name);
    if (it == m->index.end() && name[0] == '_' && name[1] == '_') {
        // host-computed tables ("__omega.<L>", "__z.<L>": FFT nodes of an S4 kernel length the model was not
        // configured with, e.g. a checkpoint trained at another l_max) are registered on first sight
        dws::ParamSpec* np_ = m->add_param(name, std::vector<int64_t>(shape, shape + ndim), dtype);
        DWS_TRY(np_->buf.ensure(np_->nbytes()));
        it = m->index.find(name);
    }
    DWS_CHECK(it != m->index.end(), DWS_ERR_INVALID, "unexpected key '%s' in state_dict", name);
    dws::ParamSpec* p = m->params[it->second];
    DWS_CHECK(dtype == p->dtype, DWS_ERR_INVALID, "'%s': dtype %d, expected %d", name, dtype, p->dtype);
    bool same = (int)p->shape.size() == ndim;
    for (int k = 0; same && k < ndim; ++k) same = p->shape[k] == shape[k];
    if (!same) {
        std::string got = "[", want = "[";
        for (int k = 0; k < ndim; ++k) got += std::to_string(shape[k]) + (k + 1 < ndim ? "," : "");
        for (size_t k = 0; k < p->shape.size(); ++k) want += std::to_string(p->shape[k]) + (k + 1 < p->shape.size() ? "," : "");
        return dws::set_error(DWS_ERR_INVALID, "size mismatch for %s: got %s], expected %s]", name, got.c_str(),
                              want.c_str());
    }
    DWS_HIP(hipMemcpyAsync(p->buf.p, data, p->nbytes(), hipMemcpyDefault, (hipStream_t)stream));
    if (dtype == 1) ++m->int_params_version;
    m->dirty = true;
    m->drop_graph();
    return DWS_OK;
}

int dws_model_set_option(dws_model* m, const char* key, const char* value) {
    DWS_CHECK(m && key && value, DWS_ERR_INVALID, "dws_model_set_option: null argument");
    m->drop_graph();
    return m->set_option(key, value);
}

int dws_model_commit(dws_model* m, void* stream) {
    DWS_CHECK(m, DWS_ERR_INVALID, "null model");
    return m->commit((hipStream_t)stream);
}

int dws_model_prepare(dws_model* m, int64_t B, int64_t L) {
    DWS_CHECK(m, DWS_ERR_INVALID, "null model");
    return m->prepare(B, L);
}

int dws_model_set_condition(dws_model* m, const float* mel, int64_t Bm, int64_t Tmel, void* stream) {
    DWS_CHECK(m, DWS_ERR_INVALID, "null model");
    m->drop_graph();
    return m->set_condition(mel, Bm, Tmel, (hipStream_t)stream);
}

int dws_model_forward(dws_model* m, const float* audio, const float* steps, float* out, void* stream) {
    DWS_CHECK(m && audio && steps && out, DWS_ERR_INVALID, "dws_model_forward: null argument");
    return m->forward(audio, steps, out, (hipStream_t)stream);
}

int dws_model_forward_train(dws_model* m, const float* audio, const float* steps, float* out, void* stream) {
    DWS_CHECK(m && audio && steps && out, DWS_ERR_INVALID, "dws_model_forward_train: null argument");
    return m->forward_train(audio, steps, out, (hipStream_t)stream);
}

int dws_model_backward(dws_model* m, const float* dout, void* stream) {
    DWS_CHECK(m && dout, DWS_ERR_INVALID, "dws_model_backward: null argument");
    m->grad_begin();
    const int rc = m->backward(dout, (hipStream_t)stream);
    if (rc != DWS_OK) { m->in_backward = false; return rc; }
    return m->grad_end((hipStream_t)stream);
}

int dws_model_set_grad_sinks(dws_model* m, int32_t count, const char* const* names, float* const* dsts, const int64_t* numels,
                             const int32_t* groups, int32_t ngroups) {
    DWS_CHECK(m && count >= 0 && ngroups >= 0 && (count == 0 || (names && dsts && numels && groups && ngroups > 0)), DWS_ERR_INVALID,
              "dws_model_set_grad_sinks: null argument");
    return m->set_grad_sinks(count, names, dsts, numels, groups, ngroups);
}

int dws_model_grad_group_wait(dws_model* m, int32_t group, void* waiting_stream) {
    DWS_CHECK(m && group >= 0 && (size_t)group < m->grad_groups.size(), DWS_ERR_INVALID, "dws_model_grad_group_wait: group %d of %zu",
              group, m ? m->grad_groups.size() : (size_t)0);
    DWS_CHECK(m->grad_groups[group].flushed, DWS_ERR_STATE, "dws_model_grad_group_wait: no backward has handed group %d over", group);
    DWS_HIP(hipStreamWaitEvent((hipStream_t)waiting_stream, m->grad_groups[group].ev, 0));
    return DWS_OK;
}

int dws_model_grad_ready_seq(dws_model* m, int32_t count, const char* const* names, int32_t* seq_out) {
    DWS_CHECK(m && names && seq_out && count >= 0, DWS_ERR_INVALID, "dws_model_grad_ready_seq: null argument");
    DWS_CHECK(!m->grad_touch_prev.empty(), DWS_ERR_STATE, "dws_model_grad_ready_seq: no backward with gradient sinks has run");
    for (int i = 0; i < count; ++i) {
        auto it = m->index.find(names[i]);
        DWS_CHECK(it != m->index.end(), DWS_ERR_INVALID, "unknown parameter '%s'", names[i]);
        seq_out[i] = m->grad_touch_prev[it->second];
    }
    return DWS_OK;
}

int dws_model_get_grad(dws_model* m, const char* name, float* dst, int64_t numel, void* stream) {
    DWS_CHECK(m && name && dst, DWS_ERR_INVALID, "dws_model_get_grad: null argument");
    auto it = m->index.find(name);
    DWS_CHECK(it != m->index.end(), DWS_ERR_INVALID, "unknown parameter '%s'", name);
    dws::ParamSpec* p = m->params[it->second];
    DWS_CHECK(p->dtype == 0 && (int64_t)p->numel() == numel, DWS_ERR_INVALID, "'%s': grad has %zu elements, got %lld", name,
              p->numel(), (long long)numel);
    float* g = m->G(name);
    DWS_CHECK(g, DWS_ERR_HIP, "could not allocate the gradient of '%s'", name);
    DWS_HIP(hipMemcpyAsync(dst, g, p->nbytes(), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DWS_OK;
}

static int multi_copy(dws_model* m, std::vector<dws::CopyJob>& jobs, hipStream_t stream);

int dws_model_update_params(dws_model* m, int32_t count, const char* const* names, const float* const* srcs, void* stream) {
    DWS_CHECK(m && names && srcs && count >= 0, DWS_ERR_INVALID, "dws_model_update_params: null argument");
    if (count == 0) return DWS_OK;
    std::vector<dws::CopyJob> jobs((size_t)count);
    for (int i = 0; i < count; ++i) {
        auto it = m->index.find(names[i]);
        DWS_CHECK(it != m->index.end(), DWS_ERR_INVALID, "unexpected key '%s' in state_dict", names[i]);
        dws::ParamSpec* p = m->params[it->second];
        DWS_CHECK(p->dtype == 0 && srcs[i], DWS_ERR_INVALID, "'%s': update_params takes float32 device tensors", names[i]);
        jobs[i] = {srcs[i], p->buf.f(), (int64_t)p->numel()};
    }
    DWS_TRY(multi_copy(m, jobs, (hipStream_t)stream));
    m->dirty = true;
    m->drop_graph();
    return DWS_OK;
}

int dws_model_get_grads(dws_model* m, int32_t count, const char* const* names, float* const* dsts, const int64_t* numels,
                        void* stream) {
    DWS_CHECK(m && names && dsts && numels && count >= 0, DWS_ERR_INVALID, "dws_model_get_grads: null argument");
    if (count == 0) return DWS_OK;
    std::vector<dws::CopyJob> jobs((size_t)count);
    for (int i = 0; i < count; ++i) {
        auto it = m->index.find(names[i]);
        DWS_CHECK(it != m->index.end(), DWS_ERR_INVALID, "unknown parameter '%s'", names[i]);
        dws::ParamSpec* p = m->params[it->second];
        DWS_CHECK(p->dtype == 0 && (int64_t)p->numel() == numels[i], DWS_ERR_INVALID, "'%s': grad has %zu elements, got %lld",
                  names[i], p->numel(), (long long)numels[i]);
        float* g = m->G(names[i]);
        DWS_CHECK(g && dsts[i], DWS_ERR_HIP, "could not allocate the gradient of '%s'", names[i]);
        jobs[i] = {g, dsts[i], numels[i]};
    }
    return multi_copy(m, jobs, (hipStream_t)stream);
}

// One kernel for a list of device-to-device copies.  The job table travels through a pinned staging buffer so nothing
// blocks the host; an event per staging slot guards its reuse.  The device-side table is written and read in stream order.
static int multi_copy(dws_model* m, std::vector<dws::CopyJob>& jobs, hipStream_t stream) {
    const int slot = m->copy_slot;
    m->copy_slot = (slot + 1) % dws_model::COPY_SLOTS;
    if (!m->copy_consumed[slot]) DWS_HIP(hipEventCreateWithFlags(&m->copy_consumed[slot], hipEventDisableTiming));
    else DWS_HIP(hipEventSynchronize(m->copy_consumed[slot]));
    if (m->copy_pinned_cap[slot] < jobs.size()) {
        if (m->copy_pinned[slot]) (void)hipHostFree(m->copy_pinned[slot]);
        m->copy_pinned_cap[slot] = jobs.size() * 2;
        DWS_HIP(hipHostMalloc(&m->copy_pinned[slot], m->copy_pinned_cap[slot] * sizeof(dws::CopyJob), hipHostMallocDefault));
    }
    std::memcpy(m->copy_pinned[slot], jobs.data(), jobs.size() * sizeof(dws::CopyJob));
    dws::DevBuf& table = m->copy_table;
    DWS_TRY(table.ensure(jobs.size() * 2 * sizeof(dws::CopyJob)));
    DWS_HIP(hipMemcpyAsync(table.p, m->copy_pinned[slot], jobs.size() * sizeof(dws::CopyJob), hipMemcpyHostToDevice, stream));
    DWS_HIP(hipEventRecord(m->copy_consumed[slot], stream));
    return dws::launch_multi_copy((const dws::CopyJob*)table.p, (int)jobs.size(), stream);
}

int dws_model_read_tap(dws_model* m, const char* tap, float* dst, int64_t capacity, void* stream) {
    DWS_CHECK(m && tap && dst, DWS_ERR_INVALID, "dws_model_read_tap: null argument");
    if (std::strcmp(tap, "sampler_eps") == 0) {     // the network output of the sampler's last reverse step
        const int64_t n = m->B * m->d.out_channels * m->L;
        DWS_CHECK(m->smp_eps.p && capacity >= n, DWS_ERR_INVALID, "tap sampler_eps: no sampler step has run, or capacity %lld < %lld",
                  (long long)capacity, (long long)n);
        // the buffer is sized by the last sampler step: after a prepare() to a larger shape it no longer matches
        DWS_CHECK(m->smp_eps.bytes >= (size_t)n * 4 && m->smp_eps_B == m->B && m->smp_eps_L == m->L, DWS_ERR_STATE,
                  "tap sampler_eps: the last sampler step ran at B=%lld L=%lld, the model is prepared for B=%lld L=%lld",
                  (long long)m->smp_eps_B, (long long)m->smp_eps_L, (long long)m->B, (long long)m->L);
        DWS_HIP(hipMemcpyAsync(dst, m->smp_eps.p, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return DWS_OK;
    }
    return m->read_tap(tap, dst, capacity, (hipStream_t)stream);
}

int dws_sampler_run(dws_model* m, float* x, const float* alpha, const float* alpha_bar, const float* sigma,
                    int32_t T, const float* noise, uint64_t seed, int32_t init_from_seed, int32_t use_graph,
                    void* stream) {
    DWS_CHECK(m && x, DWS_ERR_INVALID, "dws_sampler_run: null argument");
    return dws::sampler_run(m, x, alpha, alpha_bar, sigma, T, noise, seed, init_from_seed, use_graph,
                            (hipStream_t)stream);
}

int dws_sampler_steps(dws_model* m, float* x, const float* alpha, const float* alpha_bar, const float* sigma,
                      int32_t T, int32_t t_start, int32_t n_steps, uint64_t seed, int32_t use_graph, void* stream) {
    DWS_CHECK(m && x, DWS_ERR_INVALID, "dws_sampler_steps: null argument");
    return dws::sampler_steps(m, x, alpha, alpha_bar, sigma, T, t_start, n_steps, seed, use_graph,
                              (hipStream_t)stream);
}

}  // extern "C"
