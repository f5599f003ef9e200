// dws_model: base of the two backbones behind the C ABI (include/dws.h).
#pragma once
#include <cstring>

#include "dws_common.h"

namespace dws {

// RAII device buffer.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    ~DevBuf() { release(); }
    void release() {
        if (p) hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    // (re)allocate when the request grows; contents are not preserved.
    int ensure(size_t n) {
        if (n <= bytes && p) return DWS_OK;
        release();
        if (n == 0) n = 4;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return set_error(DWS_ERR_HIP, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        return DWS_OK;
    }
    float* f() const { return static_cast<float*>(p); }
};

struct CopyJob { const float* src; float* dst; int64_t n; };
int launch_multi_copy(const CopyJob* table_dev, int njobs, hipStream_t s);   // api.hip: one kernel for a job table

// A fixed list of device-to-device copies run as ONE launch (the per-layer fc_t / bias tensors into their stacked
// buffers and the stacked gradients back out: 4-5 hipMemcpyAsync per layer and step otherwise).  The job table is
// uploaded only when it differs from the one already on the device (buffers are allocated once, so: the first time).
struct CopyBatch {
    std::vector<CopyJob> jobs, uploaded;
    DevBuf table;
    void begin() { jobs.clear(); }
    void add(const float* src, float* dst, size_t n) { jobs.push_back({src, dst, (int64_t)n}); }
    int run(hipStream_t s) {
        if (jobs.empty()) return DWS_OK;
        const bool same = jobs.size() == uploaded.size() &&
                          std::memcmp(jobs.data(), uploaded.data(), jobs.size() * sizeof(CopyJob)) == 0;
        if (!same) {
            DWS_HIP(hipStreamSynchronize(s));   // nothing in flight may still read the old table / host copy
            uploaded = jobs;
            DWS_TRY(table.ensure(uploaded.size() * sizeof(CopyJob)));
            DWS_HIP(hipMemcpy(table.p, uploaded.data(), uploaded.size() * sizeof(CopyJob), hipMemcpyHostToDevice));
        }
        return launch_multi_copy((const CopyJob*)table.p, (int)jobs.size(), s);
    }
};

// Weight preparation of a commit as job tables (wavenet_kernels.hip: weight_prep_kernel): the weight-norm folds, MFMA fragment
// packs and row sums of ALL weights in two launches (folds, then everything that reads a folded or raw weight) instead of one
// 4-5 us launch each -- a training step commits once per step: 286 launches = 1.35 ms of a config-5 step.  Each job runs the
// very arithmetic of its single launcher (same block shape, same reduction order): results are bit-identical.
enum { PREP_FOLD = 0, PREP_PACK = 1, PREP_PACK_T = 2, PREP_ROW_SUM = 3 };
struct PrepJob { const float* a; const float* b; float* out; int n0, n1, kind, first_block; };
int prep_job_blocks(int kind, int n0, int n1);
int launch_weight_prep(const PrepJob* table_dev, int njobs, int nblocks, hipStream_t s);

struct PrepBatch {
    std::vector<PrepJob> jobs[2], uploaded[2];    // phase 0: folds; phase 1: packs / row sums (may read a phase-0 output)
    DevBuf table[2];
    int nblocks[2] = {0, 0};
    void begin() { jobs[0].clear(); jobs[1].clear(); nblocks[0] = nblocks[1] = 0; }
    void add(int kind, const float* a, const float* b, float* out, int n0, int n1) {
        const int ph = kind == PREP_FOLD ? 0 : 1;
        jobs[ph].push_back({a, b, out, n0, n1, kind, nblocks[ph]});
        nblocks[ph] += prep_job_blocks(kind, n0, n1);
    }
    void fold(const float* v, const float* g, float* out, int O, int inner) { add(PREP_FOLD, v, g, out, O, inner); }
    void pack(const float* w, float* out, int M, int K) { add(PREP_PACK, w, nullptr, out, M, K); }
    void pack_t(const float* w, float* out, int O, int K) { add(PREP_PACK_T, w, nullptr, out, O, K); }
    void row_sum(const float* w, float* rs, int O, int K) { add(PREP_ROW_SUM, w, nullptr, rs, O, K); }
    int run(hipStream_t s) {
        for (int ph = 0; ph < 2; ++ph) {
            if (jobs[ph].empty()) continue;
            const bool same = jobs[ph].size() == uploaded[ph].size() &&
                              std::memcmp(jobs[ph].data(), uploaded[ph].data(), jobs[ph].size() * sizeof(PrepJob)) == 0;
            if (!same) {      // (buffers are allocated once: the first commit of a model, or a re-shaped one)
                DWS_HIP(hipStreamSynchronize(s));
                uploaded[ph] = jobs[ph];
                DWS_TRY(table[ph].ensure(uploaded[ph].size() * sizeof(PrepJob)));
                DWS_HIP(hipMemcpy(table[ph].p, uploaded[ph].data(), uploaded[ph].size() * sizeof(PrepJob), hipMemcpyHostToDevice));
            }
            DWS_TRY(launch_weight_prep((const PrepJob*)table[ph].p, (int)jobs[ph].size(), nblocks[ph], s));
        }
        return DWS_OK;
    }
};

struct ParamSpec {
    std::string name;
    std::vector<int64_t> shape;
    int dtype = 0;  // 0 float32, 1 int64
    DevBuf buf;     // raw copy of the state-dict tensor
    DevBuf grad;    // gradient w.r.t. the raw tensor (training path; allocated on first backward)
    size_t numel() const {
        size_t n = 1;
        for (auto s : shape) n *= (size_t)s;
        return n;
    }
    size_t nbytes() const { return numel() * (dtype == 1 ? 8 : 4); }
};

}  // namespace dws

struct dws_model {
    dws_model_desc d;
    std::vector<dws::ParamSpec*> params;
    std::map<std::string, int> index;
    bool dirty = true;
    uint64_t int_params_version = 1;  // bumped whenever an int64 tensor (S4 `L` buffers) is (re)set
    int64_t B = 0, L = 0;  // prepared workspace shape

    dws::DevBuf lin_scratch;  // split-O partials of the embedding adjoint
    // batched parameter / gradient copies (api.hip): device job table, pinned staging, reuse guard
    // multi_copy staging: the job table of a call travels through one of COPY_SLOTS pinned buffers (round robin; an
    // event per slot guards its reuse, so the host only ever waits for the call made COPY_SLOTS calls ago and can run
    // a training step ahead of the GPU)
    static constexpr int COPY_SLOTS = 4;
    dws::DevBuf copy_table;
    void* copy_pinned[COPY_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t copy_pinned_cap[COPY_SLOTS] = {0, 0, 0, 0};
    hipEvent_t copy_consumed[COPY_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int copy_slot = 0;

    // sampler state (sampler.hip)
    dws::DevBuf smp_tables;   // [3][T] c1, c2, sigma
    int smp_T = 0;            // length of the uploaded tables
    std::vector<float> smp_host_tables;  // host copy of what is resident (upload skipped when identical)
    dws::DevBuf smp_state;    // int32 step index
    dws::DevBuf smp_eps;      // eps[B, Cout, L]
    int64_t smp_eps_B = 0, smp_eps_L = 0;   // shape of the sampler step that last wrote it
    const int* step_idx = nullptr;   // set by the sampler around forward(): step-table mode, row *step_idx (device memory)
    hipGraphExec_t smp_graph = nullptr;
    hipStream_t smp_stream = nullptr;  // capture/replay stream (the caller's may be the null stream)
    hipEvent_t smp_ev_in = nullptr, smp_ev_out = nullptr;
    // key of the captured graph
    int64_t g_B = 0, g_L = 0;
    int g_T = 0;
    const void* g_x = nullptr;
    const void* g_noise = nullptr;
    uint64_t g_seed = 0;

    // ---- staged gradient hand-over (data-parallel overlap, dws_model_set_grad_sinks): the host names a destination and a
    // GROUP (its all-reduce bucket) per parameter; backward() copies a group's gradients to their destinations (one launch)
    // and records the group's event the moment the group's last gradient has been produced, so the host can start that
    // bucket's all-reduce while the rest of backward still runs.  When a gradient is final is LEARNT: G() stamps every
    // access during a backward with the number of the flush point (grad_point) it precedes; the first backward after the
    // sinks (or the model's graph) changed only records and flushes everything at its end.
    struct GradGroup {
        std::vector<int> params;       // parameter indices
        dws::CopyBatch copy;           // grads -> sinks, table uploaded once
        hipEvent_t ev = nullptr;       // recorded behind the group's copy
        int ready_seq = 0;             // flush point after which every gradient of the group is final (learnt)
        bool flushed = false, touched_after_flush = false;
    };
    std::vector<GradGroup> grad_groups;
    std::vector<int> grad_group_of;        // per parameter: group or -1
    std::vector<float*> grad_sink;         // per parameter: destination or null
    std::vector<int> grad_touch, grad_touch_prev;   // per parameter: flush point of the last G() of this / the previous backward
    bool grad_order_known = false, in_backward = false;
    int grad_seq = 0;
    int set_grad_sinks(int32_t count, const char* const* names, float* const* dsts, const int64_t* numels, const int32_t* groups,
                       int32_t ngroups);
    void grad_begin();                 // start of a backward
    int grad_point(hipStream_t s);     // a flush opportunity inside backward (after a layer / block)
    int grad_end(hipStream_t s);       // end of a backward: flush what is left, learn the order
    int grad_flush(GradGroup& g, hipStream_t s);

    virtual ~dws_model();
    dws::ParamSpec* add_param(const std::string& name, std::vector<int64_t> shape, int dtype = 0);
    float* P(const std::string& name) const;  // raw device pointer of a parameter
    int alloc_params();
    void drop_graph();

    virtual int set_option(const std::string& key, const std::string& value);
    virtual int commit(hipStream_t s) = 0;
    virtual int prepare(int64_t B, int64_t L) = 0;
    virtual int set_condition(const float* mel, int64_t Bm, int64_t Tmel, hipStream_t s) = 0;
    // steps may be null only while step_idx is set (sampler): the step-only terms then come from the step table
    virtual int forward(const float* audio, const float* steps, float* out, hipStream_t s) = 0;
    // evaluate everything that depends on the diffusion step only for t = 0..T-1 (kept until the weights or T change)
    virtual int build_step_table(int T, hipStream_t s) = 0;
    virtual int read_tap(const char* tap, float* dst, int64_t capacity, hipStream_t s) = 0;
    // training path: forward that keeps what backward needs; backward fills ParamSpec::grad of every parameter
    virtual int forward_train(const float* audio, const float* steps, float* out, hipStream_t s);
    virtual int backward(const float* dout, hipStream_t s);
    float* G(const std::string& name);  // gradient buffer of a parameter (allocated, zeroed on first use)
};

namespace dws {
dws_model* make_wavenet(const dws_model_desc& d);
dws_model* make_sashimi(const dws_model_desc& d);
}  // namespace dws
