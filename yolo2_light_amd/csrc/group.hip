// group.hip -- one host process, several GPUs: the image batch of network_predict sharded over the
// devices of a node, detections gathered over xGMI with RCCL.
//
// The reference selects ONE device (`-i <n>` -> cuda_set_device, src/main.c:653-661) and its
// network_predict_* entry points take the whole batch (src/yolov2_forward_network.c:632); SURVEY 8(e)
// asks for the image batch of that same call to be split across the GPUs of the node with a final
// gather of the (small, fixed-capacity) detection records.  Images are independent, so:
//   * weights are replicated: one yl_network replica per device, built from the caller's prepared host
//     model (after fuse / quantise / any yl_network_set_* knob) with the per-device share of the batch
//     (shard_range: the first global_batch % n devices take one image more);
//   * one persistent host thread per device issues that replica's launches on its own HIP stream -- a
//     single thread walking 8 devices x ~110 launches would serialise ~0.5 ms of launch cost per device;
//     replicas share no mutable state (per-network kernel knobs, yl_internal.h);
//   * the only exchange: detection records [images][cap][6+classes] + counts, sent to the root device
//     with ncclSend/ncclRecv inside one group call (uneven shards need per-rank counts, so not
//     ncclGather), stream-ordered behind each replica's decode + NMS kernels.  The communicator is
//     RCCL's single-process form (ncclCommInitAll); librccl is resolved with dlopen at first use so a
//     single-GPU user of libyolo2hip.so never loads it (and inside a torch process torch's copy is reused).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "yl_internal.h"

namespace yl {

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl &rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.why = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
#define YL_SYM(field, sym)                                                                          \
        r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, sym));                        \
        if (!r.field) { r.why = std::string("librccl lacks ") + sym; r.handle = nullptr; return; }
        YL_SYM(CommInitAll, "ncclCommInitAll")
        YL_SYM(CommDestroy, "ncclCommDestroy")
        YL_SYM(GroupStart, "ncclGroupStart")
        YL_SYM(GroupEnd, "ncclGroupEnd")
        YL_SYM(Send, "ncclSend")
        YL_SYM(Recv, "ncclRecv")
        YL_SYM(GetErrorString, "ncclGetErrorString")
#undef YL_SYM
    });
    return r;
}

// one persistent thread per device: run(job) hands a closure over, wait() collects its return code
class Worker {
public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker()
    {
        { std::lock_guard<std::mutex> g(m_); quit_ = true; }
        cv_.notify_all();
        th_.join();
    }
    void run(std::function<int()> job)
    {
        { std::lock_guard<std::mutex> g(m_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    int wait(std::string &err)
    {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [this] { return !busy_; });
        err = err_;
        return rc_;
    }

private:
    void loop()
    {
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [this] { return quit_ || (busy_ && job_); });
                if (quit_) return;
                job = std::move(job_);
                job_ = nullptr;
            }
            const int rc = job();
            {
                std::lock_guard<std::mutex> g(m_);
                rc_ = rc;
                err_ = rc == YL_OK ? std::string() : std::string(yl_last_error());
                busy_ = false;
            }
            cv_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool busy_ = false, quit_ = false;
    int rc_ = YL_OK;
    std::string err_;
    std::thread th_;
};

}  // namespace

}  // namespace yl

struct yl_group {
    int n = 0;
    int global_batch = 0;
    std::vector<int> devices;
    std::vector<int> first, count;                   // image range of each rank
    std::vector<yl_network *> members;
    std::vector<yl::Worker *> workers;
    std::vector<float *> host_own;                   // group-owned PINNED host tensors (heads / last layer) when the model has none
    std::vector<float *> host_out_ptr;               // per layer: base of the [global_batch][outputs] host tensor (or nullptr)
    std::vector<int> host_kind;                      // per layer: HOST_PINNED (group-owned) or HOST_CALLER (the model's l.output)
    std::vector<ncclComm_t> comms;                   // RCCL, created at the first gather
    std::vector<float *> d_rec;                      // per rank: local detection records / counts staging
    std::vector<int *> d_cnt;
    size_t rec_floats = 0;                           // per image
    float *d_rec_root = nullptr;                     // yl_group_get_boxes_batch staging on the root device
    int *d_cnt_root = nullptr;
    size_t rec_root_bytes = 0;
};

using namespace yl;

namespace {

// run f(rank) on every rank's thread, return the first failure
int for_all(yl_group *g, const std::function<int(int)> &f)
{
    for (int r = 0; r < g->n; ++r) g->workers[r]->run([&f, r] { return f(r); });
    int rc = YL_OK;
    std::string first_err;
    for (int r = 0; r < g->n; ++r) {
        std::string e;
        const int rr = g->workers[r]->wait(e);
        if (rr != YL_OK && rc == YL_OK) { rc = rr; first_err = "rank " + std::to_string(r) + ": " + e; }
    }
    if (rc != YL_OK) set_error(first_err);
    return rc;
}

int ensure_comms(yl_group *g)
{
    if (!g->comms.empty()) return YL_OK;
    Rccl &R = rccl();
    if (!R.handle) { set_error(R.why); return YL_ERR_DEVICE; }
    g->comms.assign((size_t)g->n, nullptr);
    const ncclResult_t rc = R.CommInitAll(g->comms.data(), g->n, g->devices.data());
    if (rc != ncclSuccess) {
        g->comms.clear();
        set_error(std::string("ncclCommInitAll: ") + R.GetErrorString(rc));
        return YL_ERR_DEVICE;
    }
    return YL_OK;
}

}  // namespace

extern "C" {

int yl_group_create(const yl_network *model, const int *devices, int n_devices, yl_group **out)
{
    if (!model || !devices || n_devices <= 0 || !out) { set_error("bad argument"); return YL_ERR_ARG; }
    const Network &m = model->net;
    if (m.on_device) { set_error("the model handed to yl_group_create must not be on a device itself"); return YL_ERR_STATE; }
    if (m.batch < n_devices) { set_error("fewer images than devices"); return YL_ERR_ARG; }
    const int ndev = yl_device_count();
    for (int i = 0; i < n_devices; ++i) {
        if (devices[i] < 0 || devices[i] >= ndev) { set_error("device index out of range (or no GPU: there is no CPU fallback)"); return YL_ERR_DEVICE; }
        for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) { set_error("device listed twice"); return YL_ERR_ARG; }
    }
    yl_group *g = new yl_group();
    g->n = n_devices;
    g->global_batch = m.batch;
    g->devices.assign(devices, devices + n_devices);
    for (int r = 0; r < n_devices; ++r) {
        int f = 0, c = 0;
        (void)yl_shard_range(m.batch, n_devices, r, &f, &c);
        g->first.push_back(f);
        g->count.push_back(c);
    }
    // host tensors of the heads / last layer for the GLOBAL batch: the model's (the reference's l.output, given
    // through yl_layer_desc.output) or group-owned ones; every replica pulls its slice straight into them
    const size_t nl = m.layers.size();
    // Group-owned tensors are pinned (portable: every device DMAs its slice straight into them); caller memory is
    // never handed to the runtime -- each replica bounces its slice through its own pinned block (staging.hip).
    g->host_own.assign(nl, nullptr);
    g->host_out_ptr.assign(nl, nullptr);
    g->host_kind.assign(nl, HOST_NONE);
    for (size_t i = 0; i < nl; ++i) {
        const Layer &l = m.layers[i];
        const bool is_head = (l.type == YL_YOLO || l.type == YL_REGION);
        if (!(is_head || i + 1 == nl)) continue;
        if (l.host_output && l.host_kind == HOST_CALLER) { g->host_out_ptr[i] = l.host_output; g->host_kind[i] = HOST_CALLER; }
        else {
            const size_t bytes = sizeof(float) * (size_t)m.batch * l.outputs;
            if (hipSetDevice(devices[0]) != hipSuccess ||
                hipHostMalloc((void **)&g->host_own[i], bytes, hipHostMallocPortable) != hipSuccess) {
                set_error("hipHostMalloc of the group's host tensors failed");
                yl_group_destroy(g);
                return YL_ERR_DEVICE;
            }
            memset(g->host_own[i], 0, bytes);
            g->host_out_ptr[i] = g->host_own[i];
            g->host_kind[i] = HOST_PINNED;
        }
    }
    for (int r = 0; r < n_devices; ++r) {
        yl_network *rep = new yl_network();
        rep->net = m;                                   // host model: plain data, no device state
        Network &rn = rep->net;
        rn.batch = g->count[r];
        rn.stream = nullptr; rn.own_stream = false; rn.device = -1;
        for (size_t i = 0; i < nl; ++i) {
            Layer &l = rn.layers[i];
            l.batch = rn.batch;
            l.host_in_heads = false;
            l.host_output = g->host_out_ptr[i] ? g->host_out_ptr[i] + (size_t)g->first[r] * l.outputs : nullptr;
            l.host_kind = g->host_out_ptr[i] ? g->host_kind[i] : HOST_NONE;
        }
        g->members.push_back(rep);
        g->workers.push_back(new Worker());
    }
    // parameter upload + buffer planning of all replicas in parallel (packing the weights is host work)
    const int rc = for_all(g, [g](int r) { return yl_network_to_device(g->members[r], g->devices[r]); });
    if (rc != YL_OK) { const std::string keep = yl_last_error(); yl_group_destroy(g); set_error(keep); return rc; }
    g->d_rec.assign((size_t)n_devices, nullptr);
    g->d_cnt.assign((size_t)n_devices, nullptr);
    *out = g;
    return YL_OK;
}

void yl_group_destroy(yl_group *g)
{
    if (!g) return;
    // yl_group_detect_batch is asynchronous: sends / receives may still be queued on the members' streams
    for (int r = 0; r < (int)g->members.size(); ++r) {
        const Network &rn = g->members[r]->net;
        if (rn.on_device && rn.stream && hipSetDevice(g->devices[r]) == hipSuccess) (void)hipStreamSynchronize((hipStream_t)rn.stream);
    }
    if (!g->comms.empty()) {
        Rccl &R = rccl();
        for (ncclComm_t c : g->comms) if (c && R.handle) (void)R.CommDestroy(c);
    }
    for (int r = 0; r < (int)g->members.size(); ++r) {
        (void)hipSetDevice(g->devices[r]);
        if (r < (int)g->d_rec.size() && g->d_rec[r]) (void)hipFree(g->d_rec[r]);
        if (r < (int)g->d_cnt.size() && g->d_cnt[r]) (void)hipFree(g->d_cnt[r]);
        if (r == 0) {
            if (g->d_rec_root) (void)hipFree(g->d_rec_root);
            if (g->d_cnt_root) (void)hipFree(g->d_cnt_root);
        }
        yl_network_destroy(g->members[r]);
    }
    for (Worker *w : g->workers) delete w;
    for (float *p : g->host_own) if (p) (void)hipHostFree(p);
    delete g;
}

int yl_shard_range(int global_batch, int n, int rank, int *first, int *count)
{
    if (global_batch < 0 || n <= 0 || rank < 0 || rank >= n) { set_error("bad argument"); return YL_ERR_ARG; }
    const int q = global_batch / n, rem = global_batch % n;
    if (first) *first = rank * q + (rank < rem ? rank : rem);
    if (count) *count = q + (rank < rem ? 1 : 0);
    return YL_OK;
}

int yl_group_size(const yl_group *g) { return g ? g->n : YL_ERR_ARG; }

int yl_group_shard(const yl_group *g, int rank, int *first, int *count)
{
    if (!g || rank < 0 || rank >= g->n) { set_error("bad argument"); return YL_ERR_ARG; }
    if (first) *first = g->first[rank];
    if (count) *count = g->count[rank];
    return YL_OK;
}

yl_network *yl_group_member(yl_group *g, int rank)
{
    if (!g || rank < 0 || rank >= g->n) { set_error("bad argument"); return nullptr; }
    return g->members[rank];
}

float *yl_group_predict(yl_group *g, const float *input)
{
    if (!g || !input) { set_error("null argument"); return nullptr; }
    const Network &m0 = g->members[0]->net;
    const size_t per_image = (size_t)m0.c * m0.h * m0.w;
    const int rc = for_all(g, [g, input, per_image](int r) {
        return yl_network_predict(g->members[r], input + (size_t)g->first[r] * per_image) ? YL_OK : YL_ERR_DEVICE;
    });
    if (rc != YL_OK) return nullptr;
    return g->host_out_ptr.back();
}

int yl_group_forward(yl_group *g, const float *const *inputs_dev)
{
    if (!g) { set_error("null argument"); return YL_ERR_ARG; }
    return for_all(g, [g, inputs_dev](int r) {
        const float *in = (inputs_dev && inputs_dev[r]) ? inputs_dev[r] : yl_network_input_dev(g->members[r]);
        return yl_network_forward(g->members[r], in);
    });
}

int yl_group_synchronize(yl_group *g)
{
    if (!g) { set_error("null argument"); return YL_ERR_ARG; }
    return for_all(g, [g](int r) { return yl_network_synchronize(g->members[r]); });
}

int yl_group_detect_batch(yl_group *g, const int *img_w, const int *img_h, float thresh, int relative, int letter,
                          float nms, int cap, float *records_dev_root, int *counts_dev_root)
{
    if (!g || !records_dev_root || !counts_dev_root || cap <= 0) { set_error("bad argument"); return YL_ERR_ARG; }
    if ((img_w == nullptr) != (img_h == nullptr)) { set_error("img_w and img_h must both be given or both be NULL"); return YL_ERR_ARG; }
    int rc = ensure_comms(g);
    if (rc != YL_OK) return rc;
    const int classes = g->members[0]->net.layers.back().classes;
    const size_t row = (size_t)(6 + classes);
    const size_t per_image = (size_t)cap * row;
    if (g->rec_floats != per_image) {                 // (re)size the per-rank staging
        for (int r = 0; r < g->n; ++r) {
            if (hipSetDevice(g->devices[r]) != hipSuccess) { set_error("hipSetDevice failed"); return YL_ERR_DEVICE; }
            (void)hipStreamSynchronize((hipStream_t)g->members[r]->net.stream);
            if (g->d_rec[r]) (void)hipFree(g->d_rec[r]);
            if (g->d_cnt[r]) (void)hipFree(g->d_cnt[r]);
            g->d_rec[r] = nullptr; g->d_cnt[r] = nullptr;
            if (hipMalloc((void **)&g->d_rec[r], sizeof(float) * per_image * g->count[r]) != hipSuccess ||
                hipMalloc((void **)&g->d_cnt[r], sizeof(int) * g->count[r]) != hipSuccess) {
                set_error("hipMalloc of the detection staging failed"); g->rec_floats = 0; return YL_ERR_DEVICE;
            }
        }
        g->rec_floats = per_image;
    }
    // decode + NMS of every shard on its own device and stream, in parallel
    rc = for_all(g, [=](int r) {
        return yl_network_detect_batch(g->members[r], img_w ? img_w + g->first[r] : nullptr, img_h ? img_h + g->first[r] : nullptr,
                                       thresh, relative, letter, nms, cap, g->d_rec[r], g->d_cnt[r]);
    });
    if (rc != YL_OK) return rc;
    // gather to the root device, stream-ordered behind the kernels above.  Single-process RCCL: all ranks' calls
    // inside ONE group, issued by this thread (the root's own shard travels as a send-to-self).
    Rccl &R = rccl();
    ncclResult_t e = R.GroupStart();
    for (int r = 0; r < g->n && e == ncclSuccess; ++r) {
        hipStream_t s = (hipStream_t)g->members[r]->net.stream;
        e = R.Send(g->d_rec[r], per_image * g->count[r], ncclFloat, 0, g->comms[r], s);
        if (e == ncclSuccess) e = R.Send(g->d_cnt[r], (size_t)g->count[r], ncclInt32, 0, g->comms[r], s);
    }
    hipStream_t s0 = (hipStream_t)g->members[0]->net.stream;
    for (int r = 0; r < g->n && e == ncclSuccess; ++r) {
        e = R.Recv(records_dev_root + per_image * g->first[r], per_image * g->count[r], ncclFloat, r, g->comms[0], s0);
        if (e == ncclSuccess) e = R.Recv(counts_dev_root + g->first[r], (size_t)g->count[r], ncclInt32, r, g->comms[0], s0);
    }
    const ncclResult_t e2 = R.GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) { set_error(std::string("RCCL gather: ") + R.GetErrorString(e)); return YL_ERR_DEVICE; }
    return YL_OK;
}

int yl_group_get_boxes_batch(yl_group *g, const int *img_w, const int *img_h, float thresh, int relative, int letter,
                             float nms, int cap, float *rows_host, int *counts_host)
{
    if (!g || !rows_host || !counts_host || cap <= 0) { set_error("bad argument"); return YL_ERR_ARG; }
    const int classes = g->members[0]->net.layers.back().classes;
    const size_t row = (size_t)(6 + classes);
    const size_t need = sizeof(float) * (size_t)g->global_batch * cap * row;
    if (hipSetDevice(g->devices[0]) != hipSuccess) { set_error("hipSetDevice failed"); return YL_ERR_DEVICE; }
    if (g->rec_root_bytes < need) {
        if (g->d_rec_root) (void)hipFree(g->d_rec_root);
        if (g->d_cnt_root) (void)hipFree(g->d_cnt_root);
        g->d_rec_root = nullptr; g->d_cnt_root = nullptr; g->rec_root_bytes = 0;
        if (hipMalloc((void **)&g->d_rec_root, need) != hipSuccess ||
            hipMalloc((void **)&g->d_cnt_root, sizeof(int) * (size_t)g->global_batch) != hipSuccess) {
            set_error("hipMalloc of the gathered detections failed"); return YL_ERR_DEVICE;
        }
        g->rec_root_bytes = need;
    }
    const int rc = yl_group_detect_batch(g, img_w, img_h, thresh, relative, letter, nms, cap, g->d_rec_root, g->d_cnt_root);
    if (rc != YL_OK) return rc;
    if (hipSetDevice(g->devices[0]) != hipSuccess) { set_error("hipSetDevice failed"); return YL_ERR_DEVICE; }
    hipStream_t s0 = (hipStream_t)g->members[0]->net.stream;
    if (hipStreamSynchronize(s0) != hipSuccess) { set_error("stream synchronize failed"); return YL_ERR_DEVICE; }
    // caller memory: through the pinned staging chunks (staging.hip), only the filled rows travel
    int rc2 = stage_d2h(g->devices[0], counts_host, g->d_cnt_root, sizeof(int) * (size_t)g->global_batch);
    for (int b = 0; b < g->global_batch && rc2 == YL_OK; ++b) {
        const int c = counts_host[b] < cap ? counts_host[b] : cap;
        if (c > 0) rc2 = stage_d2h(g->devices[0], rows_host + (size_t)b * cap * row, g->d_rec_root + (size_t)b * cap * row, sizeof(float) * row * c);
    }
    return rc2;
}

}  // extern "C"
