// host_pool.cpp -- a small pool of host threads that moves bytes between caller memory and the library's pinned
// buffers (the bounce of staging.hip's rule: the HIP runtime never sees a pointer it did not allocate).
//
// network_predict's boundary hands over pageable host floats (284 MB in, 495 MB out for yolov3-608 at batch 64); one
// core moves ~8 GB/s, PCIe Gen5 ~50 GB/s, so the bounce has to be spread over cores AND overlapped with the DMA
// (runtime.hip: stage_input_h2d, pull_heads_overlapped).  The reference copies on the calling thread
// (cuda_push_array / cuda_pull_array, src/gpu.cu:236-266).
//
// One process-wide pool (threads are created once, detached, never joined: the HIP runtime and static destructors
// may already be gone at exit).  Jobs of different callers (two networks on two host threads, the group path's one
// thread per device) share the workers; a job's slices are counted on the job, and a waiting caller executes queued
// slices itself, so a wait never depends on a worker being free.
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>

#include "yl_internal.h"

namespace yl {

namespace {

struct Slice {
    char *dst;
    const char *src;
    size_t len;
    HostCopyJob *job;
};

struct Pool {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Slice> q;
    unsigned workers = 0;
};

Pool *g_pool = nullptr;
std::once_flag g_once;

void run(const Slice &s)
{
    memcpy(s.dst, s.src, s.len);
    s.job->pending.fetch_sub(1, std::memory_order_release);
}

void worker(Pool *p)
{
    for (;;) {
        Slice s;
        {
            std::unique_lock<std::mutex> lk(p->m);
            p->cv.wait(lk, [p] { return !p->q.empty(); });
            s = p->q.front();
            p->q.pop_front();
        }
        run(s);
    }
}

Pool *pool()
{
    std::call_once(g_once, [] {
        Pool *p = new Pool;
        unsigned hw = std::thread::hardware_concurrency();
        // a quarter of the cores, 1..16: memcpy saturates a socket's memory system long before its core count
        unsigned n = hw / 4;
        if (n < 1) n = 1;
        if (n > 16) n = 16;
        if (const char *e = getenv("YL_HOST_COPY_THREADS")) {
            int v = atoi(e);
            if (v >= 0 && v <= 64) n = (unsigned)v;
        }
        p->workers = n;
        for (unsigned i = 0; i < n; ++i) std::thread(worker, p).detach();
        g_pool = p;
    });
    return g_pool;
}

}  // namespace

unsigned host_copy_threads() { return pool()->workers; }

// queue dst <- src in slices of >= 256 KB; returns at once
void host_copy_async(HostCopyJob &job, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return;
    Pool *p = pool();
    const size_t MIN_SLICE = (size_t)256 << 10;
    size_t parts = p->workers + 1;
    if (bytes / parts < MIN_SLICE) parts = bytes / MIN_SLICE ? bytes / MIN_SLICE : 1;
    const size_t slice = ((bytes + parts - 1) / parts + 63) & ~(size_t)63;
    size_t n = 0;
    for (size_t o = 0; o < bytes; o += slice) ++n;
    job.pending.fetch_add((int)n, std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> lk(p->m);
        for (size_t o = 0; o < bytes; o += slice)
            p->q.push_back(Slice{(char *)dst + o, (const char *)src + o, bytes - o < slice ? bytes - o : slice, &job});
    }
    p->cv.notify_all();
}

// returns when every slice queued on `job` has been copied; the caller copies queued slices (of any job) meanwhile
void host_copy_wait(HostCopyJob &job)
{
    Pool *p = pool();
    while (job.pending.load(std::memory_order_acquire) > 0) {
        Slice s;
        bool have = false;
        {
            std::lock_guard<std::mutex> lk(p->m);
            if (!p->q.empty()) { s = p->q.front(); p->q.pop_front(); have = true; }
        }
        if (have) run(s);
        else std::this_thread::yield();
    }
}

}  // namespace yl
