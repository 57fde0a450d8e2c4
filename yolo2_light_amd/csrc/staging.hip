// staging.hip -- the ONLY way bytes travel between caller-owned host memory and the device.
//
// Round 2 handed pageable memory (std::vector temporaries of the weight packers, numpy arrays of the
// callers, the reference's calloc'd l.output via hipHostRegister) to hipMemcpy / hipMemcpyAsync.  Above a
// size threshold the HIP runtime locks such a range and KEEPS the lock object in a small per-queue cache keyed
// by the host address; when the allocator returns that range to the kernel (heap trim / munmap) and later hands
// the same address out for a shorter block, a later copy hits the cached lock and the copy engine walks a
// user-pointer mapping whose tail is gone: "Memory access fault by GPU node-N ... on address <host heap>".
// (Root-caused in round 3 with instrumented builds on fresh GPU boxes: DESIGN.md section 9, profiles/r3_repro_fault.txt.)
//
// Here every transfer is bounced through library-owned hipHostMalloc memory (two 8 MB chunks per device, the
// memcpy of chunk k+1 overlaps the DMA of chunk k), so the runtime never sees a pointer it did not allocate
// itself and never has to lock / unlock / cache anything.  The reference's device runtime has the same shape of
// entry points (cuda_push_array / cuda_pull_array, src/gpu.cu:236-266), on pageable memory.
//
// Concurrency: ONE stager (two bounce chunks, one copy stream, one mutex) per DEVICE.  Uploads / downloads of two
// networks on the same GPU are therefore serialised against each other; networks on different GPUs (the group path:
// one replica and one host thread per device) do not contend.  These are set-up and debug transfers (weight images,
// layer downloads): the per-step paths -- input staging, head block, detection rows -- use per-network pinned
// buffers and the network's own stream, not this stager.
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>

#include "yl_internal.h"

namespace yl {

namespace {

constexpr size_t CHUNK = (size_t)8 << 20;
constexpr int MAX_DEV = 64;

struct Stager {
    std::mutex m;
    char *buf[2] = {nullptr, nullptr};
    hipStream_t s = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ready = false;
};

Stager g_stagers[MAX_DEV];       // never destroyed: the HIP runtime may be gone before static destructors run

#define ST_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_error(std::string("staging: " #expr ": ") + hipGetErrorString(e_));    \
            return YL_ERR_DEVICE;                                                      \
        }                                                                              \
    } while (0)

int ensure(Stager &st)
{
    if (st.ready) return YL_OK;
    for (int k = 0; k < 2; ++k) {
        ST_HIP(hipHostMalloc((void **)&st.buf[k], CHUNK, hipHostMallocDefault));
        ST_HIP(hipEventCreateWithFlags(&st.ev[k], hipEventDisableTiming));
    }
    ST_HIP(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
    st.ready = true;
    return YL_OK;
}

}  // namespace

int stage_h2d(int device, void *dst_dev, const void *src_host, size_t bytes)
{
    if (bytes == 0) return YL_OK;
    if (device < 0 || device >= MAX_DEV || !dst_dev || !src_host) { set_error("staging: bad argument"); return YL_ERR_ARG; }
    Stager &st = g_stagers[device];
    std::lock_guard<std::mutex> lock(st.m);
    ST_HIP(hipSetDevice(device));
    int rc = ensure(st);
    if (rc != YL_OK) return rc;
    const char *src = static_cast<const char *>(src_host);
    char *dst = static_cast<char *>(dst_dev);
    int k = 0;
    for (size_t off = 0; off < bytes; off += CHUNK, k ^= 1) {
        const size_t len = bytes - off < CHUNK ? bytes - off : CHUNK;
        if (off >= 2 * CHUNK) ST_HIP(hipEventSynchronize(st.ev[k]));       // the DMA out of this chunk two rounds ago
        memcpy(st.buf[k], src + off, len);
        ST_HIP(hipMemcpyAsync(dst + off, st.buf[k], len, hipMemcpyHostToDevice, st.s));
        ST_HIP(hipEventRecord(st.ev[k], st.s));
    }
    ST_HIP(hipStreamSynchronize(st.s));
    return YL_OK;
}

// The producer of src_dev must have completed (the callers synchronise their stream first).
int stage_d2h(int device, void *dst_host, const void *src_dev, size_t bytes)
{
    if (bytes == 0) return YL_OK;
    if (device < 0 || device >= MAX_DEV || !dst_host || !src_dev) { set_error("staging: bad argument"); return YL_ERR_ARG; }
    Stager &st = g_stagers[device];
    std::lock_guard<std::mutex> lock(st.m);
    ST_HIP(hipSetDevice(device));
    int rc = ensure(st);
    if (rc != YL_OK) return rc;
    const char *src = static_cast<const char *>(src_dev);
    char *dst = static_cast<char *>(dst_host);
    const size_t nchunks = (bytes + CHUNK - 1) / CHUNK;
    // chunk c lands in buf[c & 1]; its DMA is issued one iteration before it is drained
    for (size_t c = 0; c <= nchunks; ++c) {
        if (c < nchunks) {
            const size_t off = c * CHUNK;
            const size_t len = bytes - off < CHUNK ? bytes - off : CHUNK;
            ST_HIP(hipMemcpyAsync(st.buf[c & 1], src + off, len, hipMemcpyDeviceToHost, st.s));
            ST_HIP(hipEventRecord(st.ev[c & 1], st.s));
        }
        if (c > 0) {
            const size_t off = (c - 1) * CHUNK;
            const size_t len = bytes - off < CHUNK ? bytes - off : CHUNK;
            ST_HIP(hipEventSynchronize(st.ev[(c - 1) & 1]));
            memcpy(dst + off, st.buf[(c - 1) & 1], len);
        }
    }
    return YL_OK;
}

}  // namespace yl
