// Host arrays over PCIe without page-locking them: staged copies through a ring of page-locked slots owned by the context.
//
// The reference's boundary is numpy arrays in, numpy arrays out (gen_cal_image.py:480-739); a 4096 x 4096 x 8 ramp is 0.48 GB in
// and 0.40 GB out.  hipMemcpy from / to PAGEABLE memory stages through the runtime's own buffers with one thread's memcpy
// (measured 44 ms per ramp = 23 ramps/s, below the 50 ramps/s floor of the north star; page-locked arrays: 17.7 ms).  Here a
// pageable array goes in chunks of 16 MiB: a few worker threads copy chunk i+1 into a page-locked slot while the DMA engine moves
// chunk i, and the other way round for results -- the PCIe rate again, whatever memory the caller's arrays live in.  Page-locked
// arrays (rip_host_alloc, or registered by the caller) are recognised and copied directly.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "rip_common.h"

namespace {

constexpr size_t HC_CHUNK = (size_t)16 << 20;
constexpr int HC_SLOTS = 3;

// a handful of threads that copy pieces of one chunk side by side (a single thread's memcpy is slower than PCIe)
class CopyPool {
  public:
    explicit CopyPool(int nworkers) {
        for (int i = 0; i < nworkers; ++i) workers_.emplace_back([this, i] { run(i); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
            ++epoch_;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void copy(void *dst, const void *src, size_t n) {
        const int parts = (int)workers_.size() + 1;
        if (n < ((size_t)1 << 20) || parts == 1) {
            memcpy(dst, src, n);
            return;
        }
        const size_t piece = ((n + parts - 1) / parts + 4095) / 4096 * 4096;
        {
            std::lock_guard<std::mutex> lk(m_);
            dst_ = (char *)dst, src_ = (const char *)src, n_ = n, piece_ = piece;
            pending_ = (int)workers_.size();
            ++epoch_;
        }
        cv_.notify_all();
        const size_t o = piece * workers_.size();   // the caller's own piece: the last one
        if (o < n) memcpy((char *)dst + o, (const char *)src + o, n - o);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

  private:
    void run(int i) {
        unsigned long seen = 0;
        for (;;) {
            char *d;
            const char *s;
            size_t n, piece;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
                d = dst_, s = src_, n = n_, piece = piece_;
            }
            const size_t o = piece * (size_t)i;
            if (o < n) memcpy(d + o, s + o, (o + piece < n) ? piece : n - o);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    unsigned long epoch_ = 0;
    bool stop_ = false;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t n_ = 0, piece_ = 0;
    int pending_ = 0;
};

}   // namespace

struct RipHostCopy {
    CopyPool pool;
    char *slot[2][HC_SLOTS] = {};          // [0]: towards the device, [1]: from the device
    hipEvent_t ev[2][HC_SLOTS] = {};
    bool busy[2][HC_SLOTS] = {};
    explicit RipHostCopy(int nworkers) : pool(nworkers) {}
};

void rip_hostcopy_release_raw(RipHostCopy *h);

static int hc_get(rip_ctx *ctx, RipHostCopy **out) {
    if (!ctx->hostcopy) {
        int nw = 6;
        if (const char *e = getenv("ROMANHIP_COPY_THREADS")) nw = atoi(e) - 1;
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && nw > hw - 1) nw = hw - 1;
        if (nw < 0) nw = 0;
        RipHostCopy *h = new RipHostCopy(nw);
        for (int d = 0; d < 2; ++d)
            for (int s = 0; s < HC_SLOTS; ++s) {
                if (hipHostMalloc((void **)&h->slot[d][s], HC_CHUNK, hipHostMallocDefault) != hipSuccess ||
                    hipEventCreateWithFlags(&h->ev[d][s], hipEventDisableTiming) != hipSuccess) {
                    rip_hostcopy_release_raw(h);
                    return rip_fail(ctx, RIP_ENOMEM, "page-locked staging ring (%d x %zu MiB)", 2 * HC_SLOTS, HC_CHUNK >> 20);
                }
            }
        ctx->hostcopy = h;
    }
    *out = ctx->hostcopy;
    return RIP_OK;
}

void rip_hostcopy_release_raw(RipHostCopy *h) {
    if (!h) return;
    for (int d = 0; d < 2; ++d)
        for (int s = 0; s < HC_SLOTS; ++s) {
            if (h->ev[d][s]) (void)hipEventDestroy(h->ev[d][s]);
            if (h->slot[d][s]) (void)hipHostFree(h->slot[d][s]);
        }
    delete h;
}

void rip_hostcopy_release(rip_ctx *ctx) {
    rip_hostcopy_release_raw(ctx->hostcopy);
    ctx->hostcopy = nullptr;
}

// is `p` page-locked (hipHostMalloc / hipHostRegister) memory?
static bool hc_pinned(const void *p) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();   // an ordinary host pointer: not an error of ours
        return false;
    }
    return at.type == hipMemoryTypeHost;
}

// host -> device on `st`; returns when the host array has been read (the DMA of the last chunks may still run)
int rip_host_to_device(rip_ctx *ctx, void *dst, const void *src, size_t bytes, hipStream_t st) {
    if (!bytes) return RIP_OK;
    if (ctx->stage_pageable == 0 || bytes < HC_CHUNK / 4 || hc_pinned(src)) {
        RIP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
        return RIP_OK;
    }
    RipHostCopy *h;
    int rc = hc_get(ctx, &h);
    if (rc) return rc;
    int k = 0;
    for (size_t o = 0; o < bytes; o += HC_CHUNK, k = (k + 1) % HC_SLOTS) {
        const size_t n = bytes - o < HC_CHUNK ? bytes - o : HC_CHUNK;
        if (h->busy[0][k]) RIP_HIP(ctx, hipEventSynchronize(h->ev[0][k]));   // the DMA that last read this slot
        h->pool.copy(h->slot[0][k], (const char *)src + o, n);
        RIP_HIP(ctx, hipMemcpyAsync((char *)dst + o, h->slot[0][k], n, hipMemcpyHostToDevice, st));
        RIP_HIP(ctx, hipEventRecord(h->ev[0][k], st));
        h->busy[0][k] = true;
    }
    return RIP_OK;
}

// device -> host on `st`, n arrays: page-locked destinations are queued and left to the caller's synchronisation; pageable ones
// are complete on return -- their chunks run through ONE pipeline across the arrays (the DMA of the next chunks is in flight
// while the workers copy a landed chunk out)
int rip_device_to_host_many(rip_ctx *ctx, int n, void *const *dst, const void *const *src, const size_t *bytes, hipStream_t st) {
    struct Chunk {
        char *d;
        const char *s;
        size_t n;
    };
    std::vector<Chunk> ch;
    for (int i = 0; i < n; ++i) {
        if (!bytes[i] || !dst[i]) continue;
        if (ctx->stage_pageable == 0 || bytes[i] < HC_CHUNK / 4 || hc_pinned(dst[i])) {
            RIP_HIP(ctx, hipMemcpyAsync(dst[i], src[i], bytes[i], hipMemcpyDeviceToHost, st));
            continue;
        }
        for (size_t o = 0; o < bytes[i]; o += HC_CHUNK)
            ch.push_back({(char *)dst[i] + o, (const char *)src[i] + o, bytes[i] - o < HC_CHUNK ? bytes[i] - o : HC_CHUNK});
    }
    if (ch.empty()) return RIP_OK;
    RipHostCopy *h;
    int rc = hc_get(ctx, &h);
    if (rc) return rc;
    auto issue = [&](size_t c) -> int {
        const int k = (int)(c % HC_SLOTS);
        RIP_HIP(ctx, hipMemcpyAsync(h->slot[1][k], ch[c].s, ch[c].n, hipMemcpyDeviceToHost, st));
        RIP_HIP(ctx, hipEventRecord(h->ev[1][k], st));
        return RIP_OK;
    };
    for (size_t c = 0; c < ch.size() && c < (size_t)HC_SLOTS; ++c)
        if ((rc = issue(c))) return rc;
    for (size_t c = 0; c < ch.size(); ++c) {
        const int k = (int)(c % HC_SLOTS);
        RIP_HIP(ctx, hipEventSynchronize(h->ev[1][k]));
        h->pool.copy(ch[c].d, h->slot[1][k], ch[c].n);
        if (c + HC_SLOTS < ch.size() && (rc = issue(c + HC_SLOTS))) return rc;
    }
    return RIP_OK;
}
