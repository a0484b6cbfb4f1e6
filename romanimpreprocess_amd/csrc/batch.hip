// A batch of ramps handed over in HOST memory, pipelined over PCIe: while ramp i runs through the chain, ramp i+1 is
// uploaded and the results of ramp i-1 are downloaded (three streams, two sets of device buffers).  The reference's driver
// calls calibrateimage file by file (runs/summer2025run/OpenUniverse_to_L1L2.py:123-137: 18 SCAs x filters per exposure);
// this entry is what a batch driver hands its arrays to.  Each ramp goes through rip_calibrate's device path unchanged, so
// results are those of single calls.  With page-locked host arrays (rip_host_alloc) the copies run at PCIe rate in both
// directions at once (measured: 10.6 ms per 4096 x 4096 x 8 ramp with all outputs, 17.6 ms for single calls); pageable
// arrays work but serialise.  Streams map onto a few hardware queues (four by default), which is why the entry makes none of
// its own: with two extra streams the copies did not overlap at all.
#include "rip_common.h"

namespace {

struct BatchSet {
    char *in = nullptr, *out = nullptr;
    hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
    bool used = false;
};

size_t al256(size_t x) { return (x + 255) / 256 * 256; }

}   // namespace

extern "C" int rip_calibrate_batch(rip_ctx *ctx, int slot, int plan_id, unsigned stages, int n, const rip_ramp_desc *in,
                                   const rip_outputs *out) {
    if (n < 0 || (n > 0 && (!in || !out))) return rip_fail(ctx, RIP_EINVAL, "calibrate_batch: bad arguments");
    if (n == 0) return RIP_OK;
    if (!(stages & RIP_STAGE_RAMPFIT)) return rip_fail(ctx, RIP_EINVAL, "calibrate_batch: the stage mask must include the ramp fit");
    if (slot < 0 || slot >= (int)ctx->cals.size() || !ctx->cals[slot].valid)
        return rip_fail(ctx, RIP_EINVAL, "calibrate_batch: caldir slot %d is empty", slot);
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const RipCal &c = ctx->cals[slot];
    const int ny = c.ny, nx = c.nx, G = in[0].ngrp;
    const size_t npix = (size_t)ny * nx;
    for (int i = 0; i < n; ++i) {
        if (in[i].location != RIP_HOST || out[i].location != RIP_HOST)
            return rip_fail(ctx, RIP_EINVAL, "calibrate_batch: ramp %d is not in host memory (use rip_calibrate for device pointers)", i);
        if (in[i].ngrp != G || in[i].data_dtype != in[0].data_dtype)
            return rip_fail(ctx, RIP_EINVAL, "calibrate_batch: ramp %d differs in group count or dtype from ramp 0", i);
        if (!in[i].data || !in[i].pixeldq || !out[i].slope || !out[i].err_read || !out[i].err_poisson || !out[i].pixeldq)
            return rip_fail(ctx, RIP_EINVAL, "calibrate_batch: ramp %d lacks a required array", i);
        if (out[i].cube) return rip_fail(ctx, RIP_EINVAL, "calibrate_batch: the corrected cube is not returned by this entry");
    }
    if (G < 1 || G > RIP_MAX_GROUPS) return rip_fail(ctx, RIP_EINVAL, "calibrate_batch: %d groups unsupported", G);
    const size_t esz = in[0].data_dtype == RIP_U16 ? 2 : 4;
    const int nch = nx / RIP_CW;
    const size_t b_data = al256((size_t)G * npix * esz), b_a33 = al256((size_t)G * ny * RIP_CW * 2), b_gdq = al256((size_t)G * npix),
                 b_pdq = al256(npix * 4), b_area = al256(npix * 8), b_lines = al256((size_t)G * nch * 16), b_pl = al256(npix * 4);
    const size_t in_bytes = b_data + b_a33 + b_gdq + b_pdq + b_area + b_lines, out_bytes = 4 * b_pl + b_gdq;

    // Streams map onto a few hardware queues (four by default), so no stream is made here: the uploads ride on the context's
    // second stream, in front of the reference-pixel pre-pass of the same ramp; the downloads have the context's third one.
    if (!ctx->stream2 || !ctx->stream3) return rip_fail(ctx, RIP_EHIP, "calibrate_batch: the context has no copy streams");
    hipStream_t s_in = ctx->stream2, s_out = ctx->stream3;
    BatchSet set[2];
    int rc = RIP_OK;
    ctx->batch_completed = 0;  // ramps whose results have been queued for download in full (valid after an error return too)
    auto cleanup = [&]() {   // waits for every stream; the device buffers stay with the context
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(s_in);
        (void)hipStreamSynchronize(s_out);
        for (auto &b : set)
            for (hipEvent_t e : {b.ev_in, b.ev_done, b.ev_out})
                if (e) (void)hipEventDestroy(e);
    };
#define BATCH_HIP(call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            rc = rip_fail(ctx, RIP_EHIP, "%s: %s", #call, hipGetErrorString(e_));               \
            cleanup();                                                                          \
            return rc;                                                                          \
        }                                                                                       \
    } while (0)
    for (int k = 0; k < 2; ++k) {
        BatchSet &b = set[k];
        for (int io = 0; io < 2; ++io) {
            const int s = 2 * k + io;
            const size_t need = io ? out_bytes : in_bytes;
            if (ctx->batch_bytes[s] < need) {
                if (ctx->batch_buf[s]) (void)hipFree(ctx->batch_buf[s]);
                ctx->batch_buf[s] = nullptr;
                ctx->batch_bytes[s] = 0;
                BATCH_HIP(hipMalloc(&ctx->batch_buf[s], need));
                ctx->batch_bytes[s] = need;
            }
        }
        b.in = (char *)ctx->batch_buf[2 * k];
        b.out = (char *)ctx->batch_buf[2 * k + 1];
        BATCH_HIP(hipEventCreateWithFlags(&b.ev_in, hipEventDisableTiming));
        BATCH_HIP(hipEventCreateWithFlags(&b.ev_done, hipEventDisableTiming));
        BATCH_HIP(hipEventCreateWithFlags(&b.ev_out, hipEventDisableTiming));
    }
    for (int i = 0; i < n; ++i) {
        BatchSet &b = set[i & 1];
        const rip_ramp_desc &ri = in[i];
        const rip_outputs &ro = out[i];
        // upload: the input buffers of this set are free once the chain of ramp i-2 has run
        if (b.used) BATCH_HIP(hipStreamWaitEvent(s_in, b.ev_done, 0));
        rip_ramp_desc rd = ri;
        rd.location = RIP_DEVICE;
        size_t o = 0;
        auto put = [&](const void *src, size_t bytes, size_t slot_bytes) -> void * {
            void *dst = b.in + o;
            o += slot_bytes;
            if (!src) return nullptr;
            if (hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s_in) != hipSuccess) rc = RIP_EHIP;
            return dst;
        };
        rd.data = put(ri.data, (size_t)G * npix * esz, b_data);
        rd.amp33 = (const uint16_t *)put(ri.amp33, (size_t)G * ny * RIP_CW * 2, b_a33);
        rd.groupdq = (const uint8_t *)put(ri.groupdq, (size_t)G * npix, b_gdq);
        rd.pixeldq = (const uint32_t *)put(ri.pixeldq, npix * 4, b_pdq);
        rd.area_factor = (const double *)put(ri.area_factor, npix * 8, b_area);
        rd.channel_lines = (const double *)put(ri.channel_lines, (size_t)G * nch * 16, b_lines);
        if (rc == RIP_OK && ri.or_first_group && rd.groupdq) rc = rip_launch_or_bytes(ctx, (uint8_t *)rd.groupdq, npix, (uint8_t)DQ_DO_NOT_USE, s_in);
        rd.or_first_group = 0;
        if (rc != RIP_OK) {
            rc = rip_fail(ctx, RIP_EHIP, "calibrate_batch: upload of ramp %d failed", i);
            cleanup();
            return rc;
        }
        BATCH_HIP(hipEventRecord(b.ev_in, s_in));
        // chain: after its inputs have landed and the previous results of this set have left
        rd.ready_event = b.ev_in;   // the pre-pass stream and the main stream wait for the upload inside rip_calibrate
        if (b.used) BATCH_HIP(hipStreamWaitEvent(ctx->stream, b.ev_out, 0));
        rip_outputs od;
        od.location = RIP_DEVICE;
        od.slope = (float *)b.out;
        od.err_read = (float *)(b.out + b_pl);
        od.err_poisson = (float *)(b.out + 2 * b_pl);
        od.pixeldq = (uint32_t *)(b.out + 3 * b_pl);
        od.groupdq = ro.groupdq ? (uint8_t *)(b.out + 4 * b_pl) : nullptr;
        od.cube = nullptr;
        if ((rc = rip_calibrate(ctx, slot, plan_id, stages, &rd, &od)) != RIP_OK) {
            cleanup();
            return rc;
        }
        BATCH_HIP(hipEventRecord(b.ev_done, ctx->stream));
        // download
        BATCH_HIP(hipStreamWaitEvent(s_out, b.ev_done, 0));
        BATCH_HIP(hipMemcpyAsync(ro.slope, od.slope, npix * 4, hipMemcpyDeviceToHost, s_out));
        BATCH_HIP(hipMemcpyAsync(ro.err_read, od.err_read, npix * 4, hipMemcpyDeviceToHost, s_out));
        BATCH_HIP(hipMemcpyAsync(ro.err_poisson, od.err_poisson, npix * 4, hipMemcpyDeviceToHost, s_out));
        BATCH_HIP(hipMemcpyAsync(ro.pixeldq, od.pixeldq, npix * 4, hipMemcpyDeviceToHost, s_out));
        if (ro.groupdq) BATCH_HIP(hipMemcpyAsync(ro.groupdq, od.groupdq, (size_t)G * npix, hipMemcpyDeviceToHost, s_out));
        BATCH_HIP(hipEventRecord(b.ev_out, s_out));
        b.used = true;
        ctx->batch_completed = i + 1;
    }
#undef BATCH_HIP
    cleanup();   // waits for every stream
    return RIP_OK;
}

extern "C" int rip_calibrate_batch_completed(rip_ctx *ctx) { return ctx ? ctx->batch_completed : RIP_EINVAL; }
