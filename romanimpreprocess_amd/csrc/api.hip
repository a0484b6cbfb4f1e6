// C-ABI of libromanhip.so (include/romanhip.h): context, device-resident CALDIR, ramp-fit plans,
// the chain driver and the stage-level entry points.  Host code only; kernels live in the other
// translation units.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <cmath>

#include "rip_common.h"

static std::string g_create_error;

int rip_fail(rip_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx)
        ctx->err = buf;
    else
        g_create_error = buf;
    return code;
}

void *rip_ws(rip_ctx *ctx, int slot, size_t bytes) {
    if (ctx->ws_bytes[slot] >= bytes && ctx->ws[slot]) return ctx->ws[slot];
    if (ctx->ws[slot]) {
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);   // (the overlapped pre-pass uses workspaces too)
        (void)hipFree(ctx->ws[slot]);
        ctx->ws[slot] = nullptr;
        ctx->ws_bytes[slot] = 0;
    }
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        rip_fail(ctx, RIP_ENOMEM, "hipMalloc(%zu bytes, workspace %d): %s", bytes, slot, hipGetErrorString(e));
        return nullptr;
    }
    ctx->ws[slot] = p;
    ctx->ws_bytes[slot] = bytes;
    return p;
}

namespace {

struct DevBuf {  // scoped device allocation for the stage-level entry points
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(rip_ctx *ctx, size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
        if (e != hipSuccess) return rip_fail(ctx, RIP_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return RIP_OK;
    }
    int upload(rip_ctx *ctx, const void *src, size_t bytes) {
        int rc = alloc(ctx, bytes);
        if (rc) return rc;
        RIP_HIP(ctx, hipMemcpyAsync(p, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return RIP_OK;
    }
    template <typename T>
    T *as() {
        return reinterpret_cast<T *>(p);
    }
};

size_t dsize(int dtype) { return dtype == RIP_F64 ? 8 : (dtype == RIP_U16 ? 2 : 4); }

int dev_copy_in(rip_ctx *ctx, void **dst, const void *src, size_t bytes) {
    *dst = nullptr;
    if (!src) return RIP_OK;
    hipError_t e = hipMalloc(dst, bytes);
    if (e != hipSuccess) return rip_fail(ctx, RIP_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    RIP_HIP(ctx, hipMemcpyAsync(*dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return RIP_OK;
}

void free_cal(RipCal &c) {
    // everything else (linearity planes, gain if f32, read noise, dark rate, flat planes) lives in the slab
    void *ptrs[] = {c.dark_data, c.dark_slope, c.dark_dq, c.amp33_med, c.ipc, c.bias, c.slab, c.sat_thr, c.sat_dq,
                    c.gain_dtype == RIP_F64 ? c.gain : nullptr};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    c = RipCal();
}

}  // namespace

extern "C" {

#ifdef RIP_TIMING_BUILD
int rip_version(void) { return RIP_VERSION + RIP_TIMING_BUILD_FLAG; }
#else
int rip_version(void) { return RIP_VERSION; }
#endif

const char *rip_last_error(const rip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rip_ctx_create(int device_id, rip_ctx **out) {
    if (!out) return rip_fail(nullptr, RIP_EINVAL, "rip_ctx_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return rip_fail(nullptr, RIP_EHIP, "no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return rip_fail(nullptr, RIP_EINVAL, "device %d out of range [0,%d)", device_id, ndev);
    e = hipSetDevice(device_id);
    if (e != hipSuccess) return rip_fail(nullptr, RIP_EHIP, "hipSetDevice(%d): %s", device_id, hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return rip_fail(nullptr, RIP_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return rip_fail(nullptr, RIP_EHIP, "device %d is %s; this library is built for gfx950 (MI355X) only", device_id,
                        prop.gcnArchName);
    rip_ctx *ctx = new rip_ctx();
    ctx->device = device_id;
    ctx->ncu = prop.multiProcessorCount;
    // (queue priorities -- main stream above the second -- and a raised wave priority of the fused kernel were both tried against
    // the 4 % the overlapped pre-pass costs the fused kernel it runs beside: no effect, profiles/r04_summary.md)
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete ctx;
        return rip_fail(nullptr, RIP_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    if (hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking) != hipSuccess) ctx->stream2 = nullptr;
    if (hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) != hipSuccess) ctx->stream3 = nullptr;
    for (int i = 0; i < 2 && ctx->stream2; ++i)
        if (hipEventCreateWithFlags(&ctx->ev_tab[i], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ctx->ev_done[i], hipEventDisableTiming) != hipSuccess) {
            ctx->use_overlap = false;
        }
    if (!ctx->stream2) ctx->use_overlap = false;
    *out = ctx;
    return RIP_OK;
}

void rip_ctx_destroy(rip_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    for (auto &c : ctx->cals) free_cal(c);
    for (auto *p : ctx->plans)
        if (p) {
            if (p->dev) (void)hipFree(p->dev);
            delete p;
        }
    rip_pink_release(ctx);
    for (hipEvent_t e : {ctx->ev_tab[0], ctx->ev_tab[1], ctx->ev_done[0], ctx->ev_done[1], ctx->ev_in, ctx->ev_pre, ctx->ev_frames, ctx->ev_fill, ctx->ev_pink, ctx->ev_ahead})
        if (e) (void)hipEventDestroy(e);
    for (void *p : ctx->ws)   // every workspace slot, the Level-1 synthesis ones included
        if (p) (void)hipFree(p);
    for (void *p : ctx->batch_buf)
        if (p) (void)hipFree(p);
    if (ctx->chain_dbg_buf) (void)hipFree(ctx->chain_dbg_buf);
    if (ctx->prepass_stamps) (void)hipFree(ctx->prepass_stamps);
    if (ctx->stream3) {
        (void)hipStreamSynchronize(ctx->stream3);
        (void)hipStreamDestroy(ctx->stream3);
    }
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamDestroy(ctx->stream2);
    }
    delete ctx;
}

int rip_synchronize(rip_ctx *ctx) {
    if (ctx->stream2) RIP_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

void *rip_stream(rip_ctx *ctx) { return (void *)ctx->stream; }

// page-locked host memory for the caller's arrays: copies from / to it run at PCIe rate and asynchronously
void *rip_host_alloc(rip_ctx *ctx, size_t bytes) {
    void *p = nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess || hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)rip_fail(ctx, RIP_ENOMEM, "rip_host_alloc: %zu bytes of page-locked memory", bytes);
        return nullptr;
    }
    return p;
}
void rip_host_free(rip_ctx *ctx, void *p) {
    (void)ctx;
    if (p) (void)hipHostFree(p);
}

int rip_set_option_f64(rip_ctx *ctx, const char *name, double value) {
    if (name && strcmp(name, "guard_band") == 0) {
        if (!(value >= 0.0)) return rip_fail(ctx, RIP_EINVAL, "guard_band must be >= 0 (INFINITY = exact path everywhere)");
        ctx->guard_band = value;
        return RIP_OK;
    }
    return rip_fail(ctx, RIP_EINVAL, "unknown option %s", name ? name : "(null)");
}

int rip_set_option(rip_ctx *ctx, const char *name, int value) {
    if (name && strcmp(name, "fused") == 0) {
        ctx->use_fused = value != 0;
        return RIP_OK;
    }
    if (name && strcmp(name, "chain2") == 0) {
        ctx->use_chain2 = value != 0;
        return RIP_OK;
    }
    if (name && strcmp(name, "chain_quad") == 0) {
        ctx->chain_quad = value != 0;
        return RIP_OK;
    }
    if (name && strcmp(name, "chain_reserve") == 0) {
        ctx->chain_reserve = value < 0 ? 0 : value;
        return RIP_OK;
    }
    if (name && strcmp(name, "prepass_form") == 0) {
        if (value < -1 || value > 1) return rip_fail(ctx, RIP_EINVAL, "prepass_form: -1 (by situation), 0 or 1");
        ctx->prepass_form = value;
        return RIP_OK;
    }
    if (name && strcmp(name, "pink_form") == 0) {
        ctx->pink_form = value;
        return RIP_OK;
    }
    if (name && strcmp(name, "overlap") == 0) {
        if (value < -1 || value > 1) return rip_fail(ctx, RIP_EINVAL, "overlap: -1 (by situation), 0 or 1");
        ctx->use_overlap = value != 0 && ctx->stream2 != nullptr;
        ctx->overlap_mode = value;
        return RIP_OK;
    }
    if (name && strcmp(name, "chain_dbg") == 0) {
        ctx->chain_dbg = value;
        return RIP_OK;
    }
    return rip_fail(ctx, RIP_EINVAL, "unknown option %s", name ? name : "(null)");
}

// diagnostic builds (-DCH_STAMP): per-phase cycle sums of the fused kernel, summed over waves; clears the buffer
int rip_chain_stamps(rip_ctx *ctx, double out[9]) {
    const size_t n = 4096 * 9;
    if (!ctx->chain_dbg_buf) {
        RIP_HIP(ctx, hipMalloc((void **)&ctx->chain_dbg_buf, n * 8));
        RIP_HIP(ctx, hipMemset(ctx->chain_dbg_buf, 0, n * 8));
        for (int i = 0; i < 9; ++i) out[i] = 0;
        return RIP_OK;
    }
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h(n);
    RIP_HIP(ctx, hipMemcpy(h.data(), ctx->chain_dbg_buf, n * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 9; ++i) out[i] = 0;
    for (size_t k = 0; k < n; ++k) out[k % 9] += (double)h[k];
    RIP_HIP(ctx, hipMemset(ctx->chain_dbg_buf, 0, n * 8));
    return RIP_OK;
}

// same buffer, per wave of a workgroup of nw waves: out[w * 9 + i] (diagnostic builds; tools/gpu_checks/stamp_roles.py)
extern "C" int rip_chain_stamps_n(rip_ctx *ctx, int nw, double *out) {
    const size_t n = 4096 * 9;
    for (int i = 0; i < nw * 9; ++i) out[i] = 0;
    if (!ctx->chain_dbg_buf || nw < 1) return RIP_OK;
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h(n);
    RIP_HIP(ctx, hipMemcpy(h.data(), ctx->chain_dbg_buf, n * 8, hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; ++k) out[((k / 9) % (size_t)nw) * 9 + k % 9] += (double)h[k];
    RIP_HIP(ctx, hipMemset(ctx->chain_dbg_buf, 0, n * 8));
    return RIP_OK;
}

// diagnostic (tools/gpu_checks/prepass_stamps.py): clock stamps of the single-launch pre-pass, 16 per workgroup; the first call
// switches them on (out may be NULL), later calls copy the last launch's stamps out (nwg workgroups)
extern "C" int rip_prepass_stamps(rip_ctx *ctx, int nwg, unsigned long long *out) {
    const size_t bytes = (size_t)4096 * 16 * 8;
    if (!ctx->prepass_stamps) {
        RIP_HIP(ctx, hipMalloc(&ctx->prepass_stamps, bytes));
        RIP_HIP(ctx, hipMemset(ctx->prepass_stamps, 0, bytes));
    }
    if (out && nwg > 0 && nwg <= 4096) {
        RIP_HIP(ctx, rip_synchronize(ctx) ? hipErrorUnknown : hipSuccess);
        RIP_HIP(ctx, hipMemcpy(out, ctx->prepass_stamps, (size_t)nwg * 16 * 8, hipMemcpyDeviceToHost));
    }
    return RIP_OK;
}

int rip_last_chain_form(rip_ctx *ctx) { return ctx ? ctx->last_form : RIP_EINVAL; }

int rip_profile_enable(rip_ctx *ctx, int on) {
    ctx->prof = on != 0;
    return RIP_OK;
}

int rip_profile_read(rip_ctx *ctx, double out_ms[4], int *ncalls) {
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 4; ++i) out_ms[i] = 0.0;
    if (ctx->stream2) RIP_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    const size_t n = ctx->prof_events.size() / 6;  // per call: pre-pass begin/end, then 4 marks on the main stream
    for (size_t c = 0; c < n; ++c) {
        const hipEvent_t *e = &ctx->prof_events[c * 6];
        float ms = 0.f;
        RIP_HIP(ctx, hipEventElapsedTime(&ms, e[0], e[1]));
        out_ms[0] += ms;
        for (int i = 1; i < 4; ++i) {
            RIP_HIP(ctx, hipEventElapsedTime(&ms, e[1 + i], e[2 + i]));
            out_ms[i] += ms;
        }
    }
    for (hipEvent_t e : ctx->prof_events) (void)hipEventDestroy(e);
    ctx->prof_events.clear();
    if (ncalls) *ncalls = (int)n;
    return RIP_OK;
}

// --------------------------------------------------------------------------- CALDIR
int rip_caldir_drop(rip_ctx *ctx, int slot) {
    if (slot < 0 || slot >= (int)ctx->cals.size() || !ctx->cals[slot].valid)
        return rip_fail(ctx, RIP_EINVAL, "caldir slot %d is empty", slot);
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    free_cal(ctx->cals[slot]);
    return RIP_OK;
}

int rip_caldir_upload(rip_ctx *ctx, int slot, const rip_caldir_desc *d) {
    if (!d || slot < 0 || slot > 255) return rip_fail(ctx, RIP_EINVAL, "caldir upload: bad arguments");
    if (d->ny < 16 || d->nx < 16 || d->nborder < 0 || 2 * d->nborder + 3 > d->ny || 2 * d->nborder + 3 > d->nx)
        return rip_fail(ctx, RIP_EINVAL, "caldir upload: bad geometry %dx%d border %d", d->ny, d->nx, d->nborder);
    if (!d->gain || !d->read_noise) return rip_fail(ctx, RIP_EINVAL, "caldir upload: gain and read noise are required");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    if ((int)ctx->cals.size() <= slot) ctx->cals.resize(slot + 1);
    if (ctx->cals[slot].valid) {
        RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        free_cal(ctx->cals[slot]);
    }
    RipCal c;
    c.ny = d->ny;
    c.nx = d->nx;
    c.nb = d->nborder;
    const size_t npix = (size_t)c.ny * c.nx;
    const int nya = c.ny - 2 * c.nb, nxa = c.nx - 2 * c.nb;
    c.gain_dtype = d->gain_dtype;
    c.ipc_dtype = d->ipc_dtype;
    c.refout_slope = d->refout_slope;
    int rc;
#define UP(dst, src, bytes)                                         \
    if ((rc = dev_copy_in(ctx, (void **)&(dst), (src), (bytes)))) { \
        free_cal(c);                                                \
        return rc;                                                  \
    }
    if (d->lin_coefs && (!d->lin_smin || !d->lin_smax || !d->lin_sref || !d->lin_dq || d->lin_nplanes < 1))
        return rip_fail(ctx, RIP_EINVAL, "caldir upload: incomplete linearity arrays");
    const int NPl = d->lin_coefs ? d->lin_nplanes : 0;
    if (hipMalloc((void **)&c.slab, (size_t)(NPl + 12) * npix * 4) != hipSuccess)
        return rip_fail(ctx, RIP_ENOMEM, "caldir upload: %zu bytes for the per-pixel planes", (size_t)(NPl + 12) * npix * 4);
    RIP_HIP(ctx, hipMemsetAsync(c.slab, 0, (size_t)(NPl + 12) * npix * 4, ctx->stream));
    float *pl = c.slab;
    auto plane = [&](int k) { return pl + (size_t)(NPl + k) * npix; };
#define UPS(dst, src, bytes)                                                                           \
    if (src) {                                                                                         \
        hipError_t e_ = hipMemcpyAsync((void *)(dst), (src), (bytes), hipMemcpyHostToDevice, ctx->stream); \
        if (e_ != hipSuccess) {                                                                        \
            free_cal(c);                                                                               \
            return rip_fail(ctx, RIP_EHIP, "caldir upload: %s", hipGetErrorString(e_));                \
        }                                                                                              \
    }
    if (d->dark_data) {
        c.ngrp_dark = d->ngrp_dark;
        UP(c.dark_data, d->dark_data, npix * 4 * (size_t)d->ngrp_dark);
    }
    UP(c.dark_slope, d->dark_slope, npix * 4);
    UP(c.dark_dq, d->dark_dq, npix * 4);
    UP(c.sat_thr, d->saturation, npix * 4);
    UP(c.sat_dq, d->saturation_dq, npix * 4);
    c.read_noise = plane(5);
    UPS(c.read_noise, d->read_noise, npix * 4);
    UP(c.amp33_med, d->amp33_med, (size_t)c.ny * RIP_CW * 4);
    c.has_amp33 = d->amp33_med != nullptr;
    if (d->gain_dtype == RIP_F64) {
        UP(c.gain, d->gain, npix * 8);
    } else {
        c.gain = plane(4);
        UPS(c.gain, d->gain, npix * 4);
    }
    if (d->lin_coefs) {
        c.lin_nplanes = d->lin_nplanes;
        c.lin_coefs = pl;
        c.lin_smin = plane(0);
        c.lin_smax = plane(1);
        c.lin_sref = plane(2);
        c.lin_dq = (uint32_t *)plane(3);
        UPS(c.lin_coefs, d->lin_coefs, npix * 4 * (size_t)d->lin_nplanes);
        UPS(c.lin_smin, d->lin_smin, npix * 4);
        UPS(c.lin_smax, d->lin_smax, npix * 4);
        UPS(c.lin_sref, d->lin_sref, npix * 4);
        UPS(c.lin_dq, d->lin_dq, npix * 4);
    }
    // dark dq: only kept if any bit is set (every dark file the reference writes has dq == 0)
    if (d->dark_dq) {
        bool any = false;
        for (size_t i = 0; i < npix && !any; ++i) any = d->dark_dq[i] != 0;
        c.has_dark_dq = any;
    }
    // ipc4d (3,3,nya,nxa) -> (9,ny,nx), biascorr (g,nya,nxa) -> (g,ny,nx): zero border, aligned rows
    if (d->ipc4d) {
        const size_t es = dsize(d->ipc_dtype);
        void *tmp = rip_ws(ctx, 2, (size_t)9 * nya * nxa * es);
        hipError_t e = tmp ? hipMalloc(&c.ipc, 9 * npix * es) : hipErrorOutOfMemory;
        if (e != hipSuccess) {
            free_cal(c);
            return rip_fail(ctx, RIP_ENOMEM, "caldir upload: ipc4d allocation failed");
        }
        RIP_HIP(ctx, hipMemcpyAsync(tmp, d->ipc4d, (size_t)9 * nya * nxa * es, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = rip_launch_embed(ctx, tmp, c.ipc, 9, c.ny, c.nx, c.nb, (int)es))) {
            free_cal(c);
            return rc;
        }
        c.has_ipc = true;
    }
    if (d->biascorr) {
        c.ngrp_bias = d->ngrp_bias;
        const size_t nb_in = (size_t)d->ngrp_bias * nya * nxa * 4;
        void *tmp = rip_ws(ctx, 2, nb_in);
        hipError_t e = tmp ? hipMalloc((void **)&c.bias, (size_t)d->ngrp_bias * npix * 4) : hipErrorOutOfMemory;
        if (e != hipSuccess) {
            free_cal(c);
            return rip_fail(ctx, RIP_ENOMEM, "caldir upload: biascorr allocation failed");
        }
        RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // tmp may still feed the ipc embed
        RIP_HIP(ctx, hipMemcpyAsync(tmp, d->biascorr, nb_in, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = rip_launch_embed(ctx, tmp, c.bias, d->ngrp_bias, c.ny, c.nx, c.nb, 4))) {
            free_cal(c);
            return rc;
        }
        c.has_bias = true;
    }
    // IPC-deconvolved dark rate (gen_cal_image.py:217-221)
    if (c.dark_slope) {
        c.dark_rate = plane(6);
        if (c.has_ipc) {
            IpcArgs ia{c.dark_slope, c.dark_rate, c.ipc, c.gain, c.ipc_dtype, c.gain_dtype, c.ny, c.nx, c.nb, 1};
            if ((rc = rip_launch_ipc_cube(ctx, ia))) {
                free_cal(c);
                return rc;
            }
        } else {
            RIP_HIP(ctx, hipMemcpyAsync(c.dark_rate, c.dark_slope, npix * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    // flat in DN units (flatutils.get_flat with pdq given) + the flags it would OR into pdq
    if (d->flat) {
        DevBuf raw, padded, gclip;
        if ((rc = raw.upload(ctx, d->flat, npix * 4)) || (rc = padded.alloc(ctx, npix * 4)) ||
            (rc = gclip.alloc(ctx, npix * dsize(c.gain_dtype)))) {
            free_cal(c);
            return rc;
        }
        c.flat_dn = plane(7);
        c.flat_flags = (uint32_t *)plane(8);
        rc = rip_launch_flat_prepare(ctx, raw.as<float>(), c.gain, c.gain_dtype, c.ny, c.nx, c.nb, padded.as<float>(),
                                     gclip.p, c.flat_flags, c.has_ipc ? 1 : 0);
        if (!rc) {
            if (c.has_ipc) {
                IpcArgs ia{padded.as<float>(), c.flat_dn, c.ipc, gclip.p, c.ipc_dtype, c.gain_dtype, c.ny, c.nx, c.nb, 1};
                rc = rip_launch_ipc_cube(ctx, ia);
            } else {
                hipError_t e = hipMemcpyAsync(c.flat_dn, padded.p, npix * 4, hipMemcpyDeviceToDevice, ctx->stream);
                if (e != hipSuccess) rc = rip_fail(ctx, RIP_EHIP, "flat copy: %s", hipGetErrorString(e));
            }
        }
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (!rc && e != hipSuccess) rc = rip_fail(ctx, RIP_EHIP, "caldir upload: %s", hipGetErrorString(e));
        if (rc) {
            free_cal(c);
            return rc;
        }
        c.has_flat = true;
    }
    // the flag words of the wave-specialised fused kernel: linearity dq merged with the flat flags and / or the dark dq
    if (c.lin_dq) {
        DevBuf clash;
        if ((rc = clash.alloc(ctx, 16))) {
            free_cal(c);
            return rc;
        }
        uint32_t h_clash[3] = {0, 0, 0};
        RIP_HIP(ctx, hipMemsetAsync(clash.p, 0, 16, ctx->stream));
        for (int combo = 1; combo < 4 && !rc; ++combo) {
            const bool ff = (combo & 1) && c.has_flat, dd = (combo & 2) && c.has_dark_dq;
            if (((combo & 1) && !c.has_flat) || ((combo & 2) && !c.has_dark_dq)) continue;   // nothing to add: see below
            rc = rip_launch_merge_dq(ctx, c.lin_dq, ff ? c.flat_flags : nullptr, dd ? c.dark_dq : nullptr, (uint32_t *)plane(8 + combo),
                                     c.ny, c.nx, c.nb, clash.as<uint32_t>() + (combo - 1));
        }
        if (!rc) {
            hipError_t em = hipMemcpyAsync(h_clash, clash.p, 12, hipMemcpyDeviceToHost, ctx->stream);
            if (em == hipSuccess) em = hipStreamSynchronize(ctx->stream);
            if (em != hipSuccess) rc = rip_fail(ctx, RIP_EHIP, "caldir upload: %s", hipGetErrorString(em));
        }
        if (rc) {
            free_cal(c);
            return rc;
        }
        for (int combo = 1; combo < 4; ++combo) {
            const int eff = (c.has_flat ? (combo & 1) : 0) | (c.has_dark_dq ? (combo & 2) : 0);   // what this set can add at all
            c.merged_plane[combo] = eff == 0 ? 3 : (h_clash[eff - 1] ? -1 : 8 + eff);
        }
    }
#undef UP
#undef UPS
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        free_cal(c);
        return rip_fail(ctx, RIP_EHIP, "caldir upload: %s", hipGetErrorString(e));
    }
    c.valid = true;
    ctx->cals[slot] = c;
    return RIP_OK;
}

// --------------------------------------------------------------------------- plans
int rip_plan_create(rip_ctx *ctx, const rip_plan_desc *d, int *plan_id) {
    if (!d || !plan_id) return rip_fail(ctx, RIP_EINVAL, "plan: NULL argument");
    const int G = d->ngrp, start = d->exclude_first ? 1 : 0;
    if (G < 2 + start || G > RIP_MAX_GROUPS) return rip_fail(ctx, RIP_EINVAL, "plan: %d groups unsupported", G);
    const int nvar = 1 + (G - 3 - start > 0 ? G - 3 - start : 0);
    if (d->nvariants != nvar) return rip_fail(ctx, RIP_EINVAL, "plan: expected %d fit variants, got %d", nvar, d->nvariants);
    RipPlan *p = new RipPlan();
    RipPlanHeader &h = p->h;
    memset(&h, 0, sizeof h);
    h.ngrp = G;
    h.start = start;
    h.nvariants = nvar;
    h.do_not_flag_first = d->do_not_flag_first;
    h.sa = d->sthresh_a;
    h.dsb = d->sthresh_b - d->sthresh_a;
    h.loglen = std::log(d->ithresh_b / d->ithresh_a);
    h.ia = (float)d->ithresh_a;
    h.ib = (float)d->ithresh_b;
    for (int i = 0; i < G; ++i) {
        h.tbar[i] = d->tbar[i];
        h.tau[i] = d->tau[i];
        h.nreads[i] = (float)d->nreads[i];
    }
    for (int v = 0; v < nvar; ++v) {
        const int g = (v == 0) ? G : G - v;  // G, G-1, ..., 3+start  (fitting.py:326)
        if (d->variant_g[v] != g) {
            delete p;
            return rip_fail(ctx, RIP_EINVAL, "plan: variant %d covers %d groups, expected %d", v, d->variant_g[v], g);
        }
        RipVariant rv;
        rv.g = g;
        rv.coef = d->variant_coef[v];
        rv.rfac = d->variant_rfac[v];
        rv.k_ofs = (int)p->kvals.size();
        std::vector<float> K(g, 0.0f);
        if (v == 0) {
            for (int i = 0; i < g; ++i) K[i] = d->K[i];
        } else {  // fitting.py:165-169
            K[g - 1] = 1.0f / (d->tbar[g - 1] - d->tbar[start]);
            K[start] = -K[g - 1];
        }
        p->kvals.insert(p->kvals.end(), K.begin(), K.end());
        rv.diff_ofs = (int)p->diffs.size();
        rv.ndiff = 0;
        for (int i = start; i < g - 1; ++i) {  // fitting.py:225-229
            const int dimax = (i == g - 2 || g - 1 - start == 2) ? 1 : 2;
            for (int di = 1; di <= dimax; ++di) {
                RipDiff df;
                df.i = i;
                df.j = i + di;
                df.dt = d->tbar[i + di] - d->tbar[i];
                const float inv = 1.0f / df.dt;
                // fast-path variance coefficients: var = A*read^2 + B*dvardt, sums in f64
                double A = 0.0, B = 0.0, Babs = 0.0;
                for (int a = 0; a < g; ++a) {
                    const double wa = ((a == df.j) ? (double)inv : (a == df.i) ? (double)(-inv) : 0.0) - (double)K[a];
                    A += wa * wa / (double)d->nreads[a];
                    B += wa * wa * (double)d->tau[a];
                    Babs += wa * wa * (double)d->tau[a];
                    for (int b = 0; b < a; ++b) {
                        const double wb = ((b == df.j) ? (double)inv : (b == df.i) ? (double)(-inv) : 0.0) - (double)K[b];
                        B += 2.0 * wa * wb * (double)d->tbar[b];
                        Babs += std::fabs(2.0 * wa * wb) * (double)d->tbar[b];
                    }
                }
                // the reference rounds each term of the variance in f32/f64 as it goes; with cancellation between
                // the terms of B the relative error of any evaluation order is amplified by Babs/B
                const double amp = (B > 0.0) ? Babs / B : 1.0;
                df.relerr = (float)(2.5e-7 * (1.0 + amp) + 5e-7);
                df.A = (float)A;
                df.B = (float)B;
                df.inv_dt = inv;
                p->diffs.push_back(df);
                rv.ndiff++;
            }
        }
        p->variants.push_back(rv);
    }
    // device image: header | variants | K | diffs (each section 16-byte aligned)
    auto al = [](size_t x) { return (x + 15) / 16 * 16; };
    const size_t o_var = al(sizeof(RipPlanHeader));
    const size_t o_k = o_var + al(p->variants.size() * sizeof(RipVariant));
    const size_t o_d = o_k + al(p->kvals.size() * sizeof(float));
    const size_t o_dense = o_d + al(p->diffs.size() * sizeof(RipDiff));
    p->bytes = o_dense + al(sizeof(RipDense));
    {  // dense view of variant 0 (the full ramp) for the register-resident fit
        RipDense &dn = p->dense;
        memset(&dn, 0, sizeof dn);
        for (int i = 0; i < G; ++i) dn.K2[i] = d->K[i];
        const RipVariant &v0 = p->variants[0];
        double amin = 1e301;
        for (int k = 0; k < v0.ndiff; ++k) {
            const RipDiff &df = p->diffs[v0.diff_ofs + k];
            const int i = df.i, di = df.j - df.i, ps = 2 * (i / 2) + (di - 1), e = i & 1;
            dn.valid |= 1u << (2 * ps + e);
            dn.kidx[2 * ps + e] = k;
            dn.pairs[ps].inv_dt[e] = df.inv_dt;
            dn.pairs[ps].A[e] = df.A;
            dn.pairs[ps].B[e] = df.B;
            // acceptance factors of the packed fast path (device_rampfit.h, fit_full_pk): r = relerr + 4.1e-7 covers
            // the variance approximation and the part of the difference's rounding that scales with the significance
            const double r = (double)df.relerr + 4.1e-7;
            if (r < 9.9e-3) {
                dn.pairs[ps].k1[e] = std::nextafter((float)((1.0 / (1.0 - r)) * (1.0 + 4e-7)), INFINITY);
            } else {  // never accepted: the exact path decides
                dn.pairs[ps].k1[e] = INFINITY;
            }
            amin = (df.B >= 0.0f) ? std::fmin(amin, (double)df.A) : 0.0;
        }
        dn.amin = (v0.ndiff > 0 && amin > 0.0 && amin < 1e300) ? (float)(amin * (1.0 - 1e-6)) : 0.0f;
        for (int ps = 0; ps < RIP_MAX_GROUPS; ++ps)
            for (int e = 0; e < 2; ++e) {
                const int bit = 2 * ps + e;
                const bool used = bit < 32 && ((dn.valid >> bit) & 1u);
                if (!used) dn.pairs[ps].A[e] = 1.0f;  // keeps the approximate variance positive for unused slots
            }
    }
    std::vector<char> img(p->bytes, 0);
    memcpy(img.data(), &h, sizeof h);
    memcpy(img.data() + o_var, p->variants.data(), p->variants.size() * sizeof(RipVariant));
    memcpy(img.data() + o_k, p->kvals.data(), p->kvals.size() * sizeof(float));
    if (!p->diffs.empty()) memcpy(img.data() + o_d, p->diffs.data(), p->diffs.size() * sizeof(RipDiff));
    memcpy(img.data() + o_dense, &p->dense, sizeof(RipDense));
    hipError_t e = hipMalloc(&p->dev, p->bytes);
    if (e == hipSuccess) e = hipMemcpy(p->dev, img.data(), p->bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (p->dev) (void)hipFree(p->dev);
        delete p;
        return rip_fail(ctx, RIP_EHIP, "plan upload: %s", hipGetErrorString(e));
    }
    p->d_variants = reinterpret_cast<const RipVariant *>((char *)p->dev + o_var);
    p->d_k = reinterpret_cast<const float *>((char *)p->dev + o_k);
    p->d_diffs = reinterpret_cast<const RipDiff *>((char *)p->dev + o_d);
    p->d_dense = reinterpret_cast<const RipDense *>((char *)p->dev + o_dense);
    int id = -1;
    for (size_t i = 0; i < ctx->plans.size(); ++i)
        if (!ctx->plans[i]) {
            id = (int)i;
            break;
        }
    if (id < 0) {
        ctx->plans.push_back(nullptr);
        id = (int)ctx->plans.size() - 1;
    }
    ctx->plans[id] = p;
    *plan_id = id;
    return RIP_OK;
}

int rip_plan_destroy(rip_ctx *ctx, int id) {
    if (id < 0 || id >= (int)ctx->plans.size() || !ctx->plans[id]) return rip_fail(ctx, RIP_EINVAL, "plan %d does not exist", id);
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->plans[id]->dev);
    delete ctx->plans[id];
    ctx->plans[id] = nullptr;
    return RIP_OK;
}

static RipPlan *get_plan(rip_ctx *ctx, int id) {
    if (id < 0 || id >= (int)ctx->plans.size() || !ctx->plans[id]) {
        rip_fail(ctx, RIP_EINVAL, "plan %d does not exist", id);
        return nullptr;
    }
    return ctx->plans[id];
}

// --------------------------------------------------------------------------- the chain
int rip_calibrate(rip_ctx *ctx, int slot, int plan_id, unsigned stages, const rip_ramp_desc *in, const rip_outputs *out) {
    if (!in || !out) return rip_fail(ctx, RIP_EINVAL, "calibrate: NULL argument");
    if (slot < 0 || slot >= (int)ctx->cals.size() || !ctx->cals[slot].valid)
        return rip_fail(ctx, RIP_EINVAL, "calibrate: caldir slot %d is empty", slot);
    if (in->location != out->location) return rip_fail(ctx, RIP_EINVAL, "calibrate: inputs and outputs must share a location");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const RipCal &c = ctx->cals[slot];
    const int G = in->ngrp, ny = c.ny, nx = c.nx;
    const size_t npix = (size_t)ny * nx;
    if (G < 1 || G > RIP_MAX_GROUPS) return rip_fail(ctx, RIP_EINVAL, "calibrate: %d groups unsupported", G);
    if (!in->data || (in->data_dtype != RIP_U16 && in->data_dtype != RIP_F32))
        return rip_fail(ctx, RIP_EINVAL, "calibrate: data must be u16 or f32");
    RipPlan *plan = nullptr;
    const bool do_fit = stages & RIP_STAGE_RAMPFIT;
    if (do_fit || (stages & RIP_STAGE_LIN)) {
        plan = get_plan(ctx, plan_id);
        if (!plan) return RIP_EINVAL;
        if (plan->h.ngrp != G) return rip_fail(ctx, RIP_EINVAL, "calibrate: ramp has %d groups, plan %d", G, plan->h.ngrp);
    }
    const bool do_ref = stages & RIP_STAGE_REFPIX, do_bias = (stages & RIP_STAGE_BIAS) && c.has_bias;
    const bool do_lin = stages & RIP_STAGE_LIN, do_ipc = (stages & RIP_STAGE_IPC) && c.has_ipc;
    if (do_ref && (!c.dark_data || c.ngrp_dark < G)) return rip_fail(ctx, RIP_EINVAL, "calibrate: dark.data has %d groups, ramp %d", c.ngrp_dark, G);
    if (do_bias && c.ngrp_bias < G) return rip_fail(ctx, RIP_EINVAL, "calibrate: biascorr has %d groups, ramp %d", c.ngrp_bias, G);
    if (do_lin && !c.lin_coefs) return rip_fail(ctx, RIP_EINVAL, "calibrate: no linearity arrays in caldir slot %d", slot);
    const bool do_sat = in->flag_saturation != 0;
    if (do_sat && !c.sat_thr) return rip_fail(ctx, RIP_EINVAL, "calibrate: flag_saturation needs the saturation array in caldir slot %d", slot);
    if ((do_fit || do_lin) && ((!in->groupdq && !do_sat) || !in->pixeldq))
        return rip_fail(ctx, RIP_EINVAL, "calibrate: groupdq/pixeldq required");
    if (do_fit && (!out->slope || !out->err_read || !out->err_poisson || !out->pixeldq))
        return rip_fail(ctx, RIP_EINVAL, "calibrate: output planes required");
    if ((stages & RIP_STAGE_DARK) && !c.dark_rate) return rip_fail(ctx, RIP_EINVAL, "calibrate: no dark_slope in caldir");

    // ---- inputs on the device
    const bool host = in->location == RIP_HOST;
    const size_t b_data = (size_t)G * npix * dsize(in->data_dtype), b_a33 = (size_t)G * ny * RIP_CW * 2;
    const size_t b_gdq = (size_t)G * npix, b_pdq = npix * 4, b_area = npix * 8;
    const void *d_data = in->data;
    const uint16_t *d_a33 = in->amp33;
    const uint8_t *d_gdq = in->groupdq;
    const uint32_t *d_pdq = in->pixeldq;
    const double *d_area = in->area_factor;
    const double *d_lines_ovr = in->channel_lines;
    const int nch = nx / RIP_CW;
    if (host) {
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        size_t tot = al(b_data) + al(b_a33) + al(b_gdq) + al(b_pdq) + al(b_area) + al((size_t)G * nch * 16);
        char *w = (char *)rip_ws(ctx, 2, tot);
        if (!w) return RIP_ENOMEM;
        size_t o = 0;
        int rc_up = RIP_OK;
        auto put = [&](const void *src, size_t bytes) -> const void * {
            if (!src) return nullptr;
            void *dst = w + o;
            o += al(bytes);
            // (pageable arrays too: the runtime's own staging runs at the page-locked rate -- 16.7 against 16.5 ms per 4096 x 4096 x 8
            // ramp; a ring of page-locked slots fed by copy threads was measured SLOWER, 18.0 ms: profiles/r04_summary.md)
            if (!rc_up && hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc_up = RIP_EHIP;
            return dst;
        };
        d_data = put(in->data, b_data);
        d_a33 = (const uint16_t *)put(in->amp33, b_a33);
        d_gdq = (const uint8_t *)put(in->groupdq, b_gdq);
        d_pdq = (const uint32_t *)put(in->pixeldq, b_pdq);
        d_area = (const double *)put(in->area_factor, b_area);
        d_lines_ovr = (const double *)put(in->channel_lines, (size_t)G * nch * 16);
        if (rc_up) return rip_fail(ctx, RIP_EHIP, "calibrate: upload of the ramp failed: %s", hipGetErrorString(hipGetLastError()));
        RIP_HIP(ctx, hipGetLastError());
        // gen_cal_image.py:142-143 (rdq[0] |= DO_NOT_USE with EXCLUDE_FIRST) on the device copy, so that the host need not copy a
        // 134 MB array to set one plane's bit
        if (in->or_first_group && d_gdq) {
            const int rco = rip_launch_or_bytes(ctx, (uint8_t *)d_gdq, npix, (uint8_t)DQ_DO_NOT_USE);
            if (rco) return rco;
        }
    }
    // ---- outputs on the device
    float *o_slope = out->slope, *o_er = out->err_read, *o_ep = out->err_poisson, *o_cube = out->cube;
    uint32_t *o_pdq = out->pixeldq;
    uint8_t *o_gdq = out->groupdq;
    if (host) {
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        size_t tot = 4 * al(npix * 4) + (out->groupdq ? al(b_gdq) : 0);
        char *w = (char *)rip_ws(ctx, 7, tot);
        if (!w) return RIP_ENOMEM;
        o_slope = (float *)w;
        o_er = (float *)(w + al(npix * 4));
        o_ep = (float *)(w + 2 * al(npix * 4));
        o_pdq = (uint32_t *)(w + 3 * al(npix * 4));
        o_gdq = out->groupdq ? (uint8_t *)(w + 4 * al(npix * 4)) : nullptr;
        o_cube = nullptr;  // taken from the workspace cube below
    }

    // inputs guarded by a caller's event: the kernels on the main stream read them as well
    if (!host && in->ready_event) RIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)in->ready_event, 0));
    int rc;
    auto mark = [&]() {
        if (!ctx->prof) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, ctx->stream);
        ctx->prof_events.push_back(e);
    };
    // ---- reference-pixel tables.  With device-resident inputs the pre-pass runs on a second stream so that it
    // overlaps the previous ramp's main kernel; the tables are double-buffered by call parity:
    //   stream2: wait(main kernels of call n-2 done) -> pre-pass -> ev_tab[p]
    //   stream : wait(ev_tab[p]) -> main kernels -> ev_done[p]
    const size_t tab_bytes = ((size_t)2 * G * ny * 8 + (size_t)G * nch * 16 + 255) / 256 * 256;  // rowcorr, its transpose, lines
    char *ws3 = (do_ref || (do_fit && (stages & RIP_STAGE_FLAT) && c.has_flat && d_area))
                    ? (char *)rip_ws(ctx, 3, 2 * tab_bytes + npix * 4 + 512)
                    : nullptr;
    const int par = ctx->parity;
    // (by situation: the f64-ipc4d form of up to 8 groups fills the 160 KB of every CU with its partial K ring -- the pre-pass of the
    // next ramp finds no room beside it, runs when it drains, and the single-launch form in front of the own ramp is the shorter way:
    // 1.121 against 1.140 ms per ramp, profiles/r04_summary.md)
    const bool lds_full = c.ipc_dtype == RIP_F64 && G <= 8 && ctx->use_fused && in->data_dtype == RIP_U16;
    const bool overlap = do_ref && !host && ctx->use_overlap && (ctx->overlap_mode == 1 || !lds_full);
    hipStream_t pre = overlap ? ctx->stream2 : ctx->stream;   // where the pre-pass and the saturation pass of THIS call are launched
    double *rowcorr = nullptr, *rowcorr_t = nullptr, *lines = nullptr;
    // dq-init + saturation flagging into workspace copies of the flag arrays (the caller's inputs stay untouched), double
    // buffered by call parity like the reference-pixel tables because the pass may run ahead on the second stream
    auto sat_pass = [&]() -> int {
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        const size_t one = al(b_gdq) + al(b_pdq);
        char *w = (char *)rip_ws(ctx, 8, 2 * one);
        if (!w) return RIP_ENOMEM;
        uint8_t *g2 = (uint8_t *)(w + (size_t)par * one);
        uint32_t *p2 = (uint32_t *)(w + (size_t)par * one + al(b_gdq));
        const int dnu_first = (plan && plan->h.start == 1) ? 1 : 0;  // the plan excludes the first group
        const int rcs = rip_launch_satflag(ctx, d_data, in->data_dtype, c.sat_thr, c.sat_dq, d_gdq, d_pdq, g2, p2, G, ny, nx,
                                           in->sat_backup, in->sat_skip_firstn, dnu_first, in->sat_dilution, pre);
        d_gdq = g2;
        d_pdq = p2;
        return rcs;
    };
    // pre-passes of consecutive calls share workspaces: one that runs on another stream than its predecessor waits for it
    auto pre_order = [&]() -> int {
        if (ctx->ev_pre_valid && ctx->pre_stream != pre) RIP_HIP(ctx, hipStreamWaitEvent(pre, ctx->ev_pre, 0));
        return RIP_OK;
    };
    auto pre_done = [&]() -> int {
        if (!ctx->ev_pre) RIP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pre, hipEventDisableTiming));
        RIP_HIP(ctx, hipEventRecord(ctx->ev_pre, pre));
        ctx->pre_stream = pre;
        ctx->ev_pre_valid = true;
        return RIP_OK;
    };
    auto mark_on = [&](hipStream_t st) {
        if (!ctx->prof) return;
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return;
        (void)hipEventRecord(e, st);
        ctx->prof_events.push_back(e);
    };
    if (do_ref) {
        if (nx % RIP_CW) return rip_fail(ctx, RIP_EINVAL, "calibrate: nx=%d is not a multiple of 128", nx);
        if (!ws3) return RIP_ENOMEM;
        if (c.has_amp33 && !d_a33) return rip_fail(ctx, RIP_EINVAL, "calibrate: the read file has amp33 but the ramp has none");
        rowcorr = (double *)(ws3 + (size_t)par * tab_bytes);
        rowcorr_t = rowcorr + (size_t)G * ny;
        lines = rowcorr_t + (size_t)G * ny;
        RefpixArgs ra{d_data, in->data_dtype, c.dark_data, c.has_amp33 ? d_a33 : nullptr, c.amp33_med, c.refout_slope,
                      d_lines_ovr, rowcorr, rowcorr_t, lines, ny, nx, G, overlap ? 1 : 0, pre};
        if (overlap) {
            if (ctx->ev_done_valid[par]) RIP_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_done[par], 0));
            // what the pre-pass stream waits for before it reads the inputs (rip_ramp_desc::inputs_ready / ready_event): the
            // caller's event, and -- unless the caller vouches for complete inputs -- everything queued on the main stream so far
            // (work of the caller's own, or of this library's device-pointer entry points: stream_dirty, which an event that
            // guards only some of the inputs does not cover)
            if (in->ready_event) RIP_HIP(ctx, hipStreamWaitEvent(ctx->stream2, (hipEvent_t)in->ready_event, 0));
            if ((!in->ready_event && in->inputs_ready != RIP_INPUTS_COMPLETE) || ctx->stream_dirty) {
                if (!ctx->ev_in) RIP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming));
                RIP_HIP(ctx, hipEventRecord(ctx->ev_in, ctx->stream));
                RIP_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ctx->ev_in, 0));
            }
        }
        mark_on(pre);
        if ((rc = pre_order()) || (rc = rip_launch_refpix_prepass(ctx, ra))) return rc;
        if (do_sat && (rc = sat_pass())) return rc;  // same stream as the pre-pass: overlaps the previous ramp's main kernel
        if ((rc = pre_done())) return rc;
        mark_on(pre);
        if (overlap) {
            RIP_HIP(ctx, hipEventRecord(ctx->ev_tab[par], ctx->stream2));
            RIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_tab[par], 0));
        }
    } else {
        mark();
        if (do_sat && ((rc = pre_order()) || (rc = sat_pass()) || (rc = pre_done()))) return rc;
        mark();
    }
    mark();
    // flat plane the slope is divided by: f32(flat_dn / AreaFactor)  (gen_cal_image.py:622)
    const float *flat_plane = nullptr;
    if (do_fit && (stages & RIP_STAGE_FLAT) && c.has_flat) {
        flat_plane = c.flat_dn;
        if (d_area) {
            if (!ws3) return RIP_ENOMEM;
            float *fl = (float *)(ws3 + 2 * tab_bytes);
            if ((rc = rip_launch_flat_area(ctx, c.flat_dn, d_area, fl, npix))) return rc;
            flat_plane = fl;
        }
    }
    const float *cur = nullptr;
    const uint32_t *pdq_mid = d_pdq;
    // the fused kernel covers the complete chain on a Level-1 (u16) cube; sub-chains and f32 cubes take the
    // stage-by-stage kernels
    ctx->last_form = 0;
    const bool fused = ctx->use_fused && do_ref && do_bias && do_lin && do_ipc && do_fit && in->data_dtype == RIP_U16 &&
                       rip_chain_supported(ctx, c.lin_nplanes, G, c.ipc_dtype, c.gain_dtype);
    bool ran_fused = false;
    if (fused) {
        // ---- one kernel: refpix apply + bias + linearity + IPC + ramp fit + finish (chain.hip)
        ChainArgs ca;
        memset(&ca, 0, sizeof ca);
        ca.data = d_data;
        ca.data_u16 = in->data_dtype == RIP_U16;
        ca.gdq = d_gdq;
        ca.pdq = d_pdq;
        ca.dark_data = c.dark_data;
        ca.rowcorr = rowcorr;
        ca.rowcorr_t = rowcorr_t;
        ca.lines = lines;
        ca.bias = c.bias + (size_t)(c.ngrp_bias - G) * npix;
        ca.planes = c.slab;
        ca.do_not_flag_first = plan->h.do_not_flag_first;
        ca.kern = c.ipc;
        ca.finish = (stages & (RIP_STAGE_DARK | RIP_STAGE_FLAT)) ? 1 : 0;
        if (stages & RIP_STAGE_DARK) {
            ca.dark_rate = 1;
            ca.dark_dq = c.has_dark_dq ? c.dark_dq : nullptr;
        }
        ca.flat = flat_plane;
        ca.slope = o_slope;
        ca.err_read = o_er;
        ca.err_poisson = o_ep;
        ca.pdq_out = o_pdq;
        ca.gdq_out = o_gdq;
        if (out->cube) {
            float *cb = host ? (float *)rip_ws(ctx, 1, (size_t)G * npix * 4) : out->cube;
            if (!cb) return RIP_ENOMEM;
            ca.cube_out = cb;
            cur = cb;
        }
        ca.ny = ny;
        ca.nx = nx;
        ca.nb = c.nb;
        ca.ngrp = G;
        ca.dense = plan->d_dense;
        // the flag word that holds what this call's finish step ORs into pixeldq (flat flags with the flat stage, dark dq with
        // the dark stage): -1 = not mergeable for this CALDIR set, the wave-specialised kernel is then not taken
        ca.merged_dq = c.merged_plane[((flat_plane ? 1 : 0) | ((stages & RIP_STAGE_DARK) ? 2 : 0))];
        ca.dbg = ctx->chain_dbg;
        ca.dbg_buf = ctx->chain_dbg_buf;
        rc = rip_launch_chain(ctx, plan, ca, c.lin_nplanes, c.ipc_dtype);
        if (rc == RIP_OK) {
            ran_fused = true;
            mark();
            mark();
            mark();
        } else if (rc != 1) {
            return rc;
        } else {   // 1: no fused kernel for this plan / CALDIR set (flag words not mergeable, unusual difference mask): stage kernels
            cur = nullptr;
            rc = RIP_OK;
        }
    }
    if (!ran_fused) {
    // ---- cube stage: refpix apply + bias + linearity (or a plain conversion to f32)
    const bool need_cube_stage = do_ref || do_bias || do_lin || in->data_dtype != RIP_F32;
    if (need_cube_stage) {
        float *cubeA = (float *)rip_ws(ctx, 0, (size_t)G * npix * 4 + npix * 4);
        if (!cubeA) return RIP_ENOMEM;
        uint32_t *pdq_ws = (uint32_t *)(cubeA + (size_t)G * npix);
        LinArgs la;
        memset(&la, 0, sizeof la);
        la.data = d_data;
        la.data_dtype = in->data_dtype;
        la.phi = cubeA;
        la.gdq = d_gdq;
        la.gdq_is_attempt = 0;
        la.pdq_in = d_pdq;
        la.pdq_out = do_lin ? pdq_ws : nullptr;
        if (do_ref) {
            la.dark_data = c.dark_data;
            la.rowcorr = rowcorr;
            la.lines = lines;
        }
        if (do_bias) la.bias = c.bias + (size_t)(c.ngrp_bias - G) * npix;  // biascorr[de:], gen_cal_image.py:561-562
        if (do_lin) {
            la.coefs = c.lin_coefs;
            la.smin = c.lin_smin;
            la.smax = c.lin_smax;
            la.sref = c.lin_sref;
            la.lin_dq = c.lin_dq;
            la.nplanes = c.lin_nplanes;
            la.do_not_flag_first = plan->h.do_not_flag_first;
        }
        la.ny = ny;
        la.nx = nx;
        la.nb = c.nb;
        la.ngrp = G;
        if ((rc = rip_launch_lin(ctx, la))) return rc;
        cur = cubeA;
        if (do_lin) pdq_mid = pdq_ws;
    } else {
        cur = (const float *)d_data;
    }
    mark();
    // ---- IPC
    if (do_ipc) {
        float *cubeB = (float *)rip_ws(ctx, 1, (size_t)G * npix * 4);
        if (!cubeB) return RIP_ENOMEM;
        IpcArgs ia{cur, cubeB, c.ipc, c.gain, c.ipc_dtype, c.gain_dtype, ny, nx, c.nb, G};
        if ((rc = rip_launch_ipc_cube(ctx, ia))) return rc;
        cur = cubeB;
    }
    mark();
    // ---- ramp fit + finish
    if (do_fit) {
        RampFitArgs fa;
        memset(&fa, 0, sizeof fa);
        fa.cube = cur;
        fa.gdq_in = d_gdq;
        fa.gdq_out = o_gdq;
        fa.pdq_in = pdq_mid;
        fa.pdq_out = o_pdq;
        fa.gain = c.gain;
        fa.read_noise = c.read_noise;
        fa.slope = o_slope;
        fa.err_read = o_er;
        fa.err_poisson = o_ep;
        fa.finish = (stages & (RIP_STAGE_DARK | RIP_STAGE_FLAT)) ? 1 : 0;
        if (stages & RIP_STAGE_DARK) {
            fa.dark_rate = c.dark_rate;
            fa.dark_dq = c.has_dark_dq ? c.dark_dq : nullptr;
        }
        fa.flat = flat_plane;
        fa.flat_flags = flat_plane ? c.flat_flags : nullptr;
        fa.ny = ny;
        fa.nx = nx;
        fa.nb = c.nb;
        fa.ngrp = G;
        if ((rc = rip_launch_rampfit(ctx, plan, fa, c.gain_dtype))) return rc;
    }
    mark();
    }  // unfused
    ctx->stream_dirty = false;
    // the main-stream kernels of this call are the last readers of the tables / flag copies of parity `par`: the
    // pre-pass of call n+2 (same parity, second stream) waits for this event before it overwrites them
    // EVERY call takes a parity and leaves its event, overlapped or not: a call whose pre-pass / flag pass ran on the main stream
    // has used the buffers of `par` too, and the next overlapped call must neither reuse them (it takes the other parity)
    // nor, two calls on, overwrite them before this call's main-stream kernels are done
    if (ctx->ev_done[par]) {
        RIP_HIP(ctx, hipEventRecord(ctx->ev_done[par], ctx->stream));
        ctx->ev_done_valid[par] = true;
    }
    ctx->parity ^= 1;
    // ---- results back
    if (host) {
        if (do_fit) {
            RIP_HIP(ctx, hipMemcpyAsync(out->slope, o_slope, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
            RIP_HIP(ctx, hipMemcpyAsync(out->err_read, o_er, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
            RIP_HIP(ctx, hipMemcpyAsync(out->err_poisson, o_ep, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
            RIP_HIP(ctx, hipMemcpyAsync(out->pixeldq, o_pdq, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
            if (out->groupdq) RIP_HIP(ctx, hipMemcpyAsync(out->groupdq, o_gdq, b_gdq, hipMemcpyDeviceToHost, ctx->stream));
        } else if (out->pixeldq) {
            RIP_HIP(ctx, hipMemcpyAsync(out->pixeldq, pdq_mid, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (out->cube) RIP_HIP(ctx, hipMemcpyAsync(out->cube, cur, (size_t)G * npix * 4, hipMemcpyDeviceToHost, ctx->stream));
        RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        if (!do_fit && out->pixeldq && out->pixeldq != pdq_mid)
            RIP_HIP(ctx, hipMemcpyAsync(out->pixeldq, pdq_mid, npix * 4, hipMemcpyDeviceToDevice, ctx->stream));
        if (o_cube && o_cube != cur)
            RIP_HIP(ctx, hipMemcpyAsync(o_cube, cur, (size_t)G * npix * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return RIP_OK;
}

// --------------------------------------------------------------------------- stage-level entry points
int rip_stage_refpix_image(rip_ctx *ctx, float *image, int ny, int nx, double slope, int do_row, int do_channel,
                           const double *lines, float *ref_med, float *ctr, float *bottom_top) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const int w = nx + RIP_CW, nch = w / RIP_CW;
    const size_t n = (size_t)ny * w;
    DevBuf img, ln, rm, ct, bt;
    int rc;
    if ((rc = img.upload(ctx, image, n * 4))) return rc;
    if (lines && (rc = ln.upload(ctx, lines, (size_t)nch * 16))) return rc;
    if ((rc = rm.alloc(ctx, (size_t)ny * 4)) || (rc = ct.alloc(ctx, 4)) || (rc = bt.alloc(ctx, (size_t)nch * 8))) return rc;
    if ((rc = rip_refpix_image(ctx, img.as<float>(), ny, nx, slope, do_row, do_channel, lines ? ln.as<double>() : nullptr,
                               rm.as<float>(), ct.as<float>(), bt.as<float>())))
        return rc;
    RIP_HIP(ctx, hipMemcpyAsync(image, img.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ref_med && do_row) RIP_HIP(ctx, hipMemcpyAsync(ref_med, rm.p, (size_t)ny * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ctr && do_row) RIP_HIP(ctx, hipMemcpyAsync(ctr, ct.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (bottom_top && do_channel) RIP_HIP(ctx, hipMemcpyAsync(bottom_top, bt.p, (size_t)nch * 8, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

int rip_stage_refpix_row(rip_ctx *ctx, float *image, int ny, int width, int nside, int use_ref_channel, int mode, double slope,
                         float *ref_med, float *sci_med, float *ctr) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    if (!image || ny < 1 || width < 1) return rip_fail(ctx, RIP_EINVAL, "refpix row: image required");
    if (mode < RIP_ROW_MEDIANS_ONLY || mode > RIP_ROW_SLOPE_F32) return rip_fail(ctx, RIP_EINVAL, "refpix row: mode %d", mode);
    const size_t n = (size_t)ny * width;
    DevBuf img, rm, sm, ct;
    int rc;
    if ((rc = img.upload(ctx, image, n * 4))) return rc;
    if ((rc = rm.alloc(ctx, (size_t)ny * 4)) || (rc = ct.alloc(ctx, 4))) return rc;
    if (sci_med && (rc = sm.alloc(ctx, (size_t)ny * 4))) return rc;
    if ((rc = rip_refpix_row_general(ctx, img.as<float>(), ny, width, nside, use_ref_channel, mode, slope, rm.as<float>(),
                                     sci_med ? sm.as<float>() : nullptr, ct.as<float>())))
        return rc;
    if (mode != RIP_ROW_MEDIANS_ONLY) RIP_HIP(ctx, hipMemcpyAsync(image, img.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ref_med) RIP_HIP(ctx, hipMemcpyAsync(ref_med, rm.p, (size_t)ny * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (sci_med) RIP_HIP(ctx, hipMemcpyAsync(sci_med, sm.p, (size_t)ny * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ctr) RIP_HIP(ctx, hipMemcpyAsync(ctr, ct.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

int rip_stage_refpix_channel(rip_ctx *ctx, float *image, int ny, int width, int channel_start, int channel_end, int nchan,
                             const double *lines, float *bottom_top) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    if (!image || ny < 1 || width < 1 || nchan < 1) return rip_fail(ctx, RIP_EINVAL, "refpix channel: image required");
    const size_t n = (size_t)ny * width;
    DevBuf img, ln, bt;
    int rc;
    if ((rc = img.upload(ctx, image, n * 4))) return rc;
    if (lines && (rc = ln.upload(ctx, lines, (size_t)nchan * 16))) return rc;
    if ((rc = bt.alloc(ctx, (size_t)nchan * 8))) return rc;
    if ((rc = rip_refpix_channel_general(ctx, img.as<float>(), ny, width, channel_start, channel_end, nchan,
                                         lines ? ln.as<double>() : nullptr, bt.as<float>())))
        return rc;
    RIP_HIP(ctx, hipMemcpyAsync(image, img.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (bottom_top) RIP_HIP(ctx, hipMemcpyAsync(bottom_top, bt.p, (size_t)nchan * 8, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

int rip_stage_refpix_tables(rip_ctx *ctx, const void *data, int data_dtype, const float *dark, const uint16_t *amp33,
                            const float *amp33_med, double slope, int ngrp, int ny, int nx, int form, double *rowcorr,
                            double *lines, int *status) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    if (!data || !dark || !amp33 || !amp33_med || !rowcorr || !lines || ngrp < 1 || ngrp > RIP_MAX_GROUPS || ny < 8 || nx < RIP_CW ||
        nx % RIP_CW || (data_dtype != RIP_U16 && data_dtype != RIP_F32))
        return rip_fail(ctx, RIP_EINVAL, "refpix tables: bad argument");
    const size_t npix = (size_t)ny * nx, nch = (size_t)nx / RIP_CW;
    DevBuf d_data, d_dark, d_a33, d_med, d_rc, d_rt, d_ln;
    int rc;
    if ((rc = d_data.upload(ctx, data, (size_t)ngrp * npix * dsize(data_dtype))) || (rc = d_dark.upload(ctx, dark, (size_t)ngrp * npix * 4)) ||
        (rc = d_a33.upload(ctx, amp33, (size_t)ngrp * ny * RIP_CW * 2)) || (rc = d_med.upload(ctx, amp33_med, (size_t)ny * RIP_CW * 4)) ||
        (rc = d_rc.alloc(ctx, (size_t)ngrp * ny * 8)) || (rc = d_rt.alloc(ctx, (size_t)ngrp * ny * 8)) || (rc = d_ln.alloc(ctx, (size_t)ngrp * nch * 16)))
        return rc;
    RefpixArgs ra{d_data.p, data_dtype, d_dark.as<float>(), d_a33.as<uint16_t>(), d_med.as<float>(), slope, nullptr,
                  d_rc.as<double>(), d_rt.as<double>(), d_ln.as<double>(), ny, nx, ngrp};
    if (form < -1 || form > 1) return rip_fail(ctx, RIP_EINVAL, "refpix tables: form %d", form);
    const int keep = ctx->prepass_form;
    if (form >= 0) ctx->prepass_form = form;
    if (form >= 1 && !rip_refpix_one_supported(ra)) {
        ctx->prepass_form = keep;
        return rip_fail(ctx, RIP_EINVAL, "refpix tables: the single-launch kernel does not cover a %d x %d frame of %d groups", ny, nx, ngrp);
    }
    rc = rip_launch_refpix_prepass(ctx, ra);
    ctx->prepass_form = keep;
    if (rc) return rc;
    RIP_HIP(ctx, hipMemcpyAsync(rowcorr, d_rc.p, (size_t)ngrp * ny * 8, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(lines, d_ln.p, (size_t)ngrp * nch * 16, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (status) return rip_refpix_one_status(ctx, status);
    return RIP_OK;
}

int rip_stage_multilin(rip_ctx *ctx, const float *S, int ngrp, int ny, int nx, int nplanes, const float *coefs,
                       const float *smin, const float *smax, const float *sref, const uint32_t *lin_dq,
                       int do_not_flag_first, const uint8_t *attempt_corr, float *phi, uint32_t *dq) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ny * nx;
    DevBuf dS, dC, dmin, dmax, dref, ddq, dac, dphi, dout;
    int rc;
    if ((rc = dS.upload(ctx, S, (size_t)ngrp * npix * 4)) || (rc = dC.upload(ctx, coefs, (size_t)nplanes * npix * 4)) ||
        (rc = dmin.upload(ctx, smin, npix * 4)) || (rc = dmax.upload(ctx, smax, npix * 4)) ||
        (rc = dref.upload(ctx, sref, npix * 4)) || (rc = ddq.upload(ctx, lin_dq, npix * 4)) ||
        (rc = dphi.alloc(ctx, (size_t)ngrp * npix * 4)) || (rc = dout.alloc(ctx, npix * 4)))
        return rc;
    if (attempt_corr && (rc = dac.upload(ctx, attempt_corr, (size_t)ngrp * npix))) return rc;
    LinArgs la;
    memset(&la, 0, sizeof la);
    la.data = dS.p;
    la.data_dtype = RIP_F32;
    la.phi = dphi.as<float>();
    la.gdq = attempt_corr ? dac.as<uint8_t>() : nullptr;
    la.gdq_is_attempt = 1;
    la.pdq_out = dout.as<uint32_t>();
    la.coefs = dC.as<float>();
    la.smin = dmin.as<float>();
    la.smax = dmax.as<float>();
    la.sref = dref.as<float>();
    la.lin_dq = ddq.as<uint32_t>();
    la.nplanes = nplanes;
    la.do_not_flag_first = do_not_flag_first;
    la.ny = ny;
    la.nx = nx;
    la.nb = 0;
    la.ngrp = ngrp;
    if ((rc = rip_launch_lin(ctx, la))) return rc;
    RIP_HIP(ctx, hipMemcpyAsync(phi, dphi.p, (size_t)ngrp * npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(dq, dout.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

int rip_stage_ipc_image(rip_ctx *ctx, int reverse, int order, const void *image, int img_dtype, int ny, int nx,
                        const void *kernel, int k_dtype, const void *gain, int g_dtype, void *outp) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ny * nx;
    const bool t64 = img_dtype == RIP_F64 || k_dtype == RIP_F64 || (gain && g_dtype == RIP_F64);
    DevBuf di, dk, dg, dout;
    int rc;
    if ((rc = di.upload(ctx, image, npix * dsize(img_dtype))) || (rc = dk.upload(ctx, kernel, 9 * npix * dsize(k_dtype))) ||
        (rc = dout.alloc(ctx, npix * (t64 ? 8 : 4))))
        return rc;
    if (gain && (rc = dg.upload(ctx, gain, npix * dsize(g_dtype)))) return rc;
    if ((rc = rip_launch_ipc_image(ctx, reverse, order, di.p, img_dtype, ny, nx, dk.p, k_dtype, gain ? dg.p : nullptr, g_dtype,
                                   dout.p, 0)))
        return rc;
    RIP_HIP(ctx, hipMemcpyAsync(outp, dout.p, npix * (t64 ? 8 : 4), hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

int rip_stage_correct_cube(rip_ctx *ctx, float *data, int ngrp, int ny, int nx, int nb, const void *kernel, int k_dtype,
                           const void *gain, int g_dtype) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ny * nx;
    const int nya = ny - 2 * nb, nxa = nx - 2 * nb;
    const size_t es = dsize(k_dtype);
    DevBuf din, dout, kraw, kemb, dg;
    int rc;
    if ((rc = din.upload(ctx, data, (size_t)ngrp * npix * 4)) || (rc = dout.alloc(ctx, (size_t)ngrp * npix * 4)) ||
        (rc = kraw.upload(ctx, kernel, (size_t)9 * nya * nxa * es)) || (rc = kemb.alloc(ctx, 9 * npix * es)))
        return rc;
    if (gain && (rc = dg.upload(ctx, gain, npix * dsize(g_dtype)))) return rc;
    if ((rc = rip_launch_embed(ctx, kraw.p, kemb.p, 9, ny, nx, nb, (int)es))) return rc;
    IpcArgs ia{din.as<float>(), dout.as<float>(), kemb.p, gain ? dg.p : nullptr, k_dtype, g_dtype, ny, nx, nb, ngrp};
    if ((rc = rip_launch_ipc_cube(ctx, ia))) return rc;
    RIP_HIP(ctx, hipMemcpyAsync(data, dout.p, (size_t)ngrp * npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

int rip_stage_ramp_fit(rip_ctx *ctx, int plan_id, const float *data, uint8_t *rdq, uint32_t *pdq, int ny, int nx, int nb,
                       const void *gain, int g_dtype, const float *read_noise, float *slope, float *err_read,
                       float *err_poisson) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    RipPlan *plan = get_plan(ctx, plan_id);
    if (!plan) return RIP_EINVAL;
    const int G = plan->h.ngrp;
    const size_t npix = (size_t)ny * nx;
    DevBuf dd, dr, dp, dg, dn, ds, de, dq2, dr2, dp2;
    int rc;
    if ((rc = dd.upload(ctx, data, (size_t)G * npix * 4)) || (rc = dr.upload(ctx, rdq, (size_t)G * npix)) ||
        (rc = dp.upload(ctx, pdq, npix * 4)) || (rc = dg.upload(ctx, gain, npix * dsize(g_dtype))) ||
        (rc = dn.upload(ctx, read_noise, npix * 4)) || (rc = ds.alloc(ctx, npix * 4)) || (rc = de.alloc(ctx, npix * 4)) ||
        (rc = dq2.alloc(ctx, npix * 4)) || (rc = dr2.alloc(ctx, (size_t)G * npix)) || (rc = dp2.alloc(ctx, npix * 4)))
        return rc;
    RampFitArgs fa;
    memset(&fa, 0, sizeof fa);
    fa.cube = dd.as<float>();
    fa.gdq_in = dr.as<uint8_t>();
    fa.gdq_out = dr2.as<uint8_t>();
    fa.pdq_in = dp.as<uint32_t>();
    fa.pdq_out = dp2.as<uint32_t>();
    fa.gain = dg.p;
    fa.read_noise = dn.as<float>();
    fa.slope = ds.as<float>();
    fa.err_read = de.as<float>();
    fa.err_poisson = dq2.as<float>();
    fa.finish = 0;
    fa.ny = ny;
    fa.nx = nx;
    fa.nb = nb;
    fa.ngrp = G;
    if ((rc = rip_launch_rampfit(ctx, plan, fa, g_dtype))) return rc;
    RIP_HIP(ctx, hipMemcpyAsync(slope, ds.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(err_read, de.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(err_poisson, dq2.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(rdq, dr2.p, (size_t)G * npix, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(pdq, dp2.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

int rip_stage_jump_detect(rip_ctx *ctx, int plan_id, const float *data, uint8_t *rdq, int ny, int nx, int nb, const void *gain,
                          int g_dtype, const float *read_noise, float *slope, float *err_read, float *err_poisson, float *smap) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    RipPlan *plan = get_plan(ctx, plan_id);
    if (!plan) return RIP_EINVAL;
    if (!data || !rdq || !gain || !read_noise || !slope || !err_read || !err_poisson || !smap)
        return rip_fail(ctx, RIP_EINVAL, "jump_detect: NULL array");
    const int G = plan->h.ngrp, nd = plan->variants[0].ndiff;
    if (G < 2 || nd <= 0) return rip_fail(ctx, RIP_EINVAL, "jump_detect: a plan of %d groups has no difference to test", G);
    const size_t npix = (size_t)ny * nx;
    DevBuf dd, dr, dg, dn, ds, de, dp, dm;
    int rc;
    if ((rc = dd.upload(ctx, data, (size_t)G * npix * 4)) || (rc = dr.upload(ctx, rdq, (size_t)G * npix)) ||
        (rc = dg.upload(ctx, gain, npix * dsize(g_dtype))) || (rc = dn.upload(ctx, read_noise, npix * 4)) ||
        (rc = ds.alloc(ctx, npix * 4)) || (rc = de.alloc(ctx, npix * 4)) || (rc = dp.alloc(ctx, npix * 4)) ||
        (rc = dm.alloc(ctx, (size_t)(nd > 0 ? nd : 1) * npix * 4)))
        return rc;
    if ((rc = rip_launch_jumpdetect(ctx, plan, dd.as<float>(), dr.as<uint8_t>(), dg.p, g_dtype, dn.as<float>(), ds.as<float>(),
                                    de.as<float>(), dp.as<float>(), dm.as<float>(), ny, nx, nb)))
        return rc;
    RIP_HIP(ctx, hipMemcpyAsync(slope, ds.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(err_read, de.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(err_poisson, dp.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipMemcpyAsync(rdq, dr.p, (size_t)G * npix, hipMemcpyDeviceToHost, ctx->stream));
    if (nd > 0) RIP_HIP(ctx, hipMemcpyAsync(smap, dm.p, (size_t)nd * npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

int rip_stage_get_flat(rip_ctx *ctx, const float *flat, int ny, int nx, int nb, const void *gain, int g_dtype,
                       const void *kernel, int k_dtype, int ipc_deconvolve, uint32_t *pdq, float *outp) {
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ny * nx;
    const int nya = ny - 2 * nb, nxa = nx - 2 * nb;
    const size_t es = dsize(k_dtype);
    DevBuf raw, padded, dg, gclip, flags, kraw, kemb, dout;
    int rc;
    if ((rc = raw.upload(ctx, flat, npix * 4)) || (rc = padded.alloc(ctx, npix * 4)) || (rc = flags.alloc(ctx, npix * 4)) ||
        (rc = dout.alloc(ctx, npix * 4)))
        return rc;
    int with_gain = 0;
    if (ipc_deconvolve) {
        if (!gain || !kernel) return rip_fail(ctx, RIP_EINVAL, "get_flat: gain and ipc4d needed for deconvolution");
        with_gain = pdq ? 1 : 2;
        if ((rc = dg.upload(ctx, gain, npix * dsize(g_dtype))) || (rc = gclip.alloc(ctx, npix * dsize(g_dtype))) ||
            (rc = kraw.upload(ctx, kernel, (size_t)9 * nya * nxa * es)) || (rc = kemb.alloc(ctx, 9 * npix * es)))
            return rc;
        if ((rc = rip_launch_embed(ctx, kraw.p, kemb.p, 9, ny, nx, nb, (int)es))) return rc;
    }
    if ((rc = rip_launch_flat_prepare(ctx, raw.as<float>(), dg.p, g_dtype, ny, nx, nb, padded.as<float>(), gclip.p,
                                      flags.as<uint32_t>(), with_gain)))
        return rc;
    const void *res = padded.p;
    if (ipc_deconvolve) {
        IpcArgs ia{padded.as<float>(), dout.as<float>(), kemb.p, gclip.p, k_dtype, g_dtype, ny, nx, nb, 1};
        if ((rc = rip_launch_ipc_cube(ctx, ia))) return rc;
        res = dout.p;
    }
    RIP_HIP(ctx, hipMemcpyAsync(outp, res, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (pdq) {
        std::vector<uint32_t> fl(npix);
        RIP_HIP(ctx, hipMemcpyAsync(fl.data(), flags.p, npix * 4, hipMemcpyDeviceToHost, ctx->stream));
        RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i < npix; ++i) pdq[i] |= fl[i];
    }
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}

}  // extern "C"
