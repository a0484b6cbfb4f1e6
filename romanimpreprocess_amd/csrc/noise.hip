// Noise layers: the arithmetic gen_noise_image.make_noise_cube applies to a Level-1 cube before it sends
// it through the chain again (gen_noise_image.py:120-134), per group k and active pixel:
//     im = f32( f64(normal) * (f64(read) / sqrt(f64(N_k))) )         numpy: f32 array *= (f32 array / np.float64 scalar)
//     r  = f32(data) + im ;  data' = u16( rint( clip(r, 0, 65535) ) )   np.round = round half to even
// The standard normal deviates come from the caller (the reference draws them with galsim.GaussianDeviate; which generator
// is used is the driver's business) -- or, normals == NULL, from a counter-based generator on the device (Philox-4x32-10 +
// Box-Muller, keyed by seed, layer, group and pixel), so that a layer is reproducible and needs no host random numbers.
// Border pixels (reference pixels) pass through unchanged.  Host arrays in and out.  Exact given the normals.
#include "rip_common.h"
#include <cstring>

namespace {

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1;
    c[3] = (uint32_t)p0;
    c[0] = n0;
    c[2] = n2;
}
__device__ __forceinline__ void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// one standard normal for (seed, layer, group, pixel): Box-Muller on two uniforms of the Philox block
__device__ __forceinline__ float device_normal(uint64_t seed, uint32_t layer, uint32_t group, uint32_t pix) {
    uint32_t c[4] = {pix, group, layer, 0x6e6f6973u};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const float u1 = ((float)(c[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
    const float u2 = ((float)(c[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cospif(2.0f * u2);
}

__global__ __launch_bounds__(256) void noise_inject_kernel(const uint16_t *__restrict__ cube, const float *__restrict__ normals,
                                                           const float *__restrict__ read, const double *__restrict__ rsqn,
                                                           int G, int ny, int nx, int nb, uint64_t seed, uint32_t layer,
                                                           uint16_t *__restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, k = blockIdx.z;
    if (x >= nx) return;
    const size_t i = ((size_t)k * ny + y) * nx + x;
    const uint16_t v = cube[i];
    if (y < nb || y >= ny - nb || x < nb || x >= nx - nb) {
        out[i] = v;
        return;
    }
    const int nxa = nx - 2 * nb, nya = ny - 2 * nb;
    const size_t ia = ((size_t)k * nya + (y - nb)) * nxa + (x - nb);
    const float nrm = normals ? normals[ia] : device_normal(seed, layer, (uint32_t)k, (uint32_t)((y - nb) * nxa + (x - nb)));
    const float im = (float)((double)nrm * ((double)read[(size_t)y * nx + x] / rsqn[k]));
    float r = __fadd_rn((float)v, im);
    r = r < 0.0f ? 0.0f : (r > 65535.0f ? 65535.0f : r);   // np.clip keeps NaN; the cast of NaN is not defined in numpy either
    out[i] = (uint16_t)rintf(r);
}

}   // namespace

extern "C" int rip_stage_noise_inject(rip_ctx *ctx, const uint16_t *cube, int ngrp, int ny, int nx, int nb, const float *read_noise,
                                      const int32_t *nreads, const float *normals, uint64_t seed, uint32_t layer, uint16_t *out) {
    if (!cube || !read_noise || !nreads || !out || ngrp < 1 || ngrp > RIP_MAX_GROUPS || ny < 1 || nx < 1 || nb < 0 || 2 * nb >= ny ||
        2 * nb >= nx)
        return rip_fail(ctx, RIP_EINVAL, "noise_inject: bad arguments");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ny * nx, nact = (size_t)(ny - 2 * nb) * (nx - 2 * nb);
    double rs[RIP_MAX_GROUPS];
    for (int k = 0; k < ngrp; ++k) {
        if (nreads[k] < 1) return rip_fail(ctx, RIP_EINVAL, "noise_inject: group %d has %d reads", k, nreads[k]);
        rs[k] = sqrt((double)nreads[k]);   // the kernel divides: read / sqrt(N) as numpy does
    }
    void *d_cube = nullptr, *d_out = nullptr, *d_read = nullptr, *d_nrm = nullptr, *d_rs = nullptr;
    int rc = RIP_OK;
    auto done = [&]() {
        for (void *p : {d_cube, d_out, d_read, d_nrm, d_rs})
            if (p) (void)hipFree(p);
    };
#define NZ_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            rc = rip_fail(ctx, RIP_EHIP, "%s: %s", #call, hipGetErrorString(e_));      \
            done();                                                                    \
            return rc;                                                                 \
        }                                                                              \
    } while (0)
    NZ_HIP(hipMalloc(&d_cube, (size_t)ngrp * npix * 2));
    NZ_HIP(hipMalloc(&d_out, (size_t)ngrp * npix * 2));
    NZ_HIP(hipMalloc(&d_read, npix * 4));
    NZ_HIP(hipMalloc(&d_rs, sizeof(double) * RIP_MAX_GROUPS));
    NZ_HIP(hipMemcpyAsync(d_cube, cube, (size_t)ngrp * npix * 2, hipMemcpyDefault, ctx->stream));
    NZ_HIP(hipMemcpyAsync(d_read, read_noise, npix * 4, hipMemcpyDefault, ctx->stream));
    NZ_HIP(hipMemcpyAsync(d_rs, rs, sizeof(double) * ngrp, hipMemcpyDefault, ctx->stream));
    if (normals) {
        NZ_HIP(hipMalloc(&d_nrm, (size_t)ngrp * nact * 4));
        NZ_HIP(hipMemcpyAsync(d_nrm, normals, (size_t)ngrp * nact * 4, hipMemcpyDefault, ctx->stream));
    }
    hipLaunchKernelGGL(noise_inject_kernel, dim3((nx + 255) / 256, ny, ngrp), dim3(256), 0, ctx->stream, (const uint16_t *)d_cube,
                       (const float *)d_nrm, (const float *)d_read, (const double *)d_rs, ngrp, ny, nx, nb, seed, layer,
                       (uint16_t *)d_out);
    NZ_HIP(hipGetLastError());
    NZ_HIP(hipMemcpyAsync(out, d_out, (size_t)ngrp * npix * 2, hipMemcpyDefault, ctx->stream));
    NZ_HIP(hipStreamSynchronize(ctx->stream));
#undef NZ_HIP
    done();
    return RIP_OK;
}

// ------------------------------------------------------------------------------------------ resampled Poisson ('P..r')
// gen_noise_image.py:262-331: per active pixel, with e = clip(skylevel * gain * t_frame, 0) electrons per frame,
//     for isamp = 0 .. lastsamp:  s = Poisson(e) - e ;  s /= gain ;  cur (f32) = f32(f64(cur) + s)            (f64 sample)
//                                 for every group j holding read isamp:  delta[j] (f32) += cur / N_j
//     diff (f32) += sum_j w[endslice][j] * delta[j]        one f32 product and one f32 addition per j, j ascending
// where w[es] is the weight vector the ramp fit used for a ramp ending at group es (processinfo weights for the full ramp,
// the two-point weights for truncated ones) and endslice comes from the L2 file (SLICEOUT).  The Poisson deviates come from
// the caller (samples (nsamp, n) f64, the reference order) or, samples == NULL, from the device generator: inversion by
// sequential search below a mean of 10, Hoermann's transformed rejection (PTRS) above, uniforms from Philox keyed by (seed,
// layer, read, pixel, attempt).  Everything but the deviates is exact.
namespace {

__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t a, uint32_t b, uint32_t c_, uint32_t d, int which) {
    uint32_t c[4] = {a, b, c_, d};
    philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return ((float)(c[which] >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

__device__ double device_poisson(double lam, uint64_t seed, uint32_t layer, uint32_t isamp, uint32_t pix) {
    if (!(lam > 0.0)) return 0.0;
    if (lam < 10.0) {   // inversion: sequential search of the cumulative distribution with one uniform (two words: 48 bits)
        uint32_t c[4] = {pix, isamp, layer, 0x706f6973u};
        philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const double u = ((double)(((uint64_t)c[0] << 16) | (c[1] >> 16)) + 0.5) * (1.0 / 281474976710656.0);
        double p = exp(-lam), cdf = p;
        int k = 0;
        while (u > cdf && k < 200) {
            ++k;
            p *= lam / k;
            cdf += p;
        }
        return (double)k;
    }
    // PTRS (W. Hoermann, "The transformed rejection method for generating Poisson random variables", 1993)
    const double slam = sqrt(lam), loglam = log(lam);
    const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b, inv_alpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
    for (uint32_t attempt = 0; attempt < 64; ++attempt) {
        uint32_t c[4] = {pix, isamp, layer ^ (attempt << 16), 0x70747273u};
        philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const double u = ((double)c[0] + 0.5) * (1.0 / 4294967296.0) - 0.5;
        const double v = ((double)c[1] + 0.5) * (1.0 / 4294967296.0);
        const double us = 0.5 - fabs(u);
        const double k = floor((2.0 * a / us + b) * u + lam + 0.43);
        if (us >= 0.07 && v <= vr) return k;
        if (k < 0.0 || (us < 0.013 && v > us)) continue;
        if (log(v) + log(inv_alpha) - log(a / (us * us) + b) <= -lam + k * loglam - lgamma(k + 1.0)) return k;
    }
    return floor(lam + 0.5);   // not reached in practice
}

struct ResampleArgs {
    const float *sky;        // (n) skylevel, DN/s
    const void *gain;        // (n) clipped gain, f32 or f64
    int gain_f64;
    double t_frame;
    const int8_t *endslice;  // (n), already mapped: <= 0 -> ngrp - 1
    const double *samples;   // (nsamp, n) or null
    float *diff;             // (n) in/out
    size_t n;
    int ngrp, nsamp;
    uint64_t seed;
    uint32_t layer;
};
struct ResampleTables {
    float w[RIP_MAX_GROUPS > 16 ? 16 : RIP_MAX_GROUPS][16];   // w[es][j]; rows without a weight vector are flagged in has
    int has[16];
    int first[16], count[16];   // reads of group j: first .. first + count - 1 (contiguous, as in every MA table)
};

template <typename GT>
__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a, ResampleTables t) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const GT g = reinterpret_cast<const GT *>(a.gain)[i];
    // e = skylevel * gain * t_frame in the promoted type, clipped at zero
    GT e = (GT)a.sky[i] * g;
    e = (GT)(e * (GT)a.t_frame);
    if (e < (GT)0) e = (GT)0;
    float cur = 0.0f;
    float delta[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) delta[j] = 0.0f;
    for (int s = 0; s < a.nsamp; ++s) {
        const double k = a.samples ? a.samples[(size_t)s * a.n + i] : device_poisson((double)e, a.seed, a.layer, (uint32_t)s, (uint32_t)i);
        double smp = k - (double)e;
        smp = smp / (double)g;
        cur = (float)((double)cur + smp);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < a.ngrp && s >= t.first[j] && s < t.first[j] + t.count[j]) delta[j] = __fadd_rn(delta[j], cur / (float)t.count[j]);
    }
    const int es = a.endslice[i];
    float d = __fadd_rn(a.diff[i], 0.0f);   // the reference adds 0.0 for every other end slice (-0 becomes +0)
    if (es >= 0 && es < a.ngrp && t.has[es]) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < a.ngrp) d = __fadd_rn(d, __fmul_rn(t.w[es][j], delta[j]));
    }
    a.diff[i] = d;
}

}   // namespace

extern "C" int rip_stage_poisson_resample(rip_ctx *ctx, const float *skylevel, const void *gain, int gain_dtype, size_t n,
                                          double frame_time, int ngrp, const int32_t *group_first, const int32_t *group_count,
                                          const float *weights, const uint8_t *has_weights, const int8_t *endslice,
                                          const double *samples, int nsamp, uint64_t seed, uint32_t layer, float *diff) {
    if (!skylevel || !gain || !group_first || !group_count || !weights || !has_weights || !endslice || !diff || ngrp < 1 || ngrp > 16 ||
        nsamp < 1 || n < 1 || (gain_dtype != RIP_F32 && gain_dtype != RIP_F64))
        return rip_fail(ctx, RIP_EINVAL, "poisson_resample: bad arguments (up to 16 groups)");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    ResampleTables t;
    memset(&t, 0, sizeof t);
    for (int j = 0; j < ngrp; ++j) {
        t.first[j] = group_first[j];
        t.count[j] = group_count[j];
        if (group_count[j] < 1) return rip_fail(ctx, RIP_EINVAL, "poisson_resample: group %d has no reads", j);
        t.has[j] = has_weights[j] ? 1 : 0;
        for (int k = 0; k < ngrp; ++k) t.w[j][k] = weights[j * ngrp + k];
    }
    const size_t gs = gain_dtype == RIP_F64 ? 8 : 4;
    void *d_sky = nullptr, *d_gain = nullptr, *d_end = nullptr, *d_smp = nullptr, *d_diff = nullptr;
    int rc = RIP_OK;
    auto done = [&]() {
        for (void *p : {d_sky, d_gain, d_end, d_smp, d_diff})
            if (p) (void)hipFree(p);
    };
#define PR_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            rc = rip_fail(ctx, RIP_EHIP, "%s: %s", #call, hipGetErrorString(e_));      \
            done();                                                                    \
            return rc;                                                                 \
        }                                                                              \
    } while (0)
    PR_HIP(hipMalloc(&d_sky, n * 4));
    PR_HIP(hipMalloc(&d_gain, n * gs));
    PR_HIP(hipMalloc(&d_end, n));
    PR_HIP(hipMalloc(&d_diff, n * 4));
    PR_HIP(hipMemcpyAsync(d_sky, skylevel, n * 4, hipMemcpyDefault, ctx->stream));
    PR_HIP(hipMemcpyAsync(d_gain, gain, n * gs, hipMemcpyDefault, ctx->stream));
    PR_HIP(hipMemcpyAsync(d_end, endslice, n, hipMemcpyDefault, ctx->stream));
    PR_HIP(hipMemcpyAsync(d_diff, diff, n * 4, hipMemcpyDefault, ctx->stream));
    if (samples) {
        PR_HIP(hipMalloc(&d_smp, (size_t)nsamp * n * 8));
        PR_HIP(hipMemcpyAsync(d_smp, samples, (size_t)nsamp * n * 8, hipMemcpyDefault, ctx->stream));
    }
    ResampleArgs a;
    a.sky = (const float *)d_sky;
    a.gain = d_gain;
    a.gain_f64 = gain_dtype == RIP_F64;
    a.t_frame = frame_time;
    a.endslice = (const int8_t *)d_end;
    a.samples = (const double *)d_smp;
    a.diff = (float *)d_diff;
    a.n = n;
    a.ngrp = ngrp;
    a.nsamp = nsamp;
    a.seed = seed;
    a.layer = layer;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (gain_dtype == RIP_F64)
        hipLaunchKernelGGL(resample_kernel<double>, grid, dim3(256), 0, ctx->stream, a, t);
    else
        hipLaunchKernelGGL(resample_kernel<float>, grid, dim3(256), 0, ctx->stream, a, t);
    PR_HIP(hipGetLastError());
    PR_HIP(hipMemcpyAsync(diff, d_diff, n * 4, hipMemcpyDefault, ctx->stream));
    PR_HIP(hipStreamSynchronize(ctx->stream));
#undef PR_HIP
    done();
    return RIP_OK;
}
