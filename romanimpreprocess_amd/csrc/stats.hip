// Per-pixel statistics over many noise realisations of one ramp: the arithmetic of the reference's
// validation_tests/many_realizations.py:57-106 on device-resident stacks of realisations.
//   :69-72   diffs[j] = f32(L1[-1]) - f32(L1[1])                                   -> rip_stats_l1_diff
//   :74-77   images[j] / err[j] embedded with a zero border, w = not PixelMask1     -> rip_stats_l2_pack
//   :78-80   moments += (w, w*x, w*x^2) in f32, realisation after realisation
//   :83-89   mean, std, -1000 sentinel, embedding                                   -> rip_stats_reduce
//   :92-101  np.median over the realisations of diffs, images, err                     (all eight planes)
// Everything is per pixel and exact: f32 operations in the reference's order (sqrtf and / are the correctly rounded forms; the
// __fsqrt_rn intrinsic is not), medians by radix selection.
// Device pointers only: the stacks (realisations x ny x nx) are far larger than what a call should move over PCIe.
#include "rip_common.h"

namespace {

// ------------------------------------------------------------------------------------------ per realisation
__global__ __launch_bounds__(256) void l1_diff_kernel(const uint16_t *__restrict__ a, const uint16_t *__restrict__ b,
                                                      float *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = __fsub_rn((float)a[i], (float)b[i]);
}

// image / err: the (ny-2nb, nx-2nb) L2 planes inside a zero border; good = 1 where the pixel passes the grown mask that the
// reference builds on the TRIMMED dq plane (zero padding starts at the edge of the active region).
__global__ __launch_bounds__(256) void l2_pack_kernel(const float *__restrict__ slope, const float *__restrict__ er,
                                                      const float *__restrict__ ep, const uint32_t *__restrict__ dq,
                                                      int ny, int nx, int nb, uint32_t m1, uint32_t m5, uint32_t m9,
                                                      uint32_t m25, float *__restrict__ image, float *__restrict__ err,
                                                      uint8_t *__restrict__ good) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const size_t i = (size_t)y * nx + x;
    if (y < nb || y >= ny - nb || x < nb || x >= nx - nb) {
        image[i] = 0.0f;
        err[i] = 0.0f;
        good[i] = 0;
        return;
    }
    image[i] = slope[i];
    const float r = er[i], p = ep[i];
    err[i] = sqrtf(__fadd_rn(__fmul_rn(r, r), __fmul_rn(p, p)));   // sqrt(var_rnoise + var_poisson) of the L2 file
    uint32_t hit = dq[i] & (m1 | m5 | m9 | m25);
    if (!hit && (m5 | m9 | m25)) {
        for (int dy = -2; dy <= 2 && !hit; ++dy) {
            const int yy = y + dy;
            if (yy < nb || yy >= ny - nb) continue;
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if (xx < nb || xx >= nx - nb) continue;
                const int ay = dy < 0 ? -dy : dy, ax = dx < 0 ? -dx : dx;
                uint32_t m = m25;
                if (ay <= 1 && ax <= 1) m |= m9;
                if (ay + ax <= 1) m |= m5;
                hit |= dq[(size_t)yy * nx + xx] & m;
            }
        }
    }
    good[i] = hit ? 0 : 1;
}

// ------------------------------------------------------------------------------------------ moments over realisations
// out planes (each npix): N, mean, std, mean - ideal   (many_realizations.py:78-89, 98-100)
__global__ __launch_bounds__(256) void seed_moments_kernel(const float *__restrict__ images, const uint8_t *__restrict__ good,
                                                           const float *__restrict__ ideal, int S, size_t npix, int y0, int ny,
                                                           int nx, int nb, float *__restrict__ oN, float *__restrict__ oMean,
                                                           float *__restrict__ oStd, float *__restrict__ oDiff) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const int y = y0 + (int)(p / nx), x = (int)(p % nx);
    float n = 0.0f, m1 = 0.0f, m2 = 0.0f;
    if (!(y < nb || y >= ny - nb || x < nb || x >= nx - nb)) {
        float s1 = 0.0f, s2 = 0.0f;
        for (int s = 0; s < S; ++s) {
            const size_t i = (size_t)s * npix + p;
            if (good[i]) {
                const float v = images[i];
                n = __fadd_rn(n, 1.0f);
                s1 = __fadd_rn(s1, v);
                s2 = __fadd_rn(s2, __fmul_rn(v, v));
            }
        }
        const float d = __fadd_rn(n, 1e-25f);
        m1 = ((s1) / (d));
        m2 = ((s2) / (d));
        float var = __fsub_rn(m2, __fmul_rn(m1, m1));
        if (var < 0.0f) var = 0.0f;   // np.clip(., 0, None): NaN stays NaN
        m2 = sqrtf(var);
        if (!(n > 0.1f)) m1 = m2 = -1000.0f;
    }
    oN[p] = n;
    oMean[p] = m1;
    oStd[p] = m2;
    oDiff[p] = __fsub_rn(m1, ideal[p]);
}

// ------------------------------------------------------------------------------------------ medians over realisations
__device__ __forceinline__ uint32_t f2key(float v) {
    const uint32_t b = __float_as_uint(v);
    return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) { return __uint_as_float(k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu)); }

// One workgroup of four waves per 64 pixels; the column of S keys of each pixel sits in LDS (tile[s*64 + lane]) or, when
// S*256 B does not fit, is re-read from the stack.  k-th smallest key by radix selection, two bits per pass: wave w counts
// over the realisations s = w (mod 4), the four partial counts meet in LDS and every wave takes the same decision.
// np.median semantics: mean of the two middle elements in f32 for even S ((a + b) / 2), NaN if the column holds a NaN.
template <bool LDS>
__global__ __launch_bounds__(256) void seed_median_kernel(const float *__restrict__ stack, int S, size_t npix,
                                                          float *__restrict__ out) {
    extern __shared__ uint32_t tile[];
    __shared__ uint32_t part[2][4][3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t p = (size_t)blockIdx.x * 64 + lane;
    const bool live = p < npix;
    const size_t pp = live ? p : npix - 1;
    uint32_t nan = 0;
    for (int s = w; s < S; s += 4) {
        const float v = stack[(size_t)s * npix + pp];
        nan |= v != v ? 1u : 0u;
        if (LDS) tile[s * 64 + lane] = f2key(v);
    }
    auto key = [&](int s) -> uint32_t { return LDS ? tile[s * 64 + lane] : f2key(stack[(size_t)s * npix + pp]); };
    int k = (S - 1) / 2, buf = 0;
    uint32_t prefix = 0;
    for (int sh = 30; sh >= 0; sh -= 2) {
        const uint32_t hi = sh == 30 ? 0u : (0xFFFFFFFFu << (sh + 2));
        uint32_t c0 = 0, c1 = 0, c2 = 0;
#pragma unroll 8
        for (int s = w; s < S; s += 4) {
            const uint32_t e = key(s);
            const bool in = (e & hi) == prefix;
            const uint32_t d = (e >> sh) & 3u;
            c0 += (in && d == 0) ? 1 : 0;
            c1 += (in && d <= 1) ? 1 : 0;
            c2 += (in && d <= 2) ? 1 : 0;
        }
        part[buf][w][0][lane] = c0;
        part[buf][w][1][lane] = c1;
        part[buf][w][2][lane] = c2;
        __syncthreads();   // also orders the tile stores of the load loop before the first reads of other waves' keys
        c0 = c1 = c2 = 0;
        for (int q = 0; q < 4; ++q) {
            c0 += part[buf][q][0][lane];
            c1 += part[buf][q][1][lane];
            c2 += part[buf][q][2][lane];
        }
        buf ^= 1;   // the other buffer is free: every wave has passed the barrier after reading it
        uint32_t d;
        if (k < (int)c0) d = 0;
        else if (k < (int)c1) { d = 1; k -= c0; }
        else if (k < (int)c2) { d = 2; k -= c1; }
        else { d = 3; k -= c2; }
        prefix |= d << sh;
    }
    uint32_t le = 0, nxt = 0xFFFFFFFFu;
    if ((S & 1) == 0) {
#pragma unroll 8
        for (int s = w; s < S; s += 4) {
            const uint32_t e = key(s);
            le += e <= prefix ? 1 : 0;
            if (e > prefix && e < nxt) nxt = e;
        }
    }
    part[buf][w][0][lane] = le;
    part[buf][w][1][lane] = nxt;
    part[buf][w][2][lane] = nan;
    __syncthreads();
    if (w != 0) return;
    le = nan = 0;
    nxt = 0xFFFFFFFFu;
    for (int q = 0; q < 4; ++q) {
        le += part[buf][q][0][lane];
        nxt = min(nxt, part[buf][q][1][lane]);
        nan |= part[buf][q][2][lane];
    }
    float med = key2f(prefix);
    if ((S & 1) == 0) med = __fadd_rn(med, key2f((int)le >= S / 2 + 1 ? prefix : nxt)) / 2.0f;
    if (nan) med = __uint_as_float(0x7FC00000u);
    if (live) out[p] = med;
}

int launch_median(rip_ctx *ctx, const float *stack, int S, size_t npix, float *out) {
    const size_t bytes = (size_t)S * 64 * sizeof(uint32_t);
    const dim3 grid((unsigned)((npix + 63) / 64));
    if (bytes <= 128 * 1024) {
        if (bytes > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute((const void *)seed_median_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)bytes));
        hipLaunchKernelGGL(seed_median_kernel<true>, grid, dim3(256), bytes, ctx->stream, stack, S, npix, out);
    } else {
        hipLaunchKernelGGL(seed_median_kernel<false>, grid, dim3(256), 0, ctx->stream, stack, S, npix, out);
    }
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

int grow_masks(rip_ctx *ctx, const uint8_t grow[32], uint32_t m[4]) {
    m[0] = m[1] = m[2] = m[3] = 0;
    for (int b = 0; b < 32; ++b) {
        const uint32_t bit = 1u << b;
        switch (grow[b]) {
            case 0: break;
            case 1: m[0] |= bit; break;
            case 5: m[1] |= bit; break;
            case 9: m[2] |= bit; break;
            case 25: m[3] |= bit; break;
            default: return rip_fail(ctx, RIP_EINVAL, "stats: growth %d of bit %d is not one of 0, 1, 5, 9, 25", grow[b], b);
        }
    }
    return RIP_OK;
}

}   // namespace

extern "C" {

int rip_stats_l1_diff(rip_ctx *ctx, const uint16_t *cube, int ngrp, int ny, int nx, int ga, int gb, float *out) {
    ctx->stream_dirty = true;
    if (!cube || !out || ny < 1 || nx < 1 || ga < 0 || gb < 0 || ga >= ngrp || gb >= ngrp)
        return rip_fail(ctx, RIP_EINVAL, "stats_l1_diff: bad arguments (groups %d, %d of %d)", ga, gb, ngrp);
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ny * nx;
    hipLaunchKernelGGL(l1_diff_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, cube + n * ga, cube + n * gb,
                       out, n);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

int rip_stats_l2_pack(rip_ctx *ctx, const float *slope, const float *err_read, const float *err_poisson,
                      const uint32_t *pixeldq, int ny, int nx, int nb, const uint8_t grow[32], float *image, float *err,
                      uint8_t *good) {
    ctx->stream_dirty = true;
    if (!slope || !err_read || !err_poisson || !pixeldq || !grow || !image || !err || !good || ny < 1 || nx < 1 || nb < 0 ||
        2 * nb >= ny || 2 * nb >= nx)
        return rip_fail(ctx, RIP_EINVAL, "stats_l2_pack: bad arguments");
    uint32_t m[4];
    int rc = grow_masks(ctx, grow, m);
    if (rc) return rc;
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(l2_pack_kernel, dim3((nx + 255) / 256, ny), dim3(256), 0, ctx->stream, slope, err_read, err_poisson, pixeldq,
                       ny, nx, nb, m[0], m[1], m[2], m[3], image, err, good);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

int rip_stats_reduce(rip_ctx *ctx, int nseeds, const float *diffs, const float *images, const float *errs, const uint8_t *good,
                     const float *ideal, int y0, int nrows, int ny, int nx, int nb, int alias_err, float *out) {
    ctx->stream_dirty = true;
    if (nseeds < 1 || !diffs || !images || !errs || !good || !ideal || !out || nrows < 1 || y0 < 0 || y0 + nrows > ny || nx < 1 ||
        nb < 0)
        return rip_fail(ctx, RIP_EINVAL, "stats_reduce: bad arguments (%d realisations, rows %d+%d of %d)", nseeds, y0, nrows, ny);
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)nrows * nx;
    int rc;
    RIP_HIP(ctx, hipMemcpyAsync(out, ideal, npix * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = launch_median(ctx, diffs, nseeds, npix, out + npix))) return rc;
    if ((rc = launch_median(ctx, errs, nseeds, npix, out + 7 * npix))) return rc;
    if (alias_err)
        RIP_HIP(ctx, hipMemcpyAsync(out + 2 * npix, out + 7 * npix, npix * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    else if ((rc = launch_median(ctx, images, nseeds, npix, out + 2 * npix)))
        return rc;
    hipLaunchKernelGGL(seed_moments_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, images, good, ideal,
                       nseeds, npix, y0, ny, nx, nb, out + 3 * npix, out + 4 * npix, out + 5 * npix, out + 6 * npix);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

}   // extern "C"
