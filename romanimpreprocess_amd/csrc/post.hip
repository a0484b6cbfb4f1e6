// Post-path 2-D reductions of the L2 image: what calibrateimage does between the chain's
// outputs and the L2 file.
//   maskhandling.py:82-117   CombinedMask.build (bit-wise grown mask)                    -> rip_stage_build_mask   (exact)
//   sky.py:20-41             binkxk of the masked image                                  -> rip_stage_bin_mean     (f32 sums; order differs)
//   sky.py:44-97             smooth_mode: percentiles + Gaussian-smoothed histogram       -> rip_stage_select_ranks (exact order statistics)
//                                                                                            rip_stage_gauss_hist   (f64 sums; order differs)
//   sky.py:100-191           medfit: block nan-medians, Legendre model, subtraction       -> rip_stage_select_ranks, rip_stage_legendre2d (exact)
//   gen_cal_image.py:697-712 SLICEOUT endslice                                            -> rip_stage_endslice     (exact)
// Host arrays in and out (these are per-image calls on planes the caller already holds).
#include "rip_common.h"

namespace {

template <typename T>
struct DevBuf {
    rip_ctx *ctx;
    T *p = nullptr;
    explicit DevBuf(rip_ctx *c) : ctx(c) {}
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t n) {
        if (hipMalloc((void **)&p, n * sizeof(T)) != hipSuccess) return rip_fail(ctx, RIP_ENOMEM, "post: %zu bytes", n * sizeof(T));
        return RIP_OK;
    }
    int upload(const T *src, size_t n) {
        int rc = alloc(n);
        if (rc) return rc;
        RIP_HIP(ctx, hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyDefault, ctx->stream));
        return RIP_OK;
    }
    int download(T *dst, size_t n) {
        RIP_HIP(ctx, hipMemcpyAsync(dst, p, n * sizeof(T), hipMemcpyDefault, ctx->stream));
        RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return RIP_OK;
    }
};

// ------------------------------------------------------------------------------------------ mask
// layer(bit) grown by its kernel: 1 = copy, 5 = plus, 9 = 3x3, 25 = 5x5 (zero padded, scipy.signal.convolve mode="same")
__global__ __launch_bounds__(256) void mask_build_kernel(const uint32_t *__restrict__ dq, uint8_t *__restrict__ out, int ny,
                                                         int nx, uint32_t m1, uint32_t m5, uint32_t m9, uint32_t m25) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    uint32_t hit = dq[(size_t)y * nx + x] & (m1 | m5 | m9 | m25);
    const uint32_t grown = m5 | m9 | m25;
    if (!hit && grown) {
        for (int dy = -2; dy <= 2 && !hit; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= ny) continue;
            for (int dx = -2; dx <= 2; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= nx) continue;
                const int ay = dy < 0 ? -dy : dy, ax = dx < 0 ? -dx : dx;
                uint32_t m = m25;
                if (ay <= 1 && ax <= 1) m |= m9;
                if (ay + ax <= 1) m |= m5;
                hit |= dq[(size_t)yy * nx + xx] & m;
            }
        }
    }
    out[(size_t)y * nx + x] = hit ? 1 : 0;
}

// ------------------------------------------------------------------------------------------ endslice
__global__ __launch_bounds__(256) void endslice_kernel(const uint8_t *__restrict__ rdq, int8_t *__restrict__ out, int G, int ny,
                                                       int nx, int nb) {
    const int xa = blockIdx.x * blockDim.x + threadIdx.x, ya = blockIdx.y;
    const int nxa = nx - 2 * nb;
    if (xa >= nxa) return;
    const size_t npix = (size_t)ny * nx, p = (size_t)(ya + nb) * nx + xa + nb;
    int8_t e = -1;
    uint8_t prev = rdq[p];
    for (int iend = 1; iend < G; ++iend) {
        const uint8_t cur = rdq[(size_t)iend * npix + p];
        if ((cur & ~prev) & DQ_SATURATED) e = (int8_t)(iend - 1);
        prev = cur;
    }
    out[(size_t)ya * nxa + xa] = e;
}

// ------------------------------------------------------------------------------------------ binned mean
// mean over k x k blocks of where(mask, nan, arr): NaN as soon as one pixel of the block is masked or NaN
__global__ __launch_bounds__(256) void bin_mean_kernel(const float *__restrict__ arr, const uint8_t *__restrict__ mask,
                                                       float *__restrict__ out, int nx, int nyo, int nxo, int k) {
    const int xo = blockIdx.x * blockDim.x + threadIdx.x, yo = blockIdx.y;
    if (xo >= nxo) return;
    float s = 0.0f;
    for (int a = 0; a < k; ++a) {
        float t = 0.0f;
        for (int b = 0; b < k; ++b) {
            const size_t p = (size_t)(yo * k + a) * nx + (size_t)xo * k + b;
            const float v = (mask && mask[p]) ? NAN : arr[p];
            t = t + v;
        }
        s = s + t;
    }
    out[(size_t)yo * nxo + xo] = s / (float)(k * k);
}

// ------------------------------------------------------------------------------------------ order statistics
// Blocks: nby x nbx rectangles of ky x kx pixels starting at (y0, x0) of an (ny, nx) f32 image.  NaNs are ignored.
// Three radix levels (11 + 11 + 10 bits) on the order-preserving integer image of the float.
#define PS_BINS 2048
__device__ __forceinline__ uint32_t ps_key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ps_unkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ int ps_shift(int level) { return level == 0 ? 21 : (level == 1 ? 10 : 0); }
__device__ __forceinline__ int ps_nbits(int level) { return level == 2 ? 10 : 11; }

struct PsGeom {
    int ny, nx, y0, x0, ky, kx, nby, nbx;
};

// level < 0: count the valid (non-NaN) elements of every block into hist[blk * PS_BINS]
__global__ __launch_bounds__(256) void ps_hist_kernel(const float *__restrict__ arr, PsGeom g, const uint32_t *__restrict__ prefix,
                                                      uint32_t *__restrict__ hist, int level) {
    __shared__ uint32_t h[PS_BINS];
    const int blk = blockIdx.y, by = blk / g.nbx, bx = blk % g.nbx;
    for (int i = threadIdx.x; i < PS_BINS; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const size_t n = (size_t)g.ky * g.kx;
    const int shift = level >= 0 ? ps_shift(level) : 0;
    const uint32_t mask = level >= 0 ? (1u << ps_nbits(level)) - 1u : 0u;
    const int above = level >= 0 ? shift + ps_nbits(level) : 32;
    const uint32_t pre = (level > 0) ? prefix[blk] : 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int yy = g.y0 + by * g.ky + (int)(i / g.kx), xx = g.x0 + bx * g.kx + (int)(i % g.kx);
        const float v = arr[(size_t)yy * g.nx + xx];
        if (v != v) continue;
        if (level < 0) {
            atomicAdd(&h[0], 1u);
        } else {
            const uint32_t key = ps_key(v);
            if (above < 32 && ((key ^ pre) >> above) != 0) continue;
            atomicAdd(&h[(key >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PS_BINS; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[(size_t)blk * PS_BINS + i], h[i]);
}

// one thread per block: walk the histogram to the bin holding `rank`, extend the key prefix, rebase the rank
__global__ void ps_scan_kernel(uint32_t *__restrict__ hist, uint32_t *__restrict__ prefix, unsigned long long *__restrict__ rank,
                               int nblk, int level) {
    const int blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk >= nblk) return;
    uint32_t *h = hist + (size_t)blk * PS_BINS;
    unsigned long long r = rank[blk], cum = 0;
    const int nb = 1 << ps_nbits(level);
    int b = 0;
    for (; b < nb - 1; ++b) {
        if (cum + h[b] > r) break;
        cum += h[b];
    }
    rank[blk] = r - cum;
    prefix[blk] = (level == 0 ? 0u : prefix[blk]) | ((uint32_t)b << ps_shift(level));
    for (int i = 0; i < PS_BINS; ++i) h[i] = 0;
}

__global__ void ps_finish_kernel(const uint32_t *__restrict__ prefix, float *__restrict__ out, int nblk) {
    const int blk = blockIdx.x * blockDim.x + threadIdx.x;
    if (blk < nblk) out[blk] = ps_unkey(prefix[blk]);
}

// ------------------------------------------------------------------------------------------ smoothed histogram
// out[i] = sum over the non-NaN x of exp(-0.5 ((z[i] - x) / scale)^2), i < nz <= 32, in f64
__global__ __launch_bounds__(256) void gauss_hist_kernel(const float *__restrict__ arr, size_t n, const double *__restrict__ z, int nz,
                                                         double scale, double *__restrict__ out) {
    __shared__ double red[256];
    double acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const float xv = arr[p];
        if (xv != xv) continue;
        const double x = (double)xv;
        for (int i = 0; i < nz; ++i) {
            const double u = (z[i] - x) / scale;
            acc[i] += exp(-0.5 * (u * u));
        }
    }
    for (int i = 0; i < nz; ++i) {
        red[threadIdx.x] = acc[i];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) atomicAdd(&out[i], red[0]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ Legendre model
// model[y,x] = sum_k coef[k] * (LPY[j_k][y] * LPX[i_k][x]) accumulated in f64 in the order k = 0.. (sky.py:183-189),
// cast to f32; arr -= model when subtract.  (i_k, j_k): i = 0..order, j = 0..order-i.
__global__ __launch_bounds__(256) void legendre2d_kernel(float *__restrict__ arr, float *__restrict__ model_out,
                                                         const double *__restrict__ LPX, const double *__restrict__ LPY,
                                                         const double *__restrict__ coef, int order, int ny, int nx, int subtract) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    double m = 0.0;
    int k = 0;
    for (int i = 0; i <= order; ++i)
        for (int j = 0; j <= order - i; ++j) {
            const double o = LPY[(size_t)j * ny + y] * LPX[(size_t)i * nx + x];
            m = m + coef[k] * o;
            ++k;
        }
    const float mf = (float)m;
    const size_t p = (size_t)y * nx + x;
    if (model_out) model_out[p] = mf;
    if (subtract) arr[p] = arr[p] - mf;
}

}  // namespace

// ============================================================================================ C-ABI

int rip_stage_build_mask(rip_ctx *ctx, const uint32_t *dq, int ny, int nx, const uint8_t grow[32], uint8_t *mask) {
    if (!dq || !grow || !mask || ny < 1 || nx < 1) return rip_fail(ctx, RIP_EINVAL, "build_mask: bad arguments");
    uint32_t m1 = 0, m5 = 0, m9 = 0, m25 = 0;
    for (int b = 0; b < 32; ++b) {
        const uint32_t bit = 1u << b;
        switch (grow[b]) {
            case 0: break;
            case 1: m1 |= bit; break;
            case 5: m5 |= bit; break;
            case 9: m9 |= bit; break;
            case 25: m25 |= bit; break;
            default: return rip_fail(ctx, RIP_EINVAL, "build_mask: growth %d of bit %d is not one of 0, 1, 5, 9, 25", grow[b], b);
        }
    }
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ny * nx;
    DevBuf<uint32_t> d(ctx);
    DevBuf<uint8_t> o(ctx);
    int rc;
    if ((rc = d.upload(dq, n)) || (rc = o.alloc(n))) return rc;
    hipLaunchKernelGGL(mask_build_kernel, dim3((nx + 255) / 256, ny), dim3(256), 0, ctx->stream, d.p, o.p, ny, nx, m1, m5, m9, m25);
    RIP_HIP(ctx, hipGetLastError());
    return o.download(mask, n);
}

int rip_stage_endslice(rip_ctx *ctx, const uint8_t *rdq, int ngrp, int ny, int nx, int nb, int8_t *out) {
    if (!rdq || !out || ngrp < 1 || nb < 0 || ny <= 2 * nb || nx <= 2 * nb) return rip_fail(ctx, RIP_EINVAL, "endslice: bad arguments");
    if (ngrp >= 128) return rip_fail(ctx, RIP_EINVAL, "too many groups");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ngrp * ny * nx, na = (size_t)(ny - 2 * nb) * (nx - 2 * nb);
    DevBuf<uint8_t> d(ctx);
    DevBuf<int8_t> o(ctx);
    int rc;
    if ((rc = d.upload(rdq, n)) || (rc = o.alloc(na))) return rc;
    hipLaunchKernelGGL(endslice_kernel, dim3((nx - 2 * nb + 255) / 256, ny - 2 * nb), dim3(256), 0, ctx->stream, d.p, o.p, ngrp, ny, nx,
                       nb);
    RIP_HIP(ctx, hipGetLastError());
    return o.download(out, na);
}

int rip_stage_bin_mean(rip_ctx *ctx, const float *arr, const uint8_t *mask, int ny, int nx, int k, float *out) {
    if (!arr || !out || k < 1 || ny < k || nx < k) return rip_fail(ctx, RIP_EINVAL, "bin_mean: bad arguments");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const int nyo = ny / k, nxo = nx / k;
    const size_t n = (size_t)ny * nx, no = (size_t)nyo * nxo;
    DevBuf<float> d(ctx), o(ctx);
    DevBuf<uint8_t> m(ctx);
    int rc;
    if ((rc = d.upload(arr, n)) || (rc = o.alloc(no))) return rc;
    if (mask && (rc = m.upload(mask, n))) return rc;
    hipLaunchKernelGGL(bin_mean_kernel, dim3((nxo + 255) / 256, nyo), dim3(256), 0, ctx->stream, d.p, mask ? m.p : nullptr, o.p, nx, nyo,
                       nxo, k);
    RIP_HIP(ctx, hipGetLastError());
    return o.download(out, no);
}

// counts[blk] = number of non-NaN elements; vals[blk * nranks + r] = element of 0-based rank ranks[blk * nranks + r]
// among them in ascending order (NaN when the rank is out of range).  ranks == NULL: only the counts.
int rip_stage_select_ranks(rip_ctx *ctx, const float *arr, int ny, int nx, int y0, int x0, int ky, int kx, int nby, int nbx,
                           int nranks, const int64_t *ranks, int64_t *counts, float *vals) {
    if (!arr || ky < 1 || kx < 1 || nby < 1 || nbx < 1 || y0 < 0 || x0 < 0 || y0 + (long)nby * ky > ny || x0 + (long)nbx * kx > nx ||
        nranks < 0 || (nranks > 0 && (!ranks || !vals)))
        return rip_fail(ctx, RIP_EINVAL, "select_ranks: bad geometry or arguments");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const int nblk = nby * nbx;
    const size_t n = (size_t)ny * nx;
    DevBuf<float> d(ctx), o(ctx);
    DevBuf<uint32_t> hist(ctx), prefix(ctx);
    DevBuf<unsigned long long> rk(ctx);
    int rc;
    if ((rc = d.upload(arr, n)) || (rc = hist.alloc((size_t)nblk * PS_BINS)) || (rc = prefix.alloc(nblk)) || (rc = rk.alloc(nblk)) ||
        (rc = o.alloc(nblk)))
        return rc;
    const PsGeom g{ny, nx, y0, x0, ky, kx, nby, nbx};
    const size_t per = (size_t)ky * kx;
    unsigned chunks = (unsigned)((per + 256 * 16 - 1) / (256 * 16));
    if (chunks < 1) chunks = 1;
    if (chunks > 1024) chunks = 1024;
    RIP_HIP(ctx, hipMemsetAsync(hist.p, 0, (size_t)nblk * PS_BINS * 4, ctx->stream));
    hipLaunchKernelGGL(ps_hist_kernel, dim3(chunks, nblk), dim3(256), 0, ctx->stream, d.p, g, prefix.p, hist.p, -1);
    std::vector<uint32_t> hh((size_t)nblk * PS_BINS);
    RIP_HIP(ctx, hipMemcpyAsync(hh.data(), hist.p, hh.size() * 4, hipMemcpyDefault, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> cnt(nblk);
    for (int b = 0; b < nblk; ++b) cnt[b] = hh[(size_t)b * PS_BINS];
    if (counts)
        for (int b = 0; b < nblk; ++b) counts[b] = cnt[b];
    std::vector<unsigned long long> r(nblk);
    std::vector<float> v(nblk);
    for (int q = 0; q < nranks; ++q) {
        for (int b = 0; b < nblk; ++b) {
            const int64_t want = ranks[(size_t)b * nranks + q];
            r[b] = (want >= 0 && want < cnt[b]) ? (unsigned long long)want : 0ull;
        }
        RIP_HIP(ctx, hipMemcpyAsync(rk.p, r.data(), (size_t)nblk * 8, hipMemcpyDefault, ctx->stream));
        RIP_HIP(ctx, hipMemsetAsync(hist.p, 0, (size_t)nblk * PS_BINS * 4, ctx->stream));
        for (int level = 0; level < 3; ++level) {
            hipLaunchKernelGGL(ps_hist_kernel, dim3(chunks, nblk), dim3(256), 0, ctx->stream, d.p, g, prefix.p, hist.p, level);
            hipLaunchKernelGGL(ps_scan_kernel, dim3((nblk + 63) / 64), dim3(64), 0, ctx->stream, hist.p, prefix.p, rk.p, nblk, level);
        }
        hipLaunchKernelGGL(ps_finish_kernel, dim3((nblk + 63) / 64), dim3(64), 0, ctx->stream, prefix.p, o.p, nblk);
        RIP_HIP(ctx, hipGetLastError());
        if ((rc = o.download(v.data(), nblk))) return rc;
        for (int b = 0; b < nblk; ++b) {
            const int64_t want = ranks[(size_t)b * nranks + q];
            vals[(size_t)b * nranks + q] = (want >= 0 && want < cnt[b]) ? v[b] : NAN;
        }
    }
    return RIP_OK;
}

int rip_stage_gauss_hist(rip_ctx *ctx, const float *arr, int64_t n, const double *z, int nz, double scale, double *out) {
    if (!arr || !z || !out || n < 1 || nz < 1 || nz > 32 || !(scale > 0.0)) return rip_fail(ctx, RIP_EINVAL, "gauss_hist: bad arguments");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf<float> d(ctx);
    DevBuf<double> dz(ctx), o(ctx);
    int rc;
    if ((rc = d.upload(arr, (size_t)n)) || (rc = dz.upload(z, nz)) || (rc = o.alloc(nz))) return rc;
    RIP_HIP(ctx, hipMemsetAsync(o.p, 0, (size_t)nz * 8, ctx->stream));
    unsigned blocks = (unsigned)(((size_t)n + 256 * 8 - 1) / (256 * 8));
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gauss_hist_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d.p, (size_t)n, dz.p, nz, scale, o.p);
    RIP_HIP(ctx, hipGetLastError());
    return o.download(out, nz);
}

int rip_stage_legendre2d(rip_ctx *ctx, float *arr, int ny, int nx, int order, const double *LPX, const double *LPY, const double *coef,
                         int subtract, float *model_out) {
    if (!LPX || !LPY || !coef || order < 0 || order > 8 || ny < 1 || nx < 1 || (subtract && !arr) || (!subtract && !model_out))
        return rip_fail(ctx, RIP_EINVAL, "legendre2d: bad arguments");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)ny * nx;
    const int nc = (order + 1) * (order + 2) / 2;
    DevBuf<float> d(ctx), mo(ctx);
    DevBuf<double> lx(ctx), ly(ctx), c(ctx);
    int rc;
    if ((rc = lx.upload(LPX, (size_t)(order + 1) * nx)) || (rc = ly.upload(LPY, (size_t)(order + 1) * ny)) || (rc = c.upload(coef, nc)))
        return rc;
    if (subtract && (rc = d.upload(arr, n))) return rc;
    if (model_out && (rc = mo.alloc(n))) return rc;
    hipLaunchKernelGGL(legendre2d_kernel, dim3((nx + 255) / 256, ny), dim3(256), 0, ctx->stream, subtract ? d.p : nullptr,
                       model_out ? mo.p : nullptr, lx.p, ly.p, c.p, order, ny, nx, subtract);
    RIP_HIP(ctx, hipGetLastError());
    if (subtract && (rc = d.download(arr, n))) return rc;
    if (model_out && (rc = mo.download(model_out, n))) return rc;
    return RIP_OK;
}
