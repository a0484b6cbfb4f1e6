// Small layout / per-SCA preparation kernels.
//
//   embed_kernel        (nplanes, ny-2nb, nx-2nb) -> (nplanes, ny, nx) with a zero border, so that the
//                       biascorr and ipc4d arrays of a CALDIR are addressed in
//                       full-frame coordinates with 16 KiB-aligned rows.
//   flat_prepare_kernel utils/flatutils.py:46-69 (pad with 1, NO_FLAT_FIELD / NO_GAIN_VALUE flags, clips)
//   flat_area_kernel    L1_to_L2/gen_cal_image.py:622  flat = f32(flat / AreaFactor)
#include "rip_common.h"

template <typename T>
__global__ void embed_kernel(const T *__restrict__ src, T *__restrict__ dst, int nplanes, int ny, int nx, int nb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)ny * nx;
    if (i >= plane * nplanes) return;
    const int pl = (int)(i / plane);
    const size_t r = i % plane;
    const int y = (int)(r / nx), x = (int)(r % nx);
    const int nya = ny - 2 * nb, nxa = nx - 2 * nb;
    T v = (T)0;
    if (y >= nb && y < ny - nb && x >= nb && x < nx - nb) v = src[((size_t)pl * nya + (y - nb)) * nxa + (x - nb)];
    dst[i] = v;
}

int rip_launch_embed(rip_ctx *ctx, const void *src, void *dst, int nplanes, int ny, int nx, int nb, int elem_size) {
    const size_t n = (size_t)ny * nx * nplanes;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (elem_size == 8)
        hipLaunchKernelGGL(embed_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, (const double *)src,
                           (double *)dst, nplanes, ny, nx, nb);
    else
        hipLaunchKernelGGL(embed_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, (const float *)src,
                           (float *)dst, nplanes, ny, nx, nb);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

template <typename GT>
__global__ void flat_prepare_kernel(const float *__restrict__ flat, const GT *__restrict__ gain, int ny, int nx, int nb,
                                    float *__restrict__ flat_padded, GT *__restrict__ gain_clipped,
                                    uint32_t *__restrict__ flags, int with_gain, int clip_gain) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ny * nx) return;
    const int y = (int)(i / nx), x = (int)(i % nx);
    const bool act = (y >= nb && y < ny - nb && x >= nb && x < nx - nb);
    float f = act ? flat[i] : 1.0f;
    uint32_t fl = (f < 0.1f || f > 10.0f) ? DQ_NO_FLAT_FIELD : 0u;
    f = f < 0.1f ? 0.1f : (f > 10.0f ? 10.0f : f);
    if (with_gain) {
        GT g = act ? gain[i] : (GT)1;
        if (act && g <= (GT)0.1) fl |= DQ_NO_GAIN_VALUE;
        if (clip_gain) g = g < (GT)0.1 ? (GT)0.1 : g;
        gain_clipped[i] = g;
    }
    flat_padded[i] = f;
    if (flags) flags[i] = fl;
}

int rip_launch_flat_prepare(rip_ctx *ctx, const float *flat, const void *gain, int g_dtype, int ny, int nx, int nb,
                            float *flat_padded, void *gain_clipped, uint32_t *flags, int with_gain) {
    const size_t n = (size_t)ny * nx;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const int use_gain = with_gain != 0, clip_gain = (with_gain == 1);
    if (g_dtype == RIP_F64)
        hipLaunchKernelGGL(flat_prepare_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, flat,
                           (const double *)gain, ny, nx, nb, flat_padded, (double *)gain_clipped, flags, use_gain,
                           clip_gain);
    else
        hipLaunchKernelGGL(flat_prepare_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, flat,
                           (const float *)gain, ny, nx, nb, flat_padded, (float *)gain_clipped, flags, use_gain,
                           clip_gain);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

__global__ void flat_area_kernel(const float *__restrict__ flat_dn, const double *__restrict__ area,
                                 float *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)((double)flat_dn[i] / area[i]);
}

int rip_launch_flat_area(rip_ctx *ctx, const float *flat_dn, const double *area, float *out, size_t n) {
    hipLaunchKernelGGL(flat_area_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, flat_dn, area,
                       out, n);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}


// ---------------------------------------------------------------------------------------------
// dq-init + saturation flagging (gen_cal_image.py:148-185; behaviour restated in
// L1_to_L2/gen_cal_image.py:flag_saturation -- PARITY UNPINNED, stcal's source is not available).
//   sat(g) = OR over the 3x3 neighbourhood (inside the frame) of [data[g] >= threshold and the pixel is checked],
//   g >= skip; sticky forward in g; set on the `backup` preceding groups (never below `skip`).
//   groupdq = gdq_in | SATURATED(sat) | DO_NOT_USE on group 0 when dnu_first; pixeldq = pdq_in | SATURATED if any.
// Two passes, one thread per pixel: (1) bit g of a per-pixel word = that pixel's own resultant g exceeds its own
// threshold (the cube is read once, coalesced); (2) OR of the nine neighbours' words, the time logic on the bits, flags.
// four consecutive pixels per thread (nx is a multiple of 4): 8-byte loads of the u16 cube, 4-byte stores of the flags
struct SatDilution {
    double f[64];  // per group: mean(read_pattern[g]) / read_pattern[g][-1], or 1
};

template <typename T>
__global__ __launch_bounds__(256) void sat_exceed_kernel(const T *__restrict__ data, const float *__restrict__ thr,
                                                         const uint32_t *__restrict__ sat_dq, uint64_t *__restrict__ ex,
                                                         int G, size_t npix, int skip, const SatDilution dil, int use_dil) {
    const size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (p >= npix) return;
    const float4 th = *reinterpret_cast<const float4 *>(thr + p);
    const float tv[4] = {th.x, th.y, th.z, th.w};
    uint4 sq = {0u, 0u, 0u, 0u};
    if (sat_dq) sq = *reinterpret_cast<const uint4 *>(sat_dq + p);
    const uint32_t sv[4] = {sq.x, sq.y, sq.z, sq.w};
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)  // NaN / inf / NO_SAT_CHECK -> never exceeded
        t[i] = (!(fabsf(tv[i]) <= 3.4e38f) || (sv[i] & (1u << 21)) != 0) ? INFINITY : tv[i];
    uint64_t m[4] = {0, 0, 0, 0};
    for (int g = skip; g < G; ++g) {
        float d[4];
        if constexpr (sizeof(T) == 2) {
            const uint2 w = *reinterpret_cast<const uint2 *>(data + (size_t)g * npix + p);
            d[0] = (float)(w.x & 0xffffu), d[1] = (float)(w.x >> 16), d[2] = (float)(w.y & 0xffffu), d[3] = (float)(w.y >> 16);
        } else {
            const float4 w = *reinterpret_cast<const float4 *>(data + (size_t)g * npix + p);
            d[0] = w.x, d[1] = w.y, d[2] = w.z, d[3] = w.w;
        }
        if (use_dil) {  // threshold of a group of several reads: f64(threshold) * mean(reads) / last read
            const double f = dil.f[g];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if ((double)d[i] >= (double)t[i] * f) m[i] |= 1ull << g;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (d[i] >= t[i]) m[i] |= 1ull << g;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) ex[p + i] = m[i];
}

__global__ __launch_bounds__(256) void sat_flags_kernel(const uint64_t *__restrict__ ex, const uint8_t *__restrict__ gdq_in,
                                                        const uint32_t *__restrict__ pdq_in, uint8_t *__restrict__ gdq_out,
                                                        uint32_t *__restrict__ pdq_out, int G, int ny, int nx, int backup,
                                                        int skip, int dnu_first) {
    const int x0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y;
    if (x0 >= nx) return;
    const size_t npix = (size_t)ny * nx, p = (size_t)y * nx + x0;
    // column-wise OR over the three rows for columns x0-1 .. x0+4, then the 3-wide OR along the row
    uint64_t c[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= ny) continue;
        const uint64_t *row = ex + (size_t)yy * nx;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int xx = x0 - 1 + i;
            if (xx >= 0 && xx < nx) c[i] |= row[xx];
        }
    }
    const uint64_t full = (G >= 64) ? ~0ull : ((1ull << G) - 1ull);
    uint64_t sat[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = c[i] | c[i + 1] | c[i + 2];
        uint64_t s = m ? (~((m & (~m + 1ull)) - 1ull)) & full : 0ull;  // every group from the first exceeding one on
        for (int b = 0; b < backup; ++b) s |= (s >> 1);
        sat[i] = s & ~((1ull << skip) - 1ull);  // never the first `skip` groups
    }
    for (int g = 0; g < G; ++g) {
        uint32_t v = gdq_in ? *reinterpret_cast<const uint32_t *>(gdq_in + (size_t)g * npix + p) : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((sat[i] >> g) & 1ull) v |= DQ_SATURATED << (8 * i);
        if (g == 0 && dnu_first) v |= 0x01010101u * DQ_DO_NOT_USE;
        *reinterpret_cast<uint32_t *>(gdq_out + (size_t)g * npix + p) = v;
    }
    uint4 pd = {0u, 0u, 0u, 0u};
    if (pdq_in) pd = *reinterpret_cast<const uint4 *>(pdq_in + p);
    pd.x |= sat[0] ? DQ_SATURATED : 0u;
    pd.y |= sat[1] ? DQ_SATURATED : 0u;
    pd.z |= sat[2] ? DQ_SATURATED : 0u;
    pd.w |= sat[3] ? DQ_SATURATED : 0u;
    *reinterpret_cast<uint4 *>(pdq_out + p) = pd;
}

// flag word of the wave-specialised fused kernel: linearity dq with the words its finish step ORs into pixeldq anyway
__global__ void merge_dq_kernel(const uint32_t *__restrict__ lin_dq, const uint32_t *__restrict__ flat_flags,
                                const uint32_t *__restrict__ dark_dq, uint32_t *__restrict__ out, int ny, int nx, int nb,
                                uint32_t *__restrict__ clash) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)ny * nx) return;
    const int y = (int)(i / nx), x = (int)(i % nx);
    const bool act = y >= nb && y < ny - nb && x >= nb && x < nx - nb;
    uint32_t add = flat_flags ? flat_flags[i] : 0u;
    if (dark_dq && act) add |= dark_dq[i];   // gen_cal_image.py:213-229: the dark step works on the active region
    out[i] = lin_dq[i] | add;
    if (add & (DQ_NO_LIN_CORR | DQ_REFERENCE_PIXEL)) atomicOr(clash, 1u);
}

int rip_launch_merge_dq(rip_ctx *ctx, const uint32_t *lin_dq, const uint32_t *flat_flags, const uint32_t *dark_dq, uint32_t *out, int ny,
                        int nx, int nb, uint32_t *d_clash) {
    const size_t n = (size_t)ny * nx;
    hipLaunchKernelGGL(merge_dq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, lin_dq, flat_flags, dark_dq, out, ny,
                       nx, nb, d_clash);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

int rip_launch_satflag(rip_ctx *ctx, const void *data, int data_dtype, const float *thr, const uint32_t *sat_dq,
                       const uint8_t *gdq_in, const uint32_t *pdq_in, uint8_t *gdq_out, uint32_t *pdq_out, int G, int ny,
                       int nx, int backup, int skip_firstn, int dnu_first, const double *dilution, hipStream_t stream) {
    hipStream_t st = stream ? stream : ctx->stream;
    if (G < 1 || G > 64 || backup < 0 || skip_firstn < 0 || skip_firstn > G)
        return rip_fail(ctx, RIP_EINVAL, "saturation flagging: bad group / backup / skip arguments");
    if (nx % 4) return rip_fail(ctx, RIP_EINVAL, "saturation flagging: nx=%d is not a multiple of 4", nx);
    const size_t npix = (size_t)ny * nx;
    uint64_t *ex = (uint64_t *)rip_ws(ctx, 9, npix * 8);
    if (!ex) return RIP_ENOMEM;
    const dim3 g1((unsigned)((npix / 4 + 255) / 256)), block(256);
    SatDilution dil;
    for (int g = 0; g < 64; ++g) dil.f[g] = (dilution && g < G) ? dilution[g] : 1.0;
    const int use_dil = dilution ? 1 : 0;
    if (data_dtype == RIP_U16)
        hipLaunchKernelGGL(sat_exceed_kernel<uint16_t>, g1, block, 0, st, (const uint16_t *)data, thr, sat_dq, ex, G, npix,
                           skip_firstn, dil, use_dil);
    else
        hipLaunchKernelGGL(sat_exceed_kernel<float>, g1, block, 0, st, (const float *)data, thr, sat_dq, ex, G, npix,
                           skip_firstn, dil, use_dil);
    hipLaunchKernelGGL(sat_flags_kernel, dim3((nx / 4 + 255) / 256, ny), block, 0, st, ex, gdq_in, pdq_in, gdq_out, pdq_out, G,
                       ny, nx, backup, skip_firstn, dnu_first);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

// bytes[0 .. n) |= bit (the DO_NOT_USE flag of an excluded first group on the library's device copy of groupdq)
__global__ __launch_bounds__(256) void or_bytes_kernel(uint32_t *__restrict__ w, size_t n4, uint32_t bits) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) w[i] |= bits;
}
int rip_launch_or_bytes(rip_ctx *ctx, uint8_t *bytes, size_t n, uint8_t bit, hipStream_t stream) {
    if ((n & 3) || ((uintptr_t)bytes & 3)) return rip_fail(ctx, RIP_EINVAL, "or_bytes: plane of %zu bytes is not a whole number of words", n);
    hipStream_t st = stream ? stream : ctx->stream;
    const uint32_t b = bit;
    hipLaunchKernelGGL(or_bytes_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, (uint32_t *)bytes, n / 4,
                       b | (b << 8) | (b << 16) | (b << 24));
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
