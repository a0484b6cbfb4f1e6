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
// One thread per pixel.  sat(g) = OR over the 3x3 neighbourhood (inside the frame) of
// [data[g] >= threshold and the pixel is checked], g >= skip; sticky forward in g; set on the `backup` preceding groups
// (never below `skip`).  groupdq = gdq_in | SATURATED(sat) | DO_NOT_USE on group 0 when dnu_first;
// pixeldq = pdq_in | SATURATED where any group is flagged.
template <typename T>
__global__ __launch_bounds__(256) void satflag_kernel(const T *__restrict__ data, const float *__restrict__ thr,
                                                      const uint32_t *__restrict__ sat_dq,
                                                      const uint8_t *__restrict__ gdq_in,
                                                      const uint32_t *__restrict__ pdq_in, uint8_t *__restrict__ gdq_out,
                                                      uint32_t *__restrict__ pdq_out, int G, int ny, int nx, int backup,
                                                      int skip, int dnu_first) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= nx) return;
    const size_t npix = (size_t)ny * nx, p = (size_t)y * nx + x;
    // thresholds of the up to nine neighbours; +inf where the neighbour is outside the frame or not checked
    float t[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        float v = INFINITY;
        if (yy >= 0 && yy < ny && xx >= 0 && xx < nx) {
            const size_t q = (size_t)yy * nx + xx;
            const float th = thr[q];
            const bool nocheck = !(fabsf(th) <= 3.4e38f) || (sat_dq && (sat_dq[q] & (1u << 21)) != 0);  // NO_SAT_CHECK
            v = nocheck ? INFINITY : th;
        }
        t[k] = v;
    }
    uint64_t sat = 0;  // bit g
    for (int g = skip; g < G; ++g) {
        bool s = false;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (t[k] == INFINITY) continue;
            const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
            const float d = (float)data[(size_t)g * npix + (size_t)yy * nx + xx];
            s = s || (d >= t[k]);
        }
        if (s) sat |= (~0ull) << g;  // sticky for the later groups
    }
    sat &= (G >= 64) ? ~0ull : ((1ull << G) - 1ull);
    for (int b = 0; b < backup; ++b) sat |= (sat >> 1);
    sat &= ~((1ull << skip) - 1ull);  // never the first `skip` groups
    for (int g = 0; g < G; ++g) {
        uint8_t v = gdq_in ? gdq_in[(size_t)g * npix + p] : (uint8_t)0;
        if ((sat >> g) & 1ull) v |= (uint8_t)DQ_SATURATED;
        if (g == 0 && dnu_first) v |= (uint8_t)DQ_DO_NOT_USE;
        gdq_out[(size_t)g * npix + p] = v;
    }
    pdq_out[p] = (pdq_in ? pdq_in[p] : 0u) | (sat ? DQ_SATURATED : 0u);
}

int rip_launch_satflag(rip_ctx *ctx, const void *data, int data_dtype, const float *thr, const uint32_t *sat_dq,
                       const uint8_t *gdq_in, const uint32_t *pdq_in, uint8_t *gdq_out, uint32_t *pdq_out, int G, int ny,
                       int nx, int backup, int skip_firstn, int dnu_first) {
    if (G < 1 || G > 64 || backup < 0 || skip_firstn < 0 || skip_firstn > G)
        return rip_fail(ctx, RIP_EINVAL, "saturation flagging: bad group / backup / skip arguments");
    const dim3 grid((nx + 255) / 256, ny), block(256);
    if (data_dtype == RIP_U16)
        hipLaunchKernelGGL(satflag_kernel<uint16_t>, grid, block, 0, ctx->stream, (const uint16_t *)data, thr, sat_dq, gdq_in,
                           pdq_in, gdq_out, pdq_out, G, ny, nx, backup, skip_firstn, dnu_first);
    else
        hipLaunchKernelGGL(satflag_kernel<float>, grid, block, 0, ctx->stream, (const float *)data, thr, sat_dq, gdq_in, pdq_in,
                           gdq_out, pdq_out, G, ny, nx, backup, skip_firstn, dnu_first);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
