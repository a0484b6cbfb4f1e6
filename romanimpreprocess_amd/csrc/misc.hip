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
