// Noise layers: the arithmetic gen_noise_image.make_noise_cube applies to a Level-1 cube before it sends
// it through the chain again (gen_noise_image.py:120-134), per group k and active pixel:
//     im = f32( f64(normal) * (f64(read) / sqrt(f64(N_k))) )         numpy: f32 array *= (f32 array / np.float64 scalar)
//     r  = f32(data) + im ;  data' = u16( rint( clip(r, 0, 65535) ) )   np.round = round half to even
// The standard normal deviates come from the caller (the reference draws them with galsim.GaussianDeviate; which generator
// is used is the driver's business) -- or, normals == NULL, from a counter-based generator on the device (Philox-4x32-10 +
// Box-Muller, keyed by seed, layer, group and pixel), so that a layer is reproducible and needs no host random numbers.
// Border pixels (reference pixels) pass through unchanged.  Host arrays in and out.  Exact given the normals.
#include "rip_common.h"

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
    NZ_HIP(hipMemcpyAsync(d_cube, cube, (size_t)ngrp * npix * 2, hipMemcpyHostToDevice, ctx->stream));
    NZ_HIP(hipMemcpyAsync(d_read, read_noise, npix * 4, hipMemcpyHostToDevice, ctx->stream));
    NZ_HIP(hipMemcpyAsync(d_rs, rs, sizeof(double) * ngrp, hipMemcpyHostToDevice, ctx->stream));
    if (normals) {
        NZ_HIP(hipMalloc(&d_nrm, (size_t)ngrp * nact * 4));
        NZ_HIP(hipMemcpyAsync(d_nrm, normals, (size_t)ngrp * nact * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    hipLaunchKernelGGL(noise_inject_kernel, dim3((nx + 255) / 256, ny, ngrp), dim3(256), 0, ctx->stream, (const uint16_t *)d_cube,
                       (const float *)d_nrm, (const float *)d_read, (const double *)d_rs, ngrp, ny, nx, nb, seed, layer,
                       (uint16_t *)d_out);
    NZ_HIP(hipGetLastError());
    NZ_HIP(hipMemcpyAsync(out, d_out, (size_t)ngrp * npix * 2, hipMemcpyDeviceToHost, ctx->stream));
    NZ_HIP(hipStreamSynchronize(ctx->stream));
#undef NZ_HIP
    done();
    return RIP_OK;
}
