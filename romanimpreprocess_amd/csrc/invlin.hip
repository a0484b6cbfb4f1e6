// Inverse linearity (the simulation-side inverse of A4): ipc_linearity.invlinearity
// (ipc_linearity.py:347-394) -- the step the reference documents as the slowest of its simulation -> Level 1 workflow.
// (per-pixel arithmetic: invlin_device.h)
// 24 bisection steps on z in (-1, 1); each evaluates the Legendre series of ipc_linearity._lin (:192-231) WITHOUT the linear
// extrapolation branch, in numpy's operation order and dtypes:
//     phi (f32) += coefs[L] (f32) * poly (ZT)          one rounding per operation; with ZT = f64 the sum is rounded back to f32
//     poly_next = c1_L * z * poly - c2_L * poly_prev    c1, c2 Python floats: cast to f32 when z is f32, exact f64 otherwise
//     z += phi < Slin ? 2^-j : -2^-j
// then S = Smin + (Smax - Smin) / 2 * (1 + z).  One thread per pixel; coefficient planes are read once per pixel (registers).
// Host arrays in and out, like the other stage entries.  Exact.
#include "rip_common.h"
#include "invlin_device.h"

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
        if (hipMalloc((void **)&p, n * sizeof(T)) != hipSuccess) return rip_fail(ctx, RIP_ENOMEM, "invlinearity: %zu bytes", n * sizeof(T));
        return RIP_OK;
    }
    int upload(const void *src, size_t n) {
        int rc = alloc(n);
        if (rc) return rc;
        RIP_HIP(ctx, hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
        return RIP_OK;
    }
};

template <typename ZT, int NP>
__global__ __launch_bounds__(256) void invlin_kernel(const ZT *__restrict__ slin, const float *__restrict__ coefs,
                                                     const float *__restrict__ smin, const float *__restrict__ smax, size_t npix,
                                                     ZT *__restrict__ out, uint8_t *__restrict__ exflag) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    float c[NP];
#pragma unroll
    for (int L = 0; L < NP; ++L) c[L] = coefs[(size_t)L * npix + i];
    bool ex;
    out[i] = rip_invlin_pixel<ZT, NP>(slin[i], c, smin[i], smax[i], ex);
    if (exflag) exflag[i] = ex ? 1 : 0;
}

template <typename ZT>
int launch(rip_ctx *ctx, int np_, const ZT *slin, const float *coefs, const float *smin, const float *smax, size_t npix, ZT *out,
           uint8_t *ex) {
    const dim3 grid((unsigned)((npix + 255) / 256)), block(256);
#define IL_CASE(N)                                                                                                       \
    case N:                                                                                                              \
        hipLaunchKernelGGL((invlin_kernel<ZT, N>), grid, block, 0, ctx->stream, slin, coefs, smin, smax, npix, out, ex); \
        break;
    switch (np_) {
        IL_CASE(2) IL_CASE(3) IL_CASE(4) IL_CASE(5) IL_CASE(6) IL_CASE(7) IL_CASE(8) IL_CASE(9) IL_CASE(10) IL_CASE(11) IL_CASE(12)
        IL_CASE(13) IL_CASE(14) IL_CASE(15) IL_CASE(16) IL_CASE(17)
        default:
            return rip_fail(ctx, RIP_EINVAL, "invlinearity: %d coefficient planes (2..17 supported)", np_);
    }
#undef IL_CASE
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}

}   // namespace

extern "C" int rip_stage_invlinearity(rip_ctx *ctx, const void *slin, int dtype, int ny, int nx, int nplanes, const float *coefs,
                                      const float *smin, const float *smax, void *S, uint8_t *exflag) {
    if (!slin || !coefs || !smin || !smax || !S || ny < 1 || nx < 1 || (dtype != RIP_F32 && dtype != RIP_F64))
        return rip_fail(ctx, RIP_EINVAL, "invlinearity: bad arguments");
    RIP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ny * nx, es = dtype == RIP_F64 ? 8 : 4;
    DevBuf<unsigned char> din(ctx), dout(ctx), dex(ctx);
    DevBuf<float> dc(ctx), dmin(ctx), dmax(ctx);
    int rc;
    if ((rc = din.upload(slin, npix * es)) || (rc = dc.upload(coefs, (size_t)nplanes * npix)) || (rc = dmin.upload(smin, npix)) ||
        (rc = dmax.upload(smax, npix)) || (rc = dout.alloc(npix * es)) || (exflag && (rc = dex.alloc(npix))))
        return rc;
    if (dtype == RIP_F64)
        rc = launch<double>(ctx, nplanes, (const double *)din.p, dc.p, dmin.p, dmax.p, npix, (double *)dout.p, exflag ? dex.p : nullptr);
    else
        rc = launch<float>(ctx, nplanes, (const float *)din.p, dc.p, dmin.p, dmax.p, npix, (float *)dout.p, exflag ? dex.p : nullptr);
    if (rc) return rc;
    RIP_HIP(ctx, hipMemcpyAsync(S, dout.p, npix * es, hipMemcpyDeviceToHost, ctx->stream));
    if (exflag) RIP_HIP(ctx, hipMemcpyAsync(exflag, dex.p, npix, hipMemcpyDeviceToHost, ctx->stream));
    RIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RIP_OK;
}
