// Per-pixel cube stage: [reference-pixel correction apply] -> [bias] -> [Legendre linearity].
// One thread per pixel walks the groups in order (the linearity flags are a running state);
// the pixel's Legendre coefficients are staged in LDS ([plane][thread]) so that the polynomial
// order is a run-time value.
//
// Replaces (reference file:line):
//   L1_to_L2/gen_cal_image.py:534-535,554-556  image = data - dark ; row/channel subtraction ; + dark
//       (the subtraction tables come from refpix.hip; the arithmetic of
//        utils/reference_subtraction.py:123 and :67-68 is applied here per pixel)
//   L1_to_L2/gen_cal_image.py:559-565          data[act] -= biascorr
//   utils/ipc_linearity.py:192-231, 276-344    _lin, multilin ; gen_cal_image.py:588 pdq |= dq_lin
// Arithmetic recipe: oracle/refpix.py, oracle/linearity.py.
//
// Roofline: HBM.  Algorithmic bytes per pixel = G*(2|4 data + 4 dark + 4 bias + 1 gdq + 4 phi)
// + 4*(nplanes + 3) + 4 (lin dq) + 4 + 4 (pdq in/out).
#include "rip_common.h"

#define LIN_THREADS 256
#define LIN_MAX_PLANES 32

template <typename T>
__device__ __forceinline__ T clip2(T x, T lo, T hi) {
    return x < lo ? lo : (x > hi ? hi : x);
}

template <typename DT>
__global__ __launch_bounds__(LIN_THREADS) void lin_kernel(LinArgs a) {
    extern __shared__ __align__(16) float lds[];
    float *CL = lds;                                       // [nplanes][LIN_THREADS]
    float *C1 = lds + (size_t)a.nplanes * LIN_THREADS;     // recurrence constants per degree
    float *C2 = C1 + LIN_MAX_PLANES;
    float *CH = C2 + LIN_MAX_PLANES;

    const int tid = threadIdx.x;
    const size_t npix = (size_t)a.ny * a.nx;
    const size_t p = (size_t)blockIdx.x * LIN_THREADS + tid;
    const bool lin = a.coefs != nullptr;
    if (lin && tid < a.nplanes && tid >= 1) {
        const int L = tid;
        C1[L] = (float)((double)(2 * L + 1) / (double)(L + 1));
        C2[L] = (float)((double)L / (double)(L + 1));
        CH[L] = (float)((double)(L * (L + 1)) / 2.0);
    }
    __syncthreads();
    if (p >= npix) return;
    const int y = (int)(p / a.nx), x = (int)(p % a.nx);
    const bool active = (y >= a.nb) && (y < a.ny - a.nb) && (x >= a.nb) && (x < a.nx - a.nb);

    float smin = 0.f, span = 1.f, sref = 0.f;
    uint32_t dq = 0;
    if (lin) {
        smin = a.smin[p];
        span = a.smax[p] - smin;
        sref = a.sref[p];
        dq = a.lin_dq[p];
        for (int L = 0; L < a.nplanes; ++L) CL[L * LIN_THREADS + tid] = a.coefs[(size_t)L * npix + p];
    }
    const uint32_t bad = DQ_NO_LIN_CORR | DQ_REFERENCE_PIXEL;
    const int nch = a.nx / RIP_CW;
    const DT *__restrict__ data = reinterpret_cast<const DT *>(a.data);

    for (int g = 0; g < a.ngrp; ++g) {
        float S = (float)data[(size_t)g * npix + p];
        if (a.rowcorr) {
            // reference_subtraction.py:123 and :67-68 in f64, cast back to f32 after each step
            const float dk = a.dark_data[(size_t)g * npix + p];
            float v = S - dk;
            v = (float)((double)v - a.rowcorr[(size_t)g * a.ny + y]);
            const double *ln = a.lines + ((size_t)g * nch + x / RIP_CW) * 2;
            const double iel = ln[0] * (double)y + ln[1];
            v = (float)((double)v - iel);
            S = v + dk;
        }
        if (a.bias && active) S = S - a.bias[(size_t)g * npix + p];
        float val = S;
        if (lin) {
            float t = S - smin;
            t = 2.0f * t;
            float z = -1.0f + t / span;
            const bool first = (g == 0) && a.do_not_flag_first;
            if (first) z = clip2<float>(z, -1.0f, 1.0f);
            const float az = fabsf(z);
            const bool ex = az > 1.0f;
            const float exc = az - 1.0f;
            const bool neg = z < 0.0f;
            float phi = CL[tid];
            float pp = 1.0f, pc = z;
            for (int L = 1; L < a.nplanes; ++L) {
                float e = 1.0f + CH[L] * exc;
                e = (neg && (L & 1)) ? -e : e;
                const float sel = ex ? e : pc;
                const float term = CL[L * LIN_THREADS + tid] * sel;
                phi = phi + term;
                const float u = C1[L] * z;
                const float pn = u * pc - C2[L] * pp;
                pp = pc;
                pc = pn;
            }
            val = ((dq & bad) == 0) ? phi : (S - sref);
            if (!first && ex) {
                bool attempt = true;
                if (a.gdq) {
                    const uint8_t q = a.gdq[(size_t)g * npix + p];
                    attempt = a.gdq_is_attempt ? (q != 0) : ((q & DQ_SATURATED) == 0);
                }
                if (attempt) dq |= DQ_NO_LIN_CORR;
            }
        }
        a.phi[(size_t)g * npix + p] = val;
    }
    if (a.pdq_out) a.pdq_out[p] = (a.pdq_in ? a.pdq_in[p] : 0u) | dq;
}

int rip_launch_lin(rip_ctx *ctx, const LinArgs &a) {
    if (a.coefs && (a.nplanes < 1 || a.nplanes > LIN_MAX_PLANES))
        return rip_fail(ctx, RIP_EINVAL, "linearity: %d Legendre planes unsupported (1..%d)", a.nplanes, LIN_MAX_PLANES);
    if (a.rowcorr && (a.nx % RIP_CW) != 0) return rip_fail(ctx, RIP_EINVAL, "refpix: nx=%d is not a multiple of 128", a.nx);
    const size_t npix = (size_t)a.ny * a.nx;
    const unsigned blocks = (unsigned)((npix + LIN_THREADS - 1) / LIN_THREADS);
    const int npl = a.coefs ? a.nplanes : 0;
    const size_t lds = ((size_t)npl * LIN_THREADS + 3 * LIN_MAX_PLANES) * sizeof(float);
    LinArgs b = a;
    b.nplanes = npl;
    if (a.data_dtype == RIP_U16)
        hipLaunchKernelGGL(lin_kernel<uint16_t>, dim3(blocks), dim3(LIN_THREADS), lds, ctx->stream, b);
    else
        hipLaunchKernelGGL(lin_kernel<float>, dim3(blocks), dim3(LIN_THREADS), lds, ctx->stream, b);
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
