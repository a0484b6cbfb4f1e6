// Up-the-ramp slope fit, jump detection, saturation-truncated refits, flag propagation and the
// post-fit algebra (dark rate, error split, flat) -- one thread per pixel, the pixel's ramp staged
// in LDS ([group][thread], conflict-free) so that the group count is a run-time value.
//
// Replaces (reference file:line):
//   utils/fitting.py:89-255   jump_detect   -> fit_variant()
//   utils/fitting.py:258-355  ramp_fit      -> rampfit_kernel() (variant loop + flag propagation)
//   L1_to_L2/gen_cal_image.py:458-475, 213-229, 607-629 -> the `finish` block
// Arithmetic recipe: oracle/rampfit.py, oracle/finish.py (bit-for-bit pinned to the reference).
//
// Roofline: HBM.  Algorithmic bytes per pixel = G*(4+1+1) + 4 (pdq) + gain + 4 (read) + 16 (outputs)
// (+ dark_rate, flat, flat_flags when finishing).
#include "rip_common.h"

#define RF_THREADS 256

#include "device_rampfit.h"

template <typename GT>
__global__ __launch_bounds__(RF_THREADS) void rampfit_kernel(RampFitArgs a, const RipPlanHeader *__restrict__ h,
                                                             const RipVariant *__restrict__ vars,
                                                             const float *__restrict__ kvals,
                                                             const RipDiff *__restrict__ diffs, double guard) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int G = a.ngrp;
    float *Dl = reinterpret_cast<float *>(lds_raw);                 // [G][RF_THREADS]
    uint8_t *Ql = lds_raw + (size_t)G * RF_THREADS * sizeof(float);  // [G][RF_THREADS] input group flags
    uint8_t *Jl = Ql + (size_t)G * RF_THREADS;                       // [G][RF_THREADS] new JUMP_DET flags

    const int tid = threadIdx.x;
    const size_t npix = (size_t)a.ny * a.nx;
    const size_t p = (size_t)blockIdx.x * RF_THREADS + tid;
    if (p >= npix) return;  // no barriers below: every thread only touches its own LDS column
    const int y = (int)(p / a.nx), x = (int)(p % a.nx);
    const bool active = (y >= a.nb) && (y < a.ny - a.nb) && (x >= a.nb) && (x < a.nx - a.nb);

    float *D = Dl + tid;
    uint8_t *Q = Ql + tid;
    uint8_t *J = Jl + tid;
    for (int g = 0; g < G; ++g) {
        D[g * RF_THREADS] = a.cube[(size_t)g * npix + p];
        Q[g * RF_THREADS] = a.gdq_in[(size_t)g * npix + p];
        J[g * RF_THREADS] = 0;
    }
    const GT gain = reinterpret_cast<const GT *>(a.gain)[p];
    const float rn = a.read_noise[p];
    float s, er, ep;
    uint32_t pdq;
    rampfit_pixel<GT, RF_THREADS>(D, Q, J, G, h, vars, kvals, diffs, gain, rn, active, guard, a.pdq_in[p],
                                  a.gdq_out ? a.gdq_out + p : nullptr, npix, s, er, ep, pdq);
    if (a.finish) finish_pixel(active, p, a.dark_rate, a.dark_dq, a.flat, a.flat_flags, s, er, ep, pdq);
    a.slope[p] = s;
    a.err_read[p] = er;
    a.err_poisson[p] = ep;
    a.pdq_out[p] = pdq;
}

int rip_launch_rampfit(rip_ctx *ctx, const RipPlan *plan, const RampFitArgs &a, int gain_dtype) {
    if (a.ngrp != plan->h.ngrp) return rip_fail(ctx, RIP_EINVAL, "ramp has %d groups, plan %d", a.ngrp, plan->h.ngrp);
    const size_t npix = (size_t)a.ny * a.nx;
    const unsigned blocks = (unsigned)((npix + RF_THREADS - 1) / RF_THREADS);
    const size_t lds = (size_t)a.ngrp * RF_THREADS * 6;
    const RipPlanHeader *h = reinterpret_cast<const RipPlanHeader *>(plan->dev);
    if (gain_dtype == RIP_F64) {
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(rampfit_kernel<double>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(rampfit_kernel<double>, dim3(blocks), dim3(RF_THREADS), lds, ctx->stream, a, h,
                           plan->d_variants, plan->d_k, plan->d_diffs, ctx->guard_band);
    } else {
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(rampfit_kernel<float>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(rampfit_kernel<float>, dim3(blocks), dim3(RF_THREADS), lds, ctx->stream, a, h,
                           plan->d_variants, plan->d_k, plan->d_diffs, ctx->guard_band);
    }
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
