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

// ------------------------------------------------------------------ fitting.jump_detect as a callable (fitting.py:89-255)
// One pass over the plan's full ramp (variant 0): slope, errors, the significance of every tested difference in the
// reference's exact operation order (the cube `smap` is an OUTPUT here, so there is no approximate fast path), JUMP_DET OR-ed
// into rdq[i] on the active region.  A ramp truncated at `truncate_ramp` is the same pass under a plan of that many groups
// with the two-point weights of fitting.py:162-167 (built by the host mirror, romanimpreprocess_amd/utils/fitting.py).
template <typename GT>
__global__ __launch_bounds__(RF_THREADS) void jumpdetect_kernel(const float *__restrict__ cube, uint8_t *__restrict__ rdq,
                                                                const void *__restrict__ gain_,
                                                                const float *__restrict__ read_noise,
                                                                float *__restrict__ slope, float *__restrict__ err_read,
                                                                float *__restrict__ err_poisson, float *__restrict__ smap,
                                                                const RipPlanHeader *__restrict__ h,
                                                                const RipVariant *__restrict__ vars,
                                                                const float *__restrict__ kvals,
                                                                const RipDiff *__restrict__ diffs, int G, int ny, int nx,
                                                                int nb) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float *D = reinterpret_cast<float *>(lds_raw) + threadIdx.x;  // [G][RF_THREADS], this thread's column
    const size_t npix = (size_t)ny * nx;
    const size_t p = (size_t)blockIdx.x * RF_THREADS + threadIdx.x;
    if (p >= npix) return;
    const int y = (int)(p / nx), x = (int)(p % nx);
    const bool active = (y >= nb) && (y < ny - nb) && (x >= nb) && (x < nx - nb);
    for (int g = 0; g < G; ++g) D[g * RF_THREADS] = cube[(size_t)g * npix + p];
    const RipVariant v = vars[0];
    const float *kv = kvals + v.k_ofs;
    const RipDiff *df = diffs + v.diff_ofs;
    const GT gain = reinterpret_cast<const GT *>(gain_)[p];
    const float rn = read_noise[p];
    const float d1 = D[RF_THREADS];
    float s = 0.0f;
    for (int t = 0; t < v.g; ++t) {
        const float diff = D[t * RF_THREADS] - d1;
        const float prod = kv[t] * diff;
        s = s + prod;
    }
    const GT gc = clip2<GT>(gain, GainConst<GT>::lo(), GainConst<GT>::hi());
    const GT dv = clip_lo<GT>((GT)s / gc, (GT)0);
    const GT pv = clip_lo<GT>((GT)v.coef * dv, (GT)0);
    float ep;
    if constexpr (sizeof(GT) == 4)
        ep = sqrtf(pv);
    else
        ep = (float)sqrt(pv);
    slope[p] = s;
    err_read[p] = rn * v.rfac;
    err_poisson[p] = ep;
    const float xc = clip2<float>(s, h->ia, h->ib);
    const float lx = log_f32(xc / h->ia);
    const double sth = h->sa + h->dsb * ((double)lx / h->loglen);
    const float s2 = rn * rn;
    for (int k = 0; k < v.ndiff; ++k) {
        const RipDiff r = df[k];
        const float num = D[r.j * RF_THREADS] - D[r.i * RF_THREADS];
        const float delta = num / r.dt - s;
        const double var = exact_variance<GT>(h, kv, v.g, r.i, r.j, r.dt, dv, s2);
        const float sme = delta / (float)sqrt(var);
        smap[(size_t)k * npix + p] = sme;
        if (active && (double)sme > sth) rdq[(size_t)r.i * npix + p] |= (uint8_t)DQ_JUMP_DET;
    }
}

int rip_launch_jumpdetect(rip_ctx *ctx, const RipPlan *plan, const float *cube, uint8_t *rdq, const void *gain, int gain_dtype,
                          const float *read_noise, float *slope, float *err_read, float *err_poisson, float *smap, int ny,
                          int nx, int nb) {
    const int G = plan->h.ngrp;
    const size_t npix = (size_t)ny * nx;
    const unsigned blocks = (unsigned)((npix + RF_THREADS - 1) / RF_THREADS);
    const size_t lds = (size_t)G * RF_THREADS * 4;
    const RipPlanHeader *h = reinterpret_cast<const RipPlanHeader *>(plan->dev);
    if (gain_dtype == RIP_F64) {
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(jumpdetect_kernel<double>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(jumpdetect_kernel<double>, dim3(blocks), dim3(RF_THREADS), lds, ctx->stream, cube, rdq, gain, read_noise,
                           slope, err_read, err_poisson, smap, h, plan->d_variants, plan->d_k, plan->d_diffs, G, ny, nx, nb);
    } else {
        if (lds > 48 * 1024)
            RIP_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(jumpdetect_kernel<float>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(jumpdetect_kernel<float>, dim3(blocks), dim3(RF_THREADS), lds, ctx->stream, cube, rdq, gain, read_noise,
                           slope, err_read, err_poisson, smap, h, plan->d_variants, plan->d_k, plan->d_diffs, G, ny, nx, nb);
    }
    RIP_HIP(ctx, hipGetLastError());
    return RIP_OK;
}
